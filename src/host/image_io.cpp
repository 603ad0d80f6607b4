// src/host/image_io.cpp -- PNG reading/writing for PathTrace/image/image_io.h, directly on zlib (no libpng in this image).
// Writes 8-bit RGBA, filter 0, one IDAT (row bands deflated in parallel); reads any non-interlaced 8-bit grey / grey+alpha / RGB / RGBA PNG.
#include <PathTrace/image/image_io.h>

#include <thread>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <limits>
#include <stdexcept>
#include <vector>

namespace {

    const unsigned char kSignature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

    void put32(std::vector<unsigned char> &out, uint32_t v) {
        for(int shift = 24; shift >= 0; shift -= 8) {
            out.push_back(static_cast<unsigned char>(v >> shift));
        }
    }
    uint32_t get32(const unsigned char *p) {
        return (static_cast<uint32_t>(p[0]) << 24) | (static_cast<uint32_t>(p[1]) << 16) | (static_cast<uint32_t>(p[2]) << 8) | p[3];
    }

    void chunk(std::vector<unsigned char> &out, const char type[4], const std::vector<unsigned char> &body) {
        put32(out, static_cast<uint32_t>(body.size()));
        const size_t start = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), body.begin(), body.end());
        put32(out, static_cast<uint32_t>(crc32(0L, out.data() + start, static_cast<uInt>(out.size() - start))));
    }

    // The reference's conversion (src/image/image_io.cpp:139-142): min(max(int(round(255.0 * v)), 0), 255) with the product in double.
    // The float -> int conversion of its x86-64 build (cvttsd2si) yields INT_MIN for NaN and for values outside the int range, which the
    // clamp then turns into 0 -- so NaN, +-inf and anything beyond about +-8.4e6 come out as 0 there, and do here.
    unsigned char quantise(float v) {
        const double rounded = std::round(255.0 * static_cast<double>(v));
        const int as_int = (rounded >= -2147483648.0 && rounded < 2147483648.0) ? static_cast<int>(rounded) : std::numeric_limits<int>::min();
        return static_cast<unsigned char>(std::min(std::max(as_int, 0), 255));
    }

    int paeth(int a, int b, int c) {
        const int p = a + b - c;
        const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
        return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
    }

} // namespace

namespace io {

    void writeRGBImage(std::basic_ostream<char> &stream, const Image<Color<float>> &image) noexcept(false) {
        const int width = image.getWidth(), height = image.getHeight();
        if(width <= 0 || height <= 0) {
            throw std::logic_error("writeRGBImage: empty image");
        }
        // The rows are cut into bands; every band is quantised and deflated on its own core (raw deflate, ended by a sync flush so
        // that the pieces can simply be concatenated; the last one ends the stream) -- SURVEY.md 8(f) rank 4.  The result is one
        // ordinary zlib stream: header, the bands, the Adler-32 of all the filtered bytes.
        const size_t row_bytes = static_cast<size_t>(width) * 4 + 1;
        const char *threads_env = std::getenv("PATHTRACE_PNG_THREADS");
        const unsigned cores = threads_env != nullptr ? static_cast<unsigned>(std::max(1, std::atoi(threads_env))) : std::max(1U, std::thread::hardware_concurrency());
        const size_t bands = std::max<size_t>(1, std::min<size_t>({static_cast<size_t>(std::min(cores, 64U)), static_cast<size_t>(height) / 16 + 1,
                                                                   row_bytes * static_cast<size_t>(height) / 65536 + 1}));
        struct Band {
            std::vector<unsigned char> packed;
            uLong adler = 1; // adler32 of the band's filtered bytes
            size_t raw_bytes = 0;
            bool ok = false;
        };
        std::vector<Band> band(bands);
        auto encode = [&](size_t b) {
            const int first = static_cast<int>(static_cast<size_t>(height) * b / bands), last = static_cast<int>(static_cast<size_t>(height) * (b + 1) / bands);
            std::vector<unsigned char> raw;
            raw.reserve(row_bytes * static_cast<size_t>(last - first));
            for(int y = first; y < last; y++) {
                raw.push_back(0); // filter type: none
                for(int x = 0; x < width; x++) {
                    const Color<float> c = image(x, y);
                    for(int k = 0; k < 4; k++) {
                        raw.push_back(quantise(c[k]));
                    }
                }
            }
            Band &out = band[b];
            out.raw_bytes = raw.size();
            out.adler = adler32(1L, raw.data(), static_cast<uInt>(raw.size()));
            z_stream z{};
            if(deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                return;
            }
            out.packed.resize(deflateBound(&z, static_cast<uLong>(raw.size())) + 16);
            z.next_in = raw.data();
            z.avail_in = static_cast<uInt>(raw.size());
            z.next_out = out.packed.data();
            z.avail_out = static_cast<uInt>(out.packed.size());
            const bool final_band = b + 1 == bands;
            const int rc = deflate(&z, final_band ? Z_FINISH : Z_SYNC_FLUSH);
            out.ok = final_band ? rc == Z_STREAM_END : (rc == Z_OK && z.avail_in == 0);
            out.packed.resize(out.packed.size() - z.avail_out);
            deflateEnd(&z);
        };
        {
            std::vector<std::thread> pool;
            for(size_t b = 1; b < bands; b++) {
                pool.emplace_back(encode, b);
            }
            encode(0);
            for(auto &t : pool) {
                t.join();
            }
        }
        std::vector<unsigned char> packed{0x78, 0x9C}; // zlib header: deflate, 32 KiB window, default level
        uLong adler = 1;
        for(size_t b = 0; b < bands; b++) {
            if(!band[b].ok) {
                throw std::logic_error("writeRGBImage: deflate failed");
            }
            packed.insert(packed.end(), band[b].packed.begin(), band[b].packed.end());
            adler = b == 0 ? band[b].adler : adler32_combine(adler, band[b].adler, static_cast<z_off_t>(band[b].raw_bytes));
        }
        put32(packed, static_cast<uint32_t>(adler));

        std::vector<unsigned char> file(kSignature, kSignature + 8);
        std::vector<unsigned char> header;
        put32(header, static_cast<uint32_t>(width));
        put32(header, static_cast<uint32_t>(height));
        header.insert(header.end(), {8, 6, 0, 0, 0}); // 8 bits, RGBA, deflate, adaptive filtering, no interlace
        chunk(file, "IHDR", header);
        chunk(file, "IDAT", packed);
        chunk(file, "IEND", {});
        stream.write(reinterpret_cast<const char *>(file.data()), static_cast<std::streamsize>(file.size()));
        if(!stream) {
            throw std::logic_error("writeRGBImage: write error");
        }
    }

    void writeRGBImage(const std::string &path, const Image<Color<float>> &image) noexcept(false) {
        writeRGBImage(std::filesystem::path(path), image);
    }

    void writeRGBImage(const std::filesystem::path &path, const Image<Color<float>> &image) noexcept(false) {
        std::ofstream stream(path, std::ios_base::out | std::ios_base::binary);
        if(!stream) {
            throw std::logic_error("writeRGBImage: cannot open " + path.string());
        }
        writeRGBImage(stream, image);
    }

    Image<Color<float>> readRGBImage(std::basic_istream<char> &stream) noexcept(false) {
        const std::vector<unsigned char> file((std::istreambuf_iterator<char>(stream)), std::istreambuf_iterator<char>());
        if(file.size() < 8 || !std::equal(kSignature, kSignature + 8, file.begin())) {
            throw std::logic_error("readRGBImage: not a PNG stream");
        }
        uint32_t width = 0, height = 0;
        int channels = 0;
        std::vector<unsigned char> packed;
        size_t pos = 8;
        bool ended = false;
        while(!ended && pos + 12 <= file.size()) {
            const uint32_t length = get32(&file[pos]);
            if(length > file.size() - pos - 12) {
                throw std::logic_error("readRGBImage: truncated chunk");
            }
            const std::string type(reinterpret_cast<const char *>(&file[pos + 4]), 4);
            const unsigned char *body = &file[pos + 8];
            if(get32(body + length) != static_cast<uint32_t>(crc32(0L, &file[pos + 4], length + 4))) {
                throw std::logic_error("readRGBImage: chunk checksum mismatch");
            }
            if(type == "IHDR") {
                if(length != 13) {
                    throw std::logic_error("readRGBImage: bad header");
                }
                width = get32(body);
                height = get32(body + 4);
                const int depth = body[8], colour = body[9], interlace = body[12];
                channels = colour == 0 ? 1 : colour == 4 ? 2 : colour == 2 ? 3 : colour == 6 ? 4 : 0;
                if(depth != 8 || channels == 0 || interlace != 0 || width == 0 || height == 0 || width > 65535 || height > 65535) {
                    throw std::logic_error("readRGBImage: unsupported PNG variant");
                }
            }
            else if(type == "IDAT") {
                packed.insert(packed.end(), body, body + length);
            }
            else if(type == "IEND") {
                ended = true;
            }
            pos += 12 + static_cast<size_t>(length);
        }
        if(channels == 0 || packed.empty()) {
            throw std::logic_error("readRGBImage: missing image data");
        }
        const size_t stride = static_cast<size_t>(width) * static_cast<size_t>(channels);
        // deflate cannot expand by more than 1032 : 1, so a header that promises more pixels than the data can hold is refused before
        // anything of that size is allocated (a 65535 x 65535 header on a 40-byte stream asked for 17 GB)
        if((stride + 1) * height > packed.size() * 1032 + 4096) {
            throw std::logic_error("readRGBImage: image data shorter than the header promises");
        }
        std::vector<unsigned char> raw((stride + 1) * height);
        uLongf raw_size = static_cast<uLongf>(raw.size());
        if(uncompress(raw.data(), &raw_size, packed.data(), static_cast<uLong>(packed.size())) != Z_OK || raw_size != raw.size()) {
            throw std::logic_error("readRGBImage: inflate failed");
        }

        Image<Color<float>> image(static_cast<int>(width), static_cast<int>(height));
        std::vector<unsigned char> previous(stride, 0), current(stride);
        for(uint32_t y = 0; y < height; y++) {
            const unsigned char *row = &raw[(stride + 1) * y];
            const int filter = row[0];
            if(filter > 4) {
                throw std::logic_error("readRGBImage: unknown filter");
            }
            for(size_t i = 0; i < stride; i++) {
                const int left = i >= static_cast<size_t>(channels) ? current[i - channels] : 0;
                const int up = previous[i];
                const int up_left = i >= static_cast<size_t>(channels) ? previous[i - channels] : 0;
                const int predictor = filter == 0 ? 0 : filter == 1 ? left : filter == 2 ? up : filter == 3 ? (left + up) / 2 : paeth(left, up, up_left);
                current[i] = static_cast<unsigned char>(row[1 + i] + predictor);
            }
            for(uint32_t x = 0; x < width; x++) {
                const unsigned char *px = &current[static_cast<size_t>(x) * channels];
                const float grey = px[0] / 255.0F;
                Color<float> c;
                if(channels <= 2) {
                    c = Color<float>(grey, grey, grey, channels == 2 ? px[1] / 255.0F : 1.0F);
                }
                else {
                    c = Color<float>(px[0] / 255.0F, px[1] / 255.0F, px[2] / 255.0F, channels == 4 ? px[3] / 255.0F : 1.0F);
                }
                image(static_cast<int>(x), static_cast<int>(y)) = c;
            }
            previous.swap(current);
        }
        return image;
    }

    Image<Color<float>> readRGBImage(const std::string &path) noexcept(false) {
        return readRGBImage(std::filesystem::path(path));
    }

    Image<Color<float>> readRGBImage(const std::filesystem::path &path) noexcept(false) {
        std::ifstream stream(path, std::ios_base::in | std::ios_base::binary);
        if(!stream) {
            throw std::logic_error("readRGBImage: cannot open " + path.string());
        }
        return readRGBImage(stream);
    }

} // namespace io
