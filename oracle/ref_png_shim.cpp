// oracle/ref_png_shim.cpp -- TEST INFRASTRUCTURE ONLY.  Plain-C doors into the reference's PNG codec (io::writeRGBImage / io::readRGBImage,
// src/image/image_io.cpp, compiled where it lies together with this file and linked against the libpng of this image, /opt/conda/lib), so
// that the tests can compare the repository's own PNG writer and reader with it byte for byte.
#include <PathTrace/image/image.h>
#include <PathTrace/image/image_io.h>

#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>

extern "C" {

// returns the size of the PNG stream (0 on failure); at most `capacity` bytes are stored
uint64_t ref_png_write(const float *rgba, int width, int height, unsigned char *out, uint64_t capacity) {
    try {
        Image<Color<float>> image(width, height);
        for(int y = 0; y < height; y++) {
            for(int x = 0; x < width; x++) {
                const float *p = rgba + 4 * (static_cast<size_t>(y) * static_cast<size_t>(width) + static_cast<size_t>(x));
                image(x, y) = Color<float>(p[0], p[1], p[2], p[3]);
            }
        }
        std::ostringstream stream;
        io::writeRGBImage(stream, image);
        const std::string bytes = stream.str();
        std::memcpy(out, bytes.data(), bytes.size() < capacity ? bytes.size() : capacity);
        return bytes.size();
    }
    catch(...) {
        return 0;
    }
}

// returns 0 and fills width, height and at most capacity_pixels rgba pixels; 1 when the reference refuses the stream
int ref_png_read(const unsigned char *data, uint64_t size, float *rgba, uint64_t capacity_pixels, int *width, int *height) {
    try {
        std::istringstream stream(std::string(reinterpret_cast<const char *>(data), size));
        const auto image = io::readRGBImage(stream);
        *width = image.getWidth();
        *height = image.getHeight();
        for(int y = 0; y < image.getHeight(); y++) {
            for(int x = 0; x < image.getWidth(); x++) {
                const uint64_t i = static_cast<uint64_t>(y) * static_cast<uint64_t>(image.getWidth()) + static_cast<uint64_t>(x);
                if(i < capacity_pixels) {
                    const auto c = image(x, y);
                    rgba[4 * i] = c[0];
                    rgba[4 * i + 1] = c[1];
                    rgba[4 * i + 2] = c[2];
                    rgba[4 * i + 3] = c[3];
                }
            }
        }
        return 0;
    }
    catch(...) {
        return 1;
    }
}

}
