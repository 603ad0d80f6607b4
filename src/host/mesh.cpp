// src/host/mesh.cpp -- OBJ loading and the plane/box generators of PathTrace/scene/mesh.h (scene construction, host only).
// The loader maps the file, reads pieces of it on all cores and validates faces / smooths normals in parallel;
// the result is what a sequential character-by-character reader produces (tests/test_oracle_vs_reference.py compares it with the
// compiled reference, also on malformed text).
#include <PathTrace/scene/mesh.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <limits>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

    // A cursor over the file text in memory.  Numbers are the longest runs of [0-9+-.eE] (floats) / [0-9+-eE] (integers);
    // whatever character ends a run is consumed with it, as a character-by-character reader would do.  The value of a run is
    // what std::stof / std::stoi make of it (a prefix may be all they use; no conversion or out of range = "unreadable").
    class ObjText {
      public:
        ObjText(const char *text, size_t size, size_t pos) : text(text), size(size), pos(pos) {}

        size_t position() const { return pos; }
        bool atEnd() const { return pos >= size; }
        char take() { return atEnd() ? static_cast<char>(-1) : text[pos++]; }
        bool takeIf(char c) {
            if(!atEnd() && text[pos] == c) {
                pos++;
                return true;
            }
            return false;
        }
        void skipBlanks() {
            while(takeIf(' ')) {
            }
        }
        void skipLine() {
            while(!atEnd()) {
                const char c = take();
                if(c == '\r' || c == '\n') {
                    return;
                }
            }
        }
        int integer() {
            size_t begin = 0, length = 0;
            run(false, begin, length);
            // plain decimal of at most nine digits: the value is immediate
            size_t i = 0;
            bool negative = false;
            if(length > 0 && (text[begin] == '-' || text[begin] == '+')) {
                negative = text[begin] == '-';
                i = 1;
            }
            if(length - i >= 1 && length - i <= 9) {
                int value = 0;
                size_t j = i;
                for(; j < length && text[begin + j] >= '0' && text[begin + j] <= '9'; j++) {
                    value = value * 10 + (text[begin + j] - '0');
                }
                if(j == length) {
                    return negative ? -value : value;
                }
            }
            try {
                return std::stoi(std::string(text + begin, length));
            }
            catch(const std::exception &) {
                return -1;
            }
        }
        float real() {
            size_t begin = 0, length = 0;
            run(true, begin, length);
            float fast;
            if(plainDecimal(text + begin, length, fast)) {
                return fast;
            }
            try {
                return std::stof(std::string(text + begin, length));
            }
            catch(const std::exception &) {
                return std::numeric_limits<float>::quiet_NaN();
            }
        }

      private:
        void run(bool allow_point, size_t &begin, size_t &length) {
            skipBlanks();
            begin = pos;
            length = 0;
            while(!atEnd()) {
                const char c = take();
                const bool part = (c >= '0' && c <= '9') || c == '-' || c == '+' || c == 'e' || c == 'E' || (allow_point && c == '.');
                if(!part) {
                    break;
                }
                length++;
            }
        }

        // [sign] digits [. digits] with at most 15 digits in all: mantissa and power of ten are exact doubles, their quotient is the
        // correctly rounded double; it is rounded once more to float unless it lies within one double ulp of the midpoint of two
        // floats, where only strtof's exact arithmetic can decide (left to std::stof)
        static bool plainDecimal(const char *p, size_t n, float &out) {
            static const double power_of_ten[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
            size_t i = 0;
            bool negative = false;
            if(n > 0 && (p[0] == '-' || p[0] == '+')) {
                negative = p[0] == '-';
                i = 1;
            }
            uint64_t mantissa = 0;
            int digits = 0, fraction_digits = 0;
            bool seen_point = false;
            for(; i < n; i++) {
                const char c = p[i];
                if(c >= '0' && c <= '9') {
                    mantissa = mantissa * 10 + static_cast<uint64_t>(c - '0');
                    digits++;
                    fraction_digits += seen_point ? 1 : 0;
                    if(digits > 15) {
                        return false;
                    }
                }
                else if(c == '.' && !seen_point) {
                    seen_point = true;
                }
                else {
                    return false;
                }
            }
            if(digits == 0) {
                return false;
            }
            double value = static_cast<double>(mantissa) / power_of_ten[fraction_digits];
            uint64_t bits;
            std::memcpy(&bits, &value, sizeof(bits));
            const uint64_t low = bits & 0x1fffffffULL; // the 29 bits a float does not keep
            if(low >= 0x0fffffffULL && low <= 0x10000001ULL) {
                return false;
            }
            const float narrowed = static_cast<float>(value);
            if(!(narrowed >= std::numeric_limits<float>::min() || narrowed == 0.0F)) {
                return false; // subnormal results: leave the range handling to the library
            }
            out = negative ? -narrowed : narrowed;
            return true;
        }

        const char *text;
        size_t size;
        size_t pos;
    };

    struct FaceLine {
        int index[3];             // zero-based
        uint32_t vertices_before; // vertices read so far within the same piece of the text
    };

    struct Piece {
        std::vector<vec3<float>> vertices;
        std::vector<FaceLine> faces;
        size_t end = 0; // where the reader stood after the last line it started
    };

    // Reads the lines that START in [begin, end).  The reader may run past `end` (a number that continues on the next line): the
    // position it stops at tells the caller whether the next piece really started at a line start.
    void readPiece(const char *text, size_t size, size_t begin, size_t end, const mat4<float> &transformation, Piece &result) {
        Piece piece; // filled locally: neighbouring entries of the callers' array share cache lines
        // one slot per line for either list: growing the lists while reading makes the threads queue up in the allocator
        size_t lines = 1;
        for(const char *p = text + begin; (p = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(text + end - p)))) != nullptr; p++) {
            lines++;
        }
        piece.vertices.reserve(lines);
        piece.faces.reserve(lines);
        ObjText reader(text, size, begin);
        while(reader.position() < end && !reader.atEnd()) {
            reader.skipBlanks();
            const char tag = reader.take();
            if(tag == '\r' || tag == '\n') {
                continue;
            }
            if(tag == 'v' && reader.takeIf(' ')) {
                const float x = reader.real();
                const float y = reader.real();
                const float z = reader.real();
                piece.vertices.push_back(vec3<float>(transformation * vec3<float>{x, y, z}));
            }
            else if(tag == 'f' && reader.takeIf(' ')) {
                FaceLine face;
                for(int &i : face.index) {
                    i = reader.integer() - 1; // OBJ indices start at 1
                    while(reader.takeIf('/')) { // texture / normal references are read and ignored
                        reader.integer();
                    }
                }
                face.vertices_before = static_cast<uint32_t>(piece.vertices.size());
                piece.faces.push_back(face);
            }
            else {
                reader.skipLine();
            }
        }
        piece.end = reader.position();
        result = std::move(piece);
    }

    template<typename Body>
    void inParallel(size_t count, unsigned threads, Body body) { // body(first, last) over [0, count) in contiguous slices
        threads = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(threads, count / 4096 + 1)));
        if(threads == 1) {
            body(size_t{0}, count);
            return;
        }
        std::vector<std::thread> pool;
        for(unsigned t = 0; t < threads; t++) {
            pool.emplace_back(body, count * t / threads, count * (t + 1) / threads);
        }
        for(auto &th : pool) {
            th.join();
        }
    }

    std::vector<Triangle> meshFromText(const char *text, size_t size, const mat4<float> &transformation, bool cull_backface, bool smooth) {
        const char *threads_env = std::getenv("PATHTRACE_LOADER_THREADS");
        unsigned threads = threads_env != nullptr ? static_cast<unsigned>(std::max(1, std::atoi(threads_env))) : std::max(1U, std::thread::hardware_concurrency());
        threads = std::min(threads, 64U);
        const bool debug = std::getenv("PATHTRACE_LOADER_DEBUG") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if(debug) {
                const auto now = std::chrono::steady_clock::now();
                std::fprintf(stderr, "[loadMesh] %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
                t_last = now;
            }
        };

        // ---- read: the text is cut at line starts and the pieces are read concurrently.  A piece is only right if the piece
        // before it stopped exactly where it starts (a line whose numbers spill over a line end breaks that); otherwise the
        // whole text is read again as one piece, which is what a sequential reader does.
        std::vector<Piece> pieces;
        {
            // PATHTRACE_LOADER_PIECE_BYTES: smallest piece (default 256 KiB; the tests use a few bytes to exercise the cutting)
            const char *piece_env = std::getenv("PATHTRACE_LOADER_PIECE_BYTES");
            const size_t piece_bytes = piece_env != nullptr ? static_cast<size_t>(std::max(1, std::atoi(piece_env))) : (size_t{1} << 18);
            const size_t wanted = size < 4 * piece_bytes ? 1 : std::min<size_t>(threads * 4, size / piece_bytes);
            std::vector<size_t> cuts{0};
            for(size_t k = 1; k < wanted; k++) {
                size_t at = size * k / wanted;
                while(at < size && text[at - 1] != '\n') {
                    at++;
                }
                if(at > cuts.back() && at < size) {
                    cuts.push_back(at);
                }
            }
            cuts.push_back(size);
            pieces.resize(cuts.size() - 1);
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for(size_t k = next.fetch_add(1); k < pieces.size(); k = next.fetch_add(1)) {
                    readPiece(text, size, cuts[k], cuts[k + 1], transformation, pieces[k]);
                }
            };
            std::vector<std::thread> pool;
            for(unsigned t = 1; t < std::min<size_t>(threads, pieces.size()); t++) {
                pool.emplace_back(worker);
            }
            worker();
            for(auto &th : pool) {
                th.join();
            }
            bool consistent = true;
            for(size_t k = 0; k + 1 < pieces.size(); k++) {
                consistent = consistent && pieces[k].end == cuts[k + 1];
            }
            if(debug) {
                std::fprintf(stderr, "[loadMesh] %zu pieces on %u threads, %s\n", pieces.size(), threads, consistent ? "consistent" : "NOT consistent: reading again as one piece");
            }
            if(!consistent) {
                pieces.assign(1, Piece{});
                readPiece(text, size, 0, size, transformation, pieces[0]);
            }
        }

        lap("read");
        // ---- vertices of the whole file, and for every face line the number of vertices that precede it
        std::vector<size_t> vertex_base(pieces.size() + 1, 0), face_base(pieces.size() + 1, 0);
        for(size_t k = 0; k < pieces.size(); k++) {
            vertex_base[k + 1] = vertex_base[k] + pieces[k].vertices.size();
            face_base[k + 1] = face_base[k] + pieces[k].faces.size();
        }
        std::vector<vec3<float>> vertices(vertex_base.back());
        inParallel(pieces.size(), threads, [&](size_t first, size_t last) {
            for(size_t k = first; k < last; k++) {
                std::copy(pieces[k].vertices.begin(), pieces[k].vertices.end(), vertices.begin() + static_cast<std::ptrdiff_t>(vertex_base[k]));
            }
        });

        // ---- faces: references to vertices that exist so far, three distinct points (NaN coordinates fail the test), not collinear
        const size_t line_count = face_base.back();
        std::vector<uint8_t> keep(line_count, 0);
        std::vector<const FaceLine *> line(line_count);
        for(size_t k = 0; k < pieces.size(); k++) {
            for(size_t i = 0; i < pieces[k].faces.size(); i++) {
                line[face_base[k] + i] = &pieces[k].faces[i];
            }
        }
        std::vector<size_t> piece_of_line(line_count);
        for(size_t k = 0; k < pieces.size(); k++) {
            std::fill(piece_of_line.begin() + static_cast<std::ptrdiff_t>(face_base[k]), piece_of_line.begin() + static_cast<std::ptrdiff_t>(face_base[k + 1]), k);
        }
        inParallel(line_count, threads, [&](size_t first, size_t last) {
            for(size_t i = first; i < last; i++) {
                const FaceLine &f = *line[i];
                const long count = static_cast<long>(vertex_base[piece_of_line[i]] + f.vertices_before);
                if(f.index[0] < 0 || f.index[0] >= count || f.index[1] < 0 || f.index[1] >= count || f.index[2] < 0 || f.index[2] >= count) {
                    continue;
                }
                const auto &pa = vertices[static_cast<size_t>(f.index[0])];
                const auto &pb = vertices[static_cast<size_t>(f.index[1])];
                const auto &pc = vertices[static_cast<size_t>(f.index[2])];
                if(!((pb - pa).getLengthSquared() > 0.0F && (pc - pa).getLengthSquared() > 0.0F && (pc - pb).getLengthSquared() > 0.0F)) {
                    continue;
                }
                if(cross(pb - pa, pc - pa).getLengthSquared() <= 0.0F) {
                    continue;
                }
                keep[i] = 1;
            }
        });
        lap("validate");
        std::vector<size_t> kept; // line numbers of the faces that stay, in file order
        kept.reserve(line_count);
        for(size_t i = 0; i < line_count; i++) {
            if(keep[i] != 0) {
                kept.push_back(i);
            }
        }
        std::vector<Triangle> faces;
        faces.reserve(kept.size());
        {
            // the storage is touched on all cores first: one thread constructing 7 M objects would otherwise spend its time taking
            // first-touch page faults on close to a gigabyte
            char *storage = reinterpret_cast<char *>(faces.data());
            const size_t bytes = kept.size() * sizeof(Triangle);
            inParallel((bytes + 4095) / 4096, threads, [&](size_t first, size_t last) {
                for(size_t page = first; page < last; page++) {
                    storage[page * 4096] = 0;
                }
            });
        }
        faces.resize(kept.size(), Triangle(vec3<float>{0.0F, 0.0F, 0.0F}, vec3<float>{1.0F, 0.0F, 0.0F}, vec3<float>{0.0F, 1.0F, 0.0F}, cull_backface));
        inParallel(kept.size(), threads, [&](size_t first, size_t last) {
            for(size_t j = first; j < last; j++) {
                const FaceLine &f = *line[kept[j]];
                // (only the geometry is written: assigning whole Triangles from all threads would make them fight over the reference
                // count of the one material handler every face shares)
                Triangle &t = faces[j];
                t.a = vertices[static_cast<size_t>(f.index[0])];
                t.b = vertices[static_cast<size_t>(f.index[1])];
                t.c = vertices[static_cast<size_t>(f.index[2])];
                t.normal_a = t.normal_b = t.normal_c = cross(t.b - t.a, t.c - t.a).normalize(); // as Triangle::Triangle does
            }
        });

        lap("triangles");
        if(smooth) {
            // every vertex gets the normalised sum of the unit normals of the faces around it, summed in file order of the faces
            std::vector<uint32_t> first_corner(vertices.size() + 1, 0);
            for(size_t j = 0; j < kept.size(); j++) {
                for(int index : line[kept[j]]->index) {
                    first_corner[static_cast<size_t>(index) + 1]++;
                }
            }
            for(size_t v = 0; v < vertices.size(); v++) {
                first_corner[v + 1] += first_corner[v];
            }
            std::vector<uint32_t> corner(first_corner.back()); // face * 4 + slot, grouped by vertex
            {
                std::vector<uint32_t> fill(first_corner.begin(), first_corner.end() - 1);
                for(size_t j = 0; j < kept.size(); j++) {
                    for(int slot = 0; slot < 3; slot++) {
                        corner[fill[static_cast<size_t>(line[kept[j]]->index[slot])]++] = static_cast<uint32_t>(j) * 4U + static_cast<uint32_t>(slot);
                    }
                }
            }
            std::vector<vec3<float>> unit_face_normal(faces.size());
            inParallel(faces.size(), threads, [&](size_t first, size_t last) {
                for(size_t j = first; j < last; j++) {
                    unit_face_normal[j] = vec3<float>(cross(faces[j].b - faces[j].a, faces[j].c - faces[j].a).normalize());
                }
            });
            inParallel(vertices.size(), threads, [&](size_t first, size_t last) {
                for(size_t v = first; v < last; v++) {
                    vec3<float> sum{};
                    for(uint32_t k = first_corner[v]; k < first_corner[v + 1]; k++) {
                        sum = sum + unit_face_normal[corner[k] >> 2];
                    }
                    if(sum.getLengthSquared() <= 0.0F) {
                        continue;
                    }
                    const vec3<float> shared = sum.normalize();
                    for(uint32_t k = first_corner[v]; k < first_corner[v + 1]; k++) {
                        Triangle &f = faces[corner[k] >> 2];
                        const uint32_t slot = corner[k] & 3U;
                        (slot == 0 ? f.normal_a : slot == 1 ? f.normal_b : f.normal_c) = shared;
                    }
                }
            });
        }
        lap("smooth");
        return faces;
    }

} // namespace

namespace io {

    std::vector<Triangle> loadMesh(std::basic_istream<char> &stream, mat4<float> transformation, bool cull_backface, bool smooth) {
        const std::string text{std::istreambuf_iterator<char>(stream), std::istreambuf_iterator<char>()};
        return meshFromText(text.data(), text.size(), transformation, cull_backface, smooth);
    }

    std::vector<Triangle> loadMesh(const std::filesystem::path &path, mat4<float> transformation, bool cull_backface, bool smooth) {
        // the file is mapped, not copied through a stream buffer
        const int fd = ::open(path.c_str(), O_RDONLY);
        if(fd < 0) {
            return {};
        }
        struct stat info {};
        if(::fstat(fd, &info) != 0 || info.st_size <= 0) {
            ::close(fd);
            return {};
        }
        const size_t size = static_cast<size_t>(info.st_size);
        void *mapped = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        ::close(fd);
        if(mapped == MAP_FAILED) {
            std::ifstream stream(path, std::ios_base::in | std::ios_base::binary);
            return loadMesh(stream, transformation, cull_backface, smooth);
        }
        auto faces = meshFromText(static_cast<const char *>(mapped), size, transformation, cull_backface, smooth);
        ::munmap(mapped, size);
        return faces;
    }

} // namespace io

std::vector<Triangle> makePlane(vec3<float> a, vec3<float> b, bool cull_backface) {
    const float tolerance = 1E-4F;
    int flat_axis = -1; // the LAST axis in which the corners coincide
    int flat_count = 0;
    for(int axis = 0; axis < 3; axis++) {
        if(std::abs(a[axis] - b[axis]) < tolerance) {
            flat_axis = axis;
            flat_count++;
        }
    }
    if(flat_count != 1) {
        return {};
    }
    const int swap_axis = flat_axis == 0 ? 1 : 0;
    vec3<float> corner_ab = a;
    vec3<float> corner_ba = b;
    corner_ab[swap_axis] = b[swap_axis];
    corner_ba[swap_axis] = a[swap_axis];
    std::vector<Triangle> triangles;
    triangles.reserve(2);
    triangles.emplace_back(a, corner_ab, b, cull_backface);
    triangles.emplace_back(b, corner_ba, a, cull_backface);
    return triangles;
}

std::vector<Triangle> makeBox(vec3<float> a, vec3<float> b, bool cull_backface) {
    const float tolerance = 1E-4F;
    for(int axis = 0; axis < 3; axis++) {
        if(std::abs(a[axis] - b[axis]) < tolerance) {
            return {};
        }
    }
    std::vector<Triangle> triangles;
    triangles.reserve(12);
    for(int axis = 0; axis < 3; axis++) {
        for(const float level : {a[axis], b[axis]}) { // the two faces perpendicular to `axis`
            vec3<float> lo = a;
            vec3<float> hi = b;
            lo[axis] = level;
            hi[axis] = level;
            const auto face = makePlane(lo, hi, cull_backface);
            triangles.insert(triangles.end(), face.begin(), face.end());
        }
    }
    return triangles;
}
