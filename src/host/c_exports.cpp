// src/host/c_exports.cpp -- plain-C views of a few scene-construction helpers of libPathTrace.so, so the Python test-suite can
// compare them with the compiled reference (tests/test_oracle_vs_reference.py).  Same argument lists as the matching
// functions of oracle/ref_shim.cpp.
#include <PathTrace/image/image.h>
#include <PathTrace/image/image_io.h>
#include <PathTrace/scene/mesh.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <sstream>
#include <string>

namespace {

    uint64_t store(const std::vector<Triangle> &triangles, uint64_t capacity, float *pos, float *nrm) {
        for(uint64_t i = 0; i < triangles.size() && i < capacity; i++) {
            const Triangle &t = triangles[i];
            const vec3<float> *points[3] = {&t.a, &t.b, &t.c};
            const vec3<float> *normals[3] = {&t.normal_a, &t.normal_b, &t.normal_c};
            for(int v = 0; v < 3; v++) {
                for(int k = 0; k < 3; k++) {
                    pos[9 * i + 3 * v + k] = (*points[v])[k];
                    nrm[9 * i + 3 * v + k] = (*normals[v])[k];
                }
            }
        }
        return triangles.size();
    }

    mat4<float> matrixFrom(const float *m16) {
        mat4<float> m{};
        for(int r = 0; r < 4; r++) {
            for(int c = 0; c < 4; c++) {
                m.rows[r][c] = m16[4 * r + c];
            }
        }
        return m;
    }

} // namespace

extern "C" {

uint64_t pth_make_plane(const float *a, const float *b, uint64_t capacity, float *pos, float *nrm) {
    return store(makePlane(vec3<float>(a[0], a[1], a[2]), vec3<float>(b[0], b[1], b[2]), false), capacity, pos, nrm);
}

uint64_t pth_make_box(const float *a, const float *b, uint64_t capacity, float *pos, float *nrm) {
    return store(makeBox(vec3<float>(a[0], a[1], a[2]), vec3<float>(b[0], b[1], b[2]), false), capacity, pos, nrm);
}

uint64_t pth_load_mesh(const char *obj_text, uint64_t len, const float *mat16, int smooth, uint64_t capacity, float *pos, float *nrm) {
    std::istringstream stream(std::string(obj_text, len));
    return store(io::loadMesh(stream, matrixFrom(mat16), false, smooth != 0), capacity, pos, nrm);
}

// Writes an indexed triangle mesh as a Wavefront OBJ file ("v x y z" with 9 significant digits -- exact for float32 -- and 1-based
// "f a b c" lines): the committed stand-in for assets/xyzrgb_dragon.obj goes to disk this way for the reference's benchmark program.
int pth_write_obj(const char *path, const float *vertices, uint64_t n_vertices, const int32_t *faces, uint64_t n_faces) {
    std::FILE *f = std::fopen(path, "w");
    if(f == nullptr) {
        return 1;
    }
    std::vector<char> buffer(1 << 22);
    std::setvbuf(f, buffer.data(), _IOFBF, buffer.size());
    for(uint64_t i = 0; i < n_vertices; i++) {
        std::fprintf(f, "v %.9g %.9g %.9g\n", vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
    }
    for(uint64_t i = 0; i < n_faces; i++) {
        std::fprintf(f, "f %d %d %d\n", faces[3 * i] + 1, faces[3 * i + 1] + 1, faces[3 * i + 2] + 1);
    }
    return std::fclose(f) == 0 ? 0 : 1;
}

// io::writeRGBImage / io::readRGBImage through memory buffers (same argument lists as ref_png_write / ref_png_read of oracle/ref_png_shim.cpp)
uint64_t pth_png_write(const float *rgba, int width, int height, unsigned char *out, uint64_t capacity) {
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

int pth_png_read(const unsigned char *data, uint64_t size, float *rgba, uint64_t capacity_pixels, int *width, int *height) {
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

void pth_mat4_apply(const float *mat16, uint64_t n, const float *in, float *out) {
    const mat4<float> m = matrixFrom(mat16);
    for(uint64_t i = 0; i < n; i++) {
        const auto v = m * vec3<float>(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
        out[3 * i] = v[0];
        out[3 * i + 1] = v[1];
        out[3 * i + 2] = v[2];
    }
}

}
