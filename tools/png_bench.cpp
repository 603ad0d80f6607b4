#include <PathTrace/image/image.h>
#include <PathTrace/image/image_io.h>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <sstream>
int main() {
    Image<> img(1024, 1024);
    unsigned s = 1;
    for(int y = 0; y < 1024; y++) for(int x = 0; x < 1024; x++) { s = s * 1664525u + 1013904223u; float n = (s >> 8) / 16777216.0f; img(x, y) = Color<float>(0.5f + 0.4f * std::sin(x * 0.01f) + 0.05f * n, 0.5f + 0.4f * std::cos(y * 0.013f) + 0.05f * n, 0.3f + 0.05f * n, 1.0f); }
    for(int rep = 0; rep < 3; rep++) {
        std::ostringstream out;
        auto t0 = std::chrono::steady_clock::now();
        io::writeRGBImage(out, img);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("writeRGBImage to memory: %.1f ms (%zu bytes)\n", ms, out.str().size());
    }
}
