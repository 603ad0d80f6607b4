// PathTrace/image/image.h -- row-major pixel container of the PathTrace API: pixel (x, y) is element y * width + x.
#ifndef PATHTRACE_IMAGE_H
#define PATHTRACE_IMAGE_H

#include <PathTrace/util/color.h>

#include <cassert>
#include <cstddef>
#include <vector>

template<typename T = Color<float>>
class Image {
  public:
    using value_type = T;

    Image() = default;
    Image(int width, int height) : width(width), height(height), pixels(static_cast<std::size_t>(width) * static_cast<std::size_t>(height)) {}

    T operator()(int x, int y) const noexcept {
        assertContainsPoint(x, y);
        return pixels[static_cast<std::size_t>(y) * width + x];
    }
    T &operator()(int x, int y) noexcept {
        assertContainsPoint(x, y);
        return pixels[static_cast<std::size_t>(y) * width + x];
    }

    std::size_t size() const noexcept { return pixels.size(); }
    const T *data() const noexcept { return pixels.data(); }
    T *data() noexcept { return pixels.data(); }
    int getWidth() const noexcept { return width; }
    int getHeight() const noexcept { return height; }

  protected:
    void assertContainsPoint([[maybe_unused]] int x, [[maybe_unused]] int y) const noexcept {
        assert(x >= 0 && x < width);
        assert(y >= 0 && y < height);
    }

  private:
    int width = 0;
    int height = 0;
    std::vector<T> pixels;
};

#endif
