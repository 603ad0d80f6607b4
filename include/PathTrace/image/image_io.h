// PathTrace/image/image_io.h -- PNG codec of the PathTrace API (8-bit RGBA out, any 8-bit non-interlaced PNG in).
// Not on the rendering path; implemented on zlib (src/host/image_io.cpp).  Errors are reported as std::logic_error.
#ifndef PATHTRACE_IMAGE_IO_H
#define PATHTRACE_IMAGE_IO_H

#include <PathTrace/image/image.h>
#include <PathTrace/util/color.h>

#include <filesystem>
#include <istream>
#include <ostream>
#include <string>

namespace io {

    Image<Color<float>> readRGBImage(std::basic_istream<char> &stream) noexcept(false);
    Image<Color<float>> readRGBImage(const std::string &path) noexcept(false);
    Image<Color<float>> readRGBImage(const std::filesystem::path &path) noexcept(false);

    void writeRGBImage(std::basic_ostream<char> &stream, const Image<Color<float>> &image) noexcept(false);
    void writeRGBImage(const std::string &path, const Image<Color<float>> &image) noexcept(false);
    void writeRGBImage(const std::filesystem::path &path, const Image<Color<float>> &image) noexcept(false);

} // namespace io

#endif
