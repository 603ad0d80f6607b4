// src/host/post_processing.cpp -- tone mapping and gamma (PathTrace/post_processing.h); host code, after rendering.
#include <PathTrace/post_processing.h>
#include <PathTrace/util/color.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace {

    float peak(const Color<float> &c) {
        return std::max({c[0], c[1], c[2]});
    }

    // alpha-weighted average of the mean and the peak of rgb
    float loudness(const Color<float> &c) {
        return c[3] * ((c[0] + c[1] + c[2]) / 3.0F + peak(c)) / 2.0F;
    }

    float bell(float t, float sigma) {
        const float z = t / sigma;
        return (1.0F / std::sqrt(2.0F * static_cast<float>(M_PI))) * std::exp(-(z * z) / 2.0F) / sigma;
    }

} // namespace

// Histogram equalisation of the loudness: the sorted loudness values are cut into up to 1024 segments whose populations follow a
// bell curve over the output range (plus a floor), and every pixel is scaled so that its peak channel lands where its loudness
// falls inside its segment.
void toneMap(Image<> &image) {
    const int width = image.getWidth(), height = image.getHeight();
    const int pixel_count = width * height;
    if(pixel_count <= 0) {
        return;
    }
    std::vector<float> sorted;
    sorted.reserve(static_cast<size_t>(pixel_count));
    float lowest = 0.0F, highest = 1E-4F;
    for(int y = 0; y < height; y++) {
        for(int x = 0; x < width; x++) {
            const float v = loudness(image(x, y));
            sorted.push_back(v);
            lowest = std::min(lowest, v);
            highest = std::max(highest, v);
        }
    }
    std::sort(sorted.begin(), sorted.end());

    const int segments = std::min(1024, pixel_count);
    std::vector<float> weight(static_cast<size_t>(segments));
    float weight_sum = 0.0F;
    for(int i = 0; i < segments; i++) {
        const float centre = 2.0F * ((static_cast<float>(i) + 0.5F) / static_cast<float>(segments) - 0.5F);
        weight[i] = 0.1F + bell(centre, 0.3F);
        weight_sum += weight[i];
    }
    std::vector<float> ceiling;
    ceiling.reserve(static_cast<size_t>(segments));
    int consumed = 0;
    float carried = 0.0F;
    for(int i = 0; i < segments - 1; i++) {
        const float share = weight[i] * static_cast<float>(pixel_count) / weight_sum;
        const int take = static_cast<int>(std::round(share + carried));
        if(take > 0) {
            ceiling.push_back(sorted[static_cast<size_t>(std::min(consumed + take - 1, pixel_count - 1))]);
            consumed += take;
            carried = 0.0F;
        }
        else {
            ceiling.push_back(i > 0 ? ceiling[static_cast<size_t>(i) - 1] : lowest);
            carried += share;
        }
    }
    ceiling.push_back(highest);

    const float tiny = std::numeric_limits<float>::min();
    for(int y = 0; y < height; y++) {
        for(int x = 0; x < width; x++) {
            Color<float> &pixel = image(x, y);
            const float v = loudness(pixel);
            const int segment = static_cast<int>(std::lower_bound(ceiling.begin(), ceiling.end(), v) - ceiling.begin());
            const int s = std::min(segment, segments - 1);
            const float upper = ceiling[static_cast<size_t>(s)];
            const float lower = s > 0 ? ceiling[static_cast<size_t>(s) - 1] : lowest;
            const float within = (v - lower) / std::max(upper - lower, tiny);
            const float out_lower = static_cast<float>(s) / static_cast<float>(segments);
            const float out_upper = static_cast<float>(s + 1) / static_cast<float>(segments);
            const float target = out_lower + within * (out_upper - out_lower);
            const float factor = target / std::max(peak(pixel), tiny);
            pixel[0] *= factor;
            pixel[1] *= factor;
            pixel[2] *= factor;
        }
    }
}

void gammaCorrect(Image<> &image, float gamma) {
    for(int y = 0; y < image.getHeight(); y++) {
        for(int x = 0; x < image.getWidth(); x++) {
            Color<float> &pixel = image(x, y);
            const float factor = std::pow(peak(pixel), 1.0F / gamma - 1.0F);
            pixel[0] *= factor;
            pixel[1] *= factor;
            pixel[2] *= factor;
        }
    }
}

void postProcess(Image<> &image) {
    toneMap(image);
    gammaCorrect(image);
}
