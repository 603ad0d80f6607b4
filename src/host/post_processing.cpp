// src/host/post_processing.cpp -- toneMap / gammaCorrect / postProcess of PathTrace/post_processing.h: the frame goes to the
// GPU, is processed there (cpupathtrace_amd/csrc/pt_post.hip through the C ABI of include/pt_hip.h) and comes back in place.
// Like the rest of this library there is no CPU path: without a HIP device the calls throw std::runtime_error.
#include <PathTrace/post_processing.h>

#include "../../include/pt_hip.h"

#include <cstdlib>
#include <stdexcept>
#include <string>

namespace {

    void runOnDevice(Image<> &image, uint32_t steps, float gamma) {
        static_assert(sizeof(Color<float>) == 4 * sizeof(float), "an Image<> is a dense rgba f32 array");
        const char *device_env = std::getenv("PATHTRACE_DEVICE");
        const int device = device_env != nullptr ? std::atoi(device_env) : 0;
        float *rgba = image.getWidth() * image.getHeight() > 0 ? &(*image.data())[0] : nullptr;
        if(pt_post_process(device, rgba, image.getWidth(), image.getHeight(), steps, gamma) != PT_OK) {
            throw std::runtime_error(std::string("PathTrace: post-processing failed: ") + pt_last_error());
        }
    }

} // namespace

void toneMap(Image<> &image) {
    runOnDevice(image, PT_POST_TONE_MAP, 1.0F);
}

void gammaCorrect(Image<> &image, float gamma) {
    runOnDevice(image, PT_POST_GAMMA, gamma);
}

void postProcess(Image<> &image) {
    runOnDevice(image, PT_POST_TONE_MAP | PT_POST_GAMMA, 1.8F); // gammaCorrect's default argument, post_processing.h:22
}
