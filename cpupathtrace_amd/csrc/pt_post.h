// pt_post.h -- device post-processing (pt_post.hip): toneMap / gammaCorrect of the reference on an rgba f32 frame in HBM.
#ifndef PT_POST_H
#define PT_POST_H

#include <hip/hip_runtime.h>

#include <cstdint>

#define PT_POST_STEP_TONE_MAP 1u
#define PT_POST_STEP_GAMMA 2u

// In place on `image` (width * height float4, row-major); tone mapping first when both steps are asked for (postProcess).
// Synchronises `stream` before returning.
hipError_t pt_post_run(hipStream_t stream, float4 *image, int32_t width, int32_t height, uint32_t steps, float gamma);

#endif
