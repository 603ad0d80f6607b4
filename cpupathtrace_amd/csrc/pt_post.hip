// pt_post.hip -- toneMap / gammaCorrect / postProcess of the reference (src/post_processing.cpp:32-187) on the device
// (SURVEY.md 8(f) rank 3), so a rendered frame can stay in HBM until it is encoded.
//
// toneMap is a histogram equalisation of a per-pixel brightness heuristic: (1) the heuristic and its range, (2) ALL heuristic values
// sorted (the reference buckets, sorts the buckets and concatenates them, :49-88 -- the result is the fully sorted list), (3) up to
// 1024 segment ceilings picked from the sorted list at positions that depend on the pixel COUNT only (:90-127), (4) every pixel's
// rgb scaled so that its peak channel lands where its heuristic falls inside its segment (:129-166).  On the device that is one
// reduction, one radix sort, one gather and one per-pixel kernel with the ceilings in LDS; the positions of step 3 are computed by the
// host exactly as the reference computes them (expf, sqrtf and roundf of the host's libm -- the same calls on the same values).
// gammaCorrect multiplies rgb by powf(peak, 1 / gamma - 1) (:171-182): ptm::powf_glibc_full, bit-exact with glibc 2.35.
// Every floating-point expression keeps the reference's operand order; the build has no contraction and no fast-math.
#include "pt_post.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "pt_libm.h"

namespace {

__device__ __forceinline__ float max_std(float a, float b) { // std::max(a, b)
    return (a < b) ? b : a;
}
// std::max({r, g, b}): the first largest element
__device__ __forceinline__ float peak_of(float4 c) {
    float m = c.x;
    m = (m < c.y) ? c.y : m;
    m = (m < c.z) ? c.z : m;
    return m;
}
// getBrightnessHeuristic, post_processing.cpp:27-30
__device__ __forceinline__ float heuristic_of(float4 c) {
    return c.w * ((c.x + c.y + c.z) / 3.0f + peak_of(c)) / 2.0f;
}

__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// range[0] = smallest key, range[1] = largest key of the heuristic values
__global__ __launch_bounds__(256) void k_heuristic(const float4 *__restrict__ image, uint32_t n, float *__restrict__ values, uint32_t *__restrict__ range) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t lo = 0xffffffffu, hi = 0u;
    if(i < n) {
        const float v = heuristic_of(image[i]);
        values[i] = v;
        lo = hi = fkey(v);
    }
    for(int off = 32; off > 0; off >>= 1) {
        const uint32_t a = __shfl_xor(lo, off), b = __shfl_xor(hi, off);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if((threadIdx.x & 63) == 0) {
        atomicMin(&range[0], lo);
        atomicMax(&range[1], hi);
    }
}

// min_brightness = min(0.0F, all values), max_brightness = max(1E-4F, all values) (:35-47), then the segment ceilings (:112-127):
// source >= 0: position in the sorted list, -1: min_brightness, -2: max_brightness
__global__ void k_ceilings(const uint32_t *__restrict__ range, const float *__restrict__ sorted, const int32_t *__restrict__ source, int segments, float *__restrict__ ceilings,
                           float *__restrict__ min_max) {
    const float vmin = fkey_inv(range[0]), vmax = fkey_inv(range[1]);
    const float min_brightness = (vmin < 0.0f) ? vmin : 0.0f;
    const float max_brightness = (1E-4f < vmax) ? vmax : 1E-4f;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i == 0) {
        min_max[0] = min_brightness;
        min_max[1] = max_brightness;
    }
    if(i < segments) {
        const int32_t src = source[i];
        ceilings[i] = src >= 0 ? sorted[src] : (src == -1 ? min_brightness : max_brightness);
    }
}

__global__ __launch_bounds__(256) void k_tone(float4 *__restrict__ image, uint32_t n, const float *__restrict__ ceilings, int segments, const float *__restrict__ min_max) {
    __shared__ float ceil_lds[1024];
    for(int i = threadIdx.x; i < segments; i += 256) {
        ceil_lds[i] = ceilings[i];
    }
    __syncthreads();
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if(p >= n) {
        return;
    }
    const float tiny = 1.17549435e-38f; // std::numeric_limits<float>::min()
    float4 c = image[p];
    const float brightness = max_std(peak_of(c), tiny);
    const float h = heuristic_of(c);
    // std::lower_bound(ceilings, h): the first ceiling that is not < h
    int lo = 0, len = segments;
    while(len > 0) {
        const int half = len >> 1;
        if(ceil_lds[lo + half] < h) {
            lo = lo + half + 1;
            len = len - half - 1;
        }
        else {
            len = half;
        }
    }
    const int segment_index = lo < segments ? lo : segments - 1; // (the last ceiling is max_brightness >= h)
    const float segment_upper = ceil_lds[segment_index];
    const float segment_lower = segment_index > 0 ? ceil_lds[segment_index - 1] : min_max[0];
    const float segment_span = max_std(segment_upper - segment_lower, tiny);
    const float segment_value = (h - segment_lower) / segment_span;
    const float mapped_upper = (float)(segment_index + 1) / (float)segments;
    const float mapped_lower = (float)segment_index / (float)segments;
    const float mapped_span = mapped_upper - mapped_lower;
    const float mapped_value = mapped_lower + segment_value * mapped_span;
    const float factor = mapped_value / brightness;
    c.x *= factor;
    c.y *= factor;
    c.z *= factor;
    image[p] = c;
}

__global__ __launch_bounds__(256) void k_gamma(float4 *__restrict__ image, uint32_t n, float exponent) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if(p >= n) {
        return;
    }
    float4 c = image[p];
    const float factor = ptm::powf_glibc_full(peak_of(c), exponent);
    c.x *= factor;
    c.y *= factor;
    c.z *= factor;
    image[p] = c;
}

// gaussian<float>(t, 0, sigma), post_processing.cpp:11-20
float bell(float t, float mu, float sigma) {
    const float pi = static_cast<float>(M_PI);
    const float fac = 1.0F / (std::sqrt(2 * pi));
    const float exponent_part = (t - mu) / (sigma);
    return fac * std::exp(-(exponent_part * exponent_part) / 2.0F) / sigma;
}

// post_processing.cpp:90-127 with the list lookups left symbolic
std::vector<int32_t> ceiling_sources(int pixel_count) {
    const int segments = std::min(1024, pixel_count);
    std::vector<float> weights;
    weights.reserve(static_cast<size_t>(segments));
    float total = 0.0F;
    for(int i = 0; i < segments; i++) {
        float x = (static_cast<float>(i) + 0.5F) / static_cast<float>(segments);
        x = 2.0F * (x - 0.5F);
        const float w = 0.1F + bell(x, 0.0F, 0.3F);
        weights.push_back(w);
        total += w;
    }
    std::vector<int32_t> source(static_cast<size_t>(segments), -2);
    int previous_index = 0;
    float missed = 0.0F;
    for(int i = 0; i < segments - 1; i++) {
        const int count = static_cast<int>(std::round(weights[i] * static_cast<float>(pixel_count) / total + missed));
        if(count > 0) {
            source[i] = std::min(previous_index + count - 1, pixel_count - 1);
            previous_index += count;
            missed = 0.0F;
        }
        else {
            source[i] = i > 0 ? source[i - 1] : -1;
            missed += weights[i] * static_cast<float>(pixel_count) / total;
        }
    }
    return source;
}

struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch() {
        for(void *p : ptrs) {
            (void)hipFree(p);
        }
    }
    template<typename T>
    hipError_t get(T **out, size_t count) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if(e == hipSuccess) {
            ptrs.push_back(p);
            *out = static_cast<T *>(p);
        }
        return e;
    }
};

} // namespace

#define PTP_TRY(call)           \
    do {                        \
        hipError_t e_ = (call); \
        if(e_ != hipSuccess) {  \
            return e_;          \
        }                       \
    } while(0)

hipError_t pt_post_run(hipStream_t stream, float4 *image, int32_t width, int32_t height, uint32_t steps, float gamma) {
    const long long count = static_cast<long long>(width) * height;
    if(count <= 0) {
        return hipSuccess;
    }
    if(count > 0x7fffffffLL) {
        return hipErrorInvalidValue;
    }
    const uint32_t n = static_cast<uint32_t>(count);
    const dim3 grid((n + 255) / 256), block(256);
    Scratch scratch;
    if(steps & PT_POST_STEP_TONE_MAP) {
        const std::vector<int32_t> source = ceiling_sources(static_cast<int>(n));
        const int segments = static_cast<int>(source.size());
        float *values = nullptr, *sorted = nullptr, *ceilings = nullptr, *min_max = nullptr;
        uint32_t *range = nullptr;
        int32_t *d_source = nullptr;
        uint8_t *temp = nullptr;
        PTP_TRY(scratch.get(&values, n));
        PTP_TRY(scratch.get(&sorted, n));
        PTP_TRY(scratch.get(&ceilings, 1024));
        PTP_TRY(scratch.get(&min_max, 2));
        PTP_TRY(scratch.get(&range, 2));
        PTP_TRY(scratch.get(&d_source, 1024));
        size_t temp_bytes = 0;
        PTP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, values, sorted, static_cast<int>(n), 0, 32, stream));
        PTP_TRY(scratch.get(&temp, temp_bytes));
        const uint32_t range_init[2] = {0xffffffffu, 0u};
        PTP_TRY(hipMemcpyAsync(range, range_init, sizeof(range_init), hipMemcpyHostToDevice, stream));
        PTP_TRY(hipMemcpyAsync(d_source, source.data(), sizeof(int32_t) * source.size(), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_heuristic, grid, block, 0, stream, image, n, values, range);
        PTP_TRY(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, values, sorted, static_cast<int>(n), 0, 32, stream));
        hipLaunchKernelGGL(k_ceilings, dim3(4), dim3(256), 0, stream, range, sorted, d_source, segments, ceilings, min_max);
        hipLaunchKernelGGL(k_tone, grid, block, 0, stream, image, n, ceilings, segments, min_max);
        PTP_TRY(hipGetLastError());
    }
    if(steps & PT_POST_STEP_GAMMA) {
        const float exponent = 1.0F / gamma - 1.0F; // post_processing.cpp:176
        hipLaunchKernelGGL(k_gamma, grid, block, 0, stream, image, n, exponent);
        PTP_TRY(hipGetLastError());
    }
    // the scratch buffers and the pageable host sources above must outlive the work
    return hipStreamSynchronize(stream);
}
