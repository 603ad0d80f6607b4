// PathTrace/detail/core.h -- Ray, the xorshift engine and the assertion helpers of the PathTrace API (reference: base.h).
#ifndef PATHTRACE_DETAIL_CORE_H
#define PATHTRACE_DETAIL_CORE_H

#include <PathTrace/detail/linear.h>

#include <cassert>
#include <cmath>
#include <cstdint>
#include <limits>
#include <random>

// origin + unit direction
struct Ray {
    vec3<float> origin;
    vec3<float> dir;
};

// 64-bit xorshift with a multiply-high output stage; 32 bits per draw.  The device kernels run the same recurrence, so an
// engine can be handed to the GPU and taken back (state()/setState()).
class xorshift {
  public:
    xorshift(uint64_t seed) noexcept : word(seed ^ (~seed << 32)) {}

    uint32_t operator()() noexcept {
        const uint64_t product = word * 0xD989BCACC137DCD5LLU;
        word ^= word >> 11;
        word ^= word << 31;
        word ^= word >> 18;
        return static_cast<uint32_t>(product >> 32);
    }

    static constexpr uint32_t min() noexcept { return std::numeric_limits<uint32_t>::min(); }
    static constexpr uint32_t max() noexcept { return std::numeric_limits<uint32_t>::max(); }

    uint64_t state() const noexcept { return word; }
    void setState(uint64_t s) noexcept { word = s; }

  private:
    uint64_t word;
};

class RandomEngine {
  public:
    RandomEngine(auto seed) noexcept : engine(seed) {}

    auto operator()() noexcept { return engine(); }
    static constexpr auto min() noexcept { return xorshift::min(); }
    static constexpr auto max() noexcept { return xorshift::max(); }

    // raw engine word: what pt_stream::rng_state carries across the C ABI
    uint64_t state() const noexcept { return engine.state(); }
    void setState(uint64_t s) noexcept { engine.setState(s); }

  private:
    xorshift engine;
};

template<typename T, int SIZE>
bool isNormalized(impl::rt_vector<T, SIZE> v) noexcept {
    return std::abs(v.getLengthSquared() - static_cast<T>(1.0)) < static_cast<T>(1E-4);
}

template<typename T, int SIZE>
bool isNonNegative(impl::rt_vector<T, SIZE> v) noexcept {
    for(int i = 0; i < SIZE; i++) {
        if(!(v[i] >= static_cast<T>(0))) { // false for NaN
            return false;
        }
    }
    return true;
}

#define assertNormalized(x) assert(isNormalized(x))   // NOLINT
#define assertNonNegative(x) assert(isNonNegative(x)) // NOLINT
#define assertFinite(x) assert(std::isfinite(x))      // NOLINT

#endif
