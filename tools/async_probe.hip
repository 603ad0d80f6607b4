// Does a wavefront have to wait for its slowest lane's record?  (gfx950)
//
// A BVH walk is a dependent chain per lane: fetch a 64-byte record, do ~100 instructions on it, learn the next address.  With the record
// in registers (4 x global_load_dwordx4) the wavefront's s_waitcnt ends when the SLOWEST lane's lines have arrived, so one lane that
// goes to HBM holds the other 63 (profiles/r03_pmc_derived.json: waves wait 63 % of their cycles).  This probe measures the alternative:
// the record is delivered straight into LDS (global_load_lds_dwordx4, one 16-byte slot per lane and quarter), the slots are pre-filled
// with a sentinel, and every iteration each lane LOOKS whether its four quarters have arrived: lanes that have theirs make their hop and
// request the next record, the others sit the iteration out -- no vmcnt wait anywhere.
//
//   mode 0: lockstep (registers, s_waitcnt vmcnt(0) before use)      mode 1: per-lane arrival through LDS sentinels
//   mode 2: lockstep, but the record comes quad-cooperatively (one 64-byte request per quad and record instead of four 16-byte requests per lane)
//           straight into LDS and is read back by its lane: a quarter of the L1 look-ups for one more LDS round trip per hop
// Access pattern: a hop goes to a COLD record (uniform over `cold` records: HBM) with probability p_cold/256, else to a HOT one (uniform
// over `hot` records: L1/L2).  The next index depends on the record just read.  Every record's words satisfy w[k] = hash(i) + k (low 31
// bits), which lets mode 1 count torn records (a quarter seen half-written).
// Build: hipcc --offload-arch=gfx950 -O3 tools/async_probe.hip -o tools/bin/async_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef const u4v __attribute__((address_space(1))) *glb_u4;
typedef u4v __attribute__((address_space(3))) *lds_u4;
typedef unsigned __attribute__((address_space(3))) *lds_u1;

#define SENT 0xffffffffu

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void fill(u4v *recs, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if(i < n) {
        const uint32_t h = hash32(i);
        for(int q = 0; q < 4; q++) {
            u4v v;
            v.x = (h + 4 * q + 0) & 0x7fffffffu;
            v.y = (h + 4 * q + 1) & 0x7fffffffu;
            v.z = (h + 4 * q + 2) & 0x7fffffffu;
            v.w = (h + 4 * q + 3) & 0x7fffffffu;
            recs[4 * (size_t)i + q] = v;
        }
    }
}

// the "work" of a hop: `alu` dependent multiply-adds on the record's words
__device__ __forceinline__ uint32_t work(u4v r0, u4v r1, u4v r2, u4v r3, int alu) {
    float a = __uint_as_float((r0.x & 0x007fffffu) | 0x3f800000u), b = __uint_as_float((r1.y & 0x007fffffu) | 0x3f800000u);
    float c = __uint_as_float((r2.z & 0x007fffffu) | 0x3f800000u), d = __uint_as_float((r3.w & 0x007fffffu) | 0x3f800000u);
#pragma unroll 1
    for(int k = 0; k < alu; k += 4) {
        a = a * 0.999f + b;
        b = b * 0.998f + c;
        c = c * 0.997f + d;
        d = d * 0.996f + a;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    return __float_as_uint(a + b + c + d);
}

__device__ __forceinline__ uint32_t next_index(uint32_t &rng, uint32_t mix, uint32_t hot_mask, uint32_t cold_mask, uint32_t p_cold, uint32_t hot_base) {
    rng = rng * 1664525u + 1013904223u + (mix & 0xffu);
    const uint32_t u = hash32(rng);
    const bool cold = ((u >> 24) & 0xffu) < p_cold;
    return cold ? (u & cold_mask) : hot_base + (u & hot_mask);
}

template<int MODE>
__global__ __launch_bounds__(256, 4) void chase(const u4v *__restrict__ recs_, uint32_t hot_mask, uint32_t cold_mask, uint32_t p_cold, int iters, int alu,
                                                 unsigned long long *out) {
    __shared__ __align__(16) unsigned char lds_raw[4 * 4224];
    glb_u4 recs = (glb_u4)recs_;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    unsigned char *region = lds_raw + wib * 4224;
    lds_u4 mine = (lds_u4)(u4v *)region + lane; // quarter q at mine[64 * q]
    uint32_t rng = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t hot_base = cold_mask + 1u; // the hot records lie behind the cold ones
    uint32_t hops = 0, torn = 0, acc = 0;
    uint32_t idx = next_index(rng, 0, hot_mask, cold_mask, p_cold, hot_base);
    if(MODE == 0) {
        u4v r0, r1, r2, r3;
        glb_u4 p = recs + 4 * (size_t)idx;
        r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3];
#pragma unroll 1
        for(int it = 0; it < iters; it++) {
            const uint32_t m = work(r0, r1, r2, r3, alu);
            acc += m;
            hops++;
            idx = next_index(rng, r0.x ^ m, hot_mask, cold_mask, p_cold, hot_base);
            p = recs + 4 * (size_t)idx;
            r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3];
        }
    }
    else if(MODE == 2) {
        // quad-cooperative fetch straight into LDS: instruction i brings the record of the quad's lane i (lane p its quarter p: one 64-byte
        // request per quad instead of four 16-byte ones per lane), then every lane reads its own record back (regions padded by 16 bytes:
        // the records of a quad's four lanes would otherwise start in the same bank)
        const uint32_t qoff = (lane & 3u) * 16u;
        unsigned char *rd = region + (lane & 3u) * 1040u + (lane >> 2) * 64u;
        lds_u4 back = (lds_u4)(u4v *)rd;
        uint32_t off = idx * 64u;
#pragma unroll 1
        for(int it = 0; it <= iters; it++) {
            {
                const uint32_t o0 = __builtin_amdgcn_update_dpp(0u, off, 0x00, 0xf, 0xf, false) + qoff;
                const uint32_t o1 = __builtin_amdgcn_update_dpp(0u, off, 0x55, 0xf, 0xf, false) + qoff;
                const uint32_t o2 = __builtin_amdgcn_update_dpp(0u, off, 0xaa, 0xf, 0xf, false) + qoff;
                const uint32_t o3 = __builtin_amdgcn_update_dpp(0u, off, 0xff, 0xf, 0xf, false) + qoff;
                const char __attribute__((address_space(1))) *base = (const char __attribute__((address_space(1))) *)recs;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + o0), (__attribute__((address_space(3))) void *)(region + 0), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + o1), (__attribute__((address_space(3))) void *)(region + 1040), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + o2), (__attribute__((address_space(3))) void *)(region + 2080), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + o3), (__attribute__((address_space(3))) void *)(region + 3120), 16, 0, 0);
            }
            if(it == iters) {
                break;
            }
            __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
            const u4v r0 = back[0], r1 = back[1], r2 = back[2], r3 = back[3];
            const uint32_t h = hash32(idx);
            if(r0.x != (h & 0x7fffffffu) || r3.w != ((h + 15) & 0x7fffffffu)) {
                torn++;
            }
            const uint32_t m = work(r0, r1, r2, r3, alu);
            acc += m;
            hops++;
            idx = next_index(rng, r0.x ^ m, hot_mask, cold_mask, p_cold, hot_base);
            off = idx * 64u;
        }
        __builtin_amdgcn_s_waitcnt(0x0000);
    }
    else {
        const u4v s = {SENT, SENT, SENT, SENT};
        mine[0] = s; mine[64] = s; mine[128] = s; mine[192] = s;
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
        {
            glb_u4 p = recs + 4 * (size_t)idx;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 0), (__attribute__((address_space(3))) void *)(region + 0), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 1), (__attribute__((address_space(3))) void *)(region + 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 2), (__attribute__((address_space(3))) void *)(region + 2048), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 3), (__attribute__((address_space(3))) void *)(region + 3072), 16, 0, 0);
        }
#pragma unroll 1
        for(int it = 0; it < iters; it++) {
            const u4v r0 = mine[0], r1 = mine[64], r2 = mine[128], r3 = mine[192];
            // a quarter has arrived when none of its words is the sentinel (the data never contains it)
            const bool here = r0.x != SENT && r0.y != SENT && r0.z != SENT && r0.w != SENT && r1.x != SENT && r1.y != SENT && r1.z != SENT && r1.w != SENT &&
                              r2.x != SENT && r2.y != SENT && r2.z != SENT && r2.w != SENT && r3.x != SENT && r3.y != SENT && r3.z != SENT && r3.w != SENT;
            if(here) {
                const uint32_t h = hash32(idx);
                if(r0.x != (h & 0x7fffffffu) || r3.w != ((h + 15) & 0x7fffffffu) || r1.y != ((h + 5) & 0x7fffffffu) || r2.z != ((h + 10) & 0x7fffffffu)) {
                    torn++;
                }
                const uint32_t m = work(r0, r1, r2, r3, alu);
                acc += m;
                hops++;
                idx = next_index(rng, r0.x ^ m, hot_mask, cold_mask, p_cold, hot_base);
                mine[0] = s; mine[64] = s; mine[128] = s; mine[192] = s;
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the sentinels are in place before the new record can land
                glb_u4 p = recs + 4 * (size_t)idx;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 0), (__attribute__((address_space(3))) void *)(region + 0), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 1), (__attribute__((address_space(3))) void *)(region + 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 2), (__attribute__((address_space(3))) void *)(region + 2048), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 3), (__attribute__((address_space(3))) void *)(region + 3072), 16, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0000); // everything landed before the LDS is given back
    }
    // wave totals
    for(int off = 32; off > 0; off >>= 1) {
        hops += __shfl_down(hops, off);
        torn += __shfl_down(torn, off);
        acc += __shfl_down(acc, off);
    }
    if(lane == 0) {
        atomicAdd(&out[0], (unsigned long long)hops);
        atomicAdd(&out[1], (unsigned long long)torn);
        atomicAdd(&out[2], (unsigned long long)acc);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t cold_n = 1u << 23; // 512 MB
    const uint32_t hot_max = 1u << 16;
    u4v *recs;
    CHECK(hipMalloc(&recs, ((size_t)cold_n + hot_max) * 64));
    hipLaunchKernelGGL(fill, dim3((cold_n + hot_max + 255) / 256), dim3(256), 0, 0, recs, cold_n + hot_max);
    unsigned long long *out;
    CHECK(hipMalloc(&out, 64));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%d CUs, %d iterations per wavefront, 4 workgroups of 256 per CU\n", cus, iters);
    for(int wg_per_cu : {4, 1}) {
        for(int alu : {40, 120}) {
            for(uint32_t hot_n : {256u, 4096u, 65536u}) {
                for(uint32_t p_cold : {0u, 4u, 16u, 64u}) {
                    double rate[3] = {0, 0, 0};
                    unsigned long long torn_total = 0, hops1 = 0;
                    for(int mode = 0; mode < 3; mode++) {
                        CHECK(hipMemset(out, 0, 64));
                        const dim3 grid(cus * wg_per_cu);
                        for(int rep = 0; rep < 2; rep++) { // first run warms up
                            CHECK(hipMemset(out, 0, 64));
                            CHECK(hipEventRecord(e0));
                            if(mode == 0) {
                                hipLaunchKernelGGL(chase<0>, grid, dim3(256), 0, 0, recs, hot_n - 1, cold_n - 1, p_cold, iters, alu, out);
                            }
                            else if(mode == 1) {
                                hipLaunchKernelGGL(chase<1>, grid, dim3(256), 0, 0, recs, hot_n - 1, cold_n - 1, p_cold, iters, alu, out);
                            }
                            else {
                                hipLaunchKernelGGL(chase<2>, grid, dim3(256), 0, 0, recs, hot_n - 1, cold_n - 1, p_cold, iters, alu, out);
                            }
                            CHECK(hipEventRecord(e1));
                            CHECK(hipEventSynchronize(e1));
                        }
                        float ms = 0;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        unsigned long long h[3];
                        CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
                        rate[mode] = (double)h[0] / (ms * 1e-3) * 1e-9;
                        if(mode == 1) {
                            hops1 = h[0];
                        }
                        if(mode >= 1) {
                            torn_total += h[1];
                        }
                    }
                    const double full = (double)cus * wg_per_cu * 256 * iters;
                    printf("wg/CU %d alu %3d hot %6u p_cold %2u/256: lockstep %7.2f G hops/s, per-lane arrival %7.2f (x%.2f; lanes hopping per iteration %.1f of 64), quad-cooperative into LDS %7.2f (x%.2f); wrong records %llu\n",
                           wg_per_cu, alu, hot_n, p_cold, rate[0], rate[1], rate[1] / rate[0], 64.0 * (double)hops1 / full, rate[2], rate[2] / rate[0], torn_total);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
