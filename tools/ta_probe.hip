// Probe of the vector-memory front end (texture addresser + L1) of one CU for the access shapes a BVH walk can use to fetch a 64-byte
// record per lane from scattered addresses.  Loads are independent (indices come from a per-lane LCG), 16 wavefronts per CU, so the
// result is a throughput, not a latency.  Build: hipcc --offload-arch=gfx950 -O3 tools/ta_probe.hip -o tools/bin/ta_probe
//   mode 0: every lane 4 x global_load_dwordx4 from its own record (what pt_path.hip does)
//   mode 1: quad-cooperative: instruction i fetches the record of the quad's lane i, lane p takes quarter p
//   mode 2: every lane 4 x global_load_dword from its own record (same lines, a quarter of the bytes)
//   mode 3: every lane 1 x global_load_dwordx4 from its own record
//   mode 4: every lane 2 x global_load_dwordx4 from its own record (32-byte records)
//   mode 5: quad-cooperative straight into LDS (global_load_lds_dwordx4: region i holds the records of the quads' lanes i), then every
//           lane reads its own record back with 4 x ds_read_b128 -- the full price of delivering a whole record to its lane
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

typedef float f4v __attribute__((ext_vector_type(4)));

template<int I>
__device__ __forceinline__ void coop_lds(const f4v *recs, uint32_t r, int lane, unsigned char *region) {
    const uint32_t ri = __builtin_amdgcn_update_dpp(0u, r, (I | I << 2 | I << 4 | I << 6), 0xf, 0xf, false);
    const f4v *a = recs + 4 * (size_t)ri + (lane & 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)a, (__attribute__((address_space(3))) void *)(region + I * 1024), 16, 0, 0);
}

template<int MODE>
__global__ __launch_bounds__(256, 4) void probe(const f4v *__restrict__ recs, uint32_t mask, int iters, float *out) {
    __shared__ __align__(16) unsigned char lds_rec[MODE == 5 ? 2 * 4 * 4096 : 16];
    unsigned char *my_region = lds_rec + (threadIdx.x >> 6) * 4096;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    f4v acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    for(int it = 0; it < iters; it++) {
#pragma unroll
        for(int u = 0; u < 2; u++) {
            idx = idx * 1664525u + 1013904223u;
            const uint32_t r = (idx >> 8) & mask;
            if(MODE == 0) {
                const f4v *p = recs + 4 * (size_t)r;
                acc += p[0] + p[1] + p[2] + p[3];
            }
            else if(MODE == 1) {
#pragma unroll
                for(int i = 0; i < 4; i++) {
                    const uint32_t ri = __shfl(r, (lane & ~3) + i);
                    acc += recs[4 * (size_t)ri + (lane & 3)];
                }
            }
            else if(MODE == 2) {
                const float *p = (const float *)(recs + 4 * (size_t)r);
                acc.x += p[1] + p[5] + p[9] + p[13];
            }
            else if(MODE == 3) {
                acc += recs[4 * (size_t)r];
            }
            else if(MODE == 5) {
                unsigned char *region = my_region + u * 16384; // two sets in flight
                coop_lds<0>(recs, r, lane, region);
                coop_lds<1>(recs, r, lane, region);
                coop_lds<2>(recs, r, lane, region);
                coop_lds<3>(recs, r, lane, region);
                const f4v *rec = (const f4v *)(region + (lane & 3) * 1024 + (lane >> 2) * 64);
                acc += rec[0] + rec[1] + rec[2] + rec[3];
            }
            else {
                const f4v *p = recs + 4 * (size_t)r;
                acc += p[0] + p[1];
            }
        }
    }
    if(acc.x + acc.y + acc.z + acc.w == 123.456f) {
        out[0] = acc.x;
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3; // Hz
    float *out;
    CHECK(hipMalloc(&out, 64));
    const uint32_t sizes[] = {128, 512, 4096, 65536, 1u << 23}; // records of 64 B: 8 KB, 32 KB, 256 KB, 4 MB, 512 MB
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for(uint32_t n : sizes) {
        f4v *recs;
        CHECK(hipMalloc(&recs, (size_t)n * 64));
        CHECK(hipMemset(recs, 0, (size_t)n * 64));
        for(int mode = 0; mode < 6; mode++) {
            float best = 1e30f;
            for(int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                const dim3 grid(cus * 4), block(256);
                switch(mode) {
                case 0: hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, recs, n - 1, iters, out); break;
                case 1: hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, recs, n - 1, iters, out); break;
                case 2: hipLaunchKernelGGL(probe<2>, grid, block, 0, 0, recs, n - 1, iters, out); break;
                case 3: hipLaunchKernelGGL(probe<3>, grid, block, 0, 0, recs, n - 1, iters, out); break;
                case 4: hipLaunchKernelGGL(probe<4>, grid, block, 0, 0, recs, n - 1, iters, out); break;
                default: hipLaunchKernelGGL(probe<5>, grid, block, 0, 0, recs, n - 1, iters, out); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double lane_records = (double)cus * 4 * 256 * iters * 2;
            const double per_clk_cu = lane_records / cus / (best * 1e-3 * clk);
            printf("records %8u (%7.0f KB) mode %d: %8.3f ms, %.3f lane-records per clock per CU (nominal %.2f GHz), %.1f G records/s\n", n, n * 64 / 1024.0,
                   mode, best, per_clk_cu, clk / 1e9, lane_records / (best * 1e-3) / 1e9);
        }
        CHECK(hipFree(recs));
    }
    return 0;
}
