// gather_bench.hip -- how fast can gfx950 fetch random 64-byte records (BVH node pairs) from a table far larger than L2?
// Variants: (A) each lane loads its own record with 4 x global_load_dwordx4 (what the path kernel does per traversal step),
//           (B) quad-cooperative: 4 adjacent lanes load the 4 x 16-byte pieces of one record in ONE instruction (the texture
//               addresser sees one 64-byte request per quad), 4 instructions cover the 4 records of the quad, pieces are
//               exchanged through LDS,
//           (C) as A but with 2 x dwordx4 + dependent chain only on the first (lower bound for "half the record").
// Each lane follows a dependent chain (next index = f(loaded data)), like a BVH walk.
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));

__device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

template<int VARIANT>
__global__ __launch_bounds__(256) void gather(const f4v *__restrict__ table, uint32_t n_records, int steps, float *out) {
    __shared__ f4v xch[256 * 4];
    const int tid = threadIdx.x;
    uint32_t idx = mix(blockIdx.x * 256 + tid) % n_records;
    float acc = 0.0f;
    for(int s = 0; s < steps; s++) {
        f4v a, b, c, d;
        if(VARIANT == 0) {
            const f4v *p = table + 4 * (size_t)idx;
            a = p[0];
            b = p[1];
            c = p[2];
            d = p[3];
        }
        else if(VARIANT == 1) {
            const int quad = tid & ~3, sub = tid & 3;
            f4v piece[4];
#pragma unroll
            for(int j = 0; j < 4; j++) {
                const uint32_t other = __shfl(idx, (tid & 63 & ~3) + j, 64);
                piece[j] = table[4 * (size_t)other + sub];
            }
#pragma unroll
            for(int j = 0; j < 4; j++) {
                xch[(quad + j) * 4 + sub] = piece[j]; // record of lane quad+j, piece sub
            }
            __builtin_amdgcn_wave_barrier();
            a = xch[tid * 4 + 0];
            b = xch[tid * 4 + 1];
            c = xch[tid * 4 + 2];
            d = xch[tid * 4 + 3];
            __builtin_amdgcn_wave_barrier();
        }
        else {
            const f4v *p = table + 4 * (size_t)idx;
            a = p[0];
            b = p[1];
            c = a;
            d = b;
        }
        const float v = a.x + b.y + c.z + d.w;
        acc += v;
        idx = mix(idx + __float_as_uint(v)) % n_records;
    }
    out[blockIdx.x * 256 + tid] = acc;
}

int main(int argc, char **argv) {
    const size_t kb = argc > 1 ? atol(argv[1]) : 461 * 1024; // table size in KiB
    const size_t mb = kb / 1024;
    const int steps = argc > 2 ? atoi(argv[2]) : 64;
    const uint32_t n_records = (uint32_t)(kb * 1024 / 64);
    f4v *table;
    float *out;
    hipMalloc(&table, (size_t)n_records * 64);
    hipMemset(table, 0x3c, (size_t)n_records * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for(int blocks_per_cu : {2, 4, 8}) {
        const int grid = 256 * blocks_per_cu;
        hipMalloc(&out, (size_t)grid * 256 * 4);
        for(int variant = 0; variant < 3; variant++) {
            float best = 1e30f;
            for(int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if(variant == 0) hipLaunchKernelGGL(gather<0>, dim3(grid), dim3(256), 0, 0, table, n_records, steps, out);
                if(variant == 1) hipLaunchKernelGGL(gather<1>, dim3(grid), dim3(256), 0, 0, table, n_records, steps, out);
                if(variant == 2) hipLaunchKernelGGL(gather<2>, dim3(grid), dim3(256), 0, 0, table, n_records, steps, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if(ms < best) best = ms;
            }
            const double recs = (double)grid * 256 * steps;
            printf("table %zu KiB (%zu MB)  blocks/CU %d  variant %d: %.3f ms  %.1f Grec/s  %.0f GB/s (64 B/record)  %.0f ns per dependent step\n", kb, mb, blocks_per_cu, variant, best,
                   recs / best / 1e6, recs * 64 / best / 1e6, best * 1e6 / steps);
        }
        hipFree(out);
    }
    return 0;
}
