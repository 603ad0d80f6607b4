// What a wavefront's instruction stream costs on gfx950 when the wavefront is ALONE on its SIMD and when it shares it: cycles per
// instruction for the instruction kinds a BVH traversal step is made of.  The strong-scaling tail of a render is a handful of such lonely
// wavefronts (tools/chain_probe.py), so their instruction latencies -- not the chip's throughput -- set the frame time.
//
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/bin/issue_probe && tools/bin/issue_probe
//
// Each test is a loop of ITER iterations of a hand-written block (inline asm, so the compiler neither removes nor reorders it), timed
// with s_memtime around the loop by lane 0 of every wavefront; the figure printed is the median over wavefronts of cycles per block
// divided by the number of instructions in the block.  Launch shapes: 1 wavefront per workgroup and 1 workgroup per CU ("alone"),
// 4 wavefronts per SIMD ("4/SIMD": 4 workgroups of 256 threads per CU), 8 per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

constexpr int ITER = 512;

template<int TEST>
__global__ void probe(unsigned long long *out, const float *table, float seed) {
    __shared__ float lds[1024];
    lds[threadIdx.x & 1023] = seed;
    __syncthreads();
    float a = seed + threadIdx.x, b = seed * 3.0f, c = 1.0f, d = 2.0f;
    uint32_t ia = threadIdx.x * 4u;
    const float *p = table + (threadIdx.x & 63);
    float l0 = 0.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for(int it = 0; it < ITER; it++) {
        if(TEST == 0) { // 64 dependent v_add_f32
            asm volatile(REP64("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        }
        else if(TEST == 1) { // 64 v_add_f32 in four independent chains
            asm volatile(REP16("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
        }
        else if(TEST == 2) { // 64 dependent v_pk_add_f32
            asm volatile(REP64("v_pk_add_f32 %0, %0, %1\n") : "+v"(*(double *)&a) : "v"(*(double *)&c));
        }
        else if(TEST == 3) { // 64 dependent s_add_u32
            uint32_t s = it;
            asm volatile(REP64("s_add_u32 %0, %0, 3\n") : "+s"(s) : : "scc");
            ia += s;
        }
        else if(TEST == 4) { // 16 x (v_cmp -> s_and_b64 -> v_cndmask): VALU -> SGPR -> SALU -> VALU hops; 48 instructions
            unsigned long long m;
            asm volatile(REP16("v_cmp_lt_f32 %1, %0, %2\n s_and_b64 %1, %1, exec\n v_cndmask_b32 %0, %0, %3, %1\n") : "+v"(a), "=&s"(m) : "v"(b), "v"(c) : "scc");
        }
        else if(TEST == 5) { // 16 x (v_cmp into vcc -> v_cndmask on vcc); 32 instructions
            asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n") : "+v"(a) : "v"(b), "v"(c) : "vcc");
        }
        else if(TEST == 6) { // 16 taken branches, each over one instruction, + 16 v_add; 32 executed instructions (+16 skipped)
            asm volatile(REP16("s_branch 1f\n v_add_f32 %0, %0, %1\n 1: v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        }
        else if(TEST == 7) { // 16 NOT taken conditional branches + 16 v_add; 48 instructions (s_cmp, s_cbranch, v_add)
            asm volatile(REP16("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n v_add_f32 %0, %0, %1\n 1:\n") : "+v"(a) : "v"(b) : "scc");
        }
        else if(TEST == 8) { // 16 x (s_and_saveexec, v_add, s_or exec): an exec-masked region; 48 instructions
            unsigned long long sv;
            asm volatile(REP16("s_and_saveexec_b64 %1, exec\n v_add_f32 %0, %0, %2\n s_or_b64 exec, exec, %1\n") : "+v"(a), "=&s"(sv) : "v"(b) : "scc");
        }
        else if(TEST == 9) { // 16 x dependent (ds_write_b32, ds_read_b32, wait): LDS round trip; 48 instructions
            asm volatile(REP16("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(ia) : "memory");
        }
        else if(TEST == 10) { // 16 x dependent (global_load_dword from one cached line, wait, v_add): L1-hit load latency; 48 instructions
            asm volatile(REP16("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)\n v_add_f32 %2, %2, %0\n") : "=&v"(l0), "+v"(p), "+v"(a)::"memory");
        }
        else if(TEST == 11) { // 16 x (4 x global_load_dwordx4 of one cached 64-byte record, wait): a record fetch that hits; 80 instructions
            float4 r0, r1, r2, r3;
            asm volatile(REP16("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n global_load_dwordx4 %3, %4, off offset:48\n s_waitcnt vmcnt(0)\n")
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(p) : "memory");
            a += r0.x + r3.w;
        }
        else if(TEST == 12) { // 64 dependent v_max3_f32 / v_min_f32 alternating
            asm volatile(REP16("v_min_f32 %0, %0, %1\n v_max3_f32 %0, %0, %1, %2\n v_max_f32 %0, %0, %2\n v_min3_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        }
        else if(TEST == 13) { // 16 x (v_readfirstlane -> s_cmp -> s_cbranch not taken): scalar decisions on vector data; 48 instructions
            uint32_t s;
            asm volatile(REP16("v_readfirstlane_b32 %1, %0\n s_cmp_eq_u32 %1, 77\n s_cbranch_scc1 1f\n 1:\n") : "+v"(ia), "=&s"(s) : : "scc");
        }
        else if(TEST == 14) { // 16 x (v_cmp to SGPR pair -> s_cmp_lg_u64 -> s_cbranch_scc1 taken over one v_add): a wave-uniform branch on a ballot
            unsigned long long m;
            asm volatile(REP16("v_cmp_lt_f32 %1, %0, %2\n s_cmp_eq_u64 %1, 0\n s_cbranch_scc1 1f\n v_add_f32 %0, %0, %3\n 1:\n") : "+v"(a), "=&s"(m) : "v"(b), "v"(c) : "scc");
        }
        else if(TEST == 15) { // 16 taken branches FAR apart (each jumps over 64 instructions = 256+ bytes): instruction fetch after a jump
            asm volatile(REP16("s_branch 1f\n" REP64("v_add_f32 %0, %0, %1\n") "1: v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if((threadIdx.x & 63) == 0) {
        out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    }
    if(a + b + c + d + l0 == 12345.678f && ia == 77u) {
        out[0] = 0; // keeps the results alive
    }
}

struct Test {
    const char *name;
    int instructions;
};
static const Test tests[] = {
    {"64 dependent v_add_f32", 64}, {"64 v_add_f32, 4 independent chains", 64}, {"64 dependent v_pk_add_f32", 64}, {"64 dependent s_add_u32", 64},
    {"16 x (v_cmp -> SGPR, s_and_b64, v_cndmask on it)", 48}, {"16 x (v_cmp -> vcc, v_cndmask)", 32}, {"16 x (taken s_branch over 1 instruction, v_add)", 32},
    {"16 x (s_cmp, not-taken s_cbranch, v_add)", 48}, {"16 x (s_and_saveexec, v_add, s_or exec)", 48}, {"16 x (ds_write, ds_read, wait) dependent", 48},
    {"16 x (global_load_dword L1 hit, wait, v_add)", 48}, {"16 x (4 x global_load_dwordx4 of a cached record, wait)", 80}, {"64 dependent v_min/v_max3/v_max/v_min3", 64},
    {"16 x (v_readfirstlane, s_cmp, not-taken s_cbranch)", 48}, {"16 x (v_cmp -> SGPR, s_cmp_eq_u64, taken s_cbranch, -)", 48}, {"16 x (taken s_branch over 64 instructions, v_add)", 32}};

template<int TEST>
void run(unsigned long long *d_out, const float *d_table, int n_cu) {
    const struct {
        const char *label;
        int blocks, threads;
    } shapes[] = {{"alone", n_cu, 64}, {"4/SIMD", n_cu * 4, 256}, {"8/SIMD", n_cu * 8, 256}};
    printf("%-62s", tests[TEST].name);
    for(const auto &sh : shapes) {
        const int waves = sh.blocks * sh.threads / 64;
        std::vector<unsigned long long> h(waves);
        for(int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL((probe<TEST>), dim3(sh.blocks), dim3(sh.threads), 0, 0, d_out, d_table, 1.5f);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double per_block = (double)h[waves / 2] / ITER;
        printf("  %s %7.1f cyc/block = %5.2f per instruction", sh.label, per_block, per_block / tests[TEST].instructions);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long *d_out;
    float *d_table;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 4 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_table, 4096));
    CHECK(hipMemset(d_table, 0, 4096));
    printf("%s, %d CUs; cycles of s_memtime per block of hand-written instructions, median over wavefronts\n", prop.name, n_cu);
    run<0>(d_out, d_table, n_cu);
    run<1>(d_out, d_table, n_cu);
    run<2>(d_out, d_table, n_cu);
    run<3>(d_out, d_table, n_cu);
    run<4>(d_out, d_table, n_cu);
    run<5>(d_out, d_table, n_cu);
    run<6>(d_out, d_table, n_cu);
    run<7>(d_out, d_table, n_cu);
    run<8>(d_out, d_table, n_cu);
    run<9>(d_out, d_table, n_cu);
    run<10>(d_out, d_table, n_cu);
    run<11>(d_out, d_table, n_cu);
    run<12>(d_out, d_table, n_cu);
    run<13>(d_out, d_table, n_cu);
    run<14>(d_out, d_table, n_cu);
    run<15>(d_out, d_table, n_cu);
    return 0;
}
