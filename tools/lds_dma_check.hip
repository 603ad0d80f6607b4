// Where does global_load_lds_dwordx4 put a lane's 16 bytes, with all lanes active and with some switched off?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__global__ void k(const u4v *src, unsigned *out, unsigned long long mask) {
    __shared__ __align__(16) unsigned lds[64 * 4 + 64];
    const int lane = threadIdx.x;
    for(int i = lane; i < 64 * 4 + 64; i += 64) {
        lds[i] = 0xdeadbeefu;
    }
    __syncthreads();
    if((mask >> lane) & 1ULL) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane), (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    for(int i = lane; i < 64 * 4 + 64; i += 64) {
        out[i] = lds[i];
    }
}
int main() {
    std::vector<unsigned> h(64 * 4);
    for(int l = 0; l < 64; l++) for(int c = 0; c < 4; c++) h[l * 4 + c] = (unsigned)(l * 16 + c); // value = lane*16 + component
    u4v *d; unsigned *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, (64 * 4 + 64) * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for(unsigned long long mask : {~0ULL, 0xff00ff00ff00ff00ULL, 0x00000000f0f0f0f0ULL}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, mask);
        std::vector<unsigned> r(64 * 4 + 64);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        int contiguous = 0, wrong = 0, untouched_ok = 0;
        for(int l = 0; l < 64; l++) {
            const bool on = (mask >> l) & 1ULL;
            for(int c = 0; c < 4; c++) {
                const unsigned v = r[l * 4 + c];
                if(on && v == (unsigned)(l * 16 + c)) contiguous++;
                else if(!on && v == 0xdeadbeefu) untouched_ok++;
                else wrong++;
            }
        }
        printf("mask %016llx: lane*16 layout matches for %d dwords, inactive slots untouched %d, other %d; first words:", mask, contiguous, untouched_ok, wrong);
        for(int i = 0; i < 12; i++) printf(" %x", r[i]);
        printf("\n");
    }
    return 0;
}
