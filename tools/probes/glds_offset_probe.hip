// Probe: does the immediate offset of global_load_lds_dwordx4 advance the LDS destination as well as the global
// source?  (If yes, the 1 KiB pieces of a chunk need ONE M0 set-up.)  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];   // 4 KiB
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned* g = src + threadIdx.x * 4;
    const unsigned dst = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)lds;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = i;
    unsigned *s, *o;
    hipMalloc(&s, 4096); hipMalloc(&o, 4096);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(s, o);
    hipMemcpy(h.data(), o, 4096, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 768; ++i) ok &= h[i] == (unsigned)i;
    printf("lds[0]=%u lds[255]=%u lds[256]=%u (expect 256 if the offset moves the LDS side) lds[512]=%u lds[768]=%x -> %s\n", h[0], h[255], h[256], h[512], h[768],
           ok ? "offset applies to BOTH sides" : "offset does NOT advance LDS");
    return 0;
}
