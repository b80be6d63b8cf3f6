// Probe: how fast can EVERY CU stream the same weight stream from the XCD's L2 into its LDS with global_load_lds_dwordx4?
// (The compensated-float16 NeRF kernel, mlp_pipe_c.h, streams 2.1 MB per 128-sample workgroup = 8.6 GB per 4096 x 128 launch; its
// ablations say the weight DMA, not the LDS reads or the matrix pipe, sets its 0.88 ms.)  One workgroup of 4 wavefronts per CU (84 KiB
// of LDS, as the kernel), 16 KiB chunks of sixteen 1 KiB pieces, 4-slot ring, counted vmcnt, optional barrier per chunk.
//   layout 0: chunk-contiguous          piece p of chunk c at c * 16 KiB + p * 1 KiB       (what the kernels do)
//   layout 1: piece-strided             piece p of chunk c at p * STRIDE + c * 1 KiB       (a chunk's pieces 16 x STRIDE apart)
//   layout 2: chunk-contiguous, every workgroup starts at its own chunk (rotation)        (not available to a real layer sequence)
//   layout 3: as 0 with plain global_load_dwordx4 into registers (no LDS write)
// hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int CB = 16384, NSLOT = 4;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int LAYOUT, bool BARRIER>
__global__ __launch_bounds__(256) void k_stream(const char* __restrict__ src, int nch, unsigned stride, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned dst0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring + wave * 4096);
    const int rot = LAYOUT == 2 ? (int)((blockIdx.x * 37u) % (unsigned)nch) : 0;
    unsigned acc = 0;
    auto issue = [&](int c) {
        int cc = c + rot; if (cc >= nch) cc -= nch;
        const unsigned dst = dst0 + (unsigned)(c & (NSLOT - 1)) * CB;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = wave * 4 + k;
            const unsigned off = (LAYOUT == 1 ? (unsigned)p * stride + (unsigned)cc * 1024u : (unsigned)cc * CB + (unsigned)p * 1024u) + lane * 16;
            if (LAYOUT == 3) {
                uint4 v = *reinterpret_cast<const uint4*>(src + off);
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
                acc += v.x;
            } else {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(src), "s"(dst + k * 1024) : "memory");
            }
        }
    };
    for (int c = 0; c < NSLOT - 1 && c < nch; ++c) issue(c);
    for (int c = 0; c < nch; ++c) {
        if (LAYOUT != 3) wait_vm<8>();                       // chunk c landed (this wavefront's pieces); c + 1, c + 2 stay in flight
        if (BARRIER) __builtin_amdgcn_s_barrier();
        if (c + NSLOT - 1 < nch) issue(c + NSLOT - 1);
    }
    if (LAYOUT != 3) wait_vm<0>();
    __syncthreads();
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = acc + *reinterpret_cast<unsigned*>(ring + threadIdx.x * 4);
}

template <int LAYOUT, bool BARRIER>
static double run(const char* src, int nch, unsigned stride, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = NSLOT * CB + 20480;
    hipFuncSetAttribute((const void*)k_stream<LAYOUT, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 2; ++i) k_stream<LAYOUT, BARRIER><<<blocks, 256, lds>>>(src, nch, stride, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) k_stream<LAYOUT, BARRIER><<<blocks, 256, lds>>>(src, nch, stride, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 4096;
    const int nchs[3] = {134, 76, 32};              // 2.1 MB (f16c stream), 1.2 MB (f16 stream), 0.5 MB
    for (int nch : nchs) {
        // piece stride: the stream's sixteenth, rounded up to 4 KiB, plus one 4 KiB + 256 B step so that neighbouring pieces differ in every plausible channel bit
        const unsigned stride = ((unsigned)nch * 1024u + 4095u) / 4096u * 4096u + 4096u + 256u;
        const size_t bytes = (size_t)16 * stride + (size_t)nch * CB;
        char* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
        const double gb = (double)blocks * nch * CB / 1e9;
        printf("stream %4d KiB x %d workgroups = %.2f GB per launch\n", nch * 16, blocks, gb);
#define RUN(L, B, name) { double ms = run<L, B>(src, nch, stride, blocks, 10); \
        printf("  %-44s %.3f ms  %6.2f TB/s  %5.1f B/ns per CU\n", name, ms, gb / ms, gb / ms * 1e3 / 256); }
        RUN(0, true, "contiguous chunks, barrier per chunk");
        RUN(0, false, "contiguous chunks, no barrier");
        RUN(1, true, "piece-strided, barrier per chunk");
        RUN(1, false, "piece-strided, no barrier");
        RUN(2, true, "rotated start per workgroup, barrier");
        RUN(3, false, "plain global_load_dwordx4 to registers");
        hipFree(src);
    }
    return 0;
}
