// Micro-probe: sustained issue rate of v_mfma_f32_32x32x16_bf16 as a function of independent accumulator
// chains per wavefront and wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(512) void k(const bf16x8* a, const bf16x8* b, float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 av = a[lane], bv = b[lane];
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS>
void run(int threads, const bf16x8* a, const bf16x8* b, float* out, long long* cyc) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<blocks, threads>>>(a, b, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS><<<blocks, threads>>>(a, b, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_wave = (double)iters * 8 * CHAINS;
    const double waves_per_simd = threads / 64 / 4.0;
    const double tf = 2.0 * 32 * 32 * 16 * mfma_per_wave * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
    printf("chains %d  waves/SIMD %.0f : %.3f ms  %.1f TF  wall-cycles/MFMA/SIMD @2.4GHz %.1f   s_memtime cyc/MFMA (wave0) %.1f\n", CHAINS, waves_per_simd, ms, tf,
           ms * 1e-3 * 2.4e9 / (mfma_per_wave * waves_per_simd), (double)c / mfma_per_wave);
}

int main() {
    bf16x8 *a, *b; float* out; long long* cyc;
    hipMalloc(&a, 64 * 16); hipMalloc(&b, 64 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    // pass 0: constant operands (minimal switching power); pass 1: random bf16 operands in (-1, 1) -- the chip clocks
    // to its power budget, so the SAME loop is slower on random data (MI355X_MICROARCH.md, DVFS give-back)
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) { hipMemset(a, 0x3c, 64 * 16); hipMemset(b, 0x3c, 64 * 16); printf("-- constant operands\n"); }
        else {
            unsigned short h[2][512];
            unsigned s = 12345u;
            for (int k = 0; k < 2; ++k)
                for (int i = 0; i < 512; ++i) {
                    s = s * 1664525u + 1013904223u;
                    const float f = ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
                    unsigned u; __builtin_memcpy(&u, &f, 4);
                    h[k][i] = (unsigned short)(u >> 16);
                }
            hipMemcpy(a, h[0], 1024, hipMemcpyHostToDevice); hipMemcpy(b, h[1], 1024, hipMemcpyHostToDevice);
            printf("-- random operands\n");
        }
        for (int threads : {256, 512}) {
            run<1>(threads, a, b, out, cyc);
            run<2>(threads, a, b, out, cyc);
            run<4>(threads, a, b, out, cyc);
        }
    }
    return 0;
}
