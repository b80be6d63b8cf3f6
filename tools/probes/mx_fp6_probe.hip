// Probe of the block-scaled MX path used by the compensated float16 mode:
//   v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_fp6_f16  (32 values -> 6 VGPRs of fp6 e2m3, scale semantics)
//   v_mfma_scale_f32_32x32x64_f8f6f4  (k mapping of the 32 values a lane holds, e8m0 scale bytes)
// and of the issue rates (f16 32x32x16 alone, fp6 32x32x64 alone, 2:1 interleaved; conversion cost).
// Dumps raw results to a binary file analysed by tools/probes/mx_fp6_probe.py.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// A [32 rows][64 k] row-major, B [64 k][32 cols] row-major; sa / sb per (row|col, half) biased exponents
__global__ void k_sem(const float* A, const float* B, const int* ea, const int* eb, int* pa_out, int* pb_out, float* D) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    v16f a0, a1, b0, b1;
    for (int e = 0; e < 16; ++e) {
        a0[e] = A[i * 64 + 32 * h + e];
        a1[e] = A[i * 64 + 32 * h + 16 + e];
        b0[e] = B[(32 * h + e) * 32 + i];
        b1[e] = B[(32 * h + 16 + e) * 32 + i];
    }
    const int sa = ea[l], sb = eb[l];
    const float fa = __int_as_float(sa << 23), fb = __int_as_float(sb << 23);
    v6i pa = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, fa);
    v32h hb;
    for (int e = 0; e < 16; ++e) { hb[e] = (_Float16)b0[e]; hb[16 + e] = (_Float16)b1[e]; }
    v6i pb = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hb, fb);
    for (int e = 0; e < 6; ++e) { pa_out[l * 6 + e] = pa[e]; pb_out[l * 6 + e] = pb[e]; }
    v8i pa8 = {pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], 0, 0}, pb8 = {pb[0], pb[1], pb[2], pb[3], pb[4], pb[5], 0, 0};
    v16f acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa8, pb8, acc, 2, 2, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

// rates: MODE 0 = f16 32x32x16 only, 1 = fp6 32x32x64 only, 2 = 2 f16 : 1 fp6 (the compensated mode's mix), 3 = mode 2 + one
// pk32 conversion per 4 MFMAs
template <int MODE> __global__ __launch_bounds__(256, 1) void k_rate(int iters, float* out, long long* cyc) {
    v16f acc[4];
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = (float)threadIdx.x;
    v8h ha, hb;
    for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(threadIdx.x * 0.001f + e); hb[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
    v8i pa = {(int)threadIdx.x * 7919, 12345, (int)threadIdx.x, 99, 1234567, 7, 0, 0}, pb = {31, (int)threadIdx.x * 31, 5, 77, 9, 1, 0, 0};
    v32h cv;
    for (int e = 0; e < 32; ++e) cv[e] = (_Float16)(e + threadIdx.x);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 0 || MODE >= 2) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[t], 0, 0, 0);
            }
            if (MODE >= 1) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa, pb, acc[t], 2, 2, 0, 127, 0, 127);
        }
        if (MODE == 3) {
            v6i q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(cv, 1.0f);
            pa[0] ^= q[0] & 1; cv[0] += (_Float16)1;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void rate(const char* name, double flop_per_iter) {
    float* o; long long* c;
    hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&c, 8);
    const int iters = 20000;
    k_rate<MODE><<<1024, 256>>>(10, o, c);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_rate<MODE><<<1024, 256>>>(iters, o, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    // 1024 blocks x 4 waves on 256 CUs x 4 SIMDs = one wave per SIMD
    printf("%-44s %8.3f ms  %9.1f TFLOP/s  %7.1f cycles/iter (wave 0)\n", name, ms, 1024.0 * 4 * iters * flop_per_iter / (ms * 1e-3) / 1e12, (double)cy / iters);
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "mx_fp6_probe.bin";
    std::vector<float> A(32 * 64), B(64 * 32);
    std::vector<int> ea(64), eb(64);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i * 64 + k] = rnd() * (1 << (i % 5)) * ((k % 7 == 0) ? 0.01f : 1.f);
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = rnd() * (0.25f * (1 + j % 3)) * ((k % 5 == 0) ? 0.02f : 1.f);
    for (int l = 0; l < 64; ++l) { ea[l] = 127 + (l & 31) % 5 - 2; eb[l] = 127 - 4 + ((l & 31) % 3 == 2 ? 1 : 0); }
    float *dA, *dB, *dD; int *dea, *deb, *dpa, *dpb;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 32 * 32 * 4);
    hipMalloc(&dea, 256); hipMalloc(&deb, 256); hipMalloc(&dpa, 64 * 6 * 4); hipMalloc(&dpb, 64 * 6 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dea, ea.data(), 256, hipMemcpyHostToDevice); hipMemcpy(deb, eb.data(), 256, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(dA, dB, dea, deb, dpa, dpb, dD);
    std::vector<float> D(32 * 32); std::vector<int> pa(64 * 6), pb(64 * 6);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(pa.data(), dpa, pa.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(pb.data(), dpb, pb.size() * 4, hipMemcpyDeviceToHost);
    FILE* f = fopen(path, "wb");
    fwrite(A.data(), 4, A.size(), f); fwrite(B.data(), 4, B.size(), f); fwrite(ea.data(), 4, 64, f); fwrite(eb.data(), 4, 64, f);
    fwrite(pa.data(), 4, pa.size(), f); fwrite(pb.data(), 4, pb.size(), f); fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("wrote %s\n", path);
    rate<0>("f16 32x32x16 (8 per iter)", 8 * 2.0 * 32 * 32 * 16 * 64 / 64);
    rate<1>("fp6 32x32x64 scaled (4 per iter)", 4 * 2.0 * 32 * 32 * 64);
    rate<2>("8 f16 + 4 fp6 per iter (flops of the f16 only)", 8 * 2.0 * 32 * 32 * 16);
    rate<3>("... + 1 pk32 fp6 conversion per iter", 8 * 2.0 * 32 * 32 * 16);
    return 0;
}
