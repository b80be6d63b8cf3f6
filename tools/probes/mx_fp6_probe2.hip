// Second probe of the MX path for the compensated float16 mode (f16c):
//  (1) layout + rounding of v_cvt_scalef32_2xpk16_fp6_f32 with an early-clobber destination (inline asm) and through the builtin
//  (2) scale byte selection (op_sel) of v_mfma_scale_f32_32x32x64_f8f6f4
//  (3) issue cost of the conversions, v_fma_mix_f32 and the 4 f16 : 2 fp6 MFMA mix with two accumulators
// hipcc --offload-arch=gfx950 -O3 -o mx_fp6_probe2 mx_fp6_probe2.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__global__ void k_cvt(const float* src, float scale, int* out_asm, int* out_bi, int* out_h) {
    const int l = threadIdx.x;
    v16f a0, a1;
    for (int e = 0; e < 16; ++e) { a0[e] = src[l * 32 + e]; a1[e] = src[l * 32 + 16 + e]; }
    v6i q;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(q) : "v"(a0), "v"(a1), "v"(scale));
    for (int e = 0; e < 6; ++e) out_asm[l * 6 + e] = q[e];
    v6i q2 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, scale);
    for (int e = 0; e < 6; ++e) out_bi[l * 6 + e] = q2[e];
    v32h hb;
    for (int e = 0; e < 16; ++e) { hb[e] = (_Float16)a0[e]; hb[16 + e] = (_Float16)a1[e]; }
    v6i q3 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hb, scale);
    for (int e = 0; e < 6; ++e) out_h[l * 6 + e] = q3[e];
}

// D = A B with all-ones fp6 operands (code 0b001000 = 1.0), scales taken from byte `sel` of per-lane words
template <int SA, int SB> __global__ void k_sel(const int* wa, const int* wb, float* D) {
    const int l = threadIdx.x;
    v8i a, b;
    // 32 codes of 1.0 (e=1,m=0 -> 0b001000 = 8): 6-bit stream
    unsigned long long bits[3] = {0, 0, 0};
    for (int i = 0; i < 32; ++i) { const int p = i * 6 + 3; bits[p / 64] |= 1ull << (p % 64); }
    for (int e = 0; e < 6; ++e) { a[e] = (int)(bits[e / 2] >> (32 * (e & 1))); b[e] = a[e]; }
    a[6] = a[7] = b[6] = b[7] = 0;
    v16f acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, SA, wa[l], SB, wb[l]);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

template <int MODE> __global__ __launch_bounds__(256, 1) void k_rate(int iters, float* out, long long* cyc) {
    v16f acc[2];
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = (float)threadIdx.x * 1e-3f;
    v8h ha, hb;
    for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(threadIdx.x * 0.001f + e); hb[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
    v8i pa = {(int)threadIdx.x * 7919, 12345, (int)threadIdx.x, 99, 1234567, 7, 0, 0}, pb = {31, (int)threadIdx.x * 31, 5, 77, 9, 1, 0, 0};
    v32h cv;
    for (int e = 0; e < 32; ++e) cv[e] = (_Float16)(e + threadIdx.x);
    v16f r0, r1;
    for (int e = 0; e < 16; ++e) { r0[e] = e * 0.37f + threadIdx.x; r1[e] = e * 0.11f - threadIdx.x; }
    float sc = 1.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 4 f16 + 2 fp6, two accumulators alternating (the f16c mix of one tile pair x k64 block ... x2)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[t], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa, pb, acc[t], 2, 2, 0, 127, 0, 127);
        } else if (MODE == 1) {   // fp6 only, two accumulators
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa, pb, acc[t], 2, 2, 0, 127, 0, 127);
        } else if (MODE == 2) {   // 8 x pk32 f16 -> fp6
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v6i q;
                asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(q) : "v"(cv), "v"(sc));
                asm volatile("" :: "v"(q));
            }
        } else if (MODE == 3) {   // 8 x 2xpk16 f32 -> fp6
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v6i q;
                asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(q) : "v"(r0), "v"(r1), "v"(sc));
                asm volatile("" :: "v"(q));
            }
        } else if (MODE == 4) {   // 64 x v_fma_mix_f32 (independent)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float y;
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(y) : "v"(pa[0]), "v"(r0[e]));
                    asm volatile("" :: "v"(y));
                }
        } else if (MODE == 5) {   // 64 x v_max_f32
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float y;
                    asm volatile("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(r0[e]));
                    asm volatile("" :: "v"(y));
                }
        } else if (MODE == 6) {   // mix of MODE 0 with the VALU load of the f16c drain between the MFMAs: 8 VALU per f16 MFMA slot
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[t], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        float y;
                        asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(y) : "v"(pa[0]), "v"(r0[(j * 2 + t + e) & 15]));
                        asm volatile("" :: "v"(y));
                    }
                }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa, pb, acc[t], 2, 2, 0, 127, 0, 127);
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        float y;
                        asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(y) : "v"(pa[0]), "v"(r1[(c * 2 + t + e) & 15]));
                        asm volatile("" :: "v"(y));
                    }
                }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void rate(const char* name) {
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
    printf("%-64s %8.3f ms  %8.1f shader-clock cycles/iter (wave 0)  %.1f ns/iter\n", name, ms, (double)cy / iters, ms * 1e6 / iters);
}

static double e2m3(int code) {
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const double v = e == 0 ? m / 8.0 : (1 + m / 8.0) * std::pow(2.0, e - 1);
    return s ? -v : v;
}
static void unpack(const int* w, int* codes) {      // 6 dwords -> 32 codes, little-endian bit stream
    for (int i = 0; i < 32; ++i) {
        int c = 0;
        for (int b = 0; b < 6; ++b) { const int p = i * 6 + b; c |= ((w[p / 32] >> (p % 32)) & 1) << b; }
        codes[i] = c;
    }
}

int main() {
    std::vector<float> src(64 * 32);
    srand(3);
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 32; ++e) src[l * 32 + e] = (l < 32 ? (e + 1) * 0.125f * (e & 1 ? -1 : 1) : ((float)rand() / RAND_MAX * 16.f - 8.f));
    float* dsrc; int *da, *db, *dh;
    hipMalloc(&dsrc, src.size() * 4); hipMalloc(&da, 64 * 6 * 4); hipMalloc(&db, 64 * 6 * 4); hipMalloc(&dh, 64 * 6 * 4);
    hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(dsrc, 1.0f, da, db, dh);
    std::vector<int> ha(64 * 6), hb(64 * 6), hh(64 * 6);
    hipMemcpy(ha.data(), da, ha.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), db, hb.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hh.data(), dh, hh.size() * 4, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 40}) {
        int ca[32], cb[32], ch[32];
        unpack(&ha[l * 6], ca); unpack(&hb[l * 6], cb); unpack(&hh[l * 6], ch);
        printf("lane %d\n  src     :", l); for (int e = 0; e < 32; ++e) printf(" %6.3f", src[l * 32 + e]);
        printf("\n  asm &v  :"); for (int e = 0; e < 32; ++e) printf(" %6.3f", e2m3(ca[e]));
        printf("\n  builtin :"); for (int e = 0; e < 32; ++e) printf(" %6.3f", e2m3(cb[e]));
        printf("\n  pk32 f16:"); for (int e = 0; e < 32; ++e) printf(" %6.3f", e2m3(ch[e]));
        printf("\n");
    }
    // scale select: per-lane words with bytes (127, 128, 129, 130) for A (+lane%2), (127, 126, 125, 124) for B
    std::vector<int> wa(64), wb(64);
    for (int l = 0; l < 64; ++l) { wa[l] = 127 | (128 << 8) | (129 << 16) | (130 << 24); wb[l] = 127 | (126 << 8) | (125 << 16) | (124 << 24); }
    int *dwa, *dwb; float* dD;
    hipMalloc(&dwa, 256); hipMalloc(&dwb, 256); hipMalloc(&dD, 4096);
    hipMemcpy(dwa, wa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dwb, wb.data(), 256, hipMemcpyHostToDevice);
    std::vector<float> D(1024);
#define SEL(SA, SB) k_sel<SA, SB><<<1, 64>>>(dwa, dwb, dD); hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost); \
    printf("op_sel a=%d b=%d: D[0][0] = %g (64 x 2^(sa-127) 2^(sb-127); byte k of A = 2^k, of B = 2^-k)\n", SA, SB, D[0]);
    SEL(0, 0) SEL(1, 0) SEL(2, 0) SEL(3, 0) SEL(0, 1) SEL(0, 2) SEL(0, 3) SEL(2, 3)
    rate<0>("8 f16 + 4 fp6 MFMA (two accumulators)");
    rate<1>("8 fp6 MFMA (two accumulators)");
    rate<2>("8 x v_cvt_scalef32_pk32_fp6_f16");
    rate<3>("8 x v_cvt_scalef32_2xpk16_fp6_f32");
    rate<4>("64 x v_fma_mix_f32");
    rate<5>("64 x v_max_f32");
    rate<6>("8 f16 + 4 fp6 MFMA with 6 VALU behind each");
    return 0;
}
