// fp4 (e2m1) as the A operand of v_mfma_scale_f32_32x32x64_f8f6f4 against an fp6 (e2m3) B operand (cbsz = 4, blgp = 2):
//  (1) which nibble of the A operand's 4 registers pairs with which fp6 element of B (same k)
//  (2) decode of the 16 e2m1 codes, scale byte semantics on the fp4 side
//  (3) pipe time of the mixed product against fp6 x fp6
// hipcc --offload-arch=gfx950 -O3 -o bin/mx_fp4_probe mx_fp4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ v8i fp6_onehot(int i0) {           // 32 fp6 codes, 1.0 (0b001000) at element i0, else 0
    unsigned long long bits[3] = {0, 0, 0};
    const int p = i0 * 6 + 3;
    bits[p / 64] |= 1ull << (p % 64);
    v8i b;
    for (int e = 0; e < 6; ++e) b[e] = (int)(bits[e / 2] >> (32 * (e & 1)));
    b[6] = b[7] = 0;
    return b;
}
// test 1: row r has fp4 code 0b0010 (= 1.0 in e2m1: e=1,m=0) at nibble r only; column n has fp6 1.0 at element n only -> D[r][n] != 0 iff nibble r <-> element n
__global__ void k_order(float* D) {
    const int l = threadIdx.x, r = l & 31;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0};
    a[r >> 3] = 2 << (4 * (r & 7));
    v8i b = fp6_onehot(l & 31);
    v16f acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 2, 0, 127, 0, 127);
    for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[q];
}
// test 2: row r (< 16) has code r at nibble 0, B = 1.0 at element 0; rows 16..31: code 2 with scale byte 127 + (r - 16)
__global__ void k_decode(float* D) {
    const int l = threadIdx.x, r = l & 31;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0};
    a[0] = r < 16 ? r : 2;
    v8i b = fp6_onehot(0);
    v16f acc = {0};
    const int sa = r < 16 ? 127 : 127 + (r - 16) - 4;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 2, 0, sa, 0, 127);
    for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[q];
}
template <int MODE> __global__ __launch_bounds__(256, 1) void k_rate(int iters, float* out, long long* cyc) {
    v16f acc[2];
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = (float)threadIdx.x * 1e-3f;
    v8i pa = {(int)threadIdx.x * 7919, 12345, (int)threadIdx.x, 99, 1234567, 7, 0, 0}, pb = {31, (int)threadIdx.x * 31, 5, 77, 9, 1, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[t] = MODE == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa, pb, acc[t], 2, 2, 0, 127, 0, 127)
                                   : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa, pb, acc[t], 4, 2, 0, 127, 0, 127);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    float *D; hipMalloc(&D, 4096); float h[1024];
    k_order<<<1, 64>>>(D); hipMemcpy(h, D, 4096, hipMemcpyDeviceToHost);
    printf("nibble r of A pairs with fp6 element: ");
    for (int r = 0; r < 32; ++r) { int hit = -1, cnt = 0; for (int n = 0; n < 32; ++n) if (h[r * 32 + n] != 0.f) { hit = n; ++cnt; } printf("%d%s ", hit, cnt == 1 ? "" : "?"); }
    printf("\n  D[0][0] = %g (both lane halves matched -> 2)\n", h[0]);
    k_decode<<<1, 64>>>(D); hipMemcpy(h, D, 4096, hipMemcpyDeviceToHost);
    printf("e2m1 codes 0..15 decode to (x2 for the two halves): ");
    for (int r = 0; r < 16; ++r) printf("%g ", h[r * 32] * 0.5f);
    printf("\ncode 2 under scale bytes 123..138: ");
    for (int r = 16; r < 32; ++r) printf("%g ", h[r * 32] * 0.5f);
    printf("\n");
    float* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16);
    for (int rep = 0; rep < 2; ++rep) { k_rate<0><<<1024, 256>>>(2000, out, cyc); k_rate<1><<<1024, 256>>>(2000, out, cyc); }
    long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("cycles per MFMA (8 per iteration, 2 accumulators): fp6 x fp6 %.1f, fp4 x fp6 %.1f\n", c[0] / 16000.0, c[1] / 16000.0);
    return 0;
}
