// v_fma_mixlo_f16 / mixhi_f16 residual pairs and v_cvt_scalef32_pk32_fp6_f16 on them: what mlp_pipe_c.h c_drain_pair / c_finish rely on
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
__global__ void k(const float* x, unsigned* out_w, unsigned* out_r, int* out_q, float scale) {
    const int l = threadIdx.x;
    float x0 = x[2 * l], x1 = x[2 * l + 1];
    unsigned w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    unsigned rw;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=&v"(rw) : "v"(w), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(rw) : "v"(w), "v"(x1));
    out_w[l] = w; out_r[l] = rw;
    u32x16 src;
    for (int e = 0; e < 16; ++e) src[e] = rw;
    i32x6 q;
    asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(q) : "v"(src), "v"(scale));
    for (int e = 0; e < 6; ++e) out_q[l * 6 + e] = q[e];
}
static float h2f(unsigned short h) { _Float16 v; __builtin_memcpy(&v, &h, 2); return (float)v; }
int main() {
    float hx[128]; for (int i = 0; i < 128; ++i) hx[i] = (i % 2 ? -1.f : 1.f) * (0.37f + 0.013f * i) * (i < 64 ? 1.f : 0.01f);
    float* dx; unsigned *dw, *dr; int* dq;
    hipMalloc(&dx, 512); hipMalloc(&dw, 256); hipMalloc(&dr, 256); hipMalloc(&dq, 64 * 24);
    hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dx, dw, dr, dq, ldexpf(1.f, -13));
    unsigned w[64], r[64]; int q[64 * 6];
    hipMemcpy(w, dw, 256, hipMemcpyDeviceToHost); hipMemcpy(r, dr, 256, hipMemcpyDeviceToHost); hipMemcpy(q, dq, 64 * 24, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 33, 40}) {
        float f0 = h2f(w[l] & 0xffff), f1 = h2f(w[l] >> 16), r0 = h2f(r[l] & 0xffff), r1 = h2f(r[l] >> 16);
        printf("lane %d: x = %.7g %.7g | f16 = %.7g %.7g | residual f16 = %.4g %.4g (exact %.4g %.4g) | fp6 word0 = %08x (codes %d %d)\n", l, hx[2 * l], hx[2 * l + 1], f0, f1, r0, r1,
               hx[2 * l] - f0, hx[2 * l + 1] - f1, q[l * 6], q[l * 6] & 63, (q[l * 6] >> 6) & 63);
    }
    return 0;
}
