// Probe: MODE.FP16_OVFL (bit 23 of HW_REG_MODE): with it set, does a float32 -> float16 conversion that overflows
// saturate to +-65504 instead of producing inf?  (If yes the float16 range guard of the MLP epilogue is free.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out, int set) {
    if (set) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const f32x2 v = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const _Float16 s = (_Float16)in[2 * threadIdx.x];
    out[3 * threadIdx.x] = (float)h[0];
    out[3 * threadIdx.x + 1] = (float)h[1];
    out[3 * threadIdx.x + 2] = (float)s;
}
int main() {
    float h[8] = {1e6f, -1e6f, 65504.f, 70000.f, 1.5f, -3.25f, 65519.f, 65520.f}, o[12];
    float *di, *dout;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        k<<<1, 4>>>(di, dout, set);
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", set);
        for (int i = 0; i < 4; ++i) printf("  (%g, %g | %g)", o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        printf("\n");
    }
    return 0;
}
