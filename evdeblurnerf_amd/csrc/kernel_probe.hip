// Measurement aid behind evd_probe_mfma_rate(): the rate at which THIS chip sustains back-to-back
// v_mfma_f32_32x32x16_bf16 issue (two wavefronts per SIMD, four independent accumulators each, every SIMD of every CU
// busy, nothing else in the loop).  MI355X clocks to its power budget, so the sustained rate depends on the operand
// data: constant operands run near the 2.4 GHz nominal peak, random operands ~20 % lower.  bench.py quotes both next to
// the nominal 2.5 PFLOP/s so that the fused MLP kernel (which multiplies random-looking data) can be read against what
// the matrix pipe can actually deliver.  (tools/probes/mfma_probe.hip is the stand-alone version.)
#include "evd_common.h"
#include "mlp_device.h"

namespace evd {

__global__ __launch_bounds__(512) void k_probe_mfma(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    const bf16x8 av = a[lane], bv = b[lane];
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace evd

using namespace evd;

extern "C" int evd_probe_mfma_rate(int random_operands, int iters, double* tflops, void* stream) {
    EVD_REQUIRE(tflops && iters > 0, "evd_probe_mfma_rate: bad arguments");
    hipStream_t st = as_stream(stream);
    int dev = 0, cus = 0;
    EVD_HIP(hipGetDevice(&dev));
    EVD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    uint16_t h[2][512];
    uint32_t s = 12345u;
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 512; ++i) {
            s = s * 1664525u + 1013904223u;
            const float f = random_operands ? ((s >> 8) & 0xffff) / 32768.0f - 1.0f : 1.0f;
            uint32_t u;
            memcpy(&u, &f, 4);
            h[k][i] = (uint16_t)(u >> 16);
        }
    DevBuf a, b, out;
    int rc = a.upload(h[0], 1024);
    if (!rc) rc = b.upload(h[1], 1024);
    if (!rc) rc = out.alloc((size_t)cus * 512 * 4);
    if (rc) { a.release(); b.release(); out.release(); return rc; }
    hipEvent_t e0, e1;
    EVD_HIP(hipEventCreate(&e0));
    EVD_HIP(hipEventCreate(&e1));
    k_probe_mfma<<<cus, 512, 0, st>>>((const bf16x8*)a.p, (const bf16x8*)b.p, (float*)out.p, iters / 8 + 1);   // warm-up / clock ramp
    EVD_HIP(hipEventRecord(e0, st));
    k_probe_mfma<<<cus, 512, 0, st>>>((const bf16x8*)a.p, (const bf16x8*)b.p, (float*)out.p, iters);
    EVD_HIP(hipEventRecord(e1, st));
    EVD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    EVD_HIP(hipEventElapsedTime(&ms, e0, e1));
    *tflops = 2.0 * 32 * 32 * 16 * (double)iters * 32 * 8 * cus / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    a.release(); b.release(); out.release();
    return EVD_OK;
}
