// Probe: what HBM rate does the compositing scan's TRAFFIC MIX reach without its arithmetic?  Per sample 16 B (raw) + 4 B (z) read and
// 4 B (weights) written, 2^20 rays x 128 samples = 3.22 GB per launch, as k_composite_rows moves it (one wavefront per RPW rays):
//   pattern 0 "consecutive": lane l owns samples 2l, 2l+1 of a ray (two 16-byte loads 32 B apart per lane, z as two dwords, w as one 8-byte store)
//   pattern 1 "interleaved": lane l owns samples l, 64 + l (every load / store instruction of the wavefront covers one contiguous run)
// each with plain and non-temporal accesses, RPW = 1, 2, 4 rays per wavefront; plus a read-only pass and a float4 copy as yardsticks.
// hipcc --offload-arch=gfx950 -O3 hbm_mix_probe.hip -o bin/hbm_mix_probe && bin/hbm_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool NT, typename T> __device__ __forceinline__ T ld(const T* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT, typename T> __device__ __forceinline__ void st(T* p, T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int PATTERN, int RPW, bool NT, bool STORE>
__global__ __launch_bounds__(256) void k_mix(const float* __restrict__ raw, const float* __restrict__ z, float* __restrict__ w, float* __restrict__ sink, long R) {
    constexpr int S = 128;
    const int lane = threadIdx.x & 63;
    const long r0 = (blockIdx.x * 4L + (threadIdx.x >> 6)) * RPW;
    if (r0 >= R) return;
    f4 v[RPW][2];
    float zz[RPW][2];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const f4* rw = reinterpret_cast<const f4*>(raw + (r0 + q) * S * 4);
        const float* zp = z + (r0 + q) * S;
        if (PATTERN == 0) {
            v[q][0] = ld<NT>(rw + 2 * lane); v[q][1] = ld<NT>(rw + 2 * lane + 1);
            const f2 t = ld<NT>(reinterpret_cast<const f2*>(zp) + lane); zz[q][0] = t.x; zz[q][1] = t.y;
        } else {
            v[q][0] = ld<NT>(rw + lane); v[q][1] = ld<NT>(rw + 64 + lane);
            zz[q][0] = ld<NT>(zp + lane); zz[q][1] = ld<NT>(zp + 64 + lane);
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const float a = v[q][0].x + v[q][0].y + v[q][0].z + v[q][0].w + zz[q][0], b = v[q][1].x + v[q][1].y + v[q][1].z + v[q][1].w + zz[q][1];
        if (STORE) {
            float* wp = w + (r0 + q) * S;
            if (PATTERN == 0) st<NT>(reinterpret_cast<f2*>(wp) + lane, f2{a, b});
            else { st<NT>(wp + lane, a); st<NT>(wp + 64 + lane, b); }
        } else acc += a + b;
    }
    if (!STORE && acc == 123.456f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ b, long n) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) b[i] = a[i];
}

template <typename F> static double timeit(F f, int iters = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main() {
    const long R = 1 << 20, S = 128;
    float *raw, *z, *w, *sink;
    hipMalloc(&raw, R * S * 16); hipMalloc(&z, R * S * 4); hipMalloc(&w, R * S * 4); hipMalloc(&sink, 64);
    hipMemset(raw, 0x3c, R * S * 16); hipMemset(z, 0x3c, R * S * 4);
    const double rd = (double)R * S * 20, wr = (double)R * S * 4;
#define RUN(P, RPW, NT, STORE) { double ms = timeit([&] { k_mix<P, RPW, NT, STORE><<<(unsigned)(R / (4 * RPW)), 256>>>(raw, z, w, sink, R); }); \
    const double b = rd + (STORE ? wr : 0); \
    printf("%-12s rpw %d %-4s %-10s %.3f ms  %5.2f TB/s = %.3f of 8 TB/s\n", P ? "interleaved" : "consecutive", RPW, NT ? "nt" : "", STORE ? "read+write" : "read only", ms, b / ms / 1e9, b / ms / 1e9 / 8.0); }
    RUN(0, 1, false, true); RUN(0, 2, false, true); RUN(0, 4, false, true);
    RUN(1, 1, false, true); RUN(1, 2, false, true); RUN(1, 4, false, true);
    RUN(0, 2, true, true); RUN(1, 2, true, true); RUN(1, 4, true, true);
    RUN(0, 2, false, false); RUN(1, 2, false, false); RUN(1, 2, true, false); RUN(1, 4, false, false);
    { const long n = R * S; double ms = timeit([&] { k_copy<<<(unsigned)(n / 256), 256>>>((const f4*)raw, (f4*)raw + n / 2 * 0 + 0, 0); });  (void)ms; }
    { const long n = R * S / 2; f4* a = (f4*)raw; f4* b = a + n; double ms = timeit([&] { k_copy<<<(unsigned)(n / 256), 256>>>(a, b, n); });
      printf("float4 copy of %.2f GB (r + w)            %.3f ms  %5.2f TB/s\n", n * 32 / 1e9, ms, n * 32.0 / ms / 1e9); }
    return 0;
}
