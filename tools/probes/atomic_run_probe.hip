// Probe (round 6): what does a float-atomic REQUEST cost as a function of the length and alignment of the run of
// consecutive floats one wavefront instruction adds to?  tools/probes/atomic_scope_probe.hip only issued 64-byte runs (16 floats per
// quarter wavefront: 20.5 G runs/s whatever the scope / table / XCD locality).  If the memory side retires a 128-byte aligned run at the
// rate of a 64-byte one, the two x-adjacent taps of a 16-channel x-z / y-z plane (8 of the scatter's 10.4 requests per sample) could go
// as one request; if the rate is per LANE (or per 64 bytes), nothing is to be had from run length.
//   RUN floats per group of lanes (8, 16, 32, 64): one wavefront instruction = 64 / RUN independent runs at random positions
//   aligned: run start a multiple of RUN floats | unaligned: start = multiple of 16 floats (64 B), i.e. a 128-byte run straddles two
//   128-byte lines half of the time (the tap pair of an odd cell)
// Reported: G runs/s, G lane-adds/s, GB/s of run bytes.  The sums are checked.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_run_probe.hip -o bin/atomic_run_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int RUN, bool ALIGNED>
__global__ __launch_bounds__(256) void k_atomic(float* __restrict__ tab, const uint32_t* __restrict__ idx, long nfloats, long nidx, int per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    constexpr int GROUPS = 64 / RUN;
    const int sub = lane / RUN, c = lane % RUN;
    constexpr int GRAN = ALIGNED ? RUN : (RUN < 16 ? RUN : 16);         // start granularity in floats
    const uint32_t nstart = (uint32_t)((nfloats - RUN) / GRAN);
    for (int k = 0; k < per_wave; ++k) {
        const long start = (long)(idx[((wave * per_wave + k) * GROUPS + sub) % nidx] % nstart) * GRAN;
        unsafeAtomicAdd(tab + start + c, 1.0f);
    }
}

__global__ void k_sum(const float* __restrict__ t, long n, double* out) {
    double s = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) s += t[i];
    atomicAdd(out, s);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const long nidx = 1L << 24;
    std::vector<uint32_t> h(nidx);
    uint64_t s = 88172645463325252ull;
    for (long i = 0; i < nidx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 16); }
    uint32_t* idx; hipMalloc(&idx, nidx * 4); hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice);
    double* dsum; hipMalloc(&dsum, 8);
    const int per_wave = 64, blocks = 8192;
    const long tab_bytes = 88L << 20, nfloats = tab_bytes / 4;
    float* tab; hipMalloc(&tab, tab_bytes);
    auto run = [&](auto kern, const char* name, int RUN) {
        hipMemset(tab, 0, tab_bytes); hipMemset(dsum, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        const int reps = 3;
        for (int rep = 0; rep < reps; ++rep) {
            hipEventRecord(e0);
            kern<<<blocks, 256>>>(tab, idx, nfloats, nidx, per_wave);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double adds = (double)blocks * 4 * per_wave * 64, runs = adds / RUN;
        k_sum<<<256, 256>>>(tab, nfloats, dsum);
        double got; hipMemcpy(&got, dsum, 8, hipMemcpyDeviceToHost);
        printf("  %-46s %8.3f ms  %7.1f G runs/s  %7.1f G lane-adds/s  %7.1f GB/s   %s\n", name, ms, runs / ms / 1e6, adds / ms / 1e6, adds * 4 / ms / 1e6,
               got == adds * reps ? "sum exact" : "<-- LOST UPDATES");
    };
    printf("float atomics (agent scope, 88 MB table), one wavefront instruction = 64 / RUN runs of RUN consecutive floats:\n");
    run(k_atomic<8, true>, "run  8 floats ( 32 B), aligned", 8);
    run(k_atomic<16, true>, "run 16 floats ( 64 B), aligned", 16);
    run(k_atomic<32, true>, "run 32 floats (128 B), aligned to 128 B", 32);
    run(k_atomic<32, false>, "run 32 floats (128 B), start on any 64 B", 32);
    run(k_atomic<64, true>, "run 64 floats (256 B), aligned to 256 B", 64);
    run(k_atomic<64, false>, "run 64 floats (256 B), start on any 64 B", 64);
    return 0;
}
