// What global float atomics sustain on this chip: the ceiling k_voxel_sample_bwd is compared with (profiles/r02_pmc_scatter.txt).
// Every wavefront instruction adds to RUN consecutive floats (16 = one 64-byte request, 64 = four) at a random aligned position of a
// table of TAB megabytes; the counters say every such request travels to the memory side (TCC_EA0_ATOMIC == TCC_ATOMIC): device-scope
// float atomics are not executed in the XCD's L2.  Table sizes: 2 MB, 88 MB (the fine 64-channel plane's gradient), 1 GB.
//     hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_probe atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void k_atomic(float* __restrict__ tab, const uint32_t* __restrict__ idx, int run, long nslots, int per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int groups = 64 / run, sub = lane / run, c = lane % run;       // `groups` independent runs per wavefront instruction
    for (int k = 0; k < per_wave; ++k) {
        const long slot = idx[((wave * per_wave + k) * groups + sub) % nslots];
        unsafeAtomicAdd(tab + slot * run + c, 1.0f);
    }
}

int main() {
    const long nidx = 1L << 24;
    for (long tab_mb : {2L, 88L, 1024L}) {
        for (int run : {16, 64}) {
            const long nslots = tab_mb * (1L << 20) / 4 / run;
            std::vector<uint32_t> h(nidx);
            uint64_t s = 88172645463325252ull;
            for (long i = 0; i < nidx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % (uint64_t)nslots); }
            float* tab; uint32_t* idx;
            hipMalloc(&tab, tab_mb << 20); hipMemset(tab, 0, tab_mb << 20);
            hipMalloc(&idx, nidx * 4); hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice);
            const int per_wave = 64, blocks = 8192;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                k_atomic<<<blocks, 256>>>(tab, idx, run, nidx, per_wave);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double adds = (double)blocks * 256 * per_wave, req = adds / 16;
            printf("table %5ld MB  runs of %2d floats: %8.3f ms  %7.1f G float adds/s  %6.2f G 64-byte requests/s\n", tab_mb, run, ms, adds / (ms * 1e-3) / 1e9, req / (ms * 1e-3) / 1e9);
            hipFree(tab); hipFree(idx);
        }
    }
    return 0;
}
