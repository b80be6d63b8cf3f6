// What a random gather of short records sustains on this chip: the ceiling k_voxel_sample is compared with.
// table of N records of REC bytes (16-byte units), every lane group of REC/16 lanes reads one random record with 16-byte loads,
// K independent records per lane in flight.  Table sizes: L2-resident (2 MB), Infinity-Cache-resident (44 MB: the fine 64-channel
// float16 plane), HBM (1 GB).     hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K> __global__ __launch_bounds__(256) void k_gather(const f32x4* __restrict__ tab, const uint32_t* __restrict__ idx, int lanes_per_rec, long nrec_reads,
                                                                 float* __restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long grp = t / lanes_per_rec;
    const int sub = (int)(t % lanes_per_rec);
    f32x4 acc = {0, 0, 0, 0};
    for (long r0 = grp * K; r0 < nrec_reads; r0 += (long)gridDim.x * 256 / lanes_per_rec * K) {
        f32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = tab[(long)idx[(r0 + k) % nrec_reads] * lanes_per_rec + sub];
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[t] = acc[0];
}

int main() {
    const long nreads = 1L << 24;
    for (long tab_mb : {2L, 16L, 44L, 1024L}) {
        for (int rec : {32, 128, 512}) {
            const long nrec = tab_mb * (1L << 20) / rec;
            std::vector<uint32_t> h(nreads);
            uint64_t s = 88172645463325252ull;
            for (long i = 0; i < nreads; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % (uint64_t)nrec); }
            f32x4* tab; uint32_t* idx; float* out;
            hipMalloc(&tab, tab_mb << 20); hipMemset(tab, 0, tab_mb << 20);
            hipMalloc(&idx, nreads * 4); hipMemcpy(idx, h.data(), nreads * 4, hipMemcpyHostToDevice);
            hipMalloc(&out, (1 << 22) * 4);
            const int lpr = rec / 16;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                k_gather<6><<<8192, 256>>>(tab, idx, lpr, nreads, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("table %5ld MB  record %4d B: %8.3f ms  %8.1f GB/s useful  %7.2f G records/s\n", tab_mb, rec, ms, nreads * (double)rec / (ms * 1e-3) / 1e9, nreads / (ms * 1e-3) / 1e9);
            hipFree(tab); hipFree(idx); hipFree(out);
        }
    }
    return 0;
}
