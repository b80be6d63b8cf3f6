// Companion of atomic_probe.hip: does the memory side execute 64-bit integer atomics at the same rate PER OPERATION as 32-bit float ones
// (then two 32-bit fixed-point channels packed into one 64-bit add would halve the scatter's atomic count), or per dword?
// Every lane adds to one element of a run of consecutive elements at a random aligned position of an 88 MB table.
//     hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_probe64 atomic_probe64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <typename T>
__global__ __launch_bounds__(256) void k_atomic(T* __restrict__ tab, const uint32_t* __restrict__ idx, int run, long nslots, int per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int groups = 64 / run, sub = lane / run, c = lane % run;
    for (int k = 0; k < per_wave; ++k) {
        const long slot = idx[((wave * per_wave + k) * groups + sub) % nslots];
        if constexpr (sizeof(T) == 8) atomicAdd(reinterpret_cast<unsigned long long*>(tab) + slot * run + c, 1ull);
        else if constexpr (__is_same(T, float)) unsafeAtomicAdd(tab + slot * run + c, 1.0f);
        else atomicAdd(tab + slot * run + c, (T)1);
    }
}

template <typename T>
void probe(const char* name, long tab_mb, int run) {
    const long nidx = 1L << 24, nslots = tab_mb * (1L << 20) / sizeof(T) / run;
    std::vector<uint32_t> h(nidx);
    uint64_t s = 88172645463325252ull;
    for (long i = 0; i < nidx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % (uint64_t)nslots); }
    T* tab; uint32_t* idx;
    hipMalloc(&tab, tab_mb << 20); hipMemset(tab, 0, tab_mb << 20);
    hipMalloc(&idx, nidx * 4); hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice);
    const int per_wave = 64, blocks = 8192;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k_atomic<T><<<blocks, 256>>>(tab, idx, run, nidx, per_wave);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double ops = (double)blocks * 256 * per_wave;
    printf("%-6s table %4ld MB  runs of %2d elements (%3d bytes): %8.3f ms  %7.1f G ops/s  %7.1f GB/s of operands\n", name, tab_mb, run, run * (int)sizeof(T), ms,
           ops / (ms * 1e-3) / 1e9, ops * sizeof(T) / (ms * 1e-3) / 1e9);
    hipFree(tab); hipFree(idx);
}

int main() {
    for (int run : {16, 32, 64}) probe<float>("f32", 88, run);
    for (int run : {8, 16, 32, 64}) probe<unsigned long long>("u64", 88, run);
    for (int run : {16, 64}) probe<unsigned int>("u32", 88, run);
    return 0;
}
