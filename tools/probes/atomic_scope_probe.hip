// Probe: are float atomics that the XCD's own L2 may execute (workgroup scope: no sc1 bit) faster than the device-scope ones the scatter
// uses (executed at the memory side: 20 G 64-byte requests/s whatever the table size, tools/probes/atomic_probe.hip)?
// Every wavefront instruction adds to 16 consecutive floats (one 64-byte request per 16 lanes, 4 independent requests per instruction)
// at random aligned positions.
//   scope:  agent (unsafeAtomicAdd, what the library issues)  |  workgroup (__hip_atomic_fetch_add ... __HIP_MEMORY_SCOPE_WORKGROUP)
//   table:  2 MB, 16 MB, 88 MB shared by all XCDs  |  "slab": 8 x 2 MB, a workgroup only touches the slab of the XCD it runs on
//           (HW_REG_XCC_ID) -- the case in which an L2-executed atomic is also CORRECT (no line is dirty in two L2s)
// The sums of the slab form are checked against the number of adds issued.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_scope_probe.hip -o bin/atomic_scope_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}

template <int SCOPE, bool SLAB>
__global__ __launch_bounds__(256) void k_atomic(float* __restrict__ tab, const uint32_t* __restrict__ idx, long nslots, long nidx, int per_wave,
                                                unsigned* __restrict__ xcc_seen) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int sub = lane >> 4, c = lane & 15;
    const unsigned x = xcc_id() & 7u;
    if (threadIdx.x == 0 && xcc_seen) atomicAdd(xcc_seen + x, 1u);
    float* base = SLAB ? tab + (long)x * nslots * 16 : tab;          // SLAB: nslots = slots of ONE slab
    for (int k = 0; k < per_wave; ++k) {
        const long slot = idx[((wave * per_wave + k) * 4 + sub) % nidx] % (uint32_t)nslots;
        float* p = base + slot * 16 + c;
        if (SCOPE == 0) unsafeAtomicAdd(p, 1.0f);
        else (void)__hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
    unsigned* seen; hipMalloc(&seen, 64); double* dsum; hipMalloc(&dsum, 8);
    const int per_wave = 64, blocks = 8192;
    const double adds = (double)blocks * 4 * per_wave * 64, reqs = adds / 16;
    auto run = [&](auto kern, const char* name, long tab_bytes, long nslots, bool check) {
        float* tab; hipMalloc(&tab, tab_bytes); hipMemset(tab, 0, tab_bytes); hipMemset(seen, 0, 64); hipMemset(dsum, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        int reps = 3;
        for (int rep = 0; rep < reps; ++rep) {
            hipEventRecord(e0);
            kern<<<blocks, 256>>>(tab, idx, nslots, nidx, per_wave, rep == 0 ? seen : nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  %-58s %8.3f ms  %7.1f G requests/s", name, ms, reqs / ms / 1e6);
        if (check) {
            k_sum<<<256, 256>>>(tab, tab_bytes / 4, dsum);
            double got; hipMemcpy(&got, dsum, 8, hipMemcpyDeviceToHost);
            printf("   sum %.0f of %.0f%s", got, adds * reps, got == adds * reps ? " (exact)" : "  <-- LOST UPDATES");
        }
        unsigned sh[8]; hipMemcpy(sh, seen, 32, hipMemcpyDeviceToHost);
        printf("\n");
        hipFree(tab);
        return sh[0];
    };
    {
        unsigned sh[8]; hipMemset(seen, 0, 64);
        float* t; hipMalloc(&t, 1 << 21);
        k_atomic<0, false><<<blocks, 256>>>(t, idx, (1 << 21) / 64, nidx, 1, seen);
        hipMemcpy(sh, seen, 32, hipMemcpyDeviceToHost);
        printf("workgroups per XCC_ID of a %d-workgroup launch:", blocks);
        for (int i = 0; i < 8; ++i) printf(" %u", sh[i]);
        printf("\n");
        hipFree(t);
    }
    for (long mb : {2L, 16L, 88L}) {
        const long bytes = mb << 20, nslots = bytes / 64;
        char nm[96];
        snprintf(nm, sizeof nm, "agent scope, %ld MB table shared by all XCDs", mb); run(k_atomic<0, false>, nm, bytes, nslots, true);
        snprintf(nm, sizeof nm, "workgroup scope, %ld MB table shared by all XCDs", mb); run(k_atomic<1, false>, nm, bytes, nslots, true);
    }
    for (long mb : {1L, 2L, 4L}) {
        const long slab = mb << 20, nslots = slab / 64;
        char nm[96];
        snprintf(nm, sizeof nm, "agent scope, 8 slabs of %ld MB, each XCD its own", mb); run(k_atomic<0, true>, nm, 8 * slab, nslots, true);
        snprintf(nm, sizeof nm, "workgroup scope, 8 slabs of %ld MB, each XCD its own", mb); run(k_atomic<1, true>, nm, 8 * slab, nslots, true);
    }
    return 0;
}
