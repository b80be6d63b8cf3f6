// Probe: what rate does the TRAFFIC of k_wgrad_dgrad (nerf_train_kernel.h) reach without its arithmetic?  The kernel reads, per 32-sample
// tile, 16 KiB of gradient fragments + 16 KiB of activation fragments by LDS-DMA (8 wavefronts x 4 x global_load_lds_dwordx4 of 1 KiB)
// into a 3-deep ring and writes 16 KiB of d X; 256 persistent workgroups of 512 threads walk the tiles with a grid stride.  Measured in
// the iteration it moves 48 KiB per tile at ~4.3 TB/s.  Variants:
//   layout 0  tile-major records (the store as shipped): tile t at t * TILE_BYTES, the three 16 KiB groups at fixed slots inside it
//   layout 1  group-major: group g of tile t at g * tiles * 16 KiB + t * 16 KiB (each layer's fragments contiguous over tiles)
//   ring depth 2 .. 5, barriers per tile 0 / 2, with / without the d X store, LDS-DMA or plain 16-byte loads into registers,
//   one workgroup per CU (160 KiB of LDS) or two (80 KiB)
// hipcc --offload-arch=gfx950 -O3 bwd_traffic_probe.hip -o bin/bwd_traffic_probe && bin/bwd_traffic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr long TILE_BYTES = 338944;        // the 8x256 NeRF's record per 32-sample tile (331 KiB)
constexpr int Y_SLOT = 40, X_SLOT = 120, O_SLOT = 200;      // three 16-slot groups somewhere in the record
struct Lay { long off[3], stride[3]; };    // operand g (y, x, out) of tile t at off[g] + t * stride[g]

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int LAYOUT, int RING, int NBAR, bool STORE, bool DMA>
__global__ __launch_bounds__(512) void k_traffic(char* __restrict__ store, long tiles, unsigned* __restrict__ sink, const Lay L) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring + wave * 4096);
    auto base = [&](long t, int slot0) -> char* {
        const int g = slot0 == Y_SLOT ? 0 : slot0 == X_SLOT ? 1 : 2;
        return store + L.off[g] + t * L.stride[g] + (long)(2 * wave) * 1024 + lane * 16;
    };
    unsigned acc = 0;
    uint4 regs[RING][4];
    auto issue = [&](long t, int stage) {
        if (t >= tiles) return;
        const char* gy = base(t, Y_SLOT);
        const char* gx = base(t, X_SLOT);
        if (DMA) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(ring0 + stage * (8 * 4096));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gy), "v"(gx), "s"(dst) : "memory");
        } else {
            regs[stage][0] = *reinterpret_cast<const uint4*>(gy);
            regs[stage][1] = *reinterpret_cast<const uint4*>(gy + 1024);
            regs[stage][2] = *reinterpret_cast<const uint4*>(gx);
            regs[stage][3] = *reinterpret_cast<const uint4*>(gx + 1024);
        }
    };
    const long stride = gridDim.x;
    long t = blockIdx.x;
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) issue(t + k * stride, k);
    int it = 0;
    for (; t < tiles; t += stride) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {                // (static stage numbers: the register form needs them)
            if (it % RING != s) continue;
            issue(t + (RING - 1) * stride, (s + RING - 1) % RING);
            if (DMA) {
                long later = 0;
                for (int k = 1; k < RING; ++k) later += t + k * stride < tiles ? 1 : 0;
                if (later >= RING - 1) wait_vm<4 * (RING - 1)>();
                else wait_vm<0>();
                if (NBAR >= 1) __syncthreads();
                const uint4 v = *reinterpret_cast<const uint4*>(ring + (s * 8 + wave) * 4096 + lane * 16);
                acc += v.x;
                if (NBAR >= 2) __syncthreads();
                if (STORE) {
                    char* o = base(t, O_SLOT);
                    *reinterpret_cast<uint4*>(o) = v;
                    *reinterpret_cast<uint4*>(o + 1024) = v;
                }
            } else {
                acc += regs[s][0].x + regs[s][1].y + regs[s][2].z + regs[s][3].w;
                if (NBAR >= 1) __syncthreads();
                if (STORE) {
                    char* o = base(t, O_SLOT);
                    *reinterpret_cast<uint4*>(o) = regs[s][0];
                    *reinterpret_cast<uint4*>(o + 1024) = regs[s][2];
                }
            }
        }
        ++it;
    }
    if (DMA) wait_vm<0>();
    if (acc == 0x12345u) sink[0] = acc;
}

template <int LAYOUT, int RING, int NBAR, bool STORE, bool DMA>
static void run(char* store, long tiles, unsigned* sink, int blocks, size_t lds, const char* name, Lay L = Lay{{0, 0, 0}, {0, 0, 0}}) {
    if (L.stride[0] == 0) {
        if (LAYOUT == 0) L = Lay{{Y_SLOT * 1024L, X_SLOT * 1024L, O_SLOT * 1024L}, {TILE_BYTES, TILE_BYTES, TILE_BYTES}};
        else L = Lay{{0, tiles * 16384, 2 * tiles * 16384}, {16384, 16384, 16384}};
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto k = k_traffic<LAYOUT, RING, NBAR, STORE, DMA>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 2; ++i) k<<<blocks, 512, lds>>>(store, tiles, sink, L);
    hipEventRecord(e0);
    const int iters = 10;
    for (int i = 0; i < iters; ++i) k<<<blocks, 512, lds>>>(store, tiles, sink, L);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    const double bytes = (double)tiles * (32768 + (STORE ? 16384 : 0));
    fprintf(stderr, "  %-86s %.3f ms  %5.2f TB/s\n", name, ms, bytes / ms / 1e9);
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { fprintf(stderr, "  (error: %s)\n", hipGetErrorString(e)); exit(1); }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    fprintf(stderr, "start\n");
    const long tiles = 16384;                    // 4096 x 128 samples
    char* store; unsigned* sink;
    if (hipMalloc(&store, tiles * TILE_BYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    fprintf(stderr, "allocated %p\n", (void*)store);
    for (long o = 0; o < tiles * TILE_BYTES; o += 1L << 30) hipMemset(store + o, 1, tiles * TILE_BYTES - o < (1L << 30) ? tiles * TILE_BYTES - o : 1L << 30);   // (one memset of > 4 GiB faulted)
    fprintf(stderr, "store ready: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    const size_t big = 160 * 1024, half = 80 * 1024;
    fprintf(stderr, "k_wgrad_dgrad traffic, %ld tiles (record %ld B): 32 KiB read (+ 16 KiB written) per tile\n", tiles, TILE_BYTES);
    run<0, 3, 2, true, true>(store, tiles, sink, 256, big, "tile-major, ring 3, 2 barriers, store, 1 workgroup per CU (as shipped)");
    run<0, 3, 0, true, true>(store, tiles, sink, 256, big, "tile-major, ring 3, no barrier, store");
    run<0, 3, 2, false, true>(store, tiles, sink, 256, big, "tile-major, ring 3, 2 barriers, no store");
    run<0, 3, 0, false, true>(store, tiles, sink, 256, big, "tile-major, ring 3, no barrier, no store");
    run<0, 4, 2, true, true>(store, tiles, sink, 256, big, "tile-major, ring 4, 2 barriers, store");
    run<0, 5, 2, true, true>(store, tiles, sink, 256, big, "tile-major, ring 5 (no room for it in the kernel), 2 barriers, store");
    run<0, 5, 0, false, true>(store, tiles, sink, 256, big, "tile-major, ring 5, no barrier, no store");
    run<0, 2, 2, true, true>(store, tiles, sink, 512, half, "tile-major, ring 2, 2 barriers, store, 2 workgroups per CU");
    run<1, 3, 2, true, true>(store, tiles, sink, 256, big, "group-major, ring 3, 2 barriers, store");
    run<1, 3, 0, false, true>(store, tiles, sink, 256, big, "group-major, ring 3, no barrier, no store");
    run<1, 5, 2, true, true>(store, tiles, sink, 256, big, "group-major, ring 5, 2 barriers, store");
    const long T = TILE_BYTES, G = 16384, far = tiles * T - tiles * G;        // a group-major region at the end of the allocation
    run<0, 3, 2, true, true>(store, tiles, sink, 256, big, "y, x tile-major; OUT group-major (own buffer)", Lay{{40 * 1024, 120 * 1024, far}, {T, T, G}});
    run<0, 3, 2, true, true>(store, tiles, sink, 256, big, "x tile-major; y and OUT group-major (gradients in their own buffers)", Lay{{far - tiles * G, 120 * 1024, far}, {G, T, G}});
    run<0, 3, 2, true, true>(store, tiles, sink, 256, big, "tile-major, out slot next to y (56)", Lay{{40 * 1024, 120 * 1024, 56 * 1024}, {T, T, T}});
    run<0, 3, 2, true, true>(store, tiles, sink, 256, big, "tile-major, out slot 201 (1 KiB off)", Lay{{40 * 1024, 120 * 1024, 201 * 1024}, {T, T, T}});
    run<0, 3, 2, true, true>(store, tiles, sink, 256, big, "tile-major, out = y (in place)", Lay{{40 * 1024, 120 * 1024, 40 * 1024}, {T, T, T}});
    for (long rec : {336L * 1024, 344L * 1024, 352L * 1024, 384L * 1024, 512L * 1024, 331L * 1024 + 256, 331L * 1024 + 512}) {
        if (rec * tiles > tiles * T + 0) { /* larger records need a larger allocation */ }
        char nm[128]; snprintf(nm, sizeof nm, "tile-major, record %ld B (%.2f KiB)", rec, rec / 1024.0);
        if (rec <= T) run<0, 3, 2, true, true>(store, tiles, sink, 256, big, nm, Lay{{40 * 1024, 120 * 1024, 200 * 1024}, {rec, rec, rec}});
        else run<0, 3, 2, true, true>(store, tiles * T / rec, sink, 256, big, nm, Lay{{40 * 1024, 120 * 1024, 200 * 1024}, {rec, rec, rec}});
    }
    return 0;
}
