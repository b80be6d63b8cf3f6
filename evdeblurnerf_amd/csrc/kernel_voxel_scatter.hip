// Tri-plane scatter without one global atomic per (tap, channel): the second pass of evd_voxel_sample_bwd (reference: the backward of
// F.grid_sample in VoxelNeRFBase.compute_appfeature, networks/pdrf/voxnerf.py:132-151, under run_nerf.py:593-601).
//
// The direct kernel (kernel_voxel.hip k_voxel_sample_bwd<false>) issues 576 float atomics per sample; they run at the L2's rate of one
// dword per clock and channel (~250 G adds/s): 1.45 - 1.9 ms per launch, half of a whole blurfactory training iteration.  Here:
//   pass 1  (k_voxel_sample_bwd<true>)  per sample one row of per-channel contributions for the plane taps, one for the line taps, the
//           tap records (clamped cells + weights) and, per plane, the key of the 16 x 16-cell tile its first tap falls in
//   sort    hipCUB radix sort of (tile key, sample) per plane
//   planes  (k_scatter_planes)  a workgroup takes 512 consecutive sorted samples: per run of one tile it accumulates the taps into a
//           17 x 17 x C tile in LDS (ds_add_f32: lanes over channels, so the 64 lanes of a wavefront hit 64 different banks) and adds
//           the touched cells to the gradient ONCE -- samples of different rays share cells (1.5 - 8 samples per cell at the
//           blurfactory sizes), so the global atomics drop 5 - 25 x
//   lines   (k_scatter_lines)  the line gradients are small (<= 586 cells): a workgroup keeps a 32-channel slice of a whole line in LDS for
//           2048 samples, then adds it once
// All loads of the second pass are streaming or row gathers issued several samples ahead (the first cut of this idea walked a
// dependent chain per sample and was latency-bound by two orders of magnitude).
//
// WHAT RUNS BY DEFAULT is the HYBRID at the end of this file (evd_voxel_sample_bwd_ws with scratch): the plane taps stay direct float
// atomics, only the line taps -- a third of the requests, onto <= 586 cells -- go through LDS (k_scatter_lines, 64-bit fixed-point
// accumulators): 1.19 -> 0.90 ms at 2^19 fine-level samples, 1.50 -> 1.10 ms at 655 k coarse-level samples, same sums to 1.2e-6.
// The full binned form below is kept behind EVD_SCATTER=binned.
//
// MEASURED, full binned form (MI355X, fine level 586 x 586 x 390, 2^19 samples, profiles/r02_scatter_binned.txt): results equal the direct kernel's to
// 1.2e-6, global atomics drop as designed -- and the whole thing takes 2.99 ms against 1.43 ms: pass 1 alone is 0.98 ms (the direct
// kernel's block-cooperative structure at one wavefront per SIMD: its gathers and two small GEMMs were HIDDEN under the atomics
// there, they are exposed here), planes 1.14 ms (two 74 KB workgroups per CU, a barrier-separated zero / search / accumulate / flush
// sequence per tile run), lines 0.50 ms, 27 merge-sort launches 0.34 ms.  So the direct form stays the default; this one is what
// selected by EVD_SCATTER=binned, kept because it is verified and because what it needs next is known: its LDS tiles used
// ds_add_f32, which turned out to run at ~0.4 lane-operations per clock and CU (k_scatter_lines below: 497 us with float, 81 us with
// integer LDS atomics) -- fixed-point tiles are the next thing to try --, then pass 1 rebuilt wavefront-autonomous like the forward gather (k_voxel_sample_w:
// 0.12 ms for the same taps), 32-channel plane slices (4 workgroups per CU), a counting sort on the 11-bit tile keys.
#include <hipcub/hipcub.hpp>

#include "evd_common.h"
#include "voxel.h"

namespace evd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SC_TW = SC_TS + 1, SC_CH = 512, SC_NT = 512, SC_LCH = 2048, SC_LDS_MAX = 150 * 1024;

struct PlaneJob {
    float* grad;
    const unsigned *sid, *skey;
    int C, Wp, Hp, tiles_x, coff, comp;
};
struct PlaneJobs { PlaneJob j[3]; };

__global__ __launch_bounds__(SC_NT) void k_scatter_planes(const PlaneJobs jobs, const float* __restrict__ rows_p, const PTap* __restrict__ ptap, long n, int ctot) {
    const PlaneJob jb = blockIdx.y == 0 ? jobs.j[0] : (blockIdx.y == 1 ? jobs.j[1] : jobs.j[2]);
    if (!jb.grad) return;
    extern __shared__ __attribute__((aligned(16))) float tile[];       // [SC_TW][SC_TW][C]
    __shared__ unsigned sid[SC_CH], skey[SC_CH];
    __shared__ int nxt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = jb.C, Wp = jb.Wp, Hp = jb.Hp;
    const long base = (long)blockIdx.x * SC_CH;
    const int cnt = (int)(n - base < SC_CH ? n - base : SC_CH);
    if (tid < cnt) { sid[tid] = jb.sid[base + tid]; skey[tid] = jb.skey[base + tid]; }
    const int spw = 64 / C, c = lane % C, sub = lane / C, cells = SC_TW * SC_TW * C;
    const int cshift = C == 64 ? 6 : (C == 32 ? 5 : 4);
    __syncthreads();
    int r0 = 0;
    while (r0 < cnt) {
        const unsigned key = skey[r0];
        if (tid == 0) nxt = cnt;
        for (int o = tid; o < cells; o += SC_NT) tile[o] = 0.f;
        __syncthreads();
        for (int j = r0 + 1 + tid; j < cnt; j += SC_NT)
            if (skey[j] != key) { atomicMin(&nxt, j); break; }          // sorted: a thread's first hit is its smallest
        __syncthreads();
        const int r1 = nxt;
        const int tx0 = (int)(key % (unsigned)jb.tiles_x) * SC_TS, ty0 = (int)(key / (unsigned)jb.tiles_x) * SC_TS;
        // the run's samples: wavefront w takes spw samples per step; the loads of 4 steps are issued before the first LDS add
        constexpr int UN = 4;
        const int step = (SC_NT / 64) * spw;
        for (int j0 = r0 + wave * spw + sub; j0 < r1; j0 += UN * step) {
            PTap t[UN];
            float v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int j = j0 + u * step;
                if (j < r1) {
                    const long s = sid[j];
                    t[u] = ptap[s * 3 + jb.comp];
                    v[u] = rows_p[s * ctot + jb.coff + c];
                } else {
                    t[u].w[0] = t[u].w[1] = t[u].w[2] = t[u].w[3] = 0.f;
                    t[u].cx0 = t[u].cx1 = (unsigned short)tx0; t[u].cy0 = t[u].cy1 = (unsigned short)ty0;
                    v[u] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int lx0 = t[u].cx0 - tx0, lx1 = t[u].cx1 - tx0, ly0 = t[u].cy0 - ty0, ly1 = t[u].cy1 - ty0;
                if (t[u].w[0] != 0.f) atomicAdd(&tile[((ly0 * SC_TW + lx0) << cshift) + c], t[u].w[0] * v[u]);
                if (t[u].w[1] != 0.f) atomicAdd(&tile[((ly0 * SC_TW + lx1) << cshift) + c], t[u].w[1] * v[u]);
                if (t[u].w[2] != 0.f) atomicAdd(&tile[((ly1 * SC_TW + lx0) << cshift) + c], t[u].w[2] * v[u]);
                if (t[u].w[3] != 0.f) atomicAdd(&tile[((ly1 * SC_TW + lx1) << cshift) + c], t[u].w[3] * v[u]);
            }
        }
        __syncthreads();
        for (int o = tid; o < cells; o += SC_NT) {
            const float a = tile[o];
            if (a != 0.f) {
                const int cell = o >> cshift, ch = o & (C - 1), ly = cell / SC_TW, lx = cell - ly * SC_TW, gx = tx0 + lx, gy = ty0 + ly;
                if (gx < Wp && gy < Hp) unsafeAtomicAdd(jb.grad + ((long)gy * Wp + gx) * C + ch, a);
            }
        }
        __syncthreads();
        r0 = r1;
    }
}

struct LineJob {
    float* grad;
    int C, Lp, coff, comp, c_lo, cg;        // channels [c_lo, c_lo + cg) of component comp
};
struct LineJobs { LineJob j[12]; };

// The line taps.  A workgroup keeps a cg-channel slice of a WHOLE line gradient (<= 586 cells) in LDS for SC_LCH samples and adds
// it to the gradient once.  The LDS accumulators are 64-bit FIXED POINT, not float: ds_add_f32 executes at ~0.4 lane-operations per
// clock and CU on this chip (100 M of them took 445 us here -- no faster than the global float atomics they were to replace), integer
// LDS atomics run 6 x faster.  The scale is a power of two chosen per workgroup from the largest contribution of its chunk
// (max |row| 2^k < 2^49, at most 2^13 terms per accumulator: no overflow), so a float32 contribution converts EXACTLY unless it is
// below 2^-49 of the chunk's maximum: the sums are more accurate than float32 atomics, and deterministic inside the workgroup.
// gmax (optional): the float bits of max |rows_l| over ALL samples, taken by the kernel that wrote the rows: the fixed-point scale is then that
// one instead of this chunk's own maximum, and the chunk's rows are read once (123 -> 89 us per 2^19 samples)
// Workgroup order (round 6): a job reads a 64-byte (16-channel) slice of every row of its chunk, i.e. HALF of each 128-byte line it pulls;
// the other half belongs to the next job of the same chunk.  With the jobs as the grid's y dimension the six jobs of a chunk ran a whole
// x sweep apart and every line came from HBM / the Infinity Cache twice (the kernel moved ~2 GB of rows per iteration at 3.0 TB/s while its LDS
// atomics, ablated, were worth 12 %).  Now the grid is one-dimensional and decoded so that ALL jobs of a chunk are neighbours in time ON ONE XCD
// (workgroup ids go round-robin over the 8 XCDs: id % 8): id = ((chunk / 8) nj + job) 8 + chunk % 8 -- the first job's miss fills that
// XCD's L2 for the others (and the four z-line jobs share their tap records the same way): 769 -> 684 us per iteration; the jobs of a chunk on
// consecutive ids, i.e. on different XCDs: 798 (profiles/r06_scatter_lines_order_ab.log).  Chunks of 4096 / 8192 samples (a quarter of the
// end-of-chunk atomics) change nothing (r06_scatter_lines_chunk_ab.log).
__global__ __launch_bounds__(SC_NT) void k_scatter_lines(const LineJobs jobs, const float* __restrict__ rows_l, const LTap* __restrict__ ltap, long n, int ctot,
                                                       const unsigned* __restrict__ gmax, int nj, int order) {
    int job, chunk;
    if (order == 1) {
        const int q = blockIdx.x >> 3;
        job = q % nj;
        chunk = (q / nj) * 8 + (blockIdx.x & 7);
    } else if (order == 0) {                        // the chunk's jobs consecutive in id: spread over the XCDs
        job = blockIdx.x % nj;
        chunk = blockIdx.x / nj;
    } else {                                        // rounds 3-5: job-major
        const int chunks = (int)((n + SC_LCH - 1) / SC_LCH);
        job = blockIdx.x / chunks;
        chunk = blockIdx.x % chunks;
    }
    if ((long)chunk * SC_LCH >= n) return;
    const LineJob jb = jobs.j[job];
    if (!jb.grad) return;
    extern __shared__ __attribute__((aligned(16))) unsigned long long lacc[];       // [Lp][cg]
    __shared__ float wmax[SC_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cg = jb.cg, total = jb.Lp * cg;
    for (int o = tid; o < total; o += SC_NT) lacc[o] = 0ull;
    const long base = (long)chunk * SC_LCH, end = base + SC_LCH < n ? base + SC_LCH : n;
    // a lane owns 4 consecutive channels of a sample (one 16-byte load): cg / 4 lanes per sample
    const int lps = cg >> 2, spw = 64 / lps, c4 = (lane % lps) * 4, sub = lane / lps, step = (SC_NT / 64) * spw;
    const float* col = rows_l + jb.coff + jb.c_lo + c4;
    float m = gmax ? __uint_as_float(*gmax) : 0.f;
    for (long s = base + wave * spw + sub; s < (gmax ? base : end); s += step) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(col + s * ctot);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = fabsf(v[k]);
            m = a != a ? __builtin_huge_valf() : fmaxf(m, a);          // (fmaxf drops NaNs: they are recorded as +inf)
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    m = 0.f;
#pragma unroll
    for (int w = 0; w < SC_NT / 64; ++w) m = fmaxf(m, wmax[w]);
    if (m > 3.0e38f) {                                  // a NaN / Inf contribution: no fixed-point scale exists -- add this chunk's taps
        for (long s = base + wave * spw + sub; s < end; s += step) {       // directly, so that the gradient shows it as the direct form would
            const LTap t = ltap[s * 3 + jb.comp];
            const f32x4 v = *reinterpret_cast<const f32x4*>(col + s * ctot);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (t.w0 != 0.f) unsafeAtomicAdd(jb.grad + (long)t.c0 * jb.C + jb.c_lo + c4 + k, t.w0 * v[k]);
                if (t.w1 != 0.f) unsafeAtomicAdd(jb.grad + (long)t.c1 * jb.C + jb.c_lo + c4 + k, t.w1 * v[k]);
            }
        }
        return;
    }
    if (!(m > 0.f)) return;                             // nothing to add
    int e;
    (void)frexpf(m, &e);                                // m = f 2^e, f in [0.5, 1)
    if (e < -77) e = -77;                               // 2^(49 - e) must stay a finite float32: a chunk whose maximum is below 2^-78 keeps the scale 2^126
                                                        // (its contributions are then resolved to 2^-126 instead of 2^-49 of the maximum: far below float32 atomics)
    const float up = ldexpf(1.f, 49 - e);               // |w v| up <= 2^49
    constexpr int UN = 4;
    for (long s0 = base + wave * spw + sub; s0 < end; s0 += UN * step) {
        LTap t[UN];
        f32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long s = s0 + u * step;
            if (s < end) {
                t[u] = ltap[s * 3 + jb.comp];
                v[u] = *reinterpret_cast<const f32x4*>(col + s * ctot);
            } else {
                t[u].c0 = t[u].c1 = 0; t[u].w0 = t[u].w1 = 0.f; v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#ifdef EVD_SL_NO_ATOMICS        // developer ablation (tools/dev/scatter_lines_ablation.sh; results wrong): the loads and conversions without the LDS atomics
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long a = (unsigned long long)__float2ll_rn((t[u].w0 * v[u][k]) * up) + (unsigned long long)__float2ll_rn((t[u].w1 * v[u][k]) * up);
                if (a == 0x123456789abcdefull) lacc[t[u].c0 * cg + c4 + k] = a;
            }
#else
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (t[u].w0 != 0.f) {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&lacc[t[u].c0 * cg + c4 + k], (unsigned long long)__float2ll_rn((t[u].w0 * v[u][k]) * up));
            }
            if (t[u].w1 != 0.f) {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&lacc[t[u].c1 * cg + c4 + k], (unsigned long long)__float2ll_rn((t[u].w1 * v[u][k]) * up));
            }
        }
#endif
    }
    __syncthreads();
    const double down = ldexp(1.0, e - 49);
    for (int o = tid; o < total; o += SC_NT) {
        const long long a = (long long)lacc[o];
        if (a != 0) unsafeAtomicAdd(jb.grad + (long)(o / cg) * jb.C + jb.c_lo + (o % cg), (float)((double)a * down));
    }
}

static const int kM0[3] = {0, 0, 1}, kM1[3] = {1, 2, 2}, kV[3] = {2, 1, 0};
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static int launch_lines(const GridParams& g, const GridGrads& gg, const float* rows_l, const LTap* ltap, long n, hipStream_t st, const unsigned* gmax = nullptr) {
    const int ctot = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
    LineJobs lj;
    int nj = 0, coff = 0;
    size_t llds = 0;
    for (int i = 0; i < 3; ++i) {
        const int C = g.n_comp[i], Lp = g.grid[kV[i]];
        int cg = 16;                                      // 64-bit accumulators: 16-channel slices (586 cells: 75 KB)
        while (cg > 4 && (size_t)Lp * cg * 8 > (size_t)SC_LDS_MAX / 2) cg /= 2;
        for (int c_lo = 0; c_lo < C && gg.line[i]; c_lo += cg) {
            if (nj >= 12) return fail(EVD_E_INVALID, "evd_voxel_sample_bwd: too many line channel groups");
            LineJob& j = lj.j[nj++];
            j.grad = gg.line[i]; j.C = C; j.Lp = Lp; j.coff = coff; j.comp = i; j.c_lo = c_lo; j.cg = cg;
            const size_t l = (size_t)Lp * cg * 8;
            llds = l > llds ? l : llds;
        }
        coff += C;
    }
    for (int k = nj; k < 12; ++k) lj.j[k].grad = nullptr;
    if (nj) {
        EVD_SET_MAX_LDS(k_scatter_lines, (size_t)SC_LDS_MAX);
        // EVD_SCATTER_LINES_ORDER (developer switch, A/B of the order above): job = the chunk's jobs on consecutive ids, old = job-major (rounds 3-5)
        static const int order = [] { const char* e = getenv("EVD_SCATTER_LINES_ORDER"); return !e ? 1 : (!strcmp(e, "job") ? 0 : (!strcmp(e, "old") ? 2 : 1)); }();
        const long chunks = cdiv(n, (long)SC_LCH), chunks8 = cdiv(chunks, 8L) * 8;
        hipLaunchKernelGGL(k_scatter_lines, dim3((unsigned)((order == 1 ? chunks8 : chunks) * nj)), dim3(SC_NT), llds, st, lj, rows_l, ltap, n, ctot, gmax, nj, order);
        EVD_LAUNCH_CHECK();
    }
    return EVD_OK;
}

static size_t sort_temp_bytes(long n) {
    size_t b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, (int)(n > 0 ? n : 1));
    return b;
}

// The binned form needs channel counts the lane mapping handles, grids whose cell coordinates fit 16 bits and lines that fit LDS.
bool voxel_scatter_binned_ok(const GridParams& g, const GridGrads& gg, long n) {
    for (int i = 0; i < 3; ++i) {
        const int C = g.n_comp[i];
        if (C != 16 && C != 32 && C != 64) return false;
        const int Wp = g.grid[kM0[i]], Hp = g.grid[kM1[i]], Lp = g.grid[kV[i]];
        if (Wp > 65535 || Hp > 65535 || (size_t)Lp * 16 * 8 > (size_t)SC_LDS_MAX / 2) return false;      // a 16-channel slice of a whole line (64-bit accumulators) must fit half the LDS
    }
    return n > 0 && n < (1L << 31);
}

size_t voxel_scatter_workspace_bytes(const GridParams& g, long n) {
    if (n <= 0) return 0;
    const size_t ctot = (size_t)(g.n_comp[0] + g.n_comp[1] + g.n_comp[2]), N = (size_t)n;
    return 2 * al256(N * ctot * 4) + al256(N * 3 * sizeof(PTap)) + al256(N * 3 * sizeof(LTap)) + 6 * al256(N * 4) + 4 * al256(N * 4) + al256(sort_temp_bytes(n)) + 512;
}

int launch_voxel_sample_bwd_binned(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                                   float* d_pts, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (workspace_bytes < voxel_scatter_workspace_bytes(g, n)) return fail(EVD_E_WORKSPACE, "evd_voxel_sample_bwd: workspace %zu < %zu bytes", workspace_bytes, voxel_scatter_workspace_bytes(g, n));
    const int ctot = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
    const size_t N = (size_t)n;
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    auto take = [&](size_t bytes) { char* p = w; w += al256(bytes); return p; };
    BinOut bo;
    bo.rows_p = (float*)take(N * ctot * 4);
    bo.rows_l = (float*)take(N * ctot * 4);
    bo.ptap = (PTap*)take(N * 3 * sizeof(PTap));
    bo.ltap = (LTap*)take(N * 3 * sizeof(LTap));
    unsigned *skey[3], *sid[3];
    for (int i = 0; i < 3; ++i) { bo.keys[i] = (unsigned*)take(N * 4); skey[i] = (unsigned*)take(N * 4); }
    bo.ids = (unsigned*)take(N * 4);
    for (int i = 0; i < 3; ++i) sid[i] = (unsigned*)take(N * 4);
    size_t tb = sort_temp_bytes(n);
    void* temp = take(tb);
    int bits[3];
    for (int i = 0; i < 3; ++i) {
        const int Wp = g.grid[kM0[i]], Hp = g.grid[kM1[i]];
        bo.tiles_x[i] = (int)cdiv(Wp, SC_TS);
        const long tiles = (long)bo.tiles_x[i] * cdiv(Hp, SC_TS);
        bits[i] = 1;
        while ((1L << bits[i]) < tiles) ++bits[i];
    }
    int rc = launch_voxel_sample_bwd_pass1(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo, st);
    if (rc) return rc;
    PlaneJobs pj;
    int coff = 0;
    size_t plds = 0;
    for (int i = 0; i < 3; ++i) {
        PlaneJob& j = pj.j[i];
        j.grad = gg.plane[i]; j.sid = sid[i]; j.skey = skey[i]; j.C = g.n_comp[i]; j.Wp = g.grid[kM0[i]]; j.Hp = g.grid[kM1[i]];
        j.tiles_x = bo.tiles_x[i]; j.coff = coff; j.comp = i;
        coff += g.n_comp[i];
        if (j.grad) {
            EVD_HIP(hipcub::DeviceRadixSort::SortPairs(temp, tb, (const unsigned*)bo.keys[i], skey[i], (const unsigned*)bo.ids, sid[i], (int)n, 0, bits[i], st));
            const size_t l = (size_t)SC_TW * SC_TW * j.C * 4;
            plds = l > plds ? l : plds;
        }
    }
    if (plds) {
        EVD_SET_MAX_LDS(k_scatter_planes, (size_t)SC_TW * SC_TW * 64 * 4);      // (the attribute is set once: the largest tile)
        hipLaunchKernelGGL(k_scatter_planes, dim3((unsigned)cdiv(n, (long)SC_CH), 3), dim3(SC_NT), plds, st, pj, (const float*)bo.rows_p, (const PTap*)bo.ptap, n, ctot);
        EVD_LAUNCH_CHECK();
    }
    if ((rc = launch_lines(g, gg, (const float*)bo.rows_l, (const LTap*)bo.ltap, n, st))) return rc;
    return EVD_OK;
}

// ---- the x-y plane's taps of the hybrid form ---------------------------------------------------------------------------------------------
// A tile is 32 consecutive samples of one ray, and the rays of an NDC scene (every shipped LLFF-type config) run along z: the 128 x-y taps
// of a tile address ~12 distinct cells.  The sum over the taps that share a cell is a small GEMM,
//     Out[slot, channel] = sum_s M[slot, s] G[s, channel],   M[slot, s] = the bilinear weight with which sample s touches window cell `slot`
// (an 8 x 8-cell window anchored at the tile's smallest cell: 64 x 32, built in LDS -- lane s writes its own column, no atomics) and G = the
// rows k_voxel_sample_bwd<3> left (d coef x line value, 64 channels): 32-64 exact-float32 MFMAs per WAVEFRONT, which owns its tile from
// the tap records to the adds (no block barrier), then ONE float atomic per occupied cell and channel: ~1.5 requests of 64 bytes per sample
// instead of 16.  A tile whose taps do not fit the window (rays across the plane), or whose sums are not finite (an Inf / NaN must land on
// the cells its sample touches, nowhere else), adds tap by tap with the lanes over the channels.
constexpr int XY_W = 8, XY_SLOTS = XY_W * XY_W, XY_STR = 33, XY_C = 64;
__global__ __launch_bounds__(256) void k_scatter_xy(const float* __restrict__ rows, const PTap* __restrict__ ptap, long n, float* __restrict__ grad, int Wp) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ float wm_all[4][XY_SLOTS * XY_STR];
    __shared__ int wi_all[4][8];
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, mn = ln & 31, kb = ln >> 5;
    float* wm = wm_all[wv];
    int* wi = wi_all[wv];
    const long t0 = ((long)blockIdx.x * 4 + wv) * 32;
    if (t0 >= n) return;                                   // wavefront-uniform: there is no block barrier in this kernel
    auto wave_sync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    PTap tp;
    tp.cx0 = tp.cx1 = tp.cy0 = tp.cy1 = 0;
    tp.w[0] = tp.w[1] = tp.w[2] = tp.w[3] = 0.f;
    if (ln < 32 && t0 + ln < n) tp = ptap[t0 + ln];
    // the rows of the tile as MFMA B operands: b[ct][j] = G[sample 2 j + kb][channel 32 ct + mn]
    float b0[16], b1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long sm = t0 + 2 * j + kb;
        b0[j] = sm < n ? rows[sm * XY_C + mn] : 0.f;
        b1[j] = sm < n ? rows[sm * XY_C + 32 + mn] : 0.f;
    }
    if (ln < 8) wi[ln] = ln < 2 ? 0x7fffffff : (ln < 4 ? -1 : 0);
    for (int o = ln; o < XY_SLOTS * XY_STR; o += 64) wm[o] = 0.f;
    wave_sync();
    int cx[4], cy[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        cx[t] = (t & 1) ? tp.cx1 : tp.cx0;
        cy[t] = (t & 2) ? tp.cy1 : tp.cy0;
        if (tp.w[t] != 0.f) {
            atomicMin(&wi[0], cx[t]); atomicMin(&wi[1], cy[t]);
            atomicMax(&wi[2], cx[t]); atomicMax(&wi[3], cy[t]);
        }
    }
    wave_sync();
    const int x0 = wi[0], y0 = wi[1], x1 = wi[2], y1 = wi[3];
    if (x1 < 0) return;                                    // no live tap in the tile
    bool done = false;
    if (x1 - x0 < XY_W && y1 - y0 < XY_W) {
        unsigned lo = 0u, hi = 0u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (tp.w[t] != 0.f) {
                const int slot = (cy[t] - y0) * XY_W + (cx[t] - x0);
                wm[slot * XY_STR + ln] += tp.w[t];         // column `ln` belongs to this lane (lanes >= 32 hold zero weights)
                if (slot < 32) lo |= 1u << slot; else hi |= 1u << (slot - 32);
            }
        }
        if (lo) atomicOr((unsigned*)&wi[5], lo);
        if (hi) atomicOr((unsigned*)&wi[6], hi);
        wave_sync();
        const unsigned occ[2] = {(unsigned)wi[5], (unsigned)wi[6]};
        f32x16 acc[2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q >> 1][q & 1][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float a0 = wm[mn * XY_STR + 2 * j + kb];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1[j], acc[0][1], 0, 0, 0);
        }
        if (occ[1]) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float a1 = wm[(32 + mn) * XY_STR + 2 * j + kb];
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[j], acc[1][1], 0, 0, 0);
            }
        }
        int bad = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) bad |= !(fabsf(acc[q >> 1][q & 1][r]) <= 3.4028234e38f);
        if (!__any(bad)) {
            done = true;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (!occ[rt]) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sr = (r & 3) + 8 * (r >> 2) + 4 * kb;          // slot inside this half (the 32 x 32 accumulator layout)
                    if ((occ[rt] >> sr) & 1u) {
                        const int slot = rt * 32 + sr;
                        float* dst = grad + ((long)(y0 + slot / XY_W) * Wp + x0 + slot % XY_W) * XY_C + mn;
                        unsafeAtomicAdd(dst, acc[rt][0][r]);
                        unsafeAtomicAdd(dst + 32, acc[rt][1][r]);
                    }
                }
            }
        }
    }
    if (!done) {                                           // tap by tap, the lanes over the 64 channels
        for (int s = 0; s < 32 && t0 + s < n; ++s) {
            const float v = rows[(t0 + s) * XY_C + ln];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tp.w[t]), s));
                if (w == 0.f) continue;
                const int px = __builtin_amdgcn_readlane(cx[t], s), py = __builtin_amdgcn_readlane(cy[t], s);
                unsafeAtomicAdd(grad + ((long)py * Wp + px) * XY_C + ln, w * v);
            }
        }
    }
}

// the x-y plane leaves the main kernel's atomic path only on request (EVD_SCATTER_WIN=1: a gain on rays along z, a loss on oblique ones; the
// numbers are at k_voxel_sample_bwd) and when it has the channel count k_scatter_xy is built for
static bool xy_deferred(const GridParams& g, const GridGrads& gg) {
    const char* e = getenv("EVD_SCATTER_WIN");
    return (e && e[0] == '1') && g.n_comp[0] == XY_C && gg.plane[0] && g.grid[0] <= 65535 && g.grid[1] <= 65535 && g.app_dim == 32 &&
           (g.n_comp[0] + g.n_comp[1] + g.n_comp[2]) % 32 == 0 && g.n_comp[0] + g.n_comp[1] + g.n_comp[2] <= 96;
}

// ---- hybrid: the 16-channel planes' taps by direct atomics (k_voxel_sample_bwd<2 / 3>), the line taps through k_scatter_lines, the x-y
// plane's through k_scatter_xy ------------------------------------------------------------------------------------------------------------
size_t voxel_scatter_hybrid_workspace_bytes(const GridParams& g, long n) {
    if (n <= 0) return 0;
    const size_t ctot = (size_t)(g.n_comp[0] + g.n_comp[1] + g.n_comp[2]);
    // rows_l, ltap, then EITHER the coefficient rows of the wavefront-autonomous form OR the x-y rows + tap records of the windowed form
    const size_t tail = al256((size_t)n * ctot * 4) > al256((size_t)n * XY_C * 4) + al256((size_t)n * sizeof(PTap)) ? al256((size_t)n * ctot * 4)
                                                                                                                   : al256((size_t)n * XY_C * 4) + al256((size_t)n * sizeof(PTap));
    return al256((size_t)n * ctot * 4) + al256((size_t)n * 3 * sizeof(LTap)) + tail + 512;
}

// developer switch: EVD_SCATTER_FORM=block selects round 2's block-cooperative main kernel (k_voxel_sample_bwd<2>) behind the hybrid entry
static bool scatter_form_block() { static const bool b = [] { const char* e = getenv("EVD_SCATTER_FORM"); return e && !strcmp(e, "block"); }(); return b; }

int launch_voxel_sample_bwd_hybrid(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                                   float* d_pts, void* workspace, size_t workspace_bytes, hipStream_t st, bool half_grids) {
    if (workspace_bytes < voxel_scatter_hybrid_workspace_bytes(g, n))
        return fail(EVD_E_WORKSPACE, "evd_voxel_sample_bwd: workspace %zu < %zu bytes", workspace_bytes, voxel_scatter_hybrid_workspace_bytes(g, n));
    const size_t ctot = (size_t)(g.n_comp[0] + g.n_comp[1] + g.n_comp[2]);
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    BinOut bo{};
    bo.rows_l = (float*)w;
    bo.ltap = (LTap*)(w + al256((size_t)n * ctot * 4));
    if (voxel_sample_bwd_w_ok(g) && !scatter_form_block() && !xy_deferred(g, gg)) {
        // round 3: wavefront-autonomous pass (plane taps incl. the run-length x-y walk, rows for the lines, coefficient rows for the basis GEMM)
        float* coef = (float*)(w + al256((size_t)n * ctot * 4) + al256((size_t)n * 3 * sizeof(LTap)));
        // the last 256 bytes of the workspace: max |line row| of the whole batch, taken by the main kernel (k_scatter_lines' scale)
        static const bool own_max = getenv("EVD_SCATTER_LINES_OWN_MAX") != nullptr;     // developer switch: every chunk finds its own maximum (round 3)
        unsigned* lmax = own_max ? nullptr : (unsigned*)(w + voxel_scatter_hybrid_workspace_bytes(g, n) - 512);
        if (lmax) EVD_HIP(hipMemsetAsync(lmax, 0, sizeof(unsigned), st));
        int rc = launch_voxel_sample_bwd_w(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo.rows_l, bo.ltap, coef, st, lmax, half_grids);
        if (rc) return rc;
        GridGrads gl = gg;
        if (voxel_sample_bwd_w_lines12(g)) gl.line[1] = gl.line[2] = nullptr;      // added inside the kernel: only the z line is left for the LDS slices
        return launch_lines(g, gl, (const float*)bo.rows_l, (const LTap*)bo.ltap, n, st, lmax);
    }
    const bool xy = xy_deferred(g, gg);
    if (xy) {
        char* w2 = w + al256((size_t)n * ctot * 4) + al256((size_t)n * 3 * sizeof(LTap));
        bo.rows_p = (float*)w2;
        bo.ptap = (PTap*)(w2 + al256((size_t)n * XY_C * 4));
    }
    int rc = launch_voxel_sample_bwd_planes(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo, st);
    if (rc) return rc;
    if (xy) {
        hipLaunchKernelGGL(k_scatter_xy, dim3((unsigned)cdiv(cdiv(n, 32L), 4L)), dim3(256), 0, st, (const float*)bo.rows_p, (const PTap*)bo.ptap, n, gg.plane[0], g.grid[0]);
        EVD_LAUNCH_CHECK();
    }
    return launch_lines(g, gg, (const float*)bo.rows_l, (const LTap*)bo.ltap, n, st);
}

}  // namespace evd
