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
// MEASURED (MI355X, fine level 586 x 586 x 390, 2^19 samples, profiles/r02_scatter_binned.txt): results equal the direct kernel's to
// 1.2e-6, global atomics drop as designed -- and the whole thing takes 2.99 ms against 1.43 ms: pass 1 alone is 0.98 ms (the direct
// kernel's block-cooperative structure at one wavefront per SIMD: its gathers and two small GEMMs were HIDDEN under the atomics
// there, they are exposed here), planes 1.14 ms (two 74 KB workgroups per CU, a barrier-separated zero / search / accumulate / flush
// sequence per tile run), lines 0.50 ms, 27 merge-sort launches 0.34 ms.  So the direct form stays the default; this one is what
// evd_voxel_sample_bwd_ws runs when the caller passes scratch (the Python mirror does under EVD_SCATTER=1), kept because it is
// verified and because what it needs next is known: pass 1 rebuilt wavefront-autonomous like the forward gather (k_voxel_sample_w:
// 0.12 ms for the same taps), 32-channel plane slices (4 workgroups per CU), a counting sort on the 11-bit tile keys.
#include <hipcub/hipcub.hpp>

#include "evd_common.h"
#include "voxel.h"

namespace evd {

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
struct LineJobs { LineJob j[8]; };

__global__ __launch_bounds__(SC_NT) void k_scatter_lines(const LineJobs jobs, const float* __restrict__ rows_l, const LTap* __restrict__ ltap, long n, int ctot) {
    const LineJob jb = jobs.j[blockIdx.y];
    if (!jb.grad) return;
    extern __shared__ __attribute__((aligned(16))) float line[];       // [Lp][cg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cg = jb.cg, total = jb.Lp * cg;
    for (int o = tid; o < total; o += SC_NT) line[o] = 0.f;
    __syncthreads();
    const long base = (long)blockIdx.x * SC_LCH, end = base + SC_LCH < n ? base + SC_LCH : n;
    const int spw = 64 / cg, c = lane % cg, sub = lane / cg, step = (SC_NT / 64) * spw;
    constexpr int UN = 4;
    for (long s0 = base + wave * spw + sub; s0 < end; s0 += UN * step) {
        LTap t[UN];
        float v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long s = s0 + u * step;
            if (s < end) {
                t[u] = ltap[s * 3 + jb.comp];
                v[u] = rows_l[s * ctot + jb.coff + jb.c_lo + c];
            } else {
                t[u].c0 = t[u].c1 = 0; t[u].w0 = t[u].w1 = 0.f; v[u] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (t[u].w0 != 0.f) atomicAdd(&line[t[u].c0 * cg + c], t[u].w0 * v[u]);
            if (t[u].w1 != 0.f) atomicAdd(&line[t[u].c1 * cg + c], t[u].w1 * v[u]);
        }
    }
    __syncthreads();
    for (int o = tid; o < total; o += SC_NT) {
        const float a = line[o];
        if (a != 0.f) unsafeAtomicAdd(jb.grad + (long)(o / cg) * jb.C + jb.c_lo + (o % cg), a);
    }
}

static const int kM0[3] = {0, 0, 1}, kM1[3] = {1, 2, 2}, kV[3] = {2, 1, 0};
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

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
        if (Wp > 65535 || Hp > 65535 || (size_t)Lp * 16 * 4 > (size_t)SC_LDS_MAX) return false;
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
    LineJobs lj;
    int nj = 0;
    size_t llds = 0;
    coff = 0;
    for (int i = 0; i < 3; ++i) {
        const int C = g.n_comp[i], Lp = g.grid[kV[i]];
        int cg = C < 32 ? C : 32;
        while ((size_t)Lp * cg * 4 > (size_t)SC_LDS_MAX) cg /= 2;
        for (int c_lo = 0; c_lo < C && gg.line[i]; c_lo += cg) {
            if (nj >= 8) return fail(EVD_E_INVALID, "evd_voxel_sample_bwd: too many line channel groups");
            LineJob& j = lj.j[nj++];
            j.grad = gg.line[i]; j.C = C; j.Lp = Lp; j.coff = coff; j.comp = i; j.c_lo = c_lo; j.cg = cg;
            const size_t l = (size_t)Lp * cg * 4;
            llds = l > llds ? l : llds;
        }
        coff += C;
    }
    for (int k = nj; k < 8; ++k) lj.j[k].grad = nullptr;
    if (nj) {
        EVD_SET_MAX_LDS(k_scatter_lines, (size_t)SC_LDS_MAX);
        hipLaunchKernelGGL(k_scatter_lines, dim3((unsigned)cdiv(n, (long)SC_LCH), nj), dim3(SC_NT), llds, st, lj, (const float*)bo.rows_l, (const LTap*)bo.ltap, n, ctot);
        EVD_LAUNCH_CHECK();
    }
    return EVD_OK;
}

}  // namespace evd
