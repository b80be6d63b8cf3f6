// The once-per-dataset tables of the event loader on the device (last clause): what LLFFEventsDataset.load_event_data
// (reference data/loader_events.py:150-257) and load_events_h5 (utils/events.py:11-69) compute with numpy on the CPU from the arrays of
// events.h5 -- the file formats themselves stay with the caller --:
//   evd_event_coord_ids   utils/events.py:39-66  the silent pixels, np.unique over [event coordinates ; silent pixels] as float64 (x, y) rows
//                                                viewed as 16 raw bytes (to_flattenvoid): ids in the BYTE order of the little-endian pairs
//   evd_event_filter      loader_events.py:191, 203-206   events inside the range of the known poses, polarity 0 -> -1
//   evd_event_color_map   loader_events.py:208-236        Bayer colour of a coordinate id: by its integer pixel, or through the ev_map
//                                                         inverse maps (the LAST pixel in raster order whose map entry equals the coordinate)
// The successor graph is evd_compute_successor (kernels_events.hip).  Integer / byte work, HBM-bound, run once: two stable 64-bit radix
// sorts (hipCUB) over N + h w keys are the cost; results are bit-identical to the reference's.
#include <hipcub/hipcub.hpp>

#include "evd_common.h"

namespace evd {

typedef unsigned long long u64;

// memcmp order of a little-endian float64 = unsigned order of its byte-swapped bits
__device__ __forceinline__ u64 et_key(double v) { return __builtin_bswap64((u64)__double_as_longlong(v)); }
constexpr u64 ET_DROP = ~0ull;              // key of an entry that is not part of the set (a pixel with events): NaN bits, no coordinate has them

__global__ __launch_bounds__(256) void k_et_mark(const float* __restrict__ x, const float* __restrict__ y, long N, int h, int w, unsigned char* __restrict__ silent) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= N) return;
    // np.clip(np.round(y).astype(np.int32), 0, h - 1): round half to even on the float32 value (utils/events.py:40-41)
    const int yy = min(max((int)rintf(y[i]), 0), h - 1), xx = min(max((int)rintf(x[i]), 0), w - 1);
    silent[(long)yy * w + xx] = 0;
}

__global__ __launch_bounds__(256) void k_et_keys(const float* __restrict__ x, const float* __restrict__ y, long N, int h, int w,
                                                 const unsigned char* __restrict__ silent, u64* __restrict__ kx, u64* __restrict__ ky, int* __restrict__ iota) {
    const long i = blockIdx.x * 256L + threadIdx.x, M = N + (long)h * w;
    if (i >= M) return;
    iota[i] = (int)i;
    double cx, cy;
    bool in = true;
    if (i < N) { cx = (double)x[i]; cy = (double)y[i]; }                 // float32 coordinates meet the int64 silent pixels: float64 rows (:52)
    else {
        const long j = i - N;
        cx = (double)(j % w); cy = (double)(j / w);                       // np.where order, [:, ::-1]: (x, y) of pixel j
        in = silent[j] != 0;
    }
    kx[i] = in ? et_key(cx) : ET_DROP;
    ky[i] = in ? et_key(cy) : ET_DROP;
}

__global__ __launch_bounds__(256) void k_et_gather(const u64* __restrict__ src, const int* __restrict__ idx, long M, u64* __restrict__ dst) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < M) dst[i] = src[idx[i]];
}

// head[k] = 1 where the sorted entry k starts a new (x, y) group
__global__ __launch_bounds__(256) void k_et_heads(const u64* __restrict__ kx_sorted, const u64* __restrict__ ky, const int* __restrict__ perm, long M, int* __restrict__ head) {
    const long k = blockIdx.x * 256L + threadIdx.x;
    if (k >= M) return;
    head[k] = (k == 0 || kx_sorted[k] != kx_sorted[k - 1] || ky[perm[k]] != ky[perm[k - 1]]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_et_write(const float* __restrict__ x, const float* __restrict__ y, long N, int h, int w, const u64* __restrict__ kx_sorted,
                                                  const int* __restrict__ perm, const int* __restrict__ head, const int* __restrict__ uid_incl,
                                                  const unsigned char* __restrict__ silent, const int* __restrict__ silent_rank_incl,
                                                  long long* __restrict__ ev_ids, long long* __restrict__ noev_ids, double* __restrict__ id_to_coords,
                                                  long long* __restrict__ counts) {
    const long k = blockIdx.x * 256L + threadIdx.x, M = N + (long)h * w;
    if (k >= M) return;
    const bool in = kx_sorted[k] != ET_DROP;                              // the dropped entries sort last, as one group
    const int uid = uid_incl[k] - 1;
    const long i = perm[k];
    if (in) {
        if (i < N) ev_ids[i] = uid;
        else noev_ids[silent_rank_incl[i - N] - 1] = uid;                 // the silent pixels keep their raster order (np.where)
        if (head[k]) {                                                     // return_index: the first occurrence (stable sorts) -- any member has the value
            const long j = i - N;
            id_to_coords[2 * (long)uid] = i < N ? (double)x[i] : (double)(j % w);
            id_to_coords[2 * (long)uid + 1] = i < N ? (double)y[i] : (double)(j / w);
        }
    }
    if (k == M - 1) {
        counts[0] = in ? uid + 1 : uid;                                    // (a dropped group at the end is not a coordinate)
        counts[1] = silent_rank_incl[(long)h * w - 1];
    }
}

__global__ __launch_bounds__(256) void k_et_u8_to_i32(const unsigned char* __restrict__ a, long n, int* __restrict__ o) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) o[i] = a[i];
}

// ---- range filter + polarity normalisation
__global__ __launch_bounds__(256) void k_ef_flags(const double* __restrict__ t, const double* __restrict__ p, long N, double tmin, double tmax, int* __restrict__ keep,
                                                  int* __restrict__ pminmax) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= N) return;
    const int kp = (t[i] >= tmin && t[i] <= tmax) ? 1 : 0;
    keep[i] = kp;
    if (kp) {
        const int pi = (int)p[i];
        if (pi < pminmax[0]) atomicMin(pminmax, pi);
        if (pi > pminmax[1]) atomicMax(pminmax + 1, pi);
    }
}

__global__ __launch_bounds__(256) void k_ef_write(const long long* __restrict__ ids, const double* __restrict__ t, const double* __restrict__ p, long N,
                                                  const int* __restrict__ keep, const int* __restrict__ pos_incl, const int* __restrict__ pminmax,
                                                  double* __restrict__ out, long long* __restrict__ count, int* __restrict__ bad) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= N) return;
    const bool zero_is_neg = pminmax[0] == 0;                              // `if events[:, -1].min() == 0: events[p == 0] = -1` (:203-205)
    if (keep[i]) {
        const long o = pos_incl[i] - 1;
        double pv = p[i];
        if (zero_is_neg && pv == 0.0) pv = -1.0;
        out[o * 3] = (double)ids[i];
        out[o * 3 + 1] = t[i];
        out[o * 3 + 2] = pv;
    }
    if (i == N - 1) {
        *count = pos_incl[i];
        // the reference asserts max == 1 and min == -1 after the normalisation (:206)
        const int mn = zero_is_neg ? -1 : pminmax[0];
        if (bad && pos_incl[i] > 0 && !(pminmax[1] == 1 && mn == -1)) *bad = 1;
    }
}

// ---- Bayer colour of a coordinate id
__device__ __forceinline__ int bayer_channel(int j, int i) { return ((j & 1) == 0) ? ((i & 1) == 0 ? 0 : 1) : ((i & 1) == 0 ? 1 : 2); }     // r g / g b (:209-213)

__global__ __launch_bounds__(256) void k_cm_int(const double* __restrict__ id_to_coords, long Nc, int h, int w, unsigned char* __restrict__ cmap) {
    const long id = blockIdx.x * 256L + threadIdx.x;
    if (id >= Nc) return;
    const long long xi = (long long)id_to_coords[2 * id], yi = (long long)id_to_coords[2 * id + 1];       // np.int64(...) truncates
    unsigned char c[3] = {0, 0, 0};
    if (xi >= 0 && xi < w && yi >= 0 && yi < h) c[bayer_channel((int)yi, (int)xi)] = 1;
    cmap[id * 3] = c[0]; cmap[id * 3 + 1] = c[1]; cmap[id * 3 + 2] = c[2];
}

// the dict lookup `(inv_mapx[j, i], inv_mapy[j, i]) in coords_to_id` (:229): id_to_coords is sorted by (key x, key y) -> binary search
__global__ __launch_bounds__(256) void k_cm_lookup(const double* __restrict__ id_to_coords, long Nc, int h, int w, const float* __restrict__ mx, const float* __restrict__ my,
                                                   int* __restrict__ last_pixel) {
    const long pix = blockIdx.x * 256L + threadIdx.x;
    if (pix >= (long)h * w) return;
    const u64 qx = et_key((double)mx[pix]), qy = et_key((double)my[pix]);
    long lo = 0, hi = Nc;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        const u64 cx = et_key(id_to_coords[2 * mid]), cy = et_key(id_to_coords[2 * mid + 1]);
        if (cx < qx || (cx == qx && cy < qy)) lo = mid + 1; else hi = mid;
    }
    if (lo < Nc && et_key(id_to_coords[2 * lo]) == qx && et_key(id_to_coords[2 * lo + 1]) == qy) atomicMax(last_pixel + lo, (int)pix);   // the later pixel overwrites (:228-230)
}

__global__ __launch_bounds__(256) void k_cm_from_pixels(const int* __restrict__ last_pixel, long Nc, int w, unsigned char* __restrict__ cmap) {
    const long id = blockIdx.x * 256L + threadIdx.x;
    if (id >= Nc) return;
    unsigned char c[3] = {0, 0, 0};
    const int pix = last_pixel[id];
    if (pix >= 0) c[bayer_channel(pix / w, pix % w)] = 1;
    cmap[id * 3] = c[0]; cmap[id * 3 + 1] = c[1]; cmap[id * 3 + 2] = c[2];
}

// every coordinate that is not a silent pixel's must have got exactly one colour (:231-234)
__global__ __launch_bounds__(256) void k_cm_check(const unsigned char* __restrict__ cmap, long Nc, const unsigned char* __restrict__ is_noev, int* __restrict__ bad) {
    const long id = blockIdx.x * 256L + threadIdx.x;
    if (id >= Nc || is_noev[id]) return;
    if (cmap[id * 3] + cmap[id * 3 + 1] + cmap[id * 3 + 2] != 1) *bad = 1;
}

__global__ __launch_bounds__(256) void k_cm_mark_noev(const long long* __restrict__ noev_ids, long n, long Nc, unsigned char* __restrict__ is_noev) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n && noev_ids[i] >= 0 && noev_ids[i] < Nc) is_noev[noev_ids[i]] = 1;
}

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace evd

using namespace evd;

extern "C" {

size_t evd_event_coord_ids_workspace_bytes(long N, int h, int w) {
    if (N < 0 || h < 1 || w < 1) return 0;
    const size_t M = (size_t)N + (size_t)h * w;
    size_t sort_bytes = 0, scan_bytes = 0;
    const u64* k = nullptr; u64* ko = nullptr; const int* v = nullptr; int* vo = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k, ko, v, vo, (int)M);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, v, vo, (int)M);
    const size_t tmp = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    // kx, ky, key scratch x 2 (u64) | iota, perm a, perm b, head, uid (int) | silent (u8) + its int image and inclusive rank
    return al256(tmp) + 4 * al256(M * 8) + 5 * al256(M * 4) + al256((size_t)h * w) + 2 * al256((size_t)h * w * 4) + 512;
}

int evd_event_coord_ids(const float* x, const float* y, long N, int h, int w, long long* ev_coord_ids, long long* noev_coord_ids, double* id_to_coords,
                        long long* counts, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(N >= 0 && h >= 1 && w >= 1 && (long)N + (long)h * w < (1L << 31), "evd_event_coord_ids: bad sizes (N + h w < 2^31)");
    EVD_REQUIRE(noev_coord_ids && id_to_coords && counts && (N == 0 || (x && y && ev_coord_ids)), "evd_event_coord_ids: null argument");
    const size_t need = evd_event_coord_ids_workspace_bytes(N, h, w);
    if (!workspace || workspace_bytes < need) return fail(EVD_E_WORKSPACE, "evd_event_coord_ids: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    const long HW = (long)h * w, M = N + HW;
    size_t sort_bytes = 0, scan_bytes = 0;
    {
        const u64* k = nullptr; u64* ko = nullptr; const int* v = nullptr; int* vo = nullptr;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k, ko, v, vo, (int)M);
        (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, v, vo, (int)M);
    }
    const size_t tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    void* tmp = p; p += al256(tmp_bytes);
    u64* kx = (u64*)p; p += al256(M * 8);
    u64* ky = (u64*)p; p += al256(M * 8);
    u64* ka = (u64*)p; p += al256(M * 8);
    u64* kb = (u64*)p; p += al256(M * 8);
    int* iota = (int*)p; p += al256(M * 4);
    int* pa = (int*)p; p += al256(M * 4);
    int* pb = (int*)p; p += al256(M * 4);
    int* head = (int*)p; p += al256(M * 4);
    int* uid = (int*)p; p += al256(M * 4);
    unsigned char* silent = (unsigned char*)p; p += al256(HW);
    int* silent_i = (int*)p; p += al256(HW * 4);
    int* silent_rank = (int*)p; p += al256(HW * 4);
    EVD_HIP(hipMemsetAsync(silent, 1, HW, st));
    if (N > 0) hipLaunchKernelGGL(k_et_mark, dim3((unsigned)cdiv(N, 256L)), dim3(256), 0, st, x, y, N, h, w, silent);
    hipLaunchKernelGGL(k_et_keys, dim3((unsigned)cdiv(M, 256L)), dim3(256), 0, st, x, y, N, h, w, silent, kx, ky, iota);
    EVD_LAUNCH_CHECK();
    // lexicographic (x bytes, then y bytes): stable sort by the minor key, then by the major one
    size_t sb = tmp_bytes;
    EVD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, sb, (const u64*)ky, ka, (const int*)iota, pa, (int)M, 0, 64, st));
    hipLaunchKernelGGL(k_et_gather, dim3((unsigned)cdiv(M, 256L)), dim3(256), 0, st, (const u64*)kx, (const int*)pa, M, kb);
    sb = tmp_bytes;
    EVD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, sb, (const u64*)kb, ka, (const int*)pa, pb, (int)M, 0, 64, st));       // ka = sorted x keys, pb = the permutation
    hipLaunchKernelGGL(k_et_heads, dim3((unsigned)cdiv(M, 256L)), dim3(256), 0, st, (const u64*)ka, (const u64*)ky, (const int*)pb, M, head);
    sb = tmp_bytes;
    EVD_HIP(hipcub::DeviceScan::InclusiveSum(tmp, sb, (const int*)head, uid, (int)M, st));
    hipLaunchKernelGGL(k_et_u8_to_i32, dim3((unsigned)cdiv(HW, 256L)), dim3(256), 0, st, (const unsigned char*)silent, HW, silent_i);
    sb = tmp_bytes;
    EVD_HIP(hipcub::DeviceScan::InclusiveSum(tmp, sb, (const int*)silent_i, silent_rank, (int)HW, st));
    hipLaunchKernelGGL(k_et_write, dim3((unsigned)cdiv(M, 256L)), dim3(256), 0, st, x, y, N, h, w, (const u64*)ka, (const int*)pb, (const int*)head, (const int*)uid,
                       (const unsigned char*)silent, (const int*)silent_rank, ev_coord_ids, noev_coord_ids, id_to_coords, counts);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

size_t evd_event_filter_workspace_bytes(long N) {
    if (N < 0) return 0;
    size_t scan_bytes = 0;
    const int* v = nullptr; int* vo = nullptr;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, v, vo, (int)(N > 0 ? N : 1));
    return al256(scan_bytes) + 2 * al256((size_t)(N > 0 ? N : 1) * 4) + 256 + 512;
}

int evd_event_filter(const long long* coord_ids, const double* t, const double* p, long N, double tmin, double tmax, double* events_out, long long* count,
                     int* bad_polarity, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(N >= 0 && N < (1L << 31) && count && (N == 0 || (coord_ids && t && p && events_out)), "evd_event_filter: bad arguments");
    hipStream_t st = as_stream(stream);
    if (bad_polarity) EVD_HIP(hipMemsetAsync(bad_polarity, 0, sizeof(int), st));
    if (N == 0) { EVD_HIP(hipMemsetAsync(count, 0, sizeof(long long), st)); return EVD_OK; }
    const size_t need = evd_event_filter_workspace_bytes(N);
    if (!workspace || workspace_bytes < need) return fail(EVD_E_WORKSPACE, "evd_event_filter: workspace %zu < %zu bytes", workspace_bytes, need);
    size_t scan_bytes = 0;
    { const int* v = nullptr; int* vo = nullptr; (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, v, vo, (int)N); }
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    void* tmp = w; w += al256(scan_bytes);
    int* keep = (int*)w; w += al256((size_t)N * 4);
    int* pos = (int*)w; w += al256((size_t)N * 4);
    int* pminmax = (int*)w;
    const int init[2] = {0x7fffffff, (int)0x80000000};
    EVD_HIP(hipMemcpyAsync(pminmax, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ef_flags, dim3((unsigned)cdiv(N, 256L)), dim3(256), 0, st, t, p, N, tmin, tmax, keep, pminmax);
    size_t sb = scan_bytes;
    EVD_HIP(hipcub::DeviceScan::InclusiveSum(tmp, sb, (const int*)keep, pos, (int)N, st));
    hipLaunchKernelGGL(k_ef_write, dim3((unsigned)cdiv(N, 256L)), dim3(256), 0, st, coord_ids, t, p, N, (const int*)keep, (const int*)pos, (const int*)pminmax, events_out,
                       count, bad_polarity);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

size_t evd_event_color_map_workspace_bytes(long n_coords) { return n_coords < 0 ? 0 : al256((size_t)(n_coords > 0 ? n_coords : 1) * 4) + al256((size_t)(n_coords > 0 ? n_coords : 1)) + 512; }

int evd_event_color_map(const double* id_to_coords, long n_coords, int h, int w, const float* inv_mapx, const float* inv_mapy, const long long* noev_coord_ids,
                        long n_noev, unsigned char* id_to_color_map, int* unmapped, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(n_coords >= 0 && h >= 1 && w >= 1 && (long)h * w < (1L << 31) && (n_coords == 0 || (id_to_coords && id_to_color_map)), "evd_event_color_map: bad arguments");
    EVD_REQUIRE((inv_mapx == nullptr) == (inv_mapy == nullptr), "evd_event_color_map: both inverse maps or none");
    hipStream_t st = as_stream(stream);
    if (unmapped) EVD_HIP(hipMemsetAsync(unmapped, 0, sizeof(int), st));
    if (n_coords == 0) return EVD_OK;
    if (!inv_mapx) {                               // integer coordinates: colour of the pixel itself (:215-218)
        hipLaunchKernelGGL(k_cm_int, dim3((unsigned)cdiv(n_coords, 256L)), dim3(256), 0, st, id_to_coords, n_coords, h, w, id_to_color_map);
        EVD_LAUNCH_CHECK();
        return EVD_OK;
    }
    const size_t need = evd_event_color_map_workspace_bytes(n_coords);
    if (!workspace || workspace_bytes < need) return fail(EVD_E_WORKSPACE, "evd_event_color_map: workspace %zu < %zu bytes", workspace_bytes, need);
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int* last_pixel = (int*)p; p += al256((size_t)n_coords * 4);
    unsigned char* is_noev = (unsigned char*)p;
    EVD_HIP(hipMemsetAsync(last_pixel, 0xff, (size_t)n_coords * 4, st));            // -1
    EVD_HIP(hipMemsetAsync(is_noev, 0, (size_t)n_coords, st));
    hipLaunchKernelGGL(k_cm_lookup, dim3((unsigned)cdiv((long)h * w, 256L)), dim3(256), 0, st, id_to_coords, n_coords, h, w, inv_mapx, inv_mapy, last_pixel);
    hipLaunchKernelGGL(k_cm_from_pixels, dim3((unsigned)cdiv(n_coords, 256L)), dim3(256), 0, st, (const int*)last_pixel, n_coords, w, id_to_color_map);
    if (unmapped) {
        if (n_noev > 0 && noev_coord_ids)
            hipLaunchKernelGGL(k_cm_mark_noev, dim3((unsigned)cdiv(n_noev, 256L)), dim3(256), 0, st, noev_coord_ids, n_noev, n_coords, is_noev);
        hipLaunchKernelGGL(k_cm_check, dim3((unsigned)cdiv(n_coords, 256L)), dim3(256), 0, st, (const unsigned char*)id_to_color_map, n_coords, (const unsigned char*)is_noev,
                           unmapped);
    }
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // extern "C"
