// Event preprocessing on the device ("next" row f-4): the per-pixel successor graph of the event stream, reference
// utils/events.py:72-120 compute_successor (a sequential numba loop over all N events on the CPU, run once per dataset).
// Integer work, HBM-bound: a stable radix sort of (pixel id, event index) pairs (hipCUB) groups every pixel's events in
// stream order; the successor of an event is then its right neighbour in the sorted order, the number of successors its
// distance to the segment end.  Results are bit-identical to the reference loop.
#include <hipcub/hipcub.hpp>

#include "evd_common.h"
#include "pose_track.h"

namespace evd {

__global__ void k_iota_i32(int* __restrict__ v, long n) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int)i;
}

__global__ void k_fill_i64(long long* __restrict__ v, long n, long long x) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) v[i] = x;
}

// seg_end[k] = index (in sorted order) of the last event of k's pixel segment: a reverse "max" scan would do; segments
// are found by a per-element search bounded by the segment length instead of a scan: every element looks only at its
// right neighbour, the segment end is propagated with one pass per power of two (pointer jumping, log2(N) passes).
__global__ void k_succ_init(const int* __restrict__ keys, long n, int* __restrict__ jump) {
    const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (k >= n) return;
    jump[k] = (k + 1 < n && keys[k + 1] == keys[k]) ? (int)(k + 1) : (int)k;     // right neighbour inside the segment, else itself
}

__global__ void k_succ_jump(const int* __restrict__ in, long n, int* __restrict__ out) {
    const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (k >= n) return;
    out[k] = in[in[k]];
}

__global__ void k_succ_write(const int* __restrict__ keys, const int* __restrict__ idx, const int* __restrict__ seg_end, long n,
                             long long* __restrict__ successor, int* __restrict__ num_successors,
                             long long* __restrict__ latest_seen, long long* __restrict__ first_seen) {
    const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int e = idx[k], end = seg_end[k];
    const bool last = end == k;
    successor[e] = last ? (long long)e : (long long)idx[k + 1];
    num_successors[e] = end - (int)k;
    if (k == 0 || keys[k - 1] != keys[k]) latest_seen[keys[k]] = e;      // first event of the pixel (events.py:112)
    if (last) first_seen[keys[k]] = e;                                     // last event of the pixel (:114-115)
}

// EventsDataset.sample_events (data/loader_events.py:259-304) for one event id per thread: the reference's gathers, gather_successor
// (utils/events.py:221-257), and get_rays_pix (utils/rays.py:25-36, the arithmetic of k_get_rays_pix) on the start and the end pose.
// TRACK: the two poses are interpolate_poses(timestamp) evaluated here (pose_track.h; the reference's scipy round trip :280-283);
// else rows of a per-event pose table.  An id outside [0, N), or a successor outside it in the single-hop branch (an event without
// successor carries -1: the reference would wrap to events[-1]), gives zero polarity sums, successor -1, the start pose twice, and sets
// the flag word like a coordinate mismatch does.
template <bool TRACK>
__global__ __launch_bounds__(256) void k_sample_events(const double* __restrict__ ev, long N, int ncol, const float* __restrict__ id_to_coords, long n_coords,
                                                       const unsigned char* __restrict__ cmap, const float* __restrict__ poses, const PoseTrackDev trk,
                                                       const long long* __restrict__ ids, const long long* __restrict__ hops, long n,
                                                       float k00, float k02, float k11, float k12, float halfpix,
                                                       float* __restrict__ rays_start, float* __restrict__ rays_end, float* __restrict__ pos_out,
                                                       float* __restrict__ neg_out, long long* __restrict__ coords_ids, unsigned char* __restrict__ cmap_out,
                                                       long long* __restrict__ succ_out, int* __restrict__ mismatch) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    const long long id = ids[i];
    // an id outside the event table, or an event whose coordinate id lies outside the coordinate tables (a corrupt column 0): nothing of
    // this event can be read
    const long long pix0 = (id >= 0 && id < N) ? (long long)ev[id * ncol] : -1;
    if (id < 0 || id >= N || pix0 < 0 || pix0 >= n_coords) {
        pos_out[i] = neg_out[i] = 0.f;
        coords_ids[i] = -1;
        if (succ_out) succ_out[i] = -1;
        if (cmap_out) cmap_out[i * 3] = cmap_out[i * 3 + 1] = cmap_out[i * 3 + 2] = 0;
        for (int k = 0; k < 6; ++k) rays_start[i * 6 + k] = rays_end[i * 6 + k] = 0.f;
        if (mismatch) atomicExch(mismatch, 1);
        return;
    }
    const double* row = ev + id * ncol;
    const long long pix = (long long)row[0];
    long long end;
    float pos = 0.f, neg = 0.f;
    bool invalid = false;
    if (!hops) {                                   // loader_events.py:272-276
        end = (long long)row[ncol - 1];
        if (end < 0 || end >= N) {
            invalid = true;
        } else {
            const double p = ev[end * ncol + ncol - 2];
            if (p > 0) pos = (float)p; else neg = (float)p;
        }
    } else {                                       // gather_successor: hops + 1 steps (h <= query_hops, events.py:242-243)
        end = id;
        const long long nh = hops[i];
        for (long long h = 0; h <= nh; ++h) {
            const long long nxt = (long long)ev[end * ncol + ncol - 1];
            if (nxt < 0 || nxt >= N) { invalid = true; break; }
            end = nxt;
            const int p = (int)ev[end * ncol + ncol - 2];      // polarities as int32 (loader_events.py:271)
            if (p > 0) pos += (float)p;
            if (p < 0) neg += (float)p;
        }
    }
    if (invalid) {
        end = -1;
        pos = neg = 0.f;
        if (mismatch && !hops) atomicExch(mismatch, 1);        // the multi-hop branch returns -1 / 0 / 0 by the reference's own rule (:252-255)
    }
    pos_out[i] = pos;
    neg_out[i] = neg;
    coords_ids[i] = pix;
    if (succ_out) succ_out[i] = end;
    if (cmap_out) {
#pragma unroll
        for (int c = 0; c < 3; ++c) cmap_out[i * 3 + c] = cmap[pix * 3 + c];
    }
    const float cx = id_to_coords[pix * 2], cy = id_to_coords[pix * 2 + 1];
    const float d0 = (cx + (halfpix - k02)) / k00, d1 = -(cy + (halfpix - k12)) / k11, d2 = -1.f;
    const long long e2 = end < 0 ? id : end;       // an invalid chain: the start pose twice (the reference would index events[-1])
    if (mismatch && end >= 0 && (long long)ev[end * ncol] != pix) atomicExch(mismatch, 1);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const long long e = which ? e2 : id;
        float pose[12];
        const float* c2w;
        if (TRACK) {
            const double te = ev[e * ncol + ncol - 3];
            if (mismatch && !(te == te && te - te == 0.0)) atomicExch(mismatch, 1);      // a non-finite timestamp gives NaN poses: flagged
            pose_at(trk, te, pose);
            c2w = pose;
        } else {
            c2w = poses + e * 12;
        }
        float* out = (which ? rays_end : rays_start) + i * 6;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            out[r * 2] = c2w[r * 4 + 3];
            out[r * 2 + 1] = __fadd_rn(__fadd_rn(__fmul_rn(d0, c2w[r * 4]), __fmul_rn(d1, c2w[r * 4 + 1])), __fmul_rn(d2, c2w[r * 4 + 2]));
        }
    }
}

// interpolate_poses (data/loader_events.py:133-148) for one timestamp per thread
__global__ __launch_bounds__(256) void k_interpolate_poses(const PoseTrackDev trk, const double* __restrict__ t, long n, float* __restrict__ out) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    float c2w[12];
    pose_at(trk, t[i], c2w);
#pragma unroll
    for (int k = 0; k < 12; ++k) out[i * 12 + k] = c2w[k];
}

}  // namespace evd

using namespace evd;

extern "C" {

size_t evd_compute_successor_workspace_bytes(long N) {
    if (N < 0) return 0;
    size_t sort_bytes = 0;
    const int* ki = nullptr; int* ko = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, ki, ko, ki, ko, (int)(N > 0 ? N : 1));
    return ((sort_bytes + 255) & ~(size_t)255) + 5 * (((size_t)(N > 0 ? N : 1) * 4 + 255) & ~(size_t)255) + 512;
}

int evd_compute_successor(const int* pixel_ids, long N, long HW, long long* successor, int* num_successors,
                          long long* latest_seen, long long* first_seen, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(N >= 0 && HW >= 1 && N < (1L << 31) && latest_seen && first_seen && (N == 0 || (pixel_ids && successor && num_successors)),
                "evd_compute_successor: bad arguments");
    hipStream_t st = as_stream(stream);
    k_fill_i64<<<cdiv(HW, 256), 256, 0, st>>>(latest_seen, HW, -1);
    k_fill_i64<<<cdiv(HW, 256), 256, 0, st>>>(first_seen, HW, -1);
    EVD_LAUNCH_CHECK();
    if (N == 0) return EVD_OK;
    const size_t need = evd_compute_successor_workspace_bytes(N);
    if (!workspace || workspace_bytes < need) return fail(EVD_E_WORKSPACE, "evd_compute_successor: workspace %zu < %zu bytes", workspace_bytes, need);
    const size_t arr = ((size_t)N * 4 + 255) & ~(size_t)255;
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int* iota = (int*)w; w += arr;
    int* keys = (int*)w; w += arr;
    int* idx = (int*)w; w += arr;
    int* ja = (int*)w; w += arr;
    int* jb = (int*)w; w += arr;
    size_t sort_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, pixel_ids, keys, iota, idx, (int)N);
    k_iota_i32<<<cdiv(N, 256), 256, 0, st>>>(iota, N);
    int bits = 1;
    while (bits < 31 && (1L << bits) < HW) ++bits;
    EVD_HIP(hipcub::DeviceRadixSort::SortPairs(w, sort_bytes, pixel_ids, keys, iota, idx, (int)N, 0, bits, st));   // stable: stream order inside a pixel
    k_succ_init<<<cdiv(N, 256), 256, 0, st>>>(keys, N, ja);
    for (long span = 1; span < N; span <<= 1) {          // pointer jumping to the segment end
        k_succ_jump<<<cdiv(N, 256), 256, 0, st>>>(ja, N, jb);
        int* t = ja; ja = jb; jb = t;
    }
    k_succ_write<<<cdiv(N, 256), 256, 0, st>>>(keys, idx, ja, N, successor, num_successors, latest_seen, first_seen);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

static int sample_events_launch(const char* who, const double* events, long N, int ncol, const float* id_to_coords, long n_coords, const unsigned char* id_to_color_map,
                                const float* poses, const evd_pose_track* track, const long long* events_ids, const long long* hops, long n,
                                const float* K, int add_halfpix, float* rays_start, float* rays_end, float* pos_cumsum, float* neg_cumsum,
                                long long* coords_ids, unsigned char* color_map, long long* successor, int* mismatch, void* stream) {
    EVD_REQUIRE(n >= 0 && N >= 0 && n_coords >= 0 && ncol >= 4 && K, "%s: bad arguments (the event table has >= 4 columns: id, .., t, p, successor)", who);
    if (track) {
        const char* why = pose_track_invalid(track);
        EVD_REQUIRE(!why, "%s: %s", who, why);
    }
    if (n == 0) return EVD_OK;
    EVD_REQUIRE(events && id_to_coords && (poses || track) && events_ids && rays_start && rays_end && pos_cumsum && neg_cumsum && coords_ids,
                "%s: null argument", who);
    EVD_REQUIRE(!color_map || id_to_color_map, "%s: a colour map output needs id_to_color_map", who);
    hipStream_t st = as_stream(stream);
    if (mismatch) EVD_HIP(hipMemsetAsync(mismatch, 0, sizeof(int), st));
    const float hp = add_halfpix ? 0.5f : 0.f;
    if (track)
        hipLaunchKernelGGL(k_sample_events<true>, dim3((unsigned)cdiv(n, 256L)), dim3(256), 0, st, events, N, ncol, id_to_coords, n_coords, id_to_color_map,
                           (const float*)nullptr, pose_track_dev(track), events_ids, hops, n, K[0], K[2], K[4], K[5], hp, rays_start, rays_end,
                           pos_cumsum, neg_cumsum, coords_ids, color_map, successor, mismatch);
    else
        hipLaunchKernelGGL(k_sample_events<false>, dim3((unsigned)cdiv(n, 256L)), dim3(256), 0, st, events, N, ncol, id_to_coords, n_coords, id_to_color_map,
                           poses, PoseTrackDev{}, events_ids, hops, n, K[0], K[2], K[4], K[5], hp, rays_start, rays_end, pos_cumsum, neg_cumsum,
                           coords_ids, color_map, successor, mismatch);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_sample_events(const double* events, long N, int ncol, const float* id_to_coords, long n_coords, const unsigned char* id_to_color_map,
                      const float* poses, const long long* events_ids, const long long* hops, long n, const float* K, int add_halfpix,
                      float* rays_start, float* rays_end, float* pos_cumsum, float* neg_cumsum, long long* coords_ids,
                      unsigned char* color_map, long long* successor, int* mismatch, void* stream) {
    return sample_events_launch("evd_sample_events", events, N, ncol, id_to_coords, n_coords, id_to_color_map, poses, nullptr, events_ids, hops, n, K, add_halfpix,
                                rays_start, rays_end, pos_cumsum, neg_cumsum, coords_ids, color_map, successor, mismatch, stream);
}

int evd_sample_events_track(const double* events, long N, int ncol, const float* id_to_coords, long n_coords, const unsigned char* id_to_color_map,
                            const evd_pose_track* track, const long long* events_ids, const long long* hops, long n, const float* K,
                            int add_halfpix, float* rays_start, float* rays_end, float* pos_cumsum, float* neg_cumsum, long long* coords_ids,
                            unsigned char* color_map, long long* successor, int* mismatch, void* stream) {
    EVD_REQUIRE(track, "evd_sample_events_track: null track");
    return sample_events_launch("evd_sample_events_track", events, N, ncol, id_to_coords, n_coords, id_to_color_map, nullptr, track, events_ids, hops, n, K,
                                add_halfpix, rays_start, rays_end, pos_cumsum, neg_cumsum, coords_ids, color_map, successor, mismatch, stream);
}

int evd_interpolate_poses(const evd_pose_track* track, const double* t, long n, float* poses, void* stream) {
    const char* why = pose_track_invalid(track);
    EVD_REQUIRE(!why, "evd_interpolate_poses: %s", why);
    EVD_REQUIRE(n >= 0, "evd_interpolate_poses: n < 0");
    if (n == 0) return EVD_OK;
    EVD_REQUIRE(t && poses, "evd_interpolate_poses: null argument");
    hipLaunchKernelGGL(k_interpolate_poses, dim3((unsigned)cdiv(n, 256L)), dim3(256), 0, as_stream(stream), pose_track_dev(track), t, n, poses);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // extern "C"
