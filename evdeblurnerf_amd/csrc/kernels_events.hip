// Event preprocessing on the device ("next" row f-4): the per-pixel successor graph of the event stream, reference
// utils/events.py:72-120 compute_successor (a sequential numba loop over all N events on the CPU, run once per dataset).
// Integer work, HBM-bound: a stable radix sort of (pixel id, event index) pairs (hipCUB) groups every pixel's events in
// stream order; the successor of an event is then its right neighbour in the sorted order, the number of successors its
// distance to the segment end.  Results are bit-identical to the reference loop.
#include <hipcub/hipcub.hpp>

#include "evd_common.h"

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

}  // extern "C"
