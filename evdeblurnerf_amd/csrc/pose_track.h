// Pose of the event camera at a timestamp, evaluated on the device (include/evdnerf.h evd_pose_track).
// Reference: data/loader_events.py:133-148 interpolate_poses -> utils/data.py:34-62 (scipy Slerp + cubic interp1d), evaluated in
// float64 like scipy, with the reference's float32 roundings at the same places (the cast at :138, the bd_scale product at :140, the
// store of the recentred pose into the float32 array at utils/data.py:177).
#pragma once

#include "evd_common.h"

namespace evd {

struct PoseTrackDev {
    int n_keys;
    const double* key_t;
    const double* key_quat;
    const double* key_rotvec;
    const double* trans_coef;
    float bd_scale;
    int recenter;
    double rinv[12];
};

inline PoseTrackDev pose_track_dev(const evd_pose_track* t) {
    PoseTrackDev d;
    d.n_keys = t->n_keys;
    d.key_t = t->key_t;
    d.key_quat = t->key_quat;
    d.key_rotvec = t->key_rotvec;
    d.trans_coef = t->trans_coef;
    d.bd_scale = t->bd_scale;
    d.recenter = t->recenter;
    for (int i = 0; i < 12; ++i) d.rinv[i] = t->recenter_inv[i];
    return d;
}

inline const char* pose_track_invalid(const evd_pose_track* t) {
    if (!t) return "null track";
    if (t->n_keys < 4) return "a track needs >= 4 key poses (cubic spline)";
    if (!t->key_t || !t->key_quat || !t->key_rotvec || !t->trans_coef) return "null table in the track";
    return nullptr;
}

// c2w[12] = rows 0..2 of interpolate_poses(t)
__device__ inline void pose_at(const PoseTrackDev& trk, double t, float c2w[12]) {
    const int M = trk.n_keys;
    const double t0 = trk.key_t[0], t1 = trk.key_t[M - 1];
    t = t < t0 ? t0 : (t > t1 ? t1 : t);                        // np.clip (loader_events.py:178, utils/data.py:56)
    // scipy Slerp.__call__: ind = searchsorted(times, t, side='left') - 1, and 0 for t == times[0]
    int lo = 0, hi = M;                                         // first index with key_t[idx] >= t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (trk.key_t[mid] < t) lo = mid + 1; else hi = mid;
    }
    int ind = lo - 1;
    if (ind < 0) ind = 0;
    if (ind > M - 2) ind = M - 2;
    const double h = trk.key_t[ind + 1] - trk.key_t[ind];
    const double alpha = (t - trk.key_t[ind]) / h;
    // Rotation.from_rotvec(rotvec * alpha)
    const double rx = trk.key_rotvec[ind * 3] * alpha, ry = trk.key_rotvec[ind * 3 + 1] * alpha, rz = trk.key_rotvec[ind * 3 + 2] * alpha;
    const double ang = sqrt(rx * rx + ry * ry + rz * rz);
    double sc;
    if (ang <= 1e-3) {
        const double a2 = ang * ang;
        sc = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
    } else {
        sc = sin(ang * 0.5) / ang;
    }
    const double qx = rx * sc, qy = ry * sc, qz = rz * sc, qw = cos(ang * 0.5);
    // key rotation * increment (Hamilton product, scipy's compose_quat)
    const double px = trk.key_quat[ind * 4], py = trk.key_quat[ind * 4 + 1], pz = trk.key_quat[ind * 4 + 2], pw = trk.key_quat[ind * 4 + 3];
    const double x = pw * qx + px * qw + py * qz - pz * qy;
    const double y = pw * qy - px * qz + py * qw + pz * qx;
    const double z = pw * qz + px * qy - py * qx + pz * qw;
    const double w = pw * qw - px * qx - py * qy - pz * qz;
    // as_matrix
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w, xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    double R[9];
    R[0] = x2 - y2 - z2 + w2; R[1] = 2 * (xy - zw);       R[2] = 2 * (xz + yw);
    R[3] = 2 * (xy + zw);     R[4] = -x2 + y2 - z2 + w2;  R[5] = 2 * (yz - xw);
    R[6] = 2 * (xz - yw);     R[7] = 2 * (yz + xw);       R[8] = -x2 - y2 + z2 + w2;
    // cubic of the interval in u = alpha (Horner)
    const double* cf = trk.trans_coef + (long)ind * 12;
    double T[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) T[c] = cf[c] + alpha * (cf[3 + c] + alpha * (cf[6 + c] + alpha * cf[9 + c]));
    // loader_events.py:137-140: columns [r1, -r0, r2, t], float32, translation *= bd_scale (float32 product)
    float P[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        P[r * 4] = (float)R[r * 3 + 1];
        P[r * 4 + 1] = (float)(-R[r * 3]);
        P[r * 4 + 2] = (float)R[r * 3 + 2];
        P[r * 4 + 3] = __fmul_rn((float)T[r], trk.bd_scale);
    }
    if (!trk.recenter) {
#pragma unroll
        for (int i = 0; i < 12; ++i) c2w[i] = P[i];
        return;
    }
    // recenter_poses: inv(c2w) @ [P; 0 0 0 1] in float64, stored float32 (utils/data.py:176-177)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double acc = __dmul_rn(trk.rinv[r * 4], (double)P[c]);
            acc = __dadd_rn(acc, __dmul_rn(trk.rinv[r * 4 + 1], (double)P[4 + c]));
            acc = __dadd_rn(acc, __dmul_rn(trk.rinv[r * 4 + 2], (double)P[8 + c]));
            if (c == 3) acc = __dadd_rn(acc, trk.rinv[r * 4 + 3]);
            c2w[r * 4 + c] = (float)acc;
        }
    }
}

}  // namespace evd
