/*
 * evd_oracle.c -- CPU restatement of the EvDeblurNeRF renderer + loss path.
 *
 * TEST INFRASTRUCTURE ONLY (see evd_oracle.h). Parity pinned by tests/golden/G*.npz,
 * which tools/gen_golden.py produced by running the imported reference.
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp; unfused mul/add like torch's
 * elementwise kernels). Citations are reference paths (file:line).
 */
#include "evd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------------------------------------------------------
 * the once-per-dataset event tables: reference utils/events.py:11-69 (load_events_h5) and data/loader_events.py:186-236 */
typedef struct { unsigned char b[16]; long idx; } evo_crow;
static int evo_crow_cmp(const void* pa, const void* pb) {
    const evo_crow *a = (const evo_crow*)pa, *b = (const evo_crow*)pb;
    const int c = memcmp(a->b, b->b, 16);                      /* np.unique on a void dtype orders by the raw bytes (utils/misc.py:143-149) */
    if (c) return c;
    return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);   /* return_index: the first occurrence leads its group */
}

long evo_event_coord_ids(const float* x, const float* y, long N, int h, int w, long long* ev_ids, long long* noev_ids, double* id_to_coords, long* n_noev) {
    const long HW = (long)h * w;
    unsigned char* silent = (unsigned char*)malloc((size_t)HW);
    memset(silent, 1, (size_t)HW);
    for (long i = 0; i < N; ++i) {                              /* utils/events.py:39-41: np.round (half to even), int32, clip */
        long yy = (long)rintf(y[i]), xx = (long)rintf(x[i]);
        yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
        xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
        silent[yy * w + xx] = 0;
    }
    long nz = 0;
    for (long j = 0; j < HW; ++j) nz += silent[j];
    const long M = N + nz;
    evo_crow* rows = (evo_crow*)malloc(sizeof(evo_crow) * (size_t)(M > 0 ? M : 1));
    double* all = (double*)malloc(sizeof(double) * 2 * (size_t)(M > 0 ? M : 1));
    for (long i = 0; i < N; ++i) { all[2 * i] = (double)x[i]; all[2 * i + 1] = (double)y[i]; }      /* float32 + int64 -> float64 rows (:52) */
    long k = N;
    for (long j = 0; j < HW; ++j)                               /* np.where order, [:, ::-1] -> (x, y) (:42) */
        if (silent[j]) { all[2 * k] = (double)(j % w); all[2 * k + 1] = (double)(j / w); ++k; }
    for (long i = 0; i < M; ++i) { memcpy(rows[i].b, all + 2 * i, 16); rows[i].idx = i; }
    qsort(rows, (size_t)M, sizeof(evo_crow), evo_crow_cmp);
    long uid = -1;
    for (long q = 0; q < M; ++q) {
        if (q == 0 || memcmp(rows[q].b, rows[q - 1].b, 16) != 0) {
            ++uid;
            id_to_coords[2 * uid] = all[2 * rows[q].idx];
            id_to_coords[2 * uid + 1] = all[2 * rows[q].idx + 1];
        }
        if (rows[q].idx < N) ev_ids[rows[q].idx] = uid; else noev_ids[rows[q].idx - N] = uid;
    }
    *n_noev = nz;
    free(rows); free(all); free(silent);
    return uid + 1;
}

long evo_event_filter(const long long* ids, const double* t, const double* p, long N, double tmin, double tmax, double* events_out) {
    long n = 0;
    double pmin = 1e300, pmax = -1e300;
    for (long i = 0; i < N; ++i)
        if (t[i] >= tmin && t[i] <= tmax) {                     /* loader_events.py:191 */
            events_out[3 * n] = (double)ids[i]; events_out[3 * n + 1] = t[i]; events_out[3 * n + 2] = p[i];
            if (p[i] < pmin) pmin = p[i];
            if (p[i] > pmax) pmax = p[i];
            ++n;
        }
    if (n > 0 && pmin == 0.0) {                                 /* :203-205 */
        for (long i = 0; i < n; ++i) if (events_out[3 * i + 2] == 0.0) events_out[3 * i + 2] = -1.0;
        pmin = -1.0;
    }
    if (n > 0 && !(pmax == 1.0 && pmin == -1.0)) return -n - 1; /* :206 */
    return n;
}

static int evo_bayer(int j, int i) { return (j % 2 == 0) ? (i % 2 == 0 ? 0 : 1) : (i % 2 == 0 ? 1 : 2); }        /* :209-213 r g / g b */

long evo_event_color_map(const double* id_to_coords, long Nc, int h, int w, const float* inv_mapx, const float* inv_mapy, const long long* noev_ids, long n_noev,
                         unsigned char* cmap) {
    memset(cmap, 0, (size_t)Nc * 3);
    if (!inv_mapx) {                                            /* :215-218 */
        for (long id = 0; id < Nc; ++id) {
            const long xi = (long)id_to_coords[2 * id], yi = (long)id_to_coords[2 * id + 1];
            if (xi >= 0 && xi < w && yi >= 0 && yi < h) cmap[3 * id + evo_bayer((int)yi, (int)xi)] = 1;
        }
        return 0;
    }
    for (int j = 0; j < h; ++j)                                 /* :226-230: the dict lookup by value, later pixels overwrite */
        for (int i = 0; i < w; ++i) {
            const double qx = (double)inv_mapx[(long)j * w + i], qy = (double)inv_mapy[(long)j * w + i];
            for (long id = 0; id < Nc; ++id)
                if (id_to_coords[2 * id] == qx && id_to_coords[2 * id + 1] == qy) {
                    cmap[3 * id] = cmap[3 * id + 1] = cmap[3 * id + 2] = 0;
                    cmap[3 * id + evo_bayer(j, i)] = 1;
                }
        }
    unsigned char* noev = (unsigned char*)calloc((size_t)(Nc > 0 ? Nc : 1), 1);
    for (long q = 0; q < n_noev; ++q) noev[noev_ids[q]] = 1;
    long badn = 0;
    for (long id = 0; id < Nc; ++id)                            /* :231-234 */
        if (!noev[id] && cmap[3 * id] + cmap[3 * id + 1] + cmap[3 * id + 2] != 1) ++badn;
    free(noev);
    return badn;
}

int evo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ activations */
static inline float act(int code, float x) {
    switch (code) {
    case EVO_ACT_RELU: return x > 0.f ? x : 0.f;
    case EVO_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case EVO_ACT_EXP: return expf(x);
    case EVO_ACT_SIGMOID1: return 1.002f / (expf(-x) + 1.f) - 0.001f;  /* nerf.py:32 */
    case EVO_ACT_SOFTPLUS: {                                           /* nn.Softplus()(x-1), nerf.py:33 */
        float y = x - 1.f;
        return y > 20.f ? y : log1pf(expf(y));
    }
    case EVO_ACT_TANH: return tanhf(x);
    default: return x;
    }
}

/* ------------------------------------------------------------------ Embedder
 * networks/embedding.py:88-98: cat([x] + [sin(x*2^k), cos(x*2^k) for k<L]); freq_bands :78 */
void evo_embed(const float* x, long n, int dim, int L, float* out) {
    const int od = dim * (1 + 2 * L);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        const float* xi = x + i * dim;
        float* o = out + i * od;
        for (int c = 0; c < dim; ++c) o[c] = xi[c];
        float freq = 1.f;
        for (int k = 0; k < L; ++k) {
            for (int c = 0; c < dim; ++c) {
                float a = xi[c] * freq;
                o[dim + (2 * k) * dim + c] = sinf(a);
                o[dim + (2 * k + 1) * dim + c] = cosf(a);
            }
            freq *= 2.f;
        }
    }
}

/* y[fo] = b + W[fo,:] . x, k ascending.  Wt is the [in,out] transpose so the inner loop vectorises
 * while every output keeps the same k-ascending summation order. */
static void linear_t(const float* Wt, const float* b, int in, int out, const float* x, float* y) {
    if (b) memcpy(y, b, sizeof(float) * out);
    else memset(y, 0, sizeof(float) * out);
    for (int k = 0; k < in; ++k) {
        const float xv = x[k];
        const float* w = Wt + (long)k * out;
        for (int j = 0; j < out; ++j) y[j] += xv * w[j];
    }
}

static float* transpose_w(const float* W, int out, int in) {
    float* t = (float*)malloc(sizeof(float) * (size_t)out * in);
    for (int o = 0; o < out; ++o)
        for (int k = 0; k < in; ++k) t[(long)k * out + o] = W[(long)o * in + k];
    return t;
}

/* Blocked variant of linear_t: NB samples share every weight row load (cache/register reuse, the CPU
 * baseline's GEMM); each output still sums k ascending, so results are bitwise those of linear_t. */
#define EVO_NB 8
typedef float evo_v8 __attribute__((vector_size(32), aligned(4)));
static void linear_blk(const float* Wt, const float* b, int in, int out, const float* x, int xs, float* y, int ys, int nb) {
    if (nb == EVO_NB && (out & 7) == 0) {
        /* 8 samples x 8 outputs register tile: the weight vector of row k is loaded once for 8 samples */
        for (int jb = 0; jb < out; jb += 8) {
            evo_v8 a0, a1, a2, a3, a4, a5, a6, a7;
            if (b) a0 = *(const evo_v8*)(b + jb); else a0 = (evo_v8){0, 0, 0, 0, 0, 0, 0, 0};
            a1 = a2 = a3 = a4 = a5 = a6 = a7 = a0;
            for (int k = 0; k < in; ++k) {
                const evo_v8 w = *(const evo_v8*)(Wt + (long)k * out + jb);
                a0 += w * x[k];          a1 += w * x[xs + k];     a2 += w * x[2 * xs + k]; a3 += w * x[3 * xs + k];
                a4 += w * x[4 * xs + k]; a5 += w * x[5 * xs + k]; a6 += w * x[6 * xs + k]; a7 += w * x[7 * xs + k];
            }
            *(evo_v8*)(y + jb) = a0;          *(evo_v8*)(y + ys + jb) = a1;     *(evo_v8*)(y + 2 * ys + jb) = a2;
            *(evo_v8*)(y + 3 * ys + jb) = a3; *(evo_v8*)(y + 4 * ys + jb) = a4; *(evo_v8*)(y + 5 * ys + jb) = a5;
            *(evo_v8*)(y + 6 * ys + jb) = a6; *(evo_v8*)(y + 7 * ys + jb) = a7;
        }
        return;
    }
    for (int s = 0; s < nb; ++s) linear_t(Wt, b, in, out, x + (long)s * xs, y + (long)s * ys);
}

/* ------------------------------------------------------------------ NeRF.eval
 * networks/nerf.py:131-162.  emb = [n, input_ch + input_ch_views]; raw = cat([rgb, alpha]) :157 */
void evo_nerf_mlp(const evo_nerf* net, const float* emb, long n, float* raw, float* feat_after, float* feat_before) {
    const int D = net->D, W = net->W, ic = net->input_ch, icv = net->input_ch_views;
    float* wt[EVO_MAX_LAYERS];
    int fin[EVO_MAX_LAYERS];
    for (int i = 0; i < D; ++i) {
        fin[i] = (i == 0) ? ic : ((i - 1) == net->skip ? W + ic : W);
        wt[i] = transpose_w(net->pts_w[i], W, fin[i]);
    }
    float *wv = NULL, *wf = NULL, *wa = NULL, *wr = NULL, *wo = NULL;
    if (net->use_viewdirs) {
        wv = transpose_w(net->views_w, W / 2, W + icv);
        wf = transpose_w(net->feature_w, W, W);
        wa = transpose_w(net->alpha_w, 1, W);
        wr = transpose_w(net->rgb_w, 3, W / 2);
    } else {
        wo = transpose_w(net->output_w, net->output_ch, W);
    }
    const int ein = ic + icv;
    const int hs = W + ic + icv;     /* row stride of the activation blocks */
    const long nblk = (n + EVO_NB - 1) / EVO_NB;
#pragma omp parallel
    {
        float* h = (float*)malloc(sizeof(float) * hs * EVO_NB);
        float* h2 = (float*)malloc(sizeof(float) * hs * EVO_NB);
#pragma omp for schedule(static)
        for (long blk = 0; blk < nblk; ++blk) {
            const long s0 = blk * EVO_NB;
            const int nb = (int)((n - s0) < EVO_NB ? (n - s0) : EVO_NB);
            const float* x = emb + s0 * ein;
            const float* in = x;
            int is = ein;
            for (int i = 0; i < D; ++i) {
                float* o = (i == net->skip) ? h2 + ic : h2;          /* cat([input_pts, h]) nerf.py:137-138 */
                linear_blk(wt[i], net->pts_b[i], fin[i], W, in, is, o, hs, nb);
                for (int s = 0; s < nb; ++s) {
                    float* os = o + (long)s * hs;
                    for (int j = 0; j < W; ++j) os[j] = os[j] > 0.f ? os[j] : 0.f;
                    if (i == net->skip) memcpy(h2 + (long)s * hs, x + (long)s * ein, sizeof(float) * ic);
                }
                float* tmp = h; h = h2; h2 = tmp;
                in = h;
                is = hs;
            }
            /* h = layer-D activations (W wide unless D-1 == skip) */
            if (feat_before) for (int s = 0; s < nb; ++s) memcpy(feat_before + (s0 + s) * W, h + (long)s * hs, sizeof(float) * W);
            if (net->use_viewdirs) {
                float alpha[EVO_NB];
                linear_blk(wa, net->alpha_b, W, 1, h, hs, alpha, 1, nb);
                linear_blk(wf, net->feature_b, W, W, h, hs, h2, hs, nb);
                for (int s = 0; s < nb; ++s) {
                    if (feat_after) memcpy(feat_after + (s0 + s) * W, h2 + (long)s * hs, sizeof(float) * W);
                    memcpy(h2 + (long)s * hs + W, x + (long)s * ein + ic, sizeof(float) * icv);   /* cat([feature, input_views]) :147 */
                }
                linear_blk(wv, net->views_b, W + icv, W / 2, h2, hs, h, hs, nb);
                for (int s = 0; s < nb; ++s) {
                    float* hh = h + (long)s * hs;
                    for (int j = 0; j < W / 2; ++j) hh[j] = hh[j] > 0.f ? hh[j] : 0.f;
                }
                float rgb[EVO_NB * 3];
                linear_blk(wr, net->rgb_b, W / 2, 3, h, hs, rgb, 3, nb);
                for (int s = 0; s < nb; ++s) {
                    float* r4 = raw + (s0 + s) * 4;
                    r4[0] = rgb[s * 3]; r4[1] = rgb[s * 3 + 1]; r4[2] = rgb[s * 3 + 2]; r4[3] = alpha[s];
                }
            } else {
                /* outputs = output_linear(h), nerf.py:158-160; output_ch is 5 with importance sampling (renderer.py:46) and raw2outputs
                 * reads channels 0..2 (rgb) and 3 (sigma) only (nerf.py:89,99): raw keeps those four */
                float ob[EVO_NB * 8];
                const int oc = net->output_ch;
                linear_blk(wo, net->output_b, W, oc, h, hs, ob, oc, nb);
                for (int s = 0; s < nb; ++s)
                    for (int c = 0; c < 4; ++c) raw[(s0 + s) * 4 + c] = c < oc ? ob[s * oc + c] : 0.f;
            }
        }
        free(h); free(h2);
    }
    for (int i = 0; i < D; ++i) free(wt[i]);
    free(wv); free(wf); free(wa); free(wr); free(wo);
}

/* ------------------------------------------------------------------ raw2outputs
 * networks/nerf.py:74-129 (sigma_ch = 3, rgb_ch0 = 0, n_rgb = 3) and
 * networks/pdrf/voxnerf.py:153-201 (sigma_ch = 0, rgb_ch0 = 1, n_rgb = C-1).
 * Density from the first S-1 samples, last alpha forced to 1 (:106,113-114); T = exclusive cumprod
 * of (1-alpha); the "+1e-10" of the reference is a no-op in float32.
 * rmnear_thresh <= 0 disables the eval-time near-plane mask (:107-111). */
void evo_composite(const float* raw, const float* z, const float* rays_d, long R, int S, int C,
                   int sigma_ch, int rgb_ch0, int n_rgb, int rgb_act, int sigma_act, int white_bkgd,
                   float rmnear_thresh, const float* noise,
                   float* out_map, float* density, float* acc, float* weights, float* depth,
                   const float* feature, int F, float* fmap) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r) {
        const float* rw = raw + r * (long)S * C;
        const float* zz = z + r * (long)S;
        const float* d = rays_d + r * 3;
        const float norm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        float T = 1.f, a_sum = 0.f, d_sum = 0.f;
        float csum[32];
        for (int c = 0; c < n_rgb && c < 32; ++c) csum[c] = 0.f;
        float* fm = fmap ? fmap + r * F : NULL;
        if (fm) for (int f = 0; f < F; ++f) fm[f] = 0.f;
        for (int i = 0; i < S; ++i) {
            float alpha;
            if (i < S - 1) {
                float dist = (zz[i + 1] - zz[i]) * norm;
                float sraw = rw[(long)i * C + sigma_ch] + (noise ? noise[r * (long)(S - 1) + i] : 0.f);
                float dens = act(sigma_act, sraw);
                if (rmnear_thresh > 0.f) dens = (zz[i + 1] > rmnear_thresh ? 1.f : 0.f) * dens;
                if (density) density[r * (long)(S - 1) + i] = dens;
                alpha = -expf(-dens * dist) + 1.f;
            } else {
                alpha = 1.f;
            }
            const float w = alpha * T;
            T = T * (-alpha + 1.f);
            if (weights) weights[r * (long)S + i] = w;
            a_sum += w;
            d_sum += w * zz[i];
            for (int c = 0; c < n_rgb && c < 32; ++c) csum[c] += w * act(rgb_act, rw[(long)i * C + rgb_ch0 + c]);
            if (fm) {
                const float* ff = feature + (r * (long)S + i) * F;
                for (int f = 0; f < F; ++f) fm[f] += w * ff[f];
            }
        }
        if (out_map)
            for (int c = 0; c < n_rgb && c < 32; ++c)
                out_map[r * n_rgb + c] = white_bkgd ? csum[c] + (1.f - a_sum) : csum[c];
        if (acc) acc[r] = a_sum;
        if (depth) depth[r] = d_sum;
    }
}

/* torch.linspace (ATen RangeFactories): symmetric fill, step = (end-start)/(steps-1) */
void evo_linspace(float start, float end, int steps, float* out) {
    if (steps == 1) { out[0] = start; return; }
    const float step = (end - start) / (float)(steps - 1);
    const int half = steps / 2;
    for (int i = 0; i < steps; ++i)
        out[i] = (i < half) ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

/* ------------------------------------------------------------------ sample_pdf
 * utils/rays.py:149-193.  bins [R,nb], w [R,nb-1]; det => u = linspace(0,1,N), else u given.
 * searchsorted(right=True) :176; below/above clamps :177-178; denom guard :187. */
void evo_sample_pdf(const float* bins, const float* w, long R, int nb, int N, int det, const float* u, float* out) {
    float* ulin = (float*)malloc(sizeof(float) * N);
    evo_linspace(0.f, 1.f, N, ulin);
#pragma omp parallel
    {
        float* cdf = (float*)malloc(sizeof(float) * nb);
#pragma omp for schedule(static)
        for (long r = 0; r < R; ++r) {
            const float* wr = w + r * (long)(nb - 1);
            const float* br = bins + r * (long)nb;
            /* torch.sum's float summation order is backend/ISA specific; the oracle uses the correctly rounded
             * sum (double accumulation is exact for <= 2^29 dynamic range), which any parallel order reproduces */
            double dsum = 0.0;
            for (int i = 0; i < nb - 1; ++i) dsum += (double)(wr[i] + 1e-5f);
            const float sum = (float)dsum;
            cdf[0] = 0.f;
            /* torch.cumsum on the CPU backend accumulates float inputs in double (acc_type<float,false>) and
             * rounds every prefix to float; the sums are exact in double, so any scan order gives these bits */
            double c = 0.0;
            for (int i = 0; i < nb - 1; ++i) { c += (double)((wr[i] + 1e-5f) / sum); cdf[i + 1] = (float)c; }
            for (int j = 0; j < N; ++j) {
                const float uu = det ? ulin[j] : u[r * (long)N + j];
                int lo = 0, hi = nb;                 /* first index with cdf[idx] > uu */
                while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] <= uu) lo = mid + 1; else hi = mid; }
                int below = lo - 1 > 0 ? lo - 1 : 0;
                int above = lo < nb - 1 ? lo : nb - 1;
                float denom = cdf[above] - cdf[below];
                if (denom < 1e-5f) denom = 1.f;
                float t = (uu - cdf[below]) / denom;
                out[r * (long)N + j] = br[below] + t * (br[above] - br[below]);
            }
        }
        free(cdf);
    }
    free(ulin);
}

/* ------------------------------------------------------------------ rays
 * utils/rays.py:8-22 get_rays; output [H,W,3] each. */
void evo_get_rays(int H, int W, const float* K, const float* c2w, int add_halfpix, float* rays_o, float* rays_d) {
    const float hp = add_halfpix ? 0.5f : 0.f;
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            float dir[3] = {((float)i + (hp - K[2])) / K[0], -((float)j + (hp - K[5])) / K[4], -1.f};
            float* o = rays_o + ((long)j * W + i) * 3;
            float* d = rays_d + ((long)j * W + i) * 3;
            for (int r = 0; r < 3; ++r) {
                d[r] = dir[0] * c2w[r * 4 + 0] + dir[1] * c2w[r * 4 + 1] + dir[2] * c2w[r * 4 + 2];
                o[r] = c2w[r * 4 + 3];
            }
        }
}

/* utils/rays.py:25-36 get_rays_pix; coords [n,2] (x,y), c2ws [n,3,4] */
void evo_get_rays_pix(const float* coords, const float* K, const float* c2ws, long n, int add_halfpix,
                      float* rays_o, float* rays_d) {
    const float hp = add_halfpix ? 0.5f : 0.f;
    for (long p = 0; p < n; ++p) {
        const float* c2w = c2ws + p * 12;
        float dir[3] = {(coords[p * 2] + (hp - K[2])) / K[0], -(coords[p * 2 + 1] + (hp - K[5])) / K[4], -1.f};
        for (int r = 0; r < 3; ++r) {
            rays_d[p * 3 + r] = dir[0] * c2w[r * 4 + 0] + dir[1] * c2w[r * 4 + 1] + dir[2] * c2w[r * 4 + 2];
            rays_o[p * 3 + r] = c2w[r * 4 + 3];
        }
    }
}

/* utils/rays.py:104-145 get_ndc_rays.  The W/(2 focal) scalars are Python doubles in the
 * reference (K is a numpy array there), rounded to float32 when they meet the tensor. */
void evo_ndc_rays(int H, int W, float focal, float near, const float* o, const float* d, long n, float* oo, float* od) {
    const float cw = (float)(-1.0 / ((double)W / (2.0 * (double)focal)));
    const float ch = (float)(-1.0 / ((double)H / (2.0 * (double)focal)));
    const float two_near = (float)(2.0 * (double)near);
    for (long i = 0; i < n; ++i) {
        const float dx = d[i * 3], dy = d[i * 3 + 1], dz = d[i * 3 + 2];
        const float t = -(near + o[i * 3 + 2]) / dz;
        const float ox = o[i * 3] + t * dx, oy = o[i * 3 + 1] + t * dy, oz = o[i * 3 + 2] + t * dz;
        const float ox_oz = ox / oz, oy_oz = oy / oz;
        const float o2 = 1.f + two_near / oz;
        oo[i * 3] = cw * ox_oz;
        oo[i * 3 + 1] = ch * oy_oz;
        oo[i * 3 + 2] = o2;
        od[i * 3] = cw * (dx / dz - ox_oz);
        od[i * 3 + 1] = ch * (dy / dz - oy_oz);
        od[i * 3 + 2] = 1.f - o2;
    }
}

/* NeRFAll.render, ray packing part: networks/renderer.py:423-446.  rays [R,3,2] (o|d interleaved in the
 * last axis).  ray_batch [R, 8 or 11] = o, d, near, far, [viewdirs]; viewdirs normalised BEFORE ndc. */
void evo_ray_batch(const evo_render_cfg* cfg, const float* rays, long R, float* ray_batch, int* ncol) {
    const int nc = cfg->use_viewdirs ? 11 : 8;
    if (ncol) *ncol = nc;
    float* o = (float*)malloc(sizeof(float) * 3 * (R > 0 ? R : 1));
    float* d = (float*)malloc(sizeof(float) * 3 * (R > 0 ? R : 1));
    for (long i = 0; i < R; ++i)
        for (int c = 0; c < 3; ++c) { o[i * 3 + c] = rays[i * 6 + c * 2]; d[i * 3 + c] = rays[i * 6 + c * 2 + 1]; }
    for (long i = 0; i < R; ++i) {
        float* rb = ray_batch + i * nc;
        if (cfg->use_viewdirs) {
            const float nrm = sqrtf(d[i * 3] * d[i * 3] + d[i * 3 + 1] * d[i * 3 + 1] + d[i * 3 + 2] * d[i * 3 + 2]);
            for (int c = 0; c < 3; ++c) rb[8 + c] = d[i * 3 + c] / nrm;
        }
        rb[6] = cfg->near; rb[7] = cfg->far;
    }
    if (cfg->ndc) {
        float* o2 = (float*)malloc(sizeof(float) * 3 * (R > 0 ? R : 1));
        float* d2 = (float*)malloc(sizeof(float) * 3 * (R > 0 ? R : 1));
        evo_ndc_rays(cfg->H, cfg->W, cfg->focal, 1.f, o, d, R, o2, d2);
        free(o); free(d); o = o2; d = d2;
    }
    for (long i = 0; i < R; ++i)
        for (int c = 0; c < 3; ++c) { ray_batch[i * nc + c] = o[i * 3 + c]; ray_batch[i * nc + 3 + c] = d[i * 3 + c]; }
    free(o); free(d);
}

/* z stratification: networks/renderer.py:163-178 */
static void make_z(const evo_render_cfg* cfg, const float* ray_batch, int nc, long R, const float* t_rand, float* z) {
    const int S = cfg->N_samples;
    float* tv = (float*)malloc(sizeof(float) * S);
    evo_linspace(0.f, 1.f, S, tv);
    for (long r = 0; r < R; ++r) {
        const float near = ray_batch[r * nc + 6], far = ray_batch[r * nc + 7];
        float* zr = z + r * (long)S;
        for (int i = 0; i < S; ++i)
            zr[i] = cfg->lindisp ? 1.f / (1.f / near * (1.f - tv[i]) + 1.f / far * tv[i])
                                 : near * (1.f - tv[i]) + far * tv[i];
        if (cfg->perturb > 0.f) {
            float* tmp = (float*)malloc(sizeof(float) * S);
            for (int i = 0; i < S; ++i) {
                float upper = (i < S - 1) ? .5f * (zr[i + 1] + zr[i]) : zr[S - 1];
                float lower = (i > 0) ? .5f * (zr[i] + zr[i - 1]) : zr[0];
                tmp[i] = lower + (upper - lower) * t_rand[r * (long)S + i];
            }
            memcpy(zr, tmp, sizeof(float) * S);
            free(tmp);
        }
    }
    free(tv);
}

static int cmp_float(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* one backbone pass of mode='nerf': NeRF.forward, networks/nerf.py:164-175 (+ mlpforward :46-72) */
static void nerf_pass(const evo_nerf* net, const evo_render_cfg* cfg, const float* ray_batch, int nc, long R, int S,
                      const float* z, float* rgb, float* depth, float* acc, float* weights, float* raw_out, float* feat_out) {
    const int L = cfg->multires, Lv = cfg->multires_views;
    const int ic = 3 * (1 + 2 * L), icv = cfg->use_viewdirs ? 3 * (1 + 2 * Lv) : 0;
    const long n = R * S;
    float* pts = (float*)malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
    float* emb = (float*)malloc(sizeof(float) * (size_t)(ic + icv) * (n > 0 ? n : 1));
    float* raw = raw_out ? raw_out : (float*)malloc(sizeof(float) * 4 * (n > 0 ? n : 1));
    float* rd = (float*)malloc(sizeof(float) * 3 * (R > 0 ? R : 1));
    for (long r = 0; r < R; ++r) {
        const float* rb = ray_batch + r * nc;
        for (int c = 0; c < 3; ++c) rd[r * 3 + c] = rb[3 + c];
        for (int i = 0; i < S; ++i)
            for (int c = 0; c < 3; ++c) pts[(r * S + i) * 3 + c] = rb[c] + rb[3 + c] * z[r * (long)S + i];  /* renderer.py:180 */
    }
    {
        float* pe = (float*)malloc(sizeof(float) * (size_t)ic * (n > 0 ? n : 1));
        evo_embed(pts, n, 3, L, pe);
        float* pev = NULL;
        if (icv) {
            float* vd = (float*)malloc(sizeof(float) * 3 * (R > 0 ? R : 1));
            for (long r = 0; r < R; ++r) for (int c = 0; c < 3; ++c) vd[r * 3 + c] = ray_batch[r * nc + 8 + c];
            pev = (float*)malloc(sizeof(float) * (size_t)icv * (R > 0 ? R : 1));
            evo_embed(vd, R, 3, Lv, pev);
            free(vd);
        }
        for (long s = 0; s < n; ++s) {
            memcpy(emb + s * (ic + icv), pe + s * ic, sizeof(float) * ic);
            if (icv) memcpy(emb + s * (ic + icv) + ic, pev + (s / S) * icv, sizeof(float) * icv);
        }
        free(pe); free(pev);
    }
    /* per-sample feature: feature_linear's output with the view branch (nerf.py:149-150); a use_viewdirs=False network only has the
     * "before_linear" one, the last hidden activations (nerf.py:141-142, 159) */
    if (net->use_viewdirs) evo_nerf_mlp(net, emb, n, raw, feat_out, NULL);
    else evo_nerf_mlp(net, emb, n, raw, NULL, feat_out);
    const float thr = (!cfg->is_train && net->rmnear > 0.f) ? (float)((double)net->rmnear / 128.0) : 0.f;
    evo_composite(raw, z, rd, R, S, 4, 3, 0, 3, net->rgb_act, net->sigma_act, cfg->white_bkgd, thr, NULL,
                  rgb, NULL, acc, weights, depth, NULL, 0, NULL);
    free(pts); free(emb); free(rd);
    if (!raw_out) free(raw);
}

static void zstd(const float* zs, long R, int N, float* out) {
    for (long r = 0; r < R; ++r) {
        double m = 0; for (int j = 0; j < N; ++j) m += zs[r * (long)N + j]; m /= N;
        double v = 0; for (int j = 0; j < N; ++j) { double dd = zs[r * (long)N + j] - m; v += dd * dd; }
        out[r] = (float)sqrt(v / N);
    }
}

/* NeRFAll.render + render_rays for mode='nerf': networks/renderer.py:399-466, 129-264 (else-branch :218-240). */
void evo_render_nerf(const evo_nerf* coarse, const evo_nerf* fine, const evo_render_cfg* cfg,
                     const float* rays, long R, const float* t_rand, const float* u, evo_render_out* out) {
    const int S = cfg->N_samples, Ni = cfg->N_importance, St = S + Ni;
    const long Rn = R > 0 ? R : 1;
    int nc;
    float* rb = (float*)malloc(sizeof(float) * 11 * Rn);
    evo_ray_batch(cfg, rays, R, rb, &nc);
    float* z = (float*)malloc(sizeof(float) * Rn * St);
    float* w0 = (float*)malloc(sizeof(float) * Rn * S);
    make_z(cfg, rb, nc, R, t_rand, z);
    if (Ni <= 0) {
        nerf_pass(coarse, cfg, rb, nc, R, S, z, out->rgb, out->depth, out->acc, w0, out->raw, out->feature);
        if (out->z_vals) memcpy(out->z_vals, z, sizeof(float) * R * S);
        if (out->weights) memcpy(out->weights, w0, sizeof(float) * R * S);
    } else {
        nerf_pass(coarse, cfg, rb, nc, R, S, z, out->rgb0, out->depth0, out->acc0, w0, NULL, NULL);
        if (out->z_vals0) memcpy(out->z_vals0, z, sizeof(float) * R * S);
        if (out->weights0) memcpy(out->weights0, w0, sizeof(float) * R * S);
        float* zmid = (float*)malloc(sizeof(float) * Rn * (S - 1));
        float* wmid = (float*)malloc(sizeof(float) * Rn * (S - 2));
        float* zs = (float*)malloc(sizeof(float) * Rn * Ni);
        for (long r = 0; r < R; ++r) {
            for (int i = 0; i < S - 1; ++i) zmid[r * (long)(S - 1) + i] = .5f * (z[r * (long)S + i + 1] + z[r * (long)S + i]);
            for (int i = 0; i < S - 2; ++i) wmid[r * (long)(S - 2) + i] = w0[r * (long)S + i + 1];   /* weights[...,1:-1] */
        }
        evo_sample_pdf(zmid, wmid, R, S - 1, Ni, cfg->perturb == 0.f, u, zs);
        if (out->z_std) zstd(zs, R, Ni, out->z_std);
        float* z2 = (float*)malloc(sizeof(float) * Rn * St);
        for (long r = 0; r < R; ++r) {
            memcpy(z2 + r * (long)St, z + r * (long)S, sizeof(float) * S);
            memcpy(z2 + r * (long)St + S, zs + r * (long)Ni, sizeof(float) * Ni);
            qsort(z2 + r * (long)St, St, sizeof(float), cmp_float);                              /* renderer.py:234 */
        }
        float* w1 = (float*)malloc(sizeof(float) * Rn * St);
        nerf_pass(fine, cfg, rb, nc, R, St, z2, out->rgb, out->depth, out->acc, w1, out->raw, out->feature);
        if (out->z_vals) memcpy(out->z_vals, z2, sizeof(float) * R * St);
        if (out->weights) memcpy(out->weights, w1, sizeof(float) * R * St);
        free(zmid); free(wmid); free(zs); free(z2); free(w1);
    }
    free(rb); free(z); free(w0);
}

/* ------------------------------------------------------------------ PDRF tri-plane features
 * networks/pdrf/voxnerf.py:203-208 sample() + :132-151 compute_appfeature().
 * F.grid_sample(bilinear, zeros padding, align_corners=True); interpolation weights follow the
 * ATen CPU kernel (w = x - floor(x), e = 1 - w).  plane i: [C_i, grid[m1], grid[m0]] sampled at
 * (x = xyz[m0], y = xyz[m1]), matMode [[0,1],[0,2],[1,2]]; line i: [C_i, grid[vec]] at xyz[vec],
 * vecMode [2,1,0] (its width-1 axis sits at x = 0 where the east tap has weight 0). */
static inline float unnorm(float c, int size) { return ((c + 1.f) / 2.f) * (float)(size - 1); }

void evo_appfeature(const evo_voxel* v, const float* pts, long n, float* out) {
    static const int mat[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    static const int vec[3] = {2, 1, 0};
    const int nc_tot = v->n_comp[0] + v->n_comp[1] + v->n_comp[2];
    float* bt = transpose_w(v->basis, v->app_dim, nc_tot);
#pragma omp parallel
    {
        float* coef = (float*)malloc(sizeof(float) * nc_tot);
#pragma omp for schedule(static)
        for (long p = 0; p < n; ++p) {
            float xyz[3];
            for (int c = 0; c < 3; ++c) {
                const float inv = 2.0f / (v->aabb[3 + c] - v->aabb[c]);          /* invaabbSize :91 */
                xyz[c] = (pts[p * 3 + c] - v->aabb[c]) * inv - 1.f;             /* :205 */
            }
            int off = 0;
            for (int i = 0; i < 3; ++i) {
                const int Wp = v->grid[mat[i][0]], Hp = v->grid[mat[i][1]], Lp = v->grid[vec[i]];
                const float ix = unnorm(xyz[mat[i][0]], Wp), iy = unnorm(xyz[mat[i][1]], Hp);
                const float fx = floorf(ix), fy = floorf(iy);
                const float ww = ix - fx, ee = 1.f - ww, nn = iy - fy, ss = 1.f - nn;
                const long x0 = (long)fx, y0 = (long)fy, x1 = x0 + 1, y1 = y0 + 1;
                const int vx0 = x0 >= 0 && x0 < Wp, vx1 = x1 >= 0 && x1 < Wp, vy0 = y0 >= 0 && y0 < Hp, vy1 = y1 >= 0 && y1 < Hp;
                const float il = unnorm(xyz[vec[i]], Lp);
                const float fl = floorf(il);
                const float ln = il - fl, ls = 1.f - ln;
                const long l0 = (long)fl, l1 = l0 + 1;
                const int vl0 = l0 >= 0 && l0 < Lp, vl1 = l1 >= 0 && l1 < Lp;
                for (int c = 0; c < v->n_comp[i]; ++c) {
                    const float* pl = v->plane[i] + (long)c * Hp * Wp;
                    float pv = 0.f;
                    if (vy0 && vx0) pv += pl[y0 * Wp + x0] * (ee * ss);
                    if (vy0 && vx1) pv += pl[y0 * Wp + x1] * (ww * ss);
                    if (vy1 && vx0) pv += pl[y1 * Wp + x0] * (ee * nn);
                    if (vy1 && vx1) pv += pl[y1 * Wp + x1] * (ww * nn);
                    const float* li = v->line[i] + (long)c * Lp;
                    float lv = 0.f;
                    if (vl0) lv += li[l0] * ls;      /* x = 0: west weight e = 1, east w = 0 */
                    if (vl1) lv += li[l1] * ln;
                    coef[off + c] = pv * lv;
                }
                off += v->n_comp[i];
            }
            float* o = out + p * v->app_dim;
            linear_t(bt, NULL, nc_tot, v->app_dim, coef, o);                    /* basis_mat :151 */
            for (int j = 0; j < v->app_dim; ++j) o[j] = act(v->app_act, o[j]);
        }
        free(coef);
    }
    free(bt);
}

/* VoxelNeRFBase.forward: networks/pdrf/voxnerf.py:210-259.  fts [R,S,F]; feature out [R,S,geo]. */
void evo_voxel_forward(const evo_voxel* v, const float* pts, const float* viewdirs, const float* fts, int F,
                       const float* z, const float* rays_d, long R, int S, int multires, int multires_views,
                       int is_train, float* color, float* depth, float* acc, float* weights, float* feature) {
    const int ic = 3 * (1 + 2 * multires), icv = 3 * (1 + 2 * multires_views);
    const long n = R * S, nn1 = n > 0 ? n : 1, Rn = R > 0 ? R : 1;
    const int G = v->geo_feat_dim, HD = v->hidden_dim;
    float* sw[EVO_MAX_LAYERS]; int sin_[EVO_MAX_LAYERS], sout[EVO_MAX_LAYERS];
    for (int l = 0; l < v->num_layers; ++l) {
        sin_[l] = l == 0 ? v->input_ch : HD;
        sout[l] = l == v->num_layers - 1 ? 1 + G : HD;
        sw[l] = transpose_w(v->sigma_w[l], sout[l], sin_[l]);
    }
    float* cw[EVO_MAX_LAYERS]; int cin[EVO_MAX_LAYERS], cout[EVO_MAX_LAYERS];
    for (int l = 0; l < v->num_layers_color; ++l) {
        cin[l] = l == 0 ? v->input_ch_views + G : HD;      /* hidden_dim, not hidden_dim_color: voxnerf.py:73,78 */
        cout[l] = l == v->num_layers_color - 1 ? 3 : HD;
        cw[l] = transpose_w(v->color_w[l], cout[l], cin[l]);
    }
    float* pe = (float*)malloc(sizeof(float) * (size_t)ic * nn1);
    float* pev = (float*)malloc(sizeof(float) * (size_t)icv * Rn);
    evo_embed(pts, n, 3, multires, pe);
    evo_embed(viewdirs, R, 3, multires_views, pev);
    float* hbuf = (float*)malloc(sizeof(float) * (size_t)(1 + G) * nn1);
    const int wide = (v->input_ch > HD ? v->input_ch : HD) + G + icv + 8;
#pragma omp parallel
    {
        float* a = (float*)malloc(sizeof(float) * wide);
        float* b = (float*)malloc(sizeof(float) * wide);
#pragma omp for schedule(static)
        for (long s = 0; s < n; ++s) {
            memcpy(a, fts + s * F, sizeof(float) * F);                           /* cat([fts, PE(pts)]) :214 */
            memcpy(a + F, pe + s * ic, sizeof(float) * ic);
            for (int l = 0; l < v->num_layers; ++l) {
                linear_t(sw[l], NULL, sin_[l], sout[l], a, b);
                if (l != v->num_layers - 1) for (int j = 0; j < sout[l]; ++j) b[j] = b[j] > 0.f ? b[j] : 0.f;
                float* t = a; a = b; b = t;
            }
            memcpy(hbuf + s * (1 + G), a, sizeof(float) * (1 + G));
            if (feature && !v->composite_feature) memcpy(feature + s * G, a + 1, sizeof(float) * G);      /* feature_map = h[...,1:] :221 */
        }
        free(a); free(b);
    }
    const float thr = (!is_train && v->rmnear > 0.f) ? (float)((double)v->rmnear / 128.0) : 0.f;
    if (v->composite_feature) {
        /* composite the (1+G)-channel h first (:223-229), then colour net per ray (:231-239) */
        float* fm = (float*)malloc(sizeof(float) * (size_t)G * Rn);
        evo_composite(hbuf, z, rays_d, R, S, 1 + G, 0, 1, G, v->rgb_act, v->sigma_act, 0, thr, NULL,
                      fm, NULL, acc, weights, depth, NULL, 0, NULL);
#pragma omp parallel
        {
            float* a = (float*)malloc(sizeof(float) * wide);
            float* b = (float*)malloc(sizeof(float) * wide);
#pragma omp for schedule(static)
            for (long r = 0; r < R; ++r) {
                memcpy(a, fm + r * G, sizeof(float) * G);
                memcpy(a + G, pev + r * icv, sizeof(float) * icv);
                for (int l = 0; l < v->num_layers_color; ++l) {
                    linear_t(cw[l], v->color_b[l], cin[l], cout[l], a, b);
                    if (l != v->num_layers_color - 1) for (int j = 0; j < cout[l]; ++j) b[j] = b[j] > 0.f ? b[j] : 0.f;
                    float* t = a; a = b; b = t;
                }
                for (int c = 0; c < 3; ++c) color[r * 3 + c] = 1.f / (1.f + expf(-a[c]));
            }
            free(a); free(b);
        }
        if (feature) memcpy(feature, fm, sizeof(float) * (size_t)G * R);       /* the returned feature_map is the composited one [R,G] :226 */
        free(fm);
    } else {
        /* per-sample colour then composite (:240-257); raw = cat([sigma, sigmoid(colour)]) */
        float* raw = (float*)malloc(sizeof(float) * 4 * nn1);
#pragma omp parallel
        {
            float* a = (float*)malloc(sizeof(float) * wide);
            float* b = (float*)malloc(sizeof(float) * wide);
#pragma omp for schedule(static)
            for (long s = 0; s < n; ++s) {
                const float* h = hbuf + s * (1 + G);
                memcpy(a, h + 1, sizeof(float) * G);
                memcpy(a + G, pev + (s / S) * icv, sizeof(float) * icv);
                for (int l = 0; l < v->num_layers_color; ++l) {
                    linear_t(cw[l], v->color_b[l], cin[l], cout[l], a, b);
                    if (l != v->num_layers_color - 1) for (int j = 0; j < cout[l]; ++j) b[j] = b[j] > 0.f ? b[j] : 0.f;
                    float* t = a; a = b; b = t;
                }
                raw[s * 4] = h[0];
                for (int c = 0; c < 3; ++c) raw[s * 4 + 1 + c] = 1.f / (1.f + expf(-a[c]));
            }
            free(a); free(b);
        }
        evo_composite(raw, z, rays_d, R, S, 4, 0, 1, 3, v->rgb_act, v->sigma_act, 0, thr, NULL,
                      color, NULL, acc, weights, depth, NULL, 0, NULL);
        free(raw);
    }
    for (int l = 0; l < v->num_layers; ++l) free(sw[l]);
    for (int l = 0; l < v->num_layers_color; ++l) free(cw[l]);
    free(pe); free(pev); free(hbuf);
}

typedef struct { float z; int idx; } zi_t;
static int cmp_zi(const void* a, const void* b) {
    const zi_t *x = (const zi_t*)a, *y = (const zi_t*)b;
    if (x->z < y->z) return -1;
    if (x->z > y->z) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* NeRFAll.render + render_rays for mode='c2f': networks/renderer.py:182-217.  Features of the merged
 * sample set are gathered by the sort order, not re-sampled (:205-213). */
void evo_render_c2f(const evo_voxel* coarse, const evo_voxel* fine, const evo_render_cfg* cfg,
                    const float* rays, long R, const float* t_rand, const float* u, evo_render_out* out) {
    const int S = cfg->N_samples, Ni = cfg->N_importance, St = S + Ni;
    const long Rn = R > 0 ? R : 1;
    const int Fc = coarse->app_dim, Ff = fine ? fine->app_dim : 0;
    int nc;
    float* rb = (float*)malloc(sizeof(float) * 11 * Rn);
    evo_ray_batch(cfg, rays, R, rb, &nc);
    float* z = (float*)malloc(sizeof(float) * Rn * S);
    make_z(cfg, rb, nc, R, t_rand, z);
    float* rd = (float*)malloc(sizeof(float) * 3 * Rn);
    float* vd = (float*)malloc(sizeof(float) * 3 * Rn);
    float* pts = (float*)malloc(sizeof(float) * 3 * Rn * S);
    for (long r = 0; r < R; ++r) {
        for (int c = 0; c < 3; ++c) { rd[r * 3 + c] = rb[r * nc + 3 + c]; vd[r * 3 + c] = rb[r * nc + 8 + c]; }
        for (int i = 0; i < S; ++i)
            for (int c = 0; c < 3; ++c) pts[(r * S + i) * 3 + c] = rb[r * nc + c] + rd[r * 3 + c] * z[r * (long)S + i];
    }
    float* ftc = (float*)malloc(sizeof(float) * (size_t)Fc * Rn * S);
    evo_appfeature(coarse, pts, R * S, ftc);
    float* w0 = (float*)malloc(sizeof(float) * Rn * S);
    if (Ni <= 0) {
        evo_voxel_forward(coarse, pts, vd, ftc, Fc, z, rd, R, S, cfg->multires, cfg->multires_views, cfg->is_train,
                          out->rgb, out->depth, out->acc, w0, out->feature);
        if (out->z_vals) memcpy(out->z_vals, z, sizeof(float) * R * S);
        if (out->weights) memcpy(out->weights, w0, sizeof(float) * R * S);
    } else {
        evo_voxel_forward(coarse, pts, vd, ftc, Fc, z, rd, R, S, cfg->multires, cfg->multires_views, cfg->is_train,
                          out->rgb0, out->depth0, out->acc0, w0, NULL);
        if (out->z_vals0) memcpy(out->z_vals0, z, sizeof(float) * R * S);
        if (out->weights0) memcpy(out->weights0, w0, sizeof(float) * R * S);
        const int Fm = Fc + Ff;
        float* ftf = (float*)malloc(sizeof(float) * (size_t)Ff * Rn * S);
        evo_appfeature(fine, pts, R * S, ftf);
        float* zmid = (float*)malloc(sizeof(float) * Rn * (S - 1));
        float* wmid = (float*)malloc(sizeof(float) * Rn * (S - 2));
        float* zs = (float*)malloc(sizeof(float) * Rn * Ni);
        for (long r = 0; r < R; ++r) {
            for (int i = 0; i < S - 1; ++i) zmid[r * (long)(S - 1) + i] = .5f * (z[r * (long)S + i + 1] + z[r * (long)S + i]);
            for (int i = 0; i < S - 2; ++i) wmid[r * (long)(S - 2) + i] = w0[r * (long)S + i + 1];
        }
        evo_sample_pdf(zmid, wmid, R, S - 1, Ni, cfg->perturb == 0.f, u, zs);
        if (out->z_std) zstd(zs, R, Ni, out->z_std);
        float* pts1 = (float*)malloc(sizeof(float) * 3 * Rn * Ni);
        for (long r = 0; r < R; ++r)
            for (int i = 0; i < Ni; ++i)
                for (int c = 0; c < 3; ++c) pts1[(r * Ni + i) * 3 + c] = rb[r * nc + c] + rd[r * 3 + c] * zs[r * (long)Ni + i];
        float* ftc1 = (float*)malloc(sizeof(float) * (size_t)Fc * Rn * Ni);
        float* ftf1 = (float*)malloc(sizeof(float) * (size_t)Ff * Rn * Ni);
        evo_appfeature(coarse, pts1, R * Ni, ftc1);
        evo_appfeature(fine, pts1, R * Ni, ftf1);
        float* z2 = (float*)malloc(sizeof(float) * Rn * St);
        float* pts2 = (float*)malloc(sizeof(float) * 3 * Rn * St);
        float* ft2 = (float*)malloc(sizeof(float) * (size_t)Fm * Rn * St);
        zi_t* zi = (zi_t*)malloc(sizeof(zi_t) * St);
        for (long r = 0; r < R; ++r) {
            for (int i = 0; i < S; ++i) { zi[i].z = z[r * (long)S + i]; zi[i].idx = i; }
            for (int i = 0; i < Ni; ++i) { zi[S + i].z = zs[r * (long)Ni + i]; zi[S + i].idx = S + i; }
            qsort(zi, St, sizeof(zi_t), cmp_zi);
            for (int k = 0; k < St; ++k) {
                const int id = zi[k].idx;
                z2[r * (long)St + k] = zi[k].z;
                float* fo = ft2 + (r * (long)St + k) * Fm;
                if (id < S) {
                    memcpy(pts2 + (r * (long)St + k) * 3, pts + (r * (long)S + id) * 3, sizeof(float) * 3);
                    memcpy(fo, ftc + (r * (long)S + id) * Fc, sizeof(float) * Fc);
                    memcpy(fo + Fc, ftf + (r * (long)S + id) * Ff, sizeof(float) * Ff);
                } else {
                    memcpy(pts2 + (r * (long)St + k) * 3, pts1 + (r * (long)Ni + id - S) * 3, sizeof(float) * 3);
                    memcpy(fo, ftc1 + (r * (long)Ni + id - S) * Fc, sizeof(float) * Fc);
                    memcpy(fo + Fc, ftf1 + (r * (long)Ni + id - S) * Ff, sizeof(float) * Ff);
                }
            }
        }
        float* w1 = (float*)malloc(sizeof(float) * Rn * St);
        evo_voxel_forward(fine, pts2, vd, ft2, Fm, z2, rd, R, St, cfg->multires, cfg->multires_views, cfg->is_train,
                          out->rgb, out->depth, out->acc, w1, out->feature);
        if (out->z_vals) memcpy(out->z_vals, z2, sizeof(float) * R * St);
        if (out->weights) memcpy(out->weights, w1, sizeof(float) * R * St);
        free(ftf); free(zmid); free(wmid); free(zs); free(pts1); free(ftc1); free(ftf1);
        free(z2); free(pts2); free(ft2); free(zi); free(w1);
    }
    free(rb); free(z); free(rd); free(vd); free(pts); free(ftc); free(w0);
}

/* ------------------------------------------------------------------ loss-side pixel ops
 * RigidBlurringModel.rbk_weighted_sum, networks/dpnerf/blurmodel.py:112-127: out[r] = sum_p ccw[r,p] x[r*P+p] */
void evo_weighted_sum(const float* x, const float* ccw, long R, int P, int C, float* out) {
    for (long r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            float s = 0.f;
            for (int p = 0; p < P; ++p) s += x[(r * P + p) * (long)C + c] * ccw[r * P + p];
            out[r * C + c] = s;
        }
}

/* CRF.forward, networks/tonemapping.py:59-93.  x [n,3]; feat NULL, [n,E] (feat_per_channel=0, repeated for
 * the 3 channels :75-76) or [n,3,E] (feat_per_channel=1).  Missing features are zero padded (:82-85). */
void evo_crf_forward(const evo_crf* crf, const float* x, const float* feat, int feat_per_channel,
                     int skip_learn, long n, float* out) {
    const int E = crf->extra_features;
    for (long i = 0; i < n * 3; ++i) {
        float v = x[i];
        if (crf->map_type == 0) { out[i] = v; continue; }                        /* 'none' :64-65 */
        if (crf->map_type == 1) v = powf(v, (float)(1.0 / (double)crf->gamma));   /* 'gamma' in map_type :67-68 */
        if (!skip_learn && crf->map_type == 2) {
            float in[1 + 16], h[16], h2[16];
            in[0] = v;
            for (int e = 0; e < E; ++e)
                in[1 + e] = feat ? (feat_per_channel ? feat[i * E + e] : feat[(i / 3) * E + e]) : 0.f;
            for (int j = 0; j < 16; ++j) {
                float s = crf->b[0][j];
                for (int k = 0; k < 1 + E; ++k) s += crf->w[0][j * (1 + E) + k] * in[k];
                h[j] = s > 0.f ? s : 0.f;
            }
            for (int l = 1; l <= 2; ++l) {
                for (int j = 0; j < 16; ++j) {
                    float s = crf->b[l][j];
                    for (int k = 0; k < 16; ++k) s += crf->w[l][j * 16 + k] * h[k];
                    h2[j] = s > 0.f ? s : 0.f;
                }
                memcpy(h, h2, sizeof(h));
            }
            float s = crf->b[3][0];
            for (int k = 0; k < 16; ++k) s += crf->w[3][k] * h[k];
            v = 1.f / (1.f + expf(-(s * 0.1f + v)));                              /* sigmoid(0.1 mlp + x) :87-88 */
        }
        out[i] = v;
    }
}

/* TonemappingTransform.encode_luma tail, networks/tonemapping.py:127-136; standard 0 rec601, 1 rec709, 2 avg */
void evo_luma(const float* x, long n, int standard, float* out) {
    for (long i = 0; i < n; ++i) {
        const float r = x[i * 3], g = x[i * 3 + 1], b = x[i * 3 + 2];
        if (standard == 0) out[i] = 0.299f * r + 0.587f * g + 0.114f * b;
        else if (standard == 1) out[i] = 0.2126f * r + 0.7152f * g + 0.0722f * b;
        else out[i] = (r + g + b) / 3.f;
    }
}

/* img2mse, utils/metrics.py:7 */
double evo_mse(const float* a, const float* b, long n) {
    double s = 0;
    for (long i = 0; i < n; ++i) { double d = (double)(a[i] - b[i]); s += d * d; }
    return n > 0 ? s / (double)n : NAN;
}

/* egm_loss, utils/events.py:260-284.  luma [n,C]; C = 1 (no mask) or 3 with one-hot color_mask [n,3]. */
double evo_egm_loss(const float* luma_start, const float* luma_end, const float* bii, long n, int C,
                    const unsigned char* color_mask, const float* color_weight) {
    double num = 0, den = 0;
    for (long i = 0; i < n; ++i) {
        int ch = 0;
        if (color_mask) for (int c = 0; c < 3; ++c) if (color_mask[i * 3 + c]) ch = c;
        const float pred = logf(luma_end[i * C + ch] + 1e-5f) - logf(luma_start[i * C + ch] + 1e-5f);
        const float w = (color_mask && color_weight) ? color_weight[ch] : 1.f;
        const float d = pred - bii[i];
        num += (double)(d * d * w);
        den += (double)w;
    }
    return num / den;
}

/* ------------------------------------------------------------------ EDI (utils/edi.py)
 * interpolate_subpixel :7-41 + brightness_increment_image :44-70 (grey events only). */
static void splat(const float* x, const float* y, const signed char* p, long n, int want_pos, int w, int h,
                  int interpolate, float* img) {
    for (long i = 0; i < n; ++i) {
        if ((p[i] > 0) != want_pos) continue;
        if (!interpolate) { img[(long)y[i] * w + (long)x[i]] += 1.f; continue; }
        for (int xr = 0; xr < 2; ++xr)
            for (int yr = 0; yr < 2; ++yr) {
                const float xf = xr ? ceilf(x[i]) : floorf(x[i]);
                const float yf = yr ? ceilf(y[i]) : floorf(y[i]);
                if (!((xf != x[i] || xr == 0) && (yf != y[i] || yr == 0) && xf < (float)w && yf < (float)h)) continue;
                const float kx = fmaxf(0.f, 1.f - fabsf(xf - x[i])), ky = fmaxf(0.f, 1.f - fabsf(yf - y[i]));
                img[(long)yf * w + (long)xf] += 1.f * kx * ky;
            }
    }
}

void evo_bii_image(const float* x, const float* y, const signed char* p, long n, int w, int h,
                   float c_pos, float c_neg, int interpolate, float* image) {
    float* ip = (float*)calloc((size_t)w * h, sizeof(float));
    float* in = (float*)calloc((size_t)w * h, sizeof(float));
    /* the reference accumulates the 4 taps as 4 passes over all events (product(floor/ceil)): keep that order */
    if (interpolate) {
        for (int pass = 0; pass < 4; ++pass) {
            const int xr = pass >> 1, yr = pass & 1;
            for (long i = 0; i < n; ++i) {
                const float xf = xr ? ceilf(x[i]) : floorf(x[i]);
                const float yf = yr ? ceilf(y[i]) : floorf(y[i]);
                if (!((xf != x[i] || xr == 0) && (yf != y[i] || yr == 0) && xf < (float)w && yf < (float)h)) continue;
                const float kx = fmaxf(0.f, 1.f - fabsf(xf - x[i])), ky = fmaxf(0.f, 1.f - fabsf(yf - y[i]));
                float* img = p[i] > 0 ? ip : in;
                img[(long)yf * w + (long)xf] += 1.f * kx * ky;
            }
        }
    } else {
        splat(x, y, p, n, 1, w, h, 0, ip);
        splat(x, y, p, n, 0, w, h, 0, in);
    }
    for (long i = 0; i < (long)w * h; ++i) image[i] = ip[i] * c_pos - in[i] * c_neg;   /* :69 */
    free(ip); free(in);
}

/* inner_double_integral, utils/edi.py:73-88.  bii [steps-1, npix] -> images [steps, npix], N = (steps-1)/2 */
void evo_inner_double_integral(const float* bii, int steps, long npix, float* images) {
    const int N = (steps - 1) / 2;
    for (long px = 0; px < npix; ++px) {
        for (int i = 0; i < N; ++i) {
            float s = 0.f;
            for (int j = i; j < N; ++j) s += bii[(long)j * npix + px];
            images[(long)i * npix + px] = -s;
        }
        images[(long)N * npix + px] = 0.f;
        for (int i = 0; i < N; ++i) {
            float s = 0.f;
            for (int j = N; j < N + 1 + i; ++j) s += bii[(long)j * npix + px];
            images[(long)(N + 1 + i) * npix + px] = s;
        }
    }
}

/* deblur_double_integral, utils/edi.py:91-95: sharp = (2N+1) blurry / sum_k exp(E_k) */
void evo_deblur_double_integral(const float* blurry, const float* bii, int steps, long npix, float* sharp) {
    float* im = (float*)malloc(sizeof(float) * (size_t)steps * npix);
    evo_inner_double_integral(bii, steps, npix, im);
    const int N = (steps - 1) / 2;
    for (long px = 0; px < npix; ++px) {
        float s = 0.f;
        for (int k = 0; k < steps; ++k) s += expf(im[(long)k * npix + px]);
        sharp[px] = (float)(2 * N + 1) * blurry[px] / s;
    }
    free(im);
}

/* TVLoss.forward, networks/pdrf/voxnerf.py:306-324 for x [1,C,H,W] */
double evo_tv_loss(const float* x, int C, int H, int W) {
    double h_tv = 0, w_tv = 0;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                const float v = x[((long)c * H + i) * W + j];
                if (i + 1 < H) { float d = x[((long)c * H + i + 1) * W + j] - v; h_tv += (double)(d * d); }
                if (j + 1 < W) { float d = x[((long)c * H + i) * W + j + 1] - v; w_tv += (double)(d * d); }
            }
    double count_h = (double)C * (H - 1) * W;
    double count_w = (double)C * H * (W - 1);
    if (count_w < 1) count_w = 1;
    return 2.0 * (h_tv / count_h + w_tv / count_w);
}

/* ------------------------------------------------------------------ AWP feature integration
 * networks/dpnerf/awp.py:49-77, restated AS WRITTEN: alpha = 1 - exp(-feat * dist) on the first S-1 samples (:65-66), a
 * ZERO alpha appended for the last one (:67); the "transmittance" is torch.cumprod(cat([ones, 1 - alpha + 1e-10], -2), -1)
 * [:, :-1, :] (:69-73) -- a cumulative product along the CHANNEL axis (dim -1) of the previous sample's row, not along the
 * samples:  Q[0, c] = 1,  Q[s, c] = prod_{c' <= c} (1 - alpha[s-1, c'] + 1e-10);  out[c] = sum_s alpha[s, c] Q[s, c] feat[s, c]. */
void evo_awp_feature_integration(const float* feat, const float* z, const float* rays_d, long N, int S, int C, float* out) {
#pragma omp parallel for schedule(static)
    for (long n = 0; n < N; ++n) {
        const float* d = rays_d + n * 3;
        const float norm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        float* o = out + n * C;
        for (int c = 0; c < C; ++c) o[c] = 0.f;
        for (int s = 0; s < S; ++s) {
            float q = 1.f;
            for (int c = 0; c < C; ++c) {
                const float f = feat[((size_t)n * S + s) * C + c];
                float alpha = 0.f;
                if (s < S - 1) alpha = -expf(-f * ((z[n * S + s + 1] - z[n * S + s]) * norm)) + 1.f;
                if (s > 0) {
                    const float fp = feat[((size_t)n * S + s - 1) * C + c];
                    const float ap = -expf(-fp * ((z[n * S + s] - z[n * S + s - 1]) * norm)) + 1.f;   /* s-1 < S-1 always */
                    q *= (-ap + (1.f + 1e-10f));
                }
                o[c] += (alpha * q) * f;
            }
        }
    }
}

/* awp.py:98-100: for l in sample_feature_embed_layer: h = relu(l(h)).  nn.Linear = x W^T + b (float32 accumulation in input order) */
void evo_awp_sample_embed(const float* x, const float* const* W, const float* const* b, long n, int in_dim, int width, int depth, float* out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        float cur[512], nxt[512];
        int d = in_dim;
        for (int c = 0; c < in_dim; ++c) cur[c] = x[(size_t)i * in_dim + c];
        for (int l = 0; l < depth; ++l) {
            for (int o = 0; o < width; ++o) {
                float acc = 0.f;
                for (int c = 0; c < d; ++c) acc += cur[c] * W[l][(size_t)o * d + c];
                acc += b[l][o];
                nxt[o] = acc > 0.f ? acc : 0.f;
            }
            d = width;
            for (int o = 0; o < width; ++o) cur[o] = nxt[o];
        }
        for (int o = 0; o < width; ++o) out[(size_t)i * width + o] = cur[o];
    }
}

/* MotionAggregationModule.forward, mam.py:72-74 (curves = self.linear(x_local), [R, P, S, 64] -> [R, P, S, M]) and CorrelationModule.forward,
 * mam.py:29-33, AS WRITTEN: curves_att = line_conv_att(curves) (a 1x1 convolution without bias: v . curves), softmax over the samples
 * (dim -1) and over the sub-exposures (dim -2), curver_inter [R, M, P] = sum_s curves softmax_s, curves_intra [R, M, S] = sum_p curves
 * softmax_p.  x_local [R P, S, C]; W [M, C], b [M] (nn.Linear), v [M]. */
void evo_mam_local(const float* x_local, const float* W, const float* b, const float* v, long R, int P, int S, int Cc, int M, float* inter,
                   float* intra) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r) {
        float* cur = (float*)malloc(sizeof(float) * (size_t)P * S * M);
        float* att = (float*)malloc(sizeof(float) * (size_t)P * S);
        for (int p = 0; p < P; ++p)
            for (int s = 0; s < S; ++s) {
                const float* x = x_local + (((size_t)r * P + p) * S + s) * Cc;
                float a = 0.f;
                for (int m = 0; m < M; ++m) {
                    float acc = 0.f;
                    for (int c = 0; c < Cc; ++c) acc += x[c] * W[(size_t)m * Cc + c];
                    acc += b[m];
                    cur[((size_t)p * S + s) * M + m] = acc;
                    a += v[m] * acc;
                }
                att[p * S + s] = a;
            }
        for (int p = 0; p < P; ++p) {                       /* softmax(dim=-1), sum over the samples */
            float mx = -INFINITY, den = 0.f;
            for (int s = 0; s < S; ++s) mx = fmaxf(mx, att[p * S + s]);
            for (int s = 0; s < S; ++s) den += expf(att[p * S + s] - mx);
            for (int m = 0; m < M; ++m) {
                float acc = 0.f;
                for (int s = 0; s < S; ++s) acc += cur[((size_t)p * S + s) * M + m] * (expf(att[p * S + s] - mx) / den);
                inter[((size_t)r * M + m) * P + p] = acc;
            }
        }
        for (int s = 0; s < S; ++s) {                       /* softmax(dim=-2), sum over the sub-exposures */
            float mx = -INFINITY, den = 0.f;
            for (int p = 0; p < P; ++p) mx = fmaxf(mx, att[p * S + s]);
            for (int p = 0; p < P; ++p) den += expf(att[p * S + s] - mx);
            for (int m = 0; m < M; ++m) {
                float acc = 0.f;
                for (int p = 0; p < P; ++p) acc += cur[((size_t)p * S + s) * M + m] * (expf(att[p * S + s] - mx) / den);
                intra[((size_t)r * M + m) * S + s] = acc;
            }
        }
        free(cur);
        free(att);
    }
}

/* The per-ray remainder of the adaptive weight proposal, AS WRITTEN:
 *   awp.py:104-109   h = cat(h_integrated [R, P, Ws], view_embedded [R, VC] repeated over P); n_mot x (Linear + ReLU) -> x_global [R, P, Cm]
 *   mam.py:35-53     CorrelationModule.forward behind the per-sample part (curver_inter [R, Cm, P] and curves_intra [R, Cm, S] are INPUTS,
 *                    evo_mam_local's outputs): conva / convb (Cm -> Cm/2), convc on x, the two softmax(bmm) attention maps, convn / convl,
 *                    the two bmm, concatenation, convd = Conv1d(Cm -> Cm, no bias) + BatchNorm1d over (ray, position) per channel --
 *                    batch statistics (biased variance) when `training`, the running estimates otherwise --, residual, leaky_relu(0.2)
 *   awp.py:112-115   mean over the P positions, w_linear, sigmoid, division by the sum
 * Every Conv1d has kernel size 1: a matrix product over the channel axis.  batch_stats [2 Cm] (may be NULL): the batch mean and the
 * UNBIASED batch variance (what BatchNorm blends into its running estimates).  float32 accumulation in input order, double for the
 * batch statistics. */
void evo_awp_per_ray(const float* h, const float* view, const float* inter, const float* intra, const float* const* mot_w,
                     const float* const* mot_b, int n_mot, const float* conva, const float* convb, const float* convc, const float* convn,
                     const float* convl, const float* convd, const float* bn_w, const float* bn_b, const float* bn_mean,
                     const float* bn_var, float bn_eps, int training, const float* wl_w, const float* wl_b, long R, int P, int S, int Ws,
                     int VC, int Cm, float* out, float* batch_stats) {
    const int mid = Cm / 2, in0 = Ws + VC;
    float* xg = (float*)malloc(sizeof(float) * (size_t)R * P * Cm);
    float* y = (float*)malloc(sizeof(float) * (size_t)R * P * Cm);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r) {
        float* a = (float*)malloc(sizeof(float) * (size_t)(in0 > Cm ? in0 : Cm));
        float* b = (float*)malloc(sizeof(float) * (size_t)Cm);
        float* kP = (float*)malloc(sizeof(float) * (size_t)P * mid), *nP = (float*)malloc(sizeof(float) * (size_t)P * mid);
        float* kS = (float*)malloc(sizeof(float) * (size_t)S * mid), *nS = (float*)malloc(sizeof(float) * (size_t)S * mid);
        float* lg = (float*)malloc(sizeof(float) * (size_t)(S > P ? S : P));
        float* f = (float*)malloc(sizeof(float) * (size_t)Cm);
        for (int p = 0; p < P; ++p) {                                          /* awp.py:104-109 */
            for (int k = 0; k < Ws; ++k) a[k] = h[((size_t)r * P + p) * Ws + k];
            for (int k = 0; k < VC; ++k) a[Ws + k] = view[(size_t)r * VC + k];
            int K = in0;
            for (int l = 0; l < n_mot; ++l) {
                for (int o = 0; o < Cm; ++o) {
                    float acc = 0.f;
                    for (int k = 0; k < K; ++k) acc += a[k] * mot_w[l][(size_t)o * K + k];
                    acc += mot_b[l][o];
                    b[o] = acc > 0.f ? acc : 0.f;
                }
                for (int o = 0; o < Cm; ++o) a[o] = b[o];
                K = Cm;
            }
            for (int o = 0; o < Cm; ++o) xg[((size_t)r * P + p) * Cm + o] = a[o];
        }
        for (int p = 0; p < P; ++p)                                            /* mam.py:38, 46: conva, convn on curver_inter[:, :, p] */
            for (int m = 0; m < mid; ++m) {
                float acc = 0.f;
                for (int c = 0; c < Cm; ++c) acc += conva[m * Cm + c] * inter[((size_t)r * Cm + c) * P + p];
                kP[p * mid + m] = acc;
            }
        for (int p = 0; p < P; ++p)
            for (int m = 0; m < mid; ++m) {
                float acc = 0.f;
                for (int c = 0; c < mid; ++c) acc += convn[m * mid + c] * kP[p * mid + c];
                nP[p * mid + m] = acc;
            }
        for (int s = 0; s < S; ++s)                                            /* mam.py:39, 47: convb, convl on curves_intra[:, :, s] */
            for (int m = 0; m < mid; ++m) {
                float acc = 0.f;
                for (int c = 0; c < Cm; ++c) acc += convb[m * Cm + c] * intra[((size_t)r * Cm + c) * S + s];
                kS[s * mid + m] = acc;
            }
        for (int s = 0; s < S; ++s)
            for (int m = 0; m < mid; ++m) {
                float acc = 0.f;
                for (int c = 0; c < mid; ++c) acc += convl[m * mid + c] * kS[s * mid + c];
                nS[s * mid + m] = acc;
            }
        for (int p = 0; p < P; ++p) {
            const float* x = xg + ((size_t)r * P + p) * Cm;
            for (int m = 0; m < mid; ++m) {                                    /* mam.py:41: x_logits */
                float acc = 0.f;
                for (int c = 0; c < Cm; ++c) acc += convc[m * Cm + c] * x[c];
                b[m] = acc;
            }
            for (int half = 0; half < 2; ++half) {                             /* mam.py:42-43, 49-50 */
                const int n = half ? S : P;
                const float* kk = half ? kS : kP;
                const float* nn = half ? nS : nP;
                float mx = -INFINITY, den = 0.f;
                for (int j = 0; j < n; ++j) {
                    float acc = 0.f;
                    for (int m = 0; m < mid; ++m) acc += b[m] * kk[j * mid + m];
                    lg[j] = acc;
                    mx = fmaxf(mx, acc);
                }
                for (int j = 0; j < n; ++j) den += expf(lg[j] - mx);
                for (int m = 0; m < mid; ++m) {
                    float acc = 0.f;
                    for (int j = 0; j < n; ++j) acc += (expf(lg[j] - mx) / den) * nn[j * mid + m];
                    f[half * mid + m] = acc;
                }
            }
            for (int o = 0; o < Cm; ++o) {                                     /* mam.py:52-53: convd[0] */
                float acc = 0.f;
                for (int c = 0; c < Cm; ++c) acc += convd[o * Cm + c] * f[c];
                y[((size_t)r * P + p) * Cm + o] = acc;
            }
        }
        free(a); free(b); free(kP); free(nP); free(kS); free(nS); free(lg); free(f);
    }
    const double cnt = (double)R * P;
    float* mean = (float*)malloc(sizeof(float) * (size_t)Cm), *rstd = (float*)malloc(sizeof(float) * (size_t)Cm);
    for (int c = 0; c < Cm; ++c) {                                             /* BatchNorm1d(Cm) on [R, Cm, P]: statistics over (R, P) */
        double s1 = 0.0, s2 = 0.0;
        for (long i = 0; i < R * P; ++i) s1 += y[(size_t)i * Cm + c];
        const double mu = s1 / cnt;
        for (long i = 0; i < R * P; ++i) { const double d = y[(size_t)i * Cm + c] - mu; s2 += d * d; }
        if (batch_stats) { batch_stats[c] = (float)mu; batch_stats[Cm + c] = (float)(s2 / (cnt > 1.0 ? cnt - 1.0 : 1.0)); }
        mean[c] = training ? (float)mu : bn_mean[c];
        rstd[c] = 1.0f / sqrtf((training ? (float)(s2 / cnt) : bn_var[c]) + bn_eps);
    }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r) {
        float hm[512], w[64];
        for (int c = 0; c < Cm; ++c) {
            float acc = 0.f;
            for (int p = 0; p < P; ++p) {
                const size_t at = ((size_t)r * P + p) * Cm + c;
                const float v = xg[at] + ((y[at] - mean[c]) * rstd[c] * bn_w[c] + bn_b[c]);
                acc += v > 0.f ? v : 0.2f * v;                                 /* mam.py:53 */
            }
            hm[c] = acc / (float)P;                                            /* awp.py:112 */
        }
        float tot = 0.f;
        for (int j = 0; j < P; ++j) {                                          /* awp.py:114-115 */
            float acc = 0.f;
            for (int c = 0; c < Cm; ++c) acc += wl_w[j * Cm + c] * hm[c];
            w[j] = 1.0f / (1.0f + expf(-(acc + wl_b[j])));
            tot += w[j];
        }
        for (int j = 0; j < P; ++j) out[(size_t)r * P + j] = w[j] / tot;
    }
    free(mean); free(rstd); free(xg); free(y);
}

/* ------------------------------------------------------------------ RBK ray warp
 * SE3Field.get_transform (rigid_warping.py:18-30): theta = |rot| + 1e-10, screw axis (rot, trans) / theta;
 * RigidBody.exp_se3 (:72-91): R = I + sin(theta) W + (1 - cos(theta)) W^2 (Rodrigues, :93-107),
 * p = (theta I + (1 - cos theta) W + (theta - sin theta) W^2) v;  warp (:32-49): homogeneous 4x4 product, divided by w.
 * rbk_warp (blurmodel.py:51-82): motion i warps the ray origin and the end point o + d; slot 0 keeps the input ray
 * when use_origin. */
static void evo_se3(const float rot[3], const float trans[3], float T[16]) {
    const float theta = sqrtf(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]) + 1.0e-10f;
    const float w[3] = {rot[0] / theta, rot[1] / theta, rot[2] / theta}, vv[3] = {trans[0] / theta, trans[1] / theta, trans[2] / theta};
    const float Wm[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
    float W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += Wm[i * 3 + k] * Wm[k * 3 + j];
            W2[i * 3 + j] = s;
        }
    const float st = sinf(theta), omc = 1.0f - cosf(theta), tms = theta - st;
    for (int i = 0; i < 3; ++i) {
        float p = 0.f;
        for (int j = 0; j < 3; ++j) {
            const float eye = i == j ? 1.f : 0.f;
            T[i * 4 + j] = eye + st * Wm[i * 3 + j] + omc * W2[i * 3 + j];
            p += (theta * eye + omc * Wm[i * 3 + j] + tms * W2[i * 3 + j]) * vv[j];
        }
        T[i * 4 + 3] = p;
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

static void evo_apply(const float T[16], const float x[3], float y[3]) {
    float h[4];
    for (int i = 0; i < 4; ++i) h[i] = T[i * 4] * x[0] + T[i * 4 + 1] * x[1] + T[i * 4 + 2] * x[2] + T[i * 4 + 3] * 1.f;
    for (int i = 0; i < 3; ++i) y[i] = h[i] / h[3];
}

void evo_rbk_warp(const float* rays, const float* r, const float* v, long R, int M, int use_origin, float* new_rays, float* transforms) {
    const int P = M + (use_origin ? 1 : 0);
#pragma omp parallel for schedule(static)
    for (long n = 0; n < R; ++n) {
        float o[3], d[3], e[3];
        for (int c = 0; c < 3; ++c) { o[c] = rays[n * 6 + c * 2]; d[c] = rays[n * 6 + c * 2 + 1]; e[c] = o[c] + d[c]; }
        if (use_origin) {
            for (int c = 0; c < 3; ++c) { new_rays[(n * P) * 6 + c * 2] = o[c]; new_rays[(n * P) * 6 + c * 2 + 1] = d[c]; }
            if (transforms) for (int k = 0; k < 16; ++k) transforms[(n * P) * 16 + k] = (k % 5 == 0) ? 1.f : 0.f;
        }
        for (int i = 0; i < M; ++i) {
            const float rot[3] = {r[(n * 3 + 0) * M + i], r[(n * 3 + 1) * M + i], r[(n * 3 + 2) * M + i]};
            const float tr[3] = {v[(n * 3 + 0) * M + i], v[(n * 3 + 1) * M + i], v[(n * 3 + 2) * M + i]};
            float T[16], wo[3], we[3];
            evo_se3(rot, tr, T);
            evo_apply(T, o, wo);
            evo_apply(T, e, we);
            const long slot = n * P + i + (use_origin ? 1 : 0);
            for (int c = 0; c < 3; ++c) { new_rays[slot * 6 + c * 2] = wo[c]; new_rays[slot * 6 + c * 2 + 1] = we[c] - wo[c]; }
            if (transforms) for (int k = 0; k < 16; ++k) transforms[slot * 16 + k] = T[k];
        }
    }
}

/* ------------------------------------------------------------------ event successor graph
 * utils/events.py:72-120, the reverse loop as written: latest_seen_idx ends up holding the FIRST event of each pixel,
 * first_seen_idx the LAST one; an event without a later event at its pixel is its own successor. */
void evo_compute_successor(const int* ids, long N, long HW, long long* successor, int* num_successors, long long* latest_seen, long long* first_seen) {
    for (long k = 0; k < HW; ++k) { latest_seen[k] = -1; first_seen[k] = -1; }
    for (long i = N - 1; i >= 0; --i) {
        const int x = ids[i];
        if (latest_seen[x] != -1) {
            successor[i] = latest_seen[x];
            num_successors[i] = num_successors[successor[i]] + 1;
        } else {
            successor[i] = i;
            num_successors[i] = 0;
        }
        latest_seen[x] = i;
        if (first_seen[x] == -1) first_seen[x] = i;
    }
}

/* ------------------------------------------------------------------ event batch assembly
 * data/loader_events.py:259-304 (EventsDataset.sample_events) with the per-event poses as a table: the start / end event of every id
 * (hops == NULL: :272-276, the successor and its polarity; else gather_successor, utils/events.py:221-257: hops + 1 steps, polarity sums
 * by sign, -1 / 0 / 0 for a chain that leaves the table), coordinates by id (:286-287), get_rays_pix on both poses (:292-297,
 * utils/rays.py:25-36; unfused float32 like the tensor ops). */
void evo_sample_events(const double* ev, long N, int ncol, const float* id_to_coords, const unsigned char* cmap, const float* poses,
                       const long long* ids, const long long* hops, long n, const float* K, int add_halfpix, float* rays_start, float* rays_end,
                       float* pos_out, float* neg_out, long long* coords_ids, unsigned char* cmap_out, long long* succ_out) {
    const float halfpix = add_halfpix ? 0.5f : 0.f;
    for (long i = 0; i < n; ++i) {
        const long long id = ids[i];
        const long long pix = (long long)ev[id * ncol];
        long long end;
        float pos = 0.f, neg = 0.f;
        if (!hops) {
            end = (long long)ev[id * ncol + ncol - 1];
            const double p = ev[end * ncol + ncol - 2];
            if (p > 0) pos = (float)p; else neg = (float)p;
        } else {
            int invalid = 0;
            end = id;
            for (long long h = 0; h <= hops[i]; ++h) {
                const long long nxt = (long long)ev[end * ncol + ncol - 1];
                if (nxt < 0 || nxt >= N) { invalid = 1; break; }
                end = nxt;
                const int p = (int)ev[end * ncol + ncol - 2];
                if (p > 0) pos += (float)p;
                if (p < 0) neg += (float)p;
            }
            if (invalid) { end = -1; pos = neg = 0.f; }
        }
        pos_out[i] = pos; neg_out[i] = neg; coords_ids[i] = pix;
        if (succ_out) succ_out[i] = end;
        if (cmap_out) for (int c = 0; c < 3; ++c) cmap_out[i * 3 + c] = cmap[pix * 3 + c];
        const float cx = id_to_coords[pix * 2], cy = id_to_coords[pix * 2 + 1];
        const float d0 = (cx + (halfpix - K[2])) / K[0], d1 = -(cy + (halfpix - K[5])) / K[4], d2 = -1.f;
        const long long e2 = end < 0 ? id : end;
        for (int which = 0; which < 2; ++which) {
            const float* c2w = poses + (which ? e2 : id) * 12;
            float* out = (which ? rays_end : rays_start) + i * 6;
            for (int r = 0; r < 3; ++r) {
                out[r * 2] = c2w[r * 4 + 3];
                out[r * 2 + 1] = d0 * c2w[r * 4 + 0] + d1 * c2w[r * 4 + 1] + d2 * c2w[r * 4 + 2];
            }
        }
    }
}

/* ------------------------------------------------------------------ camera trajectory
 * data/loader_events.py:133-148 interpolate_poses on top of utils/data.py:34-62 _get_slerp_interpolator (built at
 * loader_events.py:175-182): scipy Slerp of the key rotations, cubic interp1d (make_interp_spline k = 3, not-a-knot ends) of the key
 * translations, queries clipped to the key range; then the LLFF column change [r1, -r0, r2, t] (:137), float32 (:138), translation x
 * bd_scale in float32 (:140), recenter_poses = inv(c2w) @ pose in float64 stored float32 (utils/data.py:167-183).
 * scipy is a third-party dependency absent from /root/reference (pinned 1.9.1 in environment.yml; the build container has 1.15.3,
 * whose from_matrix orthogonalises its input first): its published algorithms are restated from the raw key poses --
 *   Rotation.from_matrix: nearest rotation (polar factor; here by Newton's iteration X <- (X + X^-T) / 2), then the quaternion branch
 *     chosen by the largest of (m00, m11, m22, trace), normalised;
 *   Slerp: rotvec_i = as_rotvec(q_i^-1 q_i+1) (w >= 0; series below 1e-3), result = q_ind * from_rotvec(alpha * rotvec_ind) with
 *     ind = searchsorted(times, t, 'left') - 1 (0 for t == times[0]), alpha = (t - t_ind) / (t_ind+1 - t_ind); as_matrix;
 *   not-a-knot cubic: second derivatives from the tridiagonal system with the end unknowns eliminated by continuity of the third
 *     derivative at x_1 and x_M-2.
 * Pinned by golden G29 (the reference's own interpolate_poses / sample_events run on scipy 1.15.3). */
static void evo_polar3(const double* m, double* r) {
    double x[9];
    memcpy(x, m, sizeof(x));
    for (int it = 0; it < 12; ++it) {
        const double c00 = x[4] * x[8] - x[5] * x[7], c01 = x[5] * x[6] - x[3] * x[8], c02 = x[3] * x[7] - x[4] * x[6];
        const double c10 = x[2] * x[7] - x[1] * x[8], c11 = x[0] * x[8] - x[2] * x[6], c12 = x[1] * x[6] - x[0] * x[7];
        const double c20 = x[1] * x[5] - x[2] * x[4], c21 = x[2] * x[3] - x[0] * x[5], c22 = x[0] * x[4] - x[1] * x[3];
        const double det = x[0] * c00 + x[1] * c01 + x[2] * c02;
        const double cof[9] = {c00, c01, c02, c10, c11, c12, c20, c21, c22};      /* cofactor matrix = det * X^-T */
        double delta = 0;
        for (int k = 0; k < 9; ++k) {
            const double nx = 0.5 * (x[k] + cof[k] / det);
            delta = fmax(delta, fabs(nx - x[k]));
            x[k] = nx;
        }
        if (delta < 1e-17) break;
    }
    memcpy(r, x, sizeof(x));
}

static void evo_quat_from_matrix(const double* m, double* q) {
    const double dec[4] = {m[0], m[4], m[8], m[0] + m[4] + m[8]};
    int c = 0;
    for (int k = 1; k < 4; ++k) if (dec[k] > dec[c]) c = k;
    if (c != 3) {
        const int i = c, j = (c + 1) % 3, k = (c + 2) % 3;
        q[i] = 1 - dec[3] + 2 * m[i * 3 + i];
        q[j] = m[j * 3 + i] + m[i * 3 + j];
        q[k] = m[k * 3 + i] + m[i * 3 + k];
        q[3] = m[k * 3 + j] - m[j * 3 + k];
    } else {
        q[0] = m[7] - m[5]; q[1] = m[2] - m[6]; q[2] = m[3] - m[1]; q[3] = 1 + dec[3];
    }
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k) q[k] /= nrm;
}

static void evo_qmul(const double* p, const double* q, double* o) {
    o[0] = p[3] * q[0] + p[0] * q[3] + p[1] * q[2] - p[2] * q[1];
    o[1] = p[3] * q[1] - p[0] * q[2] + p[1] * q[3] + p[2] * q[0];
    o[2] = p[3] * q[2] + p[0] * q[1] - p[1] * q[0] + p[2] * q[3];
    o[3] = p[3] * q[3] - p[0] * q[0] - p[1] * q[1] - p[2] * q[2];
}

/* key_t [M] ascending, key_poses [M, 3, 4] (all_poses of loader_events.py:171), recenter_c2w host [3 or 4, 4] row-major 4 x 4 or NULL,
 * t [n] -> poses float32 [n, 3, 4].  Returns 0, or -1 for M < 4 / non-ascending timestamps. */
int evo_interpolate_poses(const double* key_t, const double* key_poses, int M, float bd_scale, const double* recenter_c2w44,
                          const double* t, long n, float* poses) {
    if (M < 4) return -1;
    for (int i = 0; i + 1 < M; ++i) if (!(key_t[i + 1] > key_t[i])) return -1;
    double* q = (double*)malloc(sizeof(double) * 4 * M);
    double* rv = (double*)malloc(sizeof(double) * 3 * (M - 1));
    double* m2 = (double*)calloc((size_t)3 * M, sizeof(double));        /* second derivatives of the translation spline, [M, 3] */
    for (int i = 0; i < M; ++i) {
        double R[9], Rp[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r * 3 + c] = key_poses[i * 12 + r * 4 + c];
        evo_polar3(R, Rp);
        evo_quat_from_matrix(Rp, q + i * 4);
    }
    for (int i = 0; i + 1 < M; ++i) {
        const double qi[4] = {-q[i * 4], -q[i * 4 + 1], -q[i * 4 + 2], q[i * 4 + 3]};
        double d[4];
        evo_qmul(qi, q + (i + 1) * 4, d);
        if (d[3] < 0) for (int k = 0; k < 4; ++k) d[k] = -d[k];
        const double nv = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const double ang = 2 * atan2(nv, d[3]);
        const double sc = ang <= 1e-3 ? 2 + ang * ang / 12 + 7 * ang * ang * ang * ang / 2880 : ang / sin(ang / 2);
        for (int k = 0; k < 3; ++k) rv[i * 3 + k] = d[k] * sc;
    }
    {   /* not-a-knot cubic through (key_t, translation) */
        const int nn = M - 2;
        double* h = (double*)malloc(sizeof(double) * (M - 1));
        double* a = (double*)malloc(sizeof(double) * nn);
        double* b = (double*)malloc(sizeof(double) * nn);
        double* c = (double*)malloc(sizeof(double) * nn);
        double* cp = (double*)malloc(sizeof(double) * nn);
        double* rp = (double*)malloc(sizeof(double) * nn);
        for (int i = 0; i + 1 < M; ++i) h[i] = key_t[i + 1] - key_t[i];
        for (int i = 0; i < nn; ++i) { a[i] = h[i]; b[i] = 2 * (h[i] + h[i + 1]); c[i] = h[i + 1]; }
        b[0] += h[0] * (1 + h[0] / h[1]);
        c[0] -= h[0] * h[0] / h[1];
        b[nn - 1] += h[M - 2] * (1 + h[M - 2] / h[M - 3]);
        a[nn - 1] -= h[M - 2] * h[M - 2] / h[M - 3];
        for (int dim = 0; dim < 3; ++dim) {
#define YV(i) key_poses[(i) * 12 + dim * 4 + 3]
            for (int i = 0; i < nn; ++i) {
                const double d0 = (YV(i + 1) - YV(i)) / h[i], d1 = (YV(i + 2) - YV(i + 1)) / h[i + 1];
                const double r = 6 * (d1 - d0);
                if (i == 0) { cp[0] = c[0] / b[0]; rp[0] = r / b[0]; }
                else { const double den = b[i] - a[i] * cp[i - 1]; cp[i] = c[i] / den; rp[i] = (r - a[i] * rp[i - 1]) / den; }
            }
            m2[nn * 3 + dim] = rp[nn - 1];
            for (int i = nn - 2; i >= 0; --i) m2[(i + 1) * 3 + dim] = rp[i] - cp[i] * m2[(i + 2) * 3 + dim];
            m2[dim] = (1 + h[0] / h[1]) * m2[3 + dim] - (h[0] / h[1]) * m2[6 + dim];
            m2[(M - 1) * 3 + dim] = (1 + h[M - 2] / h[M - 3]) * m2[(M - 2) * 3 + dim] - (h[M - 2] / h[M - 3]) * m2[(M - 3) * 3 + dim];
        }
        free(h); free(a); free(b); free(c); free(cp); free(rp);
    }
    double inv[16];
    if (recenter_c2w44) {       /* np.linalg.inv(c2w) of the rigid-or-not 4 x 4: Gauss-Jordan with partial pivoting */
        double aug[4][8];
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { aug[r][c] = recenter_c2w44[r * 4 + c]; aug[r][4 + c] = r == c; }
        for (int col = 0; col < 4; ++col) {
            int piv = col;
            for (int r = col + 1; r < 4; ++r) if (fabs(aug[r][col]) > fabs(aug[piv][col])) piv = r;
            for (int c = 0; c < 8; ++c) { const double tmp = aug[col][c]; aug[col][c] = aug[piv][c]; aug[piv][c] = tmp; }
            const double d = aug[col][col];
            for (int c = 0; c < 8; ++c) aug[col][c] /= d;
            for (int r = 0; r < 4; ++r) if (r != col) { const double f = aug[r][col]; for (int c = 0; c < 8; ++c) aug[r][c] -= f * aug[col][c]; }
        }
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) inv[r * 4 + c] = aug[r][4 + c];
    }
    for (long i = 0; i < n; ++i) {
        double tq = t[i];
        if (tq < key_t[0]) tq = key_t[0];
        if (tq > key_t[M - 1]) tq = key_t[M - 1];
        int lo = 0, hi = M;
        while (lo < hi) { const int mid = (lo + hi) / 2; if (key_t[mid] < tq) lo = mid + 1; else hi = mid; }
        int ind = lo - 1;
        if (ind < 0) ind = 0;
        if (ind > M - 2) ind = M - 2;
        const double hh = key_t[ind + 1] - key_t[ind], alpha = (tq - key_t[ind]) / hh;
        const double r3[3] = {rv[ind * 3] * alpha, rv[ind * 3 + 1] * alpha, rv[ind * 3 + 2] * alpha};
        const double ang = sqrt(r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2]);
        const double sc = ang <= 1e-3 ? 0.5 - ang * ang / 48 + ang * ang * ang * ang / 3840 : sin(ang / 2) / ang;
        const double dq[4] = {r3[0] * sc, r3[1] * sc, r3[2] * sc, cos(ang / 2)};
        double qq[4];
        evo_qmul(q + ind * 4, dq, qq);
        const double x = qq[0], y = qq[1], z = qq[2], w = qq[3];
        const double R[9] = {x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w};
        double T[3];
        for (int dim = 0; dim < 3; ++dim) {      /* the interval's cubic from its end values and second derivatives */
            const double y0 = key_poses[ind * 12 + dim * 4 + 3], y1 = key_poses[(ind + 1) * 12 + dim * 4 + 3];
            const double ma = m2[ind * 3 + dim], mb = m2[(ind + 1) * 3 + dim], dd = (y1 - y0) / hh;
            const double c1 = hh * dd - hh * hh * (2 * ma + mb) / 6, c2 = hh * hh * ma / 2, c3 = hh * hh * (mb - ma) / 6;
            T[dim] = y0 + alpha * (c1 + alpha * (c2 + alpha * c3));
        }
        float P[12];
        for (int r = 0; r < 3; ++r) {
            P[r * 4] = (float)R[r * 3 + 1];
            P[r * 4 + 1] = (float)(-R[r * 3]);
            P[r * 4 + 2] = (float)R[r * 3 + 2];
            P[r * 4 + 3] = (float)T[r] * bd_scale;
        }
        float* out = poses + i * 12;
        if (!recenter_c2w44) { memcpy(out, P, sizeof(P)); continue; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double acc = inv[r * 4] * (double)P[c] + inv[r * 4 + 1] * (double)P[4 + c] + inv[r * 4 + 2] * (double)P[8 + c];
                if (c == 3) acc += inv[r * 4 + 3];
                out[r * 4 + c] = (float)acc;
            }
    }
    free(q); free(rv); free(m2);
    return 0;
}

/* ------------------------------------------------------------------ image batch assembly
 * data/loader.py:325-356 LLFFDataset.__getitem__: unravel_index in C order (:118-123), poses[img_id], images[img_id, y, x],
 * get_rays_pix on the integer pixel with the half-pixel offset (utils/rays.py:25-36: (x + (0.5 - K02)) / K00, the bracket a Python double
 * rounded to float32 when it meets the tensor), rays_x / rays_y = pixel + HALF_PIX.  Returns the number of ids outside the dataset
 * (the reference raises an IndexError for those; their outputs are zeros, image id -1). */
long evo_image_batch(const long long* ids, long n, const float* images, const float* pts0, const float* poses, int n_img, int H, int W,
                     const float* K, float* rays, float* rays_x, float* rays_y, long long* img_idx, float* rgb, float* poses_out, float* rgb0) {
    const long long hw = (long long)H * W;
    const float hx = (float)(0.5 - (double)K[2]), hy = (float)(0.5 - (double)K[5]);
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        const long long id = ids[i];
        if (id < 0 || id >= hw * n_img) {
            ++bad;
            for (int k = 0; k < 6; ++k) rays[i * 6 + k] = 0.f;
            rays_x[i] = rays_y[i] = 0.f; img_idx[i] = -1;
            for (int k = 0; k < 3; ++k) { rgb[i * 3 + k] = 0.f; if (rgb0) rgb0[i * 3 + k] = 0.f; }
            if (poses_out) for (int k = 0; k < 12; ++k) poses_out[i * 12 + k] = 0.f;
            continue;
        }
        const long long im = id / hw, rem = id % hw;
        const int y = (int)(rem / W), x = (int)(rem % W);
        const float* c2w = poses + im * 12;
        const float d0 = ((float)x + hx) / K[0], d1 = -((float)y + hy) / K[4], d2 = -1.f;
        for (int r = 0; r < 3; ++r) {
            rays[i * 6 + r * 2] = c2w[r * 4 + 3];
            rays[i * 6 + r * 2 + 1] = d0 * c2w[r * 4] + d1 * c2w[r * 4 + 1] + d2 * c2w[r * 4 + 2];
        }
        rays_x[i] = (float)x + 0.5f; rays_y[i] = (float)y + 0.5f; img_idx[i] = im;
        for (int k = 0; k < 3; ++k) { rgb[i * 3 + k] = images[id * 3 + k]; if (rgb0) rgb0[i * 3 + k] = pts0[id * 3 + k]; }
        if (poses_out) memcpy(poses_out + i * 12, c2w, sizeof(float) * 12);
    }
    return bad;
}
