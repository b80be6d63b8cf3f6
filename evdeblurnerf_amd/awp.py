"""The AWP consumer of the path's per-sample outputs (reference ``networks/dpnerf/awp.py``): the per-sample part of
``AdaptiveWeightProposal`` -- ``sample_feature_embed_layer`` (awp.py:36-37,98-100: 4 x Linear + ReLU over every sample of every
sub-exposure ray) and the ``feature_integration`` scan (awp.py:49-77) -- on hand-written kernels, forward and backward, as autograd
nodes, and the per-sample part of the ``MotionAggregationModule`` (mam.py:72-74 ``linear`` on h_local, :29-33 attention logit, the two
softmaxes and weighted sums) as one more, and the per-ray remainder (direction encoding, motion embedding, the rest of the
CorrelationModule incl. its BatchNorm over all rays, ``w_linear``, the normalisation: awp.py:89-95, 104-117, mam.py:35-53) as a third
(``evd_awp_tail_forward`` / ``_backward``).  ``FusedAWP`` wraps the reference's module, whose parameters and buffers stay where the optimizer
and the state dict expect them; a module or a shape the kernels are not built for runs the remainder on its own PyTorch layers."""
from __future__ import annotations

import ctypes as C
import os

import weakref

import torch

from . import _lib as L
from .voxnerf import GeoFragments


# data_ptr of a d h_local tensor -> (device word with the float bits of its max |.|, weak reference to the tensor, its version counter):
# _LocalConsumers.backward -> _SampleEmbed.backward.  The word is only valid for the very values the producer wrote, so the consumer
# checks that the producer's tensor is still alive (its address cannot have been handed to another tensor) and unmodified (autograd adds
# a second consumer's gradient IN PLACE, which bumps the version); anything else falls back to the kernel's own maximum pass.
_DH_ABSMAX = {}


def _take_dh_absmax(g):
    ent = _DH_ABSMAX.pop(g.data_ptr(), None)
    _DH_ABSMAX.clear()
    if ent is None:
        return None
    amax, ref, version = ent
    src = ref()
    if src is None or src.data_ptr() != g.data_ptr() or src.numel() != g.numel() or src._version != version or g._version != version:
        return None
    return amax


class _FeatureIntegration(torch.autograd.Function):
    """evd_awp_feature_integration / evd_awp_feature_integration_bwd as one autograd node: gradients to the per-sample features (the
    AWP embedding MLP and, through it, the fine level), to z_vals and to rays_d (the blur kernel's ray directions)."""

    @staticmethod
    def forward(ctx, f, z, d):
        N, S, Cc = f.shape
        out = torch.empty((N, Cc), dtype=torch.float32, device=f.device)
        L.check(L.lib().evd_awp_feature_integration(L.ptr(f), L.ptr(z), L.ptr(d), N, S, Cc, L.ptr(out), L.stream_ptr()),
                "evd_awp_feature_integration")
        ctx.save_for_backward(f, z, d)
        return out

    @staticmethod
    def backward(ctx, g):
        f, z, d = ctx.saved_tensors
        N, S, Cc = f.shape
        g = g.contiguous().float()
        df = torch.empty_like(f)
        dz = torch.empty_like(z) if ctx.needs_input_grad[1] else None
        dd = torch.empty_like(d) if ctx.needs_input_grad[2] else None
        L.check(L.lib().evd_awp_feature_integration_bwd(L.ptr(f), L.ptr(z), L.ptr(d), L.ptr(g), N, S, Cc, L.ptr(df), L.ptr(dz), L.ptr(dd),
                                                        L.stream_ptr()), "evd_awp_feature_integration_bwd")
        return df, dz, dd


def feature_integration(feat, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """awp.py:49-77: feat [N_rays, N_motion, N_samples, C], z_vals [N_rays*N_motion, N_samples], rays_d [N_rays*N_motion, 3]
    -> [N_rays, N_motion, C].  Restated as written in the reference: every channel is its own density, the last alpha is 0,
    and the cumprod of awp.py:69-73 runs along the channel axis (dim -1) of the previous sample's row."""
    n_rays, n_motion, S, Cc = feat.shape
    f = feat.reshape(-1, S, Cc).contiguous().float()
    z = z_vals.reshape(-1, S).contiguous().float()
    d = rays_d.reshape(-1, 3).contiguous().float()
    N = f.shape[0]
    if z.shape[0] != N or d.shape[0] != N:
        raise L.EvdError("feature_integration: z_vals / rays_d need one row per (ray, motion)")
    return _FeatureIntegration.apply(f, z, d).reshape(n_rays, n_motion, Cc)


class _SampleEmbed(torch.autograd.Function):
    """evd_awp_embed_forward / _backward.  `src` is either the GeoFragments token (the geo features are read as the fragments of the
    fine level's store, and their gradient is left as fragments for the level's backward) or a float32 tensor [n, 128]."""

    @staticmethod
    def forward(ctx, src, flat, embed, geo):
        lib, prec = L.lib(), L.PREC[embed.precision]
        if geo is not None:
            n, rows, dev = geo.R * geo.S, None, geo.store.device
        else:
            rows = src.reshape(-1, embed.input_ch).contiguous().float()
            n, dev = rows.shape[0], rows.device
        nb = int(lib.evd_awp_embed_store_bytes(embed._h, n))
        store = torch.empty((nb,), dtype=torch.uint8, device=dev)
        h_local = torch.empty((n, embed.width), dtype=torch.float32, device=dev)
        L.check(lib.evd_awp_embed_forward(embed._h, prec, L.ptr(rows), geo.level._h if geo is not None else None,
                                          L.ptr(geo.store) if geo is not None else None, geo.store.numel() if geo is not None else 0, n,
                                          L.ptr(h_local), L.ptr(store), nb, L.stream_ptr()), "evd_awp_embed_forward")
        ctx.embed, ctx.geo, ctx.store, ctx.n, ctx.src_shape = embed, geo, store, n, src.shape
        return h_local

    @staticmethod
    def backward(ctx, d_h):
        embed, lib, n = ctx.embed, L.lib(), ctx.n
        g = d_h.contiguous().float()
        gflat = torch.zeros((embed.nparam,), dtype=torch.float32, device=g.device)
        gs, base = L.AwpEmbedGrads(), gflat.data_ptr()
        for l, (wo, bo) in enumerate(embed.offsets):
            gs.w[l], gs.b[l] = base + 4 * wo, base + 4 * bo
        d_rows = torch.empty((n, embed.input_ch), dtype=torch.float32, device=g.device) if (ctx.geo is None and ctx.needs_input_grad[0]) else None
        nb = int(lib.evd_awp_embed_backward_workspace_bytes())
        ws = torch.empty((nb,), dtype=torch.uint8, device=g.device)
        amax = _take_dh_absmax(g)                    # the producer of d h_local took its maximum; None: evd_awp_embed_backward takes it itself
        L.check(lib.evd_awp_embed_backward(embed._h, L.PREC[embed.precision], L.ptr(g), n, L.ptr(ctx.store), ctx.store.numel(), C.byref(gs),
                                           L.ptr(d_rows), L.ptr(amax), L.ptr(ws), nb, L.stream_ptr()), "evd_awp_embed_backward")
        if ctx.geo is not None:
            ctx.geo.awp_store = ctx.store          # the level's backward (which autograd runs after this node) adds the d geo fragments
            d_src = torch.zeros(ctx.src_shape, dtype=torch.float32, device=g.device)
        else:
            d_src = d_rows.reshape(ctx.src_shape) if d_rows is not None else None
        ctx.store = None
        return d_src, gflat, None, None


class _MamLocal(torch.autograd.Function):
    """evd_mam_local_forward / _backward: h_local [R P, S, 64], u [64] -> (h_inter [R, P, 64], h_intra [R, S, 64]), the softmax-weighted
    sums of h_local along the samples / along the sub-exposures with the logits u . h_local (mam.py:29-33 before the 64 -> 32 map)."""

    @staticmethod
    def forward(ctx, h_local, u, R, P, S):
        h, uu = h_local.contiguous().float(), u.contiguous().float()
        Cc = h.shape[-1]
        if h.numel() != R * P * S * Cc:
            raise L.EvdError(f"mam_local: h_local has {h.numel()} elements, expected {R} x {P} x {S} x {Cc}")
        dev = h.device
        h_inter = torch.empty((R, P, Cc), dtype=torch.float32, device=dev)
        h_intra = torch.empty((R, S, Cc), dtype=torch.float32, device=dev)
        alpha = torch.empty((R, P, S), dtype=torch.float32, device=dev)
        beta = torch.empty((R, P, S), dtype=torch.float32, device=dev)
        L.check(L.lib().evd_mam_local_forward(L.ptr(h), L.ptr(uu), R, P, S, Cc, L.ptr(h_inter), L.ptr(h_intra), L.ptr(alpha), L.ptr(beta),
                                              L.stream_ptr()), "evd_mam_local_forward")
        ctx.save_for_backward(h, uu, alpha, beta, h_inter, h_intra)
        ctx.dims, ctx.h_shape = (R, P, S, Cc), h_local.shape
        return h_inter, h_intra

    @staticmethod
    def backward(ctx, g_inter, g_intra):
        h, uu, alpha, beta, h_inter, h_intra = ctx.saved_tensors
        R, P, S, Cc = ctx.dims
        d_h = torch.empty_like(h)
        d_u = torch.empty((R, Cc), dtype=torch.float32, device=h.device)
        L.check(L.lib().evd_mam_local_backward(L.ptr(h), L.ptr(uu), L.ptr(alpha), L.ptr(beta), L.ptr(h_inter), L.ptr(h_intra),
                                               L.ptr(g_inter.contiguous().float()), L.ptr(g_intra.contiguous().float()), R, P, S, Cc,
                                               L.ptr(d_h), L.ptr(d_u), 0, None, L.stream_ptr()), "evd_mam_local_backward")
        return d_h.reshape(ctx.h_shape), d_u.sum(0), None, None, None


class _LocalConsumers(torch.autograd.Function):
    """h_local's two consumers behind the embedding -- the feature integration (awp.py:102) and the MAM's per-sample part (mam.py:29-33,
    72-74) -- as ONE autograd node: its backward lets the second kernel ADD into the d h_local the first one wrote, where two nodes hand
    autograd two [R P, S, 64] tensors to sum (a pass over 3 x 335 MB at the blurfactory shape).  -> (h [R P, 64], h_inter, h_intra)"""

    @staticmethod
    def forward(ctx, h_local, z, d, u, R, P, S):
        f, zz, dd, uu = h_local.contiguous().float(), z.contiguous().float(), d.contiguous().float(), u.contiguous().float()
        N, Cc, dev = R * P, f.shape[-1], f.device
        if f.numel() != N * S * Cc or zz.shape[0] != N or dd.shape[0] != N:
            raise L.EvdError("FusedAWP: h_local [R P, S, C], z_vals [R P, S] and rays_d [R P, 3] do not agree")
        f32 = dict(dtype=torch.float32, device=dev)
        h = torch.empty((N, Cc), **f32)
        h_inter, h_intra = torch.empty((R, P, Cc), **f32), torch.empty((R, S, Cc), **f32)
        alpha, beta = torch.empty((R, P, S), **f32), torch.empty((R, P, S), **f32)
        lib = L.lib()
        L.check(lib.evd_awp_feature_integration(L.ptr(f), L.ptr(zz), L.ptr(dd), N, S, Cc, L.ptr(h), L.stream_ptr()), "evd_awp_feature_integration")
        L.check(lib.evd_mam_local_forward(L.ptr(f), L.ptr(uu), R, P, S, Cc, L.ptr(h_inter), L.ptr(h_intra), L.ptr(alpha), L.ptr(beta),
                                          L.stream_ptr()), "evd_mam_local_forward")
        ctx.save_for_backward(f, zz, dd, uu, alpha, beta, h_inter, h_intra)
        ctx.dims, ctx.shapes = (R, P, S, Cc), (h_local.shape, z.shape, d.shape)
        return h, h_inter, h_intra

    @staticmethod
    def backward(ctx, g_h, g_inter, g_intra):
        f, zz, dd, uu, alpha, beta, h_inter, h_intra = ctx.saved_tensors
        R, P, S, Cc = ctx.dims
        N, lib = R * P, L.lib()
        d_f = torch.empty_like(f)
        d_z = torch.empty_like(zz) if ctx.needs_input_grad[1] else None
        d_d = torch.empty_like(dd) if ctx.needs_input_grad[2] else None
        d_u = torch.empty((R, Cc), dtype=torch.float32, device=f.device)
        amax = torch.zeros((1,), dtype=torch.int32, device=f.device)
        if os.environ.get("EVD_AWP_LOCAL_BWD") == "separate":      # developer switch: the two kernels, the second adding into the first one's result
            L.check(lib.evd_awp_feature_integration_bwd(L.ptr(f), L.ptr(zz), L.ptr(dd), L.ptr(g_h.contiguous().float()), N, S, Cc, L.ptr(d_f), L.ptr(d_z),
                                                        L.ptr(d_d), L.stream_ptr()), "evd_awp_feature_integration_bwd")
            L.check(lib.evd_mam_local_backward(L.ptr(f), L.ptr(uu), L.ptr(alpha), L.ptr(beta), L.ptr(h_inter), L.ptr(h_intra),
                                               L.ptr(g_inter.contiguous().float()), L.ptr(g_intra.contiguous().float()), R, P, S, Cc, L.ptr(d_f), L.ptr(d_u),
                                               1, L.ptr(amax), L.stream_ptr()), "evd_mam_local_backward")
        else:                                                       # one launch: h_local read once, d h_local written once
            L.check(lib.evd_awp_local_consumers_backward(L.ptr(f), L.ptr(uu), L.ptr(alpha), L.ptr(beta), L.ptr(h_inter), L.ptr(h_intra),
                                                         L.ptr(g_inter.contiguous().float()), L.ptr(g_intra.contiguous().float()), L.ptr(zz), L.ptr(dd),
                                                         L.ptr(g_h.contiguous().float()), R, P, S, Cc, L.ptr(d_f), L.ptr(d_u), L.ptr(d_z), L.ptr(d_d),
                                                         L.ptr(amax), L.stream_ptr()), "evd_awp_local_consumers_backward")
        _DH_ABSMAX.clear()
        _DH_ABSMAX[d_f.data_ptr()] = (amax, weakref.ref(d_f), d_f._version)
        sh = ctx.shapes
        return (d_f.reshape(sh[0]), None if d_z is None else d_z.reshape(sh[1]), None if d_d is None else d_d.reshape(sh[2]), d_u.sum(0),
                None, None, None)


def mam_local(h_local, linear_weight, att_weight, R, P, S):
    """The per-sample part of MotionAggregationModule.forward + CorrelationModule.forward (mam.py:72-74, 29-33): h_local [R P, S, 64],
    MAM.linear.weight [M, 64], Corr.line_conv_att.weight [1, M, 1, 1] -> (h_inter [R, P, 64], h_intra [R, S, 64]);
    curver_inter = (h_inter W^T + b)^T and curves_intra = (h_intra W^T + b)^T are the caller's (per-ray sized)."""
    u = att_weight.reshape(1, -1) @ linear_weight                     # the logit of a sample is (W^T v) . h + const
    return _MamLocal.apply(h_local, u.reshape(-1), R, P, S)


class _AwpTail(torch.autograd.Function):
    """evd_awp_tail_forward / _backward: the per-ray remainder of the proposal (awp.py:89-95, 104-117; mam.py:35-53) on the library's
    kernels.  cfg = (P, S, VF, dir_freqs, n_mot, training, eps, momentum); bn = (running_mean, running_var, num_batches_tracked) or
    Nones; params in the order include/evdnerf.h documents."""

    @staticmethod
    def forward(ctx, cfg, bn, h, view_feature, rays_d, h_inter, h_intra, *params):
        P, S, VF, F, n_mot, training, eps, mom = cfg
        desc = L.AwpTailDesc(P=P, S=S, VF=VF, dir_freqs=F, n_mot=n_mot, training=int(training), bn_eps=eps, bn_momentum=mom)
        dev = h.device
        f32 = dict(dtype=torch.float32, device=dev)
        hh, rd, hi, hs = (t.contiguous().float() for t in (h, rays_d, h_inter, h_intra))
        vf = view_feature.contiguous().float() if VF else None
        R = hh.shape[0]
        ps = [t.detach().contiguous().float() for t in params]
        arr = (C.c_void_p * len(ps))(*[t.data_ptr() for t in ps])
        lib = L.lib()
        out, y, xg = torch.empty((R, P), **f32), torch.empty((R, P, 32), **f32), torch.empty((R, P, 32), **f32)
        stats = torch.empty((64,), **f32)
        need = lib.evd_awp_tail_workspace_bytes(C.byref(desc), R, 0)
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        rm, rv, nb = bn
        keep = any(ctx.needs_input_grad)
        rays = torch.empty((R, lib.evd_awp_tail_saved_floats(C.byref(desc))), **f32) if keep else None     # the rays' forward state
        L.check(lib.evd_awp_tail_forward(C.byref(desc), arr, L.ptr(hh), L.ptr(vf), L.ptr(rd), L.ptr(hi), L.ptr(hs), R, L.ptr(rm), L.ptr(rv),
                                         L.ptr(nb), L.ptr(out), L.ptr(y), L.ptr(xg), L.ptr(stats), L.ptr(rays), L.ptr(ws), need, L.stream_ptr()),
                "evd_awp_tail_forward")
        ctx.rays = rays
        ctx.save_for_backward(hh, vf, rd, hi, hs, y, xg, stats, *ps)
        ctx.desc, ctx.shapes = desc, (h.shape, None if view_feature is None else view_feature.shape, rays_d.shape, h_inter.shape, h_intra.shape,
                                      [t.shape for t in params])
        return out

    @staticmethod
    def backward(ctx, g):
        hh, vf, rd, hi, hs, y, xg, stats, *ps = ctx.saved_tensors
        desc, lib, dev = ctx.desc, L.lib(), hh.device
        R = hh.shape[0]
        arr = (C.c_void_p * len(ps))(*[t.data_ptr() for t in ps])
        total = lib.evd_awp_tail_param_count(C.byref(desc))
        d_h, d_rd, d_hi, d_hs = torch.empty_like(hh), torch.empty_like(rd), torch.empty_like(hi), torch.empty_like(hs)
        d_vf = torch.empty_like(vf) if vf is not None else None
        d_par = torch.empty((total,), dtype=torch.float32, device=dev)
        need = lib.evd_awp_tail_workspace_bytes(C.byref(desc), R, 1)
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        L.check(lib.evd_awp_tail_backward(C.byref(desc), arr, L.ptr(hh), L.ptr(vf), L.ptr(rd), L.ptr(hi), L.ptr(hs), R, L.ptr(y), L.ptr(xg),
                                          L.ptr(stats), L.ptr(ctx.rays), L.ptr(g.contiguous().float()), L.ptr(d_h), L.ptr(d_vf), L.ptr(d_rd), L.ptr(d_hi),
                                          L.ptr(d_hs), L.ptr(d_par), L.ptr(ws), need, L.stream_ptr()), "evd_awp_tail_backward")
        ctx.rays = None
        sh = ctx.shapes
        grads = [t.reshape(s) for t, s in zip(d_par.split([int(torch.Size(s).numel()) for s in sh[5]]), sh[5])]
        return (None, None, d_h.reshape(sh[0]), None if d_vf is None else d_vf.reshape(sh[1]), d_rd.reshape(sh[2]), d_hi.reshape(sh[3]),
                d_hs.reshape(sh[4]), *grads)


def _dir_freqs(fn):
    """number of frequencies F if fn is the reference's get_embedder encoding of a direction (embedding.py:65-113: [d, sin(2^k d), cos(2^k d)
    for k < F]), found by evaluating it on a probe; None for anything else (the encoding then runs in torch, its columns go in as view_feature)"""
    try:
        with torch.no_grad():
            x = torch.tensor([[0.3, -0.5, 0.8], [-0.1, 0.7, 0.2]])
            e = fn(x)
            F = (e.shape[-1] // 3 - 1) // 2
            if e.shape[-1] != 3 + 6 * F or not 0 <= F <= 4:
                return None
            want = torch.cat([x] + [f(x * 2.0 ** k) for k in range(F) for f in (torch.sin, torch.cos)], -1)
            return F if torch.allclose(e.float().cpu(), want, atol=1e-6) else None
    except Exception:
        return None


class SampleFeatureEmbed:
    """sample_feature_embed_layer (awp.py:36-37) as a library handle: weights[l] [64, in_l], biases[l] [64] (nn.Linear layouts)."""

    def __init__(self, weights, biases, precision="f16"):
        ws = [torch.as_tensor(w).detach().float().cpu().contiguous() for w in weights]
        bs = [torch.as_tensor(b).detach().float().cpu().contiguous() for b in biases]
        self.depth, self.width, self.input_ch, self.precision = len(ws), ws[0].shape[0], ws[0].shape[1], precision
        fp = C.POINTER(C.c_float)
        wa = (fp * len(ws))(*[C.cast(w.data_ptr(), fp) for w in ws])
        ba = (fp * len(bs))(*[C.cast(b.data_ptr(), fp) for b in bs])
        h = C.c_void_p()
        L.check(L.lib().evd_awp_embed_create(wa, ba, self.input_ch, self.width, self.depth, C.byref(h)), "evd_awp_embed_create")
        self._h = h
        self.nparam = int(L.lib().evd_awp_embed_param_count(h))
        self.offsets, off = [], 0
        for w, b in zip(ws, bs):
            self.offsets.append((off, off + w.numel()))
            off += w.numel() + b.numel()
        self._synced = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and L is not None and getattr(L, "lib", None) is not None:     # module globals may be gone at interpreter shutdown
            L.lib().evd_awp_embed_destroy(h)
            self._h = None

    def load_params(self, flat):
        f = flat.detach().contiguous().float()
        if f.numel() != self.nparam:
            raise L.EvdError(f"flat parameter tensor has {f.numel()} elements, the embedding {self.nparam}")
        L.check(L.lib().evd_awp_embed_load_params(self._h, L.ptr(f), L.stream_ptr()), "evd_awp_embed_load_params")

    def __call__(self, flat, geo):
        """flat: the parameters W0, b0, W1, b1, ... (re-packed into the library's streams before the launch); geo: GeoFragments or a
        float32 tensor [..., 128] -> h_local [n, 64] with autograd to flat and to the geo features"""
        self.load_params(flat)
        if isinstance(geo, GeoFragments):
            # the fragments are the level's own half-precision MFMA operands: reading a bf16 store as f16 (or the reverse) would
            # reinterpret the bits silently -- the C entry can only check the store's size
            if geo.precision not in ("f16", "bf16") or geo.precision != self.precision:
                raise L.EvdError(f"SampleFeatureEmbed[{self.precision}] cannot read the geo fragments of a level trained in {geo.precision}: "
                                 "the fragment coupling needs the same half-precision mode (f16 or bf16) on both sides")
            return _SampleEmbed.apply(geo.token, flat, self, geo)
        return _SampleEmbed.apply(geo, flat, self, None)


class _SplitKLinear(torch.autograd.Function):
    """y = x W^T (+ b) for TALL inputs ([rays x positions, <= 64 channels] against a 32..64-wide layer).  The weight gradient of such a layer
    is a [out, in] = 32 x 64 result summed over 10^4 .. 10^5 rows: the BLAS library runs it on one or two workgroups (measured: 121 us per
    call, 10 calls = 1.2 of the 2.0 ms the AWP's per-ray remainder took).  Here the rows are cut into up to 256 slabs, one batched product
    gives the slabs' partial gradients and a sum folds them: two launches of ~10 us."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        g2, x2 = gy.reshape(-1, gy.shape[-1]), x.reshape(-1, x.shape[-1])
        n = x2.shape[0]
        dx = (g2 @ w).reshape(x.shape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            G = 256
            while G > 1 and (n % G or n // G < 32):
                G //= 2
            dw = torch.bmm(g2.reshape(G, n // G, -1).transpose(1, 2), x2.reshape(G, n // G, -1)).sum(0) if G > 1 else g2.t() @ x2
        db = g2.sum(0) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return dx, dw, db


def _lin(x, w, b=None):
    return _SplitKLinear.apply(x, w, b)


class FusedAWP(torch.nn.Module):
    """Drop-in for the reference's AdaptiveWeightProposal in NeRFAll(awpnet=...): wraps THAT module (its parameters stay where the
    optimizer and the state dict expect them) and runs awp.py:79-117 with the per-sample part on the library:

        sample_feature_embed_layer   awp.py:98-100   -> evd_awp_embed_forward / _backward (reads the fine level's geo fragments)
        feature_integration          awp.py:102      -> evd_awp_feature_integration / _bwd
        MAM, per-sample part         mam.py:29-33,72-74 -> evd_mam_local_forward / _backward (one autograd node with the integration)
        motion embedding, rest of the MAM, w_linear, normalisation  awp.py:89-95,104-117, mam.py:35-53 -> evd_awp_tail_forward / _backward
                                     (tail_kernels=True and the reference's structure; else the wrapped module's own layers)

    `depth_feature` is the GeoFragments handle NeRFAll.forward_train passes when its awpnet is a FusedAWP (a float32 tensor
    [R P, S, 128] is accepted too)."""

    def __init__(self, awpnet, precision="f16", graph_per_ray=False, tail_kernels=True):
        """tail_kernels (default): the per-ray remainder (awp.py:89-95, 104-117, mam.py:35-53) on evd_awp_tail_forward / _backward -- two
        launches forward, three backward -- whenever the wrapped module has the reference's structure (Linear motion embedding of width 32,
        MotionAggregationModule with Corr / linear, BatchNorm1d with a momentum) and the ray's working set fits the LDS; otherwise, and with
        tail_kernels=False, the channel-last torch path below.
        graph_per_ray: run the PER-RAY remainder (awp.py:105-117 + mam.py:35-53: ~100 launches of [R, 32, P] / [R, 32, S]-sized
        tensors forward, twice that backward -- launch-bound, not compute-bound) as ONE captured hipGraph each way
        (torch.cuda.make_graphed_callables on a module that shares the wrapped module's parameters; built per (rays, P, S) shape on
        first use).  The arithmetic is the wrapped module's own; BatchNorm statistics update inside the graph."""
        super().__init__()
        self.ref = awpnet
        layers = list(awpnet.sample_feature_embed_layer)
        self.embed = SampleFeatureEmbed([l.weight for l in layers], [l.bias for l in layers], precision)
        self.output_ch = awpnet.output_ch
        self.graph_per_ray = bool(graph_per_ray)
        self._graphed = {}
        self.tail_kernels = bool(tail_kernels) and not self.graph_per_ray and os.environ.get("EVD_AWP_TAIL", "1") != "0" and self._tail_structure()
        self._F = _dir_freqs(awpnet.ray_dirs_embed_fn) if self.tail_kernels else None
        self._tail_refused = set()

    def _tail_structure(self):
        """does the wrapped module have the layers evd_awp_tail_* is built for (the shipped configs' AdaptiveWeightProposal)?"""
        m, nn = self.ref, torch.nn
        try:
            mot, corr, lin = list(m.motion_feature_embed_layer), m.MAM.Corr, m.MAM.linear
            bn = corr.convd[1]
            ok = 1 <= len(mot) <= 4 and all(isinstance(l, nn.Linear) and l.out_features == 32 and l.bias is not None for l in mot)
            ok = ok and all(l.in_features == 32 for l in mot[1:]) and self.embed.width == 64 and mot[0].in_features >= 64
            ok = ok and isinstance(lin, nn.Linear) and tuple(lin.weight.shape) == (32, 64) and lin.bias is not None
            ok = ok and all(tuple(getattr(corr, k).weight.shape) == (16, 32, 1) and getattr(corr, k).bias is None for k in ("conva", "convb", "convc"))
            ok = ok and all(tuple(getattr(corr, k).weight.shape) == (16, 16, 1) and getattr(corr, k).bias is None for k in ("convn", "convl"))
            ok = ok and tuple(corr.convd[0].weight.shape) == (32, 32, 1) and corr.convd[0].bias is None
            ok = ok and isinstance(bn, nn.BatchNorm1d) and bn.num_features == 32 and bn.affine and bn.momentum is not None
            ok = ok and isinstance(m.w_linear, nn.Linear) and tuple(m.w_linear.weight.shape) == (self.ref.output_ch, 32) and m.w_linear.bias is not None
            return bool(ok) and self.ref.output_ch <= 16
        except (AttributeError, IndexError, TypeError):
            return False

    def _tail_params(self):
        m = self.ref
        corr = m.MAM.Corr
        ps = [t for l in m.motion_feature_embed_layer for t in (l.weight, l.bias)] + [m.MAM.linear.weight, m.MAM.linear.bias]
        ps += [getattr(corr, k).weight for k in ("conva", "convb", "convc", "convn", "convl")]
        return ps + [corr.convd[0].weight, corr.convd[1].weight, corr.convd[1].bias, m.w_linear.weight, m.w_linear.bias]

    def _tail(self, h, view_feature, rays_d, h_inter, h_intra, n_ray, P, S):
        """-> the proposal weights [R, P] on the kernels, or None when the library refuses the shape (then the caller runs the torch path)"""
        m = self.ref
        bn = m.MAM.Corr.convd[1]
        vf, F = view_feature, self._F
        if F is None:                                                       # an encoding the kernel does not know: its columns as view_feature
            dirs = rays_d.reshape(n_ray, P, -1)[:, 0, :]
            enc = m.ray_dirs_embed_fn((dirs / torch.norm(dirs, dim=-1, keepdim=True)).reshape(-1, 3).float())
            vf, F = (enc if vf is None else torch.cat([vf, enc], dim=-1)), -1
        VF = 0 if vf is None else vf.shape[-1]
        n_mot = len(m.motion_feature_embed_layer)
        training = bn.training or bn.running_mean is None
        key = (P, S, VF, F, n_mot)
        if key in self._tail_refused or VF > 64 or m.motion_feature_embed_layer[0].in_features != 64 + VF + (3 + 6 * F if F >= 0 else 0):
            return None
        track = bn.running_mean is not None and (not training or bn.track_running_stats)
        bufs = (bn.running_mean, bn.running_var, bn.num_batches_tracked if training else None) if track else (None, None, None)
        cfg = (P, S, VF, F, n_mot, training, float(bn.eps), float(bn.momentum))
        try:
            return _AwpTail.apply(cfg, bufs, h, vf, rays_d, h_inter, h_intra, *self._tail_params())
        except L.EvdError as e:
            if "LDS" not in str(e):
                raise
            self._tail_refused.add(key)                                     # (the check runs before any launch: nothing was computed)
            return None

    @property
    def ccw_fine_scale(self):
        return self.ref.ccw_fine_scale

    def _flat(self):
        return torch.cat([t.reshape(-1) for l in self.ref.sample_feature_embed_layer for t in (l.weight, l.bias)])

    def _mam(self, x_global, h_local, n_ray, P, S):
        """MotionAggregationModule.forward (mam.py:66-83) with its per-sample part on the library (mam_local) and the per-ray remainder
        of CorrelationModule.forward (mam.py:35-53) on the wrapped module's own layers.  A MAM without the reference's structure
        (no `Corr` / `linear`) is simply called."""
        mam = self.ref.MAM
        corr, lin = getattr(mam, "Corr", None), getattr(mam, "linear", None)
        if corr is None or lin is None or h_local.shape[-1] != 64 or P > 16 or S > 512:
            return mam(x_global, h_local)
        F = torch.nn.functional
        h_inter, h_intra = mam_local(h_local, lin.weight, corr.line_conv_att.weight, n_ray, P, S)
        return self._mam_tail(x_global, h_inter, h_intra)

    def _mam_tail(self, x_global, h_inter, h_intra):
        """the per-ray remainder of CorrelationModule.forward (mam.py:35-53) with the wrapped module's own parameters, CHANNEL-LAST: every
        Conv1d there has kernel_size 1, i.e. it is a matrix product over the channel axis -- written as such ([R, P or S, C] @ W^T, no
        transposes, no convolution library: the module's own calls cost ~100 launches of layout shuffles and MIOpen kernels per
        direction for [1024, 32, 10]-sized tensors).  BatchNorm1d of convd (mam.py:24-27) normalises over (ray, position) per channel:
        F.batch_norm on the [R P, C] view is the same statistic, running estimates updated in training like the module does."""
        corr, lin = self.ref.MAM.Corr, self.ref.MAM.linear
        F = torch.nn.functional
        w = lambda conv: conv.weight.squeeze(-1)                                              # [out, in, 1] -> [out, in]
        k_inter = _lin(_lin(h_inter, lin.weight, lin.bias), w(corr.conva))                    # [R, P, mid]
        k_intra = _lin(_lin(h_intra, lin.weight, lin.bias), w(corr.convb))                    # [R, S, mid]
        q = _lin(x_global, w(corr.convc))                                                     # [R, P, mid]
        a_inter = torch.softmax(torch.bmm(q, k_inter.transpose(1, 2)), dim=-1)                # [R, P, P]
        a_intra = torch.softmax(torch.bmm(q, k_intra.transpose(1, 2)), dim=-1)                # [R, P, S]
        f = torch.cat([torch.bmm(a_inter, _lin(k_inter, w(corr.convn))), torch.bmm(a_intra, _lin(k_intra, w(corr.convl)))], dim=-1)
        y = _lin(f, w(corr.convd[0]))                                                         # [R, P, C]
        bn = corr.convd[1]
        training = bn.training or bn.running_mean is None
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        mom = bn.momentum if bn.momentum is not None else (1.0 / float(bn.num_batches_tracked) if bn.training and bn.track_running_stats else 0.0)
        y = F.batch_norm(y.reshape(-1, y.shape[-1]), bn.running_mean if (not bn.training or bn.track_running_stats) else None,
                         bn.running_var if (not bn.training or bn.track_running_stats) else None, bn.weight, bn.bias, training, mom, bn.eps).reshape(y.shape)
        return F.leaky_relu(x_global + y, negative_slope=0.2)

    def forward(self, depth_feature, z_vals, rays_d, view_feature):
        m, P = self.ref, self.output_ch
        n_ray, S = z_vals.shape[0] // P, z_vals.shape[-1]
        mam = m.MAM
        known = getattr(mam, "Corr", None) is not None and getattr(mam, "linear", None) is not None and self.embed.width == 64 and P <= 16 and S <= 512
        if self.tail_kernels and known and rays_d.is_cuda:                        # everything behind the embedding on the library
            h_local = self.embed(self._flat(), depth_feature).reshape(n_ray * P, S, self.embed.width)
            u = (mam.Corr.line_conv_att.weight.reshape(1, -1) @ mam.linear.weight).reshape(-1)      # the logit of a sample is (W^T v) . h + const
            h, h_inter, h_intra = _LocalConsumers.apply(h_local, z_vals.reshape(-1, S), rays_d.reshape(-1, 3), u, n_ray, P, S)
            h = h.reshape(n_ray, P, -1)
            out = self._tail(h, view_feature, rays_d, h_inter, h_intra, n_ray, P, S)
            if out is not None:
                return out
        dirs = rays_d.reshape(n_ray, P, -1)[:, 0, :]
        dirs = (dirs / torch.norm(dirs, dim=-1, keepdim=True)).reshape(-1, 3).float()
        view = m.ray_dirs_embed_fn(dirs)                                          # awp.py:89-95
        if view_feature is not None:
            view = torch.cat([view_feature, view], dim=-1)
        if self.tail_kernels and known and rays_d.is_cuda:                        # (the library refused the shape: the torch remainder)
            return self._per_ray(h, view, None, n_ray, P, S, h_inter, h_intra)
        graphed = self.graph_per_ray and known and view.is_cuda and torch.is_grad_enabled()
        if graphed:      # capture BEFORE this forward touches the module's parameters on the caller's stream (a live autograd graph that holds
            # their AccumulateGrad nodes on another stream breaks the capture of the backward graph)
            self._ensure_graph((n_ray, P, self.embed.width), tuple(view.shape), (n_ray, P, self.embed.width), (n_ray, S, self.embed.width), view.device)
        h_local = self.embed(self._flat(), depth_feature).reshape(n_ray * P, S, self.embed.width)     # awp.py:98-100
        h = feature_integration(h_local.reshape(n_ray, P, S, -1), z_vals, rays_d)                      # awp.py:102
        if graphed:
            h_inter, h_intra = mam_local(h_local, mam.linear.weight, mam.Corr.line_conv_att.weight, n_ray, P, S)
            return self._graphed_tail(h, view, h_inter, h_intra)
        return self._per_ray(h, view, h_local, n_ray, P, S)

    def _per_ray(self, h, view, h_local, n_ray, P, S, h_inter=None, h_intra=None):
        m = self.ref
        h = torch.cat([h, view.unsqueeze(1).repeat(1, P, 1)], dim=-1)
        for layer in m.motion_feature_embed_layer:                                # awp.py:107-109
            h = torch.relu(_lin(h, layer.weight, layer.bias) if isinstance(layer, torch.nn.Linear) else layer(h))
        h = self._mam(h, h_local, n_ray, P, S) if h_inter is None else self._mam_tail(h, h_inter, h_intra)      # awp.py:111
        h = h.mean(dim=1)                                                         # adaptive_avg_pool1d over the P positions (awp.py:112)
        w = torch.sigmoid(m.w_linear(h))
        return w / torch.sum(w, -1, keepdim=True)

    def _ensure_graph(self, sh_h, sh_view, sh_inter, sh_intra, device):
        key = (tuple(sh_h), tuple(sh_view), tuple(sh_inter), tuple(sh_intra), self.ref.training)
        if key not in self._graphed:
            outer = self

            class _Tail(torch.nn.Module):             # shares the wrapped module's parameters: make_graphed_callables treats them as graph inputs
                def __init__(self):
                    super().__init__()
                    self.ref = outer.ref

                def forward(self, h, view, h_inter, h_intra):
                    return outer._per_ray(h, view, None, h.shape[0], h.shape[1], h_intra.shape[1], h_inter, h_intra)

            tail = _Tail().train(self.ref.training)
            sample = tuple(torch.randn(sh, device=device).requires_grad_(True) for sh in (sh_h, sh_view, sh_inter, sh_intra))
            # make_graphed_callables warms the module up and captures it on these RANDOM inputs: the BatchNorm layers of the wrapped module
            # (MAM.Corr.convd) would take the random batches into running_mean / running_var / num_batches_tracked -- statistics that are
            # checkpointed.  They are put back afterwards; momentum=None (cumulative average: num_batches_tracked is read on the host inside the
            # capture) is rejected.
            bns = [m for m in self.ref.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
            if any(m.momentum is None for m in bns):
                raise L.EvdError("FusedAWP(graph_per_ray=True): BatchNorm with momentum=None cannot be captured (its average reads num_batches_tracked on the host)")
            saved = [{k: v.clone() for k, v in m._buffers.items() if v is not None} for m in bns]
            self._graphed[key] = torch.cuda.make_graphed_callables(tail, sample, allow_unused_input=True)   # (the embedding's parameters are not in this graph)
            with torch.no_grad():
                for m, sv in zip(bns, saved):
                    for k, v in sv.items():
                        m._buffers[k].copy_(v)
        return self._graphed[key]

    def _graphed_tail(self, h, view, h_inter, h_intra):
        fn = self._ensure_graph(h.shape, view.shape, h_inter.shape, h_intra.shape, h.device)
        req = lambda t: t if t.requires_grad else t.detach().requires_grad_(True)     # the capture saw inputs that require grad
        return fn(req(h.contiguous()), req(view.contiguous()), req(h_inter.contiguous()), req(h_intra.contiguous()))
