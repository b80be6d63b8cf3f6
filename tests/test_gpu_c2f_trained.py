"""GPU parity of the shipped PDRF configuration (mode='c2f', grids 293x293x195 / 586x586x390) on TRAINED parameters, and the compensated
float16 mode (EVD_PREC_F16C) of the fine level's networks (csrc/voxel_mlp_c_kernel.h).  VERDICT r2 item 1: the single-product
float16 mode holds north_star's 1e-4 RGB bound on seed-derived weights (7e-5) but not on trained ones (3.8e-4 fine / 2.3e-4 coarse after
3000 iterations of the library's own training path, tools/trained_c2f.py); f32, f16x3 and f16c must hold it on both.  The oracle is
oracle/evd_oracle.c (evo_render_c2f, reference renderer.py:182-217 + voxnerf.py), never another kernel of this library."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, maxabs
from evdeblurnerf_amd import weights as W

pytestmark = pytest.mark.gpu
AABB = W.BLURFACTORY_AABB


def T(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda")


def N(x):
    return x.detach().cpu().numpy()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def trained_c2f():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import trained_c2f as TC
    sd, rep = TC.train_c2f(iters=3000)
    print("trained c2f:", rep)
    return TC, sd, rep


def test_trained_c2f_every_mode_vs_oracle(O, trained_c2f):
    """4096 event rays x (64 + 64) samples through both trained levels: RGB L-inf (fine and coarse output) of every arithmetic mode against
    the oracle on 256 rays spread over the batch.  Bound 1e-4 for f32 / f16x3 / f16c (measured 5e-7 / 5e-7 / ~2e-5); the single-product modes
    are reported and bounded loosely -- they are throughput modes, and the test pins that f16c is what closes their gap."""
    TC, sd, rep = trained_c2f
    assert rep["loss_last"] < 0.01 * rep["loss_first"]                    # it did train (0.28 -> 6e-5)
    err, info = TC.c2f_parity(O, sd, ("f32", "f16x3", "f16c", "f16", "bf16"))
    print("RGB L-inf vs the oracle, trained c2f, 4096 x (64 + 64):", {k: {a: (f"{b:.2e}" if isinstance(b, float) else b) for a, b in v.items()} for k, v in err.items()}, info)
    assert info["rgb_std"] > 0.1                                           # a structured image, not the near-constant initial field
    # the arithmetic's error is held on the rays that carry the oracle's merged sample positions; a ray whose importance samples moved (the resampling is
    # ill-conditioned in nearly empty bins: conftest.z_mismatch; the parameters are trained in this session and differ from run to run) renders another
    # quadrature of the same field in EVERY mode, exact f32 included -- observed up to 6.4e-5 (all modes alike) in one of five runs
    for p in ("f32", "f16x3", "f16c"):
        assert err[p]["fine_on_the_oracles_samples"] < 1e-4 and err[p]["coarse"] < 1e-4, (p, err[p])
        assert err[p]["rays_with_moved_importance_samples"] <= 8, (p, err[p])                     # of 256
        assert err[p]["fine"] < (1e-4 if err[p]["rays_with_moved_importance_samples"] == 0 else 2e-3), (p, err[p])
    assert err["f16c"]["fine_on_the_oracles_samples"] < 0.5 * err["f16"]["fine"]      # (in the single-product modes every ray's samples move: their coarse weights are 1e-4-grade)
    # (reported, loosely bounded: the in-session training is not deterministic -- float atomics -- and the single-product modes' worst ray
    # moves between 2e-4 and 8e-3 from run to run, where one importance sample lands on the other side of a surface)
    assert err["f16"]["fine"] < 5e-2 and err["bf16"]["fine"] < 1e-1


def test_trained_c2f_full_frame_f16c_vs_oracle(O, trained_c2f):
    """BASELINE config 5 on the trained parameters: one 400 x 400 view (160 000 rays, 64 + 128 samples, rays generated on the device) in the
    compensated mode against the oracle on 256 rays of the frame: <= 1e-4."""
    from evdeblurnerf_amd.rays import get_rays
    TC, sd, _ = trained_c2f
    K = W.synthetic_camera()
    c2w = W.synthetic_pose(40)[:3, :4].astype(np.float32)
    o, d = get_rays(400, 400, K, T(c2w))
    rays = N(torch.stack([o, d], -1).reshape(-1, 3, 2))
    err, _ = TC.c2f_parity(O, sd, ("f16c", "f16"), Ni=128, rays=rays)
    print("RGB L-inf vs the oracle, trained c2f, full frame 64 + 128:", err)
    assert err["f16c"]["fine_on_the_oracles_samples"] < 1e-4 and err["f16c"]["coarse"] < 1e-4
    assert err["f16c"]["rays_with_moved_importance_samples"] <= 8 and err["f16c"]["fine"] < (1e-4 if err["f16c"]["rays_with_moved_importance_samples"] == 0 else 2e-3)


def _fine_level(seed, nvox=48 ** 3, bias=True):
    from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures
    gsz = W.pdrf_grid_size(AABB[0], AABB[1], nvox)
    sd = W.make_pdrf_state_dict(seed, gsz, input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=bias)
    net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=127, app_dim=32,
                                  app_n_comp=(64, 16, 16), n_voxels=nvox)
    return net, sd


def _level_inputs(rs, R, S, scale=0.3):
    pts = rs.uniform(-1, 1, (R, S, 3)).astype(np.float32)
    d = rs.normal(size=(R, 3))
    vd = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    fts = (scale * rs.normal(size=(R, S, 64))).astype(np.float32)
    z = np.sort(rs.uniform(0, 1, (R, S)).astype(np.float32), -1)
    rd = rs.uniform(-1, 1, (R, 3)).astype(np.float32)
    return T(pts), T(vd), T(fts), T(z), T(rd)


def test_fine_level_f16c_ragged_sizes_repack_and_rejections():
    """The fine level's networks alone (evd_voxel_forward) in the compensated mode against the exact-float32 kernel at sizes that are not
    multiples of its 128-sample workgroups, with colour biases, with large feature magnitudes (block scales), after a device re-pack of new
    parameters (lazy: bit-identical to a fresh handle), and its rejections."""
    from evdeblurnerf_amd import _lib as L
    net, sd = _fine_level(71)
    rs = np.random.RandomState(11)
    for R, S in ((1, 1), (1, 128), (3, 43), (70, 33), (129, 127), (531, 131), (258, 128)):       # the last one: several tiles per persistent workgroup, ragged end
        a = _level_inputs(rs, R, S)
        ref = net.forward(*a, precision="f32")
        got = net.forward(*a, precision="f16c")
        e16 = maxabs(N(net.forward(*a, precision="f16")[0]), N(ref[0]))
        ec = maxabs(N(got[0]), N(ref[0]))
        assert got[4] is None and ec < 2e-5, (R, S, ec)
        assert maxabs(N(got[3]), N(ref[3])) < 2e-5                      # compositing weights (the sigma head)
        if R * S > 1000:
            assert ec < 0.3 * e16, (ec, e16)
    a = _level_inputs(rs, 64, 64, scale=30.0)                           # features of magnitude ~100: finite, and still close
    ref, got = net.forward(*a, precision="f32"), net.forward(*a, precision="f16c")
    assert torch.isfinite(got[0]).all() and maxabs(N(got[0]), N(ref[0])) < 1e-3
    # device re-pack (evd_voxel_load_params keeps the values, the next f16c launch re-packs): equal to a handle created from those values
    net2, sd2 = _fine_level(72)
    a = _level_inputs(rs, 70, 33)
    fresh = N(net2.forward(*a, precision="f16c")[0])
    before = N(net.forward(*a, precision="f16c")[0])
    net.load_params(net.flat_params(sd2).detach())
    net.load_grids([g.detach() for g in net2.grid_params()])
    again = N(net.forward(*a, precision="f16c")[0])
    assert not np.array_equal(before, fresh) and np.array_equal(again, fresh)
    # the coarse level has no compensated kernel: it runs float32-grade (f16x3) under the same mode name
    from evdeblurnerf_amd.voxnerf import VoxelNeRFRayFeatures
    gsz = W.pdrf_grid_size(AABB[0], AABB[1], 24 ** 3)
    sdc = W.make_pdrf_state_dict(5, gsz, input_ch=95, hidden_dim=64, geo_feat_dim=15)
    coarse = VoxelNeRFRayFeatures(sdc, "", AABB, n_voxels=24 ** 3)
    pts, vd, _, z, rd = _level_inputs(rs, 40, 17)
    ftc = T((0.3 * rs.normal(size=(40, 17, 32))).astype(np.float32))
    assert torch.equal(coarse.forward(pts, vd, ftc, z, rd, precision="f16c")[0], coarse.forward(pts, vd, ftc, z, rd, precision="f16x3")[0])
    # training in this mode (round 4: tests/test_gpu_train_f16c.py) keeps the geo features as fragments: float32 feature rows are rejected
    with pytest.raises(L.EvdError):
        net.mlpforward_train(a[0], a[1], a[2], precision="f16c", want_feature=True)


def test_gather_into_a_row_window_and_merge_in_place():
    """The training render lets the fine gather write its features straight into columns 32.. of the merged row buffer and scatters the
    gradient from there (renderer._MergeFeatures with `rows`, VoxelNeRFSampleFeatures.sample(out=window)): values and every gradient
    (grids, coarse features, sample positions) equal the path that gathers into its own tensor and copies; bad windows are rejected."""
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.renderer import _MergeFeatures, _window
    net, sd = _fine_level(73)
    rs = np.random.RandomState(5)
    R, S, Nn, Fc = 37, 20, 13, 32
    pts = T(rs.uniform(-1, 1, (R, S + Nn, 3)).astype(np.float32))
    order = T(np.stack([rs.permutation(S + Nn) for _ in range(R)]).astype(np.int32))
    g_out = T(rs.standard_normal((R, S + Nn, Fc + net.app_dim)).astype(np.float32))

    def run(placed):
        grids = [g.detach().clone().requires_grad_(True) for g in net.grid_params()]
        ft0 = T(rs0.standard_normal((R, S, Fc)).astype(np.float32)).requires_grad_(True)
        ftn = T(rs0.standard_normal((R, Nn, Fc)).astype(np.float32)).requires_grad_(True)
        p = pts.clone().requires_grad_(True)
        if placed:
            rows = torch.empty((R, S + Nn, Fc + net.app_dim), dtype=torch.float32, device=pts.device)
            out = _MergeFeatures.apply(ft0, ftn, order, net.sample_train(p, grids, None, out=_window(rows, Fc, net.app_dim)), rows)
            assert out.data_ptr() == rows.data_ptr()
        else:
            out = _MergeFeatures.apply(ft0, ftn, order, net.sample_train(p, grids, None))
        (out * g_out).sum().backward()
        return [N(out.detach())] + [N(t.grad) for t in [ft0, ftn, p] + grids]

    res = []
    for placed in (False, True):
        rs0 = np.random.RandomState(9)
        res.append(run(placed))
    assert np.array_equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert maxabs(a, b) <= 1e-5 * max(1.0, float(np.abs(a).max()))          # the scatter sums in a non-deterministic order
    # plain sample() into a window == sample() into its own tensor; windows the kernel cannot write are refused
    rows = torch.zeros((R, S + Nn, 80), dtype=torch.float32, device=pts.device)
    with torch.no_grad():
        w = net.sample(pts, None, out=_window(rows, 48, net.app_dim))
        assert torch.equal(w, net.sample(pts, None)) and torch.equal(rows[..., 48:], w) and float(rows[..., :48].abs().max()) == 0.0
        with pytest.raises(L.EvdError):
            net.sample(pts, None, out=rows[:, :, :net.app_dim].transpose(0, 1))
        with pytest.raises(L.EvdError):
            net.sample(pts, None, out=torch.empty((R, S + Nn, net.app_dim), dtype=torch.float64, device=pts.device))


@pytest.mark.parametrize("prec", ["f16x3", "f16"])
def test_resident_coarse_level_equals_the_streaming_kernel(prec, tmp_path):
    """The 64-wide level's render pass on the weight stream resident in LDS (k_voxel_mlp_resident, round 6: persistent workgroups, no ring, no
    barrier) keeps the streaming kernel's layer table, arithmetic and order of operations: colour, depth, acc and weights of 4096 x 64 samples are
    BIT FOR BIT those of k_voxel_mlp_pipe (EVD_COARSE_FORM=pipe; the switch is read once per process, hence the two subprocesses)."""
    import subprocess
    outs = []
    for form in ("resident", "pipe"):
        env = dict(os.environ)
        env.pop("EVD_COARSE_FORM", None)
        if form == "pipe":
            env["EVD_COARSE_FORM"] = "pipe"
        out = tmp_path / f"{form}.npy"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "coarse_form_check.py"), str(out), prec], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert f"[{form}, {prec}] 4096 x 64" in r.stdout, r.stdout
        outs.append(np.load(out))
    a, b = outs
    assert a.size == b.size == 4096 * (3 + 1 + 1 + 64) and np.isfinite(a).all()
    assert (a.view(np.uint32) == b.view(np.uint32)).all(), float(np.abs(a - b).max())
