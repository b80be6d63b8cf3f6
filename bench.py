#!/usr/bin/env python
"""Benchmark of the hot path: rays/s for 4096 rays x 128 samples through the 8x256 NeRF MLP (PE + MLP + composite).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f16|f16x3|f32|bf16]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this script its own launcher: it re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU over RCCL (when a
launcher already set RANK / LOCAL_RANK / WORLD_SIZE, e.g. the driver's torchrun command, it just joins as that rank).
EVD_BENCH_SHARE_GPU=1 (validation of the N > 1 code path on a 1-GPU box) puts every rank on device 0 with gloo collectives.

One JSON line on rank 0 (contract in the task statement). A "step" = one pass of the render path
(ray packing -> z stratification -> fused PE+MLP kernel -> compositing scan) over one batch of 4096
synthetic LLFF-shaped rays already resident in HBM; with N > 1 every rank renders its own 4096 rays
(weak scaling) and the step ends with the packed blur-loss partial all-reduce over RCCL, the only
exchange the path has. Beside the contract keys:

  roofline       the dominant kernel (k_nerf_mlp), HIP events on the launch stream
  modes          the same kernel in every arithmetic mode
  parity         RGB L-inf of EVERY mode against the CPU oracle at the full 4096 x 128 size, on the seed-derived weights
                 and on weights trained in this run (tools/trained_weights.py)
  c2f            the shipped blurfactory configuration (PDRF levels at their real grid sizes): render step + roofline of
                 its dominant kernel, the tri-plane gather k_voxel_sample
  strong_scaling one 400x400 frame (160 000 rays, 64 + 128 samples, BASELINE config 5) with rows sharded over the ranks
  composite      the HBM-bound compositing scan on 2^20 rays;  train: the training kernels;  cpu_baseline: the C oracle
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 2 * 593408          # GEMM terms of the 8x256 net with skip and view branch
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f16c": 2500.0, "f16x3": 2500.0, "f32": 157.3}   # dense MFMA peaks (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0                 # aggregate L2 (MI355X_MICROARCH.md, "L2 (per XCD)")
DTYPE_NAME = {"f16c": "f16c (compensated f16: one f16 MFMA product + two block-scaled fp6 MFMA products of the operands' rounding residuals; f32 accumulate; ~2^-15 operands)",
              "f16": "f16 (MFMA operands; f32 accumulate)", "f16x3": "f16x3-split (3 f16 MFMA products, f32 accumulate; f32-grade)",
              "f32": "f32", "bf16": "bf16 (MFMA operands; f32 accumulate)"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="f16c", choices=["f16c", "f16", "f16x3", "f32", "bf16"],
                    help="arithmetic mode of the headline; the default is the fastest mode that holds the 1e-4 RGB bound on trained weights")
    ap.add_argument("--settle", type=int, default=300, help="untimed steps before the W warm-up steps (clock / allocator settle, ~0.3 s; 0 = none)")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--train-iters", type=int, default=1000, help="iterations of the in-run training that makes the 'trained' weights")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-zero-probe", action="store_true", help="skip the all-zero-data launches of the headline kernel (a run under rocprofv3 whose per-kernel "
                                                                    "average is to be the real-data average: tools/final_run.sh)")
    ap.add_argument("--no-modes", action="store_true")
    ap.add_argument("--no-composite", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-c2f", action="store_true")
    ap.add_argument("--no-awp", action="store_true")
    ap.add_argument("--no-strong", action="store_true")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a launcher's environment
# ---------------------------------------------------------------------------------------------------------------------

def launch_command(n, argv, port=None, script=None):
    """The command line `bench.py --gpus n` re-executes itself with (one rank per GPU; the same form the driver uses)."""
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def needs_launch(a, env=None):
    env = os.environ if env is None else env
    return a.gpus > 1 and "WORLD_SIZE" not in env


def launch(a, argv):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    env["EVD_BENCH_SELF_LAUNCH"] = "1"
    return subprocess.call(launch_command(a.gpus, argv), env=env)


# ---------------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------------

def nerf_args(n_importance=0):
    from types import SimpleNamespace
    return SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=n_importance)


def make_model(precision, sd=None):
    from evdeblurnerf_amd import weights as W
    from evdeblurnerf_amd.renderer import NeRFAll
    sd = sd if sd is not None else W.prefixed(W.make_nerf_state_dict(21), "mlp_coarse")
    return NeRFAll(nerf_args(), sd, precision=precision).eval(), sd


def _sync():
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def init_ranks():
    """(rank, world, local, dist | None, backend | None, device) from the launcher's environment; one process per GPU with RCCL
    ("nccl"), or -- EVD_BENCH_SHARE_GPU=1 / no GPU at all (the CPU test of this scaffolding) -- gloo."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    have_gpu = torch.cuda.is_available()
    share = os.environ.get("EVD_BENCH_SHARE_GPU") == "1"
    if have_gpu:
        if world > torch.cuda.device_count() and not share:
            raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s) "
                             "(EVD_BENCH_SHARE_GPU=1 shares device 0 for validation)")
        torch.cuda.set_device(0 if share else local)
    dist, backend = None, None
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if (have_gpu and not share) else "gloo"
        dist_.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_
    return rank, world, local, dist, backend, ("cuda" if have_gpu else "cpu")


def rank_devices(dist, world):
    """one entry per rank: the device it computes on (index, name, UUID, PCI bus) -- a scaling record shows N distinct GPUs by N distinct
    UUIDs (two ranks sharing a device under EVD_BENCH_SHARE_GPU=1 show the same one twice)"""
    import torch
    if not torch.cuda.is_available():
        return []
    i = torch.cuda.current_device()
    pr = torch.cuda.get_device_properties(i)
    me = {"rank": int(os.environ.get("RANK", "0")), "device": i, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
          "pci_bus_id": getattr(pr, "pci_bus_id", None)}
    if dist is None or world == 1:
        return [me]
    out = [None] * world
    dist.all_gather_object(out, me)
    return out


def time_steps(fn, steps, warmup, dist=None, drain=None, device="cuda", settle_steps=0):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device synchronize on both sides;
    returns the MAX over ranks of the elapsed seconds."""
    import torch
    barrier = (lambda: dist.barrier()) if dist else (lambda: None)
    import gc
    gc.collect()
    gc.disable()                 # the timed region is tens of milliseconds: one collector pause inside it would be a visible fraction
    # settle (untimed, before the W contractual warm-up steps): a box that just finished importing torch runs its first ~0.2 s of kernels
    # at ramping clocks and with a cold allocator -- the driver's 5 warm-up steps are 5 ms, and its first timed steps were 10 % slow
    # (0.947 ms mean vs 0.85 ms median).  Same steps, not counted; a fixed count, so that every rank issues the same collectives.
    for _ in range(settle_steps):
        fn()
    _sync()
    for _ in range(warmup):
        fn()
    barrier()
    _sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if drain:
        drain()
    barrier()
    _sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def headline(rays_per_rank, samples, world, steps, warmup, dt, precision, settle=0):
    """the contract keys of the JSON line (value = whole-job rays/s over all ranks)"""
    return {
        "metric": "rays/sec (4096x128 samples, 8x256 MLP)", "value": world * rays_per_rank * steps / dt, "unit": "rays/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "settle": settle, "ms_per_step": 1e3 * dt / steps,
        "protocol": f"{settle} untimed settle steps (clock / allocator; --settle), then the {warmup} warm-up steps, then exactly {steps} timed steps "
                    "between barrier + device synchronize",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_NAME[precision],
        "data": "synthetic",
        "config": {"workload": f"nerf8x256 render: {rays_per_rank} rays x {samples} samples per GPU, PE(10,4)+MLP+composite, ndc, viewdirs",
                   "rays_per_gpu": rays_per_rank, "samples": samples, "precision": precision},
    }


def kernel_ms(fn, steps, warmup=3):
    """Average duration of one launch, HIP events on the stream the kernel runs on (torch's current stream)."""
    import torch
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / steps


def per_step_ms(fn, steps):
    """HIP-event duration of every single step (for the median: steadier than one wall-clock mean over a 10-30 ms region)"""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    ev[-1].synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))


def oracle_parity(sd, rays, S, K, precisions):
    """RGB L-inf of each arithmetic mode's full render against the CPU oracle (oracle/evd_oracle.c) on the SAME rays and weights."""
    import numpy as np
    import torch
    from oracle import oracle as O
    ref = O.render_nerf(O.Nerf(sd, "mlp_coarse."), None, O.make_cfg(N_samples=S), rays.cpu().numpy())["rgb"]
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=S, N_importance=0, retraw=False)
    out = {}
    for prec in precisions:
        m, _ = make_model(prec, sd)
        rgb = m.render(400, 400, K, rays=rays, **kw)[0]
        torch.cuda.synchronize()
        out[prec] = float(np.abs(rgb.cpu().numpy().astype(np.float64) - ref).max())
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the shipped configuration: PDRF coarse-to-fine levels at the blurfactory grid sizes
# ---------------------------------------------------------------------------------------------------------------------

def c2f_parity(precision):
    """RGB L-inf of every arithmetic mode of the c2f render vs the CPU oracle (256 rays of 4096, 64 + 64 samples) on the seed-derived
    blurfactory parameters AND on parameters trained in this run (tools/trained_c2f.py: 3000 iterations of the library's own training
    path, ~22 s), plus one 400 x 400 frame (64 + 128) in the headline mode on the trained parameters."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import trained_c2f as TC
    from evdeblurnerf_amd import weights as W
    from evdeblurnerf_amd.rays import get_rays
    from oracle import oracle as O
    modes = ("f32", "f16x3", "f16c", "f16", "bf16")
    out = {"bound": 1e-4, "reference": "oracle/evd_oracle.c evo_render_c2f on 256 rays spread over the batch, same rays and parameters", "modes": {}}
    seed, _ = TC.c2f_parity(O, W.make_blurfactory_state_dict(31, sigma_gain=3.0), modes)
    sd, rep = TC.train_c2f(iters=3000)
    trained, info = TC.c2f_parity(O, sd, modes)
    K = W.synthetic_camera()
    c2w = torch.as_tensor(W.synthetic_pose(40)[:3, :4].astype(np.float32), device="cuda")
    o, d = get_rays(400, 400, K, c2w)
    frame, _ = TC.c2f_parity(O, sd, (precision,), Ni=128, rays=torch.stack([o, d], -1).reshape(-1, 3, 2).cpu().numpy())
    out["rgb_linf_vs_oracle"] = {"seed_parameters_4096x(64+64)": seed, "trained_parameters_4096x(64+64)": trained,
                                 "trained_parameters_frame_400x400_(64+128)": frame}
    out["trained_parameters"] = dict(rep, rgb_std_of_the_render=info["rgb_std"])
    out["headline_mode"] = precision
    out["headline_within_bound"] = {k: max(v[precision]["fine"], v[precision]["coarse"]) <= 1e-4 for k, v in out["rgb_linf_vs_oracle"].items()}
    out["note"] = ("fine / coarse: max over ALL compared rays; rays_with_moved_importance_samples: rays whose merged sample positions differ from the oracle's by more "
                   "than 5e-6 -- the inverse-CDF resampling is ill-conditioned in nearly empty bins, such a ray shows the same error in EVERY mode incl. exact f32 "
                   "(the parameters are trained in this run and differ from run to run) --; fine_on_the_oracles_samples: the arithmetic's own error")
    del sd
    torch.cuda.empty_cache()
    return out


def c2f_leg(precision, steps, parity=True):
    """BASELINE configs 2/3: one c2f render step (4096 event rays, 64 + 64 samples) and the roofline of its dominant kernel,
    the tri-plane gather (k_voxel_sample) of the FINE level over the step's 4096 x 128 merged samples."""
    import torch
    from evdeblurnerf_amd import weights as W
    from evdeblurnerf_amd.renderer import NeRFAll
    sd = W.make_blurfactory_state_dict(31)
    model = NeRFAll(W.blurfactory_args(64), sd, precision=precision).eval()
    del sd
    K = W.synthetic_camera()
    R = 4096
    rays = torch.as_tensor(W.synthetic_rays(5, R), device="cuda")
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=64, retraw=False, perturb=0., raw_noise_std=0.)
    step_ms = kernel_ms(lambda: model.render(400, 400, K, rays=rays, **kw), steps)
    # the gather alone, on the points of such a step: NDC points of the same rays at 128 depths
    rb = NeRFAll.ray_batch_train(400, 400, K, rays)
    z = torch.linspace(0, 1, 128, device="cuda")
    pts = (rb[:, None, 0:3] + rb[:, None, 3:6] * z[None, :, None]).contiguous()
    fine = model.mlp_fine
    # the single-product half-precision modes gather float16 copies of the grids, and so does the FINE level of the compensated mode
    # (evd_voxel_api.hip sample_for: its rounding costs nothing measurable there; the coarse level, which places the importance samples, stays float32)
    half_grids = precision in ("f16", "bf16", "f16c")
    g_ms = kernel_ms(lambda: fine.sample(pts, precision if half_grids else None), steps)
    n = R * 128
    esz = 2 if half_grids else 4
    taps = 4 * 96 + 2 * 96                                      # 4 plane taps + 2 line taps x 96 channels = 576 gathered values per sample
    gathered = n * taps * esz
    # PMC of THIS kernel build (profiles/r06_pmc_voxel.json, made by tools/pmc_voxel.sh: rocprofv3 cannot run inside the timed process; float16 grids:
    # k_voxel_sample_m, collected in round 6; float32 grids: k_voxel_sample_w, unchanged since round 4), per sample of the fine-level launch: 128-byte
    # line READ requests at the L2 and the lines that miss it (served by the Infinity Cache: the 165 MB of grids exceed the 32 MB of L2).  The kernel
    # is a random gather: its roofline is the RATE at which the chip serves such requests, measured by tools/probes/gather_probe.hip (random 128-byte
    # records, 16-byte lane loads), not an HBM byte rate.
    pj = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_voxel.json")))["f16_grids" if half_grids else "f32_grids"]
    pmc = {"l2_requests_per_sample": pj["l2_read_requests_per_sample"], "l2_miss_lines_per_sample": pj["l2_miss_lines_per_sample"],
           "fetch_size_bytes_per_sample": pj["fetch_size_bytes_per_sample"], "source": pj["file"] + " (committed profile of this kernel; the "
           "fraction below is counter-derived requests x the launch duration measured in this run)"}
    ceil = {"l2_resident": 146e9, "infinity_cache": 58e9, "hbm": 54e9}
    req_rate = pmc["l2_requests_per_sample"] * n / (g_ms * 1e-3)
    miss_rate = pmc["l2_miss_lines_per_sample"] * n / (g_ms * 1e-3)
    out = {"workload": "blurfactory c2f render: 4096 rays x (64 coarse + 64 importance) samples, grids 293x293x195 / 586x586x390, n_comp (64,16,16)",
           "precision": precision, "ms_per_step": step_ms, "rays_per_s": R / (step_ms * 1e-3),
           "roofline": {"kernel": "%s (fine level, 4096 x 128 samples, %s grids)" % (pj["kernel"], "float16" if half_grids else "float32"),
                        "bound": "l2-gather (128-byte line requests/s)", "kernel_ms": g_ms,
                        "achieved": req_rate / 1e9, "peak": ceil["l2_resident"] / 1e9, "unit": "G line requests/s", "frac": req_rate / ceil["l2_resident"],
                        "behind_l2": {"achieved": miss_rate / 1e9, "peak": ceil["infinity_cache"] / 1e9, "unit": "G lines/s", "frac": miss_rate / ceil["infinity_cache"]},
                        "gathered_bytes": gathered, "gathered_GBps": gathered / (g_ms * 1e-3) / 1e9, "pmc": pmc,
                        "gather_ceiling_lines_per_s": dict(ceil, probe="tools/probes/gather_probe.hip (128-byte records, 16-byte lane loads)"),
                        "note": "achieved = PMC-counted L2 line requests per sample x samples / the launch duration measured here, against the rate a bare "
                                "random gather of 128-byte records sustains when its table is L2-resident; behind_l2 = the lines that miss L2 against the "
                                "Infinity-Cache-resident ceiling.  576 grid values are gathered per sample (4 taps x 96 plane channels + 2 taps x 96 line "
                                "channels); not bandwidth-bound: k_voxel_sample_w is a latency chain per wavefront (points -> tap geometry -> gather -> basis GEMM -> "
                                "store), k_voxel_sample_m (float16 grids, round 6) is bound by the instructions its three wavefronts per SIMD issue (DESIGN.md 3.3)"}}
    out["arithmetic"] = {"f16c": "fine level: compensated float16 (k_voxel_mlp_c: f16 MFMA + two block-scaled fp6 MFMA residual products); coarse 64-wide level: "
                                 "float32-grade f16x3 products on the float32 grids (k_voxel_mlp_resident: weight stream resident in LDS; its encodings' sines on the "
                                 "hardware unit behind the two-float revolution reduction, as in the fine level's kernel); the fine level gathers the float16 copies of its grids",
                         "f16": "single-product float16 MFMA on both levels, float16 grid copies", "bf16": "bf16 MFMA on both levels, float16 grid copies",
                         "f16x3": "three float16 MFMA products per MAC on both levels, float32 grids", "f32": "exact float32 MFMA, float32 grids"}[precision]
    if parity:
        del model
        torch.cuda.empty_cache()
        out["parity"] = c2f_parity(precision)
        model = NeRFAll(W.blurfactory_args(64), W.make_blurfactory_state_dict(31), precision=precision).eval()
    return out, model


def awp_leg(precision):
    """the AWP consumer's per-sample part at the blurfactory blur-batch shape (10 240 sub-exposure rays x 128 samples):
    what it adds to the fine level's training forward + backward as torch Linear layers on depth_feature (the reference's way) and
    fused on the level's geo fragments (evd_awp_embed_forward / _backward), and the HBM roofline of the embedding kernel."""
    import numpy as np
    import torch
    from evdeblurnerf_amd import weights as W
    from evdeblurnerf_amd.awp import SampleFeatureEmbed, feature_integration
    from evdeblurnerf_amd.voxnerf import GeoFragments, VoxelNeRFSampleFeatures
    aabb = ((-1.5, -1.5, -1.0), (1.5, 1.5, 1.0))
    R, S, dev, nvox = 10240, 128, "cuda", 48 ** 3
    sd = W.make_pdrf_state_dict(71, W.pdrf_grid_size(aabb[0], aabb[1], nvox), input_ch=64 + 63, hidden_dim=256, geo_feat_dim=128, add_bias_color=True)
    net = VoxelNeRFSampleFeatures(sd, "", aabb, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=64 + 63, app_dim=32,
                                  app_n_comp=(64, 16, 16), n_voxels=nvox, precision=precision)
    flat = net.flat_params(sd)
    esd = W.make_awp_embed_state_dict(211)
    ws = [esd[f"sample_feature_embed_layer.{l}.weight"] for l in range(4)]
    bs = [esd[f"sample_feature_embed_layer.{l}.bias"] for l in range(4)]
    rows_ok = precision in ("f16", "bf16", "f16x3")            # (the mixed training modes f16c / f16m keep the geo features as float16 fragments only)
    emb = SampleFeatureEmbed(ws, bs, precision=precision if rows_ok and precision != "f16x3" else "f16")
    eflat = torch.cat([torch.tensor(t).reshape(-1) for l in range(4) for t in (ws[l], bs[l])]).to(dev).requires_grad_(True)
    lin = torch.nn.ModuleList([torch.nn.Linear(128, 64)] + [torch.nn.Linear(64, 64) for _ in range(3)]).to(dev)
    rs = np.random.RandomState(0)
    pts = torch.tensor(rs.uniform(-1, 1, (R, S, 3)).astype(np.float32), device=dev)
    d = rs.normal(size=(R, 3))
    vd = torch.tensor((d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32), device=dev)
    fts = torch.tensor((0.3 * rs.normal(size=(R, S, 64))).astype(np.float32), device=dev)
    z = torch.sort(torch.rand((R, S), device=dev), -1)[0]
    rd = torch.randn((R, 3), device=dev)
    gh, graw = torch.randn((R, 64), device=dev) * 1e-3, torch.randn((R, S, 4), device=dev) * 1e-3

    def level_only():
        (net.mlp_train(flat, pts, vd, fts) * graw).sum().backward()

    def torch_path():
        raw, feat = net.mlp_train(flat, pts, vd, fts, want_feature=True)
        h = feat
        for layer in lin:
            h = torch.relu(layer(h))
        ((raw * graw).sum() + (feature_integration(h.reshape(R, 1, S, 64), z, rd).reshape(R, 64) * gh).sum()).backward()

    def fused_path():
        geo = GeoFragments()
        raw, geo.token = net.mlp_train(flat, pts, vd, fts, want_feature=geo)
        ((raw * graw).sum() + (feature_integration(emb(eflat, geo).reshape(R, 1, S, 64), z, rd).reshape(R, 64) * gh).sum()).backward()

    # (medians of per-step HIP-event times after a warm-up, the allocator's cache dropped first: behind the c2f leg's allocations the first
    # fused steps can run into the caching allocator re-growing 5 GB of stores -- a mean over five steps once reported 24 ms for a 4.2 ms path)
    import statistics
    torch.cuda.empty_cache()

    def med(fn):
        for _ in range(3):
            fn()
        return statistics.median(per_step_ms(fn, 7))
    t0, t1, t2 = med(level_only), (med(torch_path) if rows_ok else None), med(fused_path)
    geo = GeoFragments()
    with torch.no_grad():
        _, geo.token = net.mlp_train(flat.detach().requires_grad_(True), pts, vd, fts, want_feature=geo)
    emb.load_params(eflat)
    from evdeblurnerf_amd.awp import _SampleEmbed
    k_ms = kernel_ms(lambda: _SampleEmbed.apply(geo.token, eflat.detach(), emb, geo), 10)
    n = R * S
    tile_bytes = (8 + 8 + 16) * 1024 + 32 * 64 * 4            # geo fragments in, geo copy + 4 activations out (training), h_local rows out
    algo = (n // 32) * tile_bytes
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import bench_mam
    mam = bench_mam.run(1024, 10, 128)                           # the MotionAggregationModule's per-sample part (mam.py:72-74, 29-33)
    mam["roofline"] = {"kernel": "k_mam_local_fwd / k_mam_local_bwd", "bound": "hbm", "algorithmic_bytes": 2 * mam["h_local_bytes"],
                       "achieved": [mam["fwd_GBps"], mam["bwd_GBps"]], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": [mam["fwd_GBps"] / HBM_PEAK_GBS, mam["bwd_GBps"] / HBM_PEAK_GBS],
                       "note": "forward: h_local read twice (logits, then the two softmax-weighted sums); backward: read once, d h_local written once"}
    import bench_awp_tail
    return {"workload": "AWP consumer, blurfactory blur batch: 10 240 sub-exposure rays x 128 samples, sample_feature_embed_layer 128-64-64-64-64 + feature_integration",
            "mam_per_sample_part": mam,
            "per_ray_remainder": bench_awp_tail.run(),      # awp.py:89-95, 104-117 + mam.py:35-53 on evd_awp_tail_* vs the channel-last torch path
            "precision": precision, "fine_level_fwd_bwd_ms": t0, "with_awp_torch_linear_on_depth_feature_ms": t1, "with_awp_fused_on_geo_fragments_ms": t2,
            "awp_addon_ms": {"torch": (t1 - t0) if t1 is not None else None, "fused": t2 - t0}, "depth_feature_tensor_avoided_bytes": n * 128 * 4,
            "roofline": {"kernel": "k_awp_embed (training forward, geo fragments in)", "bound": "hbm", "kernel_ms": k_ms, "algorithmic_bytes": algo,
                         "achieved": algo / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "per 32-sample tile: 8 KiB geo fragments read from the fine level's store; 8 KiB geo copy + 16 KiB activation "
                                 "fragments + 8 KiB float32 h_local rows written; 40 MFMAs per wavefront (arithmetic intensity ~ 25 FLOP/B: HBM-bound)"}}


def train_iteration_leg(precision):
    """One WHOLE blurfactory training iteration (tools/bench_train_step.py: blur batch of 1024 pixels x 10 sub-exposure rays + 2 x 4096
    event rays, 64 + 64 samples, fused losses, TV, backward on the hand-written kernels, Adam, parameter re-pack) in the training modes,
    with the parity of each mode next to its time, and the roofline of its dominant kernel, the tri-plane scatter AS SHIPPED: atomic
    requests at the L2 (PMC) over the launch duration, against the rate of a bare atomic kernel."""
    import ctypes as C
    import types
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import bench_train_step as BT
    import train_parity as TP
    from evdeblurnerf_amd import _lib as L, weights as W
    from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures, _grid_grads
    ms, nrays, _ = BT.run(types.SimpleNamespace(precision=precision, iters=10, pixels=1024, events=4096, P=10))
    torch.cuda.empty_cache()
    modes = {precision: ms}
    for other in ("f16", "f16c", "f16m"):
        if other != precision:
            modes[other], _, _ = BT.run(types.SimpleNamespace(precision=other, iters=8, pixels=1024, events=4096, P=10))
            torch.cuda.empty_cache()
    # integration option: the event batch's start and end rays rendered in ONE call (run_nerf.py:534,547 calls nerf() twice; a ray's colour does
    # not depend on its batch): fewer, larger launches.  Reported beside the figure of the reference's own call pattern, which stays `ms`.
    ms_merged, _, _ = BT.run(types.SimpleNamespace(precision=precision, iters=8, pixels=1024, events=4096, P=10, merge_events=True))
    torch.cuda.empty_cache()
    par = TP.c2f_gradient_parity(tuple(modes))
    torch.cuda.empty_cache()
    # the same iteration with the shipped configs' adaptive weight proposal on the blur batch (kernel_use_awp): the per-sample part fused
    # on the fine level's geo fragments (awp.FusedAWP: embedding MLP, scan, the MotionAggregationModule's per-sample part) vs the module's
    # plain PyTorch forward on depth_feature [R P, S, 128]; the module has the reference's structure (tools/awp_standin.py, mam="corr")
    ms_awp_f, _, _ = BT.run(types.SimpleNamespace(precision=precision, iters=5, pixels=1024, events=4096, P=10, awp="fused", mam="corr"))
    torch.cuda.empty_cache()
    ms_awp_t = None
    if precision in ("f16", "bf16", "f16x3"):        # (the torch module takes float32 feature rows, which the mixed modes do not write)
        ms_awp_t, _, _ = BT.run(types.SimpleNamespace(precision=precision, iters=5, pixels=1024, events=4096, P=10, awp="torch", mam="corr"))
        torch.cuda.empty_cache()
    aabb = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
    fv = 134217984
    g = W.pdrf_grid_size(aabb[0], aabb[1], fv)
    net = VoxelNeRFSampleFeatures(W.make_pdrf_state_dict(32, g, input_ch=127, hidden_dim=256, geo_feat_dim=128), "", aabb, num_layers=2, hidden_dim=256,
                                  geo_feat_dim=128, num_layers_color=3, input_ch=127, app_dim=32, app_n_comp=(64, 16, 16), n_voxels=fv)
    rs = np.random.RandomState(0)
    R, S = 4096, 128
    o = rs.uniform(-0.3, 0.3, (R, 1, 3)) + np.array([0, 0, 0.9])
    d = rs.normal(size=(R, 1, 3)) * np.array([0.05, 0.05, 0.0]) + np.array([0, 0, -1.0])       # slope 0.05: the PMC pass's rays (tools/pmc_scatter.sh)
    z = np.sort(rs.uniform(0.1, 1.7, (R, S, 1)), 1)
    pts = torch.as_tensor((o + d * z).astype(np.float32), device="cuda").reshape(-1, 3).contiguous()
    n = pts.shape[0]
    d_out = torch.randn((n, 32), device="cuda")
    grads, gs = _grid_grads(net, net.grid_params())
    k_ms = kernel_ms(lambda: L.check(L.lib().evd_voxel_sample_bwd(net._h, L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), None, L.stream_ptr()), "bwd"), 10)
    nb = int(L.lib().evd_voxel_sample_bwd_workspace_bytes(net._h, n))
    ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    h32_ms = kernel_ms(lambda: L.check(L.lib().evd_voxel_sample_bwd_ws(net._h, L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), None, L.ptr(ws), nb,
                                                                        L.stream_ptr()), "bwd_ws"), 10)
    # what the iteration calls: the scatter in the forward's mode (the fine level of f16c / f16 re-gathers the float16 grid copies, round 6)
    h_ms = kernel_ms(lambda: L.check(L.lib().evd_voxel_sample_bwd_prec(net._h, L.PREC[precision], L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), None, L.ptr(ws), nb,
                                                                       L.stream_ptr()), "bwd_prec"), 10)
    # atomic requests of the shipped form per 2^19 samples at this slope: profiles/r06_pmc_scatter.txt (TCC_ATOMIC_sum: main kernel + line slices)
    req = {"k_voxel_sample_bwd_w": 5416452, "k_scatter_lines": 452665}
    peak = 20.0                                                 # G atomic requests/s: a bare kernel of coalesced float atomics (tools/probes/atomic_probe.hip: 320 G adds/s in 64-byte requests)
    achieved = sum(req.values()) / (h_ms * 1e-3) / 1e9
    return {"workload": "blurfactory training iteration: 1024 pixels x 10 sub-exposure rays (stand-in RBK weights: a small learnable rigid kernel, "
                        "tools/bench_train_step.py RigidKernel, in place of RigidBlurringModel.forward's MLPs) + 2 x 4096 event rays, 64 + 64 samples, "
                        "losses, TV, backward, Adam, parameter re-pack", "precision": precision, "ms_per_iteration": ms, "rays_per_iteration": nrays,
            "rays_per_s": nrays / (ms * 1e-3),
            "ms_per_iteration_by_mode": modes,
            "ms_per_iteration_event_rays_in_one_call": ms_merged,
            "parity_holding_mode": {"mode": "f16m", "ms_per_iteration": modes.get("f16m"),
                                    "what": "the fastest mode whose GRADIENTS stay within 2e-3 of the norm of the reference's autograd (goldens G19, G30 at 16 384 "
                                            "samples; tests/test_gpu_train_f16c.py); 'precision' above (f16c) holds the rendered colours to 1e-4 (north_star's bound) "
                                            "and its gradients to 1e-2 -- the ReLU flip floor of DESIGN 3.6 -- and is NOT the gradient-parity mode.  Round 6: whether that "
                                            "floor matters for training was measured (converges_like_f32 below): it does not -- all three modes train like the "
                                            "float32-grade one"},
            "parity_by_mode": {"what": "tools/train_parity.py: 2048 rays x (16 + 16) samples, the G19 loss; rendered colours and every gradient tensor "
                                       "(30 parameters + rays) against the float32-grade mode f16x3 (= the reference's autograd to 2e-5 on golden G19)",
                               **par,
                               "note": "f16m = float32-grade forward (float32's own ReLU patterns) + float16 backward; f16c = compensated forward: ~5e-6 of the "
                                       "units flip against float32, which bounds the gradients at ~sqrt(5e-6) of their norm whatever the batch size; f16 = "
                                       "single-product float16 forward (1e-3 of the units flip)"},
            "backward_forms": "round 5: the 64-wide level's whole backward in one launch (k_voxel_bwd_fused64), sigma_net.1 as one fused launch (k_wgrad_dgrad RT_ = 5), "
                              "d c1 formed inside color_net.1's launch (YGEN), d fts as rows from the dgrad kernels, the AWP embedding's backward in one launch "
                              "(k_awp_bwd_fused); each has a switch back to the per-layer chain (INTEGRATION.md section 5) and equals it to 5e-7 "
                              "(tests/test_gpu_bwd_fusion.py)",
            "with_awp_ms_per_iteration": {"fused_on_geo_fragments": ms_awp_f, "torch_module_on_depth_feature": ms_awp_t,
                                          "note": "AWP module = tools/awp_standin.py (the reference module's surface; its per-sample embedding is the reference's)"},
            "scatter_hybrid_ms": h_ms, "scatter_hybrid_float32_grids_ms": h32_ms, "scatter_all_atomics_ms": k_ms,
            "scatter_note": "what the iteration runs: k_voxel_sample_bwd_w persistent, a wavefront owns 16 consecutive samples of a ray from the point load "
                            "to its last atomic; plane taps summed in a register along runs of samples on one cell, the basis_mat gradient accumulated in "
                            "registers by MFMA, line taps through the 64-bit fixed-point LDS slices of k_scatter_lines.  Round 6: the plane-tap walk rebuilt "
                            "(a pass's LDS operands fetched first, the 64-channel plane's four taps in one pass with scalar run ends, legacy multiply): "
                            "ablation builds had shown the walk ALONE at 33.6 k of a tile's 51.8 k cycles without any atomic "
                            "(profiles/r06_scatter_stamps_before_walk_rewrite.log); 0.543 -> 0.43 ms per 2^19 samples here, the iteration's nine scatters "
                            "5.2 -> 2.8 ms (profiles/r06_train_kernels_after_walk.txt).  Then: d coef on split-float16 MFMAs, the backward's re-gather of the grid "
                            "values on the float16 copies where the forward gathered them (evd_voxel_sample_bwd_prec; scatter_hybrid_float32_grids_ms is the "
                            "float32 re-gather), k_scatter_lines' workgroups ordered so that a chunk's jobs share one XCD's L2: the nine main launches "
                            "2.8 -> 2.45 ms, the line launches 0.77 -> 0.68 ms (profiles/r06_scatter_half_ab.log, r06_scatter_lines_order_ab.log)",
            "fastest_mode_that_trains_like_f32": {"mode": "f16", "ms_per_iteration": modes.get("f16"), "rays_per_s": nrays / (modes["f16"] * 1e-3) if modes.get("f16") else None,
                                                  "why": "converges_like_f32 below: on a fixed budget the plain float16 mode reaches the float32-grade mode's loss curve and "
                                                         "PSNR; `precision` (f16c) stays the line's primary mode because its RENDER holds north_star's 1e-4 bound"},
            "converges_like_f32": {"modes": ["f16", "f16c", "f16m"], "reference_mode": "f16x3 (float32-grade)",
                                   "what": "tools/train_synthetic.py --iters 3000 --precision f16,f16c,f16m,f16x3 (shipped network shape, same initial student, "
                                           "rays, draws and teacher): loss curves equal to 4 digits, final held-out PSNR 42.99 / 42.95 / 42.93 / 42.93 dB",
                                   "source": "profiles/r06_convergence_by_mode.txt"},
            "training_call_parity": {"what": "goldens G32 (the reference's NeRFAll.forward in training mode with its real RigidBlurringModel + AdaptiveWeightProposal) "
                                             "and G33 (five iterations of its optimisation loop): every mode's loss follows the reference's to <= 7e-6 over the "
                                             "five steps; f16x3 outputs 7e-7, level gradients 1.5e-4 of the norm in the median",
                                     "source": "tests/test_gpu_train_call.py, profiles/r06_train_call_parity.log"},
            "roofline": {"kernel": "k_voxel_sample_bwd_w<., ., true> + k_scatter_lines: the shipped hybrid (fine level 586 x 586 x 390, 4096 x 128 samples, slope 0.05)",
                         "bound": "memory-side atomics (requests/s)", "kernel_ms": h_ms, "atomic_requests": req, "achieved": achieved, "peak": peak,
                         "unit": "G atomic requests/s", "frac": achieved / peak,
                         "bytes_written": 562e6, "write_GBps": 562e6 / (h_ms * 1e-3) / 1e9,
                         "pmc_source": "profiles/r06_pmc_scatter.txt (TCC_ATOMIC_sum, WRITE_SIZE; re-collected on this round's kernels)",
                         "all_atomics_form": {"kernel_ms": k_ms, "atomic_requests": 18937705, "achieved": 18937705 / (k_ms * 1e-3) / 1e9, "frac": 18937705 / (k_ms * 1e-3) / 1e9 / peak},
                         "note": "achieved = PMC-counted atomic requests of the two launches / their duration measured here; peak = the request rate of a bare "
                                 "atomic kernel on this chip (20 G/s whatever the table size, scope, XCD locality or run length: a request costs per 64-byte "
                                 "segment, profiles/r06_atomic_run_probe.log).  The all-atomics form (every tap a float atomic, 18.9 M requests) "
                                 "runs at 84 % of that rate and takes 2.4x as long: the shipped form removes requests instead of chasing the rate"}}


def strong_leg(model_c2f, precision, world, rank, frames=3):
    """BASELINE config 5: full 400x400 frames (160 000 rays each, 64 + 128 samples, render_kwargs_test) through
    render_path(shard_rows=True): the image rows are split over the ranks and all-gathered (strong scaling: fixed total work)."""
    import numpy as np
    import torch
    from evdeblurnerf_amd import weights as W
    K = W.synthetic_camera()
    poses = [torch.as_tensor(W.synthetic_pose(40 + i)[:3, :4].astype(np.float32), device="cuda") for i in range(frames)]
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=128, perturb=0., raw_noise_std=0.)
    model_c2f.render_path(400, 400, K, 1 << 22, poses[:1], kw, shard_rows=world > 1)
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    rgbs, _ = model_c2f.render_path(400, 400, K, 1 << 22, poses, kw, shard_rows=world > 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"workload": "full-frame 400x400 render (160 000 rays, 64 + 128 samples), blurfactory c2f, rows sharded over ranks + all_gather",
            "scaling": "strong", "n_gpus": world, "frames": frames, "ms_per_frame": 1e3 * dt / frames, "rays_per_s": frames * 160000 / dt,
            "precision": precision, "frame_mean": float(rgbs.mean())}


# ---------------------------------------------------------------------------------------------------------------------

def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if needs_launch(a):
        sys.exit(launch(a, argv))
    import ctypes as C
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    rank, world, local, dist, backend, _ = init_ranks()

    from evdeblurnerf_amd import _lib as L, weights as W
    from evdeblurnerf_amd.losses import blur_loss_partials
    from evdeblurnerf_amd.tonemapping import CRF

    R, S = a.rays, a.samples
    K = W.synthetic_camera()
    rays = torch.as_tensor(W.synthetic_rays(100 + rank, R), device="cuda")
    target = torch.rand((R, 3), device="cuda")
    ones = torch.ones((R, 1), device="cuda")
    crf = CRF("gamma")
    model, sd = make_model(a.precision)
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=S, N_importance=0, retraw=False)

    pending = []

    def step():
        rgb, depth, acc, _ = model.render(400, 400, K, rays=rays, **kw)
        if dist:   # the path's one exchange: packed loss partials (evdeblurnerf_amd/dist.py)
            p, _ = blur_loss_partials(crf, rgb, ones, target)
            # the reduced loss is a logging value, nothing in the next step consumes it: the all-reduce (RCCL's own stream)
            # overlaps the next step's render and is waited for one step later (and after the last step, inside the timed region)
            if pending:
                pending.pop().wait()
            pending.append(dist.all_reduce(p, async_op=True))
        return rgb

    dt = time_steps(step, a.steps, a.warmup, dist, drain=lambda: [w.wait() for w in pending], settle_steps=a.settle)
    med = per_step_ms(lambda: model.render(400, 400, K, rays=rays, **kw), max(20, min(a.steps, 200)))
    result = headline(R, S, world, a.steps, a.warmup, dt, a.precision, a.settle)
    result["ms_per_step_median"] = med[len(med) // 2]
    result["ranks"] = {"world_size": world, "backend": backend,
                       "launcher": ("bench.py self-launch" if os.environ.get("EVD_BENCH_SELF_LAUNCH") else
                                    "external torchrun" if world > 1 else "none"),
                       "devices": rank_devices(dist, world)}
    lean = world > 1          # N > 1 runs: headline kernel + the strong-scaling leg (the other legs are N = 1 measurements)

    if rank == 0:
        # ---- roofline of the dominant kernel (fused PE+MLP), measured live with HIP events
        rb = torch.empty((R, 11), device="cuda")
        z = torch.empty((R, S), device="cuda")
        cfg = model._cfg(400, 400, float(K[0][0]), True, 0., 1., S, 0, False, 0., False)
        L.check(L.lib().evd_ray_batch(C.byref(cfg), L.ptr(rays), R, L.ptr(rb), L.stream_ptr()))
        L.check(L.lib().evd_sample_z(C.byref(cfg), L.ptr(rb), 11, R, None, L.ptr(z), L.stream_ptr()))
        modes = {}
        for prec in ([a.precision] if (a.no_modes or lean) else ["f16c", "f16", "bf16", "f16x3", "f32"]):
            net = model.mlp_coarse
            ksteps = max(3, a.steps // (10 if prec == "f32" else 1))
            ms = kernel_ms(lambda: net.mlpforward(rb, z, precision=prec), ksteps)
            tf = R * S * FLOP_PER_SAMPLE / (ms * 1e-3) / 1e12
            modes[prec] = {"kernel": "k_nerf_mlp", "ms": ms, "achieved": tf, "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                           "frac": tf / PEAK_TFLOPS[prec], "rays_per_s_kernel": R / (ms * 1e-3)}
            if prec == "f16x3":
                modes[prec]["mfma_issue_frac"] = 3 * tf / PEAK_TFLOPS[prec]      # three MFMA products per algorithmic one
            if prec == "f16c":
                modes[prec]["kernel"] = "k_nerf_mlp_c"
                modes[prec]["mfma_issue_frac"] = 1.5 * tf / PEAK_TFLOPS[prec]    # + two fp6 32x32x64 products per four f16 32x32x16 ones: 1.5x the MFMA cycles
        m = modes[a.precision]
        # HBM traffic and the hardware's own MFMA-busy fraction come from the committed PMC passes of this kernel
        # (rocprofv3 cannot run inside the timed process): profiles/r05_pmc_mlp.json, made by tools/pmc_mlp.sh + tools/pmc_mlp_json.py (newest round first)
        traffic, busy = None, None
        try:
            pmc_file = next(f for f in ("r05_pmc_mlp.json", "r03_pmc_mlp.json", "r02_pmc_mlp.json", "r01_v3_pmc_mlp.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["derived"].get(a.precision)
            if pmc and R == 4096 and S == 128:
                traffic, busy = pmc["traffic_bytes"], pmc["mfma_busy_frac"]
        except (OSError, KeyError, ValueError):
            pass
        # what the matrix pipe sustains on this chip today (power-limited clock): bare MFMA loop, constant vs random operands
        sustained = {}
        for name, rnd in (("constant_operands", 0), ("random_operands", 1)):
            tfv = C.c_double()
            L.check(L.lib().evd_probe_mfma_rate(rnd, 2000, C.byref(tfv), L.stream_ptr()))
            sustained[name] = tfv.value
        # the same binary on all-zero weights and rays: identical instruction stream, no switching activity -- what the kernel's STRUCTURE
        # costs at full clock; the difference to kernel_ms is the chip clocking down to its power budget on real data
        zero_ms = None
        if not lean and not a.no_zero_probe:
            from evdeblurnerf_amd.nerf import NeRF
            znet = NeRF({k: np.zeros_like(v) for k, v in W.make_nerf_state_dict(21).items()})
            rb0, z0 = torch.zeros_like(rb), torch.zeros_like(z)
            rb0[:, 7] = 1.0
            zero_ms = kernel_ms(lambda: znet.mlpforward(rb0, z0, precision=a.precision), max(3, a.steps))
            del znet
        result["roofline"] = {"bound": "mfma", "achieved": m["achieved"], "peak": m["peak"], "unit": "TFLOP/s",
                              "frac": m["frac"], "traffic": traffic, "kernel": m["kernel"], "kernel_ms": m["ms"],
                              "algorithmic_flop": R * S * FLOP_PER_SAMPLE, "algorithmic_hbm_bytes": R * S * 20 + R * 44,
                              "mfma_busy_frac_pmc": busy,
                              "sustained_mfma_tflops": sustained, "frac_of_sustained_random": m["achieved"] / sustained["random_operands"],
                              "kernel_ms_on_all_zero_data": zero_ms,
                              "note": "achieved = algorithmic GEMM flops (1 186 816/sample) / HIP-event launch duration, against the 2.4 GHz "
                                      "dense peak; traffic (bytes per launch: 2.8x the algorithmic bytes -- 30 MB against 10.7 MB: the 2.1 MB weight stream "
                                      "is fetched by every workgroup and served once per XCD L2) and mfma_busy_frac_pmc (SQ_VALU_MFMA_BUSY_CYCLES over "
                                      "GRBM_GUI_ACTIVE x SIMDs: the kernel keeps the pipe busier than frac says because the chip clocks "
                                      "below 2.4 GHz under this load) are from the PMC passes in profiles/r05_pmc_mlp.json (re-collected on the round-5 build); "
                                      "sustained_mfma_tflops = a bare back-to-back MFMA loop on every SIMD, measured in this run "
                                      "(evd_probe_mfma_rate): the random-operand figure is the practical ceiling for real data; "
                                      "kernel_ms_on_all_zero_data = the same launch on all-zero weights and rays (same instruction stream, no "
                                      "switching power): kernel_ms above it is the power-limited clock, not the kernel's structure"}
        result["modes"] = modes
        if not a.no_parity and not lean:
            # ---- parity of EVERY arithmetic mode at the full metric size against the CPU oracle (not against another kernel of
            # this library), on the seed-derived weights and on weights trained in this run
            precs = ["f32", "f16x3", "f16c", "f16", "bf16"]
            par = {"bound": 1e-4, "reference": "oracle/evd_oracle.c render (pinned to the imported reference by the goldens tests/golden/G1..G35), same rays and weights",
                   "rays": R, "samples": S, "rgb_linf_vs_oracle": {"seed_weights": oracle_parity(sd, rays, S, K, precs)}}
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import trained_weights as TW
            t0 = time.perf_counter()
            sd_tr, rep = TW.train_nerf(iters=a.train_iters)
            rep["train_seconds"] = time.perf_counter() - t0
            par["rgb_linf_vs_oracle"]["trained_weights"] = oracle_parity(sd_tr, rays, S, K, precs)
            par["trained_weights"] = rep
            fx = os.path.join(ROOT, "tests", "golden", "trained", "nerf_8x256_10k.npz")
            if os.path.exists(fx):          # the committed 10 000-iteration network (tools/trained_weights.py --iters 10000 --save, round 3)
                par["rgb_linf_vs_oracle"]["trained_10k_fixture"] = oracle_parity(dict(np.load(fx)), rays, S, K, precs)
            par["headline_mode"] = a.precision
            par["headline_within_bound"] = {k: v[a.precision] <= 1e-4 for k, v in par["rgb_linf_vs_oracle"].items()}
            result["parity"] = par
            del sd_tr
        if not a.no_composite and not lean:
            # ---- compositing scan alone (HBM-bound): 2^20 rays x 128 samples = 3.25 GB of algorithmic traffic
            Rc = 1 << 20
            raw_c = torch.randn((Rc, S, 4), device="cuda")
            z_c = torch.linspace(0, 1, S, device="cuda").expand(Rc, S).contiguous()
            rd_c = torch.randn((Rc, 3), device="cuda")
            o3, o1, o2, ow = (torch.empty((Rc, 3), device="cuda"), torch.empty(Rc, device="cuda"), torch.empty(Rc, device="cuda"),
                              torch.empty((Rc, S), device="cuda"))

            def comp():
                L.check(L.lib().evd_raw2outputs(L.ptr(raw_c), L.ptr(z_c), L.ptr(rd_c), 3, Rc, S, 4, 3, 0, 3, L.ACT["sigmoid"], L.ACT["relu"],
                                                0, 0.0, None, L.ptr(o3), None, L.ptr(o2), L.ptr(ow), L.ptr(o1), None, 0, None, L.stream_ptr()))
            cms = kernel_ms(comp, 10)
            cbytes = Rc * (S * 24 + 32)
            # the dispatch depends on the shape and on three developer switches (csrc/kernels_render.hip: form, non-temporal streams, rays per
            # wavefront -- 4 from 2^17 rays up): the committed PMC pass and the traffic-mix probe were taken on THIS shape with the default switches,
            # and are quoted (labelled as quoted) only when the run matches it
            comp_env = {k: os.environ[k] for k in ("EVD_COMPOSITE_FORM", "EVD_COMPOSITE_NT", "EVD_COMPOSITE_RPW") if k in os.environ}
            as_profiled = Rc == 1 << 20 and S == 128 and not comp_env
            result["composite"] = {"kernel": ("k_composite_il<2, 4, 3, sigmoid, relu, nt> (lanes interleaved over the row, non-temporal streams; the dispatch for this "
                                              "shape with default switches)" if as_profiled else "as dispatched under " + json.dumps(comp_env)),
                                   "rays": Rc, "samples": S, "ms": cms, "bound": "hbm",
                                   "achieved": cbytes / (cms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": cbytes / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": cbytes}
            if as_profiled:
                result["composite"]["traffic_quoted_from_committed_profile"] = {
                    "read_bytes": 2 * 1323072.1e3 * 1.024, "write_bytes": 544893.2e3 * 1.024, "over_algorithmic": 1.004,
                    "source": "profiles/r05_pmc_composite.txt (FETCH_SIZE x 2 per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE; KB of 1024 B); "
                              "not measured in this run -- the kernel is unchanged since that pass"}
                result["composite"]["traffic_mix_ceiling_quoted"] = {
                    "GBps": 6090, "frac_of_peak": 0.761, "what": "a bare kernel moving the same bytes in the same pattern (20 B read + 4 B written per sample, "
                    "non-temporal, no arithmetic): tools/probes/hbm_mix_probe.hip, profiles/r05_composite_ab.log; read-only it reaches 7.05 TB/s, a float4 copy 6.22 TB/s"}
            del raw_c, z_c, rd_c, o3, o1, o2, ow
        if not a.no_train and not lean:
            # ---- the training kernels on the same workload (next row): forward that keeps the activations and the
            # hand-written backward of the fused MLP (parameter gradients + the gradient reaching the rays); informational
            from evdeblurnerf_amd.nerf import NeRF
            tnet = NeRF(sd, "mlp_coarse.", precision=a.precision if a.precision in ("f16", "bf16") else "f16")
            rb_t, z_t = rb, z
            algo_flop = R * S * FLOP_PER_SAMPLE
            d_raw_t = torch.randn((R, S, 4), device="cuda") * 1e-4
            pts_t = (rb_t[:, None, 0:3] + rb_t[:, None, 3:6] * z_t[..., None]).contiguous()
            fwd_ms = kernel_ms(lambda: tnet.mlpforward_train(rb_t, z_t), 10)
            _, store_t = tnet.mlpforward_train(rb_t, z_t)
            bwd_ms = kernel_ms(lambda: tnet.mlp_backward_flat(d_raw_t, store_t), 10)
            bwd_rays_ms = kernel_ms(lambda: tnet.mlp_backward_flat(d_raw_t, store_t, pts=pts_t, ray_batch=rb_t), 10)
            result["train"] = {"forward_keeping_activations_ms": fwd_ms, "backward_params_ms": bwd_ms, "backward_params_and_rays_ms": bwd_rays_ms,
                               "activation_store_bytes": int(store_t.numel()), "algorithmic_flop_backward": 2 * algo_flop,
                               "backward_tflops": 2 * algo_flop / (bwd_ms * 1e-3) / 1e12,
                               "note": "evd_nerf_mlp_train / evd_nerf_mlp_backward on the metric workload (one network); HBM-bound by "
                                       "construction (per-layer dgrad + wgrad over the stored fragments), DESIGN.md 7"}
            del store_t
            # the mixed training modes (round 4): compensated (f16c) / split-float16 (f16m) forward, the float16 mode's store and backward
            mixed = {}
            for mp in ("f16c", "f16m"):
                fm = kernel_ms(lambda: tnet.mlpforward_train(rb_t, z_t, precision=mp), 10)
                _, st_m = tnet.mlpforward_train(rb_t, z_t, precision=mp)
                bm = kernel_ms(lambda: tnet.mlp_backward_flat(d_raw_t, st_m, precision=mp), 10)
                mixed[mp] = {"forward_keeping_activations_ms": fm, "backward_params_ms": bm}
                del st_m
            result["train"]["mixed_modes"] = dict(mixed, note="forward in the mode's arithmetic (bit-identical to its inference kernel), float16 store and "
                                                              "float16 dgrad / wgrad: G18 gradient summaries within 1.4e-3 (f16c) / 1.0e-3 (f16m) of the norm, "
                                                              "rendered colours within 1e-6 (tests/test_gpu_train_f16c.py); single-product f16: 15 % / 3e-3")
            # the float32-grade training mode (EVD_PREC_F16X3: (hi, lo) fragments, 3-MFMA products; the reference trains in float32)
            f32g = NeRF(sd, "mlp_coarse.", precision="f16x3")
            fwd3_ms = kernel_ms(lambda: f32g.mlpforward_train(rb_t, z_t), 5)
            _, store3 = f32g.mlpforward_train(rb_t, z_t)
            bwd3_ms = kernel_ms(lambda: f32g.mlp_backward_flat(d_raw_t, store3), 5)
            result["train"]["float32_grade"] = {"precision": "f16x3", "forward_keeping_activations_ms": fwd3_ms, "backward_params_ms": bwd3_ms,
                                                "activation_store_bytes": int(store3.numel()),
                                                "note": "gradients equal float64 autograd to 1e-7..1e-4 and the reference's own autograd goldens "
                                                        "(G18 / G19) to 2e-4 of the norm (tests/test_gpu_train_f32grade.py); the half-precision "
                                                        "figures above are the throughput modes"}
            del store3, f32g, tnet

    # ---- the shipped configuration (every rank builds the model: the strong-scaling leg shards one frame's rows over the ranks)
    c2f_model = None
    if not a.no_c2f or not a.no_strong:
        c2f_prec = a.precision                                          # f16c: fine level compensated (float16 grid copies), coarse level float32-grade on float32 grids
        # training: f16c = the compensated forward (the headline's arithmetic) in front of the float16 backward; f16m / f16 / f16x3 beside it
        train_prec = a.precision if a.precision in ("f16", "bf16", "f16x3", "f16c") else "f16"
        # the informational legs below must never cost the contract line: a failure is recorded in its place
        def guarded(name, fn):
            try:
                return fn()
            except Exception as e:      # noqa: BLE001
                import traceback
                sys.stderr.write(f"[bench] leg '{name}' failed:\n{traceback.format_exc()}\n")
                torch.cuda.empty_cache()
                return {"error": f"{type(e).__name__}: {e}"}
        if rank == 0 and not a.no_c2f and not lean:
            r = guarded("c2f", lambda: c2f_leg(c2f_prec, max(5, a.steps // 2), parity=not a.no_parity))
            result["c2f"], c2f_model = r if isinstance(r, tuple) else (r, None)
        if rank == 0 and not a.no_awp and not lean:
            result["awp"] = guarded("awp", lambda: awp_leg(train_prec))
        if rank == 0 and not a.no_train and not lean:
            result["train_iteration"] = guarded("train_iteration", lambda: train_iteration_leg(train_prec))
        if world > 1 and not a.no_train:
            # ---- data-parallel TRAINING: every rank runs the whole blurfactory iteration on its own batch (weak scaling),
            # loss partials all-reduced through autograd, gradients all-reduced on their persistent flat buffers before Adam
            def train_dp():
                import types
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_train_step as BT
                ns = types.SimpleNamespace(precision=train_prec, iters=10, pixels=1024, events=4096, P=10, dist=True)
                ms, nrays, _ = BT.run(ns)
                t = torch.tensor([ms, ns.allreduce_ms], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms, ar = float(t[0]), float(t[1])
                return {"workload": "blurfactory training iteration per GPU (1024 pixels x 10 sub-exposure rays + 2 x 4096 event rays, 64 + 64 samples, "
                                    "losses, TV, backward, gradient all-reduce, Adam, re-pack)", "scaling": "weak", "n_gpus": world, "precision": train_prec,
                        "ms_per_iteration": ms, "rays_per_s": world * nrays / (ms * 1e-3), "rays_per_iteration_per_gpu": nrays,
                        "gradient_allreduce_exposed_ms": ar, "gradient_allreduce_exposed_share": ar / ms, "gradient_bytes": ns.grad_bytes,
                        "gradient_messages_started_inside_backward": getattr(ns, "allreduce_early_starts", 0),
                        "note": "max over ranks; the gradient exchange runs on the persistent flat gradient buffers of the in-place mode "
                                "(4 messages: 2 levels x {networks, grids}) + one bucket for the blur kernel / CRF parameters; a level's "
                                "messages start as soon as that level's last backward node has run (dist.GradReducer.attach), "
                                "gradient_allreduce_exposed_ms = what is still waited for after loss.backward() returned"}
            r = guarded("train_scaling", train_dp)
            if rank == 0:
                result["train_scaling"] = r
            torch.cuda.empty_cache()
        if not a.no_strong:
            if c2f_model is None:
                from evdeblurnerf_amd.renderer import NeRFAll
                c2f_model = NeRFAll(W.blurfactory_args(128), W.make_blurfactory_state_dict(31), precision=c2f_prec).eval()
            strong = strong_leg(c2f_model, c2f_prec, world, rank)
            if rank == 0:
                result["strong_scaling"] = strong
        del c2f_model

    if rank == 0:
        if not a.no_cpu_baseline and not lean:
            from oracle import oracle as O
            fast = True
            try:
                O.use_fast_build()                   # the same C file built for throughput on THIS host (-O3 -march=native, FMA allowed)
            except Exception:                         # no compiler on the box: the checker build that travelled with the repo
                fast = False
            onet = O.Nerf(sd, "mlp_coarse.")
            ocfg = O.make_cfg(N_samples=S)
            sample = W.synthetic_rays(100, R)
            O.render_nerf(onet, None, ocfg, sample[:64])
            n_cpu, t_cpu = 0, 0.0
            t0 = time.perf_counter()
            while t_cpu < 10.0:                      # ~10 s of wall time on all host cores: the whole workload per call (32 rays per core on a
                O.render_nerf(onet, None, ocfg, sample)      # 128-core host; 512-ray calls left the cores mostly waiting at the OpenMP barriers)
                n_cpu += R
                t_cpu = time.perf_counter() - t0
            result["cpu_baseline"] = {"value": n_cpu / t_cpu, "unit": "rays/s", "cores": O.num_threads(), "kind": "port",
                                      "true_reference": {"rays_per_s": [379, 720], "threads": 8, "where": "the reference's NeRFAll.render (PyTorch, CPU) on this workload, "
                                                         "timed in the build container (tools/time_reference_cpu.py, BASELINE.md section 2); it cannot travel to the GPU "
                                                         "box, so it is quoted, not re-timed here", "gpu_over_reference": [result["value"] / 720, result["value"] / 379]},
                                      "sample": f"{n_cpu} rays (the workload's {R} rays x {S} samples, cycled for {t_cpu:.1f} s) through the C oracle, OpenMP over rays on all host cores",
                                      "build": "-O3 -march=native -ffp-contract=fast (throughput build of oracle/evd_oracle.c, compiled on this host)" if fast
                                               else "-O2 -march=x86-64-v3 -ffp-contract=off (the parity-checker build)"}
        print(json.dumps(result), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
