"""Data-parallel training step with REAL kernel outputs (VERDICT r1 / ADVICE: the gloo tests exchange numpy stand-ins).
Two processes share GPU 0 (gloo collectives; RCCL needs one GPU per rank): each renders its shard of the blur pixels with the
training kernels, reduces it to the packed partials of the fused loss node, the partials are all-reduced THROUGH AUTOGRAD
(dist.all_reduce_partials), the loss is back-propagated through the shard and dist.GradReducer sums the parameter gradients.
Loss and gradients must equal the single-process step on the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

R_PIX, P = 96, 4


def _step_c2f(rank, world):
    """the shipped backbone (mode='c2f') in the in-place gradient mode with the per-level early all-reduce (GradReducer.attach)"""
    from evdeblurnerf_amd import dist as D, weights as W
    from evdeblurnerf_amd.losses import blur_loss_from_partials, blur_loss_partials_autograd
    from evdeblurnerf_amd.renderer import NeRFAll
    from evdeblurnerf_amd.tonemapping import CRF
    from test_gpu_train import _c2f_model
    dev = "cuda"
    model, sd = _c2f_model("f16", 16)
    model.enable_training(sd, grads_in_place=True).train()
    red = D.GradReducer(list(model.parameters()), flat_buffers=model.grad_buffers()).attach(model)
    rs = np.random.RandomState(5)
    rays = torch.as_tensor(W.synthetic_rays(8, R_PIX * P), device=dev)
    w1 = torch.softmax(torch.as_tensor(rs.standard_normal((R_PIX, P)).astype(np.float32), device=dev), -1)
    target = torch.as_tensor(rs.uniform(0, 1, (R_PIX, 3)).astype(np.float32), device=dev)
    (plo, phi), (rlo, rhi) = D.shard_pixels(R_PIX, P, rank, world)
    n = phi - plo
    out = None
    for it in range(2):             # two iterations: the callbacks re-arm
        model.zero_grad(set_to_none=True)
        rgb, rgb0, other, _ = model(400, 400, W.synthetic_camera(), 1 << 20, rays=rays[rlo:rhi], ndc=True, near=0., far=1., N_samples=16,
                                    N_importance=16, perturb=0., raw_noise_std=0.)
        part = blur_loss_partials_autograd(CRF("gamma"), rgb.reshape(n, P, 3), w1[plo:phi], target[plo:phi], rgb0_p=rgb0.reshape(n, P, 3))
        (part,) = D.all_reduce_partials(part)
        loss, _ = blur_loss_from_partials(part)
        (loss + 0.01 * other["TV"].sum() / world).backward()
        red.start()
        red.wait()
        out = (float(loss.detach()), {k: v.grad.detach().cpu().numpy().copy() for k, v in model.named_parameters()})
    if world > 1:
        assert red.early_starts >= 4, red.early_starts          # both levels' buffers went out from inside the backward, in both iterations
    return out


def _step(rank, world):
    from types import SimpleNamespace
    from evdeblurnerf_amd import dist as D, weights as W
    from evdeblurnerf_amd.losses import blur_loss_from_partials, blur_loss_partials_autograd
    from evdeblurnerf_amd.renderer import NeRFAll
    from evdeblurnerf_amd.tonemapping import CRF
    dev = "cuda"
    sd = dict(W.prefixed(W.make_nerf_state_dict(11), "mlp_coarse"))
    sd.update(W.prefixed(W.make_nerf_state_dict(12), "mlp_fine"))
    args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=16)
    model = NeRFAll(args, sd, precision="f16").enable_training(sd).train()
    rs = np.random.RandomState(5)
    rays = torch.as_tensor(W.synthetic_rays(8, R_PIX * P), device=dev)
    w1 = torch.softmax(torch.as_tensor(rs.standard_normal((R_PIX, P)).astype(np.float32), device=dev), -1)
    target = torch.as_tensor(rs.uniform(0, 1, (R_PIX, 3)).astype(np.float32), device=dev)
    (plo, phi), (rlo, rhi) = D.shard_pixels(R_PIX, P, rank, world)
    n = phi - plo
    rgb, rgb0, _, _ = model(400, 400, W.synthetic_camera(), 1 << 20, rays=rays[rlo:rhi], ndc=True, near=0., far=1., N_samples=16,
                            N_importance=16, perturb=0., raw_noise_std=0.)
    part = blur_loss_partials_autograd(CRF("gamma"), rgb.reshape(n, P, 3), w1[plo:phi], target[plo:phi], rgb0_p=rgb0.reshape(n, P, 3))
    (part,) = D.all_reduce_partials(part)
    loss, _ = blur_loss_from_partials(part)
    loss.backward()
    params = model.parameters()
    red = D.GradReducer(params)
    red.start()
    red.wait()
    return float(loss.detach()), {k: v.grad.detach().cpu().numpy() for k, v in model.named_parameters()}


def _worker(rank, world, port, q, backend="gloo", mode="nerf", side_spin_us=0, skip_join=False):
    if side_spin_us:
        os.environ["EVD_TEST_SIDE_SPIN_US"] = str(side_spin_us)          # read once, when the library first forks to a side stream
        # the PDRF levels' backward forks to the side stream only with this switch (off by default: no gain for the level networks) --
        # without it the hook never fires in mode='c2f' and the test would prove nothing
        os.environ["EVD_BWD_OVERLAP_VOXEL"] = "1"
    if skip_join:
        os.environ["EVD_TEST_SKIP_SIDE_JOIN"] = "1"                       # negative control: the entry returns without joining its side stream
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dev = rank if backend == "nccl" else 0            # RCCL: one GPU per rank; gloo: the ranks share GPU 0
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(dev), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    loss, grads = (_step_c2f if mode == "c2f" else _step)(rank, world)
    from evdeblurnerf_amd import _lib as L
    q.put((rank, loss, grads, int(L.lib().evd_debug_side_spin_count())))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(backend, mode, side_spin_us=0, skip_join=False):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend, mode, side_spin_us, skip_join)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    loss1, grads1 = (_step_c2f if mode == "c2f" else _step)(0, 1)                      # the whole batch in this process
    got.sort(key=lambda t: t[0])
    if side_spin_us:
        assert all(g[3] > 0 for g in got), [g[3] for g in got]        # the delay kernels were really launched on the side stream
    if skip_join:               # negative control: -> the worst relative gradient error (the caller asserts that it is LARGE)
        return max(float(np.linalg.norm(got[0][2][k] - g1) / (np.linalg.norm(g1) + 1e-12)) for k, g1 in grads1.items())
    assert abs(got[0][1] - got[1][1]) < 1e-7         # every rank holds the same global loss
    assert abs(got[0][1] - loss1) < 2e-6 * max(1.0, abs(loss1))
    worst = 0.0
    for k, g1 in grads1.items():
        assert np.array_equal(got[0][2][k], got[1][2][k]), k          # identical summed gradients on both ranks
        den = np.linalg.norm(g1) + 1e-12
        worst = max(worst, float(np.linalg.norm(got[0][2][k] - g1) / den))
    print(f"2-rank vs single-process parameter gradients: worst relative L2 = {worst:.2e}")
    # float16 gradient fragments are rounded under a per-launch loss scale and the wgrad partial sums are ordered by tile:
    # sharding changes both, nothing else (c2f: + the ReLU units a different summation order of the sharded batch flips in float16)
    assert worst < (3e-3 if mode == "nerf" else 3e-2)


def test_two_rank_training_step_equals_single_process():
    _two_ranks("gloo", "nerf")


def test_two_rank_c2f_step_with_early_allreduce_equals_single_process():
    """mode='c2f', in-place gradient buffers, a level's all-reduce started from inside the backward (dist.GradReducer.attach)"""
    _two_ranks("gloo", "c2f")


def test_early_allreduce_is_ordered_behind_a_delayed_side_stream():
    """the level's all-reduce is started from inside the backward pass while the backward entry ran its wgrad
    launches on a per-handle side stream (EVD_BWD_OVERLAP_VOXEL=1 for the PDRF levels).  Every side-stream launch is delayed by 2 ms here
    (EVD_TEST_SIDE_SPIN_US, a spin kernel in front of it; the worker reports evd_debug_side_spin_count() > 0, and the next test is the
    negative control): the reduced buffers still equal the single-process gradient, i.e. the entry's join of the side stream into the
    caller's stream (hipEventRecord(side) + hipStreamWaitEvent(stream) before it returns) orders the collective behind them."""
    _two_ranks("gloo", "c2f", side_spin_us=2000)


def test_delayed_side_stream_without_the_join_gives_wrong_gradients():
    """The negative control of the test above: the same run with the join disabled (EVD_TEST_SKIP_SIDE_JOIN=1) -- the collective
    (and the gradient read-back) then race the delayed side-stream launches and the summed gradients are WRONG.  If this test ever passes
    its `>` the hook no longer exercises the edge and the positive test proves nothing."""
    worst = _two_ranks("gloo", "c2f", side_spin_us=2000, skip_join=True)
    print(f"join disabled: worst relative gradient error {worst:.2e}")
    assert worst > 0.1, worst


@pytest.mark.parametrize("spin", [0, 2000])
def test_early_allreduce_over_rccl_with_a_delayed_side_stream(spin):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL: one device per rank)")
    _two_ranks("nccl", "c2f", side_spin_us=spin)


@pytest.mark.parametrize("mode", ["nerf", "c2f"])
def test_two_rank_training_step_over_rccl(mode):
    """the same two-rank steps on the `nccl` backend (= RCCL over xGMI), one GPU per rank: runs wherever two devices are visible (the
    single-GPU boxes of this pool skip it), so that the first multi-GPU node exercises RCCL and not gloo"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL: one device per rank)")
    _two_ranks("nccl", mode)
