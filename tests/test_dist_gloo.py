"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding + the single packed loss all-reduce and the
row-tile gather of evdeblurnerf_amd/dist.py (the HIP kernels themselves need a GPU; here the per-shard partials
come from a numpy stand-in with the same packed layout, which is what the collective sees)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partials_numpy(pred, tgt):
    """packed layout of losses.blur_loss_partials: [se_rgb, 0, 0, 0, 0, n_elem, 0, 0]"""
    p = np.zeros(8, np.float32)
    p[0] = ((pred - tgt) ** 2).sum()
    p[5] = pred.size
    return p


def _worker(rank, world, port, R, P, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from evdeblurnerf_amd import dist as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    rs = np.random.RandomState(0)
    pred = rs.rand(R, 3).astype(np.float32)
    tgt = rs.rand(R, 3).astype(np.float32)
    ev = rs.rand(4).astype(np.float32)
    (plo, phi), (rlo, rhi) = D.shard_pixels(R, P, rank, world)
    assert (rlo, rhi) == (plo * P, phi * P)
    blur = torch.from_numpy(_partials_numpy(pred[plo:phi], tgt[plo:phi]))
    event = torch.from_numpy(ev * (rank + 1))
    D.all_reduce_partials(blur, event)
    rows = torch.from_numpy(pred[plo:phi])
    full = D.gather_rows(rows, R)
    q.put((rank, blur.numpy().copy(), event.numpy().copy(), full.numpy().copy()))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("R", [1024, 7, 1])
def test_two_rank_loss_allreduce_and_gather(R):
    world, P = 2, 10
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, R, P, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rs = np.random.RandomState(0)
    pred = rs.rand(R, 3).astype(np.float32)
    tgt = rs.rand(R, 3).astype(np.float32)
    ev = rs.rand(4).astype(np.float32)
    ref = _partials_numpy(pred, tgt)
    for rank, blur, event, full in res:
        assert np.allclose(blur, ref, rtol=1e-5)                       # global sums on every rank
        assert blur[5] == 3 * R
        assert np.allclose(event, ev * 3, rtol=1e-6)
        assert np.array_equal(full, pred)                              # row tiles reassemble the frame, ragged shards included
        mse = blur[0] / blur[5]
        assert abs(mse - ((pred - tgt) ** 2).mean()) < 1e-6


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from evdeblurnerf_amd import dist as D
    D.init_from_env("gloo")
    torch.manual_seed(0)
    # a stand-in model: the flat NeRF parameter tensor and three grid tensors of different sizes (one larger than a bucket)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 70000, 33, 5)]
    x = torch.randn(64, 8)
    lo, hi = D.shard_range(64, rank, world)
    loss = sum((p[:8] * x[lo:hi]).sum() * (i + 1) for i, p in enumerate(params[:3])) / 64      # global normalisation; params[3] unused
    loss.backward()
    red = D.GradReducer(params, bucket_bytes=100000)
    assert len(red.buckets) >= 2
    red.start()
    red.wait()
    q.put((rank, [p.grad.numpy().copy() for p in params]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 70000, 33, 5)]
    x = torch.randn(64, 8)
    loss = sum((p[:8] * x).sum() * (i + 1) for i, p in enumerate(params[:3])) / 64
    loss.backward()
    for rank, grads in res:
        for g, p in zip(grads[:3], params[:3]):
            assert np.allclose(g, p.grad.numpy(), rtol=1e-5, atol=1e-7)
        assert np.array_equal(grads[3], np.zeros(5, np.float32))        # a parameter without gradient on any rank stays zero


def _flat_worker(rank, world, port, q):
    """the training leg of bench.py --gpus N on CPU stand-ins: parameters whose .grad are slices of persistent flat buffers
    (NeRFAll.enable_training(grads_in_place=True) -> grad_buffers()), reduced where they lie; one group is NOT attached (falls back
    to the bucket path); plus ordinary parameters (the blur kernel's) in buckets"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from evdeblurnerf_amd import dist as D
    D.init_from_env("gloo")
    torch.manual_seed(3)
    sizes = (700, 41, 5000)
    buf_a, buf_b = torch.zeros(sum(sizes)), torch.zeros(123)
    group_a, off = [], 0
    for n in sizes:
        p = torch.nn.Parameter(torch.randn(n))
        p.grad = buf_a[off:off + n]
        off += n
        group_a.append(p)
    group_b = [torch.nn.Parameter(torch.randn(123))]                 # its buffer exists, but the gradient was produced by plain autograd
    other = [torch.nn.Parameter(torch.randn(n)) for n in (17, 300)]
    g = torch.Generator().manual_seed(100 + rank)
    for p in group_a:                                                # "backward kernels" add into the slices
        p.grad += torch.randn(p.numel(), generator=g)
    group_b[0].grad = torch.randn(123, generator=g)
    for p in other:
        p.grad = torch.randn(p.numel(), generator=g)
    ptr = buf_a.data_ptr()
    red = D.GradReducer(group_a + group_b + other, bucket_bytes=1000, flat_buffers=[(buf_a, group_a), (buf_b, group_b)])
    red.start()
    red.wait()
    assert buf_a.data_ptr() == ptr and all(p.grad.data_ptr() >= ptr for p in group_a)          # reduced in place, still attached
    q.put((rank, [p.grad.numpy().copy() for p in group_a + group_b + other]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_on_flat_buffers_in_place():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = (700, 41, 5000, 123, 17, 300)
    expect = [np.zeros(n, np.float32) for n in sizes]
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        for i, n in enumerate(sizes):
            expect[i] += torch.randn(n, generator=g).numpy()
    for rank, grads in res:
        for a, b in zip(grads, expect):
            assert np.allclose(a, b, rtol=1e-6, atol=1e-6)


def _accum_worker(rank, world, port, q):
    """GradReducer.attach with gradient accumulation: stand-in levels drive voxnerf's counting hooks the way the
    library's autograd nodes do (_fwd_noted at a forward under autograd, _bwd_done when a backward node has added into the level's
    in-place buffer).  Two micro-batches per step: the first backward inside no_sync() must NOT start a collective, the second one
    does, and the reduced buffers hold the sum over both micro-batches and both ranks.  Without no_sync() the second micro-batch's
    forward raises before anything is added."""
    import types
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("EVD_NO_EARLY_ALLREDUCE", None)
    from evdeblurnerf_amd import dist as D
    from evdeblurnerf_amd.voxnerf import _bwd_done, _fwd_noted
    D.init_from_env("gloo")
    bufs = [torch.zeros(n) for n in (300, 5000, 200, 900)]             # per level [networks, grids]
    groups = []
    for b in bufs:
        p = torch.nn.Parameter(torch.zeros(b.numel()))
        p.grad = b[:]
        groups.append((b, [p]))
    model = types.SimpleNamespace(mlp_coarse=types.SimpleNamespace(), mlp_fine=types.SimpleNamespace())
    red = D.GradReducer([], flat_buffers=groups).attach(model)
    g = torch.Generator().manual_seed(50 + rank)

    def micro_batch(levels=(model.mlp_fine, model.mlp_coarse)):
        with torch.enable_grad():
            for lv in levels:                                          # gather + networks of each level
                _fwd_noted(lv)
                _fwd_noted(lv)
        for li, lv in ((1, model.mlp_fine), (0, model.mlp_coarse)):   # backward: fine level first
            for k in range(2):
                bufs[li * 2 + k] += torch.randn(bufs[li * 2 + k].numel(), generator=g)
                _bwd_done(lv)
    with red.no_sync():
        micro_batch()
        assert red.early_starts == 0
        try:
            red.start()
            raise AssertionError("start() inside no_sync() must raise")
        except RuntimeError:
            pass
    micro_batch()
    assert red.early_starts == 4                                      # both levels' two buffers started from inside the "backward"
    red.start()
    red.wait()
    out = [b.clone() for b in bufs]
    # misuse: a second micro-batch after an armed backward
    for b in bufs:
        b.zero_()
    micro_batch()
    raised = False
    try:
        micro_batch()
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    red.start()
    red.wait()
    # EVD_NO_EARLY_ALLREDUCE: attach installs nothing
    os.environ["EVD_NO_EARLY_ALLREDUCE"] = "1"
    m2 = types.SimpleNamespace(mlp_coarse=types.SimpleNamespace(), mlp_fine=types.SimpleNamespace())
    D.GradReducer([], flat_buffers=groups).attach(m2)
    q.put((rank, [o.numpy() for o in out], raised, getattr(m2.mlp_fine, "_grads_ready_cb", None) is None))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_accumulation_with_no_sync():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = (300, 5000, 200, 900)
    expect = [np.zeros(n, np.float32) for n in sizes]
    for rank in range(world):
        g = torch.Generator().manual_seed(50 + rank)
        for _ in range(2):                                             # two micro-batches, fine level (buffers 2, 3) first
            for i in (2, 3, 0, 1):
                expect[i] += torch.randn(sizes[i], generator=g).numpy()
    for rank, out, raised, env_off in res:
        assert raised and env_off
        for a, b in zip(out, expect):
            assert np.allclose(a, b, rtol=1e-6, atol=1e-6)


def test_shard_range_partitions():
    from evdeblurnerf_amd.dist import shard_range
    for n in (0, 1, 7, 8, 4096, 160000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
