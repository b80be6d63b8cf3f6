"""Data-parallel sharding of the path across the GPUs of one node.

The reference is single-process / single-GPU (no distributed code at all); this is new functionality. Rays are
independent, so the unit index range (rays, blur pixels with their P sub-exposure rays, or start/end event pairs)
is split contiguously across ranks with no data-path collective. The ONE exchange is the loss: every rank reduces
its shard to the packed partial-sum vectors of losses.py (8 + 4 floats) and a single all-reduce (RCCL over xGMI;
gloo in the CPU tests) makes the global sums available on every rank -- one latency-bound collective per step
instead of one per loss term. Full-frame renders gather row tiles with all_gather.
One process per GPU, launched with torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).
"""
from __future__ import annotations

import os

import torch


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous [lo, hi) of the rank's units; sizes differ by at most one, empty shards allowed."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_pixels(n_pixels: int, P: int, rank: int, world: int):
    """Blur batches: keep the P sub-exposure rays of a pixel on one rank. Returns (pixel range, ray range)."""
    lo, hi = shard_range(n_pixels, rank, world)
    return (lo, hi), (lo * P, hi * P)


def init_from_env(backend: str | None = None):
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class _AllReduceSum(torch.autograd.Function):
    """sum over ranks as an autograd node.  Every rank goes on to compute the SAME global loss from the reduced vector and
    back-propagates it through its own shard only, so the backward is the identity (d global sum / d local partial = 1); the
    parameter gradients of the shards are then summed over ranks (GradReducer), which gives the single-process gradient."""

    @staticmethod
    def forward(ctx, flat):
        import torch.distributed as dist
        out = flat.detach().clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out

    @staticmethod
    def backward(ctx, g):
        return g


def all_reduce_partials(*partials):
    """Sum the packed loss partial vectors over ranks with ONE collective and return them (use the returned tensors).
    Plain tensors are also updated in place; tensors that carry an autograd graph (the outputs of the fused loss nodes
    of losses.py) come back as new tensors behind a differentiable all-reduce.  A no-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return partials
    flat = torch.cat([p.reshape(-1) for p in partials])
    if flat.requires_grad:
        red = _AllReduceSum.apply(flat)
        out, off = [], 0
        for p in partials:
            out.append(red[off:off + p.numel()].reshape(p.shape))
            off += p.numel()
        return tuple(out)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for p in partials:
        p.copy_(flat[off:off + p.numel()].reshape(p.shape))
        off += p.numel()
    return partials


class GradReducer:
    """Data-parallel training: sums the gradients of the trainable tensors over ranks (2.4 MB for the NeRF pair,
    ~150 MB for the PDRF grids).  The tensors are packed into a few large flat buckets -- xGMI is point-to-point, a ring
    all-reduce is per-link bound, so few large messages beat many small ones -- each bucket's all-reduce is launched
    asynchronously as soon as it is packed (the first collectives overlap the packing of the next buckets) and `wait()`
    scatters the sums back into the .grad tensors.  With loss terms normalised by GLOBAL counts (the packed loss partials of
    losses.py already are after all_reduce_partials) the summed gradient equals the single-process one.

        reducer = GradReducer(params)           # once
        loss.backward(); reducer.start(); ...; reducer.wait(); optimizer.step()

    flat_buffers (NeRFAll.grad_buffers() of a model trained with enable_training(grads_in_place=True)): [(buffer, params)] -- persistent
    flat gradient buffers whose slices ARE the .grad of `params`.  Such a buffer is all-reduced where it lies: no packing copy before
    and no scatter copy after the collective (the 150 MB of PDRF grid gradients are two messages, the level networks two more).  If
    the gradients of a group are not attached to its buffer at start() (the caller never ran a backward in place), the group falls
    back to the bucket path for that step.
    """

    def __init__(self, params, bucket_bytes=64 << 20, flat_buffers=()):
        self.flat_groups = [(buf, [p for p in ps if p is not None]) for buf, ps in flat_buffers]
        in_flat = {id(p) for _, ps in self.flat_groups for p in ps}
        self.bucket_bytes = bucket_bytes
        self.params = [p for p in params if p is not None and id(p) not in in_flat]
        self.buckets = self._make_buckets(self.params)
        self._flat = [None] * len(self.buckets)
        self._work = []
        self._late = []
        self._early = set()         # flat groups whose all-reduce was started from a level's "gradients complete" callback (attach)
        self.early_starts = 0
        self._armed = True          # False inside no_sync()

    def attach(self, model):
        """Start a level's in-place gradient buffers as soon as that level's LAST backward node of the iteration has run (voxnerf._bwd_done)
        instead of at start(): in the blurfactory iteration the fine level -- 150 MB of grid gradients, the largest message -- is complete
        while the coarse level's last scatter and networks still run (~1.5 ms), and the coarse level while the blur kernel's own backward
        runs.  model.grad_buffers() order: per level [network buffer, grid buffer]; levels whose nets have no callback slot are left to start().

        PRECONDITIONS (checked where they can be):
        * one optimizer step = forward(s) ... backward(s) ... start() ... wait().  Several forwards whose backwards all run before
          start() are counted per level (the callback fires when the LAST pending node of the level has run).  A step that accumulates
          gradients over micro-batches -- forward, backward, forward, backward, ... -- must run all but the last backward inside
          ``with reducer.no_sync():``; otherwise the first backward already starts the all-reduce on partial sums.  A forward of an
          attached level under autograd while one of its all-reduces is in flight raises RuntimeError (before any kernel of the second
          micro-batch adds into a buffer the collective is reading).
        * every gradient contribution to the attached levels' parameters comes from the library's autograd nodes (gather, level
          networks, TV).  A torch-side term on those leaves (e.g. a weight-decay loss written as (p ** 2).sum()) reaches .grad through
          AccumulateGrad, possibly after the level's last library node: use no_sync() for such a step, or do not attach.
        * ordering against the backward entries' side streams: evd_*_mlp_backward joins its per-handle side stream into the caller's
          stream before it returns (csrc/voxel_train_kernel.h, nerf_train_kernel.h: hipEventRecord(side) + hipStreamWaitEvent(stream)),
          and the process group orders the collective behind the work already enqueued on the current stream, so a collective started
          here sees the complete buffers.  tests/test_gpu_dist.py delays every side-stream launch by 2 ms (with the voxel levels' side
          stream switched on, asserting that the delay kernels ran) and has a negative control with the join disabled, whose gradients
          must come out wrong; over RCCL itself the test needs a second GPU and has not run yet.
        EVD_NO_EARLY_ALLREDUCE=1: attach() installs nothing (every message starts at start())."""
        if os.environ.get("EVD_NO_EARLY_ALLREDUCE", "0") not in ("", "0"):
            return self
        lv = [n for n in (getattr(model, "mlp_coarse", None), getattr(model, "mlp_fine", None)) if n is not None]
        per = len(self.flat_groups) // max(len(lv), 1) if lv else 0
        if not lv or per * len(lv) != len(self.flat_groups):
            return self
        for i, net in enumerate(lv):
            idx = list(range(i * per, (i + 1) * per))
            net._pending_bwd = 0
            net._grads_ready_cb = (lambda ids=idx: self._start_groups(ids))
            net._fwd_guard_cb = (lambda ids=idx: self._guard_forward(ids))
        self._attached_nets = lv
        return self

    def detach(self):
        """undo attach(): the levels' callbacks are removed"""
        for net in getattr(self, "_attached_nets", ()):
            net._grads_ready_cb = net._fwd_guard_cb = None
            net._pending_bwd = 0
        self._attached_nets = []

    class _NoSync:
        def __init__(self, red):
            self.red = red

        def __enter__(self):
            self.red._armed = False
            return self.red

        def __exit__(self, *exc):
            self.red._armed = True
            for net in getattr(self.red, "_attached_nets", ()):      # the next micro-batch counts its own nodes
                net._pending_bwd = 0
            return False

    def no_sync(self):
        """Context for the backward passes of all micro-batches but the last of a gradient-accumulation step (the name DDP uses): inside
        it no level starts its all-reduce, the in-place buffers keep accumulating.  start() inside the context raises."""
        return GradReducer._NoSync(self)

    def _guard_forward(self, ids):
        if any(i in self._early for i in ids):
            raise RuntimeError("GradReducer: a forward of a level whose gradient all-reduce of THIS step is already in flight -- a second "
                               "micro-batch after the first backward.  Run all but the last backward of a step inside reducer.no_sync() "
                               "(and call wait() before the next step's forward)")

    def _start_groups(self, ids):
        import torch.distributed as dist
        if not self._armed:
            return
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        for i in ids:
            buf, ps = self.flat_groups[i]
            if i in self._early:
                raise RuntimeError("GradReducer: a level finished a second backward after its all-reduce was started; gradients of this "
                                   "step are invalid.  Use reducer.no_sync() around all but the last backward of a step")
            if self._attached(buf, ps):
                self._work.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), None))
                self._early.add(i)
                self.early_starts += 1

    def _make_buckets(self, params):
        bucket_bytes = self.bucket_bytes
        buckets, cur, size = [], [], 0
        for p in params:
            nb = p.numel() * 4
            if cur and size + nb > bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nb
        if cur:
            buckets.append(cur)
        return buckets

    @staticmethod
    def _attached(buf, params):
        """every parameter's .grad is a view of `buf` (same storage, inside its extent)"""
        lo, hi = buf.data_ptr(), buf.data_ptr() + buf.numel() * buf.element_size()
        return all(p.grad is not None and p.grad.is_contiguous() and lo <= p.grad.data_ptr() and p.grad.data_ptr() + p.grad.numel() * 4 <= hi
                   for p in params)

    def start(self):
        import torch.distributed as dist
        if not self._armed:
            raise RuntimeError("GradReducer.start() inside no_sync(): the last micro-batch's backward runs outside the context")
        self._work, self._late = list(self._work) if self._early else [], []
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        for gi, (buf, ps) in enumerate(self.flat_groups):            # in place: the largest messages first
            if gi in self._early:
                continue                              # already in flight (attach)
            if self._attached(buf, ps):
                self._work.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), None))
            else:
                self._late += ps
        late_buckets = self._make_buckets(self._late) if self._late else []
        self._late_buckets, self._late_flat = late_buckets, [None] * len(late_buckets)
        for i, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            if self._flat[i] is None or self._flat[i].numel() != n:
                self._flat[i] = torch.empty((n,), dtype=torch.float32, device=bucket[0].device)
            off = 0
            for p in bucket:
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                self._flat[i][off:off + p.numel()].copy_(g.reshape(-1))
                off += p.numel()
            self._work.append((dist.all_reduce(self._flat[i], op=dist.ReduceOp.SUM, async_op=True), (bucket, self._flat[i])))
        for i, bucket in enumerate(late_buckets):
            n = sum(p.numel() for p in bucket)
            flat = torch.empty((n,), dtype=torch.float32, device=bucket[0].device)
            off = 0
            for p in bucket:
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                flat[off:off + p.numel()].copy_(g.reshape(-1))
                off += p.numel()
            self._work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), (bucket, flat)))

    def wait(self):
        for w, packed in self._work:
            w.wait()
            if packed is None:
                continue
            bucket, flat = packed
            off = 0
            for p in bucket:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + p.numel()].reshape(p.shape))
                off += p.numel()
        self._work = []
        self._early = set()
        for net in getattr(self, "_attached_nets", ()):     # a forward whose output never reached the loss leaves a count behind
            net._pending_bwd = 0


def gather_rows(local_rows: torch.Tensor, n_total: int):
    """Full-frame render: every rank holds the rows [lo, hi) of an [n_total, ...] image; returns the whole image
    on every rank (all_gather of padded tiles; xGMI moves 1.9 MB for a 400x400 RGB view)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_rows
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_total // world)
    pad = torch.zeros((per,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    dev = pad.device
    if dist.get_backend() == "gloo" and pad.is_cuda:          # gloo has no device all_gather (CPU tests, 1-GPU validation runs)
        pad = pad.cpu()
    tiles = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(tiles, pad)
    tiles = [t.to(dev) for t in tiles]
    out = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        out.append(tiles[r][: hi - lo])
    return torch.cat(out, 0)
