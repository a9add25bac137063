"""Data-parallel layer: one process per GPU, torch.distributed ('nccl' == RCCL over xGMI on ROCm; 'gloo' on
CPU for tests).  Replaces the reference's Horovod usage on the hot path:
  hvd.DistributedOptimizer (optuna_objective.py:179-186) -> DistributedOptimizer: bucketed all-reduce of the
      flat gradient buffer, launched from autograd hooks while backward is still running (RCCL runs on its own
      HIP stream; the fused Adam waits on the bucket events), averaged by folding 1/world into the Adam kernel;
  hvd.broadcast_global_variables(0) (optuna_objective.py:328,375,413) -> broadcast_global_variables;
  MPI scatter of file lists (dataset.py:307-333) -> a shared-seed permutation each rank slices (dataset.py).
xGMI is point-to-point (7 links per GPU): large buckets amortise the ring's latency, but the bucket that becomes
ready LAST is exposed (the generator's parameter-heavy low-resolution layers finish its backward), so the default
is 32 MiB (SARAGAN_BUCKET_MIB overrides); the whole gradient of a small network still goes out as one message."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, timeout_s=600):
    """Initialises torch.distributed from torchrun's env (RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, world_size, local_rank).  Single process: (0, 1, 0) without a process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        import datetime
        if backend is None:
            backend = os.environ.get('SARAGAN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s))
    return rank, world, local


def size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def reachable_leaves(roots):
    """ids of the leaf tensors whose AccumulateGrad node is reachable from the grad_fn of any of `roots`."""
    seen, out = set(), set()
    stack = [r.grad_fn for r in roots if r is not None and r.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, 'variable', None)
        if var is not None:
            out.add(id(var))
        for nxt, _ in fn.next_functions:
            if nxt is not None:
                stack.append(nxt)
    return out


class GradientAllReducer:
    """Sums contiguous slices ("buckets") of a flat gradient buffer across ranks, overlapping with backward.

    begin(flat_grad, ranges, params) arms the per-parameter countdowns; each parameter's
    post-accumulate-grad hook decrements its bucket; a bucket whose parameters are all done is all-reduced
    asynchronously.  finish() launches whatever is left and waits."""

    def __init__(self, group=None, bucket_bytes=None):
        if bucket_bytes is None:
            bucket_bytes = int(os.environ.get('SARAGAN_BUCKET_MIB', '32')) << 20
        self.group = group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self._hooked = {}
        self._plan_key = None
        self._buckets = []
        self._handles = []
        self._armed = False
        self.timing = bool(int(os.environ.get('SARAGAN_DP_TIMING', '0')))
        self._spans = []       # (event before the waits, event after): GPU time the compute stream sat in finish()

    def _plan(self, flat_grad, ranges, params):
        key = (flat_grad.data_ptr(), tuple(ranges), len(params))
        if key == self._plan_key:
            return
        self._plan_key = key
        buckets = []
        for (o, n) in ranges:
            pos = o
            while pos < o + n:
                ln = min(self.bucket_elems, o + n - pos)
                buckets.append(dict(off=pos, len=ln, params=0, left=0, launched=False))
                pos += ln
        base = flat_grad.data_ptr()
        self._owner = {}
        for p in params:
            off = (p.grad.data_ptr() - base) // 4
            for bi, b in enumerate(buckets):          # a parameter belongs to every bucket it overlaps
                if off < b['off'] + b['len'] and off + p.numel() > b['off']:
                    b['params'] += 1
                    self._owner.setdefault(id(p), []).append(bi)
            if id(p) not in self._hooked:
                self._hooked[id(p)] = p.register_post_accumulate_grad_hook(self._hook)
        self._buckets = buckets
        self._flat = flat_grad

    def begin(self, flat_grad, ranges, params, roots=None):
        """roots: the tensors backward starts from.  Parameters the graph below them does not reach (the faded-out
        branch of a stabilising phase, networks.ops.lerp) never fire their hook: they are counted as done here, so
        that their bucket goes out as soon as its live parameters are ready instead of waiting for finish()."""
        self._plan(flat_grad, ranges, params)
        for b in self._buckets:
            b['left'] = b['params']
            b['launched'] = False
        self._handles = []
        self._armed = True
        if roots is not None:
            reach = reachable_leaves(roots)
            for p in params:
                if id(p) not in reach:
                    self._hook(p)

    def _launch(self, b):
        b['launched'] = True
        if self.world_size > 1:
            view = self._flat[b['off']:b['off'] + b['len']]
            self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _hook(self, p):
        if not self._armed:
            return
        for bi in self._owner.get(id(p), ()):
            b = self._buckets[bi]
            b['left'] -= 1
            if b['left'] == 0 and not b['launched']:
                self._launch(b)

    def finish(self):
        self._armed = False
        for b in self._buckets:
            if not b['launched']:
                self._launch(b)
        span = None
        if self.timing and self._handles and torch.cuda.is_available() and self._flat.is_cuda:
            span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            span[0].record()
        for h in self._handles:
            h.wait()           # NCCL/RCCL: the current stream waits for the collective's stream; gloo: the host does
        if span is not None:
            span[1].record()
            self._spans.append(span)
        self._handles = []

    def exposed_ms(self, reset=True):
        """Per finish() call: milliseconds the compute stream spent between "backward enqueued" and "last bucket
        reduced", i.e. the all-reduce time NOT hidden behind backward (enable with .timing / SARAGAN_DP_TIMING=1;
        synchronises).  The buckets launched from the autograd hooks overlap; what is left here is the tail."""
        out = []
        for e0, e1 in self._spans:
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        if reset:
            self._spans = []
        return out


def DistributedOptimizer(optimizer, group=None, bucket_bytes=None, op=None):
    """hvd.DistributedOptimizer(optimizer): gradients are averaged over ranks before they are applied.
    (`op` is accepted for the reference's Adasum call site, optuna_objective.py:182-183, and ignored.)"""
    optimizer.distributed = GradientAllReducer(group, bucket_bytes)
    return optimizer


def broadcast_global_variables(store, root_rank=0, group=None):
    """hvd.broadcast_global_variables(root): every variable takes rank `root_rank`'s value.  Flat buffers go out
    as one message per network."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    done = set()
    for prefix, flat in store.flat.items():
        dist.broadcast(flat['param'], src=root_rank, group=group)
        done.update(flat['offsets'].keys())
    for k, v in store.vars.items():
        if k not in done:
            dist.broadcast(v.data, src=root_rank, group=group)
