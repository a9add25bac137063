"""Data-parallel layer: one process per GPU, torch.distributed ('nccl' == RCCL over xGMI on ROCm; 'gloo' on
CPU for tests).  Replaces the reference's Horovod usage on the hot path:
  hvd.DistributedOptimizer (optuna_objective.py:179-186) -> DistributedOptimizer: bucketed all-reduce of the
      flat gradient buffer, launched from autograd hooks while backward is still running (RCCL runs on its own
      HIP stream; the fused Adam waits on the bucket events), averaged by folding 1/world into the Adam kernel;
  hvd.broadcast_global_variables(0) (optuna_objective.py:328,375,413) -> broadcast_global_variables;
  MPI scatter of file lists (dataset.py:307-333) -> a shared-seed permutation each rank slices (dataset.py).
xGMI is point-to-point (7 links per GPU): large buckets amortise the ring's latency, but the bucket that becomes
ready LAST is exposed (the generator's parameter-heavy low-resolution layers finish its backward), so the default
is 32 MiB (SARAGAN_BUCKET_MIB overrides); the whole gradient of a small network still goes out as one message.
SARAGAN_DP_GRAD_DTYPE=bf16 sends every bucket as bfloat16 (Horovod's Compression.fp16 of pgan_pytorch/main.py:149, with the
wider exponent): the f32 gradients are rounded once into a staging bucket, summed by the collective in bf16, and widened back
into the f32 flat buffer that the optimiser kernel reads and accumulates from in f32 -- half the bytes per xGMI link.
SARAGAN_DP_ALGO selects how a bucket is summed: "allreduce" (default: one dist.all_reduce, RCCL picks the algorithm) or
"rs_ag" (reduce-scatter then all-gather issued by hand -- on a fully connected xGMI node each of the two is one direct
exchange over all 7 links; offered so that the first 8-GPU run can A/B it against RCCL's own choice).
hvd.DistributedOptimizer(op=hvd.Adasum) (optuna_objective.py:180-183, the discriminator under --use_adasum) ->
AdasumReducer."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, timeout_s=600):
    """Initialises torch.distributed from torchrun's env (RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, world_size, local_rank).  Single process: (0, 1, 0) without a process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or forced()) and not dist.is_initialized():
        import datetime
        if world == 1 and 'RANK' not in os.environ:      # the one-rank rehearsal started as a plain process
            import socket
            with socket.socket() as s_:
                s_.bind(('127.0.0.1', 0))
                port = s_.getsockname()[1]
            os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        if backend is None:
            backend = os.environ.get('SARAGAN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s))
    return rank, world, local


def forced():
    """SARAGAN_DP_FORCE=1: a single rank still creates its process group and issues every collective (a sum over one rank).
    A one-GPU box can then run the RCCL calls of the N-rank path -- communicator start-up, the bucket collectives on RCCL's
    stream ordered against the compute stream, the broadcast -- and must get the single-process result."""
    return os.environ.get('SARAGAN_DP_FORCE', '0') == '1'


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
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.algo = os.environ.get('SARAGAN_DP_ALGO', 'allreduce')
        if self.algo not in ('allreduce', 'rs_ag'):
            raise ValueError(f'SARAGAN_DP_ALGO must be "allreduce" or "rs_ag", got {self.algo!r}')
        self.grad_dtype = os.environ.get('SARAGAN_DP_GRAD_DTYPE', 'f32')
        if self.grad_dtype not in ('f32', 'bf16'):
            raise ValueError(f'SARAGAN_DP_GRAD_DTYPE must be "f32" or "bf16", got {self.grad_dtype!r}')
        self._widen = []       # (bf16 staging bucket, f32 slice it came from): copied back after the waits
        # in-order queues (RCCL: every collective of a group runs on that group's stream in issue order) let the
        # all-gather be issued right behind its reduce-scatter; other backends (gloo) get an explicit wait in between
        self._ordered = dist.is_initialized() and dist.get_backend(group) == 'nccl'
        self._hooked = {}
        self._plan_key = None
        self._buckets = []
        self._handles = []
        self._armed = False
        self.land = None       # optional callable(p): makes sure p.grad lives in the flat buffer before its bucket is counted down
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
        self._widen = []
        self._armed = True
        if roots is not None:
            reach = reachable_leaves(roots)
            for p in params:
                if id(p) not in reach:
                    self._hook(p)

    @property
    def grad_scale(self):
        """What the summed buffer still has to be multiplied with to become Horovod's average."""
        return 1.0 / self.world_size

    def _launch(self, b):
        b['launched'] = True
        if self.world_size > 1 or (forced() and dist.is_initialized()):
            view = self._flat[b['off']:b['off'] + b['len']]
            if self.grad_dtype == 'bf16':      # one rounding on the way out; the collective sums in bf16
                st = b.get('stage')
                if st is None or st.numel() != view.numel() or st.device != view.device:
                    st = b['stage'] = torch.empty(view.numel(), dtype=torch.bfloat16, device=view.device)
                st.copy_(view)
                self._widen.append((st, view))
                view = st
            if self.algo == 'rs_ag':
                self._launch_rs_ag(view)
            else:
                self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    @staticmethod
    def rs_ag_plan(numel, world, rank):
        """Chunk arithmetic of the in-place reduce-scatter + all-gather of a bucket of `numel` elements: (chunk, body, lo, hi):
        the first `body` = chunk * world elements are split into `world` equal chunks, rank r owns [lo, hi) = [r * chunk,
        (r + 1) * chunk); the tail [body, numel) (fewer than `world` elements) goes through a plain all_reduce."""
        chunk = numel // world
        return chunk, chunk * world, rank * chunk, (rank + 1) * chunk

    def _launch_rs_ag(self, view):
        """The same sum as all_reduce(view), as reduce-scatter + all-gather IN PLACE: rank r reduces the r-th of `world`
        equal chunks into its own slot of the bucket, then every rank gathers all slots.  A tail that does not divide by
        the world size (< world elements) goes through a plain all_reduce.
        The output of the reduce-scatter aliases its input at exactly the offset NCCL / RCCL document as their in-place form
        (ncclReduceScatter: "in-place operation will happen if recvbuff == sendbuff + rank * recvcount"; ncclAllGather:
        "sendbuff == recvbuff + rank * sendcount"): asserted below, so a change of the chunk arithmetic cannot turn it into an
        undefined overlap."""
        w = self.world_size
        chunk, nbody, lo, hi = self.rs_ag_plan(view.numel(), w, self.rank)
        if chunk:
            body = view[:nbody]
            mine = body[lo:hi]
            assert mine.data_ptr() == body.data_ptr() + self.rank * chunk * body.element_size() and mine.numel() * w == body.numel()
            h = dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if not self._ordered:
                h.wait()
            else:
                self._handles.append(h)
            self._handles.append(dist.all_gather_into_tensor(body, mine, group=self.group, async_op=True))
        if view.numel() > chunk * w:
            self._handles.append(dist.all_reduce(view[chunk * w:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _hook(self, p):
        if not self._armed:
            return
        if self.land is not None and p.grad is not None:
            self.land(p)
        for bi in self._owner.get(id(p), ()):
            b = self._buckets[bi]
            b['left'] -= 1
            if b['left'] == 0 and not b['launched']:
                self._launch(b)

    def launch_all(self):
        """Launches every bucket that has not gone out yet WITHOUT waiting (the captured step: a replayed segment has completed
        this network's gradients; the next segment is replayed while these run).  Returns the number launched."""
        n = 0
        for b in self._buckets:
            if not b['launched']:
                self._launch(b)
                n += 1
        return n

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
        for st, view in self._widen:
            view.copy_(st)     # back into the f32 buffer the optimiser reads (its state and arithmetic stay f32)
        self._widen = []
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


Average, Adasum = 'Average', 'Adasum'      # hvd.Average / hvd.Adasum


def adasum_pair(a, b, seg, nseg):
    """Adasum of two gradient vectors, per tensor (Horovod 0.19 `ops/adasum/adasum.h`, third-party and not in the
    reference tree: restated from the published rule, Maleki et al. 2020):
        a' = (1 - <a,b> / (2 |a|^2)) a + (1 - <a,b> / (2 |b|^2)) b       per tensor; a coefficient stays 1 where the norm is ~0.
    `seg` maps every element of the flat vectors to its tensor's index (alignment padding has a segment of its own)."""
    dot = torch.zeros(nseg, device=a.device, dtype=torch.float64).index_add_(0, seg, (a * b).double())
    na = torch.zeros(nseg, device=a.device, dtype=torch.float64).index_add_(0, seg, (a * a).double())
    nb = torch.zeros(nseg, device=a.device, dtype=torch.float64).index_add_(0, seg, (b * b).double())
    one = torch.ones_like(dot)
    ca = torch.where(na >= 1e-8, 1.0 - dot / (2.0 * na.clamp_min(1e-30)), one).to(a.dtype)
    cb = torch.where(nb >= 1e-8, 1.0 - dot / (2.0 * nb.clamp_min(1e-30)), one).to(a.dtype)
    return ca[seg] * a + cb[seg] * b


class AdasumReducer(GradientAllReducer):
    """hvd.DistributedOptimizer(optimizer, op=hvd.Adasum) (optuna_objective.py:182-183) in the form Horovod gives a TF1
    optimizer (`_DistributedAdasumOptimizer`, horovod/tensorflow/__init__.py of 0.19 -- third-party, not in the reference tree,
    restated from its published behaviour): compute_gradients, clipping and the gradient norms stay LOCAL; apply_gradients
    keeps the variables' start values, applies the local optimiser step, combines the ranks' weight DELTAS (var - start) by
    the Adasum rule and writes start + combined delta.  (Adam normalises the gradient's scale, so Adasum of the gradients
    followed by Adam -- what rounds 2-3 did -- is a different update.)  The rule runs along a binary tree (ranks (0,1), (2,3),
    ... first, then pairs of pairs); every rank gathers all deltas (one all_gather of the flat range: 8 x 116 MB for the
    largest discriminator of the presets) and evaluates the tree locally, so all ranks hold bit-identical weights; the world
    size must be a power of two, as Horovod requires.  optimization.StepGraph._finish drives it (`delta_form`)."""
    delta_form = True

    def __init__(self, group=None, bucket_bytes=None):
        super().__init__(group, bucket_bytes)
        if self.world_size & (self.world_size - 1):
            raise ValueError(f'Adasum needs a power-of-two number of ranks, got {self.world_size}')
        self._seg_key = None

    @property
    def grad_scale(self):
        return 1.0

    def begin(self, flat_grad, ranges, params, roots=None):
        self._flat, self._ranges, self._params = flat_grad, list(ranges), list(params)
        self._armed = False

    def finish(self):
        """Nothing to wait for: the gradients are not reduced in this form."""

    def _segments(self, lo, hi):
        key = (self._flat.data_ptr(), lo, hi, len(self._params))
        if key != self._seg_key:
            base = self._flat.data_ptr()
            seg = torch.full((hi - lo,), len(self._params), dtype=torch.int64)
            for i, p in enumerate(self._params):
                off = (p.grad.data_ptr() - base) // 4 - lo
                seg[off:off + p.numel()] = i
            self._seg, self._nseg, self._seg_key = seg.to(self._flat.device), len(self._params) + 1, key
        return self._seg, self._nseg

    def hull(self):
        """[lo, hi) of the flat buffers that the armed ranges span."""
        return min(o for o, _ in self._ranges), max(o + n for o, n in self._ranges)

    def combine(self, vec):
        """In place: vec[lo:hi] (one flat vector per rank, laid out like the gradient buffer begin() was given) becomes the
        Adasum of the ranks' vectors, per tensor."""
        if (self.world_size == 1 and not (forced() and dist.is_initialized())) or not self._ranges:
            return
        lo, hi = self.hull()
        mine = vec[lo:hi]
        allg = torch.empty(self.world_size * (hi - lo), device=mine.device, dtype=mine.dtype)
        dist.all_gather_into_tensor(allg, mine.contiguous(), group=self.group)
        allg = allg.view(self.world_size, hi - lo)
        seg, nseg = self._segments(lo, hi)
        level = [allg[r] for r in range(self.world_size)]
        while len(level) > 1:
            level = [adasum_pair(level[i], level[i + 1], seg, nseg) for i in range(0, len(level), 2)]
        mine.copy_(level[0])

    def combine_deltas(self, param_flat, start):
        """After the LOCAL optimiser step: param[lo:hi] = start + Adasum over ranks of (param[lo:hi] - start); `start` is the
        clone of param_flat[lo:hi] taken before the step."""
        lo, hi = self.hull()
        delta = torch.zeros_like(param_flat)
        delta[lo:hi] = param_flat[lo:hi] - start
        self.combine(delta)
        param_flat[lo:hi] = start + delta[lo:hi]


def DistributedOptimizer(optimizer, group=None, bucket_bytes=None, op=Average):
    """hvd.DistributedOptimizer(optimizer, op=hvd.Average | hvd.Adasum): gradients are averaged over ranks (or combined
    by Adasum) before they are applied."""
    if op in (None, Average):
        optimizer.distributed = GradientAllReducer(group, bucket_bytes)
    elif op == Adasum:
        optimizer.distributed = AdasumReducer(group, bucket_bytes)
    else:
        raise ValueError(f'unknown reduction op {op!r} (Average or Adasum)')
    return optimizer


def collective_info(group=None):
    """What bench.py records as config.collective: backend, library version and communicator size."""
    if not dist.is_initialized():
        return None
    info = dict(backend=dist.get_backend(group), world_size=dist.get_world_size(group),
                algo=os.environ.get('SARAGAN_DP_ALGO', 'allreduce'))
    if os.environ.get('SARAGAN_DP_GRAD_DTYPE', 'f32') != 'f32':
        info['grad_dtype'] = os.environ['SARAGAN_DP_GRAD_DTYPE']
    if info['backend'] == 'nccl':
        try:
            info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # pragma: no cover - informational only
            info['rccl_version'] = f'unavailable ({type(e).__name__})'
    return info


def broadcast_global_variables(store, root_rank=0, group=None):
    """hvd.broadcast_global_variables(root): every variable takes rank `root_rank`'s value.  Flat buffers go out
    as one message per network."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not forced()):
        return
    done = set()
    for prefix, flat in store.flat.items():
        dist.broadcast(flat['param'], src=root_rank, group=group)
        done.update(flat['offsets'].keys())
    for k, v in store.vars.items():
        if k not in done:
            dist.broadcast(v.data, src=root_rank, group=group)
    from . import functional as F
    F.mark_packs_stale()      # the flat buffers were rewritten behind the version counters the packed weight images are keyed by
