"""bench.py: arguments, the N-rank launch without a launcher, and the CPU rehearsal of that launch (--dry-run)."""
import argparse
import subprocess
import tempfile
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {   # BASELINE.json configs[i-1]: SURVEY.md section 8d
    1: dict(size='xs', phase=1, latent=256, batch=4, dtype='f32', alpha=0.0, dims=3),
    2: dict(size='xs', phase=4, latent=256, batch=32, dtype='bf16', alpha=0.0, dims=3),
    3: dict(size='s', phase=6, latent=512, batch=32, dtype='bf16', alpha=0.0, dims=3),
    4: dict(size='m', phase=7, latent=512, batch=2, dtype='bf16', alpha=0.5, dims=3),
    5: dict(size='xs', phase=9, latent=512, batch=4, dtype='f32', alpha=0.0, dims=2),
    # the reference's OWN operating point, the only throughput it publishes (SURFGAN_3D/out.txt:18,78,84-1639: 'xs' phase 5,
    # 64x64x16, WGAN-GP 10, latent 512, LOCAL batch 2 on each of 8 Horovod ranks: 47.15 img/s global = 5.9 per GPU).
    # `--config out_txt`; --batch 4 / 8 show what the small local batches of data parallelism at 128^2 / 256^2 cost.
    6: dict(size='xs', phase=5, latent=512, batch=2, dtype='bf16', alpha=0.0, dims=3),
}
CONFIG_NAMES = {'out_txt': 6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=lambda v: CONFIG_NAMES[v] if v in CONFIG_NAMES else int(v), default=3, choices=sorted(CONFIGS),
                    help='1..5: BASELINE.json configs[i-1]; out_txt (6): the reference log\'s own operating point')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the configuration\'s)')
    ap.add_argument('--size', default=None)
    ap.add_argument('--phase', type=int, default=None)
    ap.add_argument('--latent', type=int, default=None)
    ap.add_argument('--dtype', default=None, choices=['bf16', 'f32'])
    ap.add_argument('--loss', default='wgan', choices=['wgan', 'logistic'])
    ap.add_argument('--alpha', type=float, default=None, help='0: stabilising phase; >0: mixing (freeze ops)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the fp32 and loader-in-the-loop legs')
    ap.add_argument('--dump-prof', action='store_true', help='per-shape conv kernel table on stderr')
    ap.add_argument('--cpu-budget-s', type=float, default=14.0)
    ap.add_argument('--dry-run', action='store_true',
                    help='plumbing rehearsal of the N-rank launch on CPU tensors over gloo: no GPU, no throughput (see dry_run)')
    args = ap.parse_args()
    c = CONFIGS[args.config]
    for k in ('size', 'phase', 'latent', 'batch', 'dtype', 'alpha'):
        if getattr(args, k) is None:
            setattr(args, k, c[k])
    args.dims = c['dims']
    return args


# -----------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks (this process never touches a GPU)
# -----------------------------------------------------------------------------------------------------
def spawn_ranks(args, script):
    """The parent only counts devices and starts children; it never initialises a GPU context it would keep, and it
    never replaces its own program.  Children are polled: the first non-zero exit (a rank that died in start-up or in
    its first collective) ends the others instead of leaving them in rendezvous until the distributed timeout, and an
    overall deadline bounds the wait."""
    import socket
    import torch
    have = torch.cuda.device_count()
    stack = bool(int(os.environ.get('SARAGAN_BENCH_STACK_RANKS', '0')))    # rehearsal: several ranks share one GPU (gloo)
    if have < args.gpus and not (stack and have >= 1) and not args.dry_run:
        print(f'bench.py: --gpus {args.gpus} but only {have} device(s) are visible', file=sys.stderr)
        return 3
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode='w+')
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, script] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + float(os.environ.get('SARAGAN_BENCH_DEADLINE_S', '1500'))
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed or time.time() > deadline:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            why = 'a rank failed' if failed else 'deadline passed'
            print(f'bench.py: {why}; rank exit codes {codes}', file=sys.stderr)
            return 1
        time.sleep(0.2)
    out0.seek(0)
    line = [ln for ln in out0.read().splitlines() if ln.startswith('{')]
    if not line:
        print(f'bench.py: rank 0 printed no result line; rank exit codes {codes}', file=sys.stderr)
        return 1
    rec = json.loads(line[-1])
    if rec.get('n_gpus') != args.gpus:
        print(f"bench.py: {args.gpus} ranks requested, {rec.get('n_gpus')} took part", file=sys.stderr)
        return 1
    print(line[-1], flush=True)
    return 0


SETTLE_STEPS_MULTI_RANK = 60      # untimed steps after the warm-up when world > 1 (the count must match across ranks)


def dry_run(args, rank, world):
    """`--dry-run`: the plumbing of an N-rank launch exercised WITHOUT a GPU, so that the first real 8-GPU run cannot fail on
    it: spawn_ranks (or torch.distributed.run) -> rendezvous on 127.0.0.1 -> the gradient reducer's bucketed all-reduce (gloo,
    CPU tensors) inside every step -> W warm-up steps, the fixed SETTLE_STEPS_MULTI_RANK settle steps, K timed steps bracketed
    by barriers -> MAX-reduce of the ranks' durations -> ONE JSON line from rank 0.  The line says "dry_run": true and carries
    no throughput: nothing here measures anything but the launch path."""
    import torch
    from saragan_amd import parallel
    numel = 1 << 18
    param = torch.zeros(numel)
    grad = torch.zeros(numel)
    p_ = torch.nn.Parameter(param)
    p_.grad = grad
    red = parallel.GradientAllReducer(bucket_bytes=256 << 10)

    def step(i):
        grad.fill_(float(rank + 1) * (i + 1))
        red.begin(grad, [(0, numel)], [p_])
        red.finish()                                   # every bucket goes out here (no autograd hooks in the rehearsal)
        param.add_(grad, alpha=-1e-3 * red.grad_scale)

    def barrier():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()

    it = 0
    for _ in range(args.warmup):
        step(it)
        it += 1
    settle = SETTLE_STEPS_MULTI_RANK if world > 1 else 0
    for _ in range(settle):
        step(it)
        it += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(it)
        it += 1
    barrier()
    dt = time.perf_counter() - t0
    same = True
    if torch.distributed.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        lo, hi = param.clone(), param.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
    want = -1e-3 * sum(r + 1 for r in range(world)) / world * sum(range(1, it + 1))     # the averaged updates, in closed form
    ok = same and abs(float(param[0]) - want) <= 1e-4 * abs(want)
    if rank == 0:
        print(json.dumps(dict(dry_run=True, metric='plumbing rehearsal on CPU tensors (gloo): no GPU work, no throughput',
                              value=None, unit=None, n_gpus=world, steps=args.steps, warmup=args.warmup,
                              ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
                              vs_baseline=None, dtype=None, data='synthetic',
                              config=dict(workload='dry run', settle=dict(steps=settle), collective=parallel.collective_info(),
                                          replicas_identical=same, update_matches_closed_form=ok))), flush=True)
    if not ok:
        raise SystemExit('dry run: the ranks disagree after the all-reduced updates')
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
