"""bench.py --dry-run: the N-rank launch path -- self-spawned ranks and the torch.distributed.run form the
driver uses, rendezvous on 127.0.0.1, the bucketed all-reduce of parallel.GradientAllReducer in every step, warm-up, the fixed
settle steps, the timed steps between barriers, the MAX-reduce of the durations, ONE JSON line from rank 0 -- on CPU tensors over
gloo, so that the first real 8-GPU run cannot fail on plumbing."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_self_spawned_ranks_rehearse_the_launch_path():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--dry-run', '--steps', '3', '--warmup', '2'],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _line(r.stdout)
    assert rec['dry_run'] is True and rec['value'] is None and rec['n_gpus'] == 4 and rec['steps'] == 3 and rec['warmup'] == 2
    assert rec['config']['settle'] == {'steps': 60} and rec['config']['collective']['backend'] == 'gloo'
    assert rec['config']['collective']['world_size'] == 4
    assert rec['config']['replicas_identical'] and rec['config']['update_matches_closed_form']


def test_the_drivers_launcher_form():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '2',
                        '--warmup', '1'], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _line(r.stdout)
    assert rec['dry_run'] is True and rec['n_gpus'] == 2 and rec['config']['replicas_identical']
