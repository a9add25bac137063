"""End-to-end check of a PROGRESSIVE run with the round-4 features on (diagnostic, GPU): pgan 'xs', phases 1-4 (4x4x1 ... 32x32x8),
mixing + stabilising images per phase as given, validation metrics in the loop, checkpoints, the captured step chosen automatically.
Prints per phase: images/s, steps, how many step graphs were captured and for which keys, device memory, last losses.
usage: python tools/progressive_run_check.py [images_per_half_phase=2048] [base_batch=32]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tmp = tempfile.mkdtemp(prefix='saragan_prog_')
for ph in range(1, 5):
    xy = 4 * 2 ** (ph - 1)
    d = os.path.join(tmp, 'data', f'{xy}x{xy}')
    os.makedirs(d)
    for i in range(64):
        rng = np.random.default_rng(1234 + i)
        np.save(os.path.join(d, f'{i:04d}.npy'), np.clip(rng.normal(1024, 512, (xy // 4, xy, xy)), 0, 4095).astype(np.int16))
from saragan_amd.main import build_parser, finalize_args  # noqa: E402
from saragan_amd import optimization as opt  # noqa: E402
from saragan_amd.train import run_training  # noqa: E402

argv = ['pgan', os.path.join(tmp, 'data') + '/', '--start_shape', '(1, 1, 4, 4)', '--final_shape', '(1, 16, 64, 64)',
        '--starting_phase', '1', '--ending_phase', '4', '--base_batch_size', str(bb), '--latent_dim', '256', '--network_size', 'xs',
        '--noise_stddev', '0.01', '--mixing_nimg', str(nimg), '--stabilizing_nimg', str(nimg), '--loss_fn', 'wgan', '--gp_weight', '10',
        '--data_mean', '1024', '--data_stddev', '1024', '--logdir', os.path.join(tmp, 'log'), '--g_lr', '1e-3', '--d_lr', '1e-3',
        '--g_lr_increase', 'linear', '--g_lr_decrease', 'exponential', '--checkpoint_every_nsteps', '100000000',
        '--calc_metrics', '--compute_psnrs', '--compute_mses', '--compute_swds', '--metrics_every_nsteps', str(nimg), '--num_metric_samples', '8',
        '--metrics_batch_size', '8']
args, unknown = build_parser().parse_known_args(argv)
args = finalize_args(args)
graphs = []
real_init = opt.StepGraph.__init__


def spy(self, *a, **kw):
    real_init(self, *a, **kw)
    graphs.append(self)


opt.StepGraph.__init__ = spy
t0 = time.time()
out = run_training(args, log_every=10 ** 9)
torch.cuda.synchronize()
print(f'\n== progressive run: {time.time() - t0:.1f} s wall, SARAGAN_HIPGRAPH={os.environ.get("SARAGAN_HIPGRAPH", "(unset: auto)")}')
for ph, st in out['stats'].items():
    g = graphs[ph - 1]
    caps = g.__dict__.get('_captures', {})
    keys = [(k[3], 'captured' if 'graph' in e else e.get('decided', f'eager x{e["eager"]}'), [round(r, 2) for r in e.get('ratios', [])]) for k, e in caps.items()]
    print(f'phase {ph}: batch {st["batch_size"]}, {st["steps"]} steps, {st["img_s"]:.0f} img/s (incl. metrics and checkpoint), d_loss {st["d_loss"]}, '
          f'step graphs by alpha class: {keys}')
    for k in ('metrics_validation', 'metrics_test'):
        if k in st:
            print(f'    {k}: { {m: (np.round(v, 4).tolist() if hasattr(v, "tolist") else round(float(v), 4)) for m, v in st[k].items()} }')
print(f'device memory: allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB, peak {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB, '
      f'reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB')
for k, v in out['store'].vars.items():
    assert torch.isfinite(v).all(), k
print('all weights finite')
