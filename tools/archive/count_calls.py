"""Which Python call sites launch the elementwise kernels of one bench step, and how often?  Wraps the ctypes entry
points with a counter keyed by (symbol, element count, three innermost saragan_amd frames).  Diagnostic."""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from saragan_amd import _lib   # noqa: E402

lib = _lib.load()
counts = collections.Counter()
WATCH = ['sg_bias_act_bwd', 'sg_bias_act_bwd_bits', 'sg_axpby', 'sg_upscale_nn', 'sg_upscale2x', 'sg_upscale2x_masked', 'sg_downscale_sum',
         'sg_downscale2x', 'sg_pixel_norm_fwd', 'sg_pixel_norm_bwd', 'sg_pixel_norm_act_bwd', 'sg_bias_act_fwd', 'sg_sign_words']


class Proxy:
    def __init__(self, real):
        object.__setattr__(self, '_real', real)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name not in WATCH or not ARMED[0]:
            return fn

        def wrapped(*a):
            fr = [f for f in traceback.extract_stack()[:-1] if 'saragan_amd' in f.filename][-4:]
            site = ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in reversed(fr))
            size = ''
            if name.startswith('sg_bias_act_bwd'):
                ints = [int(getattr(v, 'value', v)) for v in a[5:7]] if name.endswith('bits') else [int(getattr(v, 'value', v)) for v in a[4:6]]
                size = f' nvox {ints[0]} c {ints[1]}'
            counts[(name, site + size)] += 1
            return fn(*a)
        return wrapped


ARMED = [False]
_lib._lib = Proxy(lib)
sys.argv = [sys.argv[0], '--batch', '4', '--no-extras', '--no-cpu-baseline']
args = bench.parse()
dev = torch.device('cuda:0')
cfg = bench.build(args, dev, 'bf16')
batches = [bench.synthetic_batch(cfg['shape'], i, dev) for i in range(2)]


def step(i):
    cfg['sess'].run(cfg['train'], feed_dict={cfg['ph']: batches[i % 2]})
    cfg['sess'].run(cfg['ema_op'])


step(0)
step(1)
ARMED[0] = True
step(2)
torch.cuda.synchronize()
for (name, site), c in sorted(counts.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f'{c:3d} {name:24s} {site}')
