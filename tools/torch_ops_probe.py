"""Diagnostic: which torch-native (non-library) GPU kernels one EAGER training step launches, by the Python source line that asked.
usage: python tools/torch_ops_probe.py [bench config, default 2]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SARAGAN_HIPGRAPH'] = '0'
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

cfg_id = sys.argv[1] if len(sys.argv) > 1 else '2'
sys.argv = ['bench.py', '--config', cfg_id, '--no-cpu-baseline', '--no-extras']
args = bench.parse()
device = torch.device('cuda:0')
cfg = bench.build(args, device, args.dtype)
sess, ph = cfg['sess'], cfg['ph']
batch = bench.synthetic_batch(cfg['shape'], 0, device)


def step():
    sess.run(cfg['train'], feed_dict={ph: batch})
    sess.run(cfg['ema_op'])


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and getattr(ev, 'kernels', None):
        st = ev.stack or []
        here = next((s for s in st if 'saragan_amd' in s or 'bench.py' in s), st[0] if st else '?')
        by[(ev.name, here.strip()[:150])] += 1
for (name, here), n in by.most_common(60):
    print(f'{n:4d}  {name:28s} {here}')
print('aten ops that launched kernels:', sum(by.values()))
