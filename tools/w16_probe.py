"""Diagnostic: the weight gradient of the 16-wide levels (4 x 16 x 16) -- conv_wgrad3l<w16> against the tap-per-wave kernel it
replaced (SG_WGRAD_NO_W16=1), whole call (memset + kernel + finalize) by HIP events; run under `rocprofv3 --kernel-trace --stats`
for the split.  usage: python tools/w16_probe.py [case indices]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from saragan_amd import _lib  # noqa: E402
from saragan_amd import functional as F  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
CASES = [(32, 128, 128), (64, 128, 128), (32, 128, 512), (64, 128, 512), (32, 64, 64), (64, 64, 64), (32, 64, 256)]
if len(sys.argv) > 1:
    CASES = [CASES[int(i)] for i in sys.argv[1:]]
for n, cin, cout in CASES:
    x = torch.randn(n, cin, 4, 16, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    gy = torch.randn(n, cout, 4, 16, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    fl = 2.0 * n * 4 * 16 * 16 * cin * cout * 27
    line = f'n{n} {cin:4d}->{cout:4d}'
    for v in (0, 1):
        os.environ['SG_WGRAD_NO_W16'] = str(v)
        lib.sg_config_reload()
        for _ in range(3):
            F.raw_wgrad(x, gy, (3, 3, 3), 0.05, want_db=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            F.raw_wgrad(x, gy, (3, 3, 3), 0.05, want_db=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        line += f'   {"w16" if v == 0 else "wgrad2"} {us:7.1f} us {fl / us / 1e6:7.1f} TF/s'
    print(line, flush=True)
