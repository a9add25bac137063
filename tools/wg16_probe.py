"""Diagnostic: the sliding-halo weight gradient on v_mfma_f32_32x32x16_bf16 against its v_mfma_f32_16x16x32_bf16 form (SG_WGRAD3L_16=1),
A/B in one process, back-to-back launches of the whole call (kernel + finalize).  usage: python tools/wg16_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from saragan_amd import _lib  # noqa: E402
from saragan_amd import functional as F  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
CASES = [(32, 32, 32, (32, 128, 128)), (32, 32, 64, (32, 128, 128)), (32, 64, 64, (16, 64, 64)), (32, 128, 128, (8, 32, 32)),
         (32, 128, 128, (4, 16, 16)), (64, 128, 512, (4, 16, 16))]
for n, cin, cout, sp in CASES:
    x = torch.randn(n, cin, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    gy = torch.randn(n, cout, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    fl = 2.0 * n * sp[0] * sp[1] * sp[2] * cin * cout * 27
    line = f'n{n} {"x".join(map(str, sp)):>11s} {cin:4d}->{cout:4d}'
    outs = {}
    res = {0: [], 1: []}
    for rep in range(3):
        for v in (0, 1):
            os.environ['SG_WGRAD3L_16'] = str(v)
            lib.sg_config_reload()
            for _ in range(3):
                outs[v] = F.raw_wgrad(x, gy, (3, 3, 3), 0.05, want_db=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                F.raw_wgrad(x, gy, (3, 3, 3), 0.05, want_db=True)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 20 * 1e3)
    for v in (0, 1):
        us = min(res[v])
        line += f'   {"16x16x32" if v else "32x32x16"} {us:8.1f} us {fl / us / 1e6:7.1f} TF/s'
    dw0, db0 = outs[0]
    dw1, db1 = outs[1]
    line += f'   max|ddw|/max|dw| {float((dw0 - dw1).abs().max() / dw0.abs().max()):.2e}  db {float((db0 - db1).abs().max() / db0.abs().max()):.2e}'
    print(line, flush=True)
    del x, gy
