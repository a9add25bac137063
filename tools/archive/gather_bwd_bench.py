"""Backward of downscale3d(leaky_relu(conv3d(x) + b)) for the 32 -> 64 layer at 128^2 (pgan/discriminator.py:39-44):
the materialised two-tensor path (_pooled_backward_planes) against the fused gather (_pooled_backward_gather), HIP-event
timed, plus a bit comparison of the data gradient.  Diagnostic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import functional as F   # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for n in (int(v) for v in os.environ.get('GB_N', '32,64').split(',')):
        d, h, w = 32, 128, 128
        x = torch.randn(n, 32, d, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        wt = torch.randn(3, 3, 3, 32, 64, device=dev) * 0.05
        gy = torch.randn(n, 64, d // 2, h // 2, w // 2, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        act = torch.randn(n, 64, d, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        signs = F.sign_words(act)
        del act
        info = F.ActInfo(0.2)
        info.bits = F.sign_words(torch.randn(n, 32, d, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d))
        info.consume(True)
        with torch.no_grad():
            tp, rp = timed(lambda: F._pooled_backward_planes(gy, x, wt, signs, 0.05, 0.2, info, True, True, True))
            tg, rg = timed(lambda: F._pooled_backward_gather(gy, x, wt, signs, 0.05, 0.2, info, True, True, True))
            tpx, _ = timed(lambda: F._pooled_backward_planes(gy, x, wt, signs, 0.05, 0.2, info, True, False, False))
            tgx, _ = timed(lambda: F._pooled_backward_gather(gy, x, wt, signs, 0.05, 0.2, info, True, False, False))
        same = torch.equal(rp[0], rg[0])
        ew = float((rp[1] - rg[1]).abs().max() / rp[1].abs().max())
        eb = float((rp[2] - rg[2]).abs().max() / rp[2].abs().max())
        print(f'n{n}: planes {tp:.3f} ms (gx only {tpx:.3f}), gather {tg:.3f} ms (gx only {tgx:.3f}); gx bit-equal {same}, '
              f'dw rel {ew:.2e}, db rel {eb:.2e}', flush=True)


if __name__ == '__main__':
    main()
