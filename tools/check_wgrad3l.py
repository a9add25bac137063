"""The lean sliding-halo weight-gradient kernel against the general one (SG_WGRAD_NO_LEAN=1) and torch fp32 on shapes
that engage it (>= 2 tile columns per block): ragged H, odd D, bias gradient.  Diagnostic."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib, functional as F   # noqa: E402

lib = None
dev = None


def run(no_lean, fn):
    os.environ['SG_WGRAD_NO_LEAN'] = '1' if no_lean else '0'
    lib.sg_config_reload()
    lib.sg_prof_enable(1)
    out = fn()
    torch.cuda.synchronize()
    ents = (_lib.ProfEntry * 64)()
    cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 64, C.byref(cnt))
    lib.sg_prof_enable(0)
    return out, sorted({ents[i].kernel.decode() for i in range(cnt.value)})


def main():
    global lib, dev
    lib = _lib.load()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    ok = True
    for (n, cin, cout, sp, ups) in [(2, 32, 64, (6, 128, 256), 0), (3, 64, 32, (5, 126, 256), 0), (2, 32, 32, (4, 128, 512), 0),
                                    (1, 64, 64, (8, 128, 256), 0), (2, 64, 32, (8, 128, 256), 1), (3, 32, 32, (4, 124, 256), 1)]:
        xsp = tuple(t // 2 for t in sp) if ups else sp
        x = torch.randn(n, cin, *xsp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        dy = torch.randn(n, cout, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        # the <ups> variant serves the shapes the sub-pixel weight gradient declines; test it on its own here
        keep, F._NO_SUBPIXEL = F._NO_SUBPIXEL, True
        try:
            (gw, gb), kg = run(False, lambda: F.raw_wgrad(x, dy, (3, 3, 3), 0.05, bool(ups), True))
            (rw, rb), kr = run(True, lambda: F.raw_wgrad(x, dy, (3, 3, 3), 0.05, bool(ups), True))
        finally:
            F._NO_SUBPIXEL = keep
        # torch fp32: dw[kd,kh,kw,ci,co] = sum_v x[v+tap,ci] dy[v,co]
        xt = x.float()
        if ups:
            xt = xt.repeat_interleave(2, 2).repeat_interleave(2, 3).repeat_interleave(2, 4)
        wt = torch.zeros(cout, cin, 3, 3, 3, device=dev, requires_grad=True)
        y = torch.nn.functional.conv3d(xt, wt, padding=1)
        (tw,) = torch.autograd.grad(y, wt, dy.float())
        tw = tw.permute(2, 3, 4, 1, 0) * 0.05
        tb = dy.float().sum(dim=(0, 2, 3, 4))
        sc = float(tw.abs().max())
        e_lean, e_gen = float((gw - tw).abs().max()) / sc, float((rw - tw).abs().max()) / sc
        eb = float((gb - tb).abs().max() / tb.abs().max())
        print(f'n{n} {cin}->{cout} {sp} ups{ups}: {kg} vs {kr}: lean err {e_lean:.2e}, general err {e_gen:.2e}, db err {eb:.2e}')
        if not any('wgrad3l' in k for k in kg) or e_lean > 5e-3 or eb > 5e-3 or not bool(torch.isfinite(gw).all()):
            ok = False
            print('   ** MISMATCH / lean kernel not engaged')
    os.environ['SG_WGRAD_NO_LEAN'] = '0'
    lib.sg_config_reload()
    print('OK' if ok else 'FAILED')
    return ok


if __name__ == '__main__':
    sys.exit(0 if main() else 1)
