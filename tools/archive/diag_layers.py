"""Diagnostic for a conv layer at an exact shape: each stage of forward / backward against the CPU oracle on a box,
with the locations of the mismatches.  usage: python tools/diag_layers.py <cin> <cout> <d> <h> <w> <dtype> [n]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pgan_oracle as O  # noqa: E402
from saragan_amd import functional as F  # noqa: E402


def where(err, tol, name):
    bad = (err > tol).nonzero()
    print(f'  {name}: max err {float(err.max()):.4g}, {len(bad)} / {err.numel()} above {tol:.3g}')
    if len(bad):
        b = bad.numpy()
        for ax, nm in enumerate(['n', 'c', 'd', 'h', 'w'][:b.shape[1]]):
            vals, cnt = np.unique(b[:, ax], return_counts=True)
            print(f'    axis {nm}: {dict(list(zip(vals.tolist(), cnt.tolist()))[:40])}')


def main():
    cin, cout, d, h, w = (int(v) for v in sys.argv[1:6])
    dtype = torch.bfloat16 if sys.argv[6] == 'bf16' else torch.float32
    n = int(sys.argv[7]) if len(sys.argv) > 7 else 2
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    sp = (d, h, w)
    x = torch.randn((n, cin, *sp), generator=g).to(dtype)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g)
    coef = O.runtime_coef(wt.shape, 'leaky_relu', 0.2)
    wq = (wt * coef).to(dtype).double()
    lo = (max(0, d // 2 - 3), max(0, h // 2 - 5), max(0, w // 2 - 20))
    hi = (lo[0] + 6, lo[1] + 10, lo[2] + 40)
    gy = torch.zeros((n, cout, *sp), dtype=dtype)
    gbox = torch.randn((n, cout, 6, 10, 40), generator=g).to(dtype)
    gy[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = gbox
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last_3d)
    xd, gyd, wd = cl(x), cl(gy), wt.to(dev)
    a0, b0, c0, a1, b1, c1 = lo[0] - 1, lo[1] - 1, lo[2] - 1, hi[0] + 1, hi[1] + 1, hi[2] + 1
    rt = 1e-4 if dtype == torch.float32 else 1e-2
    # 1. plain data-gradient conv of gy (no mask): dx = conv(gy, flip(w))
    dx, _, _ = F.raw_conv(gyd, wd, coef, True)
    xs = x.double()[:, :, a0:a1, b0:b1, c0:c1].clone().requires_grad_(True)
    yr = TF.conv3d(TF.pad(xs, (1, 1, 1, 1, 1, 1)), wq.permute(4, 3, 0, 1, 2))
    yb = yr[:, :, 1:-1, 1:-1, 1:-1]
    (gxr,) = torch.autograd.grad(yb, xs, gbox.double())
    got = dx.double().cpu()[:, :, a0:a1, b0:b1, c0:c1]
    print(f'dgrad {cout}->{cin} (transpose_flip) n={n} {sp} {sys.argv[6]}')
    where((got - gxr).abs(), rt * float(gxr.abs().max()), 'dgrad vs oracle')
    out = dx.double().cpu().clone()
    out[:, :, a0:a1, b0:b1, c0:c1] = 0
    print('  leak outside box+halo:', float(out.abs().max()))
    # 1b. same through the library's generic kernels
    lib = F._lib.load()
    for env in ({'SG_FWD_NO_V4': '1'}, {'SG_FWD_NO_V4': '1', 'SG_FWD_NO_V3': '1'}):
        os.environ.update(env)
        lib.sg_config_reload()
        F.clear_pack_cache()
        dx2, _, _ = F.raw_conv(gyd, wd, coef, True)
        print(f'  {env}: max |diff| vs default path {float((dx2.double() - dx.double()).abs().max()):.4g}')
        got2 = dx2.double().cpu()[:, :, a0:a1, b0:b1, c0:c1]
        where((got2 - gxr).abs(), rt * float(gxr.abs().max()), '  this path vs oracle')
    for k in ('SG_FWD_NO_V4', 'SG_FWD_NO_V3'):
        os.environ.pop(k, None)
    lib.sg_config_reload()
    # 2. forward
    y, _, _ = F.raw_conv(xd, wd, coef, False)
    ref = TF.conv3d(x.double()[:, :, a0:a1, b0:b1, c0:c1], wq.permute(4, 3, 0, 1, 2))     # valid conv: the box
    got = y.double().cpu()[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    where((got - ref).abs(), rt * float(ref.abs().max()), f'fwd {cin}->{cout} vs oracle')
    # 2b. fused bias + LeakyReLU forward with sign words, then the mask stage of its backward
    b = torch.randn(cout, generator=g) * 0.1
    bd = b.to(dev)
    y2, _, signs = F.raw_conv(xd, wd, coef, False, bias=bd, act=True, slope=0.2, want_signs=True)
    want = F.sign_words(y2)
    diff = (signs ^ want).to(torch.int64) & 0xFFFFFFFF
    nflip = int(sum(((diff >> k) & 1).sum() for k in range(32)))
    print(f'  sign words from the conv epilogue vs sg_sign_words(y): {nflip} differing bits of {signs.numel() * 32}')
    neg = (y2 < 0)
    bits = torch.stack([((want.to(torch.int64) >> k) & 1) for k in range(32)], dim=-1)      # [n,d,h,w,nw,32]
    bits = bits.reshape(*want.shape[:4], -1)[..., :cout].permute(0, 4, 1, 2, 3).bool()
    print('  sg_sign_words vs (y < 0):', int((bits != neg).sum()), 'differing')
    gm, _ = F.raw_bias_act_bwd(gyd, signs, 0.2, want_dx=True, want_db=False)
    exp = torch.where(neg, gyd.float() * 0.2, gyd.float())
    print('  bias_act_bwd(bits) vs where(y<0): max diff', float((gm.float() - exp).abs().max()))
    gm2, _ = F.raw_bias_act_bwd(gyd, y2, 0.2, want_dx=True, want_db=False)
    print('  bias_act_bwd(y) vs where(y<0): max diff', float((gm2.float() - exp).abs().max()))
    yref = ref + b.double().reshape(1, -1, 1, 1, 1)
    yref = torch.maximum(yref, 0.2 * yref)
    got = y2.double().cpu()[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    where((got - yref).abs(), rt * float(yref.abs().max()), 'fwd + bias + lrelu vs oracle')
    print('  sign agreement with oracle in the box:', int(((got < 0) != (yref < 0)).sum()), 'differ;',
          'min |y| oracle', float(yref.abs().min()))
    # 3. weight gradient with the sparse dy
    dw, db = F.raw_wgrad(xd, gyd, (3, 3, 3), 1.0, want_db=True)
    wr = torch.zeros((cout, cin, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    xin = x.double()[:, :, a0:a1, b0:b1, c0:c1]
    yr = TF.conv3d(xin, wr)          # valid conv over box+halo = the box
    (gw,) = torch.autograd.grad(yr, wr, gbox.double())
    refw = gw.permute(2, 3, 4, 1, 0)
    err = (dw.double().cpu() - refw).abs()
    rtw = 1e-4 if dtype == torch.float32 else 2e-3
    bad = (err > rtw * float(refw.abs().max())).nonzero()
    print(f'wgrad: max err {float(err.max()):.4g} of {float(refw.abs().max()):.4g}; {len(bad)} / {err.numel()} bad')
    if len(bad):
        b = bad.numpy()
        for ax, nm in enumerate(['kd', 'kh', 'kw', 'ci', 'co']):
            vals, cnt = np.unique(b[:, ax], return_counts=True)
            print(f'    axis {nm}: {dict(list(zip(vals.tolist(), cnt.tolist()))[:70])}')
    os.environ['SG_WGRAD_NO_V3'] = '1'
    lib.sg_config_reload()
    dw2, _ = F.raw_wgrad(xd, gyd, (3, 3, 3), 1.0, want_db=True)
    print('  wgrad without v3: max err', float((dw2.double().cpu() - refw).abs().max()))
    os.environ['SG_WGRAD_V1'] = '1'
    lib.sg_config_reload()
    dw3, _ = F.raw_wgrad(xd, gyd, (3, 3, 3), 1.0, want_db=True)
    print('  wgrad v1: max err', float((dw3.double().cpu() - refw).abs().max()))
    # dense dy for comparison between kernels
    os.environ.pop('SG_WGRAD_V1'); os.environ.pop('SG_WGRAD_NO_V3')
    lib.sg_config_reload()
    gyf = cl(torch.randn((n, cout, *sp), generator=g).to(dtype))
    dwa, _ = F.raw_wgrad(xd, gyf, (3, 3, 3), 1.0, want_db=True)
    os.environ['SG_WGRAD_V1'] = '1'
    lib.sg_config_reload()
    dwb, _ = F.raw_wgrad(xd, gyf, (3, 3, 3), 1.0, want_db=True)
    print('  dense dy: fast vs v1 wgrad max diff', float((dwa - dwb).abs().max()), 'of', float(dwb.abs().max()))


if __name__ == '__main__':
    main()
