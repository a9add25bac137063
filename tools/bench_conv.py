"""Micro-benchmark of the conv kernels through the C ABI on the layer shapes of pgan 's' phase 6.
usage: python tools/bench_conv.py [--dtype bf16] [--batch 8] [--which fwd,wgrad] [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib  # noqa: E402
from saragan_amd._lib import ConvEpilogue, ConvShape  # noqa: E402

SHAPES = [  # (d,h,w), cin, cout, k
    ((32, 128, 128), 32, 32, (3, 3, 3)),
    ((32, 128, 128), 32, 64, (3, 3, 3)),
    ((32, 128, 128), 64, 32, (3, 3, 3)),
    ((16, 64, 64), 64, 64, (3, 3, 3)),
    ((16, 64, 64), 64, 128, (3, 3, 3)),
    ((16, 64, 64), 128, 64, (3, 3, 3)),
    ((8, 32, 32), 128, 128, (3, 3, 3)),
    ((4, 16, 16), 128, 512, (3, 3, 3)),
    ((2, 8, 8), 512, 512, (1, 3, 3)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--which', default='fwd,wgrad')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--shapes', default='')
    ap.add_argument('--mask', action='store_true', help='fwd: data-gradient style epilogue (no bias/act, mask_bits)')
    ap.add_argument('--signs', action='store_true', help='fwd: also write the sign words of the output')
    a = ap.parse_args()
    lib = _lib.load()
    dt = _lib.SG_BF16 if a.dtype == 'bf16' else _lib.SG_F32
    tdt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    dev = torch.device('cuda:0')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sel = [int(s) for s in a.shapes.split(',')] if a.shapes else range(len(SHAPES))
    for si in sel:
        (d, h, w), cin, cout, k = SHAPES[si]
        n = a.batch
        shp = ConvShape(n, d, h, w, cin, cout, k[0], k[1], k[2], 0)
        x = torch.randn(n, d, h, w, cin, device=dev).to(tdt)
        dy = torch.randn(n, d, h, w, cout, device=dev).to(tdt)
        wt = torch.randn(*k, cin, cout, device=dev)
        wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
        y = torch.empty(n, d, h, w, cout, device=dev, dtype=tdt)
        bias = torch.zeros(cout, device=dev)
        nw = (cout + 31) // 32
        mbits = torch.randint(-2**31, 2**31 - 1, (n, d, h, w, nw), device=dev, dtype=torch.int32)
        sout = torch.empty((n, d, h, w, nw), device=dev, dtype=torch.int32)
        if a.mask:
            ep = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, mbits.data_ptr(), 0.2, None)
        else:
            ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr() if a.signs else None)
        flops = 2.0 * n * d * h * w * cin * cout * k[0] * k[1] * k[2]
        res = {}
        if 'fwd' in a.which:
            def f():
                _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
            res['fwd'] = timeit(f, a.iters)
        if 'wgrad' in a.which:
            wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
            ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
            dw = torch.empty(*k, cin, cout, device=dev)
            def g():
                _lib.check(lib.sg_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1.0, ws.data_ptr(), wsb, C.byref(shp), dt, st))
            res['wgrad'] = timeit(g, a.iters)
        if 'wgb' in a.which:
            wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
            ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
            dw = torch.empty(*k, cin, cout, device=dev)
            dbias = torch.empty(cout, device=dev)
            def gb():
                _lib.check(lib.sg_conv3d_wgrad_bias(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), dbias.data_ptr(), 1.0, ws.data_ptr(), wsb, C.byref(shp), dt, st))
            res['wgb'] = timeit(gb, a.iters)
        line = f'{si}: {d}x{h}x{w} {cin:4d}->{cout:4d} k{k} n={n}'
        for kk, ms in res.items():
            line += f' | {kk} {ms:8.3f} ms {flops / ms / 1e9:8.1f} TF/s'
        print(line, flush=True)


def timeit(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if __name__ == '__main__':
    main()
