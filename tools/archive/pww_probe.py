"""Diagnostic: pointwise weight gradients of the top level (from_rgb 1 -> 32 with its data gradient, to_rgb 32 -> 1).
usage: python tools/pww_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from saragan_amd import _lib  # noqa: E402
from saragan_amd._lib import ConvShape  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
dt = _lib.SG_BF16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(call):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


d, h, w = 32, 128, 128
for n, cin, cout, with_dx in ((64, 1, 32, True), (32, 1, 32, False), (32, 32, 1, False)):
    nvox = n * d * h * w
    shp = ConvShape(n, d, h, w, cin, cout, 1, 1, 1, 0)
    x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
    dy = torch.randn(n, d, h, w, cout, device=dev).bfloat16()
    dw = torch.empty(1, 1, 1, cin, cout, device=dev)
    db = torch.empty(cout, device=dev)
    wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    if with_dx:
        wm = torch.randn(cin, cout, device=dev)
        dx = torch.empty(n, d, h, w, cin, device=dev, dtype=torch.bfloat16)
        call = lambda: _lib.check(lib.sg_conv3d_pw_bwd(x.data_ptr(), dy.data_ptr(), wm.data_ptr(), dw.data_ptr(), db.data_ptr(), dx.data_ptr(),
                                                       0.5, ws.data_ptr(), wsb, C.byref(shp), dt, st))
        nbytes = nvox * (cin + cout) * 2 + nvox * cin * 2
    else:
        call = lambda: _lib.check(lib.sg_conv3d_wgrad_bias(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), 0.5, ws.data_ptr(), wsb,
                                                           C.byref(shp), dt, st))
        nbytes = nvox * (cin + cout) * 2
    us = timeit(call)
    print(f'n{n} {cin}->{cout} dx={int(with_dx)}: {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s  '
          f'dw {float(dw.double().abs().sum()):.6e} db {float(db.double().abs().sum()):.6e}', flush=True)
    del x, dy, ws
