"""Diagnostic: static s_setprio 1 for the younger wave group (waves 4-7) of the ping-pong kernels (SG_DBG_FLAGS=2048;
MI355X_MICROARCH.md 'Two waves per SIMD' item 4) -- conv_fwd3s 32 -> 32 / 32 -> 64 and the one-pass 64 -> 32 at the bench shapes,
A/B in one process.  usage: python tools/prio_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from saragan_amd import _lib  # noqa: E402
from saragan_amd._lib import ConvEpilogue, ConvShape  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
dt = _lib.SG_BF16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n, d, h, w = 32, 32, 128, 128


def bench(cin, cout, mode):
    shp = ConvShape(n, d, h, w, cin, cout, 3, 3, 3, 0)
    x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
    wt = torch.randn(3, 3, 3, cin, cout, device=dev)
    wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
    _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
    y = torch.empty(n, d, h, w, cout, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(cout, device=dev)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, (cout + 31) // 32), device=dev, dtype=torch.int32)
    sout = torch.empty_like(bits)
    ep = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, bits.data_ptr(), 0.2, None) if mode == 'mask' else \
        ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
    ws = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
    wsb = torch.empty(max(16, ws), device=dev, dtype=torch.uint8)
    if ws:
        ep.workspace, ep.workspace_bytes = wsb.data_ptr(), ws
    call = lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
    out = []
    for rep in range(3):
        for flags in (0, 2048):
            os.environ['SG_DBG_FLAGS'] = str(flags)
            lib.sg_config_reload()
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            out.append((flags, e0.elapsed_time(e1) / 20 * 1e3))
    fl = 2.0 * n * d * h * w * cin * cout * 27
    base = min(t for f, t in out if f == 0)
    prio = min(t for f, t in out if f == 2048)
    print(f'{cin}->{cout} {mode}: default {base:8.1f} us ({fl / base / 1e6:6.0f} TF/s)   setprio(1) on waves 4-7 {prio:8.1f} us ({fl / prio / 1e6:6.0f} TF/s)   '
          f'{(base / prio - 1) * 100:+.2f} %   all: {[round(t) for _, t in out]}', flush=True)


for cin, cout in ((32, 32), (32, 64), (64, 32)):
    for mode in ('signs', 'mask'):
        bench(cin, cout, mode)
os.environ['SG_DBG_FLAGS'] = '0'
lib.sg_config_reload()
