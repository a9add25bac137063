"""Diagnostic: from_rgb forward (pointwise 1 -> 32, sign words out) at the benchmarked level.  The grid rule in
launch_pw_fwd (eight trips per block) came from sweeping the block count with this script: at n32, 8192 blocks 4.4 TB/s,
2048 5.1-5.6, 32768 6.1-6.2, 131072 4.0; a block owning CONSECUTIVE voxel groups 5.5-5.9 whatever the count; plain instead of
non-temporal stores 4.0.  usage: python tools/pw_probe.py"""
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
for n in (32, 64):
    d, h, w = 32, 128, 128
    nvox = n * d * h * w
    shp = ConvShape(n, d, h, w, 1, 32, 1, 1, 1, 0)
    x = torch.randn(n, d, h, w, 1, device=dev).bfloat16()
    y = torch.empty(n, d, h, w, 32, device=dev, dtype=torch.bfloat16)
    wt = torch.randn(1, 1, 1, 1, 32, device=dev)
    wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
    _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 1.0, 0, wp.data_ptr(), C.byref(shp), dt, st))
    bias = torch.zeros(32, device=dev)
    sout = torch.empty(n, d, h, w, 1, device=dev, dtype=torch.int32)
    ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
    nbytes = nvox * (2 + 64 + 4)
    ref = None
    for flags, var in ((0, 0), (0, 0)):
        lib.sg_config_reload()
        call = lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        if ref is None:
            ref = (y.clone(), sout.clone())
        same = bool(torch.equal(ref[0], y) and torch.equal(ref[1], sout))
        print(f'n{n} nb {flags:6d} var {var:4d}: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s  identical {same}', flush=True)
    del x, y, sout
