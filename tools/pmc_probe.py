"""Launches each hot conv kernel of the pgan 's' phase-6 step a few times, preceded by a streaming kernel of known
byte count (calibration), for the HBM-traffic counter passes:

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_rd -o rd -- python tools/pmc_probe.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_wr -o wr -- python tools/pmc_probe.py

tools/pmc_summary.py turns the two counter CSVs into profiles/r01_pmc_traffic.json."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib  # noqa: E402
from saragan_amd._lib import ConvEpilogue, ConvShape  # noqa: E402

SHAPES = [  # n, (d,h,w), cin, cout  (all 3x3x3): the >= 1 ms/step entries of `bench.py --dump-prof` at the default
    # per-GPU batch 32; n = 64 is the concatenated real+fake batch of the discriminator, n = 32 the generator /
    # gradient-penalty passes
    (64, (32, 128, 128), 64, 32),
    (64, (32, 128, 128), 32, 32),
    (64, (32, 128, 128), 32, 64),
    (32, (32, 128, 128), 32, 32),
    (32, (32, 128, 128), 32, 64),
    (32, (32, 128, 128), 64, 32),
    (64, (16, 64, 64), 64, 64),
    (64, (16, 64, 64), 64, 128),
    (64, (16, 64, 64), 128, 64),
    (64, (8, 32, 32), 128, 128),
]


def main():
    lib = _lib.load()
    dt = _lib.SG_BF16
    dev = torch.device('cuda:0')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # calibration: out = 1*a + 0*b over 2^28 bf16 elements: reads 2 x 512 MiB, writes 512 MiB, 16 B per lane
    numel = 1 << 28
    a = torch.randn(numel, device=dev).to(torch.bfloat16)
    b = torch.randn(numel, device=dev).to(torch.bfloat16)
    o = torch.empty_like(a)
    for _ in range(2):
        _lib.check(lib.sg_axpby(a.data_ptr(), b.data_ptr(), o.data_ptr(), 1.0, 0.5, numel, dt, st))
    torch.cuda.synchronize()
    del a, b, o
    for n, (d, h, w), cin, cout in SHAPES:
        shp = ConvShape(n, d, h, w, cin, cout, 3, 3, 3, 0)
        x = torch.randn(n, d, h, w, cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(n, d, h, w, cout, device=dev).to(torch.bfloat16)
        wt = torch.randn(3, 3, 3, cin, cout, device=dev)
        wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
        y = torch.empty(n, d, h, w, cout, device=dev, dtype=torch.bfloat16)
        bias = torch.zeros(cout, device=dev)
        nw = (cout + 31) // 32
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, nw), device=dev, dtype=torch.int32)
        sout = torch.empty_like(bits)
        ep_plain = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
        ep_mask = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, bits.data_ptr(), 0.2, None)
        wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        dw = torch.empty(3, 3, 3, cin, cout, device=dev)
        db = torch.empty(cout, device=dev)
        for _ in range(2):
            _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep_plain), dt, st))
            _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep_mask), dt, st))
            _lib.check(lib.sg_conv3d_wgrad_bias(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), 1.0,
                                                ws.data_ptr(), wsb, C.byref(shp), dt, st))
        torch.cuda.synchronize()
        print(f'{d}x{h}x{w} {cin}->{cout} n={n} done', flush=True)


if __name__ == '__main__':
    main()
