"""Consistency of the fast kernels on tensors beyond 2 GiB (64 x 32x128x128 x 64 channels bf16 = 4.3 GB): the
persistent kernels against the library's generic kernels on the same inputs (env switches are read per call)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib  # noqa: E402
from saragan_amd._lib import ConvEpilogue, ConvShape  # noqa: E402


def main():
    lib = _lib.load()
    dt = _lib.SG_BF16
    dev = torch.device('cuda:0')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, (d, h, w) = 64, (32, 128, 128)
    for cin, cout in ((64, 32), (32, 64)):
        shp = ConvShape(n, d, h, w, cin, cout, 3, 3, 3, 0)
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(n, d, h, w, cin, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
        dy = torch.randn(n, d, h, w, cout, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
        wt = torch.randn(3, 3, 3, cin, cout, device=dev, generator=g)
        wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
        bias = torch.randn(cout, device=dev)
        ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, None)
        outs = []
        for env in ({}, {'SG_FWD_NO_V3': '1', 'SG_FWD_NO_V4': '1'}):
            for k in ('SG_FWD_NO_V3', 'SG_FWD_NO_V4'):
                os.environ.pop(k, None)
            os.environ.update(env)
            lib.sg_config_reload()
            y = torch.empty(n, d, h, w, cout, device=dev, dtype=torch.bfloat16)
            _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
            torch.cuda.synchronize()
            outs.append(y)
        for k in ('SG_FWD_NO_V3', 'SG_FWD_NO_V4'):
            os.environ.pop(k, None)
        # same f32 accumulation order is not guaranteed between kernels: compare at bf16 resolution, sample by sample
        worst = max(float((outs[0][i].float() - outs[1][i].float()).abs().max()) for i in range(0, n, 7))
        scale = float(outs[1][0].float().abs().max())
        print(f'fwd {cin}->{cout}: max |persistent - generic| = {worst:.4g} (scale {scale:.3g})', flush=True)
        assert worst <= 2e-2 * scale
        del outs
        wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        dws = []
        for env in ({}, {'SG_WGRAD_V1': '1'}):
            os.environ.pop('SG_WGRAD_V1', None)
            os.environ.update(env)
            lib.sg_config_reload()
            dw = torch.empty(3, 3, 3, cin, cout, device=dev)
            db = torch.empty(cout, device=dev)
            _lib.check(lib.sg_conv3d_wgrad_bias(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), 1.0,
                                                ws.data_ptr(), wsb, C.byref(shp), dt, st))
            torch.cuda.synchronize()
            dws.append((dw, db))
        os.environ.pop('SG_WGRAD_V1', None)
        e = float((dws[0][0] - dws[1][0]).abs().max()) / float(dws[1][0].abs().max())
        eb = float((dws[0][1] - dws[1][1]).abs().max()) / float(dws[1][1].abs().max())
        print(f'wgrad {cin}->{cout}: rel max diff dw {e:.3g}, dbias {eb:.3g}', flush=True)
        assert e < 2e-3 and eb < 2e-3
        del x, dy, dws
    print('BIG OK')


if __name__ == '__main__':
    main()
