"""Diagnostic for the one-pass 64 -> 32 kernel (csrc/conv3p.hip, conv_fwd3p): correctness against the two-pass K split
(SG_FWD_NO_3P=1) and a torch fp32 convolution, bit-equality of the fused masked gather against the materialised masked
up-scale through the same kernel, timings of both kernels, and the in-kernel phase stamps.
usage: python tools/p3_probe.py [n] [check|time|stamps ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from saragan_amd import _lib  # noqa: E402
from saragan_amd._lib import ConvEpilogue, ConvShape  # noqa: E402

lib = _lib.load()
lib.sg_debug_set_ts_buffer.argtypes = [C.c_void_p]
dev = torch.device('cuda:0')
dt = _lib.SG_BF16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
what = [a for a in sys.argv[1:] if not a.isdigit()] or ['check', 'time', 'stamps']


def set_env(**kw):
    for k, v in kw.items():
        os.environ[k] = str(v)
    lib.sg_config_reload()


def kernel_of(call):
    lib.sg_prof_enable(1)
    call()
    torch.cuda.synchronize()
    ents = (_lib.ProfEntry * 16)()
    cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 16, C.byref(cnt))
    lib.sg_prof_enable(0)
    return [ents[i].kernel.decode() for i in range(cnt.value)]


def make(n, d, h, w, ups):
    vox = n * d * h * w
    g = torch.Generator(device=dev).manual_seed(5)
    if ups:
        x = torch.randn(n, d // 2, h // 2, w // 2, 64, device=dev, generator=g).to(torch.bfloat16)
    else:
        x = torch.randn(n, d, h, w, 64, device=dev, generator=g).to(torch.bfloat16)
    bits64 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, 2), device=dev, dtype=torch.int32, generator=g)
    bits32 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, 1), device=dev, dtype=torch.int32, generator=g)
    wt = torch.randn(3, 3, 3, 32, 64, device=dev, generator=g)
    shp = ConvShape(n, d, h, w, 64, 32, 3, 3, 3, 1 if ups else 0)
    wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
    _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 1, wp.data_ptr(), C.byref(shp), dt, st))
    fws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
    fws = torch.empty(max(16, fws_bytes), device=dev, dtype=torch.uint8)
    return dict(x=x, bits64=bits64, bits32=bits32, wt=wt, shp=shp, wp=wp, fws=fws, fws_bytes=fws_bytes, vox=vox, dims=(n, d, h, w))


def launch(m, y, masked_in, masked_out):
    ep = ConvEpilogue(None, 0, 0.0, 0, 1e-8, None, m['bits32'].data_ptr() if masked_out else None, 0.2, None)
    ep.workspace, ep.workspace_bytes = m['fws'].data_ptr(), m['fws_bytes']
    if masked_in:
        ep.in_mask_bits, ep.in_mask_slope, ep.in_gain = m['bits64'].data_ptr(), 0.2, 0.125
    _lib.check(lib.sg_conv3d_fwd(m['x'].data_ptr(), m['wp'].data_ptr(), y.data_ptr(), C.byref(m['shp']), C.byref(ep), dt, st))


def ulp_report(tag, a, b):
    a32, b32 = a.float(), b.float()
    diff = (a32 - b32).abs()
    scale = a32.abs().max()
    ne = int((a.view(torch.int16) != b.view(torch.int16)).sum())
    print(f'  {tag}: {ne} of {a.numel()} elements differ ({ne / a.numel():.2e}), max |diff| / max |a| = {float(diff.max() / scale):.3e}', flush=True)
    return ne


if 'check' in what:
    from saragan_amd import functional as F
    for (nn, d, h, w) in ((2, 6, 20, 64), (4, 8, 32, 64), (3, 5, 18, 32), (8, 32, 128, 128)):
        for ups in (False, True):
            if ups and ((d | h | w) & 1):
                continue
            m = make(nn, d, h, w, ups)
            ys = {}
            for masked_in in ((False, True) if ups else (False,)):
                for masked_out in (False, True):
                    if ups and not masked_in and masked_out:
                        continue
                    for tag, env in (('3p', 0), ('ks', 1)):
                        set_env(SG_FWD_NO_3P=env)
                        y = torch.full((nn, d, h, w, 32), float('nan'), device=dev, dtype=torch.bfloat16)
                        try:
                            names = kernel_of(lambda: launch(m, y, masked_in, masked_out))
                        except _lib.SgError as e:
                            if tag == '3p':
                                raise
                            lib.sg_prof_enable(0)
                            names, y = ['declined'], None
                        ys[tag] = (y, names)
                    print(f'n{nn} {d}x{h}x{w} ups{int(ups)} in-mask {int(masked_in)} out-mask {int(masked_out)}: 3p ran {ys["3p"][1]}, ks ran {ys["ks"][1]}', flush=True)
                    assert not torch.isnan(ys['3p'][0].float()).any(), 'unwritten outputs'
                    if ys['ks'][0] is not None:
                        ulp_report('3p vs K split', ys['3p'][0], ys['ks'][0])
                    if nn * d * h * w <= 4 * 8 * 32 * 64:      # against torch fp32 on the small shapes
                        xin = m['x'].float()
                        if ups:
                            xin = xin.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
                            if masked_in:
                                bits = m['bits64']
                                ch = torch.arange(64, device=dev)
                                neg = ((bits[..., ch // 32] >> (ch % 32)) & 1).bool()
                                xin = (xin * 0.125)
                                xin = torch.where(neg, (xin * 0.2).bfloat16().float(), xin)
                            xin = xin.bfloat16().float()
                        wq = (m['wt'] * 0.05).bfloat16().float()              # [3,3,3,32(out of dgrad),64(in)]: mirrored taps, swapped I/O
                        wk = wq.flip(0, 1, 2).permute(3, 4, 0, 1, 2)          # -> [32, 64, 3, 3, 3]
                        ref = torch.nn.functional.conv3d(xin.permute(0, 4, 1, 2, 3), wk, padding=1).permute(0, 2, 3, 4, 1)
                        if masked_out:
                            neg = ((m['bits32'] >> torch.arange(32, device=dev)) & 1).bool()
                            ref = torch.where(neg, ref * 0.2, ref)
                        err = float((ys['3p'][0].float() - ref).abs().max() / ref.abs().max())
                        print(f'  3p vs torch fp32: max rel err {err:.3e}', flush=True)
                        assert err < 1e-2, err
            if ups:   # the fused masked gather against the materialised masked up-scale through the same kernel: bit-identical
                set_env(SG_FWD_NO_3P=0)
                gy = m['x'].permute(0, 4, 1, 2, 3)
                signs = m['bits64']
                full = F._Up.apply(gy, 0.125, signs, 0.2, (2, 2, 2))
                full_l = full.permute(0, 2, 3, 4, 1).contiguous()
                m2 = dict(m)
                m2['x'] = full_l
                m2['shp'] = ConvShape(nn, d, h, w, 64, 32, 3, 3, 3, 0)
                ya = torch.empty((nn, d, h, w, 32), device=dev, dtype=torch.bfloat16)
                yb = torch.empty_like(ya)
                launch(m, ya, True, True)
                names = kernel_of(lambda: launch(m2, yb, False, True))
                ne = ulp_report(f'gather vs materialised ({names})', ya, yb)
                assert ne == 0
            del m
            torch.cuda.empty_cache()

if 'time' in what:
    d, h, w = 32, 128, 128
    m = make(n, d, h, w, True)
    y = torch.empty((n, d, h, w, 32), device=dev, dtype=torch.bfloat16)
    flops = 2.0 * m['vox'] * 64 * 32 * 27
    for flags in (0, 1, 2, 3):
        for env in (0, 1):
            set_env(SG_FWD_NO_3P=env, SG_DBG_FLAGS=flags)
            res = []
            for mi, mo in ((True, True), (True, False)):
                for _ in range(3):
                    launch(m, y, mi, mo)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    launch(m, y, mi, mo)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 10 * 1e3
                res.append(f'in-mask {int(mi)} out-mask {int(mo)}: {us:8.1f} us {flops / us / 1e6:6.0f} TF/s')
            print(f'n{n} SG_DBG_FLAGS={flags} {"K split" if env else "3p     "}: ' + '   '.join(res), flush=True)
    set_env(SG_FWD_NO_3P=0, SG_DBG_FLAGS=0)

if 'stamps' in what:
    d, h, w = 32, 128, 128
    m = make(n, d, h, w, True)
    y = torch.empty((n, d, h, w, 32), device=dev, dtype=torch.bfloat16)
    set_env(SG_FWD_NO_3P=0, SG_DBG_FLAGS=128)
    ts = torch.zeros(256, dtype=torch.int64, device=dev)
    launch(m, y, True, True)
    lib.sg_debug_set_ts_buffer(ts.data_ptr())
    set_env(SG_DBG_FLAGS=128)
    launch(m, y, True, True)
    torch.cuda.synchronize()
    lib.sg_debug_set_ts_buffer(None)
    set_env(SG_DBG_FLAGS=0)
    t = ts.cpu().numpy()
    for g in range(2):
        v = t[g * 128:(g + 1) * 128]
        v = v[v > 0]
        print('group', g, 'stamps', len(v), 'deltas (shader cycles):', [int(b - a) for a, b in zip(v[:60], v[1:61])])
