"""conv_fwd5 (the streamed kernel's lean fast path) against conv_fwd4 (SG_FWD_NO_V5=1) on shapes that engage it: every
epilogue variant, ragged H and odd D, 64/128 input and output channels.  Diagnostic."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib, functional as F   # noqa: E402

lib = None
dev = None


def run(off, fn):
    os.environ['SG_FWD_NO_V5'] = '1' if off else '0'
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
    for (n, cin, cout, sp) in [(8, 64, 64, (16, 64, 64)), (4, 64, 128, (15, 62, 64)), (24, 128, 64, (8, 32, 32)), (16, 128, 128, (8, 32, 32)),
                               (3, 48, 64, (16, 64, 96))]:
        x = torch.randn(n, cin, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn(3, 3, 3, cin, cout, device=dev)
        b = torch.randn(cout, device=dev) * 0.3
        coef = (2.0 / (27 * cin)) ** 0.5
        words = torch.randint(-2 ** 31, 2 ** 31 - 1, (n * sp[0] * sp[1] * sp[2] * (cout // 32),), device=dev, dtype=torch.int64).to(torch.int32)
        cases = {
            'plain': dict(),
            'bias+act+signs': dict(bias=b, act=True, want_signs=True),
            'masked': dict(mask_bits=words, mask_slope=0.2),
            'pn+act+scale+signs': dict(bias=b, act=True, pixel_norm=True, want_scale=True, want_signs=True),
            'pn': dict(bias=b, act=True, pixel_norm=True),
        }
        for name, kw in cases.items():
            got, kg = run(False, lambda: F.raw_conv(x, w, coef, False, **kw))
            ref, kr = run(True, lambda: F.raw_conv(x, w, coef, False, **kw))
            line = f'n{n} {cin}->{cout} {sp} {name}: {kg} vs {kr}:'
            worst = 0.0
            for xg, yr in zip(got, ref):
                if xg is None and yr is None:
                    continue
                if xg.dtype == torch.int32:
                    d = float((xg != yr).float().mean())
                    line += f' signs differ {d:.1e};'
                    worst = max(worst, 10 * d)
                else:
                    d = float((xg.float() - yr.float()).abs().max() / yr.float().abs().max())
                    fin = bool(torch.isfinite(xg.float()).all())
                    line += f' err {d:.2e}{"" if fin else " NON-FINITE"};'
                    worst = max(worst, d if fin else 1e9)
            engaged = any('fwd5' in k for k in kg)
            print(line, '' if engaged else '(fwd5 not engaged)')
            if worst > 2e-2 or (not engaged and 'pn' not in name and cin % 16 == 0 and cout % 64 == 0):
                ok = False
                print('   ** MISMATCH / not engaged')
        # H x W pooled epilogue (sg_conv_epilogue.pool = 2) against the block mean of the plain launch
        if (sp[1] | sp[2]) % 2 == 0:
            for name, kw in {'pool': dict(), 'pool+bias+act+signs': dict(bias=b, act=True, want_signs=True)}.items():
                os.environ['SG_FWD_NO_V5'] = '0'
                lib.sg_config_reload()
                res = F.raw_conv(x, w, coef, False, pool=2, **kw)
                full = F.raw_conv(x, w, coef, False, **kw)
                if res is None:
                    print(f'n{n} {cin}->{cout} {sp} {name}: not fused  ** FAILED')
                    ok = False
                    continue
                ref = torch.nn.functional.avg_pool3d(full[0].float(), (1, 2, 2))
                d = float((res[0].float() - ref).abs().max() / ref.abs().max())
                sg = 0.0 if res[2] is None else float((res[2] != full[2]).float().mean())
                print(f'n{n} {cin}->{cout} {sp} {name}: err {d:.2e}; signs differ {sg:.1e}')
                if not d <= 1e-2 or sg > 0:
                    ok = False
                    print('   ** MISMATCH')
    os.environ['SG_FWD_NO_V5'] = '0'
    lib.sg_config_reload()
    print('OK' if ok else 'FAILED')
    return ok


if __name__ == '__main__':
    sys.exit(0 if main() else 1)
