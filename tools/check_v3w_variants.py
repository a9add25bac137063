"""conv_fwd3w (wave-private planes, sliding accumulators, v_mfma_f32_16x16x32_bf16; csrc/conv3w.hip) against the sliding-halo
kernel it replaces (SG_FWD3S_16=0) and, for the variants torch can state in a few lines, against torch fp32: every epilogue
variant (plain, bias + LeakyReLU + sign words, masked, pixel-norm, pooled, masked + pooled, pixel-norm backward), as forward and
as data gradient (flipped weights), on ragged H, odd D, 64 output channels, and at batch 1-2 where the columns are cut into D
segments.  Asserts that the new kernel ran.  Diagnostic; prints the worst deviations."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib, functional as F   # noqa: E402

lib = None
dev = None


def run(new, fn):
    os.environ['SG_FWD3S_16'] = '1' if new else '0'
    lib.sg_config_reload()
    lib.sg_prof_enable(1)
    out = fn()
    torch.cuda.synchronize()
    ents = (_lib.ProfEntry * 64)()
    cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 64, C.byref(cnt))
    lib.sg_prof_enable(0)
    return out, sorted({ents[i].kernel.decode() for i in range(cnt.value)})


def cmp(tag, a, b, tol=2e-2):
    worst = 0.0
    for x, y in zip(a, b):
        if x is None and y is None:
            continue
        if x.dtype in (torch.int32, torch.int64):
            d = float((x != y).float().mean())
            print(f'   {tag}: sign words differing {d:.2e}')
            worst = max(worst, d * 10)
            continue
        x, y = x.float(), y.float()
        bad = ~torch.isfinite(x)
        d = float((x - y).abs().max() / y.abs().max())
        if d > tol:
            wrong = ((x - y).abs() > tol * y.abs().max()).nonzero()
            print(f'      {wrong.shape[0]} wrong elements of {x.numel()}, shape {tuple(x.shape)} strides {x.stride()}')
            print('      first:', wrong[:6].tolist(), ' last:', wrong[-6:].tolist())
            for dim in range(wrong.shape[1]):
                u = torch.unique(wrong[:, dim])
                print(f'      dim {dim}: {u.numel()} distinct, {u[:16].tolist()}')
        ne = float((x != y).float().mean())
        print(f'   {tag}: max |diff| / max |ref| = {d:.3e}   elements differing {ne:.2e}   non-finite {int(bad.sum())}')
        worst = max(worst, d if not bool(bad.any()) else 1e9)
    return worst


def torch_ref(x, w, coef, flip, b=None, act=False, pn=False):
    wq = (w * coef).bfloat16().float()
    if flip:      # data gradient: taps mirrored, channels swapped
        wq = wq.flip(0, 1, 2).transpose(3, 4)
    z = torch.nn.functional.conv3d(x.float(), wq.permute(4, 3, 0, 1, 2).contiguous(), padding=1)
    if b is not None:
        z = z + b.view(1, -1, 1, 1, 1)
    if act:
        z = torch.nn.functional.leaky_relu(z, 0.2)
    if pn:
        z = z * torch.rsqrt((z * z).mean(1, keepdim=True) + 1e-8)
    return z


def main():
    global lib, dev
    lib = _lib.load()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    ok = True
    #            n  cout  (D, H, W)        flip
    shapes = [(2, 32, (6, 128, 256), False), (2, 64, (8, 120, 128), False), (3, 32, (5, 72, 96), True), (1, 32, (16, 64, 64), False),
              (2, 64, (32, 32, 32), True), (1, 32, (2, 16, 32), False), (5, 32, (4, 10, 32), False)]
    for (n, cout, sp, flip) in shapes:
        cin = 32
        x = torch.randn(n, cin, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn(3, 3, 3, cout, cin, device=dev) if flip else torch.randn(3, 3, 3, cin, cout, device=dev)
        b = torch.randn(cout, device=dev) * 0.3
        coef = (2.0 / (27 * cin)) ** 0.5
        nvox = n * sp[0] * sp[1] * sp[2]
        words = torch.randint(-2 ** 31, 2 ** 31 - 1, (nvox * (cout // 32),), device=dev, dtype=torch.int64).to(torch.int32)
        cases = {
            'plain': dict(),
            'bias+act+signs': dict(bias=b, act=True, want_signs=True),
            'masked': dict(mask_bits=words, mask_slope=0.2),
            'pool': dict(bias=b, act=True, pool=True),
            'pool+signs': dict(bias=b, act=True, want_signs=True, pool=True),
            'masked+pool': dict(mask_bits=words, mask_slope=0.2, pool=True),
        }
        if cout == 32:
            cases['pn+act+scale+signs'] = dict(bias=b, act=True, pixel_norm=True, want_scale=True, want_signs=True)
            cases['pn'] = dict(bias=b, act=True, pixel_norm=True)
            py = torch.randn(n, cout, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
            ps = torch.rand(nvox, device=dev) + 0.5
            cases['masked+pn_bwd'] = dict(mask_bits=words, mask_slope=0.2, pn_bwd=(py, ps))
        for name, kw in cases.items():
            if kw.get('pool') and ((sp[0] | sp[2]) & 1):
                continue
            got, kg = run(True, lambda: F.raw_conv(x, w, coef, flip, False, **kw))
            ref, kr = run(False, lambda: F.raw_conv(x, w, coef, flip, False, **kw))
            print(f'n{n} {cin}->{cout} {sp} flip{int(flip)} {name}: {kg} vs {kr}', flush=True)
            if got is None:
                ok = False
                print('   ** declined on the new path')
                continue
            if not any('conv_fwd3w' in k for k in kg):
                ok = False
                print('   ** conv_fwd3w did not run', kg)
            wst = 0.0
            if ref is not None:
                wst = cmp(name, got, ref)
            else:
                print('   (no reference path for this epilogue on this shape)')
            if name in ('plain', 'bias+act+signs', 'pn'):
                z = torch_ref(x, w, coef, flip, kw.get('bias'), kw.get('act', False), kw.get('pixel_norm', False))
                e = float((got[0].float() - z).abs().max() / z.abs().max())
                print(f'   vs torch fp32: max |diff| / max |ref| = {e:.3e}')
                wst = max(wst, e / 2)       # (one bf16 rounding of the output: 2^-8 relative to the element, bound 1e-2 of the max)
            if wst > 2e-2:
                ok = False
                print('   ** MISMATCH')
    os.environ['SG_FWD3S_16'] = '1'
    lib.sg_config_reload()
    # ---- the pointwise layers next to the 32-channel convolution, in its epilogue (round 5): to_rgb of a generator stage's output
    # (sg_conv_epilogue.rgb_*) and from_rgb's whole backward in the data gradient of its output (pw_*), against the separate passes
    for (n, sp, flip) in [(2, (6, 128, 256), False), (3, (5, 72, 96), True), (1, (16, 64, 64), False), (5, (4, 10, 32), True)]:
        cin = cout = 32
        x = torch.randn(n, cin, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn(3, 3, 3, cout, cin, device=dev) if flip else torch.randn(3, 3, 3, cin, cout, device=dev)
        b = torch.randn(cout, device=dev) * 0.3
        coef = (2.0 / (27 * cin)) ** 0.5
        nvox = n * sp[0] * sp[1] * sp[2]
        w_rgb = torch.randn(1, 1, 1, cout, 1, device=dev)
        b_rgb = torch.randn(1, device=dev)
        mat = F._rgb_matrix(w_rgb, 0.25, torch.bfloat16)
        res = F.raw_conv(x, w, coef, flip, False, bias=b, act=True, pixel_norm=True, want_scale=True, want_signs=True, rgb=(mat, b_rgb))
        y0, sc0, sg0 = F.raw_conv(x, w, coef, flip, False, bias=b, act=True, pixel_norm=True, want_scale=True, want_signs=True)
        img0 = F.raw_conv(y0, w_rgb, 0.25, False, False, bias=b_rgb)[0]
        print(f'n{n} 32->32 {sp} flip{int(flip)} pn+signs+to_rgb epilogue:', 'declined' if res is None else 'ran', flush=True)
        if res is None:
            ok = False
        else:
            wst = cmp('stage + image', [res[0], res[1], res[2], res[3]], [y0, sc0, sg0, img0])
            if wst > 2e-2:
                ok = False
                print('   ** MISMATCH')
        # from_rgb's backward: image x_img, gradient g of conv_1's output (here: x), from_rgb's sign words
        words = torch.randint(-2 ** 31, 2 ** 31 - 1, (nvox,), device=dev, dtype=torch.int64).to(torch.int32)
        x_img = torch.randn(n, 1, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        w_frgb = torch.randn(1, 1, 1, 1, cout, device=dev)
        wmat = F._rgb_matrix(w_frgb, 0.5, torch.bfloat16, small_is_cin=True)
        dw, db = torch.zeros(1, 1, 1, 1, cout, device=dev), torch.zeros(cout, device=dev)
        dimg = F.raw_conv(x, w, coef, flip, False, mask_bits=words, mask_slope=0.2,
                          pw_bwd=dict(x=x_img, wmat=wmat, want_dx=True, dw=dw, db=db, coef=0.5))
        g0 = F.raw_conv(x, w, coef, flip, False, mask_bits=words, mask_slope=0.2)[0]
        ref = F._pw_backward(x_img, g0, w_frgb, 0.5, True)
        print(f'n{n} 32->32 {sp} flip{int(flip)} masked + from_rgb backward epilogue:', 'declined' if dimg is None else 'ran', flush=True)
        if dimg is None or ref is None:
            ok = False
            print('   ** declined', dimg is None, ref is None)
        else:
            wst = cmp('image gradient / filter / bias gradient', [dimg, dw.reshape(-1), db], [ref[0], ref[1].reshape(-1), ref[2]], tol=5e-3)
            if wst > 5e-3:
                ok = False
                print('   ** MISMATCH')
    # ---- whole 2 x 2 x 2 means from the epilogue (pool = 3) against the D x W means + the H pairs
    for (n, cout, sp) in [(2, 64, (6, 128, 256)), (3, 32, (4, 72, 96)), (1, 64, (16, 64, 64))]:
        x = torch.randn(n, 32, *sp, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn(3, 3, 3, 32, cout, device=dev)
        b = torch.randn(cout, device=dev) * 0.3
        coef = (2.0 / (27 * 32)) ** 0.5
        words = torch.randint(-2 ** 31, 2 ** 31 - 1, (n * sp[0] * sp[1] * sp[2] * (cout // 32),), device=dev, dtype=torch.int64).to(torch.int32)
        for name, kw in (('pool3+signs', dict(bias=b, act=True, want_signs=True)), ('masked+pool3', dict(mask_bits=words, mask_slope=0.2))):
            got = F.raw_conv(x, w, coef, False, False, pool=3, **kw)
            ref = F.raw_conv(x, w, coef, False, False, pool=1, **kw)
            print(f'n{n} 32->{cout} {sp} {name}:', 'declined' if got is None else 'ran', flush=True)
            if got is None or ref is None:
                ok = False
                continue
            ref_y = F._Down.apply(ref[0], 0.5, None, (1, 2, 1))
            wst = cmp(name, [got[0], got[2]], [ref_y, ref[2]])
            if wst > 2e-2:
                ok = False
                print('   ** MISMATCH')
    print('OK' if ok else 'FAILED')
    return ok


if __name__ == '__main__':
    sys.exit(0 if main() else 1)
