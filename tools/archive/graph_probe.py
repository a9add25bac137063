"""Diagnostic: which kinds of launches of the library survive torch.cuda.graph capture + instantiate + replay on this ROCm.
Each case runs in its own process (a crash in hipStreamEndCapture / hipGraphInstantiate takes the process down).
usage: python tools/graph_probe.py [case]"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ['step', 'axpby', 'memset', 'conv_small', 'conv_big', 'wgrad', 'autograd_conv', 'autograd_backward', 'add_noise_dev', 'randn_outside']


def run_case(name):
    import torch
    from saragan_amd import _lib
    from saragan_amd import functional as F
    lib = _lib.load()
    dev = torch.device('cuda:0')

    if name.startswith('step'):
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from tests import test_hipgraph_gpu as T
        gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
        w, l, ncap = T._run(gd, 4, torch.float32, captured=True)
        print(name, 'captures', ncap, 'losses', l[-1], flush=True)
        print(name, 'OK', flush=True)
        return

    def body():
        if name == 'axpby':
            a = torch.randn(1 << 20, device=dev)
            out = torch.empty_like(a)
            return lambda: _lib.check(lib.sg_axpby(a.data_ptr(), None, out.data_ptr(), 2.0, 0.0, a.numel(), 0,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if name == 'memset':
            a = torch.empty(1 << 20, device=dev)
            return lambda: a.zero_()
        if name in ('conv_small', 'conv_big'):
            n, c, d, h, w = (2, 16, 4, 16, 16) if name == 'conv_small' else (4, 32, 8, 64, 64)
            x = torch.randn(n, c, d, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
            wt = torch.randn(3, 3, 3, c, c, device=dev)
            return lambda: F.raw_conv(x, wt, 0.1, False)
        if name == 'wgrad':
            x = torch.randn(2, 32, 8, 32, 32, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
            dy = torch.randn(2, 32, 8, 32, 32, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
            return lambda: F.raw_wgrad(x, dy, (3, 3, 3), 0.3, want_db=True)
        if name in ('autograd_conv', 'autograd_backward'):
            from saragan_amd.networks import ops
            x = torch.randn(2, 16, 4, 16, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
            wt = torch.randn(3, 3, 3, 16, 16, device=dev, requires_grad=True)
            wt.grad = torch.zeros_like(wt)

            def f():
                y = F.conv3d(x, wt, 0.1) if hasattr(F, 'conv3d') else F._Conv.apply(x, wt, 0.1, False, False)
                loss = y.float().square().mean()
                if name == 'autograd_backward':
                    torch.autograd.backward(loss, inputs=[wt])
                return loss
            return f
        if name == 'add_noise_dev':
            x = torch.randn(1 << 16, device=dev).bfloat16()
            ctr = torch.zeros(1, dtype=torch.int64, device=dev)
            return lambda: F.add_noise(x.view(1, 1, 1, 1, -1), 0.1, 7, offset=ctr)
        if name == 'randn_outside':
            z = torch.empty(8, 16, device=dev)
            return lambda: z.mul(2.0)
        raise SystemExit(name)
    fn = body()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r = fn()
    print(name, 'captured', flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(name, 'OK', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=120)
            tail = [ln for ln in (p.stdout + p.stderr).splitlines() if ln.strip() and 'amdgpu.ids' not in ln][-2:]
            print(f'{c:20s} rc={p.returncode} {" | ".join(t[:150] for t in tail)}', flush=True)
