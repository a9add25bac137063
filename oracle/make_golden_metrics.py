"""Generates tests/golden/ref_swd.npz by RUNNING the reference's own importable metric code
(/root/reference/pgan_pytorch/metrics/swd.py: numpy + scipy only) in the build container: pyr_down, pyr_up,
generate_laplacian_pyramid and sliced_wasserstein (seeded numpy global generator) on small random volumes.  Those
functions are identical in SURFGAN_3D/metrics/swd.py except pyr_up's gain (4 there, 8 in the 3-D tree, swd.py:70-74).
TEST INFRASTRUCTURE ONLY; only data is stored.   Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_metrics.py"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/pgan_pytorch/metrics/swd.py'

if __name__ == '__main__':
    spec = importlib.util.spec_from_file_location('ref_swd', REF)
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    rng = np.random.default_rng(7)
    x = rng.normal(size=(2, 1, 8, 16, 32)).astype(np.float32)
    down = R.pyr_down(x)
    up4 = R.pyr_up(down)
    pyr = R.generate_laplacian_pyramid(x.copy(), 2)
    a = rng.normal(size=(96, 30)).astype(np.float32)
    b = (rng.normal(size=(96, 30)) * 1.3 + 0.2).astype(np.float32)
    np.random.seed(11)
    swd = R.sliced_wasserstein(a, b, 3, 20)
    out = os.path.join(ROOT, 'tests', 'golden', 'ref_swd.npz')
    np.savez_compressed(out, x=x, down=down, up_gain4=up4, pyr0_gain4=pyr[0], pyr1=pyr[1], a=a, b=b, swd_seed11=np.float64(swd))
    print('wrote', out, down.shape, up4.shape, float(swd))
