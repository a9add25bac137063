"""INTEGRATION.md, level 1: one training step driven through the module paths the REFERENCE's loop imports
(optuna_objective.py:64-65 `importlib.import_module(f'networks.{architecture}.generator')`, `import optimization as
opt`, `from dataset import NumpyPathDataset`, `from ExtendedEMA import ExtendedEMA`), resolved through
saragan_amd/dropin on sys.path.  Runs in a child process so that top-level names like `networks`, `optimization`,
`dataset`, `utils` do not shadow anything in the test process; results are checked against the golden step fixture."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import importlib, json, os, sys, tempfile
import numpy as np
import torch
root, fixture = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import saragan_amd
sys.path.insert(0, saragan_amd.dropin_path())
# ---- the reference loop's own import statements (optuna_objective.py:1-66) --------------------------------------
import optimization as opt
from dataset import NumpyPathDataset
from ExtendedEMA import ExtendedEMA
from networks.ops import ScalarVariable
generator = importlib.import_module('networks.pgan.generator').generator
discriminator = importlib.import_module('networks.pgan.discriminator').discriminator
from networks import loss as L
assert opt.__file__.startswith(saragan_amd.dropin_path())
from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
from tests.stepfix import BASE_SHAPE, FILTER_SPEC, KERNEL_SPEC, LATENT, load_step_fixture
fx = load_step_fixture(fixture, torch.float64)
set_compute_dtype(torch.float32)
store = VariableStore('cuda', seed=0)
L.set_random_source(L.InjectedRandom({k: v.float() for k, v in fx['rnd'].items()}))
alpha = ScalarVariable(fx['alpha'], 'alpha')
import argparse
args = argparse.Namespace(optimizer='Adam', d_optimizer='Adam', adam_beta1=0.0, adam_beta2=0.9, d_adam_beta1=0.0, d_adam_beta2=0.9)
og, od = opt.get_optimizer(ScalarVariable(1e-3, 'd_lr'), ScalarVariable(1e-3, 'g_lr'), args)      # optimization.py:6 (d_lr first)
ph = opt.Placeholder([4, 1, 1, 1, 1])
with use_store(store):
    tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, alpha, fx['phase'], BASE_SHAPE, KERNEL_SPEC,
                            FILTER_SPEC, 'leaky_relu', 0.2, fx['loss_fn'], fx['cfg']['gp_weight'], 'simultaneous', False,
                            False, 0.01, None if fx['freeze'] is None else list(fx['freeze']))
store.load_state_dict(fx['p0'], strict=True)
ema = ExtendedEMA(list(store.vars.keys()), 0.99, graph=tup[0].graph)
# the batch comes from .npy files through the reference's dataset class (dataset.py:155-349)
tmp = tempfile.mkdtemp()
d = os.path.join(tmp, '8x8')
os.makedirs(d)
real = fx['real'].numpy()
for i in range(real.shape[0]):
    np.save(os.path.join(d, f'{i:04d}.npy'), real[i, 0].astype(np.float32))
ds = NumpyPathDataset(d + '/', None, False, True, seed=1)
batch = ds.batch(real.shape[0])
order = [int(np.argmin([np.abs(real[j, 0] - b).max() for j in range(real.shape[0])])) for b in batch[:, 0]]
assert sorted(order) == list(range(real.shape[0])), order
sess = opt.Session('cuda')
mixing = fx['freeze'] is not None
tg, td = (tup[12], tup[16]) if mixing else (tup[0], tup[1])
_, _, gl, dl = sess.run([tg, td, tup[2], tup[3]], feed_dict={ph: torch.as_tensor(real, dtype=torch.float32)})
sess.run(ema.apply())
out = dict(gen_loss=float(gl), disc_loss=float(dl),
           weights={k: v.detach().double().cpu().numpy().ravel()[:8].tolist() for k, v in store.vars.items()},
           ema={k: ema.average(k).detach().double().cpu().numpy().ravel()[:8].tolist() for k in store.vars})
print('RESULT ' + json.dumps(out))
'''


def _run_child(fixture):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-c', CHILD, ROOT, fixture], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
    assert line, r.stdout[-2000:]
    return json.loads(line[-1][7:])


@pytest.mark.gpu
def test_one_step_through_the_reference_module_paths(golden_dir):
    from tests.stepfix import load_step_fixture
    path = os.path.join(golden_dir, 'oracle_step_p2_wgan_a060.npz')
    out = _run_child(path)
    fx = load_step_fixture(path, torch.float64)
    np.testing.assert_allclose(out['gen_loss'], float(fx['gen_loss']), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out['disc_loss'], float(fx['disc_loss']), rtol=1e-4, atol=1e-5)
    for k, head in out['weights'].items():
        np.testing.assert_allclose(head, fx['p1'][k].numpy().ravel()[:8], rtol=2e-4, atol=5e-5, err_msg=k)
        np.testing.assert_allclose(out['ema'][k], fx['ema1'][k].numpy().ravel()[:8], rtol=2e-4, atol=5e-5, err_msg='ema:' + k)
