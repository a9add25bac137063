"""Diagnostic: which gradients of one cfg3 step (batch 2, bf16) differ between two runs from the same state in
reproducible mode."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import saragan_amd
import saragan_amd.optimization as opt
from oracle import make_loss_curve as MC
from saragan_amd.networks import loss as L
from saragan_amd.networks.ops import ScalarVariable
from saragan_amd.networks.pgan.discriminator import discriminator
from saragan_amd.networks.pgan.generator import generator
from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store

saragan_amd.set_deterministic(True)
s = MC.cfg3_setup(torch.float32)
set_compute_dtype(torch.bfloat16)
store = VariableStore('cuda', seed=0)
og = opt.AdamOptimizer(ScalarVariable(1e-3, 'g_lr'), 0.0, 0.9)
od = opt.AdamOptimizer(ScalarVariable(1e-3, 'd_lr'), 0.0, 0.9)
ph = opt.Placeholder([s['n'], *s['img']])
c = s['cfg']
with use_store(store):
    tup = opt.optimize_step(og, od, generator, discriminator, ph, s['latent'], ScalarVariable(0.0, 'alpha'), s['phase'], MC.BASE,
                            s['kernel_spec'], s['filter_spec'], 'leaky_relu', 0.2, c['loss_fn'], c['gp_weight'], 'simultaneous',
                            False, False, c['noise_stddev'], None)
store.load_state_dict(s['p0'], strict=True)
sess = opt.Session('cuda')
real, rnd = MC.cfg3_inputs(s, 0, torch.float32)
runs = []
for rep in range(2):
    L.set_random_source(L.InjectedRandom(rnd))
    gl, dl, gs, gg, dg = sess.run([tup[2], tup[3], tup[5], tup[6], tup[8]], feed_dict={ph: real})
    runs.append((float(gl), float(dl), gs.clone(), [g.clone() for g in gg], [g.clone() for g in dg]))
print('losses', runs[0][:2], runs[1][:2], 'sample equal', torch.equal(runs[0][2], runs[1][2]))
for tag, idx, hv in (('G', 3, tup[7]), ('D', 4, tup[9])):
    for v, a, b in zip(hv, runs[0][idx], runs[1][idx]):
        if not torch.equal(a, b):
            print(tag, v.key, 'differs: max', float((a - b).abs().max()), 'of', float(a.abs().max()), tuple(a.shape))
print('done')
