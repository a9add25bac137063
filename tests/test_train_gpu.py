"""GPU test of the training loop (saragan_amd.train, the optuna_objective.py:98-600 mirror): two phases on
synthetic .npy volumes, checking the schedule arithmetic, freeze-during-mixing (Q4), EMA end-of-phase checkpoint
(Q5), name-keyed phase hand-off and that losses stay finite."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(data, logdir, **kw):
    from saragan_amd.main import build_parser, finalize_args
    argv = ['pgan', str(data) + '/', '--start_shape', '(1, 1, 4, 4)', '--final_shape', '(1, 4, 16, 16)',
            '--starting_phase', '1', '--ending_phase', '2', '--base_batch_size', '4', '--latent_dim', '16',
            '--noise_stddev', '0.01', '--mixing_nimg', '16', '--stabilizing_nimg', '16', '--loss_fn', 'wgan',
            '--gp_weight', '10', '--data_mean', '1024', '--data_stddev', '1024', '--logdir', str(logdir),
            '--g_lr', '1e-3', '--d_lr', '1e-3', '--checkpoint_every_nsteps', '1000000', '--dtype', 'f32']
    args, _ = build_parser().parse_known_args(argv)
    args.kernel_spec = [[[1, 3, 3], [1, 3, 3]], [[1, 3, 3], [3, 3, 3]], [[3, 3, 3], [3, 3, 3]]]
    args.filter_spec = [[16, 16], [16, 8], [8, 8]]
    for k, v in kw.items():
        setattr(args, k, v)
    return finalize_args(args)


def _make_data(root):
    for ph, (z, xy) in enumerate([(1, 4), (2, 8), (4, 16)], 1):
        d = root / f'{xy}x{xy}'
        d.mkdir(parents=True)
        for i in range(12):
            rng = np.random.default_rng(1234 + i)
            v = np.clip(rng.normal(1024, 512, (z, xy, xy)), 0, 4095).astype(np.int16)
            np.save(d / f'{i:04d}.npy', v)


def test_two_phase_run(tmp_path):
    from saragan_amd.train import run_training
    from saragan_amd.utils import load_checkpoint
    from saragan_amd.networks.pgan.variables import pgan_variable_shapes
    data, logdir = tmp_path / 'data', tmp_path / 'log'
    _make_data(data)
    args = _args(data, logdir)
    out = run_training(args)
    st = out['stats']
    # phase 1: batch 4, (16 + 16) images -> 8 steps; phase 2: batch 2 -> 16 steps
    assert (st[1]['batch_size'], st[1]['steps']) == (4, 8) and (st[2]['batch_size'], st[2]['steps']) == (2, 16)
    assert all(np.isfinite(st[p]['d_loss']) and np.isfinite(st[p]['g_loss']) for p in (1, 2))
    base = (1, 1, 4, 4)
    for ph in (1, 2):
        ck = load_checkpoint(os.path.join(str(logdir), f'model_{ph}'))
        plan = pgan_variable_shapes(ph, base, 16, args.kernel_spec, args.filter_spec)
        assert set(ck) == set(plan)                                      # TF-name keyed, exactly the phase's variables
        assert all(tuple(ck[k].shape) == tuple(plan[k]) for k in plan)
        assert all(np.isfinite(v).all() for v in ck.values())
    store = out['store']
    ck2 = load_checkpoint(os.path.join(str(logdir), 'model_2'))
    for k, v in store.vars.items():      # Q5: weights were overwritten with the EMA before the final save
        np.testing.assert_array_equal(v.detach().cpu().numpy(), ck2[k])


def test_mixing_freezes_previous_phase_and_resume(tmp_path):
    """Phase 2 restarted from the phase-1 checkpoint (--continue_path, --starting_phase 2) with only mixing
    steps: every variable that existed in phase 1 must still equal the checkpoint (quirk Q4), new ones move."""
    from saragan_amd.train import run_training
    from saragan_amd.utils import load_checkpoint
    data, logdir = tmp_path / 'data', tmp_path / 'log'
    _make_data(data)
    out1 = run_training(_args(data, logdir, ending_phase=1))
    ck1 = load_checkpoint(os.path.join(str(logdir), 'model_1'))
    args = _args(data, tmp_path / 'log2', starting_phase=2, continue_path=os.path.join(str(logdir), 'model_1'),
                 mixing_nimg=64, stabilizing_nimg=0, ema_beta=0.0)     # ema_beta 0: the EMA copy equals the weights
    out2 = run_training(args, max_steps_per_phase=3)
    store = out2['store']
    moved = 0
    for k, v in store.vars.items():
        cur = v.detach().cpu().numpy()
        if k in ck1:
            np.testing.assert_array_equal(cur, ck1[k], err_msg=f'{k} changed during mixing')
        else:
            moved += int(np.abs(cur).sum() > 0)
    assert moved > 0
