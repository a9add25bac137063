"""The validation metrics INSIDE the training loop: --calc_metrics, --metrics_every_nsteps,
--compute_{swds,ssims,psnrs,mses,nrmses}, --num_metric_samples, --validation_fraction / --test_fraction honoured by
saragan_amd.train as optuna_objective.py:121-124,443,500-507,593-627 does -- split of the file table, metrics on the training
weights and on the EMA weights every N images, the whole test / validation subsets at the end of a phase -- with every
reported value checked against oracle/metrics_oracle.py evaluated on the very batches the loop used."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _args(data, logdir):
    from saragan_amd.main import build_parser, finalize_args
    argv = ['pgan', str(data) + '/', '--start_shape', '(1, 4, 4, 4)', '--final_shape', '(1, 32, 32, 32)',      # utils.get_num_phases = log2(32 / 4) = 3 phases: 4^3, 8^3, 16^3
           
            '--starting_phase', '2', '--ending_phase', '3', '--base_batch_size', '8', '--latent_dim', '16',
            '--noise_stddev', '0.01', '--mixing_nimg', '16', '--stabilizing_nimg', '16', '--loss_fn', 'logistic',
            '--gp_weight', '1', '--data_mean', '1024', '--data_stddev', '1024', '--logdir', str(logdir),
            '--g_lr', '1e-3', '--d_lr', '1e-3', '--checkpoint_every_nsteps', '1000000', '--dtype', 'f32',
            '--calc_metrics', '--compute_FID', '--compute_swds', '--compute_ssims', '--compute_psnrs', '--compute_mses',
            '--compute_nrmses', '--metrics_every_nsteps', '16', '--num_metric_samples', '4', '--metrics_batch_size', '2',
            '--validation_fraction', '0.2', '--test_fraction', '0.1']
    args, unknown = build_parser().parse_known_args(argv)
    assert not unknown
    args.kernel_spec = [[[3, 3, 3], [3, 3, 3]]] * 3
    args.filter_spec = [[16, 16], [16, 8], [8, 8]]
    args._metric_tap = []
    return finalize_args(args)


def test_metrics_in_the_training_loop_match_the_oracle(tmp_path, monkeypatch, capsys):
    from oracle import metrics_oracle as M
    from saragan_amd.metrics import save_metrics as SM
    from saragan_amd.train import run_training
    data, logdir = tmp_path / 'data', tmp_path / 'log'
    for z, xy in ((4, 4), (8, 8), (16, 16)):
        d = data / f'{xy}x{xy}'
        d.mkdir(parents=True)
        for i in range(20):
            rng = np.random.default_rng(1234 + i)
            np.save(d / f'{i:04d}.npy', np.clip(rng.normal(1024, 512, (z, xy, xy)), 0, 4095).astype(np.int16))
    # the sliced Wasserstein distance draws its neighbourhoods and directions from numpy's global generator: pinned per call
    real_swd = SM.SWD.get_swd_for_volumes

    def seeded(a, b):
        np.random.seed(123)
        return real_swd(a, b)
    monkeypatch.setattr(SM.SWD, 'get_swd_for_volumes', seeded)
    args = _args(data, logdir)
    out = run_training(args, log_every=10 ** 6)
    text = capsys.readouterr().out
    tap = args._metric_tap
    assert 'Split dataset of 20 samples: train 14, validation 4, test 2' in text
    assert 'FID is NOT computed' in text and 'PSNR: ' in text and 'Normalized Root MSE: ' in text and 'MSE: ' in text
    # phase 2 (8^3... 8 x 8 x 8 volumes, batch 4): 8 steps, metrics at local_step 0 and 16 -> 2 x (train weights, EMA weights);
    # end of phase: test + validation under the EMA weights.  Phase 3 (16^3, batch 2): 16 steps, metrics at 0 and 16.
    tags = [(ph, tag) for ph, tag, _, _ in tap]
    for ph in (2, 3):
        assert tags.count((ph, 'loop')) == 2 and tags.count((ph, 'loop_EMA')) == 2, tags
        assert (ph, 'test') in tags and (ph, 'validation') in tags and (ph, 'train') not in tags
    assert 'SWDS: ' in text and 'SSIM: ' in text                      # phase 3 reaches 16 voxels: both switch on there
    checked = 0
    for ph, tag, m, kept in tap:
        assert kept, (ph, tag)
        n_expected = {'loop': 4, 'loop_EMA': 4, 'test': 2, 'validation': 4}[tag]
        assert sum(r.shape[0] for r, _ in kept) == n_expected, (ph, tag, [r.shape for r, _ in kept])
        want = dict(psnr=[], mse=[], nrmse=[], ssim=[], swd=[])
        for real, fake in kept:
            real, fake = real.astype(np.float64), fake.astype(np.float64)
            assert real.shape == fake.shape and real.shape[1:] == (1, 4 * 2 ** (ph - 1), 4 * 2 ** (ph - 1), 4 * 2 ** (ph - 1))
            want['mse'].append(M.mean_squared_error(real, fake))
            want['psnr'].append(M.peak_signal_noise_ratio(real, fake, 3072))
            want['nrmse'].append(M.normalized_root_mse(real, fake))
            if ph == 3:
                want['ssim'].append(M.get_ssim(real, fake))
                np.random.seed(123)
                want['swd'].append(M.get_swd_for_volumes(real.astype(np.float32).copy(), fake.astype(np.float32).copy()))
        for k in ('psnr', 'mse', 'nrmse'):
            np.testing.assert_allclose(m[k], np.mean(want[k]), rtol=1e-6, err_msg=f'{ph} {tag} {k}')
        if ph == 3:
            np.testing.assert_allclose(m['ssim'], np.mean(want['ssim']), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(m['swd'], np.array(want['swd']).mean(axis=0), rtol=1e-3)
            assert len(m['swd']) == 2 and abs(m['swd'][-1] - np.mean(m['swd'][:-1])) < 1e-9
        else:
            assert 'ssim' not in m and 'swd' not in m              # save_metrics.py:78-79: below 16 voxels they are switched off
        assert 'FID' not in m
        checked += 1
    assert checked == 12
    # the metrics under EMA weights really used other weights: after a few steps the two fake batches differ
    loop = [t for t in tap if t[0] == 3 and t[1] == 'loop'][-1]
    ema = [t for t in tap if t[0] == 3 and t[1] == 'loop_EMA'][-1]
    assert not np.array_equal(loop[3][0][1], ema[3][0][1])
    assert 'metrics_validation' in out['stats'][3] and 'metrics_test' in out['stats'][3]
