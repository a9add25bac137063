"""CLI with the flag names of SURFGAN_3D/main.py:234-355 (hot-path subset; optuna / metrics / summary flags are
accepted and ignored so that the reference's launch lines keep working).  Launch one process per GPU with
`python -m torch.distributed.run --nproc-per-node N -m saragan_amd.main pgan <data> --horovod ...`."""
import argparse
import json

from .networks.pgan.variables import preset_specs
from .utils import get_base_shape, get_num_phases


def none_or_str(v):
    return None if v == 'None' else v


def none_or_float(v):
    return None if v == 'None' else float(v)


def _spec_loader(key):
    def load(value):
        with open(value) as f:
            return json.load(f)[key]
    return load


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('architecture', type=str)
    p.add_argument('dataset_path', type=str)
    p.add_argument('--start_shape', type=str, required=True)
    p.add_argument('--final_shape', type=str, required=True)
    p.add_argument('--starting_phase', type=int, required=True)
    p.add_argument('--ending_phase', type=int, required=True)
    p.add_argument('--scratch_path', type=str, default=None)
    p.add_argument('--base_batch_size', type=int, default=None)
    p.add_argument('--max_global_batch_size', type=int, default=None)
    p.add_argument('--mixing_nimg', type=int, default=2 ** 19)
    p.add_argument('--stabilizing_nimg', type=int, default=2 ** 19)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--horovod', default=False, action='store_true')
    p.add_argument('--checkpoint_every_nsteps', default=20000, type=int)
    p.add_argument('--logdir', default=None, type=str)
    p.add_argument('--continue_path', default=None, type=str)
    p.add_argument('--starting_alpha', default=1, type=float)
    p.add_argument('--gpu', default=False, action='store_true')
    p.add_argument('--latent_dim', type=int, required=True)
    p.add_argument('--network_size', default=None, choices=['xxs', 'xs', 's', 'm', 'l', 'xl', 'xxl'])
    p.add_argument('--activation', type=str, default='leaky_relu')
    p.add_argument('--leakiness', type=float, default=0.2)
    p.add_argument('--kernel_spec', type=_spec_loader('kernel_spec'), default=None)
    p.add_argument('--filter_spec', type=_spec_loader('filter_spec'), default=None)
    p.add_argument('--g_lr', type=float, default=1e-3)
    p.add_argument('--d_lr', type=float, default=1e-3)
    for n in ('g', 'd'):
        p.add_argument(f'--{n}_lr_increase', type=none_or_str, choices=[None, 'linear', 'exponential'], default=None)
        p.add_argument(f'--{n}_lr_decrease', type=none_or_str, choices=[None, 'linear', 'exponential'], default=None)
        p.add_argument(f'--{n}_lr_rise_niter', type=int, default=None)
        p.add_argument(f'--{n}_lr_decay_niter', type=int, default=None)
        p.add_argument(f'--{n}_scaling', default='none', choices=['linear', 'sqrt', 'none'])
        p.add_argument(f'--{n}_clipping', default=False, type=bool)
    p.add_argument('--loss_fn', default='logistic', choices=['logistic', 'wgan'])
    p.add_argument('--gp_weight', type=float, default=1)
    p.add_argument('--optim_strategy', default='simultaneous', choices=['simultaneous', 'alternate'])
    p.add_argument('--use_adasum', default=False, action='store_true')
    p.add_argument('--hipgraph', default=False, action='store_true',
                   help='(not in the reference) always replay the training step as one hipGraph (SARAGAN_HIPGRAPH=1); by default a '
                        'phase is captured when its steps measure host-bound')
    p.add_argument('--no_hipgraph', default=False, action='store_true', help='(not in the reference) never capture: SARAGAN_HIPGRAPH=0')
    p.add_argument('--ema_beta', type=float, default=0.99)
    p.add_argument('--noise_stddev', type=float, required=True)
    p.add_argument('--optimizer', type=none_or_str, choices=[None, 'Adam', 'SGD', 'Momentum', 'Adadelta'], default='Adam')
    p.add_argument('--d_use_different_optimizer', default=False, action='store_true')
    p.add_argument('--d_optimizer', type=none_or_str, choices=[None, 'Adam', 'SGD', 'Momentum', 'Adadelta'], default='Adam')
    p.add_argument('--adam_beta1', type=none_or_float, default=0)
    p.add_argument('--d_use_different_beta1', default=False, action='store_true')
    p.add_argument('--d_adam_beta1', type=none_or_float, default=0)
    p.add_argument('--adam_beta2', type=none_or_float, default=0.9)
    p.add_argument('--d_use_different_beta2', default=False, action='store_true')
    p.add_argument('--d_adam_beta2', type=none_or_float, default=0.9)
    p.add_argument('--rho', type=none_or_float, default=0.95)
    p.add_argument('--d_use_different_rho', default=False, action='store_true')
    p.add_argument('--d_rho', type=none_or_float, default=0.95)
    p.add_argument('--momentum', type=none_or_float, default=0.9)
    p.add_argument('--d_use_different_momentum', default=False, action='store_true')
    p.add_argument('--d_momentum', type=none_or_float, default=0.9)
    p.add_argument('--data_mean', default=None, type=float)
    p.add_argument('--data_stddev', default=None, type=float)
    # validation split and in-loop metrics (main.py:255-256,319-333); --compute_FID is accepted and refused (needs a download)
    p.add_argument('--validation_fraction', default=0.1, type=float)
    p.add_argument('--test_fraction', default=0.1, type=float)
    p.add_argument('--calc_metrics', default=False, action='store_true')
    p.add_argument('--compute_metrics_train', default=False, action='store_true')
    p.add_argument('--disable_compute_metrics_validation', dest='compute_metrics_validation', default=True, action='store_false')
    p.add_argument('--disable_compute_metrics_test', dest='compute_metrics_test', default=True, action='store_false')
    p.add_argument('--num_metric_samples', type=int, default=None)
    p.add_argument('--metrics_every_nsteps', default=128, type=int)
    p.add_argument('--metrics_batch_size', default=16, type=int)
    for m in ('FID', 'swds', 'ssims', 'psnrs', 'mses', 'nrmses'):
        p.add_argument(f'--compute_{m}', default=False, action='store_true')
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'], help='activation / MFMA input type (new flag)')
    p.add_argument('--max_steps_per_phase', type=int, default=None, help='smoke runs: cap the steps per phase (new flag)')
    return p


def finalize_args(args):
    """main.py:384-411 post-parse defaults: the discriminator inherits the generator's optimiser settings unless
    the --d_use_different_* switches are given; presets fill missing kernel/filter specs."""
    if not args.d_use_different_optimizer:
        args.d_optimizer = args.optimizer
    if not args.d_use_different_beta1:
        args.d_adam_beta1 = args.adam_beta1
    if not args.d_use_different_beta2:
        args.d_adam_beta2 = args.adam_beta2
    if not args.d_use_different_rho:
        args.d_rho = args.rho
    if not args.d_use_different_momentum:
        args.d_momentum = args.momentum
    for n in ('g', 'd'):       # main.py:384-399: ramp lengths default to half the mixing / stabilising images
        if getattr(args, f'{n}_lr_increase') and not getattr(args, f'{n}_lr_rise_niter'):
            setattr(args, f'{n}_lr_rise_niter', int(args.mixing_nimg / 2))
        if getattr(args, f'{n}_lr_decrease') and not getattr(args, f'{n}_lr_decay_niter'):
            setattr(args, f'{n}_lr_decay_niter', int(args.stabilizing_nimg / 2))
    if args.kernel_spec is None or args.filter_spec is None:
        if args.network_size is None:
            raise SystemExit('give --kernel_spec and --filter_spec, or --network_size for the legacy presets')
        ks, fs = preset_specs(args.network_size, get_base_shape(args.start_shape),
                              max(8, get_num_phases(args.start_shape, args.final_shape)))
        args.kernel_spec = args.kernel_spec or ks
        args.filter_spec = args.filter_spec or fs
    return args


def main(argv=None):
    p = build_parser()
    args, unknown = p.parse_known_args(argv)
    if unknown:
        print(f'ignoring flags outside the hot path: {unknown}')
    args = finalize_args(args)
    if getattr(args, 'hipgraph', False) or getattr(args, 'no_hipgraph', False):
        import os
        os.environ['SARAGAN_HIPGRAPH'] = '1' if args.hipgraph else '0'
    from .train import run_training
    out = run_training(args, max_steps_per_phase=args.max_steps_per_phase)
    for ph, st in out['stats'].items():
        print(f"phase {ph}: {st['img_s']:.2f} img/s, batch {st['batch_size']}, steps {st['steps']}")


if __name__ == '__main__':
    main()
