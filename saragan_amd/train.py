"""The training loop of SURFGAN_3D/optuna_objective.py:98-600 for a normal run (trial=None), on the HIP path:
per-phase graph build, train / validation / test split, restore from the previous phase, mixing / stabilising loop with
alpha, learning-rate and EMA ops, the validation metrics every --metrics_every_nsteps images and at the end of a phase
(SWD / SSIM / PSNR / MSE / NRMSE on the GPU: metrics/save_metrics.py), periodic and end-of-phase checkpoints.
TensorBoard summaries, FID and optuna are out of scope.  Reference quirks Q2-Q5 are
reproduced (see comments)."""
import gc
import importlib
import os
import random
import time

import numpy as np
import torch

from . import optimization as opt
from . import parallel
from .ExtendedEMA import ExtendedEMA
from .dataset import NumpyPathDataset, PinnedPrefetcher, normalize_numpy
from .metrics.save_metrics import get_compute_metrics_dict, save_metrics
from .networks import loss as L
from .networks import ops as nops
from .networks.ops import ScalarVariable
from .utils import (get_base_shape, get_current_input_shape, get_num_metric_samples, get_num_phases, get_xy_dim,
                    print_summary_to_stdout, restore_variables, save_checkpoint, scale_lr)
from .varstore import VariableStore, set_compute_dtype, use_store


def get_numpy_dataset(phase, starting_phase, start_shape, dataset_path, scratch_path, verbose, rank, world, seed,
                      local_rank=0):
    """utils.py:195-204.  One process per NODE stages the files to scratch (hvd.local_rank() == 0 in the reference)."""
    size = get_xy_dim(phase, start_shape)
    data_path = os.path.join(dataset_path, f'{size}x{size}/')
    if verbose:
        print(f'Phase {phase}: reading data from dir {data_path}')
    return NumpyPathDataset(data_path, scratch_path, copy_files=(local_rank == 0),
                            is_correct_phase=phase >= starting_phase, rank=rank, world_size=world, seed=seed)


def run_training(args, device=None, max_steps_per_phase=None, log_every=1):
    """optuna_objective(trial=None, args, config).  Returns a dict with per-phase statistics."""
    rank, world, local = parallel.init_distributed()
    horovod = bool(getattr(args, 'horovod', False)) and world > 1
    global_size = world if horovod else 1
    verbose = rank == 0
    device = torch.device(device or (f'cuda:{local}' if torch.cuda.is_available() else 'cpu'))
    if device.type != 'cuda':
        raise RuntimeError('saragan_amd trains on MI355X only: no CPU fallback')
    torch.cuda.set_device(device)
    set_compute_dtype(torch.bfloat16 if getattr(args, 'dtype', 'bf16') == 'bf16' else torch.float32)

    discriminator = importlib.import_module(f'saragan_amd.networks.{args.architecture}.discriminator').discriminator
    generator = importlib.import_module(f'saragan_amd.networks.{args.architecture}.generator').generator   # :64-65

    logdir = args.logdir or os.path.join('runs', args.architecture, time.strftime('%Y-%m-%d_%H:%M:%S', time.gmtime()))
    if verbose:
        os.makedirs(logdir, exist_ok=True)
    num_phases = get_num_phases(args.start_shape, args.final_shape)
    base_shape = get_base_shape(args.start_shape)
    ending_phase = min(getattr(args, 'ending_phase', num_phases) or num_phases, num_phases)
    seed = args.seed + (rank if horovod else 0)                      # main.py:359-370
    store = VariableStore(device, seed=seed)
    sess = opt.Session(device)
    var_list = []
    global_step = 0
    stats = {}

    for phase in range(1, ending_phase + 1):
        np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)   # :102-109
        L.set_random_source(L.RandomSource(seed * 1000 + phase, device))
        npy_data = get_numpy_dataset(phase, args.starting_phase, args.start_shape, args.dataset_path,
                                     args.scratch_path, verbose, rank if horovod else 0, global_size, seed=args.seed,
                                     local_rank=local)
        # :121-124: the split keeps the file order (consecutive scans of one patient stay on one side); quirk Q3: the
        # training batches below are still drawn from the FULL set npy_data, as the reference's are (:423-425)
        vf, tf_ = getattr(args, 'validation_fraction', 0.0) or 0.0, getattr(args, 'test_fraction', 0.0) or 0.0
        npy_data_train = npy_data_validation = npy_data_test = None
        if vf + tf_ > 0 and phase >= args.starting_phase and len(npy_data) >= 3:
            try:
                npy_data_train, npy_data_testval = npy_data.split_by_fraction(1 - (vf + tf_))
                npy_data_validation, npy_data_test = npy_data_testval.split_by_fraction(vf / (vf + tf_))
                if verbose:
                    print(f"Split dataset of {len(npy_data)} samples: train {len(npy_data_train)}, validation "
                          f"{len(npy_data_validation)}, test {len(npy_data_test)}")
            except AssertionError:      # a subset came out empty (tiny data sets): no split, no metrics
                npy_data_train = npy_data_validation = npy_data_test = None
        batch_size = max(1, args.base_batch_size // (2 ** (phase - 1)))            # :127
        if args.max_global_batch_size is not None:                                # :130-136 (quirk Q6: float)
            max_local_batch_size = args.max_global_batch_size / global_size
            if batch_size > max_local_batch_size:
                batch_size = int(max_local_batch_size)
            assert batch_size * global_size <= args.max_global_batch_size
        real_image_input = opt.Placeholder(get_current_input_shape(phase, batch_size, args.start_shape))
        num_metric_samples = get_num_metric_samples(getattr(args, 'num_metric_samples', None), batch_size, global_size)  # :144
        calc_metrics = bool(getattr(args, 'calc_metrics', False)) and npy_data_validation is not None
        metric_flags = get_compute_metrics_dict(args)
        metric_tap = getattr(args, '_metric_tap', None)      # tests: receives (tag, metrics dict, batches)

        def run_metrics(subset, nsamples, suffix='', tag='loop'):
            kept = [] if metric_tap is not None else None
            m = save_metrics(None, sess, subset, gen_sample, getattr(args, 'metrics_batch_size', 16), global_size, global_step,
                             get_xy_dim(phase, args.start_shape), horovod, False, metric_flags, nsamples, args.data_mean,
                             args.data_stddev, verbose, suffix=suffix, keep=kept)
            if metric_tap is not None:
                metric_tap.append((phase, tag + suffix, m, kept))
            return m

        g_lr0, d_lr0 = scale_lr(args.g_lr, args.d_lr, args.g_scaling, args.d_scaling, horovod, global_size)
        d_lr = ScalarVariable(d_lr0, 'd_lr')
        g_lr = ScalarVariable(g_lr0, 'g_lr')
        optimizer_gen, optimizer_disc = opt.get_optimizer(d_lr, g_lr, args)
        intra_phase_step = ScalarVariable(0, 'step', dtype=np.int64)
        update_intra_phase_step = nops.Op(lambda: intra_phase_step.assign(int(intra_phase_step.value) + batch_size * global_size))
        steps_per_phase = args.mixing_nimg + args.stabilizing_nimg
        # quirk Q2: lr_max is the UNSCALED args.*_lr, so the schedule overwrites the scaled initial value (:164-177)
        update_g_lr = opt.lr_update(g_lr, intra_phase_step, steps_per_phase, args.g_lr, args.g_lr_increase,
                                    args.g_lr_decrease, args.g_lr_rise_niter, args.g_lr_decay_niter)
        update_d_lr = opt.lr_update(d_lr, intra_phase_step, steps_per_phase, args.d_lr, args.d_lr_increase,
                                    args.d_lr_decrease, args.d_lr_rise_niter, args.d_lr_decay_niter)
        if horovod:      # optuna_objective.py:179-186: --use_adasum switches the DISCRIMINATOR's reduction only
            optimizer_gen = parallel.DistributedOptimizer(optimizer_gen)
            optimizer_disc = parallel.DistributedOptimizer(
                optimizer_disc, op=parallel.Adasum if getattr(args, 'use_adasum', False) else parallel.Average)

        alpha = ScalarVariable(1, 'alpha/alpha')
        update_alpha = nops.alpha_update(alpha, args.mixing_nimg, args.starting_alpha, batch_size, global_size)
        prev_vars = var_list
        # variables of earlier phases that this phase no longer has (old to_rgb/from_rgb) leave the store
        with use_store(store):
            tup = opt.optimize_step(optimizer_gen, optimizer_disc, generator, discriminator, real_image_input,
                                    args.latent_dim, alpha, phase, base_shape, args.kernel_spec, args.filter_spec,
                                    args.activation, args.leakiness, args.loss_fn, args.gp_weight, args.optim_strategy,
                                    args.g_clipping, args.d_clipping, args.noise_stddev, prev_vars)
        (train_gen, train_disc, gen_loss, disc_loss, gp_loss, gen_sample, _gg, g_variables, _dg, d_variables, _mg, _md,
         train_gen_freeze, _ggf, _gvf, _mgf, train_disc_freeze, _dgf, _dvf, _mdf) = tup
        graph = tup[0].graph
        var_list = [v.key for v in g_variables] + [v.key for v in d_variables]
        store.drop([k for k in list(store.vars) if k not in set(var_list)])
        if verbose:
            print(f"Generator parameters: {store.count_parameters('generator/')}")
            print(f"Discriminator parameters:: {store.count_parameters('discriminator/')}")
        ema = ExtendedEMA(var_list, decay=args.ema_beta, graph=graph)
        ema_op = ema.apply()

        if phase > args.starting_phase:
            restore_variables(store, phase, args.starting_phase, logdir, args.continue_path, prev_vars, verbose, ema)
        elif args.continue_path and phase == args.starting_phase:
            try:
                restore_variables(store, phase, args.starting_phase, logdir, args.continue_path, var_list, verbose, ema)
            except KeyError:
                restore_variables(store, phase, args.starting_phase, logdir, args.continue_path, prev_vars, verbose, ema)
        if phase < args.starting_phase:
            continue
        alpha.assign(args.starting_alpha if phase == args.starting_phase else 1)
        graph._ensure_flat()
        ema.reset_to_variables()
        if horovod:
            parallel.broadcast_global_variables(store, 0)

        loader = PinnedPrefetcher(npy_data, batch_size, horovod, args.data_mean, args.data_stddev, device)  # quirk Q3
        local_step = 0
        mixing_bool = args.mixing_nimg > 0
        t_phase, imgs, d_loss, g_loss = time.time(), 0, float('nan'), float('nan')
        while True:
            start = time.time()
            d_lr_val = sess.run(update_d_lr)
            g_lr_val = sess.run(update_g_lr)
            if not mixing_bool:
                assert alpha.eval() == 0
            if global_step % args.checkpoint_every_nsteps < (batch_size * global_size) and local_step > 0:
                if horovod:
                    parallel.broadcast_global_variables(store, 0)
                if verbose:
                    save_checkpoint(store, os.path.join(logdir, f'model_{phase}_ckpt_{global_step}'))
            batch = loader.next()
            metrics_summary_bool = local_step % getattr(args, 'metrics_every_nsteps', 128) < batch_size      # :443
            if mixing_bool:      # quirk Q4: previous-phase variables stay frozen while alpha > 0
                train_g, train_d = train_gen_freeze or train_gen, train_disc_freeze or train_disc
            else:
                train_g, train_d = train_gen, train_disc
            want_log = verbose and (local_step // batch_size) % log_every == 0
            if want_log:
                _, _, d_loss_t, g_loss_t = sess.run([train_g, train_d, disc_loss, gen_loss],
                                                    feed_dict={real_image_input: batch})
            else:
                sess.run([train_g, train_d], feed_dict={real_image_input: batch})
            sess.run(ema_op)
            if local_step == batch_size:
                # The phase's graph, variables and closures live until the phase ends: taken out of the cyclic collector's
                # sight, a full collection no longer walks them (~0.1 s of host stall each in a process of this size).
                gc.collect()
                gc.freeze()
            global_step += batch_size * global_size
            local_step += batch_size
            imgs += batch_size * global_size
            if want_log:
                d_loss, g_loss = float(d_loss_t), float(g_loss_t)      # syncs: keep log_every > 1 for speed runs
            end = time.time()
            local_img_s = batch_size / (end - start)
            img_s = global_size * local_img_s
            if mixing_bool:
                sess.run(update_alpha)
            in_phase_step = sess.run(update_intra_phase_step)
            if metrics_summary_bool and calc_metrics:      # :500-507: on the training weights, then on the EMA weights
                run_metrics(npy_data_validation, num_metric_samples)
                sess.run(ema.assign_ema_weights())
                run_metrics(npy_data_validation, num_metric_samples, suffix='_EMA')
                sess.run(ema.restore_original_weights())
            if want_log:
                print_summary_to_stdout(global_step, int(in_phase_step), img_s, local_img_s, d_loss, g_loss,
                                        float(d_lr_val), float(g_lr_val), alpha)
            if mixing_bool and (global_step >= ((phase - args.starting_phase) * (args.mixing_nimg + args.stabilizing_nimg)
                                                + args.mixing_nimg)):
                mixing_bool = False
                alpha.assign(0)
                if verbose:
                    print(f"Begin stabilizing epochs in phase {phase}")
            if global_step >= (phase - args.starting_phase + 1) * (args.stabilizing_nimg + args.mixing_nimg):
                break
            if max_steps_per_phase is not None and local_step // batch_size >= max_steps_per_phase:
                global_step = (phase - args.starting_phase + 1) * (args.stabilizing_nimg + args.mixing_nimg)
                break
        loader.close()
        gc.unfreeze()
        torch.cuda.synchronize()
        stats[phase] = dict(img_s=imgs / max(1e-9, time.time() - t_phase), d_loss=d_loss, g_loss=g_loss,
                            batch_size=batch_size, steps=local_step // batch_size)
        # quirk Q5: the end-of-phase checkpoint holds the EMA weights of G AND D; the next phase starts from them
        sess.run(ema.ema_update_weights())
        if horovod:
            parallel.broadcast_global_variables(store, 0)
        if verbose:
            print(f"Writing final checkpoint file: model_{phase}")
            save_checkpoint(store, os.path.join(logdir, f'model_{phase}'))
        # :593-627: metrics on the whole test / validation (/ training) subsets under the EMA weights.  (The reference runs
        # these loops even with no metric enabled; with none enabled nothing would be computed, so they are skipped here.)
        if npy_data_validation is not None and any(metric_flags.values()):
            if verbose:
                print(f"Computing final metrics for phase {phase} ...")
            sess.run(ema.assign_ema_weights())
            for on, subset, label, tag in ((getattr(args, 'compute_metrics_test', True), npy_data_test, 'Test', 'test'),
                                           (getattr(args, 'compute_metrics_validation', True), npy_data_validation, 'Validation', 'validation'),
                                           (getattr(args, 'compute_metrics_train', False), npy_data_train, 'Training', 'train')):
                if on and subset is not None:
                    t0 = time.time()
                    m = run_metrics(subset, len(subset), tag=tag)
                    if verbose:
                        print(f"Computing metrics on {label.lower()} set took {time.time() - t0} seconds")
                        print(f"{label} dataset metrics:")
                        print(m)
                    stats[phase][f'metrics_{tag}'] = m
            sess.run(ema.restore_original_weights())
        if world > 1:
            torch.distributed.barrier()
    return dict(stats=stats, store=store, logdir=logdir)
