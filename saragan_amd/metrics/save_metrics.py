"""Mirror of SURFGAN_3D/metrics/save_metrics.py:19-300 (called from optuna_objective.py:500-506 every
--metrics_every_nsteps images and at :593-627 at the end of a phase): draw real batches from a data subset, generate fake
images with the CURRENT generator weights, evaluate the enabled metrics on the GPU (csrc/metrics.hip through
metrics/swd.py and metrics/skim_metrics.py), average over the batches and print them in the reference's stdout format.

What differs, and why: there is no tf.summary writer (TensorBoard is out of scope: `writer` is accepted and ignored);
MPI.COMM_WORLD.Gather of the ranks' fake images (save_metrics.py:117-131) is torch.distributed.gather to rank 0; FID needs
the Inception graph the reference downloads (metrics/fid_new.py:291-318): --compute_FID is answered with a printed notice at
every evaluation and no 'FID' key."""
import time

import numpy as np
import torch

from ..dataset import normalize_numpy
from . import skim_metrics as SK
from . import swd as SWD


def get_compute_metrics_dict(args):
    """utils.py:267-277."""
    return {k: bool(getattr(args, k, False)) for k in
            ('compute_FID', 'compute_swds', 'compute_ssims', 'compute_psnrs', 'compute_mses', 'compute_nrmses')}


def _gather_to_rank0(fake, global_size):
    import torch.distributed as dist
    if dist.get_rank() == 0:
        parts = [torch.empty_like(fake) for _ in range(global_size)]
        dist.gather(fake, parts, dst=0)
        return torch.cat(parts, dim=0)
    dist.gather(fake, None, dst=0)
    return None


def save_metrics(writer, sess, npy_data, gen_sample, batch_size, global_size, global_step, imagesize_xy, horovod,
                 hyperparam_opt_inter_trial, compute_metrics, num_metric_samples, data_mean, data_stddev, verbose, suffix='',
                 keep=None):
    """Same arguments and return value as the reference (a dict with the keys swd, ssim, psnr, mse, nrmse of the enabled
    metrics).  `keep`: a list that receives (real_batch, fake_batch) of every evaluated batch (tests)."""
    def log(s):
        if verbose:
            print(s)

    compute_metrics = dict(compute_metrics)
    if compute_metrics.get('compute_FID'):      # said every time, never silently dropped: the reference's example scripts pass it
        log('FID is NOT computed: it needs the Inception graph the reference downloads (metrics/fid_new.py:291-318)')
        compute_metrics['compute_FID'] = False
    metrics = {}
    batch_size = min(batch_size, num_metric_samples)                          # save_metrics.py:71
    sample_shape = np.load(npy_data.scratch_files[0], mmap_mode='r').shape    # (the reference reads npy_data.shape)
    compute_metrics['compute_swds'] = imagesize_xy >= 16 and compute_metrics.get('compute_swds', False)       # :78
    compute_metrics['compute_ssims'] = min(sample_shape) >= 16 and compute_metrics.get('compute_ssims', False)  # :79
    rank0 = (not horovod) or hyperparam_opt_inter_trial
    if horovod and not hyperparam_opt_inter_trial:
        import torch.distributed as dist
        rank0 = dist.get_rank() == 0
    swds_local, psnrs_local, mses_local, nrmses_local, ssims_local = [], [], [], [], []
    counter = 0
    while True:
        real_batch = npy_data.batch_mpi(batch_size) if horovod else npy_data.batch(batch_size)      # :94-97
        real_batch = normalize_numpy(real_batch, data_mean, data_stddev, verbose)
        log('Generating fake images for metric computation...')
        start = time.time()
        fake = sess.run(gen_sample).float()
        if horovod and not hyperparam_opt_inter_trial:
            while fake.shape[0] * global_size < batch_size:
                fake = torch.cat((fake, sess.run(gen_sample).float()))
            log(f'Each rank generated {fake.shape[0]} images')
            fake = _gather_to_rank0(fake.contiguous(), global_size)
            if rank0:
                log(f'Gathered a total of {fake.shape[0]} images')
        else:
            while fake.shape[0] < batch_size:
                fake = torch.cat((fake, sess.run(gen_sample).float()))
                log(f'Generated {fake.shape[0]} images')
        if rank0:
            fake = fake[0:batch_size, ...]
            if verbose:
                print(f"Generating fake images took {time.time() - start}")
            real = torch.as_tensor(np.ascontiguousarray(real_batch), device=fake.device)
            real = real[0:fake.shape[0]]        # (a short last draw of a subset smaller than the batch)
            fake = fake[0:real.shape[0]]
            if keep is not None:
                keep.append((real.cpu().numpy(), fake.cpu().numpy()))
            for flag, fn, dest, name in (('compute_swds', SWD.get_swd_for_volumes, swds_local, 'swds'),
                                         ('compute_psnrs', SK.get_psnr, psnrs_local, 'psnrs'),
                                         ('compute_ssims', SK.get_ssim, ssims_local, 'ssims'),
                                         ('compute_mses', SK.get_mean_squared_error, mses_local, 'mses'),
                                         ('compute_nrmses', SK.get_normalized_root_mse, nrmses_local, 'nrmses')):
                if compute_metrics.get(flag):
                    t0 = time.time()
                    dest.append(fn(real, fake))
                    print("%s took %s" % (name, time.time() - t0))
        counter += global_size * batch_size if horovod else batch_size
        if counter >= num_metric_samples:
            break
    if rank0:
        if compute_metrics.get('compute_psnrs'):
            metrics['psnr'] = np.mean(psnrs_local)
            log(f"PSNR: {metrics['psnr']:.4f}")
        if compute_metrics.get('compute_ssims'):
            metrics['ssim'] = np.mean(ssims_local)
            log(f"SSIM: {metrics['ssim']}")
        if compute_metrics.get('compute_mses'):
            metrics['mse'] = np.mean(mses_local)
            log(f"MSE: {metrics['mse']:.4f}")
        if compute_metrics.get('compute_nrmses'):
            metrics['nrmse'] = np.mean(nrmses_local)
            log(f"Normalized Root MSE: {metrics['nrmse']:.4f}")
        if compute_metrics.get('compute_swds'):
            metrics['swd'] = np.array(swds_local).mean(axis=0)
            log(f"SWDS: {metrics['swd']}")
    return metrics
