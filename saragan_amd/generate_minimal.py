"""Mirror of SURFGAN_3D/generate_minimal.py:13-64 on the HIP path: restore the generator's variables from a checkpoint,
sample `num_samples` volumes in batches of `batch_size`, undo the input normalisation and write
`<output_dir>/generated_images/fake_images_{phase}.npy`.  Same flag names (generate_minimal.py:76-96); the checkpoint is
the `{tf variable name: ndarray}` .npz that saragan_amd.utils.save_checkpoint writes (model_{phase})."""
import argparse
import importlib
import json
import os

import numpy as np
import torch

from . import dataset as data
from .utils import get_base_shape, get_current_input_shape, load_checkpoint
from .varstore import VariableStore, set_compute_dtype, use_store


def _spec_loader(key):
    def load(value):
        with open(value) as f:
            return json.load(f)[key]
    return load


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('architecture', type=str)
    p.add_argument('--start_shape', type=str, required=True)
    p.add_argument('--final_shape', type=str, required=True)
    p.add_argument('--kernel_spec', type=_spec_loader('kernel_spec'), default=None)
    p.add_argument('--filter_spec', type=_spec_loader('filter_spec'), default=None)
    p.add_argument('--network_size', default=None, choices=['xxs', 'xs', 's', 'm', 'l', 'xl', 'xxl'], required=True)
    p.add_argument('--latent_dim', type=int, required=True)
    p.add_argument('--first_conv_nfilters', type=int, default=None)      # obsolete in the reference too (main.py:260-264)
    p.add_argument('--kernel_shape', default=[3, 3, 3])
    p.add_argument('--output_dir', type=str, default=None)
    p.add_argument('--model_path', type=str, required=True)
    p.add_argument('--num_samples', type=int, required=True)
    p.add_argument('--batch_size', default=1, type=int)
    p.add_argument('--phase', type=int, required=True)
    p.add_argument('--activation', type=str, default='leaky_relu')
    p.add_argument('--leakiness', type=float, default=0.2)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--data_mean', default=None, type=float)
    p.add_argument('--data_stddev', default=None, type=float)
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'], help='activation / MFMA input type (new flag)')
    return p


def main(args, device='cuda'):
    """generate_minimal.py:13-64.  Returns the path of the written file."""
    if not torch.cuda.is_available():
        raise RuntimeError('saragan_amd generates on MI355X only: no CPU fallback')
    generator = importlib.import_module(f'saragan_amd.networks.{args.architecture}.generator').generator
    if args.kernel_spec is None or args.filter_spec is None:
        from .networks.pgan.variables import preset_specs
        ks, fs = preset_specs(args.network_size, get_base_shape(args.start_shape), 8)
        args.kernel_spec, args.filter_spec = args.kernel_spec or ks, args.filter_spec or fs
    phase = args.phase
    logdir = os.path.join(args.output_dir or '.', 'generated_images')
    os.makedirs(logdir, exist_ok=True)
    print("Arguments passed:")
    print(args)
    print(f"Saving files to {logdir}")
    set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    store = VariableStore(device, seed=args.seed)
    rng = torch.Generator(device=device).manual_seed(args.seed)
    shape = get_current_input_shape(args.phase, args.batch_size, args.start_shape)
    base_shape = get_base_shape(args.start_shape)
    alpha = 0.0                                  # generate_minimal.py:24-25: tf.Variable(0, name='alpha')

    def sample():
        z = torch.randn(shape[0], args.latent_dim, device=device, generator=rng)
        with use_store(store), torch.no_grad():
            return generator(z, alpha, args.phase, base_shape, activation=args.activation, kernel_spec=args.kernel_spec,
                             filter_spec=args.filter_spec, param=args.leakiness, size=args.network_size, is_reuse=False)

    first = sample()                             # creates the generator's variables (tf.global_variables_initializer)
    print("Restoring variables...")
    sd = load_checkpoint(args.model_path)
    missing = store.load_state_dict({k: v for k, v in sd.items() if k.startswith('generator/')})
    if missing:
        raise KeyError(f'checkpoint {args.model_path} lacks generator variables {missing}')
    from . import functional as F
    F.clear_pack_cache()
    fake_batch = sample().float().cpu().numpy().astype(np.float32)
    del first
    i = 0
    while fake_batch.shape[0] < args.num_samples:
        i += 1
        print(f'Generating sample {i}')
        fake_batch = np.concatenate((fake_batch, sample().float().cpu().numpy().astype(np.float32)))
    print(f'Minimum of generated image 1 before inverting normalization: {np.min(fake_batch[0, ...])}')
    print(f'Maximum of generated image 1 before inverting normalization: {np.max(fake_batch[0, ...])}')
    fake_batch = data.invert_normalize_numpy(fake_batch, args.data_mean, args.data_stddev, True)
    print(f'Minimum of generated image 1 after inverting normalization: {np.min(fake_batch[0, ...])}')
    print(f'Maximum of generated image 1 after inverting normalization: {np.max(fake_batch[0, ...])}')
    out = os.path.join(logdir, f'fake_images_{phase}.npy')
    np.save(out, fake_batch)
    return out


if __name__ == '__main__':
    main(build_parser().parse_args())
