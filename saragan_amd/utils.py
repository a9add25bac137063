"""Mirror of the hot-path helpers of SURFGAN_3D/utils.py: shape/phase arithmetic, learning-rate scaling, the
stdout line format, and checkpoints keyed by the reference's TF variable names (tf.train.Saver stand-in)."""
import ast
import os
import time

import numpy as np
import torch


def parse_tuple(string):
    """utils.py: '(c, z, y, x)' strings from the CLI."""
    if isinstance(string, (tuple, list)):
        return tuple(string)
    s = ast.literal_eval(str(string))
    if not isinstance(s, tuple):
        raise ValueError(f'not a tuple: {string}')
    return s


def get_num_channels(start_shape):
    return parse_tuple(start_shape)[0]


def get_base_shape(start_shape):
    """utils.py:219-224."""
    s = parse_tuple(start_shape)
    return (s[0], s[1], s[2], s[3])


def get_num_phases(start_shape, final_shape):
    """utils.py:211-217."""
    return int(np.log2(parse_tuple(final_shape)[-1] / parse_tuple(start_shape)[-1]))


def get_current_input_shape(phase, batch_size, start_shape):
    """utils.py:163-168."""
    start_shape = parse_tuple(start_shape)
    return [batch_size, get_num_channels(start_shape), *[size * 2 ** (phase - 1) for size in get_base_shape(start_shape)[1:]]]


def get_xy_dim(phase, start_shape):
    """utils.py:188-193."""
    return parse_tuple(start_shape)[-1] * (2 ** (phase - 1))


def scale_lr(g_lr, d_lr, g_scaling, d_scaling, horovod, world_size=1):
    """utils.py:120-150 (hvd.size() -> world_size)."""
    def one(lr, how):
        if how == 'sqrt':
            return lr * np.sqrt(world_size)
        elif how == 'linear':
            return lr * world_size
        elif how == 'none':
            return lr
        raise ValueError(how)
    if horovod:
        g_lr, d_lr = one(g_lr, g_scaling), one(d_lr, d_scaling)
    return g_lr, d_lr


def get_num_metric_samples(num_metric_samples, batch_size, global_size):
    """utils.py:152-161."""
    if not num_metric_samples:
        return batch_size * global_size if batch_size > 1 else 2 * global_size
    return num_metric_samples


def format_summary_line(global_step, in_phase_step, img_s, local_img_s, d_loss, g_loss, d_lr_val, g_lr_val, alpha):
    """The stdout line of utils.py:62-73."""
    current_time = time.strftime("%Y-%m-%d_%H:%M:%S", time.gmtime())
    return (f"{current_time} \t"
            f"Step {global_step:09} \t"
            f"Step(phase) {in_phase_step:09} \t"
            f"img/s {img_s:.2f} \t "
            f"img/s/worker {local_img_s:.3f} \t"
            f"d_loss {d_loss:.4f} \t "
            f"g_loss {g_loss:.4f} \t "
            f"d_lr {d_lr_val:.5f} \t"
            f"g_lr {g_lr_val:.5f} \t"
            f"alpha {alpha:.2f}")


def print_summary_to_stdout(global_step, in_phase_step, img_s, local_img_s, d_loss, g_loss, d_lr_val, g_lr_val, alpha):
    a = float(alpha.eval()) if hasattr(alpha, 'eval') else float(alpha)
    print(format_summary_line(global_step, in_phase_step, img_s, local_img_s, d_loss, g_loss, d_lr_val, g_lr_val, a))


# ---- checkpoints: {tf variable name: ndarray}, the contract of tf.train.Saver(var_list) ---------------------
def save_checkpoint(store, path):
    """tf.train.Saver(var_list).save(sess, path): trainable G+D variables only (no optimiser slots, no EMA)."""
    arrs = {k: v.detach().float().cpu().numpy() for k, v in store.vars.items()}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path if path.endswith('.npz') else path + '.npz', **arrs)


def load_checkpoint(path):
    z = np.load(path if path.endswith('.npz') else path + '.npz')
    return {k: z[k] for k in z.files}


def restore_variables(store, phase, starting_phase, logdir, continue_path, var_list, verbose, ema=None):
    """utils.py:75-118: restore by NAME the variables listed in var_list from model_{phase-1} (or continue_path),
    then set the EMA shadows to the restored values."""
    restore_path = None
    if phase > starting_phase:
        restore_path = os.path.join(logdir, f'model_{phase - 1}')
    elif continue_path and phase == starting_phase:
        restore_path = continue_path
    if verbose:
        print("Restoring variables from:", restore_path)
    sd = load_checkpoint(restore_path)
    names = [v if isinstance(v, str) else getattr(v, 'key', v.name.replace(':0', '')) for v in var_list]
    missing = [n for n in names if n in store.vars and n not in sd]
    if missing:
        raise KeyError(f'checkpoint {restore_path} lacks variables {missing}')
    with torch.no_grad():
        for n in names:
            if n in store.vars:
                store.vars[n].copy_(torch.as_tensor(sd[n]).to(store.vars[n].device).reshape(store.vars[n].shape))
    if ema is not None:
        ema.reset_to_variables()
    if verbose:
        print("Variables restored!")
