"""Name-keyed variable store: the eager stand-in for TF1's variable scopes that the reference relies on
(`tf.variable_scope` / `tf.get_variable`, SURFGAN_3D/networks/ops.py:118,131 and pgan/*.py).  Variable names
are the reference's (without the ':0' suffix), so checkpoints, the phase hand-off and
the parameter-count KAT use the same keys.  Parameters are f32 masters; `flatten()` re-homes them in one flat
buffer per network so that the fused Adam/EMA kernel and the gradient all-reduce see contiguous memory."""
import contextlib
from collections import OrderedDict

import torch

_STATE = {'store': None, 'scope': []}
COMPUTE_DTYPE = {'dtype': torch.float32}


def set_compute_dtype(dtype):
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError('compute dtype must be torch.float32 or torch.bfloat16')
    COMPUTE_DTYPE['dtype'] = dtype


def compute_dtype():
    return COMPUTE_DTYPE['dtype']


class VariableStore:
    def __init__(self, device='cuda', seed=0):
        self.device = torch.device(device)
        self.vars = OrderedDict()          # name -> nn.Parameter (f32)
        self.gen = torch.Generator().manual_seed(seed)
        self.flat = {}                     # prefix -> dict(param=, grad=, offsets={name:(off,numel)})

    # -- creation -----------------------------------------------------------------------------------
    def get(self, name, shape, init):
        p = self.vars.get(name)
        if p is not None:
            if tuple(p.shape) != tuple(shape):
                raise ValueError(f'variable {name} exists with shape {tuple(p.shape)}, requested {tuple(shape)}')
            return p
        if init == 'normal':
            t = torch.randn(tuple(shape), generator=self.gen, dtype=torch.float32)
        elif init == 'zeros':
            t = torch.zeros(tuple(shape), dtype=torch.float32)
        else:
            raise ValueError(init)
        p = torch.nn.Parameter(t.to(self.device))
        self.vars[name] = p
        return p

    def names(self, prefix=''):
        return [k for k in self.vars if k.startswith(prefix)]

    def trainable(self, prefix=''):
        """tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, scope=prefix) in creation order."""
        return [(k, v) for k, v in self.vars.items() if k.startswith(prefix)]

    def count_parameters(self, prefix=''):
        return sum(v.numel() for k, v in self.vars.items() if k.startswith(prefix))

    # -- state dict (TF-name keyed) -----------------------------------------------------------------
    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.vars.items())

    def load_state_dict(self, sd, strict=False):
        missing = []
        for k, v in self.vars.items():
            if k in sd:
                with torch.no_grad():
                    v.copy_(torch.as_tensor(sd[k]).to(v.device, torch.float32).reshape(v.shape))
            else:
                missing.append(k)
        if strict and missing:
            raise KeyError(f'missing variables: {missing}')
        return missing

    def drop(self, names):
        for k in names:
            self.vars.pop(k, None)
        self.flat = {}

    # -- flat buffers ---------------------------------------------------------------------------------
    def flatten(self, prefix, order=None):
        """Moves every variable under `prefix` into one flat f32 buffer (16-byte aligned segments, in
        `order`) and gives each a .grad view into a matching flat gradient buffer."""
        names = order if order is not None else self.names(prefix)
        offs, total = OrderedDict(), 0
        for k in names:
            n = self.vars[k].numel()
            offs[k] = (total, n)
            total += (n + 3) // 4 * 4
        flat_p = torch.zeros(total, device=self.device, dtype=torch.float32)
        flat_g = torch.zeros(total, device=self.device, dtype=torch.float32)
        for k in names:
            p = self.vars[k]
            o, n = offs[k]
            flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat_p[o:o + n].view(p.shape)
            p.grad = flat_g[o:o + n].view(p.shape)
        self.flat[prefix] = dict(param=flat_p, grad=flat_g, offsets=offs, total=total)
        return self.flat[prefix]


@contextlib.contextmanager
def use_store(store):
    prev = _STATE['store']
    _STATE['store'] = store
    try:
        yield store
    finally:
        _STATE['store'] = prev


def current_store():
    if _STATE['store'] is None:
        raise RuntimeError('no active VariableStore: wrap network calls in `with use_store(store):`')
    return _STATE['store']


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    """tf.variable_scope(name): nests by '/'.  `reuse` is accepted for signature parity; variables are
    always shared by name (the reference passes is_reuse=True for the 2nd..4th discriminator call)."""
    _STATE['scope'].append(name)
    try:
        yield
    finally:
        _STATE['scope'].pop()


def scope_name():
    return '/'.join(_STATE['scope'])


def get_variable(name, shape, initializer='normal'):
    full = (scope_name() + '/' + name) if _STATE['scope'] else name
    return current_store().get(full, shape, initializer)
