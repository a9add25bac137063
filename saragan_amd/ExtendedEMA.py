"""Mirror of SURFGAN_3D/ExtendedEMA.py: EMA shadows of G and D variables with swap-in / backup / restore.
tf.train.ExponentialMovingAverage without num_updates or zero-debias: shadow -= (1-decay)*(shadow - var).
The shadows live in flat f32 buffers parallel to the parameter buffers, so the update rides in the fused
Adam kernel (sg_adam_ema); `apply()` returns the op the loop runs after the train step
(optuna_objective.py:467) - it only touches ranges the optimiser launch did not already cover."""
import torch

from . import functional as F
from .networks.ops import Op


class ExtendedEMA:
    def __init__(self, var_list, decay, num_updates=None, zero_debias=False, name='ExponentialMovingAverage',
                 graph=None, store=None):
        if num_updates is not None or zero_debias:
            raise NotImplementedError('the reference constructs ExtendedEMA(var_list, decay) only')
        self.var_list = var_list
        self.decay = float(decay)
        self.graph = graph
        self.store = store if store is not None else (graph.store if graph is not None else None)
        if self.store is None:
            raise ValueError('ExtendedEMA needs the StepGraph (graph=) or the VariableStore (store=)')
        self._shadow = {}
        self._backup = {}
        self._done = {}
        if graph is not None:
            graph.ema = self

    def _flat(self, prefix):
        if self.graph is not None:
            self.graph._ensure_flat()
        elif prefix not in self.store.flat:
            self.store.flatten(prefix)
        return self.store.flat[prefix]

    def shadow_flat(self, prefix):
        if prefix not in self._shadow:
            self._shadow[prefix] = self._flat(prefix)['param'].clone()   # initialised to the variables' values
        return self._shadow[prefix]

    def reset_to_variables(self):
        """utils.py:106-115: after a restore the shadows are set to the restored weights."""
        for prefix in ('generator/', 'discriminator/'):
            self.shadow_flat(prefix).copy_(self._flat(prefix)['param'])

    def mark_updated(self, prefix, ranges):
        self._done.setdefault(prefix, []).extend(ranges)

    def apply(self):
        def run():
            for prefix in ('generator/', 'discriminator/'):
                flat = self._flat(prefix)
                sh = self.shadow_flat(prefix)
                done = sorted(self._done.pop(prefix, []))
                pos = 0
                for (o, n) in done + [(flat['total'], 0)]:
                    if o > pos:   # range the optimiser did not touch this step (frozen variables, or no train op)
                        F.adam_ema_(flat['param'][pos:o], None, None, None, sh[pos:o], 0.0, 0.0, 0.0, 1,
                                    ema_decay=self.decay)
                    pos = max(pos, o + n)
        return Op(run, 'ema_apply')

    def average(self, name):
        key = name if isinstance(name, str) else name.key
        prefix = key.split('/')[0] + '/'
        o, n = self._flat(prefix)['offsets'][key]
        return self.shadow_flat(prefix)[o:o + n].view(self.store.vars[key].shape)

    def assign_ema_weights(self):
        def run():
            for prefix in ('generator/', 'discriminator/'):
                flat = self._flat(prefix)
                self._backup[prefix] = flat['param'].clone()
                flat['param'].copy_(self.shadow_flat(prefix))
            # the parameters alias the flat buffer (p.data = flat[...]): writing it moves neither data_ptr nor the version
            # counter the packed weight images are keyed by -- every writer of flat['param'] says so itself
            F.mark_packs_stale()
        return Op(run, 'assign_ema_weights')

    def restore_original_weights(self):
        def run():
            for prefix in ('generator/', 'discriminator/'):
                self._flat(prefix)['param'].copy_(self._backup[prefix])
            F.mark_packs_stale()
        return Op(run, 'restore_original_weights')

    def ema_update_weights(self):
        """tf.group([tf.assign(var, ema.average(var))]) of optuna_objective.py:280-281 (no backup)."""
        def run():
            for prefix in ('generator/', 'discriminator/'):
                self._flat(prefix)['param'].copy_(self.shadow_flat(prefix))
            F.mark_packs_stale()
        return Op(run, 'ema_update_weights')
