"""Diagnostic: which Python reference cycles a training step leaves behind (they keep device tensors alive until the
cyclic collector runs; Session.run frees them deterministically).  usage on the GPU box: python tools/cycle_probe.py"""
import collections
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0], '--batch', os.environ.get('PROBE_BATCH', '8')]
    args = bench.parse()
    device = torch.device('cuda', 0)
    cfg = bench.build(args, device, args.dtype)
    sess, ph = cfg['sess'], cfg['ph']
    batch = bench.synthetic_batch(cfg['shape'], 0, device)

    def step():
        sess.run(cfg['train'], feed_dict={ph: batch})
        sess.run(cfg['ema_op'])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.collect()
    base = torch.cuda.memory_allocated()
    gc.disable()
    step()
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated()
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    gc.set_debug(0)
    print(f'allocated: {base / 2**20:.1f} MiB before the step, {after / 2**20:.1f} MiB after it (collector off), '
          f'{n} unreachable objects')
    hist = collections.Counter(type(o).__module__ + '.' + type(o).__qualname__ for o in gc.garbage)
    for k, v in hist.most_common(40):
        print(f'  {v:6d}  {k}')
    tens = [o for o in gc.garbage if torch.is_tensor(o)]
    tot = sum(t.numel() * t.element_size() for t in tens if t.is_cuda)
    print(f'{len(tens)} tensors in cycles, {tot / 2**20:.1f} MiB on the device (views counted with their bases)')
    ids = {id(o) for o in gc.garbage}
    shown = 0
    for o in gc.garbage:
        mod = type(o).__module__
        if mod.startswith('saragan_amd') or 'Backward' in type(o).__qualname__:
            refs = [type(r).__module__ + '.' + type(r).__qualname__ for r in gc.get_referents(o) if id(r) in ids]
            print('  ', mod + '.' + type(o).__qualname__, '->', collections.Counter(refs).most_common(6))
            shown += 1
            if shown > 60:
                break
    gc.garbage.clear()
    gc.enable()


if __name__ == '__main__':
    main()
