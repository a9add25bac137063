"""Diagnostic: is the step time of bench.py's workload stationary within one process?  Runs the bench step in chunks of 10
with HIP-event timing: 8 chunks back to back, a 2 s idle pause, 8 more chunks, a pause with the caching allocator emptied,
8 more.  Prints ms/step per chunk.  usage: python tools/boost_probe.py   (see DESIGN_NOTES.md section 5)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    device = torch.device('cuda:0')
    torch.cuda.set_device(device)
    cfg = bench.build(args, device, args.dtype)
    sess, ph = cfg['sess'], cfg['ph']
    batches = [bench.synthetic_batch(cfg['shape'], i, device) for i in range(4)]

    def step(i):
        sess.run(cfg['train'], feed_dict={ph: batches[i % 4]})
        sess.run(cfg['ema_op'])

    def chunk(step, n=10):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(n):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n

    for i in range(3):
        step(i)
    state = {}
    def side_stream():
        state['s'] = torch.cuda.Stream()
        with torch.cuda.stream(state['s']):
            state['t'] = torch.zeros(1 << 20, device=device)
        state['s'].synchronize()

    def pinned_copy():
        state['pin'] = torch.empty(cfg['shape'], dtype=torch.float32).pin_memory()
        with torch.cuda.stream(state['s']):
            state['dev'] = state['pin'].to(device, non_blocking=True)
        state['s'].synchronize()

    def feed_from_pinned():        # as the loader leg: every step's input arrives by an async H2D copy on the side stream
        def step2(i):
            with torch.cuda.stream(state['s']):
                x = state['pin'].to(device, non_blocking=True).to(batches[0].dtype)
            torch.cuda.current_stream().wait_stream(state['s'])
            x.record_stream(torch.cuda.current_stream())
            sess.run(cfg['train'], feed_dict={ph: x})
            sess.run(cfg['ema_op'])
        state['step'] = step2

    import ctypes as C
    from saragan_amd import _lib
    lib = _lib.load()

    def prof_on():          # as bench.py's timed loop: per-launch profiling armed, filtered to one (kind, shape)
        lib.sg_prof_enable(1)
        step(0)
        torch.cuda.synchronize()
        ents = (_lib.ProfEntry * 256)()
        n_ent = C.c_int32(0)
        lib.sg_prof_collect(ents, 256, C.byref(n_ent))
        lib.sg_prof_enable(0)
        dom = max((ents[i] for i in range(n_ent.value)), key=lambda e: e.total_ms)
        state['dom'] = dom
        lib.sg_prof_set_filter(dom.kind, C.byref(dom.shape))
        lib.sg_prof_enable(1)

    def prof_off():
        lib.sg_prof_enable(0)
        lib.sg_prof_set_filter(0, None)

    for phase, prep in (('from start', None), ('profiling armed with a shape filter', prof_on), ('profiling off', prof_off),
                        ('after 2 s idle', lambda: time.sleep(2.0)),
                        ('after creating a side stream', side_stream), ('after a pinned H2D copy', pinned_copy),
                        ('inputs by async H2D copies', feed_from_pinned), ('resident inputs again', lambda: state.pop('step'))):
        if prep:
            prep()
        fn = state.get('step', step)
        print(phase, ' '.join('%.1f/%.1f' % chunk(fn) for _ in range(6)), flush=True)


if __name__ == '__main__':
    main()
