"""bench.py: the roofline object of the dominant kernel (HIP-event durations from the library's launch profiler, sg_prof_*; HBM
traffic from the committed counter passes)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK = 8.0e12      # bytes/s, MI355X_MICROARCH.md


def pmc_traffic(entry, dtype):
    """Bytes per launch that crossed the L2's memory side for this (kind, shape), from the committed rocprofv3 counter
    passes (profiles/r05_pmc_traffic.json, else r04 / r03 / r02 / r01: FETCH_SIZE x2 + WRITE_SIZE; tools/pmc_probe.py +
    tools/pmc_summary.py; the counters cannot be read from inside this process).  Mean over the epilogue variants
    measured; None when this shape / batch / dtype was not part of the counter run."""
    for name in ('r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(path):
            continue
        tab = json.load(open(path))
        if tab.get('dtype') != dtype:
            continue
        s = entry.shape
        key = dict(n=s.n, d=s.d, h=s.h, w=s.w, cin=s.cin, cout=s.cout, k=[s.kd, s.kh, s.kw])
        kind = 'fwd' if entry.kind == 0 else 'wgrad'
        hits = [e['traffic_bytes'] for e in tab['entries'] if e['kind'] == kind and e['shape'] == key and
                e.get('variant', '').startswith(('fwd bias', 'wgrad', 'bias', 'with'))]
        if hits and not s.upsample_in:
            return round(sum(hits) / len(hits))
    return None


def sustained_mfma_peak(dtype, kernel=''):
    """TFLOP/s this board SUSTAINS on bare v_mfma_f32_32x32x16_bf16 with random operands (register-only loop, all 256 CUs,
    6 s: tools/probe/mfma_ceiling.hip, committed as profiles/r04_mfma_ceiling.txt with the clock, power and power cap
    beside it): the chip lowers its clock under MFMA load, so the 2.5 PFLOP/s spec peak is not reachable by ANY kernel on
    random data.  None for f32 (the f32 MFMA runs at the vector rate and is not clock-limited the same way) or when the
    file is not there."""
    if dtype != 'bf16':
        return None
    path = os.path.join(ROOT, 'profiles', 'r04_mfma_ceiling.txt')
    if not os.path.exists(path):
        return None
    # kernels on v_mfma_f32_16x16x32_bf16 (conv_fwd3w, conv_fwd3p16) are priced against THAT shape's ceiling: the same loop
    # sustains 2 005 TFLOP/s on it (the board holds a higher clock), 1 848 on 32x32x16
    m16 = 'conv_fwd3w' in kernel or 'conv_fwd3p16' in kernel
    for ln in open(path):
        if m16 and ln.startswith('16x16x32 random, 1 wave/SIMD'):
            return float(ln.split('last second')[1].split()[0])
        if not m16 and ln.startswith('SUSTAINED_PEAK_32x32x16_TFLOPS'):
            return float(ln.split()[1])
    return None


def collect(lib):
    """The launch profiler's table since the last call (csrc/prof.hip), heaviest first."""
    import ctypes as C
    from saragan_amd import _lib
    ents = (_lib.ProfEntry * 256)()
    n_ent = C.c_int32(0)
    lib.sg_prof_collect(ents, 256, C.byref(n_ent))
    return sorted((ents[i] for i in range(n_ent.value)), key=lambda e: -e.total_ms)


def dominant(table):
    """The device kernel (by name) with the largest summed time over the calibration steps, its heaviest (kind, shape) entry,
    and the per-kernel sums."""
    by_kernel = {}
    for e in table:
        by_kernel[e.kernel] = by_kernel.get(e.kernel, 0.0) + e.total_ms
    name = max(by_kernel, key=by_kernel.get) if by_kernel else None
    return name, next((e for e in table if e.kernel == name), None), by_kernel


def spec_peak(dtype):
    return 2500.0 if dtype == 'bf16' else 157.3      # dense MFMA TFLOP/s, MI355X_MICROARCH.md


def _shape_dict(s, with_upsample=True):
    d = dict(n=s.n, d=s.d, h=s.h, w=s.w, cin=s.cin, cout=s.cout, k=[s.kd, s.kh, s.kw])
    if with_upsample:
        d['upsample_in'] = s.upsample_in
    return d


def roofline_object(timed, table, dom_name, dtype, hipgraph, ncal):
    """`roofline` of the JSON line: the dominant kernel's heaviest shape.  `timed`: the profiler's entries of the timed region (only
    the dominant (kind, shape) was bracketed there); `table`: the calibration steps' full table."""
    peak = spec_peak(dtype)
    timed = [e for e in timed if e.kernel == dom_name] or timed       # (one shape may run as several kernel variants)
    timing_note = 'HIP events around every launch of this (kernel, shape) inside the timed region'
    if hipgraph:      # the timed region replayed a hipGraph: the dominant kernel's duration comes from the eager calibration steps
        timed = sorted((e for e in table if e.kernel == dom_name and e.launches > 0), key=lambda r: -r.total_ms)
        timing_note = ('HIP events around every launch during the two eager calibration steps just before the timed region (the '
                       'timed region replays the step as ONE hipGraph: there is no launch to bracket)')
    if not timed or timed[0].launches <= 0:
        return None
    best = timed[0]
    avg_ms = best.total_ms / best.launches
    ach = best.flops_per_launch / (avg_ms * 1e-3) / 1e12
    s = best.shape
    alg_bytes = int(s.n * s.d * s.h * s.w * (s.cin / (8 if s.upsample_in else 1) + s.cout) * (2 if dtype == 'bf16' else 4))
    # which roof bounds this (kernel, shape): its arithmetic intensity against the machine balance (peak FLOP/s over
    # 8 TB/s of HBM).  The small-channel 2-D layers of configs[4] sit below it and are priced in bytes.
    if best.flops_per_launch / alg_bytes < peak * 1e12 / HBM_PEAK:
        gbs = alg_bytes / (avg_ms * 1e-3) / 1e9
        roof = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK / 1e9, unit='GB/s', frac=round(gbs * 1e9 / HBM_PEAK, 4))
    else:
        roof = dict(bound='mfma', achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4))
        sp = sustained_mfma_peak(dtype, best.kernel.decode())
        if sp:      # what the board sustains on bare MFMAs with random operands (profiles/r04_mfma_ceiling.txt)
            roof.update(sustained_peak=sp, frac_of_sustained=round(ach / sp, 4))
    roof.update(traffic=pmc_traffic(best, dtype), kernel=best.kernel.decode(), shape=_shape_dict(s),
                launches=int(best.launches), avg_ms=round(avg_ms, 4), flops_per_launch=best.flops_per_launch,
                algorithmic_bytes=alg_bytes, timing=timing_note)
    # the same kernel on its other shapes (calibration-step timings): the object above quotes the heaviest one by summed time
    others = sorted((e for e in table if e.kernel == dom_name and e.launches > 0), key=lambda r: -r.total_ms)[:5]
    roof['by_shape'] = [dict(n=e.shape.n, cin=e.shape.cin, cout=e.shape.cout, calls_per_step=round(e.launches / ncal, 1),
                             avg_ms=round(e.total_ms / e.launches, 4),
                             achieved=round(e.flops_per_launch / (e.total_ms / e.launches * 1e-3) / 1e12, 1),
                             frac=round(e.flops_per_launch / (e.total_ms / e.launches * 1e-3) / 1e12 / peak, 4)) for e in others]
    return roof


def small_channel_object(rows, dtype):
    """`roofline_hbm`: the bandwidth-bound kernels of the step priced in bytes (calibration-step timings; configs[4]'s 4-16-channel
    layers): the heaviest (kind, shape) of the small-channel family, algorithmic bytes = one read of x and one read / write of y."""
    peak = spec_peak(dtype)
    es = 2 if dtype == 'bf16' else 4
    small = [e for e in rows if e.kernel.startswith(b'conv_small') and e.launches > 0 and      # ... those below the machine balance
             e.flops_per_launch / (e.shape.n * e.shape.d * e.shape.h * e.shape.w * (e.shape.cin + e.shape.cout) * es) < peak * 1e12 / HBM_PEAK]
    if not small:
        return None
    e = max(small, key=lambda r: r.total_ms)
    s_ = e.shape
    nbytes = int(s_.n * s_.d * s_.h * s_.w * (s_.cin + s_.cout) * es)
    avg = e.total_ms / e.launches
    gbs = nbytes / (avg * 1e-3) / 1e9
    return dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK / 1e9, unit='GB/s', frac=round(gbs * 1e9 / HBM_PEAK, 4),
                traffic=pmc_traffic(e, dtype), kernel=e.kernel.decode(), kind='fwd' if e.kind == 0 else 'wgrad',
                shape=_shape_dict(s_, False), launches=int(e.launches), avg_ms=round(avg, 4), algorithmic_bytes=nbytes,
                note='the small-channel VALU kernels (csrc/small.hip), timed during the calibration steps')


def dump_table(rows, ncal):
    """--dump-prof: the per-shape conv kernel table on stderr."""
    for e in rows:
        s_ = e.shape
        print(f"{'fwd ' if e.kind == 0 else 'wgrd'} {e.kernel.decode():28s} n{s_.n} {s_.d}x{s_.h}x{s_.w} {s_.cin:4d}->{s_.cout:4d} "
              f"k{s_.kd}{s_.kh}{s_.kw} ups{s_.upsample_in} calls/step {e.launches / ncal:5.1f} "
              f"avg {e.total_ms / e.launches * 1e3:8.1f} us ms/step {e.total_ms / ncal:7.3f} "
              f"TF/s {e.flops_per_launch / (e.total_ms / e.launches) / 1e9:7.1f}", file=sys.stderr)
