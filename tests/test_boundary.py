"""CPU tests of the drop-in boundary: the C structs are declared three times (include/saragan_hip.h, the ctypes
binding saragan_amd/_lib.py, and the reference-side stub shown in INTEGRATION.md) and must agree field for field;
the library rejects a stale sg_conv_epilogue; the reference loop's module paths resolve through saragan_amd/dropin."""
import ctypes as C
import importlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SHIMMED = ('optimization', 'dataset', 'ExtendedEMA', 'utils')


def _c_struct_fields(header, name):
    m = re.search(r'typedef struct \{((?:(?!typedef struct).)*?)\}\s*' + name + r'\s*;', header, re.S)
    assert m, name
    body = re.sub(r'/\*.*?\*/', '', m.group(1), flags=re.S)
    out = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(',')
        out.append(re.sub(r'[\[\]\d\*]', '', names[0].split()[-1]))
        for extra in names[1:]:
            out.append(re.sub(r'[\[\]\d\*\s]', '', extra))
    return out


def _py_struct_fields(text, cls):
    m = re.search(r'class ' + cls + r'\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n', text, re.S)
    assert m, cls
    if 'for k in (' in m.group(1):
        return re.findall(r"'(\w+)'", m.group(1).split('for k in')[1])
    return re.findall(r"\('(\w+)',", m.group(1))


def test_struct_declarations_agree():
    hdr = open(os.path.join(ROOT, 'include', 'saragan_hip.h')).read()
    lib_py = open(os.path.join(ROOT, 'saragan_amd', '_lib.py')).read()
    integ = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    ep = _c_struct_fields(hdr, 'sg_conv_epilogue')
    assert ep[0] == 'struct_size' and len(ep) == 31, ep
    assert _py_struct_fields(lib_py, 'ConvEpilogue') == ep
    assert _py_struct_fields(integ, 'SgConvEpilogue') == ep, 'INTEGRATION.md stub is out of sync with the header'
    shp = _c_struct_fields(hdr, 'sg_conv_shape')
    assert _py_struct_fields(lib_py, 'ConvShape') == shp
    assert _py_struct_fields(integ, 'SgConvShape') == shp
    from saragan_amd import _lib
    # natural C layout: u32 +pad, ptr, i32, f32, i32, f32, ptr, ptr, f32 +pad, ptr, i32, i32[3], i32[3], i32, ptr, size_t, i32 +pad, ptr, ptr, ptr, f32, f32
    assert C.sizeof(_lib.ConvEpilogue) == 224 and C.sizeof(_lib.ConvShape) == 40
    assert _lib.ConvEpilogue(None, 1, 0.2).struct_size == 224
    # every entry point INTEGRATION.md names is declared by the header
    named = set(re.findall(r'`(sg_[a-z0-9_]+)', integ))
    declared = set(re.findall(r'\b(sg_[a-z0-9_]+)\s*\(', hdr)) | {'sg_conv_epilogue', 'sg_conv_shape'}
    missing = {n for n in named if n not in declared and not any(d.startswith(n) for d in declared)}
    assert not missing, missing


def test_stale_epilogue_is_rejected_without_a_gpu():
    """Argument validation runs before anything touches the device: a struct built against an older header (shorter,
    no struct_size) comes back SG_EINVAL instead of being read past its end."""
    from saragan_amd import _lib
    lib = _lib.load()
    shp = _lib.ConvShape(1, 1, 4, 4, 16, 16, 1, 3, 3, 0)
    ep = _lib.ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, None, 0.0, None)
    ep.struct_size = 68          # the 9-field struct INTEGRATION.md showed in round 1
    buf = (C.c_char * 4096)()
    addr = (C.addressof(buf) + 15) & ~15
    rc = lib.sg_conv3d_fwd(addr, addr, addr, C.byref(shp), C.byref(ep), _lib.SG_F32, None)
    assert rc == -1 and b'invalid' in lib.sg_error_string(rc)
    assert lib.sg_config_reload() == 0
    assert lib.sg_optim_step(7, addr, addr, None, None, None, 16, 0.1, 0.0, 0.0, 0, 1.0, 0.99, None) == -1


def _forget_shims():
    for k in [k for k in sys.modules if k == 'networks' or k.startswith('networks.') or k in _SHIMMED]:
        del sys.modules[k]


def test_reference_module_paths_resolve_through_dropin():
    import saragan_amd
    path = saragan_amd.dropin_path()
    sys.path.insert(0, path)
    try:
        _forget_shims()
        for arch in ('pgan', 'pgandeep'):
            g = importlib.import_module(f'networks.{arch}.generator').generator          # optuna_objective.py:64-65
            d = importlib.import_module(f'networks.{arch}.discriminator').discriminator
            assert callable(g) and callable(d)
            assert g.__module__ == f'saragan_amd.networks.{arch}.generator'
        opt = importlib.import_module('optimization')
        assert {'get_optimizer', 'minimize_with_clipping', 'optimize_step', 'lr_update'} <= set(dir(opt))
        ops = importlib.import_module('networks.ops')
        for name in ('conv3d', 'dense', 'apply_bias', 'act', 'leaky_relu', 'pixel_norm', 'upscale3d', 'downscale3d',
                     'to_rgb', 'from_rgb', 'minibatch_stddev_layer', 'alpha_update', 'num_filters', 'get_weight',
                     'calculate_gain'):
            assert hasattr(ops, name), name
        loss = importlib.import_module('networks.loss')
        assert {'forward_simultaneous', 'forward_discriminator', 'forward_generator'} <= set(dir(loss))
        assert hasattr(importlib.import_module('dataset'), 'NumpyPathDataset')
        assert hasattr(importlib.import_module('ExtendedEMA'), 'ExtendedEMA')
        assert hasattr(importlib.import_module('utils'), 'get_num_phases')
    finally:
        sys.path.remove(path)
        _forget_shims()
