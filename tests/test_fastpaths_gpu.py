"""The specialised fast paths against the general HIP kernels (and torch fp32 where cheap), on shapes large enough to
engage them (>= 256 tile columns / 512 tiles per launch): every epilogue variant of the sliding-halo forward kernel
incl. the two-pass K split and its fused nearest-x2 gather (tools/check_v3s_variants.py), the streamed kernel's lean
variant (tools/check_v5_variants.py, bit-identical to conv_fwd4 except the pixel-norm rsqrt), the lean weight-gradient
kernel with and without the x2 gather (tools/check_wgrad3l.py).  Each also asserts that the fast kernel actually ran."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')


def _run(name, sg_env):
    spec = importlib.util.spec_from_file_location(name, os.path.join(TOOLS, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        assert mod.main(), f'{name}: see the captured output'
    finally:
        for k in ('SG_FWD_NO_V3S', 'SG_FWD_NO_3P', 'SG_FWD_NO_V5', 'SG_WGRAD_NO_LEAN', 'SG_FWD3S_16'):
            os.environ.pop(k, None)
        sg_env()        # fresh configuration snapshot for the tests that follow


def test_sliding_halo_variants_match_general_kernels(sg_env):
    _run('check_v3s_variants', sg_env)


def test_wave_private_plane_variants_match_sliding_halo_and_torch(sg_env):
    """conv_fwd3w (csrc/conv3w.hip: wave-private halo planes, sliding accumulators, v_mfma_f32_16x16x32_bf16) in every epilogue
    variant, forward and data gradient, ragged H, odd D, 64 output channels, batch 1-2 (columns cut into D segments), against the
    sliding-halo kernel (SG_FWD3S_16=0) and torch fp32."""
    _run('check_v3w_variants', sg_env)


def test_streamed_lean_variants_match_conv_fwd4(sg_env):
    _run('check_v5_variants', sg_env)


def test_lean_wgrad_matches_general_and_torch(sg_env):
    _run('check_wgrad3l', sg_env)
