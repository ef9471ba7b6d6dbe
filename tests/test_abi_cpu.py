"""CPU tests of the C-ABI boundary: the library builds for gfx950 without a GPU, loads, exports every
symbol include/bhnerf_hip.h declares, and its host-side bookkeeping (parameter layout, packed sizes,
argument validation) agrees with the oracle.  No compute entry point is called (there is no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle_np as onp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as entry
    entry.build()
    from bhnerf_amd import _hip
    return _hip.lib()


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'bhnerf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bhn_\w+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from bhnerf_amd import _hip
    syms = header_symbols()
    assert len(syms) >= 16
    for name in syms:
        assert hasattr(lib, name), name
    assert sorted(_hip.SIGNATURES) == syms        # the ctypes table binds exactly the declared ABI
    assert lib.bhn_version() == _hip.ABI_VERSION == 5


def test_dynamic_symbol_table_is_the_declared_abi_and_nothing_else(lib):
    """The library is built with -fvisibility=hidden: `nm -D --defined-only` lists the entry points of include/bhnerf_hip.h
    (BHN_API) and no C++ internals (round 5 exported 31 mangled launch functions and device stubs beside them, which a second
    library in the process could interpose)."""
    import shutil
    import subprocess
    from bhnerf_amd import _hip
    nm = shutil.which('nm')
    if nm is None:
        pytest.skip('nm not available')
    so = os.path.join(_hip.CSRC, 'libbhnerf_hip.so')
    out = subprocess.run([nm, '-D', '--defined-only', so], capture_output=True, text=True, check=True).stdout
    defined = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert defined == header_symbols(), sorted(set(defined) ^ set(header_symbols()))


def test_tape_info_reports_the_layout_the_kernels_use(lib):
    """bhn_tape_info (host-only): bytes of tape per 32-point group and the layout flags, for the networks the bench line quotes."""
    from bhnerf_amd import _hip
    info = (C.c_int64 * _hip.BHN_TAPE_INFO_N)()
    F = _hip.TAPE_FLAGS

    def q(depth, width, mode, deg=3, groups=8192):
        m = _hip.make_model(depth, width, deg, True, 8.0, 0.0, 8.0, 4.0)
        assert lib.bhn_tape_info(C.byref(m), mode, groups, info, _hip.BHN_TAPE_INFO_N) == 0, lib.bhn_last_error()
        return list(info)

    # 4x256 bf16 (BASELINE config 2): h_2..h_4 + both encoded-input copies + relu bits + e; gA_1, gA_2 + dout; three dW jobs
    fw, cw, cr, dr, flags, nw = q(4, 256, _hip.BHN_BF16)[:6]
    assert flags == F['drop_h1'] | F['drop_ga'] | F['ga0_chain'] and nw == 8
    assert fw == 3 * 8 * 2048 + 2 * 2048 + 4 * 4 * 256 + 128 == 57472            # 1796 B per point
    assert cw == 2 * 8 * 2048 + 128 and cr == 4 * 4 * 256 + 128 + 2048
    assert dr == (16384 + 2048) + (16384 + 16384) + (16384 + 1024 + 16384 + 2048) == 87040      # 2720 B per point
    # 4x128 bf16: the fused backward (12-group tiles for >= 3072 groups per frame)
    fw, cw, cr, dr, flags, nw = q(4, 128, _hip.BHN_BF16)[:6]
    assert flags & F['fused128'] and flags & F['drop_hd'] and nw == 12 and q(4, 128, _hip.BHN_BF16, groups=1460)[5] == 8
    assert fw == 2 * 4 * 2048 + 512 + 2048 + 128 == 19072 and dr == fw + 128     # 596 / 600 B per point
    # f32 keeps everything; depth 6 at width 256: LBITS / DROP_HD are compiled out (round 5's Python re-derivation assumed them on)
    assert q(4, 256, _hip.BHN_F32)[4] == 0
    f6 = q(6, 256, _hip.BHN_BF16)
    assert not f6[4] & (F['lbits'] | F['drop_hd']) and f6[0] == 5 * 8 * 2048 + 2 * 2048 + 6 * 4 * 256 + 128
    assert q(4, 128, _hip.BHN_BF16, deg=5)[4] == F['general']


def test_release_library_has_no_hidden_allocation_or_environment_reads(lib):
    """include/bhnerf_hip.h conventions: the caller owns every buffer and the library keeps no hidden state.  The
    release build must not even REFERENCE an allocator or getenv (the measurement switches live in the debug build,
    include/bhnerf_hip_debug.h), and must not export a debug entry point."""
    import shutil
    import subprocess
    from bhnerf_amd import _hip
    nm = shutil.which('nm')
    if nm is None:
        pytest.skip('nm not available')
    so = os.path.join(_hip.CSRC, 'libbhnerf_hip.so')
    undefined = subprocess.run([nm, '-D', '--undefined-only', so], capture_output=True, text=True, check=True).stdout
    for banned in ('hipMalloc', 'hipFree', 'hipHostMalloc', 'getenv', 'hipStreamSynchronize'):
        assert banned not in undefined, banned
    defined = subprocess.run([nm, '-D', '--defined-only', so], capture_output=True, text=True, check=True).stdout
    assert 'bhn_debug' not in defined
    hdr = open(os.path.join(ROOT, 'include', 'bhnerf_hip.h')).read()
    assert 'bhn_debug' not in hdr


@pytest.mark.parametrize('depth,width,n', [(4, 128, 55169), (4, 256, 208641), (8, 256, 471809), (6, 64, None), (4, 100, None), (4, 7, None)])
def test_param_layout_matches_flax_tree_order(lib, depth, width, n):
    from bhnerf_amd import _hip
    m = _hip.make_model(depth, width, 3, True, 8.0, 0.0, 8.0, 4.0)
    dims = onp.mlp_layer_dims(depth, width, 21)
    total = sum(a * b + b for a, b in dims)
    assert lib.bhn_param_count(C.byref(m)) == total == (n or total)
    ko, bo, ind = (C.c_int64 * (depth + 1))(), (C.c_int64 * (depth + 1))(), (C.c_int32 * (depth + 1))()
    assert lib.bhn_param_layout(C.byref(m), ko, bo, ind) == 0
    off = 0
    for i, (fi, fo) in enumerate(dims):
        assert (ko[i], ind[i]) == (off, fi)
        off += fi * fo
        assert bo[i] == off
        off += fo
    for mode in (0, 1):
        assert lib.bhn_packed_bytes(C.byref(m), mode) > total * (2 if mode else 4)
    ws = lib.bhn_render_bwd_workspace_bytes(C.byref(m), 1, 2, 1000, 0)
    assert ws > lib.bhn_render_bwd_workspace_bytes(C.byref(m), 1, 1, 1000, 0) > 0


def test_argument_validation_reports_errors(lib):
    from bhnerf_amd import _hip
    bad_width = _hip.make_model(4, 600, 3, True, 1.0, 0.0, 1.0, 1.0)
    assert lib.bhn_param_count(C.byref(bad_width)) == -1
    assert b'net_width' in lib.bhn_last_error()
    bad_deg = _hip.make_model(4, 64, 11, True, 1.0, 0.0, 1.0, 1.0)
    assert lib.bhn_param_count(C.byref(bad_deg)) == -1 and b'posenc_deg' in lib.bhn_last_error()
    # posenc_deg 5..10 / net_width 257..512: the general path (csrc/general_mlp.hip) -- same flat parameter layout, a packed
    # image and a workspace of its own per mode; no 8-bit tape
    for general in (_hip.make_model(4, 300, 3, True, 1.0, 0.0, 1.0, 1.0), _hip.make_model(5, 128, 7, True, 1.0, 0.0, 1.0, 1.0)):
        dims = onp.mlp_layer_dims(general.net_depth, general.net_width, 3 + 6 * general.posenc_deg)
        assert lib.bhn_param_count(C.byref(general)) == sum(a * b + b for a, b in dims)
        # the bf16 mode keeps the f32 image (biases, output weights) and adds bf16 fragment images; its tape is half the f32 tape
        assert lib.bhn_packed_bytes(C.byref(general), 1) > lib.bhn_packed_bytes(C.byref(general), 0) > 4 * sum(a * b + b for a, b in dims)
        assert lib.bhn_render_bwd_workspace_bytes(C.byref(general), 0, 2, 1000, 0) > lib.bhn_render_bwd_workspace_bytes(C.byref(general), 1, 2, 1000, 0) > 0
        assert lib.bhn_render_bwd_workspace_bytes(C.byref(general), 2, 2, 1000, 0) == 0 and b'8-bit' in lib.bhn_last_error()
    skip_into_output = _hip.make_model(5, 64, 3, True, 1.0, 0.0, 1.0, 1.0)       # depth 5: concat feeds the output layer
    assert lib.bhn_param_count(C.byref(skip_into_output)) == sum(a * b + b for a, b in onp.mlp_layer_dims(5, 64, 21))
    assert onp.mlp_layer_dims(5, 64, 21)[-1] == (64 + 21, 1)
    too_deep = _hip.make_model(9, 64, 3, True, 1.0, 0.0, 1.0, 1.0)
    assert lib.bhn_param_count(C.byref(too_deep)) == -1 and b'net_depth' in lib.bhn_last_error()
    no_skip = _hip.make_model(5, 64, 3, False, 1.0, 0.0, 1.0, 1.0)
    assert lib.bhn_param_count(C.byref(no_skip)) == sum(a * b + b for a, b in onp.mlp_layer_dims(5, 64, 21, do_skip=False))
    assert lib.bhn_radiative_transfer_fwd(None, None, None, None, None, 1, 1, 1, None) == 1     # BHN_EINVAL, no launch
    with pytest.raises(_hip.HipError):
        _hip.check(lib.bhn_chi2_image(None, None, None, None, 1.0, 0, 1, 1, 1, None, None, None))


def test_tape8_mode_is_bf16_outside_the_backward_and_rejects_other_networks(lib):
    """BHN_BF16_T8 (include/bhnerf_hip.h): every entry point that does not touch the tape treats it as BHN_BF16 (same packed
    image), with or without BHN_T8_CALIBRATE; the backward exists for width 256 / depth >= 3 only and says so."""
    from bhnerf_amd import _hip
    assert _hip.MODES['bf16_t8'] == _hip.BHN_BF16_T8 == 2 and _hip.BHN_T8_CALIBRATE == 0x100
    m256 = _hip.make_model(4, 256, 3, True, 1.0, 0.0, 1.0, 1.0)
    n16 = lib.bhn_packed_bytes(C.byref(m256), _hip.BHN_BF16)
    assert n16 > 0 and lib.bhn_packed_bytes(C.byref(m256), _hip.BHN_BF16_T8) == n16
    assert lib.bhn_packed_bytes(C.byref(m256), _hip.BHN_BF16_T8 | _hip.BHN_T8_CALIBRATE) == n16
    for depth, width in ((4, 128), (2, 256)):
        m = _hip.make_model(depth, width, 3, True, 1.0, 0.0, 1.0, 1.0)
        assert lib.bhn_render_bwd_workspace_bytes(C.byref(m), _hip.BHN_BF16_T8, 8, 1 << 20, 0) == 0
        assert b'BHN_BF16_T8' in lib.bhn_last_error()


def test_device_path_fails_loudly_without_gpu():
    import torch
    from bhnerf_amd import _hip, kgeo, network
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_hip.HipError):
        network.NeRF_Predictor(8.0, 2.0, 8.0, 4.0).engine()
    with pytest.raises(_hip.HipError):
        kgeo.radiative_trasfer(torch.zeros(2, 3, 4), 1.0, 1.0, 1.0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'bhnerf_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f


def test_inline_asm_results_do_not_land_in_live_mfma_sources(lib):
    """tools/check_asm_hazard.py on the built objects: an inline-asm VALU write into a source register of the most recent
    MFMA is invisible to hipcc's hazard recogniser (measured wrong results for SrcB on gfx950)."""
    import importlib.util
    import shutil
    from bhnerf_amd import _hip
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('llvm-objdump not available')
    spec = importlib.util.spec_from_file_location('check_asm_hazard', os.path.join(ROOT, 'tools', 'check_asm_hazard.py'))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    objs = [os.path.join(_hip.CSRC, f) for f in ('fused_fwd.o', 'fused_bwd.o')]
    assert all(os.path.exists(o) for o in objs)
    assert chk.violations(objs) == []
    # the scanner itself: a write into SrcB right after the MFMA is reported, one into an unrelated register is not
    fake = '0000 <k>:\n\tv_mfma_f32_32x32x16_bf16 v[2:17], v[20:23], v[30:33], v[2:17] // 0\n\tv_pk_min_u16 v31, v5, s0 // 1\n\tv_pk_min_u16 v40, v5, s0 // 2\n'
    assert [(w, d) for _, w, d, _ in chk.scan(fake)] == [('B', 1)]
