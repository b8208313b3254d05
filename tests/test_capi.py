"""C-ABI surface checks that need no GPU: the library loads, exports every symbol the header
declares, validates arguments before touching the device, and the package fails loudly
(instead of falling back) when the library or the GPU is missing."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'dhd_amd.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|size_t)\s+(dhd_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_table_agree():
    from dhd_amd import _lib
    assert declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    from dhd_amd import _lib
    lib = _lib.load()
    assert os.path.samefile(_lib.LIB_PATH, os.path.join(ROOT, 'dhd_amd', 'csrc', 'libdhd_amd.so'))
    for name in declared_symbols():
        assert getattr(lib, name) is not None, name
    assert lib.dhd_abi_version() == _lib.ABI_VERSION
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (dhd_[a-z0-9_]+)', out))
    assert exported == set(declared_symbols()), exported ^ set(declared_symbols())


def test_sfa_stage_shape_support_and_validation():
    """dhd_sfa_stage_*: supported channel counts, sizes, and argument checks that run before any launch."""
    from dhd_amd import _lib
    lib = _lib.load()
    assert lib.dhd_sfa_stage_supported(256, 40000) == 1 and lib.dhd_sfa_stage_supported(128, 400) == 1
    assert lib.dhd_sfa_stage_supported(512, 40000) == 1
    assert lib.dhd_sfa_stage_supported(64, 40000) == 0 and lib.dhd_sfa_stage_supported(256, 402) == 0
    # a sample is addressed through 32-bit buffer offsets: 2*C*hw*4 bytes must stay below 4 GiB
    assert lib.dhd_sfa_stage_supported(512, 1 << 21) == 0 and lib.dhd_sfa_stage_supported(512, (1 << 20) - 4) == 1
    assert lib.dhd_sfa_stage_saved_bytes(4, 64, 40000, 8) == 0
    saved = lib.dhd_sfa_stage_saved_bytes(4, 256, 40000, 32)
    # y1 + y2 (2 x 164 MB) + one pass bit per activation (5 MB) + small tables
    assert 2 * 4 * 256 * 40000 * 4 < saved < 2 * 4 * 256 * 40000 * 4 + 8e6
    assert lib.dhd_sfa_stage_scratch_bytes(4, 256, 40000, 32) > 3 * 4 * 256 * 40000 * 4
    w, g = _lib.SfaWeights(), _lib.SfaGrads()
    one = C.c_void_p(16)
    assert lib.dhd_sfa_stage_forward(None, C.byref(w), one, one, one, 4, 256, 40000, None) == -1
    assert lib.dhd_sfa_stage_forward(one, C.byref(w), one, one, one, 4, 64, 40000, None) == -3   # unsupported C
    w.hidden = 32
    assert lib.dhd_sfa_stage_forward(one, C.byref(w), one, one, one, 4, 256, 40000, None) == -1  # null weights
    assert lib.dhd_sfa_stage_backward(one, C.byref(w), one, one, one, C.byref(g), one, 4, 256, 40000, None) == -1
    # the GEMM precision is a per-call field (dhd_sfa_weights.gemm), not a process-wide switch
    assert not hasattr(lib, 'dhd_sfa_set_gemm_mode') and not hasattr(lib, 'dhd_mghs_set_deterministic')
    assert _lib.SFA_GEMM == {'default': 0, 'bf16x6': 1, 'f32': 2, 'bf16x3': 3}


def test_library_is_a_gfx950_code_object():
    from dhd_amd import _lib
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob and b'gfx90a' not in blob and b'sm_' not in blob


def test_argument_validation_happens_on_the_host():
    from dhd_amd import _lib
    lib = _lib.load()
    n, m = C.c_size_t(0), C.c_size_t(0)
    d = _lib.MghsDesc()
    assert lib.dhd_mghs_workspace_bytes(C.byref(d), C.byref(n), C.byref(m)) == -1  # all-zero desc
    d.batch, d.n_cams, d.n_depth, d.fh, d.fw, d.channels, d.n_grids = 1, 6, 44, 16, 44, 64, 4
    for g, nz in zip(range(4), (1, 4, 4, 8)):
        d.grid[g].n[0], d.grid[g].n[1], d.grid[g].n[2] = 200, 200, nz
    assert lib.dhd_mghs_workspace_bytes(C.byref(d), C.byref(n), C.byref(m)) == 0
    state, scratch = n.value, m.value
    # state (what backward needs, held by autograd): slot prefix (V words) + slot -> voxel (2P) + per-point slots (2P) = 5.7 MB
    assert 5.0e6 < state < 6.5e6
    # scratch (shared per stream): counters / keys / sorted lists (~2V + 10P words = 13 MB) + the compact table sized for the
    # most slots the index rules allow: min(V0, P) + min(V1 + V2 + V3, P) = 40 000 + 185 856 rows x 256 B = 58 MB
    assert 6.5e7 < scratch < 8.0e7
    d.batch = 4
    assert lib.dhd_mghs_workspace_bytes(C.byref(d), C.byref(n), C.byref(m)) == 0
    assert 3.9 * state < n.value < 4.1 * state and 3.9 * scratch < m.value < 4.1 * scratch
    d.n_grids = 5
    assert lib.dhd_mghs_workspace_bytes(C.byref(d), C.byref(n), C.byref(m)) == -1
    d.n_grids = 4
    d.flags = 8                                                                     # unknown flag bit (1, 2, 4 are defined)
    assert lib.dhd_mghs_workspace_bytes(C.byref(d), C.byref(n), C.byref(m)) == -1
    d.flags = _lib.MGHS_DETERMINISTIC | _lib.MGHS_FEAT_GRAD_NCHW
    assert lib.dhd_mghs_workspace_bytes(C.byref(d), C.byref(n), C.byref(m)) == 0
    assert lib.dhd_mghs_prepare(C.byref(d), None, None, None, None) == -1
    assert lib.dhd_mghs_lift(C.byref(d), None, None, 65, None, None, None, None, None, None, None) == -1
    assert lib.dhd_hbm_calibrate(None, 1024, 0, None) == -1 and lib.dhd_hbm_calibrate(C.c_void_p(4096), 1000, 1, None) == -1
    assert lib.dhd_bev_pool_v2_forward(None, None, None, None, None, None, None, None, 64, 10, None) == -1
    assert lib.dhd_bev_pool_v2_forward(None, None, None, None, None, None, None, None, 64, 0, None) == 0  # nothing to do
    # fused operator: sizes on the host, shapes it does not take, missing workspace
    fs, fc = C.c_size_t(), C.c_size_t()
    assert lib.dhd_bev_pool_v2_fused_workspace_bytes(64, 4, 1, 200, 200, 81205, C.byref(fs), C.byref(fc)) == 0
    assert fs.value >= 4 * (160001 + 81206) and fs.value % 256 == 0 and fc.value >= 4 * (2 * 160000 + 64 * 81206)
    assert lib.dhd_bev_pool_v2_fused_workspace_bytes(80, 4, 1, 200, 200, 10, C.byref(fs), C.byref(fc)) == -1      # C != 64
    assert lib.dhd_bev_pool_v2_fused_workspace_bytes(64, 4, 1, 202, 200, 10, C.byref(fs), C.byref(fc)) == -3     # Dy % 4
    assert lib.dhd_bev_pool_v2_fused_workspace_bytes(64, 4, 1, 200, 260, 10, C.byref(fs), C.byref(fc)) == -3     # Dx > 256
    assert lib.dhd_bev_pool_v2_fused_workspace_bytes(64, 4, 1, 200, 200, 10, None, None) == -1
    assert lib.dhd_bev_pool_v2_fused_forward(*([None] * 8), 64, 10, 4, 1, 200, 200, None, 0, 0, None, 0, None) == -1
    assert lib.dhd_bev_pool_v2_fused_backward(*([None] * 10), 64, 10, 10, 4, 1, 200, 200, None, 0, None, 0, None) == -1
    assert lib.dhd_sfa_channel_mean(None, None, 1, 512, 40000, None) == -1
    assert lib.dhd_height_band(None, 6, 65, 16, 44, None, None, None, None) == -1
    assert lib.dhd_ema_update(None, None, None, 5, 0.5, 0.5, None) == -1
    assert lib.dhd_bn_supported(1, 24, 64, 128 * 352) == 1 and lib.dhd_bn_supported(1, 4, 64, 2500) == 0 and lib.dhd_bn_supported(0, 4, 64, 2500) == 1
    assert lib.dhd_bn_supported(0, 70000, 1, 64) == 0 and lib.dhd_bn_workspace_bytes(24, 64, 45056) == (24 * 16 * 2 * 64 + 3 * 64) * 4
    assert lib.dhd_bn_train_forward(None, 0, 2, 8, 64, None, None, None, None, 0.1, 1e-5, None, None, None, None, None) == -1
    assert lib.dhd_ema_update(None, None, None, 0, 0.5, 0.5, None) == 0  # empty state
    d.batch = 1 << 20
    assert lib.dhd_mghs_workspace_bytes(C.byref(d), C.byref(n), C.byref(m)) == -3  # beyond the int32 index space


def test_no_silent_fallback_without_gpu_or_library():
    from dhd_amd import _lib, bev_pool_v2, SFA
    from dhd_amd import mghs_op
    z = torch.zeros(0, dtype=torch.int32)
    with pytest.raises(_lib.DhdError):
        bev_pool_v2(torch.rand(1, 1, 2, 2, 2), torch.rand(1, 1, 2, 2, 4), z, z, z, (1, 1, 3, 3, 4), z, z)
    with pytest.raises(_lib.DhdError):
        SFA(32, 16)(torch.rand(1, 32, 5, 5))
    with pytest.raises(_lib.DhdError):
        mghs_op.height_band(torch.rand(2, 65, 4, 11), [0.1 * i for i in range(65)], [0, 1, 2, 3])
    from dhd_amd import label_loss, occ_loss
    from dhd_amd.depthnet import _DeformIm2col
    with pytest.raises(_lib.DhdError):
        occ_loss.occ_losses(torch.rand(10, 18), torch.zeros(10, dtype=torch.uint8), torch.ones(10, dtype=torch.uint8), torch.ones(18))
    with pytest.raises(_lib.DhdError):
        occ_loss.occ_argmax_hist(torch.rand(10, 18))
    with pytest.raises(_lib.DhdError):
        label_loss.bin_labels(torch.rand(1, 2, 32, 32), torch.rand(1, 2, 32, 32), 16, [1.0, 45.0, 1.0], 44, -1.0, 0.1, 65)
    with pytest.raises(_lib.DhdError):
        label_loss.points_to_maps(torch.rand(2, 100, 4), 64, 176)
    with pytest.raises(_lib.DhdError):
        _DeformIm2col.apply(torch.rand(1, 4, 5, 5), torch.zeros(1, 18, 5, 5), 3, 1, 1)
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['DHD_AMD_LIB'] = '/nonexistent/libdhd_amd.so'\n"
            "from dhd_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.DhdError as e:\n    print('LOUD', e)\n" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert 'LOUD' in out.stdout and 'no CPU or PyTorch fallback' in out.stdout


def test_product_code_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, 'dhd_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(base, f)).read()
                assert 'oracle' not in text.replace('oracle-free', ''), os.path.join(base, f)


def test_workspace_alignment_is_checked_on_the_host():
    """The MGHS kernels access the carved workspace arrays as 16-byte vectors: a misaligned base is refused."""
    from dhd_amd import _lib
    lib = _lib.load()
    d = _lib.MghsDesc()
    d.batch, d.n_cams, d.n_depth, d.fh, d.fw, d.channels, d.n_grids = 1, 1, 4, 4, 11, 64, 1
    d.grid[0].n[0], d.grid[0].n[1], d.grid[0].n[2] = 8, 8, 1
    n, m = C.c_size_t(0), C.c_size_t(0)
    assert lib.dhd_mghs_workspace_bytes(C.byref(d), C.byref(n), C.byref(m)) == 0 and n.value > 0 and m.value > 0
    cal = _lib.Calib()
    for f, _ in _lib.Calib._fields_:
        setattr(cal, f, 0x20000)
    for state, scratch, rc in ((0x10004, 0x40000, -1), (0x10000, 0x40004, -1)):
        ws = _lib.MghsWorkspace(state, n.value, scratch, m.value)
        assert lib.dhd_mghs_prepare(C.byref(d), C.byref(cal), None, C.byref(ws), None) == rc
    ws = _lib.MghsWorkspace(0x10000, n.value - 1, 0x40000, m.value)       # too small
    assert lib.dhd_mghs_prepare(C.byref(d), C.byref(cal), None, C.byref(ws), None) == -2


def _build_c_example(tmp_path):
    """examples/capi_kat.c with plain gcc (C11): the header is C, the only other dependency is the HIP runtime's host API."""
    exe = str(tmp_path / 'capi_kat')
    lib_dir = os.path.join(ROOT, 'dhd_amd', 'csrc')
    cmd = ['gcc', '-std=c11', '-O2', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', os.path.join(ROOT, 'examples', 'capi_kat.c'),
           '-I' + os.path.join(ROOT, 'include'), '-I/opt/rocm/include', '-L' + lib_dir, '-ldhd_amd', '-L/opt/rocm/lib', '-lamdhip64', '-lm',
           '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,/opt/rocm/lib', '-o', exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return exe


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/dhd_amd.h compiles as strict C99 (no C++ or torch types anywhere in the ABI) and a C host program written
    against it links to libdhd_amd.so with gcc alone -- the form a binding in the reference's own extension style would
    take."""
    src = tmp_path / 'h.c'
    src.write_text('#include "dhd_amd.h"\nint main(void) { return dhd_abi_version() == DHD_ABI_VERSION ? 0 : 1; }\n')
    out = subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-fsyntax-only', '-I' + os.path.join(ROOT, 'include'), str(src)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    _build_c_example(tmp_path)


@pytest.mark.gpu
def test_c_example_reproduces_the_reference_known_answer_test(gpu, tmp_path):
    """The reference's in-file KAT (ops/bev_pool_v2/bev_pool.py:163-194) from a plain C program: hipMalloc'd buffers, its own
    stream, forward + device regrouping + backward through the C ABI, no Python and no torch in the process."""
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'KAT ok' in out.stdout, (out.stdout + out.stderr)[-2000:]
