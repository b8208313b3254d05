"""ctypes binding of libdhd_amd.so (C ABI declared in include/dhd_amd.h).

There is deliberately no fallback: if the HIP library is missing or a call fails, the
error is raised to the caller.  `import torch` must precede the load so that the library
binds to the HIP runtime PyTorch already mapped (same SONAME, libamdhip64.so.7).
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DHD_AMD_LIB', os.path.join(_HERE, 'csrc', 'libdhd_amd.so'))

DHD_MAX_GRIDS = 4
ABI_VERSION = 5

_ERRORS = {-1: 'DHD_EINVAL (bad argument)', -2: 'DHD_ENOSPACE (workspace too small)',
           -3: 'DHD_EUNSUPPORTED (size outside supported range)'}


class DhdError(RuntimeError):
    pass


class Grid(C.Structure):
    _fields_ = [('lower', C.c_float * 3), ('interval', C.c_float * 3), ('size', C.c_float * 3),
                ('n', C.c_int32 * 3)]


class MghsDesc(C.Structure):
    _fields_ = [('batch', C.c_int32), ('n_cams', C.c_int32), ('n_depth', C.c_int32),
                ('fh', C.c_int32), ('fw', C.c_int32), ('channels', C.c_int32), ('n_grids', C.c_int32),
                ('grid', Grid * DHD_MAX_GRIDS), ('flags', C.c_int32)]


MGHS_DETERMINISTIC = 1     # dhd_mghs_desc.flags
MGHS_FEAT_GRAD_NCHW = 2
MGHS_DEBUG_SCAN_SELF_SERVE = 4


class MghsWorkspace(C.Structure):
    _fields_ = [('state', C.c_void_p), ('state_bytes', C.c_size_t), ('scratch', C.c_void_p), ('scratch_bytes', C.c_size_t)]


class TensorView(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('batch_stride', C.c_int64), ('z_stride', C.c_int64), ('channel_stride', C.c_int64),
                ('dtype', C.c_int32)]


DTYPE_CODE = {}   # torch dtype -> dhd_tensor_view.dtype (filled on first use: torch is imported lazily by some callers)


def dtype_code(dt):
    import torch
    if not DTYPE_CODE:
        DTYPE_CODE.update({torch.float32: 0, torch.float16: 1, torch.bfloat16: 2})
    if dt not in DTYPE_CODE:
        raise DhdError(f'pooled tensors are float32, float16 or bfloat16, not {dt}')
    return DTYPE_CODE[dt]


class Calib(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ('sensor2ego', 'intrin', 'post_rot', 'post_tran', 'bda', 'inv_post_rot', 'combine',
                 'frustum_u', 'frustum_v', 'frustum_d')]


class SfaWeights(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in
                 ('fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'conv1_w', 'conv1_b', 'bn1_w', 'bn1_b', 'bn1_mean', 'bn1_var',
                  'conv2_w', 'conv2_b', 'bn2_w', 'bn2_b', 'bn2_mean', 'bn2_var')] +
                [('hidden', C.c_int32), ('training', C.c_int32), ('eps1', C.c_float), ('eps2', C.c_float),
                 ('momentum1', C.c_float), ('momentum2', C.c_float), ('gemm', C.c_int32),
                 ('bn1_batches', C.c_void_p), ('bn2_batches', C.c_void_p), ('io_dtype', C.c_int32), ('storage_dtype', C.c_int32)])


SFA_GEMM = {'default': 0, 'bf16x6': 1, 'f32': 2, 'bf16x3': 3}   # dhd_sfa_weights.gemm


class SfaGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ('fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'conv1_w', 'conv1_b', 'bn1_w', 'bn1_b',
                 'conv2_w', 'conv2_b', 'bn2_w', 'bn2_b')]


_P = C.c_void_p
_I = C.c_int
_PROTOTYPES = {
    'dhd_abi_version': ([], _I),
    'dhd_bev_pool_v2_forward': ([_P] * 8 + [_I, _I, _P], _I),
    'dhd_bev_pool_v2_backward': ([_P] * 10 + [_I, _I, _P], _I),
    'dhd_bev_pool_v2_regroup_scratch_bytes': ([_I, _I], C.c_size_t),
    'dhd_bev_pool_v2_regroup': ([_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, C.c_size_t, _P], _I),
    'dhd_bev_pool_v2_fused_workspace_bytes': ([_I] * 6 + [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)], _I),
    'dhd_bev_pool_v2_fused_forward': ([_P] * 8 + [_I] * 6 + [_P, C.c_size_t, _I, _P, C.c_size_t, _P], _I),
    'dhd_bev_pool_v2_fused_backward': ([_P] * 10 + [_I] * 7 + [_P, C.c_size_t, _P, C.c_size_t, _P], _I),
    'dhd_mghs_workspace_bytes': ([C.POINTER(MghsDesc), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)], _I),
    'dhd_height_band': ([_P, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P], _I),
    'dhd_feat_nchw_to_nhwc': ([_P, _P, _I, _I, _I, _P], _I),
    'dhd_feat_nhwc_to_nchw': ([_P, _P, _I, _I, _I, _P], _I),
    'dhd_mghs_prepare': ([C.POINTER(MghsDesc), C.POINTER(Calib), _P, C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_lift': ([C.POINTER(MghsDesc), C.POINTER(Calib), _P, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P,
                       C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_lift_static': ([C.POINTER(MghsDesc), C.POINTER(Calib), _P, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P,
                              C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_forward': ([C.POINTER(MghsDesc), _P, _P, C.POINTER(_P * DHD_MAX_GRIDS), C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_forward_gather': ([C.POINTER(MghsDesc), _P, _P, C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_forward_stream': ([C.POINTER(MghsDesc), _P, _P, C.POINTER(_P * DHD_MAX_GRIDS), C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_forward_views': ([C.POINTER(MghsDesc), _P, _P, C.POINTER(TensorView * DHD_MAX_GRIDS), C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_forward_stream_views': ([C.POINTER(MghsDesc), _P, _P, C.POINTER(TensorView * DHD_MAX_GRIDS), C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_backward_views': ([C.POINTER(MghsDesc), _P, _P, C.POINTER(TensorView * DHD_MAX_GRIDS), _P, _P, C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_backward': ([C.POINTER(MghsDesc), _P, _P, C.POINTER(_P * DHD_MAX_GRIDS), _P, _P, C.POINTER(MghsWorkspace), _P], _I),
    'dhd_mghs_voxel_index': ([C.POINTER(MghsDesc), C.POINTER(Calib), _I, _P, _P, _P], _I),
    'dhd_mghs_stats': ([C.POINTER(MghsDesc), C.POINTER(MghsWorkspace), C.POINTER(C.c_int32 * DHD_MAX_GRIDS),
                        C.POINTER(C.c_int32 * DHD_MAX_GRIDS), _P], _I),
    'dhd_mghs_debug_keys': ([C.POINTER(MghsDesc), C.POINTER(MghsWorkspace), _P, _P], _I),
    'dhd_hbm_calibrate': ([_P, C.c_size_t, _I, _P], _I),
    'dhd_sfa_channel_mean': ([_P, _P, _I, _I, _I, _P], _I),
    'dhd_sfa_blend1': ([_P, _P, _P, _I, _I, _I, _P], _I),
    'dhd_sfa_blend2': ([_P, _P, _P, _P, _I, _I, _I, _P], _I),
    'dhd_sfa_blend2_backward': ([_P] * 7 + [_I, _I, _I, _P], _I),
    'dhd_sfa_blend1_backward': ([_P] * 5 + [_I, _I, _I, _P], _I),
    'dhd_sfa_mean_backward': ([_P, _P, _I, _I, _I, _P], _I),
    'dhd_sfa_stage_supported': ([_I, _I], _I),
    'dhd_sfa_stage_half_storage_supported': ([_I, _I], _I),
    'dhd_sfa_stage_workspace_bytes': ([_I] * 5 + [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)], _I),
    'dhd_sfa_stage_saved_bytes': ([_I, _I, _I, _I], C.c_size_t),
    'dhd_sfa_stage_scratch_bytes': ([_I, _I, _I, _I], C.c_size_t),
    'dhd_sfa_stage_forward': ([_P, C.POINTER(SfaWeights), _P, _P, _P, _I, _I, _I, _P], _I),
    'dhd_sfa_stage_backward': ([_P, C.POINTER(SfaWeights), _P, _P, _P, C.POINTER(SfaGrads), _P, _I, _I, _I, _P], _I),
    'dhd_sfa_stage_forward_phase': ([_P, C.POINTER(SfaWeights), _P, _P, _P, _I, _I, _I, _I, _P, _P], _I),
    'dhd_sfa_stage_backward_phase': ([_P, C.POINTER(SfaWeights), _P, _P, _P, C.POINTER(SfaGrads), _P, _I, _I, _I, _I, _P, _P], _I),
    'dhd_occ_loss_workspace_bytes': ([], C.c_size_t),
    'dhd_occ_loss_forward': ([_P, _P, _P, _P, C.c_int64, _I, _I, _I, _P, _P, _P], _I),
    'dhd_occ_loss_backward': ([_P, _P, _P, _P, C.c_int64, _I, _I, _I, _P, _P, _P, _P], _I),
    'dhd_occ_argmax_hist': ([_P, _P, _P, C.c_int64, _I, _P, _P, _P], _I),
    'dhd_sparse_bin_labels': ([_P, _P, _I, _I, _I, _I, C.c_float, C.c_float, _I, C.c_float, C.c_float, _I, _P, _P, _P], _I),
    'dhd_sparse_bin_labels_sid': ([_P, _P, _I, _I, _I, _I, C.c_float, C.c_float, _I, C.c_float, C.c_float, _I, _P, _P, _P], _I),
    'dhd_bin_bce_workspace_bytes': ([], C.c_size_t),
    'dhd_bin_bce_forward': ([_P, _P, _P, _I, _I, _I, C.c_float, _P, _P, _P], _I),
    'dhd_bin_bce_backward': ([_P, _P, _P, _I, _I, _I, C.c_float, _P, _P, _P, _P], _I),
    'dhd_points_to_maps': ([_P, _I, _I, _I, _I, _I, C.c_float, C.c_float, _P, _P, _P, _P], _I),
    'dhd_deform_im2col': ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    'dhd_stereo_cost_volume': ([_P, _P, _P, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P], _I),
    'dhd_deform_col2im': ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    'dhd_deform_im2col_t': ([_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    'dhd_deform_col2im_workspace_bytes': ([_I, _I, _I, _I], C.c_size_t),
    'dhd_deform_col2im_gather_supported': ([_I, _I, _I, _I], _I),
    'dhd_deform_col2im_t': ([_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, C.c_size_t, _P], _I),
    'dhd_mghs_softmax_forward': ([_P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P], _I),
    'dhd_mghs_softmax_backward': ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _I, _I, _I, _P], _I),
    'dhd_ema_update': ([_P, _P, _P, _I, C.c_float, C.c_float, _P], _I),
    'dhd_ema_update_dev': ([_P, _P, _P, _I, _P, _P], _I),
    'dhd_bn_supported': ([_I, _I, _I, _I], _I),
    'dhd_bn_workspace_bytes': ([_I, _I, _I], C.c_size_t),
    'dhd_bn_train_forward': ([_P, _I, _I, _I, _I, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P], _I),
    'dhd_bn_train_backward': ([_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    'dhd_transpose_batched': ([_P, _P, _I, C.c_long, _I, _I, _P], _I),
    'dhd_window_rows': ([_P, _P] + [_I] * 9 + [_P], _I),
    'dhd_upsample_bilinear_supported': ([_I] * 8, _I),
    'dhd_upsample_bilinear_forward': ([_P] + [_I] * 8 + [_P, _P], _I),
    'dhd_upsample_bilinear_backward': ([_P] + [_I] * 8 + [_P, _P], _I),
    'dhd_bn_nhwc_supported': ([_I, C.c_long, _I], _I),
    'dhd_bn_nhwc_workspace_bytes': ([C.c_long, _I], C.c_size_t),
    'dhd_bn_nhwc_train_forward': ([_P, _P, _I, C.c_long, _I, _I, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P], _I),
    'dhd_bn_nhwc_train_backward': ([_P, _P, _P, _I, C.c_long, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises DhdError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DhdError(
            f'{LIB_PATH} not found: build it with `make -C dhd_amd/csrc` (or __graft_entry__.build()). '
            'dhd_amd has no CPU or PyTorch fallback for its HIP kernels.')
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.dhd_abi_version() != ABI_VERSION:
        raise DhdError(f'libdhd_amd.so ABI {lib.dhd_abi_version()} != expected {ABI_VERSION}: rebuild')
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise DhdError(f'{what}: {_ERRORS.get(rc, rc)}')
    raise DhdError(f'{what}: hipError_t {rc}')


def stream_ptr(device=None):
    """The current PyTorch HIP stream as a void* for the C ABI."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_gpu_tensor(t, dtype, name):
    if not t.is_cuda:
        raise DhdError(f'{name} must live on the GPU: dhd_amd runs only as HIP kernels (got {t.device})')
    if t.dtype != dtype:
        raise DhdError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise DhdError(f'{name} must be contiguous')
    return t
