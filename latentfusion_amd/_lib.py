"""ctypes binding of liblf_hip.so (the C ABI declared in include/lf_hip.h).

The product path has NO fallback: if the shared object is missing or a symbol cannot be
resolved, `lib()` raises.  (On a box with hipcc the library is built on first use.)
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'liblf_hip.so')

LF_EPI_LRELU = 1
LF_EPI_PIXELNORM = 2
LF_EPI_ADD = 4
LF_EPI_DOT = 8
LF_RING_ADD_BF16, LF_RING_OUT_BF16, LF_RING_ROUND = 1, 2, 4
LF_RING_EX_NONE, LF_RING_EX_RH, LF_RING_EX_BLEND, LF_RING_EX_ABWD, LF_RING_EX_BLOCK, LF_RING_EX_PREV = 0, 1, 2, 3, 4, 5
LF_OUT_DEPTH_INNER = 0x100
LF_IO_IN_BF16, LF_IO_OUT_BF16, LF_IO_ADDEND_BF16 = 1, 2, 4
LF_MAP_O2C = 0
LF_MAP_C2O = 1
LF_MAP_COEFS = 20
LF_FUSE_MEAN, LF_FUSE_MAX, LF_FUSE_ABSMAX, LF_FUSE_MEDIAN = 0, 1, 2, 3
LF_AMAX_FLOATS = 2048          # floats of a max-abs side-channel buffer (include/lf_hip.h)

P = c_void_p
# name -> (restype, argtypes); mirrors include/lf_hip.h one to one (the product ABI)
SIGNATURES = {
    'lf_abi_version': (c_int, []),
    'lf_device_name': (c_int, [c_char_p, c_int]),
    'lf_resample3d_fwd': (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_resample3d_bwd_coef_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'lf_resample3d_bwd_coef': (c_int, [P, P, c_int, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_resample3d_bwd_vol_det_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'lf_resample3d_bwd_vol_det': (c_int, [P, P, c_int, P, c_int, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_conv3x3_cout_padded': (c_int, [c_int]),
    'lf_conv1x1_cout_padded': (c_int, [c_int]),
    'lf_conv3x3_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_float, c_uint, c_float, c_float, P]),
    'lf_conv1x1_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_long, c_long, c_int, c_long, c_int,
                               c_int, c_long, c_float, c_uint, c_float, c_float, P]),
    'lf_conv1x1_fwd_scaled': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_long, c_long, c_int, c_long, c_int,
                                      c_int, c_long, c_float, c_uint, c_float, c_float, P]),
    'lf_conv3x3_bwd_data': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, P, c_uint,
                                    c_float, P]),
    'lf_conv1x1_bwd_data': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_long, c_int, c_int, c_long, c_float, P, P,
                                    c_uint, c_float, P, P]),
    'lf_conv3d_c16_split_wpack_halfs': (c_size_t, []),
    'lf_conv3d_c16_split_pairs': (None, [P]),
    'lf_conv3d_c16_split': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float, P, P,
                                    c_uint, P, P, P]),
    'lf_conv3d_c16_wino_upack_floats': (c_size_t, []),
    'lf_conv3d_c16_wino': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float, P, P,
                                   c_uint, P, P]),
    'lf_conv3d_c16_wino_proj_pack_floats': (c_size_t, [c_int]),
    'lf_conv3d_c16_wino_projfwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float,
                                           P, P, P, P, c_float, c_uint, P]),
    'lf_round_bf16': (c_int, [P, P, c_long, P]),
    'lf_conv3d_c16_ring_bf16_wpack_elems': (c_size_t, []),
    'lf_conv3d_c16_ring_bf16': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float, P, c_int, P]),
    'lf_conv3d_c16_ring_bf16_io': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float, P, c_int, c_int, P]),
    'lf_conv3d_c16_ring_multi': (c_int, [P, c_int, P, c_int, P, P, c_uint, P, P, c_uint, c_int, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'lf_conv3d_c16_ring_blend': (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'lf_wino3d_tiles': (c_long, [c_int, c_int, c_int, c_int]),
    'lf_wino3d_input_transform': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_wino_fused_cout_padded': (c_int, [c_int]),
    'lf_wino_fused_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'lf_wino_fused_gemm': (c_int, [P, P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, P]),
    'lf_wino2d_tiles': (c_long, [c_int, c_int, c_int]),
    'lf_wino2d_input_transform': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'lf_gru_stage_a': (c_int, [P, P, c_int, P, P, P, c_long, c_int, c_int, c_int, P]),
    'lf_gru_stage_b': (c_int, [P, P, P, P, P, c_long, c_int, c_int, c_int, P]),
    'lf_lstm_cell_fwd': (c_int, [P, P, P, P, c_long, c_int, P]),
    'lf_lstm_cell_bwd': (c_int, [P, P, P, P, P, P, c_long, c_int, P]),
    'lf_gru_stage_b_bwd': (c_int, [P, P, P, P, P, P, P, c_long, P]),
    'lf_gru_stage_a_bwd': (c_int, [P, P, P, P, P, P, P, P, c_long, P]),
    'lf_gru_train_stage_a': (c_int, [P, P, P, c_long, c_int, P]),
    'lf_gru_train_stage_b': (c_int, [P, P, P, P, c_long, c_int, P]),
    'lf_gru_train_stage_b_bwd': (c_int, [P, P, P, P, P, P, P, P, P, c_long, c_int, P]),
    'lf_gru_train_stage_a_bwd': (c_int, [P, P, P, P, P, P, P, c_long, c_int, P]),
    'lf_lift16_fwd': (c_int, [P, P, P, P, P, c_long, c_long, c_int, c_float, c_float, c_float, P]),
    'lf_lift16_bwd_scratch_bytes': (c_size_t, [c_int]),
    'lf_lift16_bwd': (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, c_long, c_long, c_int, c_float, c_float, c_int, P]),
    'lf_proj16_fwd': (c_int, [P, P, P, P, P, c_long, c_long, c_int, c_float, c_float, c_float, P]),
    'lf_proj16_bwd_scratch_bytes': (c_size_t, [c_int]),
    'lf_proj16_bwd': (c_int, [P, P, P, P, P, P, c_size_t, c_long, c_long, c_int, c_float, P]),
    'lf_sum_views_bf16': (c_int, [P, P, c_long, c_int, c_int, P]),
    'lf_occ_input_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_long, c_float, P]),
    'lf_occ_input_bwd': (c_int, [P, P, P, P, P, P, P, c_long, c_float, P, P, c_uint, P]),
    'lf_occ_conv17_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    'lf_occ_conv17_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'lf_column_reduce_sum_fwd': (c_int, [P, P, c_int, c_int, c_long, c_int, P]),
    'lf_column_reduce_sum_bwd': (c_int, [P, P, c_int, c_int, c_long, c_int, P]),
    'lf_column_softmax_fwd': (c_int, [P, P, P, c_int, c_int, c_long, P]),
    'lf_occ_input_bwd_proj': (c_int, [P, P, P, P, P, c_float, P, P, c_int, c_int, c_long, c_float, P, P, c_uint, P]),
    'lf_occ_weight_grad': (c_int, [P, P, P, c_float, P, c_int, c_int, c_long, P]),
    'lf_pw16_fwd': (c_int, [P, c_int, P, P, c_float, c_uint, c_float, P, c_long, P]),
    'lf_pw16_bwd_scratch_bytes': (c_size_t, [c_long]),
    'lf_pw16_bwd': (c_int, [P, P, c_float, P, c_int, P, P, c_size_t, c_long, P]),
    'lf_occ_weight_grad_softmax_bwd': (c_int, [P, P, P, c_float, P, P, c_int, c_int, c_long, P]),
    'lf_occ_head_bwd': (c_int, [P, P, c_float, P, P, c_uint, c_float, P, c_long, P]),
    'lf_column_softmax_head_fwd': (c_int, [P, P, P, c_float, P, P, c_int, c_int, c_long, P]),
    'lf_column_softmax_bwd': (c_int, [P, P, P, P, c_int, c_int, c_long, P]),
    'lf_column_scale_fwd': (c_int, [P, P, P, c_long, c_int, P]),
    'lf_column_scale_bwd': (c_int, [P, P, P, P, P, c_long, c_int, P]),
    'lf_fuse_views_fwd': (c_int, [P, P, P, c_int, c_int, c_long, c_long, P]),
    'lf_fuse_views_bwd': (c_int, [P, P, P, c_int, c_int, c_long, c_long, P]),
    'lf_fuse_blend_fwd': (c_int, [P, P, P, P, c_int, c_long, c_int, c_long, c_long, P]),
    'lf_fuse_blend_bwd': (c_int, [P, P, P, P, P, c_int, c_long, c_int, c_long, c_long, P]),
    'lf_grid_sample2d_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_grid_sample2d_bwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_conv_bwd_weight_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'lf_conv_bwd_weight': (c_int, [P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'lf_conv_bwd_weight_bf16': (c_int, [P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'lf_conv_bwd_weight_bf16_io': (c_int, [P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'lf_pixelnorm_fwd': (c_int, [P, P, P, c_long, c_int, c_float, P]),
    'lf_epilogue_bwd': (c_int, [P, P, P, P, c_long, c_int, c_uint, c_float, P]),
    'lf_epilogue_bwd_c16_scratch_bytes': (c_size_t, [c_long]),
    'lf_epilogue_bwd_c16': (c_int, [P, P, P, P, P, P, c_size_t, c_long, c_uint, c_float, c_int, P]),
    'lf_resample3d_fwd_io': (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_resample3d_bwd_vol_det_io_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'lf_resample3d_bwd_vol_det_io': (c_int, [P, P, c_int, P, c_int, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_resize_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_resize_bwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_camera_coefs': (c_int, [P, P, c_float, c_float, c_int, c_int, P, P, c_int, P]),
    'lf_camera_coefs_bwd': (c_int, [P, P, P, c_int, P]),
    'lf_pose_loss_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'lf_pose_loss_fwd': (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_pose_loss_bwd': (c_int, [P, P, P, P, P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_pose_loss_fwd_depth': (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_pose_loss_bwd_depth': (c_int, [P, P, P, P, P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_pose_loss_fwd_masked': (c_int, [P, P, P, P, P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_adam_step': (c_int, [P, P, P, P, P, P, c_float, c_float, c_float, c_float, c_float, c_int, c_int, P]),
    'lf_nchw_to_nhwc': (c_int, [P, P, c_int, c_int, c_long, P]),
    'lf_nhwc_to_nchw': (c_int, [P, P, c_int, c_int, c_long, P]),
    'lf_lift_unfold': (c_int, [P, P, P, c_int, c_long, c_int, c_int, P]),
    'lf_lift_permute': (c_int, [P, P, c_int, c_long, c_int, c_int, c_int, P]),
    'lf_lift_norm_unfold': (c_int, [P, P, P, c_int, c_long, c_int, c_int, c_float, c_int, P]),
    'lf_lift_bwd': (c_int, [P, P, P, P, c_int, c_long, c_int, c_int, c_float, c_int, c_int, P]),
}

# entry points of include/lf_hip_experimental.h (A/B switches, superseded kernels): same shared object, bound for
# latentfusion_amd/experimental.py, tools/ and the tests that pin them
EXPERIMENTAL_SIGNATURES = {
    'lf_set_tuning': (c_int, [c_int, c_int]),
    'lf_resample3d_bwd_vol': (c_int, [P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'lf_conv3d_c16_wino_projbwd': (c_int, [P, P, c_float, P, P, c_uint, P, P, c_int, c_int, c_int, c_int, c_float, c_float,
                                           P, P, c_uint, P]),
    'lf_conv3d_c16_wino_split_upack_halfs': (c_size_t, []),
    'lf_conv3d_c16_wino_split': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float, P, P,
                                         c_uint, P, P, P]),
    'lf_conv3d_c16_bf16_wpack_elems': (c_size_t, []),
    'lf_conv3d_c16_bf16': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float, c_int, P]),
    'lf_wino3d_output_transform': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float, P]),
    'lf_wino2d_output_transform': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint, c_float, c_float, P]),
}

_lib = None


class LFHipError(RuntimeError):
    pass


def lib():
    """Returns the loaded library, building it first if absent and hipcc is available."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        try:
            from .csrc import build as _build
            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise LFHipError(f'liblf_hip.so is missing at {LIB_PATH} and could not be built: {e}') from e
    # PyTorch-ROCm ships its own libamdhip64.so and opens it by path.  If liblf_hip.so (linked against the
    # system HIP runtime) were loaded first, the process would end up with TWO HIP runtimes and the one torch
    # did not initialise would see no device.  Importing torch first makes the loader bind liblf_hip.so's
    # libamdhip64 dependency to the copy that is already resident.
    import torch  # noqa: F401
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise LFHipError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in list(SIGNATURES.items()) + list(EXPERIMENTAL_SIGNATURES.items()):
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise LFHipError(f'{LIB_PATH} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    if handle.lf_abi_version() != 1:
        raise LFHipError('liblf_hip.so ABI version mismatch')
    _lib = handle
    return _lib


_ERR = {-1: 'LF_EINVAL (bad size/flag combination)', -2: 'LF_EALIGN (alignment / channel count)',
        -3: 'LF_ENOSPC (scratch too small)'}


# Algorithmic-byte accounting of a step (bench.py `cfg5.roofline_step`): while BYTE_LOG is a dict, every tensor whose pointer the
# operator wrappers hand to the library (ops._ptr) is noted, and `check` -- called once per launch -- closes the launch:
# BYTE_LOG[entry point] = [launches, bytes], bytes = the sizes of the launch's input / output tensors, each once (scratch
# buffers excluded) = what the launch has to move through HBM at least.  None: no cost beyond one comparison per pointer.
BYTE_LOG = None
_PENDING = [0]


def note_bytes(t, scratch=False):
    if BYTE_LOG is not None and not scratch:
        n = t.numel() * t.element_size()
        try:                                            # (an expanded view is read once, not numel() times)
            n = min(n, t.untyped_storage().nbytes() - t.storage_offset() * t.element_size())
        except Exception:                               # noqa: BLE001
            pass
        _PENDING[0] += n


def check(rc, what):
    if BYTE_LOG is not None:
        e = BYTE_LOG.setdefault(what, [0, 0])
        e[0] += 1
        e[1] += _PENDING[0]
        _PENDING[0] = 0
    if rc != 0:
        raise LFHipError(f'{what} failed: {_ERR.get(rc, "hipError " + str(rc))}')
