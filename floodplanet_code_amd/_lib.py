"""ctypes binding of libfloodunet.so (the C ABI declared in include/floodunet.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, this
module raises.  The product path never routes through torch ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# FU_LIB_PATH selects another build of the same sources (A/B timing of kernel variants, tools/ab_libs.sh); it must
# export the full ABI of include/floodunet.h like the in-tree library.
LIB_PATH = os.environ.get("FU_LIB_PATH") or os.path.join(_HERE, "libfloodunet.so")

FU_OK, FU_ERR_INVALID, FU_ERR_HIP, FU_ERR_STATE, FU_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
FU_F32, FU_BF16, FU_F16 = 0, 1, 2
PRECISIONS = {"fp32": FU_F32, "f32": FU_F32, "float32": FU_F32, "bf16": FU_BF16, "bfloat16": FU_BF16,
              "fp16": FU_F16, "f16": FU_F16, "float16": FU_F16, "half": FU_F16}


class FuConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("n_channels", C.c_int32),
        ("n_classes", C.c_int32),
        ("base_channels", C.c_int32),
        ("bilinear", C.c_int32),
        ("max_batch", C.c_int32),
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("precision", C.c_int32),
        ("device", C.c_int32),
        ("n_encoders", C.c_int32),
        ("enc_channels", C.c_int32 * 6),
    ]


class FloodUNetError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libfloodunet error {status}: {message}")
        self.status = status


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
SYNC_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int)   # fu_sync_hook
_f = C.c_float
_d = C.c_double

# name -> (restype, argtypes).  Every symbol include/floodunet.h declares is listed here; the
# CPU test-suite checks that the library exports each of them.
SIGNATURES = {
    "fu_abi_version": (_i, []),
    "fu_last_error": (C.c_char_p, []),
    "fu_create": (_i, [C.POINTER(FuConfig), C.POINTER(_p)]),
    "fu_destroy": (_i, [_p]),
    "fu_num_params": (_i, [_p]),
    "fu_total_param_elems": (_i64, [_p]),
    "fu_param_info": (_i, [_p, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(_i64), C.POINTER(_i64)]),
    "fu_num_bn": (_i, [_p]),
    "fu_total_bn_channels": (_i64, [_p]),
    "fu_bn_info": (_i, [_p, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(_i64)]),
    "fu_bind_buffers": (_i, [_p, _p, _p, _p, _p, _p]),
    "fu_params_changed": (_i, [_p]),
    "fu_forward": (_i, [_p, _p, _i, _i, _p, _p]),
    "fu_forward_srcs": (_i, [_p, C.POINTER(_p), C.POINTER(C.c_int32), _i, _i, _i, _p, _p]),
    "fu_loss_ce": (_i, [_p, _p, _i, _p, _p, _p, _p]),
    "fu_loss_bce_dice": (_i, [_p, _p, _i, _f, _p, _p]),
    "fu_backward": (_i, [_p, _p, _p]),
    "fu_num_blocks": (_i, [_p]),
    "fu_backward_block": (_i, [_p, _i, _p, _p]),
    "fu_block_param_range": (_i, [_p, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "fu_bind_adam_state": (_i, [_p, _p, _p]),
    "fu_scale_loss_grad": (_i, [_p, _p, _p]),
    "fu_adam_step": (_i, [_p, _d, _d, _d, _d, _i64, _d, _p]),
    "fu_adam_state": (_i, [_p, C.POINTER(_p), C.POINTER(_p)]),
    "fu_fp16_guard_state": (_i, [_p, C.POINTER(_i64), C.POINTER(C.c_int32)]),
    "fu_adam_scalars": (_i, [_d, _d, _d, _d, _i64, _d, C.POINTER(C.c_float)]),
    "fu_adam_step_dev": (_i, [_p, _p, _p]),
    "fu_zero_grads": (_i, [_p, _p]),
    "fu_stitch_add": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fu_stitch_finalize": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "fu_augment": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i64, _p]),
    "fu_resize_lanczos4_tiles": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "fu_assemble_tiles": (_i, [C.POINTER(_p), C.POINTER(C.c_int32), _i, _i, _i, _i, _p, _p, _i, _p, _p, _f, _p, _p, _p, _p]),
    "fu_workspace_bytes": (_i64, [_p]),
    "fu_flops_per_tile": (_i, [_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fu_profile_enable": (_i, [_p, _i]),
    "fu_test_force_general_conv": (None, [_i]),
    "fu_test_force_lockstep_wgrad": (None, [_i]),
    "fu_test_conv_tile_mode": (None, [_i]),
    "fu_test_bnb_separate": (None, [_i]),
    "fu_test_head_store_g": (None, [_i]),
    "fu_test_force_full_taps": (None, [_i]),
    "fu_test_perturb_bnb_sums": (None, [_f]),
    "fu_test_get_buffer": (_i, [_p, _i, _i, C.POINTER(_p), C.POINTER(_i64)]),
    "fu_set_side_stream": (_i, [_p, _i]),
    "fu_backward_join": (_i, [_p, _p]),
    "fu_backward_fence": (_i, [_p, _p, _p]),
    "fu_dp_unique_id": (_i, [_p]),
    "fu_dp_init": (_i, [_p, _p, _i, _i]),
    "fu_dp_broadcast_state": (_i, [_p, _p]),
    "fu_allreduce_begin": (_i, [_p, _i64, _i64, _p]),
    "fu_allreduce_wait": (_i, [_p, _p]),
    "fu_dp_destroy": (_i, [_p]),
    "fu_set_exact_sync": (_i, [_p, SYNC_HOOK, _p, _i, _p, _i64]),
    "fu_exact_sync_bytes": (_i64, [_p]),
    "fu_profile_read": (_i, [_p, _i, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                             C.POINTER(C.c_char_p)]),
    "fu_elem_size": (_i, [_i]),
    "fu_op_nchw_to_nhwc": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fu_op_nhwc_to_nchw": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fu_op_conv3x3_fwd": (_i, [_i, _p, _i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "fu_op_conv3x3_dgrad": (_i, [_i, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _p]),
    "fu_op_conv3x3_wgrad": (_i, [_i, _p, _i, _p, _p, _p, _i, _p, _i, _p, _i, _i, _i, _p]),
    "fu_op_conv3x3_dgrad_bnsums": (_i, [_i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fu_op_head_bwd": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fu_op_bn_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fu_op_maxpool2": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fu_op_upsample2": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libfloodunet.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C floodplanet_code_amd/csrc`). "
            "There is no CPU / PyTorch fallback for this path.")
    # torch first: its libamdhip64 must be the process's HIP runtime before ours is resolved (same SONAME, so the
    # loader then binds libfloodunet.so to it).  The device pointers and stream handles that cross the C ABI come from
    # torch; with the library loaded first the process ends up with /opt/rocm's runtime beside torch's bundled one and
    # hipSetDevice fails with "no ROCm-capable device" (seen when build() and smoke() ran in one process).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != FU_OK:
        msg = load().fu_last_error()
        raise FloodUNetError(status, msg.decode("utf-8", "replace") if msg else "")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
