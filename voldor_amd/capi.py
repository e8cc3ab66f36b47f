"""ctypes binding of libvoldor_hip.so (the C-ABI declared in include/voldor_hip.h).

There is NO CPU fallback: if the HIP library is missing (not built) this module raises, and
every entry point returns a non-zero HIP error code when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvoldor_hip.so")
_F = C.POINTER(C.c_float)
_FF = C.POINTER(_F)
_I = C.POINTER(C.c_int)

# every symbol include/voldor_hip.h declares (checked by tests/test_abi.py)
C_SYMBOLS = [
    "vk_meanshift_gpu", "vk_fit_robust_gaussian", "vk_collect_p3p_instances", "vk_solve_batch_p3p_ap3p_gpu",
    "vk_solve_batch_p3p_lambdatwist_gpu", "vk_solve_batch_p3p_lambdatwist_f64_gpu", "vk_optimize_depth_gpu", "vk_gblur", "vk_fb_smooth",
    "vk_py_voldor_wrapper", "vk_voldor_device", "vk_voldor_device_block", "vk_voldor_device_batch", "vk_read_flo", "vk_write_flo", "vk_eval_covisibility", "vk_align_frame_init_gpu", "vk_align_frame_eval_gpu", "vk_last_camera_stats", "vk_estimate_pose_epipolar",
    "vk_estimate_depth_closed_form", "vk_estimate_pose_epipolar5", "vk_bootstrap_gpu_points", "vk_fivept_solve", "vk_bootstrap_gpu", "vk_get_compacted_points", "vk_set_rand_epoch", "vk_get_rand_epoch", "vk_set_strict_math", "vk_get_strict_math", "vk_set_reference_svd", "vk_get_reference_svd", "vk_set_reference_rng", "vk_get_reference_rng", "vk_set_reference_tex", "vk_get_reference_tex",
    "vk_profile_enable", "vk_profile_get", "vk_device_count", "vk_set_device", "vk_version",
    "vk_dist_get_unique_id", "vk_dist_init", "vk_dist_init_file", "vk_dist_rank", "vk_dist_world", "vk_dist_rccl_version", "vk_dist_allgather",
    "vk_dist_allreduce_max", "vk_dist_barrier", "vk_dist_allgather_stats", "vk_voldor_sharded", "vk_dist_finalize",
]
# mangled C++ symbols of include/gpu_kernels.h + include/py_export.h (what voldor/*.cpp and the .pyx link against)
CXX_SYMBOLS = [
    "_Z13meanshift_gpuPffS_S_PibiifiifS_".replace("S_S_PibiifiifS_", "S_S_Pibiifiif"),
    "_Z19fit_robust_gaussianPfS_S_ffS_Piiifi",
    "_Z21collect_p3p_instancesPPfS0_S_S_S0_S0_S_S_iiiiffffi",
    "_Z24solve_batch_p3p_ap3p_gpuPfS_S_S_S_ii",
    "_Z31solve_batch_p3p_lambdatwist_gpuPfS_S_S_S_ii",
    "_Z18optimize_depth_gpuPPfS0_S0_S0_S0_S0_S0_S_S_S_S0_S0_S0_S0_fiiiifiiiffffbfffb",
    "_Z17py_voldor_wrapperPKfS0_S0_S0_S0_S0_fffffiiiiPKcRiPfS4_S4_S4_",
    "_Z20align_frame_init_gpuPPfS0_S0_S_ffiii",
    "_Z20align_frame_eval_gpuiiPKfS0_PfS1_b",
]

_lib = None


class VoldorHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("VOLDOR_HIP_LIB", LIB_PATH)  # A/B runs of two builds on one box (scripts/ab.sh)
        if path != LIB_PATH:
            _lib = C.CDLL(path)
            _lib.vk_version.restype = C.c_char_p
            _lib.vk_get_rand_epoch.restype = C.c_uint
            return _lib
        if not os.path.exists(LIB_PATH):
            raise VoldorHipError(
                f"{LIB_PATH} not found: build it with `python -m voldor_amd.build` (needs hipcc). "
                "voldor_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.vk_version.restype = C.c_char_p
        _lib.vk_get_rand_epoch.restype = C.c_uint
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise VoldorHipError(f"{what} failed with code {rc}")


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fp(a):
    return None if a is None else a.ctypes.data_as(_F)


def ptr_table(arrs):
    """float*[] table from a list of contiguous float32 arrays (or None)."""
    if arrs is None:
        return None, None
    keep = [f32(a) for a in arrs]
    tab = (_F * len(keep))(*[a.ctypes.data_as(_F) for a in keep])
    return tab, keep
