"""ctypes binding of libmicroaligner_hip.so (the C-ABI declared in include/microaligner_hip.h).

There is no CPU fallback: if the HIP library cannot be loaded the import of any
compute entry point raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MICROALIGNER_HIP_LIB: another build of the same library (A/B timing of two builds on one box, tools/ab_libs.sh)
LIB_PATH = os.environ.get("MICROALIGNER_HIP_LIB") or os.path.join(_HERE, "libmicroaligner_hip.so")

MA_U8, MA_U16, MA_F32 = 0, 1, 2
MA_OK, MA_EINVAL, MA_ENOMEM, MA_EHIP, MA_ENODEV = 0, -1, -2, -3, -4
MA_FB_MULADD_FUSED = 1
MA_KNN_AUTO, MA_KNN_EXACT, MA_KNN_FILTERED, MA_KNN_FILTERED_F32 = 0, 1, 2, 3   # enum ma_knn_mode
MA_DOG_FUSED_BLUR, MA_DOG_FUSED_SCALE, MA_DOG_REPORT_ASYNC = 1, 2, 4
MA_FLOW_CELL_REPLICAS = 8
MA_OPT_COMPANION_STREAM, MA_OPT_WORKSPACE_LIMIT, MA_OPT_WARP_BAND_BYTES = 1, 2, 3      # enum ma_option
MA_ENGINE_COMPUTE, MA_ENGINE_H2D, MA_ENGINE_D2H = 0, 1, 2   # enum ma_engine

KERNEL_IDS = {"polyexp_m0": 0, "blur_v": 1, "blur_h_solve": 2, "warp": 3, "merge": 4, "pyr_down": 5,
              "pyr_up": 6, "dog": 7, "nmi": 8, "other": 9}

_vp, _i, _sz, _d, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_float

# name -> (restype, argtypes): exactly the symbols of include/microaligner_hip.h
SIGNATURES = {
    "ma_version": (C.c_char_p, []),
    "ma_last_error": (C.c_char_p, []),
    "ma_device_count": (_i, [C.POINTER(_i)]),
    "ma_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "ma_ctx_destroy": (None, [_vp]),
    "ma_sync": (_i, [_vp]),
    "ma_ctx_set_workspace_limit": (_i, [_vp, _sz]),
    "ma_ctx_stream": (_vp, [_vp]),
    "ma_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "ma_free": (_i, [_vp, _vp]),
    "ma_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "ma_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "ma_memcpy_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "ma_memcpy_d2h_async": (_i, [_vp, _vp, _vp, _sz]),
    "ma_host_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "ma_host_free": (_i, [_vp]),
    "ma_host_register": (_i, [_vp, _sz]),
    "ma_host_unregister": (_i, [_vp]),
    "ma_host_transfer_is_direct": (_i, [_vp, _sz, C.POINTER(C.c_int)]),
    "ma_transient_pin_stats": (_i, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(_d), C.POINTER(_d), C.POINTER(C.c_int)]),
    "ma_memset": (_i, [_vp, _vp, _i, _sz]),
    "ma_event_create": (_i, [_vp, C.POINTER(_vp)]),
    "ma_event_destroy": (_i, [_vp, _vp]),
    "ma_event_record": (_i, [_vp, _vp]),
    "ma_event_elapsed_ms": (_i, [_vp, _vp, _vp, C.POINTER(_f)]),
    "ma_profile_enable": (_i, [_vp, _i]),
    "ma_profile_reset": (_i, [_vp]),
    "ma_profile_get": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_longlong), C.POINTER(_d)]),
    "ma_farneback_tiled": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _d, _i, _vp]),
    "ma_farneback_debug": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _vp, _vp, _vp, _vp]),
    "ma_remap_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "ma_warp_tiled": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "ma_merge_flows_tiled": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ma_warp_pages_host": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _vp, _i, _i]),
    "ma_pyr_down": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ma_pyr_up_flow": (_i, [_vp, _vp, _i, _i, _f, _vp, _i, _i]),
    "ma_minmax": (_i, [_vp, _vp, _i, _sz, C.POINTER(_d), C.POINTER(_d)]),
    "ma_dog_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, C.POINTER(_i)]),
    "ma_nmi_u8": (_i, [_vp, _vp, _vp, _sz, _sz, C.POINTER(_d), _i, C.POINTER(_i)]),
    "ma_nmi_u8_pair": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, C.POINTER(_d), C.POINTER(_d), _i, C.POINTER(_i)]),
    "ma_max_project": (_i, [_vp, _vp, _i, _i, _sz, _vp]),
    "ma_normalize_minmax_u8": (_i, [_vp, _vp, _i, _sz, _vp]),
    "ma_warp_affine": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_d), _vp]),
    "ma_warp_tiled_minmax": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "ma_warp_tiled_flowcells": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "ma_merge_flows_tiled_cells": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ma_pyr_down_minmax": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ma_dog_u8_minmax": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, C.POINTER(_i)]),
    "ma_dog_u8_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, C.POINTER(_i)]),
    "ma_warp_affine_cv": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_d), _i, _i, _vp]),
    "ma_knn2_l2": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "ma_knn2_l2_ex": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "ma_match_similarity": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _f, _d, _d, _i, C.POINTER(C.c_ulonglong), C.POINTER(_d),
                                 C.POINTER(_i), C.POINTER(_i)]),
    "ma_host_pcg64_choice2": (_i, [C.POINTER(C.c_ulonglong), _i, _i, C.POINTER(_i)]),
    "ma_host_ransac_iterations": (_i, [_i, _i, _d, _i, _i, C.POINTER(_i)]),
    "ma_fast_nms": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ma_feature_extract": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(C.POINTER(_d)), C.POINTER(_i), C.POINTER(_d),
                                C.POINTER(_d), _sz, _i, _vp, _vp, _vp, C.POINTER(_i)]),
    "ma_feature_extract_enqueue": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(C.POINTER(_d)), C.POINTER(_i), C.POINTER(_d),
                                        C.POINTER(_d), _sz, _i, _vp, _vp, _vp, C.POINTER(_i)]),
    "ma_daisy_describe": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(C.POINTER(_d)), C.POINTER(_i), C.POINTER(_d), C.POINTER(_d),
                               _vp, _vp, _i, _vp]),
}



class MaParams(C.Structure):
    """ma_params of include/microaligner_hip.h."""
    _fields_ = [("num_pyr_lvl", _i), ("num_iterations", _i), ("tile_size", _i), ("overlap", _i), ("use_full_res_img", _i),
                ("use_dog", _i), ("fb_flags", _i), ("dog_flags", _i)]


class MaLevelReport(C.Structure):
    """ma_level_report of include/microaligner_hip.h."""
    _fields_ = [("factor", _i), ("h", _i), ("w", _i), ("mi_after", _d), ("mi_before", _d), ("accepted", _i)]


class MaFeatureRoundResult(C.Structure):
    """ma_feature_round_result of include/microaligner_hip.h."""
    _fields_ = [("m2x3", _d * 6), ("n_query", _i), ("n_good", _i), ("status", _i), ("is_identity", _i), ("n_scores", _i),
                ("zero_max", _i)]


SIGNATURES.update({
    "ma_feature_round": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _sz, C.POINTER(C.POINTER(_d)),
                              C.POINTER(_i), C.POINTER(_d), C.POINTER(_d), _sz, _vp, _vp, C.POINTER(_d), C.POINTER(_d), _i,
                              C.POINTER(MaFeatureRoundResult)]),
    "ma_ctx_trim": (_i, [_vp]),
    "ma_ctx_transfer_stats": (_i, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), _i]),
    "ma_params_default": (None, [C.POINTER(MaParams)]),
    "ma_optflow_register": (_i, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(MaParams), _vp, C.POINTER(MaLevelReport), _i,
                                 C.POINTER(_i)]),
    "ma_host_np_mean": (_i, [C.POINTER(_d), C.c_long, C.POINTER(_d)]),
    "ma_fast_keypoints": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, C.POINTER(_i)]),
    "ma_cut_tiles_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ma_device_info": (_i, [_i, C.c_char_p, _sz, C.c_char_p, _sz, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i)]),
    "ma_clock_probe": (_i, [_vp, _d, C.POINTER(_d)]),
    "ma_ctx_set_option": (_i, [_vp, _i, C.c_longlong]),
    "ma_ctx_get_option": (_i, [_vp, _i, C.POINTER(C.c_longlong)]),
    "ma_engine_memcpy_h2d": (_i, [_vp, _i, _vp, _vp, _sz]),
    "ma_engine_memcpy_d2h": (_i, [_vp, _i, _vp, _vp, _sz]),
    "ma_engine_record": (_i, [_vp, _i, _vp]),
    "ma_engine_wait": (_i, [_vp, _i, _vp]),
    "ma_engine_sync": (_i, [_vp, _i]),
    "ma_event_sync": (_i, [_vp, _vp]),
    "ma_host_parallel_copy": (_i, [_vp, _vp, _sz]),
    "ma_host_stream_copy": (_i, [_vp, _vp, _sz]),
    "ma_warp_pages_plan": (_i, [_i, _i, _i, _i, _i, _sz, C.POINTER(_i), C.POINTER(_i)]),
    "ma_convert_f32": (_i, [_vp, _vp, _i, _sz, _vp]),
})

_lib = None


def load():
    """Load the shared library and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m microaligner_amd.build). "
            "microaligner_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the C-ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def source_hash():
    """Hash of the kernel sources the loaded library was built from (the `src:` field of ma_version())."""
    v = load().ma_version().decode()
    return v.split("src:", 1)[1].strip() if "src:" in v else None


def check(rc):
    if rc == MA_OK:
        return
    msg = load().ma_last_error().decode("utf-8", "replace")
    if rc == MA_EINVAL:
        raise ValueError(msg)
    if rc == MA_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(f"microaligner_hip error {rc}: {msg}")
