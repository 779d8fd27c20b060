"""ctypes binding of libtinyvc_hip.so (the C ABI in include/tinyvc_hip.h).

There is no fallback: if the shared library is missing or a call fails, this raises.  The
library is built in-tree by `tinyvc_amd.build.build()` (hipcc, gfx950).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TVC_LIB_PATH") or os.path.join(_HERE, "libtinyvc_hip.so")   # TVC_LIB_PATH: same-box A/B runs of two builds

_lib = None

# name -> (restype, argtypes); mirrors include/tinyvc_hip.h one to one
SIGNATURES = {
    "tvc_version": (c_int, []),
    "tvc_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "tvc_ctx_destroy": (None, [c_void_p]),
    "tvc_last_error": (c_char_p, [c_void_p]),
    "tvc_load_tensor": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "tvc_set_pitch_table": (c_int, [c_void_p, c_void_p, c_int]),
    "tvc_finalize_weights": (c_int, [c_void_p]),
    "tvc_workspace_bytes": (c_int, [c_void_p, c_int, c_int64, c_int64, POINTER(c_size_t)]),
    "tvc_stft_mag_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_size_t]),
    "tvc_energy_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_size_t]),
    "tvc_resample_out_len": (c_int64, [c_int64, c_int, c_int]),
    "tvc_resample_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int]),
    "tvc_pcm16_to_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float]),
    "tvc_f32_to_pcm16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float]),
    "tvc_encoder_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "tvc_pitch_decode_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "tvc_knn_prepared_elems": (c_int64, [c_int64]),
    "tvc_knn_prepared_elems_f16": (c_int64, [c_int64]),
    "tvc_knn_prepare_index_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "tvc_knn_forget": (c_int, [c_void_p, c_void_p]),
    "tvc_ragged_plan": (c_int, [c_int, c_int64, POINTER(c_int64), c_int, POINTER(c_int32), POINTER(c_int)]),
    "tvc_ctx_set_ragged_batch_frames": (c_int, [c_void_p, c_int]),
    "tvc_knn_prepare_index_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "tvc_knn_match_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "tvc_knn_match_general_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "tvc_knn_topk_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "tvc_knn_gather_slots_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64]),
    "tvc_knn_finish_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "tvc_shift_frequency_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float]),
    "tvc_noise_angle_from_uniform_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "tvc_decoder_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "tvc_decoder_stages_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "tvc_filter_net_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_void_p, c_size_t]),
    "tvc_dsp_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "tvc_convert_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_uint64, c_void_p, c_int, c_int64, c_void_p, c_size_t]),
    "tvc_workspace_bytes_ragged": (c_int, [c_void_p, c_int, c_int64, POINTER(c_int64), c_int64, POINTER(c_size_t)]),
    "tvc_convert_ragged_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, POINTER(c_int64), c_void_p, c_int64, c_float, c_void_p, c_uint64, c_void_p, c_int, c_void_p, c_size_t]),
    "tvc_sola_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int]),
    "tvc_stream_push_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int]),
    "tvc_profile_enable": (c_int, [c_void_p, c_int]),
    "tvc_profile_read": (c_int, [c_void_p, ctypes.c_char_p, c_size_t]),
}


class TinyVCError(RuntimeError):
    pass


def load_library():
    """Load libtinyvc_hip.so and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles the HIP runtime (SONAME libamdhip64.so.7) and must be the one copy
    # in the process; loaded after our .so it would come in as a second runtime next to /opt/rocm's.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise TinyVCError(
            f"{LIB_PATH} is missing: build it with `python -m tinyvc_amd.build` (needs hipcc). "
            "tinyvc_amd has no CPU or eager-PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
