"""ctypes binding of the C ABI in include/amt_gpu.h (amatsukaze_amd/libamt_gpu.so).

There is no CPU fallback: if the HIP library is missing or a GPU call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMTGPU_LIB") or os.path.join(HERE, "libamt_gpu.so")   # AMTGPU_LIB: instrumented builds (tools/)

c_i, c_f, c_p, c_s = C.c_int, C.c_float, C.c_void_p, C.c_char_p
c_i64, c_u64 = C.c_int64, C.c_uint64
CB = C.CFUNCTYPE(c_i, c_f, c_i, c_i, c_i)
ALLGATHER_CB = C.CFUNCTYPE(c_i, c_p, c_p, c_p, c_i64)            # (user, send, recv, bytes)
ALLREDUCE_CB = C.CFUNCTYPE(c_i, c_p, c_p, c_i64)                 # (user, buf int64*, count)


class Scatter(C.Structure):
    """AmtGpuScatter (include/amt_gpu.h)"""
    _fields_ = [("hdst", c_p), ("dst_stride", c_i64), ("src_offset", c_u64), ("chunk_bytes", c_u64), ("nchunks", c_i)]


class Collectives(C.Structure):
    """AmtGpuCollectives (include/amt_gpu.h)"""
    _fields_ = [("rank", c_i), ("world", c_i), ("allgather", ALLGATHER_CB), ("allreduce_sum_i64", ALLREDUCE_CB), ("user", c_p)]


# name -> (restype, argtypes); mirrors include/amt_gpu.h one to one
SIGNATURES = {
    "amtgpu_abi_version": (c_i, []),
    "amtgpu_host_set_parallelism": (None, [c_i, c_i]),
    "amtgpu_context_create": (c_p, [c_i]),
    "amtgpu_context_destroy": (None, [c_p]),
    "amtgpu_last_error": (c_s, [c_p]),
    "amtgpu_context_set_stream": (c_i, [c_p, c_p]),
    "amtgpu_context_get_stream": (c_p, [c_p]),
    "amtgpu_context_synchronize": (c_i, [c_p]),
    "amtgpu_stream_create_cu_range": (c_p, [c_p, c_i, c_i]),
    "amtgpu_stream_destroy": (None, [c_p, c_p]),
    "amtgpu_device_cu_count": (c_i, [c_p]),
    "amtgpu_profile_enable": (c_i, [c_p, c_i]),
    "amtgpu_profile_report": (c_i, [c_p, c_p, c_i]),
    "amtgpu_device_alloc": (c_p, [c_p, c_u64]),
    "amtgpu_device_free": (None, [c_p, c_p]),
    "amtgpu_frames_upload": (c_i, [c_p, c_p, c_p, c_u64]),
    "amtgpu_frames_upload_strided": (c_i, [c_p, c_p, c_i64, c_p, c_i64, c_u64, c_i]),
    "amtgpu_frames_upload_wait": (c_i, [c_p]),
    "amtgpu_download": (c_i, [c_p, c_p, c_p, c_u64]),
    "amtgpu_download_strided": (c_i, [c_p, c_p, c_i64, c_p, c_i64, c_u64, c_i]),
    "amtgpu_frames_upload_gather": (c_i, [c_p, c_p, c_i64, c_p, c_i64, c_u64, c_i, c_i]),
    "amtgpu_download_pinned": (c_i, [c_p, c_p, c_u64, c_p]),
    "amtgpu_context_set_upload_threads": (c_i, [c_p, c_i]),
    "amtgpu_frames_register": (c_i, [c_p, c_p, c_u64]),
    "amtgpu_frames_unregister": (c_i, [c_p, c_p]),
    "amtgpu_download_scatter": (c_i, [c_p, c_p, c_u64, c_p, c_i]),
    "amtgpu_marker_create": (c_p, [c_p]),
    "amtgpu_marker_destroy": (None, [c_p, c_p]),
    "amtgpu_marker_record_on": (c_i, [c_p, c_p]),
    "amtgpu_marker_wait_on": (c_i, [c_p, c_p]),
    "amtgpu_marker_record": (c_i, [c_p, c_i]),
    "amtgpu_marker_wait": (c_i, [c_p, c_i]),
    "amtgpu_logo_loadW": (c_p, [c_p, c_p]),
    "amtgpu_logo_saveW": (c_i, [c_p, c_p, c_p, c_s, c_i]),
    "amtgpu_logo_get_header": (c_i, [c_p, c_p, c_i, c_p]),
    "amtgpu_logo_set_header": (c_i, [c_p, c_s, c_i]),
    "amtgpu_scanlogo_fileW": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, CB]),
    "amtgpu_weave_fields_batch": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p,
                                        c_i64, c_i64, c_i, c_i, c_i]),
    "amtgpu_amts_load": (c_p, [c_p, c_s]),
    "amtgpu_amts_destroy": (None, [c_p]),
    "amtgpu_amts_get_info": (c_i, [c_p, c_p, c_p, c_p]),
    "amtgpu_amts_get_paths": (c_i, [c_p, c_p, c_i, c_p, c_i]),
    "amtgpu_amts_get_frames": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p]),
    "amtgpu_amts_weave_plan": (c_i, [c_p, c_p, c_i, c_p, c_p]),
    "amtgpu_logo_load": (c_p, [c_p, c_s]),
    "amtgpu_logo_from_planes": (c_p, [c_p] + [c_i] * 8 + [c_p]),
    "amtgpu_logo_save": (c_i, [c_p, c_p, c_s, c_s, c_i]),
    "amtgpu_logo_destroy": (None, [c_p]),
    "amtgpu_logo_get_info": (c_i, [c_p, c_p]),
    "amtgpu_logo_get_planes": (c_i, [c_p, c_p]),
    "amtgpu_logo_mask_tables": (c_i, [c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    "amtgpu_logoframe_create": (c_p, [c_p, c_p, c_i, c_f]),
    "amtgpu_logoframe_create_from_logos": (c_p, [c_p, c_p, c_i, c_f]),
    "amtgpu_logoframe_destroy": (None, [c_p]),
    "amtgpu_logoframe_begin": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i]),
    "amtgpu_logoframe_scan_batch": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i]),
    "amtgpu_logoframe_get_results": (c_i, [c_p, c_p]),
    "amtgpu_logoframe_set_results": (c_i, [c_p, c_i, c_i, c_p]),
    "amtgpu_logoframe_select_logo": (c_i, [c_p, c_i]),
    "amtgpu_logoframe_write_result": (c_i, [c_p, c_s, c_i]),
    "amtgpu_hip_runtimes_loaded": (c_i, [c_p, c_i]),
    "amtgpu_logoframe_dump_result": (c_i, [c_p, c_s]),
    "amtgpu_logoframe_decide_host": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p]),
    "amtgpu_logoframe_best_logo": (c_i, [c_p]),
    "amtgpu_logoframe_logo_ratio": (c_f, [c_p]),
    "amtgpu_analyze_create": (c_p, [c_p, c_s, c_f]),
    "amtgpu_analyze_create_from_logo": (c_p, [c_p, c_p, c_f]),
    "amtgpu_analyze_destroy": (None, [c_p]),
    "amtgpu_analyze_batch": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "amtgpu_analyze_batch_host": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "amtgpu_analyze_set_mode": (c_i, [c_p, c_i]),
    "amtgpu_analyze_last_refined": (c_i, [c_p]),
    "amtgpu_analyze_set_fixup_queue": (c_i, [c_p, c_i]),
    "amtgpu_analyze_error_bound": (c_f, [c_p, c_i, c_i]),
    "amtgpu_erase_create": (c_p, [c_p, c_s, c_s, c_i, c_i]),
    "amtgpu_erase_create_from_logo": (c_p, [c_p, c_p, c_s, c_i, c_i]),
    "amtgpu_erase_destroy": (None, [c_p]),
    "amtgpu_erase_calc_fades": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "amtgpu_erase_batch": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "amtgpu_erase_rect_batch": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "amtgpu_erase_get_rect": (c_i, [c_p, c_p]),
    "amtgpu_erase_calc_fades_device": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "amtgpu_erase_batch_dfades": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "amtgpu_erase_batch_dfades_to": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "amtgpu_erase_rect_batch_dfades": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "amtgpu_analyze_get_rect": (c_i, [c_p, c_p]),
    "amtgpu_logoframe_get_rows": (c_i, [c_p, c_p]),
    "amtgpu_logoframe_get_columns": (c_i, [c_p, c_p]),
    "amtgpu_logoscan_create": (c_p, [c_p, c_i, c_i, c_i, c_i, c_i]),
    "amtgpu_logoscan_destroy": (None, [c_p]),
    "amtgpu_logoscan_add_batch": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "amtgpu_logoscan_nframes": (c_i, [c_p]),
    "amtgpu_logoscan_get_sums": (c_i, [c_p, c_p, c_p]),
    "amtgpu_logoscan_set_sums": (c_i, [c_p, c_p, c_p, c_i]),
    "amtgpu_logoscan_get_logo": (c_p, [c_p, c_i, c_i, c_i, c_i, c_i, c_i]),
    "amtgpu_scanlogo": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_s, c_i, c_i, c_i, c_i, c_i, c_i, CB]),
    "amtgpu_scanlogo_file": (c_i, [c_p, c_s, c_i, c_s, c_s, c_i, c_i, c_i, c_i, c_i, c_i, CB]),
    "amtgpu_logoframe_allgather_results": (c_i, [c_p, c_p, c_i, c_i]),
    "amtgpu_scanlogo_sharded": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_s, c_i, c_i, c_i, c_i, c_i, c_i, CB]),
    "amtgpu_framestats_create": (c_p, [c_p, c_i, c_i, c_i]),
    "amtgpu_framestats_destroy": (None, [c_p]),
    "amtgpu_framestats_batch": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_i, c_p]),
    "amtgpu_framestats_allgather": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "amtgpu_framestats_sharded": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_p, c_i, c_i, c_i, c_p]),
    "amtgpu_cm_scene_changes": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "amtgpu_kfm_cadence": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "amtgpu_kfm_write_durations": (c_i, [c_p, c_p, c_i, c_s, c_p]),
    "amtgpu_kfm_write_timecode": (c_i, [c_p, c_p, c_i, c_i, c_i, c_s, c_p]),
    "amtgpu_cm_write_chapter_exe": (c_i, [c_p, c_i, c_i, c_s]),
}

_lib = None


def load(path: str | None = None) -> C.CDLL:
    """Loads the HIP library; raises if it is absent (the product has no other path)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
    lib = C.CDLL(p)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        f.restype = res
        f.argtypes = args
    if missing:
        raise RuntimeError("libamt_gpu.so lacks symbols declared in include/amt_gpu.h: " + ", ".join(missing))
    if path is None:
        _lib = lib
    return lib
