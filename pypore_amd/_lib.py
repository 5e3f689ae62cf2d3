"""ctypes binding of libpypore_qc.so (include/pypore_qc.h).

There is no CPU fallback: if the shared library is missing or no HIP device is usable, the
functions here raise -- they never route through the oracle or numpy.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPQC_LIB_PATH: another build of the same library, for A/B measurements of two builds on one box
LIB_PATH = os.environ.get("PPQC_LIB_PATH") or os.path.join(_HERE, "libpypore_qc.so")

N_TOTALS = 16
N_LEN_LOG2 = 40
N_LEN_LIN = 100
N_QHIST = 8
N_GCHIST = 20
I64_MAX = (1 << 63) - 1
COMPAT_MULTI_READ_SINGLE = 1   # PPQC_COMPAT_MULTI_READ_SINGLE

TOTALS = ("n_reads", "n_bases", "n_fail", "n_gc", "n_n", "qsum", "n_binned", "bad_channel", "time_oob",
          "zero_len", "bad_mux", "no_time")

# every symbol include/pypore_qc.h declares (tests check that the built library exports all of them)
SYMBOLS = (
    "ppqc_last_error", "ppqc_abi_version", "ppqc_device_count", "ppqc_config_default", "ppqc_create",
    "ppqc_destroy", "ppqc_reset", "ppqc_alloc_pinned", "ppqc_free_pinned", "ppqc_submit", "ppqc_submit_device", "ppqc_submit_records",
    "ppqc_sync", "ppqc_n_reads", "ppqc_read_stats", "ppqc_read_hours", "ppqc_finalize_extremes",
    "ppqc_finalize_tables", "ppqc_finalize_n50", "ppqc_fetch_tables", "ppqc_fetch_tables_view", "ppqc_finalize", "ppqc_finalize_view", "ppqc_device_tables_view",
    "ppqc_timings", "ppqc_reset_timings", "ppqc_sort_u32_device", "ppqc_sort_u32_device_static", "ppqc_synth_fill_device", "ppqc_fastq_scan", "ppqc_fastq_pack",
    "ppqc_submit_fastq", "ppqc_submit_fastq_device", "ppqc_input_errors", "ppqc_time_range", "ppqc_set_ref_time", "ppqc_n50_select_hist", "ppqc_n50_select_pick", "ppqc_index_early", "ppqc_stream_wait_index", "ppqc_n50_select_early", "ppqc_cells_early", "ppqc_set_stream",
    "ppqc_mode1_dist_view", "ppqc_mode1_dist_phase", "ppqc_slot_wait", "ppqc_compat_dist_count", "ppqc_bind_thread", "ppqc_set_tuning", "ppqc_path_count",
    "ppqc_group_create", "ppqc_group_finalize", "ppqc_group_destroy",
)


class PPQCError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpypore_qc error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_channels", C.c_uint32), ("n_mux", C.c_uint32),
                ("n_time_bins", C.c_uint32), ("bin_seconds", C.c_uint32), ("time_mode", C.c_uint32),
                ("tz_offset_s", C.c_int64), ("ref_time_s", C.c_int64),
                ("fail_q", C.c_double), ("max_reads_hint", C.c_uint64), ("compat_flags", C.c_uint32),
                ("reserved0", C.c_uint32)]


class Batch(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bytes", C.c_uint64), ("bases", C.c_void_p), ("quals", C.c_void_p),
                ("offsets", C.c_void_p), ("channel", C.c_void_p), ("mux", C.c_void_p), ("start_time_s", C.c_void_p),
                ("fail_flag", C.c_void_p)]


class Records(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("length", C.c_void_p), ("mean_q", C.c_void_p), ("gc_count", C.c_void_p),
                ("channel", C.c_void_p), ("mux", C.c_void_p), ("start_time_s", C.c_void_p)]


class PerRead(C.Structure):
    _fields_ = [("length", C.c_void_p), ("gc_count", C.c_void_p), ("n_count", C.c_void_p), ("qsum", C.c_void_p),
                ("mean_q", C.c_void_p), ("err_hi", C.c_void_p), ("err_lo", C.c_void_p), ("flags", C.c_void_p)]


class Tables(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("totals", "len_hist_log2", "len_hist_lin", "q_hist", "gc_hist", "hour_reads", "hour_bases",
                 "chan_reads", "chan_bases", "chan_fail", "cmh_reads", "cmh_bases", "chan_min_len", "chan_max_len")] + \
               [(n, C.c_uint64) for n in ("max_len", "min_pos_len", "lin_bin_width", "n50")]


class DeviceTables(C.Structure):
    _fields_ = [("sum_buf", C.c_void_p), ("n_sum", C.c_uint64), ("max_buf", C.c_void_p), ("n_max", C.c_uint64),
                ("min_buf", C.c_void_p), ("n_min", C.c_uint64), ("lengths", C.c_void_p), ("n_lengths", C.c_uint64)] + \
               [(n, C.c_uint64) for n in
                ("off_totals", "off_len_log2", "off_len_lin", "off_q_hist", "off_gc_hist", "off_hour_reads",
                 "off_hour_bases", "off_chan_reads", "off_chan_bases", "off_chan_fail", "off_cmh_reads",
                 "off_cmh_bases")] + [("sel_buf", C.c_void_p), ("n_sel", C.c_uint64)]


class Mode1Tables(C.Structure):
    _fields_ = [("first", C.c_void_p), ("n_first", C.c_uint64), ("ties", C.c_void_p), ("n_ties", C.c_uint64),
                ("days", C.c_void_p), ("n_days", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [("tile_kernel_ms_total", C.c_double), ("tile_kernel_launches", C.c_uint64),
                ("tile_kernel_bytes", C.c_uint64), ("submit_ms_last", C.c_double), ("finalize_ms_last", C.c_double),
                ("sort_ms_last", C.c_double)]


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so.7; if this library pulled in
    /opt/rocm's copy first, a later `import torch` would bring up a second runtime that finds no GPU.  Loading
    torch's copy first (same soname) makes both resolve to it.  Without torch the system copy is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)


def load():
    """Load libpypore_qc.so, failing loudly if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    lib.ppqc_last_error.restype = C.c_char_p
    P = C.c_void_p
    sig = {
        "ppqc_abi_version": [],
        "ppqc_device_count": [C.POINTER(C.c_int)],
        "ppqc_config_default": [C.POINTER(Config)],
        "ppqc_create": [C.POINTER(P), C.c_int, C.POINTER(Config)],
        "ppqc_destroy": [P], "ppqc_reset": [P],
        "ppqc_alloc_pinned": [C.POINTER(P), C.c_size_t], "ppqc_free_pinned": [P],
        "ppqc_submit": [P, C.POINTER(Batch), C.c_int], "ppqc_submit_device": [P, C.POINTER(Batch)],
        "ppqc_submit_records": [P, C.POINTER(Records)],
        "ppqc_sync": [P], "ppqc_slot_wait": [P, C.c_int], "ppqc_n_reads": [P, C.POINTER(C.c_uint64)],
        "ppqc_read_stats": [P, C.POINTER(PerRead), C.c_uint64, C.c_uint64],
        "ppqc_read_hours": [P, P, C.c_uint64, C.c_uint64],
        "ppqc_finalize_extremes": [P], "ppqc_finalize_tables": [P],
        "ppqc_finalize_n50": [P, P, C.c_uint64], "ppqc_fetch_tables": [P, C.POINTER(Tables)],
        "ppqc_finalize": [P, C.POINTER(Tables)], "ppqc_finalize_view": [P, C.POINTER(Tables)],
        "ppqc_fetch_tables_view": [P, C.POINTER(Tables)], "ppqc_device_tables_view": [P, C.POINTER(DeviceTables)],
        "ppqc_timings": [P, C.POINTER(Timing)], "ppqc_reset_timings": [P],
        "ppqc_sort_u32_device": [P, P, C.c_uint64],
        "ppqc_sort_u32_device_static": [P, P, C.c_uint64],
        "ppqc_synth_fill_device": [P, P, P, C.c_uint64, C.c_uint64, C.c_uint64],
        "ppqc_submit_fastq": [P, P, C.c_uint64, C.c_int, C.c_uint8, C.POINTER(C.c_uint64)],
        "ppqc_input_errors": [P, P], "ppqc_time_range": [P, P], "ppqc_set_ref_time": [P, C.c_int64],
        "ppqc_submit_fastq_device": [P, P, C.c_uint64, C.c_uint8, C.POINTER(C.c_uint64)],
        "ppqc_n50_select_hist": [P, C.c_int], "ppqc_n50_select_pick": [P, C.c_int],
        "ppqc_index_early": [P, C.POINTER(C.c_int)], "ppqc_stream_wait_index": [P, C.c_void_p],
        "ppqc_n50_select_early": [P, C.c_int, C.c_int, C.c_void_p],
        "ppqc_cells_early": [P, C.c_int, C.c_void_p],
        "ppqc_set_stream": [P, P, C.c_int],
        "ppqc_group_create": [C.POINTER(P), C.POINTER(P), C.c_int], "ppqc_group_finalize": [P, C.POINTER(Tables)],
        "ppqc_group_destroy": [P],
        "ppqc_mode1_dist_view": [P, C.POINTER(Mode1Tables)], "ppqc_mode1_dist_phase": [P, C.c_int],
        "ppqc_compat_dist_count": [P, C.POINTER(P), C.POINTER(C.c_uint64)], "ppqc_bind_thread": [P],
        "ppqc_set_tuning": [P, C.c_char_p, C.c_int], "ppqc_path_count": [P, C.c_char_p, C.POINTER(C.c_uint64)],
        "ppqc_fastq_scan": [P, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "ppqc_fastq_pack": [P, C.c_uint64, C.c_uint64, P, P, P, P, P, P, P, C.POINTER(C.c_uint64)],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise PPQCError(rc, load().ppqc_last_error().decode("utf-8", "replace"))
