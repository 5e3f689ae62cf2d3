"""ctypes binding of oracle/libppqc_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libppqc_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


class _Cfg(C.Structure):
    _fields_ = [("n_channels", C.c_uint32), ("n_mux", C.c_uint32), ("n_time_bins", C.c_uint32),
                ("bin_seconds", C.c_uint32), ("time_mode", C.c_uint32), ("tz_offset_s", C.c_int64),
                ("ref_time_s", C.c_int64), ("fail_q", C.c_double), ("compat_flags", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_n50.restype = C.c_uint64
        _lib.orc_q_threshold.restype = C.c_uint64
        _lib.orc_q_threshold.argtypes = [C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def err_lut():
    lut = np.zeros(256, dtype=np.uint64)
    lib().orc_err_lut(_p(lut))
    return lut


def per_read(bases, quals, offsets, nthreads=1):
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    R = len(offsets) - 1
    out = dict(length=np.zeros(R, np.uint32), gc=np.zeros(R, np.uint32), n=np.zeros(R, np.uint32),
               qsum=np.zeros(R, np.int64), err_hi=np.zeros(R, np.uint64), err_lo=np.zeros(R, np.uint64),
               mean_q=np.zeros(R, np.float64))
    lib().orc_per_read(_p(bases), _p(quals), _p(offsets), C.c_uint64(R), _p(out["length"]), _p(out["gc"]),
                       _p(out["n"]), _p(out["qsum"]), _p(out["err_hi"]), _p(out["err_lo"]), _p(out["mean_q"]),
                       C.c_int(nthreads))
    out["err"] = np.array([(int(h) << 64) | int(l) for h, l in zip(out["err_hi"], out["err_lo"])], dtype=object)
    return out


def synth_fill(first, n, seed, nthreads=1):
    """(bases, quals) for stream bytes [first, first + n): host twin of the benchmark's HBM generator."""
    bases, quals = np.empty(int(n), np.uint8), np.empty(int(n), np.uint8)
    lib().orc_synth_fill(_p(bases), _p(quals), C.c_uint64(int(first)), C.c_uint64(int(n)), C.c_uint64(int(seed)),
                         C.c_int(int(nthreads)))
    return bases, quals


def n50(lengths):
    a = np.ascontiguousarray(lengths, dtype=np.uint32)
    return int(lib().orc_n50(_p(a), C.c_uint64(len(a))))


def tables(stats, channel, mux, start_time_s, fail_flag=None, cfg=None):
    from . import pypore_oracle as po
    cfg = po.default_config(**(cfg or {}))
    c = _Cfg(cfg["n_channels"], cfg["n_mux"], cfg["n_time_bins"], cfg["bin_seconds"], cfg["time_mode"],
             cfg["tz_offset_s"], cfg["ref_time_s"], cfg["fail_q"], cfg.get("compat_flags", 0))
    nch, nmux, nb = cfg["n_channels"], cfg["n_mux"], cfg["n_time_bins"]
    R = len(channel)
    channel = np.ascontiguousarray(channel, dtype=np.uint32)
    mux = np.ascontiguousarray(mux, dtype=np.uint8)
    t = np.ascontiguousarray(start_time_s, dtype=np.int64)
    ff = None if fail_flag is None else np.ascontiguousarray(fail_flag, dtype=np.uint8)
    if "err_hi" in stats:
        eh, el = stats["err_hi"], stats["err_lo"]
    else:
        eh = np.array([int(e) >> 64 for e in stats["err"]], dtype=np.uint64)
        el = np.array([int(e) & ((1 << 64) - 1) for e in stats["err"]], dtype=np.uint64)
    rec = mq = None
    if "is_record" in stats:
        rec = np.ascontiguousarray(stats["is_record"], dtype=np.uint8)
        mq = np.ascontiguousarray(stats["mean_q"], dtype=np.float64)
    z = lambda *s: np.zeros(s, dtype=np.int64)
    out = dict(totals=z(po.N_TOTALS), len_hist_log2=z(po.N_LEN_LOG2), len_hist_lin=z(po.N_LEN_LIN),
               q_hist=z(po.N_QHIST), gc_hist=z(po.N_GCHIST), hour_reads=z(nb), hour_bases=z(nb),
               chan_reads=z(nch), chan_bases=z(nch), chan_fail=z(nch), cmh_reads=z(nch, nmux, nb),
               cmh_bases=z(nch, nmux, nb), chan_min_len=z(nch), chan_max_len=z(nch))
    scal = z(4)
    lib().orc_tables(C.byref(c), C.c_uint64(R), _p(np.ascontiguousarray(stats["length"], np.uint32)),
                     _p(np.ascontiguousarray(stats["gc"], np.uint32)), _p(np.ascontiguousarray(stats["n"], np.uint32)),
                     _p(np.ascontiguousarray(stats["qsum"], np.int64)), _p(np.ascontiguousarray(eh)),
                     _p(np.ascontiguousarray(el)), _p(channel), _p(mux), _p(t),
                     _p(ff) if ff is not None else None,
                     _p(rec) if rec is not None else None, _p(mq) if mq is not None else None,
                     _p(out["totals"]), _p(out["len_hist_log2"]), _p(out["len_hist_lin"]), _p(out["q_hist"]),
                     _p(out["gc_hist"]), _p(out["hour_reads"]), _p(out["hour_bases"]), _p(out["chan_reads"]),
                     _p(out["chan_bases"]), _p(out["chan_fail"]), _p(out["cmh_reads"]), _p(out["cmh_bases"]),
                     _p(out["chan_min_len"]), _p(out["chan_max_len"]), _p(scal))
    out.update(max_len=int(scal[0]), min_pos_len=int(scal[1]), lin_bin_width=int(scal[2]), n50=int(scal[3]))
    return out
