"""Host driver of the GPU per-read QC path: one ``QCContext`` per GPU.

This is the piece that replaces the pool of ``Consumer`` processes and the per-channel tasks of the
reference (lib/seq_routines.py:916-942: ``Q.put(f5r.parsing_func(...))`` ... ``ChannelDict[...] =
result[1:6]``): loaders hand over reads packed as CSR, the library computes the per-read records
and the run-level tables on the GPU, and ``tables()`` returns what ``FastStats`` /
``result_plotting`` consume.  All compute goes through the C ABI (include/pypore_qc.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceArrayView:
    """Zero-copy view of device memory for ``torch.as_tensor(view, device='cuda')`` (RCCL all-reduce)."""

    def __init__(self, ptr, n, typestr, owner):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr or 0), False),
                                         "version": 2, "strides": None}
        self._owner = owner


SUBMIT_HOOK = None   # tests: called with (kind, {name: (address, nbytes)}) for every host batch handed to the library


class QCContext:
    DEFAULT_TUNING = {}   # tests: knobs of ``set_tuning`` applied to every new context

    def __init__(self, device=0, n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=0,
                 tz_offset_s=0, ref_time_s=0, fail_q=7.0, max_reads_hint=0, compat_flags=0):
        self._lib = L.load()
        cfg = L.Config()
        L.check(self._lib.ppqc_config_default(C.byref(cfg)))
        cfg.n_channels, cfg.n_mux, cfg.n_time_bins, cfg.bin_seconds = n_channels, n_mux, n_time_bins, bin_seconds
        cfg.time_mode, cfg.tz_offset_s, cfg.ref_time_s, cfg.fail_q = time_mode, tz_offset_s, int(ref_time_s), fail_q
        cfg.max_reads_hint = int(max_reads_hint)
        cfg.compat_flags = int(compat_flags)
        self.cfg = cfg
        self._ctx = C.c_void_p()
        L.check(self._lib.ppqc_create(C.byref(self._ctx), int(device), C.byref(cfg)))
        self.external_stream = None
        for k, v in self.DEFAULT_TUNING.items():
            self.set_tuning(k, v)
        self._keep = {}   # per staging slot: host arrays that must outlive the asynchronous submit

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        self.__dict__.pop("_dist_tensors", None)   # merge.py's tensors over the device tables (they point back at self)
        if getattr(self, "_ctx", None):
            self._lib.ppqc_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        L.check(self._lib.ppqc_reset(self._ctx))
        self._keep.clear()

    # -- submitting reads ---------------------------------------------------------------------------
    def submit(self, bases, quals, offsets, channel, mux, start_time_s, fail_flag=None, slot=0):
        """Host CSR batch (numpy arrays; see ``pack_reads``).  Asynchronous until ``sync``."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        channel = np.ascontiguousarray(channel, dtype=np.uint32)
        mux = np.ascontiguousarray(mux, dtype=np.uint8)
        t = np.ascontiguousarray(start_time_s, dtype=np.int64)
        ff = None if fail_flag is None else np.ascontiguousarray(fail_flag, dtype=np.uint8)
        R = len(offsets) - 1
        if R < 0 or len(channel) != R or len(mux) != R or len(t) != R or (ff is not None and len(ff) != R):
            raise ValueError("metadata arrays must have one entry per read")
        nbytes = int(offsets[-1]) if R > 0 else 0
        if len(bases) < nbytes or len(quals) < nbytes:
            raise ValueError("bases/quals shorter than offsets[-1]")
        b = L.Batch(R, nbytes, _ptr(bases), _ptr(quals), _ptr(offsets), _ptr(channel), _ptr(mux), _ptr(t), _ptr(ff))
        if SUBMIT_HOOK is not None:
            SUBMIT_HOOK("csr", {k: (a.ctypes.data, a.nbytes) for k, a in (("bases", bases), ("quals", quals), ("offsets", offsets),
                                                                          ("channel", channel), ("mux", mux), ("start_time", t))})
        # ppqc_submit waits for the slot's previous batch before reusing it, so only the newest batch per slot is held
        L.check(self._lib.ppqc_submit(self._ctx, C.byref(b), int(slot)))
        self._keep[int(slot)] = (bases, quals, offsets, channel, mux, t, ff)

    def submit_device(self, n_reads, n_bytes, bases_ptr, quals_ptr, offsets_ptr, channel_ptr, mux_ptr, time_ptr,
                      fail_flag_ptr=None):
        """Batch already resident in HBM (raw device pointers, e.g. ``tensor.data_ptr()``)."""
        b = L.Batch(int(n_reads), int(n_bytes), bases_ptr, quals_ptr, offsets_ptr, channel_ptr, mux_ptr, time_ptr,
                    fail_flag_ptr)
        L.check(self._lib.ppqc_submit_device(self._ctx, C.byref(b)))

    def submit_fastq(self, text, slot=0, mux=1):
        """FASTQ text (bytes / uint8 array, complete 4-line records) tokenised and reduced on the GPU.  Returns the
        number of records appended.  Channel / start time come from the ch= / start_time= header fields."""
        buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
        n = C.c_uint64()
        if SUBMIT_HOOK is not None:
            SUBMIT_HOOK("fastq", {"text": (buf.ctypes.data, buf.nbytes)})
        L.check(self._lib.ppqc_submit_fastq(self._ctx, _ptr(buf) if buf.size else None, int(buf.size), int(slot),
                                            int(mux), C.byref(n)))
        self._keep[int(slot)] = (buf,)
        return n.value

    def submit_fastq_device(self, text_ptr, n_bytes, mux=1):
        n = C.c_uint64()
        L.check(self._lib.ppqc_submit_fastq_device(self._ctx, text_ptr, int(n_bytes), int(mux), C.byref(n)))
        return n.value

    def input_errors(self):
        """dict(bad_csr_rows, bad_fastq_records, first_bad_record, bad_quality_bytes, misplaced_whitespace) found on
        the device so far."""
        out = np.zeros(5, np.uint64)
        L.check(self._lib.ppqc_input_errors(self._ctx, _ptr(out)))
        first = int(out[2])
        return dict(bad_csr_rows=int(out[0]), bad_fastq_records=int(out[1]),
                    first_bad_record=None if first == (1 << 64) - 1 else first, bad_quality_bytes=int(out[3]),
                    misplaced_whitespace=int(out[4].astype(np.int64)))

    def submit_records(self, length, mean_q, channel, mux, start_time_s, gc_count=None):
        """Reads whose statistics come from a table (albacore sequencing_summary.txt, lib/seq_routines.py:831-863)."""
        length = np.ascontiguousarray(length, dtype=np.uint32)
        mean_q = np.ascontiguousarray(mean_q, dtype=np.float64)
        channel = np.ascontiguousarray(channel, dtype=np.uint32)
        mux = np.ascontiguousarray(mux, dtype=np.uint8)
        t = np.ascontiguousarray(start_time_s, dtype=np.int64)
        gc = None if gc_count is None else np.ascontiguousarray(gc_count, dtype=np.uint32)
        n = len(length)
        if not (len(mean_q) == len(channel) == len(mux) == len(t) == n) or (gc is not None and len(gc) != n):
            raise ValueError("record columns must have equal lengths")
        r = L.Records(n, _ptr(length), _ptr(mean_q), _ptr(gc), _ptr(channel), _ptr(mux), _ptr(t))
        L.check(self._lib.ppqc_submit_records(self._ctx, C.byref(r)))

    def sync(self):
        L.check(self._lib.ppqc_sync(self._ctx))
        self._keep.clear()

    def slot_wait(self, slot):
        """Until the host batch of the newest submit on staging slot ``slot`` may be overwritten (its H2D copies are done)."""
        L.check(self._lib.ppqc_slot_wait(self._ctx, int(slot)))
        self._keep.pop(int(slot), None)

    @property
    def n_reads(self):
        n = C.c_uint64()
        L.check(self._lib.ppqc_n_reads(self._ctx, C.byref(n)))
        return n.value

    # -- results ------------------------------------------------------------------------------------
    def read_stats(self, first=0, n=None, columns=None):
        """Per-read records as a dict of numpy arrays; ``err`` is the exact fixed-point sum (Python ints).  ``columns``:
        only these (the library skips the copies of the others)."""
        if n is None:
            n = self.n_reads - first
        out = dict(length=np.zeros(n, np.uint32), gc=np.zeros(n, np.uint32), n=np.zeros(n, np.uint32),
                   qsum=np.zeros(n, np.int64), mean_q=np.zeros(n, np.float64), err_hi=np.zeros(n, np.uint64),
                   err_lo=np.zeros(n, np.uint64), flags=np.zeros(n, np.uint8))
        if columns is not None:
            out = {k: v for k, v in out.items() if k in columns}
        pr = L.PerRead(*[_ptr(out.get(k)) for k in ("length", "gc", "n", "qsum", "mean_q", "err_hi", "err_lo", "flags")])
        L.check(self._lib.ppqc_read_stats(self._ctx, C.byref(pr), int(first), int(n)))
        self._keep.clear()
        return out

    @staticmethod
    def err_exact(stats):
        return np.array([(int(h) << 32) + int(l) for h, l in zip(stats["err_hi"], stats["err_lo"])], dtype=object)

    def read_hours(self, first=0, n=None):
        if n is None:
            n = self.n_reads - first
        out = np.zeros(n, np.int32)
        L.check(self._lib.ppqc_read_hours(self._ctx, _ptr(out), int(first), int(n)))
        return out

    def _alloc_tables(self):
        nch, nmux, nb = self.cfg.n_channels, self.cfg.n_mux, self.cfg.n_time_bins
        z = lambda *s: np.empty(s, dtype=np.uint64)      # every table is overwritten by the fetch
        arr = dict(totals=z(L.N_TOTALS), len_hist_log2=z(L.N_LEN_LOG2), len_hist_lin=z(L.N_LEN_LIN),
                   q_hist=z(L.N_QHIST), gc_hist=z(L.N_GCHIST), hour_reads=z(nb), hour_bases=z(nb), chan_reads=z(nch),
                   chan_bases=z(nch), chan_fail=z(nch), cmh_reads=z(nch, nmux, nb), cmh_bases=z(nch, nmux, nb),
                   chan_min_len=z(nch), chan_max_len=z(nch))
        t = L.Tables()
        for k, a in arr.items():
            setattr(t, k, a.ctypes.data)
        return arr, t

    @staticmethod
    def _tables_dict(arr, t):
        out = {k: a.view(np.int64) for k, a in arr.items()}   # two's complement view (qsum may be negative)
        out.update(max_len=int(t.max_len), min_pos_len=int(t.min_pos_len), lin_bin_width=int(t.lin_bin_width),
                   n50=int(t.n50))
        return out

    def _tables_view(self, t):
        """Tables as zero-copy numpy views of the context's pinned host result buffer (filled by a *_view call).  The
        buffer and its layout do not move, so the arrays are made once per context; a call only fills in the scalars."""
        base = int(t.totals)
        cache = getattr(self, "_view_cache", None)
        if cache is None or cache[0] != base:
            d = L.DeviceTables()
            L.check(self._lib.ppqc_device_tables_view(self._ctx, C.byref(d)))
            words = int(d.n_sum + d.n_sel + d.n_max + d.n_min + 8)
            res = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_int64)), shape=(words,))
            nch, nmux, nb = self.cfg.n_channels, self.cfg.n_mux, self.cfg.n_time_bins
            sizes = dict(totals=L.N_TOTALS, len_hist_log2=L.N_LEN_LOG2, len_hist_lin=L.N_LEN_LIN, q_hist=L.N_QHIST,
                         gc_hist=L.N_GCHIST, hour_reads=nb, hour_bases=nb, chan_reads=nch, chan_bases=nch,
                         chan_fail=nch, cmh_reads=nch * nmux * nb, cmh_bases=nch * nmux * nb, chan_min_len=nch,
                         chan_max_len=nch)
            arrays = {}
            for k, n in sizes.items():
                o = (int(getattr(t, k)) - base) // 8
                arrays[k] = res[o:o + n]
            arrays["cmh_reads"] = arrays["cmh_reads"].reshape(nch, nmux, nb)
            arrays["cmh_bases"] = arrays["cmh_bases"].reshape(nch, nmux, nb)
            cache = self._view_cache = (base, arrays)
        out = dict(cache[1])
        out["max_len"] = int(t.max_len)
        out["min_pos_len"] = int(t.min_pos_len)
        out["lin_bin_width"] = int(t.lin_bin_width)
        out["n50"] = int(t.n50)
        return out

    def finalize(self, copy=True):
        """Single-GPU run-level tables (extremes -> tables -> N50 by radix select, or sort -> fetch).  ``copy=False`` returns views
        of the context's pinned result buffer instead of fresh arrays: valid until the next finalize / fetch of this
        context (for callers that consume the tables at once)."""
        if not copy:
            t = L.Tables()
            L.check(self._lib.ppqc_finalize_view(self._ctx, C.byref(t)))
            self._keep.clear()
            return self._tables_view(t)
        arr, t = self._alloc_tables()
        L.check(self._lib.ppqc_finalize(self._ctx, C.byref(t)))
        self._keep.clear()
        return self._tables_dict(arr, t)

    # phased finalize for multi-GPU merges
    def finalize_extremes(self):
        L.check(self._lib.ppqc_finalize_extremes(self._ctx))

    def time_range(self):
        out = np.zeros(2, np.int64)
        L.check(self._lib.ppqc_time_range(self._ctx, _ptr(out)))
        return int(out[0]), int(out[1])

    def set_ref_time(self, ref_time):
        L.check(self._lib.ppqc_set_ref_time(self._ctx, int(ref_time)))
        self.cfg.ref_time_s = int(ref_time)

    def finalize_tables(self):
        L.check(self._lib.ppqc_finalize_tables(self._ctx))

    def finalize_n50(self, lengths_ptr=None, n=0):
        L.check(self._lib.ppqc_finalize_n50(self._ctx, lengths_ptr, int(n)))

    def n50_select_hist(self, rnd):
        L.check(self._lib.ppqc_n50_select_hist(self._ctx, int(rnd)))

    def n50_select_pick(self, rnd):
        L.check(self._lib.ppqc_n50_select_pick(self._ctx, int(rnd)))

    def index_early(self):
        """True if every batch of the run left its index-only results (extremes, hours, sort keys) at submit time."""
        out = C.c_int(0)
        L.check(self._lib.ppqc_index_early(self._ctx, C.byref(out)))
        return bool(out.value)

    def stream_wait_index(self, stream_handle):
        L.check(self._lib.ppqc_stream_wait_index(self._ctx, C.c_void_p(int(stream_handle))))

    def n50_select_early(self, rnd, phase, stream_handle):
        """Histogram (phase 0) / pick (phase 1) of select round ``rnd`` on a caller's stream, independent of the tables."""
        L.check(self._lib.ppqc_n50_select_early(self._ctx, int(rnd), int(phase), C.c_void_p(int(stream_handle))))

    def cells_early(self, phase, stream_handle):
        """The (channel, mux, hour) cells on a caller's stream, beside the per-read kernel: phase 0 leaves them final in
        the ``cmh`` view of ``device_tables`` (for the all-reduce), phase 1 copies them to the host result buffer."""
        L.check(self._lib.ppqc_cells_early(self._ctx, int(phase), C.c_void_p(int(stream_handle))))

    def n50_select(self):
        """Single-GPU run of the multi-GPU N50 path (three histogram / pick rounds, no collectives in between)."""
        for r in range(3):
            self.n50_select_hist(r)
            self.n50_select_pick(r)

    def set_stream(self, stream_handle):
        """Enqueue on a caller-owned HIP stream (``torch.cuda.current_stream().cuda_stream``; 0 = null stream)."""
        L.check(self._lib.ppqc_set_stream(self._ctx, C.c_void_p(int(stream_handle)), 1))
        self.external_stream = int(stream_handle)

    def use_own_stream(self):
        L.check(self._lib.ppqc_set_stream(self._ctx, None, 0))
        self.external_stream = None

    def fetch_tables(self, copy=True):
        if not copy:
            t = L.Tables()
            L.check(self._lib.ppqc_fetch_tables_view(self._ctx, C.byref(t)))
            return self._tables_view(t)
        arr, t = self._alloc_tables()
        L.check(self._lib.ppqc_fetch_tables(self._ctx, C.byref(t)))
        return self._tables_dict(arr, t)

    def device_tables(self):
        """Zero-copy device views {sum, max, min, maxmin (int64), lengths (uint32)} plus the table offsets.
        ``min`` holds the bitwise complement of the minima, so ``maxmin`` = [max | ~min] merges with one MAX."""
        d = L.DeviceTables()
        L.check(self._lib.ppqc_device_tables_view(self._ctx, C.byref(d)))
        views = dict(sum=DeviceArrayView(d.sum_buf, d.n_sum, "<i8", self),
                     max=DeviceArrayView(d.max_buf, d.n_max, "<i8", self),
                     min=DeviceArrayView(d.min_buf, d.n_min, "<i8", self),          # complemented minima (merge by MAX)
                     maxmin=DeviceArrayView(d.max_buf, d.n_max + d.n_min, "<i8", self),
                     lengths=DeviceArrayView(d.lengths, d.n_lengths, "<i4", self),
                     sel=DeviceArrayView(d.sel_buf, d.n_sel, "<i8", self),
                     sum_sel=DeviceArrayView(d.sum_buf, d.n_sum + d.n_sel, "<i8", self),
                     # [small | cmh] = sum: the cells are all but a sliver of it and can be merged early (cells_early)
                     small=DeviceArrayView(d.sum_buf, d.off_cmh_reads, "<i8", self),
                     cmh=DeviceArrayView((d.sum_buf or 0) + 8 * d.off_cmh_reads, d.n_sum - d.off_cmh_reads, "<i8", self))
        return views, d

    def mode1_dist_tables(self):
        """time_mode 1 across ranks: zero-copy device views {first (MAX), ties (SUM), days (MAX)} (int64) that the caller
        all-reduces between the phases of ``mode1_dist_phase`` (include/pypore_qc.h, ppqc_mode1_dist_phase)."""
        m = L.Mode1Tables()
        L.check(self._lib.ppqc_mode1_dist_view(self._ctx, C.byref(m)))
        return dict(first=DeviceArrayView(m.first, m.n_first, "<i8", self), ties=DeviceArrayView(m.ties, m.n_ties, "<i8", self),
                    days=DeviceArrayView(m.days, m.n_days, "<i8", self))

    def mode1_dist_phase(self, phase):
        L.check(self._lib.ppqc_mode1_dist_phase(self._ctx, int(phase)))

    def compat_dist_count(self):
        """COMPAT_MULTI_READ_SINGLE across ranks: counts this context's reads per channel (between ``finalize_extremes``
        and ``finalize_tables``) and returns the zero-copy device view (int64) the caller all-reduces with SUM."""
        p, n = C.c_void_p(), C.c_uint64()
        L.check(self._lib.ppqc_compat_dist_count(self._ctx, C.byref(p), C.byref(n)))
        return DeviceArrayView(p.value, n.value, "<i8", self)

    def set_tuning(self, name, value=-1):
        """Test and measurement hook (ppqc_set_tuning): force one of several code paths that give the same results;
        -1 gives the choice back to the library."""
        L.check(self._lib.ppqc_set_tuning(self._ctx, name.encode(), int(value)))

    def path_count(self, name):
        """How many FASTQ texts took the named code path (ppqc_path_count)."""
        n = C.c_uint64()
        L.check(self._lib.ppqc_path_count(self._ctx, name.encode(), C.byref(n)))
        return n.value

    def bind_thread(self):
        """The context's GPU becomes the calling thread's current device (loader threads that allocate pinned memory)."""
        L.check(self._lib.ppqc_bind_thread(self._ctx))

    def timings(self):
        t = L.Timing()
        L.check(self._lib.ppqc_timings(self._ctx, C.byref(t)))
        return {k: getattr(t, k) for k, _ in L.Timing._fields_}

    def reset_timings(self):
        L.check(self._lib.ppqc_reset_timings(self._ctx))

    # -- device utilities -----------------------------------------------------------------------------
    def sort_u32_device(self, keys_ptr, n):
        L.check(self._lib.ppqc_sort_u32_device(self._ctx, keys_ptr, int(n)))

    def sort_u32_device_static(self, keys_ptr, n):
        """The same sort without any allocation (static scratch inside the library, at most 2**21 keys)."""
        L.check(self._lib.ppqc_sort_u32_device_static(self._ctx, keys_ptr, int(n)))

    def synth_fill_device(self, bases_ptr, quals_ptr, first, n, seed):
        L.check(self._lib.ppqc_synth_fill_device(self._ctx, bases_ptr, quals_ptr, int(first), int(n), int(seed)))


class Group:
    """Several contexts on several GPUs of ONE process, merged by the library itself over RCCL (ppqc_group_*): what a
    binding without torch uses.  ``finalize()`` returns the merged tables of the job."""

    def __init__(self, contexts):
        self._lib = L.load()
        self.contexts = list(contexts)
        arr = (C.c_void_p * len(self.contexts))(*[c._ctx for c in self.contexts])
        self._g = C.c_void_p()
        L.check(self._lib.ppqc_group_create(C.byref(self._g), arr, len(self.contexts)))

    def finalize(self):
        arr, t = self.contexts[0]._alloc_tables()
        L.check(self._lib.ppqc_group_finalize(self._g, C.byref(t)))
        for c in self.contexts:
            c._keep.clear()
        return QCContext._tables_dict(arr, t)

    def close(self):
        if getattr(self, "_g", None):
            self._lib.ppqc_group_destroy(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class PinnedBuffer:
    """Page-locked host memory from ``ppqc_alloc_pinned`` viewed as a numpy uint8 array: batches sliced out of it
    are copied by DMA, asynchronously, in ``submit`` (the loaders' double buffer of BASELINE config 5)."""

    def __init__(self, nbytes):
        self._lib = L.load()
        self._ptr = C.c_void_p()
        L.check(self._lib.ppqc_alloc_pinned(C.byref(self._ptr), int(nbytes)))
        self.nbytes = int(nbytes)
        self.array = np.ctypeslib.as_array(C.cast(self._ptr, C.POINTER(C.c_uint8)), shape=(max(1, self.nbytes),))[:self.nbytes]

    def close(self):
        if getattr(self, "_ptr", None):
            self.array = None
            self._lib.ppqc_free_pinned(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    n = C.c_int()
    L.check(L.load().ppqc_device_count(C.byref(n)))
    return n.value


def pack_reads(records):
    """Pack an iterable of (bases: bytes, quals: bytes) into the CSR arrays ``submit`` takes
    (the FASTQ record of lib/fastqparser.py:7-16: bases = line 2, quals = line 4)."""
    bs, qs, offs = [], [], [0]
    for b, q in records:
        if len(b) != len(q):
            raise ValueError("sequence/quality length mismatch")
        bs.append(b)
        qs.append(q)
        offs.append(offs[-1] + len(b))
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8)
    quals = np.frombuffer(b"".join(qs), dtype=np.uint8)
    return bases, quals, np.asarray(offs, dtype=np.uint64)
