"""Pinned, overlapped staging between the loaders and the GPU (BASELINE.json north_star: "loaders that pack reads into
a CSR layout of pinned buffers, double-buffered H2D").

The reference's loaders hand every read to a worker as Python strings (lib/fast5_reader.py:36-71,
lib/multi_read_fast5_reader.py:121-147, lib/seq_routines.py:896-901) and its pool pulls the next task while earlier
results are merged (:916-942).  Here a loader thread writes each batch STRAIGHT into page-locked host memory
(``ppqc_alloc_pinned``), as the CSR arrays ``ppqc_submit`` takes or as FASTQ text for ``ppqc_submit_fastq``, while the main
thread hands the previous batch to the library: one host slot per loader thread plus two, over the library's two device
staging slots, so that slots are being filled while one is being copied and one has its kernels running.  A host slot goes back to the loader when
``ppqc_slot_wait`` says its copies are done.  No statistics are computed here."""
from __future__ import annotations

import queue
import threading
import weakref

import numpy as np

from . import qc

LIVE_SLOTS = weakref.WeakSet()   # every PinnedSlot alive (tests ask whether a pointer handed to the library lies in one)


def in_pinned_slot(address, nbytes=1):
    return any(s.owns(address, nbytes) for s in list(LIVE_SLOTS))


class PinnedSlot:
    """One host slot: three page-locked arenas (bases / text, qualities, per-read metadata) that grow on demand and are
    reused from batch to batch."""

    def __init__(self):
        self._b = self._q = self._m = None
        LIVE_SLOTS.add(self)

    @staticmethod
    def _ensure(buf, nbytes):
        if buf is not None and buf.nbytes >= nbytes:
            return buf
        if buf is not None:
            buf.close()
        return qc.PinnedBuffer(max(4096, nbytes + nbytes // 2))

    def csr(self, n_bytes, n_reads):
        """Views {bases, quals, offsets, start_time, channel, mux, fail_flag} for a batch of this size."""
        self._b = self._ensure(self._b, n_bytes)
        self._q = self._ensure(self._q, n_bytes)
        need = 8 * (n_reads + 1) + 8 * n_reads + 4 * n_reads + 2 * n_reads
        self._m = self._ensure(self._m, need)
        m, o = self._m.array, 0

        def take(nb, dt):
            nonlocal o
            v = m[o:o + nb].view(dt)
            o += nb
            return v
        return dict(bases=self._b.array[:n_bytes], quals=self._q.array[:n_bytes], offsets=take(8 * (n_reads + 1), np.uint64),
                    start_time=take(8 * n_reads, np.int64), channel=take(4 * n_reads, np.uint32),
                    mux=take(n_reads, np.uint8), fail_flag=take(n_reads, np.uint8))

    def text(self, n_bytes):
        self._b = self._ensure(self._b, n_bytes)
        return self._b.array[:n_bytes]

    def text_capacity(self):
        return 0 if self._b is None else self._b.nbytes

    def owns(self, address, nbytes=1):
        for buf in (self._b, self._q, self._m):
            if buf is not None and buf.array is not None:
                base = buf.array.ctypes.data
                if base <= address and address + nbytes <= base + buf.nbytes:
                    return True
        return False

    def close(self):
        for buf in (self._b, self._q, self._m):
            if buf is not None:
                buf.close()
        self._b = self._q = self._m = None


def pack_records(recs, slot=None):
    """CSR batch from loader records ``(channel, mux, start_time, fail, title, bases, quals[, albacore])`` -- into the pinned slot's
    arenas when one is given (each read is copied once, from the bytes h5py returned into page-locked memory), else
    into fresh arrays."""
    n = len(recs)
    lens = np.fromiter((len(r[5]) for r in recs), np.int64, n)
    nb = int(lens.sum())
    if slot is not None:
        v = slot.csr(nb, n)
    else:
        v = dict(bases=np.empty(nb, np.uint8), quals=np.empty(nb, np.uint8), offsets=np.empty(n + 1, np.uint64),
                 start_time=np.empty(n, np.int64), channel=np.empty(n, np.uint32), mux=np.empty(n, np.uint8),
                 fail_flag=np.empty(n, np.uint8))
    off = v["offsets"]
    off[0] = 0
    if n:
        np.cumsum(lens, out=off[1:].view(np.int64))
    b, q, o = v["bases"], v["quals"], 0
    for r, ln in zip(recs, lens):
        if ln:
            b[o:o + ln] = np.frombuffer(r[5], np.uint8)
            q[o:o + ln] = np.frombuffer(r[6], np.uint8)
            o += ln
    v["channel"][:] = [r[0] for r in recs]
    v["mux"][:] = [r[1] for r in recs]
    v["start_time"][:] = [r[2] for r in recs]
    v["fail_flag"][:] = [r[3] for r in recs]
    v["n_reads"] = n
    v["titles"] = [r[4] for r in recs]
    # albacore's sequence_length / mean_qscore where the loader found them (-1 / NaN where not): host-side only, for the
    # cross-check of seq_routines.albacore_cross_check; never handed to the library
    alb = [r[7] if len(r) > 7 and r[7] is not None else (None, None) for r in recs]
    v["albacore_len"] = np.array([-1 if a[0] is None else a[0] for a in alb], np.int64)
    v["albacore_q"] = np.array([np.nan if a[1] is None else a[1] for a in alb], np.float64)
    return v


class Stage:
    """Loader threads + pinned host slots.  ``fill(job, slot)`` runs on a loader thread and returns the item to submit
    (or None to skip the job); ``submit(item, device_slot)`` runs on the caller's thread, in the order of the jobs, and
    must hand the item to ``ctx.submit`` / ``ctx.submit_fastq`` with that staging slot (or return False if it did not use
    the GPU slot).  ``threads`` loaders fill slots side by side (reading a file into page-locked memory and inflating a
    .gz release the GIL): two more host slots than loaders, so that one batch can be in the copy engine and one under
    the kernels while every loader fills."""

    def __init__(self, ctx, threads=1):
        self.ctx = ctx
        self.threads = max(1, int(threads))
        self.slots = [PinnedSlot() for _ in range(self.threads + 2)]

    def run(self, jobs, fill, submit):
        free, ready = queue.Queue(), queue.Queue()
        for k in range(len(self.slots)):
            free.put(k)
        stop = threading.Event()
        take = threading.Lock()
        it = iter(jobs)
        state = {"next": 0, "live": self.threads}

        def loader():
            try:
                # pinned arenas are allocated from this thread: on the context's GPU, not on a fresh thread's device 0
                if hasattr(self.ctx, "bind_thread"):
                    self.ctx.bind_thread()
                while True:
                    # a free slot FIRST, then the next job, both under one lock: the earlier job always has its slot, so
                    # the caller, who takes the items in order, never waits for a job whose loader waits for a slot
                    with take:
                        k = free.get()
                        if stop.is_set():
                            free.put(k)
                            return
                        try:
                            job = next(it)
                        except StopIteration:
                            free.put(k)
                            break
                        seq = state["next"]
                        state["next"] += 1
                    item = fill(job, self.slots[k])
                    ready.put((seq, k, item))
                with take:
                    state["live"] -= 1
                    if state["live"] == 0:
                        ready.put(None)
            except BaseException as e:      # handed to the caller's thread
                ready.put(e)

        ths = [threading.Thread(target=loader, name="ppqc-loader-%d" % i, daemon=True) for i in range(self.threads)]
        for th in ths:
            th.start()
        in_flight, n, want, early = [], 0, 0, {}
        try:
            done = False
            while not done or early:
                if want in early:
                    k, item = early.pop(want)
                else:
                    got = ready.get()
                    if got is None:
                        done = True
                        continue
                    if isinstance(got, BaseException):
                        raise got
                    seq, k, item = got
                    if seq != want:
                        early[seq] = (k, item)
                        continue
                want += 1
                if item is None:            # the job yielded nothing
                    free.put(k)
                    continue
                d = n % 2
                used = submit(item, d)
                if used is False:           # the submit did not use the GPU slot: the host slot is free at once
                    free.put(k)
                    continue
                n += 1
                in_flight.append((k, d))
                if len(in_flight) >= 2:     # the older batch's copies: done, or about to be (its slot is the other one)
                    k0, d0 = in_flight.pop(0)
                    self.ctx.slot_wait(d0)
                    free.put(k0)
        finally:
            stop.set()
            for _ in ths:
                free.put(0)                 # wakes loaders that wait for a slot
            for th in ths:
                th.join()
        for _k, d in in_flight:
            self.ctx.slot_wait(d)

    def close(self):
        for s in self.slots:
            s.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
