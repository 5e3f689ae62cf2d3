"""FASTQ / FASTQ.gz -> CSR packer (host side of the GPU path).

The reference parses FASTQ text with Biopython (``SeqIO.parse(handle, 'fastq')``, lib/seq_routines.py:50;
``SeqIO.read(StringIO(..), 'fastq')``, lib/fastqparser.py:12).  Here a native tokenizer
(``ppqc_fastq_scan`` / ``ppqc_fastq_pack`` in libpypore_qc.so) copies line 2 / line 4 of every record
straight into the CSR buffers the kernels read, and pulls ``ch=`` / ``start_time=`` out of albacore headers.
"""
from __future__ import annotations

import ctypes as C
import gzip
import os

import numpy as np

from . import _lib as L


def read_text(path):
    """Whole file as bytes; .gz is decompressed (the reference opens plain .fastq only, lib/seq_routines.py:898)."""
    if str(path).endswith(".gz"):
        with gzip.open(path, "rb") as fh:
            return fh.read()
    with open(path, "rb") as fh:
        return fh.read()


def read_text_into(path, slot):
    """The file's bytes read STRAIGHT into the pinned slot's text arena (``readinto``: no pageable copy of a plain
    .fastq; a .gz is inflated into it).  Returns the uint8 view of the text."""
    if str(path).endswith(".gz"):
        with gzip.open(path, "rb") as fh:
            n, cap = 0, max(slot.text_capacity(), 1 << 20)
            buf = slot.text(cap)
            while True:
                got = fh.readinto(memoryview(buf)[n:])
                if not got:
                    break
                n += got
                if n == len(buf):           # full: grow (the arena is reallocated, so carry the bytes over)
                    keep = buf[:n].copy()
                    buf = slot.text(2 * len(buf))
                    buf[:n] = keep
            return buf[:n]
    size = os.path.getsize(path)
    buf = slot.text(size)
    with open(path, "rb", buffering=0) as fh:
        n = 0
        while n < size:
            got = fh.readinto(memoryview(buf)[n:])
            if not got:
                break
            n += got
    return buf[:n]


def pack_fastq(text, final=True, first=True):
    """Tokenise FASTQ text (``first``: the block starts the file, where SeqIO rejects a blank line).  Returns a dict with the CSR arrays (``bases``, ``quals``, ``offsets``), per-record
    ``channel`` (0 if the header has no ch=), ``start_time`` (INT64_MIN if absent), header spans, ``n_bad_qual``
    and ``consumed`` (bytes of complete records; the rest belongs to the next block when ``final`` is False).
    Raises ValueError on malformed records (as SeqIO would)."""
    lib = L.load()
    buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    n = int(buf.size)
    if first and n and buf[0] in (10, 13):
        raise ValueError("Records in Fastq files should start with '@' character")
    p = buf.ctypes.data_as(C.c_void_p) if n else None
    nrec, nb, used = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = lib.ppqc_fastq_scan(p, n, 1 if final else 0, C.byref(nrec), C.byref(nb), C.byref(used))
    if rc != 0:
        raise ValueError(lib.ppqc_last_error().decode())
    R, B = nrec.value, nb.value
    out = dict(bases=np.empty(B, np.uint8), quals=np.empty(B, np.uint8), offsets=np.zeros(R + 1, np.uint64),
               header_off=np.zeros(R, np.uint64), header_len=np.zeros(R, np.uint32), channel=np.zeros(R, np.uint32),
               start_time=np.zeros(R, np.int64))
    bad = C.c_uint64()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.ppqc_fastq_pack(p, used.value, R, vp(out["bases"]), vp(out["quals"]), vp(out["offsets"]),
                             vp(out["header_off"]), vp(out["header_len"]), vp(out["channel"]), vp(out["start_time"]),
                             C.byref(bad))
    if rc != 0:
        raise ValueError(lib.ppqc_last_error().decode())
    out.update(n_bad_qual=bad.value, consumed=used.value, n_reads=R, text=buf)
    return out


def read_ids(packed):
    """First whitespace-delimited token of every header (the read_id column of sequencing_summary.txt)."""
    t = packed["text"]
    ids = []
    for o, n in zip(packed["header_off"], packed["header_len"]):
        h = t[int(o):int(o) + int(n)].tobytes()
        ids.append(h.split(None, 1)[0].decode("ascii", "replace") if h else "")
    return ids


def fastq_files(folder):
    """dir_surfer in albacore mode (lib/seq_routines.py:896-901): *.fastq under pass/ or fail/ (here also .fastq.gz),
    in os.walk order."""
    for subdir, _dirs, files in os.walk(folder):
        for fi in files:
            if not fi.startswith(".") and (fi.endswith(".fastq") or fi.endswith(".fastq.gz")):
                f = os.path.join(subdir, fi)
                if "pass/" in f or "fail/" in f:
                    yield f


def fastq_files_any(folder):
    """Every *.fastq / *.fastq.gz under `folder` (no pass/ fail/ requirement)."""
    for subdir, _dirs, files in os.walk(folder):
        for fi in sorted(files):
            if not fi.startswith(".") and (fi.endswith(".fastq") or fi.endswith(".fastq.gz") or fi.endswith(".fq")):
                yield os.path.join(subdir, fi)
