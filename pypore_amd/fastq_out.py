"""``-fq yes``: write the passing reads as ``<label>.fastq.gz`` (lib/fast5_reader.py:112-114, 146-150;
lib/seq_routines.py:798-816).

The reference keeps a read when ``float(mean_qscore) >= 7.0`` (:113), re-emits it with ``SeqIO.write(.., 'fastq')``
into one BGZF file per channel and concatenates those in channel order (:801-804).  Here the pass/fail decision is the
device-computed one (flag bit 0 of the per-read record: ``mean_q < 7.0`` decided by the exact integer comparison), the
reads are ordered by channel (stable, i.e. arrival order inside a channel, as ``read_lists`` is filled before the time
sort) and written as one BGZF stream.  BGZF is the blocked gzip variant of the SAM specification (what ``Bio.bgzf``
writes): every block is a gzip member with a ``BC`` extra field holding its size, at most 64 KiB of input each, and an
empty block marks the end."""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

_BLOCK = 65280       # input bytes per block (leaves room for incompressible data inside the 64 KiB block limit)
_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


class BgzfWriter:
    def __init__(self, path, level=6):
        self._fh = open(path, "wb")
        self._buf = bytearray()
        self._level = level

    def _block(self, data):
        c = zlib.compressobj(self._level, zlib.DEFLATED, -15, zlib.DEF_MEM_LEVEL, 0)
        comp = c.compress(bytes(data)) + c.flush()
        if len(comp) + 26 > 65536:      # incompressible: store
            c = zlib.compressobj(0, zlib.DEFLATED, -15, zlib.DEF_MEM_LEVEL, 0)
            comp = c.compress(bytes(data)) + c.flush()
        bsize = len(comp) + 25
        self._fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
                       struct.pack("<II", zlib.crc32(bytes(data)) & 0xffffffff, len(data)))

    def write(self, data):
        self._buf += data
        while len(self._buf) >= _BLOCK:
            self._block(self._buf[:_BLOCK])
            del self._buf[:_BLOCK]

    def flush(self):
        """End the current block here (a block boundary is a point where BGZF streams can be cut and joined)."""
        if self._buf:
            self._block(self._buf)
            self._buf = bytearray()
        return self._fh.tell()

    def close(self, eof=True):
        self.flush()
        if eof:
            self._fh.write(_EOF)
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _write(records, path, channels=None):
    """records: iterable of (title: bytes, bases: bytes, quals: bytes).  ``channels`` (one per record, non-decreasing):
    write a PART of a multi-GPU run instead -- a block ends with every channel, no end-of-file block, and
    ``path + '.idx'`` lists (channel, first byte, end byte) so that stitch_parts can cut the part at channel boundaries."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    n = 0
    out = BgzfWriter(path)
    index, cur, start = [], None, 0
    try:
        for k, (title, s, q) in enumerate(records):
            if channels is not None and channels[k] != cur:
                pos = out.flush()
                if cur is not None:
                    index.append((int(cur), start, pos))
                cur, start = channels[k], pos
            out.write(b"@" + title + b"\n" + s + b"\n+\n" + q + b"\n")
            n += 1
        if channels is not None:
            pos = out.flush()
            if cur is not None:
                index.append((int(cur), start, pos))
    finally:
        out.close(eof=channels is None)
    if channels is not None:
        import json
        with open(path + ".idx", "w") as fh:
            json.dump(index, fh)
    return n


def stitch_parts(parts, path, n_channels):
    """The ranks' parts put together: channel by channel, ranks in order (each rank holds a contiguous share of the job's
    files, so this is the order one process would have written), then the end-of-file block.  The parts are removed."""
    import json
    idx = [{c: (a, e) for c, a, e in json.load(open(p + ".idx"))} for p in parts]
    fhs = [open(p, "rb") for p in parts]
    with open(path, "wb") as out:
        for c in sorted(set().union(*[set(i) for i in idx])):
            for i, fh in zip(idx, fhs):
                if c in i:
                    a, e = i[c]
                    fh.seek(a)
                    out.write(fh.read(e - a))
        out.write(_EOF)
    for fh, p in zip(fhs, parts):
        fh.close()
        os.remove(p)
        os.remove(p + ".idx")


def passing_order(channel, flags):
    """Indices of the reads to write: not failed (bit 0), not path-flagged (bit 1), by channel, stable."""
    keep = np.nonzero((np.asarray(flags) & 3) == 0)[0]
    return keep[np.argsort(np.asarray(channel)[keep], kind="stable")]


def write_passing(packed, per_read, path, n_channels=None, part=False):
    """``packed``: list of (file path, pack_fastq() result) in submission order; ``per_read``: read_stats() of the run."""
    titles, seqs, quals, chans = [], [], [], []
    for _p, p in packed:
        t, off = p["text"], p["offsets"]
        for i in range(p["n_reads"]):
            o, n = int(p["header_off"][i]), int(p["header_len"][i])
            a, e = int(off[i]), int(off[i + 1])
            titles.append(t[o:o + n].tobytes())
            seqs.append(p["bases"][a:e].tobytes())
            quals.append(p["quals"][a:e].tobytes())
        chans.append(p["channel"])
    chan = np.concatenate(chans) if chans else np.zeros(0, np.uint32)
    order = passing_order(chan, per_read["flags"][:len(chan)])
    return _write(((titles[i], seqs[i], quals[i]) for i in order), path, chan[order] if part else None)


def write_passing_batches(batches, per_read, path, n_channels=None, part=False):
    """Same for loader batches that kept their FASTQ titles (``titles`` list per batch)."""
    titles, seqs, quals, chans = [], [], [], []
    for b in batches:
        off = b["offsets"]
        for i in range(b["n_reads"]):
            a, e = int(off[i]), int(off[i + 1])
            titles.append(b["titles"][i])
            seqs.append(b["bases"][a:e].tobytes())
            quals.append(b["quals"][a:e].tobytes())
        chans.append(b["channel"])
    chan = np.concatenate(chans) if chans else np.zeros(0, np.uint32)
    order = passing_order(chan, per_read["flags"][:len(chan)])
    return _write(((titles[i], seqs[i], quals[i]) for i in order), path, chan[order] if part else None)
