"""Host FASTQ -> CSR packer (ppqc_fastq_scan / ppqc_fastq_pack through the C ABI, no GPU involved) against the
oracle's restatement of SeqIO's FASTQ grammar (oracle/pypore_oracle.py:parse_fastq_records)."""
import gzip
import zlib

import numpy as np
import pytest

from oracle import pypore_oracle as po
from pypore_amd import fastq_io

INT64_MIN = -(1 << 63)


def _random_fastq(seed, n, crlf=False, trailing_newline=True, blank_between=False, plus_title=False):
    rng = np.random.default_rng(seed)
    recs, out = [], []
    nl = b"\r\n" if crlf else b"\n"
    for i in range(n):
        L = int(rng.choice([0, 1, 2, 15, 16, 17, 100, 1000, int(rng.integers(1, 5000))]))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgtn", np.uint8), L))
        qual = bytes(rng.integers(33, 127, L, dtype=np.uint8))   # includes '@' and '+' as quality characters
        ch = int(rng.integers(1, 513))
        sec = int(rng.integers(0, 86400 * 3))
        ts = "2018-06-%02dT%02d:%02d:%02dZ" % (3 + sec // 86400, (sec // 3600) % 24, (sec // 60) % 60, sec % 60)
        title = ("read%d runid=abc read=%d ch=%d start_time=%s" % (i, i, ch, ts)).encode()
        recs.append((title, seq, qual, ch, 1527984000 + sec))
        out.append(b"@" + title + nl + seq + nl + b"+" + (title if plus_title else b"") + nl + qual + nl)
        if blank_between and i % 3 == 0:
            out.append(nl)
    text = b"".join(out)
    if not trailing_newline and text.endswith(nl) and n and not blank_between:
        text = text[:-len(nl)]
    return text, recs


@pytest.mark.parametrize("kw", [dict(), dict(crlf=True), dict(trailing_newline=False), dict(blank_between=True),
                                dict(plus_title=True), dict(crlf=True, blank_between=True, plus_title=True)])
def test_pack_matches_oracle_grammar(kw):
    text, recs = _random_fastq(zlib.crc32(repr(sorted(kw.items())).encode()) % 1000, 200, **kw)
    want = list(po.parse_fastq_records(text))
    assert [(t, s, q) for t, s, q, _, _ in recs] == want      # the generator and the oracle agree
    p = fastq_io.pack_fastq(text)
    assert p["n_reads"] == len(want) and p["n_bad_qual"] == 0 and p["consumed"] <= len(text)
    off = p["offsets"]
    assert off[0] == 0 and int(off[-1]) == sum(len(s) for _, s, _ in want)
    for i, (t, s, q) in enumerate(want):
        a, e = int(off[i]), int(off[i + 1])
        assert p["bases"][a:e].tobytes() == s and p["quals"][a:e].tobytes() == q, i
    assert fastq_io.read_ids(p) == [t.split()[0].decode() for t, _, _ in want]
    assert p["channel"].tolist() == [r[3] for r in recs]
    assert p["start_time"].tolist() == [r[4] for r in recs]
    # per-record GC of fastq_reader (lib/seq_routines.py:52) from the packed CSR == oracle
    gc = [po.GC(p["bases"][int(off[i]):int(off[i + 1])].tobytes()) for i in range(len(want))]
    assert gc == po.fastq_reader(text)


def test_streamed_blocks_reassemble():
    text, recs = _random_fastq(5, 300)
    whole = fastq_io.pack_fastq(text)
    rng = np.random.default_rng(1)
    cuts = sorted(set(rng.integers(1, len(text), 25).tolist())) + [len(text)]
    pos, carry, seqs, first = 0, b"", [], True
    for c in cuts:
        block = carry + text[pos:c]
        final = c == len(text)
        p = fastq_io.pack_fastq(block, final=final, first=first)
        first = False
        seqs += [p["bases"][int(p["offsets"][i]):int(p["offsets"][i + 1])].tobytes() for i in range(p["n_reads"])]
        carry, pos = block[p["consumed"]:], c
    assert seqs == [s for _, s, _, _, _ in recs] and carry.strip() == b""
    assert sum(len(s) for s in seqs) == int(whole["offsets"][-1])


def test_header_fields_absent_and_bad_quality_count():
    p = fastq_io.pack_fastq(b"@r1\nACGT\n+\nII I\n@r2 ch=x start_time=yesterday\nAC\n+\n\x7f!\n")
    assert p["channel"].tolist() == [0, 0] and p["start_time"].tolist() == [INT64_MIN, INT64_MIN]
    assert p["n_bad_qual"] == 2          # ' ' and 0x7f are outside '!'..'~' (SeqIO raises ValueError)
    with pytest.raises(ValueError):
        list(po.parse_fastq_records(b"@r1\nACGT\n+\nII I\n"))
    assert fastq_io.pack_fastq(b"")["n_reads"] == 0


@pytest.mark.parametrize("bad", [
    b"@r\nACGT\n+\nII\n",                  # lengths differ
    b"r\nACGT\n+\nIIII\n",                 # no '@'
    b"@r\nACGT\n-\nIIII\n",                # no '+'
    b"@r\nACGT\n+other\nIIII\n",           # captions differ
    b"@r\nAC GT\n+\nIIIII\n",              # whitespace in the sequence
    b"@r\nACGT\n+\n",                      # truncated
    b"\n@r\nACGT\n+\nIIII\n",              # blank first line
    b"@r\nAC\nGT\n+\nIIII\n",              # multi-line record: rejected loudly (the packer is 4-line only)
])
def test_malformed_records_raise_like_seqio(bad):
    with pytest.raises(ValueError):
        fastq_io.pack_fastq(bad)
    if not bad.startswith(b"@r\nAC\nGT"):    # SeqIO accepts multi-line records; everything else it rejects too
        with pytest.raises(ValueError):
            list(po.parse_fastq_records(bad))


def test_gz_and_directory_walk(tmp_path):
    text, recs = _random_fastq(9, 20)
    (tmp_path / "pass").mkdir()
    (tmp_path / "fail").mkdir()
    (tmp_path / "other").mkdir()
    (tmp_path / "pass" / "a.fastq").write_bytes(text)
    with gzip.open(tmp_path / "fail" / "b.fastq.gz", "wb") as fh:
        fh.write(text)
    (tmp_path / "other" / "c.fastq").write_bytes(text)
    (tmp_path / "pass" / ".hidden.fastq").write_bytes(text)
    found = sorted(fastq_io.fastq_files(str(tmp_path)))
    assert [f.split("/")[-1] for f in found] == ["b.fastq.gz", "a.fastq"] or \
        [f.split("/")[-1] for f in found] == ["a.fastq", "b.fastq.gz"]
    for f in found:
        assert fastq_io.pack_fastq(fastq_io.read_text(f))["n_reads"] == 20
    assert len(list(fastq_io.fastq_files_any(str(tmp_path)))) == 3
