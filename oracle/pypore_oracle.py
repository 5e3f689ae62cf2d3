"""CPU oracle for the per-read QC statistics path of PyPore `seqstats`.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pypore_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and there only as the checker.

PARITY UNPINNED: the reference (rsemeraro/PyPore, pure Python 2.7) ships
no tests, fixtures or golden vectors, and it cannot be imported here (Python-2 semantics; h5py,
Biopython, pysam absent; no python2).  This file is therefore a line-by-line Python-3 restatement
of the reference *source*, with Python-2 semantics (``round`` half-away-from-zero, ``str(float)``
with 12 significant digits, integer ``/``) emulated explicitly.  Every function cites the
reference file:line it follows (paths relative to /root/reference).  The statistics the
reference never computes itself (Phred decode, ONT mean quality, N count, N50, explicit
histograms) follow the definitions pinned in SURVEY.md §8a-11.

Two layers:
  * "reference-shaped" functions (``GC``, ``get_content_tuple``, ``parsing_func``,
    ``channel_parser``, ``summary_reader``, ``FastStats_data``, ``result_plotting_data`` ...)
    operating on the same string tuples / dicts the reference passes around;
  * "table-shaped" functions (``per_read_stats``, ``build_tables``) that produce exactly the
    arrays the C-ABI returns, written with numpy so they finish in seconds at 1e5..1e6 reads.
    ``tests/test_oracle_*.py`` proves the two layers agree.
"""
from __future__ import annotations

import math
from decimal import Decimal, ROUND_HALF_UP

import numpy as np

# ----------------------------------------------------------------------------------------------
# Python-2 semantics helpers
# ----------------------------------------------------------------------------------------------


def py2_round(x, ndigits=0):
    """Python 2 ``round``: half away from zero, returns float (used at seq_routines.py:117,138,154)."""
    if x != x or x in (float("inf"), float("-inf")):
        return x
    q = Decimal(1).scaleb(-ndigits)
    d = Decimal(repr(float(x)))
    r = d.copy_abs().quantize(q, rounding=ROUND_HALF_UP)
    return float(-r if d < 0 else r)


def py2_str_float(x):
    """Python 2 ``str(float)``: ``'%.12g'`` plus a trailing ``.0`` when the result looks integral
    (fast5_reader.py:68 turns every field into such a string)."""
    s = "%.12g" % float(x)
    if "." not in s and "e" not in s and "n" not in s and "i" not in s:
        s += ".0"
    return s


def py2_int_div(a, b):
    """Python 2 ``int / int`` (floor division), seq_routines.py:512."""
    return a // b


# ----------------------------------------------------------------------------------------------
# 8a-2  Bio.SeqUtils.GC  (third-party Biopython <= 1.76, not under /root/reference; call sites
#       fast5_reader.py:58, multi_read_fast5_reader.py:107, seq_routines.py:52)
# ----------------------------------------------------------------------------------------------

_GC_SYMBOLS = b"GCgcSs"


def gc_count(seq: bytes) -> int:
    """Number of G/C/S symbols, either case: ``sum(seq.count(x) for x in 'GCgcSs')``."""
    if isinstance(seq, str):
        seq = seq.encode("latin-1")
    return sum(seq.count(bytes([x])) for x in _GC_SYMBOLS)


def GC(seq) -> float:
    """Published algorithm of ``Bio.SeqUtils.GC``: percentage of the *full* length, 0.0 if empty."""
    if isinstance(seq, str):
        seq = seq.encode("latin-1")
    try:
        return gc_count(seq) * 100.0 / len(seq)
    except ZeroDivisionError:
        return 0.0


# ----------------------------------------------------------------------------------------------
# 8a-11  statistics named by north_star that the reference does not compute itself
# ----------------------------------------------------------------------------------------------

PHRED_OFFSET = 33
ERR_SHIFT = 52  # fixed-point scale of the error-probability LUT: e[b] = round(2^52 * 10^(-q/10))


def phred_decode(quals: bytes):
    """(i) Sanger FASTQ: q = byte - 33 (what SeqIO 'fastq' enforces at seq_routines.py:50)."""
    return [b - PHRED_OFFSET for b in quals]


def n_count(seq: bytes) -> int:
    """(iv) number of N / n symbols."""
    return seq.count(b"N") + seq.count(b"n")


def mean_q_ont(quals: bytes) -> float:
    """(iii) albacore/ONT mean_qscore: -10*log10(mean(10^(-q/10))), 0.0 for an empty read.
    This is the *specification* in plain fp64; the device computes it through the exact
    fixed-point sum below and must agree within 1e-6 relative."""
    if len(quals) == 0:
        return 0.0
    s = 0.0
    for b in quals:
        s += 10.0 ** (-(b - PHRED_OFFSET) / 10.0)
    return -10.0 * math.log10(s / len(quals))


def err_lut():
    """256-entry LUT of the fixed-point error probability.  Bytes below '!' saturate to q=0 and
    bytes above '~' to q=93 (such bytes are not valid Sanger FASTQ; Biopython would reject the
    record, seq_routines.py:50).  Entry = round-half-even(2^52 * pow(10, -q/10)) in fp64, the same
    expression the library's host code uses to build the table it uploads."""
    lut = np.zeros(256, dtype=np.uint64)
    for b in range(256):
        q = min(max(b - PHRED_OFFSET, 0), 93)
        lut[b] = int(round(math.ldexp(math.pow(10.0, -q / 10.0), ERR_SHIFT)))
    return lut


_ERR_LUT = err_lut()


def q_threshold(q: float) -> int:
    """Fixed-point image of a quality threshold: a read of L bases has mean_q >= q  iff
    S <= L * q_threshold(q)  (exact integer comparison; this is how fail / histogram bins are
    decided so that they are bit-reproducible)."""
    return int(round(math.ldexp(math.pow(10.0, -q / 10.0), ERR_SHIFT)))


def err_sum_fixed(quals: bytes) -> int:
    """Exact integer S = sum_i e[quals[i]] (Python big int)."""
    if len(quals) == 0:
        return 0
    a = np.frombuffer(quals, dtype=np.uint8)
    # sum in Python ints to avoid any uint64 overflow for very long reads
    cnt = np.bincount(a, minlength=256)
    return int(sum(int(c) * int(e) for c, e in zip(cnt, _ERR_LUT)))


def split_err(S: int):
    """Device representation of S: S = err_hi * 2^32 + err_lo with both halves being *sums* of the
    per-lane 32-bit halves, so neither is canonical -- compare S, not the halves."""
    return S >> 32, S & 0xFFFFFFFF


def mean_q_from_fixed(S: int, length: int) -> float:
    """mean_q as the library evaluates it from the exact sum: the sum is rounded once to fp64."""
    if length == 0:
        return 0.0
    return -10.0 * math.log10(float(S) / (float(length) * float(1 << ERR_SHIFT)))


def n50(lengths) -> int:
    """(vi) sort descending, running sum c; first length with 2*c >= total; 0 if empty."""
    ls = sorted((int(x) for x in lengths), reverse=True)
    total = sum(ls)
    if total == 0:
        return 0
    c = 0
    for x in ls:
        c += x
        if 2 * c >= total:
            return x
    return 0


def per_read_stats_py(bases: bytes, quals: bytes):
    """Pure-Python per-read record (small cases)."""
    assert len(bases) == len(quals)
    L = len(bases)
    S = err_sum_fixed(quals)
    return dict(length=L, gc=gc_count(bases), n=n_count(bases),
                qsum=sum(quals) - PHRED_OFFSET * L, err=S,
                mean_q=mean_q_from_fixed(S, L), mean_q_spec=mean_q_ont(quals))


# ----------------------------------------------------------------------------------------------
# numpy per-read statistics over a CSR batch
# ----------------------------------------------------------------------------------------------


def per_read_stats(bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray):
    """Vectorised per-read statistics over CSR arrays (uint8, uint8, uint64[R+1]).

    Returns dict of arrays: length u32, gc u32, n u32, qsum i64, err (object array of Python ints,
    exact), mean_q f64.
    """
    offsets = np.asarray(offsets, dtype=np.uint64).astype(np.int64)
    R = len(offsets) - 1
    nb = int(offsets[-1])
    bases = np.asarray(bases[:nb], dtype=np.uint8)
    quals = np.asarray(quals[:nb], dtype=np.uint8)
    length = (offsets[1:] - offsets[:-1]).astype(np.int64)

    folded = bases & np.uint8(0xDF)
    is_gc = ((folded == ord("G")) | (folded == ord("C")) | (folded == ord("S"))).astype(np.int64)
    is_n = (folded == ord("N")).astype(np.int64)

    def seg_sum(x):
        c = np.concatenate(([0], np.cumsum(x, dtype=np.int64)))
        return c[offsets[1:]] - c[offsets[:-1]]

    gc = seg_sum(is_gc)
    nn = seg_sum(is_n)
    qsum = seg_sum(quals.astype(np.int64)) - PHRED_OFFSET * length
    # exact fixed-point sum: split the LUT in 26-bit halves so int64 cumsums cannot overflow
    # (2^26 * 2^37 bytes headroom), then recombine in Python ints.
    lut = _ERR_LUT.astype(np.int64)
    lo = (lut & ((1 << 26) - 1))[quals]
    hi = (lut >> 26)[quals]
    slo = seg_sum(lo)
    shi = seg_sum(hi)
    err = np.array([(int(h) << 26) + int(l) for h, l in zip(shi, slo)], dtype=object)
    mean_q = np.array([mean_q_from_fixed(int(e), int(l)) for e, l in zip(err, length)], dtype=np.float64)
    return dict(length=length.astype(np.uint32), gc=gc.astype(np.uint32), n=nn.astype(np.uint32),
                qsum=qsum.astype(np.int64), err=err, mean_q=mean_q)


# ----------------------------------------------------------------------------------------------
# 8a-1 / 8a-3  per-read tuple  (fast5_reader.py:36-71, multi_read_fast5_reader.py:89-119)
# ----------------------------------------------------------------------------------------------


def get_content_tuple(channel, mux, start_time_s, read_length, qscore, gc_pct, fail=False):
    """The 6-tuple of *strings* the reference builds per read (fast5_reader.py:68).  Reads whose
    path contains 'fail' keep the defaults ReadLength=0, QScore=6.99, GCcont=0 (:37-40, 52)."""
    if fail:
        read_length, qscore, gc_pct = 0, 6.99, 0
    def s(v):
        return py2_str_float(v) if isinstance(v, float) else str(v)
    return (str(channel), str(mux), py2_str_float(float(start_time_s)), s(read_length), s(qscore), s(gc_pct))


def local_hour(t_minus_ref: float, tz_offset_s: int = 0) -> int:
    """``int(time.strftime('%H', time.localtime(x)))`` for a machine whose local zone is a fixed
    ``tz_offset_s`` east of UTC (fast5_reader.py:118).  localtime floors fractional seconds."""
    sec = math.floor(t_minus_ref) + tz_offset_s
    return (sec // 3600) % 24


def _hour_wrap(time_vec):
    """Sequential wrap fix, fast5_reader.py:122-126 (index 0 and 1 are never fixed: ``e > 1``)."""
    tv = list(time_vec)
    for e in range(len(tv)):
        hour = tv[e]
        if e > 1:
            if hour < tv[e - 1]:
                tv[e] = hour + 24
    return tv


# ----------------------------------------------------------------------------------------------
# 8a-4  per-channel aggregation  (fast5_reader.py:88-158  ==  multi_read_fast5_reader.py:44-79)
# ----------------------------------------------------------------------------------------------


def parsing_func(tuples, ref_start, tz_offset_s=0, multi_read_single_quirk=False):
    """``tuples``: list of the 6-string tuples of ONE channel, in arrival order.

    Returns ``[ReadPerChannel, BasesPerChannel, FailedReads, ChrList, MuxProductivity]`` with the
    reference's types (ints here where the reference str()-ifies them at :152-155; its consumers
    int() them back).

    ``multi_read_single_quirk`` reproduces multi_read_fast5_reader.py:64-67 where a channel holding
    exactly one read skips the mux filter, so that read is counted once per mux k in 1..4.
    """
    ChrList = [t[1:] for t in tuples]                       # (mux, time, len, q, gc)  :108
    # stable sort by the start-time STRING (:117)
    order = [y[0] for y in sorted(enumerate([c[1] for c in ChrList]), key=lambda z: z[1])]
    ChrList = [ChrList[z] for z in order]
    TimeVec = [local_hour(float(c[1]) - float(ref_start), tz_offset_s) for c in ChrList]   # :118
    TimeVec = _hour_wrap(TimeVec)                                                          # :122-126
    ChrList = [(TimeVec[i], c[2], c[3], c[0], c[4]) for i, c in enumerate(ChrList)]        # :127
    ReadPerChannel = len(ChrList)                                                          # :128
    BasesPerChannel = sum(int(c[1]) for c in ChrList)                                      # :129
    FailedReads = 0
    MuxProductivity = {str(il): [] for il in range(1, 5)}                                  # :130
    for k in MuxProductivity.keys():
        MuxProd = {}
        MuxBase = {}
        if multi_read_single_quirk and len(ChrList) <= 1:
            MucList = ChrList
        else:
            MucList = [c for c in ChrList if c[3] == k]                                    # :134
        for hr, val, qual, muc, gcs in MucList:
            if float(qual) < 7.0:                                                          # :136
                FailedReads += 1
            if str(hr) in MuxProd:
                MuxProd[str(hr)] += int(val)
                MuxBase[str(hr)] += 1
            else:
                MuxProd[str(hr)] = int(val)
                MuxBase[str(hr)] = 1
        # {hour: (n_reads, n_bases)}   (:145; the reference's dict names are swapped)
        MuxProductivity[k].append({h: (MuxBase[h], MuxProd[h]) for h in MuxBase})
    return [ReadPerChannel, BasesPerChannel, FailedReads, ChrList, MuxProductivity]


def build_channel_dict(read_tuples, ref_start, n_channels=512, tz_offset_s=0,
                       multi_read_single_quirk=False):
    """seq_routines.py:905-942 (fast5module) / :989-1022 (multifast5reader): bucket the per-read
    tuples by ``int(channel) - 1`` and run the per-channel aggregation; inactive channels keep ()."""
    ChannelDict = {str(il): () for il in range(1, n_channels + 1)}
    lis = [[] for _ in range(n_channels)]
    for t in read_tuples:
        lis[int(t[0]) - 1].append(t)
    for e, chunk in enumerate(lis):
        if chunk != []:
            ChannelDict[str(e + 1)] = tuple(parsing_func(chunk, ref_start, tz_offset_s,
                                                         multi_read_single_quirk))
    return ChannelDict


# ----------------------------------------------------------------------------------------------
# 8a-6  albacore sequencing_summary.txt  (seq_routines.py:831-863)
# ----------------------------------------------------------------------------------------------


def summary_reader(lines, n_channels=512, tz_offset_s=0):
    """``lines``: iterable of text lines of sequencing_summary.txt, header first."""
    ChannelDict = {str(il): () for il in range(1, n_channels + 1)}
    ref_start = 1000000000
    lis = [[] for _ in range(n_channels)]
    it = iter(lines)
    next(it)                                                                 # :837
    for line in it:
        f = line.split()
        if not f:
            continue
        idx = int(f[3]) - 1                                                  # :839
        if float(f[4]) < ref_start:                                          # :840
            ref_start = float(f[4])
        lis[idx].append(f[3:-5])                                             # :842
    for l in lis:
        if l != []:
            order = [y[0] for y in sorted(enumerate([r[1] for r in l]), key=lambda z: float(z[1]))]
            l[:] = [l[z] for z in order]                                     # :845
            TimeVec = _hour_wrap([local_hour(float(r[1]) - float(ref_start), tz_offset_s) for r in l])
            chr_list = [(TimeVec[i], r[3], r[-1]) for i, r in enumerate(l)]  # :852
            ReadPerChannel = len(chr_list)
            BasesPerChannel = sum(int(c[1]) for c in chr_list)
            FailedReads = sum(float(c[2]) < 7.00 for c in chr_list)          # :855
            hour_table = {}
            for h in set(c[0] for c in chr_list):
                hour_tab = [c for c in chr_list if c[0] == int(h)]
                hour_table[str(h)] = (len(hour_tab), sum(int(c[1]) for c in hour_tab))
            ChannelDict[str(l[0][0])] = (ReadPerChannel, BasesPerChannel, FailedReads, chr_list, hour_table)
    return ChannelDict, ref_start


# ----------------------------------------------------------------------------------------------
# 8a-5  fastq_reader  (seq_routines.py:40-53): FASTQ text -> per-record GC
# ----------------------------------------------------------------------------------------------


def parse_fastq_records(text: bytes, check_quality=True):
    """FASTQ tokenizer with the semantics of ``SeqIO.parse(handle, 'fastq')`` (seq_routines.py:50,
    fastqparser.py:12).  The parser is third-party (Biopython, unpinned in setup.py:9, <= 1.76 for
    Python 2.7; not under /root/reference); this restates its published record grammar
    (Bio.SeqIO.QualityIO.FastqGeneralIterator + the Sanger quality check of FastqPhredIterator):

      * a record starts with an '@' line (title = rest, right-stripped);
      * sequence lines follow until a line starting with '+' (an optional repeated title must
        match); space or tab inside the sequence is an error;
      * quality lines follow until the quality is at least as long as the sequence AND the next
        line starts with '@' (or the file ends); blank lines there add nothing;
      * lengths must match; with ``check_quality`` every quality byte must be '!'..'~'.
    Lines are read in universal-newline mode ("rU", seq_routines.py:49).  Yields
    (title, bases, quals); raises ValueError where SeqIO would."""
    lines = text.replace(b"\r\n", b"\n").replace(b"\r", b"\n").split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()          # the text after the final newline is not a line
    i, n = 0, len(lines)
    if n == 0:
        return
    while i < n:
        line = lines[i]; i += 1
        if not line.startswith(b"@"):
            raise ValueError("Records in Fastq files should start with '@' character")
        title = line[1:].rstrip()
        seq = lines[i].rstrip() if i < n else b""
        i += 1
        while True:
            if i >= n:
                raise ValueError("End of file without quality information.")
            line = lines[i]; i += 1
            if line.startswith(b"+"):
                second = line[1:].rstrip()
                if second and second != title:
                    raise ValueError("Sequence and quality captions differ.")
                break
            seq += line.rstrip()
        if b" " in seq or b"\t" in seq:
            raise ValueError("Whitespace is not allowed in the sequence.")
        qual = lines[i].rstrip() if i < n else b""
        i += 1
        while i < n:
            line = lines[i]
            if line.startswith(b"@") and len(qual) >= len(seq):
                break
            qual += line.rstrip()
            i += 1
        if len(seq) != len(qual):
            raise ValueError("Lengths of sequence and quality values differs for %r (%i and %i)."
                             % (title, len(seq), len(qual)))
        if check_quality and any(c < 33 or c > 126 for c in qual):
            raise ValueError("Invalid character in quality string")
        yield title, seq, qual


def fastq_reader(text: bytes):
    """GC_list of one FASTQ file (seq_routines.py:47-53)."""
    return [GC(s) for _, s, _ in parse_fastq_records(text)]


# ----------------------------------------------------------------------------------------------
# 8a-7  FastStats data part  (seq_routines.py:97-144, 154-155, 207)
# ----------------------------------------------------------------------------------------------


def FastStats_data(ChannelOut, GCTable=None, n_hours=48):
    HoursDict = {str(il): [0, 0, 0] for il in range(0, n_hours)}                         # :99
    SmallDict = dict((k, v) for k, v in ChannelOut.items() if len(v) > 0)                # :100
    FailSum = sum(int(SmallDict[str(x)][2]) for x in SmallDict.keys())                   # :101
    if GCTable is None:
        for c in SmallDict:
            Sd = dict((k, v) for k, v in ChannelOut[str(c)][4].items() if len(v) > 0)
            for k, v in Sd.items():
                for o, r in v[0].items():
                    HoursDict[o][0] += r[0]
                    HoursDict[o][1] += r[1]
    else:
        for c in SmallDict:
            Sd = dict((k, v) for k, v in ChannelOut[str(c)][4].items() if len(v) > 0)
            for o, r in Sd.items():
                HoursDict[o][0] += r[0]
                HoursDict[o][1] += r[1]
    PassSum = sum(int(HoursDict[x][0]) for x in HoursDict.keys())                        # :115
    for x in sorted(int(k) for k, v in HoursDict.items() if v != [0, 0, 0]):             # :116
        HoursDict[str(x)][2] = py2_str_float(py2_round(float(HoursDict[str(x)][1]) / float(HoursDict[str(x)][0])))
    MaxMinvec = []
    LengthsVec = [0]
    QualsVec = [0]
    GCVec = [0] if GCTable is None else list(GCTable)
    for c in ChannelOut:
        try:
            Bait = [int(r[1]) for r in ChannelOut[c][3]]                                 # :127
            if not Bait:
                raise IndexError
            LengthsVec.extend(Bait)
            QualsVec.extend(float(r[2]) for r in ChannelOut[c][3])
            if GCTable is None:
                GCVec.extend(float(r[4]) for r in ChannelOut[c][3])
            MaxMinvec.append(max(Bait))
            MaxMinvec.append(min(Bait))
        except Exception:
            pass
    pos = [a for a in MaxMinvec if a > 0]
    SmallestRead = min(pos) if pos else None                                             # :135 (raises in py2)
    GCVec = [a for a in GCVec if a > 0]                                                  # :136
    BiggestRead = max(MaxMinvec) if MaxMinvec else None
    keys = sorted(int(k) for k, v in HoursDict.items() if v != [0, 0])                   # always all hours
    MeanReadLength = py2_round(float(np.mean([float(HoursDict[str(x)][2]) for x in keys])), 1)   # :138
    TotalThroughput = sum(float(HoursDict[str(x)][1]) for x in keys)                     # :140
    hours = [HoursDict[str(h)] for h in range(n_hours)]                                  # :142-144
    denom = FailSum + PassSum
    pie = None
    if denom:
        pie = [py2_round((float(PassSum) / denom) * 100, 2), py2_round((float(FailSum) / denom) * 100, 2)]  # :154-155
    table = [FailSum + PassSum, PassSum, FailSum, TotalThroughput, MeanReadLength, SmallestRead, BiggestRead]  # :207
    return dict(FailSum=FailSum, PassSum=PassSum, HoursDict=hours, table=table, pie=pie,
                LengthsVec=LengthsVec, QualsVec=QualsVec, GCVec=GCVec,
                SmallestRead=SmallestRead, BiggestRead=BiggestRead,
                MeanReadLength=MeanReadLength, TotalThroughput=TotalThroughput)


# ----------------------------------------------------------------------------------------------
# 8a-8  result_plotting / ChannelPlotter_f data part (seq_routines.py:572-680, 378-394, 507-514)
# ----------------------------------------------------------------------------------------------


def nano_layout():
    """The fixed 32-row x 16-col MinION layout of seq_routines.py:589-671, rebuilt from its
    arithmetic structure: 8 bands of 4 rows; within a band row r, columns 0-7 ascend from
    ``base + 8*r + 1`` and columns 8-15 descend from ``base + 32 + 8*r + 8``."""
    bands = [0, 448, 384, 320, 256, 192, 128, 64]
    rows = []
    for base in bands:
        for r in range(4):
            left = [base + 8 * r + 1 + c for c in range(8)]
            right = [base + 32 + 8 * r + 8 - c for c in range(8)]
            rows.append([str(x) for x in left + right])
    return rows


def result_plotting_data(ch_par):
    """NanoPlate / NanoBasePlate (seq_routines.py:673-680): int(reads), int(bases), 0 if inactive."""
    NanoPlate, NanoBasePlate = [], []
    for row in nano_layout():
        line = [ch_par[c] for c in row]
        NanoPlate.append([int(x[0]) if len(x) > 0 else 0 for x in line])
        NanoBasePlate.append([int(x[1]) if len(x) > 0 else 0 for x in line])
    return NanoPlate, NanoBasePlate


def ChannelPlotter_data(c, c_plo, summary_flag=False, n_hours=48):
    """MuxList and FailReads% of one channel (seq_routines.py:378-394, 507-514, 519-529, 561-564)."""
    if not summary_flag:
        MuxList = [[0] * n_hours for _ in range(8)]
        try:
            Bait = c_plo[str(c)][4]
            for kk, vv in Bait.items():
                if vv != []:
                    l1 = [0] * n_hours
                    l2 = [0] * n_hours
                    for u, z in vv[0].items():
                        l1[int(u)] = int(z[0])
                        l2[int(u)] = int(z[1])
                    MuxList[int(kk) - 1] = l1
                    MuxList[int(kk) - 1 + 4] = l2
        except Exception:
            pass
    else:
        MuxList = [[0] * n_hours, [0] * n_hours]
        try:
            for kk, vv in c_plo[str(c)][4].items():
                nr, nb = vv
                MuxList[0][int(kk)] = int(nr)
                MuxList[1][int(kk)] = int(nb)
        except Exception:
            pass
    FailReads = 100
    if len(c_plo[str(c)]) > 0:
        FailReads = py2_int_div(int(c_plo[str(c)][2]) * 100, int(c_plo[str(c)][0]))      # :512
    return MuxList, FailReads, 100 - FailReads


# ----------------------------------------------------------------------------------------------
# table-shaped oracle: exactly what the C-ABI returns (include/pypore_qc.h)
# ----------------------------------------------------------------------------------------------

TOT_N_READS, TOT_N_BASES, TOT_N_FAIL, TOT_N_GC, TOT_N_N, TOT_QSUM, TOT_N_BINNED, TOT_BAD_CHANNEL, \
    TOT_TIME_OOB, TOT_ZERO_LEN, TOT_BAD_MUX, TOT_NO_TIME = range(12)
N_TOTALS = 16
N_LEN_LOG2 = 40
N_LEN_LIN = 100
N_QHIST = 8
N_GCHIST = 20
I64_MAX = (1 << 63) - 1
I64_MIN = -(1 << 63)


def default_config(**kw):
    cfg = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=0,
               tz_offset_s=0, ref_time_s=0, fail_q=7.0, compat_flags=0)
    cfg.update(kw)
    return cfg


def build_tables(stats, channel, mux, start_time_s, fail_flag=None, cfg=None):
    """Run-level tables from per-read records, with the reference's semantics:

      * fail-flagged reads get len=0, q=6.99, gc=0 (fast5_reader.py:37-40);
      * chan_reads / chan_bases count every read of the channel (:128-129);
      * chan_fail, cmh_reads, cmh_bases, hour_* only see reads whose mux is in 1..n_mux (:130-145);
      * time_mode 1 = hour-of-day + sequential wrap quirk per channel in start-time order
        (:117-126); time_mode 0 = floor((t - ref + tz) / bin_seconds);
      * reads whose hour bin falls outside [0, n_time_bins) are dropped from the hour / cmh tables
        only (the reference would KeyError at seq_routines.py:107; counted in TOT_TIME_OOB);
      * reads whose channel is outside 1..n_channels are counted in TOT_N_READS and
        TOT_BAD_CHANNEL and nowhere else (the reference would IndexError at :911).
    """
    cfg = default_config(**(cfg or {}))
    nch, nmux, nb = cfg["n_channels"], cfg["n_mux"], cfg["n_time_bins"]
    R = len(channel)
    length = stats["length"].astype(np.int64).copy()
    gc = stats["gc"].astype(np.int64).copy()
    nn = stats["n"].astype(np.int64).copy()
    qsum = stats["qsum"].astype(np.int64).copy()
    err = list(stats["err"])
    channel = np.asarray(channel, dtype=np.int64)
    mux = np.asarray(mux, dtype=np.int64)
    t = np.asarray(start_time_s, dtype=np.int64)
    flagged = np.zeros(R, dtype=bool) if fail_flag is None else np.asarray(fail_flag).astype(bool)
    length[flagged] = 0
    gc[flagged] = 0
    nn[flagged] = 0
    qsum[flagged] = 0

    thr = [q_threshold(cfg["fail_q"])] + [q_threshold(6.0 + j) for j in range(N_QHIST + 1)]
    fail = np.zeros(R, dtype=bool)
    qbin = np.full(R, -1, dtype=np.int64)
    is_record = np.asarray(stats["is_record"]).astype(bool) if "is_record" in stats else np.zeros(R, bool)
    for i in range(R):
        if flagged[i]:
            fail[i] = 6.99 < cfg["fail_q"]      # default QScore of a 'fail' path, fast5_reader.py:37
            qbin[i] = 0                          # floor(6.99 - 6)
            continue
        if is_record[i]:
            # rows of sequencing_summary.txt: only the float exists (seq_routines.py:852-855)
            q = float(stats["mean_q"][i])
            fail[i] = q < cfg["fail_q"]
            if 6.0 <= q < 14.0:
                qbin[i] = int(math.floor(q - 6.0))
            continue
        L, S = int(length[i]), int(err[i])
        if L == 0:
            fail[i] = 0.0 < cfg["fail_q"]   # mean_q defined as 0.0 for an empty read
            continue
        fail[i] = S > L * thr[0]
        # bin j  <=>  6+j <= mean_q < 7+j  <=>  L*e(7+j) < S <= L*e(6+j)
        for j in range(N_QHIST):
            if S <= L * thr[1 + j] and S > L * thr[2 + j]:
                qbin[i] = j
                break

    good_ch = (channel >= 1) & (channel <= nch)
    good_mux = (mux >= 1) & (mux <= nmux)
    no_time = t == I64_MIN                      # "none": a FASTQ header without start_time= (no hour bin, not ordered)
    sec = np.where(no_time, 0, t) - cfg["ref_time_s"] + cfg["tz_offset_s"]
    hour = np.floor_divide(sec, cfg["bin_seconds"])
    if cfg["time_mode"] == 1:
        hour = np.mod(np.floor_divide(sec, cfg["bin_seconds"]), 24)
        for c in np.unique(channel[good_ch]):
            idx = np.nonzero((channel == c) & ~no_time)[0]
            idx = idx[np.argsort(t[idx], kind="stable")]
            hour[idx] = _hour_wrap([int(h) for h in hour[idx]])
    hour[no_time] = -1

    totals = np.zeros(N_TOTALS, dtype=np.int64)
    len_log2 = np.zeros(N_LEN_LOG2, dtype=np.int64)
    len_lin = np.zeros(N_LEN_LIN, dtype=np.int64)
    q_hist = np.zeros(N_QHIST, dtype=np.int64)
    gc_hist = np.zeros(N_GCHIST, dtype=np.int64)
    hour_reads = np.zeros(nb, dtype=np.int64)
    hour_bases = np.zeros(nb, dtype=np.int64)
    chan_reads = np.zeros(nch, dtype=np.int64)
    chan_bases = np.zeros(nch, dtype=np.int64)
    chan_fail = np.zeros(nch, dtype=np.int64)
    cmh_reads = np.zeros((nch, nmux, nb), dtype=np.int64)
    cmh_bases = np.zeros((nch, nmux, nb), dtype=np.int64)
    chan_min = np.full(nch, I64_MAX, dtype=np.int64)
    chan_max = np.zeros(nch, dtype=np.int64)

    totals[TOT_N_READS] = R
    totals[TOT_BAD_CHANNEL] = int((~good_ch).sum())
    sel = good_ch
    max_len = int(length[sel].max()) if sel.any() else 0
    pos = length[sel & (length > 0)]
    min_pos_len = int(pos.min()) if len(pos) else I64_MAX
    lin_w = max_len // N_LEN_LIN + 1
    for i in np.nonzero(sel)[0]:
        L = int(length[i])
        c = int(channel[i]) - 1
        totals[TOT_N_BASES] += L
        totals[TOT_N_FAIL] += int(fail[i])
        totals[TOT_N_GC] += int(gc[i])
        totals[TOT_N_N] += int(nn[i])
        totals[TOT_QSUM] += int(qsum[i])
        totals[TOT_ZERO_LEN] += int(L == 0)
        totals[TOT_NO_TIME] += int(no_time[i])
        len_log2[0 if L == 0 else L.bit_length()] += 1
        len_lin[L // lin_w] += 1
        if qbin[i] >= 0:
            q_hist[qbin[i]] += 1
        if gc[i] > 0 and L != 0:
            gc_hist[min(N_GCHIST - 1, (20 * int(gc[i])) // L)] += 1
        chan_reads[c] += 1
        chan_bases[c] += L
        chan_min[c] = min(chan_min[c], L)
        chan_max[c] = max(chan_max[c], L)
        if not good_mux[i]:
            totals[TOT_BAD_MUX] += 1
            continue
        chan_fail[c] += int(fail[i])
        h = int(hour[i])
        if h < 0 or h >= nb:
            totals[TOT_TIME_OOB] += 1
            continue
        m = int(mux[i]) - 1
        totals[TOT_N_BINNED] += 1
        hour_reads[h] += 1
        hour_bases[h] += L
        cmh_reads[c, m, h] += 1
        cmh_bases[c, m, h] += L
    if cfg["compat_flags"] & 1:
        # channel_parser, multi_read_fast5_reader.py:64-67: `if len(l) > 1` filter by mux, `else: MucList = l` -- the
        # only read of a channel is taken under every mux '1'..'4' (see parsing_func(multi_read_single_quirk=True))
        for i in np.nonzero(sel)[0]:
            c = int(channel[i]) - 1
            if chan_reads[c] != 1:
                continue
            L, h = int(length[i]), int(hour[i])
            chan_fail[c] = nmux * int(fail[i])
            if h < 0 or h >= nb:
                continue
            normal = int(good_mux[i])
            cmh_reads[c, :, h] = 1
            cmh_bases[c, :, h] = L
            hour_reads[h] += nmux - normal
            hour_bases[h] += (nmux - normal) * L
    lens_for_n50 = np.where(sel, length, 0)
    return dict(totals=totals, len_hist_log2=len_log2, len_hist_lin=len_lin, q_hist=q_hist,
                gc_hist=gc_hist, hour_reads=hour_reads, hour_bases=hour_bases,
                chan_reads=chan_reads, chan_bases=chan_bases, chan_fail=chan_fail,
                cmh_reads=cmh_reads, cmh_bases=cmh_bases, chan_min_len=chan_min, chan_max_len=chan_max,
                max_len=max_len, min_pos_len=min_pos_len, lin_bin_width=lin_w,
                n50=n50(lens_for_n50), fail=fail, qbin=qbin, hour=hour)


def channel_dict_from_tables(tables, cfg=None):
    """Rebuild the reference's ChannelDict shape (minus the per-read ChrList) from the tables, the
    way the host drop-in does, so FastStats_data / result_plotting_data can be fed from either."""
    cfg = default_config(**(cfg or {}))
    cd = {}
    for c in range(cfg["n_channels"]):
        if tables["chan_reads"][c] == 0:
            cd[str(c + 1)] = ()
            continue
        mp = {}
        for m in range(cfg["n_mux"]):
            d = {}
            for h in range(cfg["n_time_bins"]):
                r = int(tables["cmh_reads"][c, m, h])
                if r:
                    d[str(h)] = (r, int(tables["cmh_bases"][c, m, h]))
            mp[str(m + 1)] = [d]
        cd[str(c + 1)] = (int(tables["chan_reads"][c]), int(tables["chan_bases"][c]),
                          int(tables["chan_fail"][c]), None, mp)
    return cd
