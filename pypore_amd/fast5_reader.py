"""Single-read Fast5 loader: h5py on the CPU, reads packed as CSR batches for the GPU (lib/fast5_reader.py:13-71,
lib/seq_routines.py:819-828, 872-894, 908-911).

Only the I/O of the reference's ``get_content`` stays here.  What it computed per read -- ``GC()`` of FASTQ line 2
(:57-58) -- and what it read from albacore's summary attributes -- ``sequence_length``, ``mean_qscore`` (:55-56) --
is recomputed by the kernels from the FASTQ record itself (bases = line 2, quals = line 4).  h5py is imported lazily;
``h5_open`` lets tests (and other HDF5 bindings) supply the file opener."""
from __future__ import annotations

import os
import re
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .logging_module import log

FASTQ_PATH = 'Analyses/Basecall_1D_000/BaseCalled_template/Fastq'


def default_h5_open(path, mode='r'):
    try:
        import h5py
    except ImportError as e:   # loud: there is no fallback reader
        raise RuntimeError("reading Fast5 files needs h5py, which is not installed (%s); "
                           "use -a sequencing_summary.txt or -f fastq input instead" % e)
    return h5py.File(path, mode)


def _text(v):
    return v.decode('UTF-8') if isinstance(v, (bytes, bytearray)) else str(v)


def mktime_local(stamp):
    """``time.mktime(dateutil.parse(stamp).timetuple())`` (:66-67): the wall-clock fields read as local time, whole
    seconds."""
    from dateutil import parser as par
    return time.mktime(par.parse(stamp).timetuple())


def min_time_catcher(f, h5_open=None):
    """lib/seq_routines.py:819-828: exp_start_time of one file."""
    with (h5_open or default_h5_open)(f, 'r') as f5:
        start = _text(f5['UniqueGlobalKey/tracking_id'].attrs['exp_start_time'])
    return mktime_local(start)


def dir_surfer(file_folder, h5_open=None, want_ref=True):
    """lib/seq_routines.py:872-894: walk the folder, files of a directory in modification-time order, dot files
    skipped.  RefTime = min exp_start_time over the first file of every leaf directory (and of a directory named '0'
    that has sub-directories).  Returns (files, RefTime)."""
    files, ref = [], None
    for subdir, dirs, names in os.walk(file_folder):
        paths = sorted((os.path.join(subdir, y) for y in names), key=lambda x: os.path.getmtime(x))
        for en, f in enumerate(paths):
            if os.path.basename(f).startswith('.'):
                continue
            if want_ref and en == 0 and (dirs == [] or os.path.basename(subdir) == '0'):
                t = min_time_catcher(f, h5_open)
                if ref is None or float(t) < float(ref):
                    ref = t
            files.append(f)
    return files, ref


def channel_of(path):
    """lib/seq_routines.py:909-910: the channel is the second-to-last '_' field of the file name."""
    return int(path.split('_')[-2])


def Pathcheck(f5, fail):
    """lib/fast5_reader.py:13-33: locate the groups by visiting the file; the first becomes the visited name."""
    visits = []
    f5.visit(visits.append)
    paths = ['Raw/Reads', 'UniqueGlobalKey/channel_id', 'UniqueGlobalKey/tracking_id']
    if not fail:
        paths += ['Analyses/Basecall_1D_000/Summary/basecall_1d_template', FASTQ_PATH]
    return [p for p in paths if p in visits]


SUMMARY_PATH = 'Analyses/Basecall_1D_000/Summary/basecall_1d_template'


def albacore_summary(attrs):
    """(sequence_length, mean_qscore) as albacore wrote them beside the FASTQ record (lib/fast5_reader.py:55-56) -- the
    values the reference takes on trust.  The GPU path recomputes both from the record itself; the loaders carry these
    along so that the run can say where the two disagree (seq_routines.albacore_cross_check).  None where absent."""
    try:
        ln, mq = attrs.get('sequence_length'), attrs.get('mean_qscore')
        return (None if ln is None else int(ln), None if mq is None else float(mq))
    except (TypeError, ValueError, AttributeError):
        return (None, None)


def get_content(f, h5_open=None, p_check=None):
    """One read (lib/fast5_reader.py:36-71): (mux, start_time, fail, title, bases, quals, albacore summary).  A path containing 'fail'
    yields no sequence (the reference keeps ReadLength 0, QScore 6.99, GC 0 for it, :37-40, 52).  ``p_check``: the group
    names found by ONE visit of a file of the channel, as the reference passes them in (:96-105); a fail path uses its
    first three entries only."""
    fail = 'fail' in f
    read_number = int(re.sub(r'.*read', 'read', f).split('_')[1])
    with (h5_open or default_h5_open)(f, 'r') as f5:
        dp = p_check if p_check is not None else Pathcheck(f5, fail)
        mux = int(f5['%s/Read_%d' % (dp[0], read_number)].attrs['start_mux'])
        _channel = _text(f5[dp[1]].attrs['channel_number'])      # read like the reference does; unused there too (:106)
        stamp = _text(f5['Analyses/Segmentation_000'].attrs['time_stamp'])
        title = bases = quals = b""
        alb = (None, None)
        if not fail:
            lines = _text(f5[dp[4]][()]).split('\n')
            title, bases, quals = lines[0][1:].encode(), lines[1].encode(), lines[3].encode()
            if len(bases) != len(quals):
                raise ValueError("sequence/quality length mismatch in %s" % f)
            alb = albacore_summary(f5[dp[3]].attrs)
    return mux, int(mktime_local(stamp)), fail, title, bases, quals, alb


def channel_path_checks(files, h5_open=None):
    """The reference visits ONE file per channel -- the first whose path contains 'pass', else the first with 'fail',
    else the channel's first file -- and hands the group names found there to every get_content of the channel
    (lib/fast5_reader.py:95-105).  Returns {channel: names}; a channel whose probe file cannot be read gets None (its
    reads then fail one by one, as they would in the reference's ``except: del ds``)."""
    first = {}
    for f in files:
        ch = channel_of(f)
        cur = first.get(ch)
        rank = 0 if 'pass' in f else (1 if 'fail' in f else 2)
        if cur is None or rank < cur[0]:
            first[ch] = (rank, f)
    out = {}
    for ch, (_r, f) in first.items():
        try:
            with (h5_open or default_h5_open)(f, 'r') as f5:
                out[ch] = Pathcheck(f5, 'fail' in f)
        except (KeyError, IndexError, ValueError, OSError) as e:
            log.debug("path check of %s failed: %s" % (f, e))
            out[ch] = None
    return out


def iter_record_batches(files, n_channels=512, threads=1, h5_open=None, batch_reads=4000, job_files=None):
    """Lists of loader records ``(channel, mux, start_time, fail, title, bases, quals, albacore)``, ``batch_reads`` files at a
    time.  Unreadable reads are skipped as in the reference's ``except: del ds`` (lib/fast5_reader.py:104-116); a file
    name without a channel field raises like :910.  ``job_files``: all files of the job when ``files`` is one rank's
    share of them -- the probe file of a channel is picked among the job's files, so that every rank (and a run on one
    GPU, and the reference) hands the same group names to the channel's reads."""
    mine = {channel_of(f) for f in files}
    checks = channel_path_checks([f for f in (job_files if job_files is not None else files) if channel_of(f) in mine],
                                 h5_open)

    def one(f):
        ch = channel_of(f)
        if ch < 1 or ch > n_channels:
            raise IndexError("channel %d of %s outside 1..%d" % (ch, f, n_channels))
        try:
            if checks.get(ch) is None:
                raise KeyError("no usable path check for channel %d" % ch)
            mux, t, fail, title, b, q, alb = get_content(f, h5_open, checks[ch])
        except (KeyError, IndexError, ValueError, OSError) as e:
            log.debug("skipping %s: %s" % (f, e))
            return None
        return (ch, mux if 0 <= mux <= 255 else 0, t, 1 if fail else 0, title, b, q, alb)

    with ThreadPoolExecutor(max_workers=max(1, int(threads))) as pool:
        for a in range(0, len(files), batch_reads):
            yield [r for r in pool.map(one, files[a:a + batch_reads]) if r is not None]


def load_batches(files, n_channels=512, threads=1, h5_open=None, batch_reads=4000):
    """The same as CSR batches in fresh (pageable) arrays; the CLI packs into pinned slots instead (staging.py)."""
    from .staging import pack_records
    for recs in iter_record_batches(files, n_channels, threads, h5_open, batch_reads):
        yield pack_records(recs)
