"""Multi-read Fast5 loader (lib/multi_read_fast5_reader.py:81-147, lib/seq_routines.py:952-994): one CSR batch per
file, h5py on the CPU.  Fail status comes from the FILE path (lib/seq_routines.py:975); RefStart is
exp_start_time of the first read of a file (:137).  See fast5_reader.py for what moved to the GPU."""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

from .fast5_reader import default_h5_open, mktime_local, _text, FASTQ_PATH, SUMMARY_PATH, albacore_summary
from .logging_module import log


def dir_surfer(file_folder):
    """lib/seq_routines.py:872-894 with multiread_flag set: no RefTime probing, mtime order inside a directory."""
    files = []
    for subdir, _dirs, names in os.walk(file_folder):
        paths = sorted((os.path.join(subdir, y) for y in names), key=lambda x: os.path.getmtime(x))
        files += [f for f in paths if not os.path.basename(f).startswith('.')]
    return files


def read_file(path, h5_open=None):
    """mf5_reader.__call__ (:121-147): every read of one file -> (records, RefStart)."""
    fail = 'fail' in path
    recs = []
    with (h5_open or default_h5_open)(path, 'r') as f5:
        reads = list(f5.keys())
        ref_start = None
        if reads:
            ref_start = mktime_local(_text(f5[str(reads[0]) + '/tracking_id'].attrs['exp_start_time']))
        for r in reads:
            r = str(r)
            try:
                mux = int(f5[r + '/Raw'].attrs['start_mux'])
                channel = int(_text(f5[r + '/channel_id'].attrs['channel_number']))
                stamp = _text(f5[r + '/Analyses/Segmentation_000'].attrs['time_stamp'])
                title = bases = quals = b""
                alb = (None, None)
                if not fail:
                    lines = _text(f5[r + '/' + FASTQ_PATH][()]).split('\n')
                    title, bases, quals = lines[0][1:].encode(), lines[1].encode(), lines[3].encode()
                    if len(bases) != len(quals):
                        raise ValueError("sequence/quality length mismatch")
                    try:    # albacore's own length and mean quality (:104-105), for the cross-check only
                        alb = albacore_summary(f5[r + '/' + SUMMARY_PATH].attrs)
                    except KeyError:
                        pass
            except (KeyError, IndexError, ValueError) as e:
                # the reference has no handler here and the worker dies; skipping the read keeps the run alive
                log.warning("skipping %s in %s: %s" % (r, path, e))
                continue
            recs.append((channel, mux if 0 <= mux <= 255 else 0, int(mktime_local(stamp)), 1 if fail else 0, title,
                         bases, quals, alb))
    return recs, ref_start


def ref_start_of(files, h5_open=None):
    """RefStart of a run: exp_start_time of the first read of the first file that has one.  The reference ends up with
    the value of whichever file's result is scanned last (lib/seq_routines.py:989-994: every float in support_list
    overwrites it); all files of a run share one exp_start_time."""
    for f in files:
        try:
            with (h5_open or default_h5_open)(f, 'r') as f5:
                reads = list(f5.keys())
                if reads:
                    return mktime_local(_text(f5[str(reads[0]) + '/tracking_id'].attrs['exp_start_time']))
        except (KeyError, IndexError, ValueError, OSError) as e:
            log.debug("no exp_start_time in %s: %s" % (f, e))
    return None


def iter_record_batches(files, threads=1, h5_open=None):
    """One list of loader records per file, files read ``threads`` at a time, in file order."""
    with ThreadPoolExecutor(max_workers=max(1, int(threads))) as pool:
        for recs, _ref in pool.map(lambda f: read_file(f, h5_open), files):
            yield recs


def load_batches(files, threads=1, h5_open=None):
    """(list of CSR batches in fresh arrays, RefStart); the CLI packs into pinned slots instead (staging.py)."""
    from .staging import pack_records
    with ThreadPoolExecutor(max_workers=max(1, int(threads))) as pool:
        out = list(pool.map(lambda f: read_file(f, h5_open), files))
    ref = next((r for _recs, r in out if r is not None), 0)
    return [pack_records(recs) for recs, _r in out], ref
