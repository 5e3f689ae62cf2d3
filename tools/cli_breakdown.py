"""Where the wall clock of `pypore seqstats -f fastq` goes (GPU box): interpreter + imports, context, files read into
pinned slots + submitted, tables, the two HTML reports.  usage: python tools/cli_breakdown.py DIR [loader threads]"""
import os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pypore_amd import seq_routines as sq, fastq_io, qc
t1 = time.perf_counter()
files = list(fastq_io.fastq_files_any(sys.argv[1]))
sq.work_dir, sq.out_dir, sq.prefix = os.getcwd(), "bd_out", "bd"
os.makedirs("bd_out", exist_ok=True)
with qc.QCContext(device=0, n_channels=512, n_mux=4, n_time_bins=48, time_mode=1) as ctx:
    t2 = time.perf_counter()
    sq._submit_fastq_files(ctx, files, None, threads=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t3 = time.perf_counter()
    ctx.finalize_extremes(); tmin, _ = ctx.time_range(); ctx.set_ref_time(tmin)
    ctx.finalize_tables(); ctx.finalize_n50(); tables = ctx.fetch_tables()
    t4 = time.perf_counter()
cd = sq.channel_dict_from_tables(tables)
t5 = time.perf_counter()
sq.result_plotting(cd)
t6 = time.perf_counter()
sq.FastStats(cd)
t7 = time.perf_counter()
nb = int(tables["totals"][1])
print("imports %.2f s | context %.2f | read + submit %.2f (%.2f GB/s of text) | tables %.3f | ChannelDict %.2f | pore map html %.2f | "
      "summary html %.2f | total %.2f s, %.2f Gbases/s" % (t1 - t0, t2 - t1, t3 - t2, sum(os.path.getsize(f) for f in files) / (t3 - t2) / 1e9,
                                                        t4 - t3, t5 - t4, t6 - t5, t7 - t6, t7 - t0, nb / (t7 - t0) / 1e9))
