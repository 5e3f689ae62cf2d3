"""Host-side timeline of one bench step (config 2): where the wall time of reset / submit / finalize goes.
GPU box only.  usage: python tools/step_breakdown.py [reads] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import qc, synth

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed = 20240601
lens = synth.read_lengths(reads, 8.5172, 1.1774, seed)
off = synth.offsets_from_lengths(lens); nb = int(off[-1]); R = len(lens)
ch, mux, t = synth.metadata(R, 512, 48, seed)
ctx = qc.QCContext(device=0, ref_time_s=1_528_000_000, max_reads_hint=R)
db = torch.empty(nb + 64, dtype=torch.uint8, device="cuda"); dq = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
ctx.synth_fill_device(db.data_ptr(), dq.data_ptr(), 0, nb, seed)
d = [torch.from_numpy(x).cuda() for x in (off.view(np.int64), ch.view(np.int32), mux, t)]
ctx.sync()
rows = []
for s in range(steps + 2):
    t0 = time.perf_counter(); ctx.reset()
    t1 = time.perf_counter(); ctx.submit_device(R, nb, db.data_ptr(), dq.data_ptr(), *[x.data_ptr() for x in d])
    t2 = time.perf_counter(); ctx.finalize_extremes(); ctx.finalize_tables(); ctx.finalize_n50()
    t3 = time.perf_counter(); ctx.sync()
    t4 = time.perf_counter(); tab = ctx.fetch_tables(copy=False)
    t5 = time.perf_counter()
    tm = ctx.timings()
    if s >= 2:
        rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3,
                     tm["finalize_ms_last"], tm["sort_ms_last"]])
    ctx.reset_timings()
a = np.median(np.array(rows), axis=0)
print("median ms: reset %.3f submit(launch) %.3f finalize(launch) %.3f wait-for-gpu %.3f fetch(after sync) %.3f total %.3f | gpu finalize %.3f sort-chain %.3f"
      % tuple(a))
print("n50", tab["n50"], "reads", int(tab["totals"][0]))
