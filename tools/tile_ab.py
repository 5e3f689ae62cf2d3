"""A/B of tile-kernel variants in ONE process, interleaved rounds (guide rule 24). GPU box only.
usage: python tools/tile_ab.py [reads] [rounds]"""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import qc, synth

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
variants = os.environ.get("AB_VARIANTS", "0,0s,0d,0a,0p").split(",")
bpcs = os.environ.get("AB_BPC", "0").split(",")
shifts = os.environ.get("AB_SHIFT", "16").split(",")
workload = os.environ.get("AB_WORKLOAD", "config2")
sleep_ms = float(os.environ.get("AB_SLEEP_MS", "0"))
tail = int(os.environ.get("AB_TAIL", "0"))
seed = 20240601
if workload == "adv":
    lens = synth.adversarial_lengths(reads, seed)
elif workload == "config4":
    lens = synth.read_lengths(reads, 10.1266, 1.1774, seed)
elif workload.startswith("mean"):   # e.g. mean300: lognormal with that mean length (sigma 1.1774: mean = 2 e^mu)
    import math
    lens = synth.read_lengths(reads, math.log(float(workload[4:]) / 2.0), 1.1774, seed)
elif workload == "short":
    lens = synth.read_lengths(reads, 6.0, 1.0, seed)
else:
    lens = synth.read_lengths(reads, 8.5172, 1.1774, seed)
off = synth.offsets_from_lengths(lens); nb = int(off[-1]); R = len(lens)
ch, mux, t = synth.metadata(R, 512, 48, seed)
ctx = qc.QCContext(device=0, ref_time_s=1_528_000_000, max_reads_hint=R)
db = torch.empty(nb + 64, dtype=torch.uint8, device="cuda"); dq = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
ctx.synth_fill_device(db.data_ptr(), dq.data_ptr(), 0, nb, seed)
d = [torch.from_numpy(x).cuda() for x in (off.view(np.int64), ch.view(np.int32), mux, t)]
ctx.sync()
res = {}
ref = None
for rnd in range(rounds + 1):
    for v in variants:
        for bpc, sh in [(x, y) for x in bpcs for y in shifts]:
            # "0" = what ships (per-tile probe, or dense_kernel alone by the mean); "0s" / "0d" force every tile sparse /
            # dense; "0a" / "0p": dense_kernel alone / always the pair; "0:LEN" sets the density threshold
            tune = dict(tile_dense=-1, dense_len=-1, tile_all=-1)
            flags = v.split(":")[0][1:]
            if "d" in flags: tune["tile_dense"] = 1     # every tile onto the dense list
            if "s" in flags: tune["tile_dense"] = 0     # every tile sparse (tile_kernel alone)
            if "a" in flags: tune["tile_all"] = 1       # the dense kernel alone over every tile
            if "p" in flags: tune["tile_all"] = 0       # always the pair, whatever the mean
            if ":" in v: tune["dense_len"] = int(v.split(":")[1])
            tune["tile_shift"] = int(sh)
            for k, x in tune.items():
                ctx.set_tuning(k, x)
            ctx.reset(); ctx.reset_timings()
            if sleep_ms: time.sleep(sleep_ms / 1e3)      # idle gap before the launch (clock / DPM effects)
            ctx.submit_device(R, nb, db.data_ptr(), dq.data_ptr(), *[x.data_ptr() for x in d])
            tm = ctx.timings()
            if tail: ctx.finalize()                       # the step tail of bench.py in front of the next tile kernel
            if rnd == 0:
                s = ctx.read_stats(0, min(R, 20000))
                sig = (int(s["gc"].sum()), int(s["qsum"].sum()), int(s["err_lo"].sum() & 0xFFFFFFFFFFFF), int(s["n"].sum()))
                ref = ref or sig
                assert sig == ref, (v, bpc, sh, sig, ref)
                continue
            res.setdefault((v, bpc, sh), []).append(tm["tile_kernel_ms_total"])
bytes_alg = 2 * nb + 50 * R
print("workload %s reads %d bases %.2f G" % (workload, R, nb / 1e9))
for k, ms in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    med, mn = float(np.median(ms)), float(min(ms))
    print("variant %s bpc %s shift %s: median %.3f ms (%.0f GB/s, %.3f of 8 TB/s)  min %.3f ms" % (k[0], k[1], k[2], med, bytes_alg / med / 1e6, bytes_alg / med / 1e6 / 8000, mn))
