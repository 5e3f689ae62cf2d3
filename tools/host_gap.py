"""usage: python tools/host_gap.py [reads] [mu] [plain|early|behind]
Where the host's time in a step goes (config-2-shaped batch resident in HBM): reset, submit_device and finalize_view
timed apart on the host clock, against the per-read kernel's own time from the library's events.  early / behind: the
step of the multi-GPU path instead (finalize_distributed over a one-rank RCCL group, its merges beside or behind the
per-read kernel), with the time at which the host has enqueued everything and enters the fetch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pypore_amd import qc, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 8.5172
lens = synth.read_lengths(R, mu, 1.1774, 1234)
off = synth.offsets_from_lengths(lens)
n_bytes = int(off[-1])
ch, mux, t = synth.metadata(R, 512, 48, 1234)
ctx = qc.QCContext(device=0, n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=0, tz_offset_s=0,
                   ref_time_s=1_528_000_000, fail_q=7.0, max_reads_hint=R)
dev = torch.device("cuda:0")
d_b = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
d_q = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
ctx.synth_fill_device(d_b.data_ptr(), d_q.data_ptr(), 0, n_bytes, 1234)
d_off = torch.from_numpy(off.view(np.int64)).to(dev)
d_ch = torch.from_numpy(ch.view(np.int32)).to(dev)
d_mux = torch.from_numpy(mux).to(dev)
d_t = torch.from_numpy(t).to(dev)
ctx.sync(); torch.cuda.synchronize()
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
t_fetch = [0.0]
if mode != "plain":
    import torch.distributed as dist
    from pypore_amd.merge import finalize_distributed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    inner = ctx.fetch_tables
    def fetch(copy=True):
        t_fetch[0] = time.perf_counter()
        return inner(copy=copy)
    ctx.fetch_tables = fetch
acc = np.zeros(4)
N = 60
for i in range(N + 10):
    if i == 10:
        acc[:] = 0
        ctx.reset_timings()
        t_all = time.perf_counter()
    t0 = time.perf_counter()
    ctx.reset()
    t1 = time.perf_counter()
    ctx.submit_device(R, n_bytes, d_b.data_ptr(), d_q.data_ptr(), d_off.data_ptr(), d_ch.data_ptr(), d_mux.data_ptr(), d_t.data_ptr())
    t2 = time.perf_counter()
    if mode == "plain":
        ctx.finalize(copy=False)
    else:
        finalize_distributed(ctx, dist, torch, copy=False, early=None if mode == "early" else False)
    t3 = time.perf_counter()
    acc += (t1 - t0, t2 - t1, t3 - t2, (t_fetch[0] - t2) if mode != "plain" else 0.0)
step = (time.perf_counter() - t_all) / N * 1e3
tm = ctx.timings()
k = tm["tile_kernel_ms_total"] / max(1, tm["tile_kernel_launches"])
print("%s: step %.3f ms: reset %.1f us, submit %.1f us, finalize (with the wait) %.1f us (everything enqueued after %.1f us); "
      "per-read kernel %.3f ms, finalize_ms %.3f"
      % (mode, step, acc[0] / N * 1e6, acc[1] / N * 1e6, acc[2] / N * 1e6, acc[3] / N * 1e6, k, tm["finalize_ms_last"]))
if mode != "plain":
    dist.destroy_process_group()
