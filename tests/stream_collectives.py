"""Test doubles for torch.distributed that keep the STREAM ORDER of a collective and nothing else, so that the merges of
pypore_amd/merge.py can be run on one GPU exactly as RCCL orders them at N > 1: the collective is enqueued on the
issuing stream, the host never waits for the device.

``SpyDist`` wraps a real process group (1-rank "nccl") and, in front of every collective, enqueues a device-side
snapshot of the tensor on the stream the collective is issued on.  ``drain=True`` synchronises the whole device first, so
that the snapshot is the producers' final word whatever the stream order is: a run with and a run without must leave the
same snapshots, or a producer is not ordered in front of the collective.

``LocalStreamWorld`` is a world of N ranks = N threads of this process that share the GPU: ``all_reduce`` /
``all_gather_into_tensor`` meet on the host (a barrier between the threads, no stream is waited for) and are carried
out on a communication stream that waits for every rank's issuing stream by event, and every issuing stream then waits
for it -- the order RCCL gives, with real data crossing between the ranks.
"""
import threading
from types import SimpleNamespace


class SpyDist:
    def __init__(self, dist, torch, drain=False):
        self._dist, self._torch, self.drain = dist, torch, drain
        self.ReduceOp = dist.ReduceOp
        self.snaps = []          # (issuing stream handle, clone of the tensor as the collective finds it)

    def _snap(self, t):
        if self.drain:
            self._torch.cuda.synchronize()
        self.snaps.append((self._torch.cuda.current_stream().cuda_stream, t.clone()))

    def get_backend(self, group=None):
        return self._dist.get_backend(group)

    def get_world_size(self, group=None):
        return self._dist.get_world_size(group)

    def all_reduce(self, t, op=None, group=None):
        self._snap(t)
        self._dist.all_reduce(t, op=op, group=group)

    def all_gather_into_tensor(self, out, t, group=None):
        self._snap(t)
        self._dist.all_gather_into_tensor(out, t, group=group)


class LocalStreamWorld:
    def __init__(self, world, torch):
        self.world, self.torch = world, torch
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.comm = torch.cuda.Stream()
        self.done = None
        self.calls = 0

    def rank(self, r):
        return _LocalRank(self, r)


class _LocalRank:
    ReduceOp = SimpleNamespace(SUM="sum", MAX="max", MIN="min")

    def __init__(self, w, r):
        self.w, self.r = w, r

    def get_backend(self, group=None):
        return "local-stream"

    def get_world_size(self, group=None):
        return self.w.world

    def _meet(self, payload, carry_out):
        w, torch = self.w, self.w.torch
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)                              # everything this rank has enqueued in front of the collective
        w.slots[self.r] = (payload, ev)
        if w.bar.wait() == 0:                       # one of the threads carries the collective out, on the comm stream
            with torch.cuda.stream(w.comm):
                for _, e in w.slots:
                    w.comm.wait_event(e)
                carry_out([p for p, _ in w.slots])
                d = torch.cuda.Event()
                d.record(w.comm)
                w.done = d
                w.calls += 1
        w.bar.wait()
        cur.wait_event(w.done)                      # the issuing stream goes on behind the collective; the host does not wait
        w.bar.wait()                                # slots and `done` are free for the next collective

    def all_reduce(self, t, op=None, group=None):
        torch = self.w.torch
        comm = self.w.comm

        def carry_out(ts):
            acc = ts[0].clone()
            for x in ts[1:]:
                acc = acc + x if op == "sum" else (torch.maximum(acc, x) if op == "max" else torch.minimum(acc, x))
            for x in ts:
                x.copy_(acc)
                x.record_stream(comm)
        self._meet(t, carry_out)

    def all_gather_into_tensor(self, out, t, group=None):
        comm = self.w.comm
        n = t.numel()

        def carry_out(ps):
            for o, _ in ps:
                for k, (_, x) in enumerate(ps):
                    o[k * n:(k + 1) * n].copy_(x)
                o.record_stream(comm)
        self._meet((out, t), carry_out)
