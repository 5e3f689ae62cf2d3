"""pypore_amd/staging.py without a GPU: the loader-thread / slot protocol of Stage and the CSR packing of pack_records,
with page-locked memory replaced by plain numpy arrays and the context by a recorder.  (That the CLI's batches really lie
in pinned memory is asserted on the GPU: tests/test_gpu_cli.py.)"""
import threading
import time

import numpy as np
import pytest

from pypore_amd import staging, qc


class FakePinned:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.array = np.zeros(self.nbytes, np.uint8)

    def close(self):
        self.array = None


class FakeCtx:
    def __init__(self):
        self.waits = []

    def slot_wait(self, d):
        self.waits.append(d)


@pytest.fixture(autouse=True)
def no_gpu(monkeypatch):
    monkeypatch.setattr(qc, "PinnedBuffer", FakePinned)


def _recs(rng, n):
    out = []
    for i in range(n):
        L = int(rng.integers(0, 200))
        out.append((int(rng.integers(1, 513)), int(rng.integers(0, 5)), 1_528_000_000 + i, i % 7 == 0, b"read%d" % i,
                    bytes(rng.integers(65, 85, L, dtype=np.uint8)), bytes(rng.integers(33, 90, L, dtype=np.uint8))))
    return out


def test_pack_records_into_a_slot_equals_fresh_arrays():
    rng = np.random.default_rng(1)
    slot = staging.PinnedSlot()
    for n in (0, 1, 57, 400, 3):                       # growing and shrinking batches reuse the slot's arenas
        recs = _recs(rng, n)
        a, b = staging.pack_records(recs, slot), staging.pack_records(recs)
        for k in ("bases", "quals", "offsets", "channel", "mux", "start_time", "fail_flag"):
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
            if a[k].nbytes:
                assert slot.owns(a[k].ctypes.data, a[k].nbytes), k
                assert not slot.owns(b[k].ctypes.data, b[k].nbytes)
        assert a["titles"] == b["titles"] == [r[4] for r in recs] and a["n_reads"] == n
        assert int(a["offsets"][-1]) == sum(len(r[5]) for r in recs) == len(a["bases"])
        want = b"".join(r[5] for r in recs)
        assert a["bases"].tobytes() == want
    assert staging.in_pinned_slot(a["offsets"].ctypes.data, 8)
    slot.close()


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_stage_hands_every_job_over_in_order_and_recycles_its_slots(threads):
    ctx = FakeCtx()
    rng = np.random.default_rng(threads)
    delays = rng.random(40) * 0.004                     # loaders finish out of order
    seen, slots_used, device_slots = [], set(), []

    def fill(job, slot):
        if job % 5 == 4:
            return None                                # a job that yields nothing is skipped
        time.sleep(delays[job])
        v = slot.text(64)
        v[:] = job
        return job, v

    def submit(item, d):
        job, v = item
        assert (v == job).all()                        # nobody has overwritten the slot in the meantime
        seen.append(job)
        device_slots.append(d)
        slots_used.add(v.ctypes.data)
        time.sleep(0.001)
        return True

    with staging.Stage(ctx, threads=threads) as st:
        st.run(range(40), fill, submit)
        assert len(st.slots) == threads + 2
    assert seen == [j for j in range(40) if j % 5 != 4]              # in the order of the jobs, whoever loaded them
    assert device_slots == [k % 2 for k in range(len(seen))]          # the two device staging slots alternate
    assert len(slots_used) <= threads + 2
    assert len(ctx.waits) == len(seen)                                  # every batch waited for exactly once
    assert ctx.waits[:4] == [0, 1, 0, 1]


def test_stage_passes_a_loader_error_on_and_stops_the_thread():
    ctx = FakeCtx()

    def fill(job, slot):
        if job == 3:
            raise ValueError("bad file %d" % job)
        return job

    got = []
    before = threading.active_count()
    with pytest.raises(ValueError, match="bad file 3"):
        with staging.Stage(ctx, threads=2) as st:
            st.run(range(10), fill, lambda item, d: got.append(item) or True)
    assert got == [0, 1, 2][:len(got)]
    assert threading.active_count() == before

    def submit(item, d):
        if item == 2:
            raise RuntimeError("submit failed")
        return True
    with pytest.raises(RuntimeError, match="submit failed"):
        with staging.Stage(ctx, threads=4) as st:
            st.run(range(1000), lambda job, slot: job, submit)
    assert threading.active_count() == before


def test_submit_that_uses_no_device_slot_frees_the_host_slot_at_once():
    ctx = FakeCtx()
    ds = []
    with staging.Stage(ctx) as st:
        st.run(range(6), lambda job, slot: job, lambda item, d: (ds.append(d), item % 2 == 0)[1])
    # odd jobs did not use the GPU slot they were offered: the next job is offered the same one
    assert ds == [0, 1, 1, 0, 0, 1] and len(ctx.waits) == 3
