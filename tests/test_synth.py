import numpy as np

from pypore_amd import synth


def test_stream_bytes_is_a_pure_function_of_seed_and_index():
    b, q = synth.stream_bytes(0, 10_000, 42)
    for first, n in ((0, 1), (3, 5), (4, 4), (1001, 7777), (9999, 1)):
        sb, sq = synth.stream_bytes(first, n, 42)
        assert np.array_equal(sb, b[first:first + n]) and np.array_equal(sq, q[first:first + n])
    b2, _ = synth.stream_bytes(0, 10_000, 43)
    assert not np.array_equal(b, b2)
    assert len(synth.stream_bytes(5, 0, 1)[0]) == 0


def test_stream_statistics_match_baseline_md():
    b, q = synth.stream_bytes(1 << 40, 2_000_000, 20240601)
    gc = np.isin(b, np.frombuffer(b"GC", np.uint8)).mean()
    assert abs(gc - 0.41) < 0.003
    assert 1e-5 < (b == ord("N")).mean() < 3e-4
    assert set(np.unique(b).tolist()) <= set(b"ACGTN")
    ph = q.astype(int) - 33
    assert ph.min() >= 1 and ph.max() <= 60 and abs(ph.mean() - 12) < 0.3 and abs(ph.std() - 4) < 0.4


def test_lengths_and_metadata():
    L = synth.read_lengths(200_000, 8.5172, 1.1774, 20240601)
    assert np.array_equal(L, synth.read_lengths(200_000, 8.5172, 1.1774, 20240601))
    assert L.min() >= 1 and L.max() <= 2_000_000 and 9_000 < L.mean() < 11_000
    off = synth.offsets_from_lengths(L)
    assert off[0] == 0 and off[-1] == L.sum(dtype=np.uint64) and off.dtype == np.uint64
    ch, mux, t = synth.metadata(100_000, 512, 48, 1)
    assert ch.min() == 1 and ch.max() == 512 and set(np.unique(mux).tolist()) == {0, 1, 2, 3, 4}
    assert 0.005 < (mux == 0).mean() < 0.02 and (t.max() - t.min()) < 48 * 3600
    adv = synth.adversarial_lengths(200_000, 3)
    assert (adv == 2_000_000).sum() == 100 and (adv == 200).sum() == 199_900


def test_c_generator_of_the_cpu_leg_equals_stream_bytes(c_oracle):
    for first, n, seed, th in ((0, 10_001, 42, 1), (3, 5, 7, 4), (1001, 77_777, 20240601, 3), (1 << 33, 4096, 9, 16),
                               (5, 0, 1, 2)):
        b, q = synth.stream_bytes(first, n, seed)
        cb, cq = c_oracle.synth_fill(first, n, seed, th)
        assert np.array_equal(b, cb) and np.array_equal(q, cq), (first, n, seed, th)
