"""ppqc_sort_u32_device_static -- the entry point that allocates nothing (static device scratch inside the library) --
against ppqc_sort_u32_device, the entry point it is a variant of, through the C ABI.  Bit-exact (integer keys).  Two
calls in a row on the same GPU, the second with a smaller input, so that whatever the first call left in the shared
scratch must not leak into the second; then the size limit.  Collected last (file name) so that a failure here cannot
hide the other GPU tests behind `pytest -x`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _keys(seed, n):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    # the shapes the product sorts: read lengths (small values, many ties) mixed with full-range keys
    k[: n // 3] = np.minimum(rng.lognormal(8.5, 1.2, size=n // 3), 2e6).astype(np.uint32)
    return k


def test_static_sort_matches_allocating_sort_two_calls_in_a_row():
    import torch
    from pypore_amd import qc, _lib
    dev = torch.device("cuda", 0)
    with qc.QCContext(device=0) as ctx:
        for seed, n in ((1, 1_500_000), (2, 70_001), (3, 1), (4, 0), (5, 2 * 1024 * 1024)):
            k = _keys(seed, n)
            a = torch.from_numpy(k.view(np.int32)).to(dev)
            b = a.clone()
            torch.cuda.synchronize()
            ctx.sort_u32_device_static(a.data_ptr() if n else None, n)
            ctx.sort_u32_device(b.data_ptr() if n else None, n)
            ctx.sync()
            ga = a.cpu().numpy().view(np.uint32)
            gb = b.cpu().numpy().view(np.uint32)
            assert np.array_equal(ga, gb), "static sort differs from ppqc_sort_u32_device at n=%d" % n
            assert np.array_equal(ga, np.sort(k)), "n=%d" % n
        # beyond the stated maximum: a clear error, nothing launched
        n = 2 * 1024 * 1024 + 1
        a = torch.zeros(n, dtype=torch.int32, device=dev)
        with pytest.raises(_lib.PPQCError) as e:
            ctx.sort_u32_device_static(a.data_ptr(), n)
        assert e.value.code == -6 and "static scratch" in str(e.value)
