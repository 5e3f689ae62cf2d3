"""The C-ABI boundary without a GPU: the library loads, exports every symbol the header declares, the ctypes
structures match the C layouts, and -- with no device -- the compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pypore_qc.h")


@pytest.fixture(scope="module")
def lib():
    from pypore_amd import build, _lib
    build.build()
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppqc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    from pypore_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libpypore_qc.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "pypore_amd/_lib.py SYMBOLS out of sync with include/pypore_qc.h"
    assert lib.ppqc_abi_version() == 7


def test_ctypes_structs_match_header_layout(lib, tmp_path):
    from pypore_amd import _lib
    structs = ["ppqc_config", "ppqc_batch", "ppqc_per_read", "ppqc_tables", "ppqc_device_tables", "ppqc_timing"]
    prog = '#include <stdio.h>\n#include "pypore_qc.h"\nint main(void){' + "".join(
        'printf("%%zu\\n", sizeof(%s));' % s for s in structs) + "return 0;}"
    c = tmp_path / "sz.c"
    c.write_text(prog)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mine = [C.sizeof(x) for x in (_lib.Config, _lib.Batch, _lib.PerRead, _lib.Tables, _lib.DeviceTables, _lib.Timing)]
    assert sizes == mine


def test_config_default_and_validation(lib):
    from pypore_amd import _lib
    cfg = _lib.Config()
    assert lib.ppqc_config_default(C.byref(cfg)) == 0
    # the reference's hard-coded geometry: 512 channels, 4 muxes, 48 hourly bins, fail below 7.0
    assert (cfg.n_channels, cfg.n_mux, cfg.n_time_bins, cfg.bin_seconds, cfg.fail_q) == (512, 4, 48, 3600, 7.0)
    assert cfg.struct_size == C.sizeof(_lib.Config)
    ctx = C.c_void_p()
    bad = _lib.Config.from_buffer_copy(cfg)
    bad.n_channels = 0
    assert lib.ppqc_create(C.byref(ctx), 0, C.byref(bad)) == -1 and b"config" in lib.ppqc_last_error()
    bad = _lib.Config.from_buffer_copy(cfg)
    bad.time_mode = 2
    assert lib.ppqc_create(C.byref(ctx), 0, C.byref(bad)) == -1
    bad = _lib.Config.from_buffer_copy(cfg)
    bad.struct_size = 8
    assert lib.ppqc_create(C.byref(ctx), 0, C.byref(bad)) == -1
    assert lib.ppqc_create(None, 0, C.byref(cfg)) == -1
    assert lib.ppqc_submit(None, None, 0) == -1 and lib.ppqc_sync(None) == -1


def test_no_device_fails_loudly(lib):
    """On a machine without a GPU the product path must raise, never fall back to a CPU implementation."""
    from pypore_amd import qc, _lib
    n = C.c_int(-1)
    rc = lib.ppqc_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.PPQCError) as e:
        qc.QCContext(device=0)
    assert e.value.code == -2 and "no HIP device" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pypore_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "ppqc_oracle" not in txt and "libppqc_oracle" not in txt, f
