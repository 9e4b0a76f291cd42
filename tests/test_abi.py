"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/ivfhnsw_hip.h
declares, and fails loudly -- never silently falls back -- when no gfx950 device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_is_built_and_loads(pkg):
    assert os.path.exists(pkg.LIB_PATH), "run __graft_entry__.build()"
    lib = pkg.lib()
    assert lib.ivfhnsw_gpu_abi_version() == 9


def test_every_declared_symbol_is_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "ivfhnsw_hip.h")).read()
    declared = set(re.findall(r"\b(ivfhnsw_gpu_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(pkg.ABI_SYMBOLS), declared ^ set(pkg.ABI_SYMBOLS)
    raw = ctypes.CDLL(pkg.LIB_PATH)
    for sym in declared:
        assert hasattr(raw, sym), sym


def test_header_cites_the_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "ivfhnsw_hip.h")).read()
    for cite in ("IndexIVF_HNSW.cpp:234-296", "IndexIVF_HNSW_Grouping.cpp:188-363", "IndexIVF_HNSW.cpp:453-492",
                 "hnswalg.cpp:227-234", "IndexIVF_HNSW.h:50-66"):
        assert cite in hdr, cite


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C with no C++ or torch types."""
    src = tmp_path / "t.c"
    src.write_text('#include "ivfhnsw_hip.h"\nint main(void){ivfhnsw_search_params p; (void)p; return 0;}\n')
    import subprocess
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                    str(src)], check=True)


def test_no_device_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device error path is covered on the CPU box")
    with pytest.raises(pkg.IvfHnswError) as e:
        pkg.GpuIndex(0)
    assert e.value.code == pkg.ERR_HIP
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """oracle/ is test infrastructure: nothing under ivf-hnsw_amd/ or include/ may import, link or call it."""
    bad = []
    for base in ("ivf-hnsw_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"liborc|orc_search|from oracle|import oracle|oracle/", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
