"""CPU-side checks of the drop-in boundary: the built C-ABI library loads, exports every symbol declared in
include/vorbis_synth_hip.h, mirrors the POD layouts, and FAILS LOUDLY (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry
from parseoggvorbis_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    entry.build_hip()
    return binding.load()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "vorbis_synth_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vsyn_[a-z_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    names = header_functions()
    assert len(names) >= 14
    for nm in names:
        assert hasattr(lib, nm), nm
    assert sorted(binding.declared_symbols()) == names
    assert lib.vsyn_abi_version() == 5
    assert b"gfx950" in lib.vsyn_version()


def test_pod_layouts_match_header():
    assert binding.PACKET_DTYPE.itemsize == 16 and binding.PACKET_DTYPE.fields["granule"][1] == 8
    assert binding.SEGMENT_DTYPE.itemsize == 24 and binding.SEGMENT_DTYPE.fields["residue_off"][1] == 16
    assert C.sizeof(binding.Status) == 8 and C.sizeof(binding.Taps) == 32 and binding.VQ_PACKET_DTYPE.itemsize == 16


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tests.workloads import fixture_like_spec
    with pytest.raises(binding.VsynError) as ei:
        binding.Synth(fixture_like_spec(2))
    assert ei.value.code == binding.VSYN_ERR_NO_DEVICE
    assert "no CPU path" in str(ei.value) or "device" in str(ei.value)


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "parseoggvorbis_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".inc", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_binding" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("oracle/gen_inverse_db.py", ""), os.path.join(dp, f)
