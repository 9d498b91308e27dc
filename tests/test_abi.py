"""The C-ABI library without a GPU: it loads, exports every symbol the header declares, and
refuses to create a handle when no device is present (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from gatk_amd import pairhmm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gatk_pairhmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gatk_pairhmm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = pairhmm.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gatk_pairhmm.h but not exported"
    assert sorted(pairhmm.EXPORTS) == names


def test_smith_waterman_header_symbols_are_exported():
    text = open(os.path.join(ROOT, "include", "gatk_smithwaterman.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(gatk_sw_[a-z0-9_]+)\s*\(", text)))
    assert names == ["gatk_sw_align", "gatk_sw_last_error"]
    lib = pairhmm.load_library()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gatk_smithwaterman.h but not exported"


def test_version_string():
    assert b"gfx950" in pairhmm.load_library().gatk_pairhmm_version()


def test_no_device_means_error_not_fallback():
    if pairhmm.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pairhmm.DeviceError):
        pairhmm.HipLoglessPairHMM(0)
    assert b"no CPU fallback" in pairhmm.load_library().gatk_pairhmm_last_error()


def test_null_handle_is_rejected():
    lib = pairhmm.load_library()
    assert lib.gatk_pairhmm_initialize(None, 10, 10) == pairhmm.ILLEGAL_ARGUMENT
    assert lib.gatk_pairhmm_finalize_region(None) == pairhmm.ILLEGAL_ARGUMENT
    assert lib.gatk_pairhmm_close(None) == pairhmm.OK
    assert lib.gatk_pairhmm_get_stats(None, None) == pairhmm.ILLEGAL_ARGUMENT


def test_product_does_not_reference_the_oracle():
    """The engine must never route through the CPU oracle."""
    for sub in ("gatk_amd", os.path.join("gatk_amd", "csrc"), "include"):
        d = os.path.join(ROOT, sub)
        for f in os.listdir(d):
            p = os.path.join(d, f)
            if os.path.isfile(p) and f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                assert "oracle" not in open(p, errors="replace").read().lower().replace("no cpu fallback", ""), p
