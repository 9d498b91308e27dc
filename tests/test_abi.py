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


def _kernel_metadata(path):
    """(name, vgpr_spill_count, private_segment_fixed_size) of every gfx950 kernel in a built library, read from the
    code object's notes; None when the binutils / ROCm LLVM tools are not on this machine."""
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    bundler, readelf = os.path.join(llvm, "clang-offload-bundler"), os.path.join(llvm, "llvm-readelf")
    if not (shutil.which("objcopy") and os.path.exists(bundler) and os.path.exists(readelf)):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "code.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        subprocess.check_call([bundler, "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                              stderr=subprocess.DEVNULL)
        notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1))
        out.append((name, spill, scratch))
    return out


def test_register_spills_of_the_shipped_kernels_go_to_scratch_memory():
    """The fp64 passes from 10 rows per lane on (and the fp64 pair pass from 12) spill registers.  Built with spills into
    accumulation registers (the compiler's default) they return wrong likelihoods on MI355X for reads of some 600 bases
    and more (tests/test_gpu_parity.py::test_long_reads_at_many_rows_per_lane fails at 760 and 830 bases); the Makefile
    therefore passes -amdgpu-spill-vgpr-to-agpr=0.  A build that lost the flag shows up here without a GPU: its
    spilling kernels have less scratch memory than spills (none at all for the pair pass)."""
    kernels = _kernel_metadata(pairhmm.LIB_PATH)
    if kernels is None:
        pytest.skip("objcopy / clang-offload-bundler / llvm-readelf not available")
    assert len(kernels) > 100
    spilling = [k for k in kernels if k[1] > 0]
    assert spilling, "the fp64 instantiations of 12 and more rows per lane are expected to spill"
    for name, spill, scratch in spilling:
        assert scratch >= spill, f"{name}: {spill} spills but {scratch} bytes of scratch: spills went to accumulation registers"
        # only the instantiations known to need more than the register file: fp64, 9 rows per lane and more
        m = re.search(r"fwd_(?:direct|list)IdLi(\d+)|fwd_pairsILi(\d+)", name)
        assert m and int(m.group(1) or m.group(2)) >= 9, f"unexpected register spills in {name}"
