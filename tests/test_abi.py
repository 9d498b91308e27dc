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


def _part_cuts(weights, shared=False):
    import ctypes as C
    lib = pairhmm.load_library()
    lib.gatk_pairhmm_debug_part_cuts.restype = C.c_int
    w = (C.c_int64 * len(weights))(*weights)
    out = (C.c_int32 * (len(weights) + 2))()
    n = lib.gatk_pairhmm_debug_part_cuts(w, len(weights), 1 if shared else 0, out, len(out))
    assert n >= 2
    cuts = list(out[:n])
    assert cuts[0] == 0 and cuts[-1] == len(weights) and cuts[1:] == sorted(set(cuts[1:]))
    return cuts


def test_where_a_call_is_cut_into_overlapped_parts():
    """Host logic of the overlapped call (no device involved): a small call stays whole, from 12 288 pairs on it is cut
    in two halves, from 30 720 in four growing parts, and a large one into parts of 98 304 pairs behind a ramp of an
    eighth, a quarter and a half; a call that shares the device is cut at full parts only; the last part is never
    a sliver."""
    region = 600  # pairs per region
    parts = lambda cuts: [region * (b - a) for a, b in zip(cuts, cuts[1:])]
    assert _part_cuts([region] * 15) == [0, 15]                       # 9 000 pairs: one part
    halves = parts(_part_cuts([region] * 30))                         # 18 000 pairs
    assert len(halves) == 2 and min(halves) >= 6144
    four = parts(_part_cuts([region] * 100))                          # 60 000 pairs
    assert len(four) == 4 and four[:3] == sorted(four[:3]) and four[0] <= 60000 // 8 + region and min(four) >= 60000 // 8 and sum(four) == 60000
    mid = parts(_part_cuts([region] * 400))                           # 240 000 pairs: still four parts, none above a full part
    assert len(mid) == 4 and max(mid) <= 98304
    large = parts(_part_cuts([region] * 2000))                        # 1.2 M pairs
    assert large[0] <= 98304 // 8 + region and large[1] <= 98304 // 4 + region and large[2] <= 98304 // 2 + region
    assert all(98304 <= p < 98304 + region for p in large[3:-1]) and large[-1] >= 98304 // 2 and sum(large) == 1200000
    shared = parts(_part_cuts([region] * 400, shared=True))
    assert all(98304 <= p < 98304 + region for p in shared[:-1]) and len(shared) == 2
    assert _part_cuts([region] * 100, shared=True) == [0, 100]
    # one region heavier than a part is a part of its own; empty regions change nothing
    lumpy = _part_cuts([0, 500000, 0, 10, 0, 200000, 3, 0])
    assert lumpy[1] == 2 and lumpy[-1] == 8
    assert _part_cuts([]) == [0, 0]  # no regions: one empty part


def test_rows_per_lane_of_a_wave_group():
    """Host logic (no device involved): whatever rows-per-lane value the planner gives a wave-group, the group's longest
    read with its pad and sum rows fits the 64 lanes; the value is one the kernels are built for; a restricted set is
    honoured wherever one of its values fits, and ignored (not violated into a misfit) where none does."""
    lib = pairhmm.load_library()
    built = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16}
    for longest in range(1, 1023):
        k = lib.gatk_pairhmm_debug_group_rows(longest, 0)
        assert k in built and 64 * k >= longest + 2, (longest, k)
        assert k <= 12 or longest + 2 > 64 * 12  # 14 and 16 rows cost a wave per SIMD: only for reads that need them
    allowed = (1 << 9) | (1 << 10)
    for longest in range(1, 1023):
        k = lib.gatk_pairhmm_debug_group_rows(longest, allowed)
        assert k in built and 64 * k >= longest + 2
        if longest + 2 <= 64 * 10:
            assert k in (9, 10), (longest, k)
    assert lib.gatk_pairhmm_debug_group_rows(0, 0) == -1 and lib.gatk_pairhmm_debug_group_rows(1023, 0) == -1
    # the HaplotypeCaller read lengths: five values in use, each a tight fit
    assert {lib.gatk_pairhmm_debug_group_rows(n, 0) for n in range(121, 152)} == {7, 8, 9, 10, 11}
