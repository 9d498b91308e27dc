"""Parity of the HIP engine with the CPU oracle, through the C-ABI, on a real MI355X.

Tolerance: |log10 lk (engine) - log10 lk (fp64 linear-space oracle)| <= 1e-3 absolute, the bound
stated for the path (fp32 evaluation with an fp64 rerun below the underflow threshold).  -infinity
must match exactly.  Where the engine is forced to fp64 the bound is 1e-9.
"""
import ctypes as C
import os

import numpy as np
import pytest

from gatk_amd import pairhmm, sharding, synth
from gatk_amd.batch import Batch, pack
from gatk_amd.pairhmm import ReadLikelihoodsMatrix

pytestmark = pytest.mark.gpu

TOL = 1e-3
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pairhmm_golden.npz")


def ref(oracle, b, threads=8):
    rc, out = oracle.compute_likelihoods(b, oracle.LOGLESS, fast_threads=threads)
    assert rc == 0
    return out


def assert_parity(got, want, tol=TOL):
    assert got.shape == want.shape
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), "-infinity pattern differs"
    assert not np.isnan(got).any()
    assert (got <= 0).all()
    if fin.any():
        err = np.abs(got[fin] - want[fin]).max()
        assert err <= tol, f"max |d log10 lk| = {err}"


def test_closed_form_known_answers(hmm):
    b = pack([("A", [30], [45], [45], [10])], ["A", "C", "AC"])
    got = hmm.compute_batch(b)
    np.testing.assert_allclose(got[0], [-0.0461920023, -3.5228787453, -0.3470771124], atol=1e-6)


@pytest.mark.parametrize("name", ["c1_slice", "c5_slice", "uniform_quals", "ragged", "wildcards"])
def test_golden_fixtures(hmm, name):
    z = np.load(GOLDEN)
    b = Batch(*[z[f"{name}.{f}"] for f in ("read_bases", "read_quals", "ins_gop", "del_gop", "gcp", "read_offsets", "hap_bases", "hap_offsets")])
    got = hmm.compute_batch(b)
    assert_parity(got, z[f"{name}.expected_logless"])
    # the reference's own Java on the same inputs, once a machine with a JDK has produced it (tests/golden_reference.py)
    from golden_reference import load_reference_outputs
    ref_out = load_reference_outputs()
    if ref_out is not None and "exact" in ref_out.get(name, {}):
        assert_parity(got, ref_out[name]["exact"])


def test_config_c1_full(hmm, oracle):
    b = synth.config_c1()
    got = hmm.compute_batch(b)
    assert_parity(got, ref(oracle, b))
    st = hmm.stats()
    assert st["pairs"] == 128 * 16 and st["cells"] == b.cells


@pytest.mark.parametrize("stream_mode", [1, 0, -1])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16])
def test_every_rows_per_lane_variant(hmm, oracle, k, stream_mode):
    # reads of 1..60 bases (k = 1 needs <= 62 rows); every lane-segment packing is exercised, with
    # haplotype runs and with one haplotype per work item
    b = synth.make_batch(40, 5, 60, 130, seed=30 + k, len_jitter=59, gop_model="random")
    want = ref(oracle, b)
    hmm.set_option("force_rows_per_lane", k)
    hmm.set_option("stream_mode", stream_mode)
    try:
        assert_parity(hmm.compute_batch(b), want)
    finally:
        hmm.set_option("force_rows_per_lane", 0)
        hmm.set_option("stream_mode", 1)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_sweep_without_lane_masks_on_short_and_ragged_shapes(hmm, oracle, k):
    """Up to 6 rows per lane the one-haplotype table-mode kernel runs fill, steady steps and drain in one loop without
    lane masks (a table column stands for the positions outside a haplotype).  Haplotypes shorter than the pipeline,
    of 1 to 3 bases, of lengths around the four- and eight-step blocks, reads of every length a lane segment can hold,
    several segments of different lengths in one wave-group."""
    rng = np.random.default_rng(700 + k)
    longest = min(64 * k - 2, 150)
    reads = []
    for n in [1, 2, 3, longest, longest - 1, max(1, longest // 2)] + [int(rng.integers(1, longest + 1)) for _ in range(34)]:
        bases = rng.choice(list(b"ACGTN"), size=n, p=[0.24, 0.24, 0.24, 0.24, 0.04]).astype(np.uint8)
        quals = rng.integers(2, 42, size=n).astype(np.uint8)
        ins = rng.integers(1, 60, size=n).astype(np.uint8)
        dele = rng.integers(1, 60, size=n).astype(np.uint8)
        gcp = rng.integers(5, 30, size=n).astype(np.uint8)
        reads.append((bases, quals, ins, dele, gcp))
    for hap_lens in ([1, 2, 3], [4, 5, 7, 8, 9], [11, 12, 13, 15, 16, 17, 40], [63, 64, 65, 130, 131]):
        haps = [rng.choice(list(b"ACGT"), size=n).astype(np.uint8) for n in hap_lens]
        b = pack(reads, haps)
        want = ref(oracle, b)
        hmm.set_option("force_rows_per_lane", k)
        hmm.set_option("stream_mode", 0)  # one haplotype per work item
        try:
            assert_parity(hmm.compute_batch(b), want)
            st = hmm.stats()
            assert st["main_rows_per_lane"] == k and st["main_stream_pass"] == 0 and st["main_emit_mode"] in (2, 3)
        finally:
            hmm.set_option("force_rows_per_lane", 0)
            hmm.set_option("stream_mode", 1)


def test_long_read_indel_stress_uses_the_fp64_rerun(hmm, oracle):
    b = synth.config_c5(num_reads=96, num_haps=8)
    got = hmm.compute_batch(b)
    st = hmm.stats()
    assert_parity(got, ref(oracle, b))
    assert st["rerun_tasks"] > 0, "this workload is meant to push pairs under the fp32 threshold"


def test_threshold_sweep(hmm, oracle):
    """The rerun branch is data dependent: force it for every pair (threshold = +inf -> all fp64) and
    for none that the default would not take, and require both to meet the bound."""
    b = synth.config_c5(seed=77, num_reads=48, num_haps=6)
    want = ref(oracle, b)
    try:
        hmm.set_option("fp32_threshold", 3e38)
        all64 = hmm.compute_batch(b)
        assert hmm.stats()["rerun_pairs"] == hmm.stats()["pairs"]
        assert_parity(all64, want, tol=1e-9)
        hmm.set_option("fp32_threshold", 1e-28)
        default = hmm.compute_batch(b)
        assert_parity(default, want)
        assert np.abs(default - all64)[np.isfinite(want)].max() <= TOL
    finally:
        hmm.set_option("fp32_threshold", 1e-28)


def test_forced_fp64_matches_oracle_closely(hmm, oracle):
    b = synth.make_batch(32, 6, 151, 420, seed=41, gop_model="random")
    hmm.set_option("force_fp64", 1)
    try:
        assert_parity(hmm.compute_batch(b), ref(oracle, b), tol=1e-9)
    finally:
        hmm.set_option("force_fp64", 0)


@pytest.mark.parametrize("length", [300, 640, 700, 760, 830, 900, 1000, 1022])
def test_long_reads_at_many_rows_per_lane(hmm, oracle, length):
    """Reads that need 5 to 16 rows per lane with all 64 lanes at work, close enough to their haplotypes for finite
    likelihoods (a random long read underflows to -infinity, which hides a wrong sum): through the fp32 pass with
    its fp64 pair pass behind it, through one haplotype per work item, and with everything forced through fp64."""
    rng = np.random.default_rng(1000 + length)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 1100)]
    read = hap[30:30 + length].copy()
    read[::37] = acgt[(np.searchsorted(acgt, read[::37]) + 1) % 4]  # a mismatch every 37 bases
    b = pack([(read, rng.integers(10, 40, length), None, None, None)], [hap, hap[:1000]])
    want = ref(oracle, b, 1)
    assert np.isfinite(want).all()
    try:
        for mode in (1, 0, -1):
            hmm.set_option("stream_mode", mode)
            assert_parity(hmm.compute_batch(b), want)
        hmm.set_option("force_fp64", 1)
        for mode in (1, 0):
            hmm.set_option("stream_mode", mode)
            assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    finally:
        hmm.set_option("force_fp64", 0)
        hmm.set_option("stream_mode", 1)


@pytest.mark.parametrize("read_len,jitter", [(620, 40), (800, 150), (1000, 300)])
def test_batches_of_long_reads(hmm, oracle, read_len, jitter):
    """Several wave-groups of long reads (11 to 16 rows per lane, lengths mixed) against haplotype runs, read from
    their haplotypes so that the likelihoods are finite: stream pass with its pair pass, one haplotype per work
    item with its fp64 list pass, and everything in fp64."""
    b = synth.make_batch(24, 6, read_len, 1100, seed=read_len, len_jitter=jitter, gop_model="random")
    want = ref(oracle, b)
    assert np.isfinite(want).all()
    try:
        for mode in (1, 0):
            hmm.set_option("stream_mode", mode)
            assert_parity(hmm.compute_batch(b), want)
        hmm.set_option("force_fp64", 1)
        assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    finally:
        hmm.set_option("force_fp64", 0)
        hmm.set_option("stream_mode", 1)


def _near_copy(rng, hap, start, length, every=37):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    read = hap[start:start + length].copy()
    read[::every] = acgt[(np.searchsorted(acgt, read[::every]) + 1) % 4]
    return read


@pytest.mark.parametrize("length", [1023, 1030, 1534, 2047, 2500, 4099])
def test_reads_longer_than_a_wave_holds(hmm, oracle, length):
    """A read of more than 1022 bases has more rows than the 64 x 16 of one wave: its pairs go through the long-read
    pass (double, chunk of rows by chunk of rows).  Lengths around the chunk edges (a chunk is 512 rows: 1023 and
    1030 take three, 1534 exactly three with one pad row, 2047 four, 4099 nine), each against a longer haplotype,
    a shorter one and one with 'N's, cut from their haplotype so that the likelihoods are finite."""
    rng = np.random.default_rng(7000 + length)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, length + 170)]
    with_n = hap[:length + 90].copy()
    with_n[::101] = ord("N")
    read = _near_copy(rng, hap, 40, length)
    quals = rng.integers(10, 40, length)
    gop = rng.integers(30, 50, length)
    b = pack([(read, quals, gop, gop[::-1].copy(), rng.integers(8, 12, length))], [hap, hap[:length - 5], with_n])
    want = ref(oracle, b, 1)
    assert np.isfinite(want).all()
    assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    hmm.set_option("force_fp64", 1)
    try:
        assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    finally:
        hmm.set_option("force_fp64", 0)


@pytest.mark.parametrize("num_haps", [150, 300, 400, 600])
def test_long_read_workgroups_of_every_size(hmm, oracle, num_haps):
    """The waves of a long-read workgroup share a pair's chunks of rows, and the fewer pairs a call has the more waves
    a workgroup gets: one read of 1600 bases (four chunks) against 150, 300, 400 and 600 haplotypes runs with four,
    three, two and one wave per pair - with three and two the boundary rows are reused within a pair.  Workgroups of
    one and two waves take their emission terms from a table in LDS when all pairs are on the device at once, and
    compare bytes like the larger ones when the engine is told to (emit_mode 0) or a haplotype holds another letter."""
    rng = np.random.default_rng(7600 + num_haps)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 1700)]
    haps = []
    for a in range(num_haps):
        x = hap[:1700 - a % 7].copy()
        x[3 * a % 1600] = acgt[(np.searchsorted(acgt, x[3 * a % 1600]) + 1) % 4]
        haps.append(x)
    b = pack([(_near_copy(rng, hap, 30, 1600), rng.integers(10, 40, 1600), None, None, None)], haps)
    want = ref(oracle, b)
    assert np.isfinite(want).all()
    assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    hmm.set_option("emit_mode", 0)
    try:
        assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    finally:
        hmm.set_option("emit_mode", -1)
    if num_haps == 150:  # raw bytes are compared (Log10PairHMM.java:156): these two match nothing in the read
        first, second = haps[0].copy(), haps[1].copy()
        first[5], second[9] = ord("a"), ord("R")
        odd = pack([(b.read_bases, b.read_quals, None, None, None)], [first, second, haps[2]])
        assert_parity(hmm.compute_batch(odd), ref(oracle, odd), tol=1e-9)


def test_long_read_against_short_haplotypes(hmm, oracle):
    """Haplotypes shorter than one block of 64 columns, of exactly one, and a little more, against a read of four
    chunks of rows: the hand-off between chunks when a block is cut short.  Cheap gaps (open and continuation quality 1)
    keep the thousand insertions such a pair needs finite."""
    rng = np.random.default_rng(75)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    n = 1700
    hap = acgt[rng.integers(0, 4, 200)]
    read = np.resize(hap, n).copy()
    ones = np.ones(n, np.uint8)
    b = pack([(read, rng.integers(10, 40, n), ones, ones, ones)], [hap[:1], hap[:5], hap[:63], hap[:64], hap[:65], hap[:130], hap])
    want = ref(oracle, b, 1)
    assert np.isfinite(want).all()
    assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    hmm.set_option("emit_mode", 0)
    try:
        assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    finally:
        hmm.set_option("emit_mode", -1)


def test_long_reads_through_the_other_entry_points(oracle):
    """A region with two reads for the long-read pass among ordinary ones through the per-pair call, BatchPairHMM,
    a resident batch computed twice, and the region-level hook followed by computeLikelihoods: the same likelihoods as
    from compute_batch, which the oracle vouches for."""
    rng = np.random.default_rng(76)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 1500)]
    lens = [140, 1300, 151, 1023, 90]
    recs = []
    for n in lens:
        gop = rng.integers(30, 50, n).astype(np.uint8)
        recs.append((_near_copy(rng, hap, int(rng.integers(0, 100)), n), rng.integers(10, 40, n).astype(np.uint8), gop, gop[::-1].copy(),
                     np.full(n, 10, np.uint8)))
    haps = [hap, hap[:1450], hap[25:]]
    b = pack(recs, haps)
    want = ref(oracle, b, 1)
    assert np.isfinite(want).all()
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        assert_parity(h.compute_batch(b), want)
        h.initialize(1400, 1600)
        for r in (1, 3):
            lk = h.computeReadLikelihoodGivenHaplotypeLog10(bytes(haps[1]), bytes(recs[r][0]), *recs[r][1:])
            assert abs(lk - want[r, 1]) <= 1e-9
        for rd in recs:
            h.batchAdd([bytes(x) for x in haps], bytes(rd[0]), *rd[1:])
        assert_parity(h.batchGetResult().reshape(want.shape), want)
        res = h.upload([b, b])
        try:
            for _ in range(2):
                res.compute()
                for got in res.download():
                    assert_parity(got, want)
        finally:
            res.free()
        h.initialize_region_batches([b], 1400, 1600)
        assert_parity(h.compute_batch(b), want)
        assert h.stats()["region_hits"] >= 1
        h.finalizeRegion()
    finally:
        h.close()


def test_a_long_read_wave_that_waits_in_vain_gives_up(oracle):
    """Inside a long-read workgroup a wave waits for the chunk above its own.  Every such wait is bounded: with the
    first chunk's progress withheld (a test hook) the second chunk's wave gives up after about half a second, the
    launch drains, and the call comes back as a device error instead of never; the handle works afterwards."""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 1400)]
    b = pack([(_near_copy(rng, hap, 10, 1300), rng.integers(10, 40, 1300), None, None, None)], [hap])
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        h.set_option("debug_long_stall", 1)
        with pytest.raises(pairhmm.DeviceError, match="gave up waiting"):
            h.compute_batch(b)
        h.set_option("debug_long_stall", 0)
        assert_parity(h.compute_batch(b), ref(oracle, b, 1), tol=1e-9)
    finally:
        h.close()


def test_long_reads_among_ordinary_ones(hmm, oracle):
    """Reads for the long-read pass, ordinary reads, a read of exactly 1022 bases (the longest a wave-group takes) and
    a zero-length read in one region, next to a region of ordinary reads and one that holds a long read only: every
    row of the result is its read's, whichever pass scored it, in one call and in regions."""
    rng = np.random.default_rng(71)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 2100)]
    other = hap.copy()
    other[500] = acgt[(np.searchsorted(acgt, other[500]) + 2) % 4]
    def rec(start, n):
        return (_near_copy(rng, hap, start, n), rng.integers(10, 40, n), None, None, None)
    mixed = pack([rec(0, 150), rec(10, 1800), rec(300, 1022), ("", [], [], [], []), rec(100, 1023), rec(700, 90), rec(5, 2000)],
                 [hap, other, hap[:1900]])
    ordinary = synth.make_batch(12, 5, 120, 300, seed=72)
    alone = pack([rec(20, 1300)], [hap[:1500]])
    wants = [ref(oracle, x, 1) for x in (mixed, ordinary, alone)]
    assert np.isfinite(wants[0][[0, 1, 2, 4, 5, 6]]).all() and np.isneginf(wants[0][3]).all()
    assert_parity(hmm.compute_batch(mixed), wants[0])
    for got, want in zip(hmm.compute_regions([mixed, ordinary, alone]), wants):
        assert_parity(got, want)
    for got, want in zip(hmm.compute_regions([alone, mixed]), (wants[2], wants[0])):
        assert_parity(got, want)


def test_the_longest_reads_the_engine_takes(hmm, oracle):
    """15900 bases (32 chunks of rows) against the longest haplotype the engine takes, which the read was cut from: a
    finite likelihood, the oracle's.  And 32766 bases, the longest read: against a haplotype of 16000 bases half of
    it is insertions, and the sum underflows to zero in the reference's arithmetic as in the engine's."""
    rng = np.random.default_rng(73)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 16000)]
    n = 15900
    read = _near_copy(rng, hap, 50, n, every=997)
    b = pack([(read, rng.integers(25, 40, n), None, None, None)], [hap])
    want = ref(oracle, b, 1)
    assert np.isfinite(want).all()
    assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    n = 32766
    read = np.concatenate([hap, hap, hap[:766]])
    b = pack([(read, rng.integers(25, 40, n), None, None, None)], [hap, hap[:200]])
    want = ref(oracle, b, 1)
    assert np.isneginf(want).all()
    assert_parity(hmm.compute_batch(b), want)


def test_long_reads_in_a_call_cut_into_overlapped_parts(hmm, oracle):
    """Regions with reads for the long-read pass in a call that is cut into parts (each part plans its own long-read
    launch on its own stream and boundary buffer), blocking and through submit / fetch."""
    rng = np.random.default_rng(74)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    regions = []
    for k in range(12):
        hap = acgt[rng.integers(0, 4, 1700)]
        recs = [(_near_copy(rng, hap, int(rng.integers(0, 100)), int(n)), rng.integers(10, 40, int(n)), None, None, None)
                for n in rng.choice([80, 151, 151, 1100, 1500], size=8)]
        regions.append(pack(recs, [hap, hap[:1650], hap[20:]]))
        regions.append(synth.make_batch(30, 6, 151, 400, seed=740 + k))
    wants = [ref(oracle, b) for b in regions]
    for got, want in zip(with_options_pipeline(hmm, 200, lambda: hmm.compute_regions(regions)), wants):
        assert_parity(got, want)
    tickets = [hmm.submit(regions[i:i + 6]) for i in range(0, len(regions), 6)]
    outs = [o for t in tickets for o in t.fetch()]
    for got, want in zip(outs, wants):
        assert_parity(got, want)


def test_a_long_read_with_a_bad_quality_raises_as_any_other(hmm):
    n = 1500
    quals = [30] * n
    quals[1400] = 255
    with pytest.raises(pairhmm.ArrayIndexOutOfBoundsException):
        hmm.compute_batch(pack([("ACGT" * (n // 4), quals, None, None, None)], ["ACGT" * 400]))
    gop = [45] * n
    gop[1234] = 200
    with pytest.raises(pairhmm.IllegalArgumentException):
        hmm.compute_batch(pack([("ACGT" * (n // 4), [30] * n, gop, None, None)], ["ACGT" * 400]))


def test_underflow_to_minus_infinity(hmm, oracle):
    b = pack([("A" * 400, [93] * 400, [93] * 400, [93] * 400, [93] * 400), ("ACGT" * 20, [30] * 80, None, None, None)],
             ["C" * 400, "ACGT" * 30])
    got = hmm.compute_batch(b)
    want = ref(oracle, b, threads=1)
    assert want[0, 0] == -np.inf
    assert_parity(got, want)


def test_wildcards_and_case(hmm, oracle):
    # read 'N' and haplotype 'N' match everything; comparison is on raw bytes (Log10PairHMM.java:156)
    reads = [("ACNTNACGTA", [30, 2, 40, 0, 11, 25, 254, 33, 7, 60], [45] * 10, [40] * 10, [10] * 10),
             ("NNNN", [20] * 4, None, None, None), ("acgt", [20] * 4, None, None, None), ("ACGT", [20] * 4, None, None, None)]
    with_n = pack(reads, ["ACGTNACGTAAC", "NNNNNNNN", "ACGT", "acgtacgt"])
    assert_parity(hmm.compute_batch(with_n), ref(oracle, with_n, 1))
    without_n = pack(reads, ["ACGTAACGTAAC", "ACGT", "acgtacgt"])
    assert_parity(hmm.compute_batch(without_n), ref(oracle, without_n, 1))


def test_shapes_at_the_edges(hmm, oracle):
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    def seq(n):
        return acgt[rng.integers(0, 4, n)]
    cases = [
        pack([(seq(1), [30], None, None, None)], [seq(1)]),                       # 1 x 1
        pack([(seq(1), [30], None, None, None)], [seq(700)]),                     # one row, long haplotype
        pack([(seq(200), [25] * 200, None, None, None)], [seq(3)]),               # haplotype shorter than read
        pack([(seq(64), [30] * 64, None, None, None), (seq(65), [30] * 65, None, None, None)], [seq(90)]),  # 64 / 65 rows
        pack([(seq(511), rng.integers(10, 40, 511), None, None, None)], [seq(600), seq(513)]),  # 8 rows per lane
        pack([(seq(1022), rng.integers(10, 40, 1022), None, None, None)], [seq(1100), seq(700)]),  # longest read
        pack([(seq(33), [30] * 33, None, None, None)] * 70, [seq(50)] * 3),       # many equal reads: several groups
    ]
    for b in cases:
        assert_parity(hmm.compute_batch(b), ref(oracle, b, 1))


def test_zero_length_read_and_empty_batch(hmm, oracle):
    b = pack([("ACGT", [30] * 4, None, None, None), ("", [], [], [], []), ("GG", [30] * 2, None, None, None)], ["ACGTA", "CC"])
    got = hmm.compute_batch(b)
    assert_parity(got, ref(oracle, b, 1))
    assert (got[1] == -np.inf).all()
    # no reads: computeLikelihoods returns at once (PairHMM.java:193)
    empty = pack([], ["ACGT"])
    assert hmm.compute_batch(empty).shape == (0, 1)


def test_staging_buffer_that_has_to_grow_while_the_read_copies_run(oracle):
    """The staging buffer is sized from the call's reads, haplotypes and regions before the per-base arrays are copied
    into it; the output slots of zero-length reads are listed in it as well, and thousands of them against a hundred
    haplotypes outgrow that size on a fresh handle: the buffer moves with the read arrays already in it.  Every
    ordinary read's likelihoods must come out as from a call of their own."""
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        front = synth.make_batch(20, 4, 80, 200, seed=5)
        mixed = synth.make_batch(6, 100, 60, 150, seed=6)
        lens = np.concatenate([np.diff(mixed.read_offsets), np.zeros(4000, dtype=np.int64)])
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        many_empty = Batch(mixed.read_bases, mixed.read_quals, mixed.ins_gop, mixed.del_gop, mixed.gcp, offsets, mixed.hap_bases, mixed.hap_offsets)
        got = h.compute_regions([front, many_empty])
        assert_parity(got[0], ref(oracle, front))
        assert_parity(got[1][:6], ref(oracle, mixed))
        assert got[1].shape == (4006, 100) and np.isneginf(got[1][6:]).all()
    finally:
        h.close()


def test_error_conventions(hmm):
    ok = ("ACGT", [30] * 4, [45] * 4, [45] * 4, [10] * 4)
    for k in (2, 3, 4):  # negative Java byte in ins / del / gcp (PairHMMModel.java:136-138)
        rec = list(ok)
        rec[k] = [45, 45, 200, 45]
        with pytest.raises(pairhmm.IllegalArgumentException):
            hmm.compute_batch(pack([tuple(rec)], ["ACGTA"]))
    rec = list(ok)
    rec[1] = [30, 255, 30, 30]  # QualityUtils.java:63,173
    with pytest.raises(pairhmm.ArrayIndexOutOfBoundsException):
        hmm.compute_batch(pack([tuple(rec)], ["ACGTA"]))
    rec[1] = [30, 254, 30, 30]
    assert np.isfinite(hmm.compute_batch(pack([tuple(rec)], ["ACGTA"]))).all()
    with pytest.raises(pairhmm.IllegalArgumentException):  # READ_MAX_LENGTH must be > 0 (PairHMM.java:105)
        hmm.compute_batch(pack([("", [], [], [], [])], ["ACGT"]))
    with pytest.raises(pairhmm.IllegalArgumentException):  # HAPLOTYPE_MAX_LENGTH must be > 0 (PairHMM.java:106)
        hmm.compute_batch(pack([ok], []))
    with pytest.raises(pairhmm.IllegalArgumentException):  # zero-length haplotype: undefined in the reference
        hmm.compute_batch(pack([ok], ["ACGT", ""]))
    with pytest.raises(pairhmm.IllegalArgumentException, match="too long"):  # read longer than the engine's maximum
        hmm.compute_batch(pack([("A" * 32767, [30] * 32767, None, None, None)], ["ACGT"]))
    with pytest.raises(pairhmm.IllegalArgumentException):
        hmm.initialize(0, 10)
    with pytest.raises(pairhmm.IllegalArgumentException):
        hmm.initialize(10, -1)
    # the handle still works after errors
    assert np.isfinite(hmm.compute_batch(pack([ok], ["ACGTA"]))).all()


def test_single_pair_entry_point(oracle):
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        args = (b"ACGTTGCA", b"ACGTGCA", [30] * 7, [45] * 7, [45] * 7, [10] * 7)
        with pytest.raises(pairhmm.IllegalStateException):  # PairHMM.java:271
            h.computeReadLikelihoodGivenHaplotypeLog10(*args)
        h.initialize(7, 8)
        got = h.computeReadLikelihoodGivenHaplotypeLog10(*args)
        want = ref(oracle, pack([(args[1], args[2], args[3], args[4], args[5])], [args[0]]), 1)[0, 0]
        assert abs(got - want) <= TOL
        with pytest.raises(pairhmm.IllegalArgumentException):  # PairHMM.java:273
            h.computeReadLikelihoodGivenHaplotypeLog10(b"ACGTTGCAA", *args[1:])
        with pytest.raises(pairhmm.IllegalArgumentException):  # PairHMM.java:275
            h.computeReadLikelihoodGivenHaplotypeLog10(args[0], b"ACGTGCAA", [30] * 8, [45] * 8, [45] * 8, [10] * 8)
        with pytest.raises(pairhmm.IllegalArgumentException):  # PairHMM.java:276
            h.computeReadLikelihoodGivenHaplotypeLog10(args[0], args[1], [30] * 6, *args[3:])
    finally:
        h.close()


def test_compute_likelihoods_fills_matrix_and_array(hmm, oracle):
    reads = [("ACGTACGTAC", [30] * 10, None, None), ("TTGCA", [25] * 5, [40] * 5, [41] * 5), ("GGGGGGG", [35] * 7, None, None)]
    haps = ["ACGTACGTACGT", "TTGCATTGCA"]
    m = ReadLikelihoodsMatrix(haps, reads)
    hmm.computeLikelihoods(m, reads, gcp={0: [10] * 10, 1: [12] * 5, 2: [10] * 7})
    b = pack([(r[0], r[1], r[2], r[3], g) for r, g in zip(reads, ([10] * 10, [12] * 5, [10] * 7))], haps)
    want = ref(oracle, b, 1)
    assert m.values.shape == (2, 3)  # [allele][read], ReadLikelihoods.java:1638-1640
    assert_parity(m.values.T.copy(), want)
    flat = hmm.getLikelihoodArray()  # read-major, PairHMM.java:206,225
    assert flat.shape == (6,) and np.array_equal(flat.reshape(3, 2), m.values.T)
    assert m.get(1, 2) == m.values[1, 2]
    hmm.computeLikelihoods(ReadLikelihoodsMatrix(haps, []), [])  # no-op
    hmm.finalizeRegion()


def test_batch_pair_hmm_interface(hmm, oracle):
    haps = [b"ACGTACGTACGT", b"TTGCATTGCA", b"ACGTAC"]
    r1 = (b"ACGTACG", [30] * 7, [45] * 7, [45] * 7, [10] * 7)
    r2 = (b"TTGCA", [20] * 5, [40] * 5, [40] * 5, [10] * 5)
    hmm.batchAdd(haps, *r1)
    hmm.batchAdd(haps[:2], *r2)
    got = hmm.batchGetResult()
    want = np.concatenate([ref(oracle, pack([r1], haps), 1)[0], ref(oracle, pack([r2], haps[:2]), 1)[0]])
    assert got.shape == (5,) and np.abs(got - want).max() <= TOL
    assert hmm.batchGetResult().shape == (0,)  # queue was cleared


def test_batch_pair_hmm_reads_against_one_haplotype_list_run_as_one_region(oracle):
    """BatchPairHMM.batchAdd once per read of a region against the region's haplotypes (BatchPairHMM.java:33-40):
    the queued reads are scored as one region, so batchGetResult returns what one computeLikelihoods call over the
    same reads returns, in about the same time; a change of haplotype list in the middle starts a new region."""
    import time
    b = max(synth.make_regions(6, seed=123, coverage=28, read_len=140), key=lambda x: x.num_reads)
    other = synth.make_regions(1, seed=124, coverage=6, read_len=100)[0]
    assert b.num_reads >= 40

    def reads_of(x):
        o = x.read_offsets
        return [(bytes(x.read_bases[o[r]:o[r + 1]]), x.read_quals[o[r]:o[r + 1]], x.ins_gop[o[r]:o[r + 1]], x.del_gop[o[r]:o[r + 1]],
                 x.gcp[o[r]:o[r + 1]]) for r in range(x.num_reads)]

    def haps_of(x):
        o = x.hap_offsets
        return [bytes(x.hap_bases[o[a]:o[a + 1]]) for a in range(x.num_haps)]

    h = pairhmm.HipLoglessPairHMM(0)
    try:
        whole = h.compute_batch(b)
        for rd in reads_of(b):
            h.batchAdd(haps_of(b), *rd)
        got = h.batchGetResult()
        assert np.array_equal(got, whole.reshape(-1))
        assert h.stats()["pairs"] == b.num_reads * b.num_haps  # one region, not one per read
        assert_parity(got.reshape(whole.shape), ref(oracle, b))
        # two lists in turn: the first half of b, all of `other`, the second half of b
        half = b.num_reads // 2
        for rd in reads_of(b)[:half]:
            h.batchAdd(haps_of(b), *rd)
        for rd in reads_of(other):
            h.batchAdd(haps_of(other), *rd)
        for rd in reads_of(b)[half:]:
            h.batchAdd(haps_of(b), *rd)
        got = h.batchGetResult()
        n1, n2 = half * b.num_haps, other.num_reads * other.num_haps
        want = np.concatenate([ref(oracle, b).reshape(-1)[:n1], ref(oracle, other).reshape(-1), ref(oracle, b).reshape(-1)[n1:]])
        assert got.shape == want.shape
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin) and np.abs(got[fin] - want[fin]).max() <= TOL
        assert n2 > 0
        # time: the queue against one call (marshalling of the adds not counted: the clock is on batchGetResult)
        keep, best_q, best_c = reads_of(b), 1e9, 1e9
        for _ in range(5):
            for rd in keep:
                h.batchAdd(haps_of(b), *rd)
            t0 = time.perf_counter()
            h.batchGetResult()
            best_q = min(best_q, time.perf_counter() - t0)
            t0 = time.perf_counter()
            h.compute_batch(b)
            best_c = min(best_c, time.perf_counter() - t0)
        assert best_q <= 1.3 * best_c + 30e-6, (best_q, best_c)
    finally:
        h.close()


def test_pointer_array_entry_point(hmm, oracle):
    b = synth.make_batch(9, 4, 50, 100, seed=8, len_jitter=30)
    lib = pairhmm.load_library()
    nr, nh = b.num_reads, b.num_haps
    def ptrs(arr, offs):
        a = np.ascontiguousarray(arr)
        return a, (C.c_void_p * (len(offs) - 1))(*[a.ctypes.data + int(o) for o in offs[:-1]])
    keep = [ptrs(x, b.read_offsets) for x in (b.read_bases, b.read_quals, b.ins_gop, b.del_gop, b.gcp)]
    hp = ptrs(b.hap_bases, b.hap_offsets)
    rl, hl = np.diff(b.read_offsets).astype(np.int32), np.diff(b.hap_offsets).astype(np.int32)
    out = np.empty((nr, nh))
    rc = lib.gatk_pairhmm_compute_likelihoods_ptrs(hmm._h, nr, nh, keep[0][1], keep[1][1], keep[2][1], keep[3][1], keep[4][1],
                                                   rl.ctypes.data, hp[1], hl.ctypes.data, out.ctypes.data)
    assert rc == 0
    assert_parity(out, ref(oracle, b, 1))


def test_many_regions_in_one_pass(hmm, oracle):
    regions = synth.make_regions(24, seed=11, coverage=8, read_len=100)
    regions.insert(5, pack([], ["ACGT"]))  # a region without reads is skipped
    outs = hmm.compute_regions(regions)
    assert len(outs) == len(regions)
    for b, o in zip(regions, outs):
        if b.num_reads:
            assert_parity(o, ref(oracle, b))
    res = hmm.upload(regions)
    try:
        res.compute(sync=True)
        for o, o2 in zip(outs, res.download()):
            if o.size:
                np.testing.assert_array_equal(o, o2)
    finally:
        res.free()


def test_sharded_regions_single_rank(hmm, oracle):
    regions = synth.make_regions(6, seed=12, coverage=6, read_len=80)
    outs = sharding.compute_sharded(regions, hmm.compute_regions, rank=0, world=1)
    for b, o in zip(regions, outs):
        assert_parity(o, ref(oracle, b))
    big = synth.make_batch(40, 4, 90, 200, seed=13)
    assert_parity(sharding.compute_split_batch(big, hmm.compute_regions), ref(oracle, big))


def test_prepared_call_scores_into_a_given_buffer(hmm):
    """prepare_regions marshals a compute_regions call once; with flat= it scores into the caller's buffer (the
    slice of a shared-memory segment when processes shard a list, sharding.SharedOutput)."""
    regions = synth.make_regions(7, seed=33, coverage=8, read_len=100)
    want = hmm.compute_regions(regions)
    total = sum(b.num_reads * b.num_haps for b in regions)
    buf = np.full(total + 3, -7.0)
    call = hmm.prepare_regions(regions, flat=buf[1:1 + total])
    for _ in range(2):
        outs = call.compute()
        for o, w in zip(outs, want):
            np.testing.assert_array_equal(o, w)
    assert buf[0] == -7.0 and (buf[1 + total:] == -7.0).all()  # nothing outside the slice
    assert np.array_equal(buf[1:1 + total], np.concatenate([w.reshape(-1) for w in want]))
    own = hmm.prepare_regions(regions).compute()
    assert all(np.array_equal(o, w) for o, w in zip(own, want))
    with pytest.raises(ValueError):
        hmm.prepare_regions(regions, flat=np.zeros(total + 1))


def test_long_haplotype_and_length_cap(hmm, oracle):
    rng = np.random.default_rng(6)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 16000)]
    read = hap[5000:5100].copy()
    b = pack([(read, [30] * 100, None, None, None)], [hap, hap[:15000], hap[4000:9000]])
    want = ref(oracle, b, 1)
    assert_parity(hmm.compute_batch(b), want)                                              # haplotype runs
    assert_parity(with_options(hmm, {"stream_mode": 0}, lambda: hmm.compute_batch(b)), want)  # one haplotype per item
    # a long read against the same haplotypes: 16 rows per lane next to the largest LDS images
    long_read = hap[7000:7900].copy()
    long_read[::41] = acgt[(np.searchsorted(acgt, long_read[::41]) + 1) % 4]
    b2 = pack([(long_read, rng.integers(10, 40, 900), None, None, None), (read, [30] * 100, None, None, None)], [hap, hap[:15000], hap[4000:9000]])
    want2 = ref(oracle, b2, 1)
    assert np.isfinite(want2).all()
    assert_parity(hmm.compute_batch(b2), want2)
    assert_parity(with_options(hmm, {"stream_mode": 0}, lambda: hmm.compute_batch(b2)), want2)
    assert_parity(with_options(hmm, {"force_fp64": 1}, lambda: hmm.compute_batch(b2)), want2, tol=1e-9)
    with pytest.raises(pairhmm.IllegalArgumentException):
        hmm.compute_batch(pack([(read, [30] * 100, None, None, None)], [acgt[rng.integers(0, 4, 16001)]]))


def test_full_size_large_batch_properties(hmm, oracle):
    """4096 x 64, 151 bp / 600 bp (the benchmark workload), checked without scoring all of it on the
    CPU: (a) a random sample of reads against every haplotype vs the oracle; (b) read order does not
    change any value (pairs are independent, packing must not leak between lane segments);
    (c) every value is a finite log10 probability; (d) the resident path returns the same bits."""
    b = synth.config_c3()
    assert b.num_reads == 4096 and b.num_haps == 64 and b.cells == 4096 * 151 * 64 * 600
    got = hmm.compute_batch(b)
    st = hmm.stats()
    assert np.isfinite(got).all() and (got <= 0).all()
    rng = np.random.default_rng(99)
    pick = np.sort(rng.choice(b.num_reads, size=48, replace=False))
    sub = subset_reads(b, pick)
    assert_parity(got[pick], ref(oracle, sub))
    perm = rng.permutation(b.num_reads)
    shuffled = hmm.compute_batch(subset_reads(b, perm))
    np.testing.assert_array_equal(shuffled, got[perm])
    res = hmm.upload([b])
    try:
        res.compute(sync=True)
        np.testing.assert_array_equal(res.download()[0], got)
    finally:
        res.free()
    assert st["rerun_pairs"] / st["pairs"] < 0.01


def test_full_size_long_read_indel_stress_properties(hmm, oracle):
    """4096 x 32, 250 bp / 1000 bp with 4 % indels (the configuration with the fp64 reruns) at full size: a random
    sample of reads against every haplotype vs the oracle, read order changes no value (the pair pass redoes single
    pairs, whatever shared a wave with them), every value a finite log10 probability, and the rerun rate in the
    range the workload is built for."""
    b = synth.config_c5(num_reads=4096, num_haps=32)
    got = hmm.compute_batch(b)
    st = hmm.stats()
    assert np.isfinite(got).all() and (got <= 0).all()
    assert 0.02 < st["rerun_pairs"] / st["pairs"] < 0.10
    rng = np.random.default_rng(123)
    pick = np.sort(rng.choice(b.num_reads, size=40, replace=False))
    assert_parity(got[pick], ref(oracle, subset_reads(b, pick)))
    perm = rng.permutation(b.num_reads)
    shuffled = hmm.compute_batch(subset_reads(b, perm))
    np.testing.assert_array_equal(shuffled, got[perm])


def subset_reads(b, idx):
    parts = [slice(int(b.read_offsets[i]), int(b.read_offsets[i + 1])) for i in idx]
    cat = lambda a: np.concatenate([a[p] for p in parts])
    offs = np.zeros(len(idx) + 1, np.int32)
    offs[1:] = np.cumsum([p.stop - p.start for p in parts])
    return Batch(cat(b.read_bases), cat(b.read_quals), cat(b.ins_gop), cat(b.del_gop), cat(b.gcp), offs, b.hap_bases, b.hap_offsets)


def test_cxx_host_mirror():
    """include/gatk_pairhmm.hpp (the C++ stand-in for the Java PairHMM subclass) against closed-form answers."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gatk_amd", "cxx_driver")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "cxx_driver ok" in r.stdout, r.stdout + r.stderr


def test_zero_match_to_match_rows_go_through_fp64(hmm, oracle):
    """Insertion and deletion quality both <= 3 clamp matchToMatch to 0 (PairHMMModel.java:119): the fp32
    pass divides by it, must notice the NaN and hand the work item to the fp64 pass."""
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 120)]
    reads = []
    for k in range(12):
        n = 40 + k
        start = int(rng.integers(0, 120 - n))
        ins = rng.integers(20, 46, n)
        dele = rng.integers(20, 46, n)
        if k % 2 == 0:  # every other read has a stretch of rows with t0 = 0 (and one at the first row)
            ins[0] = dele[0] = 0
            ins[10:14] = rng.integers(0, 4, 4)
            dele[10:14] = rng.integers(0, 4, 4)
        reads.append((hap[start:start + n], rng.integers(10, 41, n), ins, dele, [10] * n))
    b = pack(reads, [hap, hap[::-1].copy()])
    want = ref(oracle, b, 1)
    got = hmm.compute_batch(b)
    assert hmm.stats()["rerun_tasks"] > 0
    assert_parity(got, want)
    hmm.set_option("emit_mode", 0)  # and through the byte-compare kernel variant
    try:
        assert_parity(hmm.compute_batch(b), want)
    finally:
        hmm.set_option("emit_mode", -1)


def test_extreme_gap_quality_steps(hmm, oracle):
    """Insertion quality jumping between 1 and 127 from row to row stretches the rescaled fp32 form
    (F = I / t2) over ~12 decades; whatever fp32 cannot hold must come back through the fp64 pass."""
    rng = np.random.default_rng(18)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 200)]
    reads = []
    for k in range(8):
        n = 90
        ins = np.where(np.arange(n) % 2 == 0, 127, 4 + k)
        dele = np.where(np.arange(n) % 3 == 0, 4, 127)
        reads.append((hap[20 + k:20 + k + n], rng.integers(5, 41, n), ins, dele, rng.integers(1, 60, n)))
    b = pack(reads, [hap, hap[10:150]])
    assert_parity(hmm.compute_batch(b), ref(oracle, b, 1))


# ---- haplotype runs (the fp32 stream pass) ------------------------------------------------------

def with_options(hmm, opts, fn):
    defaults = {"stream_mode": 1, "max_stream_haps": 16, "force_rows_per_lane": 0, "emit_mode": -1, "force_fp64": 0,
                "stream_run_divisor": 2, "group_rows_allowed": 0, "per_group_factor": 32}  # the session handle's
    for k, v in opts.items():
        hmm.set_option(k, v)
    try:
        return fn()
    finally:
        for k in opts:
            hmm.set_option(k, defaults[k])


@pytest.mark.parametrize("run", [1, 2, 3, 5, 16])
def test_haplotype_runs_of_every_length(hmm, oracle, run):
    # 23 haplotypes of ragged length: runs of `run` haplotypes plus a shorter last run
    b = synth.make_batch(50, 23, 101, 300, seed=500 + run, len_jitter=40, hap_len_jitter=120, gop_model="random")
    want = ref(oracle, b)
    got = with_options(hmm, {"max_stream_haps": run}, lambda: hmm.compute_batch(b))
    assert_parity(got, want)


@pytest.mark.parametrize("num_haps,num_reads", [(200, 50), (513, 7)])
def test_regions_with_hundreds_of_haplotypes(hmm, oracle, num_haps, num_reads):
    # far more haplotypes than the assembler's cap per region (128 per k-mer size)
    b = synth.make_batch(num_reads, num_haps, 80, 200, seed=num_haps, len_jitter=20)
    want = ref(oracle, b)
    for mode in (-1, 1, 0):
        got = with_options(hmm, {"stream_mode": mode}, lambda: hmm.compute_batch(b))
        assert_parity(got, want)


def test_runs_and_single_haplotype_items_agree(hmm, oracle):
    b = synth.make_batch(300, 40, 151, 420, seed=77, len_jitter=30, hap_len_jitter=60)
    runs = hmm.compute_batch(b)
    single = with_options(hmm, {"stream_mode": 0}, lambda: hmm.compute_batch(b))
    compare = with_options(hmm, {"emit_mode": 0}, lambda: hmm.compute_batch(b))
    want = ref(oracle, b)
    for got in (runs, single, compare):
        assert_parity(got, want)
    # the three fp32 passes evaluate the same expression tree cell by cell
    np.testing.assert_allclose(runs, single, rtol=0, atol=1e-6)
    np.testing.assert_allclose(runs, compare, rtol=0, atol=1e-6)


@pytest.mark.parametrize("k", [1, 2, 4, 8, 9, 10, 11, 12, 14, 16])
def test_runs_with_haplotypes_shorter_than_the_pipeline(hmm, oracle, k):
    # haplotypes of 1..12 bases against reads of up to 60 bases: with few rows per lane a read spans more
    # lanes than a haplotype has columns, so several gaps are inside the wave at once
    rng = np.random.default_rng(k)
    haps = ["".join(rng.choice(list("ACGT"), size=n)) for n in (1, 2, 3, 1, 12, 5, 7, 1, 1, 4, 9, 2, 6)]
    src = synth.make_batch(30, 1, 60, 200, seed=900 + k, len_jitter=59)
    b = Batch(src.read_bases, src.read_quals, src.ins_gop, src.del_gop, src.gcp, src.read_offsets,
              np.frombuffer("".join(haps).encode(), dtype=np.uint8).copy(),
              np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.int32))
    want = ref(oracle, b)
    got = with_options(hmm, {"force_rows_per_lane": k}, lambda: hmm.compute_batch(b))
    assert_parity(got, want)


def test_only_underflowing_pairs_go_through_the_fp64_pair_pass(hmm, oracle):
    b = synth.config_c5(num_reads=96, num_haps=12)
    want = ref(oracle, b)
    got = hmm.compute_batch(b)
    assert_parity(got, want)
    st = hmm.stats()
    assert 0 < st["rerun_pairs"] < st["pairs"]
    # the one-haplotype passes queue whole wave-groups: more pairs recomputed for the same result
    single = with_options(hmm, {"stream_mode": 0}, lambda: (hmm.compute_batch(b), hmm.stats()))
    np.testing.assert_allclose(single[0], got, rtol=0, atol=1e-6)
    assert st["rerun_pairs"] <= single[1]["rerun_pairs"]
    # the pair pass is exact to the oracle where it ran: force it everywhere
    hmm.set_option("fp32_threshold", 3e38)
    try:
        assert_parity(hmm.compute_batch(b), want, tol=1e-9)
    finally:
        hmm.set_option("fp32_threshold", 1e-28)


def test_fp64_pair_pass_with_ragged_reads_and_haplotypes(hmm, oracle):
    # every pair through the pair pass: reads of 1..300 bases share waves, haplotypes of 20..700 bases
    b = synth.make_batch(70, 9, 300, 360, seed=41, len_jitter=299, hap_len_jitter=340, gop_model="random")
    want = ref(oracle, b)
    hmm.set_option("fp32_threshold", 3e38)
    try:
        for k in (0, 5, 16):
            got = with_options(hmm, {"force_rows_per_lane": k}, lambda: hmm.compute_batch(b))
            assert hmm.stats()["rerun_pairs"] == b.num_reads * b.num_haps
            assert_parity(got, want, tol=1e-9)
    finally:
        hmm.set_option("fp32_threshold", 1e-28)


def test_runs_across_regions_with_n_in_haplotypes(hmm, oracle):
    regions = synth.make_regions(12, seed=5, coverage=10, read_len=120)
    # an 'N' in one haplotype switches every launch of the call to the five-letter table
    hb = regions[3].hap_bases.copy()
    hb[7] = ord("N")
    r3 = regions[3]
    regions[3] = Batch(r3.read_bases, r3.read_quals, r3.ins_gop, r3.del_gop, r3.gcp, r3.read_offsets, hb, r3.hap_offsets)
    outs = hmm.compute_regions(regions)
    for b, o in zip(regions, outs):
        assert_parity(o, ref(oracle, b))


def test_many_region_call_in_overlapped_parts(hmm, oracle):
    regions = synth.make_regions(40, seed=21, coverage=6, read_len=90)
    regions.insert(7, pack([], ["ACGT"]))
    whole = with_options_pipeline(hmm, 0, lambda: hmm.compute_regions(regions))
    for pairs in (1, 300, 5000):  # one region per part; a few regions per part; two or three parts
        parts = with_options_pipeline(hmm, pairs, lambda: hmm.compute_regions(regions))
        for a, b in zip(whole, parts):
            np.testing.assert_array_equal(a, b)
    st = with_options_pipeline(hmm, 300, lambda: (hmm.compute_regions(regions), hmm.stats())[1])
    assert st["pairs"] == sum(b.num_reads * b.num_haps for b in regions)
    assert st["cells"] == sum(b.cells for b in regions)
    for b, o in zip(regions[:6], whole[:6]):
        assert_parity(o, ref(oracle, b))


def test_error_in_a_later_part_of_an_overlapped_call(hmm):
    regions = synth.make_regions(12, seed=22, coverage=5, read_len=80)
    bad = regions[8]
    q = bad.read_quals.copy()
    q[3] = 255
    regions[8] = Batch(bad.read_bases, q, bad.ins_gop, bad.del_gop, bad.gcp, bad.read_offsets, bad.hap_bases, bad.hap_offsets)
    with pytest.raises(pairhmm.ArrayIndexOutOfBoundsException):
        with_options_pipeline(hmm, 1, lambda: hmm.compute_regions(regions))
    # the handle is usable afterwards
    good = synth.make_regions(3, seed=23, coverage=5, read_len=80)
    assert all(np.isfinite(o).all() for o in hmm.compute_regions(good))


def test_one_large_region_is_cut_by_read_ranges(hmm, oracle):
    """A region with more pairs than a part holds goes through the overlapped path in pieces of consecutive
    reads (every pair is independent, PairHMM.java:217); zero-length reads and the per-read mapping-quality
    cap travel with their piece."""
    b = synth.make_regions(1, seed=41, coverage=40, read_len=120)[0]
    lens = np.diff(b.read_offsets)
    # reads 3 and 4 become zero-length: their bytes are dropped from the five arrays
    keep = np.ones(b.read_offsets[-1], dtype=bool)
    keep[b.read_offsets[3]:b.read_offsets[5]] = False
    lens[3] = lens[4] = 0
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    mapq = (20 + np.arange(b.num_reads) % 30).astype(np.uint8)
    cut = Batch(b.read_bases[keep], b.read_quals[keep], b.ins_gop[keep], b.del_gop[keep], b.gcp[keep], offsets, b.hap_bases, b.hap_offsets, mapq)
    capped = Batch(cut.read_bases, np.minimum(cut.read_quals, np.repeat(mapq, lens)), cut.ins_gop, cut.del_gop, cut.gcp, offsets, b.hap_bases, b.hap_offsets)
    want = ref(oracle, capped)
    for pairs in (0, 7, cut.num_reads * cut.num_haps // 3):
        got = with_options_pipeline(hmm, pairs, lambda: hmm.compute_batch(cut))
        assert np.isneginf(got[3:5]).all() and np.isneginf(want[3:5]).all()
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin)
        assert np.abs(got[fin] - want[fin]).max() <= TOL


def with_options_pipeline(hmm, pairs, fn):
    hmm.set_option("pipeline_pairs", pairs)
    try:
        return fn()
    finally:
        hmm.set_option("pipeline_pairs", 98304)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PAIRHMM_FUZZ_SEEDS", "24"))))
def test_random_shapes_and_options(hmm, oracle, seed):
    """Seeded fuzz: random region shapes, quality models, wildcard rates and engine options; every result
    within the bound of the oracle, and the fp64 passes within 1e-9."""
    rng = np.random.default_rng(7000 + seed)
    regions = []
    for _ in range(int(rng.integers(1, 6))):
        read_len = int(rng.choice([1, 7, 36, 76, 101, 151, 250, 400, 640, 900]))
        hap_len = max(read_len, int(rng.choice([8, 60, 200, 420, 900, 1100])))
        b = synth.make_batch(int(rng.integers(1, 70)), int(rng.integers(1, 40)), read_len, hap_len, seed=int(rng.integers(1 << 30)),
                             len_jitter=int(rng.integers(0, read_len)), hap_len_jitter=int(rng.integers(0, max(1, hap_len // 3))),
                             qual_model=str(rng.choice(["illumina", "uniform"])), gop_model=str(rng.choice(["const", "random"])),
                             indel_rate=float(rng.choice([0.0, 0.0, 0.03])), n_rate=float(rng.choice([0.0, 0.001, 0.05])))
        if rng.random() < 0.3:  # an 'N' in a haplotype
            hb = b.hap_bases.copy()
            hb[int(rng.integers(len(hb)))] = ord("N")
            b = Batch(b.read_bases, b.read_quals, b.ins_gop, b.del_gop, b.gcp, b.read_offsets, hb, b.hap_offsets)
        regions.append(b)
    extra = np.random.default_rng(9000 + seed)  # draws of its own: the shapes above stay what they were per seed
    if extra.random() < 0.3:  # a read for the long-read pass in one of the regions, tiled from the region's first haplotype
        k = int(extra.integers(len(regions)))
        b = regions[k]
        n = int(extra.integers(1023, 1700))
        hap = b.hap_bases[int(b.hap_offsets[0]):int(b.hap_offsets[1])]
        read = np.resize(hap, n).copy()
        read[::53] = ord("A")
        cat = lambda old, new: np.concatenate([old, new.astype(old.dtype)])
        regions[k] = Batch(cat(b.read_bases, read), cat(b.read_quals, extra.integers(10, 41, n)), cat(b.ins_gop, extra.integers(30, 50, n)),
                           cat(b.del_gop, extra.integers(30, 50, n)), cat(b.gcp, np.full(n, 10)),
                           cat(b.read_offsets, np.array([int(b.read_offsets[-1]) + n])), b.hap_bases, b.hap_offsets)
    grouped = [int(x) for b in regions for x in np.diff(b.read_offsets) if x <= 1022]
    opts = {"stream_mode": int(rng.choice([1, 1, 0, -1])), "max_stream_haps": int(rng.choice([1, 3, 16])),
            "emit_mode": int(rng.choice([-1, -1, 0]))}
    opts["stream_run_divisor"] = int(rng.choice([2, 2, 1, 7]))
    if rng.random() < 0.3:  # rows per lane chosen per wave-group as in a large call, from every value or from a few
        opts["per_group_factor"] = 0
        opts["group_rows_allowed"] = int(rng.choice([0, (1 << 9) | (1 << 10), 1 << 16, 1 << 3, (1 << 5) | (1 << 12)]))
    elif rng.random() < 0.4:
        longest = max(grouped)
        ks = [k for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16) if 64 * k >= longest + 2]
        opts["force_rows_per_lane"] = int(rng.choice(ks))
    hmm.set_option("pipeline_pairs", int(rng.choice([0, 200, 98304])))
    try:
        got = with_options(hmm, opts, lambda: hmm.compute_regions(regions))
        hmm.set_option("fp32_threshold", 3e38)
        got64 = with_options(hmm, opts, lambda: hmm.compute_regions(regions))
    finally:
        hmm.set_option("pipeline_pairs", 98304)
        hmm.set_option("fp32_threshold", 1e-28)
    for b, g, g64 in zip(regions, got, got64):
        want = ref(oracle, b)
        assert_parity(g, want)
        # below 10^-614.7 the sum times 2^1020 is a subnormal double: the reference's own arithmetic keeps fewer and
        # fewer bits there, and the order of its additions shows
        assert_parity(g64, want)
        normal = want > -614.0
        if normal.any():
            assert np.abs(g64[normal] - want[normal]).max() <= 1e-9


@pytest.mark.parametrize("combine", [False, True])
def test_concurrent_handles_one_per_thread(oracle, combine):
    """The reference makes one PairHMM per map thread (nanoScheduler/NanoScheduler.java:112, pairhmm/PairHMM.java:297):
    handles used from different threads at the same time must not disturb one another, and a failing call on
    one thread must leave the other threads' results and error text alone.  With the handle option
    combine_calls the region-sized calls of all threads share device passes: a bad batch in a pass fails its
    own caller only."""
    import threading

    regions = synth.make_regions(24, seed=77, coverage=8, read_len=100)
    want = [ref(oracle, b, threads=2) for b in regions]
    bad = pack([("ACGT", [30] * 4, [45, 45, 200, 45], [45] * 4, [10] * 4)], ["ACGTA"])
    failures = []

    def work(tid):
        h = pairhmm.HipLoglessPairHMM(0)
        if combine:
            h.set_option("combine_calls", 1)
        try:
            mine = list(range(tid, len(regions), 4))
            for rep in range(3):
                if tid == 0:  # this thread keeps provoking the reference's IllegalArgumentException
                    try:
                        h.compute_batch(bad)
                        failures.append("the bad batch was accepted")
                    except pairhmm.IllegalArgumentException:
                        pass
                got = h.compute_regions([regions[i] for i in mine]) if rep % 2 == 0 else [h.compute_batch(regions[i]) for i in mine]
                for i, g in zip(mine, got):
                    fin = np.isfinite(want[i])
                    if not (np.array_equal(np.isfinite(g), fin) and np.abs(g[fin] - want[i][fin]).max() <= TOL):
                        failures.append(f"thread {tid} region {i} rep {rep}")
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            failures.append(f"thread {tid}: {e!r}")
        finally:
            h.close()

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures


def test_region_sized_call_on_a_default_handle(oracle):
    """Left to itself (stream_mode -1) the engine gives a region-sized call to the one-haplotype pass and lets
    that kernel expand the reads and check the qualities itself: parity, the -infinity of zero-length reads,
    missing optional arrays, the mapping-quality cap with squashing, the reference's two input exceptions and
    a pair that needs the fp64 rerun, all on that path."""
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        b = synth.make_regions(1, seed=91, coverage=12, read_len=110)[0]
        assert_parity(h.compute_batch(b), ref(oracle, b))
        # zero-length read next to ordinary ones, no insertion / deletion / GCP arrays
        z = pack([("", [], None, None, None), ("ACGTACGT", [30] * 8, None, None, None), ("TTGCA", [20] * 5, None, None, None)], ["ACGTACGTAC", "TTGCATT"])
        got = h.compute_batch(z)
        zd = pack([("", [], [], [], []), ("ACGTACGT", [30] * 8, [45] * 8, [45] * 8, [10] * 8), ("TTGCA", [20] * 5, [45] * 5, [45] * 5, [10] * 5)], ["ACGTACGTAC", "TTGCATT"])
        want = ref(oracle, zd)
        assert np.isneginf(got[0]).all() and np.abs(got[1:] - want[1:]).max() <= TOL
        # mapping-quality cap and squashing
        m = synth.make_batch(30, 4, 90, 250, seed=92, mean_qual=22.0)
        mapq = (np.arange(m.num_reads) % 4 * 15 + 5).astype(np.uint8)
        quals = m.read_quals.copy()
        for r in range(m.num_reads):
            lo, hi = m.read_offsets[r], m.read_offsets[r + 1]
            quals[lo:hi] = oracle.modify_read_qualities(quals[lo:hi], mapq[r], True)
        h.set_option("squash_base_quals", 1)
        got = h.compute_batch(Batch(m.read_bases, m.read_quals, m.ins_gop, m.del_gop, m.gcp, m.read_offsets, m.hap_bases, m.hap_offsets, mapq))
        h.set_option("squash_base_quals", 0)
        assert_parity(got, ref(oracle, Batch(m.read_bases, quals, m.ins_gop, m.del_gop, m.gcp, m.read_offsets, m.hap_bases, m.hap_offsets)))
        # the two input exceptions, then a good call on the same handle
        ok = ("ACGT", [30] * 4, [45] * 4, [45] * 4, [10] * 4)
        with pytest.raises(pairhmm.IllegalArgumentException):
            h.compute_batch(pack([("ACGT", [30] * 4, [45, 200, 45, 45], [45] * 4, [10] * 4)], ["ACGTA"]))
        with pytest.raises(pairhmm.ArrayIndexOutOfBoundsException):
            h.compute_batch(pack([("ACGT", [30, 255, 30, 30], [45] * 4, [45] * 4, [10] * 4)], ["ACGTA"]))
        assert np.isfinite(h.compute_batch(pack([ok], ["ACGTA"]))).all()
        # a read far from its haplotype at high quality: fp32 gives up, the rerun list carries it
        far = synth.config_c5(num_reads=96, num_haps=12)
        got = h.compute_batch(far)
        assert h.stats()["rerun_tasks"] > 0
        assert_parity(got, ref(oracle, far))
    finally:
        h.close()


def test_small_results_come_home_the_same_through_flag_event_and_copy():
    """A region-sized result reaches the caller through one block that raises a flag in pinned host memory, through a
    copy kernel and an event, or through the runtime's copy: the same doubles every way, call after call on one handle
    (the flag's sequence number moves on), results of different sizes in turn, an input error in between."""
    regions = synth.make_regions(6, seed=77, coverage=25, read_len=140) + [synth.config_c1()]
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        got = {}
        for name, opts in (("flag", {}), ("event", {"signal_download_bytes": 0}), ("copy", {"signal_download_bytes": 0, "kernel_download_bytes": 0})):
            h.set_option("signal_download_bytes", opts.get("signal_download_bytes", 65536))
            h.set_option("kernel_download_bytes", opts.get("kernel_download_bytes", 524288))
            got[name] = [h.compute_batch(b).copy() for _ in range(3) for b in regions][-len(regions):]
            with pytest.raises(pairhmm.ArrayIndexOutOfBoundsException):
                h.compute_batch(pack([("ACGT", [30, 255, 30, 30], [45] * 4, [45] * 4, [10] * 4)], ["ACGTA"]))
            again = h.compute_batch(regions[0])
            assert np.array_equal(again, got[name][0])
        for name in ("event", "copy"):
            for a, b in zip(got["flag"], got[name]):
                assert np.array_equal(a, b), name
    finally:
        h.close()


@pytest.mark.parametrize("num_regions", [10, 25, 40, 70, 150])
def test_mid_sized_calls_on_a_default_handle(oracle, num_regions):
    """Calls of some thousand to a hundred thousand pairs on a handle with default options: one rows-per-lane
    value for the whole call, from about 12 000 pairs on two overlapped halves, from about 30 000 on four growing
    parts.  Regions with reads of different
    lengths, an empty region and a zero-length read among them; same results as with one region per call."""
    regions = synth.make_regions(num_regions, seed=300 + num_regions, coverage=30, read_len=151)
    regions += synth.make_regions(3, seed=17, coverage=10, read_len=60)
    regions.insert(2, pack([], ["ACGTACGT"]))
    regions.append(pack([("", [], [], [], []), ("ACGTTGCA", [30] * 8, [45] * 8, [45] * 8, [10] * 8)], ["ACGTTGCAAC", "ACGTGCAAC"]))
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        got = h.compute_regions(regions)
        st = h.stats()
        assert st["pairs"] == sum(b.num_reads * b.num_haps for b in regions)
        for k in list(range(0, len(regions), 9)) + [len(regions) - 1]:
            if regions[k].num_reads:
                assert_parity(got[k], ref(oracle, regions[k], threads=2))
        for k, b in enumerate(regions):
            one = h.compute_batch(b) if b.num_reads else np.empty((0, b.num_haps))
            fin = np.isfinite(one)
            assert np.array_equal(np.isfinite(got[k]), fin)
            assert np.abs(got[k][fin] - one[fin]).max(initial=0.0) <= 2e-6
        assert np.isneginf(got[-1][0]).all()
    finally:
        h.close()
