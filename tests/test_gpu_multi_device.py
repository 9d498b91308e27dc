"""The multi-device handle (gatk_pairhmm_create_multi) and the two region-stream workloads at their real sizes:
a HaplotypeCaller-shaped stream of about 2000 active regions (30x, 151-base reads) through one compute_regions
call, and the whole-genome-shaped 10 000 regions sharded over devices with the results written straight into the
caller's arrays.  The GPU box has one GPU, so the multi-device handle is given device 0 twice: two engines, two
host threads, the same sharding code as with eight distinct devices.

Tolerance as in test_gpu_parity: 1e-3 in log10 against the fp64 oracle.  Two engine runs of the same region may
differ by the fp32 error (about 2e-6) where one of them recomputed a read in fp64 because a wave-mate underflowed.
"""
import numpy as np
import pytest

from gatk_amd import pairhmm, sharding, synth
from gatk_amd.batch import Batch
from gatk_amd.pairhmm import HipLoglessPairHMM

pytestmark = pytest.mark.gpu

TOL = 1e-3
SAME_ENGINE_TOL = 1e-5


@pytest.fixture(scope="session")
def hc_regions():
    """2000 HaplotypeCaller-shaped regions (about 130 k reads, 1 M pairs, 56 G cells), generated once per session."""
    return synth.make_regions(2000, seed=synth.DEFAULT_SEED)


@pytest.fixture(scope="module")
def multi():
    h = HipLoglessPairHMM([0, 0])
    yield h
    h.close()


def ref(oracle, b):
    rc, out = oracle.compute_likelihoods(b, oracle.LOGLESS, fast_threads=8)
    assert rc == 0
    return out


def check_against_oracle(oracle, regions, outs, sample):
    for k in sample:
        want, got = ref(oracle, regions[k]), outs[k]
        assert got.shape == want.shape
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin)
        assert float(np.abs(got[fin] - want[fin]).max()) <= TOL, f"region {k}"


def check_valid(regions, outs):
    assert len(outs) == len(regions)
    for b, o in zip(regions, outs):
        assert o.shape == (b.num_reads, b.num_haps)
        assert not np.isnan(o).any() and (o <= 0).all()


def assert_same(a, b, tol=SAME_ENGINE_TOL):
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(np.isfinite(x), np.isfinite(y))
        fin = np.isfinite(x)
        if fin.any():
            assert float(np.abs(x[fin] - y[fin]).max()) <= tol


def test_haplotypecaller_stream_at_size(hmm, oracle, hc_regions):
    outs = hmm.compute_regions(hc_regions)
    st = hmm.stats()
    assert st["pairs"] == sum(b.num_reads * b.num_haps for b in hc_regions)
    assert st["cells"] == sum(b.cells for b in hc_regions)
    check_valid(hc_regions, outs)
    rng = np.random.default_rng(5)
    check_against_oracle(oracle, hc_regions, outs, rng.choice(len(hc_regions), size=50, replace=False))


def test_multi_device_handle_matches_single_device(hmm, multi, oracle, hc_regions):
    # 300 regions plus one batch large enough to be cut by read ranges between the engines, and two empty regions
    big = synth.make_batch(1536, 32, 151, 600, seed=21, hap_len_jitter=0)
    empty = synth.make_batch(0, 3, 10, 30, seed=1)
    regions = hc_regions[:150] + [empty, big] + hc_regions[150:300] + [empty]
    single = hmm.compute_regions(regions)
    got = multi.compute_regions(regions)
    st = multi.stats()
    assert st["pairs"] == sum(b.num_reads * b.num_haps for b in regions) and st["cells"] == sum(b.cells for b in regions)
    check_valid(regions, got)
    assert_same(got, single)
    check_against_oracle(oracle, regions, got, [0, 77, 151, 152, 299])
    # one computeLikelihoods call on the multi-device handle: the large batch is split, the small one goes to one engine
    assert_same([multi.compute_batch(big)], [single[151]])
    small = hc_regions[7]
    for _ in range(3):  # the engines take small calls in turn
        assert_same([multi.compute_batch(small)], [single[7]])
    assert np.array_equal(multi.getLikelihoodArray(), multi.compute_batch(small).reshape(-1))


def test_multi_device_handle_options_errors_and_single_pair(multi, oracle, hc_regions):
    regions = hc_regions[300:420]
    multi.set_option("force_fp64", 1)
    try:
        got = multi.compute_regions(regions)
    finally:
        multi.set_option("force_fp64", 0)
    for k in (0, 60, 119):
        want = ref(oracle, regions[k])
        fin = np.isfinite(want)
        assert float(np.abs(got[k][fin] - want[fin]).max()) <= 1e-9
    # the reference's exception reaches the caller whichever engine met it
    bad = synth.make_batch(40, 6, 100, 300, seed=3)
    bad.gcp = bad.gcp.copy(); bad.gcp[11] = 200
    # ... under the caller's own region and read index, although the engine that met it held a share of the regions
    with pytest.raises(pairhmm.IllegalArgumentException, match=r"\(region 60, read 0\)"):
        multi.compute_regions(regions[:60] + [bad] + regions[60:])
    with pytest.raises(pairhmm.IllegalArgumentException, match="region 3"):  # host-side checks name the caller's region index
        worse = synth.make_batch(4, 2, 20, 50, seed=3)
        worse.hap_offsets = worse.hap_offsets.copy(); worse.hap_offsets[1] = worse.hap_offsets[0]
        multi.compute_regions(regions[:3] + [worse])
    # and the handle is usable afterwards
    assert_same(multi.compute_regions(regions[:5]), [ref(oracle, b) for b in regions[:5]], tol=TOL)
    # initialize and the single-pair call go to every engine
    multi.initialize(200, 700)
    b = regions[0]
    lo, hi = int(b.read_offsets[0]), int(b.read_offsets[1])
    hap = bytes(b.hap_bases[int(b.hap_offsets[0]):int(b.hap_offsets[1])])
    lk = multi.computeReadLikelihoodGivenHaplotypeLog10(hap, bytes(b.read_bases[lo:hi]), b.read_quals[lo:hi], b.ins_gop[lo:hi],
                                                        b.del_gop[lo:hi], b.gcp[lo:hi])
    assert abs(lk - ref(oracle, b)[0, 0]) <= TOL


def test_a_malformed_region_late_in_a_large_call_is_named(hmm, oracle, hc_regions):
    """A call of many regions is checked part by part while the parts before are already on the device: a region
    whose offsets decrease, four hundred regions in, is still reported under its own index, as is one with a
    zero-length haplotype in the last part; nothing is left in flight, and the next call is right."""
    worse = synth.make_batch(4, 2, 20, 50, seed=3)
    worse.read_offsets = worse.read_offsets.copy(); worse.read_offsets[2] = worse.read_offsets[1] - 1
    with pytest.raises(pairhmm.IllegalArgumentException, match=r"read offsets must not decrease \(region 400\)"):
        hmm.compute_regions(hc_regions[:400] + [worse] + hc_regions[400:500])
    empty_hap = synth.make_batch(4, 2, 20, 50, seed=4)
    empty_hap.hap_offsets = empty_hap.hap_offsets.copy(); empty_hap.hap_offsets[1] = empty_hap.hap_offsets[0]
    with pytest.raises(pairhmm.IllegalArgumentException, match=r"zero-length haplotype.*\(region 1199\)"):
        hmm.compute_regions(hc_regions[:1199] + [empty_hap])
    got = hmm.compute_regions(hc_regions[:80])
    check_against_oracle(oracle, hc_regions[:80], got, [0, 41, 79])


def test_reads_for_the_long_read_pass_on_the_multi_device_handle(hmm, multi, oracle, hc_regions):
    """Reads of more than 1022 bases in regions that go whole to one engine, and in one large region that is cut by
    read ranges between the engines: each engine runs its own long-read launch on its share."""
    rng = np.random.default_rng(31)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, 1500)]
    def near(start, n):
        read = hap[start:start + n].copy()
        read[::41] = acgt[(np.searchsorted(acgt, read[::41]) + 1) % 4]
        return read
    lens = [int(x) for x in rng.choice([100, 151, 1023, 1200, 1400], size=400, p=[0.45, 0.45, 0.04, 0.03, 0.03])]
    lens[0], lens[-1] = 1300, 1250  # a long read at either end of the read ranges
    reads = [near(int(rng.integers(0, 90)), n) for n in lens]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    total = int(offsets[-1])
    haps = [hap, hap[:1450], hap[10:]] * 8
    hap_offsets = np.concatenate([[0], np.cumsum([len(x) for x in haps])]).astype(np.int32)
    big = Batch(np.concatenate(reads), rng.integers(10, 40, total).astype(np.uint8), np.full(total, 45, np.uint8),
                      np.full(total, 45, np.uint8), np.full(total, 10, np.uint8), offsets, np.concatenate(haps), hap_offsets)
    small = Batch(big.read_bases[:int(offsets[3])], big.read_quals[:int(offsets[3])], big.ins_gop[:int(offsets[3])],
                        big.del_gop[:int(offsets[3])], big.gcp[:int(offsets[3])], offsets[:4].copy(), big.hap_bases[:int(hap_offsets[3])],
                        hap_offsets[:4].copy())
    regions = hc_regions[:20] + [small] + hc_regions[20:40] + [big]
    got = multi.compute_regions(regions)
    check_valid(regions, got)
    check_against_oracle(oracle, regions, got, [3, 20, 41])
    assert_same(got, hmm.compute_regions(regions))
    assert_same([multi.compute_batch(big)], [got[41]])


def test_whole_genome_shaped_10k_regions_sharded(hmm, multi, oracle, hc_regions):
    # 10 000 regions: five differently ordered passes over the 2000 generated ones
    rng = np.random.default_rng(9)
    index = np.concatenate([rng.permutation(len(hc_regions)) for _ in range(5)])
    regions = [hc_regions[i] for i in index]
    pairs = sum(b.num_reads * b.num_haps for b in regions)
    cells = sum(b.cells for b in regions)
    # through the process-per-GPU sharding helper (world 1: everything on this rank) ...
    sharded = sharding.compute_sharded(regions, hmm.compute_regions, rank=0, world=1)
    st = hmm.stats()
    assert st["pairs"] == pairs and st["cells"] == cells
    check_valid(regions, sharded)
    # ... and through the multi-device handle, which shards inside the C-ABI
    got = multi.compute_regions(regions)
    st = multi.stats()
    assert st["pairs"] == pairs and st["cells"] == cells
    check_valid(regions, got)
    assert_same(got, sharded)
    # order independence: the five copies of a region agree wherever they landed
    first = {}
    for pos, i in enumerate(index):
        if i in first:
            assert_same([got[pos]], [got[first[i]]])
        else:
            first[i] = pos
    sample = rng.choice(len(regions), size=50, replace=False)
    check_against_oracle(oracle, regions, got, sample)


def test_device_list_of_the_whole_node():
    h = HipLoglessPairHMM("all")  # every visible device
    try:
        b = synth.config_c1()
        assert h.compute_batch(b).shape == (128, 16)
    finally:
        h.close()
    with pytest.raises(pairhmm.DeviceError):
        HipLoglessPairHMM([0, 99])


@pytest.mark.parametrize("share", [0, 1])
def test_an_exception_inside_one_engines_share_comes_back_as_a_code(multi, hc_regions, share):
    """Nothing may cross the C boundary as a C++ exception, and a failing share must not pull the other engine's
    work from under it: the call returns an error code after both engines are done, and the handle works afterwards."""
    lib = pairhmm.load_library()
    regions = hc_regions[:120]
    lib.gatk_pairhmm_debug_fail_share(share)
    try:
        with pytest.raises(pairhmm.DeviceError, match="injected failure"):
            multi.compute_regions(regions)
    finally:
        lib.gatk_pairhmm_debug_fail_share(-1)
    outs = multi.compute_regions(regions)
    check_valid(regions, outs)


def test_distinct_devices_when_the_box_has_them(hmm, hc_regions):
    """With more than one GPU visible: a handle over every device gives what one device gives."""
    n = pairhmm.device_count()
    if n < 2:
        pytest.skip("one GPU visible: the multi-device path is only exercised with device 0 listed twice")
    regions = hc_regions[:400]
    h = HipLoglessPairHMM(list(range(n)))
    try:
        assert_same(hmm.compute_regions(regions), h.compute_regions(regions))
        tickets = [h.submit([b]) for b in regions[:2 * n]]
        assert_same([t.fetch()[0] for t in tickets], hmm.compute_regions(regions[:2 * n]))
    finally:
        h.close()


def test_an_error_in_a_region_cut_by_read_ranges_names_the_callers_read(multi, hmm):
    """A region large enough to be cut by read ranges, on one engine (parts of a cut call) and over several (one range
    per engine): a base quality of 255 deep inside it is reported as read 3000 of region 1, not as a read of a piece."""
    big = synth.make_batch(4096, 64, 151, 600, seed=21)
    big.read_quals = big.read_quals.copy()
    big.read_quals[int(big.read_offsets[3000]) + 5] = 255
    small = synth.make_batch(6, 3, 50, 120, seed=2)
    for engine in (hmm, multi):
        with pytest.raises(pairhmm.ArrayIndexOutOfBoundsException, match=r"\(region 1, read 3000\)"):
            engine.compute_regions([small, big])
        assert np.isfinite(engine.compute_regions([small])[0]).all()
