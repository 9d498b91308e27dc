"""submit / fetch and the region-level hook, through the C-ABI on a real MI355X, against the CPU oracle.

gatk_pairhmm_initialize_region is PairHMM.initialize(haplotypes, perSampleReadList, readMaxLength,
haplotypeMaxLength) (pairhmm/PairHMM.java:129-139); the computeLikelihoods calls behind it (PairHMM.java:190-236) must
return what they return without it, exceptions included.  Same tolerance as the parity tests: 1e-3 absolute on log10
likelihoods, -infinity exact.
"""
import threading

import numpy as np
import pytest

from gatk_amd import pairhmm, synth
from gatk_amd.batch import Batch, pack

pytestmark = pytest.mark.gpu

TOL = 1e-3


def ref(oracle, b, threads=4):
    rc, out = oracle.compute_likelihoods(b, oracle.LOGLESS, fast_threads=threads)
    assert rc == 0
    return out


def close_enough(got, want, tol=TOL):
    fin = np.isfinite(want)
    return got.shape == want.shape and np.array_equal(np.isfinite(got), fin) and not np.isnan(got).any() and \
        (not fin.any() or np.abs(got[fin] - want[fin]).max() <= tol)


def samples_of(region, num_samples):
    """The reads of one region dealt out to samples: every sample carries the region's haplotypes."""
    cuts = np.linspace(0, region.num_reads, num_samples + 1).astype(int)
    return [region.read_slice(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:])]


def test_tickets_fetched_in_any_order(oracle):
    regions = synth.make_regions(9, seed=5, coverage=14, read_len=120)
    want = [ref(oracle, b) for b in regions]
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        tickets = [h.submit([b]) for b in regions[:6]] + [h.submit(regions[6:])]  # six one-region requests and a three-region one
        got = {}
        for k in (3, 0, 6, 5, 1, 2, 4):
            got[k] = tickets[k].fetch()
        for k in range(6):
            assert close_enough(got[k][0], want[k]), k
        for j in range(3):
            assert close_enough(got[6][j], want[6 + j]), 6 + j
        with pytest.raises(pairhmm.IllegalStateException):
            tickets[0].fetch()
        lib = pairhmm.load_library()
        assert lib.gatk_pairhmm_fetch(h._h, 12345) == pairhmm.ILLEGAL_ARGUMENT
        st = h.stats()
        assert st["shared_passes"] >= 1 and st["shared_requests"] >= 7
        # the blocking call gives the same likelihoods (a pass over several regions may choose other rows per lane
        # than a call of one: the fp32 sums then differ in their last bits)
        for k in (0, 4):
            assert np.abs(h.compute_batch(regions[k]) - got[k][0]).max() <= 2e-6
    finally:
        h.close()


def test_submit_reports_shape_errors_and_fetch_reports_data_errors(oracle):
    good = synth.make_regions(2, seed=11, coverage=10, read_len=100)
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        with pytest.raises(pairhmm.IllegalArgumentException):  # zero-length haplotype: refused before anything is queued
            h.submit([pack([("ACGT", [30] * 4, None, None, None)], [""])])
        bad_gap = pack([("ACGT", [30] * 4, [45, 45, 200, 45], [45] * 4, [10] * 4)], ["ACGTA"])
        bad_qual = pack([("ACGT", [30, 255, 30, 30], [45] * 4, [45] * 4, [10] * 4)], ["ACGTA"])
        t_good, t_gap, t_qual, t_good2 = h.submit([good[0]]), h.submit([bad_gap]), h.submit([bad_qual]), h.submit([good[1]])
        with pytest.raises(pairhmm.IllegalArgumentException):
            t_gap.fetch()
        assert close_enough(t_good.fetch()[0], ref(oracle, good[0]))
        with pytest.raises(pairhmm.ArrayIndexOutOfBoundsException):
            t_qual.fetch()
        assert close_enough(t_good2.fetch()[0], ref(oracle, good[1]))
        # empty requests and regions without reads
        assert h.submit([]).fetch() == []
        e = h.submit([pack([], ["ACGT"]), good[0]]).fetch()
        assert e[0].shape == (0, 1) and close_enough(e[1], ref(oracle, good[0]))
    finally:
        h.close()


def test_requests_of_many_threads_share_passes(oracle):
    """Eight threads, each with its own handle and three regions in flight, one of them provoking the reference's
    IllegalArgumentException over and over: every good request gets its own result, the bad ones their exception, and
    the dispatcher's counters show passes that carried several requests."""
    regions = synth.make_regions(32, seed=21, coverage=9, read_len=100)
    want = [ref(oracle, b, threads=2) for b in regions]
    bad = pack([("ACGT", [30] * 4, [45] * 4, [45, 130, 45, 45], [10] * 4)], ["ACGTA"])
    failures, counts = [], {}

    def work(tid):
        h = pairhmm.HipLoglessPairHMM(0)
        try:
            mine = list(range(tid, len(regions), 8))
            for rep in range(6):
                tickets = [(i, h.submit([regions[i]])) for i in mine[:3]]
                t_bad = h.submit([bad]) if tid == 3 else None
                tickets += [(i, h.submit([regions[i]])) for i in mine[3:]]
                if t_bad is not None:
                    try:
                        t_bad.fetch()
                        failures.append("the bad request was accepted")
                    except pairhmm.IllegalArgumentException:
                        pass
                for i, t in tickets:
                    if not close_enough(t.fetch()[0], want[i]):
                        failures.append(f"thread {tid} region {i} rep {rep}")
            counts[tid] = h.stats()
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            failures.append(f"thread {tid}: {e!r}")
        finally:
            h.close()

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures
    st = counts[0]
    assert st["shared_passes"] > 0 and st["shared_requests"] >= st["shared_passes"]


def test_blocking_calls_with_combine_calls_go_through_the_dispatcher(oracle):
    """The handle option combine_calls: blocking computeLikelihoods calls of eight threads share the device's
    dispatcher; the counters prove the shared passes ran, and a failing caller fails alone."""
    regions = synth.make_regions(24, seed=31, coverage=8, read_len=100)
    want = [ref(oracle, b, threads=2) for b in regions]
    bad = pack([("ACGT", [30, 255, 30, 30], [45] * 4, [45] * 4, [10] * 4)], ["ACGTA"])
    probe = pairhmm.HipLoglessPairHMM(0)
    before = probe.stats()
    failures = []

    def work(tid):
        h = pairhmm.HipLoglessPairHMM(0)
        h.set_option("combine_calls", 1)
        try:
            for rep in range(4):
                if tid == 5:
                    try:
                        h.compute_batch(bad)
                        failures.append("the bad batch was accepted")
                    except pairhmm.ArrayIndexOutOfBoundsException:
                        pass
                for i in range(tid, len(regions), 8):
                    if not close_enough(h.compute_batch(regions[i]), want[i]):
                        failures.append(f"thread {tid} region {i} rep {rep}")
        except Exception as e:  # noqa: BLE001
            failures.append(f"thread {tid}: {e!r}")
        finally:
            h.close()

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    after = probe.stats()
    probe.close()
    assert not failures, failures
    assert after["shared_requests"] - before["shared_requests"] >= 8 * 4 * 3
    assert after["shared_passes"] > before["shared_passes"]


@pytest.mark.parametrize("num_samples", [1, 3, 8])
def test_region_sent_ahead_serves_every_sample(oracle, num_samples):
    region = synth.make_regions(1, seed=40 + num_samples, coverage=40, read_len=130)[0]
    samples = samples_of(region, num_samples)
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        ahead = [Batch(s.read_bases, s.read_quals, s.ins_gop, s.del_gop, None, s.read_offsets, s.hap_bases, s.hap_offsets) for s in samples]
        h.initialize_region_batches(ahead, int(region.read_lengths.max()), int(region.hap_lengths.max()))
        for s in samples:  # computeLikelihoods brings the GCP array: all 10, the default the region was sent with
            s.gcp = np.full_like(s.read_bases, 10)
            got = h.compute_batch(s)
            assert close_enough(got, ref(oracle, s))
            assert np.array_equal(h.getLikelihoodArray(), got.reshape(-1))
        st = h.stats()
        assert st["region_hits"] == num_samples and st["region_misses"] == 0
        h.finalizeRegion()
        # after finalizeRegion nothing is served from the region any more
        assert close_enough(h.compute_batch(samples[0]), ref(oracle, samples[0]))
        assert h.stats()["region_hits"] == num_samples
    finally:
        h.close()


def test_region_sent_ahead_with_other_gcp_samples_out_of_order_and_strangers(oracle):
    region = synth.make_regions(1, seed=77, coverage=30, read_len=110)[0]
    samples = samples_of(region, 4)
    other = synth.make_regions(1, seed=78, coverage=10, read_len=90)[0]
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        ahead = [Batch(s.read_bases, s.read_quals, None, None, None, s.read_offsets, s.hap_bases, s.hap_offsets) for s in samples]
        h.initialize_region_batches(ahead, 200, 1000)
        flat45 = [Batch(s.read_bases, s.read_quals, np.full_like(s.read_bases, 45), np.full_like(s.read_bases, 45), np.full_like(s.read_bases, 10),
                        s.read_offsets, s.hap_bases, s.hap_offsets) for s in samples]
        # out of order, and explicit default arrays against the ones left out: still the same computation
        for k in (2, 0, 3):
            assert close_enough(h.compute_batch(flat45[k]), ref(oracle, flat45[k]))
        assert h.stats()["region_hits"] == 3
        # a GCP that is not the default: scored the ordinary way, with that GCP
        g = flat45[1]
        g.gcp = np.full_like(g.read_bases, 12)
        assert close_enough(h.compute_batch(g), ref(oracle, g))
        # insertion qualities that differ from the flat ones sent ahead: a miss, scored with its own
        own = samples[1]
        own.ins_gop = np.full_like(own.read_bases, 40)
        assert close_enough(h.compute_batch(own), ref(oracle, own))
        # a batch that was never sent ahead
        assert close_enough(h.compute_batch(other), ref(oracle, other))
        st = h.stats()
        assert st["region_hits"] == 3 and st["region_misses"] == 3
        # a new region without finalizeRegion in between replaces the old one
        h.initialize_region_batches([other], 200, 1000)
        assert close_enough(h.compute_batch(other), ref(oracle, other))
        assert h.stats()["region_hits"] == 4
        h.compute_batch(flat45[0])
        assert h.stats()["region_misses"] == 4
    finally:
        h.close()


def test_region_sent_ahead_with_an_input_error_in_one_sample(oracle):
    """The reference raises from the computeLikelihoods call whose reads hold the bad byte, not from initialize."""
    region = synth.make_regions(1, seed=91, coverage=20, read_len=100)[0]
    samples = samples_of(region, 3)
    bad = samples[1]
    bad.ins_gop = bad.ins_gop.copy()
    bad.ins_gop[7] = 200
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        h.initialize_region_batches(samples, 200, 1000)
        assert close_enough(h.compute_batch(samples[0]), ref(oracle, samples[0]))
        with pytest.raises(pairhmm.IllegalArgumentException):
            h.compute_batch(bad)
        assert close_enough(h.compute_batch(samples[2]), ref(oracle, samples[2]))
        h.finalizeRegion()
        with pytest.raises(pairhmm.IllegalArgumentException):  # PairHMM.java:105-106 through initialize(int, int)
            h.initialize_region_batches(samples[:1], 0, 10)
    finally:
        h.close()


def test_finalize_or_close_without_asking_for_the_results(oracle):
    regions = synth.make_regions(6, seed=3, coverage=10, read_len=100)
    h = pairhmm.HipLoglessPairHMM(0)
    h.initialize_region_batches(regions[:3], 200, 1000)
    h.finalizeRegion()  # nobody asked
    h.initialize_region_batches(regions[3:], 200, 1000)
    keep = [h.submit([b]) for b in regions]  # never fetched
    assert close_enough(h.compute_batch(regions[4]), ref(oracle, regions[4]))
    h.close()
    del keep
    h2 = pairhmm.HipLoglessPairHMM(0)  # the device and its dispatcher are fine afterwards
    try:
        assert close_enough(h2.submit([regions[0]]).fetch()[0], ref(oracle, regions[0]))
    finally:
        h2.close()


def test_large_and_non_default_requests_are_scored_at_submit(oracle):
    big = synth.make_batch(700, 160, 100, 300, seed=9)  # 112 000 pairs: more than a shared pass takes
    small = synth.make_regions(1, seed=8, coverage=10, read_len=100)[0]
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        before = h.stats()["shared_requests"]
        got = h.submit([big]).fetch()[0]
        assert h.stats()["shared_requests"] == before
        want = ref(oracle, big, threads=8)
        assert close_enough(got, want)
        h.set_option("force_fp64", 1)  # the shared engines do not run with this: the handle's own engine does
        got = h.submit([small]).fetch()[0]
        assert h.stats()["shared_requests"] == before
        assert close_enough(got, ref(oracle, small), tol=1e-9)
    finally:
        h.close()


def test_multi_device_handle_takes_tickets(oracle):
    regions = synth.make_regions(8, seed=14, coverage=10, read_len=100)
    h = pairhmm.HipLoglessPairHMM([0, 0])
    try:
        tickets = [h.submit([b]) for b in regions]
        for b, t in zip(regions, tickets):
            assert close_enough(t.fetch()[0], ref(oracle, b))
        h.initialize_region_batches(regions[:2], 200, 1000)
        assert close_enough(h.compute_batch(regions[1]), ref(oracle, regions[1]))
        assert h.stats()["region_hits"] == 1
    finally:
        h.close()


def test_region_sent_ahead_raw_is_found_by_the_reads_the_caller_processed_itself(oracle):
    """The upstream likelihood engine caps a read's base qualities by its mapping quality and squashes the ones under
    BASE_QUALITY_SCORE_THRESHOLD (PairHMM.java:54) between initialize(haplotypes, perSampleReadList, ...) and
    computeLikelihoods.  A region sent ahead raw, with the mapping qualities and the squashing option, is the same
    computation as the processed reads that are asked for: served from the region, equal to scoring them directly."""
    region = synth.make_batch(40, 5, 100, 260, seed=17, mean_qual=21.0)
    mapq = (np.arange(region.num_reads) % 5 * 12 + 8).astype(np.uint8)
    processed = region.read_quals.copy()
    for r in range(region.num_reads):
        lo, hi = region.read_offsets[r], region.read_offsets[r + 1]
        processed[lo:hi] = oracle.modify_read_qualities(region.read_quals[lo:hi], mapq[r], True)
    assert (processed != region.read_quals).any()
    raw = Batch(region.read_bases, region.read_quals, region.ins_gop, region.del_gop, None, region.read_offsets, region.hap_bases,
                region.hap_offsets, mapq)
    asked = Batch(region.read_bases, processed, region.ins_gop, region.del_gop, region.gcp, region.read_offsets, region.hap_bases, region.hap_offsets)
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        h.set_option("squash_base_quals", 1)
        h.initialize_region_batches([raw], 200, 1000)
        got = h.compute_batch(asked)
        assert h.stats()["region_hits"] == 1
        assert close_enough(got, ref(oracle, asked))
        # a quality that differs after the treatment as well: not the same computation
        other = processed.copy()
        other[3] = 41
        h.compute_batch(Batch(region.read_bases, other, region.ins_gop, region.del_gop, region.gcp, region.read_offsets, region.hap_bases, region.hap_offsets))
        assert h.stats()["region_misses"] == 1
        h.finalizeRegion()
        # without the option the raw bytes are what counts: the processed reads are strangers
        h.set_option("squash_base_quals", 0)
        h.initialize_region_batches([Batch(region.read_bases, region.read_quals, region.ins_gop, region.del_gop, None, region.read_offsets,
                                           region.hap_bases, region.hap_offsets)], 200, 1000)
        assert close_enough(h.compute_batch(asked), ref(oracle, asked))
        assert h.stats()["region_hits"] == 1 and h.stats()["region_misses"] == 2
    finally:
        h.close()
