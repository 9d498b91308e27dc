"""The entry point that allocates nothing (gatk_pairhmm_compute_likelihoods_noalloc) against
gatk_pairhmm_compute_likelihoods on the same inputs: two calls in a row, the second smaller, so that what the
first call left in the per-device scratch array must not leak into the second."""
import numpy as np
import pytest

from gatk_amd import synth
from gatk_amd.pairhmm import CBatch, HipLoglessPairHMM

pytestmark = pytest.mark.gpu

STATUS_ILLEGAL_ARGUMENT, STATUS_INDEX, STATUS_SHAPE = 1, 2, 4


def device_batch(torch, b):
    """The batch's arrays as torch tensors on the GPU (plumbing only) and a CBatch of their device addresses."""
    dev = torch.device("cuda", 0)
    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    t = {k: up(getattr(b, k), np.uint8) for k in ("read_bases", "read_quals", "ins_gop", "del_gop", "gcp", "hap_bases")}
    t["read_offsets"] = up(b.read_offsets, np.int32)
    t["hap_offsets"] = up(b.hap_offsets, np.int32)
    ptr = lambda k: t[k].data_ptr() if t[k].numel() else None
    cb = CBatch(b.num_reads, b.num_haps, ptr("read_bases"), ptr("read_quals"), ptr("ins_gop"), ptr("del_gop"), ptr("gcp"),
                ptr("read_offsets"), ptr("hap_bases"), ptr("hap_offsets"), None)
    return t, cb


def run_noalloc(torch, hmm, b, read_max=None, hap_max=None, fill=np.nan):
    keep, cb = device_batch(torch, b)
    out = torch.full((b.num_reads, b.num_haps), fill, dtype=torch.float64, device="cuda:0")
    status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.current_stream()
    hmm.compute_device_batch_noalloc(cb, int(read_max or b.read_lengths.max()), int(hap_max or b.hap_lengths.max()),
                                     out.data_ptr(), status.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    del keep
    return out.cpu().numpy(), int(status.item())


def test_two_calls_in_a_row_match_the_allocating_entry_point():
    torch = pytest.importorskip("torch")
    hmm = HipLoglessPairHMM(0)
    try:
        # first call: long, indel-rich reads (some pairs underflow fp32 and go through the fp64 list), an N-rich
        # haplotype set, a zero-length read in the middle
        big = synth.make_batch(96, 12, 250, 1000, seed=11, indel_rate=0.04, mean_qual=20.0, gop_model="random", n_rate=0.01)
        offs = np.asarray(big.read_offsets).copy()
        cut = int(offs[41] - offs[40])   # make read 40 empty by removing its bytes
        lo, hi = int(offs[40]), int(offs[41])
        for name in ("read_bases", "read_quals", "ins_gop", "del_gop", "gcp"):
            setattr(big, name, np.concatenate([getattr(big, name)[:lo], getattr(big, name)[hi:]]))
        offs[41:] -= cut
        big.read_offsets = offs.astype(np.int32)
        big.hap_bases = big.hap_bases.copy()
        big.hap_bases[5::97] = ord("N")
        # second call: smaller in every dimension, so stale descriptors, haplotypes and list entries lie behind it
        small = synth.make_batch(7, 3, 60, 140, seed=12, len_jitter=30)
        for b in (big, small):
            want = hmm.compute_batch(b)
            rerun = hmm.stats()["rerun_pairs"]
            got, status = run_noalloc(torch, hmm, b)
            assert status == 0
            assert got.shape == want.shape and not np.isnan(got).any()
            assert np.array_equal(np.isneginf(got), np.isneginf(want))
            finite = np.isfinite(want)
            err = float(np.abs(got[finite] - want[finite]).max())
            # Bar: 1e-3 in log10 (fp32 path).  Both entry points redo in fp64 every read that shares a wave with an
            # underflowing pair, and they pack reads into waves differently, so on a batch with underflowing pairs a
            # few results are the fp64 value on one side and the fp32 value (within 2e-6 of it) on the other;
            # without such pairs the two run the same arithmetic and must agree to the bit.
            assert err <= (1e-5 if rerun > 0 else 0.0), f"allocation-free entry point differs from compute_likelihoods by {err}"
            if b is big:
                assert rerun > 0 and np.isneginf(got[40]).all()
            else:
                assert rerun == 0
        # a declared maximum larger than any read packs differently and must not change a value
        again, status = run_noalloc(torch, hmm, small, read_max=151, hap_max=600)
        assert status == 0 and np.array_equal(again, got)
    finally:
        hmm.close()


@pytest.mark.parametrize("read_len,jitter", [(640, 60), (1000, 300)])
def test_long_reads(read_len, jitter):
    """11 to 16 rows per lane (the variants whose fp64 pass does not fit the register file) through the
    allocation-free entry point: same values as compute_likelihoods, and a smaller call behind it."""
    torch = pytest.importorskip("torch")
    hmm = HipLoglessPairHMM(0)
    try:
        long_reads = synth.make_batch(20, 5, read_len, 1100, seed=read_len + 1, len_jitter=jitter, gop_model="random")
        small = synth.make_batch(5, 2, 90, 300, seed=3)
        for b in (long_reads, small):
            want = hmm.compute_batch(b)
            rerun = hmm.stats()["rerun_pairs"]
            got, status = run_noalloc(torch, hmm, b)
            assert status == 0 and np.isfinite(got).all()
            assert float(np.abs(got - want).max()) <= (1e-5 if rerun > 0 else 0.0)
    finally:
        hmm.close()


def test_device_side_errors_come_back_in_the_status_word():
    torch = pytest.importorskip("torch")
    from gatk_amd.pairhmm import IllegalArgumentException
    hmm = HipLoglessPairHMM(0)
    try:
        b = synth.make_batch(9, 4, 80, 200, seed=13)
        _, status = run_noalloc(torch, hmm, b, read_max=int(b.read_lengths.max()) - 1)   # a read longer than declared
        assert status & STATUS_SHAPE
        _, status = run_noalloc(torch, hmm, b, hap_max=int(b.hap_lengths.max()) - 1)     # a haplotype longer than declared
        assert status & STATUS_SHAPE
        bad = synth.make_batch(9, 4, 80, 200, seed=13)
        bad.gcp = bad.gcp.copy(); bad.gcp[17] = 200
        _, status = run_noalloc(torch, hmm, bad)
        assert status == STATUS_ILLEGAL_ARGUMENT
        bad = synth.make_batch(9, 4, 80, 200, seed=13)
        bad.read_quals = bad.read_quals.copy(); bad.read_quals[3] = 255
        _, status = run_noalloc(torch, hmm, bad)
        assert status == STATUS_INDEX
        # and a clean call right behind them is clean
        got, status = run_noalloc(torch, hmm, b)
        assert status == 0 and np.array_equal(got, hmm.compute_batch(b))
        # sizes beyond the scratch array are refused on the host, before anything is enqueued
        keep, cb = device_batch(torch, b)
        cb.num_reads = 4097
        with pytest.raises(IllegalArgumentException, match="at most 4096 reads"):
            hmm.compute_device_batch_noalloc(cb, 80, 200, 1, None, None)
        cb.num_reads = b.num_reads
        with pytest.raises(IllegalArgumentException, match="too long"):
            hmm.compute_device_batch_noalloc(cb, 80, 5000, 1, None, None)
    finally:
        hmm.close()
