"""Host-side logic that needs no GPU: batch packing, synthetic workloads, sharding."""
import numpy as np
import pytest

from gatk_amd import sharding, synth
from gatk_amd.batch import DEFAULT_GCP, DEFAULT_INSERTION_DELETION_QUAL, pack


def test_pack_defaults_and_offsets():
    b = pack([("ACGT", [30, 31, 32, 33], None, None, None), (b"GG", [10, 11], [40, 41], [42, 43], [9, 8])], ["ACGTAC", "TT"])
    assert b.num_reads == 2 and b.num_haps == 2
    assert list(b.read_offsets) == [0, 4, 6] and list(b.hap_offsets) == [0, 6, 8]
    assert bytes(b.read_bases) == b"ACGTGG" and bytes(b.hap_bases) == b"ACGTACTT"
    # reads without BI/BD tags get flat Q45 (GATKSAMRecord.java:61,298-322); GCP defaults to 10
    assert list(b.ins_gop) == [DEFAULT_INSERTION_DELETION_QUAL] * 4 + [40, 41]
    assert list(b.del_gop) == [DEFAULT_INSERTION_DELETION_QUAL] * 4 + [42, 43]
    assert list(b.gcp) == [DEFAULT_GCP] * 4 + [9, 8]
    assert b.cells == (4 + 2) * (6 + 2)


def test_pack_rejects_mismatched_lengths():
    for bad in (("ACGT", [30] * 3, None, None, None), ("ACGT", [30] * 4, [45] * 5, None, None),
                ("ACGT", [30] * 4, None, [45] * 2, None), ("ACGT", [30] * 4, None, None, [10])):
        with pytest.raises(ValueError):
            pack([bad], ["ACGT"])


def test_read_slice():
    b = synth.make_batch(10, 3, 50, 120, seed=3, len_jitter=20)
    s = b.read_slice(4, 9)
    assert s.num_reads == 5 and s.num_haps == 3 and s.read_offsets[0] == 0
    lo = int(b.read_offsets[4])
    assert np.array_equal(s.read_bases, b.read_bases[lo:int(b.read_offsets[9])])
    assert np.array_equal(np.diff(s.read_offsets), np.diff(b.read_offsets)[4:9])


def test_read_slice_keeps_optional_arrays_optional_and_carries_mapq():
    b = synth.make_batch(10, 3, 50, 120, seed=3, len_jitter=20)
    assert b.ins_gop is not None
    b.read_mapq = np.arange(10, dtype=np.uint8) + 20
    s = b.read_slice(2, 7)
    assert list(s.read_mapq) == list(range(22, 27))
    b.ins_gop = b.del_gop = b.gcp = None
    s = b.read_slice(2, 7)
    assert s.ins_gop is None and s.del_gop is None and s.gcp is None and list(s.read_mapq) == list(range(22, 27))
    b.read_mapq = None
    assert b.read_slice(0, 3).read_mapq is None


def test_named_configs_have_the_quoted_shapes():
    c1 = synth.config_c1()
    assert c1.num_reads == 128 and c1.num_haps == 16 and set(c1.read_lengths) == {101}
    assert abs(c1.hap_lengths.mean() - 400) < 12
    c3 = synth.config_c3(num_reads=64)
    assert set(c3.read_lengths) == {151} and c3.num_haps == 64
    assert c3.cells // 64 == 151 * int(c3.hap_lengths.sum())
    c5 = synth.config_c5(num_reads=8, num_haps=4)
    assert c5.read_lengths.max() <= 250 and abs(c5.hap_lengths.mean() - 1000) < 12
    assert set(np.unique(c5.read_bases)) <= set(b"ACGTN")
    regions = synth.make_regions(5, seed=9)
    assert len(regions) == 5 and all(r.num_reads >= 4 and 2 <= r.num_haps <= 32 for r in regions)
    # same seed, same bytes
    assert np.array_equal(synth.config_c1().read_quals, c1.read_quals)


def test_lpt_assign_balances_and_covers():
    rng = np.random.default_rng(0)
    w = [int(x) for x in rng.integers(1, 1000, size=200)]
    for n in (1, 2, 3, 8):
        shards = sharding.lpt_assign(w, n)
        assert sorted(i for s in shards for i in s) == list(range(200))
        loads = [sum(w[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(w)


def test_split_reads_covers_every_read():
    b = synth.make_batch(37, 2, 80, 100, seed=4, len_jitter=60)
    for n in (1, 2, 5, 8, 64):
        ranges = sharding.split_reads(b, n)
        assert len(ranges) == n and ranges[0][0] == 0 and ranges[-1][1] == b.num_reads
        assert all(ranges[k][1] == ranges[k + 1][0] for k in range(n - 1))
