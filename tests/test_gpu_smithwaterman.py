"""The gfx950 Smith-Waterman aligner against the oracle's restatement of
smithwaterman/SWPairwiseAlignment.java, through the C-ABI.  Integer scores and back-tracks: the CIGAR
and the alignment offset must be identical."""
import numpy as np
import pytest

from gatk_amd import smithwaterman as sw
from gatk_amd.pairhmm import IllegalArgumentException

pytestmark = pytest.mark.gpu

STRATEGIES = [sw.SOFTCLIP, sw.INDEL, sw.LEADING_INDEL, sw.IGNORE]


def mutate(rng, seq, snps, indels):
    s = list(seq)
    for _ in range(snps):
        s[rng.integers(len(s))] = "ACGT"[rng.integers(4)]
    for _ in range(indels):
        if len(s) < 4:
            break
        p = int(rng.integers(1, len(s) - 1))
        if rng.random() < 0.5:
            del s[p:p + int(rng.integers(1, 12))]
        else:
            s[p:p] = list("".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 12)))))
    return "".join(s) or "A"


def random_pairs(rng, count, ref_len, jitter):
    refs, alts = [], []
    for _ in range(count):
        n = int(ref_len + rng.integers(-jitter, jitter + 1))
        ref = "".join(rng.choice(list("ACGT"), size=max(n, 1)))
        lo = int(rng.integers(0, max(1, n // 5)))
        hi = n - int(rng.integers(0, max(1, n // 5)))
        alt = mutate(rng, ref[lo:max(hi, lo + 1)] if rng.random() < 0.5 else ref, int(rng.integers(0, 5)), int(rng.integers(0, 4)))
        if rng.random() < 0.3:
            alt = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 9)))) + alt
        refs.append(ref.encode())
        alts.append(alt.encode())
    return refs, alts


@pytest.mark.parametrize("parameters", [sw.ORIGINAL_DEFAULT, sw.STANDARD_NGS])
@pytest.mark.parametrize("strategy", STRATEGIES)
def test_random_haplotype_like_pairs(oracle, parameters, strategy):
    rng = np.random.default_rng(100 + strategy)
    refs, alts = random_pairs(rng, 60, 300, 120)
    got = sw.align_batch(refs, alts, parameters, strategy)
    for r, a, (cigar, off) in zip(refs, alts, got):
        rc, want_cigar, want_off = oracle.sw_align(r, a, parameters, strategy)
        assert rc == 0
        assert (cigar, off) == (want_cigar, want_off), (r, a)


# one length per fill-kernel instantiation (rows per lane 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32) and the edges between them
@pytest.mark.parametrize("ref_len", [1, 2, 63, 64, 65, 128, 129, 192, 193, 300, 380, 440, 500, 600, 700, 1000, 1025, 1536, 1537, 2048])
def test_every_rows_per_lane_instantiation(oracle, ref_len):
    rng = np.random.default_rng(ref_len)
    refs, alts = random_pairs(rng, 6, ref_len, 0)
    alts[0] = b"A"                       # one-base alternate
    alts[1] = refs[1]                    # identical
    for strategy in (sw.SOFTCLIP, sw.INDEL):
        got = sw.align_batch(refs, alts, sw.STANDARD_NGS, strategy)
        for r, a, g in zip(refs, alts, got):
            rc, c, o = oracle.sw_align(r, a, sw.STANDARD_NGS, strategy)
            assert rc == 0 and g == (c, o)


def test_one_call_with_references_of_many_lengths(oracle):
    """Every pair takes the fill kernel that fits its own reference: one call, several launches, results in
    the caller's order."""
    rng = np.random.default_rng(12)
    refs, alts = [], []
    for ref_len in (40, 700, 130, 260, 2000, 64, 450, 330, 1100, 90, 520, 400):
        r, a = random_pairs(rng, 3, ref_len, 10)
        refs += r
        alts += a
    for strategy in (sw.SOFTCLIP, sw.LEADING_INDEL):
        got = sw.align_batch(refs, alts, sw.STANDARD_NGS, strategy)
        for r, a, g in zip(refs, alts, got):
            rc, c, o = oracle.sw_align(r, a, sw.STANDARD_NGS, strategy)
            assert rc == 0 and g == (c, o)


def test_alignments_with_many_cigar_elements(oracle):
    """Lists longer than what the first fetch of a call brings back (many indels), next to short ones."""
    rng = np.random.default_rng(3)
    refs, alts = [], []
    for n_indels in (0, 14, 2, 20):
        ref = "".join(rng.choice(list("ACGT"), size=900))
        alt, at = [], 0
        for k in range(n_indels):
            nxt = at + 900 // (n_indels + 1)
            alt.append(ref[at:nxt - 6] if k % 2 == 0 else ref[at:nxt] + "TTTTT")  # a deletion or an insertion
            at = nxt
        alt.append(ref[at:])
        refs.append(ref.encode())
        alts.append("".join(alt).encode())
    got = sw.align_batch(refs, alts, sw.STANDARD_NGS, sw.SOFTCLIP)
    assert max(c.count("D") + c.count("I") for c, _ in got) >= 14
    for r, a, g in zip(refs, alts, got):
        rc, c, o = oracle.sw_align(r, a, sw.STANDARD_NGS, sw.SOFTCLIP)
        assert rc == 0 and g == (c, o)


def test_the_reference_benchmark_sequences(oracle):
    # smithwaterman/SmithWatermanBenchmark.java:48-72: two SNPs ('X' against 'Y') in runs of 'A'
    refs, alts = [], []
    for mid in (10, 50, 100, 500):
        for end in (10, 50, 100, 500):
            refs.append(b"A" * end + b"X" + b"A" * mid + b"X" + b"A" * end)
            alts.append(b"A" * end + b"Y" + b"A" * mid + b"Y" + b"A" * end)
    got = sw.align_batch(refs, alts)
    for r, a, g in zip(refs, alts, got):
        assert g == (f"{len(r)}M", 0)
        assert g == oracle.sw_align(r, a)[1:]


def test_known_small_cases():
    a = sw.SWPairwiseAlignment(b"AAAAACCCCCGGGGG", b"AAAAAGGGGG")
    assert (a.getCigar(), a.getAlignmentStart2wrt1()) == ("5M5D5M", 0)
    a = sw.SWPairwiseAlignment(b"TTTTACGTACGTTTTT", b"ACGTACGT")
    assert (a.getCigar(), a.getAlignmentStart2wrt1()) == ("8M", 4)
    a = sw.SWPairwiseAlignment(b"ACGTACGT", b"GGACGTACGTGG", sw.STANDARD_NGS, sw.IGNORE)
    assert (a.getCigar(), a.getAlignmentStart2wrt1()) == ("12M", -2)


def test_errors():
    with pytest.raises(IllegalArgumentException):
        sw.align_batch([b""], [b"ACGT"])
    with pytest.raises(IllegalArgumentException):
        sw.align_batch([b"ACGT"], [b""])
    with pytest.raises(IllegalArgumentException):
        sw.align_batch([b"A" * 2049], [b"ACGT"])
    with pytest.raises(sw.CigarTooLong):
        sw.align_batch([b"AAAAACCCCCGGGGG"], [b"AAAAAGGGGG"], cigar_capacity=2)
    assert sw.align_batch([], []) == []


NEW_SW = (200, -150, -260, -11)  # the heaviest weights GATK ships: they overflow the packed cell's score field on long pairs


def test_both_cell_forms_in_one_call(oracle):
    """Short pairs take the packed cell form, pairs whose scores or lengths do not fit its fields the plain one."""
    rng = np.random.default_rng(21)
    refs, alts = [], []
    for ref_len in (30, 250, 400, 640, 1000, 1500, 2040, 505, 509, 1019, 1021):
        r, a = random_pairs(rng, 2, ref_len, 3)
        refs += r
        alts += a
    refs.append(b"ACGT" * 40)
    alts.append(b"ACGT" * 3000)  # an alternate far longer than any reference row count
    for parameters in (NEW_SW, sw.ORIGINAL_DEFAULT):
        for strategy in (sw.SOFTCLIP, sw.INDEL):
            got = sw.align_batch(refs, alts, parameters, strategy, cigar_capacity=256)
            for r, a, g in zip(refs, alts, got):
                rc, c, o = oracle.sw_align(r, a, parameters, strategy)
                assert rc == 0 and g == (c, o), (len(r), len(a), parameters, strategy)


def test_weights_beyond_the_packed_fields(oracle):
    rng = np.random.default_rng(22)
    refs, alts = random_pairs(rng, 12, 200, 60)
    heavy = (200000, -150000, -260000, -11000)
    got = sw.align_batch(refs, alts, heavy, sw.SOFTCLIP)
    for r, a, g in zip(refs, alts, got):
        rc, c, o = oracle.sw_align(r, a, heavy, sw.SOFTCLIP)
        assert rc == 0 and g == (c, o)


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_ties_in_low_complexity_sequences(oracle, strategy):
    """Repeats make equal scores everywhere: the order diagonal, horizontal gap, vertical gap and 'an older gap wins a
    tie with a newly opened one' decide the CIGAR."""
    rng = np.random.default_rng(23 + strategy)
    refs, alts = [], []
    for unit in ("A", "AC", "ACG", "AACC", "ACGTT"):
        for n in (17, 64, 130, 333):
            ref = (unit * (n // len(unit) + 1))[:n]
            alt = mutate(rng, ref, 2, 3)
            refs.append(ref.encode())
            alts.append(alt.encode())
            refs.append(alt.encode())
            alts.append(ref.encode())
    for parameters in (sw.ORIGINAL_DEFAULT, (1, -1, -1, -1), (2, -2, -2, 0), sw.STANDARD_NGS):
        got = sw.align_batch(refs, alts, parameters, strategy, cigar_capacity=512)
        for r, a, g in zip(refs, alts, got):
            rc, c, o = oracle.sw_align(r, a, parameters, strategy)
            assert rc == 0 and g == (c, o), (r, a, parameters)
