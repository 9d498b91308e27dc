"""The oracle's restatement of the steps right behind the path (genotyper/ReadLikelihoods.java:
normalizeLikelihoods :353-404, filterPoorlyModeledReads :1059-1093, marginalLikelihoods :911-939)
against hand-worked cases.  The reference tree holds no test of these methods: parity unpinned beyond
what is worked out here from the Java text."""
import numpy as np

INF = np.inf


def test_normalize_caps_against_best_alternative(oracle):
    # alleles x reads; reference allele = 0
    v = np.array([[-1.0, -9.0, -INF], [-3.0, -2.0, -INF], [-20.0, -2.5, -INF]])
    rc, out = oracle.normalize_likelihoods(v, 0, False, -4.0)
    assert rc == 0
    # read 0: best alternative -3 -> cap -7: only -20 is raised; the reference value is never a "best alternative"
    np.testing.assert_array_equal(out[:, 0], [-1.0, -3.0, -7.0])
    # read 1: best alternative -2 -> cap -6: the reference -9 is raised to -6
    np.testing.assert_array_equal(out[:, 1], [-6.0, -2.0, -2.5])
    # read 2: everything -inf stays -inf (cap is -inf, nothing is below it)
    assert np.all(np.isneginf(out[:, 2]))


def test_normalize_best_to_zero(oracle):
    v = np.array([[-1.0, -9.0, -INF], [-3.0, -2.0, -INF], [-20.0, -2.5, -INF]])
    rc, out = oracle.normalize_likelihoods(v, 0, True, -4.0)
    assert rc == 0
    np.testing.assert_array_equal(out[:, 0], [0.0, -2.0, -6.0])   # best absolute is the reference, -1
    np.testing.assert_array_equal(out[:, 1], [-4.0, 0.0, -0.5])   # capped at -6, then minus -2
    np.testing.assert_array_equal(out[:, 2], [0.0, 0.0, 0.0])     # :388-390
    rc, out = oracle.normalize_likelihoods(v[:, :2], -1, True, -INF)  # no reference allele, no cap: plain shift
    np.testing.assert_array_equal(out, [[0.0, -7.0], [-2.0, 0.0], [-19.0, -0.5]])


def test_normalize_trivial_and_illegal_cases(oracle):
    v = np.array([[-1.0, -2.0]])
    for cap in (0.0, 1.0, np.nan):
        assert oracle.normalize_likelihoods(v, 0, True, cap)[0] == 1  # IllegalArgumentException
    np.testing.assert_array_equal(oracle.normalize_likelihoods(v, 0, False, -3.0)[1], v)   # one allele, not bestToZero
    np.testing.assert_array_equal(oracle.normalize_likelihoods(v, 0, True, -3.0)[1], [[0.0, 0.0]])
    two = np.array([[-1.0], [-5.0]])
    np.testing.assert_array_equal(oracle.normalize_likelihoods(two, 0, False, -INF)[1], two)


def test_poorly_modelled_reads(oracle):
    # threshold = min(2, ceil(len * rate)) * -4: len 100 @ 0.02 -> 2 errors -> -8; len 30 @ 0.02 -> 1 -> -4
    v = np.array([[-8.0, -8.1, -4.0, -4.5], [-30.0, -9.0, -50.0, -4.01]])
    rc, flags = oracle.poorly_modelled_reads(v, [100, 100, 30, 30], 0.02)
    assert rc == 0
    np.testing.assert_array_equal(flags, [0, 1, 0, 1])
    assert oracle.poorly_modelled_reads(v, [100] * 4, 0.0)[0] == 1
    assert oracle.poorly_modelled_reads(v, [100] * 4, np.nan)[0] == 1
    assert oracle.poorly_modelled_reads(np.zeros((0, 4)), [100] * 4, 0.02)[0] == 2  # IllegalStateException


def test_marginal_likelihoods(oracle):
    v = np.array([[-1.0, -9.0], [-3.0, -2.0], [-20.0, -2.5], [-0.5, -INF]])
    out = oracle.marginal_likelihoods(v, [0, 1, 1, -1], 3)
    np.testing.assert_array_equal(out, [[-1.0, -9.0], [-3.0, -2.0], [-INF, -INF]])
