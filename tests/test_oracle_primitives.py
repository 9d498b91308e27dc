"""The oracle's primitives against the assertions the reference's own unit tests make.

Replays public/gatk-utils/src/test/java/org/broadinstitute/gatk/utils/MathUtilsUnitTest.java
(:332-417 approximateLog10SumLog10, tolerance 1e-4; :420-485 log10sumLog10, tolerance 1e-14) and
QualityUtilsUnitTest.java (:53-80 every qual 0..254, tolerance 1e-9; :118-133 Q20/30/40), which
are the only tests in the reference tree that pin any part of the PairHMM path.
"""
import math

import numpy as np
import pytest


def _real_sum(vals):
    return math.log10(sum(10.0 ** v for v in vals))


PAIRS = [(0.0, 0.0), (-1.0, 0.0), (0.0, -1.0), (-2.2, -3.5), (-1.0, -7.1), (5.0, 6.2), (38.1, 16.2), (-38.1, 6.2),
         (-19.1, -37.1), (-29.1, -27.6), (-0.12345, -0.23456), (-15.7654, -17.0101)]
TRIPLES = [(0.0, 0.0, 0.0), (-1.0, 0.0, 0.0), (0.0, -1.0, -2.5), (-2.2, -3.5, -1.1), (-1.0, -7.1, 0.5), (5.0, 6.2, 1.3),
           (38.1, 16.2, 18.1), (-38.1, 6.2, 26.6), (-19.1, -37.1, -45.1), (-29.1, -27.6, -26.2),
           (-0.12345, -0.23456, -0.34567), (-15.7654, -17.0101, -17.9341)]
SINGLES = [0.0, -5.15, 130.0, -0.145]


def _partitions():
    # MathUtilsUnitTest.java:395-416: partitions of unity
    for alpha in (0.999, 0.98, 0.95, 0.90, 0.8, 0.5, 0.3, 0.1, 0.05, 0.001):
        la, l1a = math.log10(alpha), math.log10(1 - alpha)
        for npart in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1028):
            equal = [-math.log10(npart)] * npart
            mult, rem = [], 0.0
            for _ in range(npart - 1):
                mult.append(rem + la)
                rem += l1a
            mult.append(rem)
            yield equal, mult


def test_approximate_log10_sum_log10(oracle):
    L = oracle.lib()
    tol = 1e-4
    for v in SINGLES:
        assert oracle.arr_fn("oracle_approximateLog10SumLog10_arr", [v], 1) == pytest.approx(v, abs=tol)
    for a, b in PAIRS:
        want = _real_sum((a, b))
        assert L.oracle_approximateLog10SumLog10_2(a, b) == pytest.approx(want, abs=tol)
        assert oracle.arr_fn("oracle_approximateLog10SumLog10_arr", [a, b], 2) == pytest.approx(want, abs=tol)
    # MathUtilsUnitTest.java:353-354
    assert L.oracle_approximateLog10SumLog10_2(-0.12345, -math.inf) == pytest.approx(-0.12345, abs=tol)
    assert L.oracle_approximateLog10SumLog10_2(-15.7654, -math.inf) == pytest.approx(-15.7654, abs=tol)
    for t in TRIPLES:
        want = _real_sum(t)
        assert L.oracle_approximateLog10SumLog10_3(*t) == pytest.approx(want, abs=tol)
        assert oracle.arr_fn("oracle_approximateLog10SumLog10_arr", list(t), 3) == pytest.approx(want, abs=tol)
    for equal, mult in _partitions():
        assert oracle.arr_fn("oracle_approximateLog10SumLog10_arr", equal, len(equal)) == pytest.approx(0.0, abs=tol)
        assert oracle.arr_fn("oracle_approximateLog10SumLog10_arr", mult, len(mult)) == pytest.approx(0.0, abs=tol)


def test_approximate_sum_all_minus_infinity(oracle):
    # Log10PairHMM.java:185-190: the recursion relies on an all -Infinity input giving -Infinity
    assert oracle.arr_fn("oracle_approximateLog10SumLog10_arr", [-math.inf] * 3, 3) == -math.inf
    assert oracle.arr_fn("oracle_log10sumLog10", [-math.inf] * 3, 0, 3) == -math.inf


def test_log10_sum_log10(oracle):
    tol = 1e-14
    f = lambda v, s=0, e=None: oracle.arr_fn("oracle_log10sumLog10", v, s, len(v) if e is None else e)
    log3, log2 = 0.477121254719662, 0.301029995663981
    assert f([0.0, 0.0, 0.0]) == pytest.approx(log3, abs=tol)
    assert f([0.0, 0.0, 0.0], 0, 3) == pytest.approx(log3, abs=tol)
    assert f([0.0, 0.0, 0.0], 0, 2) == pytest.approx(log2, abs=tol)
    assert f([0.0, 0.0, 0.0], 0, 1) == pytest.approx(0.0, abs=tol)
    for v in SINGLES:
        assert f([v]) == pytest.approx(v, abs=tol)
    for p in PAIRS:
        assert f(list(p)) == pytest.approx(_real_sum(p), abs=tol * max(1.0, abs(_real_sum(p))))
    for t in TRIPLES:
        assert f(list(t)) == pytest.approx(_real_sum(t), abs=tol * max(1.0, abs(_real_sum(t))))
    for equal, mult in _partitions():
        assert f(equal) == pytest.approx(0.0, abs=tol)
        assert f(mult) == pytest.approx(0.0, abs=tol)


def test_quality_utils_every_qual(oracle):
    # QualityUtilsUnitTest.java:53-80
    L = oracle.lib()
    tol = 1e-9
    for q in range(255):
        err = 10.0 ** (q / -10.0)
        assert L.oracle_qualToErrorProb(q) == pytest.approx(err, abs=tol)
        assert L.oracle_qualToProb(q) == pytest.approx(1 - err, abs=tol)
        assert L.oracle_qualToErrorProbLog10(q) == pytest.approx(math.log10(err), abs=tol)
        if q > 0:
            assert L.oracle_qualToProbLog10(q) == pytest.approx(math.log10(1 - err), abs=tol)
        else:
            assert L.oracle_qualToProbLog10(q) == -math.inf


def test_qual_caches(oracle):
    # QualityUtilsUnitTest.java:118-133
    L = oracle.lib()
    for q, e, le, p, lp in ((20, 0.01, -2.0, 0.99, -0.0043648054), (30, 0.001, -3.0, 0.999, -0.000434511774),
                            (40, 0.0001, -4.0, 0.9999, -4.34316198e-5)):
        assert L.oracle_qualToErrorProb(q) == pytest.approx(e, abs=1e-6)
        assert L.oracle_qualToErrorProbLog10(q) == pytest.approx(le, abs=1e-6)
        assert L.oracle_qualToProb(q) == pytest.approx(p, abs=1e-6)
        assert L.oracle_qualToProbLog10(q) == pytest.approx(lp, abs=1e-6)


def test_qual_to_error_prob_double(oracle):
    # QualityUtilsUnitTest.java:167-173
    L = oracle.lib()
    q = 3.0
    while q < 255.0:
        assert L.oracle_qualToErrorProb_d(q) == pytest.approx(10.0 ** (q / -10.0), abs=1e-9)
        q += 0.1


def test_jacobian_table(oracle):
    # MathUtils.java:124-147: 80001 entries, log10(1 + 10^(-k * 1e-4))
    L = oracle.lib()
    assert L.oracle_jacobian_table_size() == 80001
    for k in (0, 1, 1000, 40000, 80000):
        assert L.oracle_jacobian_entry(k) == pytest.approx(math.log10(1 + 10 ** (-k * 1e-4)), abs=1e-15)


def test_match_to_match_table(oracle):
    # PairHMMModel.java:114-122,387-434
    L = oracle.lib()
    worst = 0.0
    clamped = 0
    for mx in range(255):
        for mn in range(mx + 1):
            v = L.oracle_matchToMatchProb(mn, mx)
            assert v == L.oracle_matchToMatchProb(mx, mn)
            closed = 1.0 - 10 ** (-0.1 * mn) - 10 ** (-0.1 * mx)
            if closed <= 0:
                assert v == 0.0  # log1p(-1) = -Infinity -> 10^-Infinity = 0, never negative
                assert L.oracle_matchToMatchProbLog10(mn, mx) == -math.inf
                clamped += 1
            else:
                worst = max(worst, abs(v - closed))
                assert L.oracle_matchToMatchProbLog10(mn, mx) == pytest.approx(math.log10(v), abs=1e-12)
            assert L.oracle_mm_table_entry(mx * (mx + 1) // 2 + mn, 0) == v
    assert clamped == 265
    # the only deviation from the closed form is the dropped smaller term when |a - b| >= 80
    assert 7e-9 < worst < 8e-9


def test_trans_probs_layout(oracle):
    # PairHMMModel.java:54-79,135-144,235-244
    L = oracle.lib()
    rc, t = oracle.trans_probs(45, 40, 10)
    assert rc == 0
    assert t[0] == L.oracle_matchToMatchProb(45, 40)
    assert t[1] == pytest.approx(1 - 10 ** -1.0)
    assert t[2] == pytest.approx(10 ** -4.5)
    assert t[3] == t[5] == pytest.approx(10 ** -1.0)
    assert t[4] == pytest.approx(10 ** -4.0)
    rc, tl = oracle.trans_probs(45, 40, 10, log10=True)
    assert rc == 0
    np.testing.assert_allclose(tl, np.log10(t), atol=1e-12)
    for bad in ((128, 0, 0), (0, 200, 0), (0, 0, 255)):  # negative Java bytes
        assert oracle.trans_probs(*bad)[0] == oracle.ILLEGAL_ARGUMENT
        assert oracle.trans_probs(*bad, log10=True)[0] == oracle.ILLEGAL_ARGUMENT


def test_good_log10_probability(oracle):
    # MathUtils.java:1241-1254
    L = oracle.lib()
    assert L.oracle_goodLog10Probability(0.0) and L.oracle_goodLog10Probability(-3.2) and L.oracle_goodLog10Probability(-math.inf)
    assert not L.oracle_goodLog10Probability(0.1) and not L.oracle_goodLog10Probability(math.nan) and not L.oracle_goodLog10Probability(math.inf)
