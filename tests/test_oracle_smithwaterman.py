"""The oracle's restatement of smithwaterman/SWPairwiseAlignment.java against cases worked out by hand
from the Java text.  The reference tree holds no test of the aligner: parity unpinned beyond these."""
import numpy as np


def test_identical_and_simple_indels(oracle):
    assert oracle.sw_align(b"ACGTACGT", b"ACGTACGT") == (0, "8M", 0)
    assert oracle.sw_align(b"AAAAACCCCCGGGGG", b"AAAAAGGGGG") == (0, "5M5D5M", 0)
    assert oracle.sw_align(b"AAAAAGGGGG", b"AAAAACCCCCGGGGG") == (0, "5M5I5M", 0)


def test_overhang_strategies(oracle):
    ngs = oracle.SW_STANDARD_NGS
    # alternate inside the reference: the offset says where it starts
    assert oracle.sw_align(b"TTTTACGTACGTTTTT", b"ACGTACGT") == (0, "8M", 4)
    # alternate hanging over both ends of the reference (documented at SWPairwiseAlignment.java:440-446)
    assert oracle.sw_align(b"ACGTACGT", b"GGACGTACGTGG", ngs, oracle.SW_SOFTCLIP) == (0, "2S8M2S", 0)
    assert oracle.sw_align(b"ACGTACGT", b"GGACGTACGTGG", ngs, oracle.SW_IGNORE) == (0, "12M", -2)
    assert oracle.sw_align(b"ACGTACGT", b"GGACGTACGTGG", ngs, oracle.SW_INDEL) == (0, "2I8M2I", 0)
    assert oracle.sw_align(b"GGACGTACGTGG", b"ACGTACGT", ngs, oracle.SW_INDEL) == (0, "2D8M2D", 0)
    assert oracle.sw_align(b"GGACGTACGTGG", b"ACGTACGT", ngs, oracle.SW_LEADING_INDEL) == (0, "2D8M", 0)


def test_matrix_of_a_tiny_case(oracle):
    # ORIGINAL_DEFAULT (3, -1, -4, -3), SOFTCLIP: first row and column stay 0
    rc, cigar, off, m, bt = oracle.sw_align(b"AC", b"AGC", matrices=True)
    assert rc == 0
    np.testing.assert_array_equal(m[0], [0, 0, 0, 0])
    np.testing.assert_array_equal(m[:, 0], [0, 0, 0])
    # row 1 ('A'): match 3; then a horizontal gap from it: 3-4 = -1 beats the mismatch -1? equal -> diagonal wins
    np.testing.assert_array_equal(m[1], [0, 3, -1, -1])
    assert bt[1, 1] == 0 and bt[1, 2] == 0
    # row 2 ('C'): under the match a vertical gap 3-4 = -1 ties with the mismatch -1 -> diagonal;
    # 'C' against 'G' continues the diagonal from 3: 2; 'C' against 'C' from -1: 2
    np.testing.assert_array_equal(m[2], [0, -1, 2, 2])
    # end cell: the last column's best is sw[2][3] = 2; the bottom row holds the same score at j = 2, closer to
    # the diagonal (|2 - 2| < |2 - 3|, :383-384), so the alignment ends there and the last base is clipped
    assert (cigar, off) == ("2M1S", 0)


def test_empty_sequences_are_rejected(oracle):
    assert oracle.sw_align(b"", b"A")[0] == 1
    assert oracle.sw_align(b"A", b"")[0] == 1
