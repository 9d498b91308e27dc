"""Generates tests/golden/pairhmm_golden.npz.

SELF-GENERATED vectors: the reference tree holds no PairHMM fixtures and cannot be run here
(Java, no JDK), so these are outputs of the CPU oracle (oracle/pairhmm_oracle.c), written only
after its exact-log10 restatement of the in-tree Log10PairHMM and its linear-space pass agree
to 1e-9 on every pair.  They pin the oracle and the engine against silent drift between rounds;
they are not reference outputs.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gatk_amd import synth  # noqa: E402
from gatk_amd.batch import pack  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    # name: batch
    "c1_slice": synth.make_batch(24, 6, 101, 400, seed=101),
    "c5_slice": synth.config_c5(seed=102, num_reads=6, num_haps=3),
    "uniform_quals": synth.make_batch(12, 4, 76, 180, seed=103, qual_model="uniform", gop_model="random"),
    "ragged": synth.make_batch(16, 5, 90, 150, seed=104, len_jitter=80),
    "wildcards": pack([("ACNTNACGTA", [30, 2, 40, 0, 11, 25, 254, 33, 7, 60], [45] * 10, [40] * 10, [10] * 10),
                       ("NNNN", [20] * 4, None, None, None),
                       ("acgt", [20] * 4, None, None, None)],
                      ["ACGTNACGTAAC", "NNNNNNNN", "ACGT", "TTTTTTTTTTTTTTTTTTTT"]),
}


def main():
    out = {}
    for name, b in CASES.items():
        rc, exact = O.compute_likelihoods(b, O.EXACT)
        assert rc == 0
        rc, logless = O.compute_likelihoods(b, O.LOGLESS)
        assert rc == 0
        fin = np.isfinite(exact)
        assert np.array_equal(fin, np.isfinite(logless))
        assert np.abs(exact[fin] - logless[fin]).max() < 1e-9, name
        for f in ("read_bases", "read_quals", "ins_gop", "del_gop", "gcp", "read_offsets", "hap_bases", "hap_offsets"):
            out[f"{name}.{f}"] = getattr(b, f)
        out[f"{name}.expected_logless"] = logless
        out[f"{name}.expected_exact"] = exact
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pairhmm_golden.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
