"""Loader of tests/golden/reference_outputs.txt: the log10 likelihoods the reference's own Java (Log10PairHMM, exact and
approximate) gives on the golden inputs, printed by java/test/PinOracleTest.java on a machine with a JDK.  The file is
absent until someone has run that kit; the tests that use it skip, and the golden fixtures stay self-generated
("parity unpinned") until then."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_OUTPUTS = os.path.join(HERE, "golden", "reference_outputs.txt")
GOLDEN = os.path.join(HERE, "golden", "pairhmm_golden.npz")


def parse_double(text):
    """Java's Double.toString spellings as well as Python's."""
    t = text.strip()
    if t in ("-Infinity", "-inf"):
        return -np.inf
    if t in ("Infinity", "inf"):
        return np.inf
    if t in ("NaN", "nan"):
        return np.nan
    return float(t)


def load_reference_outputs(path=REFERENCE_OUTPUTS, golden=GOLDEN):
    """{case: {"exact": [reads, haps] array, "original": ...}}, or None when the file does not exist.  Every pair of
    every mode a case appears with must be present exactly once."""
    if not os.path.exists(path):
        return None
    z = np.load(golden)
    shapes = {f.split(".")[0]: (len(z[f]) - 1) for f in z.files if f.endswith(".read_offsets")}
    haps = {f.split(".")[0]: (len(z[f]) - 1) for f in z.files if f.endswith(".hap_offsets")}
    out, seen = {}, {}
    with open(path) as f:
        for line in f:
            if not line.strip() or line.startswith("#"):
                continue
            name, mode, r, a, value = line.split()
            if name not in shapes:
                raise ValueError(f"{path}: unknown case {name}")
            if mode not in ("exact", "original"):
                raise ValueError(f"{path}: unknown mode {mode}")
            arr = out.setdefault(name, {}).setdefault(mode, np.full((shapes[name], haps[name]), np.nan))
            cnt = seen.setdefault((name, mode), np.zeros((shapes[name], haps[name]), dtype=np.int32))
            arr[int(r), int(a)] = parse_double(value)
            cnt[int(r), int(a)] += 1
    for (name, mode), cnt in seen.items():
        if not (cnt == 1).all():
            raise ValueError(f"{path}: case {name} mode {mode} does not list every pair exactly once")
    return out
