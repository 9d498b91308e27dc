"""ctypes binding of the CPU oracle (oracle/pairhmm_oracle.c) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package gatk_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpairhmm_oracle.so")

EXACT, ORIGINAL, LOGLESS = 0, 1, 2  # HMM_IMPLEMENTATION EXACT / ORIGINAL / LOGLESS_CACHING
OK, ILLEGAL_ARGUMENT, ILLEGAL_STATE, INDEX_OUT_OF_BOUNDS = 0, 1, 2, 3

_lib = None


def build():
    """Compile the oracle with the Makefile beside it."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, i32p, f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.oracle_init.restype = None
        L.oracle_computeLikelihoods.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, u8p, u8p, u8p, i32p, u8p, i32p, f64p]
        L.oracle_computeLikelihoods.restype = C.c_int
        L.oracle_logless_batch_fast.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p, u8p, u8p, i32p, u8p, i32p, f64p, C.c_int]
        L.oracle_logless_batch_fast.restype = C.c_int
        for name, at in [("oracle_qualToErrorProb", [C.c_uint8]), ("oracle_qualToProb", [C.c_uint8]),
                         ("oracle_qualToProbLog10", [C.c_uint8]), ("oracle_qualToErrorProbLog10", [C.c_uint8]),
                         ("oracle_qualToErrorProb_d", [C.c_double]),
                         ("oracle_approximateLog10SumLog10_2", [C.c_double, C.c_double]),
                         ("oracle_approximateLog10SumLog10_3", [C.c_double, C.c_double, C.c_double]),
                         ("oracle_approximateLog10SumLog10_arr", [f64p, C.c_int]),
                         ("oracle_log10sumLog10", [f64p, C.c_int, C.c_int]),
                         ("oracle_matchToMatchProb", [C.c_int, C.c_int]),
                         ("oracle_matchToMatchProbLog10", [C.c_int, C.c_int]),
                         ("oracle_jacobian_entry", [C.c_int]), ("oracle_mm_table_entry", [C.c_int, C.c_int])]:
            f = getattr(L, name)
            f.argtypes = at
            f.restype = C.c_double
        L.oracle_doNotUseTristateCorrection.argtypes = [C.c_int]
        L.oracle_doNotUseTristateCorrection.restype = None
        L.oracle_goodLog10Probability.argtypes = [C.c_double]
        L.oracle_goodLog10Probability.restype = C.c_int
        L.oracle_jacobian_table_size.restype = C.c_int
        L.oracle_qualToTransProbs.argtypes = [f64p, C.c_uint8, C.c_uint8, C.c_uint8]
        L.oracle_qualToTransProbs.restype = C.c_int
        L.oracle_qualToTransProbsLog10.argtypes = [f64p, C.c_uint8, C.c_uint8, C.c_uint8]
        L.oracle_qualToTransProbsLog10.restype = C.c_int
        L.oracle_normalizeLikelihoods.argtypes = [f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        L.oracle_normalizeLikelihoods.restype = C.c_int
        L.oracle_poorlyModelledReads.argtypes = [f64p, C.c_int, C.c_int, i32p, C.c_double, u8p]
        L.oracle_poorlyModelledReads.restype = C.c_int
        L.oracle_marginalLikelihoods.argtypes = [f64p, C.c_int, C.c_int, C.c_int, i32p, f64p]
        L.oracle_marginalLikelihoods.restype = None
        L.oracle_modifyReadQualities.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
        L.oracle_modifyReadQualities.restype = None
        L.oracle_defaultIndelQualities.argtypes = [u8p, C.c_int]
        L.oracle_defaultIndelQualities.restype = None
        L.oracle_sw_align.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      i32p, i32p, C.c_int, i32p, i32p, i32p, i32p]
        L.oracle_sw_align.restype = C.c_int
        L.simd_pairhmm_batch.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p, u8p, u8p, i32p, u8p, i32p, f64p, C.c_int, C.POINTER(C.c_int64)]
        L.simd_pairhmm_batch.restype = C.c_int
        L.simd_pairhmm_isa.restype = C.c_char_p
        L.oracle_init()
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def arr_fn(name, vals, *extra):
    v = np.ascontiguousarray(vals, dtype=np.float64)
    return getattr(lib(), name)(v.ctypes.data_as(C.POINTER(C.c_double)), *extra)


def trans_probs(ins, dele, gcp, log10=False):
    out = np.zeros(6, dtype=np.float64)
    f = lib().oracle_qualToTransProbsLog10 if log10 else lib().oracle_qualToTransProbs
    rc = f(out.ctypes.data_as(C.POINTER(C.c_double)), ins, dele, gcp)
    return rc, out


def do_not_use_tristate_correction(on=True):
    """PairHMM.doNotUseTristateCorrection (PairHMM.java:83-84): a switch of the whole oracle; clear it after use."""
    lib().oracle_doNotUseTristateCorrection(1 if on else 0)


def compute_likelihoods(batch, mode=LOGLESS, fast_threads=0):
    """Run PairHMM.computeLikelihoods on a packed batch (gatk_amd.batch.Batch-like object with
    read_bases/read_quals/ins_gop/del_gop/gcp/read_offsets/hap_bases/hap_offsets numpy arrays).
    Returns (rc, out[numReads, numHaps])."""
    nr, nh = len(batch.read_offsets) - 1, len(batch.hap_offsets) - 1
    out = np.full((nr, nh), np.nan, dtype=np.float64)
    keep = [_u8(batch.read_bases), _u8(batch.read_quals), _u8(batch.ins_gop), _u8(batch.del_gop), _u8(batch.gcp),
            _i32(batch.read_offsets), _u8(batch.hap_bases), _i32(batch.hap_offsets)]
    p = [k[1] for k in keep]
    op = out.ctypes.data_as(C.POINTER(C.c_double))
    if fast_threads > 0:
        assert mode == LOGLESS
        rc = lib().oracle_logless_batch_fast(nr, nh, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], op, fast_threads)
    else:
        rc = lib().oracle_computeLikelihoods(mode, nr, nh, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], op)
    return rc, out


def simd_compute_likelihoods(batch, threads=1):
    """The vectorised fp32-with-fp64-redo CPU baseline (pairhmm_simd.cpp) on a packed batch.
    Returns (rc, out[numReads, numHaps], pairs redone in fp64)."""
    nr, nh = len(batch.read_offsets) - 1, len(batch.hap_offsets) - 1
    out = np.full((nr, nh), np.nan, dtype=np.float64)
    keep = [_u8(batch.read_bases), _u8(batch.read_quals), _u8(batch.ins_gop), _u8(batch.del_gop), _u8(batch.gcp),
            _i32(batch.read_offsets), _u8(batch.hap_bases), _i32(batch.hap_offsets)]
    p = [k[1] for k in keep]
    redo = C.c_int64(0)
    rc = lib().simd_pairhmm_batch(nr, nh, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7],
                                  out.ctypes.data_as(C.POINTER(C.c_double)), threads, C.byref(redo))
    return rc, out, redo.value


def simd_isa():
    return lib().simd_pairhmm_isa().decode()


# ---- downstream of the path (genotyper/ReadLikelihoods.java); matrices are values[allele][read] as in the Java ----

def normalize_likelihoods(values, reference_allele_index, best_to_zero, cap):
    """ReadLikelihoods.normalizeLikelihoods on one sample's matrix.  Returns (rc, new matrix)."""
    v = np.array(values, dtype=np.float64, order="C")
    a, r = v.shape
    rc = lib().oracle_normalizeLikelihoods(v.ctypes.data_as(C.POINTER(C.c_double)), a, r, reference_allele_index, 1 if best_to_zero else 0, cap)
    return rc, v


def poorly_modelled_reads(values, read_lengths, maximum_error_per_base):
    """Flags of the reads ReadLikelihoods.filterPoorlyModeledReads removes.  Returns (rc, uint8[reads])."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    a, r = v.shape
    rl = np.ascontiguousarray(read_lengths, dtype=np.int32)
    out = np.zeros(r, dtype=np.uint8)
    rc = lib().oracle_poorlyModelledReads(v.ctypes.data_as(C.POINTER(C.c_double)), a, r, rl.ctypes.data_as(C.POINTER(C.c_int32)),
                                          maximum_error_per_base, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return rc, out


def marginal_likelihoods(values, old_to_new, new_allele_count):
    """ReadLikelihoods.marginalLikelihoods keeping every read.  Returns the new values[allele][read]."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    a, r = v.shape
    m = np.ascontiguousarray(old_to_new, dtype=np.int32)
    assert len(m) == a
    out = np.empty((new_allele_count, r), dtype=np.float64)
    lib().oracle_marginalLikelihoods(v.ctypes.data_as(C.POINTER(C.c_double)), a, new_allele_count, r, m.ctypes.data_as(C.POINTER(C.c_int32)),
                                     out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def modify_read_qualities(quals, mapq=None, squash=False):
    """Base qualities of one read as the PairHMM receives them: capped by mapq (None: no cap), squashed under 18."""
    q = np.array(quals, dtype=np.uint8)
    lib().oracle_modifyReadQualities(q.ctypes.data_as(C.POINTER(C.c_uint8)), len(q), -1 if mapq is None else int(mapq), 1 if squash else 0)
    return q


def default_indel_qualities(n):
    q = np.zeros(n, dtype=np.uint8)
    lib().oracle_defaultIndelQualities(q.ctypes.data_as(C.POINTER(C.c_uint8)), n)
    return q


# ---- Smith-Waterman (smithwaterman/SWPairwiseAlignment.java) ----
SW_SOFTCLIP, SW_INDEL, SW_LEADING_INDEL, SW_IGNORE = 0, 1, 2, 3
SW_ORIGINAL_DEFAULT = (3, -1, -4, -3)      # SWParameterSet.java:38
SW_STANDARD_NGS = (25, -50, -110, -6)      # SWParameterSet.java:43


def sw_align(reference, alternate, parameters=SW_ORIGINAL_DEFAULT, strategy=SW_SOFTCLIP, matrices=False):
    """SWPairwiseAlignment(reference, alternate, parameters, strategy): returns (rc, cigar string, alignment offset)
    and, with matrices=True, also the sw and btrack matrices."""
    ref, alt = _u8(np.frombuffer(bytes(reference), dtype=np.uint8)), _u8(np.frombuffer(bytes(alternate), dtype=np.uint8))
    n, m = len(ref[0]), len(alt[0])
    cap = n + m + 8
    ops, lens = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    cnt, off = C.c_int32(0), C.c_int32(0)
    sw = np.zeros((n + 1, m + 1), dtype=np.int32) if matrices else None
    bt = np.zeros((n + 1, m + 1), dtype=np.int32) if matrices else None
    i32 = C.POINTER(C.c_int32)
    rc = lib().oracle_sw_align(ref[1], n, alt[1], m, *[int(x) for x in parameters], strategy, ops.ctypes.data_as(i32), lens.ctypes.data_as(i32), cap,
                               C.byref(cnt), C.byref(off), sw.ctypes.data_as(i32) if matrices else None, bt.ctypes.data_as(i32) if matrices else None)
    cigar = "".join(f"{lens[k]}{'MIDS'[ops[k]]}" for k in range(cnt.value)) if rc == 0 else None
    return (rc, cigar, off.value, sw, bt) if matrices else (rc, cigar, off.value)
