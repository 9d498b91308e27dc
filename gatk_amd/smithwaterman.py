"""Python mirror of the reference's Smith-Waterman aligner interface over the C-ABI of
include/gatk_smithwaterman.h (smithwaterman/SmithWaterman.java:38-56: getCigar(),
getAlignmentStart2wrt1(); constructors smithwaterman/SWPairwiseAlignment.java:132-175).

The GPU does the work; without one the calls raise (there is no CPU path)."""
import ctypes as C
import time

import numpy as np

from .pairhmm import DeviceError, IllegalArgumentException, load_library

# SWPairwiseAlignment.OVERHANG_STRATEGY (SWPairwiseAlignment.java:69-93)
SOFTCLIP, INDEL, LEADING_INDEL, IGNORE = 0, 1, 2, 3
# SWParameterSet.java:38,43
ORIGINAL_DEFAULT = (3, -1, -4, -3)
STANDARD_NGS = (25, -50, -110, -6)

_OPS = "MIDS"


class CParameters(C.Structure):
    _fields_ = [("w_match", C.c_int32), ("w_mismatch", C.c_int32), ("w_open", C.c_int32), ("w_extend", C.c_int32)]


last_call_seconds = 0.0  # wall time of the last gatk_sw_align call, without this module's marshalling


class CigarTooLong(RuntimeError):
    pass


def _lib():
    L = load_library()
    L.gatk_sw_align.restype = C.c_int
    L.gatk_sw_last_error.restype = C.c_char_p
    return L


def align_batch(references, alternates, parameters=ORIGINAL_DEFAULT, strategy=SOFTCLIP, device=0, cigar_capacity=64):
    """Aligns alternates[k] to references[k] (bytes-like).  Returns a list of (cigar string, alignment offset)."""
    n = len(references)
    assert len(alternates) == n
    if n == 0:
        return []
    refs = [np.frombuffer(bytes(r), dtype=np.uint8) for r in references]
    alts = [np.frombuffer(bytes(a), dtype=np.uint8) for a in alternates]
    dummy = np.zeros(1, dtype=np.uint8)
    rp = (C.c_void_p * n)(*[(r.ctypes.data if r.size else dummy.ctypes.data) for r in refs])
    ap = (C.c_void_p * n)(*[(a.ctypes.data if a.size else dummy.ctypes.data) for a in alts])
    rl = np.array([r.size for r in refs], dtype=np.int32)
    al = np.array([a.size for a in alts], dtype=np.int32)
    ops = np.zeros((n, cigar_capacity), dtype=np.int32)
    lens = np.zeros((n, cigar_capacity), dtype=np.int32)
    counts = np.zeros(n, dtype=np.int32)
    offsets = np.zeros(n, dtype=np.int32)
    par = CParameters(*[int(x) for x in parameters])
    L = _lib()
    global last_call_seconds
    t0 = time.perf_counter()
    rc = L.gatk_sw_align(C.c_int32(device), C.c_int32(n), rp, C.c_void_p(rl.ctypes.data), ap, C.c_void_p(al.ctypes.data), C.byref(par),
                         C.c_int32(strategy), C.c_int32(cigar_capacity), C.c_void_p(ops.ctypes.data), C.c_void_p(lens.ctypes.data),
                         C.c_void_p(counts.ctypes.data), C.c_void_p(offsets.ctypes.data))
    last_call_seconds = time.perf_counter() - t0
    if rc != 0:
        msg = L.gatk_sw_last_error().decode()
        raise {1: IllegalArgumentException, 5: CigarTooLong}.get(rc, DeviceError)(msg)
    return [("".join(f"{lens[k, x]}{_OPS[ops[k, x]]}" for x in range(counts[k])), int(offsets[k])) for k in range(n)]


class SWPairwiseAlignment:
    """new SWPairwiseAlignment(seq1, seq2, parameters, strategy) (SWPairwiseAlignment.java:149-153)."""

    def __init__(self, seq1, seq2, parameters=ORIGINAL_DEFAULT, strategy=SOFTCLIP, device=0):
        (self._cigar, self._offset), = align_batch([seq1], [seq2], parameters, strategy, device)

    def getCigar(self):
        return self._cigar

    def getAlignmentStart2wrt1(self):
        return self._offset
