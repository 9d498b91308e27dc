"""Host-side mirror of the reference's PairHMM plugin interface over the C-ABI library.

Class and method names follow pairhmm/PairHMM.java of the reference: `initialize`,
`computeLikelihoods` (fills a `ReadLikelihoodsMatrix`, the stand-in for
genotyper/ReadLikelihoods.java's `Matrix`), `getLikelihoodArray`, `finalizeRegion`, `close`;
`batchAdd` / `batchGetResult` follow pairhmm/BatchPairHMM.java.  All arithmetic happens in
libgatk_pairhmm_hip.so (hand-written HIP kernels); this module only marshals bytes.  If the
library or a GPU is missing, construction raises: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from .batch import Batch, pack

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GATK_PAIRHMM_LIB", os.path.join(_HERE, "libgatk_pairhmm_hip.so"))  # override: experimental builds

OK, ILLEGAL_ARGUMENT, ILLEGAL_STATE, INDEX_OUT_OF_BOUNDS, DEVICE_ERROR, OUT_OF_MEMORY = range(6)

BASE_QUALITY_SCORE_THRESHOLD = 18  # PairHMM.java:54


class HMM_IMPLEMENTATION:  # PairHMM.java:56-75; the entry this engine adds is the last one
    EXACT, ORIGINAL, LOGLESS_CACHING, VECTOR_LOGLESS_CACHING = "EXACT", "ORIGINAL", "LOGLESS_CACHING", "VECTOR_LOGLESS_CACHING"
    HIP_LOGLESS_CACHING = "HIP_LOGLESS_CACHING"


class IllegalArgumentException(ValueError):
    pass


class IllegalStateException(RuntimeError):
    pass


class ArrayIndexOutOfBoundsException(IndexError):
    pass


class DeviceError(RuntimeError):
    pass


_EXC = {ILLEGAL_ARGUMENT: IllegalArgumentException, ILLEGAL_STATE: IllegalStateException,
        INDEX_OUT_OF_BOUNDS: ArrayIndexOutOfBoundsException, DEVICE_ERROR: DeviceError, OUT_OF_MEMORY: MemoryError}


class CBatch(C.Structure):
    _fields_ = [("num_reads", C.c_int32), ("num_haps", C.c_int32),
                ("read_bases", C.c_void_p), ("read_quals", C.c_void_p), ("ins_gop", C.c_void_p),
                ("del_gop", C.c_void_p), ("gcp", C.c_void_p), ("read_offsets", C.c_void_p),
                ("hap_bases", C.c_void_p), ("hap_offsets", C.c_void_p), ("read_mapq", C.c_void_p)]


class CStats(C.Structure):
    _fields_ = [("pairs", C.c_int64), ("cells", C.c_int64), ("wave_tasks", C.c_int64), ("rerun_tasks", C.c_int64),
                ("rerun_pairs", C.c_int64), ("fp32_kernel_ms", C.c_double), ("fp64_kernel_ms", C.c_double),
                ("host_plan_ms", C.c_double), ("host_upload_ms", C.c_double), ("aux_kernel_ms", C.c_double),
                ("timed_passes", C.c_int64), ("main_rows_per_lane", C.c_int32), ("main_emit_mode", C.c_int32),
                ("main_stream_pass", C.c_int32), ("main_fp64", C.c_int32),
                ("lane_rows_allocated", C.c_int64), ("lane_rows_used", C.c_int64),
                ("shared_passes", C.c_int64), ("shared_requests", C.c_int64),
                ("region_hits", C.c_int64), ("region_misses", C.c_int64)]


# every symbol include/gatk_pairhmm.h declares
EXPORTS = [
    "gatk_pairhmm_create", "gatk_pairhmm_initialize", "gatk_pairhmm_compute_likelihoods",
    "gatk_pairhmm_compute_likelihoods_ptrs", "gatk_pairhmm_compute_regions", "gatk_pairhmm_compute_pair",
    "gatk_pairhmm_batch_add", "gatk_pairhmm_batch_get_result", "gatk_pairhmm_get_likelihood_array",
    "gatk_pairhmm_finalize_region", "gatk_pairhmm_close", "gatk_pairhmm_set_number_of_threads",
    "gatk_pairhmm_get_stats", "gatk_pairhmm_last_error", "gatk_pairhmm_version", "gatk_pairhmm_device_count",
    "gatk_pairhmm_set_option", "gatk_pairhmm_upload", "gatk_pairhmm_compute_resident", "gatk_pairhmm_download",
    "gatk_pairhmm_resident_free", "gatk_pairhmm_get_stream", "gatk_pairhmm_resident_device_output",
    "gatk_pairhmm_wait_resident", "gatk_pairhmm_normalize_resident", "gatk_pairhmm_poorly_modeled_resident",
    "gatk_pairhmm_marginalize_resident", "gatk_pairhmm_compute_likelihoods_noalloc", "gatk_pairhmm_create_multi",
    "gatk_pairhmm_initialize_region", "gatk_pairhmm_submit", "gatk_pairhmm_fetch",
]

_lib = None


def load_library():
    """dlopen the engine; raises OSError when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.gatk_pairhmm_last_error.restype = C.c_char_p
        L.gatk_pairhmm_version.restype = C.c_char_p
        for name in EXPORTS:
            f = getattr(L, name)
            if name not in ("gatk_pairhmm_last_error", "gatk_pairhmm_version"):
                f.restype = C.c_int
        L.gatk_pairhmm_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.gatk_pairhmm_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.gatk_pairhmm_create_multi.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.gatk_pairhmm_initialize.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.gatk_pairhmm_compute_likelihoods.argtypes = [C.c_void_p, C.POINTER(CBatch), C.c_void_p]
        L.gatk_pairhmm_compute_regions.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CBatch), C.POINTER(C.c_void_p)]
        L.gatk_pairhmm_upload.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CBatch), C.POINTER(C.c_void_p)]
        L.gatk_pairhmm_compute_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.gatk_pairhmm_download.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.gatk_pairhmm_resident_free.argtypes = [C.c_void_p, C.c_void_p]
        L.gatk_pairhmm_wait_resident.argtypes = [C.c_void_p, C.c_void_p]
        L.gatk_pairhmm_normalize_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double]
        L.gatk_pairhmm_poorly_modeled_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_void_p)]
        L.gatk_pairhmm_marginalize_resident.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p)]
        L.gatk_pairhmm_get_stats.argtypes = [C.c_void_p, C.POINTER(CStats)]
        L.gatk_pairhmm_close.argtypes = [C.c_void_p]
        L.gatk_pairhmm_finalize_region.argtypes = [C.c_void_p]
        L.gatk_pairhmm_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.gatk_pairhmm_resident_device_output.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.gatk_pairhmm_get_likelihood_array.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.gatk_pairhmm_batch_get_result.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.gatk_pairhmm_compute_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
        L.gatk_pairhmm_compute_likelihoods_ptrs.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 9
        L.gatk_pairhmm_batch_add.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.gatk_pairhmm_compute_likelihoods_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CBatch), C.c_int32, C.c_int32,
                                                               C.c_void_p, C.c_void_p]
        L.gatk_pairhmm_initialize_region.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CBatch), C.c_int32, C.c_int32]
        L.gatk_pairhmm_submit.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CBatch), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.gatk_pairhmm_fetch.argtypes = [C.c_void_p, C.c_int64]
        L.gatk_pairhmm_device_count.argtypes = [C.POINTER(C.c_int32)]
        L.gatk_pairhmm_set_number_of_threads.argtypes = [C.c_int32]
        _lib = L
    return _lib


def _check(rc):
    if rc != OK:
        msg = load_library().gatk_pairhmm_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, RuntimeError)(msg)


def _ptr(a):
    return a.ctypes.data if a.size else None


def to_cbatch(b: Batch, keep: list) -> CBatch:
    # ins_gop / del_gop / gcp set to None ask for the engine's defaults (Q45, Q45, 10)
    arrs = [None if x is None else np.ascontiguousarray(x, dtype=np.uint8) for x in (b.read_bases, b.read_quals, b.ins_gop, b.del_gop, b.gcp)]
    mapq = getattr(b, "read_mapq", None)
    mapq = None if mapq is None else np.ascontiguousarray(mapq, dtype=np.uint8)
    ro = np.ascontiguousarray(b.read_offsets, dtype=np.int32)
    hb = np.ascontiguousarray(b.hap_bases, dtype=np.uint8)
    ho = np.ascontiguousarray(b.hap_offsets, dtype=np.int32)
    # a zero-size numpy array has a dangling data pointer: give the C side a real one
    dummy = np.zeros(1, dtype=np.uint8)
    keep.extend(arrs + [ro, hb, ho, dummy, mapq])
    p = [None if a is None else (_ptr(a) or dummy.ctypes.data) for a in arrs]
    return CBatch(b.num_reads, b.num_haps, p[0], p[1], p[2], p[3], p[4], ro.ctypes.data,
                  _ptr(hb) or dummy.ctypes.data, ho.ctypes.data, None if mapq is None else (_ptr(mapq) or dummy.ctypes.data))


class ReadLikelihoodsMatrix:
    """Likelihood matrix of one sample: `values[a][r]`, allele-major as in
    genotyper/ReadLikelihoods.java:56-64,1638-1640, with the `Matrix` accessors the PairHMM uses
    (genotyper/ReadLikelihoods.java:652-756): alleles(), reads(), set(a, r, v), get(a, r)."""

    def __init__(self, alleles, reads):
        self._alleles = list(alleles)
        self._reads = list(reads)
        self.values = np.full((len(self._alleles), len(self._reads)), np.nan, dtype=np.float64)

    def alleles(self):
        return self._alleles

    def reads(self):
        return self._reads

    def alleleCount(self):
        return len(self._alleles)

    def readCount(self):
        return len(self._reads)

    def set(self, allele_index, read_index, value):
        self.values[allele_index, read_index] = value

    def get(self, allele_index, read_index):
        return float(self.values[allele_index, read_index])


class Resident:
    """Region batches kept in HBM (gatk_pairhmm_upload)."""

    def __init__(self, hmm, handle, shapes):
        self.hmm, self.handle, self.shapes = hmm, handle, shapes

    def compute(self, sync=True):
        _check(load_library().gatk_pairhmm_compute_resident(self.hmm._h, self.handle, 1 if sync else 0))

    def wait(self):
        _check(load_library().gatk_pairhmm_wait_resident(self.hmm._h, self.handle))

    def download(self):
        outs = [np.empty(nr * nh, dtype=np.float64) for nr, nh in self.shapes]
        arr = (C.c_void_p * max(1, len(outs)))(*[o.ctypes.data if o.size else None for o in outs])
        _check(load_library().gatk_pairhmm_download(self.hmm._h, self.handle, arr))
        return [o.reshape(nr, nh) for o, (nr, nh) in zip(outs, self.shapes)]

    # -- the steps right behind the path (genotyper/ReadLikelihoods.java), on the device results --
    def normalizeLikelihoods(self, bestToZero, maximumLikelihoodDifferenceCap, referenceAlleleIndex=None):
        """ReadLikelihoods.normalizeLikelihoods (:353) on every region, in place; referenceAlleleIndex: one
        index per region (-1: none), or None for no reference allele anywhere."""
        ref = None if referenceAlleleIndex is None else np.ascontiguousarray(referenceAlleleIndex, dtype=np.int32)
        assert ref is None or len(ref) == len(self.shapes)
        _check(load_library().gatk_pairhmm_normalize_resident(self.hmm._h, self.handle, None if ref is None else ref.ctypes.data,
                                                              1 if bestToZero else 0, maximumLikelihoodDifferenceCap))

    def poorlyModeledReads(self, maximumErrorPerBase):
        """Per region, uint8 flags of the reads ReadLikelihoods.filterPoorlyModeledReads (:1059) removes."""
        flags = [np.zeros(nr if nh else 0, dtype=np.uint8) for nr, nh in self.shapes]
        arr = (C.c_void_p * max(1, len(flags)))(*[f.ctypes.data if f.size else None for f in flags])
        _check(load_library().gatk_pairhmm_poorly_modeled_resident(self.hmm._h, self.handle, maximumErrorPerBase, arr))
        return flags

    def marginalize(self, oldToNewAlleleIndexMaps, newAlleleCounts):
        """ReadLikelihoods.marginalLikelihoods (:911) keeping every read: per region a [reads, new alleles] array."""
        maps = [np.ascontiguousarray(m, dtype=np.int32) for m in oldToNewAlleleIndexMaps]
        counts = np.ascontiguousarray(newAlleleCounts, dtype=np.int32)
        outs = [np.empty((nr if nh else 0, int(c)), dtype=np.float64) for (nr, nh), c in zip(self.shapes, counts)]
        marr = (C.c_void_p * max(1, len(maps)))(*[m.ctypes.data if m.size else None for m in maps])
        oarr = (C.c_void_p * max(1, len(outs)))(*[o.ctypes.data if o.size else None for o in outs])
        _check(load_library().gatk_pairhmm_marginalize_resident(self.hmm._h, self.handle, marr, counts.ctypes.data, oarr))
        return outs

    def device_output(self):
        p, n = C.c_void_p(), C.c_int64()
        _check(load_library().gatk_pairhmm_resident_device_output(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def free(self):
        if self.handle:
            _check(load_library().gatk_pairhmm_resident_free(self.hmm._h, self.handle))
            self.handle = None


class RegionResults(list):
    """Per-region [reads, haplotypes] arrays that are consecutive views of one buffer (`flat_buffer`)."""
    flat_buffer = None


class PreparedRegions:
    """A compute_regions call whose argument marshalling is done once: the ctypes views of the caller's arrays and
    one flat output buffer.  compute() is then the C-ABI call alone, as a host that already holds its arrays (the
    Java side) would make it."""

    def __init__(self, hmm, regions, flat=None):
        """flat: a float64 buffer of exactly the call's size to score into (e.g. this rank's slice of a shared-memory
        segment, sharding.SharedOutput), or None for a buffer of its own."""
        self.hmm, self.regions, self._keep = hmm, list(regions), []
        n = len(self.regions)
        self._cbs = (CBatch * max(1, n))(*[to_cbatch(b, self._keep) for b in self.regions])
        sizes = [b.num_reads * b.num_haps for b in self.regions]
        if flat is None:
            flat = np.full(int(sum(sizes)), np.nan, dtype=np.float64)
        if flat.dtype != np.float64 or flat.ndim != 1 or flat.size != int(sum(sizes)) or not flat.flags.c_contiguous:
            raise ValueError("flat must be a contiguous float64 vector with one entry per (read, haplotype) pair of the call")
        self.flat = flat
        self.outs = RegionResults()
        off = 0
        for b, size in zip(self.regions, sizes):
            self.outs.append(self.flat[off:off + size].reshape(b.num_reads, b.num_haps))
            off += size
        self.outs.flat_buffer = self.flat
        self._arr = (C.c_void_p * max(1, n))(*[o.ctypes.data if o.size else None for o in self.outs])

    def compute(self):
        _check(self.hmm._lib.gatk_pairhmm_compute_regions(self.hmm._h, len(self.regions), self._cbs, self._arr))
        return self.outs


class Ticket:
    """An outstanding gatk_pairhmm_submit call; the output arrays stay alive with it."""

    def __init__(self, hmm, ticket, outs):
        self.hmm, self.ticket, self.outs = hmm, ticket, outs

    def fetch(self):
        t, self.ticket = self.ticket, None
        if t is None:
            raise IllegalStateException("the ticket has been fetched already")
        _check(self.hmm._lib.gatk_pairhmm_fetch(self.hmm._h, t))
        return self.outs


class HipLoglessPairHMM:
    """PairHMM (PairHMM.java:46) implemented by the gfx950 engine."""

    def __init__(self, device=0):
        """device: a GPU index, or a list of them (a multi-device instance, gatk_pairhmm_create_multi), or "all"."""
        self._lib = load_library()
        h = C.c_void_p()
        if isinstance(device, (list, tuple)) or device == "all":
            devs = np.asarray([] if device == "all" else device, dtype=np.int32)
            _check(self._lib.gatk_pairhmm_create_multi(devs.ctypes.data if devs.size else None, int(devs.size), C.byref(h)))
        else:
            _check(self._lib.gatk_pairhmm_create(device, C.byref(h)))
        self._h = h
        self.mLikelihoodArray = None

    # -- PairHMM.java:104 --
    def initialize(self, readMaxLength, haplotypeMaxLength):
        _check(self._lib.gatk_pairhmm_initialize(self._h, readMaxLength, haplotypeMaxLength))

    # -- PairHMM.java:137 --
    def initializeRegion(self, haplotypes, perSampleReadList, readMaxLength, haplotypeMaxLength):
        """PairHMM.initialize(haplotypes, perSampleReadList, readMaxLength, haplotypeMaxLength): sends every sample's
        reads (perSampleReadList: dict sample -> list of (bases, quals, insQuals|None, delQuals|None)) and the region's
        haplotypes to the device in one pass and returns without waiting; the computeLikelihoods calls that follow
        find their likelihoods there."""
        batches = [pack([(r[0], r[1], r[2], r[3], None) for r in reads], haplotypes) for reads in perSampleReadList.values()]
        for b in batches:
            b.gcp = None  # the GCP map only comes with computeLikelihoods (PairHMM.java:190-192)
        self.initialize_region_batches(batches, readMaxLength, haplotypeMaxLength)

    def initialize_region_batches(self, batches, readMaxLength, haplotypeMaxLength):
        keep = []
        cbs = (CBatch * max(1, len(batches)))(*[to_cbatch(b, keep) for b in batches])
        _check(self._lib.gatk_pairhmm_initialize_region(self._h, len(batches), cbs, readMaxLength, haplotypeMaxLength))

    # -- submit / fetch: compute_regions in two halves --
    def submit(self, regions):
        """gatk_pairhmm_submit: returns a ticket object at once; ticket.fetch() returns the per-region arrays."""
        keep = []
        cbs = (CBatch * max(1, len(regions)))(*[to_cbatch(b, keep) for b in regions])
        outs = [np.full((b.num_reads, b.num_haps), np.nan, dtype=np.float64) for b in regions]
        arr = (C.c_void_p * max(1, len(regions)))(*[o.ctypes.data if o.size else None for o in outs])
        t = C.c_int64()
        _check(self._lib.gatk_pairhmm_submit(self._h, len(regions), cbs, arr, C.byref(t)))
        return Ticket(self, t.value, outs)

    # -- PairHMM.java:190 --
    def computeLikelihoods(self, likelihoods: ReadLikelihoodsMatrix, processedReads, gcp=None):
        """processedReads: list of (bases, quals, insQuals|None, delQuals|None); gcp: dict read-index ->
        byte array, or None for the constant 10.  Fills `likelihoods` (set(a, r, lk)) and
        mLikelihoodArray (read-major)."""
        if len(processedReads) == 0:
            return
        reads = [(r[0], r[1], r[2], r[3], None if gcp is None else gcp[k]) for k, r in enumerate(processedReads)]
        out = self.compute_batch(pack(reads, likelihoods.alleles()))
        likelihoods.values[:, :] = out.T
        self.mLikelihoodArray = out.reshape(-1).copy()

    def compute_batch(self, b: Batch) -> np.ndarray:
        keep = []
        cb = to_cbatch(b, keep)
        out = np.full((b.num_reads, b.num_haps), np.nan, dtype=np.float64)
        _check(self._lib.gatk_pairhmm_compute_likelihoods(self._h, C.byref(cb), out.ctypes.data if out.size else None))
        return out

    def compute_regions(self, regions):
        keep = []
        cbs = (CBatch * max(1, len(regions)))(*[to_cbatch(b, keep) for b in regions])
        outs = [np.full((b.num_reads, b.num_haps), np.nan, dtype=np.float64) for b in regions]
        arr = (C.c_void_p * max(1, len(regions)))(*[o.ctypes.data if o.size else None for o in outs])
        _check(self._lib.gatk_pairhmm_compute_regions(self._h, len(regions), cbs, arr))
        return outs

    def compute_device_batch_noalloc(self, cbatch: CBatch, read_max_length, haplotype_max_length, out_ptr, status_ptr=None, stream=None):
        """gatk_pairhmm_compute_likelihoods_noalloc: `cbatch` holds DEVICE pointers, out_ptr / status_ptr are device
        addresses, stream a hipStream_t address (None: the default stream).  Enqueues and returns; allocates nothing."""
        _check(self._lib.gatk_pairhmm_compute_likelihoods_noalloc(self._h, stream, C.byref(cbatch), read_max_length,
                                                                  haplotype_max_length, out_ptr, status_ptr))

    def prepare_regions(self, regions, flat=None) -> PreparedRegions:
        return PreparedRegions(self, regions, flat)

    def upload(self, regions) -> Resident:
        keep = []
        cbs = (CBatch * max(1, len(regions)))(*[to_cbatch(b, keep) for b in regions])
        r = C.c_void_p()
        _check(self._lib.gatk_pairhmm_upload(self._h, len(regions), cbs, C.byref(r)))
        return Resident(self, r, [(b.num_reads, b.num_haps if b.num_reads else 0) for b in regions])

    # -- PairHMM.java:313 (single pair through the checks of :262-306) --
    def computeReadLikelihoodGivenHaplotypeLog10(self, haplotypeBases, readBases, readQuals, insertionGOP, deletionGOP, overallGCP):
        hb, rb = np.frombuffer(bytes(haplotypeBases), np.uint8), np.frombuffer(bytes(readBases), np.uint8)
        arrs = [np.ascontiguousarray(x, dtype=np.uint8) for x in (readQuals, insertionGOP, deletionGOP, overallGCP)]
        for name, a in zip(("read quals", "read insertion quals", "read deletion quals", "overall GCP"), arrs):
            if len(a) != len(rb):  # PairHMM.java:276-279
                raise IllegalArgumentException(f"Read bases and {name} aren't the same size: {len(rb)} vs {len(a)}")
        out = C.c_double()
        dummy = np.zeros(1, np.uint8)
        g = lambda a: a.ctypes.data if a.size else dummy.ctypes.data
        _check(self._lib.gatk_pairhmm_compute_pair(self._h, g(hb), len(hb), g(rb), g(arrs[0]), g(arrs[1]), g(arrs[2]), g(arrs[3]), len(rb), C.byref(out)))
        return out.value

    # -- BatchPairHMM.java:33-40 --
    def batchAdd(self, haplotypes, readBases, readQuals, insertionGOP, deletionGOP, overallGCP):
        haps = [np.frombuffer(bytes(h), np.uint8) for h in haplotypes]
        hp = (C.c_void_p * max(1, len(haps)))(*[h.ctypes.data for h in haps])
        hl = np.asarray([len(h) for h in haps], dtype=np.int32)
        arrs = [np.ascontiguousarray(np.frombuffer(bytes(readBases), np.uint8))] + [np.ascontiguousarray(x, dtype=np.uint8) for x in (readQuals, insertionGOP, deletionGOP, overallGCP)]
        _check(self._lib.gatk_pairhmm_batch_add(self._h, len(haps), hp, hl.ctypes.data, *[a.ctypes.data for a in arrs], len(arrs[0])))

    def batchGetResult(self):
        p, n = C.c_void_p(), C.c_int64()
        _check(self._lib.gatk_pairhmm_batch_get_result(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.float64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n.value,)).copy()

    # -- PairHMM.java:360 --
    def getLikelihoodArray(self):
        p, n = C.c_void_p(), C.c_int64()
        _check(self._lib.gatk_pairhmm_get_likelihood_array(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.float64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n.value,)).copy()

    # -- PairHMM.java:124 --
    def finalizeRegion(self):
        _check(self._lib.gatk_pairhmm_finalize_region(self._h))

    # -- PairHMM.java:364 --
    def close(self):
        if self._h:
            _check(self._lib.gatk_pairhmm_close(self._h))
            self._h = None

    def set_option(self, name, value):
        _check(self._lib.gatk_pairhmm_set_option(self._h, name.encode(), float(value)))

    def stats(self):
        s = CStats()
        _check(self._lib.gatk_pairhmm_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in CStats._fields_}

    def stream(self):
        p = C.c_void_p()
        _check(self._lib.gatk_pairhmm_get_stream(self._h, C.byref(p)))
        return p.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def setNumberOfThreads(numThreads):  # PairHMM.java:350
    _check(load_library().gatk_pairhmm_set_number_of_threads(int(numThreads)))


def device_count():
    n = C.c_int32()
    _check(load_library().gatk_pairhmm_device_count(C.byref(n)))
    return n.value
