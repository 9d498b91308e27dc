/*
 * C-ABI of the MI355X (gfx950) PairHMM engine: libgatk_pairhmm_hip.so.
 *
 * These entry points are what a native PairHMM implementation of GATK 3.x binds: one
 * opaque handle per Java PairHMM instance, the batch call that replaces
 * PairHMM.computeLikelihoods, and the region / close hooks the reference reserves for JNI
 * back ends.  Citations are file:line under
 *   public/gatk-utils/src/main/java/org/broadinstitute/gatk/utils/pairhmm/
 * of the reference tree.  Plain pointers and sizes only; all byte arrays are caller-owned
 * and are not retained after a call returns (except through the explicit resident API).
 *
 * Every function returns one of the GATK_PAIRHMM_* codes; the three *_ILLEGAL_* /
 * *_INDEX_* codes stand for the Java exception the reference would have thrown at the same
 * point.  gatk_pairhmm_last_error() returns the message for the calling thread.
 *
 * Threading: distinct handles may be used from distinct threads concurrently (the reference
 * creates one PairHMM per map thread, PairHMM.java:297).  One handle must not be used from
 * two threads at once.
 */
#ifndef GATK_PAIRHMM_H
#define GATK_PAIRHMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GATK_PAIRHMM_OK 0
#define GATK_PAIRHMM_ILLEGAL_ARGUMENT 1    /* IllegalArgumentException (PairHMM.java:105-106,272-279; PairHMMModel.java:136-138) */
#define GATK_PAIRHMM_ILLEGAL_STATE 2       /* IllegalStateException (PairHMM.java:271,292-295) */
#define GATK_PAIRHMM_INDEX_OUT_OF_BOUNDS 3 /* ArrayIndexOutOfBoundsException: base quality 255 (QualityUtils.java:63,173) */
#define GATK_PAIRHMM_DEVICE_ERROR 4        /* a HIP runtime call failed, or the device is absent or not a gfx950 */
#define GATK_PAIRHMM_OUT_OF_MEMORY 5

typedef struct gatk_pairhmm gatk_pairhmm;                   /* one PairHMM instance bound to one GPU */
typedef struct gatk_pairhmm_resident gatk_pairhmm_resident; /* a set of region batches kept in HBM */

/*
 * One computeLikelihoods call: the reads of one sample in one active region against the
 * haplotypes of that region (PairHMM.java:190-214).  Each of the five per-read arrays is the
 * concatenation of the reads' byte[] (getReadBases, getBaseQualities,
 * getBaseInsertionQualities, getBaseDeletionQualities, the GCP map entry); read r occupies
 * [read_offsets[r], read_offsets[r+1]).  Haplotype a occupies hap_bases[hap_offsets[a] ..
 * hap_offsets[a+1]).  Quality bytes are used unsigned; insertion/deletion/GCP values >= 128
 * are the Java "negative byte" error and base quality 255 is the out-of-bounds error.
 *
 * ins_gop and del_gop may be NULL: the read then has the flat Q45 GATKSAMRecord.getBaseInsertionQualities /
 * getBaseDeletionQualities give a read without them (sam/GATKSAMRecord.java:61,298-322).  gcp may be NULL:
 * 10 for every base, the constant HaplotypeCaller passes (upstream of the reference tree).  An array no
 * region of a call supplies is neither staged nor copied.  read_mapq (one byte per read) may be NULL; when
 * given, a read's base qualities are capped by it, and with the handle option "squash_base_quals" base
 * qualities under BASE_QUALITY_SCORE_THRESHOLD = 18 (pairhmm/PairHMM.java:54) count as
 * MIN_USABLE_Q_SCORE = 6 (QualityUtils.java:57) - the two steps the upstream likelihood engine applies to
 * a read before the PairHMM sees it.
 *
 * Limits, each an ILLEGAL_ARGUMENT ("... is too long", the text of PairHMM.java:273-279): a read of more than 32766
 * bases, a haplotype of more than 16000.  A haplotype of no bases is rejected too (the reference's matrix has no
 * column for it).  Reads of up to 1022 bases share wavefronts in the fp32 pass (fp64 for the pairs it cannot hold);
 * a longer read has more rows than one wavefront carries, and each of its pairs is scored in fp64, 512 rows at a
 * time, by one to eight wavefronts - the same results, at a fraction of the rate.  A read of no bases scores log10(0) against
 * every haplotype.
 */
typedef struct gatk_pairhmm_batch {
    int32_t num_reads;
    int32_t num_haps;
    const uint8_t* read_bases;
    const uint8_t* read_quals;
    const uint8_t* ins_gop;
    const uint8_t* del_gop;
    const uint8_t* gcp;
    const int32_t* read_offsets; /* num_reads + 1 entries */
    const uint8_t* hap_bases;
    const int32_t* hap_offsets;  /* num_haps + 1 entries */
    const uint8_t* read_mapq;    /* num_reads entries, or NULL */
} gatk_pairhmm_batch;

/* Counters of the last compute call on a handle. */
typedef struct gatk_pairhmm_stats {
    int64_t pairs;          /* (read, haplotype) pairs scored */
    int64_t cells;          /* sum of readLen * hapLen over those pairs */
    int64_t wave_tasks;     /* wavefront work items launched in fp32 */
    int64_t rerun_tasks;    /* wavefront work items of the fp64 pass */
    int64_t rerun_pairs;    /* pairs recomputed in fp64 because their fp32 sum fell under the threshold */
    double fp32_kernel_ms;  /* device time of the fp32 forward kernels (HIP events), 0 if not timed */
    double fp64_kernel_ms;  /* device time of the fp64 rerun kernels (HIP events), 0 if not timed */
    double host_plan_ms;    /* host time spent checking, planning and packing the call */
    double host_upload_ms;  /* host time of the single host-to-device copy (including the wait for it) */
    double aux_kernel_ms;   /* device time of the small kernels around them: validation, expansion of reads and
                               haplotypes, conversion of the fp32 sums to log10 */
    int64_t timed_passes;   /* passes the three *_kernel_ms sums cover: every compute_resident since the last wait */
    /* the forward launch that held most of the call's work items (of the last part, for a call cut into parts) */
    int32_t main_rows_per_lane;  /* read rows per lane of that launch (the K of the kernel instantiation) */
    int32_t main_emit_mode;      /* 0 compares bytes, 1 same with haplotype N, 2 emission table A/C/G/T, 3 table with N */
    int32_t main_stream_pass;    /* 1: the fp32 stream pass (haplotype runs), 0: the one-haplotype pass */
    int32_t main_fp64;           /* 1: everything was scored in fp64 (force_fp64) */
    /* Rows the forward launches of the call offered (64 lanes x rows per lane, per wave-group and haplotype) and
     * the rows among them that hold a base of a read: their ratio is how full the lanes ran (a read also takes a
     * pad row and a sum row, so 151-base reads at 10 rows per lane reach 0.944). */
    int64_t lane_rows_allocated;
    int64_t lane_rows_used;
    /* Since the process started, on this handle's device: device passes run for calls that share the device's
     * dispatcher (submit / fetch, regions sent ahead, "combine_calls") and the calls those passes carried. */
    int64_t shared_passes;
    int64_t shared_requests;
    /* On this handle: computeLikelihoods calls answered from a region sent ahead with initialize_region, and calls
     * made while such a region was held that matched none of its samples. */
    int64_t region_hits;
    int64_t region_misses;
} gatk_pairhmm_stats;

/* Constructor of a PairHMM instance on GPU `device` (replaces the JNI library's one-time
 * set-up, recalled upstream as jniInitializeClassFieldsAndMachineMask).  Fails with
 * GATK_PAIRHMM_DEVICE_ERROR when the device is absent or is not a gfx950: there is no CPU fallback. */
int gatk_pairhmm_create(int device, gatk_pairhmm** out);

/* One PairHMM instance over several GPUs of the node, for a host that is a single process (one JVM whose -nct map
 * threads, nanoScheduler/NanoScheduler.java:112, share the node): the region-level hook the reference reserves
 * for native back ends (PairHMM.java:129-139, "transfer all data once to the native code") hands over many
 * regions, and regions are independent (traversals/TraverseActiveRegions.java:705-709).  devices[0 .. num_devices)
 * lists the GPUs (a device may be listed more than once: an engine each); NULL or num_devices <= 0 = every
 * visible device.  The handle owns one single-device engine and one host thread per entry.  compute_regions,
 * compute_likelihoods and the calls built on them shard their regions over the engines, longest processing
 * time first by cell count; a region worth more than half a device's share is cut by read ranges; each engine
 * writes straight into the caller's likelihoods[k].  No gather copy and nothing between GPUs.  A call of fewer
 * than "multi_min_pairs" pairs (option, default 8192) goes to one engine, the engines taking such calls in
 * turn.  set_option and initialize apply to every engine; stats add the counts up and report the slowest
 * device's times.  The resident interface, the allocation-free entry point and get_stream are served by the
 * first device's engine.  The handle option "combine_calls" has no effect on such a handle.  Same threading
 * rule: one handle, one thread at a time. */
int gatk_pairhmm_create_multi(const int32_t* devices, int32_t num_devices, gatk_pairhmm** out);

/* PairHMM.initialize(int readMaxLength, int haplotypeMaxLength), PairHMM.java:104-119.
 * ILLEGAL_ARGUMENT when either is <= 0.  Optional, because a compute call whose longest read or longest
 * haplotype exceeds the current maxima re-initialises the instance with that call's two maxima, as
 * PairHMM.java:198-201 does (so the other maximum may shrink; compute_pair checks against them). */
int gatk_pairhmm_initialize(gatk_pairhmm* h, int32_t read_max_length, int32_t haplotype_max_length);

/* PairHMM.initialize(List<Haplotype> haplotypes, Map<String, List<GATKSAMRecord>> perSampleReadList, int readMaxLength,
 * int haplotypeMaxLength), PairHMM.java:129-139: "used by the JNI implementations to transfer all data once to the
 * native code".  per_sample[s] holds sample s's reads against the region's haplotypes (every entry carries the
 * haplotypes; gcp is normally NULL here, the GCP map only arrives with computeLikelihoods, PairHMM.java:190-192, and
 * NULL stands for the constant 10 HaplotypeCaller passes).  The call does what initialize(int, int) does, copies the
 * bytes, starts ONE device pass over all samples and returns without waiting for it.  A later
 * gatk_pairhmm_compute_likelihoods on this handle whose batch is byte for byte one of these samples (lengths, bases,
 * qualities, haplotypes; an array left out on one side equals an array of its default on the other) waits for that
 * pass and copies the sample's likelihoods out; any other batch (another sample, a GCP that is not the default) is
 * scored the ordinary way, and so is every sample when the pass reported an input error, so that the reference's
 * exception comes from the computeLikelihoods call whose reads raise it.  Shape errors (PairHMM.java:105-106, null or
 * over-long arrays) are returned here.  The region is forgotten at gatk_pairhmm_finalize_region, at the next
 * initialize_region and at close. */
int gatk_pairhmm_initialize_region(gatk_pairhmm* h, int32_t num_samples, const gatk_pairhmm_batch* per_sample,
                                   int32_t read_max_length, int32_t haplotype_max_length);

/* gatk_pairhmm_compute_regions in two halves, for a host thread that keeps region n + 1 on the device while it consumes
 * region n (one map thread of nanoScheduler/NanoScheduler.java:112 with two PairHMM.computeLikelihoods calls,
 * PairHMM.java:190-236, in flight).  submit checks the shapes (their errors are returned here), copies every input
 * byte (the caller's input arrays are free again when it returns), hands the regions to the device's dispatcher and
 * returns a ticket without waiting.  fetch(ticket) returns when likelihoods[k] (which must stay valid until then;
 * another thread of the process may be the one that writes them) hold the results, with the code of the reference's
 * exception if the data raised one (negative gap qualities, base quality 255), and forgets the ticket.  Any number
 * of tickets may be outstanding on a handle, fetched in any order; close drops the ones never fetched.
 * The dispatcher is per device and shared by all handles: requests of several threads that are waiting when an
 * engine falls free go through ONE device pass, so region-sized calls fill the device the better the more callers
 * there are.  A call of more than "pipeline_pairs" pairs, or on a handle with non-default result options, is scored
 * before submit returns.  On a multi-device handle the requests go to the devices in turn. */
int gatk_pairhmm_submit(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions, double* const* likelihoods,
                        int64_t* ticket);
int gatk_pairhmm_fetch(gatk_pairhmm* h, int64_t ticket);

/* PairHMM.computeLikelihoods, PairHMM.java:190-236 (upstream JNI: jniComputeLikelihoods).
 * likelihoods[r * num_haps + a] = log10 P(read r | haplotype a): the read-major order of
 * mLikelihoodArray (PairHMM.java:206,225).  num_reads == 0 is a no-op (PairHMM.java:193).
 * Values are <= 0, never NaN, and may be -infinity (PairHMM.java:292-295). */
int gatk_pairhmm_compute_likelihoods(gatk_pairhmm* h, const gatk_pairhmm_batch* batch, double* likelihoods);

/* The same call with the per-read / per-haplotype byte[] passed as pointer arrays, the shape
 * the upstream JNI holder classes (JNIReadDataHolderClass / JNIHaplotypeDataHolderClass) have. */
int gatk_pairhmm_compute_likelihoods_ptrs(gatk_pairhmm* h, int32_t num_reads, int32_t num_haps,
                                          const uint8_t* const* read_bases, const uint8_t* const* read_quals,
                                          const uint8_t* const* ins_gop, const uint8_t* const* del_gop,
                                          const uint8_t* const* gcp, const int32_t* read_lengths,
                                          const uint8_t* const* hap_bases, const int32_t* hap_lengths,
                                          double* likelihoods);

/* Many computeLikelihoods calls (regions x samples) in one device pass: what the concurrent
 * map threads of the reference (nanoScheduler/NanoScheduler.java:112) would each submit, batched
 * so that small regions fill the GPU.  likelihoods[k] receives region k's read-major array.
 * When the call fails (a malformed region, an input error in a read), the arrays of regions in front of the
 * failing one may already hold their results: a call of many regions is checked and scored part by part. */
int gatk_pairhmm_compute_regions(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions,
                                 double* const* likelihoods);

/* PairHMM.subComputeReadLikelihoodGivenHaplotypeLog10 for a single pair (PairHMM.java:313-321)
 * through the checks of computeReadLikelihoodGivenHaplotypeLog10 (PairHMM.java:262-306). */
int gatk_pairhmm_compute_pair(gatk_pairhmm* h, const uint8_t* hap_bases, int32_t hap_length,
                              const uint8_t* read_bases, const uint8_t* read_quals, const uint8_t* ins_gop,
                              const uint8_t* del_gop, const uint8_t* gcp, int32_t read_length, double* likelihood);

/* BatchPairHMM.batchAdd / batchGetResult, BatchPairHMM.java:33-40: queue one read against a
 * haplotype list, then fetch all queued results (haplotype-minor per queued read) and clear the
 * queue.  *results points into handle-owned memory valid until the next call on the handle. */
int gatk_pairhmm_batch_add(gatk_pairhmm* h, int32_t num_haps, const uint8_t* const* hap_bases,
                           const int32_t* hap_lengths, const uint8_t* read_bases, const uint8_t* read_quals,
                           const uint8_t* ins_gop, const uint8_t* del_gop, const uint8_t* gcp, int32_t read_length);
int gatk_pairhmm_batch_get_result(gatk_pairhmm* h, const double** results, int64_t* count);

/* PairHMM.getLikelihoodArray(), PairHMM.java:360: the read-major array of the last
 * compute_likelihoods call; handle-owned. */
int gatk_pairhmm_get_likelihood_array(gatk_pairhmm* h, const double** array, int64_t* count);

/* PairHMM.finalizeRegion(), PairHMM.java:121-127: end of a region; drops per-region state (the BatchPairHMM queue, the
 * region sent ahead with gatk_pairhmm_initialize_region and its results). */
int gatk_pairhmm_finalize_region(gatk_pairhmm* h);

/* PairHMM.close(), PairHMM.java:364 (upstream JNI: jniClose): frees the handle and its
 * device memory. */
int gatk_pairhmm_close(gatk_pairhmm* h);

/* PairHMM.setNumberOfThreads, PairHMM.java:350-355: profiling (HIP-event kernel timing in the
 * stats) is kept only when num_threads == 1. */
int gatk_pairhmm_set_number_of_threads(int32_t num_threads);

int gatk_pairhmm_get_stats(gatk_pairhmm* h, gatk_pairhmm_stats* out);
const char* gatk_pairhmm_last_error(void);
const char* gatk_pairhmm_version(void);
int gatk_pairhmm_device_count(int32_t* count);

/* Engine tunables (all optional).  Planning options take effect at the next upload or host-pointer call.
 *   "fp32_threshold"      default 1e-28: a pair whose scaled fp32 sum falls below it is recomputed in fp64.
 *   "force_fp64"          1 = score everything in fp64 (batches planned while it is set take at most 8 rows per
 *                         lane where the reads fit, the most the fp64 pass holds in registers).
 *   "force_rows_per_lane" 0 = automatic.
 *   "emit_mode"           -1 = automatic, 0 = always compare bases instead of using the per-wave emission table.
 *   "timing"              1 = fill the *_kernel_ms stats.
 *   "stream_mode"         1 = a wave scores a run of haplotypes back to back; 0 = one haplotype per work item;
 *                         -1 (default) = runs, except in a call with fewer pairs than two per wave slot of the
 *                         device, where no wave would get a run and the one-haplotype pass saves a kernel launch.
 *   "max_stream_haps"     longest run, 1..16.
 *   "stream_run_divisor"  default 2: a run holds the work items still to hand out divided by this many times the
 *                         device's wave slots, so that runs shorten towards the end of a launch.
 *   "group_rows_allowed"  bit k set = k rows per lane may be chosen for a wave-group of a large call (fewer values in
 *                         use = fewer, larger launches at lower lane use); 0 (default) = every value.
 *   "squash_base_quals"   1 = base qualities under 18 count as 6, see gatk_pairhmm_batch.
 *   "do_not_use_tristate_correction"  1 = PairHMM.doNotUseTristateCorrection() (PairHMM.java:83-84, "only used for
 *                         debugging purposes"): the prior of a mismatching base is the error probability itself, not
 *                         a third of it (Log10PairHMM.java:157).
 *   "pipeline_pairs"      default 98304: a compute call with more pairs is cut, between regions and inside a
 *                         large region between reads, into parts of about that many pairs, and one part is
 *                         planned and copied while the kernels of the parts before it run; 0 = never cut.
 *   "pipeline_cells_per_pair" default 80000: in that count a region of long reads against long haplotypes (more than
 *                         100 000 cells per pair) stands for its cells divided by this instead of its pairs, so that
 *                         such a call is cut into parts of about the same device time and the fp64 pair pass of
 *                         one part runs beside the fp32 pass of the next; 0 = count pairs only.
 *   "pipeline_ramp"       default 1: the first parts are smaller, so that the device starts early: "pipeline_ramp_steps"
 *                         (default 3) parts of an eighth, a quarter and a half of a full part; a call of 30 000 pairs to
 *                         three parts' worth is cut in four parts of 1/8, 1/4, 5/16 and 5/16 of its pairs.
 *   "pipeline_depth"      default 3: parts of a cut call that are in flight or being planned at once, each with
 *                         buffers and a stream of its own; 2 makes the host wait for part p before it plans
 *                         part p + 2.
 *   "pipeline_min_pairs"  default 6144: a compute_regions call of less than two parts' worth of pairs that has
 *                         the device to itself is cut in two halves of at least this many pairs, so that
 *                         planning the second and fetching the first hide behind kernels.
 *   "per_group_factor"    default 32: a call, or a part of a cut call, with fewer pairs than this many per wave
 *                         slot uses one rows-per-lane value throughout, so that it costs one forward launch; a
 *                         larger one chooses the value per wave-group.
 *   "combine_calls"       default 0; 1 = gatk_pairhmm_compute_likelihoods calls too small to fill the device,
 *                         made at the same time by several threads on their own handles, go through the device's
 *                         dispatcher (see gatk_pairhmm_submit) as a submit and a fetch in one: a call that finds an
 *                         engine free runs alone, calls that find none share the next pass; same results and
 *                         exceptions, per caller; only on handles whose other options are the defaults,
 *                         squash_base_quals aside.
 *   "kernel_upload_bytes" default 1048576: an upload of at most that many bytes is copied out of the pinned
 *                         staging buffer by a kernel on the compute stream instead of by the DMA engine, whose
 *                         hand-over to the first kernel costs more than the copy of a region-sized call.
 *   "kernel_download_bytes" default 524288: results of at most that many bytes (counters included) come back through
 *                         the same kind of kernel, device to pinned host memory: the runtime's copy starts 5-6 us
 *                         after the kernel in front of it.
 *   "signal_download_bytes" default 65536: results of at most that many bytes come back through one block that,
 *                         after its stores, raises a flag in pinned host memory which the calling thread polls (for
 *                         2 ms at most, then it waits through the runtime): the caller sees a region-sized result
 *                         about 5 us earlier than through an event.  0 turns it off.
 *   "wave_slots"         waves the planner assumes the device holds at once, default 12 per CU (occupancy
 *                         experiments).
 *   "stream_lds_floor"    least bytes of LDS a wave of the fp32 stream pass asks for; a larger value lowers the
 *                         number of waves a CU holds (occupancy experiments).
 *   "multi_min_pairs"     multi-device handles only, default 8192: a call with fewer pairs goes to one engine on the
 *                         calling thread.
 *   "debug_long_stall"    tests only: 1 = in the pass for reads of more than 1022 bases the first 512 rows of every
 *                         pair never announce their progress, so that the wavefront behind them runs into the limit of
 *                         its wait (about half a second) and the call returns GATK_PAIRHMM_DEVICE_ERROR. */
int gatk_pairhmm_set_option(gatk_pairhmm* h, const char* name, double value);

/*
 * Resident API: keep region batches in HBM across calls, score them without touching host
 * memory, and fetch results separately.  This is the path whose device time is the GCUPS
 * figure (inputs already resident when the clock starts).
 *   upload   : validate shapes, plan wave tasks, copy bytes and plan to the device
 *   compute  : enqueue validation + forward kernels (+ fp64 rerun) on the handle's stream;
 *              returns without waiting unless sync != 0
 *   wait     : wait for the enqueued passes and check the device-side error word
 *   download : wait, check the device-side error word, copy region k's likelihoods out
 */
int gatk_pairhmm_upload(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions,
                        gatk_pairhmm_resident** out);
int gatk_pairhmm_compute_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r, int32_t sync);
/* Wait for every pass enqueued on r, report device-side input errors of the last one, refresh the stats. */
int gatk_pairhmm_wait_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r);
int gatk_pairhmm_download(gatk_pairhmm* h, gatk_pairhmm_resident* r, double* const* likelihoods);
int gatk_pairhmm_resident_free(gatk_pairhmm* h, gatk_pairhmm_resident* r);
/* The HIP stream (hipStream_t) the handle enqueues on, for callers that time it with events. */
int gatk_pairhmm_get_stream(gatk_pairhmm* h, void** stream);
/*
 * The steps right behind the path, applied to the likelihoods of a resident batch while they are still on
 * the device (after compute_resident, before or after download).  Matrices are per region; the engine's
 * layout is read-major (values[r * alleles + a]), the Java's is values[a][r].  There is no <NON_REF>
 * allele on this path.
 *
 * normalize      : ReadLikelihoods.normalizeLikelihoods(bestToZero, cap), genotyper/ReadLikelihoods.java:353-404,
 *                  in place; reference_allele_index[k] is region k's reference allele or -1 (NULL: -1
 *                  everywhere).  IllegalArgumentException for a cap that is not negative.
 * poorly_modeled : flags[k][r] = 1 for the reads ReadLikelihoods.filterPoorlyModeledReads(maximumErrorPerBase)
 *                  removes, :1059-1093 (the reads stay in the batch: dropping them is the caller's bookkeeping).
 * marginalize    : ReadLikelihoods.marginalLikelihoods keeping every read, :911-939:
 *                  new_likelihoods[k][r * new_allele_count[k] + n] = max over the region's alleles a with
 *                  old_to_new_allele_index[k][a] == n (-1: allele dropped), -infinity when there is none.
 */
int gatk_pairhmm_normalize_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r, const int32_t* reference_allele_index,
                                    int32_t best_to_zero, double maximum_likelihood_difference_cap);
int gatk_pairhmm_poorly_modeled_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r, double maximum_error_per_base,
                                         uint8_t* const* flags);
int gatk_pairhmm_marginalize_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r, const int32_t* const* old_to_new_allele_index,
                                      const int32_t* new_allele_count, double* const* new_likelihoods);

/*
 * gatk_pairhmm_compute_likelihoods for callers that cannot allocate.  No counterpart in the reference: it is for
 * callers that may not call hipMalloc or wait on the device - code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  One computeLikelihoods call (one region) whose inputs and outputs are already
 * in device memory: every pointer of device_batch, device_likelihoods and device_status is a DEVICE pointer on the
 * handle's GPU (read_mapq may be NULL; ins_gop, del_gop and gcp may not: no defaults are filled in here).  The
 * call enqueues its kernels on `stream` (a hipStream_t; NULL = the default stream) and returns without waiting;
 * likelihoods[r * num_haps + a] is complete once the stream has passed them.  It calls no hipMalloc, hipFree or
 * other allocating routine: descriptors, the haplotype copy and the fp64 rerun list live in a fixed 1.7 MB
 * array in each GPU's memory that the library holds from load to exit.
 *
 * The host cannot see lengths that live in device memory, so the caller declares them: no read is longer than
 * read_max_length (<= 1022), no haplotype longer than haplotype_max_length (<= 4096).  Limits per call:
 * 4096 reads, 256 haplotypes, 262144 pairs, 512 KiB of haplotype bases (each haplotype rounded up to 16);
 * a call beyond them fails with ILLEGAL_ARGUMENT before anything is enqueued.
 *
 * Errors that only the device can see come back in *device_status (may be NULL: then they are lost), valid
 * once the stream has passed the call: bit 0 = negative insertion / deletion / GCP byte
 * (IllegalArgumentException, PairHMMModel.java:136-138), bit 1 = base quality 255
 * (ArrayIndexOutOfBoundsException, QualityUtils.java:173), bit 2 = a length contradicts the declared maxima,
 * an empty haplotype, decreasing offsets or too many haplotype bytes: nothing was scored and the likelihoods
 * are unspecified.
 *
 * Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams, two threads) race.
 * Calls on one stream, or otherwise ordered one after the other, are safe.  Handle options honoured:
 * fp32_threshold, squash_base_quals.
 */
int gatk_pairhmm_compute_likelihoods_noalloc(gatk_pairhmm* h, void* stream, const gatk_pairhmm_batch* device_batch,
                                             int32_t read_max_length, int32_t haplotype_max_length,
                                             double* device_likelihoods, int32_t* device_status);

/* Device pointer to the resident likelihood array (all regions concatenated, region-major). */
int gatk_pairhmm_resident_device_output(gatk_pairhmm_resident* r, void** device_ptr, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* GATK_PAIRHMM_H */
