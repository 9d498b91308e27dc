// Host-side internals of the PairHMM engine shared by its translation units: the handle and resident-batch
// structures behind the opaque types of include/gatk_pairhmm.h, device and pinned buffers, error plumbing, and
// the planner / sequencing functions the entry points are built from.  Not installed; the C-ABI is the interface.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <emmintrin.h>

#include <array>
#include <functional>
#include <thread>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/gatk_pairhmm.h"
#include "pairhmm_device.h"
#include "pairhmm_tables.h"

namespace pairhmm_hip {
hipError_t launch_forward(const DeviceBatch& B, int rows_per_lane, int emit_mode, bool fp64, bool list_mode,
                          int blocks, hipStream_t stream);
int lds_bytes_per_wave(int rows_per_lane, int emit_mode, int max_hap_len);
int stream_lds_bytes_per_wave(int rows_per_lane, int emit_mode);
hipError_t launch_stream(const DeviceBatch& B, int rows_per_lane, int emit_mode, hipStream_t stream);
hipError_t launch_finish_stream(const DeviceBatch& B, hipStream_t stream);
hipError_t launch_pairs(const DeviceBatch& B, int rows_per_lane, int waves, hipStream_t stream);
hipError_t launch_long(const DeviceBatch& B, int blocks, int waves_per_block, int max_hap_len, hipStream_t stream);
bool long_table_fits(int waves_per_block);
int long_lds_bytes(int waves_per_block, int max_hap_len, int table_codes);
hipError_t launch_normalize_reads(const DeviceBatch& B, const int32_t* reference_allele_index, int best_to_zero, double cap, hipStream_t stream);
hipError_t launch_flag_poorly_modelled_reads(const DeviceBatch& B, double maximum_error_per_base, uint8_t* flags, hipStream_t stream);
hipError_t launch_marginalize_reads(const DeviceBatch& B, const int32_t* old_to_new, const int64_t* new_base, const int32_t* new_count,
                                    double* new_out, hipStream_t stream);
hipError_t launch_prepare(const DeviceBatch& B, bool expand_rows, bool stream, hipStream_t s);
hipError_t launch_upload(const void* pinned_host, void* device, size_t bytes, hipStream_t s);
hipError_t launch_download(void* device, void* pinned_host, size_t bytes, size_t clear_bytes, hipStream_t s);
hipError_t launch_download_and_signal(void* device, void* pinned_host, size_t bytes, size_t clear_bytes, unsigned* pinned_flag,
                                      unsigned seq, hipStream_t s);
}  // namespace pairhmm_hip

namespace pairhmm_host {
using namespace pairhmm_hip;

// Stores the calling thread's error message (gatk_pairhmm_last_error) and returns `code`.
int fail(int code, const std::string& msg);
const std::string& last_error();
// The handler of every entry point: no C++ exception (a failed host allocation, a thread that cannot be
// started) crosses the C boundary into a caller that cannot unwind it.
int on_exception() noexcept;

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? GATK_PAIRHMM_OUT_OF_MEMORY : GATK_PAIRHMM_DEVICE_ERROR, \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                             \
    } while (0)

// Grow-only device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr; cap = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Grow-only pinned host buffer (staging for the one H2D and the one D2H copy of a call).
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    // coherent: fine-grained whatever the process environment says about the default, for memory the host reads
    // while the kernel that writes it is still running (download_and_signal)
    hipError_t ensure(size_t bytes, bool coherent = false) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipHostFree(p);
            p = nullptr; cap = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipHostMalloc(&p, want, coherent ? hipHostMallocCoherent : hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    // the same, keeping the first `keep` bytes of what the buffer holds
    hipError_t ensure_keep(size_t bytes, size_t keep) {
        if (bytes <= cap) return hipSuccess;
        void* old = p;
        p = nullptr; cap = 0;
        hipError_t e = ensure(bytes);
        if (e == hipSuccess && old && keep) memcpy(p, old, keep);
        if (old) { const hipError_t f = hipHostFree(old); if (e == hipSuccess) e = f; }
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
    }
};

constexpr int kLongWaves = 2048;                       // most waves of a long-read launch (one pair each at a time): two per SIMD
constexpr size_t kLongBoundaryBytes = size_t{256} << 20;  // most memory for their boundary rows (fewer waves when haplotypes are long)
constexpr size_t kCounterBytes = 256;  // device counters of a pass: error word, rerun-list lengths, pair-list lengths, first bad read
constexpr int kErrorReadSlot = 1 + 2 * kMaxBuckets;  // counters[kErrorReadSlot]: INT32_MAX - index of the first read with an input error, 0 = none
static_assert((kErrorReadSlot + 1) * sizeof(int32_t) <= kCounterBytes, "counter block");

struct Bucket {
    int rows_per_lane = 0;
    int task_begin = 0;   // offset of this bucket in the rerun-list array
    int num_tasks = 0;
    int desc_begin = 0;   // first RegionDesc of this bucket
    int num_descs = 0;
    int max_hap_len = 0;
    int emit_mode = kEmitCompare;       // fp32 pass
    int emit_mode_fp64 = kEmitCompare;  // fp64 pass: always compares bytes
    // fp32 stream pass (table modes only): haplotype runs instead of single haplotypes
    bool stream = false;
    int stream_desc_begin = 0, num_stream_descs = 0, num_stream_tasks = 0;
    int64_t pair_begin = 0, num_pairs = 0;  // its part of the list of pairs queued for the fp64 pair pass
    int pair_rows_per_lane = 1;             // rows per lane of the fp64 pair pass (its own choice, see plan)
    int slot_lanes = 1;                     // lanes a pair gets there
};

// One host thread of a multi-device handle: runs the jobs it is given one at a time on its own device's engine.
struct DeviceWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, stop = false;
    void start() { th = std::thread([this] { loop(); }); }
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_job || stop; });
            if (!has_job) return;  // stop
            std::function<void()> j = std::move(job);
            has_job = false;
            lk.unlock();
            try { j(); } catch (...) {}  // jobs report through what they capture; an exception must not end the process
            lk.lock();
            done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<void()> j) {
        { std::lock_guard<std::mutex> lk(m); job = std::move(j); has_job = true; done = false; }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
    }
    void shutdown() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

// One gatk_pairhmm_submit call (or the regions a gatk_pairhmm_initialize_region call sent ahead): the library's own
// copy of the inputs, where the results go, and how far it has got.  The request belongs to its handle; while it is
// pending or in flight the device's dispatcher also points at it, and state, engine, rc and error change under the
// dispatcher's mutex only.
struct AsyncRequest {
    int64_t id = 0;
    std::vector<uint8_t> blob;                 // every byte array of every region, copied at submit
    std::vector<int32_t> offsets;              // their read / haplotype offset tables, rebased to 0
    std::vector<gatk_pairhmm_batch> batches;   // pointers into blob and offsets
    std::vector<double*> outs;                 // per region: the caller's array, or a slice of own_out
    std::vector<double> own_out;               // results of regions sent ahead, kept until they are asked for
    std::vector<int64_t> out_off;              // [regions + 1] offsets of the regions in own_out
    int64_t pairs = 0;
    enum State { kPending, kInFlight, kDone } state = kDone;
    int engine = -1;                           // engine of the pass that carries it, while in flight
    int rc = GATK_PAIRHMM_OK;
    std::string error;
    void* dispatcher = nullptr;                // the device dispatcher it was handed to, null when scored at submit
};

}  // namespace pairhmm_host

struct gatk_pairhmm_resident {
    // ---- host-side plan ----
    hipStream_t stream = nullptr;  // every copy and kernel of this batch is ordered on it
    int num_regions = 0;
    int region_label_base = 0;       // index, in the caller's region list, of this batch's first region (error messages)
    bool regions_are_pieces = false; // the call cut large regions by read ranges: indices count the pieces
    std::vector<int64_t> region_out_off;  // [num_regions + 1] offsets into the concatenated output
    int64_t total_pairs = 0, cells = 0, total_tasks = 0;
    int64_t lane_rows_used = 0, lane_rows_allocated = 0;  // read rows the forward launches hold / rows their lanes offer (x haplotypes)
    int64_t total_read_bytes = 0;
    std::vector<pairhmm_host::Bucket> buckets;
    std::vector<pairhmm_hip::RegionDesc> descs;      // as uploaded, bucket by bucket
    std::vector<uint8_t> group_segs;    // reads packed in each wave-group
    int32_t num_neg_inf = 0;            // output slots of zero-length reads
    bool computed = false;
    double plan_ms = 0, upload_ms = 0;
    // ---- device ----
    pairhmm_host::DevBuf d_arena;  // every input array of the call, one H2D copy
    // d_out: [pairhmm_host::kCounterBytes of counters][likelihoods]: one device-to-host copy fetches both
    pairhmm_host::DevBuf d_out, d_rerun_lists;
    int32_t* counters() const { return d_out.as<int32_t>(); }
    double* out() const { return reinterpret_cast<double*>(d_out.as<uint8_t>() + pairhmm_host::kCounterBytes); }
    pairhmm_host::DevBuf d_row_consts;  // fp32 row constants of every read, written on the device
    pairhmm_host::DevBuf d_hap_stream, d_raw_out, d_pair_lists;  // fp32 stream pass and the fp64 pair pass behind it
    pairhmm_host::DevBuf d_long_boundary;  // long-read pass: two boundary rows per wave
    int32_t num_long_pairs = 0;            // pairs of reads too long for a wave-group
    int long_hap_max = 0;                  // longest haplotype among them
    int long_read_max = 0;                 // longest of those reads
    bool haps_acgtn_only = false;          // no haplotype of the call holds a byte other than A, C, G, T, N
    bool haps_have_n = false;              // one of them holds an N
    size_t hap_stream_entries = 0;
    std::vector<int32_t> region_reads, region_haps;     // per region of the call (0 haplotypes when it has no reads)
    std::vector<int32_t> region_read_begin, region_hap_begin;  // global index of a region's first read / haplotype
    int64_t total_reads = 0, total_haps = 0;
    pairhmm_host::DevBuf d_epilogue;  // small arrays of the steps behind the path: per-region parameters, flags, marginal likelihoods
    bool any_stream = false;
    bool streamed = false;  // the last pass enqueued used the stream pass and the fp64 pair pass
    pairhmm_hip::DeviceBatch base{};  // pointers into d_arena
    bool counters_clear = false;        // the counters were zeroed behind the last download: the next pass need not
    hipEvent_t ev_downloaded = nullptr; // recorded behind a download's device-to-host copy
    // HIP events of the passes enqueued since the last wait (timing option): start, forward kernels
    // begin / end, fp64 pass begin, end
    using EventSet = std::array<hipEvent_t, 5>;
    std::vector<EventSet> ev_pending, ev_free;
    gatk_pairhmm_stats last_timing{};
    void release() {
        for (pairhmm_host::DevBuf* b : {&d_arena, &d_out, &d_rerun_lists, &d_row_consts, &d_hap_stream, &d_raw_out, &d_pair_lists, &d_long_boundary, &d_epilogue}) b->release();
        counters_clear = false;
        if (ev_downloaded) (void)hipEventDestroy(ev_downloaded);
        ev_downloaded = nullptr;
        for (auto* v : {&ev_pending, &ev_free}) {
            for (auto& set : *v)
                for (auto& e : set)
                    if (e) (void)hipEventDestroy(e);
            v->clear();
        }
    }
};

struct gatk_pairhmm {
    // A multi-device handle (gatk_pairhmm_create_multi) owns one single-device engine per entry of its device
    // list and computes nothing itself: its compute calls shard regions over the engines.  children[0] runs on
    // the calling thread, children[k > 0] on workers[k - 1].
    std::vector<gatk_pairhmm*> children;
    std::vector<std::unique_ptr<pairhmm_host::DeviceWorker>> workers;
    unsigned next_child = 0;     // small calls go to the engines in turn
    bool stats_in_first_engine = false;  // the last call went through the resident interface, which the first engine serves
    int64_t multi_min_pairs = 8192;  // a call with fewer pairs goes to one engine on the calling thread
    bool is_shared_engine = false;  // an engine of a device's dispatcher, not a caller's handle
    bool counted = false;           // counted among the live callers' handles (see release_shared_engines_if_last)
    int device = 0;
    hipStream_t stream = nullptr;
    int wave_slots = 3072;  // waves the device holds at once at three per SIMD
    size_t kernel_upload_bytes = size_t{1} << 20;  // uploads up to this size are copied by a kernel, not by the DMA engine
    size_t kernel_download_bytes = size_t{512} << 10;  // results of at most this many bytes are copied back by a kernel
    size_t signal_download_bytes = size_t{64} << 10;   // ... and of at most this many by one block that then raises a flag the host polls
    pairhmm_host::PinnedBuf done_flag;                               // that flag (one word of pinned host memory)
    pairhmm_host::PinnedBuf signal_out;                              // and where such a result lands (coherent pinned memory)
    unsigned done_seq = 0;                             // value the next such download stores there
    bool debug_long_stall = false;  // test hook: the long-read pass withholds a chunk's progress, so that the wait for it runs into its limit
    int stream_run_divisor = 2;  // a haplotype run of the stream pass is the items still to hand out over this many times the wave slots
    std::vector<uint8_t> group_rows_table;  // rows per lane by longest read of a wave-group when only some values are allowed (empty: all are)
    int64_t per_group_factor = 32;  // calls with at least this many pairs per wave slot choose rows per lane per wave-group
    bool combine_calls = false;  // small compute_likelihoods calls of all threads on this device share the device's dispatcher
    int64_t pipeline_mid_ramp_min = 3840;  // a call of at least eight times this many pairs (and less than three parts' worth) is cut in four
    int pipeline_ramp_steps = 3;  // parts of a large cut call that come before the first full one, each half the size of the next
    bool pipeline_ramp = true;  // the first two parts of a cut call are a quarter and a half of pipeline_pairs
    int stream_lds_floor = 0;  // least LDS bytes per wave of the stream pass (caps the waves a CU holds)
    double* d_err = nullptr;
    double* d_mm = nullptr;
    bool initialized = false;
    int max_read = 0, max_hap = 0;
    float fp32_threshold = 1e-28f;
    int force_rows_per_lane = 0;
    int emit_mode = -1;  // -1: table when the haplotype alphabet allows it; 0: always compare bytes
    bool squash_base_quals = false;  // base qualities under 18 count as 6 (PairHMM.java:54, QualityUtils.java:57)
    bool no_tristate_correction = false;  // PairHMM.doNotUseTristateCorrection (PairHMM.java:83-84): mismatch prior = error probability, not a third of it
    int stream_mode = -1;  // -1: haplotype runs in table modes; 0: one haplotype per work item
    int max_stream_haps = pairhmm_hip::kMaxStreamHaps;
    bool force_fp64 = false;
    bool timing = false;
    // buffers reused by the host-pointer entry points; three of them, each with a stream of its own, so that
    // a call over many regions can plan and copy one part while the kernels of the two parts before it run
    gatk_pairhmm_resident scratch, scratch2, scratch3;
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    int pipeline_depth = 3;  // parts of a cut call in flight or in planning at once (2 or 3)
    int64_t pipeline_pairs = 98304;  // pairs per part of such a call; 0: never split a call
    int64_t pipeline_cells_per_pair = 80000;  // pairs of more than 100 000 cells count as cells / this towards a part of a cut call; 0: count pairs only
    int64_t pipeline_min_pairs = 6144;  // the halves of a call with less than two parts' worth of pairs are at least this large
    // Pinned staging shared by every upload / download on this handle.  Uploads alternate between two staging buffers:
    // the image of one part of a cut call is put together while the copy of the part before it is still reading the
    // other buffer (with one buffer the planner waited for that copy before it could write the first byte).
    pairhmm_host::PinnedBuf stage_in_buffers[2], stage_out;
    int stage_in_slot = 0;                 // the buffer of the latest upload
    hipEvent_t stage_in_copied[2] = {nullptr, nullptr};  // recorded behind the host-to-device copy out of each buffer
    bool stage_in_busy[2] = {false, false};
    hipStream_t stage_in_unrecorded = nullptr;  // stream of an upload whose event is still to be recorded
    bool stage_in_record_due = false;
    bool in_cut_call = false;  // a call cut into overlapped parts is under way: the next part waits for an upload's copy, not for its kernels
    std::vector<double> likelihood_array;
    // where the last call's input error was, as indices into the region list the failing pass was given (-1: the call
    // did not fail on a read's qualities): callers that cut or deal out regions translate them into the caller's own
    int error_region = -1, error_read = -1;
    gatk_pairhmm_stats stats{};
    // BatchPairHMM queue: one single-read region per batchAdd
    struct Queued {
        std::vector<uint8_t> bases, quals, ins, del, gcp, haps;
        std::vector<int32_t> hap_off;
    };
    std::vector<Queued> queue;
    std::vector<double> queue_results;
    // gatk_pairhmm_submit / gatk_pairhmm_fetch: requests by ticket, and retired ones whose buffers the next submit reuses
    std::map<int64_t, std::unique_ptr<pairhmm_host::AsyncRequest>> tickets;
    std::vector<std::unique_ptr<pairhmm_host::AsyncRequest>> spare_requests;
    int64_t next_ticket = 1;
    // gatk_pairhmm_initialize_region: ticket of the region sent ahead (0: none), whether it has been waited for, and
    // the sample the next computeLikelihoods call is expected to ask for
    int64_t region_ticket = 0;
    bool region_fetched = false;
    int region_next_sample = 0;
    int64_t region_hits = 0, region_misses = 0;  // computeLikelihoods calls served from / not found in a region sent ahead
};

namespace pairhmm_host {

double now_ms();
// fails with the message of an input error found on the device (code: ILLEGAL_ARGUMENT for a negative gap quality,
// INDEX_OUT_OF_BOUNDS for base quality 255) at read `read` of region `region`, and records the two in h
int fail_input_error(gatk_pairhmm* h, int code, int region, int read, bool regions_are_pieces);
int check_handle(gatk_pairhmm* h);
// The resident interface, the allocation-free entry point and the stream getter work on one device: on a
// multi-device handle they are served by the engine of the first device in the handle's list.
gatk_pairhmm* first_engine(gatk_pairhmm* h);
// Shape checks of one computeLikelihoods call with the reference's exceptions as return codes; k labels the
// region in the message.
int check_region(const gatk_pairhmm_batch& b, int k, int* read_max_out = nullptr, int* hap_max_out = nullptr);
void note_call_maxima(gatk_pairhmm* h, const gatk_pairhmm_batch& b, int read_max, int hap_max);
// Plans the regions into wave tasks, packs them into the handle's pinned staging buffer and enqueues the one
// host-to-device copy on R's stream.  checked: check_region has been run on every region already.
int plan_and_upload(gatk_pairhmm* h, gatk_pairhmm_resident* R, int num_regions, const gatk_pairhmm_batch* regions, bool checked = false);
// Enqueues validation, forward kernels and the fp64 rerun of a planned batch; does not wait.
int enqueue_compute(gatk_pairhmm* h, gatk_pairhmm_resident* R);
// Waits for R's stream, turns the device error word into the reference's exception code, fills the stats.
int finish_compute(gatk_pairhmm* h, gatk_pairhmm_resident* R);

// A result's way home in two halves, so that a caller can do other work between them.  begin enqueues the copy
// of counters and likelihoods into pinned host memory behind the kernels of the pass; ready tells, without
// blocking, whether it has landed; end waits for it, digests the counters (stats, the reference's exception for
// an input error) and scatters the likelihoods into outs[k].  download_into is begin + end.
struct PendingDownload {
    bool active = false;
    bool direct = false;     // one large matrix: end copies it straight into the caller's array
    bool by_signal = false;  // the copy kernel raises a flag in pinned memory that the host polls
    unsigned seq = 0;        // the value that flag takes
    const uint8_t* landed = nullptr;  // where counters + likelihoods arrive
};
int download_begin(gatk_pairhmm* h, gatk_pairhmm_resident* R, double* const* outs, PendingDownload& d);
bool download_ready(gatk_pairhmm* h, gatk_pairhmm_resident* R, const PendingDownload& d);
int download_end(gatk_pairhmm* h, gatk_pairhmm_resident* R, double* const* outs, PendingDownload& d);
int download_into(gatk_pairhmm* h, gatk_pairhmm_resident* R, double* const* outs);

// compute_regions on a single-device handle (blocking; cuts large calls into overlapped parts).
int compute_regions_single(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions,
                           double* const* likelihoods, bool checked);

// ---- submit / fetch and the region sent ahead (pairhmm_async.cpp) ----
// True when the handle's options are the ones the shared per-device engines run with (squash_base_quals aside).
bool has_default_options(const gatk_pairhmm* h);
// gatk_pairhmm_compute_likelihoods asks whether `batch` is a sample of the region sent ahead on this handle; when it
// is, *hit = true, the sample's likelihoods are in `likelihoods` and the return value is the call's code.
int serve_from_region(gatk_pairhmm* h, const gatk_pairhmm_batch* batch, double* likelihoods, bool* hit);
// Forgets the region sent ahead / every ticket of the handle, waiting for passes that still write into them.
void drop_region(gatk_pairhmm* h);
void drop_tickets(gatk_pairhmm* h);
// Frees the shared engines of every device (called when the last caller's handle closes).
void release_dispatch_engines();
// Passes the dispatcher of (device, squash) has run and the requests they carried, since the process started.
void dispatch_counts(int device, bool squash, int64_t* passes, int64_t* requests);
// Blocking submit + fetch of one region through the device's dispatcher (the handle option "combine_calls").
int compute_shared(gatk_pairhmm* h, const gatk_pairhmm_batch* batch, double* likelihoods);
// An engine of a device's dispatcher: a single-device handle that is not counted among the callers' handles.
int create_shared_engine(int device, bool squash, gatk_pairhmm** out);

}  // namespace pairhmm_host
