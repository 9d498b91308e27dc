// C-ABI of the PairHMM engine (include/gatk_pairhmm.h): argument checks with the reference's
// error behaviour, the host-side planner that turns region batches into wavefront tasks, device
// buffer management and kernel sequencing.  There is no CPU compute path in this file: every
// likelihood comes out of the HIP kernels, and handle creation fails when no device is present.
#include "pairhmm_engine.h"

using namespace pairhmm_hip;
using namespace pairhmm_host;

// ---------------------------------------------------------------------------------------------
// Working memory of gatk_pairhmm_compute_likelihoods_noalloc, the entry point for callers that cannot
// allocate: every device holds its own copy of this array for the life of the process.  Host code never
// uses the array's name as an address: gatk_pairhmm_create asks the runtime for the device's address once
// (hipGetSymbolAddress) and keeps it in g_scratch_ptr[device].
// ---------------------------------------------------------------------------------------------
namespace noalloc {
constexpr int kMaxReads = 4096;          // reads per call
constexpr int kMaxHaps = 256;            // haplotypes per call
constexpr int64_t kMaxPairs = 262144;    // reads x haplotypes per call (4096 x 64)
constexpr int kMaxHapLen = 4096;         // haplotype_max_length a caller may declare
constexpr size_t kHapPoolBytes = size_t{512} << 10;  // haplotypes, each padded to a multiple of 16 bytes
constexpr size_t at_counters = 0;                                  // kCounterBytes: error word, rerun-list length
constexpr size_t at_region_desc = 256;                             // one RegionDesc
constexpr size_t at_read_out_base = 512;                           // int32[kMaxReads]
constexpr size_t at_groups = at_read_out_base + kMaxReads * 4;     // GroupDesc[kMaxReads]
constexpr size_t at_segs = at_groups + kMaxReads * 8;              // SegDesc[kMaxReads]
constexpr size_t at_hap_off = at_segs + kMaxReads * 8;             // int32[kMaxHaps]
constexpr size_t at_hap_len = at_hap_off + kMaxHaps * 4;           // int32[kMaxHaps]
constexpr size_t at_hap_pool = at_hap_len + kMaxHaps * 4;          // kHapPoolBytes + 64
constexpr size_t at_rerun_list = at_hap_pool + kHapPoolBytes + 256;  // int32[kMaxPairs]
constexpr size_t kScratchBytes = at_rerun_list + static_cast<size_t>(kMaxPairs) * 4;  // 1.66 MB
static_assert(sizeof(GroupDesc) == 8 && sizeof(SegDesc) == 8 && sizeof(RegionDesc) == 32, "scratch layout");
static_assert(at_groups % 256 == 0 && at_segs % 256 == 0 && at_hap_off % 256 == 0 && at_hap_pool % 256 == 0 && at_rerun_list % 256 == 0, "scratch layout");
}  // namespace noalloc

static __device__ __align__(256) unsigned char scratch[noalloc::kScratchBytes];

namespace {

thread_local std::string g_last_error;
std::atomic<int> g_num_threads{1};
std::atomic<int> g_live_handles{0};  // callers' single-device handles (the dispatchers' shared engines are not counted)
void release_shared_engines_if_last();
int g_fail_share_for_test = -1;  // gatk_pairhmm_debug_fail_share: the multi-device share of this engine throws once it is done
// compute_regions calls under way per device (all handles): a call that has the device to itself overlaps
// its own halves; one that shares it is overlapped with the calls of the other handles already
std::atomic<int> g_region_calls[64];  // indexed by device & 63: devices 64 apart would share a count, which only costs overlap
struct RegionCallCount {
    std::atomic<int>& n;
    int others;
    explicit RegionCallCount(int device) : n(g_region_calls[device & 63]), others(n.fetch_add(1)) {}
    ~RegionCallCount() { n.fetch_sub(1); }
};


// Address of `scratch` on each device, filled once per device by setup_scratch with that device current.
constexpr int kMaxScratchDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxScratchDevices] = {};

int setup_scratch(int device) {
    if (device < 0 || device >= kMaxScratchDevices) return GATK_PAIRHMM_OK;  // such a device has no allocation-free entry point
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return GATK_PAIRHMM_OK;
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(GATK_PAIRHMM_DEVICE_ERROR, "hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    return GATK_PAIRHMM_OK;
}

}  // namespace

namespace pairhmm_host {

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

const std::string& last_error() { return g_last_error; }

int on_exception() noexcept {
    try {
        try { throw; }
        catch (const std::bad_alloc&) { return fail(GATK_PAIRHMM_OUT_OF_MEMORY, "host memory exhausted"); }
        catch (const std::exception& e) { return fail(GATK_PAIRHMM_DEVICE_ERROR, std::string("internal error: ") + e.what()); }
        catch (...) { return fail(GATK_PAIRHMM_DEVICE_ERROR, "internal error: unknown exception"); }
    } catch (...) {
        return GATK_PAIRHMM_OUT_OF_MEMORY;  // not even the message could be stored
    }
}

}  // namespace pairhmm_host


namespace pairhmm_host {

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// Planner
// ---------------------------------------------------------------------------------------------

// Rows per lane for a region: pack the region's reads (lengths sorted longest first) into wave-groups
// for every candidate and take the one with the fewest modelled issue slots, keeping enough waves in
// flight.  call_pairs: (read, haplotype) pairs of the whole call, all regions together: regions share
// launches, so whether the GPU is full is a property of the call, not of one region.
int choose_rows_per_lane(const std::vector<int>& lens_desc, int64_t num_haps, double mean_hap_len, int64_t call_pairs, int forced) {
    const int max_read_len = lens_desc.empty() ? 1 : lens_desc.front();
    const int min_k = (max_read_len + 1 + kWave) / kWave;  // lanes * k >= len + 2 (a pad row in front, the sum row behind)
    if (forced >= min_k && rows_per_lane_supported(forced)) return forced;
    int best_k = kMaxRowsPerLane;
    double best = std::numeric_limits<double>::infinity();
    for (int k : kRowsPerLaneChoices) {
        if (k < min_k) continue;
        if (k > 12 && best_k != kMaxRowsPerLane) break;  // more rows cost a wave per SIMD: only for long reads
        // the packing the planner will do: segments in order, a new wave-group when 64 lanes are full
        int64_t groups = 0;
        double group_steps = 0;
        int used = kWave;
        for (int len : lens_desc) {
            const int seg = (len + 1 + k) / k;
            if (used + seg > kWave) { ++groups; used = 0; group_steps += mean_hap_len + 1 + seg; }
            used += seg;
        }
        const double per_wave = std::max(1.0, static_cast<double>(lens_desc.size()) / std::max<int64_t>(groups, 1));
        double slots = group_steps * (5.0 * k + 6.0) * static_cast<double>(num_haps);  // vector instructions of the region
        if (k > 10) slots *= 1.03;  // the fp64 pass spills beyond 10 rows
        if (k > 12) slots *= 1.15;  // two waves per SIMD instead of three
        const double tasks = static_cast<double>(groups) * static_cast<double>(num_haps);
        // two waves on each of the 1024 SIMDs saturate issue
        const double in_flight = std::min(std::max(tasks, static_cast<double>(call_pairs) / per_wave), 2048.0);
        const double t = slots / in_flight;
        if (t < best * 0.99) { best = t; best_k = k; }
    }
    return best_k;
}

// Rows per lane for a whole call that is too small to keep the device busy for long: one value for every
// region, because every value in use costs a launch of its own whose waves run from start to end with the
// device partly empty.  Measured on calls of 1 to 128 HaplotypeCaller-sized regions: while the call's work
// items (one wave-group on one haplotype) number at most two per SIMD, a wave issues at the rate of a wave
// alone, so the value with the shortest wave wins, which is the smallest that still has so few items; once
// no value has so few, the one with the fewest vector instructions in total wins.
// A call of fewer pairs than this per wave slot of the device goes through the one-haplotype pass (measured with the
// one-haplotype kernel of up to 6 rows per lane against haplotype runs: 16 regions 147 against 157 us, 32 regions 310
// against 317 us, equal from 128 regions, runs ahead beyond: 1000 regions 4.7 against 5.4 ms).
constexpr int kRunsFromPairsPerSlot = 4;
struct RegionShape { int num_reads, num_haps; double packing_read_len, mean_hap_len; };
// one_haplotype_pass: the call is too small for haplotype runs and goes through the one-haplotype kernel, whose
// instantiations up to 6 rows per lane keep their table reads a block ahead and sweep without lane masks (measured
// about a quarter faster per instruction than the others on such calls).  There the choice is by modelled time: the
// vector instructions of an average wave, times how many waves a SIMD gets beyond the two that issue at a lone wave's
// rate.  (Calls of 8 to 12 regions: 5 rows per lane instead of 7 or 8, 128 -> 116 us at 12 regions.)
int choose_call_rows_per_lane(const std::vector<RegionShape>& shapes, int longest_read, bool one_haplotype_pass) {
    const int min_k = (longest_read + 1 + kWave) / kWave;
    int best_k = 0;
    double best = std::numeric_limits<double>::infinity();
    for (int k : kRowsPerLaneChoices) {
        if (k < min_k) continue;
        if (k > 12 && best_k != 0) break;  // more rows cost a wave per SIMD: only for long reads
        double items = 0, work = 0;
        for (const RegionShape& r : shapes) {
            const int seg = static_cast<int>((r.packing_read_len + 1 + k) / k);
            const int per_wave = std::max(1, kWave / std::max(seg, 1));
            const double w = static_cast<double>((r.num_reads + per_wave - 1) / per_wave) * r.num_haps;
            items += w;
            work += w * (r.mean_hap_len + seg + 1) * (5.0 * k + 6.0);
        }
        if (one_haplotype_pass) {
            const double time = work / std::max(items, 1.0) * std::max(1.0, items / (2.1 * 1024)) * (k > 6 ? 1.25 : 1.0);
            if (time < best * 0.99) { best = time; best_k = k; }
            continue;
        }
        if (items <= 2.1 * 1024) return k;
        if (work < best * 0.99) { best = work; best_k = k; }
    }
    return best_k != 0 ? best_k : kMaxRowsPerLane;
}

// Rows per lane for a wave-group whose longest read has `longest` bases, in a call large enough to keep the
// device full: the value with the fewest modelled vector instructions per read, taking the group's other
// reads to be as long as its longest (reads are packed longest first, so they are close) and a typical
// haplotype length for the weight of the pipeline fill.  A table: the planner asks once per wave-group.
// allowed: bit k set = k rows per lane may be chosen (a read too long for every allowed value takes the best of all).
std::array<uint8_t, kWave * kMaxRowsPerLane> group_rows_table(uint32_t allowed) {
    std::array<uint8_t, kWave * kMaxRowsPerLane> t{};
    for (int len = 1; len <= kWave * kMaxRowsPerLane - 2; ++len) {
        const int min_k = (len + 1 + kWave) / kWave;
        for (int pass = 0; pass < 2 && t[len] == 0; ++pass) {
            int best_k = 0;
            double best = std::numeric_limits<double>::infinity();
            for (int k : kRowsPerLaneChoices) {
                if (k < min_k || (pass == 0 && !((allowed >> k) & 1u))) continue;
                if (k > 12 && best_k != 0) break;  // more rows cost a wave per SIMD: only for long reads
                const int lanes = (len + 1 + k) / k;
                double cost = (5.0 * k + 6.0) * (400.0 + lanes) / (kWave / lanes);
                if (k > 10) cost *= 1.03;  // the one-haplotype fp64 pass spills beyond 10 rows
                if (k > 12) cost *= 1.15;  // two waves per SIMD instead of three
                if (cost < best * 0.995) { best = cost; best_k = k; }
            }
            t[len] = static_cast<uint8_t>(best_k);
        }
        if (t[len] == 0) t[len] = kMaxRowsPerLane;
    }
    return t;
}

int choose_group_rows_per_lane(int longest) {
    static const std::array<uint8_t, kWave * kMaxRowsPerLane> table = group_rows_table(~0u);
    return table[longest];
}

// What can be said about a region without walking its reads: counts and pointers.  After it the region's offset
// tables may be read at their two ends.
int check_region_arrays(const gatk_pairhmm_batch& b, int k) {
    const auto where = [k] { return " (region " + std::to_string(k) + ")"; };  // built only for a failure
    if (b.num_reads < 0 || b.num_haps < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("negative read or haplotype count") + where());
    if (b.num_reads == 0) return GATK_PAIRHMM_OK;  // PairHMM.java:193
    if (!b.read_offsets || !b.read_bases || !b.read_quals)  // insertion / deletion / GCP arrays may be null: defaults
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("readBases cannot be null") + where());
    if (b.num_haps > 0 && (!b.hap_offsets || !b.hap_bases))
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("haplotypeBases cannot be null") + where());
    return GATK_PAIRHMM_OK;
}

int check_region(const gatk_pairhmm_batch& b, int k, int* read_max_out, int* hap_max_out) {
    const auto where = [k] { return " (region " + std::to_string(k) + ")"; };  // built only for a failure
    if (b.num_reads < 0 || b.num_haps < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("negative read or haplotype count") + where());
    if (b.num_reads == 0) return GATK_PAIRHMM_OK;  // PairHMM.java:193
    if (!b.read_offsets || !b.read_bases || !b.read_quals)  // insertion / deletion / GCP arrays may be null: defaults
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("readBases cannot be null") + where());
    // branch-free passes over the offset tables (they vectorise): a large call walks some hundred thousand entries here
    int read_max = 0, hap_max = 0, read_min = 0, hap_min = std::numeric_limits<int>::max();
    for (int r = 0; r < b.num_reads; ++r) {
        const int len = b.read_offsets[r + 1] - b.read_offsets[r];
        read_min = std::min(read_min, len);
        read_max = std::max(read_max, len);
    }
    if (read_min < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("read offsets must not decrease") + where());
    if (b.num_haps > 0 && (!b.hap_offsets || !b.hap_bases))
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("haplotypeBases cannot be null") + where());
    for (int a = 0; a < b.num_haps; ++a) {
        const int len = b.hap_offsets[a + 1] - b.hap_offsets[a];
        hap_min = std::min(hap_min, len);
        hap_max = std::max(hap_max, len);
    }
    if (b.num_haps > 0 && hap_min < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("haplotype offsets must not decrease") + where());
    // PairHMM.initialize, reached through PairHMM.java:200-201
    if (read_max <= 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "READ_MAX_LENGTH must be > 0 but got " + std::to_string(read_max) + where());
    if (hap_max <= 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "HAPLOTYPE_MAX_LENGTH must be > 0 but got " + std::to_string(hap_max) + where());
    if (b.num_haps > 0 && hap_min == 0)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, std::string("zero-length haplotype: the reference result is undefined") + where());
    if (read_max > kMaxReadLen)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "readBases is too long, got " + std::to_string(read_max) + " but max is " +
                                                       std::to_string(kMaxReadLen) + where());
    if (hap_max > kMaxHapLen)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "Haplotype bases is too long, got " + std::to_string(hap_max) + " but max is " +
                                                       std::to_string(kMaxHapLen) + where());
    if (read_max_out) *read_max_out = read_max;
    if (hap_max_out) *hap_max_out = hap_max;
    return GATK_PAIRHMM_OK;
}

// PairHMM.java:198-201: a computeLikelihoods call whose longest read or longest haplotype exceeds the instance's
// maxima (or the first call) re-initialises the instance with THIS call's two maxima, so the other one may shrink.
void note_call_maxima(gatk_pairhmm* h, const gatk_pairhmm_batch& b, int read_max, int hap_max) {
    if (b.num_reads == 0 || b.num_haps == 0) return;
    if (!h->initialized || read_max > h->max_read || hap_max > h->max_hap) {
        h->max_read = read_max;
        h->max_hap = hap_max;
        h->initialized = true;
    }
}

// Carves aligned sub-arrays out of one staging buffer; the same offsets are valid in the device arena.
struct ArenaLayout {
    size_t size = 0;
    size_t add(size_t bytes) {
        const size_t at = size;
        size = (size + bytes + 255) & ~size_t{255};
        return at;
    }
};

// Which letters does a haplotype use?  has_n: an 'N'; has_other: anything outside A, C, G, T, N.
void scan_alphabet(const uint8_t* p, int len, bool& has_n, bool& has_other) {
    int x = 0;
    const __m128i a = _mm_set1_epi8('A'), c = _mm_set1_epi8('C'), g = _mm_set1_epi8('G'), t = _mm_set1_epi8('T'), n = _mm_set1_epi8('N');
    __m128i all_ok = _mm_set1_epi8(-1), any_n = _mm_setzero_si128();
    for (; x + 16 <= len; x += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + x));
        const __m128i is_n = _mm_cmpeq_epi8(v, n);
        const __m128i acgt = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, a), _mm_cmpeq_epi8(v, c)), _mm_or_si128(_mm_cmpeq_epi8(v, g), _mm_cmpeq_epi8(v, t)));
        any_n = _mm_or_si128(any_n, is_n);
        all_ok = _mm_and_si128(all_ok, _mm_or_si128(acgt, is_n));
    }
    if (_mm_movemask_epi8(any_n) != 0) has_n = true;
    if (_mm_movemask_epi8(all_ok) != 0xffff) has_other = true;
    for (; x < len; ++x) {
        const uint8_t b = p[x];
        if (b == 'N') has_n = true;
        else if (b != 'A' && b != 'C' && b != 'G' && b != 'T') has_other = true;
    }
}


// A memcpy into the staging buffer.
struct CopyJob {
    uint8_t* dst;
    const uint8_t* src;
    size_t bytes;
};

// Large calls are bound by these copies (five bytes per read base): they go to a few threads, which start as soon
// as the jobs are known and run beside the planner's walk over the regions; a small call copies on its own thread.
struct CopyCrew {
    std::vector<CopyJob> jobs;
    std::vector<std::thread> pool;
    bool started = false;
    static void run(const std::vector<CopyJob>* jobs, size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) memcpy((*jobs)[k].dst, (*jobs)[k].src, (*jobs)[k].bytes);
    }
    // jobs is complete and stays untouched until finish()
    void start() {
        size_t total = 0;
        for (const CopyJob& j : jobs) total += j.bytes;
        const int threads = total < (size_t{2} << 20) ? 0 : static_cast<int>(std::min<size_t>(4, 1 + (total >> 21)));
        started = true;
        if (threads == 0) return;  // finish() copies
        size_t begin = 0, acc = 0;
        for (size_t k = 0; k < jobs.size(); ++k) {
            acc += jobs[k].bytes;
            if (acc >= total / threads || k + 1 == jobs.size()) {
                pool.emplace_back(run, &jobs, begin, k + 1);
                begin = k + 1; acc = 0;
            }
        }
    }
    void finish() {
        if (!started) return;
        if (pool.empty()) run(&jobs, 0, jobs.size());
        for (auto& t : pool) t.join();
        pool.clear();
        started = false;
    }
    ~CopyCrew() {  // an early return between start() and finish(): the threads still hold the staging buffer
        for (auto& t : pool) t.join();
    }
};

// Per-region planning record (host only).
struct RegionPlan {
    int region = 0, K = 0, nr = 0, nh = 0;
    int read_begin = 0;   // global index of the region's first read
    int hap_begin = 0;    // global index of its first haplotype
    int group_begin = 0, num_groups = 0;
    int hap_max = 0, cost = 0;
    int read_max = 0;
    int longest = 0;      // longest read in this plan's wave-groups
    int seg_reads = 0;    // reads in them
};

// Records the event that guards the staging buffer behind whatever is enqueued on the uploading stream so
// far.  Later than the copy itself is safe (the next upload waits a little longer); it is called after the
// kernels of a region-sized pass have been enqueued, so that the record does not delay them.
int record_stage_in_event(gatk_pairhmm* h) {
    if (!h->stage_in_record_due) return GATK_PAIRHMM_OK;
    hipEvent_t& copied = h->stage_in_copied[h->stage_in_slot];
    if (!copied) HIP_TRY(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(copied, h->stage_in_unrecorded));
    h->stage_in_record_due = false;
    h->stage_in_busy[h->stage_in_slot] = true;
    return GATK_PAIRHMM_OK;
}

// Weight of a region towards a part of a cut call: see compute_regions_single.
constexpr int64_t kLongPairCells = 100000;
int64_t pipeline_weight(const gatk_pairhmm_batch& b, int64_t ref_cells) {
    if (b.num_reads <= 0 || b.num_haps <= 0) return 0;
    const int64_t pairs = static_cast<int64_t>(b.num_reads) * b.num_haps;
    if (ref_cells <= 0) return pairs;
    const int64_t cells = static_cast<int64_t>(b.read_offsets[b.num_reads] - b.read_offsets[0]) * (b.hap_offsets[b.num_haps] - b.hap_offsets[0]);
    return cells > pairs * kLongPairCells ? std::max(pairs, cells / ref_cells) : pairs;
}

// Where a call is cut into parts that are planned, copied and scored one behind the other: the regions of part p are
// [cuts[p], cuts[p + 1]).  weights: what each region counts towards a part (pipeline_weight).
//   * A large call: parts of about pipeline_pairs, the first ramp_steps of them an eighth, a quarter, a half ... of
//     that, so that the device starts early (nothing overlaps the planning of the first part).
//   * A call of less than three parts' worth that has the device to itself and at least 8 mid_ramp_min pairs: four
//     parts, an eighth, a quarter and twice five sixteenths of it.  (Against two halves, on 64 / 128 / 192 / 256 / 320
//     HaplotypeCaller-sized regions per call: 3 / 9 / 7 / 16 / 13 % less time.)
//   * A smaller one that has the device to itself: two halves of at least min_pairs, so that the planning of the
//     second and the copy back of the first hide behind kernels; below two min_pairs, one part.
// No cut leaves less than half a part behind it.
constexpr int kCheckByPartFromRegions = 64;  // calls of at least this many regions are checked as their parts are planned
struct CutRules { int64_t pipeline_pairs, min_pairs, mid_ramp_min; bool ramp; int ramp_steps; bool device_is_shared; };
std::vector<int> part_cuts(const std::vector<int64_t>& weights, const CutRules& rules) {
    const int n = static_cast<int>(weights.size());
    std::vector<int> cuts{0};
    if (rules.pipeline_pairs > 0) {
        int64_t acc = 0, total = 0, taken = 0;
        for (const int64_t w : weights) total += w;
        const int64_t part_pairs = rules.device_is_shared ? rules.pipeline_pairs
                                 : std::max(std::min(rules.min_pairs, rules.pipeline_pairs), std::min(rules.pipeline_pairs, total / 2));
        const bool ramp = rules.ramp && total >= 3 * part_pairs;
        const bool mid_ramp = rules.ramp && !ramp && !rules.device_is_shared && total >= 8 * rules.mid_ramp_min;
        for (int k = 0; k < n; ++k) {
            acc += weights[static_cast<size_t>(k)]; taken += weights[static_cast<size_t>(k)];
            const int part = static_cast<int>(cuts.size()) - 1;
            const int64_t want = mid_ramp ? (part == 0 ? total / 8 : part == 1 ? total / 4 : 5 * total / 16)
                                          : !ramp || part >= rules.ramp_steps ? part_pairs : part_pairs >> (rules.ramp_steps - part);
            if (acc >= want && k + 1 < n && 2 * (total - taken) >= (mid_ramp ? want : part_pairs)) { cuts.push_back(k + 1); acc = 0; }
        }
    }
    cuts.push_back(n);
    return cuts;
}

// checked: the caller has run check_region on these regions already (under their own indices)
#ifdef PAIRHMM_CALL_CLOCKS
#define PLAN_CLOCK(i) plan_clock[i] = now_ms()
#else
#define PLAN_CLOCK(i)
#endif
int plan_and_upload(gatk_pairhmm* h, gatk_pairhmm_resident* R, int num_regions, const gatk_pairhmm_batch* regions, bool checked) {
    const double t_begin = now_ms();
#ifdef PAIRHMM_CALL_CLOCKS
    double plan_clock[8] = {};
#endif
    if (num_regions < 0 || (num_regions > 0 && !regions)) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "regions cannot be null");
    if (!checked)
        for (int k = 0; k < num_regions; ++k) {
            int read_max = 0, hap_max = 0;
            if (int rc = check_region(regions[k], R->region_label_base + k, &read_max, &hap_max)) return rc;
            note_call_maxima(h, regions[k], read_max, hap_max);
        }

    R->num_regions = num_regions;
    R->region_out_off.assign(num_regions + 1, 0);
    R->buckets.clear(); R->descs.clear(); R->group_segs.clear();
    R->any_stream = false;
    R->computed = false;
    R->last_timing = gatk_pairhmm_stats{};

    // ---- pass 1: sizes, rows per lane, wave-groups ------------------------------------------------
    int64_t call_pairs = 0, total_reads = 0, total_haps = 0, total_read_bytes = 0, hap_pool_bytes = 0, call_cells = 0;
    for (int k = 0; k < num_regions; ++k) {
        const gatk_pairhmm_batch& b = regions[k];
        const int nr = b.num_reads, nh = nr == 0 ? 0 : b.num_haps;
        R->region_out_off[k + 1] = R->region_out_off[k] + static_cast<int64_t>(nr) * nh;
        if (nr == 0 || nh == 0) continue;
        call_pairs += static_cast<int64_t>(nr) * nh;
        total_reads += nr;
        total_haps += nh;
        total_read_bytes += b.read_offsets[nr] - b.read_offsets[0];
        call_cells += static_cast<int64_t>(b.read_offsets[nr] - b.read_offsets[0]) * (b.hap_offsets[nh] - b.hap_offsets[0]);
        for (int a = 0; a < nh; ++a) hap_pool_bytes += (b.hap_offsets[a + 1] - b.hap_offsets[a] + 15) & ~15;
    }
    if (call_pairs >= (int64_t{1} << 31) || total_read_bytes >= (int64_t{1} << 31) || hap_pool_bytes >= (int64_t{1} << 31))
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "more than 2^31 pairs or bytes in one call");

    // ---- the per-base read arrays lead the staging buffer: where they go depends on sizes only, so their copies
    // start here, on threads of their own when they are large, and run beside the walk over the regions ----------
    ArenaLayout L;
    const size_t rbytes = static_cast<size_t>(total_read_bytes);
    // An optional per-base array that no region of the call supplies is not staged or copied at all: the
    // kernels get a null pointer and take the default for every base.
    bool have_ins = false, have_del = false, have_gcp = false, have_mapq = false;
    for (int k = 0; k < num_regions; ++k) {
        const gatk_pairhmm_batch& b = regions[k];
        if (b.num_reads == 0 || b.num_haps == 0) continue;
        have_ins |= b.ins_gop != nullptr; have_del |= b.del_gop != nullptr; have_gcp |= b.gcp != nullptr;
        have_mapq |= b.read_mapq != nullptr;
    }
    const size_t o_rb = L.add(rbytes + 16), o_rq = L.add(rbytes + 16);
    const size_t o_ri = have_ins ? L.add(rbytes + 16) : 0, o_rd = have_del ? L.add(rbytes + 16) : 0, o_rg = have_gcp ? L.add(rbytes + 16) : 0;
    const size_t bulk_bytes = L.size;
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = record_stage_in_event(h)) return rc;
    h->stage_in_slot ^= 1;  // the other staging buffer: what may still be read out of it is the upload before the last
    if (h->stage_in_busy[h->stage_in_slot]) {
        HIP_TRY(hipEventSynchronize(h->stage_in_copied[h->stage_in_slot]));
        h->stage_in_busy[h->stage_in_slot] = false;
    }
    pairhmm_host::PinnedBuf& stage_in = h->stage_in_buffers[h->stage_in_slot];
    // room for the rest as well, generously: the buffer is not to move while the copies run
    HIP_TRY(stage_in.ensure(bulk_bytes + static_cast<size_t>(hap_pool_bytes) + static_cast<size_t>(total_reads) * 64 +
                               static_cast<size_t>(total_haps) * 400 + static_cast<size_t>(num_regions) * 512 + (size_t{64} << 10)));
    CopyCrew crew;
    {
        uint8_t* const S0 = static_cast<uint8_t*>(stage_in.p);
        crew.jobs.reserve(static_cast<size_t>(num_regions) * 5);
        size_t rcur = 0;
        for (int k = 0; k < num_regions; ++k) {
            const gatk_pairhmm_batch& b = regions[k];
            if (b.num_reads == 0 || b.num_haps == 0) continue;
            const int bo = b.read_offsets[0];
            const size_t total = static_cast<size_t>(b.read_offsets[b.num_reads] - bo);
            crew.jobs.push_back({S0 + o_rb + rcur, b.read_bases + bo, total});
            crew.jobs.push_back({S0 + o_rq + rcur, b.read_quals + bo, total});
            // defaults for a region without the array, when other regions of the call have it
            if (b.ins_gop) crew.jobs.push_back({S0 + o_ri + rcur, b.ins_gop + bo, total});
            else if (have_ins) memset(S0 + o_ri + rcur, kDefaultIndelQual, total);
            if (b.del_gop) crew.jobs.push_back({S0 + o_rd + rcur, b.del_gop + bo, total});
            else if (have_del) memset(S0 + o_rd + rcur, kDefaultIndelQual, total);
            if (b.gcp) crew.jobs.push_back({S0 + o_rg + rcur, b.gcp + bo, total});
            else if (have_gcp) memset(S0 + o_rg + rcur, kDefaultGcp, total);
            rcur += total;
        }
        crew.start();
    }

    // a call that does not fill the device for long: one rows-per-lane value for all of it
    // (pairs count by their length here as they do where a call is cut into parts: see compute_regions_single)
    const int64_t call_weight = h->pipeline_cells_per_pair > 0 && call_cells > call_pairs * kLongPairCells
                                    ? std::max(call_pairs, call_cells / h->pipeline_cells_per_pair) : call_pairs;
    int call_k = h->force_rows_per_lane;
    if (call_k == 0 && call_pairs > 0 && call_weight < h->per_group_factor * static_cast<int64_t>(h->wave_slots)) {
        std::vector<RegionShape> shapes;
        shapes.reserve(num_regions);
        int longest = 1;
        for (int k = 0; k < num_regions; ++k) {
            const gatk_pairhmm_batch& b = regions[k];
            if (b.num_reads == 0 || b.num_haps == 0) continue;
            int nonempty = 0, region_longest = 0;
            int64_t grouped_bytes = 0;
            for (int r = 0; r < b.num_reads; ++r) {
                const int len = b.read_offsets[r + 1] - b.read_offsets[r];
                if (len <= 0 || len > kMaxGroupedReadLen) continue;  // no rows, or the long-read pass: no wave-group either way
                region_longest = std::max(region_longest, len);
                grouped_bytes += len;
                ++nonempty;
            }
            if (nonempty == 0) continue;
            longest = std::max(longest, region_longest);
            // reads are packed longest first: between the mean and the longest is what decides how many share a wave
            const double mean_len = static_cast<double>(grouped_bytes) / nonempty;
            shapes.push_back(RegionShape{nonempty, b.num_haps, 0.5 * (mean_len + region_longest),
                                         static_cast<double>(b.hap_offsets[b.num_haps] - b.hap_offsets[0]) / b.num_haps});
        }
        call_k = choose_call_rows_per_lane(shapes, longest, h->stream_mode == 0 || (h->stream_mode < 0 && call_pairs < kRunsFromPairsPerSlot * static_cast<int64_t>(h->wave_slots)));
    }

    PLAN_CLOCK(0);
    std::vector<RegionPlan> plans;
    plans.reserve(num_regions);
    std::vector<GroupDesc> groups;
    std::vector<SegDesc> segs;
    segs.reserve(static_cast<size_t>(total_reads));
    std::vector<int32_t> neg_inf_idx;
    std::vector<RerunPair> long_pairs;  // the pairs of reads too long for a wave-group
    int long_hap_max = 0, long_read_max = 0;
    std::vector<int> order, lens_desc, len_slot;
    std::vector<ReadInfo> read_info(static_cast<size_t>(total_reads));
    std::vector<StreamDesc> stream_descs;
    bool any_hap_n = false, any_hap_other = false;
    int64_t cells = 0, rows_used = 0, rows_allocated = 0;
    int read_cursor = 0, hap_cursor = 0;
    for (int k = 0; k < num_regions; ++k) {
        const gatk_pairhmm_batch& b = regions[k];
        const int nr = b.num_reads, nh = nr == 0 ? 0 : b.num_haps;
        if (nr == 0 || nh == 0) continue;
        RegionPlan rp;
        rp.region = k; rp.nr = nr; rp.nh = nh;
        rp.read_begin = read_cursor; rp.hap_begin = hap_cursor;
        int64_t hap_total = 0;
        for (int a = 0; a < nh; ++a) {
            const int o = b.hap_offsets[a], len = b.hap_offsets[a + 1] - o;
            hap_total += len;
            rp.hap_max = std::max(rp.hap_max, len);
            scan_alphabet(b.hap_bases + o, len, any_hap_n, any_hap_other);  // every haplotype: an 'N' must be seen wherever it is
        }
        int read_max = 0, read_min = std::numeric_limits<int>::max(), nonempty = 0;
        int64_t len_total = 0;
        for (int r = 0; r < nr; ++r) {
            const int len = b.read_offsets[r + 1] - b.read_offsets[r];
            len_total += len;
            read_info[rp.read_begin + r] = ReadInfo{rp.hap_begin, nh, -1, k};
            if (len > kMaxGroupedReadLen) {  // more rows than a wave holds: every pair of the read goes to the long-read pass
                for (int a = 0; a < nh; ++a) long_pairs.push_back(RerunPair{rp.read_begin + r, a});
                long_hap_max = std::max(long_hap_max, rp.hap_max);
                long_read_max = std::max(long_read_max, len);
            } else if (len > 0) {
                read_max = std::max(read_max, len);
                read_min = std::min(read_min, len);
                ++nonempty;
            } else {  // a zero-length read sums an empty last row: log10(0) (Log10PairHMM.java:128-134 on row 0)
                for (int a = 0; a < nh; ++a)
                    neg_inf_idx.push_back(static_cast<int32_t>(R->region_out_off[k] + static_cast<int64_t>(r) * nh + a));
            }
        }
        cells += len_total * hap_total;
        rp.read_max = read_max;
        // wave-groups: longest reads first (equal lengths in read order), filled lane segment by lane segment
        // up to 64 lanes.  Lengths are small integers: a counting sort.
        order.assign(static_cast<size_t>(nonempty), 0);
        lens_desc.assign(static_cast<size_t>(nonempty), 0);
        if (nonempty > 0) {
            len_slot.assign(static_cast<size_t>(read_max - read_min + 2), 0);
            for (int r = 0; r < nr; ++r) {
                const int len = b.read_offsets[r + 1] - b.read_offsets[r];
                if (len > 0 && len <= kMaxGroupedReadLen) ++len_slot[read_max - len + 1];  // slot 0: the longest
            }
            for (size_t i = 1; i < len_slot.size(); ++i) len_slot[i] += len_slot[i - 1];
            for (int r = 0; r < nr; ++r) {
                const int len = b.read_offsets[r + 1] - b.read_offsets[r];
                if (len <= 0 || len > kMaxGroupedReadLen) continue;
                const int at = len_slot[read_max - len]++;
                order[at] = r;
                lens_desc[at] = len;
            }
        }
        // Rows per lane: the call's value (or the forced one; a region whose reads are too long for it gets
        // one of its own from the region model); in a large call every wave-group takes the value that
        // suits its own reads, so a region whose reads differ in length spreads over buckets.
        const double mean_hap = static_cast<double>(hap_total) / nh;
        const bool per_group = call_k == 0;
        const int region_k = per_group ? 0 : choose_rows_per_lane(lens_desc, nh, mean_hap, call_pairs, call_k);
        size_t pos = 0;
        while (pos < order.size()) {
            int K = !per_group ? region_k : h->group_rows_table.empty() ? choose_group_rows_per_lane(lens_desc[pos]) : h->group_rows_table[lens_desc[pos]];
            // everything in double: from 10 rows per lane on the fp64 pass no longer fits the register file and
            // spills, so reads that fit 8 rows per lane (up to 510 bases) take 8
            if (h->force_fp64 && h->force_rows_per_lane == 0 && K > 8 && lens_desc[pos] + 2 <= 8 * kWave &&
                (per_group || lens_desc.front() + 2 <= 8 * kWave)) K = 8;
            if (plans.empty() || plans.back().region != k || plans.back().K != K) {
                rp.K = K;
                rp.group_begin = static_cast<int>(groups.size());
                rp.num_groups = 0;
                rp.cost = 0;
                rp.longest = lens_desc[pos];
                rp.seg_reads = 0;
                plans.push_back(rp);
            }
            GroupDesc g{static_cast<int32_t>(segs.size()), 0, 0};
            int used = 0, max_seg = 0;
            rows_allocated += static_cast<int64_t>(kWave) * K * nh;
            while (pos < order.size()) {
                const int r = order[pos];
                const int len = lens_desc[pos];
                const int seg = (len + 1 + K) / K;  // rows >= len + 2: a pad row in front, the sum row behind
                if (used + seg > kWave) break;
                rows_used += static_cast<int64_t>(len) * nh;
                segs.push_back(SegDesc{rp.read_begin + r, static_cast<uint8_t>(used), static_cast<uint8_t>(seg), static_cast<uint16_t>(len)});
                read_info[rp.read_begin + r].rows_per_lane = K;
                used += seg;
                max_seg = std::max(max_seg, seg);
                ++g.num_segs; ++pos;
            }
            g.fill = static_cast<int16_t>(max_seg - 1);
            groups.push_back(g);
            R->group_segs.push_back(static_cast<uint8_t>(g.num_segs));
            ++plans.back().num_groups;
            plans.back().seg_reads += g.num_segs;
            plans.back().cost = std::max(plans.back().cost, rp.hap_max + max_seg - 1);
        }
        read_cursor += nr;
        hap_cursor += nh;
    }

    PLAN_CLOCK(1);
    // ---- buckets: one launch per rows-per-lane value; longest work items first -----------------
    std::vector<int> by_cost(plans.size());
    {
        // rows per lane ascending, cost descending, plan order on ties: one integer key per plan
        std::vector<uint64_t> keys(plans.size());
        for (size_t i = 0; i < plans.size(); ++i)
            keys[i] = (static_cast<uint64_t>(plans[i].K) << 52) | (static_cast<uint64_t>(0xFFFFF - std::min(plans[i].cost, 0xFFFFF)) << 32) | i;
        std::sort(keys.begin(), keys.end());
        for (size_t i = 0; i < keys.size(); ++i) by_cost[i] = static_cast<int>(keys[i] & 0xFFFFFFFFu);
    }
    int64_t total_tasks = 0, stream_pairs = 0;
    int8_t stream_bucket_of_rows[sizeof(DeviceBatch::stream_bucket_of_rows)];
    memset(stream_bucket_of_rows, -1, sizeof stream_bucket_of_rows);
    for (size_t i = 0; i < by_cost.size();) {
        const int K = plans[by_cost[i]].K;
        Bucket bk;
        bk.rows_per_lane = K;
        bk.task_begin = static_cast<int>(total_tasks);
        bk.desc_begin = static_cast<int>(R->descs.size());
        int64_t t = 0;
        const size_t bucket_first = i;
        for (; i < by_cost.size() && plans[by_cost[i]].K == K; ++i) {
            const RegionPlan& rp = plans[by_cost[i]];
            RegionDesc d{};
            d.task_begin = static_cast<int32_t>(t);
            d.group_begin = rp.group_begin;
            d.num_groups = rp.num_groups;
            d.hap_begin = rp.hap_begin;
            d.num_haps = rp.nh;
            R->descs.push_back(d);
            t += static_cast<int64_t>(rp.num_groups) * rp.nh;
            bk.max_hap_len = std::max(bk.max_hap_len, rp.hap_max);
        }
        if (total_tasks + t >= (int64_t{1} << 31)) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "more than 2^31 work items in one call");
        bk.num_tasks = static_cast<int>(t);
        bk.num_descs = static_cast<int>(R->descs.size()) - bk.desc_begin;
        bk.emit_mode_fp64 = any_hap_n ? kEmitCompareN : kEmitCompare;
        bk.emit_mode = bk.emit_mode_fp64;
        // Haplotype runs (the stream pass), unless the call is too small to give any wave a run: it then takes
        // the one-haplotype pass, which writes log10 results itself - one kernel less on a call whose time is
        // the chain of its launches.
        const bool runs = h->stream_mode > 0 || (h->stream_mode < 0 && call_pairs >= kRunsFromPairsPerSlot * static_cast<int64_t>(h->wave_slots));
        if (!any_hap_other && h->emit_mode != 0) {
            const int lut = any_hap_n ? kEmitLut5 : kEmitLut4;
            // a workgroup's LDS must fit the 64 KiB a block may ask for; the stream pass keeps only the table
            // there, the one-haplotype pass also the longest haplotype of the launch
            const int lds = runs ? stream_lds_bytes_per_wave(K, lut) : lds_bytes_per_wave(K, lut, bk.max_hap_len);
            if (lds * kWavesPerBlock <= 64 * 1024) bk.emit_mode = lut;
        }
        // fp32 stream pass: the haplotype axis of every region is cut into runs; a run is as long as
        // keeps every wave slot of the device supplied twice over, so runs shrink towards the end of
        // the launch and the last waves to finish are short ones
        bk.stream = (bk.emit_mode == kEmitLut4 || bk.emit_mode == kEmitLut5) && runs;
        const int bucket_index = static_cast<int>(R->buckets.size());
        if (bucket_index >= kMaxBuckets) return fail(GATK_PAIRHMM_ILLEGAL_STATE, "more launch buckets than rows-per-lane choices");
        if (bk.stream) {
            bk.stream_desc_begin = static_cast<int>(stream_descs.size());
            int64_t remaining = t, st = 0;
            for (size_t j = bucket_first; j < i; ++j) {
                const RegionPlan& rp = plans[by_cost[j]];
                for (int a = 0; a < rp.nh;) {
                    int64_t c = remaining / (h->stream_run_divisor * static_cast<int64_t>(h->wave_slots));
                    c = std::max<int64_t>(1, std::min<int64_t>(c, std::min(h->max_stream_haps, kMaxStreamHaps)));
                    c = std::min<int64_t>(c, rp.nh - a);
                    StreamDesc d{};
                    d.task_begin = static_cast<int32_t>(st);
                    d.group_begin = rp.group_begin;
                    d.num_groups = rp.num_groups;
                    d.hap_begin = rp.hap_begin + a;
                    d.num_haps = static_cast<int32_t>(c);
                    d.hap_local_begin = a;
                    stream_descs.push_back(d);
                    st += rp.num_groups;
                    remaining -= static_cast<int64_t>(rp.num_groups) * c;
                    a += static_cast<int>(c);
                }
            }
            bk.num_stream_descs = static_cast<int>(stream_descs.size()) - bk.stream_desc_begin;
            bk.num_stream_tasks = static_cast<int>(st);
            R->any_stream = true;
        }
        if (bk.stream) {
            bk.pair_begin = stream_pairs;
            int read_max = 1;
            for (size_t j = bucket_first; j < i; ++j) {
                const RegionPlan& rp = plans[by_cost[j]];
                read_max = std::max(read_max, rp.longest);
                bk.num_pairs += static_cast<int64_t>(rp.seg_reads) * rp.nh;
            }
            stream_bucket_of_rows[K] = static_cast<int8_t>(bucket_index);
            // The pair pass picks its own rows per lane: at most 8 (more spills in double), the value that
            // wastes the fewest lane-rows for the bucket's longest read, the smaller one on a tie (the
            // pass usually has too few pairs to fill the device, so a short sweep matters more than
            // instruction count).
            double best_util = -1.0;
            for (int k : kRowsPerLaneChoices) {
                if (k < (read_max + 1 + kWave) / kWave) continue;
                if (k > 8 && best_util >= 0.0) break;
                const int lanes = (read_max + 1 + k) / k;
                const double util = static_cast<double>(kWave / lanes) * (read_max + 2) / (static_cast<double>(kWave) * k);
                if (util > best_util + 0.02) { best_util = util; bk.pair_rows_per_lane = k; bk.slot_lanes = lanes; }
            }
            stream_pairs += bk.num_pairs;
        }
        total_tasks += t;
        R->buckets.push_back(bk);
    }
    R->total_pairs = call_pairs;
    R->total_reads = total_reads;
    R->total_haps = total_haps;
    R->region_reads.assign(num_regions, 0); R->region_haps.assign(num_regions, 0);
    R->region_read_begin.assign(num_regions, 0); R->region_hap_begin.assign(num_regions, 0);
    {
        int rb = 0, hb = 0;
        for (int k = 0; k < num_regions; ++k) {
            const int nr = regions[k].num_reads, nh = nr == 0 ? 0 : regions[k].num_haps;
            R->region_read_begin[k] = rb; R->region_hap_begin[k] = hb;
            if (nr == 0 || nh == 0) continue;
            R->region_reads[k] = nr; R->region_haps[k] = nh;
            rb += nr; hb += nh;
        }
    }
    R->cells = cells;
    R->lane_rows_used = rows_used;
    R->lane_rows_allocated = rows_allocated;
    R->total_tasks = total_tasks;
    R->total_read_bytes = total_read_bytes;
    R->num_neg_inf = static_cast<int32_t>(neg_inf_idx.size());

    PLAN_CLOCK(2);
    // ---- pass 2: the rest of the call's image in the pinned staging buffer, one H2D copy -------------
    const size_t o_mapq = have_mapq ? L.add(static_cast<size_t>(total_reads) + 16) : 0;
    const size_t o_roff = L.add((total_reads + 1) * sizeof(int32_t));
    const size_t o_rout = L.add((total_reads + 1) * sizeof(int32_t));
    const size_t o_hpool = L.add(static_cast<size_t>(hap_pool_bytes) + 32);
    const size_t o_hoff = L.add((total_haps + 1) * sizeof(int32_t));
    const size_t o_hlen = L.add((total_haps + 1) * sizeof(int32_t));
    const size_t o_groups = L.add((groups.size() + 1) * sizeof(GroupDesc));
    const size_t o_segs = L.add((segs.size() + 1) * sizeof(SegDesc));
    const size_t o_descs = L.add((R->descs.size() + 1) * sizeof(RegionDesc));
    const size_t o_neg = L.add((neg_inf_idx.size() + 1) * sizeof(int32_t));
    const size_t o_hsoff = L.add((total_haps + 1) * sizeof(int32_t));
    const size_t o_sdescs = L.add((stream_descs.size() + 1) * sizeof(StreamDesc));
    const size_t o_rinfo = L.add((total_reads + 1) * sizeof(ReadInfo));
    const size_t o_long = L.add((long_pairs.size() + 1) * sizeof(RerunPair));
    const size_t upload_bytes = L.size;
    if (upload_bytes > stage_in.cap) {  // the estimate fell short: let the copies finish, then move what they wrote
        crew.finish();
        HIP_TRY(stage_in.ensure_keep(upload_bytes, bulk_bytes));
    }
    uint8_t* S = static_cast<uint8_t*>(stage_in.p);
    int32_t* roff = reinterpret_cast<int32_t*>(S + o_roff);
    int32_t* rout = reinterpret_cast<int32_t*>(S + o_rout);
    int32_t* hoff = reinterpret_cast<int32_t*>(S + o_hoff);
    int32_t* hlen = reinterpret_cast<int32_t*>(S + o_hlen);
    int32_t* hsoff = reinterpret_cast<int32_t*>(S + o_hsoff);
    size_t scur = kStreamFrontPad;  // haplotype stream: [pad] then per region  gap hap gap hap ... gap
    size_t rcur = 0, hcur = 0;
    int ri = 0, hi = 0;
    roff[0] = 0;
    for (int k = 0; k < num_regions; ++k) {
        const gatk_pairhmm_batch& b = regions[k];
        const int nr = b.num_reads, nh = nr == 0 ? 0 : b.num_haps;
        if (nr == 0 || nh == 0) continue;
        const int bo = b.read_offsets[0];
        const size_t total = static_cast<size_t>(b.read_offsets[nr] - bo);
        if (have_mapq) {
            if (b.read_mapq) memcpy(S + o_mapq + ri, b.read_mapq, static_cast<size_t>(nr));
            else memset(S + o_mapq + ri, 255, static_cast<size_t>(nr));
        }
        for (int r = 0; r < nr; ++r) {
            rout[ri] = static_cast<int32_t>(R->region_out_off[k] + static_cast<int64_t>(r) * nh);
            roff[++ri] = static_cast<int32_t>(rcur + (b.read_offsets[r + 1] - bo));
        }
        rcur += total;
        ++scur;  // the region's first gap
        for (int a = 0; a < nh; ++a) {
            const int o = b.hap_offsets[a], len = b.hap_offsets[a + 1] - o;
            hsoff[hi] = static_cast<int32_t>(scur);
            scur += static_cast<size_t>(len) + 1;
            hoff[hi] = static_cast<int32_t>(hcur);
            hlen[hi++] = len;
            memcpy(S + o_hpool + hcur, b.hap_bases + o, len);
            const size_t padded = (static_cast<size_t>(len) + 15) & ~size_t{15};
            memset(S + o_hpool + hcur + len, 0, padded - len);
            hcur += padded;
        }
    }
    memset(S + o_hpool + hcur, 0, 32);
    PLAN_CLOCK(3);
    crew.finish();
    PLAN_CLOCK(4);
    if (!groups.empty()) memcpy(S + o_groups, groups.data(), groups.size() * sizeof(GroupDesc));
    if (!segs.empty()) memcpy(S + o_segs, segs.data(), segs.size() * sizeof(SegDesc));
    if (!R->descs.empty()) memcpy(S + o_descs, R->descs.data(), R->descs.size() * sizeof(RegionDesc));
    if (!neg_inf_idx.empty()) memcpy(S + o_neg, neg_inf_idx.data(), neg_inf_idx.size() * sizeof(int32_t));
    if (!stream_descs.empty()) memcpy(S + o_sdescs, stream_descs.data(), stream_descs.size() * sizeof(StreamDesc));
    if (!read_info.empty()) memcpy(S + o_rinfo, read_info.data(), read_info.size() * sizeof(ReadInfo));
    if (!long_pairs.empty()) memcpy(S + o_long, long_pairs.data(), long_pairs.size() * sizeof(RerunPair));
    R->num_long_pairs = static_cast<int32_t>(long_pairs.size());
    R->long_hap_max = long_hap_max;
    R->long_read_max = long_read_max;
    R->haps_acgtn_only = !any_hap_other;
    R->haps_have_n = any_hap_n;
    R->hap_stream_entries = scur + kStreamBackPad;
    if (R->hap_stream_entries >= (size_t{1} << 31)) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "more than 2^31 haplotype bases in one call");
    const double t_planned = now_ms();

    if (!R->stream) R->stream = h->stream;
    hipStream_t s = R->stream;
    HIP_TRY(R->d_arena.ensure(L.size));
    const void* const out_before = R->d_out.p;
    HIP_TRY(R->d_out.ensure(kCounterBytes + std::max<size_t>(static_cast<size_t>(call_pairs) * sizeof(double), 16)));
    if (R->d_out.p != out_before) R->counters_clear = false;
    HIP_TRY(R->d_rerun_lists.ensure(std::max<size_t>(static_cast<size_t>(total_tasks) * sizeof(int32_t), 16)));
    HIP_TRY(R->d_row_consts.ensure((rbytes + 2 * static_cast<size_t>(total_reads) + 2) * sizeof(RowConst)));
    if (R->any_stream) {
        HIP_TRY(R->d_hap_stream.ensure(R->hap_stream_entries * sizeof(uint16_t)));
        HIP_TRY(R->d_raw_out.ensure(std::max<size_t>(static_cast<size_t>(call_pairs) * sizeof(float), 16)));
        HIP_TRY(R->d_pair_lists.ensure(std::max<size_t>(static_cast<size_t>(stream_pairs) * sizeof(RerunPair), 16)));
    }
    // upload_bytes is a multiple of 256 and the staging buffer is at least as large
    if (upload_bytes > 0 && upload_bytes <= h->kernel_upload_bytes) HIP_TRY(launch_upload(S, R->d_arena.p, upload_bytes, s));
    else if (upload_bytes) HIP_TRY(hipMemcpyAsync(R->d_arena.p, S, upload_bytes, hipMemcpyHostToDevice, s));
    const uint8_t* D = R->d_arena.as<uint8_t>();
    DeviceBatch& B = R->base;
    B = DeviceBatch{};
    B.read_bases = D + o_rb; B.read_quals = D + o_rq;
    B.ins_gop = have_ins ? D + o_ri : nullptr; B.del_gop = have_del ? D + o_rd : nullptr; B.gcp = have_gcp ? D + o_rg : nullptr;
    B.read_off = reinterpret_cast<const int32_t*>(D + o_roff);
    B.read_out_base = reinterpret_cast<const int32_t*>(D + o_rout);
    B.hap_pool = D + o_hpool;
    B.hap_off = reinterpret_cast<const int32_t*>(D + o_hoff);
    B.hap_len = reinterpret_cast<const int32_t*>(D + o_hlen);
    B.groups = reinterpret_cast<const GroupDesc*>(D + o_groups);
    B.segs = reinterpret_cast<const SegDesc*>(D + o_segs);
    B.region_descs = reinterpret_cast<const RegionDesc*>(D + o_descs);
    B.total_read_bytes = total_read_bytes;
    B.read_mapq = have_mapq ? D + o_mapq : nullptr;
    B.num_reads = static_cast<int32_t>(total_reads);
    B.num_haps = static_cast<int32_t>(total_haps);
    B.hap_stream_off = reinterpret_cast<const int32_t*>(D + o_hsoff);
    B.stream_descs = reinterpret_cast<const StreamDesc*>(D + o_sdescs);
    B.read_info = reinterpret_cast<const ReadInfo*>(D + o_rinfo);
    B.long_pairs = reinterpret_cast<const RerunPair*>(D + o_long);
    B.num_long_pairs = R->num_long_pairs;
    B.hap_stream = R->d_hap_stream.as<uint16_t>();
    B.raw_out = R->d_raw_out.as<float>();
    B.pair_lists = R->d_pair_lists.as<RerunPair>();
    for (size_t bi = 0; bi < R->buckets.size(); ++bi) B.bucket_pair_begin[bi] = static_cast<int32_t>(R->buckets[bi].pair_begin);
    memcpy(B.stream_bucket_of_rows, stream_bucket_of_rows, sizeof stream_bucket_of_rows);
    B.row_consts = R->d_row_consts.as<RowConst>();
    B.neg_inf_idx = reinterpret_cast<const int32_t*>(D + o_neg);
    B.num_neg_inf = R->num_neg_inf;
    B.hap_stream_len = static_cast<int32_t>(R->hap_stream_entries);
    // the staging buffer is reused by the next upload on this handle: that upload waits for this copy
    // A region-sized upload (the kernel copy above) has its event recorded behind the kernels of the pass that
    // follows at once, so that the record does not hold them up; behind a large copy it goes in right away:
    // the next part of an overlapped call must wait for the copy, not for this part's kernels.
    h->stage_in_unrecorded = s;
    h->stage_in_record_due = true;
    if (upload_bytes > h->kernel_upload_bytes || h->in_cut_call)
        if (int rc = record_stage_in_event(h)) return rc;
    R->plan_ms = t_planned - t_begin;
    R->upload_ms = now_ms() - t_planned;
#ifdef PAIRHMM_CALL_CLOCKS
    if (num_regions > 100)
        std::fprintf(stderr, "    plan of %d regions: sizes %.0f us, region walk %.0f, buckets %.0f, layout and small copies %.0f, bulk copies %.0f, descriptors %.0f, enqueue %.0f\n",
                     num_regions, (plan_clock[0] - t_begin) * 1e3, (plan_clock[1] - plan_clock[0]) * 1e3, (plan_clock[2] - plan_clock[1]) * 1e3,
                     (plan_clock[3] - plan_clock[2]) * 1e3, (plan_clock[4] - plan_clock[3]) * 1e3, (t_planned - plan_clock[4]) * 1e3, R->upload_ms * 1e3);
#endif
    return GATK_PAIRHMM_OK;
}

// counters layout: [0] error word, [1 + b] rerun-list length of bucket b (one-haplotype passes),
// [1 + kMaxBuckets + b] pairs queued for the fp64 pair pass of bucket b (stream pass)

int enqueue_compute(gatk_pairhmm* h, gatk_pairhmm_resident* R) {
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = R->stream;
    const bool timing = h->timing && g_num_threads.load() == 1;
    gatk_pairhmm_resident::EventSet ev{};
    if (timing) {
        if (!R->ev_free.empty()) {
            ev = R->ev_free.back();
            R->ev_free.pop_back();
        } else {
            for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
        }
        R->ev_pending.push_back(ev);
        HIP_TRY(hipEventRecord(ev[0], s));
    }
    if (!R->counters_clear) HIP_TRY(hipMemsetAsync(R->counters(), 0, kCounterBytes, s));
    R->counters_clear = false;
    DeviceBatch B = R->base;
    B.err_prob = h->d_err;
    B.match_to_match = h->d_mm;
    B.out = R->out();
    B.error_word = R->counters();
    B.fp32_threshold = h->fp32_threshold;
    B.squash_base_quals = h->squash_base_quals ? 1 : 0;
    B.mismatch_divisor = h->no_tristate_correction ? 1.0 : 3.0;
    B.error_read = R->counters() + kErrorReadSlot;
    B.pair_counts = R->counters() + 1 + kMaxBuckets;
    const bool streaming = R->any_stream && !h->force_fp64;
    // A region-sized call on the one-haplotype pass skips the expansion launch: the forward kernel works the
    // row constants out itself (and makes the quality checks), one launch less in a chain of five.  Not with reads
    // of the long-read pass in the call: their qualities are checked by the expansion launch only.
    const bool rows_in_kernel = !streaming && !h->force_fp64 && R->num_long_pairs == 0 && R->total_pairs < 2 * static_cast<int64_t>(h->wave_slots);
    if (rows_in_kernel) {
        B.row_consts = nullptr;
        if (R->num_neg_inf > 0) {  // only the -infinity fill of zero-length reads is left of the prepare pass
            DeviceBatch P = B;
            P.num_reads = 0;
            HIP_TRY(launch_prepare(P, false, false, s));
        }
    } else {
        HIP_TRY(launch_prepare(B, !h->force_fp64, streaming, s));
    }
    if (R->num_long_pairs > 0) {
        // At most kLongWaves waves at once; a workgroup owns a boundary row of {diagonal term, insertion mass} per wave and
        // one more.  On the call's own stream.  (Tried on MI355X and not kept: a stream of its own, so that a call's few long reads
        // run beside its fp32 pass.  At equal priority nothing overlapped - a wave of this pass needs half a SIMD's
        // registers, the forward launch refills every slot one of its waves leaves, and the pass started when that
        // launch had handed out its last work item; on a stream of the highest priority the call did not come back
        // within three minutes.)
        // Few pairs: a workgroup of several waves works on the chunks of one pair at once, as many as bring the launch
        // up to a wave per SIMD; with enough pairs for that, a wave per pair.
        const int stride = (R->long_hap_max + 63) & ~63;
        const int chunks = ((R->long_read_max + 1 + 8) / 8 + kWave - 1) / kWave;  // 64 lanes x 8 rows each
        const int simds = std::max(h->wave_slots / 3, 1);
        const int waves_per_block = std::max(1, std::min({kLongMaxWaves, chunks, simds / R->num_long_pairs}));
        const size_t per_block = static_cast<size_t>(waves_per_block + 1) * 2 * static_cast<size_t>(stride) * sizeof(double);
        const int blocks = static_cast<int>(std::min<size_t>({static_cast<size_t>(R->num_long_pairs), static_cast<size_t>(kLongWaves / waves_per_block),
                                                               std::max<size_t>(kLongBoundaryBytes / per_block, 32)}));
        HIP_TRY(R->d_long_boundary.ensure(static_cast<size_t>(blocks) * per_block));
        DeviceBatch Lg = B;
        Lg.long_boundary = R->d_long_boundary.as<double>();
        Lg.long_boundary_stride = stride;
        Lg.long_withhold_progress = h->debug_long_stall ? 1 : 0;
        // Emission terms from a per-wave LDS table (a quarter faster per wave, but 16 KiB of LDS a wave, 20 with a column
        // for 'N') when a CU holds as many waves with it as without, or when every pair of the call is resident at once
        // even so; byte compares otherwise.
        int table_codes = 0;
        if (R->haps_acgtn_only && h->emit_mode != 0 && long_table_fits(waves_per_block)) {
            const int codes = R->haps_have_n ? 5 : 4;
            const int lds_cu = 160 * 1024, most = 8 / waves_per_block;  // two waves a SIMD by registers
            const int with_table = std::min(lds_cu / long_lds_bytes(waves_per_block, R->long_hap_max, codes), most);
            const int without = std::min(lds_cu / long_lds_bytes(waves_per_block, R->long_hap_max, 0), most);
            if (with_table >= without || R->num_long_pairs <= with_table * (simds / 4)) table_codes = codes;
        }
        Lg.long_table_codes = table_codes;
        HIP_TRY(launch_long(Lg, blocks, waves_per_block, R->long_hap_max, s));
    }
    if (timing) HIP_TRY(hipEventRecord(ev[1], s));
    auto bucket_args = [&](size_t bi, int mode) {
        const Bucket& bk = R->buckets[bi];
        DeviceBatch L = B;
        L.region_descs = B.region_descs + bk.desc_begin;
        L.num_region_descs = bk.num_descs;
        L.num_tasks = bk.num_tasks;
        L.rerun_count = R->counters() + 1 + bi;
        L.rerun_list = R->d_rerun_lists.as<int32_t>() + bk.task_begin;
        L.lds_per_wave = lds_bytes_per_wave(bk.rows_per_lane, mode, bk.max_hap_len);
        return L;
    };
    for (size_t bi = 0; bi < R->buckets.size(); ++bi) {
        const Bucket& bk = R->buckets[bi];
        const int mode = h->force_fp64 ? bk.emit_mode_fp64 : bk.emit_mode;
        if (streaming && bk.stream) {
            DeviceBatch L = bucket_args(bi, mode);
            L.stream_descs = B.stream_descs + bk.stream_desc_begin;
            L.num_stream_descs = bk.num_stream_descs;
            L.num_tasks = bk.num_stream_tasks;
            L.lds_per_wave = std::max(stream_lds_bytes_per_wave(bk.rows_per_lane, mode), h->stream_lds_floor);
            HIP_TRY(launch_stream(L, bk.rows_per_lane, mode, s));
            continue;
        }
        const int blocks = (bk.num_tasks + kWavesPerBlock - 1) / kWavesPerBlock;
        HIP_TRY(launch_forward(bucket_args(bi, mode), bk.rows_per_lane, mode, h->force_fp64, false, blocks, s));
    }
    if (timing) HIP_TRY(hipEventRecord(ev[2], s));
    if (streaming) HIP_TRY(launch_finish_stream(B, s));
    if (timing) HIP_TRY(hipEventRecord(ev[3], s));
    if (!h->force_fp64) {
        for (size_t bi = 0; bi < R->buckets.size(); ++bi) {
            const Bucket& bk = R->buckets[bi];
            if (streaming && bk.stream) {  // only the pairs that asked for it, repacked
                DeviceBatch L = B;
                L.pair_count = B.pair_counts + bi;
                L.pair_list = B.pair_lists + bk.pair_begin;
                L.slot_lanes = bk.slot_lanes;
                const int64_t waves = (bk.num_pairs + kWave / bk.slot_lanes - 1) / (kWave / bk.slot_lanes);
                HIP_TRY(launch_pairs(L, bk.pair_rows_per_lane, static_cast<int>(std::min<int64_t>(waves, 8192)), s));
                continue;
            }
            if (bk.rows_per_lane <= kInlineRerunRows) continue;  // the fp32 pass has redone those items itself
            const int blocks = std::min((bk.num_tasks + kWavesPerBlock - 1) / kWavesPerBlock, 1024);
            HIP_TRY(launch_forward(bucket_args(bi, bk.emit_mode_fp64), bk.rows_per_lane, bk.emit_mode_fp64, true, true, blocks, s));
        }
    }
    if (timing) HIP_TRY(hipEventRecord(ev[4], s));
    if (int rc = record_stage_in_event(h)) return rc;
    // Hand the queued work to the device now: the runtime may otherwise hold launches back until the
    // host next waits on this stream, which in an overlapped call is only after the next part has been planned.
    {
        const hipError_t q = hipStreamQuery(s);
        if (q != hipSuccess && q != hipErrorNotReady) HIP_TRY(q);
    }
    R->streamed = streaming;
    R->computed = true;
    return GATK_PAIRHMM_OK;
}

// Which wave-group a bucket-local work item belongs to (host copy of the kernel's decoding).
int group_of_task(const gatk_pairhmm_resident* R, const Bucket& bk, int task) {
    int lo = bk.desc_begin, hi = bk.desc_begin + bk.num_descs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (R->descs[mid].task_begin <= task) lo = mid; else hi = mid - 1;
    }
    const RegionDesc& d = R->descs[lo];
    return d.group_begin + (task - d.task_begin) % d.num_groups;
}

int digest_counters(gatk_pairhmm* h, gatk_pairhmm_resident* R, const int32_t* counters);

// Wait for the stream, turn the device error word into the reference's exception code, fill stats.
int finish_compute(gatk_pairhmm* h, gatk_pairhmm_resident* R) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(R->stream));
    int32_t counters[kErrorReadSlot + 1] = {0};
    HIP_TRY(hipMemcpy(counters, R->counters(), sizeof counters, hipMemcpyDeviceToHost));
    return digest_counters(h, R, counters);
}

// The stream is idle and `counters` is the host copy of the pass's counters: stats and error code.
int digest_counters(gatk_pairhmm* h, gatk_pairhmm_resident* R, const int32_t* counters) {
    gatk_pairhmm_stats st{};
    st.pairs = R->total_pairs;
    st.cells = R->cells;
    st.wave_tasks = R->total_tasks;
    st.host_plan_ms = R->plan_ms;
    st.host_upload_ms = R->upload_ms;
    st.lane_rows_used = R->lane_rows_used;
    st.lane_rows_allocated = R->lane_rows_allocated;
    {
        int64_t most = -1;
        for (const Bucket& bk : R->buckets) {
            const bool stream = R->streamed && bk.stream;
            const int64_t items = stream ? bk.num_stream_tasks : bk.num_tasks;
            if (items <= most) continue;
            most = items;
            st.main_rows_per_lane = bk.rows_per_lane;
            st.main_emit_mode = h->force_fp64 ? bk.emit_mode_fp64 : bk.emit_mode;
            st.main_stream_pass = stream ? 1 : 0;
            st.main_fp64 = h->force_fp64 ? 1 : 0;
        }
    }
    for (size_t bi = 0; bi < R->buckets.size(); ++bi) {
        const Bucket& bk = R->buckets[bi];
        if (R->streamed && bk.stream) {
            const int64_t n = std::min<int64_t>(counters[1 + kMaxBuckets + bi], bk.num_pairs);
            const int per_wave = kWave / bk.slot_lanes;
            st.rerun_pairs += n;
            st.rerun_tasks += (n + per_wave - 1) / per_wave;
            continue;
        }
        const int n = std::min(counters[1 + bi], bk.num_tasks);
        st.rerun_tasks += n;
        if (n > 0) {
            std::vector<int32_t> list(n);
            HIP_TRY(hipMemcpy(list.data(), R->d_rerun_lists.as<int32_t>() + bk.task_begin, n * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (int t : list) st.rerun_pairs += R->group_segs[group_of_task(R, bk, t)];
        }
    }
    if (R->ev_pending.empty()) {  // nothing timed since the last wait on this batch: its timings still stand
        st.fp32_kernel_ms = R->last_timing.fp32_kernel_ms;
        st.fp64_kernel_ms = R->last_timing.fp64_kernel_ms;
        st.aux_kernel_ms = R->last_timing.aux_kernel_ms;
        st.timed_passes = R->last_timing.timed_passes;
    }
    for (const auto& ev : R->ev_pending) {  // every pass enqueued since the last wait
        float pre = 0, fwd = 0, fin = 0, rerun = 0;
        HIP_TRY(hipEventElapsedTime(&pre, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&fwd, ev[1], ev[2]));
        HIP_TRY(hipEventElapsedTime(&fin, ev[2], ev[3]));
        HIP_TRY(hipEventElapsedTime(&rerun, ev[3], ev[4]));
        st.fp32_kernel_ms += fwd;
        st.fp64_kernel_ms += rerun;
        st.aux_kernel_ms += pre + fin;
        ++st.timed_passes;
        R->ev_free.push_back(ev);
    }
    R->ev_pending.clear();
    R->last_timing = st;
    h->stats = st;
    h->error_region = h->error_read = -1;
    if (counters[0] & kStallBit)
        return fail(GATK_PAIRHMM_DEVICE_ERROR, "the long-read pass gave up waiting for a wave of its workgroup; the call's results are not valid");
    if (counters[0] & 3) {
        // which read: the first one (in the call's read order) that had an input error of either kind
        int region = -1, read_in_region = -1;
        if (counters[kErrorReadSlot] > 0) {
            const int read = 0x7fffffff - counters[kErrorReadSlot];
            int k = static_cast<int>(std::upper_bound(R->region_read_begin.begin(), R->region_read_begin.end(), read) - R->region_read_begin.begin()) - 1;
            while (k > 0 && R->region_reads[k] == 0) --k;  // regions without reads share their successor's first index
            if (k >= 0 && k < R->num_regions) { region = R->region_label_base + k; read_in_region = read - R->region_read_begin[k]; }
        }
        return fail_input_error(h, (counters[0] & 1) ? GATK_PAIRHMM_ILLEGAL_ARGUMENT : GATK_PAIRHMM_INDEX_OUT_OF_BOUNDS, region, read_in_region, R->regions_are_pieces);
    }
    return GATK_PAIRHMM_OK;
}

int fail_input_error(gatk_pairhmm* h, int code, int region, int read, bool regions_are_pieces) {
    h->error_region = region;
    h->error_read = read;
    std::string where;
    if (region >= 0)
        where = " (region " + std::to_string(region) + (regions_are_pieces ? " of the call's read-range pieces" : "") + ", read " + std::to_string(read) + ")";
    if (code == GATK_PAIRHMM_ILLEGAL_ARGUMENT) return fail(code, "insert quality, deletion quality or gcp cannot be less than 0 (byte >= 128)" + where);
    return fail(code, "base quality 255 is outside the 255-entry error-probability cache" + where);
}

int download_begin(gatk_pairhmm* h, gatk_pairhmm_resident* R, double* const* outs, PendingDownload& d) {
    d = PendingDownload{};
    d.active = true;
    if (R->total_pairs == 0) return GATK_PAIRHMM_OK;  // end waits for the stream and digests
    int with_output = 0;
    for (int k = 0; k < R->num_regions; ++k) {
        if (R->region_out_off[k + 1] == R->region_out_off[k]) continue;
        if (!outs || !outs[k]) {
            (void)finish_compute(h, R);
            d.active = false;
            return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "likelihoods cannot be null");
        }
        ++with_output;
    }
    const size_t bytes = static_cast<size_t>(R->total_pairs) * sizeof(double);
    if (with_output == 1 && bytes > (size_t{256} << 10)) {  // a large single matrix: straight into the caller's array
        d.direct = true;
        return GATK_PAIRHMM_OK;
    }
    // counters and likelihoods in one device-to-host copy into pinned memory, one wait, then scatter on the host
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(h->stage_out.ensure(kCounterBytes + bytes));
    // A small result goes home through a kernel on the compute stream (device to pinned host memory; it zeroes the
    // counters for the next pass on its way): the runtime's own copy starts 5-6 us after the kernel in front of it, a
    // kernel follows a kernel at once.
    const size_t down_bytes = (kCounterBytes + bytes + 255) & ~size_t{255};
    const bool by_kernel = down_bytes <= h->kernel_download_bytes && down_bytes <= h->stage_out.cap && down_bytes <= R->d_out.cap &&
                           static_cast<void*>(R->counters()) == R->d_out.p;
    // (not with pass timings pending: their events are read right after, and only a wait through the runtime makes them readable)
    d.by_signal = by_kernel && down_bytes <= h->signal_download_bytes && R->ev_pending.empty();
    d.landed = static_cast<const uint8_t*>(h->stage_out.p);
    if (d.by_signal) {
        if (!h->done_flag.p) {
            HIP_TRY(h->done_flag.ensure(64, true));
            *static_cast<volatile unsigned*>(h->done_flag.p) = 0;
        }
        HIP_TRY(h->signal_out.ensure(down_bytes, true));
        d.landed = static_cast<const uint8_t*>(h->signal_out.p);
        d.seq = ++h->done_seq ? h->done_seq : ++h->done_seq;  // never 0, the flag's initial value
        HIP_TRY(launch_download_and_signal(R->d_out.p, h->signal_out.p, down_bytes, kCounterBytes, static_cast<unsigned*>(h->done_flag.p), d.seq, R->stream));
        R->counters_clear = true;
        return GATK_PAIRHMM_OK;
    }
    if (by_kernel) HIP_TRY(launch_download(R->d_out.p, h->stage_out.p, down_bytes, kCounterBytes, R->stream));
    else HIP_TRY(hipMemcpyAsync(h->stage_out.p, R->d_out.p, kCounterBytes + bytes, hipMemcpyDeviceToHost, R->stream));
    // the host waits for the copy only; the runtime's copy leaves the counters to a memset behind it
    if (!R->ev_downloaded) HIP_TRY(hipEventCreateWithFlags(&R->ev_downloaded, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(R->ev_downloaded, R->stream));
    if (!by_kernel) HIP_TRY(hipMemsetAsync(R->counters(), 0, kCounterBytes, R->stream));
    R->counters_clear = true;
    return GATK_PAIRHMM_OK;
}

bool download_ready(gatk_pairhmm* h, gatk_pairhmm_resident* R, const PendingDownload& d) {
    if (!d.active) return true;
    if (R->total_pairs == 0 || d.direct) return hipStreamQuery(R->stream) != hipErrorNotReady;
    if (d.by_signal) return __atomic_load_n(static_cast<const unsigned*>(h->done_flag.p), __ATOMIC_ACQUIRE) == d.seq || hipStreamQuery(R->stream) != hipErrorNotReady;
    return hipEventQuery(R->ev_downloaded) != hipErrorNotReady;
}

int download_end(gatk_pairhmm* h, gatk_pairhmm_resident* R, double* const* outs, PendingDownload& d) {
    d.active = false;
    if (R->total_pairs == 0) return finish_compute(h, R);
    if (d.direct) {
        if (int rc = finish_compute(h, R)) return rc;
        int last = -1;
        for (int k = 0; k < R->num_regions; ++k)
            if (R->region_out_off[k + 1] != R->region_out_off[k]) last = k;
        HIP_TRY(hipMemcpy(outs[last], R->out(), static_cast<size_t>(R->total_pairs) * sizeof(double), hipMemcpyDeviceToHost));
        return GATK_PAIRHMM_OK;
    }
    HIP_TRY(hipSetDevice(h->device));
    if (d.by_signal) {
        // A region-sized pass is over in tens of microseconds: poll for them, then leave the rest to the runtime
        // (which also reports a failed launch; the flag alone would never come).
        const unsigned* const flag = static_cast<const unsigned*>(h->done_flag.p);
        bool waited = false;
        const double t_poll = now_ms();
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == d.seq) { waited = true; break; }
            __builtin_ia32_pause();
            if ((spins & 1023u) == 1023u && now_ms() - t_poll > 2.0) break;
        }
        if (!waited) HIP_TRY(hipStreamSynchronize(R->stream));
    } else {
        HIP_TRY(hipEventSynchronize(R->ev_downloaded));
    }
    if (int rc = digest_counters(h, R, reinterpret_cast<const int32_t*>(d.landed))) return rc;
    const double* src = reinterpret_cast<const double*>(d.landed + kCounterBytes);
    for (int k = 0; k < R->num_regions; ++k) {
        const int64_t n = R->region_out_off[k + 1] - R->region_out_off[k];
        if (n > 0) memcpy(outs[k], src + R->region_out_off[k], static_cast<size_t>(n) * sizeof(double));
    }
    return GATK_PAIRHMM_OK;
}

int download_into(gatk_pairhmm* h, gatk_pairhmm_resident* R, double* const* outs) {
    PendingDownload d;
    if (int rc = download_begin(h, R, outs, d)) return rc;
    return download_end(h, R, outs, d);
}

int check_handle(gatk_pairhmm* h) {
    if (!h) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "PairHMM handle cannot be null");
    return GATK_PAIRHMM_OK;
}

gatk_pairhmm* first_engine(gatk_pairhmm* h) {
    if (h && !h->children.empty()) {
        h->stats_in_first_engine = true;
        return h->children[0];
    }
    return h;
}

}  // namespace pairhmm_host

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* gatk_pairhmm_last_error(void) { return g_last_error.c_str(); }
const char* gatk_pairhmm_version(void) { return "gatk_pairhmm_hip 0.1 (gfx950)"; }

int gatk_pairhmm_device_count(int32_t* count) try {
    if (!count) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "count cannot be null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_set_number_of_threads(int32_t num_threads) try {
    g_num_threads.store(num_threads);
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_create(int device, gatk_pairhmm** out) try {
    if (!out) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "out cannot be null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(GATK_PAIRHMM_DEVICE_ERROR, "no HIP device is visible: the PairHMM engine has no CPU fallback");
    if (device < 0 || device >= n) return fail(GATK_PAIRHMM_DEVICE_ERROR, "device index out of range");
    {   // the library holds gfx950 code only: say so here rather than at the first launch
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(GATK_PAIRHMM_DEVICE_ERROR, "device " + std::to_string(device) + " is " + prop.gcnArchName +
                        ": the PairHMM engine is built for gfx950 (MI355X) only");
    }
    HIP_TRY(hipSetDevice(device));
    gatk_pairhmm* h = new gatk_pairhmm();
    h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking);
    int cus = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (cus > 0) h->wave_slots = cus * 12;
    const Tables& t = tables();
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_err), sizeof t.err_prob);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_mm), t.match_to_match.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(h->d_err, t.err_prob, sizeof t.err_prob, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_mm, t.match_to_match.data(), t.match_to_match.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        const int rc = fail(GATK_PAIRHMM_DEVICE_ERROR, std::string("device set-up failed: ") + hipGetErrorString(e));
        gatk_pairhmm_close(h);
        return rc;
    }
    if (const int rc = setup_scratch(device)) {  // once per device, with the device current
        gatk_pairhmm_close(h);
        return rc;
    }
    h->counted = true;
    g_live_handles.fetch_add(1);
    *out = h;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_initialize(gatk_pairhmm* h, int32_t read_max_length, int32_t haplotype_max_length) try {
    if (int rc = check_handle(h)) return rc;
    if (read_max_length <= 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "READ_MAX_LENGTH must be > 0 but got " + std::to_string(read_max_length));
    if (haplotype_max_length <= 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "HAPLOTYPE_MAX_LENGTH must be > 0 but got " + std::to_string(haplotype_max_length));
    h->max_read = read_max_length;
    h->max_hap = haplotype_max_length;
    h->initialized = true;
    for (gatk_pairhmm* c : h->children) { c->max_read = read_max_length; c->max_hap = haplotype_max_length; c->initialized = true; }
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

static int compute_regions_multi(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions, double* const* likelihoods);

}  // extern "C"

// compute_regions on a single-device handle.  checked: the caller has run check_region on every region already.
#ifdef PAIRHMM_CALL_CLOCKS
#define CALL_CLOCK(label) std::fprintf(stderr, "  %8.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - call_t0).count(), label)
#else
#define CALL_CLOCK(label)
#endif
int pairhmm_host::compute_regions_single(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions,
                                  double* const* likelihoods, bool checked) {
#ifdef PAIRHMM_CALL_CLOCKS
    const auto call_t0 = std::chrono::steady_clock::now();
#endif
    // A call of many regions is checked part by part, as each part is planned: walking the offset tables of two
    // thousand regions first would keep the device waiting for a fifth of a millisecond, on cache lines the planner
    // then has to fetch again.  Only counts and pointers are looked at beforehand (what the cut rules read).
    bool checked_by_part = false;
    if (!checked) {
        checked_by_part = num_regions >= kCheckByPartFromRegions && h->pipeline_pairs > 0;
        for (int k = 0; k < num_regions; ++k) {
            if (checked_by_part) {
                if (int rc = check_region_arrays(regions[k], k)) return rc;
                continue;
            }
            int read_max = 0, hap_max = 0;
            if (int rc = check_region(regions[k], k, &read_max, &hap_max)) return rc;
            note_call_maxima(h, regions[k], read_max, hap_max);
        }
    }
    CALL_CLOCK("checked");
    const RegionCallCount in_flight(h->device);
    // How much of a part a region takes: its pairs, and for a region of long reads against long haplotypes (more
    // than kLongPairCells cells per pair; "pipeline_pairs" was measured on pairs of 50 to 90 thousand) its cells over
    // "pipeline_cells_per_pair", so that such a call is cut into parts of about the same device time: the fp64 pair
    // pass of one part, which long high-indel reads need, then runs beside the fp32 pass of the next.
    const int64_t ref_cells = h->pipeline_cells_per_pair;
    auto weight_of = [ref_cells](const gatk_pairhmm_batch& b) -> int64_t { return pipeline_weight(b, ref_cells); };
    // A region worth more than a part holds is cut by read ranges (its pairs are independent,
    // PairHMM.java:217): the pieces share its haplotypes and write into consecutive rows of its array.
    std::vector<gatk_pairhmm_batch> pieces;
    std::vector<double*> piece_outs;
    std::vector<std::pair<int, int>> piece_origin;  // the caller's region and the first of its reads a piece holds
    h->error_region = h->error_read = -1;
    if (h->pipeline_pairs > 0) {
        bool any_large = false;
        for (int k = 0; k < num_regions && !any_large; ++k)
            any_large = regions[k].num_reads > 1 && regions[k].num_haps > 0 && weight_of(regions[k]) >= 2 * h->pipeline_pairs;
        if (any_large) {
            if (checked_by_part) {  // the pieces are not the caller's regions: check those while they are still at hand
                for (int k = 0; k < num_regions; ++k) {
                    int read_max = 0, hap_max = 0;
                    if (int rc = check_region(regions[k], k, &read_max, &hap_max)) return rc;
                    note_call_maxima(h, regions[k], read_max, hap_max);
                }
                checked_by_part = false;
            }
            for (int k = 0; k < num_regions; ++k) {
                const gatk_pairhmm_batch& b = regions[k];
                const int n_pieces = static_cast<int>(std::min<int64_t>(std::max<int64_t>(weight_of(b) / h->pipeline_pairs, 1), std::max(b.num_reads, 1)));
                for (int q = 0; q < n_pieces; ++q) {
                    const int r0 = static_cast<int>(static_cast<int64_t>(b.num_reads) * q / n_pieces);
                    const int r1 = static_cast<int>(static_cast<int64_t>(b.num_reads) * (q + 1) / n_pieces);
                    gatk_pairhmm_batch piece = b;
                    piece.num_reads = r1 - r0;
                    if (b.read_offsets) piece.read_offsets = b.read_offsets + r0;  // offsets stay absolute
                    if (b.read_mapq) piece.read_mapq = b.read_mapq + r0;
                    pieces.push_back(piece);
                    piece_origin.emplace_back(k, r0);
                    double* out = likelihoods ? likelihoods[k] : nullptr;
                    piece_outs.push_back(out ? out + static_cast<int64_t>(r0) * b.num_haps : nullptr);
                }
            }
            regions = pieces.data();
            likelihoods = piece_outs.data();
            num_regions = static_cast<int>(pieces.size());
        }
    }
    std::vector<int64_t> weights(static_cast<size_t>(num_regions));
    for (int k = 0; k < num_regions; ++k) weights[static_cast<size_t>(k)] = weight_of(regions[k]);
    std::vector<int> cuts = part_cuts(weights, CutRules{h->pipeline_pairs, h->pipeline_min_pairs, h->pipeline_mid_ramp_min, h->pipeline_ramp,
                                                        h->pipeline_ramp_steps, in_flight.others > 0});
    const bool are_pieces = !pieces.empty();
    for (gatk_pairhmm_resident* R : {&h->scratch, &h->scratch2, &h->scratch3}) { R->region_label_base = 0; R->regions_are_pieces = false; }
    // an input error found in a read-range piece is reported under the caller's own region and read index
    auto in_callers_terms = [&](int rc) {
        if (rc == GATK_PAIRHMM_OK || !are_pieces || h->error_region < 0 || h->error_region >= static_cast<int>(piece_origin.size())) return rc;
        const std::pair<int, int> at = piece_origin[static_cast<size_t>(h->error_region)];
        return fail_input_error(h, rc, at.first, at.second + h->error_read, false);
    };
    if (cuts.size() <= 2) {
        if (int rc = plan_and_upload(h, &h->scratch, num_regions, regions, !checked_by_part)) return rc;
        if (h->scratch.total_pairs == 0) { h->stats = gatk_pairhmm_stats{}; return GATK_PAIRHMM_OK; }
        if (int rc = enqueue_compute(h, &h->scratch)) return rc;
        return in_callers_terms(download_into(h, &h->scratch, likelihoods));
    }
    // part p is planned and copied while the kernels of the parts before it run; its results are fetched
    // while the kernels of the parts behind it run
    h->scratch.stream = h->stream;
    h->scratch2.stream = h->stream2;
    h->scratch3.stream = h->stream3;
    gatk_pairhmm_resident* slot[3] = {&h->scratch, &h->scratch2, &h->scratch3};
    // depth 3: the host plans part p + 2 without waiting for part p first, so that the device always has a
    // whole part queued behind the one it is working on
    const int depth = h->pipeline_depth >= 3 ? 3 : 2;
    struct InCutCall { bool& flag; explicit InCutCall(bool& f) : flag(f) { flag = true; } ~InCutCall() { flag = false; } } in_cut_call(h->in_cut_call);
    gatk_pairhmm_stats total{};
    int rc = GATK_PAIRHMM_OK;
    const int parts = static_cast<int>(cuts.size()) - 1;
    auto fetch = [&](int p) {
        gatk_pairhmm_resident* R = slot[p % depth];
        const int r = R->total_pairs > 0 ? download_into(h, R, likelihoods ? likelihoods + cuts[p] : nullptr) : GATK_PAIRHMM_OK;
        if (R->total_pairs > 0) {
            const gatk_pairhmm_stats& st = h->stats;
            total.pairs += st.pairs; total.cells += st.cells; total.wave_tasks += st.wave_tasks;
            total.rerun_tasks += st.rerun_tasks; total.rerun_pairs += st.rerun_pairs;
            total.lane_rows_used += st.lane_rows_used; total.lane_rows_allocated += st.lane_rows_allocated;
            total.fp32_kernel_ms += st.fp32_kernel_ms; total.fp64_kernel_ms += st.fp64_kernel_ms; total.aux_kernel_ms += st.aux_kernel_ms;
            total.timed_passes += st.timed_passes;
            total.main_rows_per_lane = st.main_rows_per_lane; total.main_emit_mode = st.main_emit_mode;
            total.main_stream_pass = st.main_stream_pass; total.main_fp64 = st.main_fp64;
        }
        total.host_plan_ms += R->plan_ms; total.host_upload_ms += R->upload_ms;
        return r;
    };
    int fetched = 0;  // parts [0, fetched) are done; parts [fetched, enqueued) are in flight
    int enqueued = 0;
    for (int p = 0; p < parts && rc == GATK_PAIRHMM_OK; ++p) {
        gatk_pairhmm_resident* R = slot[p % depth];
        R->region_label_base = cuts[p];
        CALL_CLOCK("part begins");
        rc = plan_and_upload(h, R, cuts[p + 1] - cuts[p], regions + cuts[p], !checked_by_part);
        CALL_CLOCK("  planned, upload enqueued");
        if (rc == GATK_PAIRHMM_OK && R->total_pairs > 0) rc = enqueue_compute(h, R);
        CALL_CLOCK("  kernels enqueued");
        if (rc == GATK_PAIRHMM_OK) enqueued = p + 1;
        if (rc == GATK_PAIRHMM_OK && p >= depth - 1) { rc = fetch(p - (depth - 1)); fetched = p - (depth - 1) + 1; CALL_CLOCK("  fetched an earlier part"); }
    }
    while (rc == GATK_PAIRHMM_OK && fetched < enqueued) { rc = fetch(fetched); ++fetched; CALL_CLOCK("fetched"); }
    if (rc != GATK_PAIRHMM_OK) {  // leave nothing in flight on buffers the next call will reuse
        const std::string msg = g_last_error;
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->stream2);
        (void)hipStreamSynchronize(h->stream3);
        for (auto* R : slot) {
            for (auto& ev : R->ev_pending) R->ev_free.push_back(ev);
            R->ev_pending.clear();
        }
        g_last_error = msg;
        return in_callers_terms(rc);
    }
    h->stats = total;
    return GATK_PAIRHMM_OK;
}

extern "C" {

int gatk_pairhmm_compute_regions(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions,
                                 double* const* likelihoods) try {
    if (int rc = check_handle(h)) return rc;
    if (num_regions < 0 || (num_regions > 0 && !regions)) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "regions cannot be null");
    if (!h->children.empty()) return compute_regions_multi(h, num_regions, regions, likelihoods);
    return compute_regions_single(h, num_regions, regions, likelihoods, false);
} catch (...) { return on_exception(); }

int gatk_pairhmm_create_multi(const int32_t* devices, int32_t num_devices, gatk_pairhmm** out) try {
    if (!out) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "out cannot be null");
    *out = nullptr;
    std::vector<int> list;
    if (!devices || num_devices <= 0) {  // every visible device
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
            return fail(GATK_PAIRHMM_DEVICE_ERROR, "no HIP device is visible: the PairHMM engine has no CPU fallback");
        for (int d = 0; d < n; ++d) list.push_back(d);
    } else {
        list.assign(devices, devices + num_devices);
    }
    gatk_pairhmm* h = new gatk_pairhmm();
    h->device = list[0];
    for (int d : list) {
        gatk_pairhmm* child = nullptr;
        if (const int rc = gatk_pairhmm_create(d, &child)) {
            const std::string msg = g_last_error;
            gatk_pairhmm_close(h);
            return fail(rc, msg);
        }
        h->children.push_back(child);
    }
    for (size_t k = 1; k < h->children.size(); ++k) {
        h->workers.emplace_back(new DeviceWorker());
        h->workers.back()->start();
    }
    *out = h;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

// compute_regions on a multi-device handle: the regions go to the engines longest-processing-time first by cell
// count (sum of readLen * hapLen), a region worth more than half a device's share is cut by read ranges first
// (its pairs are independent, PairHMM.java:217), every engine plans, copies and scores its share on a host
// thread of its own with its own overlapped pipeline, and writes into the caller's likelihoods[k] directly:
// no gather copy, nothing between devices.
static int compute_regions_multi(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions, double* const* likelihoods) {
    const int n_dev = static_cast<int>(h->children.size());
    std::vector<int64_t> cells(static_cast<size_t>(num_regions), 0);
    int64_t total_cells = 0, total_pairs = 0;
    for (int k = 0; k < num_regions; ++k) {
        int read_max = 0, hap_max = 0;
        if (int rc = check_region(regions[k], k, &read_max, &hap_max)) return rc;
        note_call_maxima(h, regions[k], read_max, hap_max);
        const gatk_pairhmm_batch& b = regions[k];
        if (b.num_reads == 0 || b.num_haps == 0) continue;
        if (!likelihoods || !likelihoods[k]) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "likelihoods cannot be null");
        cells[k] = static_cast<int64_t>(b.read_offsets[b.num_reads] - b.read_offsets[0]) * (b.hap_offsets[b.num_haps] - b.hap_offsets[0]);
        total_cells += cells[k];
        total_pairs += static_cast<int64_t>(b.num_reads) * b.num_haps;
    }
    for (gatk_pairhmm* c : h->children) { c->initialized = h->initialized; c->max_read = h->max_read; c->max_hap = h->max_hap; }
    h->stats_in_first_engine = false;
    if (total_pairs == 0) { h->stats = gatk_pairhmm_stats{}; return GATK_PAIRHMM_OK; }
    if (n_dev == 1 || total_pairs < h->multi_min_pairs) {  // not worth a hand-over between threads
        gatk_pairhmm* c = h->children[h->next_child++ % n_dev];
        const int rc = compute_regions_single(c, num_regions, regions, likelihoods, true);
        h->stats = c->stats;
        return rc;
    }
    // pieces: whole regions, or read ranges of a region worth more than half a device's share
    struct Piece { gatk_pairhmm_batch b; double* out; int64_t cells; int region, first_read; };
    std::vector<Piece> pieces;
    pieces.reserve(static_cast<size_t>(num_regions) + 4 * n_dev);
    const int64_t share = std::max<int64_t>(total_cells / n_dev, 1);
    for (int k = 0; k < num_regions; ++k) {
        const gatk_pairhmm_batch& b = regions[k];
        if (cells[k] == 0) continue;
        int n_pieces = 1;
        if (2 * cells[k] > share) n_pieces = static_cast<int>(std::min<int64_t>((4 * cells[k] + share - 1) / share, b.num_reads));
        for (int q = 0; q < n_pieces; ++q) {
            const int r0 = static_cast<int>(static_cast<int64_t>(b.num_reads) * q / n_pieces);
            const int r1 = static_cast<int>(static_cast<int64_t>(b.num_reads) * (q + 1) / n_pieces);
            if (r1 == r0) continue;
            Piece pc{b, likelihoods[k] + static_cast<int64_t>(r0) * b.num_haps, 0, k, r0};
            pc.b.num_reads = r1 - r0;
            pc.b.read_offsets = b.read_offsets + r0;  // offsets stay absolute
            if (b.read_mapq) pc.b.read_mapq = b.read_mapq + r0;
            pc.cells = static_cast<int64_t>(b.read_offsets[r1] - b.read_offsets[r0]) * (b.hap_offsets[b.num_haps] - b.hap_offsets[0]);
            pieces.push_back(pc);
        }
    }
    // longest processing time first; ties in input order, so the assignment is deterministic
    std::vector<int> order(pieces.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pieces[a].cells > pieces[b].cells; });
    std::vector<std::vector<int>> mine(static_cast<size_t>(n_dev));
    std::vector<int64_t> load(static_cast<size_t>(n_dev), 0);
    for (int i : order) {
        int best = 0;
        for (int d = 1; d < n_dev; ++d)
            if (load[d] < load[best]) best = d;
        mine[best].push_back(i);
        load[best] += pieces[i].cells;
    }
    struct Share {
        std::vector<gatk_pairhmm_batch> batches;
        std::vector<double*> outs;
        std::vector<std::pair<int, int>> origin;  // the caller's region and the first of its reads each batch holds
        int rc = GATK_PAIRHMM_OK;
        std::string error;
    };
    std::vector<Share> shares(static_cast<size_t>(n_dev));
    for (int d = 0; d < n_dev; ++d) {
        std::sort(mine[d].begin(), mine[d].end());  // input order within a device: neighbouring read ranges stay neighbours
        for (int i : mine[d]) {
            shares[d].batches.push_back(pieces[i].b);
            shares[d].outs.push_back(pieces[i].out);
            shares[d].origin.emplace_back(pieces[i].region, pieces[i].first_read);
        }
    }
    // Nothing may leave a share's job as an exception: on a worker thread it would end the process, and on this
    // thread it would unwind `shares` from under the workers that are still scoring into it.
    auto run = [&](int d) noexcept {
        Share& sh = shares[d];
        try {
            if (sh.batches.empty()) { h->children[d]->stats = gatk_pairhmm_stats{}; return; }
            sh.rc = compute_regions_single(h->children[d], static_cast<int32_t>(sh.batches.size()), sh.batches.data(), sh.outs.data(), true);
            if (g_fail_share_for_test == d) throw std::runtime_error("injected failure");
        } catch (...) {
            sh.rc = on_exception();
        }
        if (sh.rc != GATK_PAIRHMM_OK) {
            try { sh.error = g_last_error; } catch (...) {}  // this thread's message
        }
    };
    {
        struct WaitForWorkers {  // every worker is done before `run` and `shares` go, whatever happens on this thread
            gatk_pairhmm* h; int started = 0;
            ~WaitForWorkers() { for (int k = 0; k < started; ++k) h->workers[k]->wait(); }
        } guard{h};
        for (int d = 1; d < n_dev; ++d) { h->workers[d - 1]->submit([&run, d] { run(d); }); guard.started = d; }
        run(0);
    }
    gatk_pairhmm_stats total{};
    for (int d = 0; d < n_dev; ++d) {  // counts add up; times are the slowest device's
        const gatk_pairhmm_stats& st = h->children[d]->stats;
        total.pairs += st.pairs; total.cells += st.cells; total.wave_tasks += st.wave_tasks;
        total.rerun_tasks += st.rerun_tasks; total.rerun_pairs += st.rerun_pairs;
        total.lane_rows_used += st.lane_rows_used; total.lane_rows_allocated += st.lane_rows_allocated;
        total.fp32_kernel_ms = std::max(total.fp32_kernel_ms, st.fp32_kernel_ms);
        total.fp64_kernel_ms = std::max(total.fp64_kernel_ms, st.fp64_kernel_ms);
        total.aux_kernel_ms = std::max(total.aux_kernel_ms, st.aux_kernel_ms);
        total.host_plan_ms = std::max(total.host_plan_ms, st.host_plan_ms);
        total.host_upload_ms = std::max(total.host_upload_ms, st.host_upload_ms);
        total.timed_passes = std::max(total.timed_passes, st.timed_passes);
        if (d == 0) { total.main_rows_per_lane = st.main_rows_per_lane; total.main_emit_mode = st.main_emit_mode;
                      total.main_stream_pass = st.main_stream_pass; total.main_fp64 = st.main_fp64; }
    }
    h->stats = total;
    h->error_region = h->error_read = -1;
    for (int d = 0; d < n_dev; ++d) {
        if (shares[d].rc == GATK_PAIRHMM_OK) continue;
        // an input error is reported under the caller's region and read index, not under the engine's share of them
        const gatk_pairhmm* c = h->children[d];
        if (c->error_region >= 0 && c->error_region < static_cast<int>(shares[d].origin.size())) {
            const std::pair<int, int> at = shares[d].origin[static_cast<size_t>(c->error_region)];
            return fail_input_error(h, shares[d].rc, at.first, at.second + c->error_read, false);
        }
        return fail(shares[d].rc, shares[d].error + " (device " + std::to_string(c->device) + ")");
    }
    return GATK_PAIRHMM_OK;
}

namespace {

// The shared engines of the devices' dispatchers and their device buffers go when the last caller's handle of the
// process closes (no call can be under way then: a call needs a handle); the next shared call creates them again.
void release_shared_engines_if_last() {
    if (g_live_handles.fetch_sub(1) != 1) return;
    release_dispatch_engines();
}

}  // namespace

int gatk_pairhmm_compute_likelihoods(gatk_pairhmm* h, const gatk_pairhmm_batch* batch, double* likelihoods) try {
    if (int rc = check_handle(h)) return rc;
    if (!batch) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "batch cannot be null");
    if (batch->num_reads == 0) return GATK_PAIRHMM_OK;  // PairHMM.java:193
    const int64_t n = static_cast<int64_t>(batch->num_reads) * std::max(batch->num_haps, 0);
    if (n > 0 && !likelihoods) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "likelihoods cannot be null");
    double* outs[1] = {likelihoods};
    // a sample of the region sent ahead with gatk_pairhmm_initialize_region: its likelihoods are there already, or on their way
    bool served = false;
    if (h->region_ticket != 0)
        if (int rc = serve_from_region(h, batch, likelihoods, &served)) return rc;
    if (served) {
        h->stats = gatk_pairhmm_stats{};
        h->stats.pairs = n;
    } else if (h->combine_calls && h->children.empty() && has_default_options(h) && n < 4 * static_cast<int64_t>(h->wave_slots)) {
        // a call that cannot fill the device by itself, on a handle whose other options are the defaults: it shares
        // the device's dispatcher with the region-sized calls other threads make at the same time
        int read_max = 0, hap_max = 0;
        if (int rc = check_region(*batch, 0, &read_max, &hap_max)) return rc;
        if (int rc = compute_shared(h, batch, likelihoods)) return rc;
        h->stats = gatk_pairhmm_stats{};
        h->stats.pairs = n;
        note_call_maxima(h, *batch, read_max, hap_max);
    } else if (int rc = gatk_pairhmm_compute_regions(h, 1, batch, outs)) return rc;
    h->likelihood_array.assign(likelihoods, likelihoods + n);  // mLikelihoodArray, PairHMM.java:206,225
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_compute_likelihoods_ptrs(gatk_pairhmm* h, int32_t num_reads, int32_t num_haps,
                                          const uint8_t* const* read_bases, const uint8_t* const* read_quals,
                                          const uint8_t* const* ins_gop, const uint8_t* const* del_gop,
                                          const uint8_t* const* gcp, const int32_t* read_lengths,
                                          const uint8_t* const* hap_bases, const int32_t* hap_lengths,
                                          double* likelihoods) try {
    if (int rc = check_handle(h)) return rc;
    if (num_reads == 0) return GATK_PAIRHMM_OK;
    if (num_reads < 0 || num_haps < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "negative read or haplotype count");
    if (!read_bases || !read_quals || !ins_gop || !del_gop || !gcp || !read_lengths)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "readBases cannot be null");
    if (num_haps > 0 && (!hap_bases || !hap_lengths)) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "haplotypeBases cannot be null");
    std::vector<uint8_t> rb, rq, ri, rd, rg, hb;
    std::vector<int32_t> ro{0}, ho{0};
    for (int r = 0; r < num_reads; ++r) {
        const int len = read_lengths[r];
        if (len < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "negative read length");
        if (len > 0 && (!read_bases[r] || !read_quals[r] || !ins_gop[r] || !del_gop[r] || !gcp[r]))
            return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "readBases cannot be null");
        rb.insert(rb.end(), read_bases[r], read_bases[r] + len);
        rq.insert(rq.end(), read_quals[r], read_quals[r] + len);
        ri.insert(ri.end(), ins_gop[r], ins_gop[r] + len);
        rd.insert(rd.end(), del_gop[r], del_gop[r] + len);
        rg.insert(rg.end(), gcp[r], gcp[r] + len);
        ro.push_back(static_cast<int32_t>(rb.size()));
    }
    for (int a = 0; a < num_haps; ++a) {
        const int len = hap_lengths[a];
        if (len < 0 || (len > 0 && !hap_bases[a])) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "haplotypeBases cannot be null");
        hb.insert(hb.end(), hap_bases[a], hap_bases[a] + len);
        ho.push_back(static_cast<int32_t>(hb.size()));
    }
    // keep the data pointers non-null for empty payloads
    rb.push_back(0); rq.push_back(0); ri.push_back(0); rd.push_back(0); rg.push_back(0); hb.push_back(0);
    gatk_pairhmm_batch b{num_reads, num_haps, rb.data(), rq.data(), ri.data(), rd.data(), rg.data(), ro.data(), hb.data(), ho.data()};
    return gatk_pairhmm_compute_likelihoods(h, &b, likelihoods);
} catch (...) { return on_exception(); }

int gatk_pairhmm_compute_pair(gatk_pairhmm* h, const uint8_t* hap_bases, int32_t hap_length,
                              const uint8_t* read_bases, const uint8_t* read_quals, const uint8_t* ins_gop,
                              const uint8_t* del_gop, const uint8_t* gcp, int32_t read_length, double* likelihood) try {
    if (int rc = check_handle(h)) return rc;
    // PairHMM.java:271-279
    if (!h->initialized) return fail(GATK_PAIRHMM_ILLEGAL_STATE, "Must call initialize before calling computeReadLikelihoodGivenHaplotypeLog10");
    if (!hap_bases) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "haplotypeBases cannot be null");
    if (hap_length > h->max_hap) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "Haplotype bases is too long, got " + std::to_string(hap_length) + " but max is " + std::to_string(h->max_hap));
    if (!read_bases) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "readBases cannot be null");
    if (read_length > h->max_read) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "readBases is too long, got " + std::to_string(read_length) + " but max is " + std::to_string(h->max_read));
    if (!read_quals || !ins_gop || !del_gop || !gcp) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "Read bases and read quals aren't the same size");
    if (!likelihood) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "likelihood cannot be null");
    if (read_length == 0 && hap_length > 0) { *likelihood = -std::numeric_limits<double>::infinity(); return GATK_PAIRHMM_OK; }
    const int32_t ro[2] = {0, read_length}, ho[2] = {0, hap_length};
    gatk_pairhmm_batch b{1, 1, read_bases, read_quals, ins_gop, del_gop, gcp, ro, hap_bases, ho};
    double* outs[1] = {likelihood};
    return gatk_pairhmm_compute_regions(h, 1, &b, outs);
} catch (...) { return on_exception(); }

int gatk_pairhmm_batch_add(gatk_pairhmm* h, int32_t num_haps, const uint8_t* const* hap_bases,
                           const int32_t* hap_lengths, const uint8_t* read_bases, const uint8_t* read_quals,
                           const uint8_t* ins_gop, const uint8_t* del_gop, const uint8_t* gcp, int32_t read_length) try {
    if (int rc = check_handle(h)) return rc;
    if (num_haps < 0 || read_length < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "negative count");
    if (!read_bases || !read_quals || !ins_gop || !del_gop || !gcp) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "readBases cannot be null");
    if (num_haps > 0 && (!hap_bases || !hap_lengths)) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "haplotypeBases cannot be null");
    gatk_pairhmm::Queued q;
    q.bases.assign(read_bases, read_bases + read_length);
    q.quals.assign(read_quals, read_quals + read_length);
    q.ins.assign(ins_gop, ins_gop + read_length);
    q.del.assign(del_gop, del_gop + read_length);
    q.gcp.assign(gcp, gcp + read_length);
    q.hap_off.push_back(0);
    for (int a = 0; a < num_haps; ++a) {
        if (hap_lengths[a] < 0 || !hap_bases[a]) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "haplotypeBases cannot be null");
        q.haps.insert(q.haps.end(), hap_bases[a], hap_bases[a] + hap_lengths[a]);
        q.hap_off.push_back(static_cast<int32_t>(q.haps.size()));
    }
    for (auto* v : {&q.bases, &q.quals, &q.ins, &q.del, &q.gcp, &q.haps}) v->push_back(0);
    h->queue.push_back(std::move(q));
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_batch_get_result(gatk_pairhmm* h, const double** results, int64_t* count) try {
    if (int rc = check_handle(h)) return rc;
    if (!results || !count) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "results cannot be null");
    const size_t n = h->queue.size();
    int64_t total = 0;
    for (size_t k = 0; k < n; ++k) total += static_cast<int64_t>(h->queue[k].hap_off.size()) - 1;
    h->queue_results.assign(static_cast<size_t>(total), 0.0);
    // Queued reads that follow one another with byte-identical haplotype lists (the usual case: every read of a
    // region is added against the region's haplotypes) become ONE region: its reads then share wave-groups and
    // haplotype runs, instead of one 16-lane read per wavefront.  The results of a region are read-major, which is
    // the queue's order: haplotype-minor per queued read.
    struct Merged {
        std::vector<uint8_t> bases, quals, ins, del, gcp;
        std::vector<int32_t> read_off{0};
        const gatk_pairhmm::Queued* haps = nullptr;
    };
    std::vector<Merged> merged;
    std::vector<int64_t> merged_out;
    int64_t off = 0;
    for (size_t k = 0; k < n; ++k) {
        const gatk_pairhmm::Queued& q = h->queue[k];
        const bool same = !merged.empty() && merged.back().haps->hap_off == q.hap_off && merged.back().haps->haps == q.haps;
        if (!same) {
            merged.emplace_back();
            merged.back().haps = &q;
            merged_out.push_back(off);
        }
        Merged& m = merged.back();
        const size_t len = q.bases.size() - 1;  // every queued array carries one byte of padding
        m.bases.insert(m.bases.end(), q.bases.begin(), q.bases.begin() + len);
        m.quals.insert(m.quals.end(), q.quals.begin(), q.quals.begin() + len);
        m.ins.insert(m.ins.end(), q.ins.begin(), q.ins.begin() + len);
        m.del.insert(m.del.end(), q.del.begin(), q.del.begin() + len);
        m.gcp.insert(m.gcp.end(), q.gcp.begin(), q.gcp.begin() + len);
        m.read_off.push_back(static_cast<int32_t>(m.bases.size()));
        off += static_cast<int64_t>(q.hap_off.size()) - 1;
    }
    std::vector<gatk_pairhmm_batch> regions(merged.size());
    std::vector<double*> outs(merged.size());
    for (size_t g = 0; g < merged.size(); ++g) {
        Merged& m = merged[g];
        for (auto* v : {&m.bases, &m.quals, &m.ins, &m.del, &m.gcp}) v->push_back(0);  // data pointers stay non-null for empty reads
        regions[g] = gatk_pairhmm_batch{static_cast<int32_t>(m.read_off.size()) - 1, static_cast<int32_t>(m.haps->hap_off.size()) - 1,
                                        m.bases.data(), m.quals.data(), m.ins.data(), m.del.data(), m.gcp.data(), m.read_off.data(),
                                        m.haps->haps.data(), m.haps->hap_off.data(), nullptr};
        outs[g] = h->queue_results.data() + merged_out[g];
    }
    const int rc = n ? gatk_pairhmm_compute_regions(h, static_cast<int32_t>(regions.size()), regions.data(), outs.data()) : GATK_PAIRHMM_OK;
    h->queue.clear();
    if (rc) return rc;
    *results = h->queue_results.data();
    *count = total;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_get_likelihood_array(gatk_pairhmm* h, const double** array, int64_t* count) try {
    if (int rc = check_handle(h)) return rc;
    if (!array || !count) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "array cannot be null");
    *array = h->likelihood_array.data();
    *count = static_cast<int64_t>(h->likelihood_array.size());
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_finalize_region(gatk_pairhmm* h) try {
    if (int rc = check_handle(h)) return rc;
    h->queue.clear();
    drop_region(h);
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_close(gatk_pairhmm* h) try {
    if (!h) return GATK_PAIRHMM_OK;
    drop_tickets(h);  // nothing may write into this handle's requests any more
    if (!h->children.empty() || !h->workers.empty()) {  // a multi-device handle owns engines and threads only
        for (auto& w : h->workers) w->shutdown();
        for (gatk_pairhmm* c : h->children) gatk_pairhmm_close(c);
        delete h;
        return GATK_PAIRHMM_OK;
    }
    if (h->counted) release_shared_engines_if_last();
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    if (h->stream3) (void)hipStreamSynchronize(h->stream3);
    h->scratch.release();
    h->scratch2.release();
    h->scratch3.release();
    for (int k = 0; k < 2; ++k) {
        if (h->stage_in_copied[k]) { (void)hipEventSynchronize(h->stage_in_copied[k]); (void)hipEventDestroy(h->stage_in_copied[k]); }
        h->stage_in_buffers[k].release();
    }
    h->stage_out.release();
    h->done_flag.release();
    h->signal_out.release();
    if (h->d_err) (void)hipFree(h->d_err);
    if (h->d_mm) (void)hipFree(h->d_mm);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream3) (void)hipStreamDestroy(h->stream3);
    delete h;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_get_stats(gatk_pairhmm* h, gatk_pairhmm_stats* out) try {
    if (int rc = check_handle(h)) return rc;
    if (!out) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "out cannot be null");
    *out = (!h->children.empty() && h->stats_in_first_engine) ? h->children[0]->stats : h->stats;
    const gatk_pairhmm* e = h->children.empty() ? h : h->children[0];
    dispatch_counts(e->device, e->squash_base_quals, &out->shared_passes, &out->shared_requests);
    out->region_hits = h->region_hits;
    out->region_misses = h->region_misses;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_set_option(gatk_pairhmm* h, const char* name, double value) try {
    if (int rc = check_handle(h)) return rc;
    if (!name) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "name cannot be null");
    const std::string n(name);
    for (gatk_pairhmm* c : h->children)
        if (n != "multi_min_pairs")
            if (int rc = gatk_pairhmm_set_option(c, name, value)) return rc;
    if (n == "multi_min_pairs") h->multi_min_pairs = std::max<int64_t>(0, static_cast<int64_t>(value));
    else if (n == "fp32_threshold") h->fp32_threshold = static_cast<float>(value);
    else if (n == "debug_long_stall") h->debug_long_stall = value != 0;
    else if (n == "force_rows_per_lane") h->force_rows_per_lane = static_cast<int>(value);
    else if (n == "force_fp64") h->force_fp64 = value != 0;
    else if (n == "emit_mode") h->emit_mode = static_cast<int>(value);
    else if (n == "timing") h->timing = value != 0;
    else if (n == "stream_mode") h->stream_mode = static_cast<int>(value);
    else if (n == "squash_base_quals") h->squash_base_quals = value != 0;
    else if (n == "do_not_use_tristate_correction") h->no_tristate_correction = value != 0;
    else if (n == "pipeline_pairs") h->pipeline_pairs = std::max<int64_t>(0, static_cast<int64_t>(value));
    else if (n == "pipeline_cells_per_pair") h->pipeline_cells_per_pair = std::max<int64_t>(0, static_cast<int64_t>(value));
    else if (n == "pipeline_ramp") h->pipeline_ramp = value != 0;
    else if (n == "pipeline_depth") h->pipeline_depth = value >= 3 ? 3 : 2;
    else if (n == "pipeline_ramp_steps") h->pipeline_ramp_steps = std::max(0, std::min(4, static_cast<int>(value)));
    else if (n == "pipeline_mid_ramp_min") h->pipeline_mid_ramp_min = static_cast<int64_t>(value);
    else if (n == "stream_run_divisor") h->stream_run_divisor = std::max(1, static_cast<int>(value));
    else if (n == "group_rows_allowed") {  // bit k: k rows per lane may be chosen per wave-group; 0: all
        const uint32_t allowed = static_cast<uint32_t>(value);
        h->group_rows_table.clear();
        if (allowed) { const auto t = group_rows_table(allowed); h->group_rows_table.assign(t.begin(), t.end()); }
    }
    else if (n == "pipeline_min_pairs") h->pipeline_min_pairs = std::max<int64_t>(1, static_cast<int64_t>(value));
    else if (n == "combine_calls") h->combine_calls = value != 0;
    else if (n == "per_group_factor") h->per_group_factor = std::max<int64_t>(0, static_cast<int64_t>(value));  // 0: every call chooses per wave-group
    else if (n == "kernel_download_bytes") h->kernel_download_bytes = static_cast<size_t>(std::max(0.0, value));
    else if (n == "signal_download_bytes") h->signal_download_bytes = static_cast<size_t>(std::max(0.0, value));
    else if (n == "kernel_upload_bytes") h->kernel_upload_bytes = static_cast<size_t>(std::max(0.0, value));
    else if (n == "wave_slots") h->wave_slots = std::max(64, static_cast<int>(value));
    else if (n == "stream_lds_floor") h->stream_lds_floor = std::max(0, static_cast<int>(value));
    else if (n == "max_stream_haps") h->max_stream_haps = std::max(1, static_cast<int>(value));
    else return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "unknown option " + n);
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_upload(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions,
                        gatk_pairhmm_resident** out) try {
    if (int rc = check_handle(h)) return rc;
    h = first_engine(h);
    if (!out) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "out cannot be null");
    *out = nullptr;
    auto* R = new gatk_pairhmm_resident();
    if (int rc = plan_and_upload(h, R, num_regions, regions)) {
        R->release();
        delete R;
        return rc;
    }
    *out = R;
    return record_stage_in_event(h);  // no pass follows at once
} catch (...) { return on_exception(); }

int gatk_pairhmm_compute_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r, int32_t sync) try {
    if (int rc = check_handle(h)) return rc;
    h = first_engine(h);
    if (!r) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "resident batch cannot be null");
    if (r->total_pairs > 0)
        if (int rc = enqueue_compute(h, r)) return rc;
    if (sync) return finish_compute(h, r);
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_wait_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r) try {
    if (int rc = check_handle(h)) return rc;
    h = first_engine(h);
    if (!r) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "resident batch cannot be null");
    if (r->total_pairs == 0) return GATK_PAIRHMM_OK;
    if (!r->computed) return fail(GATK_PAIRHMM_ILLEGAL_STATE, "wait before compute_resident");
    return finish_compute(h, r);
} catch (...) { return on_exception(); }

int gatk_pairhmm_download(gatk_pairhmm* h, gatk_pairhmm_resident* r, double* const* likelihoods) try {
    if (int rc = check_handle(h)) return rc;
    h = first_engine(h);
    if (!r) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "resident batch cannot be null");
    if (r->total_pairs > 0 && !r->computed) return fail(GATK_PAIRHMM_ILLEGAL_STATE, "download before compute_resident");
    return download_into(h, r, likelihoods);
} catch (...) { return on_exception(); }

int gatk_pairhmm_resident_free(gatk_pairhmm* h, gatk_pairhmm_resident* r) try {
    if (int rc = check_handle(h)) return rc;
    h = first_engine(h);
    if (!r) return GATK_PAIRHMM_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    r->release();
    delete r;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_get_stream(gatk_pairhmm* h, void** stream) try {
    if (int rc = check_handle(h)) return rc;
    h = first_engine(h);
    if (!stream) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "stream cannot be null");
    *stream = h->stream;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

// ---- the steps right behind the path, on results still on the device -----------------------------

namespace {
int check_epilogue(gatk_pairhmm*& h, gatk_pairhmm_resident* r) {
    if (int rc = check_handle(h)) return rc;
    h = first_engine(h);
    if (!r) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "resident batch cannot be null");
    if (r->total_pairs > 0 && !r->computed) return fail(GATK_PAIRHMM_ILLEGAL_STATE, "likelihoods have not been computed yet");
    return GATK_PAIRHMM_OK;
}
DeviceBatch epilogue_batch(gatk_pairhmm_resident* r) {
    DeviceBatch B = r->base;
    B.out = r->out();
    return B;
}
}  // namespace

int gatk_pairhmm_normalize_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r, const int32_t* reference_allele_index,
                                    int32_t best_to_zero, double maximum_likelihood_difference_cap) try {
    if (int rc = check_epilogue(h, r)) return rc;
    const double cap = maximum_likelihood_difference_cap;
    if (cap >= 0.0 || std::isnan(cap)) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "the minimum reference likelihood fall cannot be positive");
    if (cap == -std::numeric_limits<double>::infinity() && !best_to_zero) return GATK_PAIRHMM_OK;  // ReadLikelihoods.java:357-358
    if (r->total_pairs == 0) return GATK_PAIRHMM_OK;
    HIP_TRY(hipSetDevice(h->device));
    const int32_t* d_ref = nullptr;
    if (reference_allele_index) {
        for (int k = 0; k < r->num_regions; ++k)
            if (r->region_haps[k] > 0 && (reference_allele_index[k] < -1 || reference_allele_index[k] >= r->region_haps[k]))
                return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "reference allele index out of range (region " + std::to_string(k) + ")");
        HIP_TRY(r->d_epilogue.ensure(static_cast<size_t>(r->num_regions) * sizeof(int32_t)));
        HIP_TRY(hipStreamSynchronize(r->stream));  // the buffer may still be in use by an earlier step
        HIP_TRY(hipMemcpyAsync(r->d_epilogue.p, reference_allele_index, static_cast<size_t>(r->num_regions) * sizeof(int32_t), hipMemcpyHostToDevice, r->stream));
        d_ref = r->d_epilogue.as<int32_t>();
    }
    HIP_TRY(launch_normalize_reads(epilogue_batch(r), d_ref, best_to_zero ? 1 : 0, cap, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));  // reference_allele_index is the caller's memory
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_poorly_modeled_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r, double maximum_error_per_base, uint8_t* const* flags) try {
    if (int rc = check_epilogue(h, r)) return rc;
    if (std::isnan(maximum_error_per_base) || maximum_error_per_base <= 0.0)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "the maximum error per base must be a positive number");
    if (r->total_pairs == 0) return GATK_PAIRHMM_OK;
    if (!flags) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "flags cannot be null");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(r->d_epilogue.ensure(static_cast<size_t>(r->total_reads)));
    HIP_TRY(launch_flag_poorly_modelled_reads(epilogue_batch(r), maximum_error_per_base, r->d_epilogue.as<uint8_t>(), r->stream));
    std::vector<uint8_t> host(static_cast<size_t>(r->total_reads));
    HIP_TRY(hipMemcpyAsync(host.data(), r->d_epilogue.p, host.size(), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (int k = 0; k < r->num_regions; ++k) {
        if (r->region_haps[k] == 0) continue;
        if (!flags[k]) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "flags cannot be null (region " + std::to_string(k) + ")");
        memcpy(flags[k], host.data() + r->region_read_begin[k], static_cast<size_t>(r->region_reads[k]));
    }
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_marginalize_resident(gatk_pairhmm* h, gatk_pairhmm_resident* r, const int32_t* const* old_to_new_allele_index,
                                      const int32_t* new_allele_count, double* const* new_likelihoods) try {
    if (int rc = check_epilogue(h, r)) return rc;
    if (r->total_pairs == 0) return GATK_PAIRHMM_OK;
    if (!old_to_new_allele_index || !new_allele_count || !new_likelihoods) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "argument cannot be null");
    // one upload: [map per global haplotype][new count per region][new row offset per read]
    std::vector<int32_t> map(static_cast<size_t>(r->total_haps)), counts(static_cast<size_t>(r->num_regions), 0);
    std::vector<int64_t> base(static_cast<size_t>(r->total_reads));
    std::vector<int64_t> region_new_off(static_cast<size_t>(r->num_regions) + 1, 0);
    for (int k = 0; k < r->num_regions; ++k) {
        region_new_off[k + 1] = region_new_off[k];
        if (r->region_haps[k] == 0) continue;
        const int n_new = new_allele_count[k];
        if (n_new < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "negative allele count (region " + std::to_string(k) + ")");
        if (!old_to_new_allele_index[k] || (!new_likelihoods[k] && n_new > 0))
            return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "argument cannot be null (region " + std::to_string(k) + ")");
        counts[k] = n_new;
        for (int a = 0; a < r->region_haps[k]; ++a) {
            const int n = old_to_new_allele_index[k][a];
            if (n < -1 || n >= n_new) return fail(GATK_PAIRHMM_INDEX_OUT_OF_BOUNDS, "new allele index out of range (region " + std::to_string(k) + ")");
            map[r->region_hap_begin[k] + a] = n;
        }
        for (int x = 0; x < r->region_reads[k]; ++x) base[r->region_read_begin[k] + x] = region_new_off[k] + static_cast<int64_t>(x) * n_new;
        region_new_off[k + 1] += static_cast<int64_t>(r->region_reads[k]) * n_new;
    }
    const int64_t total_new = region_new_off[r->num_regions];
    if (total_new == 0) return GATK_PAIRHMM_OK;
    HIP_TRY(hipSetDevice(h->device));
    ArenaLayout L;
    const size_t o_base = L.add(base.size() * sizeof(int64_t)), o_map = L.add(map.size() * sizeof(int32_t)),
                 o_counts = L.add(counts.size() * sizeof(int32_t)), o_out = L.add(static_cast<size_t>(total_new) * sizeof(double));
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(r->d_epilogue.ensure(L.size));
    uint8_t* D = r->d_epilogue.as<uint8_t>();
    HIP_TRY(hipMemcpyAsync(D + o_base, base.data(), base.size() * sizeof(int64_t), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipMemcpyAsync(D + o_map, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipMemcpyAsync(D + o_counts, counts.data(), counts.size() * sizeof(int32_t), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(launch_marginalize_reads(epilogue_batch(r), reinterpret_cast<const int32_t*>(D + o_map), reinterpret_cast<const int64_t*>(D + o_base),
                                     reinterpret_cast<const int32_t*>(D + o_counts), reinterpret_cast<double*>(D + o_out), r->stream));
    std::vector<double> host(static_cast<size_t>(total_new));
    HIP_TRY(hipMemcpyAsync(host.data(), D + o_out, host.size() * sizeof(double), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (int k = 0; k < r->num_regions; ++k) {
        const int64_t n = region_new_off[k + 1] - region_new_off[k];
        if (n > 0) memcpy(new_likelihoods[k], host.data() + region_new_off[k], static_cast<size_t>(n) * sizeof(double));
    }
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

int gatk_pairhmm_resident_device_output(gatk_pairhmm_resident* r, void** device_ptr, int64_t* count) try {
    if (!r || !device_ptr || !count) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "argument cannot be null");
    *device_ptr = r->out();
    *count = r->total_pairs;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

}  // extern "C"

// Test hook (not in the header): makes the share of engine `index` of the next multi-device calls end in a C++
// exception, which must come back as an error code with every other engine's work finished.  -1 turns it off.
extern "C" int gatk_pairhmm_debug_fail_share(int index) {
    g_fail_share_for_test = index;
    return GATK_PAIRHMM_OK;
}

// Test hook (not in the header, touches no device): where a call whose regions weigh `weights` would be cut into parts
// under the default rules of a handle (shared != 0: other calls are under way on the device).  Writes the cut
// positions, the first 0 and the last num_regions, and returns how many there are (or -1 when `capacity` is too small).
extern "C" int gatk_pairhmm_debug_part_cuts(const int64_t* weights, int32_t num_regions, int32_t shared, int32_t* cuts, int32_t capacity) try {
    const gatk_pairhmm defaults;
    const std::vector<int> c = part_cuts(std::vector<int64_t>(weights, weights + std::max(num_regions, 0)),
                                         CutRules{defaults.pipeline_pairs, defaults.pipeline_min_pairs, defaults.pipeline_mid_ramp_min,
                                                  defaults.pipeline_ramp, defaults.pipeline_ramp_steps, shared != 0});
    if (static_cast<int>(c.size()) > capacity) return -1;
    for (size_t k = 0; k < c.size(); ++k) cuts[k] = c[k];
    return static_cast<int>(c.size());
} catch (...) { return -1; }

// Test hook (not in the header, touches no device): rows per lane of a wave-group whose longest read has `longest`
// bases in a large call, when only the values whose bits are set in `allowed` may be chosen (0: every value).
extern "C" int gatk_pairhmm_debug_group_rows(int32_t longest, uint32_t allowed) try {
    if (longest < 1 || longest > kWave * kMaxRowsPerLane - 2) return -1;
    return allowed ? group_rows_table(allowed)[static_cast<size_t>(longest)] : choose_group_rows_per_lane(longest);
} catch (...) { return -1; }

// An engine of a device's dispatcher: a single-device handle that is nobody's own.
int pairhmm_host::create_shared_engine(int device, bool squash, gatk_pairhmm** out) {
    const int rc = gatk_pairhmm_create(device, out);
    if (rc != GATK_PAIRHMM_OK) return rc;
    (*out)->squash_base_quals = squash;
    (*out)->is_shared_engine = true;  // not a caller's handle
    (*out)->counted = false;
    g_live_handles.fetch_sub(1);
    return GATK_PAIRHMM_OK;
}

// =================================================================================================
// The entry point that allocates nothing
// =================================================================================================
namespace {

// What the device-side planner needs: the caller's offset tables and haplotype bytes (device memory), the
// shape the host chose from the declared maxima, and where to put the descriptors.
struct NoAllocPlan {
    const int32_t* read_offsets;
    const int32_t* hap_offsets;
    const uint8_t* hap_bases;
    int32_t num_reads, num_haps;
    int32_t rows_per_lane;   // K of the launch
    int32_t slot_lanes;      // lanes a read of the declared maximum length takes at that K
    int32_t reads_per_wave;  // 64 / slot_lanes
    int32_t read_max, hap_max;
    unsigned char* scratch;  // this device's address of the scratch array
    double* out;
    int32_t* error_word;
};

// Writes, into the scratch array, what plan_and_upload lays out on the host for the one-haplotype pass: one
// RegionDesc, a GroupDesc per wave-group, a SegDesc per read, the haplotypes on 16-byte boundaries with their
// offsets and lengths, and -infinity for the pairs of zero-length reads.  The host cannot see the lengths
// (they are in device memory and it may not wait), so reads go into fixed lane slots sized for the declared
// maximum read length, reads_per_wave to a wave in read order, and every length is checked here: a read or a
// haplotype longer than declared, an empty haplotype, decreasing offsets or more haplotype bytes than the pool
// holds set kAbortBit in the error word, and the forward kernels behind this launch then do nothing.
// Block 0 takes the haplotypes, the other blocks 256 reads each.
__global__ void __launch_bounds__(256) noalloc_plan(const NoAllocPlan a) {
    const int tid = threadIdx.x;
    unsigned char* const S = a.scratch;
    if (blockIdx.x == 0) {
        __shared__ int s_off[noalloc::kMaxHaps + 1];
        __shared__ int s_bad;
        if (tid == 0) s_bad = 0;
        __syncthreads();
        for (int h = tid; h < a.num_haps; h += 256) {
            const int len = a.hap_offsets[h + 1] - a.hap_offsets[h];
            if (len <= 0 || len > a.hap_max) s_bad = 1;  // every writer stores the same value
        }
        __syncthreads();
        if (tid == 0 && s_bad == 0) {
            int off = 0;
            for (int h = 0; h < a.num_haps; ++h) {
                s_off[h] = off;
                off += (a.hap_offsets[h + 1] - a.hap_offsets[h] + 15) & ~15;
            }
            s_off[a.num_haps] = off;
            if (static_cast<size_t>(off) > noalloc::kHapPoolBytes) s_bad = 1;
        }
        __syncthreads();
        if (s_bad != 0) {
            if (tid == 0) atomicOr(a.error_word, kAbortBit);
            return;
        }
        int32_t* const hap_off = reinterpret_cast<int32_t*>(S + noalloc::at_hap_off);
        int32_t* const hap_len = reinterpret_cast<int32_t*>(S + noalloc::at_hap_len);
        unsigned char* const pool = S + noalloc::at_hap_pool;
        for (int h = tid; h < a.num_haps; h += 256) {
            hap_off[h] = s_off[h];
            hap_len[h] = a.hap_offsets[h + 1] - a.hap_offsets[h];
        }
        for (int h = 0; h < a.num_haps; ++h) {
            const uint8_t* src = a.hap_bases + a.hap_offsets[h];
            const int len = a.hap_offsets[h + 1] - a.hap_offsets[h];
            const int padded = s_off[h + 1] - s_off[h];
            for (int j = tid; j < padded; j += 256) pool[s_off[h] + j] = j < len ? src[j] : 0;
        }
        if (tid == 0) {
            const int groups = (a.num_reads + a.reads_per_wave - 1) / a.reads_per_wave;
            RegionDesc d{};
            d.task_begin = 0; d.group_begin = 0; d.num_groups = groups; d.hap_begin = 0; d.num_haps = a.num_haps;
            *reinterpret_cast<RegionDesc*>(S + noalloc::at_region_desc) = d;
        }
        return;
    }
    const int r = (blockIdx.x - 1) * 256 + tid;
    if (r >= a.num_reads) return;
    const int K = a.rows_per_lane;
    const int len = a.read_offsets[r + 1] - a.read_offsets[r];
    if (len < 0 || len > a.read_max) {
        atomicOr(a.error_word, kAbortBit);
        return;
    }
    reinterpret_cast<int32_t*>(S + noalloc::at_read_out_base)[r] = r * a.num_haps;
    if (len == 0)  // the last matrix row is row 0: log10(0) (Log10PairHMM.java:128-134)
        for (int h = 0; h < a.num_haps; ++h) a.out[static_cast<int64_t>(r) * a.num_haps + h] = -__builtin_inf();
    const int slot = r % a.reads_per_wave;
    SegDesc sd;
    sd.read_idx = r;
    sd.lane_begin = static_cast<uint8_t>(slot * a.slot_lanes);
    sd.seg_lanes = static_cast<uint8_t>(len > 0 ? (len + 1 + K) / K : 0);  // no lanes for a zero-length read
    sd.read_len = static_cast<uint16_t>(len);
    reinterpret_cast<SegDesc*>(S + noalloc::at_segs)[r] = sd;
    if (slot == 0) {  // the first read of a wave-group describes the group
        const int g = r / a.reads_per_wave;
        const int end = min(r + a.reads_per_wave, a.num_reads);
        int max_seg = 1;
        for (int q = r; q < end; ++q) {
            const int l = a.read_offsets[q + 1] - a.read_offsets[q];
            if (l > 0 && l <= a.read_max) max_seg = max(max_seg, (l + 1 + K) / K);
        }
        GroupDesc gd;
        gd.seg_begin = r;
        gd.num_segs = static_cast<int16_t>(end - r);
        gd.fill = static_cast<int16_t>(max_seg - 1);
        reinterpret_cast<GroupDesc*>(S + noalloc::at_groups)[g] = gd;
    }
}

}  // namespace

extern "C" int gatk_pairhmm_compute_likelihoods_noalloc(gatk_pairhmm* h, void* stream, const gatk_pairhmm_batch* device_batch,
                                                        int32_t read_max_length, int32_t haplotype_max_length,
                                                        double* device_likelihoods, int32_t* device_status) try {
    if (int rc = check_handle(h)) return rc;
    h = first_engine(h);
    const gatk_pairhmm_batch* b = device_batch;
    if (!b) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "batch cannot be null");
    if (b->num_reads < 0 || b->num_haps < 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "negative read or haplotype count");
    if (b->num_reads == 0 || b->num_haps == 0) return GATK_PAIRHMM_OK;  // PairHMM.java:193
    if (b->num_reads > noalloc::kMaxReads || b->num_haps > noalloc::kMaxHaps ||
        static_cast<int64_t>(b->num_reads) * b->num_haps > noalloc::kMaxPairs)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "the allocation-free entry point takes at most " + std::to_string(noalloc::kMaxReads) +
                    " reads, " + std::to_string(noalloc::kMaxHaps) + " haplotypes and " + std::to_string(noalloc::kMaxPairs) +
                    " pairs per call, got " + std::to_string(b->num_reads) + " x " + std::to_string(b->num_haps));
    if (read_max_length <= 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "READ_MAX_LENGTH must be > 0 but got " + std::to_string(read_max_length));
    if (haplotype_max_length <= 0) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "HAPLOTYPE_MAX_LENGTH must be > 0 but got " + std::to_string(haplotype_max_length));
    if (read_max_length > kWave * kMaxRowsPerLane - 2)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "readBases is too long, got " + std::to_string(read_max_length) + " but max is " +
                    std::to_string(kWave * kMaxRowsPerLane - 2));
    if (haplotype_max_length > noalloc::kMaxHapLen)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "Haplotype bases is too long for the allocation-free entry point, got " +
                    std::to_string(haplotype_max_length) + " but max is " + std::to_string(noalloc::kMaxHapLen));
    if (!b->read_bases || !b->read_quals || !b->ins_gop || !b->del_gop || !b->gcp || !b->read_offsets)
        return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "readBases cannot be null (the allocation-free entry point fills in no defaults)");
    if (!b->hap_bases || !b->hap_offsets) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "haplotypeBases cannot be null");
    if (!device_likelihoods) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "likelihoods cannot be null");
    if (h->device < 0 || h->device >= kMaxScratchDevices || !g_scratch_ptr[h->device])
        return fail(GATK_PAIRHMM_ILLEGAL_STATE, "no scratch array was set up for this device");
    unsigned char* const S = g_scratch_ptr[h->device];  // the device's own copy of the array
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(h->device));

    // shape of the launch from the declared maxima: one rows-per-lane value, reads in fixed lane slots
    const int K = choose_group_rows_per_lane(read_max_length);
    const int slot_lanes = (read_max_length + 1 + K) / K;
    if (!rows_per_lane_supported(K) || slot_lanes < 1 || slot_lanes > kWave) return fail(GATK_PAIRHMM_ILLEGAL_STATE, "no rows-per-lane value for this read length");
    const int reads_per_wave = kWave / slot_lanes;
    const int groups = (b->num_reads + reads_per_wave - 1) / reads_per_wave;
    const int64_t tasks = static_cast<int64_t>(groups) * b->num_haps;  // <= pairs <= kMaxPairs: the rerun list holds them all
    const int lds = lds_bytes_per_wave(K, kEmitCompareN, haplotype_max_length);
    if (tasks > noalloc::kMaxPairs || lds * kWavesPerBlock > 64 * 1024) return fail(GATK_PAIRHMM_ILLEGAL_STATE, "launch shape outside the scratch array");

    int32_t* const counters = reinterpret_cast<int32_t*>(S + noalloc::at_counters);
    int32_t* const error_word = device_status ? device_status : counters;
    // the array keeps the last call's contents: the counters must start at zero (descriptors are rewritten)
    HIP_TRY(hipMemsetAsync(counters, 0, kCounterBytes, s));
    if (device_status) HIP_TRY(hipMemsetAsync(device_status, 0, sizeof(int32_t), s));

    NoAllocPlan plan{b->read_offsets, b->hap_offsets, b->hap_bases, b->num_reads, b->num_haps, K, slot_lanes, reads_per_wave,
                     read_max_length, haplotype_max_length, S, device_likelihoods, error_word};
    hipLaunchKernelGGL(noalloc_plan, dim3(1 + (b->num_reads + 255) / 256), dim3(256), 0, s, plan);
    HIP_TRY(hipGetLastError());

    DeviceBatch B{};
    B.read_bases = b->read_bases; B.read_quals = b->read_quals; B.ins_gop = b->ins_gop; B.del_gop = b->del_gop; B.gcp = b->gcp;
    B.read_mapq = b->read_mapq;
    B.squash_base_quals = h->squash_base_quals ? 1 : 0;
    B.mismatch_divisor = h->no_tristate_correction ? 1.0 : 3.0;
    B.read_off = b->read_offsets;
    B.read_out_base = reinterpret_cast<const int32_t*>(S + noalloc::at_read_out_base);
    B.hap_pool = S + noalloc::at_hap_pool;
    B.hap_off = reinterpret_cast<const int32_t*>(S + noalloc::at_hap_off);
    B.hap_len = reinterpret_cast<const int32_t*>(S + noalloc::at_hap_len);
    B.groups = reinterpret_cast<const GroupDesc*>(S + noalloc::at_groups);
    B.segs = reinterpret_cast<const SegDesc*>(S + noalloc::at_segs);
    B.region_descs = reinterpret_cast<const RegionDesc*>(S + noalloc::at_region_desc);
    B.row_consts = nullptr;  // the forward kernel works the row constants out itself, quality checks included
    B.num_reads = b->num_reads;
    B.num_haps = b->num_haps;
    B.num_region_descs = 1;
    B.num_tasks = static_cast<int32_t>(tasks);
    B.err_prob = h->d_err;
    B.match_to_match = h->d_mm;
    B.out = device_likelihoods;
    B.rerun_count = counters + 1;
    B.rerun_list = reinterpret_cast<int32_t*>(S + noalloc::at_rerun_list);
    B.error_word = error_word;
    B.abort_word = error_word;
    B.fp32_threshold = h->fp32_threshold;
    B.lds_per_wave = lds;
    const int blocks = static_cast<int>((tasks + kWavesPerBlock - 1) / kWavesPerBlock);
    HIP_TRY(launch_forward(B, K, kEmitCompareN, false, false, blocks, s));
    if (K > kInlineRerunRows)  // with fewer rows per lane the fp32 pass redoes its underflowing items itself
        HIP_TRY(launch_forward(B, K, kEmitCompareN, true, true, std::min(blocks, 1024), s));
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

