// Structures shared by the host planner and the HIP kernels.
#pragma once
#include <cstdint>

namespace pairhmm_hip {

// Read rows a lane can own: the template instantiations of the forward kernel.
constexpr int kMaxRowsPerLane = 16;
constexpr int kRowsPerLaneChoices[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16};
inline bool rows_per_lane_supported(int k) {
    for (int c : kRowsPerLaneChoices)
        if (c == k) return true;
    return false;
}
constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kInlineRerunRows = 5;  // up to this many rows per lane the fp32 one-haplotype pass redoes an item in fp64 itself
constexpr int kHapFrontPad = 64;  // entries in front of the LDS haplotype image: lanes ahead of column 1 read here
constexpr int kHapBackPad = 80;   // entries behind it: draining lanes and the read-ahead of the sweep land here

// How a cell gets its emission term prior[i][j].
enum EmitMode : int {
    kEmitCompare = 0,   // compare read byte with haplotype byte (any alphabet)
    kEmitCompareN = 1,  // same, and a haplotype 'N' matches everything
    kEmitLut4 = 2,      // haplotypes are A/C/G/T only: per-wave LDS table indexed by base code
    kEmitLut5 = 3,      // haplotypes are A/C/G/T/N only: table with an 'N' column
};

constexpr int kMaxGroupedReadLen = kWave * kMaxRowsPerLane - 2;  // longest read a wave-group holds (a pad row in front, the sum row behind)
constexpr int kMaxReadLen = 32766;  // longer reads are refused; reads beyond kMaxGroupedReadLen go through the long-read pass
constexpr int kMaxHapLen = 16000;  // one wave's LDS image must fit in a quarter of a 64 KiB block allocation

// A wave-group packs one or more reads into the 64 lanes of a wavefront: a read of length R takes
// ceil((R + 1) / K) consecutive lanes ("segment"), each lane owning K consecutive read rows; the
// unused rows are put in FRONT of row 1 so that the read's last row is always row K-1 of the
// segment's last lane.
struct SegDesc {
    int32_t read_idx;    // global read index
    uint8_t lane_begin;  // first lane of the segment
    uint8_t seg_lanes;   // lanes in the segment
    uint16_t read_len;
};

// Constants of one matrix row in the fp32 (rescaled) arithmetic, expanded from a read's quality
// bytes once per read and call (the reference expands them once per read too:
// PairHMM.java:210-222 keeps one read against all haplotypes).  Read r owns entries
// [read_off[r] + 2 r, read_off[r + 1] + 2 (r + 1)): entry 0 is the pad row above row 1, entries
// 1..R are the read's rows, entry R + 1 is the sum row.
struct RowConst {
    float p_match;     // prior of a matching base times the row's matchToMatch
    float p_mismatch;  // prior of a mismatching base times the row's matchToMatch
    float t5;          // deletionToDeletion
    float c_f;         // matchToInsertion * insertionToInsertion' / matchToInsertion'   (' = next row)
    float c_i;         // matchToInsertion * indelToMatch' / matchToMatch'
    float c_e;         // matchToDeletion * indelToMatch' / matchToMatch'
    uint32_t base;     // read base, or 0x100 for a row that matches nothing / everything alike
    uint32_t reserved;
};
static_assert(sizeof(RowConst) == 32, "two 16-byte loads per row");

struct GroupDesc {
    int32_t seg_begin;  // first SegDesc of the group
    int16_t num_segs;
    int16_t fill;       // lanes of the longest segment minus one: extra steps to drain the pipeline
};

// Work items of one region inside a launch: item t in [task_begin, task_begin + num_groups * num_haps)
// scores wave-group group_begin + (t - task_begin) % num_groups against haplotype
// hap_begin + (t - task_begin) / num_groups.  Descriptors are sorted by task_begin.
struct RegionDesc {
    int32_t task_begin;
    int32_t group_begin;
    int32_t num_groups;
    int32_t hap_begin;
    int32_t num_haps;
    int32_t reserved[3];
};

// Work items of the fp32 stream pass: item t in [task_begin, task_begin + num_groups) sweeps wave-group
// group_begin + (t - task_begin) over haplotypes [hap_begin, hap_begin + num_haps) back to back.
// hap_local_begin: index of hap_begin among its region's haplotypes.  Sorted by task_begin.
constexpr int kMaxStreamHaps = 16;
struct StreamDesc {
    int32_t task_begin;
    int32_t group_begin;
    int32_t num_groups;
    int32_t hap_begin;
    int32_t num_haps;
    int32_t hap_local_begin;
    int32_t reserved[2];
};

// What the pass that turns the stream pass's raw sums into log10 needs to know about a read.
struct ReadInfo {
    int32_t hap_begin;   // global index of the first haplotype of the read's region
    int32_t num_haps;    // haplotypes of the read's region
    int32_t rows_per_lane;  // rows per lane of the read's wave-group (which names its launch bucket), -1 for a zero-length read
    int32_t region;      // index of the read's region among the call's regions
};

// A (read, haplotype) pair fp32 could not hold, queued for the fp64 pair pass.
struct RerunPair {
    int32_t read_idx;   // global read index
    int32_t hap_local;  // haplotype index within the read's region
};

constexpr int kMaxBuckets = 16;  // one bucket per rows-per-lane value in use
constexpr int kStreamFrontPad = 64;  // entries in front of the haplotype stream: lanes ahead of the first column read here
constexpr int kStreamBackPad = 128;  // entries behind it: draining lanes and the read-ahead land here

// Base quality as the forward pass uses it: capped by the read's mapping quality when the caller
// supplies one, and squashed to the minimum usable quality when under the threshold
// (pairhmm/PairHMM.java:54 BASE_QUALITY_SCORE_THRESHOLD = 18, QualityUtils.java:57 MIN_USABLE_Q_SCORE = 6;
// the method that applies them is upstream of the reference tree).
constexpr unsigned kBaseQualityScoreThreshold = 18;
constexpr unsigned kMinUsableQScore = 6;
constexpr int kDefaultIndelQual = 45;  // sam/GATKSAMRecord.java:61,298-322: reads without insertion / deletion qualities get a flat Q45
constexpr int kDefaultGcp = 10;        // the constant gap-continuation penalty HaplotypeCaller passes (upstream; not in the reference tree)

// Kernel arguments.  All pointers are device pointers.
struct DeviceBatch {
    const uint8_t* read_bases;
    const uint8_t* read_quals;
    const uint8_t* ins_gop;        // null: every base kDefaultIndelQual
    const uint8_t* del_gop;        // null: every base kDefaultIndelQual
    const uint8_t* gcp;            // null: every base kDefaultGcp
    const uint8_t* read_mapq;      // [num_reads] cap on the read's base qualities, or null
    int32_t squash_base_quals;     // 1: base qualities under 18 count as 6
    double mismatch_divisor;       // 3: the tristate correction of Log10PairHMM.java:157; 1 with doNotUseTristateCorrection
    const int32_t* read_off;       // [num_reads + 1]
    const int32_t* read_out_base;  // [num_reads] index in out[] of (read, haplotype 0 of its region)
    const uint8_t* hap_pool;       // haplotype bytes, each haplotype starting on a 16-byte boundary
    const int32_t* hap_off;        // [num_haps] byte offset into hap_pool
    const int32_t* hap_len;        // [num_haps]
    const GroupDesc* groups;       // [num_groups]
    const SegDesc* segs;           // [num_reads with length > 0]
    const RegionDesc* region_descs;  // this launch's regions
    RowConst* row_consts;          // [total_read_bytes + 2 * num_reads], written by the expansion kernel
    int32_t num_reads;
    int32_t num_haps;
    // fp32 stream pass
    uint16_t* hap_stream;          // table offsets (base code * 512) of every haplotype, one zero entry between neighbours
    const int32_t* hap_stream_off; // [num_haps] index of a haplotype's first base in hap_stream
    int32_t hap_stream_len;        // entries of hap_stream, pads included
    const StreamDesc* stream_descs;
    int32_t num_stream_descs;
    float* raw_out;                // [pairs] scaled likelihood sums, laid out like out
    const ReadInfo* read_info;     // [num_reads]
    int32_t* pair_counts;          // [kMaxBuckets] pairs queued for the fp64 pair pass
    RerunPair* pair_lists;         // their lists, bucket b starting at bucket_pair_begin[b]
    int32_t bucket_pair_begin[kMaxBuckets];
    int8_t stream_bucket_of_rows[20];  // launch bucket of the stream pass by rows per lane, -1 when there is none
    // fp64 pair pass of one bucket
    const int32_t* pair_count;     // = pair_counts + bucket
    const RerunPair* pair_list;    // = pair_lists + bucket_pair_begin[bucket]
    int32_t slot_lanes;            // lanes every pair gets: enough for the bucket's longest read
    // long-read pass: the pairs of reads too long for one wave's 64 x 16 rows
    const RerunPair* long_pairs;   // [num_long_pairs]
    int32_t num_long_pairs;
    int32_t long_boundary_stride;  // entries of one boundary row (the launch's longest haplotype, rounded up)
    int32_t long_table_codes;      // 0: the long-read pass compares bytes; 4 or 5: base codes of its emission table (5: with 'N')
    int32_t long_withhold_progress;  // test hook: the first chunk of every pair never says how far it has come, so that the chunk below gives up waiting
    double* long_boundary;         // [workgroups of the launch][waves of a workgroup + 1 rows][long_boundary_stride][2]: what a chunk of rows hands the chunk below it, per column
    int32_t num_region_descs;
    int32_t num_tasks;             // this launch's work items
    const double* err_prob;        // [256]
    const double* match_to_match;  // [128 * 128]
    double* out;
    int32_t* rerun_count;          // this launch's rerun list length
    int32_t* rerun_list;           // this launch's rerun list: work item indices
    int32_t* error_word;
    float fp32_threshold;
    int32_t lds_per_wave;          // bytes of LDS each wave owns
    int64_t total_read_bytes;
    const int32_t* neg_inf_idx;    // [num_neg_inf] indices in out[] of the pairs of zero-length reads
    int32_t num_neg_inf;
    int32_t* error_read;           // slot for the first read with an input error, kept as INT32_MAX - read index (atomicMax); may be null
    // Set only by passes whose descriptors are written on the device: the one-haplotype kernels leave at once
    // when bit kAbortBit of this word is set (the descriptors are then not to be trusted).  Null otherwise.
    const int32_t* abort_word;
};

constexpr int32_t kStallBit = 8;  // error-word bit: a wave of the long-read pass gave up waiting for its workgroup (a device error)
constexpr int kLongMaxWaves = 8;  // most waves of a long-read workgroup, which share one pair
constexpr int32_t kAbortBit = 4;  // error-word bit: a length exceeds what the caller declared, nothing was scored

}  // namespace pairhmm_hip
