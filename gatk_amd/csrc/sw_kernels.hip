// Smith-Waterman pairwise alignment for gfx950 (wave64): kernels and the C-ABI of
// include/gatk_smithwaterman.h.
//
// The algorithm is the reference's (smithwaterman/SWPairwiseAlignment.java): an integer affine-gap matrix
// in which a cell takes the best of the diagonal step, the best horizontal gap ending in it and the best
// vertical gap ending in it, the two gaps tracked by running "best gap so far" values and their lengths
// (calculateMatrix, :223-332); then a back-track from an end cell chosen per overhang strategy
// (calculateCigar, :350-470).
//
// Mapping: one wave per pair.  Lane t owns reference rows t*K+1 .. t*K+K; at step s it is at alternate
// column s - t + 1, so the three values that flow down a column (the cell above, the best vertical gap
// and its length) cross lanes once per step with v_mov_b32_dpp wave_shr:1, and everything that flows
// along a row (the cell to the left, the best horizontal gap and its length) stays in the lane's
// registers.  The alternate sequence sits in LDS.  The back-track matrix goes to global memory as int16
// (a gap is never longer than a sequence) in the order it is produced, [step][row of the lane][lane], so
// that the 64 lanes of a store write 128 consecutive bytes; a second kernel, one wave per pair, picks
// the end cell and walks the matrix through the same index map.  The host reverses and consolidates the
// few CIGAR elements.
//
// Two cell forms.  The plain one keeps a gap's score and its length in two registers and follows the
// reference statement by statement (about 30 vector instructions per cell).  The packed one keeps them in one
// 32-bit word, score * 2^(S+2) + priority * 2^S + length, so that "the better score, and on equal scores the
// longer (older) gap" is one signed max, and the three-way choice diagonal > horizontal > vertical, which the
// reference settles with >= tests in that order, is one max of three such words whose priority fields are
// 2, 1 and 0: 12 vector instructions per cell, and the low 16 bits of the winning word are the back-track
// entry.  A pair takes the packed form when its lengths fit S <= 14 bits and every score the matrix can
// hold fits the 30 - S bits left (checked on the host from the weights and the lengths); any other pair
// takes the plain form.  Both give the reference's integers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "gatk_smithwaterman.h"

namespace {

constexpr int kWave = 64;
constexpr int kMaxRowsPerLane = 32;
constexpr int kMaxRef = kWave * kMaxRowsPerLane;  // 2048
constexpr int kHeadElems = 16;  // raw CIGAR elements of a pair that come back with the first copy
constexpr int kNumRowChoices = 13;
constexpr int kNumGroups = 2 * kNumRowChoices;  // launch groups: (rows per lane, cell form), the packed form the odd ones
constexpr int kRowChoices[kNumRowChoices] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32};  // rows per lane a fill kernel exists for
constexpr int kMaxAlt = 16000;
constexpr size_t kStagedPoolBytes = 1 << 20;      // sequence pools from this size on go up one launch at a time
constexpr int kLowInit = INT_MIN / 2;             // lowInitValue, :235
constexpr int kMatrixMinCutoff = -100000000;      // (int) -1e8, :233 (the static cutoff flag is false)

struct PairDesc {
    int64_t ref_off, alt_off;      // byte offsets into the sequence pool
    int64_t btrack_off;            // int16 offset of this pair's back-track matrix, [steps][rows_per_lane][64]
    int32_t rows_per_lane;         // of the launch that fills it
    int64_t edge_off;              // int32 offset of its last-column (n+1) and bottom-row (m+1) scores
    int64_t elem_off;              // int32 offset of its raw CIGAR element list (2 ints per element)
    int32_t n, m;                  // reference / alternate length
    int32_t elem_cap;
    int32_t len_bits;              // S of the packed cell form this pair's matrix was filled in; 0: the plain form
};

struct PairResult {
    int32_t num_elems;   // raw elements written, last CIGAR element first
    int32_t offset;      // alignment offset
};

struct SwArgs {
    const uint8_t* pool;
    const PairDesc* pairs;
    int16_t* btrack;
    int32_t* edges;
    int32_t* elems;
    int32_t* heads;        // [num_pairs][kHeadElems][2]: the first raw elements of every pair again, packed for one copy
    PairResult* results;
    const int32_t* order;  // pair indices grouped by rows per lane: a launch takes a contiguous range of it
    int32_t num_pairs;
    int32_t w_match, w_mismatch, w_open, w_extend;
    int32_t strategy;
};

// Where cell (i, j) of a pair's back-track matrix lives: lane t = (i - 1) / K reaches column j at step
// j - 1 + t and holds row i as its row r = (i - 1) % K.
__device__ __forceinline__ int64_t btrack_index(int i, int j, int K) {
    const int t = (i - 1) / K, r = (i - 1) - t * K;
    return (static_cast<int64_t>(j - 1 + t) * K + r) * kWave + t;
}

__device__ __forceinline__ int shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }

// First-row / first-column scores: gap penalties when overhangs count as indels (:244-261), else 0.
__device__ __forceinline__ int edge_score(const SwArgs& A, int k) {
    if (k <= 0 || (A.strategy != GATK_SW_INDEL && A.strategy != GATK_SW_LEADING_INDEL)) return 0;
    return A.w_open + (k - 1) * A.w_extend;
}

template <int K>
__global__ void __launch_bounds__(kWave) sw_fill(const SwArgs A, const int first_pair) {
    extern __shared__ uint8_t alt_lds[];  // [64 pad][alternate][64 pad]
    const PairDesc P = A.pairs[A.order[first_pair + blockIdx.x]];  // the grid is exactly this launch's range
    const int lane = threadIdx.x;
    const int n = P.n, m = P.m;
    const uint8_t* ref = A.pool + P.ref_off;
    const uint8_t* alt = A.pool + P.alt_off;
    for (int x = lane; x < m + 2 * kWave; x += kWave) {
        const int j = x - kWave;
        alt_lds[x] = (j >= 0 && j < m) ? alt[j] : 0;
    }
    __syncthreads();

    const int row0 = lane * K + 1;  // this lane's first row (1-based)
    int a_base[K], left[K], bgh[K], gsh[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const int i = row0 + r;
        a_base[r] = i <= n ? ref[i - 1] : 0x100;  // rows past the end match nothing
        left[r] = edge_score(A, i);               // sw[i][0]
        bgh[r] = kLowInit;
        gsh[r] = 0;
    }
    int diag0 = edge_score(A, row0 - 1);  // sw[row0 - 1][0]
    int out_sw = 0, out_bgv = kLowInit, out_gsv = 0;  // what this lane hands down: its last row at its current column
    int16_t* const bt = A.btrack + P.btrack_off;
    int32_t* const last_col = A.edges + P.edge_off;
    int32_t* const bottom = last_col + (n + 1);
    const int steps = m + (n + K - 1) / K - 1;
    const int last_row_lane = (n - 1) / K, last_row_r = (n - 1) % K;  // where row n lives
    for (int s = 0; s < steps; ++s) {
        int up = shr1(out_sw), bgv = shr1(out_bgv), gsv = shr1(out_gsv);
        const int j = s - lane + 1;  // 1-based column
        if (lane == 0) { up = edge_score(A, j); bgv = kLowInit; gsv = 0; }  // row 0: sw[0][j], no vertical gap yet
        if (j >= 1 && j <= m) {
            const int b_base = alt_lds[kWave + j - 1];
            const int up_in = up;
            int diag = diag0;
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const int old_left = left[r];
                const int step_diag = diag + (a_base[r] == b_base ? A.w_match : A.w_mismatch);
                // best vertical gap ending here, :283-298
                int prev_gap = up + A.w_open;
                bgv += A.w_extend;
                if (prev_gap > bgv) { bgv = prev_gap; gsv = 1; } else ++gsv;
                // best horizontal gap ending here, :305-318
                prev_gap = old_left + A.w_open;
                bgh[r] += A.w_extend;
                if (prev_gap > bgh[r]) { bgh[r] = prev_gap; gsh[r] = 1; } else ++gsh[r];
                // priority: diagonal, then right, then down, :320-331
                int val, btr;
                if (step_diag >= bgv && step_diag >= bgh[r]) { val = step_diag; btr = 0; }
                else if (bgh[r] >= bgv) { val = bgh[r]; btr = -gsh[r]; }
                else { val = bgv; btr = gsv; }
                val = val > kMatrixMinCutoff ? val : kMatrixMinCutoff;
                bt[(static_cast<int64_t>(s) * K + r) * kWave + lane] = static_cast<int16_t>(btr);  // = btrack_index(i, j, K)
                left[r] = val;
                diag = old_left;  // sw[i][j-1] is the diagonal of row i + 1
                up = val;
            }
            diag0 = up_in;
            out_sw = up; out_bgv = bgv; out_gsv = gsv;
            // the two edges calculateCigar looks at: the last column (once per lane) and the last row
            // (one row of one lane), kept out of the cell loop so that a cell has no branch
            if (j == m) {
#pragma unroll
                for (int r = 0; r < K; ++r)
                    if (row0 + r <= n) last_col[row0 + r] = left[r];
            }
            if (lane == last_row_lane) {
                int v = left[0];
#pragma unroll
                for (int r = 1; r < K; ++r) v = (r == last_row_r) ? left[r] : v;
                bottom[j] = v;
            }
        }
    }
}

// The packed cell form (see the head of the file).  A word is score * 2^Q + priority * 2^S + length with
// Q = S + 2; scores that carry no gap (cells, the first row and column) are score * 2^Q with the low bits zero.
//   vertical gap    Wv = max(up + open_v, Wv + ext)      open_v = w_open * 2^Q + 1, ext = w_extend * 2^Q + 1
//   horizontal gap  Wh = max(left + open_h, Wh + ext)    open_h = open_v + 2^S (priority 1)
//   diagonal        sd = diag + (match ? w_match : w_mismatch) * 2^Q + 2 * 2^S
//   cell            best = max(sd, Wh, Wv); value = best with the low Q bits cleared; back-track = low 16 bits
// On equal scores the extended gap has the larger length field (>= 2 against 1), which is the reference's
// "prev_gap > best_gap" (:283-298, :305-318); between the three candidates the priority field decides as the
// reference's ">=" tests do (:320-331).  The "no gap yet" value is -1.5 * 2^30: below every word a real score
// makes, and one extension (the only thing that happens to it before a real gap replaces it) cannot wrap it.
// MATRIX_MIN_CUTOFF (:233) cannot be reached by a pair the host lets in here.
template <int K>
__global__ void __launch_bounds__(kWave) sw_fill_packed(const SwArgs A, const int first_pair) {
    extern __shared__ uint8_t alt_lds[];  // [64 pad][alternate][64 pad]
    const PairDesc P = A.pairs[A.order[first_pair + blockIdx.x]];
    const int lane = threadIdx.x;
    const int n = P.n, m = P.m;
    const uint8_t* ref = A.pool + P.ref_off;
    const uint8_t* alt = A.pool + P.alt_off;
    for (int x = lane; x < m + 2 * kWave; x += kWave) {
        const int j = x - kWave;
        alt_lds[x] = (j >= 0 && j < m) ? alt[j] : 0;
    }
    __syncthreads();

    const int S = P.len_bits, Q = S + 2;
    const int unit = 1 << Q;
    int w_eq = A.w_match * unit + (2 << S), w_ne = A.w_mismatch * unit + (2 << S);
    asm volatile("" : "+v"(w_eq), "+v"(w_ne));  // two finished constants, not a select of weights with the shift behind it
    const int ext = A.w_extend * unit + 1;
    const int open_v = A.w_open * unit + 1, open_h = open_v + (1 << S);
    const int none_v = -(3 << 29), none_h = none_v + (1 << S);
    const int value_mask = ~(unit - 1);

    const int row0 = lane * K + 1;
    int a_base[K], left[K], wh[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const int i = row0 + r;
        a_base[r] = i <= n ? ref[i - 1] : 0x100;
        left[r] = edge_score(A, i) * unit;
        wh[r] = none_h;
    }
    int diag0 = edge_score(A, row0 - 1) * unit;
    int out_sw = 0, out_wv = none_v;
    int16_t* const bt = A.btrack + P.btrack_off;
    int32_t* const last_col = A.edges + P.edge_off;
    int32_t* const bottom = last_col + (n + 1);
    const int steps = m + (n + K - 1) / K - 1;
    const int last_row_lane = (n - 1) / K, last_row_r = (n - 1) % K;
    // the alternate's base of a step is read a step ahead (a wave alone on its SIMD would wait out the LDS latency
    // every step; volatile, or the compiler moves the read back to its use)
    const volatile __attribute__((address_space(3))) uint8_t* const alt_of_lane =
        (const volatile __attribute__((address_space(3))) uint8_t*)alt_lds + (kWave - lane);
    int b_next = alt_of_lane[0];  // column 1 - lane: the pad in front for every lane but the first
    for (int s = 0; s < steps; ++s) {
        int up = shr1(out_sw), wv = shr1(out_wv);
        const int j = s - lane + 1;
        const int b_base = b_next;
        b_next = alt_of_lane[s + 1];  // column j + 1; behind the alternate: the pad (64 entries, and a lane is at most 63 columns behind lane 0)
        if (lane == 0) { up = edge_score(A, j) * unit; wv = none_v; }
        if (j >= 1 && j <= m) {
            const int up_in = up;
            int16_t* const bt_step = bt + static_cast<int64_t>(s) * K * kWave + lane;
            // everything that reads the previous column first (after it a row's old value is dead and the new one
            // takes its register), then the chain down the rows
            int sd[K];
#pragma unroll
            for (int r = 0; r < K; ++r) {
                sd[r] = (r == 0 ? diag0 : left[r - 1]) + (a_base[r] == b_base ? w_eq : w_ne);
                wh[r] = max(left[r] + open_h, wh[r] + ext);
            }
#pragma unroll
            for (int r = 0; r < K; ++r) {
                wv = max(up + open_v, wv + ext);
                const int best = max(max(sd[r], wh[r]), wv);
                // the two SW_TIMING_ builds store half the bytes or none (wrong alignments: for timing what the stores cost)
#if defined(SW_TIMING_BYTE_STORES)
                reinterpret_cast<int8_t*>(bt)[(static_cast<int64_t>(s) * K + r) * kWave + lane] = static_cast<int8_t>(best);
#elif !defined(SW_TIMING_WITHOUT_STORES)
                bt_step[r * kWave] = static_cast<int16_t>(best);
#else
                if (best == 0x12345678) bt_step[r * kWave] = 1;
#endif
                up = best & value_mask;
                left[r] = up;
            }
            diag0 = up_in;
            out_sw = up; out_wv = wv;
            if (j == m) {
#pragma unroll
                for (int r = 0; r < K; ++r)
                    if (row0 + r <= n) last_col[row0 + r] = left[r] >> Q;
            }
            if (lane == last_row_lane) {
                int v = left[0];
#pragma unroll
                for (int r = 1; r < K; ++r) v = (r == last_row_r) ? left[r] : v;
                bottom[j] = v >> Q;
            }
        }
    }
}

// calculateCigar, :350-470, one wave per pair: the raw element list (last element first) and the offset.
//
// The reference's two scans for the end cell and its walk are sequential loops; a lane working through them alone makes
// one trip to memory per handful of entries (a region-sized call spent 140 us here, as much as in the fill).  The wave
// does the same work side by side:
//   * last column (:362-372, "score >= best" keeps the LAST row among equal scores): every lane takes the rows
//     lane + 1, lane + 65, ...; the lanes' bests are merged by (score, then larger row);
//   * bottom row (:375-392: a column replaces the best so far when its score is higher, or equal and its cell strictly
//     closer to the diagonal): the outcome of that loop is the least of (score descending, distance from the diagonal
//     ascending, scan order ascending) over the last column's winner and all columns, so the lanes' bests merge by that key;
//   * the walk (:401-430) runs mostly along diagonals: the 64 lanes fetch the next 64 cells of the diagonal, a ballot
//     finds the first that is not a diagonal step, and all lanes keep the same walk state (lane 0 writes the elements).
__device__ __forceinline__ int wave_first_lane_value(int v, int src_lane) { return __shfl(v, src_lane, kWave); }

__global__ void __launch_bounds__(kWave) sw_backtrack(const SwArgs A, const int first_pair, const int num) {
    const int x = blockIdx.x;
    if (x >= num) return;
    const int lane = threadIdx.x;
    const int k = A.order[first_pair + x];
    const PairDesc P = A.pairs[k];
    const int refLength = P.n, altLength = P.m;
    const int16_t* bt = A.btrack + P.btrack_off;
    const int32_t* last_col = A.edges + P.edge_off;
    const int32_t* bottom = last_col + (refLength + 1);
    int32_t* elems = A.elems + P.elem_off;
    int32_t* heads = A.heads + static_cast<int64_t>(k) * 2 * kHeadElems;
    int count = 0;
    auto add = [&](int op, int len) {
        if (lane == 0) {
            if (count < P.elem_cap) { elems[2 * count] = op; elems[2 * count + 1] = len; }
            if (count < kHeadElems) { heads[2 * count] = op; heads[2 * count + 1] = len; }
        }
        ++count;
    };
    // A packed entry is the low 16 bits of the winning word: length in the low len_bits bits, the priority (2 diagonal,
    // 1 horizontal gap, 0 vertical gap) above them.  diag_code is what (entry & diag_mask) is for a diagonal step: the
    // plain form's entries are the reference's btrack values themselves, 0 for a diagonal step.
    const int len_bits = P.len_bits;
    const int diag_mask = len_bits ? 3 << len_bits : -1, diag_code = len_bits ? 2 << len_bits : 0;
    int p1 = 0, p2 = 0, maxscore = INT_MIN, segment_length = 0;
    if (A.strategy == GATK_SW_INDEL) {
        p1 = refLength; p2 = altLength;
    } else {
        p2 = altLength;
        {   // last column
            int best_v = INT_MIN, best_i = 0;
            for (int i = 1 + lane; i <= refLength; i += kWave) {
                const int v = last_col[i];
                if (v >= best_v) { best_v = v; best_i = i; }
            }
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const int ov = __shfl_xor(best_v, off, kWave), oi = __shfl_xor(best_i, off, kWave);
                if (oi > 0 && (best_i == 0 || ov > best_v || (ov == best_v && oi > best_i))) { best_v = ov; best_i = oi; }
            }
            p1 = best_i; maxscore = best_v;  // refLength >= 1: some row always wins
        }
        if (A.strategy != GATK_SW_LEADING_INDEL) {
            // candidates: columns 1 .. altLength of the bottom row; key (score descending, distance ascending, column ascending)
            int best_v = INT_MIN, best_d = INT_MAX, best_j = 0;
            for (int j = 1 + lane; j <= altLength; j += kWave) {
                const int v = bottom[j], d = abs(refLength - j);
                if (best_j == 0 || v > best_v || (v == best_v && d < best_d)) { best_v = v; best_d = d; best_j = j; }
            }
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const int ov = __shfl_xor(best_v, off, kWave), od = __shfl_xor(best_d, off, kWave), oj = __shfl_xor(best_j, off, kWave);
                const bool better = oj > 0 && (best_j == 0 || ov > best_v || (ov == best_v && (od < best_d || (od == best_d && oj < best_j))));
                if (better) { best_v = ov; best_d = od; best_j = oj; }
            }
            // against the last column's winner, which came first in the reference's order
            if (best_j > 0 && (best_v > maxscore || (best_v == maxscore && best_d < abs(p1 - p2)))) {
                p1 = refLength; p2 = best_j; maxscore = best_v;
                segment_length = altLength - best_j;
            }
        }
    }
    if (segment_length > 0 && A.strategy == GATK_SW_SOFTCLIP) { add(GATK_SW_OP_S, segment_length); segment_length = 0; }
    int state = GATK_SW_OP_M;
    // every round moves at least one cell towards row 0 or column 0; the bound only keeps a damaged matrix from
    // holding the wave forever
    int rounds_left = refLength + altLength + 1;
    do {
        const int span = min(kWave, min(p1, p2));
        const int raw_mine = lane < span ? bt[btrack_index(p1 - lane, p2 - lane, P.rows_per_lane)] : diag_code;
        const unsigned long long not_diag = __ballot(lane < span && (raw_mine & diag_mask) != diag_code);
        const int matches = not_diag ? __builtin_ctzll(not_diag) : span;  // leading diagonal steps
        int btr = 0;  // the reference's btrack value of the gap behind them: > 0 that many rows up, < 0 that many columns left
        if (not_diag) {
            const int raw = wave_first_lane_value(raw_mine, matches);
            if (len_bits == 0) btr = raw;
            else { const int len = raw & ((1 << len_bits) - 1); btr = ((raw >> len_bits) & 3) == 1 ? -len : len; }
        }
        if (matches > 0) {
            p1 -= matches; p2 -= matches;
            if (state == GATK_SW_OP_M) segment_length += matches;
            else { add(state, segment_length); segment_length = matches; state = GATK_SW_OP_M; }
        }
        if (btr != 0) {
            const int new_state = btr > 0 ? GATK_SW_OP_D : GATK_SW_OP_I;
            const int step_length = btr > 0 ? btr : -btr;
            if (new_state == GATK_SW_OP_I) p2 -= step_length; else p1 -= step_length;
            if (new_state == state) segment_length += step_length;
            else { add(state, segment_length); segment_length = step_length; state = new_state; }
        }
    } while (p1 > 0 && p2 > 0 && --rounds_left > 0);
    int offset;
    if (A.strategy == GATK_SW_SOFTCLIP) {
        add(state, segment_length);
        if (p2 > 0) add(GATK_SW_OP_S, p2);
        offset = p1;
    } else if (A.strategy == GATK_SW_IGNORE) {
        add(state, segment_length + p2);
        offset = p1 - p2;
    } else {
        add(state, segment_length);
        if (p1 > 0) add(GATK_SW_OP_D, p1);
        else if (p2 > 0) add(GATK_SW_OP_I, p2);
        offset = 0;
    }
    if (lane == 0) A.results[k] = PairResult{count, offset};
}

thread_local std::string g_sw_error;
// GATK_SW_PLAIN_CELLS=1 in the environment keeps every pair on the plain cell form (cross-checks, timing).
const bool g_force_plain = [] { const char* e = std::getenv("GATK_SW_PLAIN_CELLS"); return e && e[0] == '1'; }();

int fail(int code, const std::string& msg) {
    g_sw_error = msg;
    return code;
}

#define SW_TRY(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(GATK_SW_DEVICE_ERROR, std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

constexpr int kSideStreams = 3;

// Grow-only device scratch of the calling thread (one per device it has used), and its streams: the fill
// launches of a call (one per rows-per-lane value) are short on waves each, so they run side by side.
struct Scratch {
    int device = -1;
    void* p = nullptr;
    size_t cap = 0;
    hipStream_t main = nullptr, side[kSideStreams] = {};
    hipEvent_t fork = nullptr, join[kSideStreams] = {};
    void* up = nullptr;    // pinned host staging of what a call sends ...
    size_t up_cap = 0;
    void* down = nullptr;  // ... and of what comes back with the first copy
    size_t down_cap = 0;
};
thread_local std::vector<Scratch> g_scratch;

hipError_t scratch_for(int device, size_t bytes, void** out, Scratch** scratch) {
    Scratch* s = nullptr;
    for (auto& x : g_scratch)
        if (x.device == device) s = &x;
    if (!s) {
        Scratch fresh;
        fresh.device = device;
        hipError_t e = hipStreamCreateWithFlags(&fresh.main, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&fresh.fork, hipEventDisableTiming);
        for (int k = 0; k < kSideStreams && e == hipSuccess; ++k) {
            e = hipStreamCreateWithFlags(&fresh.side[k], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&fresh.join[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) return e;
        g_scratch.push_back(fresh);
        s = &g_scratch.back();
    }
    *scratch = s;
    if (bytes > s->cap) {
        if (s->p) { hipError_t e = hipFree(s->p); s->p = nullptr; s->cap = 0; if (e != hipSuccess) return e; }
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc(&s->p, want);
        if (e != hipSuccess) return e;
        s->cap = want;
    }
    *out = s->p;
    return hipSuccess;
}

// Grow-only pinned host buffer (contents are not kept).
hipError_t pinned_for(void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return hipSuccess;
    if (*p) { hipError_t e = hipHostFree(*p); *p = nullptr; *cap = 0; if (e != hipSuccess) return e; }
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(p, want, hipHostMallocDefault);
    if (e == hipSuccess) *cap = want;
    return e;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <int K>
hipError_t launch_fill(const SwArgs& A, bool packed, int first, int count, size_t lds, hipStream_t stream) {
    if (packed) hipLaunchKernelGGL((sw_fill_packed<K>), dim3(count), dim3(kWave), lds, stream, A, first);
    else hipLaunchKernelGGL((sw_fill<K>), dim3(count), dim3(kWave), lds, stream, A, first);
    return hipGetLastError();
}

}  // namespace

extern "C" {

const char* gatk_sw_last_error(void) { return g_sw_error.c_str(); }

int gatk_sw_align(int32_t device, int32_t num_pairs, const uint8_t* const* references, const int32_t* reference_lengths,
                  const uint8_t* const* alternates, const int32_t* alternate_lengths, const gatk_sw_parameters* parameters,
                  int32_t overhang_strategy, int32_t cigar_capacity, int32_t* cigar_ops, int32_t* cigar_lengths,
                  int32_t* cigar_counts, int32_t* alignment_offsets) try {
    if (num_pairs < 0) return fail(GATK_SW_ILLEGAL_ARGUMENT, "negative pair count");
    if (num_pairs == 0) return GATK_SW_OK;
    if (!references || !reference_lengths || !alternates || !alternate_lengths || !parameters || !cigar_ops || !cigar_lengths ||
        !cigar_counts || !alignment_offsets)
        return fail(GATK_SW_ILLEGAL_ARGUMENT, "argument cannot be null");
    if (overhang_strategy < GATK_SW_SOFTCLIP || overhang_strategy > GATK_SW_IGNORE) return fail(GATK_SW_ILLEGAL_ARGUMENT, "unknown overhang strategy");
    if (cigar_capacity <= 0) return fail(GATK_SW_ILLEGAL_ARGUMENT, "cigar_capacity must be positive");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(GATK_SW_DEVICE_ERROR, "no HIP device is visible: the Smith-Waterman engine has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(GATK_SW_DEVICE_ERROR, "device index out of range");

    // ---- layout -----------------------------------------------------------------------------------
    std::vector<PairDesc> pairs(num_pairs);
    size_t pool_bytes = 0, bt_entries = 0, edge_ints = 0, elem_ints = 0;
    const int elem_stride = 2 * cigar_capacity + 16;  // raw elements kept per pair
    int max_m = 0;
    for (int k = 0; k < num_pairs; ++k) {
        const int n = reference_lengths[k], m = alternate_lengths[k];
        if (!references[k] || n <= 0 || !alternates[k] || m <= 0)  // SWPairwiseAlignment.java:183-184
            return fail(GATK_SW_ILLEGAL_ARGUMENT, "Non-null, non-empty sequences are required for the Smith-Waterman calculation (pair " + std::to_string(k) + ")");
        if (n > kMaxRef || m > kMaxAlt)
            return fail(GATK_SW_ILLEGAL_ARGUMENT, "sequence too long: reference up to " + std::to_string(kMaxRef) + ", alternate up to " +
                                                      std::to_string(kMaxAlt) + " bases (pair " + std::to_string(k) + ")");
        PairDesc& P = pairs[k];
        P.n = n; P.m = m;
        P.edge_off = static_cast<int64_t>(edge_ints); edge_ints += static_cast<size_t>(n + 1) + (m + 1);
        // an element is added when the back-track changes state, plus a few at the ends: a list that outgrows
        // this cannot consolidate into cigar_capacity elements
        P.elem_cap = std::min(n + m + 8, elem_stride);
        P.elem_off = static_cast<int64_t>(elem_ints); elem_ints += 2 * static_cast<size_t>(elem_stride);  // every pair the same stride: see the fetch
        max_m = std::max(max_m, m);
    }
    // rows per lane: per pair, the smallest instantiation that holds its reference; cell form: packed where the
    // pair's lengths and scores fit its fields (see sw_fill_packed).  One launch per (rows per lane, form) in use.
    std::vector<int32_t> order(num_pairs);
    std::array<int, kNumGroups + 1> group_begin{};
    {
        const int64_t w_abs = std::max(std::abs(static_cast<int64_t>(parameters->w_match)), std::abs(static_cast<int64_t>(parameters->w_mismatch)));
        const int64_t open_abs = std::abs(static_cast<int64_t>(parameters->w_open)), ext_abs = std::abs(static_cast<int64_t>(parameters->w_extend));
        const bool gaps_cost = parameters->w_open <= 0 && parameters->w_extend <= 0;  // what the score bound below assumes
        std::array<int, kNumGroups> count{};
        std::vector<uint8_t> group(num_pairs);
        for (int k = 0; k < num_pairs; ++k) {
            PairDesc& P = pairs[k];
            const int need = (P.n + kWave - 1) / kWave;
            int c = 0;
            while (kRowChoices[c] < need) ++c;
            const int K = kRowChoices[c];
            // Packed form: a gap length (at most the rows the wave computes, 64 K, or the alternate's length) must
            // stay below 2^S - 1, and every score of the matrix below 2^(28 - S) in magnitude.  A cell is at most
            // min(rows, m) diagonal steps of the larger weight above its edge, and at least one gap along its row
            // below it: value(i, j) >= edge(i) + w_open + (j - 1) w_extend.
            const int64_t rows = static_cast<int64_t>(kWave) * K, longest = std::max<int64_t>(rows, P.m);
            int S = 4;
            while ((int64_t{1} << S) < longest + 3) ++S;
            const int64_t bound = w_abs * (std::min<int64_t>(rows, P.m) + 2) + 2 * open_abs + (rows + P.m + 2) * ext_abs;
            const bool packed = !g_force_plain && gaps_cost && S <= 14 && bound < (int64_t{1} << (28 - S)) &&
                                std::max(w_abs, std::max(open_abs, ext_abs)) < (int64_t{1} << (26 - S));
            P.len_bits = packed ? S : 0;
            const int g = 2 * c + (packed ? 1 : 0);
            group[k] = static_cast<uint8_t>(g);
            ++count[g];
            const size_t steps = static_cast<size_t>(P.m) + (P.n + K - 1) / K - 1;
            P.rows_per_lane = K;
            P.btrack_off = static_cast<int64_t>(bt_entries);
            bt_entries += steps * K * kWave;
        }
        for (int g = 0; g < kNumGroups; ++g) group_begin[g + 1] = group_begin[g] + count[g];
        std::array<int, kNumGroups> at{};
        for (int k = 0; k < num_pairs; ++k) order[group_begin[group[k]] + at[group[k]]++] = k;
    }
    // the sequence pool in launch order (most rows per lane first), so that a launch's sequences are one range of it
    std::array<size_t, kNumGroups + 1> group_pool{};  // group g's sequences: [group_pool[g + 1], group_pool[g])
    for (int g = kNumGroups - 1; g >= 0; --g) {
        group_pool[g + 1] = pool_bytes;
        for (int x = group_begin[g]; x < group_begin[g + 1]; ++x) {
            PairDesc& P = pairs[order[x]];
            P.ref_off = static_cast<int64_t>(pool_bytes); pool_bytes += align_up(P.n, 16);
            P.alt_off = static_cast<int64_t>(pool_bytes); pool_bytes += align_up(P.m, 16);
        }
    }
    group_pool[0] = pool_bytes;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off = align_up(off + bytes, 256); return at; };
    const size_t o_pairs = carve(pairs.size() * sizeof(PairDesc)), o_order = carve(order.size() * sizeof(int32_t)), o_pool = carve(pool_bytes);
    const size_t upload_bytes = off;
    const size_t o_bt = carve(bt_entries * sizeof(int16_t)), o_edges = carve(edge_ints * sizeof(int32_t)),
                 o_elems = carve(elem_ints * sizeof(int32_t)), o_results = carve(pairs.size() * sizeof(PairResult)),
                 o_heads = carve(pairs.size() * 2 * kHeadElems * sizeof(int32_t));

    // ---- device -----------------------------------------------------------------------------------
    SW_TRY(hipSetDevice(device));
    void* base = nullptr;
    Scratch* sc = nullptr;
    SW_TRY(scratch_for(device, off, &base, &sc));
    uint8_t* D = static_cast<uint8_t*>(base);
    hipStream_t stream = sc->main;  // this thread's stream on the device: its calls are ordered, results are fetched synchronously
    // The sequences go straight into pinned memory (the bytes between them are never read).  A large pool goes
    // up launch by launch on the launch's own stream, so that the host packs the next launch's sequences while
    // the first one runs; a small one goes up whole with the descriptors.
    SW_TRY(pinned_for(&sc->up, &sc->up_cap, upload_bytes));
    uint8_t* host = static_cast<uint8_t*>(sc->up);
    const bool staged = pool_bytes >= kStagedPoolBytes;
    auto pack = [&](int g) {
        for (int x = group_begin[g]; x < group_begin[g + 1]; ++x) {
            const int k = order[x];
            memcpy(host + o_pool + pairs[k].ref_off, references[k], pairs[k].n);
            memcpy(host + o_pool + pairs[k].alt_off, alternates[k], pairs[k].m);
        }
    };
    memcpy(host + o_pairs, pairs.data(), pairs.size() * sizeof(PairDesc));
    memcpy(host + o_order, order.data(), order.size() * sizeof(int32_t));
    if (!staged)
        for (int g = 0; g < kNumGroups; ++g) pack(g);
    SW_TRY(hipMemcpyAsync(D, host, staged ? o_pool : upload_bytes, hipMemcpyHostToDevice, stream));
    SwArgs A{};
    A.pool = D + o_pool;
    A.pairs = reinterpret_cast<const PairDesc*>(D + o_pairs);
    A.btrack = reinterpret_cast<int16_t*>(D + o_bt);
    A.edges = reinterpret_cast<int32_t*>(D + o_edges);
    A.elems = reinterpret_cast<int32_t*>(D + o_elems);
    A.heads = reinterpret_cast<int32_t*>(D + o_heads);
    A.results = reinterpret_cast<PairResult*>(D + o_results);
    A.order = reinterpret_cast<const int32_t*>(D + o_order);
    A.num_pairs = num_pairs;
    A.w_match = parameters->w_match; A.w_mismatch = parameters->w_mismatch; A.w_open = parameters->w_open; A.w_extend = parameters->w_extend;
    A.strategy = overhang_strategy;
    const size_t lds = static_cast<size_t>(max_m) + 2 * kWave;
    SW_TRY(hipEventRecord(sc->fork, stream));
    int launches = 0;
    bool side_used[kSideStreams] = {};
    for (int g = kNumGroups - 1; g >= 0; --g) {  // most rows per lane first: those waves run longest
        const int first = group_begin[g], count = group_begin[g + 1] - first;
        if (count == 0) continue;
        hipStream_t stream = sc->main;  // the first launch stays on the main stream, the others fan out
        if (launches > 0) {
            const int k = (launches - 1) % kSideStreams;
            stream = sc->side[k];
            if (!side_used[k]) SW_TRY(hipStreamWaitEvent(stream, sc->fork, 0));
            side_used[k] = true;
        }
        ++launches;
        if (staged) {
            pack(g);
            SW_TRY(hipMemcpyAsync(D + o_pool + group_pool[g + 1], host + o_pool + group_pool[g + 1], group_pool[g] - group_pool[g + 1],
                                  hipMemcpyHostToDevice, stream));
        }
        const bool packed = (g & 1) != 0;
        hipError_t e = hipErrorInvalidValue;
        switch (kRowChoices[g / 2]) {
            case 1: e = launch_fill<1>(A, packed, first, count, lds, stream); break;
            case 2: e = launch_fill<2>(A, packed, first, count, lds, stream); break;
            case 3: e = launch_fill<3>(A, packed, first, count, lds, stream); break;
            case 4: e = launch_fill<4>(A, packed, first, count, lds, stream); break;
            case 5: e = launch_fill<5>(A, packed, first, count, lds, stream); break;
            case 6: e = launch_fill<6>(A, packed, first, count, lds, stream); break;
            case 7: e = launch_fill<7>(A, packed, first, count, lds, stream); break;
            case 8: e = launch_fill<8>(A, packed, first, count, lds, stream); break;
            case 10: e = launch_fill<10>(A, packed, first, count, lds, stream); break;
            case 12: e = launch_fill<12>(A, packed, first, count, lds, stream); break;
            case 16: e = launch_fill<16>(A, packed, first, count, lds, stream); break;
            case 24: e = launch_fill<24>(A, packed, first, count, lds, stream); break;
            case 32: e = launch_fill<32>(A, packed, first, count, lds, stream); break;
        }
        SW_TRY(e);
    }
    for (int k = 0; k < kSideStreams; ++k) {
        if (!side_used[k]) continue;
        SW_TRY(hipEventRecord(sc->join[k], sc->side[k]));
        SW_TRY(hipStreamWaitEvent(stream, sc->join[k], 0));
    }
    // one back-track launch behind all fills (run next to them it slows them more than it gains)
    hipLaunchKernelGGL(sw_backtrack, dim3(num_pairs), dim3(kWave), 0, stream, A, 0, num_pairs);
    SW_TRY(hipGetLastError());
    // Raw element lists: alignments of haplotypes have a handful of elements, so the first kHeadElems of every
    // pair come back packed in one copy and the rest of the few longer lists afterwards.
    const int head = std::min(kHeadElems, elem_stride);
    const size_t results_bytes = align_up(pairs.size() * sizeof(PairResult), 256), heads_bytes = pairs.size() * 2 * kHeadElems * sizeof(int32_t);
    static_assert(sizeof(PairResult) == 8, "results and heads come back in one copy");
    SW_TRY(pinned_for(&sc->down, &sc->down_cap, results_bytes + heads_bytes));
    SW_TRY(hipMemcpyAsync(sc->down, D + o_results, results_bytes + heads_bytes, hipMemcpyDeviceToHost, stream));  // o_heads follows o_results
    SW_TRY(hipStreamSynchronize(stream));
    const PairResult* results = static_cast<const PairResult*>(sc->down);
    const int32_t* heads = reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(sc->down) + results_bytes);
    std::vector<int32_t> tails;                          // elements head.. of the pairs that have them, back to back
    std::vector<int64_t> tail_at(num_pairs, -1);         // a pair's offset in tails
    {
        size_t tail_ints = 0;
        for (int k = 0; k < num_pairs; ++k) {
            const int have = std::min(results[k].num_elems, pairs[k].elem_cap);
            if (have > head) { tail_at[k] = static_cast<int64_t>(tail_ints); tail_ints += 2 * static_cast<size_t>(have - head); }
        }
        if (tail_ints > 0) {
            tails.resize(tail_ints);
            for (int k = 0; k < num_pairs; ++k) {
                if (tail_at[k] < 0) continue;
                const int have = std::min(results[k].num_elems, pairs[k].elem_cap);
                const size_t src = (static_cast<size_t>(pairs[k].elem_off) + 2 * static_cast<size_t>(head)) * sizeof(int32_t);
                SW_TRY(hipMemcpyAsync(tails.data() + tail_at[k], D + o_elems + src, static_cast<size_t>(have - head) * 8, hipMemcpyDeviceToHost, stream));
            }
            SW_TRY(hipStreamSynchronize(stream));
        }
    }

    // ---- Collections.reverse + AlignmentUtils.consolidateCigar (sam/AlignmentUtils.java:698-727) -------
    int rc = GATK_SW_OK;
    for (int k = 0; k < num_pairs; ++k) {
        const int32_t* raw_head = heads + static_cast<size_t>(k) * 2 * kHeadElems;
        const int32_t* raw_tail = tail_at[k] >= 0 ? tails.data() + tail_at[k] : nullptr;
        if (results[k].num_elems > pairs[k].elem_cap)
            rc = fail(GATK_SW_CIGAR_TOO_LONG, "pair " + std::to_string(k) + " needs more than cigar_capacity elements");
        const int count = std::min(results[k].num_elems, pairs[k].elem_cap);
        int32_t* ops = cigar_ops + static_cast<size_t>(k) * cigar_capacity;
        int32_t* lens = cigar_lengths + static_cast<size_t>(k) * cigar_capacity;
        int out = 0, sum = 0, last = -1;
        auto emit = [&](int op, int len) {
            if (out < cigar_capacity) { ops[out] = op; lens[out] = len; }
            else rc = fail(GATK_SW_CIGAR_TOO_LONG, "pair " + std::to_string(k) + " needs more than cigar_capacity elements");
            ++out;
        };
        for (int x = count - 1; x >= 0; --x) {
            const int32_t* e = x < head ? raw_head + 2 * x : raw_tail + 2 * (x - head);
            const int op = e[0], len = e[1];
            if (len == 0) continue;
            if (last != -1 && last != op) { emit(last, sum); sum = 0; }
            sum += len;
            last = op;
        }
        if (sum > 0) emit(last, sum);
        cigar_counts[k] = out;
        alignment_offsets[k] = results[k].offset;
    }
    return rc;
} catch (const std::bad_alloc&) {  // no C++ exception crosses the C boundary
    return fail(GATK_SW_DEVICE_ERROR, "host memory exhausted");
} catch (const std::exception& e) {
    return fail(GATK_SW_DEVICE_ERROR, std::string("internal error: ") + e.what());
}

}  // extern "C"
