// PairHMM forward kernels for gfx950 (MI355X, wave64).
//
// What is computed (linear-space form of the reference recurrence,
// pairhmm/Log10PairHMM.java:108-113,211-219 with pairhmm/PairHMMModel.java:135-144 transitions):
//   M[i][j] = prior[i][j] * ( M[i-1][j-1]*t[i][0] + (I[i-1][j-1] + D[i-1][j-1])*t[i][1] )
//   I[i][j] =                 M[i-1][j]  *t[i][2] +  I[i-1][j]  *t[i][3]
//   D[i][j] =                 M[i][j-1]  *t[i][4] +  D[i][j-1]  *t[i][5]
//   result  = log10( sum_j M[R][j] + I[R][j] ) - log10(scale),  D[0][j] = scale / H
//
// How it is laid out on a wavefront:
//   * a wave owns a wave-group (one or more reads packed into lane segments, K read rows per
//     lane) and one haplotype; at step s the lane at segment position t works on haplotype
//     column s - t, so the wave sweeps anti-diagonals and a segment of L lanes needs H + L - 1
//     steps;
//   * the only values that cross lanes are, for the lane's LAST row, what the next read row
//     needs from it: Sn = M*t0' + (I + D)*t1' (the bracket of the next row's M one column later)
//     and In = M*t2' + I*t3' (the next row's I in this column), primed = next row's transitions.
//     Both move one lane up with a single v_mov_b32_dpp wave_shr:1 each per step (fp32 carries
//     them in a rescaled form, see Rows);
//   * the rows inside a lane hand the same two values down through registers;
//   * haplotype bases sit in an LDS image private to the wave; per-row transition constants are
//     expanded once per task from the read's quality bytes through look-up tables and stay in
//     registers;
//   * the emission term prior[i][j] is either selected per cell by comparing bytes (any
//     alphabet), or, when the haplotypes are made of A/C/G/T(/N) only, read from a per-wave LDS
//     table [row group][base code][lane][4 rows] that the lane fills once per task: one conflict-free
//     ds_read_b128 per four cells replaces the compare + select pairs on the vector ALU;
//   * lanes outside 1 <= column <= H are masked off, and every lane segment begins with a pad
//     row that reproduces matrix row 0, so every pair starts from exact boundary values and no
//     pair ever reads another pair's state.
//
// fp32 runs with scale 2^120; a task in which some pair's final sum is below the threshold (or
// is not finite) is queued and recomputed by the same code in fp64 with scale 2^1020.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pairhmm_device.h"

namespace pairhmm_hip {

// Profiling build only (make phase): shader-clock cycles each wave spends in the phases of a task,
// summed over waves.  0 decode+haplotype image, 1 row constants+table, 2 fill, 3 steady, 4 drain,
// 5 termination, 6 tasks.
#ifdef PAIRHMM_PHASE_CLOCKS
__device__ unsigned long long g_phase_clocks[8];
// One entry per wave of the stream pass, in order of completion: {start, end} on the constant 100 MHz counter,
// rows per lane, and the hardware id word (which XCD, CU and SIMD the wave ran on).
constexpr unsigned kWaveLogEntries = 1u << 18;
__device__ unsigned g_wave_log_count;
__device__ uint4 g_wave_log[kWaveLogEntries];
// The shares are kept in registers and added to the counters once, when the work item is over: an atomic in the
// middle of the item would stand in front of every later global load of the wave (they return in order).
#define PHASE_BEGIN() long long phase_t0 = clock64(); long long phase_acc[6] = {0, 0, 0, 0, 0, 0}
#define PHASE_END(n)                                  \
    do {                                              \
        const long long phase_t1 = clock64();         \
        phase_acc[n] += phase_t1 - phase_t0;          \
        phase_t0 = phase_t1;                          \
    } while (0)
#define PHASE_FLUSH()                                                                              \
    do {                                                                                           \
        if (lane == 0)                                                                             \
            for (int n_ = 0; n_ < 6; ++n_)                                                         \
                if (phase_acc[n_] != 0) atomicAdd(&g_phase_clocks[n_], (unsigned long long)phase_acc[n_]); \
    } while (0)
#else
#define PHASE_BEGIN() do {} while (0)
#define PHASE_END(n) do {} while (0)
#define PHASE_FLUSH() do {} while (0)
#endif

namespace {

__device__ __forceinline__ float shr1(float v) {
    // lane l receives lane l-1's value; lane 0 receives 0
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    return __builtin_bit_cast(float, r);
}

__device__ __forceinline__ double shr1(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(b), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(b >> 32), 0x138, 0xf, 0xf, true);
    const long long r = (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo);
    return __builtin_bit_cast(double, r);
}

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T> struct Scale;
template <> struct Scale<float> { static constexpr int kExp2 = 120; };
template <> struct Scale<double> { static constexpr int kExp2 = 1020; };

constexpr unsigned kNoBase = 0x100;  // compares unequal to every haplotype byte

// Base quality k of a read as the recurrence uses it (see kBaseQualityScoreThreshold).
__device__ __forceinline__ unsigned effective_qual(const DeviceBatch& B, const int k, const int read) {
    unsigned q = B.read_quals[k];
    if (B.read_mapq) { const unsigned m = B.read_mapq[read]; q = q < m ? q : m; }
    if (B.squash_base_quals && q < kBaseQualityScoreThreshold) q = kMinUsableQScore;
    return q;
}

// An input error found on the device: the kind goes into the error word, the smallest read index that had one
// into error_read (kept as INT32_MAX - index, so that zero means none and atomicMax keeps the first read).
__device__ __forceinline__ void report_input_error(const DeviceBatch& B, const int bad, const int read) {
    atomicOr(B.error_word, bad);
    if (B.error_read != nullptr && read >= 0) atomicMax(B.error_read, 0x7fffffff - read);
}

// Keeps a per-row constant in a register of its own type: without it the compiler carries the
// double it was converted from through the loop and converts after every select.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(unsigned& x) { asm volatile("" : "+v"(x)); }
constexpr int kFirstSpillingRowsFp64 = 10;  // from this many rows per lane on the fp64 one-haplotype pass no longer fits the register file

constexpr bool is_lut(int mode) { return mode == kEmitLut4 || mode == kEmitLut5; }
constexpr int lut_codes(int mode) { return mode == kEmitLut5 ? 5 : 4; }
// The emission table keeps kLutRowsPerRead read rows side by side, [row group][base code][lane][rows], so that one
// LDS read fetches several cells' terms.
// Four rows side by side: one ds_read_b128 fetches four cells' terms (conflict-free: consecutive lanes, consecutive
// 16-byte slots).  The same bytes as two ds_read_b64 in half the LDS instructions, which measured 3-4 % less
// kernel time on a device that runs this kernel at its power cap.  -DPAIRHMM_LUT_PAIRS builds the two-row layout.
#ifdef PAIRHMM_LUT_PAIRS
constexpr int kLutRowsPerRead = 2;
#else
constexpr int kLutRowsPerRead = 4;
#endif
constexpr int lut_row_pairs(int k) { return (k + kLutRowsPerRead - 1) / kLutRowsPerRead; }
constexpr int kLutCodeStride = kWave * 4 * kLutRowsPerRead;  // bytes between two base codes of one row pair
constexpr int lut_table_bytes(int k, int mode) { return is_lut(mode) ? lut_row_pairs(k) * lut_codes(mode) * kLutCodeStride : 0; }
// The one-haplotype kernel up to kUnmaskedRows rows per lane (the values region-sized calls take) runs its whole sweep
// without lane masks: its table has one more column, for the positions in front of and behind a haplotype, in which
// every row's emission term is 0 except the sum row's, which is 1.  A lane that is not inside its haplotype yet then
// recomputes column 0 (M = 0 stays 0, a pad row's deletion mass stays, the hand-offs are those of column 0), and a
// lane that has left it lets its sum row take in what is still on its way (the sum row's M + E + waiting hand-off is
// unchanged by such a step; the rows above it only produce values that go to lanes which have left as well).
constexpr int kUnmaskedRows = 6;
constexpr bool unmasked_sweep(int k, int mode) { return is_lut(mode) && k <= kUnmaskedRows; }
constexpr int direct_lut_codes(int k, int mode) { return lut_codes(mode) + (unmasked_sweep(k, mode) ? 1 : 0); }
constexpr int direct_table_bytes(int k, int mode) { return is_lut(mode) ? lut_row_pairs(k) * direct_lut_codes(k, mode) * kLutCodeStride : 0; }
// index, in floats, of the table entry of row r against base code c for this lane
__device__ __forceinline__ int lut_index(const int r, const int c, const int codes, const int lane) {
    return (((r / kLutRowsPerRead) * codes + c) * kWave + lane) * kLutRowsPerRead + (r % kLutRowsPerRead);
}
// the emission terms of the lane's K rows against the base code whose table offset is code_off.
// AHEAD: the read is one the caller issues a block of steps before it uses the terms.  The compiler otherwise sinks
// such a read to its first use, one loop iteration later, and the wave then waits out the LDS latency there: nothing
// to a SIMD with three waves, a quarter of the step time to a wave that has its SIMD to itself.  A volatile read
// stays where it is written.
// ROWS table entries of one lane from p (a pointer into LDS), read where the statement stands (volatile: see lut_emit;
// the LDS address space is spelled out because address-space inference leaves volatile accesses as flat ones)
template <int ROWS>
__device__ __forceinline__ float4 lut_read_in_place(const unsigned char* p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f3 __attribute__((ext_vector_type(3)));
    typedef float f2 __attribute__((ext_vector_type(2)));
#define LDS_VOLATILE(type) (const volatile __attribute__((address_space(3))) type*)
    if constexpr (ROWS >= 4) { const f4 w = *(LDS_VOLATILE(f4) p); return make_float4(w.x, w.y, w.z, w.w); }
    else if constexpr (ROWS == 3) { const f3 w = *(LDS_VOLATILE(f3) p); return make_float4(w.x, w.y, w.z, 0.0f); }
    else if constexpr (ROWS == 2) { const f2 w = *(LDS_VOLATILE(f2) p); return make_float4(w.x, w.y, 0.0f, 0.0f); }
    else { return make_float4(*(LDS_VOLATILE(float) p), 0.0f, 0.0f, 0.0f); }
#undef LDS_VOLATILE
}

template <typename T, int K, int CODES, bool AHEAD = false>
__device__ __forceinline__ void lut_emit(const unsigned char* tbl_lane, const unsigned code_off, T (&prior)[K]) {
#ifndef PAIRHMM_LUT_PAIRS
    if constexpr (AHEAD && K <= 8) {
        const float4 a = lut_read_in_place<(K < 4 ? K : 4)>(tbl_lane + code_off);
        prior[0] = a.x;
        if constexpr (K > 1) prior[1] = a.y;
        if constexpr (K > 2) prior[2] = a.z;
        if constexpr (K > 3) prior[3] = a.w;
        if constexpr (K > 4) {
            const float4 b = lut_read_in_place<K - 4>(tbl_lane + code_off + CODES * kLutCodeStride);
            prior[4] = b.x;
            if constexpr (K > 5) prior[5] = b.y;
            if constexpr (K > 6) prior[6] = b.z;
            if constexpr (K > 7) prior[7] = b.w;
        }
        return;
    }
#endif
#pragma unroll
    for (int g = 0; g < lut_row_pairs(K); ++g) {
        const unsigned char* p = tbl_lane + code_off + g * CODES * kLutCodeStride;
#ifndef PAIRHMM_LUT_PAIRS
        const float4 v = *reinterpret_cast<const float4*>(p);
        prior[4 * g] = v.x;
        if (4 * g + 1 < K) prior[4 * g + 1] = v.y;
        if (4 * g + 2 < K) prior[4 * g + 2] = v.z;
        if (4 * g + 3 < K) prior[4 * g + 3] = v.w;
#else
        const float2 v = *reinterpret_cast<const float2*>(p);
        prior[2 * g] = v.x;
        if (2 * g + 1 < K) prior[2 * g + 1] = v.y;
#endif
    }
}

// Registers of one lane: K read rows.  Two arithmetic forms share the layout:
//
// plain (fp64):  E = D / t4 (deletion recurrence is one fma), hand-offs Sn and In as in the header.
//
// scaled (fp32): every quantity that is only ever multiplied by a per-row constant is carried
//   divided by that constant, so that the constant disappears from the inner loop:
//     E  = D / t4                        E'  = M + E * t5
//     F  = I / t2                        Fn  = M + F * cF,  cF = t2 * t3' / t2'      (next row's F)
//     Sn = bracket of next row's M / t0' Sn  = M + F * cI + E * cE,
//                                        cI = t2 * t1' / t0',  cE = t4 * t1' / t0'
//     M  = prior * t0 * Sd               (t0 folded into the row's two prior constants)
//   5 arithmetic instructions per cell (mul, 4 fma) plus the emission term.  A row with t0 = 0
//   (insertion and deletion quality both <= 3) makes cI / cE infinite; the NaN that follows
//   reaches the final sum, fails the threshold test and sends the task to the plain fp64 pass.
template <typename T, int K>
struct Rows {
    T T5[K];
    T A[K], B[K], C[K], D[K], E[K];  // plain: t0', t1', t2', t3', t4*t1'   scaled: cF, cI, cE, -, -
    T Ml[K], El[K], Sd[K];
    T Sout, Iout;
};

template <typename T> struct IsScaled { static constexpr bool value = false; };
template <> struct IsScaled<float> { static constexpr bool value = true; };

// One haplotype column for all K rows of the lane.  prior[r]: emission term of row r in this
// column; Sin / Iin: hand-off of the row above the lane's first row in this column.
template <typename T, int K>
__device__ __forceinline__ void column(Rows<T, K>& w, const T (&prior)[K], const T Sin, const T Iin) {
    T Iup = Iin;    // I (plain) or F (scaled) of the current row in this column
    T Snext = Sin;  // Sn of the row above in this column: next column's diagonal term
    if constexpr (!IsScaled<T>::value) {
        // The plain form runs under a lane mask in most of its callers (a lane outside its haplotype skips the
        // column), where every new value that cannot take the register of the old one costs a copy at the end of
        // the masked block.  In this order every state register is dead when its new value is made: E first (the
        // last use of the old M), then M (the last use of the old diagonal term), then the hand-off chain, whose
        // results are the next column's diagonal terms.  Same operations, same results.
#pragma unroll
        for (int r = 0; r < K; ++r) w.El[r] = fma_(w.El[r], w.T5[r], w.Ml[r]);
#pragma unroll
        for (int r = 0; r < K; ++r) w.Ml[r] = prior[r] * w.Sd[r];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const T In = fma_(w.Ml[r], w.C[r], Iup * w.D[r]);
            const T Sn = fma_(w.El[r], w.E[r], fma_(Iup, w.B[r], w.Ml[r] * w.A[r]));
            w.Sd[r] = Snext;
            Snext = Sn;
            Iup = In;
        }
        w.Sout = Snext;
        w.Iout = Iup;
        return;
    }
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const T M = prior[r] * w.Sd[r];
        const T E = fma_(w.El[r], w.T5[r], w.Ml[r]);
        T In, Sn;
        if (IsScaled<T>::value) {
            In = fma_(Iup, w.A[r], M);
            Sn = fma_(E, w.C[r], fma_(Iup, w.B[r], M));
        } else {
            In = fma_(M, w.C[r], Iup * w.D[r]);
            Sn = fma_(E, w.E[r], fma_(Iup, w.B[r], M * w.A[r]));
        }
        w.Ml[r] = M;
        w.El[r] = E;
        w.Sd[r] = Snext;
        Snext = Sn;
        Iup = In;
    }
    w.Sout = Snext;
    w.Iout = Iup;
}

// The scaled column of the stream pass.  The diagonal term of a lane's first row is the hand-off Sn that the lane
// above made two steps earlier; column() has it fetched one step ahead into Sd[0] (a DPP move of its own) and
// multiplies a step later.  Here the lane above keeps that hand-off for one more step (Sheld), and the multiply of the
// first row reads it through DPP itself: one vector instruction less per step.  Sd0 is the value to multiply with
// (the caller passes shr1(w.Sheld), or 0 for a lane at a gap); Sd[0] is not used.
template <int K>
__device__ __forceinline__ void column_stream(Rows<float, K>& w, float& Sheld, const float (&prior)[K], const float Sd0, const float Iin) {
    float Iup = Iin;
    float Snext = 0.0f;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const float M = prior[r] * (r == 0 ? Sd0 : w.Sd[r]);
        const float E = fma_(w.El[r], w.T5[r], w.Ml[r]);
        const float In = fma_(Iup, w.A[r], M);
        const float Sn = fma_(E, w.C[r], fma_(Iup, w.B[r], M));
        w.Ml[r] = M;
        w.El[r] = E;
        if (r > 0) w.Sd[r] = Snext;
        Snext = Sn;
        Iup = In;
    }
    Sheld = w.Sout;  // the hand-off of the step before this one: what the lane below multiplies with in its next step
    w.Sout = Snext;
    w.Iout = Iup;
}

// The column for loops whose steps alternate PARITY 0, 1, 0, 1 ... and run under a lane mask (a lane outside its
// haplotype skips the column).  Under a mask every new state value that cannot take the register of the old one costs a
// copy at the end of the masked block, and the deletion term E = E' * t5 + M' is an FMA whose addend is the old M: the
// two-address v_fmac the compiler emits for it leaves the result in the addend's register.  So the operations come in
// an order in which every state register is dead when its new value is made (E first: the last use of the old M; then
// M: the last use of the old diagonal term; then the hand-off chain, whose results are the next column's diagonal
// terms), and M and E trade registers every step: PARITY 0 finds (M', E') in (Ml, El) and leaves (E, M) there, PARITY 1
// finds (E', M') and leaves (M, E).  A lane whose first column falls on an odd step starts with the two swapped; the
// final sum adds both, so it does not care.  Same operations as column(), same results.
template <typename T, int K, int PARITY>
__device__ __forceinline__ void column_swapping(Rows<T, K>& w, const T (&prior)[K], const T Sin, const T Iin) {
    T (&Mslot)[K] = PARITY == 0 ? w.El : w.Ml;  // where this column's M goes: the register that held the old E
    T (&Eslot)[K] = PARITY == 0 ? w.Ml : w.El;  // where its E goes: the register that held the old M
#pragma unroll
    for (int r = 0; r < K; ++r) Eslot[r] = fma_(Mslot[r], w.T5[r], Eslot[r]);  // old E sits in Mslot, old M in Eslot
#pragma unroll
    for (int r = 0; r < K; ++r) Mslot[r] = prior[r] * w.Sd[r];
    T Iup = Iin, Snext = Sin;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        T In, Sn;
        if (IsScaled<T>::value) {
            In = fma_(Iup, w.A[r], Mslot[r]);
            Sn = fma_(Eslot[r], w.C[r], fma_(Iup, w.B[r], Mslot[r]));
        } else {
            In = fma_(Mslot[r], w.C[r], Iup * w.D[r]);
            Sn = fma_(Eslot[r], w.E[r], fma_(Iup, w.B[r], Mslot[r] * w.A[r]));
        }
        w.Sd[r] = Snext;
        Snext = Sn;
        Iup = In;
    }
    w.Sout = Snext;
    w.Iout = Iup;
}

// LDS image of one wave.
//   compare modes: [64 pad][haplotype bytes][pad]                       (byte per base)
//   table modes  : [K/2][codes][64][2] emission table | [64 pad][code*512 as u16 per base][pad]
// Base codes are (byte >> 1) & 3 (A 0, C 1, T 2, G 3) and 4 for 'N'; the host only selects a table
// mode when every haplotype byte of the launch is one of those letters.
template <int K, int MODE>
struct Image {
    static constexpr int kTableBytes = direct_table_bytes(K, MODE);
};

// What a lane knows about its place in the wave-group.
struct LaneInfo {
    int read_idx;   // read owning this lane, or -1 for an unused lane
    int first_row;  // 1-based read row of this lane's row 0; <= 0 means leading pad rows
    int seg_lane;   // lane index inside the segment (0 = segment start)
    int seg_last;   // 1 if this lane holds the read's last row
};

// The optional per-base arrays: a call in which no read has one passes a null pointer and every base takes the
// default (flat Q45 insertion / deletion qualities, sam/GATKSAMRecord.java:61,298-322; gap-continuation penalty 10).
__device__ __forceinline__ unsigned ins_byte_at(const DeviceBatch& B, const int idx) { return B.ins_gop ? B.ins_gop[idx] : static_cast<unsigned>(kDefaultIndelQual); }
__device__ __forceinline__ unsigned del_byte_at(const DeviceBatch& B, const int idx) { return B.del_gop ? B.del_gop[idx] : static_cast<unsigned>(kDefaultIndelQual); }
__device__ __forceinline__ unsigned gcp_byte_at(const DeviceBatch& B, const int idx) { return B.gcp ? B.gcp[idx] : static_cast<unsigned>(kDefaultGcp); }

// Entry i of a read's RowConst list (0: the pad row above row 1, 1..R: the read's rows, R + 1: the sum row)
// from the quality bytes, with the checks the reference makes while expanding a read's qualities
// (PairHMMModel.java:136-138: negative ins/del/GCP byte -> IllegalArgumentException, bit 0 of `bad`;
// QualityUtils.java:173: base quality 255 indexes past the 255-entry cache, bit 1).  The arithmetic is done
// in double and rounded once (PairHMMModel.java:109-143 for the transitions, QualityUtils.java:169-176 for
// the priors).
__device__ __forceinline__ RowConst make_row_const(const DeviceBatch& B, const int read, const int ro, const int R, const int i, int& bad) {
    RowConst c{0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, kNoBase, 0u};
    double t2 = 0.0, t4 = 1.0;  // a pad row hands its deletion mass down unweighted
    if (i >= 1 && i <= R) {
        const unsigned x = B.read_bases[ro + i - 1];
        const unsigned ins_byte = ins_byte_at(B, ro + i - 1), del_byte = del_byte_at(B, ro + i - 1), gcp_byte = gcp_byte_at(B, ro + i - 1);
        if ((ins_byte | del_byte | gcp_byte) & 0x80u) bad |= 1;
        const unsigned qual = effective_qual(B, ro + i - 1, read);  // what the PairHMM is handed: after the mapping-quality cap
        if (qual == 255) bad |= 2;
        const double e = B.err_prob[qual];
        const unsigned ins = ins_byte & 127u, del = del_byte & 127u;
        const double t0 = B.match_to_match[ins * 128u + del];
        const double p = 1.0 - e;
        const double q = (x == 'N') ? p : e / B.mismatch_divisor;  // a read 'N' matches every base
        t2 = B.err_prob[ins];
        t4 = B.err_prob[del];
        c.p_match = static_cast<float>(p * t0);
        c.p_mismatch = static_cast<float>(q * t0);
        c.t5 = static_cast<float>(B.err_prob[gcp_byte]);
        c.base = x;
    }
    if (i < R) {  // transitions of read row i + 1, taken from quals[i]
        const unsigned ins = ins_byte_at(B, ro + i) & 127u, del = del_byte_at(B, ro + i) & 127u;
        const double eg = B.err_prob[gcp_byte_at(B, ro + i)];
        const double t0n = B.match_to_match[ins * 128u + del];
        const double t1n = 1.0 - eg;
        if (i >= 1) {
            c.c_f = static_cast<float>(t2 * eg / B.err_prob[ins]);
            c.c_i = static_cast<float>(t2 * t1n / t0n);
        }
        c.c_e = static_cast<float>(t4 * t1n / t0n);
    } else if (i == R) {
        c.c_i = static_cast<float>(t2);  // last row: Sn = M + F * t2 = M + I, what the sum row adds up
    } else {
        c.p_match = c.p_mismatch = 1.0f;  // sum row
    }
    return c;
}

// The RowConst entries of a lane's K rows (entry of row r: first_row + r, kept inside 0 .. R + 1), for the kernel that
// works them out itself.  The same arithmetic as make_row_const, entry by entry, laid out for a wave that is alone on
// its SIMD: all quality bytes of all rows are fetched first, then all table entries, then the arithmetic, and what kind
// of row an entry is (pad row, read row, last row, sum row) is settled by selects.  Row by row with a branch per kind,
// the wave made some twenty dependent trips to memory, which was most of its set-up time.  Indices of rows that have no
// byte of their own are held inside the read, so every load is valid and its result is dropped by the selects.
template <int K>
__device__ __forceinline__ void row_consts_of_lane(const DeviceBatch& B, const int read, const int ro, const int R, const int first_row,
                                                   float4 (&c0)[K], float4 (&c1)[K], int& bad) {
    int i_of[K], ia[K], ib[K];
    const int last_byte = R > 0 ? R - 1 : 0;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        int i = first_row + r;
        i = i < 0 ? 0 : (i > R + 1 ? R + 1 : i);
        i_of[r] = i;
        const int a = i - 1 < 0 ? 0 : (i - 1 > last_byte ? last_byte : i - 1);
        const int b = i > last_byte ? last_byte : i;
        ia[r] = ro + a;  // byte of row i
        ib[r] = ro + b;  // byte of row i + 1
    }
    unsigned x[K], qual[K], ins[K], del[K], gcp[K], ins_n[K], del_n[K], gcp_n[K];
#pragma unroll
    for (int r = 0; r < K; ++r) { x[r] = B.read_bases[ia[r]]; qual[r] = B.read_quals[ia[r]]; }
    if (B.ins_gop) {
#pragma unroll
        for (int r = 0; r < K; ++r) { ins[r] = B.ins_gop[ia[r]]; ins_n[r] = B.ins_gop[ib[r]]; }
    } else {
#pragma unroll
        for (int r = 0; r < K; ++r) ins[r] = ins_n[r] = static_cast<unsigned>(kDefaultIndelQual);
    }
    if (B.del_gop) {
#pragma unroll
        for (int r = 0; r < K; ++r) { del[r] = B.del_gop[ia[r]]; del_n[r] = B.del_gop[ib[r]]; }
    } else {
#pragma unroll
        for (int r = 0; r < K; ++r) del[r] = del_n[r] = static_cast<unsigned>(kDefaultIndelQual);
    }
    if (B.gcp) {
#pragma unroll
        for (int r = 0; r < K; ++r) { gcp[r] = B.gcp[ia[r]]; gcp_n[r] = B.gcp[ib[r]]; }
    } else {
#pragma unroll
        for (int r = 0; r < K; ++r) gcp[r] = gcp_n[r] = static_cast<unsigned>(kDefaultGcp);
    }
    const unsigned mapq = (B.read_mapq && read >= 0) ? B.read_mapq[read] : 255u;
    double e[K], t0[K], t2[K], t4[K], t5[K], eg[K], t0n[K], t2n[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const bool real = i_of[r] >= 1 && i_of[r] <= R;
        if (real && ((ins[r] | del[r] | gcp[r]) & 0x80u)) bad |= 1;
        unsigned q = qual[r] < mapq ? qual[r] : mapq;  // what the PairHMM is handed: after the mapping-quality cap
        if (B.squash_base_quals && q < kBaseQualityScoreThreshold) q = kMinUsableQScore;
        if (real && q == 255) bad |= 2;
        e[r] = B.err_prob[q];
        t0[r] = B.match_to_match[(ins[r] & 127u) * 128u + (del[r] & 127u)];
        t2[r] = B.err_prob[ins[r] & 127u];
        t4[r] = B.err_prob[del[r] & 127u];
        t5[r] = B.err_prob[gcp[r]];
        eg[r] = B.err_prob[gcp_n[r]];
        t0n[r] = B.match_to_match[(ins_n[r] & 127u) * 128u + (del_n[r] & 127u)];
        t2n[r] = B.err_prob[ins_n[r] & 127u];
    }
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const int i = i_of[r];
        const bool real = i >= 1 && i <= R, has_next = i < R, last = i == R, sum_row = i > R;
        const double p = 1.0 - e[r];
        const double q = (x[r] == 'N') ? p : e[r] / B.mismatch_divisor;  // a read 'N' matches every base
        const double m2 = real ? t2[r] : 0.0, m4 = real ? t4[r] : 1.0;    // a pad row hands its deletion mass down unweighted
        const double t1n = 1.0 - eg[r];
        const float c_f = static_cast<float>(m2 * eg[r] / t2n[r]);
        const float c_i = static_cast<float>(m2 * t1n / t0n[r]);
        const float c_e = static_cast<float>(m4 * t1n / t0n[r]);
        c0[r].x = real ? static_cast<float>(p * t0[r]) : (sum_row ? 1.0f : 0.0f);
        c0[r].y = real ? static_cast<float>(q * t0[r]) : (sum_row ? 1.0f : 0.0f);
        c0[r].z = real ? static_cast<float>(t5[r]) : 1.0f;
        c0[r].w = (has_next && i >= 1) ? c_f : 0.0f;
        c1[r].x = has_next ? (i >= 1 ? c_i : 0.0f) : (last ? static_cast<float>(m2) : 0.0f);  // last row: Sn = M + F * t2 = M + I
        c1[r].y = has_next ? c_e : 0.0f;
        c1[r].z = __uint_as_float(real ? x[r] : kNoBase);
        c1[r].w = 0.0f;
    }
}

template <typename T, int K, int MODE>
__device__ __forceinline__ bool run_task(const DeviceBatch& B, const int task_idx, unsigned char* slice,
                                         const int lane, const bool queue_underflow) {
    constexpr bool kScaled = IsScaled<T>::value;
    constexpr bool kLut = is_lut(MODE);
    constexpr int kCodes = lut_codes(MODE);
    constexpr bool kUnmasked = unmasked_sweep(K, MODE);       // the whole sweep without lane masks, see direct_lut_codes
    constexpr int kTblCodes = direct_lut_codes(K, MODE);      // columns of this kernel's table
    constexpr unsigned kOutside = kCodes * kLutCodeStride;    // table offset of the column for positions outside the haplotype
    static_assert(!kLut || std::is_same<T, float>::value, "the emission table holds floats");
    PHASE_BEGIN();
    // ---- which wave-group and haplotype is this work item? (wave-uniform, scalar) ----------------
    int lo = 0, hi = B.num_region_descs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (B.region_descs[mid].task_begin <= task_idx) lo = mid; else hi = mid - 1;
    }
    const RegionDesc rdesc = B.region_descs[lo];
    const int local = task_idx - rdesc.task_begin;
    const int hap_local = local / rdesc.num_groups;
    const int hap_idx = rdesc.hap_begin + hap_local;
    const GroupDesc grp = B.groups[rdesc.group_begin + (local - hap_local * rdesc.num_groups)];
    const int H = B.hap_len[hap_idx];
    const int fill = grp.fill;
    // ---- which read segment does this lane belong to? ----------------------------------------------
    LaneInfo ld{-1, 1, 0, 0};
    for (int k = 0; k < grp.num_segs; ++k) {
        const SegDesc sd = B.segs[grp.seg_begin + k];
        const int t = lane - static_cast<int>(sd.lane_begin);
        if (t >= 0 && t < static_cast<int>(sd.seg_lanes)) {
            const int len = sd.read_len;
            ld.read_idx = sd.read_idx;
            ld.seg_lane = t;
            ld.first_row = t * K + 1 - (static_cast<int>(sd.seg_lanes) * K - len - 1);  // pads | rows 1..len | sum row
            ld.seg_last = (t == static_cast<int>(sd.seg_lanes) - 1) ? 1 : 0;
        }
    }
    unsigned char* const hap_img = slice + Image<K, MODE>::kTableBytes;

    // ---- haplotype bases -> this wave's LDS image --------------------------------------------
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(B.hap_pool + B.hap_off[hap_idx]);
        const int words = (H + 3) >> 2;
        if (kLut) {
            // every entry the sweep can touch must hold a valid table offset: the pads hold column 0, or, for the
            // sweep without lane masks, the column for positions outside the haplotype (as do the entries behind
            // base H - 1 in the haplotype's last word)
            uint16_t* img = reinterpret_cast<uint16_t*>(hap_img);
            constexpr unsigned kPad = kUnmasked ? kOutside : 0u;
            img[lane] = static_cast<uint16_t>(kPad);
            for (int d = lane; d < kHapBackPad; d += kWave) img[kHapFrontPad + words * 4 + d] = static_cast<uint16_t>(kPad);
            uint2* dst = reinterpret_cast<uint2*>(img + kHapFrontPad);
            for (int d = lane; d < words; d += kWave) {
                const uint32_t v = src[d];
                unsigned c[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const unsigned byte = (v >> (8 * b)) & 0xffu;
                    c[b] = ((byte == 'N' && kCodes == 5) ? 4u : ((byte >> 1) & 3u)) * kLutCodeStride;
                    if (kUnmasked && 4 * d + b >= H) c[b] = kOutside;
                }
                dst[d] = make_uint2(c[0] | (c[1] << 16), c[2] | (c[3] << 16));
            }
        } else {
            uint32_t* dst = reinterpret_cast<uint32_t*>(hap_img + kHapFrontPad);
            for (int d = lane; d < words; d += kWave) dst[d] = src[d];
        }
    }

    PHASE_END(0);
    // ---- per-row constants ---------------------------------------------------------------------
    // Every segment starts with at least one pad row.  A pad row behaves as matrix row 0
    // (M = I = 0, D = scale / H in every column) whatever the lane above hands down, so the value
    // a segment's first lane receives from the previous segment never reaches a real row.
    //
    // Every segment ends with a sum row (row R + 1, always row K-1 of the segment's last lane).  Read
    // row R hands it Sn = M[R][j] + I[R][j]; with emission 1 and a deletion-to-deletion weight of 1
    // its M is that hand-off one column later and its E is the running sum of its M, so after the
    // last column  sum_j (M[R][j] + I[R][j]) = E + M + (the hand-off still waiting in Sd).  The final
    // sum therefore costs no instruction in the sweep.
    const T initD = static_cast<T>(::ldexp(1.0, Scale<T>::kExp2) / static_cast<double>(H));
    int ro = 0, R = 0;
    int out_base = 0;  // fetched here, with the other per-read words: at the end of the sweep the load would stand alone
    if (ld.read_idx >= 0) {
        ro = B.read_off[ld.read_idx];
        R = B.read_off[ld.read_idx + 1] - ro;
        out_base = B.read_out_base[ld.read_idx];
    }
    Rows<T, K> w;
    unsigned XB[K];
    T P[K], Q[K];
    if constexpr (kScaled) {
        // expanded once per read by prepare_read: two 16-byte loads per row, no branches, every
        // load of the lane in flight at once
        const int rc_base = ro + 2 * (ld.read_idx >= 0 ? ld.read_idx : 0);
        float4 c0[K], c1[K];
        if (B.row_consts != nullptr) {
#pragma unroll
            for (int r = 0; r < K; ++r) {
                int e = ld.first_row + r;
                e = e < 0 ? 0 : (e > R + 1 ? R + 1 : e);
                const float4* src = reinterpret_cast<const float4*>(B.row_consts + rc_base + e);
                c0[r] = src[0];
                c1[r] = src[1];
            }
        } else {
            // a region-sized call: no expansion pass ran, the same entries are worked out here (and the
            // quality checks with them)
            int bad = 0;
            row_consts_of_lane<K>(B, ld.read_idx, ro, R, ld.first_row, c0, c1, bad);
            if (bad) report_input_error(B, bad, ld.read_idx);
        }
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int i = ld.first_row + r;
            const bool used = ld.read_idx >= 0;
            const bool deep_pad = i < 0;  // the row below is a pad row too: hand down anything finite
            P[r] = (used && !deep_pad) ? c0[r].x : 0.0f;
            Q[r] = (used && !deep_pad) ? c0[r].y : 0.0f;
            w.T5[r] = !used ? 0.0f : deep_pad ? 1.0f : c0[r].z;
            w.A[r] = (used && !deep_pad) ? c0[r].w : 0.0f;
            w.B[r] = (used && !deep_pad) ? c1[r].x : 0.0f;
            w.C[r] = !used ? 0.0f : deep_pad ? 1.0f : c1[r].y;
            XB[r] = (used && !deep_pad) ? __float_as_uint(c1[r].z) : kNoBase;
            pin(w.A[r]); pin(w.B[r]); pin(w.C[r]); pin(w.T5[r]);
            w.Ml[r] = 0.0f;
            w.El[r] = (used && i <= 0) ? initD : 0.0f;  // column 0
            if (kLut) {
                // emission table: entry [r / 2][code][lane][r % 2] = prior of row r against base code
                float* tbl = reinterpret_cast<float*>(slice);
                const unsigned letters[5] = {'A', 'C', 'T', 'G', 'N'};
#pragma unroll
                for (int c = 0; c < kCodes; ++c)
                    tbl[lut_index(r, c, kTblCodes, lane)] = (XB[r] == letters[c] || c == 4) ? P[r] : Q[r];
                if (kUnmasked) tbl[lut_index(r, kCodes, kTblCodes, lane)] = (used && i == R + 1) ? 1.0f : 0.0f;  // outside the haplotype
            } else {
                pin(XB[r]); pin(P[r]); pin(Q[r]);
            }
        }
    } else {
        // plain form, straight from the quality bytes (rare: only pairs the fp32 pass gave up on)
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int i = ld.first_row + r;  // 1-based read row; <= 0: pad row
            double p = 0, q = 0, t4 = 0, t5 = 0, t0n = 0, t1n = 0, t2n = 0, t3n = 0;
            bool pad = false, has_next = false;
            unsigned xb = kNoBase;
            const bool sum_row = ld.read_idx >= 0 && i == R + 1;
            const bool last_row = ld.read_idx >= 0 && i == R;
            if (sum_row) {
                p = q = 1.0;
                t5 = 1.0;
            } else if (ld.read_idx >= 0) {
                if (i >= 1) {
                    const unsigned x = B.read_bases[ro + i - 1];
                    const double e = B.err_prob[effective_qual(B, ro + i - 1, ld.read_idx)];
                    xb = x;
                    p = 1.0 - e;                         // QualityUtils.qualToProb
                    q = (x == 'N') ? p : e / B.mismatch_divisor;        // a read 'N' matches every base
                    t4 = B.err_prob[del_byte_at(B, ro + i - 1) & 127u];  // matchToDeletion
                    t5 = B.err_prob[gcp_byte_at(B, ro + i - 1)];             // deletionToDeletion
                } else {
                    pad = true;
                    t4 = 1.0;
                    t5 = 1.0;
                    if (i < 0) t1n = 1.0;  // next row is a pad row too: anything finite
                }
                if (i >= 0 && i < R) {  // transitions of read row i+1, taken from quals[i]
                    const unsigned ins = ins_byte_at(B, ro + i) & 127u, del = del_byte_at(B, ro + i) & 127u;
                    const double eg = B.err_prob[gcp_byte_at(B, ro + i)];
                    has_next = true;
                    t0n = B.match_to_match[ins * 128u + del];
                    t1n = 1.0 - eg;          // indelToMatch
                    t2n = B.err_prob[ins];   // matchToInsertion
                    t3n = eg;                // insertionToInsertion
                }
            }
            const bool real_next = has_next && i >= 1;
            XB[r] = xb;
            P[r] = static_cast<T>(p);
            Q[r] = static_cast<T>(q);
            w.T5[r] = static_cast<T>(t5);
            w.A[r] = static_cast<T>(real_next ? t0n : last_row ? 1.0 : 0.0);  // last row: Sn = M + I
            w.B[r] = static_cast<T>(last_row ? 1.0 : t1n);
            w.C[r] = static_cast<T>(real_next ? t2n : 0.0);
            w.D[r] = static_cast<T>(real_next ? t3n : 0.0);
            w.E[r] = static_cast<T>(t4 * t1n);
            // Not in the fp64 instantiations that spill registers (10 rows per lane and more): built with spills into
            // accumulation registers, those returned wrong sums with these empty asm statements on doubles in them.
            if constexpr (!(std::is_same<T, double>::value && K >= kFirstSpillingRowsFp64)) {
                pin(w.A[r]); pin(w.B[r]); pin(w.C[r]); pin(w.D[r]); pin(w.E[r]); pin(w.T5[r]);
                pin(XB[r]); pin(P[r]); pin(Q[r]);
            }
            w.Ml[r] = static_cast<T>(0);
            w.El[r] = pad ? initD : static_cast<T>(0);  // column 0
        }
    }
    // column-0 hand-offs: Sn of the row above, where M = I = 0
    const T* cS = kScaled ? w.C : w.E;
#pragma unroll
    for (int r = 1; r < K; ++r) w.Sd[r] = w.El[r - 1] * cS[r - 1];
    w.Sout = w.El[K - 1] * cS[K - 1];
    w.Iout = static_cast<T>(0);
    w.Sd[0] = shr1(w.Sout);
    __builtin_amdgcn_wave_barrier();

    // ---- anti-diagonal sweep -------------------------------------------------------------------
    // step s: the lane at segment position t works on haplotype column s - t (0-based).
    //   [0, fill)            pipeline fill: lanes with s < t wait              -> masked
    //   [fill, H)            every lane of every segment is inside 0..H-1      -> no mask
    //   [max(H, fill), H+fill) pipeline drain: lanes with s - t >= H are done  -> masked
    const int steady_begin = fill < H ? fill : H;
    int s = 0;
    // Every column of the sweep is a column_swapping of its step's parity (in the unrolled blocks, which start on even
    // steps, the parity is a constant): a lane whose first column is an odd step starts with M and E swapped.
    // (A sweep without lane masks runs every lane through every step from step 0 on: nothing to swap.)
    if (!kUnmasked && (ld.seg_lane & 1)) {
#pragma unroll
        for (int r = 0; r < K; ++r) { const T e = w.El[r]; w.El[r] = w.Ml[r]; w.Ml[r] = e; }
    }
    auto col = [&](const int parity, const T (&prior)[K], const T Sin, const T Iin) {
        if (parity & 1) column_swapping<T, K, 1>(w, prior, Sin, Iin);
        else column_swapping<T, K, 0>(w, prior, Sin, Iin);
    };
    PHASE_END(1);
    if (kLut) {
        const uint16_t* hp = reinterpret_cast<const uint16_t*>(hap_img) + (kHapFrontPad - ld.seg_lane);
        const unsigned char* tbl_lane = slice + lane * 4 * kLutRowsPerRead;
        auto emit = [&](const unsigned code_off, T (&prior)[K]) { lut_emit<T, K, kTblCodes>(tbl_lane, code_off, prior); };
        // reads issued a block ahead of their use; pinned in place where the wave is likely alone on its SIMD
        auto emit_ahead = [&](const unsigned code_off, T (&prior)[K]) { lut_emit<T, K, kTblCodes, (K <= 6)>(tbl_lane, code_off, prior); };
        if constexpr (kUnmasked) {
            // One loop for fill, steady steps and drain: H + fill steps, every lane in every one of them; outside its
            // haplotype a lane reads the table's extra column.  Eight steps per turn, the two sets of emission terms
            // taking turns, reads a block ahead of their use.
            const int end = H + fill;
            T pa[4][K], pb[4][K];
            unsigned yn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) emit_ahead(hp[u], pa[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) yn[u] = hp[4 + u];
            auto block = [&](T (&cur)[4][K], T (&nxt)[4][K]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) emit_ahead(yn[u], nxt[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) yn[u] = hp[s + 8 + u];  // at most fill + 11 entries behind the haplotype: inside the back pad
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const T Sin = shr1(w.Sout);
                    const T Iin = shr1(w.Iout);
                    col(u, cur[u], Sin, Iin);
                }
                s += 4;
            };
            auto rest = [&](T (&cur)[4][K]) {  // the last 0 to 3 steps: their terms are in hand
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (s + u < end) {
                        const T Sin = shr1(w.Sout);
                        const T Iin = shr1(w.Iout);
                        col(u, cur[u], Sin, Iin);
                    }
                s = end;
            };
            while (s + 8 <= end) { block(pa, pb); block(pb, pa); }
            if (s + 4 <= end) { block(pa, pb); rest(pb); } else { rest(pa); }
            PHASE_END(3);
        }
        auto masked_step = [&](const int step) {
            const T Sin = shr1(w.Sout);
            const T Iin = shr1(w.Iout);
            T prior[K];
            emit(hp[step], prior);
            if (static_cast<unsigned>(step - ld.seg_lane) < static_cast<unsigned>(H)) col(step, prior, Sin, Iin);
        };
        // Fill and drain in blocks of four steps with the same read-ahead as the steady blocks, up to 6 rows per lane
        // (the values region-sized calls take: their waves are alone on a SIMD and would wait out two dependent LDS
        // reads per step; with more rows the second set of emission terms costs registers that larger calls, whose
        // other waves cover the latency, need for occupancy).  Every entry the read-ahead can touch holds a valid
        // table offset: the pads in front of column 0 and behind column H - 1 are zeroed.
        auto masked_run = [&](const int end) {
            if constexpr (K <= 6) {
                if ((s & 1) && s < end) { masked_step(s); ++s; }
                if (s + 4 <= end) {
                    // two sets of emission terms take turns (no copy from "next" to "current" at the end of a block,
                    // which would wait for the reads it has just issued)
                    T ma[4][K], mb[4][K];
                    unsigned mn[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) emit_ahead(hp[s + u], ma[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) mn[u] = hp[s + 4 + u];
                    auto masked_block = [&](T (&cur)[4][K], T (&nxt)[4][K]) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) emit_ahead(mn[u], nxt[u]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) mn[u] = hp[s + 8 + u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const T Sin = shr1(w.Sout);
                            const T Iin = shr1(w.Iout);
                            if (static_cast<unsigned>(s + u - ld.seg_lane) < static_cast<unsigned>(H)) col(u, cur[u], Sin, Iin);
                        }
                        s += 4;
                    };
                    while (s + 8 <= end) { masked_block(ma, mb); masked_block(mb, ma); }
                    if (s + 4 <= end) masked_block(ma, mb);
                }
            }
            for (; s < end; ++s) masked_step(s);
        };
        if constexpr (!kUnmasked) {
        masked_run(steady_begin);
        if (s & 1) { masked_step(s); ++s; }  // the steady blocks start on an even step
        PHASE_END(2);
        if (s + 4 <= H) {
            // two-deep software pipeline: emission terms of the next four columns and base codes of
            // the four after them are in flight while the current four columns are computed
            T pc[4][K];
            unsigned yn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) emit_ahead(hp[s + u], pc[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) yn[u] = hp[s + 4 + u];
            if constexpr (K <= 6) {
                // a wave that is likely alone on its SIMD: the two sets of terms take turns, so that no block ends
                // with a copy that waits for the reads it has just issued
                T pd[4][K];
                auto steady_block = [&](T (&cur)[4][K], T (&nxt)[4][K]) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) emit_ahead(yn[u], nxt[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) yn[u] = hp[s + 8 + u];  // past H: lands in the zeroed tail pad
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const T Sin = shr1(w.Sout);
                        const T Iin = shr1(w.Iout);
                        col(u, cur[u], Sin, Iin);
                    }
                    s += 4;
                };
                while (s + 8 <= H) { steady_block(pc, pd); steady_block(pd, pc); }
                if (s + 4 <= H) steady_block(pc, pd);
            } else {
                for (; s + 4 <= H; s += 4) {
                    T pn[4][K];
#pragma unroll
                    for (int u = 0; u < 4; ++u) emit_ahead(yn[u], pn[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) yn[u] = hp[s + 8 + u];  // past H: lands in the zeroed tail pad
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const T Sin = shr1(w.Sout);
                        const T Iin = shr1(w.Iout);
                        col(u, pc[u], Sin, Iin);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int r = 0; r < K; ++r) pc[u][r] = pn[u][r];
                }
            }
        }
        PHASE_END(3);
        masked_run(H + fill);
        PHASE_END(4);
        }
    } else {
        const unsigned char* hp = hap_img + (kHapFrontPad - ld.seg_lane);
        auto emit = [&](const unsigned yb, T (&prior)[K]) {
#pragma unroll
            for (int r = 0; r < K; ++r) {
                bool match = (yb == XB[r]);
                if (MODE == kEmitCompareN) match = match || (yb == 'N');
                prior[r] = match ? P[r] : Q[r];
            }
        };
        auto masked_step = [&](const int step) {
            const T Sin = shr1(w.Sout);
            const T Iin = shr1(w.Iout);
            const unsigned yb = hp[step];
            if (static_cast<unsigned>(step - ld.seg_lane) < static_cast<unsigned>(H)) {
                T prior[K];
                emit(yb, prior);
                col(step, prior, Sin, Iin);
            }
        };
        for (; s < steady_begin; ++s) masked_step(s);
        if (s & 1) { masked_step(s); ++s; }  // the steady blocks start on an even step
        if (s + 4 <= H) {
            unsigned n0 = hp[s], n1 = hp[s + 1], n2 = hp[s + 2], n3 = hp[s + 3];
            for (; s + 4 <= H; s += 4) {
                const unsigned y0 = n0, y1 = n1, y2 = n2, y3 = n3;
                // next iteration's bases; reads past H land in the image's tail padding
                n0 = hp[s + 4]; n1 = hp[s + 5]; n2 = hp[s + 6]; n3 = hp[s + 7];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const T Sin = shr1(w.Sout);
                    const T Iin = shr1(w.Iout);
                    T prior[K];
                    emit(u == 0 ? y0 : u == 1 ? y1 : u == 2 ? y2 : y3, prior);
                    col(u, prior, Sin, Iin);
                }
            }
        }
        for (; s < H + fill; ++s) masked_step(s);
    }

    // ---- termination ---------------------------------------------------------------------------
    bool under = false;
    if (ld.seg_last != 0 && ld.read_idx >= 0) {
        const T total = w.El[K - 1] + w.Ml[K - 1] + w.Sd[K - 1];  // the sum row, see above
        const double res = ::log10(static_cast<double>(total)) - Scale<T>::kExp2 * 0.30102999566398119521;
        B.out[out_base + hap_local] = res;
        // too small for fp32 to be trusted, or NaN / infinity from the scaled form: redo in fp64
        if (kScaled) under = !(static_cast<float>(total) >= B.fp32_threshold && static_cast<float>(total) <= 3.0e38f);
    }
    bool again = false;  // wave-uniform: some pair of this work item has to be redone in fp64
    if (queue_underflow) {
        again = __any(under ? 1 : 0) != 0;
        if (again) {
            if (lane == 0) {
                const int pos = atomicAdd(B.rerun_count, 1);
                B.rerun_list[pos] = task_idx;
            }
        }
    }
    PHASE_END(5);
    PHASE_FLUSH();
#ifdef PAIRHMM_PHASE_CLOCKS
    if (lane == 0) atomicAdd(&g_phase_clocks[6], 1ull);
#endif
    return again;
}

// ---------------------------------------------------------------------------------------------
// fp32 stream pass
// ---------------------------------------------------------------------------------------------
// A wave keeps its wave-group (row constants in registers, emission table in LDS) and sweeps a run of
// haplotypes back to back.  The region's haplotypes lie in hap_stream as
//     gap hap0 gap hap1 gap ... gap            (gap: one entry)
// and the sweep treats the run as one long haplotype: at step s the lane at segment position t is at
// stream position s - t.  A gap is column 0 of the haplotype behind it and the end of the one in front
// of it: a lane arriving there hands in the finished haplotype's sum (segment-last lanes) and
// re-initialises its rows; everywhere else it computes a column.  So the pipeline fills and drains
// once per run instead of once per haplotype, and nothing per read is redone between haplotypes.
//
// Steps are taken four at a time.  A block of four steps in which no lane is at a gap runs the
// unmasked column code; a block that has some lane at a gap takes the masked path.  Haplotype codes
// come straight from global memory, kStreamAhead blocks ahead of their use (the queue of fetched codes moves up
// one block per block of steps: every block of read-ahead is two register copies per block of steps; measured on the
// 4096 x 64 batch, two blocks ahead 2.71 ms, three 2.70, four 2.74: eight steps cover the load).
#ifndef PAIRHMM_STREAM_AHEAD
#define PAIRHMM_STREAM_AHEAD 2
#endif
constexpr int kStreamAhead = PAIRHMM_STREAM_AHEAD;

struct StreamLane {
    int next_gap;  // stream position (relative to the run's first gap) of the next gap this lane meets
    int gap_idx;   // its index: gap c is column 0 of haplotype c
    float init_d;  // scale / H of the haplotype behind that gap
};

__device__ __forceinline__ uint2 load_codes(const uint16_t* p) {
    uint2 v;
    __builtin_memcpy(&v, p, sizeof v);  // 2-byte aligned
    return v;
}

template <int K, int MODE>
__device__ __forceinline__ void run_stream(const DeviceBatch& B, const int task_idx, unsigned char* slice, const int lane) {
    using T = float;
    constexpr int kCodes = lut_codes(MODE);
    static_assert(is_lut(MODE), "the stream pass reads emission terms from the table");
    PHASE_BEGIN();
    // ---- which wave-group and haplotype run? (wave-uniform) ------------------------------------------
    // The last descriptor that begins at or before this item, by a search that narrows 64-fold per round: every
    // lane probes one descriptor, so tens of thousands of descriptors take three dependent loads where halving
    // takes fifteen (a wave holds its slot through every one of them).
    int lo = 0;
    for (int n = B.num_stream_descs; n > 1;) {
        const int stride = (n + kWave - 1) / kWave;
        const bool at_or_before = lane * stride < n && B.stream_descs[lo + lane * stride].task_begin <= task_idx;
        // descriptors are sorted: the lanes that answer yes are lane 0 (the invariant) and a run behind it
        const int last = __popcll(__ballot(at_or_before)) - 1;
        lo += last * stride;
        n = n - last * stride < stride ? n - last * stride : stride;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const StreamDesc sdesc = B.stream_descs[lo];
    const GroupDesc grp = B.groups[sdesc.group_begin + (task_idx - sdesc.task_begin)];
    const int fill = grp.fill;
    const int C = sdesc.num_haps;
    LaneInfo ld{-1, 1, 0, 0};
    int R = 0;
    for (int k = 0; k < grp.num_segs; ++k) {
        const SegDesc sd = B.segs[grp.seg_begin + k];
        const int t = lane - static_cast<int>(sd.lane_begin);
        if (t >= 0 && t < static_cast<int>(sd.seg_lanes)) {
            R = sd.read_len;
            ld.read_idx = sd.read_idx;
            ld.seg_lane = t;
            ld.first_row = t * K + 1 - (static_cast<int>(sd.seg_lanes) * K - R - 1);  // pads | rows 1..R | sum row
            ld.seg_last = (t == static_cast<int>(sd.seg_lanes) - 1) ? 1 : 0;
        }
    }
    const bool used = ld.read_idx >= 0;
    // ---- the run's gaps -> LDS: {position relative to the first gap, scale / H of the haplotype behind it}
    int2* const gaps = reinterpret_cast<int2*>(slice + lut_table_bytes(K, MODE));
    const int run_base = B.hap_stream_off[sdesc.hap_begin] - 1;  // the run's first gap
    if (lane <= C + 1) {
        const int a = sdesc.hap_begin + (lane < C ? lane : C - 1);
        const int off = B.hap_stream_off[a], len = B.hap_len[a];
        int2 g;
        g.x = lane < C ? off - 1 - run_base : lane == C ? off + len - run_base : 0x7fffffff;
        g.y = __float_as_int(static_cast<float>(::ldexp(1.0, Scale<T>::kExp2) / static_cast<double>(len)));
        gaps[lane] = g;
    }
    PHASE_END(0);
    // ---- per-row constants and the emission table -------------------------------------------------------
    Rows<T, K> w;
    int out_base = 0;
    {
        int ro = 0;
        if (used) {
            ro = B.read_off[ld.read_idx];
            out_base = B.read_out_base[ld.read_idx] + sdesc.hap_local_begin;
        }
        const int rc_base = ro + 2 * (used ? ld.read_idx : 0);
        float4 c0[K], c1[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            int e = ld.first_row + r;
            e = e < 0 ? 0 : (e > R + 1 ? R + 1 : e);
            const float4* src = reinterpret_cast<const float4*>(B.row_consts + rc_base + e);
            c0[r] = src[0];
            c1[r] = src[1];
        }
        float* tbl = reinterpret_cast<float*>(slice);
        const unsigned letters[5] = {'A', 'C', 'T', 'G', 'N'};
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const bool deep_pad = ld.first_row + r < 0;  // the row below is a pad row too: hand down anything finite
            const float P = (used && !deep_pad) ? c0[r].x : 0.0f;
            const float Q = (used && !deep_pad) ? c0[r].y : 0.0f;
            const unsigned xb = (used && !deep_pad) ? __float_as_uint(c1[r].z) : kNoBase;
            w.T5[r] = !used ? 0.0f : deep_pad ? 1.0f : c0[r].z;
            w.A[r] = (used && !deep_pad) ? c0[r].w : 0.0f;
            w.B[r] = (used && !deep_pad) ? c1[r].x : 0.0f;
            w.C[r] = !used ? 0.0f : deep_pad ? 1.0f : c1[r].y;
            pin(w.A[r]); pin(w.B[r]); pin(w.C[r]); pin(w.T5[r]);
            w.Ml[r] = 0.0f; w.El[r] = 0.0f; w.Sd[r] = 0.0f;
#pragma unroll
            for (int c = 0; c < kCodes; ++c)
                tbl[lut_index(r, c, kCodes, lane)] = (xb == letters[c] || c == 4) ? P : Q;
        }
        w.Sout = 0.0f; w.Iout = 0.0f;
    }
    float sheld = 0.0f;  // see column_stream
    // pad rows of this lane: rows [0, num_pad)
    const int num_pad = used ? (1 - ld.first_row < 0 ? 0 : (1 - ld.first_row > K ? K : 1 - ld.first_row)) : 0;
    __builtin_amdgcn_wave_barrier();
    const int run_end = __builtin_amdgcn_readfirstlane(gaps[C].x);  // the gap behind the last haplotype
    PHASE_END(1);

    // ---- sweep -----------------------------------------------------------------------------------------
    const int t = ld.seg_lane;
    const uint16_t* const hp = B.hap_stream + run_base - t;  // hp[s]: table offset of this lane's position at step s
    const unsigned char* const tbl_lane = slice + lane * 4 * kLutRowsPerRead;
    auto emit = [&](const unsigned code_off, T (&prior)[K]) { lut_emit<T, K, kCodes>(tbl_lane, code_off, prior); };
    // The table position of a step is the lane's table base plus the base code's offset, one of a packed pair of
    // 16-bit offsets: added as 32-bit LDS addresses, the unpacking folds into the add (v_add_u32 with a 16-bit
    // operand select) instead of costing an instruction of its own.
    const unsigned tbl_lane_lds = static_cast<unsigned>(reinterpret_cast<size_t>((const __attribute__((address_space(3))) unsigned char*)tbl_lane));
    auto table_at = [](const unsigned lds_addr) {
        return (const unsigned char*)reinterpret_cast<const __attribute__((address_space(3))) unsigned char*>(static_cast<size_t>(lds_addr));
    };
    auto emit4 = [&](const uint2 codes, T (&prior)[4][K]) {
        lut_emit<T, K, kCodes>(table_at(tbl_lane_lds + (codes.x & 0xffffu)), 0u, prior[0]);
        lut_emit<T, K, kCodes>(table_at(tbl_lane_lds + (codes.x >> 16)), 0u, prior[1]);
        lut_emit<T, K, kCodes>(table_at(tbl_lane_lds + (codes.y & 0xffffu)), 0u, prior[2]);
        lut_emit<T, K, kCodes>(table_at(tbl_lane_lds + (codes.y >> 16)), 0u, prior[3]);
    };
    StreamLane sl{0, 0, __int_as_float(gaps[0].y)};
    // One step of a block in which some lane is at a gap.  A lane at a gap hands in the finished
    // haplotype's sum, then clears M, E and the diagonal terms (pad rows: E = scale / H) and takes
    // no insertion mass from above: the column it computes next is then exactly column 0 of the
    // next haplotype (M = I = 0, D as initialised, hand-offs to the right and downwards included),
    // whatever base code the gap entry holds.  Lanes outside the run compute on, unobserved.
    auto masked_step = [&](const int step, const unsigned code_off) {
        T Sd0 = shr1(sheld);
        T Iin = shr1(w.Iout);
        if (step - t == sl.next_gap) {
            if (sl.gap_idx > 0 && ld.seg_last != 0 && used)
                B.raw_out[out_base + sl.gap_idx - 1] = w.El[K - 1] + w.Ml[K - 1] + (K == 1 ? Sd0 : w.Sd[K - 1]);  // the sum row
#pragma unroll
            for (int r = 0; r < K; ++r) {
                w.Ml[r] = 0.0f;
                w.El[r] = r < num_pad ? sl.init_d : 0.0f;
                w.Sd[r] = 0.0f;
            }
            Sd0 = 0.0f;
            Iin = 0.0f;
            const int2 next = gaps[sl.gap_idx + 1];
            sl.next_gap = next.x;
            sl.init_d = __int_as_float(next.y);
            ++sl.gap_idx;
        }
        T prior[K];
        emit(code_off, prior);
        column_stream<K>(w, sheld, prior, Sd0, Iin);
    };

    const int num_blocks = (run_end + fill + 4) >> 2;  // steps 0 .. run_end + fill
    // codes: table offsets of block q; ahead[d]: those of block q + 1 + d, on their way from global memory
    uint2 codes = load_codes(hp);
    uint2 ahead[kStreamAhead];
#pragma unroll
    for (int d = 0; d < kStreamAhead; ++d) ahead[d] = load_codes(hp + 4 * (d + 1));
    auto advance_codes = [&](const int q) {
        codes = ahead[0];
#pragma unroll
        for (int d = 0; d + 1 < kStreamAhead; ++d) ahead[d] = ahead[d + 1];
        ahead[kStreamAhead - 1] = load_codes(hp + 4 * (q + kStreamAhead + 1));
    };
    int gap_s = 0;       // wave-uniform: index of the first gap that some lane has still to pass
    int gap_pos_s = 0;   // and its position
    int q = 0;
    while (q < num_blocks) {
        const int s0 = q << 2;
        while (s0 - fill > gap_pos_s) {  // every lane is past that gap
            ++gap_s;
            gap_pos_s = __builtin_amdgcn_readfirstlane(gaps[gap_s].x);
        }
        // blocks from here on in which no lane meets a gap: s0' + 3 < gap_pos_s
        const int steady = s0 + 3 < gap_pos_s ? (gap_pos_s - s0) >> 2 : 0;
        if (steady > 0) {
            // two-deep software pipeline: emission terms of the next block are in flight while the
            // current block is computed
            T pc[4][K];
            emit4(codes, pc);
            for (const int q_end = q + steady; q < q_end; ++q) {
                T pn[4][K];
                advance_codes(q);
                emit4(codes, pn);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const T Iin = shr1(w.Iout);
                    column_stream<K>(w, sheld, pc[u], shr1(sheld), Iin);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < K; ++r) pc[u][r] = pn[u][r];
            }
            PHASE_END(3);
        } else {
            masked_step(s0, codes.x & 0xffffu);
            masked_step(s0 + 1, codes.x >> 16);
            masked_step(s0 + 2, codes.y & 0xffffu);
            masked_step(s0 + 3, codes.y >> 16);
            advance_codes(q);
            ++q;
            PHASE_END(2);
        }
    }
    PHASE_END(3);
    PHASE_FLUSH();
}

// One work item per wave.  (Tried on MI355X and not kept: persistent waves that take their items from a device-wide
// counter.  With one item per wave a slot stands empty for 9 to 30 us between a wave and its successor, and workgroups
// are dealt to the eight XCDs in turn whatever their load, so residency is 88 % on the 4096 x 64 batch and 78 % on
// HaplotypeCaller-shaped work; persistent waves hold 100 % until the last tenth of the launch and are 8 to 14 % slower
// all the same: waves of equal priority issue oldest first, a persistent wave keeps its age, two waves fill a SIMD's
// issue slots, and the youngest of three ran at 0.4 of its share; stepping every wave through the four issue
// priorities, one step per haplotype, brought the launch back to the time of one item per wave, not below it.  Dealing
// the items to the XCDs one by one instead of four by four changed nothing either: on uniform work the XCDs in the
// middle of the package finish 15 to 35 % later than the outer ones, and a launch that keeps all of them busy to the
// end takes as long as one whose fast XCDs idle for its last tenth: the device's power budget sets the rate.)
template <int K, int MODE>
__global__ void __launch_bounds__(kWave* kWavesPerBlock) fwd_stream(const DeviceBatch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & (kWave - 1);
    const int task = blockIdx.x * kWavesPerBlock + wave;
    if (task >= B.num_tasks) return;
#ifdef PAIRHMM_PHASE_CLOCKS
    const long long life_c0 = clock64(), life_w0 = wall_clock64();
#endif
    run_stream<K, MODE>(B, task, lds + wave * B.lds_per_wave, lane);
#ifdef PAIRHMM_PHASE_CLOCKS
    if (lane == 0) {  // wave lifetime in shader-clock ticks and in ticks of the constant 100 MHz counter
        atomicAdd(&g_phase_clocks[4], (unsigned long long)(clock64() - life_c0));
        atomicAdd(&g_phase_clocks[7], (unsigned long long)(wall_clock64() - life_w0));
        atomicAdd(&g_phase_clocks[6], 1ull);
        const unsigned slot = atomicAdd(&g_wave_log_count, 1u);
        unsigned hw_id = 0, xcc_id = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        if (slot < kWaveLogEntries)
            g_wave_log[slot] = make_uint4(static_cast<unsigned>(life_w0), static_cast<unsigned>(wall_clock64()), static_cast<unsigned>(K) | (xcc_id << 8), hw_id);
    }
#endif
}

// Haplotype a -> table offsets in the stream, by one wave, which also zeroes the entries between the
// previous haplotype and this one (gaps, or the front pad) and, behind the last haplotype, the rest
// of the stream: every entry of the stream is written by exactly one wave.
__device__ __forceinline__ void expand_hap(const DeviceBatch& B, const int a, const int lane) {
    const uint8_t* src = B.hap_pool + B.hap_off[a];
    const int begin = B.hap_stream_off[a];
    const int len = B.hap_len[a];
    const int prev_end = a == 0 ? 0 : B.hap_stream_off[a - 1] + B.hap_len[a - 1];
    for (int j = prev_end + lane; j < begin; j += kWave) B.hap_stream[j] = 0;
    uint16_t* dst = B.hap_stream + begin;
    for (int j = lane; j < len; j += kWave) {
        const unsigned byte = src[j];
        dst[j] = static_cast<uint16_t>((byte == 'N' ? 4u : ((byte >> 1) & 3u)) * kLutCodeStride);
    }
    if (a == B.num_haps - 1)
        for (int j = begin + len + lane; j < B.hap_stream_len; j += kWave) B.hap_stream[j] = 0;
}

// Raw sums of the stream pass -> log10 likelihoods; pairs fp32 could not hold go on their bucket's
// list for the fp64 pair pass (one counter update per wave).
__global__ void __launch_bounds__(256) finish_stream(const DeviceBatch B) {
    const int read = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (read >= B.num_reads) return;
    const ReadInfo info = B.read_info[read];
    const int bucket = info.rows_per_lane < 0 ? -1 : B.stream_bucket_of_rows[info.rows_per_lane];
    if (bucket < 0) return;  // the stream pass did not score this read
    const int ob = B.read_out_base[read];
    const int lane = threadIdx.x & 63;
    RerunPair* const list = B.pair_lists + B.bucket_pair_begin[bucket];
    for (int a0 = 0; a0 < info.num_haps; a0 += kWave) {  // wave-uniform trip count
        const int a = a0 + lane;
        bool push = false;
        if (a < info.num_haps) {
            const float total = B.raw_out[ob + a];
            B.out[ob + a] = ::log10(static_cast<double>(total)) - Scale<float>::kExp2 * 0.30102999566398119521;
            push = !(total >= B.fp32_threshold && total <= 3.0e38f);
        }
        const unsigned long long pushing = __ballot(push);
        if (pushing != 0) {
            const int leader = __ffsll(static_cast<long long>(pushing)) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&B.pair_counts[bucket], __popcll(pushing));
            base = __shfl(base, leader);
            if (push) list[base + __popcll(pushing & ((1ull << lane) - 1ull))] = RerunPair{read, a};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fp64 pair pass
// ---------------------------------------------------------------------------------------------
// Scores the queued (read, haplotype) pairs in double, plain arithmetic form (the reference's own operations in
// linear space, so that results and the point of underflow to zero are those of a double-precision
// LoglessPairHMM with the 2^1020 initial condition).  A wave takes 64 / slot_lanes consecutive pairs of the
// list, whatever reads and haplotypes they are: every pair gets slot_lanes lanes, and each lane follows its own
// haplotype through the code stream.  Only the pairs that need it are recomputed, however they were grouped in
// the fp32 pass.  The emission terms come from a per-wave LDS table of doubles [row pair][base code][lane][2]
// read with one ds_read_b128 per two cells, one step ahead of their use.  A base code's table offset is the one
// the haplotype stream holds (64 lanes x 16 bytes = kLutCodeStride), so the codes need no conversion.
constexpr int kPairCodes = 5;  // A, C, T, G, N: the pass does not know which letters its haplotypes use
static_assert(kLutCodeStride == kWave * 16, "the pair pass reads the stream's table offsets as offsets into a table of double pairs");
constexpr int pair_table_bytes(int k) { return ((k + 1) / 2) * kPairCodes * kLutCodeStride; }

template <int K>
__device__ __forceinline__ void run_pairs(const DeviceBatch& B, const int first_pair, const int num_pairs, unsigned char* slice, const int lane) {
    using T = double;
    const int slot = lane / B.slot_lanes;
    const int t = lane - slot * B.slot_lanes;
    const bool have = slot < num_pairs;
    RerunPair pr{0, 0};
    if (have) pr = B.pair_list[first_pair + slot];
    const int ro = B.read_off[pr.read_idx];
    const int R = have ? B.read_off[pr.read_idx + 1] - ro : 0;
    const int seg_lanes = (R + 1 + K) / K;
    const bool used = have && t < seg_lanes;
    const int first_row = t * K + 1 - (seg_lanes * K - R - 1);  // pads | rows 1..R | sum row
    const bool seg_last = t == seg_lanes - 1;
    const int hap = B.read_info[pr.read_idx].hap_begin + pr.hap_local;
    const int H = used ? B.hap_len[hap] : 0;
    const uint16_t* const hp = B.hap_stream + (used ? B.hap_stream_off[hap] : kStreamFrontPad) - t;  // hp[s]: code of column s - t (0-based)

    // per-row constants straight from the quality bytes (as the one-haplotype fp64 pass does), and the emission table
    Rows<T, K> w;
    const T initD = static_cast<T>(::ldexp(1.0, Scale<T>::kExp2) / static_cast<double>(H > 0 ? H : 1));
    double* const tbl = reinterpret_cast<double*>(slice);
    const unsigned letters[kPairCodes] = {'A', 'C', 'T', 'G', 'N'};
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const int i = first_row + r;
        double p = 0, q = 0, t4 = 0, t5 = 0, t0n = 0, t1n = 0, t2n = 0, t3n = 0;
        bool pad = false, has_next = false;
        unsigned xb = kNoBase;
        const bool sum_row = used && i == R + 1;
        const bool last_row = used && i == R;
        if (sum_row) {
            p = q = 1.0;
            t5 = 1.0;
        } else if (used) {
            if (i >= 1) {
                const unsigned x = B.read_bases[ro + i - 1];
                const double e = B.err_prob[effective_qual(B, ro + i - 1, pr.read_idx)];
                xb = x;
                p = 1.0 - e;
                q = (x == 'N') ? p : e / B.mismatch_divisor;
                t4 = B.err_prob[del_byte_at(B, ro + i - 1) & 127u];
                t5 = B.err_prob[gcp_byte_at(B, ro + i - 1)];
            } else {
                pad = true;
                t4 = 1.0;
                t5 = 1.0;
                if (i < 0) t1n = 1.0;
            }
            if (i >= 0 && i < R) {
                const unsigned ins = ins_byte_at(B, ro + i) & 127u, del = del_byte_at(B, ro + i) & 127u;
                const double eg = B.err_prob[gcp_byte_at(B, ro + i)];
                has_next = true;
                t0n = B.match_to_match[ins * 128u + del];
                t1n = 1.0 - eg;
                t2n = B.err_prob[ins];
                t3n = eg;
            }
        }
        const bool real_next = has_next && i >= 1;
        w.T5[r] = t5;
        w.A[r] = real_next ? t0n : last_row ? 1.0 : 0.0;
        w.B[r] = last_row ? 1.0 : t1n;
        w.C[r] = real_next ? t2n : 0.0;
        w.D[r] = real_next ? t3n : 0.0;
        w.E[r] = t4 * t1n;
        if constexpr (K < kFirstSpillingRowsFp64) { pin(w.A[r]); pin(w.B[r]); pin(w.C[r]); pin(w.D[r]); pin(w.E[r]); pin(w.T5[r]); }
        w.Ml[r] = 0.0;
        w.El[r] = pad ? initD : 0.0;  // column 0
#pragma unroll
        for (int cd = 0; cd < kPairCodes; ++cd)  // a haplotype 'N' matches every row, a read 'N' (q = p) every column
            tbl[(((r >> 1) * kPairCodes + cd) * kWave + lane) * 2 + (r & 1)] = (xb == letters[cd] || cd == 4) ? p : q;
    }
#pragma unroll
    for (int r = 1; r < K; ++r) w.Sd[r] = w.El[r - 1] * w.E[r - 1];
    w.Sout = w.El[K - 1] * w.E[K - 1];
    w.Iout = 0.0;
    w.Sd[0] = shr1(w.Sout);
    if (t & 1) {  // this lane's first column is an odd step: column_swapping<K, 1> expects M and E in each other's place
#pragma unroll
        for (int r = 0; r < K; ++r) { const T e = w.El[r]; w.El[r] = w.Ml[r]; w.Ml[r] = e; }
    }
    __builtin_amdgcn_wave_barrier();

    // the longest haplotype of the wave sets the number of steps
    int hmax = H;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) hmax = max(hmax, __shfl_xor(hmax, d));
    const int steps = hmax + B.slot_lanes - 1;
    const unsigned char* const tbl_lane = slice + lane * 16;
    auto emit = [&](const unsigned code_off, T (&prior)[K]) {
#pragma unroll
        for (int g = 0; g < (K + 1) / 2; ++g) {
            const double2 v = *reinterpret_cast<const double2*>(tbl_lane + code_off + g * kPairCodes * kLutCodeStride);
            prior[2 * g] = v.x;
            if (2 * g + 1 < K) prior[2 * g + 1] = v.y;
        }
    };
    // a lane whose haplotype is shorter than the wave's longest stops reading the stream behind its own
    const int last_block = H + t + 8;
    uint2 codes = load_codes(hp);
    uint2 next_codes = load_codes(hp + 4);
    T pn[K];  // emission terms one step ahead of their use
    emit(codes.x & 0xffffu, pn);
    for (int s0 = 0; s0 < steps; s0 += 4) {
        const uint2 after = load_codes(hp + min(s0 + 8, last_block));
        auto step = [&](auto parity_tag, const int u, const unsigned y_next) {
            constexpr int kParity = decltype(parity_tag)::value;
            const T Sin = shr1(w.Sout);
            const T Iin = shr1(w.Iout);
            T pc[K];
#pragma unroll
            for (int r = 0; r < K; ++r) pc[r] = pn[r];
            emit(y_next, pn);
            if (static_cast<unsigned>(s0 + u - t) < static_cast<unsigned>(H)) column_swapping<T, K, kParity>(w, pc, Sin, Iin);
        };
        step(std::integral_constant<int, 0>{}, 0, codes.x >> 16);
        step(std::integral_constant<int, 1>{}, 1, codes.y & 0xffffu);
        step(std::integral_constant<int, 0>{}, 2, codes.y >> 16);
        step(std::integral_constant<int, 1>{}, 3, next_codes.x & 0xffffu);
        codes = next_codes;
        next_codes = after;
    }
    if (used && seg_last) {
        const T total = w.El[K - 1] + w.Ml[K - 1] + w.Sd[K - 1];  // the sum row
        B.out[B.read_out_base[pr.read_idx] + pr.hap_local] = ::log10(total) - Scale<T>::kExp2 * 0.30102999566398119521;
    }
}

// One wave per block: a wave's table of doubles is up to 40 KiB (16 rows per lane).
template <int K>
__global__ void __launch_bounds__(kWave) fwd_pairs(const DeviceBatch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    const int n = *B.pair_count;
    const int per_wave = kWave / B.slot_lanes;
    const int stride = gridDim.x * per_wave;
    for (int first = blockIdx.x * per_wave; first < n; first += stride) {
        run_pairs<K>(B, first, min(per_wave, n - first), lds, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// Long-read pass
// ---------------------------------------------------------------------------------------------
// A read of more than 64 x 16 - 2 bases does not fit the rows of one wave.  Its pairs are scored in double and in the
// plain arithmetic form (as the fp64 pair pass: the reference's operations in linear space with the 2^1020 initial
// condition), chunk of rows by chunk of rows: a chunk is 64 lanes x kLongRows rows, swept by one wave over the whole
// haplotype like any wave-group; what the chunk's last lane hands downwards in every column (the diagonal term and the
// insertion mass, the two values a lane takes from the lane above it) goes to a boundary row in global memory, and the
// first lane of the chunk below reads it there in the same column.  Rows are laid out as everywhere else: pad rows in
// front, the read's rows, the sum row last.  Emission terms come from comparing the bytes (no table: the rows change
// with every chunk).
//
// A workgroup of W waves shares a pair: wave w takes chunks w, w + W, ...; chunk c + 1 follows chunk c a few blocks of
// 64 columns behind, so W chunks are under way at once (with few pairs in a call the time of a pair is what counts;
// a launch with enough pairs to fill the device uses W = 1).  The chunks of a pair use W + 1 boundary rows in turn:
// chunk c writes row c mod (W + 1), which chunk c - W, the last reader of that row, has left by then (it ran on the
// same wave as c).  How far a row has been written is kept in LDS, one counter per row holding
// (chunk + 1) << 16 | columns written, raised by the writer after every block; a reader waits for its columns.  Every
// wait ends: chunk 0 waits for nothing, every other chunk only for the chunk before it.  A wait that nevertheless goes on
// for about half a second gives up and sets kStallBit in the error word (the host reports a device error) so that the
// grid always drains.
// A correctness path, not a fast one: such reads are no HaplotypeCaller input.
constexpr int kLongRows = 8;
constexpr int kLongStageBytes = kWave * 16;  // one block of boundary columns: {diagonal term, insertion mass} per column
constexpr int kLongProgressBytes = 64;       // kLongMaxWaves + 1 counters
constexpr int kLongSpinLimit = 1 << 21;      // waits of a quarter of a microsecond

// stage: this wave's kLongStageBytes of LDS; progress: the workgroup's counters; hb: the haplotype in LDS, its bytes or,
// with a table, its base codes; table: this wave's emission table in LDS, as the fp64 pair pass lays it out
// ([row pair][base code][lane][2] doubles; `codes` base codes: A C T G, and N when a haplotype of the call holds one),
// written anew for every chunk.
template <int kCodes>  // 0: compare bytes; 4 or 5: base codes of the table
__device__ __forceinline__ double run_long_chunk(const DeviceBatch& B, const RerunPair pr, const int chunk, const int num_waves,
                                                 double* boundary, double2* stage, int* progress, const uint8_t* hb, unsigned char* table,
                                                 const int lane, bool& stalled) {
    constexpr bool kTable = kCodes != 0;
    constexpr int codes = kCodes;
    using T = double;
    constexpr int K = kLongRows;
    const int ro = B.read_off[pr.read_idx];
    const int R = B.read_off[pr.read_idx + 1] - ro;
    const int hap = B.read_info[pr.read_idx].hap_begin + pr.hap_local;
    const int H = B.hap_len[hap];
    const int total_lanes = (R + 1 + K) / K;  // rows >= R + 2
    const int chunks = (total_lanes + kWave - 1) / kWave;
    const T initD = static_cast<T>(::ldexp(1.0, Scale<T>::kExp2) / static_cast<double>(H));
    const int stride = B.long_boundary_stride;
    const int g = chunk * kWave + lane;  // this lane's place among all lanes of the pair
    const bool used = g < total_lanes;
    const int first_row = g * K + 1 - (total_lanes * K - R - 1);  // pads | rows 1..R | sum row
    const bool seg_last = g == total_lanes - 1;
    // what the chunk above left per column, and where this chunk leaves its own: {diagonal term, insertion mass}
    // (read and written through the device-coherent cache level)
    const int num_rows = num_waves + 1;
    const bool from_above = chunk > 0;
    const bool to_below = chunk + 1 < chunks;
    const int row_in = (chunk + num_rows - 1) % num_rows, row_out = chunk % num_rows;
    double* const above = boundary + static_cast<size_t>(row_in) * 2 * stride;
    double* const below = boundary + static_cast<size_t>(row_out) * 2 * stride;
    const int tag_in = chunk << 16, tag_out = (chunk + 1) << 16;
    Rows<T, K> w;
    T P[K], Q[K];             // compare mode only
    unsigned xw[K / 4] = {};  // the rows' read bytes, four to a word (0: a row without a base, whose terms are equal anyway)
    const unsigned letters[kPairCodes] = {'A', 'C', 'T', 'G', 'N'};
    double* const tbl = reinterpret_cast<double*>(table);
    if constexpr (kTable) {  // the table of the chunk before has been read to the end
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const int i = first_row + r;
        double p = 0, q = 0, t4 = 0, t5 = 0, t0n = 0, t1n = 0, t2n = 0, t3n = 0;
        bool pad = false, has_next = false;
        unsigned xb = kNoBase;
        const bool sum_row = used && i == R + 1;
        const bool last_row = used && i == R;
        if (sum_row) {
            p = q = 1.0;
            t5 = 1.0;
        } else if (used) {
            if (i >= 1) {
                const unsigned x = B.read_bases[ro + i - 1];
                const double e = B.err_prob[effective_qual(B, ro + i - 1, pr.read_idx)];
                xw[r >> 2] |= x << (8 * (r & 3));
                xb = x;
                p = 1.0 - e;
                q = (x == 'N') ? p : e / B.mismatch_divisor;
                t4 = B.err_prob[del_byte_at(B, ro + i - 1) & 127u];
                t5 = B.err_prob[gcp_byte_at(B, ro + i - 1)];
            } else {
                pad = true;
                t4 = 1.0;
                t5 = 1.0;
                if (i < 0) t1n = 1.0;
            }
            if (i >= 0 && i < R) {
                const unsigned ins = ins_byte_at(B, ro + i) & 127u, del = del_byte_at(B, ro + i) & 127u;
                const double eg = B.err_prob[gcp_byte_at(B, ro + i)];
                has_next = true;
                t0n = B.match_to_match[ins * 128u + del];
                t1n = 1.0 - eg;
                t2n = B.err_prob[ins];
                t3n = eg;
            }
        }
        const bool real_next = has_next && i >= 1;
        if constexpr (kTable) {
#pragma unroll
            for (int cd = 0; cd < kPairCodes; ++cd)  // a haplotype 'N' matches every row, a read 'N' (q = p) every column
                if (cd < codes) tbl[(((r >> 1) * codes + cd) * kWave + lane) * 2 + (r & 1)] = (xb == letters[cd] || cd == 4) ? p : q;
        } else {
            P[r] = p;
            Q[r] = q;
        }
        w.T5[r] = t5;
        w.A[r] = real_next ? t0n : last_row ? 1.0 : 0.0;
        w.B[r] = last_row ? 1.0 : t1n;
        w.C[r] = real_next ? t2n : 0.0;
        w.D[r] = real_next ? t3n : 0.0;
        w.E[r] = t4 * t1n;
        w.Ml[r] = 0.0;
        w.El[r] = pad ? initD : 0.0;  // column 0
    }
#pragma unroll
    for (int r = 1; r < K; ++r) w.Sd[r] = w.El[r - 1] * w.E[r - 1];
    w.Sout = w.El[K - 1] * w.E[K - 1];
    w.Iout = 0.0;
    w.Sd[0] = shr1(w.Sout);  // the first lane of a chunk below the first: the row above it is a read row, whose column 0 holds nothing
    const int steps = H + kWave - 1;
    const unsigned char* const tbl_lane = table + lane * 16;
    auto emit = [&](const unsigned y, T (&prior)[K]) {  // y: a haplotype byte, or with a table its base code
        if constexpr (kTable) {
#pragma unroll
            for (int g2 = 0; g2 < K / 2; ++g2) {
                const double2 v = *reinterpret_cast<const double2*>(tbl_lane + y * kLutCodeStride + g2 * codes * kLutCodeStride);
                prior[2 * g2] = v.x;
                prior[2 * g2 + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < K; ++r) prior[r] = (((xw[r >> 2] >> (8 * (r & 3))) & 0xffu) == y || y == 'N') ? P[r] : Q[r];  // a read 'N' has Q = P
        }
    };
    if constexpr (kTable) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // Blocks of 64 steps.  Lane 0 of a chunk below the first takes its hand-off from the boundary row: the 64
    // columns of a block are fetched by the 64 lanes one block ahead of their use and passed through LDS.
    auto fetch = [&](const int s0, double2& v) {
        v = double2{0.0, 0.0};
        if (!from_above || s0 >= H) return;  // wave-uniform
        const int need = tag_in + min(s0 + kWave, H);
        for (int spins = 0; !stalled && __hip_atomic_load(progress + row_in, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++spins) {
            if (spins > kLongSpinLimit) stalled = true;  // once per wave: no later wait of this wave spins again
            __builtin_amdgcn_s_sleep(8);
        }
        const int j = s0 + lane;
        if (j < H) {
            v.x = __hip_atomic_load(above + 2 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v.y = __hip_atomic_load(above + 2 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    double2 ahead;
    fetch(0, ahead);
    unsigned y_next = hb[min(max(-lane, 0), H - 1)];
    for (int s0 = 0; s0 < steps; s0 += kWave) {
        // this wave's lanes only: LDS serves a wave's reads and writes in the order it issued them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        stage[lane] = ahead;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fetch(s0 + kWave, ahead);
        const int block_end = min(s0 + kWave, steps);
        for (int s = s0; s < block_end; ++s) {
            T Sin = shr1(w.Sout);
            T Iin = shr1(w.Iout);
            const int j = s - lane;  // this lane's column (0-based haplotype position)
            const bool in_hap = used && j >= 0 && j < H;
            const unsigned y = y_next;
            y_next = hb[min(max(j + 1, 0), H - 1)];
            const double2 handed = stage[s - s0];  // one address for the wave
            if (from_above && lane == 0) {
                Sin = handed.x;
                Iin = handed.y;
            }
            if (in_hap) {
                T prior[K];
                emit(y, prior);
                column<T, K>(w, prior, Sin, Iin);
                if (lane == kWave - 1 && to_below) {
                    __hip_atomic_store(below + 2 * j, w.Sout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(below + 2 * j + 1, w.Iout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (to_below) {  // the last lane has finished the columns up to block_end - 64: its stores first, then the count
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == kWave - 1 && !(B.long_withhold_progress && chunk == 0))  // the lane whose stores it vouches for
                __hip_atomic_store(progress + row_out, tag_out + min(max(block_end - (kWave - 1), 0), H), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    T total = 0.0;
    if (used && seg_last) total = w.El[K - 1] + w.Ml[K - 1] + w.Sd[K - 1];  // the sum row
    return total;
}

// Workgroups of 1 to kLongMaxWaves waves, a pair at a time; a workgroup owns waves + 1 boundary rows.
// Dynamic LDS: [waves x kLongStageBytes][kLongProgressBytes][with a table: waves x long_table_bytes][the longest
// haplotype's bytes, rounded up to 16].  kTable: emission terms from a per-wave table instead of comparing bytes, a third
// fewer vector instructions per step; for launches of one or two waves per workgroup (the table is 16 KiB a wave, 20
// with a column for 'N') whose haplotypes hold A, C, G, T and N only.
constexpr int long_table_bytes(int codes) { return (kLongRows / 2) * codes * kLutCodeStride; }
constexpr int kLongTableMaxWaves = 2;

template <int kCodes>
__global__ void __launch_bounds__(kWave* kLongMaxWaves) fwd_long(const DeviceBatch B) {
    constexpr bool kTable = kCodes != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int num_waves = static_cast<int>(blockDim.x) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int lane = threadIdx.x & (kWave - 1);
    double* const boundary = B.long_boundary + static_cast<size_t>(blockIdx.x) * (num_waves + 1) * 2 * B.long_boundary_stride;
    double2* const stage = reinterpret_cast<double2*>(lds + wave * kLongStageBytes);
    int* const progress = reinterpret_cast<int*>(lds + num_waves * kLongStageBytes);
    unsigned char* const tables = lds + num_waves * kLongStageBytes + kLongProgressBytes;
    constexpr int table_bytes = kTable ? long_table_bytes(kCodes) : 0;
    uint8_t* const hb = tables + num_waves * table_bytes;
    bool stalled = false;
    for (int p = blockIdx.x; p < B.num_long_pairs; p += gridDim.x) {
        const RerunPair pr = B.long_pairs[p];
        const int hap = B.read_info[pr.read_idx].hap_begin + pr.hap_local;
        const int H = B.hap_len[hap];
        const int R = B.read_off[pr.read_idx + 1] - B.read_off[pr.read_idx];
        const int total_lanes = (R + 1 + kLongRows) / kLongRows;
        const int chunks = (total_lanes + kWave - 1) / kWave;
        {   // the haplotype into LDS, four bytes a thread (haplotypes start at multiples of 16 in the pool and are padded to one)
            const uint8_t* const src = B.hap_pool + B.hap_off[hap];
            for (int j = threadIdx.x * 4; j < H; j += blockDim.x * 4) {
                uint32_t four = *reinterpret_cast<const uint32_t*>(src + j);
                if constexpr (kTable) {  // bytes -> base codes, the order of the table's columns: A C T G N
                    uint32_t codes = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t byte = (four >> (8 * k)) & 0xffu;
                        codes |= (byte == 'N' ? 4u : ((byte >> 1) & 3u)) << (8 * k);
                    }
                    four = codes;
                }
                *reinterpret_cast<uint32_t*>(hb + j) = four;
            }
        }
        if (threadIdx.x <= static_cast<unsigned>(num_waves)) progress[threadIdx.x] = 0;
        __syncthreads();
        for (int chunk = wave; chunk < chunks; chunk += num_waves) {
            const double total = run_long_chunk<kCodes>(B, pr, chunk, num_waves, boundary, stage, progress, hb, tables + wave * table_bytes, lane, stalled);
            if (chunk == chunks - 1 && lane == ((total_lanes - 1) & (kWave - 1)))
                B.out[B.read_out_base[pr.read_idx] + pr.hap_local] = ::log10(total) - Scale<double>::kExp2 * 0.30102999566398119521;
        }
        __syncthreads();  // the next pair overwrites the haplotype and the counters
    }
    if (stalled && lane == 0) report_input_error(B, kStallBit, -1);
}

// ---------------------------------------------------------------------------------------------
// The steps right behind the path (genotyper/ReadLikelihoods.java), on the results still on the device.
// out is read-major: a read's likelihoods against its region's alleles are consecutive.  One thread per read.
// ---------------------------------------------------------------------------------------------

// ReadLikelihoods.normalizeLikelihoodsPerRead, :375-403 (best alternative allele: :454-484 with
// canBeReference = false; there is no <NON_REF> allele on this path).
__global__ void __launch_bounds__(256) normalize_reads(const DeviceBatch B, const int32_t* reference_allele_index, const int best_to_zero,
                                                       const double cap) {
    const int read = blockIdx.x * blockDim.x + threadIdx.x;
    if (read >= B.num_reads) return;
    const ReadInfo info = B.read_info[read];
    const int A = info.num_haps;
    if (A == 0 || (A == 1 && !best_to_zero)) return;  // :361-364
    const int ref = reference_allele_index ? reference_allele_index[info.region] : -1;
    double* const v = B.out + B.read_out_base[read];
    double best_alt = -__builtin_inf();
    if (!(A == 1 && ref == 0)) {
        int first = ref != 0 ? 0 : 1;
        best_alt = v[first];
        for (int a = first + 1; a < A; ++a)
            if (a != ref && v[a] > best_alt) best_alt = v[a];
    }
    const double worst_cap = best_alt + cap;
    const double ref_lk = ref == -1 ? -__builtin_inf() : v[ref];
    const double best_abs = best_alt > ref_lk ? best_alt : ref_lk;
    for (int a = 0; a < A; ++a) {
        double x = v[a];
        if (best_to_zero) {
            if (best_abs == -__builtin_inf()) x = 0.0;
            else if (worst_cap != -__builtin_inf()) x = (x < worst_cap ? worst_cap : x) - best_abs;
            else x -= best_abs;
        } else if (x < worst_cap) {
            x = worst_cap;
        }
        v[a] = x;
    }
}

// ReadLikelihoods.readIsPoorlyModelled, :1081-1093: 1 for the reads filterPoorlyModeledReads removes.
__global__ void __launch_bounds__(256) flag_poorly_modelled_reads(const DeviceBatch B, const double maximum_error_per_base, uint8_t* flags) {
    const int read = blockIdx.x * blockDim.x + threadIdx.x;
    if (read >= B.num_reads) return;
    const ReadInfo info = B.read_info[read];
    const int len = B.read_off[read + 1] - B.read_off[read];
    const double max_errors = ::fmin(2.0, ::ceil(len * maximum_error_per_base));
    const double threshold = max_errors * -4.0;
    const double* const v = B.out + B.read_out_base[read];
    uint8_t poorly = 1;
    for (int a = 0; a < info.num_haps; ++a)
        if (v[a] >= threshold) poorly = 0;
    flags[read] = poorly;
}

// ReadLikelihoods.marginalLikelihoods, :911-939, keeping every read.  old_to_new[global haplotype]: new
// allele index within the region or -1; new_base[read]: where the read's new row starts in new_out;
// new_count[region]: new alleles of the region.
__global__ void __launch_bounds__(256) marginalize_reads(const DeviceBatch B, const int32_t* old_to_new, const int64_t* new_base,
                                                         const int32_t* new_count, double* new_out) {
    const int read = blockIdx.x * blockDim.x + threadIdx.x;
    if (read >= B.num_reads) return;
    const ReadInfo info = B.read_info[read];
    double* const dst = new_out + new_base[read];
    const int n_new = new_count[info.region];
    for (int n = 0; n < n_new; ++n) dst[n] = -__builtin_inf();
    const double* const v = B.out + B.read_out_base[read];
    for (int a = 0; a < info.num_haps; ++a) {
        const int n = old_to_new[info.hap_begin + a];
        if (n >= 0 && v[a] > dst[n]) dst[n] = v[a];
    }
}

// One task per wave, tasks taken in launch order.
template <typename T, int K, int MODE>
__global__ void __launch_bounds__(kWave* kWavesPerBlock) fwd_direct(const DeviceBatch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & (kWave - 1);
    const int task = blockIdx.x * kWavesPerBlock + wave;
    if (task >= B.num_tasks) return;
    if (B.abort_word != nullptr && (*B.abort_word & kAbortBit) != 0) return;
    const bool again = run_task<T, K, MODE>(B, task, lds + wave * B.lds_per_wave, lane, std::is_same<T, float>::value);
    // Up to kInlineRerunRows rows per lane (the values region-sized calls take) the wave redoes its work item in fp64
    // itself instead of leaving it to a launch of fwd_list: on a call whose time is the chain of its launches that
    // launch, empty almost always, is 4 us.  The item still goes on the list, which the host reads for its statistics.
    if constexpr (std::is_same<T, float>::value && K <= kInlineRerunRows) {
        if (again) {
            __builtin_amdgcn_wave_barrier();
            constexpr int kMode64 = (MODE == kEmitLut5 || MODE == kEmitCompareN) ? kEmitCompareN : kEmitCompare;
            run_task<double, K, kMode64>(B, task, lds + wave * B.lds_per_wave, lane, false);
        }
    }
}

// Tasks taken from the device-side rerun list; every wave strides over the list, so the grid
// size does not depend on the (device-only) list length and each wave leaves when the list ends.
template <typename T, int K, int MODE>
__global__ void __launch_bounds__(kWave* kWavesPerBlock) fwd_list(const DeviceBatch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & (kWave - 1);
    if (B.abort_word != nullptr && (*B.abort_word & kAbortBit) != 0) return;
    int n = *B.rerun_count;
    if (n > B.num_tasks) n = B.num_tasks;
    const int stride = gridDim.x * kWavesPerBlock;
    for (int it = blockIdx.x * kWavesPerBlock + wave; it < n; it += stride) {
        const int task = __builtin_amdgcn_readfirstlane(B.rerun_list[it]);
        run_task<T, K, MODE>(B, task, lds + wave * B.lds_per_wave, lane, false);
        __builtin_amdgcn_wave_barrier();
    }
}

// Read r, by one wave: the checks the reference makes while expanding a read's qualities
// (PairHMMModel.java:136-138: negative ins/del/GCP byte -> IllegalArgumentException;
// QualityUtils.java:173: base quality 255 indexes past the 255-entry cache) and, when `expand` is
// set, the RowConst entries of the fp32 pass, one lane per entry.  The arithmetic is done in double
// and rounded once, so an entry holds exactly what the forward kernel would have derived from the
// quality bytes (PairHMMModel.java:109-143 for the transitions, QualityUtils.java:169-176 for the
// priors).
__device__ __forceinline__ void prepare_read(const DeviceBatch& B, const int read, const int lane, const bool expand) {
    const int ro = B.read_off[read];
    const int R = B.read_off[read + 1] - ro;
    RowConst* dst = B.row_consts + ro + 2 * static_cast<int64_t>(read);
    int bad = 0;
    for (int i = lane; i <= R + 1; i += kWave) {
        const RowConst c = make_row_const(B, read, ro, R, i, bad);
        if (expand) dst[i] = c;
    }
    if (bad) report_input_error(B, bad, read);
}

// Everything a pass needs before its forward kernels, in one launch.  Blocks [0, read_blocks) take
// four reads each (prepare_read), blocks [read_blocks, read_blocks + hap_blocks) four haplotypes each
// (expand_hap), the rest the pairs of zero-length reads: the last matrix row is row 0, whose M and I
// are all zero -> log10(0).
__global__ void __launch_bounds__(256) prepare_batch(const DeviceBatch B, const int read_blocks, const int hap_blocks, const int expand) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int block = blockIdx.x;
    if (block < read_blocks) {
        const int read = block * 4 + wave;
        if (read < B.num_reads) prepare_read(B, read, lane, expand != 0);
        return;
    }
    block -= read_blocks;
    if (block < hap_blocks) {
        const int a = block * 4 + wave;
        if (a < B.num_haps) expand_hap(B, a, lane);
        return;
    }
    block -= hap_blocks;
    const int k = block * 256 + threadIdx.x;
    if (k < B.num_neg_inf) B.out[B.neg_inf_idx[k]] = -__builtin_inf();
}

template <typename T, int K, int MODE>
hipError_t launch_m(const DeviceBatch& B, bool list_mode, int blocks, hipStream_t stream) {
    const size_t lds = static_cast<size_t>(B.lds_per_wave) * kWavesPerBlock;
    const dim3 grid(blocks), block(kWave * kWavesPerBlock);
    if (list_mode) {
        // only the fp64 pass walks a rerun list
        if constexpr (std::is_same<T, double>::value) hipLaunchKernelGGL((fwd_list<T, K, MODE>), grid, block, lds, stream, B);
        else return hipErrorInvalidValue;
    } else {
        hipLaunchKernelGGL((fwd_direct<T, K, MODE>), grid, block, lds, stream, B);
    }
    return hipGetLastError();
}

template <typename T, int K>
hipError_t launch_k(const DeviceBatch& B, int mode, bool list_mode, int blocks, hipStream_t stream) {
    if constexpr (std::is_same<T, float>::value) {
        if (mode == kEmitLut4) return launch_m<T, K, kEmitLut4>(B, list_mode, blocks, stream);
        if (mode == kEmitLut5) return launch_m<T, K, kEmitLut5>(B, list_mode, blocks, stream);
    }
    if (mode == kEmitCompare) return launch_m<T, K, kEmitCompare>(B, list_mode, blocks, stream);
    if (mode == kEmitCompareN) return launch_m<T, K, kEmitCompareN>(B, list_mode, blocks, stream);
    return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_t(const DeviceBatch& B, int K, int mode, bool list_mode, int blocks, hipStream_t stream) {
    switch (K) {
        case 1: return launch_k<T, 1>(B, mode, list_mode, blocks, stream);
        case 2: return launch_k<T, 2>(B, mode, list_mode, blocks, stream);
        case 3: return launch_k<T, 3>(B, mode, list_mode, blocks, stream);
        case 4: return launch_k<T, 4>(B, mode, list_mode, blocks, stream);
        case 5: return launch_k<T, 5>(B, mode, list_mode, blocks, stream);
        case 6: return launch_k<T, 6>(B, mode, list_mode, blocks, stream);
        case 7: return launch_k<T, 7>(B, mode, list_mode, blocks, stream);
        case 8: return launch_k<T, 8>(B, mode, list_mode, blocks, stream);
        case 9: return launch_k<T, 9>(B, mode, list_mode, blocks, stream);
        case 10: return launch_k<T, 10>(B, mode, list_mode, blocks, stream);
        case 11: return launch_k<T, 11>(B, mode, list_mode, blocks, stream);
        case 12: return launch_k<T, 12>(B, mode, list_mode, blocks, stream);
        case 14: return launch_k<T, 14>(B, mode, list_mode, blocks, stream);
        case 16: return launch_k<T, 16>(B, mode, list_mode, blocks, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int K>
hipError_t launch_stream_k(const DeviceBatch& B, int mode, int blocks, hipStream_t stream) {
    const size_t lds = static_cast<size_t>(B.lds_per_wave) * kWavesPerBlock;
    const dim3 grid(blocks), block(kWave * kWavesPerBlock);
    if (mode == kEmitLut4) hipLaunchKernelGGL((fwd_stream<K, kEmitLut4>), grid, block, lds, stream, B);
    else if (mode == kEmitLut5) hipLaunchKernelGGL((fwd_stream<K, kEmitLut5>), grid, block, lds, stream, B);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace

// Bytes of LDS one wave of the stream pass needs: the emission table and the run's gap list.
int stream_lds_bytes_per_wave(int rows_per_lane, int emit_mode) {
    return (lut_table_bytes(rows_per_lane, emit_mode) + (kMaxStreamHaps + 2) * 8 + 15) & ~15;
}

// Enqueue the fp32 stream pass for one bucket (B.stream_descs / num_stream_descs / num_tasks set for it).
hipError_t launch_stream(const DeviceBatch& B, int rows_per_lane, int emit_mode, hipStream_t stream) {
    if (B.num_tasks <= 0) return hipSuccess;
    const int blocks = (B.num_tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    switch (rows_per_lane) {
        case 1: return launch_stream_k<1>(B, emit_mode, blocks, stream);
        case 2: return launch_stream_k<2>(B, emit_mode, blocks, stream);
        case 3: return launch_stream_k<3>(B, emit_mode, blocks, stream);
        case 4: return launch_stream_k<4>(B, emit_mode, blocks, stream);
        case 5: return launch_stream_k<5>(B, emit_mode, blocks, stream);
        case 6: return launch_stream_k<6>(B, emit_mode, blocks, stream);
        case 7: return launch_stream_k<7>(B, emit_mode, blocks, stream);
        case 8: return launch_stream_k<8>(B, emit_mode, blocks, stream);
        case 9: return launch_stream_k<9>(B, emit_mode, blocks, stream);
        case 10: return launch_stream_k<10>(B, emit_mode, blocks, stream);
        case 11: return launch_stream_k<11>(B, emit_mode, blocks, stream);
        case 12: return launch_stream_k<12>(B, emit_mode, blocks, stream);
        case 14: return launch_stream_k<14>(B, emit_mode, blocks, stream);
        case 16: return launch_stream_k<16>(B, emit_mode, blocks, stream);
        default: return hipErrorInvalidValue;
    }
}

// Enqueue the fp64 pair pass of one bucket (B.pair_count / pair_list / slot_lanes set for it); waves: one-wave blocks.
hipError_t launch_pairs(const DeviceBatch& B, int rows_per_lane, int waves, hipStream_t stream) {
    if (waves <= 0) return hipSuccess;
    const dim3 grid(waves), block(kWave);
    const size_t lds = static_cast<size_t>(pair_table_bytes(rows_per_lane));
    switch (rows_per_lane) {
#define PAIRHMM_PAIRS_CASE(k) case k: hipLaunchKernelGGL((fwd_pairs<k>), grid, block, lds, stream, B); break;
        PAIRHMM_PAIRS_CASE(1) PAIRHMM_PAIRS_CASE(2) PAIRHMM_PAIRS_CASE(3) PAIRHMM_PAIRS_CASE(4) PAIRHMM_PAIRS_CASE(5)
        PAIRHMM_PAIRS_CASE(6) PAIRHMM_PAIRS_CASE(7) PAIRHMM_PAIRS_CASE(8) PAIRHMM_PAIRS_CASE(9) PAIRHMM_PAIRS_CASE(10)
        PAIRHMM_PAIRS_CASE(11) PAIRHMM_PAIRS_CASE(12) PAIRHMM_PAIRS_CASE(14) PAIRHMM_PAIRS_CASE(16)
#undef PAIRHMM_PAIRS_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Enqueue the long-read pass (B.long_pairs / num_long_pairs / long_boundary / long_boundary_stride set); waves: one-wave blocks.
bool long_table_fits(int waves_per_block) { return waves_per_block <= kLongTableMaxWaves; }

// table_codes: 0 = no table, 4 or 5 = a table with that many base codes
int long_lds_bytes(int waves_per_block, int max_hap_len, int table_codes) {
    return waves_per_block * kLongStageBytes + kLongProgressBytes + (table_codes ? waves_per_block * long_table_bytes(table_codes) : 0) + ((max_hap_len + 15) & ~15);
}

hipError_t launch_long(const DeviceBatch& B, int blocks, int waves_per_block, int max_hap_len, hipStream_t stream) {
    if (blocks <= 0 || B.num_long_pairs <= 0) return hipSuccess;
    const int codes = B.long_table_codes;
    if (waves_per_block < 1 || waves_per_block > kLongMaxWaves || (codes != 0 && codes != 4 && codes != 5) || (codes && !long_table_fits(waves_per_block)))
        return hipErrorInvalidValue;
    const size_t lds = static_cast<size_t>(long_lds_bytes(waves_per_block, max_hap_len, codes));
    const dim3 grid(blocks), block(kWave * waves_per_block);
    if (codes == 5) hipLaunchKernelGGL(fwd_long<5>, grid, block, lds, stream, B);
    else if (codes == 4) hipLaunchKernelGGL(fwd_long<4>, grid, block, lds, stream, B);
    else hipLaunchKernelGGL(fwd_long<0>, grid, block, lds, stream, B);
    return hipGetLastError();
}

hipError_t launch_normalize_reads(const DeviceBatch& B, const int32_t* reference_allele_index, int best_to_zero, double cap, hipStream_t stream) {
    if (B.num_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(normalize_reads, dim3((B.num_reads + 255) / 256), dim3(256), 0, stream, B, reference_allele_index, best_to_zero, cap);
    return hipGetLastError();
}

hipError_t launch_flag_poorly_modelled_reads(const DeviceBatch& B, double maximum_error_per_base, uint8_t* flags, hipStream_t stream) {
    if (B.num_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(flag_poorly_modelled_reads, dim3((B.num_reads + 255) / 256), dim3(256), 0, stream, B, maximum_error_per_base, flags);
    return hipGetLastError();
}

hipError_t launch_marginalize_reads(const DeviceBatch& B, const int32_t* old_to_new, const int64_t* new_base, const int32_t* new_count,
                                    double* new_out, hipStream_t stream) {
    if (B.num_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(marginalize_reads, dim3((B.num_reads + 255) / 256), dim3(256), 0, stream, B, old_to_new, new_base, new_count, new_out);
    return hipGetLastError();
}

hipError_t launch_finish_stream(const DeviceBatch& B, hipStream_t stream) {
    if (B.num_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(finish_stream, dim3((B.num_reads + 3) / 4), dim3(256), 0, stream, B);
    return hipGetLastError();
}

// Bytes of LDS one wave needs for a launch with these parameters.
int lds_bytes_per_wave(int rows_per_lane, int emit_mode, int max_hap_len) {
    const int entries = kHapFrontPad + ((max_hap_len + 3) & ~3) + kHapBackPad;
    const bool lut = is_lut(emit_mode);
    const int table = direct_table_bytes(rows_per_lane, emit_mode);
    return (table + entries * (lut ? 2 : 1) + 15) & ~15;
}

// Enqueue the forward kernel for one bucket of tasks.  list_mode: take task indices from the
// bucket's rerun list (fp64 pass) instead of 0..num_tasks-1.  The fp64 pass always compares bytes.
hipError_t launch_forward(const DeviceBatch& B, int rows_per_lane, int emit_mode, bool fp64, bool list_mode,
                          int blocks, hipStream_t stream) {
    if (B.num_tasks <= 0 || blocks <= 0) return hipSuccess;
    return fp64 ? launch_t<double>(B, rows_per_lane, emit_mode, list_mode, blocks, stream)
                : launch_t<float>(B, rows_per_lane, emit_mode, list_mode, blocks, stream);
}

// Copies a small upload from pinned host memory (which the device can read) into the arena.  On the
// stream the kernels run on: a copy through the DMA engine costs its own latency plus a hand-over between
// engines that is several times the copy itself for a region-sized call.
__global__ void __launch_bounds__(256) upload_from_host(const uint4* __restrict__ src, uint4* __restrict__ dst, const int n16) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += gridDim.x * blockDim.x) dst[k] = src[k];
}

hipError_t launch_upload(const void* pinned_host, void* device, size_t bytes, hipStream_t s) {
    const int n16 = static_cast<int>((bytes + 15) / 16);
    if (n16 <= 0) return hipSuccess;
    hipLaunchKernelGGL(upload_from_host, dim3(std::min((n16 + 255) / 256, 512)), dim3(256), 0, s,
                       static_cast<const uint4*>(pinned_host), static_cast<uint4*>(device), n16);
    return hipGetLastError();
}

// The way back for a small result: counters and likelihoods from the device into pinned host memory, and the first
// clear16 16-byte words of the source (the counters) zeroed for the next pass, in one launch.
__global__ void __launch_bounds__(256) download_to_host(uint4* src, uint4* __restrict__ dst, const int n16, const int clear16) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += gridDim.x * blockDim.x) {
        dst[k] = src[k];
        if (k < clear16) src[k] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// The same in one block that then announces itself: every thread makes its stores visible to the host, the block
// meets, and one thread stores `seq` into a word of pinned host memory the caller is polling.  The host sees the
// result a fraction of a microsecond after the last store instead of after the end-of-kernel signal has made its
// way through the runtime.
__global__ void __launch_bounds__(1024) download_and_signal(uint4* src, uint4* __restrict__ dst, const int n16, const int clear16,
                                                            unsigned* flag, const unsigned seq) {
    for (int k = threadIdx.x; k < n16; k += blockDim.x) {
        dst[k] = src[k];
        if (k < clear16) src[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_download_and_signal(void* device, void* pinned_host, size_t bytes, size_t clear_bytes, unsigned* pinned_flag,
                                      unsigned seq, hipStream_t s) {
    const int n16 = static_cast<int>((bytes + 15) / 16);
    hipLaunchKernelGGL(download_and_signal, dim3(1), dim3(n16 <= 256 ? 256 : n16 <= 512 ? 512 : 1024), 0, s, static_cast<uint4*>(device),
                       static_cast<uint4*>(pinned_host), n16, static_cast<int>(clear_bytes / 16), pinned_flag, seq);
    return hipGetLastError();
}

hipError_t launch_download(void* device, void* pinned_host, size_t bytes, size_t clear_bytes, hipStream_t s) {
    const int n16 = static_cast<int>((bytes + 15) / 16);
    if (n16 <= 0) return hipSuccess;
    hipLaunchKernelGGL(download_to_host, dim3(std::min((n16 + 255) / 256, 512)), dim3(256), 0, s,
                       static_cast<uint4*>(device), static_cast<uint4*>(pinned_host), n16, static_cast<int>(clear_bytes / 16));
    return hipGetLastError();
}

// expand_rows: write the RowConst entries (the fp32 passes read them); stream: fill the haplotype stream.
hipError_t launch_prepare(const DeviceBatch& B, bool expand_rows, bool stream, hipStream_t s) {
    const int read_blocks = (B.num_reads + 3) / 4;
    const int hap_blocks = stream ? (B.num_haps + 3) / 4 : 0;
    const int neg_blocks = (B.num_neg_inf + 255) / 256;
    const int blocks = read_blocks + hap_blocks + neg_blocks;
    if (blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(prepare_batch, dim3(blocks), dim3(256), 0, s, B, read_blocks, hap_blocks, expand_rows ? 1 : 0);
    return hipGetLastError();
}

}  // namespace pairhmm_hip

#ifdef PAIRHMM_PHASE_CLOCKS
extern "C" int gatk_pairhmm_debug_phase_clocks(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(pairhmm_hip::g_phase_clocks), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long zero[8] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(pairhmm_hip::g_phase_clocks), zero, sizeof(zero));
    }
    return e == hipSuccess ? 0 : -1;
}
// Copies out up to `capacity` wave-log entries (4 words each) and empties the log; returns how many there were.
extern "C" int gatk_pairhmm_debug_wave_log(unsigned* out, unsigned capacity) {
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(pairhmm_hip::g_wave_log_count), sizeof n) != hipSuccess) return -1;
    n = n < pairhmm_hip::kWaveLogEntries ? n : pairhmm_hip::kWaveLogEntries;
    n = n < capacity ? n : capacity;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(pairhmm_hip::g_wave_log), static_cast<size_t>(n) * 16) != hipSuccess) return -1;
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(pairhmm_hip::g_wave_log_count), &zero, sizeof zero) != hipSuccess) return -1;
    return static_cast<int>(n);
}
#endif
