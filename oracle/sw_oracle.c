/* CPU restatement of the reference's Smith-Waterman pairwise aligner (TEST INFRASTRUCTURE ONLY:
 * nothing in the product links or calls this).
 *
 * Follows smithwaterman/SWPairwiseAlignment.java of the reference:
 *   calculateMatrix  :223-332   affine-gap matrix with the running best-gap arrays and the back-track matrix
 *   calculateCigar   :350-470   choice of the end cell per overhang strategy, back-track, overhang handling
 *   wd               :484-486   match / mismatch score
 * and sam/AlignmentUtils.java:698-727 (consolidateCigar) for the final merge of equal neighbours.
 * Parity unpinned: the reference tree holds no test of the aligner (only a timing benchmark,
 * smithwaterman/SmithWatermanBenchmark.java); the tests of this file work cases out by hand.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { SW_OK = 0, SW_ILLEGAL_ARGUMENT = 1, SW_CIGAR_TOO_LONG = 5 };
enum { SW_SOFTCLIP = 0, SW_INDEL = 1, SW_LEADING_INDEL = 2, SW_IGNORE = 3 };  /* OVERHANG_STRATEGY, :69-93 */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_S = 3 };                                /* State -> CigarOperator, :472-481 */

/* Fills sw[(n+1) x (m+1)] and btrack (same shape), row-major.  :223-332 */
static void calculateMatrix(const uint8_t *reference, int refLength, const uint8_t *alternate, int altLength,
                            int w_match, int w_mismatch, int w_open, int w_extend, int strategy, int *sw, int *btrack) {
    const int nrow = refLength + 1, ncol = altLength + 1;
    const int MATRIX_MIN_CUTOFF = (int)-1e8;          /* cutoff is false, :231-233 */
    const int lowInitValue = INT32_MIN / 2;            /* :235 */
    int *best_gap_v = (int *)malloc(sizeof(int) * (ncol + 1));
    int *gap_size_v = (int *)calloc(ncol + 1, sizeof(int));
    int *best_gap_h = (int *)malloc(sizeof(int) * (nrow + 1));
    int *gap_size_h = (int *)calloc(nrow + 1, sizeof(int));
    for (int k = 0; k <= ncol; k++) best_gap_v[k] = lowInitValue;
    for (int k = 0; k <= nrow; k++) best_gap_h[k] = lowInitValue;
    memset(sw, 0, sizeof(int) * (size_t)nrow * ncol);
    memset(btrack, 0, sizeof(int) * (size_t)nrow * ncol);
    if (strategy == SW_INDEL || strategy == SW_LEADING_INDEL) {  /* :244-261 */
        int cur = w_open;
        sw[1] = w_open;
        for (int j = 2; j < ncol; j++) { cur += w_extend; sw[j] = cur; }
        sw[(size_t)1 * ncol] = w_open;
        cur = w_open;
        for (int i = 2; i < nrow; i++) { cur += w_extend; sw[(size_t)i * ncol] = cur; }
    }
    for (int i = 1; i < nrow; i++) {                   /* :263-331 */
        const uint8_t a_base = reference[i - 1];
        const int *lastRow = sw + (size_t)(i - 1) * ncol;
        int *curRow = sw + (size_t)i * ncol;
        int *curBackTrackRow = btrack + (size_t)i * ncol;
        for (int j = 1; j < ncol; j++) {
            const uint8_t b_base = alternate[j - 1];
            const int step_diag = lastRow[j - 1] + (a_base == b_base ? w_match : w_mismatch);
            int prev_gap = lastRow[j] + w_open;
            best_gap_v[j] += w_extend;
            if (prev_gap > best_gap_v[j]) { best_gap_v[j] = prev_gap; gap_size_v[j] = 1; }
            else gap_size_v[j]++;
            const int step_down = best_gap_v[j];
            const int kd = gap_size_v[j];
            prev_gap = curRow[j - 1] + w_open;
            best_gap_h[i] += w_extend;
            if (prev_gap > best_gap_h[i]) { best_gap_h[i] = prev_gap; gap_size_h[i] = 1; }
            else gap_size_h[i]++;
            const int step_right = best_gap_h[i];
            const int ki = gap_size_h[i];
            const int diagHighestOrEqual = (step_diag >= step_down) && (step_diag >= step_right);
            if (diagHighestOrEqual) {
                curRow[j] = step_diag > MATRIX_MIN_CUTOFF ? step_diag : MATRIX_MIN_CUTOFF;
                curBackTrackRow[j] = 0;
            } else if (step_right >= step_down) {
                curRow[j] = step_right > MATRIX_MIN_CUTOFF ? step_right : MATRIX_MIN_CUTOFF;
                curBackTrackRow[j] = -ki;
            } else {
                curRow[j] = step_down > MATRIX_MIN_CUTOFF ? step_down : MATRIX_MIN_CUTOFF;
                curBackTrackRow[j] = kd;
            }
        }
    }
    free(best_gap_v); free(gap_size_v); free(best_gap_h); free(gap_size_h);
}

struct elems { int32_t *ops, *lens; int n, cap, overflow; };
static void add(struct elems *e, int op, int len) {
    if (e->n >= e->cap) { e->overflow = 1; return; }
    e->ops[e->n] = op; e->lens[e->n] = len; e->n++;
}

/* Aligns alternate to reference.  Writes the consolidated CIGAR, first element first, into
 * cigar_ops / cigar_lens (capacity elements) and returns its length in *num_elems, the
 * alignment offset (getAlignmentStart2wrt1) in *alignment_offset.  When sw_out / btrack_out are not
 * NULL they receive the two (n+1) x (m+1) matrices. */
int oracle_sw_align(const uint8_t *reference, int refLength, const uint8_t *alternate, int altLength,
                    int w_match, int w_mismatch, int w_open, int w_extend, int strategy,
                    int32_t *cigar_ops, int32_t *cigar_lens, int capacity, int32_t *num_elems, int32_t *alignment_offset,
                    int32_t *sw_out, int32_t *btrack_out) {
    if (!reference || refLength <= 0 || !alternate || altLength <= 0) return SW_ILLEGAL_ARGUMENT;  /* :183-184 */
    const int nrow = refLength + 1, ncol = altLength + 1;
    int *sw = (int *)malloc(sizeof(int) * (size_t)nrow * ncol);
    int *btrack = (int *)malloc(sizeof(int) * (size_t)nrow * ncol);
    calculateMatrix(reference, refLength, alternate, altLength, w_match, w_mismatch, w_open, w_extend, strategy, sw, btrack);
    if (sw_out) memcpy(sw_out, sw, sizeof(int) * (size_t)nrow * ncol);
    if (btrack_out) memcpy(btrack_out, btrack, sizeof(int) * (size_t)nrow * ncol);

    /* calculateCigar, :350-470 */
    int p1 = 0, p2 = 0;
    int maxscore = INT32_MIN;
    int segment_length = 0;
    if (strategy == SW_INDEL) {
        p1 = refLength; p2 = altLength;
    } else {
        p2 = altLength;
        for (int i = 1; i < nrow; i++) {
            const int curScore = sw[(size_t)i * ncol + altLength];
            if (curScore >= maxscore) { p1 = i; maxscore = curScore; }
        }
        if (strategy != SW_LEADING_INDEL) {
            const int *bottomRow = sw + (size_t)refLength * ncol;
            for (int j = 1; j < ncol; j++) {
                const int curScore = bottomRow[j];
                if (curScore > maxscore || (curScore == maxscore && abs(refLength - j) < abs(p1 - p2))) {
                    p1 = refLength; p2 = j; maxscore = curScore;
                    segment_length = altLength - j;
                }
            }
        }
    }
    const int cap = refLength + altLength + 8;
    struct elems lce = { (int32_t *)malloc(sizeof(int32_t) * cap), (int32_t *)malloc(sizeof(int32_t) * cap), 0, cap, 0 };
    if (segment_length > 0 && strategy == SW_SOFTCLIP) { add(&lce, OP_S, segment_length); segment_length = 0; }
    int state = OP_M;
    do {
        const int btr = btrack[(size_t)p1 * ncol + p2];
        int new_state, step_length = 1;
        if (btr > 0) { new_state = OP_D; step_length = btr; }
        else if (btr < 0) { new_state = OP_I; step_length = -btr; }
        else new_state = OP_M;
        switch (new_state) {
            case OP_M: p1--; p2--; break;
            case OP_I: p2 -= step_length; break;
            case OP_D: p1 -= step_length; break;
        }
        if (new_state == state) segment_length += step_length;
        else { add(&lce, state, segment_length); segment_length = step_length; state = new_state; }
    } while (p1 > 0 && p2 > 0);
    int offset;
    if (strategy == SW_SOFTCLIP) {
        add(&lce, state, segment_length);
        if (p2 > 0) add(&lce, OP_S, p2);
        offset = p1;
    } else if (strategy == SW_IGNORE) {
        add(&lce, state, segment_length + p2);
        offset = p1 - p2;
    } else {
        add(&lce, state, segment_length);
        if (p1 > 0) add(&lce, OP_D, p1);
        else if (p2 > 0) add(&lce, OP_I, p2);
        offset = 0;
    }
    /* Collections.reverse + AlignmentUtils.consolidateCigar (sam/AlignmentUtils.java:698-727):
     * zero-length elements dropped, equal neighbours merged */
    int out = 0, rc = SW_OK;
    int sum = 0, last = -1;
    for (int k = lce.n - 1; k >= 0; k--) {
        if (lce.lens[k] == 0) continue;
        if (last != -1 && last != lce.ops[k]) {
            if (out < capacity) { cigar_ops[out] = last; cigar_lens[out] = sum; } else rc = SW_CIGAR_TOO_LONG;
            out++; sum = 0;
        }
        sum += lce.lens[k];
        last = lce.ops[k];
    }
    if (sum > 0) {
        if (out < capacity) { cigar_ops[out] = last; cigar_lens[out] = sum; } else rc = SW_CIGAR_TOO_LONG;
        out++;
    }
    *num_elems = out;
    *alignment_offset = offset;
    free(lce.ops); free(lce.lens); free(sw); free(btrack);
    return rc;
}
