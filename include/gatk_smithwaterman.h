/*
 * C-ABI of the MI355X (gfx950) Smith-Waterman pairwise aligner: the other dynamic programme of a
 * HaplotypeCaller region (every assembled haplotype is aligned to the reference window).  It stands
 * where the reference constructs a SWPairwiseAlignment
 * (smithwaterman/SWPairwiseAlignment.java:132-175: new SWPairwiseAlignment(seq1, seq2, parameters,
 * strategy), then getCigar() :177 and getAlignmentStart2wrt1() :180), many pairs per call.
 *
 * Results are identical to the reference algorithm's (integer scores, the same tie-breaking:
 * diagonal, then horizontal, then vertical, :316-331; the same choice of the end cell, :362-392).
 */
#ifndef GATK_SMITHWATERMAN_H
#define GATK_SMITHWATERMAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* smithwaterman/Parameters.java:35-61; named sets in SWParameterSet.java:38,43:
 * ORIGINAL_DEFAULT {3, -1, -4, -3}, STANDARD_NGS {25, -50, -110, -6}. */
typedef struct gatk_sw_parameters {
    int32_t w_match;
    int32_t w_mismatch;
    int32_t w_open;
    int32_t w_extend;
} gatk_sw_parameters;

/* SWPairwiseAlignment.OVERHANG_STRATEGY, :69-93 */
enum {
    GATK_SW_SOFTCLIP = 0,
    GATK_SW_INDEL = 1,
    GATK_SW_LEADING_INDEL = 2,
    GATK_SW_IGNORE = 3
};

/* CIGAR operators of the result (State -> CigarOperator, :472-481) */
enum { GATK_SW_OP_M = 0, GATK_SW_OP_I = 1, GATK_SW_OP_D = 2, GATK_SW_OP_S = 3 };

/* Return codes: 0, GATK_SW_ILLEGAL_ARGUMENT for what the Java answers with IllegalArgumentException
 * (a null or empty sequence, :183-184) and for sequences beyond the limits below,
 * GATK_SW_CIGAR_TOO_LONG when a result needs more than cigar_capacity elements,
 * GATK_SW_DEVICE_ERROR for HIP failures (no GPU: there is no CPU fallback). */
enum { GATK_SW_OK = 0, GATK_SW_ILLEGAL_ARGUMENT = 1, GATK_SW_DEVICE_ERROR = 4, GATK_SW_CIGAR_TOO_LONG = 5 };

/* Limits: reference (seq1) up to 2048 bases, alternate (seq2) up to 16000 bases. */

/*
 * Aligns alternates[k] to references[k] for k in [0, num_pairs) on GPU `device`.
 * Pair k's consolidated CIGAR (sam/AlignmentUtils.java:698-727) goes to
 * cigar_ops / cigar_lengths [k * cigar_capacity ...], first element first, its length to
 * cigar_counts[k], and getAlignmentStart2wrt1() to alignment_offsets[k].
 */
int gatk_sw_align(int32_t device, int32_t num_pairs,
                  const uint8_t* const* references, const int32_t* reference_lengths,
                  const uint8_t* const* alternates, const int32_t* alternate_lengths,
                  const gatk_sw_parameters* parameters, int32_t overhang_strategy,
                  int32_t cigar_capacity, int32_t* cigar_ops, int32_t* cigar_lengths, int32_t* cigar_counts,
                  int32_t* alignment_offsets);

/* Message of the last failing call on this thread. */
const char* gatk_sw_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GATK_SMITHWATERMAN_H */
