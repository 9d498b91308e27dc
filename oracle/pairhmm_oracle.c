/*
 * CPU oracle for the PairHMM path — TEST INFRASTRUCTURE ONLY.
 *
 * This file restates, in plain C and in IEEE double arithmetic, the algorithm of the
 * reference's PairHMM path so that the HIP engine can be checked against it.  Nothing in
 * the product (gatk_amd/, the C-ABI library) may call, link or import it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: the reference holds no golden vectors for the end-to-end path (there is
 * no PairHMM unit test in its tree) and cannot be built here (Java, no JDK), so the
 * end-to-end result of this oracle is "parity unpinned".  Its primitives ARE pinned by the
 * reference's own unit-test assertions (MathUtilsUnitTest.java:332-485,
 * QualityUtilsUnitTest.java:53-133), which tests/test_oracle_primitives.py replays, and
 * the three forward implementations below (exact log10, approximate log10, linear
 * "logless") are cross-checked against each other and against closed-form answers.
 *
 * All file:line citations are under
 *   /root/reference/public/gatk-utils/src/main/java/org/broadinstitute/gatk/utils/
 *
 * The linear-space ("logless") forward pass is NOT in the reference tree (it lives in
 * GATK's protected module); it is restated here from the in-tree linear transition
 * function PairHMMModel.qualToTransProbs (pairhmm/PairHMMModel.java:135-144), the in-tree
 * recurrence Log10PairHMM.updateCell (pairhmm/Log10PairHMM.java:211-219) taken out of
 * log space, and the published upstream constant INITIAL_CONDITION = 2^1020.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_QUAL 254 /* QualityUtils.java:63 */
#define TRANS_PROB_ARRAY_LENGTH 6 /* pairhmm/PairHMMModel.java:49 */
enum { matchToMatch = 0, indelToMatch = 1, matchToInsertion = 2, insertionToInsertion = 3,
       matchToDeletion = 4, deletionToDeletion = 5 }; /* pairhmm/PairHMMModel.java:54-79 */

/* error codes returned by the batch entry points; they stand for the Java exceptions */
enum { ORACLE_OK = 0,
       ORACLE_ILLEGAL_ARGUMENT = 1,  /* IllegalArgumentException */
       ORACLE_ILLEGAL_STATE = 2,     /* IllegalStateException    */
       ORACLE_INDEX_OUT_OF_BOUNDS = 3 /* ArrayIndexOutOfBoundsException (base qual 255) */ };

/* ------------------------------------------------------------------------------------
 * QualityUtils
 * ---------------------------------------------------------------------------------- */
static double qualToErrorProbCache[MAX_QUAL + 1]; /* QualityUtils.java:68 */
static double qualToProbLog10Cache[MAX_QUAL + 1]; /* QualityUtils.java:69 */

/* QualityUtils.java:153-156 */
double oracle_qualToErrorProb_d(double qual) { return pow(10.0, qual / -10.0); }

/* ------------------------------------------------------------------------------------
 * MathUtils
 * ---------------------------------------------------------------------------------- */
#define MAX_TOLERANCE 8.0 /* MathUtils.java:126 */
static const double TABLE_STEP = 0.0001;   /* MathUtils.java:144 */
static double INV_STEP;                    /* MathUtils.java:145 */
static double *jacobianCache = NULL;       /* MathUtils.java:146 */
static int jacobianSize = 0;

/* MathUtils.java:152-154 */
static int fastRound(double d) { return (d > 0.0) ? (int)(d + 0.5) : (int)(d - 0.5); }

/* MathUtils.java:128-133 */
static double jacobianGet(double difference) { return jacobianCache[fastRound(difference * INV_STEP)]; }

/* MathUtils.java:895-909 (start = 0) */
static int maxElementIndex(const double *a, int endIndex) {
    int maxI = 0;
    for (int i = 1; i < endIndex; i++)
        if (a[i] > a[maxI]) maxI = i;
    return maxI;
}

/* MathUtils.java:186-203 */
double oracle_approximateLog10SumLog10_arr(const double *vals, int endIndex) {
    const int maxElementIdx = maxElementIndex(vals, endIndex);
    double approxSum = vals[maxElementIdx];
    for (int i = 0; i < endIndex; i++) {
        if (i == maxElementIdx || vals[i] == -INFINITY) continue;
        const double diff = approxSum - vals[i];
        if (diff < MAX_TOLERANCE) approxSum += jacobianGet(diff);
    }
    return approxSum;
}

/* MathUtils.java:209-231 */
double oracle_approximateLog10SumLog10_2(double small, double big) {
    if (small > big) { const double t = big; big = small; small = t; }
    if (small == -INFINITY || big == -INFINITY) return big;
    const double diff = big - small;
    if (diff >= MAX_TOLERANCE) return big;
    return big + jacobianGet(diff);
}

/* MathUtils.java:205-207 */
double oracle_approximateLog10SumLog10_3(double a, double b, double c) {
    return oracle_approximateLog10SumLog10_2(a, oracle_approximateLog10SumLog10_2(b, c));
}

/* MathUtils.java:300-323.  The Java throws IllegalArgumentException when the running sum
 * is NaN or +Inf; here that case returns NaN and the callers below turn it into an error. */
double oracle_log10sumLog10(const double *log10p, int start, int finish) {
    if (start >= finish) return -INFINITY;
    int maxElementIdx = start;
    for (int i = start + 1; i < finish; i++)
        if (log10p[i] > log10p[maxElementIdx]) maxElementIdx = i;
    const double maxValue = log10p[maxElementIdx];
    if (maxValue == -INFINITY) return maxValue;
    double sum = 1.0;
    for (int i = start; i < finish; i++) {
        const double curVal = log10p[i];
        const double scaled_val = curVal - maxValue;
        if (i == maxElementIdx || curVal == -INFINITY) continue;
        sum += pow(10.0, scaled_val);
    }
    if (isnan(sum) || sum == INFINITY) return NAN;
    return maxValue + (sum != 1.0 ? log10(sum) : 0.0);
}

/* MathUtils.java:1241-1254 (allowNegativeInfinity = true) */
int oracle_goodLog10Probability(double result) {
    return result <= 0.0 && result != INFINITY && !isnan(result);
}

/* ------------------------------------------------------------------------------------
 * PairHMMModel
 * ---------------------------------------------------------------------------------- */
#define MM_TABLE_SIZE (((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1)
static double matchToMatchProbTab[MM_TABLE_SIZE];  /* pairhmm/PairHMMModel.java:99  */
static double matchToMatchLog10Tab[MM_TABLE_SIZE]; /* pairhmm/PairHMMModel.java:109 */
static double LN10, INV_LN10;                      /* pairhmm/PairHMMModel.java:84,89 */

static int oracle_ready = 0;

/* PairHMM.doNotUseTristateCorrection, pairhmm/PairHMM.java:83-84 ("only used for debugging purposes"): when set,
 * the mismatch prior is the error probability itself, not a third of it (pairhmm/Log10PairHMM.java:157).  A switch
 * of the whole oracle, as the Java field is a switch of a PairHMM instance; tests set it and clear it again. */
static int doNotUseTristateCorrection = 0;
void oracle_doNotUseTristateCorrection(int on) { doNotUseTristateCorrection = on != 0; }

/* Static initialisers of QualityUtils (:72-77), MathUtils.JacobianLogTable (:135-142)
 * and PairHMMModel (:114-122), in dependency order. */
void oracle_init(void) {
    if (oracle_ready) return;
    for (int i = 0; i <= MAX_QUAL; i++) {
        qualToErrorProbCache[i] = oracle_qualToErrorProb_d((double)i);
        qualToProbLog10Cache[i] = log10(1.0 - qualToErrorProbCache[i]);
    }
    INV_STEP = 1.0 / TABLE_STEP;
    jacobianSize = (int)(MAX_TOLERANCE / TABLE_STEP) + 1;
    jacobianCache = (double *)malloc(sizeof(double) * (size_t)jacobianSize);
    for (int k = 0; k < jacobianSize; k++)
        jacobianCache[k] = log10(1.0 + pow(10.0, -((double)k) * TABLE_STEP));
    LN10 = log(10.0);
    INV_LN10 = 1.0 / LN10;
    for (int i = 0, offset = 0; i <= MAX_QUAL; offset += ++i)
        for (int j = 0; j <= i; j++) {
            const double log10Sum = oracle_approximateLog10SumLog10_2(-0.1 * i, -0.1 * j);
            /* Math.log1p(-1) is -Infinity in Java; C's log1p(-1) returns -HUGE_VAL too */
            matchToMatchLog10Tab[offset + j] = log1p(-fmin(1.0, pow(10.0, log10Sum))) * INV_LN10;
            matchToMatchProbTab[offset + j] = pow(10.0, matchToMatchLog10Tab[offset + j]);
        }
    oracle_ready = 1;
}

int oracle_jacobian_table_size(void) { oracle_init(); return jacobianSize; }

/* QualityUtils.java:172-174; the Java indexes a 255-entry array with (qual & 0xff), so a
 * byte of 255 raises ArrayIndexOutOfBoundsException: callers check for it first. */
double oracle_qualToErrorProb(uint8_t qual) { return qualToErrorProbCache[qual]; }
/* QualityUtils.java:104-106 */
double oracle_qualToProb(uint8_t qual) { return 1.0 - oracle_qualToErrorProb(qual); }
/* QualityUtils.java:138-140 */
double oracle_qualToProbLog10(uint8_t qual) { return qualToProbLog10Cache[qual]; }
/* QualityUtils.java:191-193,206-209 */
double oracle_qualToErrorProbLog10(uint8_t qual) { return (double)qual * -0.1; }

/* pairhmm/PairHMMModel.java:387-402 */
double oracle_matchToMatchProb(int insQual, int delQual) {
    int minQual, maxQual;
    if (insQual <= delQual) { minQual = insQual; maxQual = delQual; }
    else { minQual = delQual; maxQual = insQual; }
    if (minQual < 0) return NAN;
    return (MAX_QUAL < maxQual)
        ? 1.0 - pow(10.0, oracle_approximateLog10SumLog10_2(-0.1 * minQual, -0.1 * maxQual))
        : matchToMatchProbTab[((maxQual * (maxQual + 1)) >> 1) + minQual];
}

/* pairhmm/PairHMMModel.java:420-434 */
double oracle_matchToMatchProbLog10(int insQual, int delQual) {
    int minQual, maxQual;
    if (insQual <= delQual) { minQual = insQual; maxQual = delQual; }
    else { minQual = delQual; maxQual = insQual; }
    return (MAX_QUAL < maxQual)
        ? log1p(-fmin(1.0, pow(10.0, oracle_approximateLog10SumLog10_2(-.1 * minQual, -.1 * maxQual)))) * INV_LN10
        : matchToMatchLog10Tab[((maxQual * (maxQual + 1)) >> 1) + minQual];
}

/* pairhmm/PairHMMModel.java:135-144.  Java bytes are signed: a stored value >= 128 is
 * negative and raises IllegalArgumentException. */
int oracle_qualToTransProbs(double *dest, uint8_t insQual, uint8_t delQual, uint8_t gcp) {
    if (insQual >= 128 || delQual >= 128 || gcp >= 128) return ORACLE_ILLEGAL_ARGUMENT;
    dest[matchToMatch] = oracle_matchToMatchProb(insQual, delQual);
    dest[matchToInsertion] = oracle_qualToErrorProb(insQual);
    dest[matchToDeletion] = oracle_qualToErrorProb(delQual);
    dest[indelToMatch] = oracle_qualToProb(gcp);
    dest[insertionToInsertion] = dest[deletionToDeletion] = oracle_qualToErrorProb(gcp);
    return ORACLE_OK;
}

/* pairhmm/PairHMMModel.java:235-244 */
int oracle_qualToTransProbsLog10(double *dest, uint8_t insQual, uint8_t delQual, uint8_t gcp) {
    if (insQual >= 128 || delQual >= 128 || gcp >= 128) return ORACLE_ILLEGAL_ARGUMENT;
    dest[matchToMatch] = oracle_matchToMatchProbLog10(insQual, delQual);
    dest[matchToInsertion] = oracle_qualToErrorProbLog10(insQual);
    dest[matchToDeletion] = oracle_qualToErrorProbLog10(delQual);
    dest[indelToMatch] = oracle_qualToProbLog10(gcp);
    dest[insertionToInsertion] = dest[deletionToDeletion] = oracle_qualToErrorProbLog10(gcp);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------
 * N2MemoryPairHMM / Log10PairHMM / logless forward pass for ONE (read, haplotype) pair
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int paddedMaxReadLength, paddedMaxHaplotypeLength; /* pairhmm/PairHMM.java:112-113 */
    double *M, *I, *D, *prior; /* pairhmm/N2MemoryPairHMM.java:62-67, row-major [R+1][H+1] */
    double *transition;        /* [R+1][6], pairhmm/PairHMMModel.java:329-331 */
} n2mem;

static int n2mem_alloc(n2mem *m, int readMaxLength, int haplotypeMaxLength) {
    m->paddedMaxReadLength = readMaxLength + 1;
    m->paddedMaxHaplotypeLength = haplotypeMaxLength + 1;
    const size_t n = (size_t)m->paddedMaxReadLength * (size_t)m->paddedMaxHaplotypeLength;
    m->M = (double *)calloc(n, sizeof(double));
    m->I = (double *)calloc(n, sizeof(double));
    m->D = (double *)calloc(n, sizeof(double));
    m->prior = (double *)calloc(n, sizeof(double));
    m->transition = (double *)calloc((size_t)m->paddedMaxReadLength * 6, sizeof(double));
    return (m->M && m->I && m->D && m->prior && m->transition) ? 0 : -1;
}
static void n2mem_free(n2mem *m) {
    free(m->M); free(m->I); free(m->D); free(m->prior); free(m->transition);
    memset(m, 0, sizeof(*m));
}

enum { MODE_LOG10_EXACT = 0,   /* HMM_IMPLEMENTATION.EXACT    pairhmm/PairHMM.java:57-58 */
       MODE_LOG10_APPROX = 1,  /* HMM_IMPLEMENTATION.ORIGINAL pairhmm/PairHMM.java:59-60 */
       MODE_LOGLESS = 2        /* HMM_IMPLEMENTATION.LOGLESS_CACHING pairhmm/PairHMM.java:61-62 */ };

/* pairhmm/Log10PairHMM.java:196-198 */
static double myLog10SumLog10(int mode, const double *v, int n) {
    return mode == MODE_LOG10_EXACT ? oracle_log10sumLog10(v, 0, n)
                                    : oracle_approximateLog10SumLog10_arr(v, n);
}

/* Log10PairHMM.subComputeReadLikelihoodGivenHaplotypeLog10 with hapStartIndex = 0 and
 * recacheReadValues = true, as PairHMM.computeLikelihoods always calls it
 * (pairhmm/PairHMM.java:217,284; pairhmm/Log10PairHMM.java:89-119). */
static double log10_pair(int mode, n2mem *m, const uint8_t *hap, int H, const uint8_t *readBases,
                         const uint8_t *readQuals, const uint8_t *insGOP, const uint8_t *delGOP,
                         const uint8_t *gcp, int R, int *err) {
    const int W = m->paddedMaxHaplotypeLength;
    const int paddedReadLength = R + 1, paddedHaplotypeLength = H + 1;
    /* Log10PairHMM.initialize :75-83 — every cell starts at -Infinity */
    for (int i = 0; i < paddedReadLength; i++)
        for (int j = 0; j < paddedHaplotypeLength; j++)
            m->M[i * W + j] = m->I[i * W + j] = m->D[i * W + j] = -INFINITY;
    /* initializeProbabilities :175-179 -> PairHMMModel.qualToTransProbsLog10 :286-294 */
    for (int i = 0; i < R; i++) {
        const int e = oracle_qualToTransProbsLog10(m->transition + (size_t)(i + 1) * 6, insGOP[i], delGOP[i], gcp[i]);
        if (e) { *err = e; return NAN; }
    }
    /* initializePriors :146-160 */
    const double log10_3 = log10(3.0); /* :52 */
    for (int i = 0; i < R; i++) {
        const uint8_t x = readBases[i], qual = readQuals[i];
        if (qual > MAX_QUAL) { *err = ORACLE_INDEX_OUT_OF_BOUNDS; return NAN; }
        for (int j = 0; j < H; j++) {
            const uint8_t y = hap[j];
            m->prior[(i + 1) * W + (j + 1)] = (x == y || x == (uint8_t)'N' || y == (uint8_t)'N')
                ? oracle_qualToProbLog10(qual) : (oracle_qualToErrorProbLog10(qual) - (doNotUseTristateCorrection ? 0.0 : log10_3));
        }
    }
    /* initializeMatrixValues :121-126 */
    const double initialValue = log10(1.0 / H);
    for (int j = 0; j < paddedHaplotypeLength; j++) m->D[0 * W + j] = initialValue;
    /* the double loop :108-113 with updateCell :211-219 */
    for (int i = 1; i < paddedReadLength; i++) {
        const double *t = m->transition + (size_t)i * 6;
        for (int j = 1; j < paddedHaplotypeLength; j++) {
            double v3[3], v2[2];
            v3[0] = m->M[(i - 1) * W + j - 1] + t[matchToMatch];
            v3[1] = m->I[(i - 1) * W + j - 1] + t[indelToMatch];
            v3[2] = m->D[(i - 1) * W + j - 1] + t[indelToMatch];
            m->M[i * W + j] = m->prior[i * W + j] + myLog10SumLog10(mode, v3, 3);
            v2[0] = m->M[(i - 1) * W + j] + t[matchToInsertion];
            v2[1] = m->I[(i - 1) * W + j] + t[insertionToInsertion];
            m->I[i * W + j] = myLog10SumLog10(mode, v2, 2);
            v2[0] = m->M[i * W + j - 1] + t[matchToDeletion];
            v2[1] = m->D[i * W + j - 1] + t[deletionToDeletion];
            m->D[i * W + j] = myLog10SumLog10(mode, v2, 2);
        }
    }
    /* finalLikelihoodCalculation :128-134 */
    const int endI = paddedReadLength - 1;
    double v2[2] = { m->M[endI * W + 1], m->I[endI * W + 1] };
    double finalSumProbabilities = myLog10SumLog10(mode, v2, 2);
    for (int j = 2; j < paddedHaplotypeLength; j++) {
        double v3[3] = { finalSumProbabilities, m->M[endI * W + j], m->I[endI * W + j] };
        finalSumProbabilities = myLog10SumLog10(mode, v3, 3);
    }
    return finalSumProbabilities;
}

/* Linear-space forward pass (see the header: restated, not in the reference tree).
 * Same matrices, boundaries, recurrence and termination as log10_pair, in probability
 * space, with the first deletion row scaled by 2^1020 so that long products stay inside
 * the double range; the scale is taken out again in log space at the end. */
static double logless_pair(n2mem *m, const uint8_t *hap, int H, const uint8_t *readBases,
                           const uint8_t *readQuals, const uint8_t *insGOP, const uint8_t *delGOP,
                           const uint8_t *gcp, int R, int *err) {
    const int W = m->paddedMaxHaplotypeLength;
    const int paddedReadLength = R + 1, paddedHaplotypeLength = H + 1;
    const double INITIAL_CONDITION = ldexp(1.0, 1020);
    const double INITIAL_CONDITION_LOG10 = log10(INITIAL_CONDITION);
    /* PairHMMModel.qualToTransProbs(double[][],...) pairhmm/PairHMMModel.java:187-196 */
    for (int i = 0; i < R; i++) {
        const int e = oracle_qualToTransProbs(m->transition + (size_t)(i + 1) * 6, insGOP[i], delGOP[i], gcp[i]);
        if (e) { *err = e; return NAN; }
    }
    /* priors: linear form of pairhmm/Log10PairHMM.java:146-160 */
    for (int i = 0; i < R; i++) {
        const uint8_t x = readBases[i], qual = readQuals[i];
        if (qual > MAX_QUAL) { *err = ORACLE_INDEX_OUT_OF_BOUNDS; return NAN; }
        for (int j = 0; j < H; j++) {
            const uint8_t y = hap[j];
            m->prior[(i + 1) * W + (j + 1)] = (x == y || x == (uint8_t)'N' || y == (uint8_t)'N')
                ? oracle_qualToProb(qual) : (oracle_qualToErrorProb(qual) / (doNotUseTristateCorrection ? 1.0 : 3.0));
        }
    }
    /* boundaries: linear form of pairhmm/Log10PairHMM.java:75-83,121-126 */
    for (int i = 0; i < paddedReadLength; i++)
        m->M[i * W] = m->I[i * W] = m->D[i * W] = 0.0;
    const double initialValue = INITIAL_CONDITION / H;
    for (int j = 0; j < paddedHaplotypeLength; j++) {
        m->M[j] = 0.0; m->I[j] = 0.0; m->D[j] = initialValue;
    }
    /* recurrence: linear form of pairhmm/Log10PairHMM.java:108-113,211-219 */
    for (int i = 1; i < paddedReadLength; i++) {
        const double *t = m->transition + (size_t)i * 6;
        for (int j = 1; j < paddedHaplotypeLength; j++) {
            m->M[i * W + j] = m->prior[i * W + j] * (m->M[(i - 1) * W + j - 1] * t[matchToMatch] +
                                                     m->I[(i - 1) * W + j - 1] * t[indelToMatch] +
                                                     m->D[(i - 1) * W + j - 1] * t[indelToMatch]);
            m->I[i * W + j] = m->M[(i - 1) * W + j] * t[matchToInsertion] + m->I[(i - 1) * W + j] * t[insertionToInsertion];
            m->D[i * W + j] = m->M[i * W + j - 1] * t[matchToDeletion] + m->D[i * W + j - 1] * t[deletionToDeletion];
        }
    }
    /* termination: linear form of pairhmm/Log10PairHMM.java:128-134 */
    const int endI = paddedReadLength - 1;
    double finalSumProbabilities = 0.0;
    for (int j = 1; j < paddedHaplotypeLength; j++)
        finalSumProbabilities += m->M[endI * W + j] + m->I[endI * W + j];
    return log10(finalSumProbabilities) - INITIAL_CONDITION_LOG10;
}

/* ------------------------------------------------------------------------------------
 * PairHMM.computeLikelihoods — the batch loop (pairhmm/PairHMM.java:190-236) with the
 * per-pair wrapper computeReadLikelihoodGivenHaplotypeLog10 (:262-306).
 *
 * Reads and haplotypes arrive as concatenated byte arrays with offset tables
 * (offsets[n] .. offsets[n+1]).  out is read-major: out[r * numHaps + a] (:206,225).
 * Returns ORACLE_OK or the code of the first exception the Java would have thrown.
 * ---------------------------------------------------------------------------------- */
int oracle_computeLikelihoods(int mode, int numReads, int numHaps,
                              const uint8_t *readBases, const uint8_t *readQuals,
                              const uint8_t *insGOP, const uint8_t *delGOP, const uint8_t *gcp,
                              const int32_t *readOffsets,
                              const uint8_t *hapBases, const int32_t *hapOffsets,
                              double *out) {
    oracle_init();
    if (numReads == 0) return ORACLE_OK; /* :193 */
    int readMaxLength = 0, haplotypeMaxLength = 0; /* :198-199 */
    for (int r = 0; r < numReads; r++) {
        const int len = readOffsets[r + 1] - readOffsets[r];
        if (len > readMaxLength) readMaxLength = len;
    }
    for (int a = 0; a < numHaps; a++) {
        const int len = hapOffsets[a + 1] - hapOffsets[a];
        if (len > haplotypeMaxLength) haplotypeMaxLength = len;
    }
    /* initialize(readMaxLength, haplotypeMaxLength) :104-106, as a fresh PairHMM instance
     * reaches it through :200-201 */
    if (readMaxLength <= 0 || haplotypeMaxLength <= 0) return ORACLE_ILLEGAL_ARGUMENT;
    /* A zero-length haplotype makes the Java read matrix column 1 of a row it never wrote
     * (pairhmm/Log10PairHMM.java:130 with paddedHaplotypeLength == 1): the value returned is
     * whatever an earlier pair left there.  No defined answer exists, so it is an error here. */
    for (int a = 0; a < numHaps; a++)
        if (hapOffsets[a + 1] - hapOffsets[a] <= 0) return ORACLE_ILLEGAL_ARGUMENT;
    n2mem m;
    if (n2mem_alloc(&m, readMaxLength, haplotypeMaxLength)) return -1;
    int err = ORACLE_OK;
    for (int r = 0; r < numReads && !err; r++) { /* :209 */
        const int ro = readOffsets[r], R = readOffsets[r + 1] - ro;
        for (int a = 0; a < numHaps && !err; a++) { /* :218 */
            const int ho = hapOffsets[a], H = hapOffsets[a + 1] - ho;
            double lk;
            if (mode == MODE_LOGLESS)
                lk = logless_pair(&m, hapBases + ho, H, readBases + ro, readQuals + ro, insGOP + ro, delGOP + ro, gcp + ro, R, &err);
            else
                lk = log10_pair(mode, &m, hapBases + ho, H, readBases + ro, readQuals + ro, insGOP + ro, delGOP + ro, gcp + ro, R, &err);
            if (err) break;
            /* :292-295 */
            if (lk > 0.0 || !oracle_goodLog10Probability(lk)) { err = ORACLE_ILLEGAL_STATE; break; }
            out[(size_t)r * numHaps + a] = lk; /* :224-225 */
        }
    }
    n2mem_free(&m);
    return err;
}

/* One pair through the same wrapper; convenience for known-answer tests. */
int oracle_pair(int mode, const uint8_t *hap, int H, const uint8_t *readBases, const uint8_t *readQuals,
                const uint8_t *insGOP, const uint8_t *delGOP, const uint8_t *gcp, int R, double *out) {
    const int32_t ro[2] = { 0, R }, ho[2] = { 0, H };
    return oracle_computeLikelihoods(mode, 1, 1, readBases, readQuals, insGOP, delGOP, gcp, ro, hap, ho, out);
}

/* Linear-space batch with O(H) working memory per pair and optional OpenMP over reads:
 * the same arithmetic as logless_pair in the same order, kept for timing the CPU baseline
 * (the N^2 matrices above are what the reference allocates, pairhmm/N2MemoryPairHMM.java:59-68,
 * but they make the baseline memory-bound for no algorithmic reason). */
int oracle_logless_batch_fast(int numReads, int numHaps,
                              const uint8_t *readBases, const uint8_t *readQuals,
                              const uint8_t *insGOP, const uint8_t *delGOP, const uint8_t *gcp,
                              const int32_t *readOffsets,
                              const uint8_t *hapBases, const int32_t *hapOffsets,
                              double *out, int numThreads) {
    oracle_init();
    int Hmax = 0, Rmax = 0;
    for (int a = 0; a < numHaps; a++) { int l = hapOffsets[a + 1] - hapOffsets[a]; if (l > Hmax) Hmax = l; }
    for (int r = 0; r < numReads; r++) { int l = readOffsets[r + 1] - readOffsets[r]; if (l > Rmax) Rmax = l; }
    const double INITIAL_CONDITION = ldexp(1.0, 1020);
    const double INITIAL_CONDITION_LOG10 = log10(INITIAL_CONDITION);
    int err = ORACLE_OK;
    if (numThreads < 1) numThreads = 1;
#pragma omp parallel num_threads(numThreads)
    {
        double *buf = (double *)malloc(sizeof(double) * ((size_t)6 * (Hmax + 1) + (size_t)8 * (Rmax + 1)));
        double *Mp = buf, *Ip = Mp + Hmax + 1, *Dp = Ip + Hmax + 1, *Mc = Dp + Hmax + 1, *Ic = Mc + Hmax + 1, *Dc = Ic + Hmax + 1;
        double *tr = Dc + Hmax + 1, *pm = tr + (size_t)6 * (Rmax + 1), *pmm = pm + Rmax + 1;
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < numReads; r++) {
            const int ro = readOffsets[r], R = readOffsets[r + 1] - ro;
            int bad = 0;
            for (int i = 0; i < R; i++) {
                if (oracle_qualToTransProbs(tr + (size_t)(i + 1) * 6, insGOP[ro + i], delGOP[ro + i], gcp[ro + i])) { bad = ORACLE_ILLEGAL_ARGUMENT; break; }
                if (readQuals[ro + i] > MAX_QUAL) { bad = ORACLE_INDEX_OUT_OF_BOUNDS; break; }
                pm[i + 1] = oracle_qualToProb(readQuals[ro + i]);
                pmm[i + 1] = oracle_qualToErrorProb(readQuals[ro + i]) / (doNotUseTristateCorrection ? 1.0 : 3.0);
            }
            if (bad) {
#pragma omp critical
                err = bad;
                continue;
            }
            for (int a = 0; a < numHaps; a++) {
                const int ho = hapOffsets[a], H = hapOffsets[a + 1] - ho;
                const uint8_t *hap = hapBases + ho;
                double *mp = Mp, *ip = Ip, *dp = Dp, *mc = Mc, *ic = Ic, *dc = Dc;
                for (int j = 0; j <= H; j++) { mp[j] = 0.0; ip[j] = 0.0; dp[j] = INITIAL_CONDITION / H; }
                for (int i = 1; i <= R; i++) {
                    const double *t = tr + (size_t)i * 6;
                    const uint8_t x = readBases[ro + i - 1];
                    mc[0] = ic[0] = dc[0] = 0.0;
                    for (int j = 1; j <= H; j++) {
                        const uint8_t y = hap[j - 1];
                        const double prior = (x == y || x == (uint8_t)'N' || y == (uint8_t)'N') ? pm[i] : pmm[i];
                        mc[j] = prior * (mp[j - 1] * t[matchToMatch] + ip[j - 1] * t[indelToMatch] + dp[j - 1] * t[indelToMatch]);
                        ic[j] = mp[j] * t[matchToInsertion] + ip[j] * t[insertionToInsertion];
                        dc[j] = mc[j - 1] * t[matchToDeletion] + dc[j - 1] * t[deletionToDeletion];
                    }
                    double *s;
                    s = mp; mp = mc; mc = s; s = ip; ip = ic; ic = s; s = dp; dp = dc; dc = s;
                }
                double fin = 0.0;
                for (int j = 1; j <= H; j++) fin += mp[j] + ip[j];
                out[(size_t)r * numHaps + a] = log10(fin) - INITIAL_CONDITION_LOG10;
            }
        }
        free(buf);
    }
    return err;
}

/* Table accessors for tests */
double oracle_jacobian_entry(int k) { oracle_init(); return jacobianCache[k]; }
double oracle_mm_table_entry(int idx, int log10form) { oracle_init(); return log10form ? matchToMatchLog10Tab[idx] : matchToMatchProbTab[idx]; }

/* ----------------------------------------------------------------------------------
 * Downstream of the path: genotyper/ReadLikelihoods.java, one sample.
 * values is the Java layout, values[a * readCount + r] = sampleValues[a][r].
 * nonRefAlleleIndex is -1 throughout (no <NON_REF> allele is added on this path).
 * ---------------------------------------------------------------------------------- */

/* ReadLikelihoods.java:454-484 with canBeReference = false and every allele considered:
 * the best likelihood among the non-reference alleles, -inf when there is none. */
static double bestAlternativeLikelihood(const double *values, int alleleCount, int readCount, int referenceAlleleIndex, int r) {
    if (alleleCount == 0 || (alleleCount == 1 && referenceAlleleIndex == 0)) return -INFINITY;  /* :457-458 */
    int best = referenceAlleleIndex != 0 ? 0 : 1;                                                /* :461 */
    double bestLikelihood = values[(size_t)best * readCount + r];
    for (int a = best + 1; a < alleleCount; a++) {
        if (referenceAlleleIndex == a) continue;                                                 /* :466-467 */
        const double c = values[(size_t)a * readCount + r];
        if (c > bestLikelihood) bestLikelihood = c;                                              /* :473-477 */
    }
    return bestLikelihood;
}

/* ReadLikelihoods.normalizeLikelihoods, :353-404.  Returns ORACLE_ILLEGAL_ARGUMENT for a cap that
 * is not negative (:354-355). */
int oracle_normalizeLikelihoods(double *values, int alleleCount, int readCount, int referenceAlleleIndex,
                                int bestToZero, double maximumLikelihoodDifferenceCap) {
    if (maximumLikelihoodDifferenceCap >= 0.0 || isnan(maximumLikelihoodDifferenceCap)) return ORACLE_ILLEGAL_ARGUMENT;
    if (maximumLikelihoodDifferenceCap == -INFINITY && !bestToZero) return ORACLE_OK;            /* :357-358 */
    if (alleleCount == 0) return ORACLE_OK;                                                      /* :361-362 */
    if (alleleCount == 1 && !bestToZero) return ORACLE_OK;                                       /* :363-364 */
    for (int r = 0; r < readCount; r++) {                                                        /* :375-403 */
        const double bestAlt = bestAlternativeLikelihood(values, alleleCount, readCount, referenceAlleleIndex, r);
        const double worstLikelihoodCap = bestAlt + maximumLikelihoodDifferenceCap;
        const double referenceLikelihood = referenceAlleleIndex == -1 ? -INFINITY : values[(size_t)referenceAlleleIndex * readCount + r];
        const double bestAbsoluteLikelihood = bestAlt > referenceLikelihood ? bestAlt : referenceLikelihood;  /* Math.max */
        for (int a = 0; a < alleleCount; a++) {
            double *v = &values[(size_t)a * readCount + r];
            if (bestToZero) {
                if (bestAbsoluteLikelihood == -INFINITY) *v = 0;
                else if (worstLikelihoodCap != -INFINITY) *v = (*v < worstLikelihoodCap ? worstLikelihoodCap : *v) - bestAbsoluteLikelihood;
                else *v -= bestAbsoluteLikelihood;
            } else if (*v < worstLikelihoodCap) *v = worstLikelihoodCap;
        }
    }
    return ORACLE_OK;
}

/* ReadLikelihoods.filterPoorlyModeledReads, :1059-1093, as flags: poorlyModelled[r] = 1 for the reads
 * the Java removes.  ORACLE_ILLEGAL_STATE without alleles (:1060-1061), ORACLE_ILLEGAL_ARGUMENT for a
 * rate that is not positive (:1062-1063). */
int oracle_poorlyModelledReads(const double *values, int alleleCount, int readCount, const int32_t *readLengths,
                               double maximumErrorPerBase, uint8_t *poorlyModelled) {
    if (alleleCount == 0) return ORACLE_ILLEGAL_STATE;
    if (isnan(maximumErrorPerBase) || maximumErrorPerBase <= 0.0) return ORACLE_ILLEGAL_ARGUMENT;
    for (int r = 0; r < readCount; r++) {
        const double maxErrorsForRead = fmin(2.0, ceil(readLengths[r] * maximumErrorPerBase));    /* :1082 */
        const double log10MaxLikelihoodForTrueAllele = maxErrorsForRead * -4.0;                  /* :1083-1084 */
        uint8_t poorly = 1;
        for (int a = 0; a < alleleCount; a++)
            if (values[(size_t)a * readCount + r] >= log10MaxLikelihoodForTrueAllele) { poorly = 0; break; }  /* :1088-1090 */
        poorlyModelled[r] = poorly;
    }
    return ORACLE_OK;
}

/* ReadLikelihoods.marginalLikelihoods, :911-939, keeping every read: the new allele's likelihood is the
 * maximum over the old alleles mapped to it (oldToNewAlleleIndexMap[a] = -1: dropped). */
void oracle_marginalLikelihoods(const double *oldValues, int oldAlleleCount, int newAlleleCount, int readCount,
                                const int32_t *oldToNewAlleleIndexMap, double *newValues) {
    for (size_t k = 0; k < (size_t)newAlleleCount * readCount; k++) newValues[k] = -INFINITY;
    for (int r = 0; r < readCount; r++)
        for (int a = 0; a < oldAlleleCount; a++) {
            const int n = oldToNewAlleleIndexMap[a];
            if (n == -1) continue;
            const double lk = oldValues[(size_t)a * readCount + r];
            if (lk > newValues[(size_t)n * readCount + r]) newValues[(size_t)n * readCount + r] = lk;
        }
}

/* ----------------------------------------------------------------------------------
 * Upstream of the path: what a read's qualities look like when the PairHMM receives them.
 * In the reference tree: the flat Q45 default of reads without insertion / deletion qualities
 * (sam/GATKSAMRecord.java:61,298-322) and the two constants of the base-quality squashing
 * (pairhmm/PairHMM.java:54 BASE_QUALITY_SCORE_THRESHOLD = 18, QualityUtils.java:57
 * MIN_USABLE_Q_SCORE = 6).  The method that applies the constants (cap by mapping quality, then
 * squash) belongs to the likelihood engine upstream of the tree; it is restated from its
 * documented behaviour, not from code.
 * ---------------------------------------------------------------------------------- */
void oracle_defaultIndelQualities(uint8_t *quals, int n) {
    for (int i = 0; i < n; i++) quals[i] = 45;
}

/* mapq < 0: no cap. */
void oracle_modifyReadQualities(uint8_t *quals, int n, int mapq, int squash) {
    for (int i = 0; i < n; i++) {
        int q = quals[i];
        if (mapq >= 0 && q > mapq) q = mapq;
        if (squash && q < 18) q = 6;
        quals[i] = (uint8_t)q;
    }
}
