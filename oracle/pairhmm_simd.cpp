// CPU baseline for timing (TEST / BENCHMARK INFRASTRUCTURE, never linked into the product):
// a vectorised PairHMM forward pass in the style of the AVX library GATK 3.x loads for
// VECTOR_LOGLESS_CACHING (named in pairhmm/PairHMM.java:66-69, not in the reference tree):
// fp32 evaluation in linear space with scale 2^120 and an fp64 redo of every pair whose fp32 sum
// falls below 1e-28.  Written from the recurrence of pairhmm/Log10PairHMM.java:211-219 and the linear
// transitions of pairhmm/PairHMMModel.java:135-144; the tables come from the oracle
// (pairhmm_oracle.c), which is linked into the same library.
//
// Vector lanes are haplotypes: one read against 16 haplotypes at a time, column by column, so a
// row's transition constants are broadcast scalars and the working set (three arrays of R + 1
// vectors) stays in L1.  OpenMP spreads reads over the cores.  The hot function
// (pairhmm_simd_block.cpp) is built for AVX-512, AVX2 and plain x86-64; the CPU it runs on picks one.
#include <omp.h>
#include <xmmintrin.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
// from pairhmm_oracle.c
void oracle_init(void);
double oracle_qualToProb(uint8_t q);
double oracle_qualToErrorProb(uint8_t q);
int oracle_qualToTransProbs(double* dest, uint8_t insQual, uint8_t delQual, uint8_t gcp);
}

#include "pairhmm_simd.h"

namespace {

constexpr int W = kSimdLanes;

struct ReadConsts {
    std::vector<float> t;        // 6 per row, rows 1..R
    std::vector<float> pm, pmm;  // prior of a match / a mismatch per row
    std::vector<float> pmm_n;    // pmm, with pm in the rows whose read base is 'N'
    std::vector<float> xs;       // read bases as floats
    std::vector<double> td, pmd, pmmd;
};


// fp64 redo of one pair, same recurrence, scale 2^1020.
double forward_pair_fp64(const uint8_t* bases, int R, const ReadConsts& rc, const uint8_t* hap, int H, double* M, double* I, double* D) {
    const double initD = std::ldexp(1.0, 1020) / H;
    for (int i = 0; i <= R; ++i) M[i] = I[i] = D[i] = 0.0;
    double acc = 0.0;
    for (int j = 1; j <= H; ++j) {
        const int y = hap[j - 1];
        double diagM = 0, diagI = 0, diagD = initD, upM = 0, upI = 0;
        for (int i = 1; i <= R; ++i) {
            const double* t = &rc.td[static_cast<size_t>(i) * 6];
            const int x = bases[i - 1];
            const double prior = (x == y || x == 'N' || y == 'N') ? rc.pmd[i] : rc.pmmd[i];
            const double lM = M[i], lI = I[i], lD = D[i];
            const double m = prior * (diagM * t[matchToMatch] + (diagI + diagD) * t[indelToMatch]);
            const double ins = upM * t[matchToInsertion] + upI * t[insertionToInsertion];
            const double del = lM * t[matchToDeletion] + lD * t[deletionToDeletion];
            diagM = lM; diagI = lI; diagD = lD;
            upM = m; upI = ins;
            M[i] = m; I[i] = ins; D[i] = del;
        }
        acc += M[R] + I[R];
    }
    return std::log10(acc) - 1020.0 * std::log10(2.0);
}

}  // namespace

extern "C" {

// Which build of the hot loop this CPU runs.
const char* simd_pairhmm_isa(void) {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
        __builtin_cpu_supports("avx512vl"))
        return "avx512";
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return "avx2";
    return "generic";
}

// Same arguments and output layout as oracle_logless_batch_fast.  Returns 0, or 1 / 3 for the input
// errors the reference raises (negative ins/del/gcp byte; base quality 255).  reruns: pairs redone in fp64.
int simd_pairhmm_batch(int numReads, int numHaps, const uint8_t* readBases, const uint8_t* readQuals, const uint8_t* insGOP,
                       const uint8_t* delGOP, const uint8_t* gcp, const int32_t* readOffsets, const uint8_t* hapBases,
                       const int32_t* hapOffsets, double* out, int numThreads, int64_t* reruns) {
    oracle_init();
    const std::string isa = simd_pairhmm_isa();
    const simd_block_fn forward_block = isa == "avx512" ? simd_block_avx512 : isa == "avx2" ? simd_block_avx2 : simd_block_generic;
    if (numThreads < 1) numThreads = 1;
    int Rmax = 0;
    for (int r = 0; r < numReads; ++r) Rmax = std::max(Rmax, readOffsets[r + 1] - readOffsets[r]);
    // haplotype blocks, transposed once
    const int numBlocks = (numHaps + W - 1) / W;
    std::vector<std::vector<uint8_t>> hapT(numBlocks);
    std::vector<int32_t> hlen(static_cast<size_t>(numBlocks) * W, 0), hmax(numBlocks, 0);
    for (int b = 0; b < numBlocks; ++b) {
        for (int l = 0; l < W && b * W + l < numHaps; ++l) {
            hlen[b * W + l] = hapOffsets[b * W + l + 1] - hapOffsets[b * W + l];
            hmax[b] = std::max(hmax[b], hlen[b * W + l]);
        }
        hapT[b].assign(static_cast<size_t>(hmax[b]) * W, 0);
        for (int l = 0; l < W && b * W + l < numHaps; ++l)
            for (int j = 0; j < hlen[b * W + l]; ++j) hapT[b][static_cast<size_t>(j) * W + l] = hapBases[hapOffsets[b * W + l] + j];
    }
    int err = 0;
    int64_t redo = 0;
    const double log10_scale = 120.0 * std::log10(2.0);
#pragma omp parallel num_threads(numThreads) reduction(+ : redo)
    {
        // cells far from the alignment path decay into the fp32 denormal range, where x86 arithmetic takes
        // a microcode assist per operation: flush them (they are > 2^100 below the sums that count)
        const unsigned saved_csr = _mm_getcsr();
        _mm_setcsr(saved_csr | 0x8040u);  // FTZ | DAZ
        char* buf = static_cast<char*>(std::aligned_alloc(4 * W, static_cast<size_t>(4 * W) * 3 * (Rmax + 1)));
        std::vector<double> dbuf(3 * (Rmax + 1));
        ReadConsts rc;
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < numReads; ++r) {
            const int ro = readOffsets[r], R = readOffsets[r + 1] - ro;
            rc.t.assign(static_cast<size_t>(R + 1) * 6, 0.0f); rc.pm.assign(R + 1, 0.0f); rc.pmm.assign(R + 1, 0.0f); rc.pmm_n.assign(R + 1, 0.0f); rc.xs.assign(R + 1, 0.0f);
            rc.td.assign(static_cast<size_t>(R + 1) * 6, 0.0); rc.pmd.assign(R + 1, 0.0); rc.pmmd.assign(R + 1, 0.0);
            int bad = 0;
            for (int i = 0; i < R && !bad; ++i) {
                if (oracle_qualToTransProbs(&rc.td[static_cast<size_t>(i + 1) * 6], insGOP[ro + i], delGOP[ro + i], gcp[ro + i])) bad = 1;
                else if (readQuals[ro + i] > 254) bad = 3;
                else {
                    rc.pmd[i + 1] = oracle_qualToProb(readQuals[ro + i]);
                    rc.pmmd[i + 1] = oracle_qualToErrorProb(readQuals[ro + i]) / 3.0;
                    rc.pm[i + 1] = static_cast<float>(rc.pmd[i + 1]);
                    rc.pmm[i + 1] = static_cast<float>(rc.pmmd[i + 1]);
                    rc.xs[i + 1] = static_cast<float>(readBases[ro + i]);
                    rc.pmm_n[i + 1] = readBases[ro + i] == 'N' ? rc.pm[i + 1] : rc.pmm[i + 1];
                    for (int k = 0; k < 6; ++k) rc.t[static_cast<size_t>(i + 1) * 6 + k] = static_cast<float>(rc.td[static_cast<size_t>(i + 1) * 6 + k]);
                }
            }
            if (bad) {
#pragma omp critical
                err = bad;
                continue;
            }
            for (int b = 0; b < numBlocks; ++b) {
                float acc[W];
                const SimdReadConsts view{rc.t.data(), rc.pm.data(), rc.pmm_n.data(), rc.xs.data()};
                if (R > 0) forward_block(R, &view, hapT[b].data(), &hlen[b * W], hmax[b], buf, buf + static_cast<size_t>(4 * W) * (Rmax + 1), buf + static_cast<size_t>(8 * W) * (Rmax + 1), acc);
                for (int l = 0; l < W && b * W + l < numHaps; ++l) {
                    const int a = b * W + l;
                    double res;
                    if (R == 0) res = -INFINITY;  // the last row is row 0: log10(0)
                    else if (acc[l] >= 1e-28f && acc[l] <= 3.0e38f) res = std::log10(static_cast<double>(acc[l])) - log10_scale;
                    else {
                        res = forward_pair_fp64(readBases + ro, R, rc, hapBases + hapOffsets[a], hlen[a], dbuf.data(), dbuf.data() + (Rmax + 1), dbuf.data() + 2 * (Rmax + 1));
                        ++redo;
                    }
                    out[static_cast<size_t>(r) * numHaps + a] = res;
                }
            }
        }
        std::free(buf);
        _mm_setcsr(saved_csr);
    }
    if (reruns) *reruns = redo;
    return err;
}

}  // extern "C"
