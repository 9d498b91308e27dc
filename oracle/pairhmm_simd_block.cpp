// Hot loop of the vectorised CPU baseline (see pairhmm_simd.cpp).  TEST / BENCHMARK INFRASTRUCTURE.
// Compiled once per instruction set (-march=skylake-avx512, -march=haswell, plain x86-64) with
// SIMD_BLOCK_NAME set to a different symbol each time; pairhmm_simd.cpp picks one at run time.
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "pairhmm_simd.h"

namespace {
constexpr int W = kSimdLanes;
typedef float vf __attribute__((vector_size(W * 4)));
typedef int32_t vi __attribute__((vector_size(W * 4)));
#define splat(x) (vf{} + static_cast<float>(x))
}  // namespace

// One read against W haplotypes (hapT[j * W + lane], zero padded; hlen[lane] = 0 for an unused lane).
// Returns the scaled sums in acc[lane].
extern "C" void SIMD_BLOCK_NAME(int R, const SimdReadConsts* rcp, const uint8_t* hapT, const int32_t* hlen, int Hmax, void* Mv, void* Iv, void* Dv, float* acc_out) {
    const SimdReadConsts& rc = *rcp;
    vf* M = static_cast<vf*>(Mv);
    vf* I = static_cast<vf*>(Iv);
    vf* D = static_cast<vf*>(Dv);
    vf initD, hl;
    for (int l = 0; l < W; ++l) {
        initD[l] = hlen[l] > 0 ? std::ldexp(1.0f, 120) / static_cast<float>(hlen[l]) : 0.0f;
        hl[l] = static_cast<float>(hlen[l]);
    }
    const vf zero = splat(0.0f);
    for (int i = 0; i <= R; ++i) { M[i] = zero; I[i] = zero; D[i] = zero; }
    const float* const tr = rc.t;
    const float* const pm = rc.pm;
    const float* const xs = rc.xs;
    // a read 'N' matches everything: give its row the match prior on both sides
    const float* const pmm = rc.pmm_n;
    vf acc = zero;
    for (int j = 1; j <= Hmax; ++j) {
        // haplotype bases as floats: the comparison with the read base is then one vcmpps
        vf y;
        for (int l = 0; l < W; ++l) y[l] = static_cast<float>(hapT[static_cast<size_t>(j - 1) * W + l]);
        const vi y_is_n = y == splat('N');
        vf diagM = zero, diagI = zero, diagD = initD;  // row 0 of column j - 1
        vf upM = zero, upI = zero;                     // row 0 of column j
        for (int i = 1; i <= R; ++i) {
            const float* t = &tr[static_cast<size_t>(i) * 6];
            const float xf = xs[i];  // read base, or 'N' replaced by a value that matches through pm == pmm
            const vi match = (y == splat(xf)) | y_is_n;
            const vf prior = match ? splat(pm[i]) : splat(pmm[i]);
            const vf leftM = M[i], leftI = I[i], leftD = D[i];
            const vf m = prior * (diagM * t[matchToMatch] + (diagI + diagD) * t[indelToMatch]);
            const vf ins = upM * t[matchToInsertion] + upI * t[insertionToInsertion];
            const vf del = leftM * t[matchToDeletion] + leftD * t[deletionToDeletion];
            diagM = leftM; diagI = leftI; diagD = leftD;
            upM = m; upI = ins;
            M[i] = m; I[i] = ins; D[i] = del;
        }
        const vi inside = splat(static_cast<float>(j)) <= hl;
        acc += inside ? (M[R] + I[R]) : zero;
    }
    for (int l = 0; l < W; ++l) acc_out[l] = acc[l];
}

