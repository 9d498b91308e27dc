// Host-side construction of the Phred look-up tables the kernels read.
//
// The values follow the reference's static initialisers bit for bit in IEEE double:
//   errProb[q]      = 10^(q / -10)                     QualityUtils.java:72-77,153-156
//   matchToMatch    = 10^( log1p(-min(1, 10^s)) / ln 10 ), s = Jacobian-table log10 sum of
//                     (-0.1*ins, -0.1*del)             pairhmm/PairHMMModel.java:114-122
//   Jacobian table  = log10(1 + 10^(-k * 1e-4)), k = 0..80000, looked up at
//                     fastRound(diff * 10000); the smaller term is dropped when the
//                     difference is >= 8.0             MathUtils.java:124-154,209-231
// so that e.g. matchToMatch(1, 81) keeps the reference's dropped 10^-8.1 term rather than
// the closed form 1 - 10^-0.1 - 10^-8.1.
#include "pairhmm_tables.h"

#include <cmath>
#include <mutex>
#include <vector>

namespace pairhmm_hip {

namespace {

struct Jacobian {
    static constexpr double kMaxTolerance = 8.0;
    static constexpr double kTableStep = 0.0001;
    double inv_step;
    std::vector<double> cache;
    Jacobian() {
        inv_step = 1.0 / kTableStep;
        const int n = static_cast<int>(kMaxTolerance / kTableStep) + 1;
        cache.resize(n);
        for (int k = 0; k < n; ++k) cache[k] = std::log10(1.0 + std::pow(10.0, -static_cast<double>(k) * kTableStep));
    }
    static int fast_round(double d) { return d > 0.0 ? static_cast<int>(d + 0.5) : static_cast<int>(d - 0.5); }
    double log10_sum(double small, double big) const {
        if (small > big) std::swap(small, big);
        if (std::isinf(small) && small < 0) return big;
        if (std::isinf(big) && big < 0) return big;
        const double diff = big - small;
        if (diff >= kMaxTolerance) return big;
        return big + cache[fast_round(diff * inv_step)];
    }
};

Tables build() {
    Tables t;
    for (int q = 0; q <= kMaxQual; ++q) t.err_prob[q] = std::pow(10.0, static_cast<double>(q) / -10.0);
    t.err_prob[255] = 0.0;  // never used: a 255 quality is rejected before any kernel reads the table
    const Jacobian jac;
    const double inv_ln10 = 1.0 / std::log(10.0);
    t.match_to_match.assign(static_cast<size_t>(kGopLimit) * kGopLimit, 0.0);
    // Insertion/deletion GOPs reach the table only when < 128 (PairHMMModel.java:136-138), so a
    // square 128x128 table indexed [ins][del] holds every entry of the reference's triangular one
    // that the path can touch; it is symmetric because the reference orders the pair by min/max.
    for (int mx = 0; mx < kGopLimit; ++mx)
        for (int mn = 0; mn <= mx; ++mn) {
            const double s = jac.log10_sum(-0.1 * mx, -0.1 * mn);
            const double as_log10 = std::log1p(-std::fmin(1.0, std::pow(10.0, s))) * inv_ln10;
            const double v = std::pow(10.0, as_log10);
            t.match_to_match[static_cast<size_t>(mx) * kGopLimit + mn] = v;
            t.match_to_match[static_cast<size_t>(mn) * kGopLimit + mx] = v;
        }
    return t;
}

}  // namespace

const Tables& tables() {
    static const Tables t = build();
    return t;
}

}  // namespace pairhmm_hip
