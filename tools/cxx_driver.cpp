// Exercises the C++ host mirror (include/gatk_pairhmm.hpp) end to end on GPU 0: the closed-form
// single-base answers, the read-major array, an error path and the BatchPairHMM interface.
// Prints "cxx_driver ok" and exits 0 when everything matches.
#include <cmath>
#include <cstdio>

#include "gatk_pairhmm.hpp"

static gatk::Bytes B(const char* s) { return gatk::Bytes(s, s + std::char_traits<char>::length(s)); }

int main() {
    try {
        gatk::HipLoglessPairHMM hmm(0);
        hmm.initialize(4, 8);
        gatk::LikelihoodMatrix m({B("A"), B("C"), B("AC")}, 2);
        std::vector<gatk::Read> reads = {{B("A"), {30}, {45}, {45}}, {B("C"), {30}, {45}, {45}}};
        hmm.computeLikelihoods(m, reads, {{10}, {10}});
        const double want[3] = {-0.0461920023, -3.5228787453, -0.3470771124};
        for (int a = 0; a < 3; ++a)
            if (std::fabs(m.get(a, 0) - want[a]) > 1e-6) { std::printf("mismatch allele %d: %.10f\n", a, m.get(a, 0)); return 1; }
        if (std::fabs(m.get(1, 1) - want[0]) > 1e-6 || std::fabs(m.get(0, 1) - want[1]) > 1e-6) { std::printf("read 1 mismatch\n"); return 1; }
        const auto& flat = hmm.getLikelihoodArray();  // read-major
        if (flat.size() != 6 || flat[3] != m.get(0, 1) || flat[2] != m.get(2, 0)) { std::printf("array order mismatch\n"); return 1; }
        bool threw = false;
        try {
            std::vector<gatk::Read> bad = {{B("A"), {30}, {200}, {45}}};
            gatk::LikelihoodMatrix m2({B("A")}, 1);
            hmm.computeLikelihoods(m2, bad, {{10}});
        } catch (const gatk::IllegalArgumentException&) { threw = true; }
        if (!threw) { std::printf("negative insertion quality was accepted\n"); return 1; }
        hmm.batchAdd({B("A"), B("AC")}, B("A"), {30}, {45}, {45}, {10});
        const auto res = hmm.batchGetResult();
        if (res.size() != 2 || std::fabs(res[0] - want[0]) > 1e-6 || std::fabs(res[1] - want[2]) > 1e-6) { std::printf("batch mismatch\n"); return 1; }
        hmm.finalizeRegion();
        hmm.close();
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf("cxx_driver ok\n");
    return 0;
}
