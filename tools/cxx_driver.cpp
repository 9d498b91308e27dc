// Exercises the C++ host mirrors (include/gatk_pairhmm.hpp, include/gatk_smithwaterman.hpp) end to end on
// GPU 0: the closed-form single-base answers, the read-major array, an error path, the BatchPairHMM
// interface, the region-level hook and submit / fetch, the ReadLikelihoods steps on resident results, three Smith-Waterman cases and a multi-device instance.
// Prints "cxx_driver ok" and exits 0 when everything matches.
#include <cmath>
#include <cstdio>
#include <map>
#include <string>

#include "gatk_pairhmm.hpp"
#include "gatk_smithwaterman.hpp"

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
        {
            // the region-level hook: two samples sent ahead, then one computeLikelihoods call per sample (served from the
            // region: the stats count the hits), and a call in two halves through submit / fetch
            const std::vector<gatk::Bytes> alleles = {B("A"), B("C"), B("AC")};
            const std::map<std::string, std::vector<gatk::Read>> perSample = {{"s1", {reads[0]}}, {"s2", {reads[1], reads[0]}}};
            hmm.initialize(alleles, perSample, 4, 8);
            gatk::LikelihoodMatrix s1(alleles, 1), s2(alleles, 2);
            hmm.computeLikelihoods(s1, perSample.at("s1"), {{10}});
            hmm.computeLikelihoods(s2, perSample.at("s2"), {{10}, {10}});
            gatk_pairhmm_stats st{};
            gatk_pairhmm_get_stats(hmm.handle(), &st);
            hmm.finalizeRegion();
            for (int a = 0; a < 3; ++a)
                if (s1.get(a, 0) != m.get(a, 0) || s2.get(a, 0) != m.get(a, 1) || s2.get(a, 1) != m.get(a, 0)) { std::printf("region hook mismatch\n"); return 1; }
            if (st.region_hits != 2) { std::printf("region hook: %lld hits\n", static_cast<long long>(st.region_hits)); return 1; }
            gatk::LikelihoodMatrix later(alleles, 2);
            auto ticket = hmm.submit(later, reads, {{10}, {10}});
            hmm.fetch(*ticket);
            for (int a = 0; a < 3; ++a)
                if (later.get(a, 0) != m.get(a, 0) || later.get(a, 1) != m.get(a, 1)) { std::printf("submit / fetch mismatch\n"); return 1; }
        }
        {
            // the steps behind the path on resident results: alleles A (reference), C, AC against reads A and C
            const std::vector<gatk::Bytes> alleles = {B("A"), B("C"), B("AC")};
            gatk::ResidentReadLikelihoods rl(hmm, alleles, reads, {{10}, {10}}, 0);
            const auto before = rl.matrix(alleles);
            for (int a = 0; a < 3; ++a)
                if (before.get(a, 0) != m.get(a, 0) || before.get(a, 1) != m.get(a, 1)) { std::printf("resident matrix differs\n"); return 1; }
            const auto merged = rl.marginalize({0, 1, 1}, {B("A"), B("C")});
            if (merged.get(0, 0) != m.get(0, 0) || merged.get(1, 0) != std::fmax(m.get(1, 0), m.get(2, 0))) { std::printf("marginalize mismatch\n"); return 1; }
            const auto poorly = rl.poorlyModeledReads(0.02);  // one-base reads: threshold -4
            if (poorly[0] || poorly[1]) { std::printf("poorly-modelled flags mismatch\n"); return 1; }
            rl.normalizeLikelihoods(true, -1.0);
            const auto after = rl.matrix(alleles);
            // read 0: best alternative is AC (-0.347): cap -1.347 lifts C (-3.52); best overall A -> 0
            if (after.get(0, 0) != 0.0 || std::fabs(after.get(1, 0) - (m.get(2, 0) - 1.0 - m.get(0, 0))) > 1e-12 ||
                std::fabs(after.get(2, 0) - (m.get(2, 0) - m.get(0, 0))) > 1e-12) { std::printf("normalize mismatch\n"); return 1; }
        }
        {
            const gatk::SWPairwiseAlignment del(B("AAAAACCCCCGGGGG"), B("AAAAAGGGGG"));
            if (del.getCigar() != "5M5D5M" || del.getAlignmentStart2wrt1() != 0) { std::printf("SW deletion case: %s\n", del.getCigar().c_str()); return 1; }
            const gatk::SWPairwiseAlignment inner(B("TTTTACGTACGTTTTT"), B("ACGTACGT"));
            if (inner.getCigar() != "8M" || inner.getAlignmentStart2wrt1() != 4) { std::printf("SW offset case\n"); return 1; }
            const gatk::SWPairwiseAlignment over(B("ACGTACGT"), B("GGACGTACGTGG"), gatk::STANDARD_NGS, gatk::OverhangStrategy::INDEL);
            if (over.getCigar() != "2I8M2I") { std::printf("SW overhang case: %s\n", over.getCigar().c_str()); return 1; }
        }
        hmm.close();
        {
            // one instance over two engines (device 0 listed twice), without the tristate correction:
            // read A (Q30) against haplotype C scores log10(0.001 * 0.9)
            gatk::HipLoglessPairHMM multi(std::vector<int32_t>{0, 0});
            multi.doNotUseTristateCorrection();
            gatk::LikelihoodMatrix m3({B("A"), B("C")}, 1);
            multi.computeLikelihoods(m3, {{B("A"), {30}, {45}, {45}}}, {{10}});
            if (std::fabs(m3.get(0, 0) - want[0]) > 1e-6 || std::fabs(m3.get(1, 0) - (-3.0457574906)) > 1e-6) { std::printf("multi-device / tristate mismatch: %.10f\n", m3.get(1, 0)); return 1; }
        }
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf("cxx_driver ok\n");
    return 0;
}
