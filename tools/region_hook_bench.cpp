// Multi-sample regions through the region-level hook: gatk_pairhmm_initialize_region (PairHMM.initialize(haplotypes,
// perSampleReadList, ...), pairhmm/PairHMM.java:129-139) followed by one gatk_pairhmm_compute_likelihoods per sample
// and gatk_pairhmm_finalize_region, against the same per-sample calls without the hook.  The reads of a synthetic
// HaplotypeCaller-sized region (64 reads of 121-151 bases, 8 haplotypes of 250-500 bases) are dealt out to S samples,
// so every S scores the same pairs.  usage: region_hook_bench REGIONS
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/gatk_pairhmm.h"

struct Rng {
    uint64_t s;
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return static_cast<uint32_t>(s >> 33); }
    int range(int lo, int hi) { return lo + static_cast<int>(next() % static_cast<uint32_t>(hi - lo + 1)); }
};

struct Region {
    std::vector<uint8_t> rb, rq, hb;
    std::vector<int32_t> roff{0}, hoff{0};
    int nr = 0, nh = 0;
};

int main(int argc, char** argv) {
    const int num_regions = argc > 1 ? std::atoi(argv[1]) : 500;
    Rng g{7};
    std::vector<Region> regions(num_regions);
    for (Region& r : regions) {
        const int span = g.range(250, 500);
        r.nh = 8; r.nr = 64;
        std::vector<uint8_t> ref(span);
        for (auto& b : ref) b = "ACGT"[g.range(0, 3)];
        for (int a = 0; a < r.nh; ++a) {
            std::vector<uint8_t> h = ref;
            for (int e = g.range(0, 3); e > 0; --e) h[g.range(0, span - 1)] = "ACGT"[g.range(0, 3)];
            r.hb.insert(r.hb.end(), h.begin(), h.end());
            r.hoff.push_back(static_cast<int32_t>(r.hb.size()));
        }
        for (int k = 0; k < r.nr; ++k) {
            const int len = g.range(121, 151), start = g.range(0, span - len);
            for (int i = 0; i < len; ++i) {
                r.rb.push_back(g.range(0, 99) == 0 ? "ACGT"[g.range(0, 3)] : ref[start + i]);
                r.rq.push_back(static_cast<uint8_t>(g.range(10, 40)));
            }
            r.roff.push_back(static_cast<int32_t>(r.rb.size()));
        }
    }
    gatk_pairhmm* h = nullptr;
    if (gatk_pairhmm_create(0, &h)) { std::printf("%s\n", gatk_pairhmm_last_error()); return 1; }
    std::vector<double> out(64 * 8);
    double base_us = 0;
    for (int samples : {1, 2, 4, 8, 16}) {
        for (int hook = 1; hook >= 0; --hook) {
            double best = 1e30, checksum = 0;
            for (int rep = 0; rep < 4; ++rep) {
                checksum = 0;
                const auto t0 = std::chrono::steady_clock::now();
                for (const Region& r : regions) {
                    std::vector<gatk_pairhmm_batch> per_sample(samples);
                    for (int s = 0; s < samples; ++s) {
                        const int r0 = r.nr * s / samples, r1 = r.nr * (s + 1) / samples;
                        per_sample[s] = gatk_pairhmm_batch{r1 - r0, r.nh, r.rb.data(), r.rq.data(), nullptr, nullptr, nullptr, r.roff.data() + r0,
                                                           r.hb.data(), r.hoff.data(), nullptr};
                    }
                    if (hook && gatk_pairhmm_initialize_region(h, samples, per_sample.data(), 151, 500)) { std::printf("%s\n", gatk_pairhmm_last_error()); return 1; }
                    for (int s = 0; s < samples; ++s) {
                        if (gatk_pairhmm_compute_likelihoods(h, &per_sample[s], out.data())) { std::printf("%s\n", gatk_pairhmm_last_error()); return 1; }
                        for (int k = 0; k < per_sample[s].num_reads * r.nh; ++k) checksum += out[k];
                    }
                    gatk_pairhmm_finalize_region(h);
                }
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rep > 0 && s < best) best = s;
            }
            const double us = best * 1e6 / num_regions;
            if (samples == 1 && hook) base_us = us;
            gatk_pairhmm_stats st{};
            gatk_pairhmm_get_stats(h, &st);
            std::printf("%2d samples per region, %s: %7.1f us per region (%.2f x the one-sample region with the hook), checksum %.4f, hits %lld misses %lld\n",
                        samples, hook ? "sent ahead with initialize_region" : "one call per sample, no hook      ", us, us / base_us, checksum,
                        static_cast<long long>(st.region_hits), static_cast<long long>(st.region_misses));
        }
    }
    gatk_pairhmm_close(h);
    return 0;
}
