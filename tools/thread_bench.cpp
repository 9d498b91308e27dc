// Region-at-a-time calls from T host threads, one handle per thread (how the reference's map threads would
// drive a PairHMM, nanoScheduler/NanoScheduler.java:112 and pairhmm/PairHMM.java:297), with and without the
// handle option "combine_calls".  Synthetic HaplotypeCaller-sized regions: 30-100 reads of 121-151 bases,
// 2-16 haplotypes of 250-500 bases.  usage: thread_bench REGIONS THREADS COMBINE [OPTION VALUE ...]
// COMBINE: 0 = blocking calls, each on its own handle; 1 = blocking calls that share the device's dispatcher
// ("combine_calls"); -D = gatk_pairhmm_submit / gatk_pairhmm_fetch with D regions in flight per thread.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
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
    std::vector<double> out;
    gatk_pairhmm_batch c{};
    int64_t cells = 0;
};

int main(int argc, char** argv) {
    const int num_regions = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int threads = argc > 2 ? std::atoi(argv[2]) : 4;
    const int combine = argc > 3 ? std::atoi(argv[3]) : 0;
    Rng g{99};
    std::vector<Region> regions(num_regions);
    int64_t cells = 0;
    for (Region& r : regions) {
        const int span = g.range(250, 500), nh = g.range(2, 16), nr = g.range(30, 100);
        std::vector<uint8_t> ref(span);
        for (auto& b : ref) b = "ACGT"[g.range(0, 3)];
        int64_t hap_total = 0;
        for (int a = 0; a < nh; ++a) {
            std::vector<uint8_t> h = ref;
            for (int e = g.range(0, 3); e > 0; --e) h[g.range(0, span - 1)] = "ACGT"[g.range(0, 3)];
            r.hb.insert(r.hb.end(), h.begin(), h.end());
            r.hoff.push_back(static_cast<int32_t>(r.hb.size()));
            hap_total += span;
        }
        for (int k = 0; k < nr; ++k) {
            const int len = g.range(121, 151), start = g.range(0, span - len);
            for (int i = 0; i < len; ++i) {
                r.rb.push_back(g.range(0, 99) == 0 ? "ACGT"[g.range(0, 3)] : ref[start + i]);
                r.rq.push_back(static_cast<uint8_t>(g.range(10, 40)));
            }
            r.roff.push_back(static_cast<int32_t>(r.rb.size()));
            r.cells += static_cast<int64_t>(len) * hap_total;
        }
        r.out.resize(static_cast<size_t>(nr) * nh);
        r.c = gatk_pairhmm_batch{nr, nh, r.rb.data(), r.rq.data(), nullptr, nullptr, nullptr, r.roff.data(), r.hb.data(), r.hoff.data(), nullptr};
        cells += r.cells;
    }
    std::vector<gatk_pairhmm*> handles(threads, nullptr);
    for (auto& h : handles) {
        if (gatk_pairhmm_create(0, &h)) { std::printf("%s\n", gatk_pairhmm_last_error()); return 1; }
        gatk_pairhmm_set_option(h, "combine_calls", combine > 0 ? combine : 0);
        for (int a = 4; a + 1 < argc; a += 2) gatk_pairhmm_set_option(h, argv[a], std::atof(argv[a + 1]));  // further engine options: NAME VALUE ...
    }
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {  // the first repetition warms buffers and clocks
        std::atomic<int> next{0}, failed{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                if (combine >= 0) {
                    for (int k; (k = next.fetch_add(1)) < num_regions;)
                        if (gatk_pairhmm_compute_likelihoods(handles[t], &regions[k].c, regions[k].out.data())) failed.store(1);
                    return;
                }
                // submit / fetch: -combine regions in flight, the oldest fetched before the next goes out
                std::vector<int64_t> ring(static_cast<size_t>(-combine), 0);
                size_t head = 0, live = 0;
                for (int k; (k = next.fetch_add(1)) < num_regions;) {
                    if (live == ring.size()) {
                        if (gatk_pairhmm_fetch(handles[t], ring[head])) failed.store(1);
                        head = (head + 1) % ring.size(); --live;
                    }
                    double* out = regions[k].out.data();
                    int64_t ticket = 0;
                    if (gatk_pairhmm_submit(handles[t], 1, &regions[k].c, &out, &ticket)) { failed.store(1); continue; }
                    ring[(head + live) % ring.size()] = ticket; ++live;
                }
                for (; live > 0; --live, head = (head + 1) % ring.size())
                    if (gatk_pairhmm_fetch(handles[t], ring[head])) failed.store(1);
            });
        for (auto& th : pool) th.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (failed.load()) { std::printf("a call failed\n"); return 1; }
        if (rep > 0 && s < best) best = s;
    }
    double sum = 0;
    for (const Region& r : regions) for (double v : r.out) sum += v;
    gatk_pairhmm_stats st{};
    gatk_pairhmm_get_stats(handles[0], &st);
    if (st.shared_passes) std::printf("(%.2f requests per shared pass) ", double(st.shared_requests) / double(st.shared_passes));
    std::printf("regions %d threads %d combine_calls %d: %.2f ms, %.1f us per region, %.0f GCUPS (checksum %.6f)\n", num_regions, threads, combine,
                best * 1e3, best * 1e6 / num_regions, cells / best / 1e9, sum);
    for (auto h : handles) gatk_pairhmm_close(h);
    return 0;
}
