// Stress driver for the C-ABI's host code, meant to be built with the host side under AddressSanitizer and
// UndefinedBehaviorSanitizer (tools/build_stress_asan.sh) and run on a GPU.  It pushes randomly shaped region
// batches (empty regions, zero-length reads, missing optional arrays, 'N' bases, mapping-quality caps, a few
// reads that underflow fp32) through every entry point and checks that differently planned engine
// configurations agree: automatic planning against forced rows per lane with one haplotype per work item,
// one big call against region-at-a-time calls and against calls of a few dozen regions on a default handle,
// host-pointer calls against the resident path, a multi-device handle (three engines on GPU 0) against one engine
// for calls of many regions, few regions and one region cut by read ranges, and eight threads whose one-region
// calls are combined into shared passes, then ten threads on submit / fetch and the region-level hook with a failing
// request among them.  Last, Smith-Waterman calls of random sizes (a handful of pairs, thousands of
// them with a sequence pool large enough to go up launch by launch, weights on both sides of the packed cell form's
// limits, alignments with more elements than come back with the first copy): an alignment's elements must add up to its
// sequences.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../include/gatk_pairhmm.h"
#include "../include/gatk_smithwaterman.h"

namespace {

struct Rng {
    uint64_t s;
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return static_cast<uint32_t>(s >> 33); }
    int range(int lo, int hi) { return lo + static_cast<int>(next() % static_cast<uint32_t>(hi - lo + 1)); }  // inclusive
};

struct Region {
    std::vector<uint8_t> rb, rq, ri, rd, rg, hb, mapq;
    std::vector<int32_t> roff{0}, hoff{0};
    bool with_ins, with_del, with_gcp, with_mapq;
    gatk_pairhmm_batch c{};
    void finish() {
        c.num_reads = static_cast<int32_t>(roff.size()) - 1;
        c.num_haps = static_cast<int32_t>(hoff.size()) - 1;
        c.read_bases = rb.data(); c.read_quals = rq.data();
        c.ins_gop = with_ins ? ri.data() : nullptr;
        c.del_gop = with_del ? rd.data() : nullptr;
        c.gcp = with_gcp ? rg.data() : nullptr;
        c.read_offsets = roff.data();
        c.hap_bases = hb.data(); c.hap_offsets = hoff.data();
        c.read_mapq = with_mapq ? mapq.data() : nullptr;
    }
};

Region make_region(Rng& g, bool allow_empty) {
    static const char letters[] = "ACGT";
    Region r;
    r.with_ins = g.range(0, 3) != 0; r.with_del = g.range(0, 3) != 0; r.with_gcp = g.range(0, 3) != 0; r.with_mapq = g.range(0, 4) == 0;
    const bool long_region = g.range(0, 39) == 0;  // a region with reads of more than 1022 bases: the long-read pass
    const int span = long_region ? g.range(1100, 1800) : g.range(30, 500);
    std::vector<uint8_t> ref(span);
    for (auto& b : ref) b = letters[g.range(0, 3)];
    const int nh = g.range(1, 20);
    for (int a = 0; a < nh; ++a) {
        std::vector<uint8_t> h = ref;
        for (int e = g.range(0, 4); e > 0; --e) {
            const int p = g.range(0, static_cast<int>(h.size()) - 1);
            switch (g.range(0, 3)) {
                case 0: h[p] = letters[g.range(0, 3)]; break;
                case 1: h.insert(h.begin() + p, letters[g.range(0, 3)]); break;
                case 2: if (h.size() > 12) h.erase(h.begin() + p); break;
                default: if (g.range(0, 9) == 0) h[p] = 'N';
            }
        }
        r.hb.insert(r.hb.end(), h.begin(), h.end());
        r.hoff.push_back(static_cast<int32_t>(r.hb.size()));
    }
    const int nr = allow_empty && g.range(0, 15) == 0 ? 0 : g.range(1, 90);
    for (int k = 0; k < nr; ++k) {
        // a region whose reads are all empty is the reference's "READ_MAX_LENGTH must be > 0": keep read 0 non-empty
        int len = (k > 0 && g.range(0, 30) == 0) ? 0 : g.range(1, std::min(span, g.range(0, 5) == 0 ? 260 : 151));
        if (long_region && k % 16 == 1) len = g.range(1023, span);
        const int start = g.range(0, span - std::max(len, 1));
        const bool noisy = g.range(0, 25) == 0;  // many errors at high quality: the fp32 sum underflows
        for (int i = 0; i < len; ++i) {
            uint8_t b = ref[start + i];
            if (g.range(0, noisy ? 2 : 60) == 0) b = letters[g.range(0, 3)];
            if (g.range(0, 400) == 0) b = 'N';
            r.rb.push_back(b);
            r.rq.push_back(static_cast<uint8_t>(noisy ? g.range(35, 41) : g.range(2, 41)));
            r.ri.push_back(static_cast<uint8_t>(g.range(0, 8) == 0 ? g.range(1, 45) : 45));
            r.rd.push_back(static_cast<uint8_t>(g.range(0, 8) == 0 ? g.range(1, 45) : 45));
            r.rg.push_back(static_cast<uint8_t>(g.range(0, 8) == 0 ? g.range(5, 20) : 10));
        }
        r.roff.push_back(static_cast<int32_t>(r.rb.size()));
        r.mapq.push_back(static_cast<uint8_t>(g.range(10, 60)));
    }
    r.finish();
    return r;
}

#define CHECK(call)                                                                          \
    do {                                                                                     \
        const int rc_ = (call);                                                              \
        if (rc_ != 0) { std::printf("%s -> %d: %s\n", #call, rc_, gatk_pairhmm_last_error()); return 1; } \
    } while (0)

double worst_difference(const std::vector<std::vector<double>>& a, const std::vector<std::vector<double>>& b) {
    double worst = 0.0;
    for (size_t k = 0; k < a.size(); ++k)
        for (size_t i = 0; i < a[k].size(); ++i) {
            if (std::isinf(a[k][i]) || std::isinf(b[k][i])) {
                if (a[k][i] != b[k][i]) return INFINITY;
                continue;
            }
            if (std::isnan(a[k][i]) || std::isnan(b[k][i])) return INFINITY;
            worst = std::fmax(worst, std::fabs(a[k][i] - b[k][i]));
        }
    return worst;
}

}  // namespace

int main(int argc, char** argv) {
    const int num_regions = argc > 1 ? std::atoi(argv[1]) : 400;
    const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 7;
    Rng g{seed};
    std::vector<Region> regions;
    regions.reserve(num_regions);
    for (int k = 0; k < num_regions; ++k) regions.push_back(make_region(g, true));
    for (auto& r : regions) r.finish();  // pointers into the vectors after the moves
    std::vector<gatk_pairhmm_batch> batches;
    int64_t pairs = 0;
    for (auto& r : regions) { batches.push_back(r.c); pairs += static_cast<int64_t>(r.c.num_reads) * r.c.num_haps; }
    auto outputs = [&]() {
        std::vector<std::vector<double>> o(regions.size());
        for (size_t k = 0; k < regions.size(); ++k) o[k].assign(static_cast<size_t>(regions[k].c.num_reads) * regions[k].c.num_haps, NAN);
        return o;
    };
    auto pointers = [](std::vector<std::vector<double>>& o) {
        std::vector<double*> p;
        for (auto& v : o) p.push_back(v.empty() ? nullptr : v.data());
        return p;
    };

    gatk_pairhmm* h = nullptr;
    CHECK(gatk_pairhmm_create(0, &h));
    CHECK(gatk_pairhmm_set_option(h, "pipeline_pairs", 20000));  // several parts: the pipelined path
    auto all = outputs();
    { auto p = pointers(all); CHECK(gatk_pairhmm_compute_regions(h, num_regions, batches.data(), p.data())); }

    // region at a time, one haplotype per work item, forced rows per lane
    gatk_pairhmm* h2 = nullptr;
    CHECK(gatk_pairhmm_create(0, &h2));
    CHECK(gatk_pairhmm_set_option(h2, "stream_mode", 0));
    CHECK(gatk_pairhmm_set_option(h2, "force_rows_per_lane", 5));
    auto single = outputs();
    for (int k = 0; k < num_regions; ++k)
        CHECK(gatk_pairhmm_compute_likelihoods(h2, &batches[k], single[k].empty() ? nullptr : single[k].data()));
    const double d1 = worst_difference(all, single);

    // resident path with everything in fp64
    gatk_pairhmm* h3 = nullptr;
    CHECK(gatk_pairhmm_create(0, &h3));
    CHECK(gatk_pairhmm_set_option(h3, "force_fp64", 1));
    gatk_pairhmm_resident* res = nullptr;
    CHECK(gatk_pairhmm_upload(h3, num_regions, batches.data(), &res));
    CHECK(gatk_pairhmm_compute_resident(h3, res, 0));
    CHECK(gatk_pairhmm_compute_resident(h3, res, 1));
    auto fp64 = outputs();
    { auto p = pointers(fp64); CHECK(gatk_pairhmm_download(h3, res, p.data())); }
    const double d2 = worst_difference(all, fp64);
    // the steps behind the path, on the resident results
    std::vector<std::vector<uint8_t>> flags(regions.size());
    std::vector<uint8_t*> flag_ptrs;
    for (size_t k = 0; k < regions.size(); ++k) { flags[k].assign(regions[k].c.num_reads, 2); flag_ptrs.push_back(flags[k].empty() ? nullptr : flags[k].data()); }
    CHECK(gatk_pairhmm_poorly_modeled_resident(h3, res, 0.02, flag_ptrs.data()));
    std::vector<std::vector<int32_t>> maps(regions.size());
    std::vector<const int32_t*> map_ptrs;
    std::vector<int32_t> new_counts;
    auto merged = outputs();
    for (size_t k = 0; k < regions.size(); ++k) {
        const int nh = regions[k].c.num_haps, n_new = std::max(1, nh / 2);
        for (int a = 0; a < nh; ++a) maps[k].push_back(g.range(-1, n_new - 1));
        map_ptrs.push_back(maps[k].data());
        new_counts.push_back(n_new);
        merged[k].assign(static_cast<size_t>(regions[k].c.num_reads) * n_new, NAN);
    }
    { auto p = pointers(merged); CHECK(gatk_pairhmm_marginalize_resident(h3, res, map_ptrs.data(), new_counts.data(), p.data())); }
    CHECK(gatk_pairhmm_normalize_resident(h3, res, nullptr, 1, -4.5));
    CHECK(gatk_pairhmm_resident_free(h3, res));

    // default options: one call over everything, then calls of 1 to 80 regions (one rows-per-lane value per
    // call, two overlapped halves from some ten thousand pairs on)
    gatk_pairhmm* h4 = nullptr;
    CHECK(gatk_pairhmm_create(0, &h4));
    auto dflt = outputs();
    { auto p = pointers(dflt); CHECK(gatk_pairhmm_compute_regions(h4, num_regions, batches.data(), p.data())); }
    const double d3 = worst_difference(all, dflt);
    auto mid = outputs();
    {
        auto p = pointers(mid);
        for (int at = 0; at < num_regions;) {
            const int n = std::min(num_regions - at, g.range(1, 80));
            CHECK(gatk_pairhmm_compute_regions(h4, n, batches.data() + at, p.data() + at));
            at += n;
        }
    }
    const double d4 = worst_difference(all, mid);
    CHECK(gatk_pairhmm_close(h4));
    if (!(d3 <= 2e-3) || !(d4 <= 2e-3)) { std::printf("default handle differs: %g %g\nFAILED\n", d3, d4); return 1; }

    // one handle over three engines on GPU 0: the whole list (regions dealt out by cells), calls of a few regions
    // (regions cut by read ranges), one region per call
    {
        gatk_pairhmm* hm = nullptr;
        const int32_t devices[3] = {0, 0, 0};
        CHECK(gatk_pairhmm_create_multi(devices, 3, &hm));
        CHECK(gatk_pairhmm_set_option(hm, "multi_min_pairs", 64));
        auto multi = outputs();
        { auto p = pointers(multi); CHECK(gatk_pairhmm_compute_regions(hm, num_regions, batches.data(), p.data())); }
        const double d5 = worst_difference(all, multi);
        auto few = outputs();
        {
            auto p = pointers(few);
            for (int at = 0; at < num_regions;) {
                const int n = std::min(num_regions - at, g.range(1, 4));
                CHECK(gatk_pairhmm_compute_regions(hm, n, batches.data() + at, p.data() + at));
                at += n;
            }
        }
        const double d6 = worst_difference(all, few);
        CHECK(gatk_pairhmm_close(hm));
        if (!(d5 <= 2e-3) || !(d6 <= 2e-3)) { std::printf("multi-device handle differs: %g %g\nFAILED\n", d5, d6); return 1; }
    }

    // eight caller threads, a handle each, one region per call, calls combined into shared passes
    {
        constexpr int kThreads = 8;
        auto combined = outputs();
        std::vector<int> failed(kThreads, 0);
        std::vector<std::thread> threads;
        for (int t = 0; t < kThreads; ++t)
            threads.emplace_back([&, t]() {
                gatk_pairhmm* hc = nullptr;
                if (gatk_pairhmm_create(0, &hc) != 0 || gatk_pairhmm_set_option(hc, "combine_calls", 1) != 0) { failed[t] = 1; return; }
                for (int k = t; k < num_regions; k += kThreads)
                    if (gatk_pairhmm_compute_likelihoods(hc, &batches[k], combined[k].empty() ? nullptr : combined[k].data()) != 0) {
                        std::printf("thread %d region %d: %s\n", t, k, gatk_pairhmm_last_error());
                        failed[t] = 1;
                        break;
                    }
                if (gatk_pairhmm_close(hc) != 0) failed[t] = 1;
            });
        for (auto& th : threads) th.join();
        for (int t = 0; t < kThreads; ++t)
            if (failed[t]) { std::printf("combined calls: thread %d failed\nFAILED\n", t); return 1; }
        const double d7 = worst_difference(all, combined);
        if (!(d7 <= 2e-3)) { std::printf("combined calls differ: %g\nFAILED\n", d7); return 1; }
    }

    // eight caller threads on submit / fetch with three regions in flight each, one of them submitting a request with a
    // negative gap quality now and then (which must fail alone), while two more threads send regions ahead through the
    // region-level hook in groups of up to five samples and ask for them sample by sample
    {
        constexpr int kThreads = 10;
        auto tickets_out = outputs();
        std::vector<int> failed(kThreads, 0);
        std::vector<std::thread> threads;
        std::vector<uint8_t> bad_bases = {'A', 'C', 'G', 'T'}, bad_quals = {30, 30, 30, 30}, bad_ins = {45, 200, 45, 45}, bad_hap = {'A', 'C', 'G', 'T', 'A'};
        const int32_t bad_roff[2] = {0, 4}, bad_hoff[2] = {0, 5};
        const gatk_pairhmm_batch bad{1, 1, bad_bases.data(), bad_quals.data(), bad_ins.data(), nullptr, nullptr, bad_roff, bad_hap.data(), bad_hoff, nullptr};
        for (int t = 0; t < kThreads; ++t)
            threads.emplace_back([&, t]() {
                gatk_pairhmm* hc = nullptr;
                if (gatk_pairhmm_create(0, &hc) != 0) { failed[t] = 1; return; }
                if (t < 8) {
                    std::vector<int64_t> ring;
                    double bad_out = 0.0;
                    for (int k = t; k < num_regions; k += kThreads) {
                        if (ring.size() == 3) {
                            if (gatk_pairhmm_fetch(hc, ring.front()) != 0) { std::printf("fetch: %s\n", gatk_pairhmm_last_error()); failed[t] = 1; }
                            ring.erase(ring.begin());
                        }
                        if (t == 3 && (k / kThreads) % 7 == 0) {
                            double* bo = &bad_out;
                            int64_t tb = 0;
                            if (gatk_pairhmm_submit(hc, 1, &bad, &bo, &tb) != 0 || gatk_pairhmm_fetch(hc, tb) != GATK_PAIRHMM_ILLEGAL_ARGUMENT) failed[t] = 1;
                        }
                        double* out = tickets_out[k].empty() ? nullptr : tickets_out[k].data();
                        int64_t ticket = 0;
                        if (gatk_pairhmm_submit(hc, 1, &batches[k], &out, &ticket) != 0) { std::printf("submit: %s\n", gatk_pairhmm_last_error()); failed[t] = 1; break; }
                        ring.push_back(ticket);
                    }
                    // the last tickets of every other thread are left to close()
                    if (t % 2 == 0) for (int64_t ticket : ring) if (gatk_pairhmm_fetch(hc, ticket) != 0) failed[t] = 1;
                    if (t % 2 == 1) for (size_t i = 0; i + 1 < ring.size(); ++i) if (gatk_pairhmm_fetch(hc, ring[i]) != 0) failed[t] = 1;
                } else {
                    int k = t;
                    while (k < num_regions) {
                        std::vector<int> mine;
                        for (int n = 1 + (k % 5); n > 0 && k < num_regions; --n, k += kThreads) mine.push_back(k);
                        std::vector<gatk_pairhmm_batch> ahead;
                        for (int r : mine) ahead.push_back(batches[r]);
                        if (gatk_pairhmm_initialize_region(hc, static_cast<int32_t>(ahead.size()), ahead.data(), 1022, 16000) != 0) { std::printf("initialize_region: %s\n", gatk_pairhmm_last_error()); failed[t] = 1; break; }
                        for (int r : mine)
                            if (gatk_pairhmm_compute_likelihoods(hc, &batches[r], tickets_out[r].empty() ? nullptr : tickets_out[r].data()) != 0) failed[t] = 1;
                        if (mine.size() % 2 == 0 && gatk_pairhmm_finalize_region(hc) != 0) failed[t] = 1;  // every other region is replaced without it
                    }
                }
                if (gatk_pairhmm_close(hc) != 0) failed[t] = 1;
            });
        for (auto& th : threads) th.join();
        for (int t = 0; t < kThreads; ++t)
            if (failed[t]) { std::printf("submit / fetch / region hook: thread %d failed\nFAILED\n", t); return 1; }
        // every other thread's last ticket was dropped at close: its array may hold anything, so compare the rest
        for (int t = 1; t < 8; t += 2) {
            int last = -1;
            for (int k = t; k < num_regions; k += kThreads) last = k;
            if (last >= 0) tickets_out[last] = all[last];
        }
        const double d8 = worst_difference(all, tickets_out);
        if (!(d8 <= 2e-3)) { std::printf("submit / fetch / region hook differ: %g\nFAILED\n", d8); return 1; }
    }

    // the queueing interface and the single-pair call on the first region that has a read
    for (auto& r : regions) {
        if (r.c.num_reads == 0 || r.roff[1] == 0 || !r.with_ins || !r.with_del || !r.with_gcp) continue;
        std::vector<const uint8_t*> hp;
        std::vector<int32_t> hl;
        for (int a = 0; a < r.c.num_haps; ++a) { hp.push_back(r.hb.data() + r.hoff[a]); hl.push_back(r.hoff[a + 1] - r.hoff[a]); }
        CHECK(gatk_pairhmm_batch_add(h, r.c.num_haps, hp.data(), hl.data(), r.rb.data(), r.rq.data(), r.ri.data(), r.rd.data(), r.rg.data(), r.roff[1]));
        const double* got = nullptr;
        int64_t count = 0;
        CHECK(gatk_pairhmm_batch_get_result(h, &got, &count));
        if (count != r.c.num_haps) { std::printf("batch_get_result count %lld\n", static_cast<long long>(count)); return 1; }
        double one = 0.0;
        CHECK(gatk_pairhmm_compute_pair(h, hp[0], hl[0], r.rb.data(), r.rq.data(), r.ri.data(), r.rd.data(), r.rg.data(), r.roff[1], &one));
        if (std::fabs(one - got[0]) > 1e-3) { std::printf("pair call differs from the queue: %g vs %g\n", one, got[0]); return 1; }
        break;
    }
    // Smith-Waterman
    for (int round = 0; round < 6; ++round) {
        const int n_pairs = round == 5 ? 3000 : g.range(1, 40);
        std::vector<std::vector<uint8_t>> refs(n_pairs), alts(n_pairs);
        std::vector<const uint8_t*> rp(n_pairs), ap(n_pairs);
        std::vector<int32_t> rl(n_pairs), al(n_pairs);
        for (int k = 0; k < n_pairs; ++k) {
            const int n = round == 5 ? g.range(200, 500) : g.range(1, g.range(0, 9) == 0 ? 2048 : 600);
            for (int i = 0; i < n; ++i) refs[k].push_back("ACGT"[g.range(0, 3)]);
            alts[k] = refs[k];
            const int indels = round == 4 ? 24 : g.range(0, 3);  // round 4: more elements than the first copy brings back
            for (int e = 0; e < indels && alts[k].size() > 40; ++e) {
                const size_t at = static_cast<size_t>(g.range(1, static_cast<int>(alts[k].size()) - 8));
                if (g.range(0, 1)) alts[k].erase(alts[k].begin() + at, alts[k].begin() + at + static_cast<size_t>(g.range(1, 6)));
                else alts[k].insert(alts[k].begin() + at, static_cast<size_t>(g.range(1, 6)), static_cast<uint8_t>('T'));
            }
            for (int e = g.range(0, 4); e > 0; --e) alts[k][static_cast<size_t>(g.range(0, static_cast<int>(alts[k].size()) - 1))] = "ACGT"[g.range(0, 3)];
            rp[k] = refs[k].data(); ap[k] = alts[k].data();
            rl[k] = static_cast<int32_t>(refs[k].size()); al[k] = static_cast<int32_t>(alts[k].size());
        }
        const gatk_sw_parameters sets[4] = {{3, -1, -4, -3}, {25, -50, -110, -6}, {200, -150, -260, -11}, {200000, -150000, -260000, -11000}};
        const gatk_sw_parameters par = sets[g.range(0, 3)];
        const int strategy = g.range(0, 3);
        const int cap = 128;
        std::vector<int32_t> ops(static_cast<size_t>(n_pairs) * cap), lens(ops.size()), counts(n_pairs), offs(n_pairs);
        const int rc = gatk_sw_align(0, n_pairs, rp.data(), rl.data(), ap.data(), al.data(), &par, strategy, cap, ops.data(), lens.data(), counts.data(), offs.data());
        if (rc != GATK_SW_OK) { std::printf("gatk_sw_align: %s\nFAILED\n", gatk_sw_last_error()); return 1; }
        for (int k = 0; k < n_pairs; ++k) {
            int64_t on_alt = 0, on_ref = 0;
            for (int e = 0; e < counts[k]; ++e) {
                const int op = ops[static_cast<size_t>(k) * cap + e], len = lens[static_cast<size_t>(k) * cap + e];
                if (op == GATK_SW_OP_M) { on_alt += len; on_ref += len; }
                else if (op == GATK_SW_OP_I || op == GATK_SW_OP_S) on_alt += len;
                else on_ref += len;
            }
            // the elements account for the whole alternate; the reference span fits the reference from the offset on
            // (IGNORE folds the unaligned head of the alternate into its first element, whatever that is, and its
            // offset may be negative: there only the presence of an alignment is checked)
            const bool ok = strategy == GATK_SW_IGNORE ? counts[k] >= 1 : (on_alt == al[k] && offs[k] >= 0 && offs[k] + on_ref <= rl[k]);
            if (!ok) { std::printf("Smith-Waterman pair %d of round %d: elements cover %lld alternate / %lld reference bases of %d / %d, offset %d\nFAILED\n", k, round, static_cast<long long>(on_alt), static_cast<long long>(on_ref), al[k], rl[k], offs[k]); return 1; }
        }
    }

    CHECK(gatk_pairhmm_close(h));
    CHECK(gatk_pairhmm_close(h2));
    CHECK(gatk_pairhmm_close(h3));
    std::printf("regions %d pairs %lld: pipelined vs region-at-a-time max |d| %.3g, vs fp64 %.3g\n", num_regions, static_cast<long long>(pairs), d1, d2);
    if (!(d1 <= 2e-3) || !(d2 <= 1e-3)) { std::printf("FAILED\n"); return 1; }
    std::printf("ok\n");
    // leave without the exit handlers: under AddressSanitizer the HSA runtime's own teardown can trip a check in the
    // sanitizer's allocator after everything here has been released
    std::fflush(stdout);
    std::_Exit(0);
}
