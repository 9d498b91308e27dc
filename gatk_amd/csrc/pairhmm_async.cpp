// gatk_pairhmm_submit / gatk_pairhmm_fetch and the region-level hook gatk_pairhmm_initialize_region
// (include/gatk_pairhmm.h): region-sized calls that return before the device has scored them, and the
// per-device dispatcher that carries them.
//
// A HaplotypeCaller-sized region (some hundred pairs) keeps a tenth of the device busy for some thirty
// microseconds, and a caller that waits for each region before it sends the next leaves the device idle in
// between.  The dispatcher owns a few engines per device.  A request goes to an idle engine at once; when all are
// busy it waits in a queue, and whoever next needs an engine (a caller fetching a result, or submitting) first
// takes the oldest pass home and then leads ONE pass over everything that has queued up meanwhile, whichever
// thread submitted it.  So the passes grow with the load and no thread is set aside to drive the device.
// Results and errors stay per request: a pass that reports an input error is repeated request by request, so that
// the reference's exception reaches the caller whose data raised it, and only that caller.
#include "pairhmm_engine.h"

using namespace pairhmm_hip;
using namespace pairhmm_host;

namespace pairhmm_host {
int create_shared_engine(int device, bool squash, gatk_pairhmm** out);  // pairhmm_api.cpp
}

namespace {

struct Dispatcher {
    static constexpr int kEngines = 4;               // as many as the runtime has hardware queues by default
    static constexpr size_t kMaxRequestsPerPass = 64;
    static constexpr int64_t kMaxPairsPerPass = 32768;  // a pass takes requests until it holds this many pairs
    static constexpr int kSpinMicroseconds = 250;    // a waiting caller spins this long for a change before it sleeps,
    static constexpr int kMaxSpinners = 12;          // and at most this many do so at once
    int device = 0;
    bool squash = false;
    std::mutex m;
    std::condition_variable cv;
    std::deque<AsyncRequest*> pending;
    struct Engine {
        gatk_pairhmm* h = nullptr;
        enum State { kIdle, kBusy, kInFlight } state = kIdle;  // kBusy: a thread is planning or fetching on it
        uint64_t pass_no = 0;                                   // age of the pass in flight
        std::vector<AsyncRequest*> reqs;
        std::vector<gatk_pairhmm_batch> batches;
        std::vector<double*> outs;
        PendingDownload dl;
    } eng[kEngines];
    uint64_t next_pass_no = 1;
    std::atomic<uint64_t> changes{0};  // moves (under m) whenever an engine changes state; waiting callers spin on it
    int spinners = 0;
    int64_t passes = 0, requests = 0;

    int idle_engine() const {
        for (int e = 0; e < kEngines; ++e)
            if (eng[e].state == Engine::kIdle) return e;
        return -1;
    }
    int oldest_in_flight() const {
        int best = -1;
        for (int e = 0; e < kEngines; ++e)
            if (eng[e].state == Engine::kInFlight && (best < 0 || eng[e].pass_no < eng[best].pass_no)) best = e;
        return best;
    }
    void changed() {
        changes.fetch_add(1, std::memory_order_release);
        cv.notify_all();
    }
    // Waits (lock held on entry and exit) until some engine has changed state.
    void wait_change(std::unique_lock<std::mutex>& lock) {
        const uint64_t seen = changes.load(std::memory_order_acquire);
        const bool spin = spinners < kMaxSpinners;  // the rest sleep: spinning callers must not take every core
        if (spin) ++spinners;
        lock.unlock();
        if (spin) {
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(kSpinMicroseconds);
            for (int k = 0; changes.load(std::memory_order_acquire) == seen; ++k) {
                _mm_pause();
                if ((k & 63) == 63 && std::chrono::steady_clock::now() >= until) break;
            }
        }
        lock.lock();
        if (spin) --spinners;
        if (changes.load(std::memory_order_acquire) == seen) cv.wait(lock);  // the counter moves under this mutex
    }

    // Every request of a failed pass again, alone and to the end, so that each gets its own code and message.
    static void rerun_one_by_one(Engine& E) {
        for (AsyncRequest* r : E.reqs) {
            r->rc = compute_regions_single(E.h, static_cast<int32_t>(r->batches.size()), r->batches.data(), r->outs.data(), true);
            r->error = r->rc == GATK_PAIRHMM_OK ? std::string() : last_error();
        }
    }

    void finish_pass(Engine& E) {  // lock held
        for (AsyncRequest* r : E.reqs) { r->state = AsyncRequest::kDone; r->engine = -1; }
        ++passes;
        requests += static_cast<int64_t>(E.reqs.size());
        E.reqs.clear();
        E.state = Engine::kIdle;
        changed();
    }

    // Leads a pass on idle engine e over what is queued (lock held on entry and exit, released while planning).
    void lead(std::unique_lock<std::mutex>& lock, int e) {
        Engine& E = eng[e];
        E.reqs.clear(); E.batches.clear(); E.outs.clear();
        int64_t pairs = 0;
        while (!pending.empty() && E.reqs.size() < kMaxRequestsPerPass && (E.reqs.empty() || pairs + pending.front()->pairs <= kMaxPairsPerPass)) {
            AsyncRequest* r = pending.front();
            pending.pop_front();
            r->state = AsyncRequest::kInFlight;
            r->engine = e;
            pairs += r->pairs;
            E.reqs.push_back(r);
        }
        if (E.reqs.empty()) return;
        E.state = Engine::kBusy;
        E.pass_no = next_pass_no++;
        lock.unlock();
        int rc = GATK_PAIRHMM_OK;
        std::string error;
        if (!E.h) {  // only the thread that made the engine busy touches it
            rc = create_shared_engine(device, squash, &E.h);
            if (rc != GATK_PAIRHMM_OK) error = last_error();
        }
        bool in_flight = false;
        if (rc == GATK_PAIRHMM_OK) {
            for (AsyncRequest* r : E.reqs) {
                E.batches.insert(E.batches.end(), r->batches.begin(), r->batches.end());
                E.outs.insert(E.outs.end(), r->outs.begin(), r->outs.end());
            }
            gatk_pairhmm_resident* R = &E.h->scratch;
            R->region_label_base = 0;
            R->regions_are_pieces = false;
            int prc = plan_and_upload(E.h, R, static_cast<int>(E.batches.size()), E.batches.data(), true);
            if (prc == GATK_PAIRHMM_OK && R->total_pairs > 0) prc = enqueue_compute(E.h, R);
            if (prc == GATK_PAIRHMM_OK) prc = download_begin(E.h, R, E.outs.data(), E.dl);
            if (prc == GATK_PAIRHMM_OK) in_flight = true;
            else {  // could not even be enqueued: let every request find out for itself
                (void)hipStreamSynchronize(R->stream);
                rerun_one_by_one(E);
            }
        } else {
            for (AsyncRequest* r : E.reqs) { r->rc = rc; r->error = error; }
        }
        lock.lock();
        if (in_flight) {
            E.state = Engine::kInFlight;
            changed();
        } else {
            finish_pass(E);
        }
    }

    // Takes the pass in flight on engine e home (lock held on entry and exit, released while waiting for it).
    void complete(std::unique_lock<std::mutex>& lock, int e) {
        Engine& E = eng[e];
        E.state = Engine::kBusy;
        lock.unlock();
        const int rc = download_end(E.h, &E.h->scratch, E.outs.data(), E.dl);
        if (rc != GATK_PAIRHMM_OK) {
            if (E.reqs.size() == 1) { E.reqs[0]->rc = rc; E.reqs[0]->error = last_error(); }
            else rerun_one_by_one(E);
        }
        lock.lock();
        finish_pass(E);
    }

    void submit(AsyncRequest* r) {
        std::unique_lock<std::mutex> lock(m);
        r->state = AsyncRequest::kPending;
        r->rc = GATK_PAIRHMM_OK;
        r->error.clear();
        pending.push_back(r);
        const int e = idle_engine();
        if (e >= 0) lead(lock, e);
    }

    // Returns when r is done.  A caller whose request still waits, or whose pass is still out, works for it: it
    // takes passes home and leads the next one over everything that has queued up.
    void fetch(AsyncRequest* r) {
        std::unique_lock<std::mutex> lock(m);
        while (r->state != AsyncRequest::kDone) {
            if (r->state == AsyncRequest::kPending) {
                int e = idle_engine();
                if (e >= 0) { lead(lock, e); continue; }
                e = oldest_in_flight();
                if (e >= 0) { complete(lock, e); continue; }
                wait_change(lock);  // every engine is in another thread's hands
                continue;
            }
            const int e = r->engine;
            if (eng[e].state == Engine::kInFlight) {
                complete(lock, e);
                if (!pending.empty() && eng[e].state == Engine::kIdle) lead(lock, e);  // keep the device fed
                continue;
            }
            wait_change(lock);  // another thread is taking that pass home
        }
    }

    // Forget r without its results: out of the queue, or wait for the pass that would write into it.
    void cancel(AsyncRequest* r) {
        {
            std::unique_lock<std::mutex> lock(m);
            if (r->state == AsyncRequest::kPending) {
                pending.erase(std::remove(pending.begin(), pending.end(), r), pending.end());
                r->state = AsyncRequest::kDone;
                return;
            }
        }
        fetch(r);
    }
};

std::mutex g_dispatchers_mutex;
std::map<int, Dispatcher*> g_dispatchers;  // by device * 2 + squash; never freed (their engines are)

// One dispatcher per device and per value of the one result-changing option the shared engines support.
Dispatcher* dispatcher_for(int device, bool squash) {
    std::lock_guard<std::mutex> lock(g_dispatchers_mutex);
    Dispatcher*& d = g_dispatchers[device * 2 + (squash ? 1 : 0)];
    if (!d) {
        d = new Dispatcher();
        d->device = device;
        d->squash = squash;
    }
    return d;
}

// A byte array a caller may leave out against one it may give: equal when both are there and equal, or when the
// one that is there holds nothing but the default the other stands for.
bool same_or_default(const uint8_t* a, const uint8_t* b, size_t n, uint8_t dflt) {
    if (a && b) return memcmp(a, b, n) == 0;
    const uint8_t* p = a ? a : b;
    if (!p) return true;
    for (size_t k = 0; k < n; ++k)
        if (p[k] != dflt) return false;
    return true;
}

// Base quality as the kernels use it (pairhmm_device.h, effective_qual): capped by the read's mapping quality when
// one is given, then squashed to the minimum usable quality when under the threshold and the handle asks for it.
inline unsigned effective_qual(unsigned q, unsigned mapq, bool squash) {
    q = q < mapq ? q : mapq;
    return squash && q < kBaseQualityScoreThreshold ? kMinUsableQScore : q;
}

// Base qualities of two batches with the same read lengths: the same bytes, or, where a mapping-quality cap or the
// squashing option is in play, the same qualities once those have been applied (a caller may send a region's raw
// reads ahead with their mapping qualities and ask with the qualities it has capped and squashed itself).
bool same_base_quals(const gatk_pairhmm_batch& kept, const gatk_pairhmm_batch& asked, int32_t asked_base, bool squash) {
    const int nr = kept.num_reads;
    const size_t rbytes = static_cast<size_t>(kept.read_offsets[nr]);
    const uint8_t* const aq = asked.read_quals + asked_base;
    if (memcmp(kept.read_quals, aq, rbytes) == 0 && same_or_default(kept.read_mapq, asked.read_mapq, static_cast<size_t>(nr), 255)) return true;
    if (!squash && !kept.read_mapq && !asked.read_mapq) return false;
    for (int r = 0; r < nr; ++r) {
        const unsigned km = kept.read_mapq ? kept.read_mapq[r] : 255u, am = asked.read_mapq ? asked.read_mapq[r] : 255u;
        for (int32_t i = kept.read_offsets[r]; i < kept.read_offsets[r + 1]; ++i) {
            // 255 is the out-of-bounds quality whatever the cap: it must reach the kernels' check as it is
            if ((kept.read_quals[i] == 255) != (aq[i] == 255)) return false;
            if (effective_qual(kept.read_quals[i], km, squash) != effective_qual(aq[i], am, squash)) return false;
        }
    }
    return true;
}

// Does `asked` describe exactly the computation `kept` (a region sent ahead) stands for?
bool same_inputs(const gatk_pairhmm_batch& kept, const gatk_pairhmm_batch& asked, bool squash) {
    if (kept.num_reads != asked.num_reads || kept.num_haps != asked.num_haps) return false;
    if (asked.num_reads <= 0 || asked.num_haps <= 0 || !asked.read_offsets || !asked.hap_offsets || !asked.read_bases || !asked.read_quals || !asked.hap_bases)
        return false;
    const int nr = asked.num_reads, nh = asked.num_haps;
    const int32_t rb = asked.read_offsets[0], hb = asked.hap_offsets[0];
    for (int r = 1; r <= nr; ++r)
        if (asked.read_offsets[r] - rb != kept.read_offsets[r]) return false;
    for (int a = 1; a <= nh; ++a)
        if (asked.hap_offsets[a] - hb != kept.hap_offsets[a]) return false;
    const size_t rbytes = static_cast<size_t>(kept.read_offsets[nr]), hbytes = static_cast<size_t>(kept.hap_offsets[nh]);
    return memcmp(kept.read_bases, asked.read_bases + rb, rbytes) == 0 && memcmp(kept.hap_bases, asked.hap_bases + hb, hbytes) == 0 &&
           same_or_default(kept.ins_gop, asked.ins_gop ? asked.ins_gop + rb : nullptr, rbytes, kDefaultIndelQual) &&
           same_or_default(kept.del_gop, asked.del_gop ? asked.del_gop + rb : nullptr, rbytes, kDefaultIndelQual) &&
           same_or_default(kept.gcp, asked.gcp ? asked.gcp + rb : nullptr, rbytes, kDefaultGcp) &&
           same_base_quals(kept, asked, rb, squash);
}

// The library's own copy of the regions of a submit call.  own_results: the results stay with the request
// (regions sent ahead) instead of going to arrays of the caller.
void copy_regions(AsyncRequest* q, int num_regions, const gatk_pairhmm_batch* regions, double* const* likelihoods, bool own_results) {
    size_t bytes = 0, ints = 0;
    int64_t pairs = 0;
    for (int k = 0; k < num_regions; ++k) {
        const gatk_pairhmm_batch& b = regions[k];
        if (b.num_reads <= 0 || b.num_haps <= 0) continue;
        const size_t rbytes = static_cast<size_t>(b.read_offsets[b.num_reads] - b.read_offsets[0]);
        bytes += rbytes * (2 + (b.ins_gop ? 1 : 0) + (b.del_gop ? 1 : 0) + (b.gcp ? 1 : 0)) + (b.read_mapq ? b.num_reads : 0) +
                 static_cast<size_t>(b.hap_offsets[b.num_haps] - b.hap_offsets[0]);
        ints += static_cast<size_t>(b.num_reads) + b.num_haps + 2;
        pairs += static_cast<int64_t>(b.num_reads) * b.num_haps;
    }
    q->blob.resize(bytes + 1);
    q->offsets.resize(ints + 1);
    q->batches.assign(static_cast<size_t>(num_regions), gatk_pairhmm_batch{});
    q->outs.assign(static_cast<size_t>(num_regions), nullptr);
    q->out_off.assign(static_cast<size_t>(num_regions) + 1, 0);
    q->pairs = pairs;
    if (own_results) q->own_out.resize(static_cast<size_t>(pairs));
    uint8_t* p = q->blob.data();
    int32_t* o = q->offsets.data();
    auto take = [&p](const uint8_t* src, size_t n) { uint8_t* at = p; memcpy(at, src, n); p += n; return at; };
    for (int k = 0; k < num_regions; ++k) {
        const gatk_pairhmm_batch& b = regions[k];
        gatk_pairhmm_batch& c = q->batches[k];
        q->out_off[k + 1] = q->out_off[k];
        if (b.num_reads <= 0 || b.num_haps <= 0) { c.num_reads = 0; c.num_haps = b.num_haps; continue; }
        c.num_reads = b.num_reads;
        c.num_haps = b.num_haps;
        const int32_t rb = b.read_offsets[0], hb = b.hap_offsets[0];
        const size_t rbytes = static_cast<size_t>(b.read_offsets[b.num_reads] - rb);
        c.read_offsets = o;
        for (int r = 0; r <= b.num_reads; ++r) *o++ = b.read_offsets[r] - rb;
        c.hap_offsets = o;
        for (int a = 0; a <= b.num_haps; ++a) *o++ = b.hap_offsets[a] - hb;
        c.read_bases = take(b.read_bases + rb, rbytes);
        c.read_quals = take(b.read_quals + rb, rbytes);
        if (b.ins_gop) c.ins_gop = take(b.ins_gop + rb, rbytes);
        if (b.del_gop) c.del_gop = take(b.del_gop + rb, rbytes);
        if (b.gcp) c.gcp = take(b.gcp + rb, rbytes);
        if (b.read_mapq) c.read_mapq = take(b.read_mapq, static_cast<size_t>(b.num_reads));
        c.hap_bases = take(b.hap_bases + hb, static_cast<size_t>(b.hap_offsets[b.num_haps] - hb));
        q->out_off[k + 1] += static_cast<int64_t>(b.num_reads) * b.num_haps;
        q->outs[k] = own_results ? q->own_out.data() + q->out_off[k] : likelihoods[k];
    }
}

AsyncRequest* new_request(gatk_pairhmm* h) {
    std::unique_ptr<AsyncRequest> q;
    if (!h->spare_requests.empty()) {
        q = std::move(h->spare_requests.back());
        h->spare_requests.pop_back();
    } else {
        q.reset(new AsyncRequest());
    }
    q->id = h->next_ticket++;
    q->state = AsyncRequest::kDone;
    q->engine = -1;
    q->rc = GATK_PAIRHMM_OK;
    q->error.clear();
    q->dispatcher = nullptr;
    AsyncRequest* raw = q.get();
    h->tickets[raw->id] = std::move(q);
    return raw;
}

void retire(gatk_pairhmm* h, int64_t ticket) {
    auto it = h->tickets.find(ticket);
    if (it == h->tickets.end()) return;
    if (h->spare_requests.size() < 8) h->spare_requests.push_back(std::move(it->second));
    h->tickets.erase(it);
}

// The single-device engine a request of this handle goes to: the handle itself, or for a multi-device handle its
// engines in turn.
gatk_pairhmm* engine_for_request(gatk_pairhmm* h) {
    if (h->children.empty()) return h;
    return h->children[h->next_child++ % h->children.size()];
}

// Checks, copies and sends off the regions; *ticket names the request.
int submit_regions(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions, double* const* likelihoods, bool own_results, int64_t* ticket) {
    if (num_regions < 0 || (num_regions > 0 && !regions)) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "regions cannot be null");
    for (int k = 0; k < num_regions; ++k) {
        int read_max = 0, hap_max = 0;
        if (int rc = check_region(regions[k], k, &read_max, &hap_max)) return rc;
        note_call_maxima(h, regions[k], read_max, hap_max);
        if (!own_results && regions[k].num_reads > 0 && regions[k].num_haps > 0 && (!likelihoods || !likelihoods[k]))
            return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "likelihoods cannot be null");
    }
    AsyncRequest* q = new_request(h);
    copy_regions(q, num_regions, regions, likelihoods, own_results);
    *ticket = q->id;
    if (q->pairs == 0) return GATK_PAIRHMM_OK;  // nothing to score: done as it stands
    gatk_pairhmm* target = engine_for_request(h);
    for (gatk_pairhmm* c : h->children) { c->initialized = h->initialized; c->max_read = h->max_read; c->max_hap = h->max_hap; }
    // A call that fills the device by itself, or a handle whose options the shared engines do not run with, is
    // scored here and now on the handle's own engine; fetch then only hands the outcome over.
    if (!has_default_options(target) || q->pairs > target->pipeline_pairs || target->pipeline_pairs == 0) {
        const int32_t n = static_cast<int32_t>(q->batches.size());
        q->rc = h->children.empty() ? compute_regions_single(h, n, q->batches.data(), q->outs.data(), true)
                                    : gatk_pairhmm_compute_regions(h, n, q->batches.data(), q->outs.data());  // sharded over the devices
        if (q->rc != GATK_PAIRHMM_OK) q->error = last_error();
        return GATK_PAIRHMM_OK;
    }
    Dispatcher* d = dispatcher_for(target->device, target->squash_base_quals);
    q->dispatcher = d;
    d->submit(q);
    return GATK_PAIRHMM_OK;
}

int fetch_request(AsyncRequest* q) {
    if (q->dispatcher) static_cast<Dispatcher*>(q->dispatcher)->fetch(q);
    return q->rc == GATK_PAIRHMM_OK ? GATK_PAIRHMM_OK : fail(q->rc, q->error);
}

}  // namespace

namespace pairhmm_host {

bool has_default_options(const gatk_pairhmm* h) {
    return h->fp32_threshold == 1e-28f && h->force_rows_per_lane == 0 && !h->force_fp64 && h->emit_mode == -1 &&
           h->stream_mode == -1 && !h->timing && !h->no_tristate_correction;
}

void drop_region(gatk_pairhmm* h) {
    if (h->region_ticket == 0) return;
    auto it = h->tickets.find(h->region_ticket);
    if (it != h->tickets.end()) {
        AsyncRequest* q = it->second.get();
        if (q->dispatcher && !h->region_fetched) static_cast<Dispatcher*>(q->dispatcher)->cancel(q);
        retire(h, h->region_ticket);
    }
    h->region_ticket = 0;
    h->region_fetched = false;
    h->region_next_sample = 0;
}

void drop_tickets(gatk_pairhmm* h) {
    drop_region(h);
    for (auto& kv : h->tickets) {
        AsyncRequest* q = kv.second.get();
        if (q->dispatcher) static_cast<Dispatcher*>(q->dispatcher)->cancel(q);
    }
    h->tickets.clear();
    h->spare_requests.clear();
}

void release_dispatch_engines() {
    std::lock_guard<std::mutex> lock(g_dispatchers_mutex);
    for (auto& kv : g_dispatchers) {
        Dispatcher* d = kv.second;
        std::unique_lock<std::mutex> inner(d->m);
        for (auto& E : d->eng) {
            // no caller's handle is left, so no request is: an engine still in flight carries nothing anyone waits for
            if (E.state == Dispatcher::Engine::kInFlight) {
                inner.unlock();
                (void)download_end(E.h, &E.h->scratch, E.outs.data(), E.dl);
                inner.lock();
                E.reqs.clear();
                E.state = Dispatcher::Engine::kIdle;
            }
            if (E.h && E.state == Dispatcher::Engine::kIdle) {
                gatk_pairhmm_close(E.h);
                E.h = nullptr;
            }
        }
    }
}

void dispatch_counts(int device, bool squash, int64_t* passes, int64_t* requests) {
    Dispatcher* d = dispatcher_for(device, squash);
    std::lock_guard<std::mutex> lock(d->m);
    *passes = d->passes;
    *requests = d->requests;
}

int compute_shared(gatk_pairhmm* h, const gatk_pairhmm_batch* batch, double* likelihoods) {
    AsyncRequest q;  // the call blocks, so the caller's arrays serve as they are: nothing is copied
    q.batches.assign(1, *batch);
    q.outs.assign(1, likelihoods);
    q.pairs = static_cast<int64_t>(batch->num_reads) * batch->num_haps;
    Dispatcher* d = dispatcher_for(h->device, h->squash_base_quals);
    q.dispatcher = d;
    d->submit(&q);
    return fetch_request(&q);
}

int serve_from_region(gatk_pairhmm* h, const gatk_pairhmm_batch* batch, double* likelihoods, bool* hit) {
    *hit = false;
    if (h->region_ticket == 0) return GATK_PAIRHMM_OK;
    auto it = h->tickets.find(h->region_ticket);
    if (it == h->tickets.end()) return GATK_PAIRHMM_OK;
    AsyncRequest* q = it->second.get();
    const int n = static_cast<int>(q->batches.size());
    const bool squash = (h->children.empty() ? h : h->children[0])->squash_base_quals;
    int found = -1;
    for (int step = 0; step < n && found < 0; ++step) {  // samples are asked for in the order they were sent: try that one first
        const int s = (h->region_next_sample + step) % n;
        if (q->batches[s].num_reads > 0 && same_inputs(q->batches[s], *batch, squash)) found = s;
    }
    if (found < 0) { ++h->region_misses; return GATK_PAIRHMM_OK; }
    if (!h->region_fetched) {
        (void)fetch_request(q);  // its outcome stays in the request
        h->region_fetched = true;
    }
    // An input error somewhere in the region: this sample goes the ordinary way, which raises the reference's
    // exception if the sample is the one that holds the error and scores it otherwise.
    if (q->rc != GATK_PAIRHMM_OK) { ++h->region_misses; return GATK_PAIRHMM_OK; }
    const int64_t count = q->out_off[found + 1] - q->out_off[found];
    memcpy(likelihoods, q->own_out.data() + q->out_off[found], static_cast<size_t>(count) * sizeof(double));
    h->region_next_sample = (found + 1) % n;
    ++h->region_hits;
    *hit = true;
    return GATK_PAIRHMM_OK;
}

}  // namespace pairhmm_host

extern "C" {

int gatk_pairhmm_submit(gatk_pairhmm* h, int32_t num_regions, const gatk_pairhmm_batch* regions, double* const* likelihoods,
                        int64_t* ticket) try {
    if (int rc = check_handle(h)) return rc;
    if (!ticket) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "ticket cannot be null");
    *ticket = 0;
    return submit_regions(h, num_regions, regions, likelihoods, false, ticket);
} catch (...) { return on_exception(); }

int gatk_pairhmm_fetch(gatk_pairhmm* h, int64_t ticket) try {
    if (int rc = check_handle(h)) return rc;
    auto it = h->tickets.find(ticket);
    if (it == h->tickets.end() || ticket == h->region_ticket) return fail(GATK_PAIRHMM_ILLEGAL_ARGUMENT, "no such ticket on this handle: " + std::to_string(ticket));
    AsyncRequest* q = it->second.get();
    const int rc = fetch_request(q);
    h->stats = gatk_pairhmm_stats{};
    h->stats.pairs = q->pairs;
    retire(h, ticket);
    return rc;
} catch (...) { return on_exception(); }

int gatk_pairhmm_initialize_region(gatk_pairhmm* h, int32_t num_samples, const gatk_pairhmm_batch* per_sample,
                                   int32_t read_max_length, int32_t haplotype_max_length) try {
    if (int rc = check_handle(h)) return rc;
    if (int rc = gatk_pairhmm_initialize(h, read_max_length, haplotype_max_length)) return rc;  // PairHMM.java:138
    drop_region(h);
    if (num_samples <= 0) return GATK_PAIRHMM_OK;
    int64_t ticket = 0;
    if (int rc = submit_regions(h, num_samples, per_sample, nullptr, true, &ticket)) return rc;
    h->region_ticket = ticket;
    h->region_fetched = false;
    h->region_next_sample = 0;
    return GATK_PAIRHMM_OK;
} catch (...) { return on_exception(); }

}  // extern "C"
