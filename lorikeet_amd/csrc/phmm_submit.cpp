// phmm_submit / phmm_wait / phmm_engine_submit (include/phmm.h): many host threads, one shared engine handle.
// Built on the two internal entry points of the host path (phmm_host.hpp); there is no CPU fallback here either.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <unordered_map>

#include "phmm_host.hpp"

using namespace phmm_host;

// ---- phmm_submit / phmm_wait: cross-thread batching -------------------------------------------------------------------
// The reference calls the PairHMM once per region from every rayon worker (assembly_region_walker.rs:210-273), and one
// region fills a quarter of the chip for a few tens of microseconds.  Here the workers share one handle: a submission is
// only queued; the first thread that waits while a lane is free becomes the leader of one flush, takes everything queued
// so far -- its own region and those of the threads that arrived meanwhile -- and computes it as ONE batch (one H2D copy,
// one set of launches, one D2H copy), then hands every region's results to its owner.  Nobody waits on a timer: batches
// grow exactly as large as the number of threads that were waiting anyway.  Four lanes (private engine handles) let the
// next flush stage and copy while the previous one computes.
constexpr uint64_t kServerTicket = 1ull << 63;  // tickets of calls the region server has: the rest is the ServerPending's address

struct Submission {
    uint32_t n_regions = 0, n_reads = 0, n_haps = 0;
    const uint32_t *region_read_off = nullptr, *region_hap_off = nullptr, *read_off = nullptr, *hap_off = nullptr;
    const uint8_t *payload[6] = {};
    const uint64_t *out_off = nullptr;
    double *out = nullptr;
    size_t read_bytes = 0, hap_bytes = 0;
    uint64_t n_out = 0;
    // phmm_engine_submit: payload = {read_bases, base_q, ins_q | NULL, del_q | NULL, unused, hap_bases}
    bool engine = false;
    // phmm_region_submit: everything is in `ra` (the fields above but read_bytes are unused)
    bool region = false;
    RegionArgs ra;
    phmm_engine_config cfg{};
    const uint8_t *mapq = nullptr;
    const int32_t *ref_hap = nullptr;
    uint8_t *keep = nullptr;
    enum State { QUEUED, RUNNING, DONE } state = QUEUED;
    int status = PHMM_OK;
    std::string err;
    // The thread in phmm_wait sleeps on a condition variable of its submission's own: it is woken when its flush is done,
    // or when it is the oldest submission in the queue and a lane has become free (it then leads the next flush).  (One
    // condition variable for all, notify_all at the end of every flush, woke every waiting worker 6 000 times a second at
    // 32 threads, each to take the mutex and go back to sleep.)
    std::shared_ptr<std::condition_variable> cv = std::make_shared<std::condition_variable>();
    bool waiting = false;  // a thread sleeps on `cv` right now (a submission nobody waits for yet cannot lead a flush)
    // may share a flush with `o`: same entry point and, for the engine-level call, the same configuration and the same
    // optional arrays present
    bool compatible(const Submission &o) const {
        if (engine != o.engine || region != o.region) return false;
        if (region)
            return memcmp(&ra.cfg, &o.ra.cfg, sizeof ra.cfg) == 0 && memcmp(&ra.rcfg, &o.ra.rcfg, sizeof ra.rcfg) == 0 &&
                   !ra.ins_q == !o.ra.ins_q && !ra.del_q == !o.ra.del_q && !ra.hap_priority == !o.ra.hap_priority &&
                   !ra.read_soft_clip == !o.ra.read_soft_clip;
        if (!engine) return true;
        return memcmp(&cfg, &o.cfg, sizeof cfg) == 0 && !payload[2] == !o.payload[2] && !payload[3] == !o.payload[3] &&
               !ref_hap == !o.ref_hap;
    }
};

struct Combiner {
    static constexpr int kMaxLanes = 8;
    static constexpr size_t kMaxParts = 256;
    std::mutex mu;
    std::condition_variable gather_cv;  // the one leader that is letting submissions arrive (below) sleeps here; phmm_submit wakes it
    bool gathering = false;
    int gather_us = 40;  // (0 / 40 / 100 measured in round 5, tools/run/threads.sh: 40)
    std::deque<uint64_t> queue;  // tickets nobody has picked up yet, in submission order
    std::unordered_map<uint64_t, Submission> live;  // until phmm_wait returns them (element addresses are stable)
    uint64_t next_ticket = 1;
    int n_lanes = 4;  // PHMM_SUBMIT_LANES: 2 flushes in flight leave the GPU idle during their copies, 4 match private handles at 4 threads
    phmm_handle *lane[kMaxLanes] = {};
    bool lane_busy[kMaxLanes] = {};
    uint64_t n_flushes = 0, n_parts = 0;  // statistics (phmm_submit_stats)
    double flush_us = 0;  // time spent inside flushes (PHMM_TRACE prints the mean when the handle is destroyed)
    bool trace = false;
    struct Scratch {  // per lane, reused between flushes
        std::vector<uint32_t> rro, rho, ro, ho;
        std::vector<uint64_t> oo;
        Parts parts;
        // engine-level flushes are concatenated on the host (every array, the optional ones included) and scattered back
        std::vector<uint8_t> bytes[5], mapq, keep;  // bytes: read_bases, base_q, ins_q, del_q, hap_bases
        std::vector<int32_t> ref;
        std::vector<double> out;
        // region-level flushes: only the offset arrays are concatenated, payload and results go straight from / to their owners
        std::vector<uint32_t> hco, oco;
        std::vector<uint64_t> outco;
        std::vector<RegionArgs> rparts;
    } scratch[kMaxLanes];
};

// a lane is free (or about to be) and work is queued: its oldest submission's waiter leads the next flush
static void wake_queue_head(Combiner *c) {
    for (uint64_t t : c->queue) {
        auto it = c->live.find(t);
        if (it != c->live.end() && it->second.waiting) {
            it->second.cv->notify_one();
            return;
        }
    }
}

namespace phmm_host {

void combiner_destroy(Combiner *c) {
    if ((c->trace || getenv("PHMM_SUBMIT_STATS")) && c->n_flushes)  // (PHMM_SUBMIT_STATS: this line alone, for A/B runs)
        fprintf(stderr, "phmm_submit: %llu flushes carried %llu submissions, mean flush %.1f us\n",
                (unsigned long long)c->n_flushes, (unsigned long long)c->n_parts, c->flush_us / c->n_flushes);
    for (int l = 0; l < Combiner::kMaxLanes; ++l)
        if (c->lane[l]) phmm_destroy(c->lane[l]);
    delete c;
}

void combiner_set_switches(Combiner *c, const Switches &sw) {
    std::lock_guard<std::mutex> lk(c->mu);
    for (int l = 0; l < Combiner::kMaxLanes; ++l)
        if (c->lane[l]) c->lane[l]->sw = sw;
}

uint64_t combiner_stat(Combiner *c, const char *name) {
    std::lock_guard<std::mutex> lk(c->mu);
    uint64_t n = 0;
    for (int l = 0; l < Combiner::kMaxLanes; ++l)
        if (c->lane[l]) n += phmm_get_stat(c->lane[l], name);
    return n;
}

}  // namespace phmm_host

namespace {

int submission_alone(phmm_handle *lane, Submission *s) {
    if (s->region)
        s->status = region_compute(lane, s->ra);
    else if (s->engine)
        s->status = phmm_engine_compute(lane, &s->cfg, s->n_regions, s->region_read_off, s->region_hap_off, s->read_off,
                                        s->payload[0], s->payload[1], s->payload[2], s->payload[3], s->mapq, s->hap_off,
                                        s->payload[5], s->ref_hap, s->out_off, s->out, s->keep);
    else
        s->status = phmm_compute(lane, s->n_regions, s->region_read_off, s->region_hap_off, s->read_off, s->payload[0], s->payload[1],
                             s->payload[2], s->payload[3], s->payload[4], s->hap_off, s->payload[5], s->out_off, s->out);
    if (s->status != PHMM_OK) s->err = lane->err;
    return s->status;
}

// One flush on one lane, outside the combiner's lock: `subs` are RUNNING and belong to this call.
void run_flush(phmm_handle *lane, Combiner::Scratch &w, std::vector<Submission *> &subs) {
    if (subs.size() == 1) {
        submission_alone(lane, subs[0]);
        return;
    }
    struct Defer {  // combined flushes happen when other threads are busy too: keep the copy engines unblocked
        phmm_handle *h;
        explicit Defer(phmm_handle *hh) : h(hh) { h->defer_d2h = true; }
        ~Defer() { h->defer_d2h = false; }
    } defer(lane);
    // concatenate the offset arrays (every submission starts at 0, so each is shifted by what came before it)
    w.rro.assign(1, 0);
    w.rho.assign(1, 0);
    w.ro.assign(1, 0);
    w.ho.assign(1, 0);
    w.oo.assign(1, 0);
    Parts &parts = w.parts;
    for (int i = 0; i < 6; ++i) parts.src[i].clear();
    parts.read_bytes.clear();
    parts.hap_bytes.clear();
    parts.out.clear();
    parts.n_out.clear();
    parts.first_region.assign(1, 0);
    for (const Submission *s : subs) {
        if (s->region) break;  // (region-level flushes build their own, below)
        const uint32_t r0 = w.rro.back(), h0 = w.rho.back(), rb0 = w.ro.back(), hb0 = w.ho.back();
        const uint64_t o0 = w.oo.back();
        for (uint32_t g = 1; g <= s->n_regions; ++g) {
            w.rro.push_back(r0 + s->region_read_off[g]);
            w.rho.push_back(h0 + s->region_hap_off[g]);
            w.oo.push_back(o0 + s->out_off[g]);
        }
        for (uint32_t r = 1; r <= s->n_reads; ++r) w.ro.push_back(rb0 + s->read_off[r]);
        for (uint32_t a = 1; a <= s->n_haps; ++a) w.ho.push_back(hb0 + s->hap_off[a]);
        for (int i = 0; i < 6; ++i) parts.src[i].push_back(s->payload[i]);
        parts.read_bytes.push_back(s->read_bytes);
        parts.hap_bytes.push_back(s->hap_bytes);
        parts.out.push_back(s->out);
        parts.n_out.push_back(s->n_out);
        parts.first_region.push_back(parts.first_region.back() + s->n_regions);
    }
    if (subs[0]->region) {
        // phmm_region_submit: the regions of all waiting workers as ONE enqueue of the whole per-region pipeline
        w.rro.assign(1, 0);
        w.rho.assign(1, 0);
        w.ro.assign(1, 0);
        w.ho.assign(1, 0);
        w.oo.assign(1, 0);
        w.hco.assign(1, 0);
        w.oco.assign(1, 0);
        w.outco.assign(1, 0);
        w.rparts.clear();
        bool holes = false;
        for (const Submission *s : subs) {
            const RegionArgs &q = s->ra;
            const uint32_t r0 = w.rro.back(), h0 = w.rho.back(), rb0 = w.ro.back(), hb0 = w.ho.back(), hc0 = w.hco.back(), oc0 = w.oco.back();
            const uint64_t o0 = w.oo.back(), oc_out0 = w.outco.back();
            const uint32_t nr = q.region_read_off[q.n_regions], nh = q.region_hap_off[q.n_regions];
            for (uint32_t g = 1; g <= q.n_regions; ++g) {
                w.rro.push_back(r0 + q.region_read_off[g]);
                w.rho.push_back(h0 + q.region_hap_off[g]);
                // (the combined matrix is tight: a submission's own gaps in out_off are honoured when its results are handed back)
                w.oo.push_back(w.oo.back() + (uint64_t)(q.region_read_off[g] - q.region_read_off[g - 1]) * (q.region_hap_off[g] - q.region_hap_off[g - 1]));
            }
            (void)o0;
            for (uint32_t r = 1; r <= nr; ++r) {
                w.ro.push_back(rb0 + q.read_off[r]);
                w.oco.push_back(oc0 + (nr ? q.orig_cigar_off[r] : 0));
                w.outco.push_back(oc_out0 + (nr ? q.out_cigar_off[r] : 0));
            }
            for (uint32_t a = 1; a <= nh; ++a) {
                w.ho.push_back(hb0 + q.hap_off[a]);
                w.hco.push_back(hc0 + (q.hap_cigar_off ? q.hap_cigar_off[a] : 0));
            }
            holes = holes || (nh && !q.hap_cigar_off);
            w.rparts.push_back(q);
        }
        if (holes) {  // (a submission without reads may come without the realignment arrays: nothing to combine it with)
            for (Submission *s : subs) submission_alone(lane, s);
            return;
        }
        RegionArgs c = subs[0]->ra;  // configuration and which optional arrays are present
        c.n_regions = (uint32_t)w.rro.size() - 1;
        c.region_read_off = w.rro.data();
        c.region_hap_off = w.rho.data();
        c.read_off = w.ro.data();
        c.hap_off = w.ho.data();
        c.out_off = w.oo.data();
        c.hap_cigar_off = w.hco.data();
        c.orig_cigar_off = w.oco.data();
        c.out_cigar_off = w.outco.data();
        const int st = region_compute_parts(lane, c, w.rparts);
        if (st != PHMM_OK && st != PHMM_ERR_HIP) {  // somebody's region is at fault: find out whose
            for (Submission *s : subs) submission_alone(lane, s);
            return;
        }
        for (Submission *s : subs) {
            s->status = st;
            if (st != PHMM_OK) s->err = lane->err;
        }
        return;
    }
    if (subs[0]->engine) {
        const Submission &f = *subs[0];
        static const int which[5] = {0, 1, 2, 3, 5};
        for (int i = 0; i < 5; ++i) w.bytes[i].clear();
        w.mapq.clear();
        w.ref.clear();
        for (const Submission *s : subs) {
            for (int i = 0; i < 5; ++i)
                if (s->payload[which[i]])
                    w.bytes[i].insert(w.bytes[i].end(), s->payload[which[i]], s->payload[which[i]] + (i < 4 ? s->read_bytes : s->hap_bytes));
            w.mapq.insert(w.mapq.end(), s->mapq, s->mapq + s->n_reads);
            if (s->ref_hap) w.ref.insert(w.ref.end(), s->ref_hap, s->ref_hap + s->n_regions);
        }
        w.out.resize(w.oo.back());
        w.keep.resize(w.rro.back());
        int st = phmm_engine_compute(lane, &f.cfg, (uint32_t)w.rro.size() - 1, w.rro.data(), w.rho.data(), w.ro.data(),
                                     w.bytes[0].data(), w.bytes[1].data(), f.payload[2] ? w.bytes[2].data() : nullptr,
                                     f.payload[3] ? w.bytes[3].data() : nullptr, w.mapq.data(), w.ho.data(), w.bytes[4].data(),
                                     f.ref_hap ? w.ref.data() : nullptr, w.oo.data(), w.out.data(), w.keep.data());
        if (st != PHMM_OK && st != PHMM_ERR_HIP) {  // somebody's region is at fault: find out whose
            for (Submission *s : subs) submission_alone(lane, s);
            return;
        }
        size_t o = 0, r = 0;
        for (Submission *s : subs) {
            s->status = st;
            if (st != PHMM_OK) s->err = lane->err;
            if (st == PHMM_OK && s->n_out) memcpy(s->out, w.out.data() + o, s->n_out * 8);
            if (st == PHMM_OK && s->n_reads) memcpy(s->keep, w.keep.data() + r, s->n_reads);
            o += s->n_out;
            r += s->n_reads;
        }
        return;
    }
    lane->slot = 0;
    PendingCompute p;
    int st = enqueue_compute(lane, (uint32_t)w.rro.size() - 1, w.rro.data(), w.rho.data(), w.ro.data(), nullptr, nullptr, nullptr,
                             nullptr, nullptr, w.ho.data(), nullptr, w.oo.data(), nullptr, &p, &parts);
    if (st == PHMM_OK) st = finish_compute(lane, &p);
    if (st != PHMM_OK && st != PHMM_ERR_HIP) {
        // some region of the batch tripped the reference's assert (pair_hmm.rs:478-481): find out whose it was, the
        // other submitters get their (valid) results
        for (Submission *s : subs) submission_alone(lane, s);
        return;
    }
    for (Submission *s : subs) {
        s->status = st;
        if (st != PHMM_OK) s->err = lane->err;
    }
}

int submit_fail(phmm_handle *h, int code, const char *msg) {
    set_thread_error(h, msg);
    return code;
}

}  // namespace

namespace {

int submit_impl(phmm_handle *h, Submission &s, uint64_t *ticket) {
    std::call_once(h->comb_once, [h] {
        Combiner *c = new Combiner();
        c->n_lanes = 4;  // (2 / 3 / 6 / 8 lanes measured in round 5: none better at 8 or 16 callers, NOTEBOOK 19.2)
        c->trace = h->sw.trace != 0;
        h->comb = c;
    });
    Combiner *c = h->comb;
    std::lock_guard<std::mutex> lk(c->mu);
    for (int l = 0; l < c->n_lanes; ++l)
        if (!c->lane[l]) {  // first submission: the lanes are engines of their own on the same device
            c->lane[l] = create_internal(h->device, h->flags);
            if (!c->lane[l]) return submit_fail(h, PHMM_ERR_HIP, phmm_last_error(nullptr));
            c->lane[l]->sw = h->sw;  // the lanes follow the shared handle's developer switches
        }
    const uint64_t t = c->next_ticket++;
    c->live.emplace(t, std::move(s));
    c->queue.push_back(t);
    *ticket = t;
    if (c->gathering) c->gather_cv.notify_one();
    return PHMM_OK;
}

}  // namespace

extern "C" {

int phmm_submit(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q,
                const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off, const uint8_t *hap_bases,
                const uint64_t *out_off, double *out, uint64_t *ticket) {
    if (!h || !ticket) return PHMM_ERR_INVALID_ARG;
    if (const char *bad = validate_offsets(n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, nullptr))
        return submit_fail(h, PHMM_ERR_INVALID_ARG, bad);
    Submission s;
    s.n_regions = n_regions;
    s.n_reads = region_read_off[n_regions];
    s.n_haps = region_hap_off[n_regions];
    s.region_read_off = region_read_off;
    s.region_hap_off = region_hap_off;
    s.read_off = read_off;
    s.hap_off = hap_off;
    s.out_off = out_off;
    s.out = out;
    s.read_bytes = read_off[s.n_reads];
    s.hap_bytes = hap_off[s.n_haps];
    s.n_out = out_off[n_regions];
    const uint8_t *pl[6] = {read_bases, base_q, ins_q, del_q, gcp, hap_bases};
    for (int i = 0; i < 6; ++i) s.payload[i] = pl[i];
    if ((s.read_bytes && (!read_bases || !base_q || !ins_q || !del_q || !gcp)) || (s.hap_bytes && !hap_bases) ||
        (s.n_out && !out))
        return submit_fail(h, PHMM_ERR_INVALID_ARG, "phmm_submit: null pointer");
    return submit_impl(h, s, ticket);
}

int phmm_engine_submit(phmm_handle *h, const phmm_engine_config *cfg, uint32_t n_regions, const uint32_t *region_read_off,
                       const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases,
                       const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *mapq,
                       const uint32_t *hap_off, const uint8_t *hap_bases, const int32_t *region_ref_hap,
                       const uint64_t *out_off, double *out, uint8_t *keep, uint64_t *ticket) {
    if (!h || !cfg || !ticket) return PHMM_ERR_INVALID_ARG;
    if (cfg->pcr_error_model > 3) return submit_fail(h, PHMM_ERR_INVALID_ARG, "phmm_engine_compute: Unknown PCR Error Model");
    if (const char *bad = validate_offsets(n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, nullptr))
        return submit_fail(h, PHMM_ERR_INVALID_ARG, bad);
    Submission s;
    s.engine = true;
    s.cfg = *cfg;
    s.n_regions = n_regions;
    s.n_reads = region_read_off[n_regions];
    s.n_haps = region_hap_off[n_regions];
    s.region_read_off = region_read_off;
    s.region_hap_off = region_hap_off;
    s.read_off = read_off;
    s.hap_off = hap_off;
    s.out_off = out_off;
    s.out = out;
    s.read_bytes = read_off[s.n_reads];
    s.hap_bytes = hap_off[s.n_haps];
    s.n_out = out_off[n_regions];
    const uint8_t *pl[6] = {read_bases, base_q, ins_q, del_q, nullptr, hap_bases};
    for (int i = 0; i < 6; ++i) s.payload[i] = pl[i];
    s.mapq = mapq;
    s.ref_hap = region_ref_hap;
    s.keep = keep;
    if ((s.read_bytes && (!read_bases || !base_q)) || (s.n_reads && (!mapq || !keep)) || (s.hap_bytes && !hap_bases) ||
        (s.n_out && !out))
        return submit_fail(h, PHMM_ERR_INVALID_ARG, "phmm_engine_submit: null pointer");
    return submit_impl(h, s, ticket);
}

int phmm_region_submit(phmm_handle *h, const phmm_engine_config *cfg, const phmm_realign_config *rcfg, uint32_t n_regions,
                       const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                       const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q,
                       const uint8_t *mapq, const uint32_t *read_soft_clip, const uint32_t *hap_off, const uint8_t *hap_bases,
                       const int32_t *region_ref_hap, const uint64_t *out_off, const int32_t *hap_priority,
                       const uint64_t *region_reference_start, const uint32_t *hap_cigar_off, const uint32_t *hap_cigar,
                       const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar,
                       const uint64_t *out_cigar_off, double *out, uint8_t *keep, int32_t *best_allele, double *likelihood,
                       double *confidence, uint32_t *out_cigar, uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status,
                       uint64_t *ticket) {
    if (!h || !cfg || !rcfg || !ticket) return PHMM_ERR_INVALID_ARG;
    try {
        Submission s;
        s.region = true;
        RegionArgs &a = s.ra;
        a.cfg = *cfg;
        a.rcfg = *rcfg;
        a.n_regions = n_regions;
        a.region_read_off = region_read_off;
        a.region_hap_off = region_hap_off;
        a.read_off = read_off;
        a.read_bases = read_bases;
        a.base_q = base_q;
        a.ins_q = ins_q;
        a.del_q = del_q;
        a.mapq = mapq;
        a.read_soft_clip = read_soft_clip;
        a.hap_off = hap_off;
        a.hap_bases = hap_bases;
        a.region_ref_hap = region_ref_hap;
        a.out_off = out_off;
        a.hap_priority = hap_priority;
        a.region_reference_start = region_reference_start;
        a.hap_cigar_off = hap_cigar_off;
        a.hap_cigar = hap_cigar;
        a.hap_start_wrt_ref = hap_start_wrt_ref;
        a.orig_cigar_off = orig_cigar_off;
        a.orig_cigar = orig_cigar;
        a.out_cigar_off = out_cigar_off;
        a.out = out;
        a.keep = keep;
        a.best_allele = best_allele;
        a.likelihood = likelihood;
        a.confidence = confidence;
        a.out_cigar = out_cigar;
        a.n_out_cigar = n_out_cigar;
        a.new_pos = new_pos;
        a.status = status;
        const std::string bad = region_validate(a);
        if (!bad.empty()) return submit_fail(h, PHMM_ERR_INVALID_ARG, bad.c_str());
        // The device's resident server takes the call if it is within its limits (phmm_server.cpp): staged by THIS thread, one
        // ring entry, no flush to wait for; phmm_wait polls for its finish word.  The ticket's top bit tells the two kinds apart.
        {
            ServerPending *pending = nullptr;
            const int st = server_region_submit(h, a, &pending, true);
            if (st == PHMM_OK) {
                *ticket = kServerTicket | (uint64_t)(uintptr_t)pending;
                return PHMM_OK;
            }
            if (st != kServerNotTaken) return submit_fail(h, st, "phmm_region_submit: the region server refused the call");
        }
        s.n_regions = n_regions;
        s.n_reads = region_read_off[n_regions];
        s.n_haps = region_hap_off[n_regions];
        s.read_bytes = read_off[s.n_reads];
        return submit_impl(h, s, ticket);
    } catch (const std::exception &e) {
        return submit_fail(h, PHMM_ERR_NO_MEMORY, e.what());
    }
}

int phmm_wait(phmm_handle *h, uint64_t ticket) {
    if (!h) return PHMM_ERR_INVALID_ARG;
    if (ticket & kServerTicket) {  // a call the region server has (phmm_region_submit)
        ServerPending *pending = (ServerPending *)(uintptr_t)(ticket & ~kServerTicket);
        std::string err;
        RegionArgs again;
        int st = server_region_wait(h, pending, &err, &again);
        if (st == kServerRedo) {  // (an alignment outgrew its slot: once more through the combiner, whose pipeline grows the slots)
            try {
                Submission s;
                s.region = true;
                s.ra = again;
                s.n_regions = again.n_regions;
                s.n_reads = again.region_read_off[again.n_regions];
                s.n_haps = again.region_hap_off[again.n_regions];
                s.read_bytes = again.read_off[s.n_reads];
                uint64_t t2 = 0;
                st = submit_impl(h, s, &t2);
                return st == PHMM_OK ? phmm_wait(h, t2) : st;
            } catch (const std::exception &e) {
                return submit_fail(h, PHMM_ERR_NO_MEMORY, e.what());
            }
        }
        if (st != PHMM_OK)
            set_thread_error(h, err);
        else
            clear_thread_error(h);
        return st;
    }
    Combiner *c = h->comb;
    if (!c) return submit_fail(h, PHMM_ERR_INVALID_ARG, "phmm_wait: nothing was submitted on this handle");
    std::unique_lock<std::mutex> lk(c->mu);
    auto it = c->live.find(ticket);
    if (it == c->live.end()) return submit_fail(h, PHMM_ERR_INVALID_ARG, "phmm_wait: unknown ticket (already waited for?)");
    Submission *me = &it->second;
    std::vector<Submission *> subs;
    for (;;) {
        if (me->state == Submission::DONE) {
            const int st = me->status;
            if (st != PHMM_OK)
                set_thread_error(h, me->err);
            else
                clear_thread_error(h);
            c->live.erase(ticket);
            return st;
        }
        int lane = -1;
        if (me->state == Submission::QUEUED && !c->gathering)
            for (int l = 0; l < c->n_lanes && lane < 0; ++l)
                if (!c->lane_busy[l]) lane = l;
        if (lane < 0) {  // my region is in somebody's flush (or about to be), or every lane is taken: the finishing leader wakes me
            me->waiting = true;
            me->cv->wait(lk);
            me->waiting = false;
            continue;
        }
        // Under load the workers a finished flush has just released re-submit within microseconds of each other; the first
        // of them to get here would lead a flush of one or two regions and leave the others to wait for the next free lane
        // (32 workers, 4 lanes: 5.3 regions per flush where 8 are outstanding per lane).  So a leader that sees more work
        // outstanding than is queued lets it arrive: until the queue holds an equal share of everything outstanding, for
        // as long as submissions keep coming (15 us without one ends it), at most gather_us.  A lone caller, or as many
        // workers as lanes, never waits: their share is one region.
        {
            // (the per-region pipeline is a chain of five dependent kernels, ~140 us however small the call: a flush of four
            // regions costs 1.5 x one region's time, so from eight callers on four go together before lanes are spread -- 8
            // threads: 2 lanes x 4 regions instead of 4 x 2, 28.2 -> 30.0 k regions/s; with fewer callers the wait costs more
            // than the sharing brings: 4 threads 19.3 -> 15.2 k)
            const size_t even = (c->live.size() + (size_t)c->n_lanes - 1) / (size_t)c->n_lanes;
            const size_t share = me->region && c->live.size() >= 8 ? std::max<size_t>(even, 4) : even;
            if (c->gather_us > 0 && share > 1 && c->queue.size() < share) {
                const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(c->gather_us);
                c->gathering = true;
                for (;;) {
                    const size_t before = c->queue.size();
                    const auto t_now = std::chrono::steady_clock::now();
                    if (before >= share || t_now >= t_end) break;
                    c->gather_cv.wait_until(lk, std::min(t_end, t_now + std::chrono::microseconds(15)));
                    if (c->queue.size() == before) break;  // nobody came
                }
                c->gathering = false;
            }
        }
        // Lead one flush: everything queued so far, in order, while it fits one staging pass.  (Measured and dropped:
        // polling instead of sleeping, and a leader that keeps the lane for further flushes -- neither changes the
        // rates of tools/threads_bench, the flush itself is what takes the time.)
        subs.clear();
        size_t bytes = 0;
        while (!c->queue.empty() && subs.size() < Combiner::kMaxParts) {
            Submission *s = &c->live.find(c->queue.front())->second;
            if (!subs.empty() && (bytes + s->read_bytes > kCombineBytes || !s->compatible(*subs[0]))) break;
            bytes += s->read_bytes;
            s->state = Submission::RUNNING;
            subs.push_back(s);
            c->queue.pop_front();
        }
        c->lane_busy[lane] = true;
        {   // (what did not fit this flush may find another free lane)
            bool free_lane = false;
            for (int l = 0; l < c->n_lanes; ++l) free_lane |= !c->lane_busy[l];
            if (free_lane) wake_queue_head(c);
        }
        {
            // A combined flush means the handle is under load: plan it for the share of the chip it will get (the lanes
            // computing right now, this one included) rather than for an empty chip -- 16 threads 53-60 k -> 59-68 k
            // regions/s.  Single regions keep the wide latency shape: narrowing those costs 25 % at 2-4 threads.
            uint32_t busy = 0;
            for (int l = 0; l < c->n_lanes; ++l) busy += c->lane_busy[l] ? 1u : 0u;
            c->lane[lane]->gpu_sharers = subs.size() >= 2 ? busy : 1u;
            c->lane[lane]->busy_lanes = busy;
        }
        c->n_flushes += 1;
        c->n_parts += subs.size();
        auto nowus = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tb = nowus();
        lk.unlock();
        run_flush(c->lane[lane], c->scratch[lane], subs);
        lk.lock();
        c->flush_us += nowus() - tb;
        c->lane_busy[lane] = false;
        for (Submission *s : subs) {
            s->state = Submission::DONE;
            if (s != me) s->cv->notify_one();
        }
        wake_queue_head(c);  // this lane is free again
    }
}

void phmm_submit_stats(phmm_handle *h, uint64_t *n_flushes, uint64_t *n_submissions) {
    uint64_t f = 0, n = 0;
    if (h && h->comb) {
        std::lock_guard<std::mutex> lk(h->comb->mu);
        f = h->comb->n_flushes;
        n = h->comb->n_parts;
    }
    if (n_flushes) *n_flushes = f;
    if (n_submissions) *n_submissions = n;
}

}  // extern "C"
