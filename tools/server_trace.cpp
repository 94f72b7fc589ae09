// Developer tool: where the time of region calls goes inside the resident region server (lorikeet_amd/csrc/phmm_server.cpp).
// T threads call phmm_region_compute on regions of one shape for a while with PHMM_SERVER_TRACE=1; then, per kind of task:
// how many ran, their mean duration, the mean wait between a stage becoming claimable and its tasks starting; per call: the
// time from its first task to its last; over the run: how many of the chip's worker waves were inside a task (busy) at a time.
// usage: server_trace [threads] [calls per thread] [Nr Nh R H]
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

#include "../include/phmm.h"

struct Rec {
    uint32_t seq, kind, idx, worker;
    uint64_t t_claim, t_begin, t_end;
    uint64_t t_mid[4];  // chain tasks: pre-step done, PairHMM done, helpers in + post-step done, aligner done
};
struct Region {
    std::vector<uint32_t> rro, rho, ro, ho, hco, hc, hs, oco, oc, cig, ncig;
    std::vector<uint64_t> oo, rstart, outco;
    std::vector<uint8_t> bases, q, iq, dq, haps, mapq, keep;
    std::vector<int32_t> refhap, best, status;
    std::vector<int64_t> pos;
    std::vector<double> out, lk, conf;
};
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() {
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (uint32_t)(rs >> 11);
}
static Region make(uint32_t nr, uint32_t nh, uint32_t R, uint32_t H) {
    Region g;
    const char A[] = "ACGT";
    g.rro = {0, nr};
    g.rho = {0, nh};
    g.oo = {0, (uint64_t)nr * nh};
    g.ro.resize(nr + 1);
    g.ho.resize(nh + 1);
    for (uint32_t i = 0; i <= nr; ++i) g.ro[i] = i * R;
    for (uint32_t i = 0; i <= nh; ++i) g.ho[i] = i * H;
    g.haps.resize((size_t)nh * H);
    for (uint32_t j = 0; j < H; ++j) g.haps[j] = A[rnd() & 3];
    for (uint32_t a = 1; a < nh; ++a)
        for (uint32_t j = 0; j < H; ++j) g.haps[a * H + j] = rnd() % 50 == 0 ? A[rnd() & 3] : g.haps[j];
    g.bases.resize((size_t)nr * R);
    g.q.assign((size_t)nr * R, 30);
    g.iq.assign((size_t)nr * R, 45);
    g.dq.assign((size_t)nr * R, 45);
    for (uint32_t r = 0; r < nr; ++r) {
        const uint32_t a = rnd() % nh, at = rnd() % (H - R + 1);
        for (uint32_t i = 0; i < R; ++i) g.bases[r * R + i] = rnd() % 100 == 0 ? A[rnd() & 3] : g.haps[a * H + at + i];
    }
    g.mapq.assign(nr, 60);
    g.keep.resize(nr);
    g.refhap = {0};
    g.rstart = {1000};
    g.hco.resize(nh + 1);
    g.hc.resize(nh);
    g.hs.assign(nh, 0);
    for (uint32_t a = 0; a <= nh; ++a) g.hco[a] = a;
    for (uint32_t a = 0; a < nh; ++a) g.hc[a] = H << 4;
    g.oco.resize(nr + 1);
    g.oc.resize(nr);
    g.outco.resize(nr + 1);
    for (uint32_t r = 0; r <= nr; ++r) g.oco[r] = r, g.outco[r] = 16ull * r;
    for (uint32_t r = 0; r < nr; ++r) g.oc[r] = R << 4;
    g.cig.resize(16 * nr);
    g.ncig.resize(nr);
    g.best.resize(nr);
    g.status.resize(nr);
    g.pos.resize(nr);
    g.out.resize((size_t)nr * nh);
    g.lk.resize(nr);
    g.conf.resize(nr);
    return g;
}
static int call(phmm_handle *h, Region &g) {
    phmm_engine_config cfg{};
    cfg.constant_gcp = 10;
    cfg.pcr_error_model = 3;
    cfg.base_quality_score_threshold = 18;
    cfg.symmetrically_normalize_alleles_to_reference = 1;
    cfg.log10_global_read_mismapping_rate = -4.5;
    cfg.read_disqualification_scale = 1.0;
    cfg.expected_error_rate_per_base = 0.02;
    phmm_realign_config rc{};
    rc.sw_parameters = {10, -15, -30, -5};
    rc.overhang_strategy = PHMM_SW_SOFTCLIP;
    rc.informative_threshold = 0.2;
    return phmm_region_compute(h, &cfg, &rc, 1, g.rro.data(), g.rho.data(), g.ro.data(), g.bases.data(), g.q.data(), g.iq.data(), g.dq.data(), g.mapq.data(),
                               nullptr, g.ho.data(), g.haps.data(), g.refhap.data(), g.oo.data(), nullptr, g.rstart.data(), g.hco.data(), g.hc.data(), g.hs.data(),
                               g.oco.data(), g.oc.data(), g.outco.data(), g.out.data(), g.keep.data(), g.best.data(), g.lk.data(), g.conf.data(), g.cig.data(),
                               g.ncig.data(), g.pos.data(), g.status.data());
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 1, calls = argc > 2 ? atoi(argv[2]) : 50;
    const uint32_t nr = argc > 6 ? atoi(argv[3]) : 128, nh = argc > 6 ? atoi(argv[4]) : 8, R = argc > 6 ? atoi(argv[5]) : 150, H = argc > 6 ? atoi(argv[6]) : 300;
    setenv("PHMM_SERVER_TRACE", "1", 1);
    setenv("PHMM_REGION_SERVER", "1", 1);  // (every call through the server, a lone thread's too)
    if (!getenv("PHMM_SERVER_IDLE_US")) setenv("PHMM_SERVER_IDLE_US", "20000", 1);  // (one launch for the whole run: the trace starts over with every launch)
    std::vector<phmm_handle *> hs(T);
    std::vector<Region> gs;
    for (int t = 0; t < T; ++t) {
        hs[t] = phmm_create(0, 0);
        if (!hs[t]) return fprintf(stderr, "phmm_create: %s\n", phmm_last_error(nullptr)), 1;
        gs.push_back(make(nr, nh, R, H));
    }
    for (int t = 0; t < T; ++t)
        if (call(hs[t], gs[t]) != PHMM_OK) return fprintf(stderr, "call: %s\n", phmm_last_error(hs[t])), 1;
    std::this_thread::sleep_for(std::chrono::milliseconds(100));  // (the warm-up's launch has left: the run below is one launch of its own)
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (int k = 0; k < calls; ++k)
                if (call(hs[t], gs[t]) != PHMM_OK) fprintf(stderr, "call: %s\n", phmm_last_error(hs[t]));
        });
    for (auto &x : th) x.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<Rec> recs(1u << 20);
    const uint32_t n = std::min<uint32_t>(phmm_server_trace(0, recs.data(), (uint32_t)recs.size()), (uint32_t)recs.size());
    recs.resize(n);
    printf("%d thread(s) x %d calls of %u x %u (R %u, H %u): %.0f regions/s, %.1f us per call per thread; %u task records, server launches %llu, jobs %llu\n",
           T, calls, nr, nh, R, H, T * calls / secs, secs / calls * 1e6, n, (unsigned long long)phmm_get_stat(hs[0], "server_launches"),
           (unsigned long long)phmm_get_stat(hs[0], "server_jobs"));
    {
        const double jobs = (double)phmm_get_stat(hs[0], "server_jobs");
        printf("host side, mean per call: staging + publishing %.1f us, polling %.1f us, results out %.1f us\n", phmm_get_stat(hs[0], "server_stage_ns") / jobs / 1e3,
               phmm_get_stat(hs[0], "server_wait_ns") / jobs / 1e3, phmm_get_stat(hs[0], "server_out_ns") / jobs / 1e3);
    }
    if (!n) return 0;
    const char *names[] = {"stage-in", "chain"};
    // per call and kind: first claim, first begin, last end
    struct Span {
        uint64_t first_claim = ~0ull, first_begin = ~0ull, last_end = 0, busy = 0;
        uint32_t n = 0;
    };
    std::map<uint32_t, std::vector<Span>> by_call;
    uint64_t t_min = ~0ull, t_max = 0;
    double main_n = 0, helper_n = 0, ph[6] = {0, 0, 0, 0, 0, 0}, helper_us = 0;
    for (const Rec &r : recs) {
        auto &v = by_call[r.seq];
        v.resize(2);
        Span &s = v[r.kind % 2];
        s.first_claim = std::min(s.first_claim, r.t_claim);
        s.first_begin = std::min(s.first_begin, r.t_begin);
        s.last_end = std::max(s.last_end, r.t_end);
        s.busy += r.t_end - r.t_begin;
        s.n += 1;
        t_min = std::min(t_min, r.t_begin);
        t_max = std::max(t_max, r.t_end);
        if (r.kind == 1 && r.t_mid[3]) {  // a main wave
            main_n += 1;
            ph[0] += r.t_mid[0] - r.t_begin;
            ph[1] += r.t_mid[1] - r.t_mid[0];
            ph[2] += r.t_mid[2] - r.t_mid[1];
            ph[3] += r.t_mid[3] - r.t_mid[2];
            ph[4] += r.t_end - r.t_mid[3];
            ph[5] += r.t_end - r.t_begin;
        } else if (r.kind == 1) {
            helper_n += 1;
            helper_us += r.t_end - r.t_begin;
        }
    }
    printf("%-10s %8s %12s %14s %16s %14s\n", "kind", "tasks", "mean task us", "span us", "start after prev", "claim -> begin");
    for (int k = 0; k < 2; ++k) {
        double tasks = 0, dur = 0, span = 0, gap = 0, cb = 0;
        int calls_with = 0;
        for (auto &c : by_call) {
            const Span &s = c.second[k];
            if (!s.n) continue;
            calls_with += 1;
            tasks += s.n;
            dur += (double)s.busy / s.n;
            span += (double)(s.last_end - s.first_begin);
            gap += k ? (double)((int64_t)s.first_begin - (int64_t)c.second[0].last_end) : 0;
            cb += (double)(s.first_begin - s.first_claim);
        }
        if (!calls_with) continue;
        printf("%-10s %8.0f %12.2f %14.2f %16.2f %14.2f\n", names[k], tasks / calls_with, dur / calls_with / 100, span / calls_with / 100, gap / calls_with / 100,
               cb / calls_with / 100);
    }
    if (main_n)
        printf("a read's main wave, mean us: inputs + pre-step %.1f, PairHMM %.1f, helpers in + post-step %.1f, aligner %.1f, projection + out %.1f = %.1f; a helper wave %.1f\n",
               ph[0] / main_n / 100, ph[1] / main_n / 100, ph[2] / main_n / 100, ph[3] / main_n / 100, ph[4] / main_n / 100, ph[5] / main_n / 100,
               helper_n ? helper_us / helper_n / 100 : 0.0);
    double in_server = 0;
    for (auto &c : by_call) {
        uint64_t b = ~0ull, e = 0;
        for (const Span &s : c.second)
            if (s.n) b = std::min(b, s.first_claim), e = std::max(e, s.last_end);
        in_server += (double)(e - b);
    }
    printf("first claim -> last task end, mean over %zu calls: %.1f us\n", by_call.size(), in_server / by_call.size() / 100);
    // occupancy: worker-waves inside a task over the run (sampled per 1 us bucket)
    const uint64_t span = t_max - t_min;
    uint64_t busy_ticks = 0;
    for (const Rec &r : recs) busy_ticks += r.t_end - r.t_begin;
    std::vector<int32_t> delta((size_t)(span / 100) + 2, 0);
    for (const Rec &r : recs) {
        delta[(size_t)((r.t_begin - t_min) / 100)] += 1;
        delta[(size_t)((r.t_end - t_min) / 100) + 1] -= 1;
    }
    size_t idle_buckets = 0, level = 0;
    for (size_t i = 0; i + 1 < delta.size(); ++i) {
        level += delta[i];
        if (level == 0) idle_buckets += 1;
    }
    printf("run %.0f us on the device: mean %.1f worker waves inside a task, no task running %.1f %% of the time\n", span / 100.0, (double)busy_ticks / span,
           100.0 * idle_buckets / (delta.size() - 1));
    for (auto h : hs) phmm_destroy(h);
    return 0;
}
