// Developer tool: the reference's call pattern without an interpreter in the way -- T host threads, ONE region per call
// (host buffers, PCIe included), as rayon workers call PairHMM::compute_likelihoods (reference pair_hmm.rs:345-375 from
// assembly_region_walker.rs:210-273).  Two ways to serve it through include/phmm.h:
//   own     every thread has its own handle and calls phmm_compute
//   shared  all threads share one handle and call phmm_submit + phmm_wait (cross-thread batching)
//   pipeline (TB_MODE=pipeline only)  own handles; per call phmm_compute and then phmm_realign_reads with its likelihoods
//   realign  (TB_MODE=realign only)   own handles; phmm_realign_reads alone, on likelihoods computed beforehand
//   fused    (TB_MODE=fused only)     own handles; phmm_region_compute: pre-step, PairHMM, post-step, best alleles, alignments,
//                                     projection in ONE call -- the likelihoods never leave the device in between
//   gshared  (TB_MODE=gshared only)   one shared handle; phmm_region_submit + phmm_wait (the fused call, batched across threads)
// usage: threads_bench [seconds per point] [Nr Nh R H [regions per call]]      (default 1.0 s, 128 8 150 300 1 = config 2)
// env: TB_THREADS=4,8,16 (thread counts), TB_MODE=own|shared (only that mode), TB_FLAGS=<phmm_create flags>,
//      TB_SHAPE=ragged (every call a region of the long-tailed mix), TB_DEVICES=n (the binding's pattern on a multi-GPU node,
//      integration/hip_backend.rs: thread t works on device t % n -- its own handle there, or that device's ONE shared handle),
//      TB_DEPTH=2 (shared / gshared: every thread keeps two tickets in flight -- submits region k+1 before it waits for region k)
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <algorithm>
#include <thread>
#include <vector>

#include <cstring>
#include "../include/phmm.h"

struct Region {
    std::vector<uint32_t> rro, rho, ro, ho;
    std::vector<uint64_t> oo;
    std::vector<uint8_t> bases, q, iq, dq, gcp, haps, mapq, keep;
    std::vector<double> out;
    uint64_t cells = 0;
    // what phmm_realign_reads takes besides (mode "pipeline"): priorities, reference haplotype / start per region, the
    // haplotypes' CIGARs (SNVs only: one M element), the reads' original CIGARs, and room for the results
    std::vector<int32_t> pri, ref_hap, best, status;
    std::vector<uint64_t> rstart, out_cig_off;
    std::vector<uint32_t> hc_off, hc, hs, oc_off, oc, cig, n_cig;
    std::vector<int64_t> pos;
    std::vector<double> lk, conf;
};

static uint64_t rng_state;
static uint32_t rnd() {  // splitmix64
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}

static double rnd_normal() {  // Box-Muller
    const double u = (rnd() + 1.0) / 4294967296.0, v = rnd() / 4294967296.0;  // u in (0, 1]
    return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v);
}

// `per_call` regions of the same shape, as one call's arrays.  TB_SHAPE=ragged (nr_in == 0): every region draws its own shape
// from the long-tailed mix of the bench's ragged workload (lorikeet_amd/synthetic.py: reads per region log-normal around 60,
// haplotypes log-normal around 5, haplotype length 60 ... 500, read lengths 30 ... 250 mixed inside a region; the tail is
// cut at 2 000 reads so that a thread's cycle of regions stays a few megabytes).
static Region make_region(uint64_t seed, int nr_in, int nh_in, int R_in, int H_in, int per_call) {
    rng_state = seed;
    Region g;
    const bool ragged = nr_in == 0;
    int nr = nr_in, nh = nh_in, R = R_in, H = H_in;
    std::vector<uint32_t> read_len;
    const char acgt[] = "ACGT";
    g.ho.push_back(0);
    g.ro.push_back(0);
    g.rro.push_back(0);
    g.rho.push_back(0);
    g.oo.push_back(0);
  size_t n_reads = 0, n_haps = 0;
  for (int reg = 0; reg < per_call; ++reg) {
    if (ragged) {
        nr = (int)std::fmin(2000.0, std::fmax(3.0, std::exp(std::log(60.0) + 1.3 * rnd_normal())));
        nh = (int)std::fmin(128.0, std::fmax(1.0, std::exp(std::log(5.0) + 0.9 * rnd_normal())));
        H = 60 + (int)(rnd() % 441);
    }
    n_reads += (size_t)nr;
    n_haps += (size_t)nh;
    std::vector<uint8_t> root(H);
    for (auto &b : root) b = acgt[rnd() & 3];
    const size_t hap0 = g.haps.size();
    for (int a = 0; a < nh; ++a) {
        std::vector<uint8_t> h = root;
        if (a)
            for (int k = 0; k < 1 + (int)(rnd() % 3); ++k) h[rnd() % H] = acgt[rnd() & 3];
        g.haps.insert(g.haps.end(), h.begin(), h.end());
        g.ho.push_back((uint32_t)g.haps.size());
    }
    uint64_t sum_r = 0;
    for (int r = 0; r < nr; ++r) {
        if (ragged) R = 30 + (int)(rnd() % (uint32_t)(std::min(250, H) - 30 + 1));
        sum_r += (uint64_t)R;
        read_len.push_back((uint32_t)R);
        const int a = rnd() % nh, s = rnd() % (H - R + 1);
        for (int i = 0; i < R; ++i) {
            uint8_t b = g.haps[hap0 + (size_t)a * H + s + i];
            if (rnd() % 100 == 0) b = acgt[rnd() & 3];
            g.bases.push_back(b);
            const uint32_t u = rnd() % 100;
            g.q.push_back(u < 60 ? 37 : u < 75 ? 32 : u < 85 ? 27 : u < 93 ? 22 : 6);
            g.iq.push_back(rnd() % 10 ? 40 : 30 + rnd() % 10);
            g.dq.push_back(rnd() % 10 ? 40 : 30 + rnd() % 10);
            g.gcp.push_back(10);
        }
        g.ro.push_back((uint32_t)g.bases.size());
    }
    g.rro.push_back(g.rro.back() + nr);
    g.rho.push_back(g.rho.back() + nh);
    g.oo.push_back(g.oo.back() + (uint64_t)nr * nh);
    g.cells += sum_r * (uint64_t)nh * H;
    for (int a = 0; a < nh; ++a) g.hc.push_back((uint32_t)H << 4);
  }
    g.out.assign((size_t)g.oo.back(), 0.0);
    // (distinct priorities inside a region, the first haplotype -- the reference -- highest, as haplotype_alignment_tiebreaking
    // gives them: alleles whose likelihoods tie to within the informative threshold are then chosen by priority, not by the
    // last bits of an f64 sum -- which differ by lane geometry, i.e. by who else is in a combined flush)
    g.pri.assign(n_haps, 0);
    for (size_t reg = 0; reg + 1 < g.rho.size(); ++reg)
        for (uint32_t k = g.rho[reg]; k < g.rho[reg + 1]; ++k) g.pri[k] = (int32_t)(g.rho[reg + 1] - k);
    g.ref_hap.assign(per_call, 0);
    for (int reg = 0; reg < per_call; ++reg) g.rstart.push_back(1000ull * (reg + 1));
    for (size_t a = 0; a <= n_haps; ++a) g.hc_off.push_back((uint32_t)a);
    g.hs.assign(n_haps, 0);
    for (size_t r = 0; r <= n_reads; ++r) g.oc_off.push_back((uint32_t)r), g.out_cig_off.push_back(8ull * r);
    for (size_t r = 0; r < n_reads; ++r) g.oc.push_back(read_len[r] << 4);
    g.cig.assign(8 * n_reads, 0);
    g.n_cig.assign(n_reads, 0);
    g.pos.assign(n_reads, 0);
    g.status.assign(n_reads, 0);
    g.best.assign(n_reads, 0);
    g.lk.assign(n_reads, 0.0);
    g.conf.assign(n_reads, 0.0);
    g.mapq.assign(n_reads, 60);
    g.keep.assign(n_reads, 0);
    return g;
}

static int call_own(phmm_handle *h, Region &g) {
    return phmm_compute(h, (uint32_t)g.rro.size() - 1, g.rro.data(), g.rho.data(), g.ro.data(), g.bases.data(), g.q.data(), g.iq.data(), g.dq.data(),
                        g.gcp.data(), g.ho.data(), g.haps.data(), g.oo.data(), g.out.data());
}

static int submit_shared(phmm_handle *h, Region &g, uint64_t *t) {
    return phmm_submit(h, (uint32_t)g.rro.size() - 1, g.rro.data(), g.rho.data(), g.ro.data(), g.bases.data(), g.q.data(), g.iq.data(), g.dq.data(),
                       g.gcp.data(), g.ho.data(), g.haps.data(), g.oo.data(), g.out.data(), t);
}
static int call_shared(phmm_handle *h, Region &g) {
    uint64_t t = 0;
    int st = submit_shared(h, g, &t);
    return st ? st : phmm_wait(h, t);
}

// likelihoods, then the reads realigned with them (best alleles, alignments, projection onto the reference)
static int call_pipeline(phmm_handle *h, Region &g) {
    int st = call_own(h, g);
    if (st) return st;
    static const phmm_sw_parameters prm{10, -15, -30, -5};
    return phmm_realign_reads(h, (uint32_t)g.rro.size() - 1, g.rro.data(), g.rho.data(), g.ro.data(), g.bases.data(), g.ho.data(), g.haps.data(),
                              g.oo.data(), g.out.data(), nullptr, g.pri.data(), 0.2, &prm, PHMM_SW_SOFTCLIP, g.ref_hap.data(), g.rstart.data(),
                              g.hc_off.data(), g.hc.data(), g.hs.data(), g.oc_off.data(), g.oc.data(), g.out_cig_off.data(), g.cig.data(),
                              g.n_cig.data(), g.pos.data(), g.status.data(), g.best.data(), g.lk.data(), g.conf.data());
}

static int call_realign(phmm_handle *h, Region &g) {
    static const phmm_sw_parameters prm{10, -15, -30, -5};
    return phmm_realign_reads(h, (uint32_t)g.rro.size() - 1, g.rro.data(), g.rho.data(), g.ro.data(), g.bases.data(), g.ho.data(), g.haps.data(),
                              g.oo.data(), g.out.data(), nullptr, g.pri.data(), 0.2, &prm, PHMM_SW_SOFTCLIP, g.ref_hap.data(), g.rstart.data(),
                              g.hc_off.data(), g.hc.data(), g.hs.data(), g.oc_off.data(), g.oc.data(), g.out_cig_off.data(), g.cig.data(),
                              g.n_cig.data(), g.pos.data(), g.status.data(), g.best.data(), g.lk.data(), g.conf.data());
}

// the whole per-region path in one call (the engine-level pre- and post-step included)
static const phmm_engine_config kCfg{10, 3, 18, 0, 1, 0, {0, 0}, -4.5, 1.0, 0.02};
static const phmm_realign_config kRcfg{{10, -15, -30, -5}, PHMM_SW_SOFTCLIP, 0u, 0.2};
static int call_fused(phmm_handle *h, Region &g) {
    return phmm_region_compute(h, &kCfg, &kRcfg, (uint32_t)g.rro.size() - 1, g.rro.data(), g.rho.data(), g.ro.data(), g.bases.data(), g.q.data(),
                               g.iq.data(), g.dq.data(), g.mapq.data(), nullptr, g.ho.data(), g.haps.data(), g.ref_hap.data(), g.oo.data(), g.pri.data(),
                               g.rstart.data(), g.hc_off.data(), g.hc.data(), g.hs.data(), g.oc_off.data(), g.oc.data(), g.out_cig_off.data(),
                               g.out.data(), g.keep.data(), g.best.data(), g.lk.data(), g.conf.data(), g.cig.data(), g.n_cig.data(), g.pos.data(),
                               g.status.data());
}
static int submit_fused_shared(phmm_handle *h, Region &g, uint64_t *t) {
    return phmm_region_submit(h, &kCfg, &kRcfg, (uint32_t)g.rro.size() - 1, g.rro.data(), g.rho.data(), g.ro.data(), g.bases.data(), g.q.data(),
                              g.iq.data(), g.dq.data(), g.mapq.data(), nullptr, g.ho.data(), g.haps.data(), g.ref_hap.data(), g.oo.data(),
                              g.pri.data(), g.rstart.data(), g.hc_off.data(), g.hc.data(), g.hs.data(), g.oc_off.data(), g.oc.data(),
                              g.out_cig_off.data(), g.out.data(), g.keep.data(), g.best.data(), g.lk.data(), g.conf.data(), g.cig.data(),
                              g.n_cig.data(), g.pos.data(), g.status.data(), t);
}
static int call_fused_shared(phmm_handle *h, Region &g) {
    uint64_t t = 0;
    int st = submit_fused_shared(h, g, &t);
    return st ? st : phmm_wait(h, t);
}

int main(int argc, char **argv) {
    const double dur = argc > 1 ? atof(argv[1]) : 1.0;
    const int nr = argc > 5 ? atoi(argv[2]) : 128, nh = argc > 5 ? atoi(argv[3]) : 8, R = argc > 5 ? atoi(argv[4]) : 150,
              H = argc > 5 ? atoi(argv[5]) : 300, per_call = argc > 6 ? atoi(argv[6]) : 1;
    const bool ragged = getenv("TB_SHAPE") && std::string(getenv("TB_SHAPE")) == "ragged";
    const int cycle = ragged ? 32 : 4;  // regions a thread cycles through
    if (phmm_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    if (ragged)
        printf("%d region(s) per call: the ragged mix (reads ~ logN(60), haplotypes ~ logN(5), H 60-500, R 30-250), %.1f s per point\n", per_call, dur);
    else
        printf("%d region(s) per call: %d reads x %d haplotypes, R=%d, H=%d (%.3g cells each), %.1f s per point\n", per_call, nr, nh,
               R, H, (double)nr * R * nh * H, dur);
    std::vector<int> Ts = {1, 2, 4, 8, 16, 32, 64};
    if (const char *e = getenv("TB_THREADS")) {  // e.g. TB_THREADS=4,8,16
        Ts.clear();
        for (const char *q = e; *q; q += (*q == ',')) {
            Ts.push_back((int)strtol(q, (char **)&q, 10));
            if (Ts.back() < 1) return 2;
        }
    }
    const bool verify = getenv("TB_VERIFY") != nullptr;
    const int depth = std::max(1, std::min(getenv("TB_DEPTH") ? atoi(getenv("TB_DEPTH")) : 1, 4));
    const char *only = getenv("TB_MODE");  // "own", "shared" or "pipeline": just that one (pipeline only when asked for)
    for (int mode = 0; mode < 6; ++mode) {
        if (only ? only[0] != "osprfg"[mode] : mode >= 2) continue;
        const bool one_handle = mode == 1 || mode == 5;
        for (int T : Ts) {
            std::vector<phmm_handle *> hs;
            const int n_dev = std::max(1, std::min(getenv("TB_DEVICES") ? atoi(getenv("TB_DEVICES")) : 1, phmm_device_count()));
            for (int i = 0; i < (!one_handle ? T : n_dev); ++i) {
                hs.push_back(phmm_create(i % n_dev, getenv("TB_FLAGS") ? (unsigned)atoi(getenv("TB_FLAGS")) : 0u));
                if (!hs.back()) {
                    fprintf(stderr, "phmm_create: %s\n", phmm_last_error(nullptr));
                    return 2;
                }
            }
            // every thread cycles through regions of its own (different data, same shape)
            std::vector<std::vector<Region>> regs(T);
            for (int t = 0; t < T; ++t)
                for (int k = 0; k < cycle; ++k) regs[t].push_back(make_region(1000 + 64 * t + k, ragged ? 0 : nr, nh, R, H, per_call));
            std::atomic<uint64_t> n_calls{0}, n_cells{0};
            std::atomic<int> failed{0};
            std::atomic<bool> go{false}, stop{false};
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    phmm_handle *h = hs[!one_handle ? t : t % n_dev];
                    auto call = mode == 0 ? call_own : mode == 1 ? call_shared : mode == 2 ? call_pipeline : mode == 3 ? call_realign : mode == 4 ? call_fused : call_fused_shared;
                    for (int k = 0; k < cycle; ++k)  // warm the arenas (and compute the likelihoods mode "realign" starts from)
                        if ((mode == 3 && call_own(h, regs[t][k])) || call(h, regs[t][k])) failed = 1;
                    // TB_VERIFY=1: what the first pass gave is what every later call must give, bit for bit, whichever way the
                    // library takes it under load (all-pairs call or chain, own queue or not, alone or in a combined flush)
                    std::vector<Region> first;
                    if (verify) first = regs[t];
                    // (likelihoods within 1e-9: a combined flush may sweep a region with another lane geometry than a lone call,
                    // which changes the order of the f64 sums by ~1e-13; everything discrete must be equal)
                    auto same = [&](const Region &a, const Region &b) {
                        auto close = [](const std::vector<double> &x, const std::vector<double> &y) {
                            for (size_t i = 0; i < x.size(); ++i)
                                if (!(x[i] == y[i] || std::fabs(x[i] - y[i]) <= 1e-9 || (std::isnan(x[i]) && std::isnan(y[i])))) return false;
                            return true;
                        };
                        const char *what = nullptr;
                        if (!close(a.out, b.out)) {
                            what = "likelihoods";
                            size_t n_bad = 0, first_bad = 0, last_bad = 0;
                            for (size_t i = 0; i < a.out.size(); ++i)
                                if (!(a.out[i] == b.out[i] || std::fabs(a.out[i] - b.out[i]) <= 1e-9)) {
                                    if (!n_bad++) first_bad = i;
                                    last_bad = i;
                                }
                            fprintf(stderr, "TB_VERIFY: %zu of %zu likelihoods differ, [%zu, %zu]: out[%zu] = %.17g, first pass %.17g (regions %zu, reads %u, haps %u); keep %s best %s pos %s status %s\n",
                                    n_bad, a.out.size(), first_bad, last_bad, first_bad, a.out[first_bad], b.out[first_bad], a.rro.size() - 1, a.rro.back(), a.rho.back(),
                                    a.keep == b.keep ? "=" : "DIFF", a.best == b.best ? "=" : "DIFF", a.pos == b.pos ? "=" : "DIFF", a.status == b.status ? "=" : "DIFF");
                            for (size_t i = first_bad; i < std::min(first_bad + 8, a.out.size()); ++i) fprintf(stderr, "   out[%zu] %.6f / %.6f\n", i, a.out[i], b.out[i]);
                        }
                        else if (mode < 4) return true;
                        else if (a.keep != b.keep) what = "keep";
                        else if (a.best != b.best) {
                            what = "best allele";
                            for (size_t r = 0; r < a.best.size(); ++r)
                                if (a.best[r] != b.best[r]) {
                                    const size_t nh_r = a.out.size() / a.best.size();  // (uniform shapes)
                                    fprintf(stderr, "TB_VERIFY: read %zu: best %d (first pass %d), lk %.17g / %.17g, conf %.17g / %.17g; row:", r, a.best[r], b.best[r], a.lk[r], b.lk[r], a.conf[r], b.conf[r]);
                                    for (size_t k = 0; k < nh_r; ++k) fprintf(stderr, " %.17g", a.out[r * nh_r + k]);
                                    fprintf(stderr, " | first pass:");
                                    for (size_t k = 0; k < nh_r; ++k) fprintf(stderr, " %.17g", b.out[r * nh_r + k]);
                                    fprintf(stderr, " | priorities:");
                                    for (size_t k = 0; k < nh_r; ++k) fprintf(stderr, " %d", a.pri[k]);
                                    fprintf(stderr, "\n");
                                    break;
                                }
                        }
                        else if (a.n_cig != b.n_cig) what = "cigar lengths";
                        else if (a.pos != b.pos) what = "positions";
                        else if (a.status != b.status) what = "status";
                        else if (!close(a.lk, b.lk)) what = "best likelihood";
                        else if (!close(a.conf, b.conf)) what = "confidence";
                        else
                            for (size_t r = 0; r < a.n_cig.size() && !what; ++r)
                                if (memcmp(a.cig.data() + a.out_cig_off[r], b.cig.data() + b.out_cig_off[r], 4 * (size_t)a.n_cig[r])) what = "cigars";
                        if (what) fprintf(stderr, "TB_VERIFY: %s differ\n", what);
                        return what == nullptr;
                    };
                    while (!go.load()) std::this_thread::yield();
                    uint64_t n = 0, cells = 0;
                    // TB_DEPTH=d (shared handles): a worker keeps d tickets in flight -- region k + d - 1 is submitted before region k
                    // is waited for (what a caller with more work than cores does: twice the regions outstanding per thread)
                    if (one_handle && depth > 1) {
                        std::vector<uint64_t> tk((size_t)depth, 0);
                        auto submit = [&](size_t k) {
                            Region &g = regs[t][k % (size_t)cycle];
                            return mode == 1 ? submit_shared(h, g, &tk[k % (size_t)depth]) : submit_fused_shared(h, g, &tk[k % (size_t)depth]);
                        };
                        size_t head = 0;  // next region to submit
                        for (; head + 1 < (size_t)depth; ++head)
                            if (submit(head)) failed = 1;
                        for (size_t k = 0; !failed; ++k) {
                            const bool more = !stop.load(std::memory_order_relaxed);
                            if (more && submit(head++)) {
                                failed = 1;
                                break;
                            }
                            if (k >= head) break;  // (stopped and drained)
                            Region &g = regs[t][k % (size_t)cycle];
                            if (phmm_wait(h, tk[k % (size_t)depth])) {
                                failed = 1;
                                break;
                            }
                            if (verify && !same(g, first[k % (size_t)cycle])) {
                                fprintf(stderr, "TB_VERIFY: thread %d, call %zu: results differ from the first pass\n", t, k);
                                failed = 1;
                                break;
                            }
                            ++n;
                            cells += g.cells;
                        }
                        n_calls += n;
                        n_cells += cells;
                        return;
                    }
                    for (size_t k = 0; !stop.load(std::memory_order_relaxed); ++k) {
                        Region &g = regs[t][k % (size_t)cycle];
                        if (call(h, g)) {
                            failed = 1;
                            break;
                        }
                        if (verify && !same(g, first[k % (size_t)cycle])) {
                            fprintf(stderr, "TB_VERIFY: thread %d, call %zu: results differ from the first pass\n", t, k);
                            failed = 1;
                            break;
                        }
                        ++n;
                        cells += g.cells;
                    }
                    n_calls += n;
                    n_cells += cells;
                });
            std::this_thread::sleep_for(std::chrono::milliseconds(200));
            uint64_t f0 = 0, s0 = 0, f1 = 0, s1 = 0;
            auto submit_stats = [&](uint64_t *f, uint64_t *n) {
                for (auto *h : hs) {
                    uint64_t a = 0, b = 0;
                    phmm_submit_stats(h, &a, &b);
                    *f += a;
                    *n += b;
                }
            };
            if (one_handle) submit_stats(&f0, &s0);
            const auto t0 = std::chrono::steady_clock::now();
            go = true;
            std::this_thread::sleep_for(std::chrono::duration<double>(dur));
            stop = true;
            for (auto &x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (one_handle) submit_stats(&f1, &s1);
            if (failed) {
                fprintf(stderr, "a call failed: %s\n", phmm_last_error(hs[0]));
                return 1;
            }
            const double rate = n_calls * per_call / dt;
            printf("%-8s %2d threads: %8.0f regions/s  %7.1f GCUPS  %6.1f us per call per thread", mode == 0 ? "own" : mode == 1 ? "shared" : mode == 2 ? "pipeline" : mode == 3 ? "realign" : mode == 4 ? "fused" : "gshared", T,
                   rate, (double)n_cells / dt / 1e9, dt * T / (double)n_calls * 1e6);
            if (one_handle) printf("   %.2f regions per flush", f1 > f0 ? (double)(s1 - s0) / (double)(f1 - f0) : 0.0);
            if (one_handle && depth > 1) printf("   %d tickets in flight per thread", depth);
            if (n_dev > 1) printf("   %d devices", n_dev);
            printf("\n");
            fflush(stdout);
            for (auto *h : hs) phmm_destroy(h);
        }
    }
    return 0;
}
