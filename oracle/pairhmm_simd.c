/*
 * pairhmm_simd.c -- CPU stand-in for the reference's VECTOR arm (AVXMode::AVX, the default).
 *
 * THIS FILE IS TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE (same rules as pairhmm_oracle.c).
 *
 * The reference's production PairHMM is not its scalar loop but `gkl::pairhmm::forward()` (called at
 * src/pair_hmm/pair_hmm.rs:348-366, detected at pair_hmm_likelihood_calculation_engine.rs:654-672); the crate `gkl ^0.1.1`
 * (Cargo.toml:42) is not under /root/reference and cannot be built here.  SURVEY.md 8(d) therefore asks for a second,
 * clearly labelled CPU line: "an AVX2/AVX-512 inter-pair SIMD variant as a stand-in for the gkl default mode (our
 * restatement, not gkl)".  This is that line: OUR restatement of the published Intel-GKL scheme --
 *   * the same M/I/D recurrence in f32 under a 2^120 initial scale,
 *   * a pair whose f32 result is too small to trust (scaled row sum < 1e-28, GKL's MIN_ACCEPTED) is recomputed in f64
 *     under the 2^1020 scale by the scalar oracle,
 * vectorised ACROSS pairs (one SIMD lane per (read, haplotype) pair of the region, 16 lanes = one 512-bit register;
 * gcc vector extensions, so -march=native picks AVX-512 / AVX2 as the host offers) instead of GKL's intra-pair
 * anti-diagonal scheme: inter-pair lanes need no shuffles at all and are always full, so on the region shapes of
 * BASELINE.json this is at least as fast as the intra-pair scheme, i.e. the stronger baseline.
 * Two bundles of 16 pairs are walked side by side to give the serial D chain (one FMA per column) some
 * instruction-level parallelism.
 * It is pinned by the reference's 104 known-answer vectors at the reference's own tolerance for this arm, 1e-5
 * (tests/test_oracle_simd.py), and against the scalar oracle.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))
#define V 16
typedef float vf __attribute__((vector_size(V * sizeof(float))));
typedef int32_t vi __attribute__((vector_size(V * sizeof(int32_t))));

/* scalar oracle (pairhmm_oracle.c): tables and the f64 pair */
extern double oracle_qual_to_error_prob(uint8_t qual);
extern double oracle_match_to_match_prob(unsigned ins_qual, unsigned del_qual);
typedef struct oracle_pairhmm oracle_pairhmm;
extern oracle_pairhmm *oracle_pairhmm_new(size_t max_read_length, size_t haplotype_max_length);
extern void oracle_pairhmm_free(oracle_pairhmm *h);
extern void oracle_pairhmm_do_not_use_tristate_correction(oracle_pairhmm *h);
extern double oracle_compute_read_likelihood_given_haplotype_log10(oracle_pairhmm *h, const uint8_t *hap, size_t H,
                                                                   const uint8_t *read, size_t R, const uint8_t *quals,
                                                                   const uint8_t *ins, const uint8_t *del,
                                                                   const uint8_t *gcp, int recache_read_values,
                                                                   const uint8_t *next_hap, size_t next_H, int *status);

static float eps_f[256], eps3_f[256], pm_f[256];
static pthread_once_t simd_once = PTHREAD_ONCE_INIT;
static void simd_tables(void) {
    for (int q = 0; q < 256; ++q) {
        double e = oracle_qual_to_error_prob((uint8_t)q);
        eps_f[q] = (float)e;
        eps3_f[q] = (float)(e / 3.0);
        pm_f[q] = (float)(1.0 - e);
    }
}

/* Per read row, per-lane constants (lane l of a bundle works on its own (read, haplotype) pair). */
typedef struct {
    vf mm, mi, md, im, ii, pm, px;
    vi x;
} rowv;

typedef struct {
    uint32_t n_regions;
    const uint32_t *region_read_off, *region_hap_off, *read_off, *hap_off;
    const uint8_t *read_bases, *base_q, *ins_q, *del_q, *gcp, *hap_bases;
    const uint64_t *out_off;
    double *out;
    int disable_tristate;
    volatile uint32_t next;
    volatile uint64_t redone;
} simd_ctx;

/* A bundle = V consecutive (read, haplotype) pairs of a region in read-major order, one per lane: 2 reads x 8
 * haplotypes for the 8-haplotype regions of BASELINE.json, 16 haplotypes of one read for the 64-haplotype ones. */
typedef struct {
    rowv *rows;   /* [Rmax]   */
    vi *hapT;     /* [Hmax]   column j of every lane's haplotype (256 past its end: matches nothing) */
    vi *hapN;     /* [Hmax]   -1 where that column is 'N' (wildcard, pair_hmm.rs:643) */
    vf *M, *I, *D; /* [Hmax+1] */
    vi Rl, Hl;
    vf init, sum;
    size_t Rmax, Hmax;
    uint32_t lanes;
    uint32_t hap_of_lane[V];
} bundle;

static void bundle_setup(const simd_ctx *c, uint32_t r0, uint32_t h0, uint32_t nh, uint64_t p0, uint32_t lanes, bundle *b,
                         int reuse_haps) {
    const float INIT = 0x1p120f; /* GKL: ldexpf(1.f, 120) */
    b->lanes = lanes;
    b->Rmax = b->Hmax = 0;
    uint32_t read_of_lane[V];
    int same_haps = reuse_haps;
    for (uint32_t l = 0; l < V; ++l) {
        const uint64_t p = p0 + (l < lanes ? l : 0);  /* idle lanes repeat lane 0 (results ignored) */
        read_of_lane[l] = r0 + (uint32_t)(p / nh);
        const uint32_t a = h0 + (uint32_t)(p % nh);
        if (b->hap_of_lane[l] != a) same_haps = 0;
        b->hap_of_lane[l] = a;
        const size_t R = c->read_off[read_of_lane[l] + 1] - c->read_off[read_of_lane[l]];
        const size_t H = c->hap_off[a + 1] - c->hap_off[a];
        b->Rl[l] = (int32_t)R;
        b->Hl[l] = (int32_t)H;
        b->init[l] = INIT / (float)H;
        if (R > b->Rmax) b->Rmax = R;
        if (H > b->Hmax) b->Hmax = H;
    }
    if (!same_haps)
        for (size_t j = 0; j < b->Hmax; ++j) {
            vi y, yn;
            for (uint32_t l = 0; l < V; ++l) {
                const int32_t v = (int32_t)j < b->Hl[l] ? c->hap_bases[c->hap_off[b->hap_of_lane[l]] + j] : 256;
                y[l] = v;
                yn[l] = v == 'N' ? -1 : 0;
            }
            b->hapT[j] = y;
            b->hapN[j] = yn;
        }
    for (size_t i = 0; i < b->Rmax; ++i) {
        rowv r;
        for (uint32_t l = 0; l < V; ++l) {
            if ((int32_t)i >= b->Rl[l]) {  /* past this lane's read: its sum has been taken, anything goes */
                r.mm[l] = r.mi[l] = r.md[l] = r.im[l] = r.ii[l] = r.pm[l] = r.px[l] = 0.f;
                r.x[l] = 257;
                continue;
            }
            const size_t o = c->read_off[read_of_lane[l]] + i;
            const uint8_t q = c->base_q[o], iq = c->ins_q[o], dq = c->del_q[o], g = c->gcp[o], x = c->read_bases[o];
            r.mm[l] = (float)oracle_match_to_match_prob(iq, dq);
            r.mi[l] = eps_f[iq];
            r.md[l] = eps_f[dq];
            r.ii[l] = eps_f[g];
            r.im[l] = pm_f[g];
            r.pm[l] = pm_f[q];
            r.px[l] = x == 'N' ? pm_f[q] : (c->disable_tristate ? eps_f[q] : eps3_f[q]);
            r.x[l] = x;
        }
        b->rows[i] = r;
    }
}

#define CELL(B, j)                                                                        \
    {                                                                                     \
        const vf uM = B->M[j], uI = B->I[j], uD = B->D[j];                                \
        const vi eq = (B->hapT[j - 1] == r##B.x) | B->hapN[j - 1];                        \
        const vf prior = (vf)((eq & (vi)r##B.pm) | (~eq & (vi)r##B.px));                  \
        const vf m = prior * (dM##B * r##B.mm + (dI##B + dD##B) * r##B.im);               \
        const vf in = uM * r##B.mi + uI * r##B.ii;                                        \
        const vf d = lM##B * r##B.md + lD##B * r##B.ii;                                   \
        B->M[j] = m;                                                                      \
        B->I[j] = in;                                                                     \
        B->D[j] = d;                                                                      \
        dM##B = uM; dI##B = uI; dD##B = uD; lM##B = m; lD##B = d;                         \
    }

static void take_sums(bundle *b, size_t i) { /* lanes whose read ends with row i */
    const vi ends = b->Rl == ((vi){0} + (int32_t)(i + 1));
    int any = 0;
    for (int l = 0; l < V; ++l) any |= ends[l];
    if (!any) return;
    vf s = {0};
    for (size_t j = 1; j <= b->Hmax; ++j) {
        const vi in = (((vi){0} + (int32_t)j) <= b->Hl) & ends;
        s += (vf)(in & (vi)(b->M[j] + b->I[j]));
    }
    b->sum += s;
}

/* Two bundles walked side by side: the serial D chain (one FMA per column) gets some instruction-level parallelism. */
static void sweep2(bundle *a, bundle *b /* may be NULL */) {
    const vf zero = {0};
    bundle *both[2] = {a, b};
    for (int k = 0; k < 2; ++k)
        if (both[k]) {
            for (size_t j = 0; j <= both[k]->Hmax; ++j) {
                both[k]->M[j] = both[k]->I[j] = zero;
                both[k]->D[j] = both[k]->init;
            }
            both[k]->sum = zero;
        }
    const size_t Ra = a->Rmax, Rb = b ? b->Rmax : 0, Rmax = Ra > Rb ? Ra : Rb;
    const size_t Ha = a->Hmax, Hb = b ? b->Hmax : 0, Hmin = b ? (Ha < Hb ? Ha : Hb) : 0;
    for (size_t i = 0; i < Rmax; ++i) {
        const int la = i < Ra, lb = i < Rb;
        vf dMa = zero, dIa = zero, dDa = zero, lMa = zero, lDa = zero, dMb = zero, dIb = zero, dDb = zero, lMb = zero, lDb = zero;
        rowv ra = a->rows[la ? i : 0], rb = ra;
        if (la) {
            dMa = a->M[0]; dIa = a->I[0]; dDa = a->D[0];
            a->M[0] = a->I[0] = a->D[0] = zero;  /* column 0 of rows >= 1 is zero */
        }
        if (lb) {
            rb = b->rows[i];
            dMb = b->M[0]; dIb = b->I[0]; dDb = b->D[0];
            b->M[0] = b->I[0] = b->D[0] = zero;
        }
        size_t j = 1;
        if (la && lb)
            for (; j <= Hmin; ++j) {
                CELL(a, j)
                CELL(b, j)
            }
        if (la) {
            for (size_t k = j; k <= Ha; ++k) CELL(a, k)
            take_sums(a, i);
        }
        if (lb) {
            for (size_t k = j; k <= Hb; ++k) CELL(b, k)
            take_sums(b, i);
        }
    }
}

static void bundle_alloc(bundle *b, size_t max_r, size_t max_h) {
    b->rows = (rowv *)aligned_alloc(64, sizeof(rowv) * (max_r + 1));
    b->hapT = (vi *)aligned_alloc(64, sizeof(vi) * (max_h + 1) * 2);
    b->hapN = b->hapT + max_h + 1;
    b->M = (vf *)aligned_alloc(64, sizeof(vf) * (max_h + 1) * 3);
    b->I = b->M + max_h + 1;
    b->D = b->I + max_h + 1;
    for (int l = 0; l < V; ++l) b->hap_of_lane[l] = 0xffffffffu;
}
static void bundle_free(bundle *b) {
    free(b->rows);
    free(b->hapT);
    free(b->M);
}

static void region_simd(const simd_ctx *c, uint32_t g, uint64_t *redone) {
    const uint32_t r0 = c->region_read_off[g], r1 = c->region_read_off[g + 1];
    const uint32_t h0 = c->region_hap_off[g], h1 = c->region_hap_off[g + 1];
    const uint32_t nr = r1 - r0, nh = h1 - h0;
    if (!nr || !nh) return;
    double *out = c->out + c->out_off[g];
    size_t max_r = 0, max_h = 0;
    for (uint32_t r = r0; r < r1; ++r) {
        size_t len = c->read_off[r + 1] - c->read_off[r];
        if (len > max_r) max_r = len;
    }
    for (uint32_t a = h0; a < h1; ++a) {
        size_t len = c->hap_off[a + 1] - c->hap_off[a];
        if (len > max_h) max_h = len;
    }
    bundle B[2];
    bundle_alloc(&B[0], max_r, max_h);
    bundle_alloc(&B[1], max_r, max_h);
    oracle_pairhmm *scalar = NULL;
    const double INIT_LOG10 = 120.0 * 0.30102999566398119521;
    const float MIN_ACCEPTED = 1e-28f; /* GKL: below this the f32 result is redone in f64 */
    const uint64_t n_pairs = (uint64_t)nr * nh;
    for (uint64_t p0 = 0; p0 < n_pairs; p0 += 2 * V) {
        const uint64_t left = n_pairs - p0;
        const uint32_t la = left < V ? (uint32_t)left : V;
        const uint32_t lb = left > V ? (left - V < V ? (uint32_t)(left - V) : V) : 0;
        bundle_setup(c, r0, h0, nh, p0, la, &B[0], p0 > 0);
        if (lb) bundle_setup(c, r0, h0, nh, p0 + V, lb, &B[1], p0 > 0);
        sweep2(&B[0], lb ? &B[1] : NULL);
        for (int k = 0; k < 2; ++k) {
            const uint32_t lanes = k ? lb : la;
            for (uint32_t l = 0; l < lanes; ++l) {
                const uint64_t p = p0 + (uint64_t)k * V + l;
                const float s = B[k].sum[l];
                double v;
                if (!(s >= MIN_ACCEPTED) || isinf(s)) { /* not trusted in f32: the scalar f64 arm */
                    if (!scalar) {
                        scalar = oracle_pairhmm_new(max_r, max_h);
                        if (c->disable_tristate) oracle_pairhmm_do_not_use_tristate_correction(scalar);
                    }
                    const uint32_t rr = r0 + (uint32_t)(p / nh), a = h0 + (uint32_t)(p % nh);
                    const size_t ro = c->read_off[rr], R = c->read_off[rr + 1] - ro;
                    const size_t ho = c->hap_off[a], H = c->hap_off[a + 1] - ho;
                    v = oracle_compute_read_likelihood_given_haplotype_log10(
                        scalar, c->hap_bases + ho, H, c->read_bases + ro, R, c->base_q + ro, c->ins_q + ro, c->del_q + ro,
                        c->gcp + ro, 1, NULL, 0, NULL);
                    ++*redone;
                } else {
                    v = log10((double)s) - INIT_LOG10;
                }
                out[p] = v;
            }
        }
    }
    if (scalar) oracle_pairhmm_free(scalar);
    bundle_free(&B[0]);
    bundle_free(&B[1]);
}

static void *simd_worker(void *p) {
    simd_ctx *c = (simd_ctx *)p;
    uint64_t redone = 0;
    for (;;) {
        uint32_t g = __sync_fetch_and_add(&c->next, 1u);
        if (g >= c->n_regions) break;
        region_simd(c, g, &redone);
    }
    __sync_fetch_and_add(&c->redone, redone);
    return NULL;
}

/* Same SoA layout as oracle_compute / phmm_compute.  *n_redone (may be NULL) = pairs recomputed in f64. */
ORACLE_API int oracle_simd_compute(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                                   const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                                   const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp,
                                   const uint32_t *hap_off, const uint8_t *hap_bases, const uint64_t *out_off,
                                   double *out, int disable_tristate, int n_threads, uint64_t *n_redone) {
    pthread_once(&simd_once, simd_tables);
    (void)oracle_match_to_match_prob(40, 40); /* builds the scalar oracle's tables before the threads start */
    simd_ctx c = {n_regions, region_read_off, region_hap_off, read_off, hap_off, read_bases, base_q, ins_q,
                  del_q,     gcp,             hap_bases,      out_off,  out,     disable_tristate, 0, 0};
    if (n_threads <= 1) {
        simd_worker(&c);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, simd_worker, &c);
        for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    if (n_redone) *n_redone = c.redone;
    return 0;
}

ORACLE_API int oracle_simd_lanes(void) { return V; }
