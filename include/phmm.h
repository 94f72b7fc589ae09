/*
 * phmm.h -- C ABI of the MI355X-native PairHMM read x haplotype likelihood engine.
 *
 * This is the drop-in boundary for Lorikeet's PairHMM path.  The reference has no FFI of its
 * own on this path; the seam it does have is the function pointer returned by
 * `gkl::pairhmm::forward()` and the batch loop around it:
 *
 *   reference/src/pair_hmm/pair_hmm.rs:345-375   PairHMM::compute_likelihoods
 *       for read { for hap { forward(hap, read, read_quals, ins_gop, del_gop, gcp) -> f64 } }
 *       => m_log_likelihood_array, Nr*Nh f64, READ-MAJOR, haplotypes in initialize() order
 *   reference/src/pair_hmm/pair_hmm.rs:217-341   PairHMM::compute_log10_likelihoods (caller)
 *   reference/tests/vector_pair_hmm_unit_tests.rs:51-59   the same call shape in the tests
 *
 * Every entry point below takes plain pointers and sizes (no C++/torch types).  Inputs are the
 * arrays the Rust call site already holds (`ReadDataHolder`, pair_hmm.rs:720-745, and
 * `m_haplotype_data_array`, :71-79) flattened into struct-of-arrays with prefix-sum offsets, so
 * that any number of assembly regions travel in one call.  INTEGRATION.md shows the Rust
 * `extern "C"` block and the `AVXMode::Hip` arm that binds them.
 *
 * Semantics (identical to the reference):
 *   - log10 Pr(read | haplotype) of the M/I/D forward recurrence of pair_hmm.rs:503-615,
 *     tristate correction ON unless PHMM_FLAG_NO_TRISTATE (pair_hmm.rs:189-191, :643-651);
 *   - base comparison is raw byte equality, uppercase 'N' on either side is a wildcard (:643);
 *   - qualities are full u8 (0..=255); reads longer than the haplotype are legal; an empty read
 *     gives -inf; an empty read list is a no-op (:224); an EMPTY HAPLOTYPE is rejected with
 *     PHMM_ERR_INVALID_ARG by every entry point (the reference would divide 2^1020 by zero and return
 *     -inf for every read of the region, :515-517; Lorikeet's assembler never produces one);
 *   - every result satisfies <= 0.0; a violation (the reference asserts, :478-481) is reported
 *     as PHMM_ERR_POSITIVE_RESULT.
 * Results agree with the reference's scalar f64 path to ~1e-13 absolute in log10 (FMA contraction,
 * exact rescalings of the DP state and the order of the final row sum are the only differences,
 * NOTEBOOK.md §4); the reference's own gate is 1e-5 abs.  Pairs whose log10 likelihood is below -600 --
 * where the reference's 2^1020-scaled sums approach the denormal range and every rounding shows -- are recomputed
 * in the reference's own operation order, so the underflow band (down to and including the point where the result
 * turns to -inf, ~1e-628) agrees with the scalar arm as well.
 *
 * Every entry point returns with the calling thread's current HIP device restored, and no C++ exception crosses
 * the boundary (PHMM_ERR_NO_MEMORY / PHMM_ERR_INTERNAL).  Slots of `out` that out_off leaves between regions
 * (gaps) are never written.
 *
 * Threading: a handle may be used by one thread at a time; create one per host thread (the
 * reference clones its engine per rayon task, assembly_region_walker.rs:227) or serialise.  The exception is
 * phmm_submit / phmm_engine_submit / phmm_wait: any number of threads may call them on ONE shared handle, and the
 * library computes the regions of all waiting threads together.  phmm_compute_multi spreads one call over several
 * handles (one per device).
 * There is NO CPU fallback: without a HIP device phmm_create() fails.
 */
#ifndef PHMM_H
#define PHMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHMM_VERSION 1

/* flags for phmm_create */
#define PHMM_FLAG_NO_TRISTATE 1u /* PairHMM::do_not_use_tristate_correction (pair_hmm.rs:189) */
#define PHMM_FLAG_F32_FIRST 2u   /* opt-in: what the reference's vector arm does (gkl, called at pair_hmm.rs:345-375):
                                  * large batches are swept in f32 first and every read whose result is too small to
                                  * trust in f32 (< ~1e-59 / haplotype length), or that needs the general path, is
                                  * recomputed in f64.  Results of the f32 pairs differ from the f64 path by f32
                                  * rounding (<= 2e-6 in log10 measured, the reference's gate is 1e-5); default OFF:
                                  * everything in f64.  Small calls, and region calls that go through the resident region
                                  * server, are computed in f64 under this flag too (NOTEBOOK.md 20.7). */

/* status codes (0 == success) */
#define PHMM_OK 0
#define PHMM_ERR_INVALID_ARG 1      /* null pointer, non-monotonic offsets, size mismatch      */
#define PHMM_ERR_NO_DEVICE 2        /* no HIP device / bad device id                           */
#define PHMM_ERR_HIP 3              /* a HIP runtime call failed; see phmm_last_error          */
#define PHMM_ERR_POSITIVE_RESULT 4  /* some log10 likelihood > 0 (reference asserts, :478-481) */
#define PHMM_ERR_NOT_BOUND 5        /* phmm_batch_launch before device buffers were bound      */
#define PHMM_ERR_NO_MEMORY 6        /* a host allocation failed (std::bad_alloc never crosses the ABI)    */
#define PHMM_ERR_INTERNAL 7         /* any other C++ exception inside the library; see phmm_last_error    */
#define PHMM_ERR_CIGAR_CAPACITY 8   /* phmm_sw_align: a CIGAR did not fit its slot; n_cigar has the sizes */

typedef struct phmm_handle phmm_handle;
typedef struct phmm_batch phmm_batch;

/* Number of HIP devices visible to the process (0 if none / no driver). */
int phmm_device_count(void);

/* Create an engine on HIP device `device_id`.  Builds the quality->probability tables
 * (quality_utils.rs:82-104, pair_hmm_model.rs:47-78) once and keeps them resident in HBM;
 * owns a stream, pinned staging and device arenas that grow on demand.  While at most four engines are alive on a device,
 * each runs its one-enqueue calls on a hardware queue of its own (callers with an engine each then run side by side whatever
 * the runtime does with ordinary streams); env PHMM_REGION_OWN_QUEUE=0 at creation turns that off.
 * While MORE than five of the caller's engines are alive on a device (an engine per worker thread at Lorikeet's --threads 10),
 * the one-shot phmm_region_compute calls of such a private engine are served by the device's resident REGION SERVER (a kernel
 * that stays on the chip while calls keep coming: the region is staged into a slot of pinned memory, every read runs as one wave
 * from pre-step to projection, nothing is launched; phmm_server.cpp).  Results and error reporting are the call's own, and the
 * likelihoods are the region's own bits whatever else is in flight (16 lanes x ceil(H / 16) columns per pair: a function of the
 * region's longest haplotype).  Calls outside the server's limits (reads beyond 268 bases, haplotypes beyond 512, more than 1 MB of
 * inputs) take the engine's own streams.  env PHMM_REGION_SERVER=0 turns the server off, =1 sends every region call through it.
 * While the caller holds more engines on a device than the process has cores (its affinity mask, its container's CPU quota), a
 * one-shot call waits for its kernels in 20 us naps instead of spinning: spinning waiters beyond the cores are throttled together
 * with the callers that stage (32 engines on 16 cores ran at half the rate of 16; NOTEBOOK.md 20.5).
 * Opt-in, env PHMM_ROUTE_SHARED=n: while more than n engines are alive, the one-shot calls of private engines -- phmm_compute,
 * phmm_engine_compute, phmm_region_compute on up to eight regions or 512 KB per array -- go through ONE shared engine of the same
 * flags inside the library instead (the queue of phmm_submit: callers that are waiting anyway share a flush); which regions share a
 * launch depends on timing, so likelihoods are then reproducible to ~1e-13 rather than bit for bit.  An engine whose developer
 * switches were set (phmm_set_switch) keeps to its own streams.
 * Debug environment: PHMM_MIRROR_CANARY=1 makes a device store that lands in the pinned hand-over buffer outside its call fail
 * that call (PHMM_ERR_INTERNAL; =2: abort); PHMM_REGION_PICK_TIMEOUT_US (default 5 000) bounds the small region call's wait for
 * its second hardware queue -- out of time, the call is redone the chained way (phmm_get_stat "region_pick_timeouts").
 * Replaces PairHMM::initialize's per-region table/matrix construction (pair_hmm.rs:63-165).
 * Returns NULL on failure (phmm_last_error(NULL) has the message). */
phmm_handle *phmm_create(int device_id, unsigned flags);
void phmm_destroy(phmm_handle *h);

/* Last error message of this handle (or of the failed phmm_create when h == NULL).
 * Never NULL; valid until the next call on the same handle. */
const char *phmm_last_error(phmm_handle *h);

/*
 * Synchronous whole-batch call on HOST buffers (pageable is fine): plan, H2D, kernels, D2H.
 * Replaces PairHMM::compute_likelihoods (pair_hmm.rs:345-375) for `n_regions` regions at once.
 *
 *   region_read_off[n_regions+1]  prefix sums: region g owns reads  [region_read_off[g], region_read_off[g+1])
 *   region_hap_off [n_regions+1]  prefix sums: region g owns haps   [region_hap_off[g],  region_hap_off[g+1])
 *   read_off[n_reads+1]           byte offsets of each read into the five per-base read arrays
 *   read_bases, base_q, ins_q, del_q, gcp   read_off[n_reads] bytes each (ReadDataHolder fields)
 *   hap_off[n_haps+1], hap_bases  byte offsets / bases of each haplotype
 *   out_off[n_regions+1]          element offsets into out; region g needs Nr_g*Nh_g doubles
 *   out                           per region row-major [read][hap] == m_log_likelihood_array order
 *
 * All pointers are caller-owned and not retained.  Returns a PHMM_* status.
 */
int phmm_compute(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off,
                 const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases,
                 const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp,
                 const uint32_t *hap_off, const uint8_t *hap_bases, const uint64_t *out_off, double *out);

/*
 * Cross-thread batching: the same call as phmm_compute, split into submit + wait, and -- unlike every other entry
 * point -- safe to call from many threads on ONE shared handle.  This is the call pattern of the reference as it
 * stands: every rayon worker calls PairHMM::compute_likelihoods (pair_hmm.rs:345-375) with one region at a time
 * (assembly_region_walker.rs:210-273).  A submission is only queued.  The first thread to wait while an engine lane is
 * free leads one flush: it takes everything queued so far (its own region plus those of the workers that arrived in
 * the meantime), computes it as one batch and hands every region's results to its owner; the other threads sleep until
 * theirs are in.  No timer, no background thread: batches become as large as the number of workers that were waiting
 * anyway, a lone caller pays nothing.  Results are those of phmm_compute on the same regions (every region is
 * independent; which regions share a launch only selects the kernel shape, like batch size does in phmm_compute).
 *
 *   phmm_submit   same arrays as phmm_compute; ALL of them, inputs and `out`, stay caller-owned and must remain valid
 *                 until phmm_wait(ticket) has returned.  Argument errors are reported here (nothing is queued).
 *   phmm_wait     blocks until the ticket's results are in `out`; returns that submission's own status (a failure in
 *                 another thread's region does not leak into it).  Each ticket must be waited for exactly once.
 * phmm_last_error() after a failed phmm_submit / phmm_wait returns the calling thread's message.
 */
int phmm_submit(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off,
                const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases,
                const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp,
                const uint32_t *hap_off, const uint8_t *hap_bases, const uint64_t *out_off, double *out,
                uint64_t *ticket);
int phmm_wait(phmm_handle *h, uint64_t ticket);
/* How many flushes have run on this handle and how many submissions they carried (batching achieved). */
void phmm_submit_stats(phmm_handle *h, uint64_t *n_flushes, uint64_t *n_submissions);

/*
 * Several devices, one process (regions shard, nothing is exchanged).  phmm_compute_multi is phmm_compute over
 * `n_handles` engines, normally one per device.  Whole regions go to engines in CONTIGUOUS ranges balanced by
 * cells(region) = sum of read lengths x sum of haplotype lengths (phmm_split_regions); only when such ranges come out
 * more than 5 % uneven -- a heavy-tailed set -- are regions dealt out one by one by greedy longest-processing-time
 * (phmm_assign_regions: heaviest region first onto the least loaded engine).  Either way every engine stages its share
 * straight from the caller's arrays (no gather: each payload byte is copied once, into that engine's pinned staging),
 * computes it concurrently on a host thread of its own pinned to the CPUs local to its GPU, and the results land in the
 * caller's `out` exactly where phmm_compute would put them.  The handles must not be in use by other threads during the
 * call; on failure the message is phmm_last_error(handles[0]).  phmm_assign_regions / phmm_split_regions expose the two
 * assignments alone (host only, no device needed): part_of_region[g] in [0, n_parts); first_region[0..n_parts], part k
 * owning the regions [first_region[k], first_region[k+1]).
 */
int phmm_assign_regions(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                        const uint32_t *read_off, const uint32_t *hap_off, uint32_t n_parts,
                        uint32_t *part_of_region);
int phmm_split_regions(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                       const uint32_t *read_off, const uint32_t *hap_off, uint32_t n_parts,
                       uint32_t *first_region);
int phmm_compute_multi(phmm_handle *const *handles, uint32_t n_handles, uint32_t n_regions,
                       const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                       const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q,
                       const uint8_t *gcp, const uint32_t *hap_off, const uint8_t *hap_bases,
                       const uint64_t *out_off, double *out);

/*
 * Split-phase interface for device-resident data and for overlapping transfers with compute.
 * A batch owns the launch plan (regions binned into kernel shape classes) and the device copy
 * of the offset arrays; the byte payload and the output live in device memory that is either
 * caller-owned (phmm_batch_bind_device) or uploaded from host buffers (phmm_batch_upload).
 *
 *   b = phmm_batch_create(h, <offset arrays on the host>);
 *   phmm_batch_bind_device(b, d_read_bases, ..., d_out);   // or phmm_batch_upload(b, host ptrs)
 *   phmm_batch_launch(b, stream);                           // async: kernels only
 *   ... hipStreamSynchronize / phmm_batch_download(b, out) ...
 *   phmm_batch_destroy(b);
 */
phmm_batch *phmm_batch_create(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off,
                              const uint32_t *region_hap_off, const uint32_t *read_off,
                              const uint32_t *hap_off, const uint64_t *out_off);
void phmm_batch_destroy(phmm_batch *b);

/* Device pointers (on the handle's device), caller-owned, must stay valid until the launch completes. */
int phmm_batch_bind_device(phmm_batch *b, const uint8_t *d_read_bases, const uint8_t *d_base_q,
                           const uint8_t *d_ins_q, const uint8_t *d_del_q, const uint8_t *d_gcp,
                           const uint8_t *d_hap_bases, double *d_out);

/* Copy host payload into batch-owned device buffers (async on the handle's stream) and bind them. */
int phmm_batch_upload(phmm_batch *b, const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q,
                      const uint8_t *del_q, const uint8_t *gcp, const uint8_t *hap_bases);

/* Enqueue the forward kernels on `stream` (a hipStream_t; NULL = the handle's own stream).
 * Asynchronous; no host<->device copies, no allocation. */
int phmm_batch_launch(phmm_batch *b, void *stream);

/* Wait for the handle's stream, copy batch-owned output to `out` (host) and check the device
 * status word.  Only valid after phmm_batch_upload + phmm_batch_launch(b, NULL). */
int phmm_batch_download(phmm_batch *b, double *out);

/* Read and clear the device status word after the caller synchronised its own stream
 * (for the bind_device flow).  Returns PHMM_OK or PHMM_ERR_POSITIVE_RESULT. */
int phmm_batch_status(phmm_batch *b);

/* Introspection for tests / bench: totals of the plan. */
uint64_t phmm_batch_cells(const phmm_batch *b);           /* sum over regions of (sum R)*(sum H)      */
uint64_t phmm_batch_algorithmic_bytes(const phmm_batch *b); /* sum 5R + sum H + 8*Nr*Nh    */
uint32_t phmm_batch_num_launches(const phmm_batch *b);    /* kernel launches one phmm_batch_launch does */
/* What the planned launches sweep, in lane-cells: every row of every wave x 64 lanes x its columns per lane -- the columns beyond a
 * haplotype's end, the haplotype slots a wave leaves empty.  executed / phmm_batch_cells is what a batch's shapes cost in padding
 * (1.01 for the uniform 150 / 300 batch).  (The reference's scalar arm skips the columns a haplotype shares with the one before it,
 * find_first_position_where_haplotypes_differ, pair_hmm.rs:452-464, 706-717; here every pair is swept in full: a device form of the
 * sharing was built in round 4, measured at x 0.96-1.10 and removed in round 6, NOTEBOOK.md 18.4.) */
uint64_t phmm_batch_executed_cells(const phmm_batch *b);
/* Name of the kernel doing most cells of this batch as rocprofv3 reports it, e.g. "phmm_forward_chain_k<16,19>". */
const char *phmm_batch_dominant_kernel(const phmm_batch *b);

/*
 * The launch plan of a batch WITHOUT a device (host only, like phmm_split_regions): what phmm_batch_create would decide
 * for these offsets on an engine created with `flags`, `concurrent_callers` flows sharing the chip (1 = alone).  For sizing
 * shards before any GPU is touched: e.g. that each rank's share of a set still takes the chained kernel with enough work
 * items (one wave each) to put two waves on every one of the chip's 1 024 SIMDs.
 */
typedef struct phmm_plan_info {
    uint64_t cells;              /* sum over regions of (sum R) x (sum H)                                   */
    uint64_t chain_cells;        /* ... of which the chained kernels sweep                                  */
    uint64_t chain_items;        /* their work items: one wave each (a run of reads x one load of haplotypes) */
    uint32_t n_launches;         /* kernel launches of one phmm_batch_launch                                */
    uint32_t n_chain_launches;
    uint32_t min_reads_per_run;  /* shortest run of reads a chained work item holds (0: nothing chains)     */
    uint32_t reserved;
    char dominant_kernel[64];
    /* What the launches sweep (phmm_batch_executed_cells), in lane-cells = steps x 64 lanes x K columns of every wave ...          */
    uint64_t swept_cells;
    uint64_t pad_column_cells;   /* ... of which columns beyond a haplotype's end (lanes x K - H per pair)                      */
    uint64_t pad_slot_cells;     /* ... haplotype slots a wave leaves empty; the rest above `cells` is steps without a read row */
                                 /*     (SUM / RESET rows between the reads of a run, the fill of the lane pipeline)            */
} phmm_plan_info;
int phmm_plan_describe(unsigned flags, uint32_t concurrent_callers, uint32_t n_regions, const uint32_t *region_read_off,
                       const uint32_t *region_hap_off, const uint32_t *read_off, const uint32_t *hap_off, phmm_plan_info *info);

/*
 * Engine-level call: everything PairHMMLikelihoodCalculationEngine::compute_read_likelihoods does with
 * the numbers (reference src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs:195-242), for
 * n_regions regions at once, on the device:
 *   1. modify_read_qualities (:352-388, default branch): PCR indel error model (:502-611) and the
 *      quality caps (:428-466) on copies of the read qualities; gcp = constant (:649-651);
 *   2. the PairHMM forward kernels on the modified qualities (pair_hmm.rs:345-375);
 *   3. normalize_likelihoods (src/model/allele_likelihoods.rs:378-508) with
 *      log10_global_read_mismapping_rate as the cap;
 *   4. the keep / remove decision of filter_poorly_modeled_evidence (:925-1041) with the static or
 *      dynamic threshold of :229-239 (computed from the ORIGINAL base qualities, as the reference does).
 * The caller keeps ownership of the evidence lists: `keep[r] == 0` marks reads the reference would move
 * to filtered_evidence_by_sample_index; compaction of the [allele, read] matrix happens when the caller
 * scatters `out` (read-major) into its AlleleLikelihoods.  Per-sample structure does not matter here:
 * every step is per read.
 */
typedef struct phmm_engine_config {
    uint8_t constant_gcp;                                   /* engine.rs:130  (CLI default 10)            */
    uint8_t pcr_error_model;                                /* :61-70  0 None 1 Hostile 2 Aggressive 3 Conservative */
    uint8_t base_quality_score_threshold;                   /* :133    (CLI default 18)                   */
    uint8_t dynamic_read_disqualification;                  /* :134                                       */
    uint8_t symmetrically_normalize_alleles_to_reference;   /* :137                                       */
    uint8_t disable_cap_read_qualities_to_mapq;             /* :138                                       */
    uint8_t reserved[2];
    double log10_global_read_mismapping_rate;               /* :131    cap of normalize_likelihoods       */
    double read_disqualification_scale;                     /* :135                                       */
    double expected_error_rate_per_base;                    /* :136                                       */
} phmm_engine_config;

/*
 *   base_q            ORIGINAL base qualities (read.qual())
 *   ins_q, del_q      BI / BD tags, or NULL for the reference's flat Q45 default (read_utils.rs:23,372-416)
 *   mapq[n_reads]     mapping quality per read (cap_minimum_read_qualities, :438)
 *   region_ref_hap    [n_regions] index INSIDE the region of the reference haplotype, -1 if none; may be
 *                     NULL (only consulted when symmetric normalisation is off)
 *   out               per region row-major [read][hap], NORMALISED log10 likelihoods
 *   keep[n_reads]     1 = evidence kept, 0 = removed as poorly modelled
 * Haplotypes of a region must be distinct (the reference de-duplicates them, haplotype.rs:263-275).
 */
int phmm_engine_compute(phmm_handle *h, const phmm_engine_config *cfg, uint32_t n_regions,
                        const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                        const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q,
                        const uint8_t *del_q, const uint8_t *mapq, const uint32_t *hap_off,
                        const uint8_t *hap_bases, const int32_t *region_ref_hap, const uint64_t *out_off,
                        double *out, uint8_t *keep);

/* phmm_engine_compute through the shared, thread-safe queue of phmm_submit (same rules: every array stays caller-owned
 * until phmm_wait(ticket) has returned; phmm_wait is the one above).  Engine-level submissions share a flush when their
 * configurations are equal and the same optional arrays (ins_q / del_q, region_ref_hap) are present; plain and
 * engine-level submissions on one handle are flushed separately, in submission order. */
int phmm_engine_submit(phmm_handle *h, const phmm_engine_config *cfg, uint32_t n_regions,
                       const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                       const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q,
                       const uint8_t *del_q, const uint8_t *mapq, const uint32_t *hap_off,
                       const uint8_t *hap_bases, const int32_t *region_ref_hap, const uint64_t *out_off,
                       double *out, uint8_t *keep, uint64_t *ticket);
/* ... and over several engines, one per device, like phmm_compute_multi (contiguous cell-balanced ranges of regions, one
 * pinned host thread per engine, nothing gathered); on failure the message is phmm_last_error(handles[0]). */
int phmm_engine_compute_multi(phmm_handle *const *handles, uint32_t n_handles, const phmm_engine_config *cfg, uint32_t n_regions,
                              const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                              const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q,
                              const uint8_t *mapq, const uint32_t *hap_off, const uint8_t *hap_bases,
                              const int32_t *region_ref_hap, const uint64_t *out_off, double *out, uint8_t *keep);

/*
 * Smith-Waterman alignment: SmithWatermanAligner::align of the reference
 * (src/smith_waterman/smith_waterman_aligner.rs:47-107 dispatch and exact-substring shortcut, :124-271 calculate_matrix,
 * :273-443 calculate_cigar) for a batch of (reference, alternate) pairs under one parameter set and one overhang
 * strategy -- what each of its call sites has in hand (read -> best haplotype: src/reads/alignment_utils.rs:40-70 via
 * src/assembly/assembly_based_caller_utils.rs:208-246; haplotype -> reference: src/reads/cigar_utils.rs:358-405).
 * Matrix, backtrack and CIGAR assembly all run on the device; the arithmetic is i32 and every tie rule is the
 * reference's, so CIGAR and offset EQUAL the reference's scalar arm (which its own tests assert equal to the vector
 * arm, tests/smith_waterman_aligner_unit_tests.rs:999-1103).
 *
 *   ref_off / alt_off [n+1]   byte offsets of each pair's reference / alternate sequence; what is aligned must be non-empty
 *                             (the reference asserts, :65-68).  Any lengths: up to ~8 000 bases everything of an alignment
 *                             lives in LDS, beyond that its bottom row and strip edges move to device memory (the two
 *                             sequences themselves must fit LDS together: ~80 000 bases each)
 *   params                    gkl::smithwaterman::Parameters::new(match, mismatch, gap open, gap extend).  While
 *                             max |weight| x (longest ref + longest alt + 2) stays below 1e8 -- the range in which the
 *                             reference's clamp at -1e8 (:31) cannot act -- scores travel times four with the winning
 *                             candidate in their low bits; beyond that a wide instance carries them as they are and applies
 *                             the clamp; from 1e9 on, where the reference's own 32-bit sums overflow, PHMM_ERR_INVALID_ARG
 *   overhang_strategy         PHMM_SW_* below == gkl::smithwaterman::OverhangStrategy
 *   cigar_off [n+1]           element offsets into `cigar`: alignment a may use cigar_off[a+1] - cigar_off[a] elements
 *                             (ref_len + alt_len + 3 always suffices; real CIGARs have a handful)
 *   cigar                     elements in BAM encoding, (length << 4) | op with M = 0, I = 1, D = 2, S = 4
 *   n_cigar [n]               elements of each CIGAR; if one exceeds its slot the call returns
 *                             PHMM_ERR_CIGAR_CAPACITY, every other alignment is valid and n_cigar tells the size to retry with
 *   alignment_offset [n]      SmithWatermanAlignmentResult::alignment_offset
 */
#define PHMM_SW_SOFTCLIP 0
#define PHMM_SW_INDEL 1
#define PHMM_SW_LEADING_INDEL 2
#define PHMM_SW_IGNORE 3
typedef struct phmm_sw_parameters {
    int32_t match_value, mismatch_penalty, gap_open_penalty, gap_extend_penalty;
} phmm_sw_parameters;
int phmm_sw_align(phmm_handle *h, uint32_t n_alignments, const uint32_t *ref_off, const uint8_t *ref_bases,
                  const uint32_t *alt_off, const uint8_t *alt_bases, const phmm_sw_parameters *params,
                  int overhang_strategy, const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar,
                  int32_t *alignment_offset);

/*
 * The same with shared references: alignment a pairs alternate sequence a with reference ref_index[a] -- reads against
 * the few haplotypes of their region, which then cross the bus once instead of once per read.  ref_index[a] ==
 * PHMM_SW_NO_REFERENCE skips the alignment (n_cigar 0, offset 0).  ref_off has n_references + 1 entries.
 */
#define PHMM_SW_NO_REFERENCE 0xffffffffu
int phmm_sw_align_indexed(phmm_handle *h, uint32_t n_references, const uint32_t *ref_off, const uint8_t *ref_bases,
                          uint32_t n_alignments, const uint32_t *ref_index, const uint32_t *alt_off,
                          const uint8_t *alt_bases, const phmm_sw_parameters *params, int overhang_strategy,
                          const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar, int32_t *alignment_offset);

/*
 * Best allele per read, ties broken by priority: AlleleLikelihoods::best_alleles_breaking_ties_main
 * (src/model/allele_likelihoods.rs:1043-1095) = search_best_allele (:457-554, can_be_reference = true) + BestAllele::new
 * (:1142-1160) for every read of n_regions regions -- the first step of realign_reads_to_their_best_haplotype
 * (src/assembly/assembly_based_caller_utils.rs:208-246) and of the reference's read-allele maps.
 *   likelihoods        per region row-major [read][hap] at out_off[g]: what phmm_engine_compute / phmm_compute return
 *   keep [n_reads]     or NULL: 0 = evidence removed by filter_poorly_modeled_evidence, no best allele (-1)
 *   hap_priority       [n_haps] one i32 per haplotype (haplotype_alignment_tiebreaking_priority, :187-195: is_ref +
 *                      1 - cigar elements; reference_tiebreaking_priority, :197-199), or NULL: no tie breaking
 *   informative_threshold   LOG_10_INFORMATIVE_THRESHOLD = 0.2 (:17) for log10 likelihoods
 *   best_allele [n_reads]   index inside the region, -1 where there is none; likelihood / confidence as BestAllele
 *                      holds them (BestAllele::is_informative: confidence > threshold)
 */
int phmm_best_alleles(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                      const uint64_t *out_off, const double *likelihoods, const uint8_t *keep, const int32_t *hap_priority,
                      double informative_threshold, int32_t *best_allele, double *likelihood, double *confidence);

/*
 * Both steps of realign_reads_to_their_best_haplotype that are arithmetic, in one call: the best allele of every read
 * (as phmm_best_alleles) and the read's Smith-Waterman alignment to that haplotype (as phmm_sw_align_indexed; the
 * reference: SoftClip, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, src/reads/alignment_utils.rs:40-70).  The index never
 * leaves the device.  `read_bases` are the reads WITHOUT their soft clips (the caller hard-clips them, :47-50); reads
 * without a best allele get n_cigar 0.  Projecting the read -> haplotype CIGAR onto the reference (:83-140) stays with
 * the caller.
 */
int phmm_realign_to_best(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                         const uint32_t *read_off, const uint8_t *read_bases, const uint32_t *hap_off,
                         const uint8_t *hap_bases, const uint64_t *out_off, const double *likelihoods, const uint8_t *keep,
                         const int32_t *hap_priority, double informative_threshold, const phmm_sw_parameters *params,
                         int overhang_strategy, const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar,
                         int32_t *alignment_offset, int32_t *best_allele, double *likelihood, double *confidence);

/*
 * The rest of AlignmentUtils::create_read_aligned_to_ref (src/reads/alignment_utils.rs:40-165), for every read of n_regions
 * regions: the read -> haplotype alignment phmm_realign_to_best returned is projected onto the reference through the
 * haplotype's own CIGAR -- CigarBuilder clean-up (src/reads/cigar_builder.rs), get_consolidated_padded_cigar(1000)
 * (src/haplotype/haplotype.rs:248-256), read_start_on_reference_haplotype (:283-311), trim_cigar_by_bases (:321-386),
 * apply_cigar_to_cigar (:240-281), left_align_indels against the reference haplotype (:425-566), the clips of the read's
 * original CIGAR put back (:173-213) and the length check (:151-161) -- one lane per read on the device.
 *   read_off / read_bases     the reads minus their soft clips (what was aligned)
 *   region_ref_hap [n_regions]          index INSIDE the region of the reference haplotype (left-alignment reads its bases)
 *   region_reference_start [n_regions]  padded_reference_loc.get_start()
 *   hap_cigar_off [n_haps+1], hap_cigar   Haplotype::cigar of every haplotype, BAM-encoded elements
 *   hap_start_wrt_ref [n_haps]          Haplotype::alignment_start_hap_wrt_ref
 *   best_allele, sw_cigar_off / sw_cigar / n_sw_cigar / sw_offset   as phmm_realign_to_best filled them
 *   orig_cigar_off [n_reads+1], orig_cigar   the reads' CIGARs before realignment (only their clips are used)
 *   out_cigar_off [n_reads+1]           element offsets into out_cigar (sw elements + haplotype elements + clips + 4 suffices)
 *   status [n_reads]   PHMM_PROJECT_REALIGNED: new_pos / out_cigar / n_out_cigar are the read's new alignment;
 *                      PHMM_PROJECT_UNCHANGED: no best allele or alignment_offset == -1, the read stays as it is (:60-63);
 *                      negative: the reference panics or returns Err for this read (-1 ... -4 CigarBuilder errors in the
 *                      order of cigar_builder.rs, -5 an assert such as "Read goes past end of reference")
 * Returns PHMM_ERR_CIGAR_CAPACITY when an output slot is too small (n_out_cigar holds the sizes).
 */
#define PHMM_PROJECT_REALIGNED 0
#define PHMM_PROJECT_UNCHANGED 1
int phmm_project_to_reference(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                              const uint32_t *read_off, const uint8_t *read_bases, const uint32_t *hap_off,
                              const uint8_t *hap_bases, const int32_t *region_ref_hap, const uint64_t *region_reference_start,
                              const uint32_t *hap_cigar_off, const uint32_t *hap_cigar, const uint32_t *hap_start_wrt_ref,
                              const int32_t *best_allele, const uint64_t *sw_cigar_off, const uint32_t *sw_cigar,
                              const uint32_t *n_sw_cigar, const int32_t *sw_offset, const uint32_t *orig_cigar_off,
                              const uint32_t *orig_cigar, const uint64_t *out_cigar_off, uint32_t *out_cigar,
                              uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status);

/*
 * realign_reads_to_their_best_haplotype (src/assembly/assembly_based_caller_utils.rs:208-246) in one call: the best allele
 * of every read (phmm_best_alleles), the read's alignment to that haplotype (phmm_sw_align_indexed) and the alignment
 * projected onto the reference (phmm_project_to_reference) -- the reads and haplotypes cross the bus once, the best
 * alleles and the read -> haplotype alignments never leave the device.  Arguments as in those three calls; per read it
 * returns BestAllele (best_allele / likelihood / confidence) and status / new_pos / out_cigar.  An alignment whose CIGAR
 * outgrows the library's own slots (24 elements) is handled inside (the call runs once more with larger ones).
 */
int phmm_realign_reads(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                       const uint32_t *read_off, const uint8_t *read_bases, const uint32_t *hap_off, const uint8_t *hap_bases,
                       const uint64_t *out_off, const double *likelihoods, const uint8_t *keep, const int32_t *hap_priority,
                       double informative_threshold, const phmm_sw_parameters *params, int overhang_strategy,
                       const int32_t *region_ref_hap, const uint64_t *region_reference_start, const uint32_t *hap_cigar_off,
                       const uint32_t *hap_cigar, const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off,
                       const uint32_t *orig_cigar, const uint64_t *out_cigar_off, uint32_t *out_cigar, uint32_t *n_out_cigar,
                       int64_t *new_pos, int32_t *status, int32_t *best_allele, double *likelihood, double *confidence);

/*
 * One call per region for the whole arithmetic path: what the reference does between
 * PairHMMLikelihoodCalculationEngine::compute_read_likelihoods and AssemblyBasedCallerUtils::
 * realign_reads_to_their_best_haplotype (src/haplotype/haplotype_caller_engine.rs:1311-1357 -- nothing lies between the
 * two but an early return when only one allele is left) in ONE enqueue on one stream:
 *   pre-step (phmm_engine_compute 1.) -> PairHMM forward kernels -> the exact pass below -600 -> normalize_likelihoods +
 *   filter_poorly_modeled_evidence -> best allele per read (phmm_best_alleles) -> the read's Smith-Waterman alignment to
 *   that haplotype -> its projection onto the reference (phmm_project_to_reference).
 * The likelihood matrix, the keep flags, the best alleles and the read -> haplotype alignments never leave the device
 * between the steps; the reads and haplotypes cross the bus once.  Returns what phmm_engine_compute and phmm_realign_reads
 * return together, and equals them field by field (`keep` of the first feeding the second).
 *
 *   cfg / region arrays / read arrays / haplotype arrays / out_off / out / keep     as phmm_engine_compute
 *   read_soft_clip [2 n_reads] or NULL   (leading, trailing) bases of each read that are soft clips: the reference aligns the
 *                      read minus its soft clips (src/reads/alignment_utils.rs:47-50) while the PairHMM sees what
 *                      modify_read_qualities leaves (engine.rs:352-423: all of it with modify_soft_clipped_bases, else the
 *                      clipped read -- then there is nothing to clip here: NULL)
 *   region_ref_hap     REQUIRED here for every region with reads and haplotypes (left-alignment reads the reference haplotype)
 *   rcfg               Smith-Waterman parameters + overhang strategy (the reference: ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS,
 *                      SoftClip), LOG_10_INFORMATIVE_THRESHOLD, flags
 *   hap_priority ... out_cigar_off, best_allele ... status                          as phmm_realign_reads
 * PHMM_REGION_SKIP_SINGLE_ALLELE: a region with exactly one haplotype is not realigned (the reference returns before it
 * gets there, haplotype_caller_engine.rs:1339-1345): its reads keep status PHMM_PROJECT_UNCHANGED, BestAllele is still filled.
 * Any number of regions per call; large calls are pipelined in chunks of regions like phmm_engine_compute.
 *
 * phmm_region_submit is the same call through the shared, thread-safe queue of phmm_submit (phmm_wait is the one above;
 * submissions with equal configurations and the same optional arrays share a flush).
 */
#define PHMM_REGION_SKIP_SINGLE_ALLELE 1u
typedef struct phmm_realign_config {
    phmm_sw_parameters sw_parameters;   /* cigar_utils.rs:22 ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS = (10, -15, -30, -5) */
    int32_t overhang_strategy;          /* PHMM_SW_SOFTCLIP at the reference's call site (alignment_utils.rs:58) */
    uint32_t flags;                     /* PHMM_REGION_* */
    double informative_threshold;       /* LOG_10_INFORMATIVE_THRESHOLD = 0.2 (allele_likelihoods.rs:17) */
} phmm_realign_config;
int phmm_region_compute(phmm_handle *h, const phmm_engine_config *cfg, const phmm_realign_config *rcfg, uint32_t n_regions,
                        const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                        const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q,
                        const uint8_t *mapq, const uint32_t *read_soft_clip, const uint32_t *hap_off, const uint8_t *hap_bases,
                        const int32_t *region_ref_hap, const uint64_t *out_off, const int32_t *hap_priority,
                        const uint64_t *region_reference_start, const uint32_t *hap_cigar_off, const uint32_t *hap_cigar,
                        const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar,
                        const uint64_t *out_cigar_off, double *out, uint8_t *keep, int32_t *best_allele, double *likelihood,
                        double *confidence, uint32_t *out_cigar, uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status);
int phmm_region_submit(phmm_handle *h, const phmm_engine_config *cfg, const phmm_realign_config *rcfg, uint32_t n_regions,
                       const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                       const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q,
                       const uint8_t *mapq, const uint32_t *read_soft_clip, const uint32_t *hap_off, const uint8_t *hap_bases,
                       const int32_t *region_ref_hap, const uint64_t *out_off, const int32_t *hap_priority,
                       const uint64_t *region_reference_start, const uint32_t *hap_cigar_off, const uint32_t *hap_cigar,
                       const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar,
                       const uint64_t *out_cigar_off, double *out, uint8_t *keep, int32_t *best_allele, double *likelihood,
                       double *confidence, uint32_t *out_cigar, uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status,
                       uint64_t *ticket);
/* ... and over several engines, normally one per device, like phmm_compute_multi: whole regions in contiguous ranges
 * balanced by cells (phmm_split_regions), every engine on a host thread of its own next to its GPU, each range staged
 * straight from the caller's arrays, results where phmm_region_compute would put them; the arguments are checked once
 * for the whole call; on failure the message is phmm_last_error(handles[0]).  (The reference reaches several devices
 * through its rayon workers instead -- one shared handle per device, phmm_region_submit, integration/hip_backend.rs --
 * which tools/threads_bench TB_DEVICES=n imitates; this entry point is for callers that hold many regions at once.) */
int phmm_region_compute_multi(phmm_handle *const *handles, uint32_t n_handles, const phmm_engine_config *cfg,
                              const phmm_realign_config *rcfg, uint32_t n_regions, const uint32_t *region_read_off,
                              const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases,
                              const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *mapq,
                              const uint32_t *read_soft_clip, const uint32_t *hap_off, const uint8_t *hap_bases,
                              const int32_t *region_ref_hap, const uint64_t *out_off, const int32_t *hap_priority,
                              const uint64_t *region_reference_start, const uint32_t *hap_cigar_off, const uint32_t *hap_cigar,
                              const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar,
                              const uint64_t *out_cigar_off, double *out, uint8_t *keep, int32_t *best_allele, double *likelihood,
                              double *confidence, uint32_t *out_cigar, uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status);

/*
 * CigarUtils::calculate_cigar (src/reads/cigar_utils.rs:358-457) for n (reference, haplotype) pairs: the haplotype's CIGAR
 * against the reference -- the two shortcuts (empty haplotype: one D; equal lengths and at most two mismatches: one M),
 * otherwise Smith-Waterman between the sequences padded with "NNNNNNNNNN" on both sides (as phmm_sw_align; the reference's
 * callers use NEW_SW_PARAMETERS and OverhangStrategy::InDel or SoftClip), the padding trimmed off, indels left-aligned,
 * the leading / trailing deletions kept.  Empty sequences are allowed here.
 *   status [n]   0: cigar / n_cigar hold the result; 1: None (is_s_w_failure, :469-487); negative: the reference panics
 * Returns PHMM_ERR_CIGAR_CAPACITY when a slot is too small (n_cigar holds the sizes; ref + alt elements + 2 always suffice).
 */
int phmm_calculate_cigar(phmm_handle *h, uint32_t n, const uint32_t *ref_off, const uint8_t *ref_bases, const uint32_t *alt_off,
                         const uint8_t *alt_bases, const phmm_sw_parameters *params, int overhang_strategy,
                         const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar, int32_t *status);

/*
 * Developer switches and counters (tests, A/B measurements; never needed in production, NOTEBOOK.md section 11).
 * The PHMM_* environment variables of the same names (upper case) are read once, by phmm_create; phmm_set_switch changes one
 * switch of one handle afterwards.  What is left of them after round 6 (every switch whose A/B was closed went with its code):
 *   planner        "force_L" (16 / 32 / 64 lanes per pair), "force_chain" (reads per run of the chained kernel; 0 = per-read kernel
 *                  only), "force_streams", "no_pipeline" (a host-buffer call in one shot whatever its size), "no_rescue", "trace"
 *   aligner        "sw_lite" (the tags-only first pass: -1 where it pays, 0 never, 1 always), "sw_chunks", "sw_lanes",
 *                  "sw_transpose", "sw_no_zero_copy", "sw_clock"
 *   region call    "region_server" (the resident region server: -1 the one-shot calls of private handles past five alive on the
 *                  device, 0 never, 1 every call its limits admit), "server_idle_us", "server_trace";
 *                  "region_sw_all" (pairs up to which a lone launched call aligns every read against every haplotype beside the
 *                  PairHMM kernels: -1 by load, 0 never), "region_flag_wait", "region_pick_timeout_us", "region_debug_pick" (tests),
 *                  "mirror_canary", "region_own_queue" (environment only)
 *   many callers   "route_shared" (opt-in: private handles' small calls through the shared combiner)
 * Value -1 / 0 = back to the planner's choice as documented there.  Not to be called while another thread computes on the handle.
 * Returns PHMM_ERR_INVALID_ARG for an unknown name.
 * phmm_get_stat: "staged_bytes" (payload bytes this handle -- for a shared handle, its lanes -- copied into pinned
 * staging so far), "rescue_passes" (batches that needed the exact pass below -600), "sw_kernel_us" / "sw_backtrack_bytes" /
 * "sw_clock_mhz" (device time of the last phmm_sw_align's kernels by HIP events, the backtrack bytes they stored, the shader
 * clock one of their blocks saw), "sw_second_pass" (alignments of the last aligner call whose walk met a gap behind the
 * tags-only sweep and which the full instance aligned again; 0 when the call took one pass), "region_sw_all" (region calls that
 * aligned every pair beside the PairHMM kernels so far), "server_jobs" / "server_launches" / "server_broken" (the device's region
 * server: calls it has taken, times it was launched, whether it gave up); unknown names give 0.
 */
int phmm_set_switch(phmm_handle *h, const char *name, int value);
uint64_t phmm_get_stat(phmm_handle *h, const char *name);
/* Developer runs (switch "server_trace" / PHMM_SERVER_TRACE=1): the tasks the device's region server has run since its last
 * launch, one record each -- {u32 sequence number of the call, u32 kind (0 stage-in, 1 a read's chain), u32 index, u32 worker,
 * u64 claimed, u64 begun, u64 ended, u64[4] inside a chain: pre-step done, PairHMM done, post-step done, aligner done} in ticks
 * of the device's 100 MHz clock.  Copies up to `cap` records (72 bytes each) into `out`, returns how many exist; waits for
 * the server to leave the chip first.  tools/server_trace.cpp prints a call's timeline from it. */
uint32_t phmm_server_trace(int device_id, void *out, uint32_t cap);

/* What the library was built from: "cigar=<hash> pairhmm=<hash> server=<hash> sw=<hash>", the hashes of the kernel sources of each
 * family (tools/source_hash.py) at compile time.  smoke() and bench.py compare it with the tree they run in. */
const char *phmm_build_info(void);

/* Host copies of the device tables, for parity tests against the oracle:
 * eps[q] = 10^(-q/10) for q in 0..=255, mm = triangular match->match table incl. row 255. */
size_t phmm_table_eps(const double **eps);
size_t phmm_table_match_to_match(const double **mm);

#ifdef __cplusplus
}
#endif
#endif /* PHMM_H */
