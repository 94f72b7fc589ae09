//! `extern "C"` bindings of libphmm.so -- the MI355X-native PairHMM engine -- for Lorikeet.
//!
//! One declaration per export of `include/phmm.h`, same order, same argument order; nothing elided.
//! `tests/test_integration_artifacts.py` parses both files and fails when a name, an arity or a scalar width
//! differs.  This file is added to Lorikeet as `src/pair_hmm/hip_ffi.rs` by `integration/lorikeet-hip.patch`
//! (behind the cargo feature `hip`); the safe wrapper the PairHMM arm calls is `src/pair_hmm/hip_backend.rs`
//! of the same patch.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_uint, c_void};

/// Opaque engine handle (`phmm_handle`).
#[repr(C)]
pub struct phmm_handle {
    _private: [u8; 0],
}
/// Opaque launch plan (`phmm_batch`).
#[repr(C)]
pub struct phmm_batch {
    _private: [u8; 0],
}

pub const PHMM_VERSION: c_int = 1;

pub const PHMM_FLAG_NO_TRISTATE: c_uint = 1;
pub const PHMM_FLAG_F32_FIRST: c_uint = 2;

pub const PHMM_OK: c_int = 0;
pub const PHMM_ERR_INVALID_ARG: c_int = 1;
pub const PHMM_ERR_NO_DEVICE: c_int = 2;
pub const PHMM_ERR_HIP: c_int = 3;
pub const PHMM_ERR_POSITIVE_RESULT: c_int = 4;
pub const PHMM_ERR_NOT_BOUND: c_int = 5;
pub const PHMM_ERR_NO_MEMORY: c_int = 6;
pub const PHMM_ERR_INTERNAL: c_int = 7;
pub const PHMM_ERR_CIGAR_CAPACITY: c_int = 8;

/// `overhang_strategy` of `phmm_sw_align` == gkl::smithwaterman::OverhangStrategy
pub const PHMM_SW_SOFTCLIP: c_int = 0;
pub const PHMM_SW_INDEL: c_int = 1;
pub const PHMM_SW_LEADING_INDEL: c_int = 2;
pub const PHMM_SW_IGNORE: c_int = 3;
/// `ref_index` value of `phmm_sw_align_indexed`: this alignment is skipped
pub const PHMM_SW_NO_REFERENCE: u32 = 0xffff_ffff;
/// `status` of `phmm_project_to_reference` (negative values: the read is one the reference panics on)
pub const PHMM_PROJECT_REALIGNED: c_int = 0;
pub const PHMM_PROJECT_UNCHANGED: c_int = 1;

/// `phmm_sw_parameters` == gkl::smithwaterman::Parameters::new(match, mismatch, gap open, gap extend)
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct phmm_sw_parameters {
    pub match_value: i32,
    pub mismatch_penalty: i32,
    pub gap_open_penalty: i32,
    pub gap_extend_penalty: i32,
}

/// `phmm_plan_info`: the launch plan of a batch, computed on the host alone (`phmm_plan_describe`)
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct phmm_plan_info {
    pub cells: u64,
    pub chain_cells: u64,
    pub chain_items: u64,
    pub n_launches: u32,
    pub n_chain_launches: u32,
    pub min_reads_per_run: u32,
    pub reserved: u32,
    pub dominant_kernel: [c_char; 64],
}

/// `flags` of `phmm_realign_config`: a region with exactly one haplotype is not realigned
/// (src/haplotype/haplotype_caller_engine.rs:1339-1345 returns before it gets there)
pub const PHMM_REGION_SKIP_SINGLE_ALLELE: c_uint = 1;

/// `phmm_realign_config`: what `realign_reads_to_their_best_haplotype` fixes at its call site
/// (src/reads/alignment_utils.rs:52-58, src/model/allele_likelihoods.rs:17)
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct phmm_realign_config {
    pub sw_parameters: phmm_sw_parameters,
    pub overhang_strategy: i32,
    pub flags: u32,
    pub informative_threshold: f64,
}

/// `phmm_engine_config`: the arguments of `PairHMMLikelihoodCalculationEngine::new`
/// (src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs:129-141) the device needs.
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct phmm_engine_config {
    pub constant_gcp: u8,
    pub pcr_error_model: u8,
    pub base_quality_score_threshold: u8,
    pub dynamic_read_disqualification: u8,
    pub symmetrically_normalize_alleles_to_reference: u8,
    pub disable_cap_read_qualities_to_mapq: u8,
    pub reserved: [u8; 2],
    pub log10_global_read_mismapping_rate: f64,
    pub read_disqualification_scale: f64,
    pub expected_error_rate_per_base: f64,
}

extern "C" {
    pub fn phmm_device_count() -> c_int;
    pub fn phmm_create(device_id: c_int, flags: c_uint) -> *mut phmm_handle;
    pub fn phmm_destroy(h: *mut phmm_handle);
    pub fn phmm_last_error(h: *mut phmm_handle) -> *const c_char;

    pub fn phmm_compute(
        h: *mut phmm_handle,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        gcp: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        out_off: *const u64,
        out: *mut f64,
    ) -> c_int;

    pub fn phmm_submit(
        h: *mut phmm_handle,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        gcp: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        out_off: *const u64,
        out: *mut f64,
        ticket: *mut u64,
    ) -> c_int;
    pub fn phmm_wait(h: *mut phmm_handle, ticket: u64) -> c_int;
    pub fn phmm_submit_stats(h: *mut phmm_handle, n_flushes: *mut u64, n_submissions: *mut u64);

    pub fn phmm_assign_regions(
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        hap_off: *const u32,
        n_parts: u32,
        part_of_region: *mut u32,
    ) -> c_int;
    pub fn phmm_split_regions(
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        hap_off: *const u32,
        n_parts: u32,
        first_region: *mut u32,
    ) -> c_int;
    pub fn phmm_compute_multi(
        handles: *const *mut phmm_handle,
        n_handles: u32,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        gcp: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        out_off: *const u64,
        out: *mut f64,
    ) -> c_int;

    pub fn phmm_batch_create(
        h: *mut phmm_handle,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        hap_off: *const u32,
        out_off: *const u64,
    ) -> *mut phmm_batch;
    pub fn phmm_batch_destroy(b: *mut phmm_batch);
    pub fn phmm_batch_bind_device(
        b: *mut phmm_batch,
        d_read_bases: *const u8,
        d_base_q: *const u8,
        d_ins_q: *const u8,
        d_del_q: *const u8,
        d_gcp: *const u8,
        d_hap_bases: *const u8,
        d_out: *mut f64,
    ) -> c_int;
    pub fn phmm_batch_upload(
        b: *mut phmm_batch,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        gcp: *const u8,
        hap_bases: *const u8,
    ) -> c_int;
    pub fn phmm_batch_launch(b: *mut phmm_batch, stream: *mut c_void) -> c_int;
    pub fn phmm_batch_download(b: *mut phmm_batch, out: *mut f64) -> c_int;
    pub fn phmm_batch_status(b: *mut phmm_batch) -> c_int;
    pub fn phmm_batch_cells(b: *const phmm_batch) -> u64;
    pub fn phmm_batch_algorithmic_bytes(b: *const phmm_batch) -> u64;
    pub fn phmm_batch_num_launches(b: *const phmm_batch) -> u32;
    pub fn phmm_batch_executed_cells(b: *const phmm_batch) -> u64;
    pub fn phmm_batch_dominant_kernel(b: *const phmm_batch) -> *const c_char;
    pub fn phmm_plan_describe(
        flags: c_uint,
        concurrent_callers: u32,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        hap_off: *const u32,
        info: *mut phmm_plan_info,
    ) -> c_int;

    pub fn phmm_engine_compute(
        h: *mut phmm_handle,
        cfg: *const phmm_engine_config,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        mapq: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        region_ref_hap: *const i32,
        out_off: *const u64,
        out: *mut f64,
        keep: *mut u8,
    ) -> c_int;
    pub fn phmm_engine_submit(
        h: *mut phmm_handle,
        cfg: *const phmm_engine_config,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        mapq: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        region_ref_hap: *const i32,
        out_off: *const u64,
        out: *mut f64,
        keep: *mut u8,
        ticket: *mut u64,
    ) -> c_int;

    pub fn phmm_engine_compute_multi(
        handles: *const *mut phmm_handle,
        n_handles: u32,
        cfg: *const phmm_engine_config,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        mapq: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        region_ref_hap: *const i32,
        out_off: *const u64,
        out: *mut f64,
        keep: *mut u8,
    ) -> c_int;

    pub fn phmm_sw_align(
        h: *mut phmm_handle,
        n_alignments: u32,
        ref_off: *const u32,
        ref_bases: *const u8,
        alt_off: *const u32,
        alt_bases: *const u8,
        params: *const phmm_sw_parameters,
        overhang_strategy: c_int,
        cigar_off: *const u64,
        cigar: *mut u32,
        n_cigar: *mut u32,
        alignment_offset: *mut i32,
    ) -> c_int;

    pub fn phmm_sw_align_indexed(
        h: *mut phmm_handle,
        n_references: u32,
        ref_off: *const u32,
        ref_bases: *const u8,
        n_alignments: u32,
        ref_index: *const u32,
        alt_off: *const u32,
        alt_bases: *const u8,
        params: *const phmm_sw_parameters,
        overhang_strategy: c_int,
        cigar_off: *const u64,
        cigar: *mut u32,
        n_cigar: *mut u32,
        alignment_offset: *mut i32,
    ) -> c_int;

    pub fn phmm_best_alleles(
        h: *mut phmm_handle,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        out_off: *const u64,
        likelihoods: *const f64,
        keep: *const u8,
        hap_priority: *const i32,
        informative_threshold: f64,
        best_allele: *mut i32,
        likelihood: *mut f64,
        confidence: *mut f64,
    ) -> c_int;

    pub fn phmm_realign_to_best(
        h: *mut phmm_handle,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        out_off: *const u64,
        likelihoods: *const f64,
        keep: *const u8,
        hap_priority: *const i32,
        informative_threshold: f64,
        params: *const phmm_sw_parameters,
        overhang_strategy: c_int,
        cigar_off: *const u64,
        cigar: *mut u32,
        n_cigar: *mut u32,
        alignment_offset: *mut i32,
        best_allele: *mut i32,
        likelihood: *mut f64,
        confidence: *mut f64,
    ) -> c_int;

    pub fn phmm_project_to_reference(
        h: *mut phmm_handle,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        region_ref_hap: *const i32,
        region_reference_start: *const u64,
        hap_cigar_off: *const u32,
        hap_cigar: *const u32,
        hap_start_wrt_ref: *const u32,
        best_allele: *const i32,
        sw_cigar_off: *const u64,
        sw_cigar: *const u32,
        n_sw_cigar: *const u32,
        sw_offset: *const i32,
        orig_cigar_off: *const u32,
        orig_cigar: *const u32,
        out_cigar_off: *const u64,
        out_cigar: *mut u32,
        n_out_cigar: *mut u32,
        new_pos: *mut i64,
        status: *mut i32,
    ) -> c_int;

    pub fn phmm_realign_reads(
        h: *mut phmm_handle,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        hap_off: *const u32,
        hap_bases: *const u8,
        out_off: *const u64,
        likelihoods: *const f64,
        keep: *const u8,
        hap_priority: *const i32,
        informative_threshold: f64,
        params: *const phmm_sw_parameters,
        overhang_strategy: c_int,
        region_ref_hap: *const i32,
        region_reference_start: *const u64,
        hap_cigar_off: *const u32,
        hap_cigar: *const u32,
        hap_start_wrt_ref: *const u32,
        orig_cigar_off: *const u32,
        orig_cigar: *const u32,
        out_cigar_off: *const u64,
        out_cigar: *mut u32,
        n_out_cigar: *mut u32,
        new_pos: *mut i64,
        status: *mut i32,
        best_allele: *mut i32,
        likelihood: *mut f64,
        confidence: *mut f64,
    ) -> c_int;

    pub fn phmm_region_compute(
        h: *mut phmm_handle,
        cfg: *const phmm_engine_config,
        rcfg: *const phmm_realign_config,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        mapq: *const u8,
        read_soft_clip: *const u32,
        hap_off: *const u32,
        hap_bases: *const u8,
        region_ref_hap: *const i32,
        out_off: *const u64,
        hap_priority: *const i32,
        region_reference_start: *const u64,
        hap_cigar_off: *const u32,
        hap_cigar: *const u32,
        hap_start_wrt_ref: *const u32,
        orig_cigar_off: *const u32,
        orig_cigar: *const u32,
        out_cigar_off: *const u64,
        out: *mut f64,
        keep: *mut u8,
        best_allele: *mut i32,
        likelihood: *mut f64,
        confidence: *mut f64,
        out_cigar: *mut u32,
        n_out_cigar: *mut u32,
        new_pos: *mut i64,
        status: *mut i32,
    ) -> c_int;

    pub fn phmm_region_submit(
        h: *mut phmm_handle,
        cfg: *const phmm_engine_config,
        rcfg: *const phmm_realign_config,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        mapq: *const u8,
        read_soft_clip: *const u32,
        hap_off: *const u32,
        hap_bases: *const u8,
        region_ref_hap: *const i32,
        out_off: *const u64,
        hap_priority: *const i32,
        region_reference_start: *const u64,
        hap_cigar_off: *const u32,
        hap_cigar: *const u32,
        hap_start_wrt_ref: *const u32,
        orig_cigar_off: *const u32,
        orig_cigar: *const u32,
        out_cigar_off: *const u64,
        out: *mut f64,
        keep: *mut u8,
        best_allele: *mut i32,
        likelihood: *mut f64,
        confidence: *mut f64,
        out_cigar: *mut u32,
        n_out_cigar: *mut u32,
        new_pos: *mut i64,
        status: *mut i32,
        ticket: *mut u64,
    ) -> c_int;

    pub fn phmm_region_compute_multi(
        handles: *const *mut phmm_handle,
        n_handles: u32,
        cfg: *const phmm_engine_config,
        rcfg: *const phmm_realign_config,
        n_regions: u32,
        region_read_off: *const u32,
        region_hap_off: *const u32,
        read_off: *const u32,
        read_bases: *const u8,
        base_q: *const u8,
        ins_q: *const u8,
        del_q: *const u8,
        mapq: *const u8,
        read_soft_clip: *const u32,
        hap_off: *const u32,
        hap_bases: *const u8,
        region_ref_hap: *const i32,
        out_off: *const u64,
        hap_priority: *const i32,
        region_reference_start: *const u64,
        hap_cigar_off: *const u32,
        hap_cigar: *const u32,
        hap_start_wrt_ref: *const u32,
        orig_cigar_off: *const u32,
        orig_cigar: *const u32,
        out_cigar_off: *const u64,
        out: *mut f64,
        keep: *mut u8,
        best_allele: *mut i32,
        likelihood: *mut f64,
        confidence: *mut f64,
        out_cigar: *mut u32,
        n_out_cigar: *mut u32,
        new_pos: *mut i64,
        status: *mut i32,
    ) -> c_int;

    pub fn phmm_calculate_cigar(
        h: *mut phmm_handle,
        n: u32,
        ref_off: *const u32,
        ref_bases: *const u8,
        alt_off: *const u32,
        alt_bases: *const u8,
        params: *const phmm_sw_parameters,
        overhang_strategy: c_int,
        cigar_off: *const u64,
        cigar: *mut u32,
        n_cigar: *mut u32,
        status: *mut i32,
    ) -> c_int;

    pub fn phmm_set_switch(h: *mut phmm_handle, name: *const c_char, value: c_int) -> c_int;
    pub fn phmm_get_stat(h: *mut phmm_handle, name: *const c_char) -> u64;
    /// (developer runs: the task records of the device's region server, 72 bytes each)
    pub fn phmm_server_trace(device_id: c_int, out: *mut c_void, cap: u32) -> u32;
    /// "cigar=<hash> pairhmm=<hash> server=<hash> sw=<hash>": the kernel sources the library was built from
    pub fn phmm_build_info() -> *const c_char;

    pub fn phmm_table_eps(eps: *mut *const f64) -> usize;
    pub fn phmm_table_match_to_match(mm: *mut *const f64) -> usize;
}
