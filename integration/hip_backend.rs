//! Safe side of the MI355X PairHMM binding: flattens the data `PairHMM::compute_likelihoods` already holds into the
//! struct-of-arrays `phmm_compute` takes (include/phmm.h) and turns a non-zero status into the panic the rest of
//! the PairHMM path uses for violated preconditions.
//!
//! Two kinds of engine handles.  The per-region pipeline (`region_compute`: everything between
//! `compute_read_likelihoods` and `change_evidence`) goes through ONE shared handle per device with
//! `phmm_region_submit` / `phmm_wait`, so that the regions of all rayon workers waiting at a moment are computed as one
//! batch.  The single-step entry points (`compute_likelihoods`, `realign_reads`, `calculate_cigars`) use one private
//! handle per rayon worker: Lorikeet clones its likelihood engine per region task
//! (src/assembly/assembly_region_walker.rs) and calls synchronously from every worker, and such a handle must not be
//! shared between threads.  Workers are spread over the visible devices round-robin.
use std::cell::RefCell;
use std::ffi::CStr;
use std::sync::atomic::{AtomicU32, Ordering};

use crate::pair_hmm::hip_ffi::*;

/// Flags every engine of this process is created with (`phmm_create`, include/phmm.h).
static ENGINE_FLAGS: AtomicU32 = AtomicU32::new(0);

/// `--pairhmm-backend hip-f32`: engines compute in f32 first (state scaled by 2^120) and redo in f64 what underflows --
/// the arithmetic of the reference's production arm (gkl's AVX kernel, called at src/pair_hmm/pair_hmm.rs:348-366), within
/// 1e-5 of the scalar arm like that one, and 1.5x the f64 rate.  `--pairhmm-backend hip` is f64 throughout (the scalar
/// arm's arithmetic).  Takes effect for engines created afterwards: `AVXMode::select` calls it before any region is computed.
pub fn set_f32_first(on: bool) {
    if on {
        ENGINE_FLAGS.fetch_or(PHMM_FLAG_F32_FIRST as u32, Ordering::Relaxed);
    } else {
        ENGINE_FLAGS.fetch_and(!(PHMM_FLAG_F32_FIRST as u32), Ordering::Relaxed);
    }
}

/// ... or LORIKEET_HIP_F32_FIRST=1 in the environment (read when an engine is created), for runs that cannot change the
/// command line.
fn engine_flags() -> std::os::raw::c_uint {
    let mut flags = ENGINE_FLAGS.load(Ordering::Relaxed);
    if std::env::var("LORIKEET_HIP_F32_FIRST").map(|v| !v.is_empty() && v != "0").unwrap_or(false) {
        flags |= PHMM_FLAG_F32_FIRST as u32;
    }
    flags as std::os::raw::c_uint
}

struct Engine(*mut phmm_handle);

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { phmm_destroy(self.0) }
    }
}

thread_local! {
    static ENGINE: RefCell<Option<Engine>> = RefCell::new(None);
}

/// Number of MI355X devices the process can use (0: the backend is unavailable).
pub fn device_count() -> i32 {
    unsafe { phmm_device_count() }
}

fn last_error(h: *mut phmm_handle) -> String {
    unsafe { CStr::from_ptr(phmm_last_error(h)).to_string_lossy().into_owned() }
}

fn with_engine<R>(f: impl FnOnce(*mut phmm_handle) -> R) -> R {
    ENGINE.with(|cell| {
        let mut slot = cell.borrow_mut();
        if slot.is_none() {
            let n = device_count().max(1);
            let device = (rayon::current_thread_index().unwrap_or(0) as i32) % n;
            let h = unsafe { phmm_create(device, engine_flags()) };
            if h.is_null() {
                panic!("HIP PairHMM: {}", last_error(std::ptr::null_mut()));
            }
            *slot = Some(Engine(h));
        }
        f(slot.as_ref().unwrap().0)
    })
}

/// log10 Pr(read | haplotype) for every read x haplotype of one region: read-major, haplotypes in list order --
/// the layout of `PairHMM::m_log_likelihood_array`.
pub fn compute_likelihoods(
    haplotypes: &[&[u8]],
    read_bases: &[&[u8]],
    read_quals: &[&[u8]],
    insertion_gop: &[&[u8]],
    deletion_gop: &[&[u8]],
    overall_gcp: &[&[u8]],
) -> Vec<f64> {
    let n_reads = read_bases.len();
    let n_haps = haplotypes.len();
    let mut read_off: Vec<u32> = Vec::with_capacity(n_reads + 1);
    read_off.push(0);
    let total: usize = read_bases.iter().map(|r| r.len()).sum();
    let (mut bases, mut quals, mut ins, mut del, mut gcp) = (
        Vec::with_capacity(total),
        Vec::with_capacity(total),
        Vec::with_capacity(total),
        Vec::with_capacity(total),
        Vec::with_capacity(total),
    );
    for r in 0..n_reads {
        let n = read_bases[r].len();
        assert!(
            read_quals[r].len() == n && insertion_gop[r].len() == n && deletion_gop[r].len() == n && overall_gcp[r].len() == n,
            "Read bases and read quals aren't the same size"
        );
        bases.extend_from_slice(read_bases[r]);
        quals.extend_from_slice(read_quals[r]);
        ins.extend_from_slice(insertion_gop[r]);
        del.extend_from_slice(deletion_gop[r]);
        gcp.extend_from_slice(overall_gcp[r]);
        read_off.push(bases.len() as u32);
    }
    let mut hap_off: Vec<u32> = Vec::with_capacity(n_haps + 1);
    hap_off.push(0);
    let mut haps: Vec<u8> = Vec::with_capacity(haplotypes.iter().map(|h| h.len()).sum());
    for h in haplotypes {
        haps.extend_from_slice(h);
        hap_off.push(haps.len() as u32);
    }
    let mut out = vec![0.0f64; n_reads * n_haps];
    let region_read_off = [0u32, n_reads as u32];
    let region_hap_off = [0u32, n_haps as u32];
    let out_off = [0u64, (n_reads * n_haps) as u64];
    with_engine(|h| {
        let rc = unsafe {
            phmm_compute(
                h,
                1,
                region_read_off.as_ptr(),
                region_hap_off.as_ptr(),
                read_off.as_ptr(),
                bases.as_ptr(),
                quals.as_ptr(),
                ins.as_ptr(),
                del.as_ptr(),
                gcp.as_ptr(),
                hap_off.as_ptr(),
                haps.as_ptr(),
                out_off.as_ptr(),
                out.as_mut_ptr(),
            )
        };
        if rc != PHMM_OK {
            // the scalar arm asserts the same conditions (pair_hmm.rs: argument sizes, result <= 0)
            panic!("HIP PairHMM failed ({}): {}", rc, last_error(h));
        }
    });
    out
}

/// One unit of evidence after `realign_to_best`: `AlleleLikelihoods::BestAllele` (allele index, likelihood,
/// confidence) and the read's Smith-Waterman alignment to that haplotype (BAM-encoded CIGAR elements,
/// `(length << 4) | op`, and `SmithWatermanAlignmentResult::alignment_offset`).  `allele_index` is `None`, and the
/// CIGAR empty, where the reference has no best allele for the read.
pub struct RealignedToBest {
    pub allele_index: Option<usize>,
    pub likelihood: f64,
    pub confidence: f64,
    pub cigar: Vec<u32>,
    pub alignment_offset: i32,
}

/// The arithmetic of `AssemblyBasedCallerUtils::realign_reads_to_their_best_haplotype`
/// (src/assembly/assembly_based_caller_utils.rs:208-246) for one region in one call: the best allele of every read
/// with ties broken by `priorities` (`AlleleLikelihoods::best_alleles_breaking_ties_main`,
/// src/model/allele_likelihoods.rs:1043-1095) and the read's alignment to that haplotype with
/// `ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS` and `OverhangStrategy::SoftClip`
/// (src/reads/alignment_utils.rs:52-58).  `likelihoods` is read-major (`[read][haplotype]`, what
/// `compute_likelihoods` above returns); `reads_minus_soft_clips` are the hard-clipped reads of
/// alignment_utils.rs:47-50; `priorities` one value per haplotype
/// (`haplotype_alignment_tiebreaking_priority`, assembly_based_caller_utils.rs:187-195).
pub fn realign_to_best(
    haplotypes: &[&[u8]],
    reads_minus_soft_clips: &[&[u8]],
    likelihoods: &[f64],
    priorities: &[i32],
) -> Vec<RealignedToBest> {
    let n_reads = reads_minus_soft_clips.len();
    let n_haps = haplotypes.len();
    assert!(likelihoods.len() == n_reads * n_haps && priorities.len() == n_haps, "one likelihood per read and haplotype, one priority per haplotype");
    let mut read_off: Vec<u32> = Vec::with_capacity(n_reads + 1);
    read_off.push(0);
    let mut bases: Vec<u8> = Vec::with_capacity(reads_minus_soft_clips.iter().map(|r| r.len()).sum());
    for r in reads_minus_soft_clips {
        bases.extend_from_slice(r);
        read_off.push(bases.len() as u32);
    }
    let mut hap_off: Vec<u32> = Vec::with_capacity(n_haps + 1);
    hap_off.push(0);
    let mut haps: Vec<u8> = Vec::with_capacity(haplotypes.iter().map(|h| h.len()).sum());
    for h in haplotypes {
        haps.extend_from_slice(h);
        hap_off.push(haps.len() as u32);
    }
    let region_read_off = [0u32, n_reads as u32];
    let region_hap_off = [0u32, n_haps as u32];
    let out_off = [0u64, (n_reads * n_haps) as u64];
    // ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS (src/smith_waterman/smith_waterman_aligner.rs:23-26)
    let params = phmm_sw_parameters { match_value: 10, mismatch_penalty: -15, gap_open_penalty: -30, gap_extend_penalty: -5 };
    let mut capacity = vec![16u64; n_reads];
    let mut best = vec![0i32; n_reads];
    let mut likelihood = vec![0.0f64; n_reads];
    let mut confidence = vec![0.0f64; n_reads];
    let mut n_cigar = vec![0u32; n_reads];
    let mut offset = vec![0i32; n_reads];
    let mut cigar_off = vec![0u64; n_reads + 1];
    let mut cigar: Vec<u32> = Vec::new();
    with_engine(|h| {
        for attempt in 0..2 {
            for r in 0..n_reads {
                cigar_off[r + 1] = cigar_off[r] + capacity[r];
            }
            cigar = vec![0u32; cigar_off[n_reads] as usize];
            let rc = unsafe {
                phmm_realign_to_best(
                    h,
                    1,
                    region_read_off.as_ptr(),
                    region_hap_off.as_ptr(),
                    read_off.as_ptr(),
                    bases.as_ptr(),
                    hap_off.as_ptr(),
                    haps.as_ptr(),
                    out_off.as_ptr(),
                    likelihoods.as_ptr(),
                    std::ptr::null(),
                    priorities.as_ptr(),
                    0.2, // LOG_10_INFORMATIVE_THRESHOLD (src/model/allele_likelihoods.rs:17)
                    &params,
                    PHMM_SW_SOFTCLIP,
                    cigar_off.as_ptr(),
                    cigar.as_mut_ptr(),
                    n_cigar.as_mut_ptr(),
                    offset.as_mut_ptr(),
                    best.as_mut_ptr(),
                    likelihood.as_mut_ptr(),
                    confidence.as_mut_ptr(),
                )
            };
            if rc == PHMM_ERR_CIGAR_CAPACITY && attempt == 0 {
                // the library reports the sizes: once more with those
                for r in 0..n_reads {
                    capacity[r] = capacity[r].max(n_cigar[r] as u64);
                }
                continue;
            }
            if rc != PHMM_OK {
                panic!("HIP realignment failed ({}): {}", rc, last_error(h));
            }
            break;
        }
    });
    (0..n_reads)
        .map(|r| {
            let start = cigar_off[r] as usize;
            RealignedToBest {
                allele_index: if best[r] >= 0 { Some(best[r] as usize) } else { None },
                likelihood: likelihood[r],
                confidence: confidence[r],
                cigar: cigar[start..start + n_cigar[r] as usize].to_vec(),
                alignment_offset: offset[r],
            }
        })
        .collect()
}

/// One read after `realign_reads`: its `BestAllele` and, when `realigned`, the position and CIGAR
/// `AlignmentUtils::create_read_aligned_to_ref` would give it (BAM-encoded elements, clips of the original CIGAR
/// included).  `realigned == false`: the reference returns the read unchanged (no best allele, or
/// `alignment_offset == -1`, src/reads/alignment_utils.rs:60-63).
pub struct RealignedRead {
    pub allele_index: Option<usize>,
    pub likelihood: f64,
    pub confidence: f64,
    pub realigned: bool,
    pub pos: i64,
    pub cigar: Vec<u32>,
}

/// `AssemblyBasedCallerUtils::realign_reads_to_their_best_haplotype`
/// (src/assembly/assembly_based_caller_utils.rs:208-246) for one region in one call of the library: best alleles,
/// the reads' alignments to them and `create_read_aligned_to_ref`'s projection onto the reference
/// (src/reads/alignment_utils.rs:83-165).  `haplotype_cigars[a]` / `alignment_start_hap_wrt_ref[a]` are
/// `Haplotype::cigar` / `alignment_start_hap_wrt_ref` of haplotype `a`, `reference_haplotype` its index,
/// `reference_start` is `padded_reference_loc.get_start()`, `original_cigars[r]` the read's CIGAR before realignment.
/// Panics where the reference panics ("Read goes past end of reference", builder errors ...).
#[allow(clippy::too_many_arguments)]
pub fn realign_reads(
    haplotypes: &[&[u8]],
    haplotype_cigars: &[&[u32]],
    alignment_start_hap_wrt_ref: &[u32],
    reference_haplotype: usize,
    reference_start: u64,
    reads_minus_soft_clips: &[&[u8]],
    original_cigars: &[&[u32]],
    likelihoods: &[f64],
    priorities: &[i32],
) -> Vec<RealignedRead> {
    let n_reads = reads_minus_soft_clips.len();
    let n_haps = haplotypes.len();
    assert!(likelihoods.len() == n_reads * n_haps && priorities.len() == n_haps, "one likelihood per read and haplotype, one priority per haplotype");
    assert!(haplotype_cigars.len() == n_haps && alignment_start_hap_wrt_ref.len() == n_haps && original_cigars.len() == n_reads);
    let flatten_u8 = |parts: &[&[u8]]| {
        let mut off: Vec<u32> = vec![0];
        let mut all: Vec<u8> = Vec::new();
        for p in parts {
            all.extend_from_slice(p);
            off.push(all.len() as u32);
        }
        (off, all)
    };
    let flatten_u32 = |parts: &[&[u32]]| {
        let mut off: Vec<u32> = vec![0];
        let mut all: Vec<u32> = Vec::new();
        for p in parts {
            all.extend_from_slice(p);
            off.push(all.len() as u32);
        }
        (off, all)
    };
    let (read_off, bases) = flatten_u8(reads_minus_soft_clips);
    let (hap_off, haps) = flatten_u8(haplotypes);
    let (hap_cigar_off, hap_cigar) = flatten_u32(haplotype_cigars);
    let (orig_cigar_off, orig_cigar) = flatten_u32(original_cigars);
    let region_read_off = [0u32, n_reads as u32];
    let region_hap_off = [0u32, n_haps as u32];
    let out_off = [0u64, (n_reads * n_haps) as u64];
    let region_ref_hap = [reference_haplotype as i32];
    let region_reference_start = [reference_start];
    let params = phmm_sw_parameters { match_value: 10, mismatch_penalty: -15, gap_open_penalty: -30, gap_extend_penalty: -5 };
    let mut capacity = vec![16u64; n_reads];
    let mut best = vec![0i32; n_reads];
    let mut likelihood = vec![0.0f64; n_reads];
    let mut confidence = vec![0.0f64; n_reads];
    let mut n_out = vec![0u32; n_reads];
    let mut pos = vec![0i64; n_reads];
    let mut status = vec![0i32; n_reads];
    let mut out_cigar_off = vec![0u64; n_reads + 1];
    let mut out_cigar: Vec<u32> = Vec::new();
    with_engine(|h| {
        for attempt in 0..2 {
            for r in 0..n_reads {
                out_cigar_off[r + 1] = out_cigar_off[r] + capacity[r];
            }
            out_cigar = vec![0u32; out_cigar_off[n_reads] as usize];
            let rc = unsafe {
                phmm_realign_reads(
                    h,
                    1,
                    region_read_off.as_ptr(),
                    region_hap_off.as_ptr(),
                    read_off.as_ptr(),
                    bases.as_ptr(),
                    hap_off.as_ptr(),
                    haps.as_ptr(),
                    out_off.as_ptr(),
                    likelihoods.as_ptr(),
                    std::ptr::null(),
                    priorities.as_ptr(),
                    0.2,
                    &params,
                    PHMM_SW_SOFTCLIP,
                    region_ref_hap.as_ptr(),
                    region_reference_start.as_ptr(),
                    hap_cigar_off.as_ptr(),
                    hap_cigar.as_ptr(),
                    alignment_start_hap_wrt_ref.as_ptr(),
                    orig_cigar_off.as_ptr(),
                    orig_cigar.as_ptr(),
                    out_cigar_off.as_ptr(),
                    out_cigar.as_mut_ptr(),
                    n_out.as_mut_ptr(),
                    pos.as_mut_ptr(),
                    status.as_mut_ptr(),
                    best.as_mut_ptr(),
                    likelihood.as_mut_ptr(),
                    confidence.as_mut_ptr(),
                )
            };
            if rc == PHMM_ERR_CIGAR_CAPACITY && attempt == 0 {
                for r in 0..n_reads {
                    capacity[r] = capacity[r].max(n_out[r] as u64);
                }
                continue;
            }
            if rc != PHMM_OK {
                panic!("HIP realignment failed ({}): {}", rc, last_error(h));
            }
            break;
        }
    });
    (0..n_reads)
        .map(|r| {
            if status[r] < 0 {
                // the reference panics on this read (builder error, read past the end of the reference, ...)
                panic!("Failed to realign read {} (status {})", r, status[r]);
            }
            let start = out_cigar_off[r] as usize;
            RealignedRead {
                allele_index: if best[r] >= 0 { Some(best[r] as usize) } else { None },
                likelihood: likelihood[r],
                confidence: confidence[r],
                realigned: status[r] == PHMM_PROJECT_REALIGNED,
                pos: pos[r],
                cigar: out_cigar[start..start + n_out[r] as usize].to_vec(),
            }
        })
        .collect()
}

// ---- the whole per-region path in one call (phmm_region_submit / phmm_wait on ONE shared handle) ----------------------

/// BAM encoding of a CIGAR: `(length << 4) | op`, M = 0, I = 1, D = 2, N = 3, S = 4, H = 5, P = 6, `=` = 7, X = 8.
pub fn encode_cigar(cigar: &[rust_htslib::bam::record::Cigar]) -> Vec<u32> {
    use rust_htslib::bam::record::Cigar;
    cigar
        .iter()
        .map(|c| {
            let op = match c {
                Cigar::Match(_) => 0u32,
                Cigar::Ins(_) => 1,
                Cigar::Del(_) => 2,
                Cigar::RefSkip(_) => 3,
                Cigar::SoftClip(_) => 4,
                Cigar::HardClip(_) => 5,
                Cigar::Pad(_) => 6,
                Cigar::Equal(_) => 7,
                Cigar::Diff(_) => 8,
            };
            (c.len() << 4) | op
        })
        .collect()
}

/// The inverse of `encode_cigar`.
pub fn decode_cigar(elements: &[u32]) -> rust_htslib::bam::record::CigarString {
    use rust_htslib::bam::record::{Cigar, CigarString};
    CigarString(
        elements
            .iter()
            .map(|e| {
                let len = e >> 4;
                match e & 0xf {
                    0 => Cigar::Match(len),
                    1 => Cigar::Ins(len),
                    2 => Cigar::Del(len),
                    3 => Cigar::RefSkip(len),
                    4 => Cigar::SoftClip(len),
                    5 => Cigar::HardClip(len),
                    6 => Cigar::Pad(len),
                    7 => Cigar::Equal(len),
                    8 => Cigar::Diff(len),
                    other => panic!("Unknown CIGAR operator code {}", other),
                }
            })
            .collect(),
    )
}

/// (leading, trailing) soft-clipped bases of a read: what `ReadClipper::hard_clip_soft_clipped_bases`
/// (src/reads/read_clipper.rs:395-435) cuts off its two ends.
pub fn soft_clips(cigar: &[rust_htslib::bam::record::Cigar]) -> (u32, u32) {
    use rust_htslib::bam::record::Cigar;
    let (mut lead, mut trail, mut right_tail) = (0u32, 0u32, false);
    for c in cigar {
        match c {
            Cigar::SoftClip(n) => {
                if right_tail {
                    trail += *n;
                } else {
                    lead += *n;
                }
            }
            Cigar::HardClip(_) => {}
            _ => {
                right_tail = true;
                trail = 0;
            }
        }
    }
    (lead, trail)
}

struct Shared(Vec<*mut phmm_handle>);
// phmm_submit / phmm_region_submit / phmm_wait are the entry points of the library that any number of threads may call
// on one handle (include/phmm.h)
unsafe impl Send for Shared {}
unsafe impl Sync for Shared {}

lazy_static! {
    /// One engine per device for ALL rayon workers: a worker's region is only queued, and the first worker that waits
    /// while an engine lane is free computes the regions of every worker waiting at that moment as ONE batch and hands
    /// each its results (include/phmm.h, phmm_submit).  Lorikeet's call pattern -- one region per call from every
    /// worker, src/assembly/assembly_region_walker.rs:210-273 -- then fills the device instead of a fraction of it.
    static ref SHARED: Shared = {
        let n = device_count().max(1);
        Shared(
            (0..n)
                .map(|device| {
                    let h = unsafe { phmm_create(device, engine_flags()) };
                    if h.is_null() {
                        panic!("HIP PairHMM: {}", last_error(std::ptr::null_mut()));
                    }
                    h
                })
                .collect(),
        )
    };
}

/// The shared engine of this worker's device (workers are spread over the visible devices round-robin).
fn shared_engine() -> *mut phmm_handle {
    let handles = &SHARED.0;
    handles[rayon::current_thread_index().unwrap_or(0) % handles.len()]
}

/// What `PairHMMLikelihoodCalculationEngine::compute_read_likelihoods` and
/// `AssemblyBasedCallerUtils::realign_reads_to_their_best_haplotype` produce together for one region.
pub struct RegionOutput {
    /// normalised log10 likelihoods, read-major (`[read][haplotype]`)
    pub likelihoods: Vec<f64>,
    /// `false`: `filter_poorly_modeled_evidence` removes the read (src/model/allele_likelihoods.rs:925-964)
    pub keep: Vec<bool>,
    /// per read: `BestAllele` and the realigned position / CIGAR (`realigned == false` for removed reads)
    pub reads: Vec<RealignedRead>,
}

/// One call for everything `HaplotypeCallerEngine::call_region` does with numbers between
/// `compute_read_likelihoods` and `change_evidence` (src/haplotype/haplotype_caller_engine.rs:1311-1357): the PCR
/// indel model and quality caps (`modify_read_qualities`), the PairHMM, `normalize_likelihoods`, the keep / remove
/// decision of `filter_poorly_modeled_evidence`, the best allele of every surviving read with
/// `haplotype_alignment_tiebreaking_priority`, its Smith-Waterman alignment to that haplotype and
/// `create_read_aligned_to_ref`'s projection onto the reference.  The likelihood matrix never leaves the device in
/// between.  `read_quals` / `ins_quals` / `del_quals` are the ORIGINAL qualities of what the PairHMM sees
/// (`read.qual()`, BI / BD or the flat Q45 default), `soft_clips[r]` the (leading, trailing) soft-clipped bases
/// inside `read_bases[r]` (all zero when the reads were hard-clipped beforehand), `cfg` the engine's parameters.
#[allow(clippy::too_many_arguments)]
pub fn region_compute(
    cfg: &phmm_engine_config,
    haplotypes: &[&[u8]],
    haplotype_cigars: &[Vec<u32>],
    alignment_start_hap_wrt_ref: &[u32],
    reference_haplotype: usize,
    reference_start: u64,
    priorities: &[i32],
    read_bases: &[&[u8]],
    read_quals: &[&[u8]],
    ins_quals: &[Vec<u8>],
    del_quals: &[Vec<u8>],
    mapq: &[u8],
    soft_clips: &[(u32, u32)],
    original_cigars: &[Vec<u32>],
) -> RegionOutput {
    let n_reads = read_bases.len();
    let n_haps = haplotypes.len();
    assert!(
        read_quals.len() == n_reads && ins_quals.len() == n_reads && del_quals.len() == n_reads && mapq.len() == n_reads
            && soft_clips.len() == n_reads && original_cigars.len() == n_reads,
        "one entry per read"
    );
    assert!(haplotype_cigars.len() == n_haps && alignment_start_hap_wrt_ref.len() == n_haps && priorities.len() == n_haps, "one entry per haplotype");
    let mut read_off: Vec<u32> = vec![0];
    let (mut bases, mut quals, mut ins, mut del) = (Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for r in 0..n_reads {
        let n = read_bases[r].len();
        assert!(read_quals[r].len() == n && ins_quals[r].len() == n && del_quals[r].len() == n, "Read bases and read quals aren't the same size");
        bases.extend_from_slice(read_bases[r]);
        quals.extend_from_slice(read_quals[r]);
        ins.extend_from_slice(&ins_quals[r]);
        del.extend_from_slice(&del_quals[r]);
        read_off.push(bases.len() as u32);
    }
    let mut hap_off: Vec<u32> = vec![0];
    let mut haps: Vec<u8> = Vec::new();
    for h in haplotypes {
        haps.extend_from_slice(h);
        hap_off.push(haps.len() as u32);
    }
    let flatten_u32 = |parts: &[Vec<u32>]| {
        let mut off: Vec<u32> = vec![0];
        let mut all: Vec<u32> = Vec::new();
        for p in parts {
            all.extend_from_slice(p);
            off.push(all.len() as u32);
        }
        (off, all)
    };
    let (hap_cigar_off, hap_cigar) = flatten_u32(haplotype_cigars);
    let (orig_cigar_off, orig_cigar) = flatten_u32(original_cigars);
    let clips: Vec<u32> = soft_clips.iter().flat_map(|c| [c.0, c.1]).collect();
    let any_clip = clips.iter().any(|c| *c != 0);
    let region_read_off = [0u32, n_reads as u32];
    let region_hap_off = [0u32, n_haps as u32];
    let out_off = [0u64, (n_reads * n_haps) as u64];
    let region_ref_hap = [reference_haplotype as i32];
    let region_reference_start = [reference_start];
    let rcfg = phmm_realign_config {
        // ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, OverhangStrategy::SoftClip (src/reads/alignment_utils.rs:52-58)
        sw_parameters: phmm_sw_parameters { match_value: 10, mismatch_penalty: -15, gap_open_penalty: -30, gap_extend_penalty: -5 },
        overhang_strategy: PHMM_SW_SOFTCLIP,
        // the caller returns before realigning when one allele is left (haplotype_caller_engine.rs:1339-1345)
        flags: PHMM_REGION_SKIP_SINGLE_ALLELE,
        informative_threshold: 0.2, // LOG_10_INFORMATIVE_THRESHOLD (src/model/allele_likelihoods.rs:17)
    };
    let mut likelihoods = vec![0.0f64; n_reads * n_haps];
    let mut keep = vec![0u8; n_reads];
    let mut best = vec![0i32; n_reads];
    let mut likelihood = vec![0.0f64; n_reads];
    let mut confidence = vec![0.0f64; n_reads];
    let mut n_out = vec![0u32; n_reads];
    let mut pos = vec![0i64; n_reads];
    let mut status = vec![0i32; n_reads];
    let mut capacity = vec![16u64; n_reads];
    let mut out_cigar_off = vec![0u64; n_reads + 1];
    let mut out_cigar: Vec<u32> = Vec::new();
    let h = shared_engine();
    for attempt in 0..2 {
        for r in 0..n_reads {
            out_cigar_off[r + 1] = out_cigar_off[r] + capacity[r];
        }
        out_cigar = vec![0u32; out_cigar_off[n_reads] as usize];
        let mut ticket = 0u64;
        let mut rc = unsafe {
            phmm_region_submit(
                h,
                cfg,
                &rcfg,
                1,
                region_read_off.as_ptr(),
                region_hap_off.as_ptr(),
                read_off.as_ptr(),
                bases.as_ptr(),
                quals.as_ptr(),
                ins.as_ptr(),
                del.as_ptr(),
                mapq.as_ptr(),
                if any_clip { clips.as_ptr() } else { std::ptr::null() },
                hap_off.as_ptr(),
                haps.as_ptr(),
                region_ref_hap.as_ptr(),
                out_off.as_ptr(),
                priorities.as_ptr(),
                region_reference_start.as_ptr(),
                hap_cigar_off.as_ptr(),
                hap_cigar.as_ptr(),
                alignment_start_hap_wrt_ref.as_ptr(),
                orig_cigar_off.as_ptr(),
                orig_cigar.as_ptr(),
                out_cigar_off.as_ptr(),
                likelihoods.as_mut_ptr(),
                keep.as_mut_ptr(),
                best.as_mut_ptr(),
                likelihood.as_mut_ptr(),
                confidence.as_mut_ptr(),
                out_cigar.as_mut_ptr(),
                n_out.as_mut_ptr(),
                pos.as_mut_ptr(),
                status.as_mut_ptr(),
                &mut ticket,
            )
        };
        if rc == PHMM_OK {
            // every array above stays alive and untouched until the ticket has been waited for
            rc = unsafe { phmm_wait(h, ticket) };
        }
        if rc == PHMM_ERR_CIGAR_CAPACITY && attempt == 0 {
            for r in 0..n_reads {
                capacity[r] = capacity[r].max(n_out[r] as u64);
            }
            continue;
        }
        if rc != PHMM_OK {
            // the scalar arm asserts the same conditions (argument sizes, result <= 0)
            panic!("HIP region pipeline failed ({}): {}", rc, last_error(h));
        }
        break;
    }
    let reads = (0..n_reads)
        .map(|r| {
            if status[r] < 0 {
                // the reference panics on this read (builder error, read past the end of the reference, ...)
                panic!("Failed to realign read {} (status {})", r, status[r]);
            }
            let start = out_cigar_off[r] as usize;
            RealignedRead {
                allele_index: if best[r] >= 0 { Some(best[r] as usize) } else { None },
                likelihood: likelihood[r],
                confidence: confidence[r],
                realigned: status[r] == PHMM_PROJECT_REALIGNED,
                pos: pos[r],
                cigar: out_cigar[start..start + n_out[r] as usize].to_vec(),
            }
        })
        .collect();
    RegionOutput { likelihoods, keep: keep.iter().map(|k| *k != 0).collect(), reads }
}

/// The realignment half of `region_compute`, carried from `compute_read_likelihoods` to
/// `realign_reads_to_their_best_haplotype`: both run back to back on the same rayon worker
/// (src/haplotype/haplotype_caller_engine.rs:1311-1357), so the token is thread-local, and it names the evidence it
/// belongs to (surviving reads per sample, number of alleles) so that a stale one is never used.
pub struct RegionRealignment {
    pub evidence_counts: Vec<usize>,
    pub n_alleles: usize,
    /// per sample, per surviving read in evidence order
    pub reads: Vec<Vec<RealignedRead>>,
}

thread_local! {
    static REALIGNMENT: RefCell<Option<RegionRealignment>> = RefCell::new(None);
}

pub fn stash_realignment(r: RegionRealignment) {
    REALIGNMENT.with(|cell| *cell.borrow_mut() = Some(r));
}

pub fn clear_realignment() {
    REALIGNMENT.with(|cell| *cell.borrow_mut() = None);
}

/// The stashed result if it belongs to exactly this evidence; consumed either way.
pub fn take_realignment(evidence_counts: &[usize], n_alleles: usize) -> Option<Vec<Vec<RealignedRead>>> {
    REALIGNMENT.with(|cell| match cell.borrow_mut().take() {
        Some(r) if r.evidence_counts.as_slice() == evidence_counts && r.n_alleles == n_alleles => Some(r.reads),
        _ => None,
    })
}

/// `CigarUtils::calculate_cigar` (src/reads/cigar_utils.rs:358-457) for a batch of (reference, haplotype) pairs under
/// one parameter set and one overhang strategy (`PHMM_SW_*`): `None` where the reference returns `None`
/// (`is_s_w_failure`), BAM-encoded elements otherwise.  Panics where the reference panics.
pub fn calculate_cigars(pairs: &[(&[u8], &[u8])], parameters: (i32, i32, i32, i32), overhang_strategy: i32) -> Vec<Option<Vec<u32>>> {
    let n = pairs.len();
    let mut ref_off: Vec<u32> = vec![0];
    let mut alt_off: Vec<u32> = vec![0];
    let (mut refs, mut alts): (Vec<u8>, Vec<u8>) = (Vec::new(), Vec::new());
    for (r, a) in pairs {
        refs.extend_from_slice(r);
        alts.extend_from_slice(a);
        ref_off.push(refs.len() as u32);
        alt_off.push(alts.len() as u32);
    }
    let params = phmm_sw_parameters { match_value: parameters.0, mismatch_penalty: parameters.1, gap_open_penalty: parameters.2, gap_extend_penalty: parameters.3 };
    let mut capacity = vec![16u64; n];
    let mut cigar_off = vec![0u64; n + 1];
    let mut cigar: Vec<u32> = Vec::new();
    let mut n_cigar = vec![0u32; n];
    let mut status = vec![0i32; n];
    with_engine(|h| {
        for attempt in 0..2 {
            for a in 0..n {
                cigar_off[a + 1] = cigar_off[a] + capacity[a];
            }
            cigar = vec![0u32; cigar_off[n] as usize];
            let rc = unsafe {
                phmm_calculate_cigar(
                    h,
                    n as u32,
                    ref_off.as_ptr(),
                    refs.as_ptr(),
                    alt_off.as_ptr(),
                    alts.as_ptr(),
                    &params,
                    overhang_strategy,
                    cigar_off.as_ptr(),
                    cigar.as_mut_ptr(),
                    n_cigar.as_mut_ptr(),
                    status.as_mut_ptr(),
                )
            };
            if rc == PHMM_ERR_CIGAR_CAPACITY && attempt == 0 {
                for a in 0..n {
                    capacity[a] = capacity[a].max(n_cigar[a] as u64);
                }
                continue;
            }
            if rc != PHMM_OK {
                panic!("HIP calculate_cigar failed ({}): {}", rc, last_error(h));
            }
            break;
        }
    });
    (0..n)
        .map(|a| {
            if status[a] < 0 {
                panic!("calculate_cigar: the alignment of pair {} is one the reference panics on (status {})", a, status[a]);
            }
            if status[a] == 1 {
                None
            } else {
                let start = cigar_off[a] as usize;
                Some(cigar[start..start + n_cigar[a] as usize].to_vec())
            }
        })
        .collect()
}
