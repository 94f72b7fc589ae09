//! Safe side of the MI355X PairHMM binding: flattens the data `PairHMM::compute_likelihoods` already holds into the
//! struct-of-arrays `phmm_compute` takes (include/phmm.h) and turns a non-zero status into the panic the rest of
//! the PairHMM path uses for violated preconditions.
//!
//! One engine handle per rayon worker: Lorikeet clones its likelihood engine per region task
//! (src/assembly/assembly_region_walker.rs) and calls the PairHMM synchronously from every worker, and a handle must
//! not be shared between threads.  Workers are spread over the visible devices round-robin.
use std::cell::RefCell;
use std::ffi::CStr;

use crate::pair_hmm::hip_ffi::*;

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
            let h = unsafe { phmm_create(device, 0) };
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
