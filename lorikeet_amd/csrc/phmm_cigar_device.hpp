// Device-side pieces of the projection of read -> haplotype alignments onto the reference (phmm_cigar_kernels.hip; the
// resident region server's tasks, phmm_server_kernels.hip): the CIGAR builders, project_read, pick_read.  Files that include
// this are compiled with -ffp-contract=off (pick_read carries the post-step).
#pragma once
#include "phmm_cigar_internal.hpp"
#include "phmm_post_device.hpp"

namespace phmm {

namespace cigdev {

enum : int { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };
enum : int { LEFT_HARD, LEFT_SOFT, MIDDLE, RIGHT_SOFT, RIGHT_HARD };

__device__ __forceinline__ uint32_t len_of(uint32_t e) { return e >> 4; }
__device__ __forceinline__ int op_of(uint32_t e) { return (int)(e & 15u); }
__device__ __forceinline__ uint32_t elem(int op, uint32_t len) { return (len << 4) | (uint32_t)op; }
// cigar_utils.rs:105-127, :459-464, :536-541, :661-666, as bit sets over the operator
__device__ __forceinline__ bool on_read(int op) { return (0x193u >> op) & 1u; }    // M I S = X
__device__ __forceinline__ bool on_ref(int op) { return (0x18du >> op) & 1u; }     // M D N = X
__device__ __forceinline__ bool clipping(int op) { return op == OP_S || op == OP_H; }
__device__ __forceinline__ bool alignment_op(int op) { return (0x181u >> op) & 1u; }  // M = X
__device__ __forceinline__ uint32_t read_len_of(uint32_t e) { return on_read(op_of(e)) ? len_of(e) : 0; }
__device__ __forceinline__ uint32_t ref_len_of(uint32_t e) { return on_ref(op_of(e)) ? len_of(e) : 0; }

// CigarBuilder (cigar_builder.rs:30-372); `last` is the operator of last_operator, -1 = None
struct Builder {
    uint32_t *el;
    uint32_t n, cap;
    int last, section;
    bool strip;  // remove_deletions_at_ends
    uint32_t leading_removed, trailing_removed;
    int error;

    __device__ void init(uint32_t *storage, uint32_t capacity, bool remove_deletions_at_ends) {
        el = storage;
        n = 0;
        cap = capacity;
        last = -1;
        section = LEFT_HARD;
        strip = remove_deletions_at_ends;
        leading_removed = trailing_removed = 0;
        error = CIGAR_OK;
    }
    __device__ bool deletion_then_insertion() const { return last == OP_I && n > 1 && op_of(el[n - 2]) == OP_D; }  // :201-220
    __device__ int append(uint32_t e) {
        if (n >= cap) return error = CIGAR_ERR_WORKSPACE;
        el[n++] = e;
        return CIGAR_OK;
    }
    __device__ int add(uint32_t e) {  // :59-186
        const int op = op_of(e);
        if (!len_of(e)) return CIGAR_OK;
        if (strip && op == OP_D &&  // a deletion before anything aligned goes (:61-86)
            (last < 0 || clipping(last) || (last == OP_I && (n == 1 || clipping(op_of(el[n - 2])))))) {
            leading_removed += len_of(e);
            return CIGAR_OK;
        }
        // [hard clip] [soft clip] aligned [soft clip] [hard clip] (:223-273)
        if (op == OP_H) {
            if (section == LEFT_SOFT || section == MIDDLE || section == RIGHT_SOFT) section = RIGHT_HARD;
        } else if (op == OP_S) {
            if (section == RIGHT_HARD) return error = CIGAR_ERR_ORDER;
            section = section == LEFT_HARD ? LEFT_SOFT : section == MIDDLE ? RIGHT_SOFT : section;
        } else {
            if (section == RIGHT_SOFT || section == RIGHT_HARD) return error = CIGAR_ERR_ORDER;
            if (section == LEFT_HARD || section == LEFT_SOFT) section = MIDDLE;
        }
        if (last == op) {  // consecutive elements of one type merge (:93-97)
            if (op_of(el[n - 1]) == op) el[n - 1] = elem(op, len_of(el[n - 1]) + len_of(e));
            return CIGAR_OK;
        }
        if (last >= 0 && clipping(op)) {  // clipping starts on the right: a deletion in front of it goes (:105-131)
            if (strip && !on_read(last) && !clipping(last)) {
                trailing_removed += len_of(el[n - 1]);
                el[n - 1] = e;
                last = op;
                return CIGAR_OK;
            }
            if (strip && deletion_then_insertion()) {  // (last_operator stays the insertion, as in the reference)
                trailing_removed += len_of(el[n - 2]);
                el[n - 2] = el[n - 1];
                el[n - 1] = e;
                return CIGAR_OK;
            }
        } else if (last == OP_I && op == OP_D) {  // deletions move to the left of an adjacent insertion (:133-170)
            if (n > 1 && op_of(el[n - 2]) == OP_D) {
                el[n - 2] = elem(OP_D, len_of(el[n - 2]) + len_of(e));
                return CIGAR_OK;
            }
            if (n >= cap) return error = CIGAR_ERR_WORKSPACE;
            el[n] = el[n - 1];
            el[n - 1] = e;
            ++n;
            return CIGAR_OK;
        }
        last = op;
        return append(e);
    }
    // make(false) (:275-324); *trailing = get_trailing_deletion_bases_removed()
    __device__ int make(uint32_t *trailing = nullptr) {
        if (error != CIGAR_OK) return error;
        if (section == LEFT_SOFT && n && op_of(el[0]) == OP_S) return CIGAR_ERR_SOFT_CLIPPED;
        uint32_t in_make = 0;
        if (strip) {
            if (last < 0) return CIGAR_ERR_NONE;
            if (last == OP_D) {
                in_make = len_of(el[n - 1]);
                --n;
            } else if (deletion_then_insertion()) {
                in_make = len_of(el[n - 2]);
                el[n - 2] = el[n - 1];
                --n;
            }
        }
        if (!n) return CIGAR_ERR_EMPTY;
        if (trailing) *trailing = trailing_removed + in_make;
        return CIGAR_OK;
    }
};

__device__ __forceinline__ int32_t range_len(int32_t start, int32_t end) { return end > start ? end - start : 0; }  // Range<i32>::len()

// trim_cigar(cigar, start, end, by_reference = false).make_and_record_deletions_removed_result() (alignment_utils.rs:334-386)
__device__ int trim_by_bases(Builder &T, const uint32_t *src, uint32_t n, uint64_t start, uint64_t end, uint32_t *leading, uint32_t *trailing) {
    if (end < start) return CIGAR_ERR_PANIC;  // "End position cannot be before start position"
    uint64_t e_start, e_end = 0;
    for (uint32_t i = 0; i < n; ++i) {
        e_start = e_end;
        e_end = e_start + read_len_of(src[i]);
        if (e_end < start || (e_end == start && e_start < start)) continue;  // zero-length elements at both ends stay
        if (e_start > end && e_end > end + 1) break;
        const uint64_t overlap = e_end == e_start ? (uint64_t)len_of(src[i]) : (end + 1 < e_end ? end + 1 : e_end) - (start > e_start ? start : e_start);
        const int st = T.add(elem(op_of(src[i]), (uint32_t)overlap));
        if (st != CIGAR_OK) return st;
    }
    if (e_end < end) return CIGAR_ERR_PANIC;  // "Cigar elements don't reach end position (inclusive)"
    const int st = T.make(trailing);
    if (st == CIGAR_OK && leading) *leading = T.leading_removed;
    return st;
}

// left_align_indels(cigar, ref, read, read_start) (alignment_utils.rs:425-566, normalize_alleles :585-640) into T;
// `rtl` holds result_right_to_left.  The source may be T's own storage's neighbour, never T itself.
__device__ int left_align(Builder &T, uint32_t *t_storage, uint32_t capacity, uint32_t *rtl, const uint32_t *src, uint32_t n,
                          const uint8_t *ref_seq, uint32_t ref_seq_len, const uint8_t *read, uint32_t read_len, uint32_t read_start,
                          uint32_t *leading, uint32_t *trailing) {
    int last_indel = -1;
    for (uint32_t i = 0; i < n; ++i)
        if (op_of(src[i]) == OP_D || op_of(src[i]) == OP_I) last_indel = (int)i;
    if (last_indel < 0) {  // unchanged (:431-433)
        if (n > capacity) return CIGAR_ERR_WORKSPACE;
        T.init(t_storage, capacity, true);
        for (uint32_t i = 0; i < n; ++i) T.el[i] = src[i];
        T.n = n;
        *leading = *trailing = 0;
        return CIGAR_OK;
    }
    uint64_t needed = read_start, ref_length = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if ((int)i <= last_indel) needed += ref_len_of(src[i]);
        ref_length += ref_len_of(src[i]);
    }
    if (needed > ref_seq_len) return CIGAR_ERR_PANIC;  // "Read goes past end of reference"
    uint32_t n_rtl = 0;
    bool ok = true, full = false;  // ok: cleared where the reference would index outside a sequence (or panic otherwise)
    auto emit = [&](uint32_t e) {
        if (n_rtl < capacity) rtl[n_rtl++] = e;
        else ok = false, full = true;
    };
    auto ref_at = [&](int32_t i) -> int {
        if (i < 0 || (uint32_t)i >= ref_seq_len) return ok = false, 256;
        return ref_seq[i];
    };
    auto read_at = [&](int32_t i) -> int {
        if (i < 0 || (uint32_t)i >= read_len) return ok = false, 257;
        return read[i];
    };
    int32_t ref_s = (int32_t)(read_start + ref_length), ref_e = ref_s;  // the indel's range on the reference ...
    int32_t rd_s = (int32_t)read_len, rd_e = rd_s;                       // ... and on the read
    for (int k = (int)n - 1; k >= 0 && ok; --k) {
        const uint32_t e = src[k];
        const int op = op_of(e);
        const int32_t on_r = (int32_t)read_len_of(e), on_f = (int32_t)ref_len_of(e);
        if (op == OP_D || op == OP_I) {  // accumulate; shifted when an alignment block or the start is reached
            ref_s -= on_f;
            rd_s -= on_r;
        } else if (range_len(ref_s, ref_e) == 0 && range_len(rd_s, rd_e) == 0) {
            ref_s -= on_f, ref_e -= on_f, rd_s -= on_r, rd_e -= on_r;
            emit(e);
        } else {
            // normalize_alleles({reference, read}, ranges, max_shift, trim = true) (:585-640)
            const uint32_t max_shift = alignment_op(op) ? len_of(e) : 0;
            if (max_shift > (uint32_t)ref_s || max_shift > (uint32_t)rd_s) ok = false;  // "maxShift goes past the start of a sequence"
            int32_t start_shift = 0, end_shift = 0;
            int32_t min_size = min(range_len(ref_s, ref_e), range_len(rd_s, rd_e));
            while (ok && min_size > 0 && ref_at(ref_e - 1) == read_at(rd_e - 1) && ok) {  // shared bases at the end
                --ref_e, --rd_e, --min_size, ++end_shift;
            }
            while (ok && min_size > 0 && ref_at(ref_s) == read_at(rd_s) && ok) {          // ... and at the start
                ++ref_s, ++rd_s, --min_size, --start_shift;
            }
            while (ok && start_shift < (int32_t)max_shift && ref_at(ref_s - 1) == read_at(rd_s - 1) &&
                   ref_at(ref_e - 1) == read_at(rd_e - 1) && ok) {                         // shift left
                --ref_s, --ref_e, --rd_s, --rd_e, ++start_shift, ++end_shift;
            }
            if (!ok) break;
            emit(elem(OP_M, (uint32_t)end_shift));  // new matches on the right of the shifted indel
            const bool emit_indel = k == 0 || start_shift < (int32_t)max_shift || !alignment_op(op);
            const int32_t new_match_left = start_shift < 0 ? -start_shift : 0;
            const int32_t remaining_left = start_shift < 0 ? (int32_t)len_of(e) : (int32_t)len_of(e) - start_shift;
            if (emit_indel) {
                emit(elem(OP_D, (uint32_t)range_len(ref_s, ref_e)));
                emit(elem(OP_I, (uint32_t)range_len(rd_s, rd_e)));
                ref_e -= range_len(ref_s, ref_e);
                rd_e -= range_len(rd_s, rd_e);
                const int32_t d_ref = new_match_left + (on_ref(op) ? remaining_left : 0);
                const int32_t d_read = new_match_left + (on_read(op) ? remaining_left : 0);
                ref_s -= d_ref, ref_e -= d_ref, rd_s -= d_read, rd_e -= d_read;
            }
            emit(elem(OP_M, (uint32_t)new_match_left));
            if (remaining_left < 0) ok = false;
            emit(elem(op, (uint32_t)max(remaining_left, 0)));
        }
    }
    if (ok) {
        emit(elem(OP_D, (uint32_t)range_len(ref_s, ref_e)));
        emit(elem(OP_I, (uint32_t)range_len(rd_s, rd_e)));
    }
    if (!ok) return full ? CIGAR_ERR_WORKSPACE : CIGAR_ERR_PANIC;
    if (rd_s != 0) return CIGAR_ERR_PANIC;  // "Given cigar does not account for all bases of the read"
    T.init(t_storage, capacity, true);
    for (uint32_t i = n_rtl; i-- > 0;)
        if (T.add(rtl[i]) != CIGAR_OK) break;
    if (T.error != CIGAR_OK) return T.error;
    const int st = T.make(trailing);
    if (st == CIGAR_OK) *leading = T.leading_removed;
    return st;
}


// read r (one lane): `ws` = its four builders of p.capacity elements each
__device__ __forceinline__ void project_read(const ProjectParams &p, const uint32_t r, uint32_t *ws) {
    int status = CIGAR_UNCHANGED;
    int64_t new_pos = 0;
    uint32_t n_out = 0;
    Builder A, B, T;
    uint32_t *rtl = ws + 3 * (size_t)p.capacity;

    // the region of read r: the last g with region_read_off[g] <= r
    uint32_t lo = 0, hi = p.n_regions;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (p.region_read_off[mid] <= r) lo = mid;
        else hi = mid;
    }
    const uint32_t g = lo;
    // (phmm_region_compute keeps the best alleles in the caller's pinned memory and the reads' reference index on the device)
    const int32_t best = p.best_allele ? p.best_allele[r]
                         : p.ref_index[r] == SW_NO_REFERENCE ? -1 : (int32_t)(p.ref_index[r] - p.region_hap_off[g]);
    // where the aligner left this read's alignment: its own slot, or (every read against every haplotype, SwParams::pair_stride)
    // the slot of the read's best haplotype
    const uint64_t sw_at = p.sw_pair_stride ? (uint64_t)r * p.sw_pair_stride + (uint32_t)(best < 0 ? 0 : best) : (uint64_t)r;
    const int32_t sw_offset = p.sw_offset[sw_at];
    const int32_t ref_in_region = p.region_ref_hap[g];
    uint32_t *out = p.out_cigar + p.out_cigar_off[r];
    const uint64_t out_cap = p.out_cigar_off[r + 1] - p.out_cigar_off[r];

#define CHECK(x)                                             \
    do {                                                     \
        const int st_ = (x);                                 \
        if (st_ != CIGAR_OK) {                               \
            status = st_ < 0 ? st_ : CIGAR_ERR_PANIC;        \
            goto done;                                       \
        }                                                    \
    } while (0)

    if (best < 0 || sw_offset == -1) goto done;  // no best allele / "sw can fail ... just don't realign the read" (:60-63)
    if (p.ref_index && p.ref_index[r] == SW_NO_REFERENCE) goto done;  // not aligned (a region with a single allele is not realigned)
    if (sw_offset < 0 || ref_in_region < 0 || (uint32_t)best >= p.region_hap_off[g + 1] - p.region_hap_off[g]) {
        status = CIGAR_ERR_PANIC;
        goto done;
    }
    {
        const uint32_t hp = p.region_hap_off[g] + (uint32_t)best, hr = p.region_hap_off[g] + (uint32_t)ref_in_region;
        const uint8_t *ref_seq = p.hap_bases + p.hap_off[hr];
        const uint32_t ref_seq_len = p.hap_off[hr + 1] - p.hap_off[hr];
        const uint32_t clip_l = p.read_clip ? p.read_clip[2 * r] : 0u, clip_r = p.read_clip ? p.read_clip[2 * r + 1] : 0u;
        const uint8_t *read = p.read_bases + p.read_off[r] + clip_l;  // the read minus its soft clips (:47-50)
        const uint32_t read_len = p.read_off[r + 1] - p.read_off[r] - clip_l - clip_r;

        const uint32_t *sw = p.sw_cigar + (p.sw_cigar_off ? p.sw_cigar_off[r] : sw_at * p.sw_cigar_slot);
        const uint32_t n_sw = p.n_sw_cigar[sw_at];
        const uint32_t *hc = p.hap_cigar + p.hap_cigar_off[hp];
        const uint32_t nhc = p.hap_cigar_off[hp + 1] - p.hap_cigar_off[hp];
        uint64_t start_on_reference = 0;
        uint32_t leading_removed = 0, trailing_removed = 0;
        // The usual read: aligned to its haplotype without a gap or a clip (one M element over the whole read), the haplotype
        // without a gap against the reference (one M element).  Every step below then returns what went in -- the padded cigar
        // trimmed to the read is one M run, the two cigars applied to each other are M x read length, nothing to left-align --
        // and the start on the reference is the haplotype's plus the alignment offset.  A lane of its own is ~1 500 dependent
        // instructions and LDS operations for the general case (8 us of every region call's tail); this is a dozen.
        bool plain = false;
        if (n_sw == 1 && nhc == 1) {
            const uint32_t e = sw[0], hce = hc[0];
            plain = op_of(e) == OP_M && read_len > 0 && len_of(e) == read_len && op_of(hce) == OP_M && len_of(hce) > 0 &&
                    (uint64_t)(uint32_t)sw_offset + read_len <= (uint64_t)len_of(hce) + 1000u;
        }
        if (plain) {
            T.init(ws + 2 * (size_t)p.capacity, p.capacity, true);
            if (p.capacity < 1) CHECK(CIGAR_ERR_WORKSPACE);
            T.el[0] = elem(OP_M, read_len);
            T.n = 1;
            start_on_reference = p.region_reference_start[g] + p.hap_start_wrt_ref[hp] + (uint32_t)sw_offset;
        } else {
        // :65-72 the alignment's cigar through a builder
        A.init(ws, p.capacity, true);
        {
            for (uint32_t i = 0; i < n_sw; ++i)
                if (A.add(sw[i]) != CIGAR_OK) break;
            CHECK(A.error);
            CHECK(A.make());
        }
        // :84-100 the haplotype's cigar, padded with 1000M, and the read's start on the reference
        B.init(ws + p.capacity, p.capacity, true);
        {
            for (uint32_t i = 0; i < nhc; ++i)
                if (B.add(hc[i]) != CIGAR_OK) break;
            CHECK(B.error);
            CHECK(B.add(elem(OP_M, 1000)));
            CHECK(B.make());
        }
        uint32_t start_on_ref_hap = 0;
        if (sw_offset != 0) {  // :283-311
            uint32_t ref_used = 0, hap_used = 0;
            bool reached = false;
            for (uint32_t i = 0; i < B.n && !reached; ++i) {
                ref_used += ref_len_of(B.el[i]);
                hap_used += read_len_of(B.el[i]);
                if (hap_used >= (uint32_t)sw_offset) {
                    const uint32_t excess = on_ref(op_of(B.el[i])) ? hap_used - (uint32_t)sw_offset : 0;
                    start_on_ref_hap = ref_used >= excess ? ref_used - excess : 0;
                    reached = true;
                }
            }
            if (!reached) CHECK(CIGAR_ERR_PANIC);  // "Cigar doesn't reach the read start"
        }
        start_on_reference = p.region_reference_start[g] + p.hap_start_wrt_ref[hp] + start_on_ref_hap;

        // :107-113 trim_cigar_by_bases(padded cigar, offset, its read length - 1): elements behind the read's end stay
        T.init(ws + 2 * (size_t)p.capacity, p.capacity, true);
        {
            uint32_t padded_len = 0;
            for (uint32_t i = 0; i < B.n; ++i) padded_len += read_len_of(B.el[i]);
            CHECK(trim_by_bases(T, B.el, B.n, (uint32_t)sw_offset, (uint64_t)padded_len - 1, nullptr, nullptr));
        }

        // :115 apply_cigar_to_cigar(read -> haplotype, haplotype -> reference) into B, by runs
        B.init(ws + p.capacity, p.capacity, true);
        {
            uint32_t i12 = 0, i23 = 0, rem12 = A.n ? len_of(A.el[0]) : 0, rem23 = T.n ? len_of(T.el[0]) : 0;
            while (i12 < A.n && i23 < T.n) {
                const int a = op_of(A.el[i12]), c = op_of(T.el[i23]);
                const bool a_m = alignment_op(a), a_i = a == OP_I || a == OP_S, a_d = a == OP_D;
                const bool c_m = alignment_op(c), c_i = c == OP_I || c == OP_S, c_d = c == OP_D;
                if (!(a_m || a_i || a_d) || !(c_m || c_i || c_d)) CHECK(CIGAR_ERR_PANIC);
                int op13;                  // CigarPairTransform::new (:974-1049)
                bool adv12, adv23;
                if (a_i) {
                    op13 = OP_I, adv12 = true, adv23 = false;
                } else if (c_d) {
                    op13 = OP_D, adv12 = false, adv23 = true;
                } else if (a_m) {
                    op13 = c_m ? OP_M : OP_I, adv12 = adv23 = true;
                } else {
                    op13 = c_m ? OP_D : -1, adv12 = adv23 = true;
                }
                const uint32_t run = adv12 && adv23 ? min(rem12, rem23) : adv12 ? rem12 : rem23;
                if (op13 >= 0) CHECK(B.add(elem(op13, run)) == CIGAR_OK ? CIGAR_OK : (B.error == CIGAR_ERR_WORKSPACE ? CIGAR_ERR_WORKSPACE : CIGAR_ERR_PANIC));
                if (adv12 && !(rem12 -= run)) rem12 = ++i12 < A.n ? len_of(A.el[i12]) : 0;
                if (adv23 && !(rem23 -= run)) rem23 = ++i23 < T.n ? len_of(T.el[i23]) : 0;
            }
            const int st = B.make();
            CHECK(st == CIGAR_OK || st == CIGAR_ERR_WORKSPACE ? st : CIGAR_ERR_PANIC);
        }

        // :116-122 left_align_indels(B, reference haplotype, read, start on the reference haplotype) into T
        CHECK(left_align(T, ws + 2 * (size_t)p.capacity, p.capacity, rtl, B.el, B.n, ref_seq, ref_seq_len, read, read_len, start_on_ref_hap,
                         &leading_removed, &trailing_removed));
        }  // (the general case)

        // :126-130 left-alignment may have moved a deletion to the front of the read and dropped it
        new_pos = (int64_t)(start_on_reference + leading_removed);
        // :151-161 the realigned cigar must cover the read
        uint32_t aligned_len = 0;
        for (uint32_t i = 0; i < T.n; ++i) aligned_len += read_len_of(T.el[i]);
        if (aligned_len != read_len) CHECK(CIGAR_ERR_PANIC);  // (+ soft-clipped bases == original read length)
        // :135-143 the clips of the original cigar go back on (:173-213)
        const uint32_t *oc = p.orig_cigar + p.orig_cigar_off[r];
        const uint32_t n_oc = p.orig_cigar_off[r + 1] - p.orig_cigar_off[r];
        if (!n_oc) CHECK(CIGAR_ERR_PANIC);
        uint32_t first = 0, last = n_oc - 1;
        auto put = [&](uint32_t e) {
            if (n_out < out_cap) out[n_out] = e;
            ++n_out;
        };
        while (clipping(op_of(oc[first])) && first != last) put(oc[first++]);
        for (uint32_t i = 0; i < T.n; ++i) put(T.el[i]);
        uint32_t right = last + 1;
        while (clipping(op_of(oc[last])) && first != last) right = last--;
        for (uint32_t i = right; i < n_oc; ++i) put(oc[i]);
        status = CIGAR_OK;
    }
done:
#undef CHECK
    p.status[r] = status;
    p.new_pos[r] = status == CIGAR_OK ? new_pos : 0;
    p.n_out_cigar[r] = status == CIGAR_OK ? n_out : 0;
    if (status == CIGAR_OK && n_out > out_cap) *p.flags = 1u;
}

// Post-step, best allele and projection of a read in ONE launch (phmm_region_compute, small calls whose alignments were made
// for every haplotype beside the PairHMM kernels): lane r normalises its row of likelihoods, decides keep[r], finds the best
// allele and projects the alignment the aligner left in THAT haplotype's slot -- phmm_post_best_reads and
// phmm_project_kernel, statement for statement, without the launch in between.
// `ws`: the lane's builders (4 x p.capacity words: LDS for small launches, else its share of p.workspace)
__device__ __forceinline__ void pick_read(const PostBestParams &pb, const ProjectParams &p, uint32_t r, uint32_t *ws) {
    const uint32_t g = pb.post.read_region[r];
    const uint32_t nh = pb.post.region_hap_off[g + 1] - pb.post.region_hap_off[g];
    if (nh <= 16) {
        post_best_in_registers<16>(pb, r, g, nh);
    } else {
        const uint8_t keep = post_read(pb.post, r, true);
        if (pb.keep_final) pb.keep_final[r] = keep;
        best_allele_of(pb.best, r, g, keep != 0, !(pb.skip_single_allele && nh == 1));
    }
    if (p.wait_counter) {  // the alignments come from a kernel on another stream: until every block of it has counted itself in
        // The wait is BOUNDED (ProjectParams::wait_ticks of the 100 MHz clock): forward progress of that other kernel rests on
        // the two streams owning different hardware queues, which is how this runtime maps CU-masked streams, not a contract.
        // Out of time (the queues were multiplexed into one, the aligner faulted, its queue is not mapped): the lane raises
        // flags[1], projects nothing and leaves -- the kernel always ends, the host synchronises both streams
        // and runs the call again the chained way (region_one_shot).
        const uint64_t t0 = wall_clock64();
        bool in_time = true;
        while ((int32_t)(__hip_atomic_load(p.wait_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.wait_target) < 0) {
            if (wall_clock64() - t0 > (uint64_t)p.wait_ticks) {
                in_time = false;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (!in_time) {
            p.flags[1] = 1u;  // (a word of its own, a plain store: the block may live in the caller's pinned memory)
            p.status[r] = CIGAR_UNCHANGED;
            p.new_pos[r] = 0;
            p.n_out_cigar[r] = 0;
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    project_read(p, r, ws);
}

}  // namespace cigdev

}  // namespace phmm
