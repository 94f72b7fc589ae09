// gfx950 chained forward kernel: the same register-resident sweep as phmm_forward<L,K>, but a wave keeps
// its 64/L haplotypes and streams MANY reads through them back to back, so the (L-1)-step fill / drain of the
// lane pipeline is paid once per chain instead of once per read.  Compiled once per lanes-per-pair value
// (-DPHMM_CHAIN_L=16|32|64), like phmm_kernels.hip.
//
// The reads of a chain form one stream of rows in an LDS ring:
//     [rows of read 0][SUM][RESET][rows of read 1][SUM][RESET] ...
// and lane l simply works on stream position t - l at step t.  Nothing in the steady state is predicated
// or per-lane conditional; the hand-over between reads happens through the recurrence itself:
//   * the SUM row's dDp (the previous row's M->D coefficient, applied by the consumer) is 0, so the last row's
//     deletion state drops out, and its I^ is I itself (im(R+1) = 1);
//   * the SUM row (mm = bI = pm(R), gI = dd = 1, prior 1) turns M(k) into M(R,k-1) + I(R,k-1) and I(k) into
//     M(R,k) + I(R,k), and lets the D chain -- which already runs left to right through columns AND lanes --
//     accumulate the M's: after the SUM step of the lane that owns the last haplotype column c, D' + M + I of that
//     column is sum_j M[R][j] + I[R][j] in the reference's order (pair_hmm.rs:598-603); that lane takes the
//     log10 and stores the result (columns right of c carry don't-care values until the RESET row);
//   * the RESET row (prior = 0, bI = gI = 0, dd = 1) rebuilds the row-0 state (0, 0, c0) of the next
//     read: M and I^ vanish and the D chain copies the value injected at the group's first lane.
// Rows are produced 64 at a time (one per lane) into a 256-row ring, 64..128 rows ahead of lane 0; the stream
// is padded with CL-1 neutral rows in front and behind, so a lane's slot is just (position & 255).
// D(0,j): every value of the recurrence is linear in the initial D(0,j) = 2^1020 / H (pair_hmm.rs:515-529), so
// the chained kernel starts all haplotypes from the same 2^1010 (which lets the RESET row carry the injected
// value itself) and subtracts log10(H) at the end; the difference to dividing first is one rounding.
//
// Streams.  With 16 lanes per pair a wave has four haplotype slots; a region with 1, 2, 3, 5, ... haplotypes would
// leave slots idle.  So the run of reads of a work item can be split into S = 1, 2 or 4 sub-runs ("streams") that are
// swept side by side: stream s owns 4/S slots (the same haplotypes as the other streams, its own reads), its own
// quarter/half of the ring (RING/S rows), and the producer makes 64/S rows per stream and tick (a tick is then 64/S
// steps), so everything scales and the sweep loop itself is unchanged.
//
// Chains with a haplotype containing 'N' or a read with gcp == 0 / base quality 0 fall back, inside the same wave,
// to the plain per-read sweep (sweep_general): rare, kept for exactness.
#include "phmm_device.hpp"

namespace phmm {

namespace {

#ifndef PHMM_CHAIN_L
#define PHMM_CHAIN_L 16
#endif
constexpr int CL = PHMM_CHAIN_L;     // lanes per pair
#ifndef PHMM_RING
#define PHMM_RING 256
#endif
constexpr int RING = PHMM_RING;      // ring rows (power of two), shared by the streams
// Every stream's rows are followed by a GUARD slot that repeats its row 0, so the sweep fetches the two rows of a
// ping-pong pair from ONE address (the second through the immediate offset of the LDS read): 3 instead of 10 address
// instructions per pair of steps.  Slots: S streams x (RING / S rows + guard), at most RING + 4; then the neutral row.
constexpr int RING_SLOTS = RING + 4 + 1;
constexpr int CHAIN_META = CHAIN_MAX_READS + 8;  // per-read offsets of all streams: n_chain + S entries
constexpr uint32_t X_PAD = 0x100u;   // base code of padding columns (>= H) and read-side code of the SUM row
constexpr uint32_t X_NONE = 0x102u;  // read-side code that matches nothing
constexpr int LEAD = CL - 1;          // neutral rows in front of the stream (lane l starts LEAD - l rows early)

// Left neighbour's value; the group's first lane (no neighbour inside its group) receives `inject`: through the
// DPP `old` operand where the shift has no source lane (row_shr:1 for 16-lane groups, wave_shr:1 for lane 0),
// through a select for lane 32 of two 32-lane groups.
__device__ __forceinline__ double from_left_inject(double v, double inject, bool group_head) {
    constexpr int ctrl = CL == 16 ? 0x111 : 0x138;
    int lo = __builtin_amdgcn_update_dpp(__double2loint(inject), __double2loint(v), ctrl, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(inject), __double2hiint(v), ctrl, 0xf, 0xf, false);
    if constexpr (CL == 32) {
        lo = group_head ? __double2loint(inject) : lo;
        hi = group_head ? __double2hiint(inject) : hi;
    }
    return __hiloint2double(hi, lo);
}

template <int K>
__device__ __forceinline__ void chain_fallback(const ForwardParams &p, const ChainItem &it, RowConst *ring, int lane,
                                               int grp, int l, const HapCols<K> &hc, int H, bool hv, int a, int Nh) {
    // Per read: stage plain or pre-scaled rows linearly (slot 0 neutral, row r at r+1), general sweep, reduce.
    const uint32_t reg = it.region;
    for (uint32_t r = it.read_begin; r < it.read_end; ++r) {
        const uint32_t ro = p.read_off[r];
        const int R = (int)(p.read_off[r + 1] - ro);
        bool z = false;
        for (int row = lane; row < R; row += WAVE) z |= row_blocks_prescale(p, ro + row);
        const bool scaled = __ballot(z) == 0ull;
        const bool staged = R + 1 <= RING;  // longer reads build their rows on the fly (slow, but this path is rare)
        lds_wave_sync();
        if (staged) {
            if (lane == 0) ring[0] = neutral_row();
            for (int row = lane; row < R; row += WAVE) ring[row + 1] = make_row(p, ro, row, R, scaled);
        }
        lds_wave_sync();
        const double scale0 = (scaled && R > 0) ? 1.0 - p.eps[p.gcp[ro]] : 1.0;
        const double fin = (scaled && R > 0) ? 1.0 - p.eps[p.base_q[ro + R - 1]] : 1.0;
        const double c0 = p.initial_condition / (double)H * scale0;
        const bool group_head = (CL == 32) && (lane == 32);
        double s = staged ? sweep_general<CL, K>(LdsView{ring}, R, l, group_head, hc, H, c0, scaled, fin)
                          : sweep_general<CL, K>(GlobalRowView{p, ro, R, scaled}, R, l, group_head, hc, H, c0, scaled, fin);
#pragma unroll
        for (int off = CL / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        if (l == 0 && hv) {
            const double v = log10(s) - p.initial_condition_log10;
            p.out[p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh + a] = v;
            if (const uint32_t sb = status_bits(v)) atomicOr(p.status, sb);
        }
    }
    (void)grp;
}

// The same for a work item split into streams: every 16-lane group walks the reads of ITS stream, rows built on the
// fly from HBM (no staging: the groups of a wave are at different reads).  Slow and rare.
template <int K>
__device__ __forceinline__ void chain_fallback_streams(const ForwardParams &p, uint32_t reg, uint32_t r_begin, int n_mine,
                                                       int n_iter, int lane, int l, const HapCols<K> &hc, int H, bool hv,
                                                       int a, int Nh) {
    const uint64_t group_bits = 0xffffull << (lane & ~15);
    for (int i = 0; i < n_iter; ++i) {  // wave-uniform trip count; groups past their last read idle
        const bool valid = i < n_mine;
        const uint32_t r = r_begin + (uint32_t)i;
        const uint32_t ro = valid ? p.read_off[r] : 0u;
        const int R = valid ? (int)(p.read_off[r + 1] - ro) : 0;
        bool z = false;
        for (int row = l; row < R; row += 16) z |= row_blocks_prescale(p, ro + row);
        const bool scaled = (__ballot(z) & group_bits) == 0ull;
        const double scale0 = (scaled && R > 0) ? 1.0 - p.eps[p.gcp[ro]] : 1.0;
        const double fin = (scaled && R > 0) ? 1.0 - p.eps[p.base_q[ro + R - 1]] : 1.0;
        const double c0 = p.initial_condition / (double)H * scale0;
        double s = sweep_general<16, K>(GlobalRowView{p, ro, R, scaled}, R, l, false, hc, H, c0, scaled, fin);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        if (l == 0 && hv && valid) {
            const double v = log10(s) - p.initial_condition_log10;
            p.out[p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh + a] = v;
            if (const uint32_t sb = status_bits(v)) atomicOr(p.status, sb);
        }
    }
}

}  // namespace

// One work item = one wave: a run of reads of one region against one group of its haplotypes, K columns per lane.
// MODE (phmm_internal.hpp, shared haplotype prefixes): CHAIN_PARK items store M~ / D' of the flagged lanes' last column for
// every stream row (`x`: where), CHAIN_SUFFIX items start at haplotype column x.col0 and take their left neighbour from there.
template <int CLT, int K, int MODE = CHAIN_PLAIN>  // CLT == CL of this compilation unit (keeps the three units' kernel symbols apart)
__device__ __forceinline__ void chain_body(const ForwardParams &p, const ChainItem it, unsigned char *smem, const ChainItemX *x = nullptr,
                                           double *park = nullptr) {
    static_assert(CLT == CL, "one lanes-per-pair value per compilation unit");
    static_assert(MODE == CHAIN_PLAIN || CL == 16, "prefix sharing: 16 lanes per pair");
    const int col0 = MODE == CHAIN_SUFFIX ? (int)x->mask_or_col0 : 0;
    const uint32_t park_rows = MODE == CHAIN_PLAIN ? 0u : 16u * x->park_rows16;
    const uint32_t park_mask = MODE == CHAIN_PARK ? (uint32_t)x->mask_or_col0 : 0u;
    const int lane = threadIdx.x;
    const int grp = lane / CL, l = lane % CL;
    const bool group_head = (CL == 32) && (lane == 32);
    const uint32_t reg = it.region;
    const int n_chain = (int)(it.read_end - it.read_begin);
    const uint32_t h0 = p.region_hap_off[reg];
    const int Nh = (int)(p.region_hap_off[reg + 1] - h0);
    // streams (see the header): S sub-runs of reads side by side, each on G/S haplotype slots
    const int S = (CL == 16) ? (int)it.streams : 1;
    const int GS = (WAVE / CL) / S;             // haplotype slots per stream
    const int sid = grp / GS;                   // stream of this lane's group
    const int a = MODE == CHAIN_PLAIN ? (int)it.quad * GS + grp % GS : (int)x->hap[grp & 3];  // (the sharing kernels name their haplotypes)
    const bool hv = a < Nh;
    const int n_sub = (n_chain + S - 1) / S;    // reads per stream (the last streams may get fewer, or none)
    const int TPS = (RING / 4) / S;             // rows produced per stream and tick == steps per tick (a quarter of the stream's ring)
    const int NM = RING / S - 1;                // ring rows per stream - 1 (mask)
    auto n_of = [&](int s) { return max(0, min(n_sub, n_chain - s * n_sub)); };
    const int n_mine = n_of(sid);
    uint32_t ho = 0;
    int H = 0;
    if (hv) {
        ho = p.hap_off[h0 + a];
        H = (int)(p.hap_off[h0 + a + 1] - ho);
    }
    RowConst *ring = reinterpret_cast<RowConst *>(smem);          // RING_SLOTS records
    uint32_t *roff = reinterpret_cast<uint32_t *>(ring + RING_SLOTS);  // per stream s at s*(n_sub+1): byte offset of each read
    uint32_t *stot = roff + CHAIN_META;                              // [4] rows of each stream

    // ---- haplotype columns: real bases, one EDGE column, then padding -------------------------------
    HapCols<K> hc;
    bool lane_n = false;
#pragma unroll
    for (int w = 0; w < HapCols<K>::W; ++w) hc.y[w] = 0u;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int col = col0 + l * K + k;
        uint32_t y = col < H ? (uint32_t)p.hap_bases[ho + col] : X_PAD;
        const bool is_n = (y == 'N');
        lane_n |= is_n;
        hc.set(k, y);
    }

    // ---- the chain: stream offsets, and whether every read can be pre-scaled ------------------------
    const uint32_t rb = it.read_begin;
    const uint32_t byte0 = p.read_off[rb];
    const uint32_t bytes = p.read_off[it.read_end] - byte0;
    bool z = false;
    for (uint32_t i = lane; i < bytes; i += WAVE) z |= row_blocks_prescale(p, byte0 + i);
    if ((__ballot(z) | __ballot(lane_n)) != 0ull) {  // rare: exact but unchained
        if constexpr (MODE == CHAIN_SUFFIX) {
            // (a suffix cannot be swept alone the general way: its pairs are left to the exact pass -- NaN asks for it)
            for (uint32_t r = rb + (uint32_t)l; r < it.read_end && hv; r += CL)
                p.out[p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh + a] = __longlong_as_double(0x7ff8000000000000ll);
            if (lane == 0) atomicOr(p.status, STATUS_RESCUE);
            return;
        }
        if (S == 1)
            chain_fallback<K>(p, it, ring, lane, grp, l, hc, H, hv, a, Nh);
        else if constexpr (CL == 16)
            chain_fallback_streams<K>(p, reg, rb + (uint32_t)(sid * n_sub), n_mine, n_sub, lane, l, hc, H, hv, a, Nh);
        return;
    }
    {   // lane j describes read j of the run: its stream sj and index ij there; offsets by a wave scan
        const int sj = lane / n_sub, ij = lane % n_sub;
        uint32_t len = lane < n_chain ? p.read_off[rb + lane + 1] - p.read_off[rb + lane] + 2u : 0u;  // + SUM + RESET
        uint32_t incl = len;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += v;
        }
        const uint32_t before = __shfl(incl, max(sj * n_sub - 1, 0), WAVE);  // scan value just before my stream starts
        const int cb = sj * (n_sub + 1);
        if (lane < 4) stot[lane] = 0u;  // rows of each stream (streams without reads stay 0)
        if (lane < n_chain) {
            roff[cb + ij] = p.read_off[rb + lane];
            if (ij + 1 == n_of(sj)) {
                roff[cb + ij + 1] = p.read_off[rb + lane + 1];
                stot[sj] = incl - (sj > 0 ? before : 0u);
            }
        }
        if (lane == 0) ring[RING_SLOTS - 1] = neutral_row();
    }
    lds_wave_sync();
    const int S_max = (int)max(max(stot[0], stot[1]), max(stot[2], stot[3]));  // longest stream (rows)
    const double c_unit = ldexp(1.0, 1010);  // common D(0,j) of every haplotype, see the header

    // ---- row producer: stream positions [P0, P0 + 64) -> ring ----------------------------------------
    // Two-phase producer, one ring row per lane: `issue` starts the (cold) loads of a row's quality bytes, `finish`
    // -- one tick = 64 steps later, when they have long arrived -- looks the (hot, L1/L2-resident) table values
    // up, builds the record and writes it to the ring.  Only the six bytes live in registers in between.
    uint32_t pb_x = 0, pb_q = 0, pb_qp = 0, pb_i = 0, pb_d = 0, pb_dp = 0, pb_g = 0, pb_gn = 0;
    double pb_bM = 0.0, pb_bD = 0.0;  // CHAIN_SUFFIX: the parked column's M~, D' of the row in the making
    int p_Q = 0;
    // producer side of this lane: row (lane % TPS) of stream (lane / TPS) of each tick.  Its position in that stream,
    // (p_lo = read of the stream, p_row = row of that read, counting SUM and RESET), moves on by TPS rows per tick.
    const bool producer = lane < S * TPS;       // RING / 4 lanes build a row per tick (all 64 with the 256-row ring)
    const int ps = producer ? lane / TPS : 0, pj = lane % TPS;
    const int pcb = ps * (n_sub + 1), pn = n_of(ps);
    int p_lo = 0, p_row = pj - LEAD - TPS;  // before the first advance(); rows < 0 are the neutral lead-in
    p_Q = pj - TPS;                         // ring position of that row
    uint32_t p_ro = pn > 0 ? roff[pcb] : 0u;
    int p_R = pn > 0 ? (int)(roff[pcb + 1] - p_ro) : 0;
    auto advance = [&]() {
        p_row += TPS;
        p_Q += TPS;
        while (p_lo < pn && p_row >= p_R + 2) {  // past SUM and RESET of the current read: on to the next one(s)
            p_row -= p_R + 2;
            ++p_lo;
            if (p_lo < pn) {
                p_ro = roff[pcb + p_lo];
                p_R = (int)(roff[pcb + p_lo + 1] - p_ro);
            }
        }
    };
    auto issue = [&]() {  // next position of this lane: start the loads of its bytes
        advance();
        const int row = p_row, R = p_R;
        const uint32_t ro = p_ro;
        if constexpr (MODE == CHAIN_SUFFIX) {
            if (producer && p_lo < pn && row >= 0 && (uint32_t)p_Q < park_rows) {  // every row of the stream, SUM and RESET rows included
                const double *at = park + 2 * ((size_t)x->park_row0 + (size_t)p_Q);
                pb_bM = at[0];
                pb_bD = at[1];
            }
        }
        if (producer && p_lo < pn && row >= 0 && row <= R) {
            pb_qp = row > 0 ? (uint32_t)p.base_q[ro + row - 1] : 0u;  // row == R: the SUM row needs pm(R)
            if (row < R) {
                pb_x = p.read_bases[ro + row];
                pb_q = p.base_q[ro + row];
                pb_i = p.ins_q[ro + row];
                pb_d = p.del_q[ro + row];
                pb_dp = row > 0 ? (uint32_t)p.del_q[ro + row - 1] : 0u;
                pb_g = p.gcp[ro + row];
                pb_gn = row + 1 < R ? (uint32_t)p.gcp[ro + row + 1] : 0u;
            }
        }
    };
    auto finish = [&](int Q0) {  // the issued position = ring positions [Q0, Q0 + TPS) of every stream
        const int Q = Q0 + pj;     // stream position = ring position - LEAD
        const int lo = p_lo, row = p_row, R = p_R;
        RowConst n;
        if (lo < pn && row >= 0) {
            if (row < R) {
                n = make_row_bytes(p, pb_x, pb_q, pb_qp, pb_i, pb_d, pb_dp, pb_g, pb_gn, row == 0, row + 1 >= R, true);
            } else if (row == R) {  // SUM row: M_S(k) = M(R,k-1) + I(R,k-1), I_S(k) = M(R,k) + I(R,k); pad0 = read index in the chain
                n.mm = 1.0 - p.eps[pb_qp]; n.bI = n.mm; n.gI = 1.0; n.dDp = 0.0; n.dd = 1.0; n.pm = 1.0; n.px = 1.0;
                n.x = X_PAD; n.pad0 = (uint32_t)lo; n.pad1 = 0.0;
            } else {                // RESET row; pad1 = D'(0,.) of the next read = 2^1010 * im of its first row
                n.mm = 0.0; n.bI = 0.0; n.gI = 0.0; n.dDp = 0.0; n.dd = 1.0; n.pm = 0.0; n.px = 0.0;
                n.x = X_NONE; n.pad0 = 0;
                n.pad1 = c_unit * (lo + 1 < pn ? 1.0 - p.eps[p.gcp[roff[pcb + lo + 1]]] : 1.0);
            }
            if constexpr (MODE == CHAIN_SUFFIX) {  // the left neighbour of the item's first lanes in this row (pm is not read by the sweep)
                n.pm = pb_bM;
                n.pad1 = pb_bD;
            }
        } else {
            n = neutral_row();
        }
        if (producer) {
            const int slot = ps * (NM + 2) + (Q & NM);
            ring[slot] = n;
            if ((Q & NM) == 0) ring[slot + NM + 1] = n;  // the stream's guard slot
        }
    };
    issue();
    finish(0);
    issue();
    finish(TPS);
    issue();
    finish(2 * TPS);
    lds_wave_sync();
    issue();

    // ---- state ---------------------------------------------------------------------------------------
    // planner guarantees every read has >= 1 base; a stream without reads only ever sees neutral rows
    const double c0 = c_unit * (n_mine > 0 ? 1.0 - p.eps[p.gcp[roff[sid * (n_sub + 1)]]] : 1.0);
    double Mp[K], Ip[K], Dp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Mp[k] = 0.0;
        Ip[k] = 0.0;
        Dp[k] = c0;
    }
    double aM, aI, aD, bM = 0.0, bI = 0.0, bD = c0;
    // After the SUM row the last haplotype column c = H-1 holds I_S = M(R,c)+I(R,c), M_S = M(R,c-1)+I(R,c-1) and
    // D' = the sum over everything further left: the lane that owns it emits the result.
    const int edge_lane = (H > col0 ? H - 1 - col0 : 0) / K, edge_k = (H > col0 ? H - 1 - col0 : 0) % K;
    const bool last_lane = (l == edge_lane);
    const uint32_t sum_code = last_lane ? X_PAD : 0xffffffffu;  // == c.x exactly when this lane has to emit
    const double log10_scale = log10(c_unit) + log10((double)H);  // result = log10(sum) - log10(2^1010 * H)
    auto emit = [&](const RowConst &c) {
        // after the owning lane's SUM step, see above
        {
            if (last_lane && c.x == X_PAD && hv) {
                const uint32_t r = rb + (uint32_t)(sid * n_sub) + c.pad0;
                // the column is a per-lane value: a K-way select.  `ek` is made opaque so that the K compare masks are
                // built here, once per read, instead of living in 2*K SGPRs across the sweep loop
                int ek = edge_k;
                asm volatile("" : "+v"(ek));
                double sum = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (k == ek) sum = (Dp[k] + Mp[k]) + Ip[k];
                const double v = log10(sum) - log10_scale;
                p.out[p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh + a] = v;
                if (const uint32_t sb = status_bits(v)) atomicOr(p.status, sb);  // reference asserts result <= 0 (pair_hmm.rs:478-481)
            }
        }
    };

    // lane l works on ring position q = t + LEAD - l  (stream position t - l); rows outside the stream are neutral
    int q = LEAD - l;
    const int my_ring = sid * (NM + 2);  // first slot of this lane's stream (an index, so the LDS address stays 32-bit math)
    // (LDS byte address of the stream's first slot; q1 = q + 1 is what the loop advances)
    const uint32_t my_rows = (uint32_t)(uintptr_t)(ring + my_ring);  // (the low half of a flat address into LDS is the LDS address)
    RowConst cA = lds_row(ring, my_ring + (q & NM)), cB;
    int q1 = q + 1;
    const int T = (S_max + CL - 1 + 1) & ~1;  // even number of steps; surplus steps run neutral rows
    // CHAIN_PARK: this lane's column of the parking area, indexed by ring position (the lane is at q now, q + 1 next, ...);
    // only the trunk -- slot 0 of the wave -- is parked
    bool park_lane = false;
    double2 *park_at = nullptr;
    if constexpr (MODE == CHAIN_PARK) {
        park_lane = grp == 0 && ((park_mask >> l) & 1u) != 0u && hv && (uint32_t)(q + T) <= park_rows;
        const int b = __popc(park_mask & ((1u << l) - 1u));
        park_at = reinterpret_cast<double2 *>(park) + ((size_t)x->park_row0 + (size_t)b * park_rows + (size_t)q);
    }
    // Outer loop = one producer tick (TPS steps), inner loop = the sweep.  The producer's pending bytes are
    // defined before the inner loop and first used after it, so their loads have a tick to land.
    // The first tick is shortened by a per-block even phase (the ring only gets further ahead), so that the two
    // waves sharing a SIMD do not run their producers -- the one latency-exposed part -- at the same time.
    const int phase = (int)((blockIdx.x * 2654435761u) >> 26) & (TPS - 2);
    for (int t0 = 0, t1 = min(T, TPS - phase), tick = 0; t0 < T; t0 = t1, t1 = min(T, t1 + TPS), ++tick) {
        if (tick >= 1) {  // keep the ring one to two ticks (+ phase) ahead of the first lane
            finish(TPS * tick + 2 * TPS);
            lds_wave_sync();
            issue();
        }
        for (int t = t0; t < t1; t += 2) {
            // rows q + 1 and q + 2 lie next to each other (the guard slot repeats row 0 behind the stream's last row)
            // (one full-rate v_mad_u32_u24: the compiler turns the 24-bit multiply back into a 32-bit v_mul_lo_u32, a
            // quarter-rate instruction, plus an add)
            uint32_t pair_at;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(pair_at) : "v"((uint32_t)(q1 & NM)), "s"((uint32_t)sizeof(RowConst)), "v"(my_rows));
            cB = lds_row_at(pair_at, 0);
            if constexpr (MODE == CHAIN_SUFFIX) {  // the first lane's neighbour is the parked column: M~ and D' of this row from the
                // record, I^ of this row from the row before (the operations the owner of that column performed: mul, then fma)
                aI = from_left_inject(Ip[K - 1], fma(bM, cA.bI, bI * cA.gI), group_head);
                aM = from_left_inject(Mp[K - 1], cA.pm, group_head);
            } else {
                aM = from_left<CL>(Mp[K - 1], group_head);
                aI = from_left<CL>(Ip[K - 1], group_head);
            }
            aD = from_left_inject(Dp[K - 1], cA.pad1, group_head);  // column 0 has D = 0; a RESET row injects the next read's D(0,0)
            row_update<K, ROW_FAST_EXEC>(Mp, Ip, Dp, bM, bI, bD, aM, aD, cA, hc, 1.0);
            if constexpr (MODE == CHAIN_PARK)
                if (park_lane) park_at[0] = make_double2(Mp[K - 1], Dp[K - 1]);
            // a read's SUM row reaches its emitting lane once per read: a wave-uniform test per step, each on the row
            // that was just consumed (testing cB here as well would wait for its LDS load right after issuing it)
            if (__ballot(cA.x == sum_code) != 0ull) emit(cA);
            cA = lds_row_at(pair_at, 1);
            if constexpr (MODE == CHAIN_SUFFIX) {
                bI = from_left_inject(Ip[K - 1], fma(aM, cB.bI, aI * cB.gI), group_head);
                bM = from_left_inject(Mp[K - 1], cB.pm, group_head);
            } else {
                bM = from_left<CL>(Mp[K - 1], group_head);
                bI = from_left<CL>(Ip[K - 1], group_head);
            }
            bD = from_left_inject(Dp[K - 1], cB.pad1, group_head);
            row_update<K, ROW_FAST_EXEC>(Mp, Ip, Dp, aM, aI, aD, bM, bD, cB, hc, 1.0);
            if constexpr (MODE == CHAIN_PARK) {
                if (park_lane) park_at[1] = make_double2(Mp[K - 1], Dp[K - 1]);
                park_at += 2;
            }
            if (__ballot(cB.x == sum_code) != 0ull) emit(cB);
            q1 += 2;
        }
    }
}


// ---- the kernel: every item carries its own K (and stream count), so ONE launch per lanes-per-pair value covers
// every shape class of a batch.  A long-tailed mix of regions (3 x 2 ... 5 000 x 128, haplotypes of 60 ... 500 bases)
// falls into dozens of <K, streams> classes; launched one after the other, each left most of the chip idle and had a
// tail of its own (1 536 mixed regions: 72 launches, 145 ms for 12.6 ms of work).  The wave reads its item, branches
// once (scalar) to the body compiled for that K, and never meets another K again; the launch is sorted longest item
// first across all classes.  Registers are those of the largest body, which costs nothing: the 19 KB LDS ring
// already limits the kernel to two waves per SIMD.
#define PHMM_CHAIN_K_LIST(X) \
    X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) \
    X(23) X(24) X(25)

// (Round 3: one kernel per RANGE of K -- kChainRange below -- instead of one for all 24 bodies: that one carried 725
// spilled SGPRs and 12 bytes of scratch per lane and ran a uniform batch 1.5-2 % slower than the body alone.  The
// planner groups a launch's items by range; launches of different groups go out on parallel streams.)
template <int CLT, int KLO, int KHI>
__global__ __launch_bounds__(WAVE, 2) void phmm_forward_chain(const ChainParams cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ChainItem it = cp.items[blockIdx.x];
    switch (__builtin_amdgcn_readfirstlane((int)it.k)) {
#define PHMM_CASE(KK)                                                  \
    case KK:                                                           \
        if constexpr (KK >= KLO && KK <= KHI) chain_body<CLT, KK>(cp.f, it, smem); \
        break;
        PHMM_CHAIN_K_LIST(PHMM_CASE)
#undef PHMM_CASE
        default:
            break;
    }
}

// The same body as a kernel of its own: what a launch whose items all share one K uses (the uniform batches of
// BASELINE.json).  Compiled alone a body keeps its own register allocation -- the any-K kernel above carries the
// scalar-register pressure of 24 bodies and runs the very same items 1.5-2 % slower (config 2: 10.69 vs 10.50 ms).
template <int CLT, int K>
__global__ __launch_bounds__(WAVE, 2) void phmm_forward_chain_k(const ChainParams cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    chain_body<CLT, K>(cp.f, cp.items[blockIdx.x], smem);
}

#define PHMM_CHAIN_CAT2(a, b) a##b
#define PHMM_CHAIN_CAT(a, b) PHMM_CHAIN_CAT2(a, b)
#define PHMM_CHAIN_LAUNCH PHMM_CHAIN_CAT(launch_chain_L, PHMM_CHAIN_L)

// single_k: the K every item of the launch has, or 0 for a mixed launch
hipError_t PHMM_CHAIN_LAUNCH(int single_k, const ChainParams &cp, hipStream_t stream) {
    const size_t lds = (size_t)RING_SLOTS * sizeof(RowConst) + (CHAIN_META + 4) * sizeof(uint32_t);
#define PHMM_CASE(KK)                                                                                     \
    if (single_k == KK) {                                                                                 \
        hipLaunchKernelGGL((phmm_forward_chain_k<CL, KK>), dim3(cp.n_items), dim3(WAVE), lds, stream, cp); \
        return hipGetLastError();                                                                         \
    }
    PHMM_CHAIN_K_LIST(PHMM_CASE)
#undef PHMM_CASE
    // a mixed launch: single_k = -(range index + 1)
#define PHMM_RANGE(R, LO, HI)                                                                                       \
    if (single_k == -(R + 1)) {                                                                                     \
        hipLaunchKernelGGL((phmm_forward_chain<CL, LO, HI>), dim3(cp.n_items), dim3(WAVE), lds, stream, cp);        \
        return hipGetLastError();                                                                                   \
    }
    PHMM_CHAIN_RANGES(PHMM_RANGE)
#undef PHMM_RANGE
    return hipErrorInvalidValue;
}

#if PHMM_CHAIN_L == 16
hipError_t launch_chain_L32(int single_k, const ChainParams &cp, hipStream_t stream);
hipError_t launch_chain_L64(int single_k, const ChainParams &cp, hipStream_t stream);
int chain_max_k() { return 25; }
hipError_t launch_chain(int L, int single_k, const ChainParams &cp, hipStream_t stream) {
    if (!cp.n_items) return hipSuccess;
    return L == 16 ? launch_chain_L16(single_k, cp, stream) : L == 32 ? launch_chain_L32(single_k, cp, stream)
         : L == 64 ? launch_chain_L64(single_k, cp, stream) : hipErrorInvalidValue;
}
#endif

}  // namespace phmm
