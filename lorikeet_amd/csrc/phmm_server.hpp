// The resident region server (phmm_server_kernels.hip, phmm_server.cpp): ONE kernel per device that stays on the chip while
// region calls keep coming and runs every step of a call -- stage-in, pre-step, PairHMM, post-step / best allele, aligner,
// projection -- as TASKS its waves take from a ready list in device memory.  What a call costs on the host is a memcpy into
// its slot of pinned memory, one 64-byte ring entry and a poll of one word; nothing is launched, no stream is synchronised,
// no caller waits for another caller's flush (NOTEBOOK section 20; SURVEY section 7 step 4, "persistent-kernel work queue
// across regions").  The sequence it replaces is haplotype_caller_engine.rs:1311-1357 as one worker of
// assembly_region_walker.rs:210-273 runs it per region.
//
// Shared by the kernel file and the host side; plain data only.
#pragma once
#include "phmm_cigar_internal.hpp"
#include "phmm_internal.hpp"
#include "phmm_sw_internal.hpp"

namespace phmm {

constexpr uint32_t SRV_RING = 256;        // ring entries between host and dispatcher (a power of two, > the slots a device has)
constexpr uint32_t SRV_MAIL = 1u << 17;   // mailboxes (a power of two): ticket t is served through mailbox t % SRV_MAIL; the host admits calls only while
constexpr uint32_t SRV_MAIL_TASKS = SRV_MAIL / 2;  // the tasks of all calls in flight stay below this, so a mailbox is read before its turn comes again
constexpr uint32_t SRV_LDS_BYTES = 19456;  // LDS of one worker wave: 8 waves per CU; 270 row records of the forward sweep
constexpr uint32_t SRV_STAGE_UNITS = 1024;  // 16-byte units one stage-in task copies
constexpr int SRV_MAX_K = 25;             // forward instances <16, 2..25> and <32, 2..13>: haplotypes up to 400 bases
constexpr uint32_t SRV_MAX_ROWS = SRV_LDS_BYTES / LDS_ROW_BYTES - 2;  // longest read

// The kinds of task, in the order a region's stages can become ready.
enum : uint32_t {
    SRV_STAGE = 0,  // 16 KB of the slot's pinned mirror -> its device arena (the job record travels with the inputs)
    SRV_PREP,       // one read: PCR indel model, quality caps, disqualification threshold (phmm_prep_device.hpp)
    SRV_FWD,        // one read x one group of four (two) haplotypes: forward_read<16 | 32, K>
    SRV_SWALL,      // (a call alone on the chip) one read x one haplotype: the aligner beside the PairHMM tasks
    SRV_POST,       // up to 64 reads: normalise, filter, best allele; with SRV_SWALL before it also the projection (pick_read)
    SRV_SW,         // one read against its best haplotype
    SRV_PROJ,       // some reads: the alignment onto the reference (project_read)
    SRV_KINDS
};

// One submission, as the host writes it into the ring (pinned host memory; `valid` last).  64 bytes = one line.
struct alignas(64) SrvEntry {
    uint32_t valid;            // sequence number + 1 once everything else is in place
    uint32_t slot;
    uint32_t n[SRV_KINDS];     // tasks per kind
    uint32_t flags;            // bit 0: trace this job's tasks
    uint32_t stage_n16;        // 16-byte units to stage in
    uint32_t job_off;          // where the SrvJob lies inside the staged block
    uint64_t stage_src;        // the mirror as the device sees it
    uint64_t stage_dst;        // the slot's device arena
};
static_assert(sizeof(SrvEntry) == 64, "ring entry");

// Everything the tasks of one submission need, made by the host inside the slot's mirror and staged with the inputs.
struct SrvJob {
    PrepParams prep;
    ForwardParams fwd;
    PostBestParams pb;
    SwParams sw;
    ProjectParams pj;
    uint32_t fwd_l, fwd_k;    // lanes per pair and columns per lane of the forward sweep: a function of the job's own shape (its pairs, its longest
                              // haplotype), never of the load
    uint32_t fwd_quads;       // groups of 64 / fwd_l haplotypes per read (tasks per read)
    uint32_t sw_k;            // rows per lane of the aligner's <64, k, transposed> instance
    uint32_t proj_per_task;   // reads one SRV_PROJ / picking SRV_POST task takes (their builders share the wave's LDS)
    uint32_t all_pairs;       // 1: SRV_SWALL beside SRV_FWD, SRV_POST picks
    uint32_t *finish_flag;    // a word of the mirror: 1 when the last task is through (the caller polls it)
};

// Device side of a submission (device memory, written by the dispatcher).
struct alignas(128) SrvRegion {
    uint32_t seq;
    uint32_t flags;
    uint32_t n[SRV_KINDS];
    uint32_t stage_n16;
    const SrvJob *job;         // the staged copy
    const void *stage_src;
    void *stage_dst;
    // (a line of their own: the counters are what every finishing task touches)
    alignas(64) uint32_t done[SRV_KINDS];
    uint32_t arrived[SRV_KINDS];  // predecessor stages that have completed (SRV_POST after SRV_SWALL waits for two)
};

// How a task reaches a worker.  A worker that wants work takes a TICKET (one atomic add on SrvCtl::next_ticket) and polls
// mailbox ticket % SRV_MAIL -- a line nobody else polls.  Whoever makes a stage of n tasks ready reserves n tickets' worth of
// mailboxes (one atomic add on SrvCtl::posted) and fills them in, a lane each.  Tickets are served in the order they were
// taken, so the tasks of a stage go to the workers that have been idle longest, each told through its own word: no worker
// ever polls a word another worker polls, and nobody claims a task somebody else gets (a first version with one shared
// ready list had two thousand idle waves polling one address; a region call took 1 ms, NOTEBOOK 20.2).
struct alignas(16) SrvMail {
    uint32_t tag;     // ticket + 1 once the other words are in place; SRV_MAIL_EXIT: leave
    uint32_t region;  // index into the region ring
    uint32_t kind;
    uint32_t idx;
};
constexpr uint32_t SRV_MAIL_EXIT = 0xffffffffu;

// Two CLASSES of worker, a ticket line and a ring of mailboxes each: the first wave of the server on a SIMD is that SIMD's
// PRIMARY worker, the second its SECONDARY.  Two waves that share a SIMD share its issue slots -- two PairHMM tasks side by side
// take 85 us each, one alone 55 -- so tasks go to primaries while primaries are waiting, and to secondaries only when the chip
// has more tasks than SIMDs (post(), phmm_server_kernels.hip).
struct SrvCtl {
    alignas(128) uint32_t next_ticket0;  // tickets taken by primary workers
    alignas(128) uint32_t next_ticket1;  // ... by secondary workers
    alignas(128) uint32_t posted0;       // tasks posted to the primaries' ring (mailboxes [0, posted) have been, or are being, filled in)
    alignas(128) uint32_t posted1;
    alignas(128) uint32_t closed;        // the dispatcher is leaving: a worker that takes a ticket now leaves too
    uint32_t fault;
    alignas(128) uint32_t finished;      // submissions whose last task is through
    alignas(128) uint32_t trace_count;
    alignas(128) uint32_t simd_waves[8192];  // by (XCC, SE, SH, CU, SIMD) of HW_ID: how many worker waves have reported from there
};

// What a server tells the host when it leaves (pinned host memory).
struct SrvExit {
    uint32_t epoch;      // the launch that wrote this (0: none yet)
    uint32_t consumed;   // ring entries taken by all launches so far: the next launch starts there
    uint32_t fault;      // != 0: the launch gave up (no progress within its time limit)
    uint32_t pad;
};

struct SrvTrace {  // one task, where tracing is on (developer runs: tools/server_trace.py)
    uint32_t seq, kind, idx, worker;
    uint64_t t_claim, t_begin, t_end;  // 100 MHz ticks
};

struct SrvParams {
    SrvCtl *ctl;
    SrvRegion *regions;        // [SRV_RING]
    SrvMail *mail;             // [2][SRV_MAIL]: the primaries' ring, the secondaries'
    const SrvEntry *ring;      // [SRV_RING], pinned host memory by its device address
    SrvExit *exit_word;        // pinned host memory by its device address
    uint32_t start_seq;        // first ring entry this launch looks at
    uint32_t epoch;
    uint32_t idle_ticks;       // the dispatcher leaves when nothing is in flight and nothing has arrived for this long (100 MHz ticks)
    uint32_t stall_ticks;      // ... and gives up (fault) when something is in flight and nothing has finished for this long
    SrvTrace *trace;           // or null
    uint32_t trace_cap;
};

hipError_t launch_server(const SrvParams &p, uint32_t n_blocks, hipStream_t stream);
int server_blocks_per_cu();  // what a CU holds of the server's one-wave blocks (registers, LDS)

}  // namespace phmm
