// The resident region server (phmm_server_kernels.hip, phmm_server.cpp): ONE kernel per device that stays on the chip while
// region calls keep coming.  A call's inputs are copied from its slot of pinned host memory by a few STAGE tasks; then every
// READ of the call is one CHAIN task -- a wave that runs the read's whole path by itself: pre-step, PairHMM against (a group
// of) its region's haplotypes, post-step and best allele, the alignment to that haplotype, the projection onto the reference --
// with helper waves for the further haplotype groups.  Nothing a read's steps hand each other ever leaves its wave (beyond four
// likelihoods per helper), so there are no stages to wait for, no launches, no stream synchronisation, and no caller waits for
// another caller's flush: what a call costs on the host is a memcpy into its slot, one 64-byte ring entry and a poll of one word
// (NOTEBOOK section 20; SURVEY section 7 step 4, "persistent-kernel work queue across regions").  The sequence is
// haplotype_caller_engine.rs:1311-1357 as one worker of assembly_region_walker.rs:210-273 runs it per region.
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
constexpr int SRV_MAX_K = 25;             // forward instances <16, 2..25> (haplotypes up to 400 bases) and <32, 13..16> (up to 512)
constexpr uint32_t SRV_MAX_ROWS = SRV_LDS_BYTES / LDS_ROW_BYTES - 2;  // longest read

enum : uint32_t {
    SRV_STAGE = 0,  // 16 KB of the slot's pinned mirror -> its device arena (the job record travels with the inputs)
    SRV_CHAIN,      // task i: read i / groups, one group of its region's haplotypes (four, or two long ones); the group-0 wave is the read's MAIN wave
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
    uint32_t pad[5];
    uint64_t stage_src;        // the mirror as the device sees it
    uint64_t stage_dst;        // the slot's device arena
};
static_assert(sizeof(SrvEntry) == 64, "ring entry");

// Everything the tasks of one submission need, made by the host inside the slot's mirror and staged with the inputs.
struct SrvJob {
    PrepParams prep;          // (waves_per_read = 1: the chain's wave takes the whole read)
    ForwardParams fwd;
    PostBestParams pb;
    SwParams sw;
    ProjectParams pj;
    uint32_t fwd_k;           // columns per lane of the forward sweep: 16 lanes per pair up to 400 bases, 32 beyond -- a function of the job's
                              // longest haplotype alone
    uint32_t group_haps;      // haplotypes a wave sweeps side by side: 4 (16 lanes per pair) or 2
    uint32_t groups;          // such groups per read = chain tasks per read
    uint32_t sw_k;            // rows per lane of the aligner's <64, k, transposed> instance
    uint32_t n_reads;
    // Words that waves of different XCDs hand each other are touched by agent-scope atomics ONLY, from the moment the stage-in's
    // last wave has zeroed them (a plain store leaves a line in some L2 that a later agent-scope load of another call may still find):
    uint32_t *group_done;     // [n_reads] helpers of the read that have stored their likelihoods
    double *helper_out;       // [n_reads][helper_stride] the helpers' likelihoods, column = haplotype index inside the region
    uint32_t helper_stride;
    uint32_t pad0;
    uint32_t *status_in;      // the forward sweeps' status word (device) ...
    uint32_t *status_out;     // ... which the call's last wave hands on to the caller's mirror
    uint32_t *finish_flag;    // a word of the mirror: 1 when the last task is through (the caller polls it)
    uint32_t wait_ticks;      // how long a main wave waits for its helpers at most (100 MHz ticks)
    uint32_t pad1;
};

// Device side of a submission (device memory, written by the dispatcher).
struct alignas(128) SrvRegion {
    uint32_t seq;
    uint32_t flags;
    uint32_t n[SRV_KINDS];
    uint32_t stage_n16;
    uint32_t pad0;
    const SrvJob *job;         // the staged copy
    const void *stage_src;
    void *stage_dst;
    // (a line of their own: the counters are what every finishing task touches)
    alignas(64) uint32_t done[SRV_KINDS];
    uint32_t timed_out;        // a main wave gave up waiting for a helper: the call fails
};

// How a task reaches a worker.  A worker that wants work takes a TICKET (one atomic add on SrvCtl::next_ticket) and polls
// mailbox ticket % SRV_MAIL -- a line nobody else polls.  Whoever makes n tasks ready reserves n tickets' worth of mailboxes
// (one atomic add on SrvCtl::posted) and fills them in, a lane each.  Tickets are served in the order they were taken, so
// tasks go to the workers that have been idle longest, each told through its own word: no worker ever polls a word another
// worker polls, and nobody claims a task somebody else gets (a first version with one shared ready list had two thousand idle
// waves polling one address; a region call took 1 ms, NOTEBOOK 20.2).
struct alignas(16) SrvMail {
    uint32_t tag;     // ticket + 1 once the other words are in place; SRV_MAIL_EXIT: leave
    uint32_t region;  // index into the region ring
    uint32_t kind;
    uint32_t idx;
};
constexpr uint32_t SRV_MAIL_EXIT = 0xffffffffu;

struct SrvCtl {
    alignas(128) uint32_t next_ticket;   // tickets taken
    alignas(128) uint32_t posted;        // tasks posted (mailboxes [0, posted) have been, or are being, filled in)
    alignas(128) uint32_t closed;        // the dispatcher is leaving: a worker that takes a ticket now leaves too
    uint32_t fault;
    alignas(128) uint32_t finished;      // submissions whose last task is through
    alignas(128) uint32_t trace_count;
};

// What a server tells the host when it leaves (pinned host memory).
struct SrvExit {
    uint32_t epoch;      // the launch that wrote this (0: none yet)
    uint32_t consumed;   // ring entries taken by all launches so far: the next launch starts there
    uint32_t fault;      // != 0: the launch gave up (no progress within its time limit)
    uint32_t pad;
};

struct SrvTrace {  // one task, where tracing is on (developer runs: tools/server_trace.cpp)
    uint32_t seq, kind, idx, worker;
    uint64_t t_claim, t_begin, t_end;  // 100 MHz ticks
    uint64_t t_mid[4];                 // chain tasks: pre-step done, PairHMM done, helpers in + post-step done, aligner done
};

struct SrvParams {
    SrvCtl *ctl;
    SrvRegion *regions;        // [SRV_RING]
    SrvMail *mail;             // [SRV_MAIL]
    const SrvEntry *ring;      // [SRV_RING], pinned host memory by its device address
    SrvExit *exit_word;        // pinned host memory by its device address
    const uint32_t *yield_word;  // pinned host memory: the host adds one when kernels of the launched kind need the chip -- the dispatcher
                               // then takes no further call, lets those in flight finish and leaves (the next launch follows them)
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
