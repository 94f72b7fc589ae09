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
constexpr uint32_t SRV_RECS = 4096;       // ready records in flight (a power of two; six per region at most)
constexpr uint32_t SRV_LDS_BYTES = 19456;  // LDS of one worker wave: 8 waves per CU; 270 row records of the forward sweep
constexpr uint32_t SRV_STAGE_UNITS = 1024;  // 16-byte units one stage-in task copies
constexpr int SRV_MAX_K = 25;             // forward instances <16, 2..25>: haplotypes up to 400 bases
constexpr uint32_t SRV_MAX_ROWS = SRV_LDS_BYTES / LDS_ROW_BYTES - 2;  // longest read

// The kinds of task, in the order a region's stages can become ready.
enum : uint32_t {
    SRV_STAGE = 0,  // 16 KB of the slot's pinned mirror -> its device arena (the job record travels with the inputs)
    SRV_PREP,       // one read: PCR indel model, quality caps, disqualification threshold (phmm_prep_device.hpp)
    SRV_FWD,        // one read x one group of four haplotypes: forward_read<16, K>
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
    uint32_t fwd_k;           // columns per lane of the forward sweep (16 lanes per pair): a function of the job's longest haplotype alone
    uint32_t fwd_quads;       // groups of four haplotypes per read (tasks per read)
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

// One ready stage: `n` tasks anybody may claim (next is the claim counter; it overshoots).
struct alignas(32) SrvRec {
    uint32_t valid;   // index + 1 once the record is complete
    uint32_t n, next;
    uint32_t region;  // index into the region ring
    uint32_t kind;
    uint32_t pad[3];
};

struct SrvCtl {
    alignas(128) uint32_t rec_reserved;  // records appended (or being appended)
    alignas(128) uint32_t head_rec;      // a record index nobody needs to look below (monotonic hint)
    alignas(128) uint32_t closed;        // the dispatcher has left: idle workers leave too
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

struct SrvTrace {  // one task, where tracing is on (developer runs: tools/server_trace.py)
    uint32_t seq, kind, idx, worker;
    uint64_t t_claim, t_begin, t_end;  // 100 MHz ticks
};

struct SrvParams {
    SrvCtl *ctl;
    SrvRegion *regions;        // [SRV_RING]
    SrvRec *recs;              // [SRV_RECS]
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
