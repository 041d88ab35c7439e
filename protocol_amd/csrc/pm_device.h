// pm_device.h — argument blocks shared by the kernels (pm_kernels.hip) and the engine (pm_engine.cpp).
#ifndef PM_DEVICE_H
#define PM_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pm_engine.h"

namespace pm {

static constexpr double PM_RAD = 3.14159265358979323846 / 180.0;  // f64::to_radians factor
static constexpr uint64_t PM_KEY_NOLOC = 0x7FEFFFFFFFFFFFFFull;   // bits of f64::MAX (mod.rs:244,249)
static constexpr double PM_A_MAX_SAFE = 1.0 - 1.0 / 1048576.0;    // near-antipodal => settle on host
// below this Haversine term (about 10 km) the chord-length form of the key is not accurate enough for the
// certificate band and the proposer evaluates the sine form instead (see prox_key in pm_kernels.hip)
static constexpr double PM_A_CHORD_MIN = 6.2e-7;
// carve kernel geometry
static constexpr uint32_t PM_CARVE_SLOTS = 8192;       // candidate slots with positions/bitmaps in LDS, keys in VGPRs
static constexpr uint32_t PM_CARVE_PART = 64;          // per-wave partial selection capacity (max_group_size - 1)
static constexpr uint32_t PM_CARVE_SEL_CAP = 256;      // selected slots staged in LDS before the member stores
static constexpr uint32_t PM_CARVE_SLOT_BITS = 13;     // log2(PM_CARVE_SLOTS): low key bits that hold the slot
static constexpr uint32_t PM_CARVE_BIG_SLOTS = 262144; // lists up to here keep their bitmaps + staged rows in LDS
static constexpr uint32_t PM_CARVE_SLOT_BITS_BIG = 18; // log2(PM_CARVE_BIG_SLOTS)
static constexpr uint32_t PM_CARVE_SLOT_BITS_MEM = 21; // same for lists kept in HBM (up to 2M candidates)
// certificate bands: 8x the truncation step of the packed key (2^-(52-bits)) — see carve_kernel
static constexpr double PM_TIE_BAND = 1.0 / 68719476736.0;      // 2^-36
static constexpr double PM_TIE_BAND_BIG = 1.0 / 2147483648.0;   // 2^-31
static constexpr double PM_TIE_BAND_MEM = 1.0 / 268435456.0;    // 2^-28
static constexpr uint32_t PM_PROP_ROW = 96;            // proposal row stride (u64 words): word 0 = flags, words 1..63 = keys,
static constexpr uint32_t PM_PROP_SLOTS = 64;          // then from this word on 64 x u32: flags, the slots of entries 0..62
static constexpr uint32_t PM_PROP_KMAX = 63;           // neighbours a row can list (word 0 carries the flags)
// flags word of a proposal row (low 32 bits of word 0): bits 0..7 entries, bits 8..15 first entry that lies within
// the certificate band of the LAST entry (a selection that ends in front of it never touches the row's tail)
static constexpr uint32_t PM_ROW_SAFE = 1u << 27;        // no listed Haversine term beyond PM_A_MAX_SAFE
static constexpr uint32_t PM_ROW_TAIL_CLEAR = 1u << 28;  // the first unlisted candidate is beyond the band of the last entry
static constexpr uint32_t PM_ROW_CLEAN = 1u << 29;       // no two entries within the band of each other at different sites
static constexpr uint32_t PM_ROW_TAIL_OK = 1u << 30;     // everything unlisted within the band of the last entry sits at its site
static constexpr uint32_t PM_ROW_COMPLETE = 1u << 31;    // the row lists every live candidate
static constexpr uint32_t PM_PROP_RESERVE = 64;        // entries beyond max_group_size - 1: the proposer's register holds 64 sorted
                                                       // keys whatever K is, so every row is as long as a row can be (K = 63)
static constexpr uint32_t PM_PROP_MAX_SEEDS = 16384;   // located slots that get a proposal per configuration
// spatial index of a carve's located positions (cell_*_kernel): a G x G x G grid over the unit-vector cube [-1, 1]^3,
// x fastest, so a run of cells along x is one contiguous range of the cell-sorted entries
static constexpr uint32_t PM_CELL_G_MAX = 64;
static constexpr uint32_t PM_CELL_TABLE = PM_CELL_G_MAX * PM_CELL_G_MAX * PM_CELL_G_MAX + 2;  // starts of every cell + the end
static constexpr uint32_t PM_CELL_MIN_N = 6000;        // eligible positions below which the grid is 32^3 whatever is asked for (forced modes)
// eligible positions from which a carve builds the index of its own accord (prune_mode 1).  Measured with the streaming
// carve (profiles/r05_index_crossover.txt, tools/index_crossover.py): with and without the index a cold match costs
// the same within +-3 % from 6,000 to 30,000 workers (10,000: 1.074 against 1.047 ms — the four cell_* launches are
// not paid back), and the index wins from 50,000 on (3.88 against 3.99 ms; 100,000: 6.9 against 9.1)
static constexpr uint32_t PM_CELL_AUTO_N = 40000;
static constexpr uint32_t PM_CELL_BIG_N = 40000;       // ... from which the grid is 64^3 instead of 32^3
static constexpr uint32_t PM_CELL_RMAX = 14;           // rings of cells a seed walks before it falls back to the whole list
// part | alive, loc bitmaps | wid | site | key | sel_out | BlockRed + s_n
static constexpr size_t PM_CARVE_LDS_BYTES = size_t(16) * PM_CARVE_PART * 8 + size_t(PM_CARVE_SLOTS / 64) * 16 +
                                             size_t(PM_CARVE_SLOTS) * 16 + size_t(PM_CARVE_SEL_CAP) * 4 + 1024;
// ---- streaming carve (carve_stream_kernel): one launch per try_form_new_groups pass; workgroup 0 validates, the
// others compute neighbour rows for the seeds a bounded look-ahead in front of the chain
static constexpr uint32_t PM_STREAM_SQ = 8192;      // seed-ticket ring (8-byte granules): > waves that can hold a claim
static constexpr uint32_t PM_STREAM_RQ = 8192;      // row ring: rows of tickets t, t + RQ share a slot
static constexpr uint32_t PM_STREAM_TP = 4096;      // tickets whose seed position the validator remembers (LDS)
static constexpr uint32_t PM_STREAM_LA_MAX = 3072;  // look-ahead: tickets issued and not yet handed to the chain
static constexpr uint32_t PM_STREAM_CTL_WORDS = 64; // control block (u32): see the SC_* indices
#ifndef PM_STREAM_PROP_WAVES_N  // (settable in a variant build: with few row-making workgroups per pool — K pools on one
#define PM_STREAM_PROP_WAVES_N 4  // GPU — eight waves a workgroup trade a row's latency for rows per second)
#endif
static constexpr uint32_t PM_STREAM_PROP_WAVES = PM_STREAM_PROP_WAVES_N; // waves of a proposer workgroup that build rows (one per SIMD)
static constexpr uint32_t PM_STREAM_SLW_TREQ = PM_STREAM_TP;      // words of the validator's ticket state that
static constexpr uint32_t PM_STREAM_SLW_PAY = PM_STREAM_TP + 1;   // carve_fast_steps<STREAM> reads (pm_stream.inc: SLW_*)
static constexpr size_t PM_STREAM_LDS_BYTES = PM_CARVE_LDS_BYTES + size_t(PM_STREAM_TP) * 4 + 128;
// control words in global memory (each hot word on its own 64-byte line)
enum { SC_CLAIM = 0,    // next ticket a proposer wave takes (atomicAdd)
       SC_QUIT = 16,    // the validator is through: proposers leave
       SC_DROP = 32,    // configuration << 26 | ticket: that configuration's tickets below this one are not wanted any more (its
                        // tickets were issued afresh, or it is over) — a row maker that holds one lets it be
       SC_ROWS = 48,    // rows the proposers delivered (statistics)
       SC_LET_BE = 53,  // rows given up half-made because their ticket was dropped (statistics)
       SC_GAVE_UP = 49, // proposer waves that left because nothing was asked of them for too long
       SC_FINISH = 50,  // the carve launch ran (carve_finish_kernel has records to finish)
       SC_FINISH_TICKET = 52,  // carve_finish_kernel: blocks through (the last one mirrors the status to the host)
       SC_TRACE = 51,   // PM_CARVE_PROF builds: events written to the trace buffer
       SC_BATCHES = 54  // PM_ROW_REC builds: batches of steps the chain has recorded
};
static constexpr uint32_t PM_STREAM_TRACE_CAP = 1u << 17;  // events (two u64 each) of a PM_CARVE_PROF build's timeline
// how the rows of a ticket are made (bits 24..25 of the ticket's payload)
enum { SROW_NONE = 0, SROW_BITMAP = 1, SROW_WALK = 2, SROW_SWEEP = 3 };

struct CompatArgs {
  uint32_t W, n_cfgs, model_words;
  const uint32_t *flags, *gpu_count, *gpu_mem, *gpu_cls, *cpu_cores, *ram, *storage;
  const pm_config_row* cfgs;
  const pm_gpu_alt_row* alts;
  const uint32_t* model_bits;
  uint64_t* compat;
};

struct ClaimArgs {
  uint32_t R;            // rows: all workers, or the workers `rows` lists (multi-GPU: the ones this rank owns)
  const uint32_t* rows;  // nullptr = row r is worker r
  const int32_t* group_of;
  const uint32_t *g_n, *g_off, *g_task;
  const uint64_t* g_id;
  uint32_t* g_task_next;
  const uint32_t* chosen;         // per row
  const uint32_t* rank_in_group;  // per worker
  const uint32_t* by_rank;        // per member slot: worker of that rank
  const uint64_t* t_live;     // task table: live bitmap and live-prefix per word (handle -> published position)
  const uint32_t* t_prefix;
  pm_assignment* table;  // per worker; with `rows`: per row (this rank's segment of the exchange buffer)
  uint32_t* task_col;    // compact per-worker task column (device-side consumers; not written with `rows`)
  // The snapshot buffer the look-ups will read (pinned host memory, the one that is not current) and the groups' task words
  // beside it: written by the claim itself — a row is 32 bytes of a coalesced store over PCIe — instead of by two copies
  // queued behind it (each a launch of the runtime's copy kernel with a 12 us gap in front).  nullptr: the caller copies.
  pm_assignment* h_table;
  uint32_t* h_gtask;     // [groups]: written by the group's member of rank 0
};

// Behind the carve, in front of the pair sweep (one engine, every worker a row): what is per worker or per group and needs
// no pair — the worker's selector, the sweep's outputs at their neutral values, GROUP_INDEX and the members by rank, the
// groups' task words carried over — in ONE launch where there were three kernels and a copy (each 4 - 5 us with 5 - 6 us of an
// idle GPU in front: profiles/r06_timeline.txt).
struct MatchPrepArgs {
  uint32_t W, G;
  const int32_t* group_of;
  const uint32_t *g_cfg, *g_n, *g_off, *members, *addr_rank, *g_task;
  uint64_t* sel;                      // [W]
  uint32_t *first, *count;            // [W]: PM_NONE / 0
  uint32_t *rank_in_group, *by_rank;  // group_rank_kernel's outputs
  uint32_t* g_task_next;              // [G] <- g_task
};

// pm_update_workers / pm_append_workers: n packed rows (one H2D copy) scattered into the worker columns
struct RowUpdateArgs {
  uint32_t n;
  const double *lat_in, *lon_in;  // [n] each
  const uint32_t* u32_in;         // 10 columns of n: idx, flags, gpu_count, gpu_mem, gpu_cls, cpu_cores, ram, storage, addr_rank, site
  uint32_t *flags, *gpu_count, *gpu_mem, *gpu_cls, *cpu_cores, *ram, *storage, *addr_rank, *site;
  double *lat, *lon, *coslat;
  double *ux, *uy, *uz;  // unit vector of the location (cos lat cos lon, cos lat sin lon, sin lat)
};

enum { CARVE_MODE_FORM = 0, CARVE_MODE_MERGE = 1 };
enum { CARVE_STATE_RUNNING = 0, CARVE_STATE_DONE = 1, CARVE_STATE_UNCERTAIN = 2, CARVE_STATE_OVERFLOW = 3,
       CARVE_STATE_ABORTED = 4 };  // ABORTED: a bounded wait inside the launch gave up (what was committed stands)
// launch flags of carve_kernel
enum {
  CARVE_F_INIT = 1u << 0,   // build the eligible list + position columns (first launch of a carve)
  CARVE_F_RUN = 1u << 1,    // process the prepared configuration
  CARVE_F_ALL = 1u << 2,    // keep going through every configuration in this launch (no proposals)
  CARVE_F_PROPS = 1u << 3,  // neighbour-list proposals of carve_propose_kernel are available
  CARVE_F_EXTPREP = 1u << 4 // the candidate lists are prepared by carve_prep_*_kernel between the launches
};
// Device-resident state of one carve (try_form_new_groups / one merge configuration); it persists across
// the launches of the propose / validate sequence.
struct CarveStatus {
  uint32_t state;
  uint32_t stop_ci;      // configuration (position in the carve order) to resume at after UNCERTAIN
  uint32_t n_groups;     // records written so far (in/out)
  uint32_t n_members;    // member slots used so far (in/out)
  uint32_t steps_total;  // committed steps over all launches of this tick
  uint32_t stop_seed;    // worker index of the seed of the uncertain step (diagnostic)
  uint32_t n_eligible;   // length of the eligible list
  uint32_t cur_ci;       // prepared configuration (position in the carve order); n_avail = none left
  unsigned long long cand_sum;  // sum over committed steps of the candidates scanned
  uint32_t n_list;       // slots of the prepared candidate list
  uint32_t prop_k;       // entries per proposal for the prepared configuration (0 = no proposals)
  uint32_t prop_limit;   // proposals exist for located slots below this slot number
  uint32_t rows_pr;      // rows per rank of this batch in the proposal buffer: ceil(seeds / world)
  uint32_t n_seeds;      // seeds of the batch
  uint32_t total_available;
  uint32_t fast_steps;   // steps committed from proposals
  uint32_t slow_steps;   // steps that needed the full key sweep
  uint32_t n_solo;       // single-node groups carved (the merge pass only runs when there are two or more)
  uint32_t n_props;      // neighbour lists computed by this rank's proposer
  uint32_t need_prep;    // (external preparation) the next candidate list has not been prepared yet
  uint32_t g_lo, g_hi;   // groups appended by the last validation launch (their group_of is written by the prep kernels)
  uint32_t n_batches;    // validation launches that had a prepared list to run (the host sizes its next queue by it)
  uint32_t n_void;       // validation launches whose batch had been prepared for another configuration, or for nothing
  // how validation launches ended: 0 chain: list thinned out, 1 seeds of the batch used up, 2 exact step: thinned out,
  // 3 configuration exhausted, 4 batch prepared for another configuration, 5 batch too stale, 6 configuration not entered
  uint32_t why[8];
  uint32_t stream_timeouts; // streaming carve: rows the validator stopped waiting for (the step took the exact sweep)
  uint32_t stream_tickets;  // streaming carve: seeds handed to the proposers
  uint32_t stream_switches; // streaming carve: configurations that went from walking the index to sweeping the bitmap
  uint32_t stream_listed;   // streaming carve: configurations entered sweeping the bitmap
  uint32_t stream_pre_used; // streaming carve: configurations that started on tickets issued ahead
  uint32_t stream_pre_lost; // streaming carve: tickets issued ahead for a configuration that was not the next one
  uint32_t stream_refreshes; // streaming carve: times a configuration's outstanding tickets were dropped and issued afresh
  uint32_t _pad_sr;
  uint32_t cell_g;          // grid size of the spatial index built for this carve's positions (0 = none)
  uint32_t n_indexed;       // located positions in the index
  uint32_t pruned_batches;  // batches whose proposals walked the index instead of the whole list
  uint32_t prune_fallbacks; // seeds of such batches that gave up on the index (rings exhausted) and swept the list
#ifdef PM_BATCH_LOG  // (experiment builds, tools/prune_probe.py: what every preparation of the carve produced)
  uint32_t blog_n;
  uint32_t blog[3 * 512];   // per preparation: list length (0 = none), seeds, grid of the walk (0 = whole-list sweep)
#endif
  unsigned long long prop_keys;  // keys (Haversine terms) those sweeps evaluated
  unsigned long long prof[64];  // PM_CARVE_PROF builds: accumulated s_memtime ticks per phase
};

// A proposal batch as its preparation describes it (carve_plan_kernel + carve_prep_*_kernel write it, the proposer
// and the validator read it).  A batch may be prepared while the batch in front of it is still being validated —
// from a snapshot of the position bitmap, for the configuration the carve is expected to be at — so the validator
// accepts it only if it started from the configuration the carve really is at (ci0), and treats what has been
// removed since the snapshot as dead slots.
struct BatchDesc {
  uint32_t planned;          // the plan ran (the carve was RUNNING)
  uint32_t ci0;              // configuration (position in the carve order) the preparation started from
  uint32_t total_available;  // as of the plan (an upper bound of what the validator will find)
  uint32_t valid;            // a candidate list was prepared: configuration `ci`, the first from ci0 on that can be entered
  uint32_t none;             // no configuration from ci0 on can be entered any more
  uint32_t ci, n_list, prop_k, prop_limit, rows_pr, n_seeds;
  uint32_t cell_g;           // the proposals of this batch walk the spatial index (grid size; 0 = sweep the whole list)
};

struct CarveArgs {
  uint32_t mode;   // CARVE_MODE_*
  uint32_t _flags_unused;
  uint32_t W;
  uint32_t proximity;
  uint32_t debug_uncertain_every;
  uint32_t _pad_re;
  // worker columns
  const uint32_t* wflags;
  const double *lat, *lon, *coslat;
  const double *ux, *uy, *uz;  // unit vectors: the proposer's chord-length key (see prox_key)
  const uint32_t* site;  // equal (lat, lon) bit patterns <=> equal site id (host-interned)
  const uint64_t* compat;
  int32_t* group_of;  // FORM: read (eligibility) and written (commit)
  // ordered eligible list: FORM -> written by the kernel (eligible rows in input order);
  // MERGE -> supplied by the engine (nodes of the compatible solo groups in group-id order)
  uint32_t* order;
  uint32_t n_order;
  uint32_t _pad2;
  // scratch (capacity W each): columns indexed by position in the eligible list ...
  double *c_lat, *c_lon, *c_cos;
  double *c_ux, *c_uy, *c_uz;
  uint32_t* c_site;
  uint64_t* c_compat;
  uint64_t *alive_g, *loc_g;  // bitmaps over positions (bits_stride words each)
  // ... and by candidate slot of the prepared configuration
  double *cc_lat, *cc_lon, *cc_cos;
  double *cc_ux, *cc_uy, *cc_uz;
  uint32_t* cc_site;
  uint32_t* slot_pos;      // slot -> position (for the alive_g write-back)
  uint32_t* slot_wid;      // slot -> worker id
  uint64_t* bits_scratch;  // slot bitmaps: alive, loc (bits_stride words each)
  uint64_t* keys;          // packed keys when the candidate list does not fit in LDS
  uint32_t bits_stride;
  uint32_t _pad0;
  // proposals (carve_propose_kernel): one row of PM_PROP_ROW u64 per seed of the batch — the flags word (PM_ROW_*)
  // in word 0, then the packed keys sorted ascending.  Seed i of a batch (rank among the live located slots
  // below prop_limit) belongs to rank i % world and is row i / world of that rank's segment; `prop` holds all
  // segments back to back ([world][rows_pr] rows — what the all-gather delivers), `prop_send` is this rank's
  // segment (the same memory as `prop` when world == 1).
  uint64_t* prop;
  uint64_t* prop_send;
  uint64_t* seed_map;      // per bitmap word below prop_limit: the batch's seeds (live & located at preparation)
  uint32_t* seed_prefix;   // per bitmap word: seeds in front of the word
  uint32_t* seed_slots;    // seed number -> slot (PM_PROP_MAX_SEEDS + 64 entries)
  uint32_t dist_rank, dist_world;
  uint32_t count_keys, _pad_ck;  // proposer: count the keys it sweeps (bench bookkeeping)
  uint32_t* prep_block_counts;   // [blocks][PM_MAX_CONFIGS] live compatible positions per block and configuration
  uint32_t* prep_counts;         // [PM_MAX_CONFIGS] totals, [PM_MAX_CONFIGS] = finished-blocks ticket
  BatchDesc* desc;               // this argument block's batch (the per-batch scratch above belongs to it)
  uint64_t* alive_snap;          // the position bitmap as the preparation saw it (bits_stride words)
  uint32_t _pad_sp;
  uint32_t debug_mem_above;      // test hook: candidate lists longer than this take the all-in-HBM path (0 = off)
  // spatial index over the located positions (built once per carve behind the eligible list, see cell_count_kernel)
  uint32_t* cell_cnt;            // [PM_CELL_TABLE] members per cell while the index is built; zero between builds
  uint32_t* cell_start;          // [PM_CELL_TABLE] first entry of every cell (+ the end)
  uint32_t *pos_cell, *pos_rank; // per position: its cell, its rank among the cell's members
  uint32_t* cs_of_pos;           // per position: its entry in cell order (located positions only)
  uint32_t* cs_slot;             // per entry: the candidate slot it has in the prepared list, ~0 = none (per batch)
  double *cs_ux, *cs_uy, *cs_uz; // per entry: unit vector
  uint32_t* cs_site;             // per entry: site id
  uint32_t prune_mode;           // 0 never, 1 when it pays (list length vs live fraction), 2 whenever there is an index,
                                 // 3 = 2 with every seed forced through the whole-list fallback (test hooks)
  uint32_t prune_factor;         // mode 1: walk when n_list^2 >= prune_factor x (indexed positions)
  uint32_t walk_cap_div, _pad_w; // seeds of a batch that walks the index: n_list / walk_cap_div
  // ---- streaming carve (carve_stream_kernel).  Slot == position: cc_* alias c_*, slot_wid aliases order,
  // bits_scratch = {the published `free` bitmap (positions no group holds yet — the proposers' view, a few commits
  // behind), loc_g}, alive_g = the validator's own master copy of it (brought up to date between configurations).
  uint32_t stream, stream_tag0;  // stream: 1 = this argument block drives carve_stream_kernel; tag0: first ticket tag
  // gather copies of the columns a row maker reads per candidate, 32 bytes a record = ONE memory line per candidate
  // instead of five (unit vector x 3, site, located bit): {ux, uy, uz, site | located << 32} — a row is a chain of
  // gathers, and its latency is what the chain waits for at every cold start
  double* c_pack;                // [n][4] by position (carve_elig_place_kernel)
  double* cs_pack;               // [n_indexed][4] by entry of the spatial index (cell_place_kernel)
  uint64_t* cfgbits;             // [n_avail][bits_stride] per configuration (carve order): compatible positions
  unsigned long long* stream_sq;      // [PM_STREAM_SQ] seed tickets: {tag, position | ci << 18 | mode << 24}
  unsigned long long* stream_row_lo;  // [PM_STREAM_RQ][64] rows: {tag, flags word | low half of the packed key of entry g - 1}
  unsigned long long* stream_row_hi;  // [PM_STREAM_RQ][64]       {tag, high half}
  uint32_t* stream_ctl;               // [PM_STREAM_CTL_WORDS] SC_*
  uint32_t stream_la, stream_row_spins;  // look-ahead cap (0 = default); polls before the validator gives a row up
  uint32_t stream_la_div;                // look-ahead = candidates / (la_div x (max_group_size - 1)) (0 = default)
  uint32_t debug_abort_after;            // test hook (pm_debug_stream_abort_after): the chain gives the launch up — CARVE_STATE_ABORTED —
                                         // once this many steps of the carve are committed (0 = off)
  // behind the streaming carve, carve_finish_kernel completes the records of the groups the launch appended — worker ids for
  // positions, group_of, and: their ids (generate_group_id's stream: output k + 1 of id_state for group id_g0 + k), an
  // empty task word, and a copy of the records and of the status in pinned HOST memory (stage_*, h_status: written by the
  // kernel over PCIe) — so that nothing has to be copied, and no launch queued, between the carve and the pair sweep
  unsigned long long id_state;
  uint32_t id_g0, stage_m0;              // first group / member slot of this carve (the staging arrays start there)
  uint32_t stage_cap_g, stage_cap_m;     // entries the staging arrays hold
  uint32_t *stage_cfg, *stage_n, *stage_off, *stage_mem;  // pinned host memory (null: no host copy)
  unsigned long long* g_id_out;          // [cap_groups] ids (null: none written)
  uint32_t* g_task_out;                  // [cap_groups] task words
  CarveStatus* h_status;                 // pinned host memory (null: no mirror)
  unsigned long long* stream_trace;      // PM_CARVE_PROF builds: [PM_STREAM_TRACE_CAP][2] {s_memtime, type | a << 8 | b << 32}
  // configurations in carve order (get_available_configurations, mod.rs:399-418)
  uint32_t n_avail, start_ci;
  uint32_t avail_cfg[PM_MAX_CONFIGS];
  uint32_t min_size[PM_MAX_CONFIGS];
  uint32_t max_size[PM_MAX_CONFIGS];
  // output group records (slots n_groups.. are appended)
  uint32_t *g_cfg, *g_n, *g_off, *members;
  uint32_t cap_groups, cap_members;
  CarveStatus* status;
};

void launch_compat(const CompatArgs& a, hipStream_t s);
void launch_geo(const double* lat, const double* lon, double* coslat, double* ux, double* uy, double* uz, uint32_t W,
                hipStream_t s);
void launch_triad(const double* b, const double* c, double* a, size_t n, hipStream_t s);
#ifdef PM_ROW_BENCH
hipError_t launch_row_bench(const CarveArgs* d_args, uint32_t seed, uint32_t ci, uint32_t reps, uint32_t mode, unsigned long long* d_out, hipStream_t s);
#endif
void launch_row_network_test(const uint64_t* keys, const uint32_t* sites, uint32_t n_waves, uint32_t n_per_wave, uint32_t slot_bits,
                             uint64_t ulps, uint32_t upto, uint64_t* rows_out, uint32_t* mismatches, hipStream_t s);
void launch_update_rows(const RowUpdateArgs& a, hipStream_t s);
void launch_worker_selector(const int32_t* group_of, const uint32_t* g_cfg, uint32_t R, const uint32_t* rows,
                            uint64_t* sel, hipStream_t s);
void launch_eligible_selector(const uint32_t* wflags, const int32_t* group_of, const uint64_t* compat,
                              uint64_t enabled, uint32_t W, const uint8_t* shard, uint32_t my_rank, uint64_t* sel,
                              hipStream_t s);
void launch_chooser_rank(const int32_t* group_of, const uint64_t* g_id, const uint32_t* count, uint32_t R,
                         const uint32_t* rows, uint64_t seed, uint32_t* rank, hipStream_t s);
void launch_table_scatter(const pm_assignment* x, const uint32_t* xrow, uint32_t W, pm_assignment* table,
                          uint32_t* task_col, uint32_t* g_task_next, const uint64_t* t_live, const uint32_t* t_prefix,
                          hipStream_t s);
void launch_group_rank(const int32_t* group_of, const uint32_t* g_n, const uint32_t* g_off, const uint32_t* members,
                       const uint32_t* addr_rank, uint32_t W, uint32_t* rank_in_group, uint32_t* by_rank,
                       hipStream_t s);
void launch_claim_publish(const ClaimArgs& a, hipStream_t s);
void launch_match_prep(const MatchPrepArgs& a, hipStream_t s);
void launch_build_planes(const uint64_t* col_mask, uint32_t n_cols, uint32_t c_begin, uint32_t c_end, uint32_t stride,
                         uint32_t n_planes, uint64_t* planes, hipStream_t s);
void launch_pair_sweep(int variant, const uint64_t* row_sel, uint32_t R, const uint64_t* col_mask,
                       const uint64_t* planes, uint32_t c_begin, uint32_t c_end, uint32_t stride, uint32_t n_planes,
                       uint32_t* first, uint32_t* count, hipStream_t s, bool inited = false);
void launch_pair_select(int variant, const uint64_t* row_sel, uint32_t R, const uint64_t* col_mask,
                        const uint64_t* planes, uint32_t c_begin, uint32_t c_end, uint32_t stride, uint32_t n_planes,
                        const uint32_t* rank, uint32_t* out, hipStream_t s);
void launch_task_prefix(const uint64_t* live, uint32_t w_begin, uint32_t w_end, uint32_t* prefix, hipStream_t s);
void launch_task_delete(const uint32_t* slots, uint32_t n, uint64_t* tmask, uint64_t* live, uint64_t* planes,
                        uint32_t stride, uint32_t n_planes, hipStream_t s);
void launch_task_intern(const uint64_t* tmask, const uint64_t* live, uint32_t u_begin, uint32_t u_end, uint64_t valid,
                        uint64_t* keys, uint32_t n_slots, uint32_t* vals, uint32_t* counter_and_overflow, uint64_t* umask,
                        uint32_t cap_u, hipStream_t s);
void launch_task_compact_class(const uint32_t* first_c, const uint32_t* count_c, const uint64_t* tmask, uint64_t valid,
                               const uint64_t* keys, const uint32_t* vals, uint32_t n_slots, uint32_t u_begin,
                               uint32_t u_end, const uint64_t* live, const uint32_t* prefix, uint32_t* first_out,
                               uint32_t* count_out, hipStream_t s);
void launch_task_compact(const uint32_t* first_u, const uint32_t* count_u, uint32_t u_begin, uint32_t u_end,
                         const uint64_t* live, const uint32_t* prefix, uint32_t* first_out, uint32_t* count_out,
                         hipStream_t s);
void launch_newest(const int64_t* created_at, const uint64_t* live, uint32_t t_begin, uint32_t t_end,
                   uint32_t* idx_by_block, long long* val_by_block, uint32_t n_blocks, hipStream_t s);
void launch_group_ids(uint64_t* g_id, uint32_t* g_task, uint32_t n, uint64_t rng_state, hipStream_t s);
hipError_t carve_kernels_init();  // per device: the carve kernels' dynamic LDS sizes (pm_engine_create)
hipError_t launch_carve(const CarveArgs* d_args, uint32_t flags, uint32_t start_ci, size_t lds_bytes, hipStream_t s);
uint32_t launch_carve_prep(const CarveArgs* d_args, uint32_t W, hipStream_t s);  // count + place
// eligible list [+ spatial index]; returns the launches.  fresh: the status block is initialised by the first kernel
// (state RUNNING, n_groups0 groups, n_members0 member slots, everything else zero) instead of by a copy in front of it
hipError_t launch_carve_args_put(const CarveArgs& a, CarveArgs* d_args, hipStream_t s);
uint32_t launch_carve_elig(CarveArgs* d_args, const CarveArgs* put, uint32_t W, uint32_t n_bound, uint32_t index_min, uint32_t start_ci,
                           bool fresh, uint32_t n_groups0, uint32_t n_members0, hipStream_t s);
void launch_carve_apply(const CarveArgs* d_args, uint32_t W, hipStream_t s);
void launch_carve_propose(const CarveArgs* d_args, uint32_t W, hipStream_t s);
hipError_t launch_carve_stream(const CarveArgs* d_args, uint32_t start_ci, uint32_t n_prop_wgs, hipStream_t s);
void launch_scatter_const(uint32_t* dst, const uint32_t* idx, uint32_t n, uint32_t v, hipStream_t s);
void launch_scatter_pairs(uint32_t* dst, const uint32_t* pairs, uint32_t n, hipStream_t s);
void launch_merge_place(const CarveArgs* d_args, uint32_t n_order, uint32_t n_groups0, uint32_t n_members0, uint32_t steps0,
                        hipStream_t s);

}  // namespace pm
#endif
