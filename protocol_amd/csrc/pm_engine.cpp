// pm_engine.cpp — host side of libpm_engine.so: owns the HBM-resident SoA tables, drives the kernels of
// pm_kernels.hip on one HIP stream and implements the C ABI of include/pm_engine.h.
//
// Reference behaviour mirrored here (paths relative to /root/reference/crates/orchestrator/src):
//   plugins/node_groups/mod.rs:113-175   constructor checks + config sort      -> pm_set_configs
//   plugins/node_groups/mod.rs:478-628   try_form_new_groups                   -> pm_form_groups
//   plugins/node_groups/mod.rs:631-971   try_merge_solo_groups                 -> pm_merge_solo_groups
//   plugins/node_groups/mod.rs:1423-1487 dissolve_group                        -> pm_dissolve_group
//   plugins/node_groups/scheduler_impl.rs:11-110  filter_tasks (all workers)   -> pm_match
//   plugins/node_groups/status_update_impl.rs:8-39 handle_status_change        -> pm_on_worker_status
//   plugins/newest_task/mod.rs:8-19                                            -> pm_newest_task
//   scheduler/mod.rs:26-36               get_task_for_node                     -> pm_lookup_task_for_worker
// There is no CPU fallback: without a gfx950 device pm_engine_create fails with PM_ENODEV.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "pm_device.h"
#include "pm_engine.h"
#include "pm_internal.h"
#include "pm_members.h"

namespace pm {

static thread_local std::string g_last_error;

int32_t set_error(int32_t code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return set_error(PM_ENODEV, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  // capacity for n elements, keeping the first `keep` (device-to-device copy on `s` when the buffer moves)
  hipError_t grow_keep(size_t n, size_t keep, hipStream_t s) {
    if (n <= cap) return hipSuccess;
    size_t want = n + n / 4 + 64;
    T* q = nullptr;
    hipError_t e = hipMalloc((void**)&q, want * sizeof(T));
    if (e != hipSuccess) return e;
    if (p && keep) {
      e = hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);  // the old buffer is freed right below
      if (e != hipSuccess) {
        (void)hipFree(q);
        return e;
      }
    }
    if (p) (void)hipFree(p);
    p = q;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Group {
  uint64_t id;
  uint32_t cfg;
  uint32_t task;      // index into the current task table or PM_NONE
  uint64_t task_uid;  // identity of the claimed task across uploads
  MemberList members;  // carve order; BTreeSet order is derived from addr_rank (pm_members.h: small lists need no heap)
  bool dead = false;              // dissolved, not yet removed from the list (compact_groups)
};

// Published assignment table: two buffers, each guarded by a sequence counter (seqlock).  pm_tick writes the
// buffer that is NOT current, then flips `pub_cur`; a reader copies its 32-byte row with relaxed atomic loads
// between two reads of the buffer's counter and retries if the counter moved or was odd.  A buffer is rewritten
// only by the publish AFTER the next one, so a retry needs two publishes within one row copy.  No lock, no
// reference count, no allocation on the read path.  Buffers only grow; replaced allocations are retired, not
// freed, until the engine is destroyed (a reader may still hold the old pointer — its sequence check fails).
struct PubTable {
  std::atomic<uint64_t> seq{0};
  std::atomic<uint64_t*> words{nullptr};  // 4 x u64 per row (pm_assignment is 32 bytes, 8-byte aligned)
  std::atomic<uint32_t> n{0};
  std::atomic<uint32_t> task_shift{0};  // added to every task position read from this buffer: tasks inserted in front
                                        // of the list since it was written (pm_tasks_insert_front) move them all alike
  std::atomic<uint32_t> cleared{0};     // the groups the rows name are gone (pm_reset_groups): every row reads as "no
                                        // group, no task" until the next publish
  size_t cap_rows = 0;
};
static_assert(sizeof(pm_assignment) == 32, "published rows are copied as four 64-bit words");
static_assert(offsetof(pm_assignment, task) == 0 && offsetof(pm_assignment, group_slot) == 4 && offsetof(pm_assignment, group_index) == 8 &&
                  offsetof(pm_assignment, group_size) == 12 && offsetof(pm_assignment, next_worker) == 16 &&
                  offsetof(pm_assignment, group_id) == 24,
              "the words pub_patch clears in place");

}  // namespace pm

using namespace pm;

namespace pm {
struct FormRun;
}

struct pm_engine {
  pm_engine_config cfg{};
  hipStream_t stream = nullptr;        // the stream every kernel and copy of this engine goes to
  hipStream_t stream_owned = nullptr;  // created with the engine; `stream` unless pm_set_stream handed one in
  bool own_stream = true;
  hipEvent_t ev[6]{};
  hipEvent_t kev[6]{};  // kernel-only brackets: compat, carve, sweep
  float k_ms_compat = 0, k_ms_carve = 0, k_ms_sweep = 0;
  bool k_sweep_recorded = false, k_compat_recorded = false;
  uint64_t tick_cand_sum = 0;
  unsigned long long carve_prof[64]{};
  unsigned long long carve_why[24]{};  // CarveStatus::why of the last carve, its batches and void launches, the spatial
                                       // index, the streaming carve's counters
  uint32_t debug_mem_above = 0;  // pm_debug_mem_lists_above
  uint32_t debug_abort_after = 0;  // pm_debug_stream_abort_after
  uint32_t delta_pushes = 0;       // push_groups calls that went up as a delta (since creation)
  uint32_t merge_streamed = 0;     // merge configurations whose selections went through the streaming carve (since creation)
  uint32_t merge_stream_min = 512; // PM_MERGE_STREAM_MIN: compatible solo groups from which a merge configuration does (tests: 8)
  uint32_t prune_mode = 1;       // pm_debug_prune_mode / PM_PRUNE_MODE: CarveArgs::prune_mode
  uint32_t prune_factor = 512;   // PM_PRUNE_FACTOR: CarveArgs::prune_factor (measured crossover)
  uint32_t walk_cap_div = 0;     // PM_WALK_CAP_DIV: CarveArgs::walk_cap_div (0 = the kernels' default)
#ifdef PM_BATCH_LOG
  std::vector<uint32_t> blog;    // the preparations of the last carve (tools/prune_probe.py)
#endif
  std::mutex mu;

  // ---- configuration tables
  std::vector<pm_config_row> cfgs;
  std::vector<pm_gpu_alt_row> alts;
  std::vector<uint32_t> model_bits;
  uint32_t model_rows = 0, model_classes = 0;
  bool cls_check_dirty = true;  // gpu_model_class of some row, or the model table, changed since ensure_compat last checked the column
  uint64_t enabled = ~0ull;
  DevBuf<pm_config_row> d_cfgs;
  DevBuf<pm_gpu_alt_row> d_alts;
  DevBuf<uint32_t> d_model_bits;
  bool have_cfgs = false;

  // ---- worker table (host mirror + HBM columns)
  uint32_t W = 0;
  bool have_workers = false;
  std::vector<uint32_t> h_flags, h_gpu_count, h_gpu_mem, h_gpu_cls, h_cpu_cores, h_ram, h_storage, h_price,
      h_addr_rank;
  std::vector<double> h_lat, h_lon;
  std::vector<uint32_t> h_site;  // equal (lat, lon) bit patterns <=> equal site id
  DevBuf<uint32_t> d_site, d_c_site, d_cc_site, d_seed_prefix, d_seed_slots, d_prep_block_counts, d_prep_counts;
  DevBuf<uint64_t> d_prop, d_prop_send, d_seed_map;
  uint32_t tick_fast_steps = 0;
  DevBuf<uint32_t> d_flags, d_gpu_count, d_gpu_mem, d_gpu_cls, d_cpu_cores, d_ram, d_storage, d_addr_rank;
  DevBuf<double> d_lat, d_lon, d_coslat, d_ux, d_uy, d_uz;
  DevBuf<uint64_t> d_compat;
  bool compat_dirty = true;
  std::vector<uint64_t> h_compat;
  bool h_compat_valid = false;
  bool any_price = false, price_dirty = true;
  std::vector<uint32_t> price_perm;  // workers sorted by (price, index); rebuilt lazily (ensure_price_order)
  // coordinates -> site id (identical bit patterns <=> identical id); located population per site; bit 31 of a
  // worker's site word marks a site shared by >= 2 located workers (a hint for the carve: only those can have
  // same-site neighbours).  The bits are refreshed for the whole column when a site crosses the 1 <-> 2 line.
  struct SiteKeyHash {
    size_t operator()(const std::pair<uint64_t, uint64_t>& k) const {
      return size_t(splitmix64_mix(k.first ^ splitmix64_mix(k.second)));
    }
  };
  std::unordered_map<std::pair<uint64_t, uint64_t>, uint32_t, SiteKeyHash> site_map;
  std::vector<uint32_t> site_pop;
  bool site_bits_dirty = false;
  DevBuf<unsigned char> d_row_stage;  // packed rows of pm_update_workers / pm_append_workers

  // ---- task table
  // The table lives in a fixed-capacity index space [0, t_cap) and is filled from the TOP: the used part is
  // [t_lo, t_cap), ascending index u = get_all_tasks order (created_at desc).  New tasks are the newest, so they
  // go in FRONT — at lower indices — without moving anything (pm_tasks_insert_front); a deleted task leaves a
  // tombstone (mask 0, not live).  A task's index u is therefore a stable HANDLE (what groups store as their
  // claim); its position in the caller's current list — what the ABI reports — is the number of live tasks in
  // front of it (live bitmap + per-word prefix, on the device and lazily on the host).
  uint32_t T = 0;  // live tasks
  uint32_t t_cap = 0, t_lo = 0, t_dead = 0;
  bool have_tasks = false;
  std::vector<uint64_t> h_tmask, h_tuid, h_tlive;  // indexed by u (h_tlive: bitmap words)
  std::vector<int64_t> h_created;
  std::vector<uint32_t> h_tprefix;
  bool h_tprefix_valid = false;
  bool tasks_have_uid = false;
  std::unordered_map<uint64_t, uint32_t> uid_to_u;  // built on the first pm_tasks_delete
  bool uid_map_valid = false;
  uint32_t cfg_app_count[PM_MAX_CONFIGS]{};         // live tasks applicable per configuration (merge path)
  bool cfg_app_valid = false;
  DevBuf<uint64_t> d_tmask, d_tplanes, d_tlive;
  DevBuf<int64_t> d_created;
  DevBuf<uint32_t> d_tprefix, d_first_c, d_count_c, d_tdel;
  bool tplanes_dirty = true, tprefix_dirty = true;

  // ---- groups: the host vector is the source of truth between calls; device arrays mirror it
  std::vector<Group> groups;
  size_t n_dead_groups = 0;  // dissolved entries still in `groups` (slots shift only when they are removed)
  bool flags_dirty = false;  // h_flags changed since the last upload of the column
  std::vector<int32_t> h_group_of;
  uint64_t id_rng = 0;
  bool groups_dirty = true;
  // Delta sync of the device's group state (the churn path: a tick's status changes and new rows touch a few thousand
  // entries of tables that hold a hundred thousand).  While groups_delta_ok the device arrays are the host list as of the
  // last full push_groups / carve, except for what these record: workers whose group was dissolved since (group_of -> -1;
  // the dissolved group stays in both lists as a tombstone — no slot moves), rows appended since (group_of -> -1 for the
  // tail).  Anything else that changes the list clears the flag, and the next push_groups compacts and uploads it whole.
  bool groups_delta_ok = false;
  std::vector<uint32_t> delta_free;
  uint32_t delta_tail_from = PM_NONE;
  // ... and of the flags column: rows whose flags changed since the last upload, while flags_delta_ok
  bool flags_delta_ok = false;
  std::vector<uint32_t> delta_flags;
  uint32_t* h_delta_pin[2] = {nullptr, nullptr};  // pinned staging: [0] freed workers (indices), [1] flags ({index, value} pairs)
  size_t h_delta_cap[2] = {0, 0};
  DevBuf<uint32_t> d_delta[2];
  DevBuf<int32_t> d_group_of;
  DevBuf<uint32_t> d_g_cfg, d_g_n, d_g_off, d_g_task, d_g_task_next, d_members, d_by_rank, d_rank_in_group;
  DevBuf<uint64_t> d_g_id;
  uint32_t d_n_groups = 0, d_n_members = 0;

  // ---- carve scratch
  DevBuf<uint32_t> d_order;
  uint32_t form_rounds_hint = 0;  // validation rounds the last proposal-driven carve needed (0 = unknown)
  bool tick_needs_merge = false;  // the last pm_tick found two or more single-node groups (the merge pass ran): the next one does
                                  // not queue its pair sweep before it has seen the carve's result
  // group life-cycle feed (pm_enable_group_events / pm_drain_group_events)
  bool events_on = false;
  std::vector<pm_group_event> ev_log;
  std::vector<uint32_t> ev_members;
  DevBuf<double> d_c_lat, d_c_lon, d_c_cos, d_cc_lat, d_cc_lon, d_cc_cos, d_c_u[3], d_cc_u[3];
  DevBuf<uint64_t> d_c_compat, d_keys, d_bits;
  DevBuf<uint32_t> d_slot_pos, d_slot_wid;
  DevBuf<CarveStatus> d_status;
  DevBuf<CarveArgs> d_carve_args;   // [2]: one argument block per proposal batch in flight (the second one only
                                    // differs in the per-batch scratch, see CarveSet)
  DevBuf<BatchDesc> d_desc;         // [2]
  DevBuf<uint64_t> d_snap;          // [2][stride] position-bitmap snapshots of the preparations
  // spatial index of a carve's located positions (cell_*_kernel)
  DevBuf<uint32_t> d_cell_cnt, d_cell_start, d_pos_cell, d_pos_rank, d_cs_of_pos, d_cs_slot, d_cs_site;
  DevBuf<double> d_cs_u[3];
  DevBuf<double> d_c_pack, d_cs_pack;  // streaming carve: 32-byte gather records by position / by index entry
  // streaming carve (carve_stream_kernel): per-configuration bitmaps, ticket and row rings, control block, candidate list
  DevBuf<uint64_t> d_cfgbits, d_stream_sq, d_stream_row_lo, d_stream_row_hi, d_stream_trace;
  DevBuf<uint32_t> d_stream_ctl;
  uint32_t stream_seq = 0;       // launches so far: the tags of a launch's tickets start at stream_seq << 25
  uint32_t stream_wgs_env = 0;   // PM_STREAM_WGS: proposer workgroups (0 = by the size of the eligible list)
  uint32_t stream_la_env = 0;    // PM_STREAM_LA: look-ahead cap (0 = the kernel's default)
  uint32_t stream_la_div_env = 0;  // PM_STREAM_LA_DIV: look-ahead divisor (0 = the kernel's default)
  uint32_t stream_row_spins_env = 0;  // PM_STREAM_ROW_SPINS: polls before the validator gives a row up (0 = default)
  uint32_t n_cus = 256;
  uint32_t tick_stream_timeouts = 0, tick_stream_tickets = 0, tick_stream_aborts = 0;
  DevBuf<uint64_t> d_ikeys, d_umask;  // per-task orientation: table of the distinct topology masks, the masks densely
  DevBuf<uint32_t> d_ivals;
  DevBuf<uint32_t> d_m_cfg, d_m_n, d_m_off, d_m_members;  // MERGE batches

  // ---- sweep scratch
  DevBuf<uint64_t> d_sel, d_wplanes, d_sel_perm;
  DevBuf<uint32_t> d_first, d_count, d_rank, d_chosen, d_perm;
  DevBuf<pm_assignment> d_table;
  DevBuf<uint32_t> d_task_col;
  const pm_assignment* h_table = nullptr;  // the snapshot published last (host side of the lock-free look-up)
  uint64_t groups_epoch = 0, pub_groups_epoch = ~0ull;  // slot numbering of e->groups / the one the published rows use
  std::unordered_map<uint64_t, uint32_t> slot_of_id;    // pub_patch: group id -> slot under numbering slot_of_id_epoch
  uint64_t slot_of_id_epoch = ~0ull;
  // pinned staging for the group records a carve appended (absorbed into the host list by absorb_groups)
  uint32_t* h_gstage = nullptr;
  size_t h_gstage_cap = 0;
  const uint32_t *ab_cfg_p = nullptr, *ab_n_p = nullptr, *ab_off_p = nullptr, *ab_mem_p = nullptr;  // the staged records absorb_groups reads
  CarveStatus* h_status = nullptr;  // pinned: carve_finish_kernel mirrors the status block here (streaming carve)
  hipEvent_t ev_groups = nullptr;
  bool absorb_pending = false;
  uint32_t ab_g0 = 0, ab_g1 = 0, ab_m0 = 0, ab_m1 = 0, ab_solo = 0;
  uint32_t* h_gtask_pinned = nullptr;
  size_t h_gtask_cap = 0;
  DevBuf<uint32_t> d_nb_idx;
  DevBuf<long long> d_nb_val;

  PubTable pub[2];
  std::atomic<int> pub_cur{-1};
  std::vector<uint64_t*> pub_retired;
  pm_stats last_stats{};

  // ---- multi-GPU (pm_dist_*): this engine is rank `dist_rank` of `dist_world`, every rank holds the whole swarm
  uint32_t dist_rank = 0, dist_world = 1;
  std::vector<uint8_t> h_shard;        // owner rank of every worker
  std::vector<uint32_t> h_own_rows;    // workers owned by this rank, ascending
  uint32_t dist_cap_t = 0;             // table rows per rank in the exchange buffer (largest shard)
  DevBuf<uint8_t> d_shard;
  DevBuf<uint32_t> d_own_rows, d_xrow; // d_xrow[w] = shard * cap_t + index within the shard
  DevBuf<uint64_t> d_sel_own;
  DevBuf<pm_assignment> d_table_x;     // [world][cap_t] exchange buffer of published rows
  pm::FormRun* form = nullptr;         // carve in progress (stepwise tick)
  int dist_phase = 0;                  // 0 idle, 1 carving, 2 carve done, 3 match queued
  uint32_t dist_n_formed = 0, dist_n_merged = 0;
  uint32_t tick_host_resolved = 0, tick_carve_launches = 0, tick_carve_steps = 0;
  uint32_t tick_props = 0;
  uint64_t tick_prop_keys = 0;
  float k_ms_propose = 0;
  std::vector<hipEvent_t> prop_ev;  // (start, stop) pairs of the proposer launches of the current poll interval
  size_t prop_ev_used = 0;
};

namespace pm {

// ------------------------------------------------------------------------------------------------
// small helpers

template <typename T>
static int32_t upload(DevBuf<T>& d, const T* src, size_t n, hipStream_t s) {
  HIPCHK(d.ensure(n ? n : 1));
  if (n) HIPCHK(hipMemcpyAsync(d.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
  return PM_OK;
}

static void reset_groups_locked(pm_engine* e) {
  e->groups.clear();
  e->n_dead_groups = 0;
  e->absorb_pending = false;  // records of a carve that was never absorbed belong to the old list
  e->h_group_of.assign(e->W, -1);
  e->id_rng = e->cfg.group_id_seed;
  e->groups_dirty = true, e->groups_delta_ok = false;
  // The published rows name slots and ids of the list that just went (and the id stream restarts: the same ids will
  // name other groups): a heartbeat before the next publish is told "no group", and pub_patch resolves nothing
  // against the new list.
  e->groups_epoch++;
  const int cur = e->pub_cur.load(std::memory_order_relaxed);
  if (cur >= 0) {
    PubTable& t = e->pub[cur];
    const uint64_t s0 = t.seq.load(std::memory_order_relaxed);
    t.seq.store(s0 + 1, std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_release);
    t.cleared.store(1u, std::memory_order_relaxed);
    t.seq.store(s0 + 2, std::memory_order_release);
  }
}

// dissolve_group (mod.rs:1423-1487).  A status storm dissolves hundreds of groups per tick; removing each from
// the list right away would renumber every later slot (O(groups + workers) per call).  The entry is only
// marked here — the members are free at once — and compact_groups() removes all marked entries in one pass,
// in list order, before anything looks at slot numbers again.
// one entry of the group life-cycle feed; members in BTreeSet<String> order (address rank), like pm_get_groups
static void log_group_event(pm_engine* e, uint32_t kind, const Group& gr) {
  if (!e->events_on) return;
  pm_group_event ev{};
  ev.group_id = gr.id;
  ev.kind = kind;
  ev.config = gr.cfg;
  ev.member_begin = uint32_t(e->ev_members.size());
  ev.n_members = uint32_t(gr.members.size());
  e->ev_members.insert(e->ev_members.end(), gr.members.begin(), gr.members.end());
  std::sort(e->ev_members.end() - ptrdiff_t(gr.members.size()), e->ev_members.end(),
            [&](uint32_t a, uint32_t b) { return e->h_addr_rank[a] < e->h_addr_rank[b]; });
  e->ev_log.push_back(ev);
}

static void dissolve_locked(pm_engine* e, uint32_t slot) {
  if (slot >= e->groups.size() || e->groups[slot].dead) return;
  log_group_event(e, PM_GROUP_DESTROYED, e->groups[slot]);  // mod.rs:1469-1481
  for (uint32_t w : e->groups[slot].members) e->h_group_of[w] = -1;
  if (e->groups_delta_ok) e->delta_free.insert(e->delta_free.end(), e->groups[slot].members.begin(), e->groups[slot].members.end());
  e->groups[slot].dead = true;
  e->n_dead_groups++;
  e->groups_dirty = true;  // (a delta while groups_delta_ok: see push_groups)
}

static void compact_groups(pm_engine* e) {
  if (!e->n_dead_groups) return;
  e->groups.erase(std::remove_if(e->groups.begin(), e->groups.end(), [](const Group& g) { return g.dead; }),
                  e->groups.end());
  e->groups_epoch++;  // (published group slots no longer index this list)
  std::fill(e->h_group_of.begin(), e->h_group_of.end(), -1);
  for (size_t g = 0; g < e->groups.size(); ++g)
    for (uint32_t w : e->groups[g].members) e->h_group_of[w] = int32_t(g);
  e->n_dead_groups = 0;
  e->groups_dirty = true, e->groups_delta_ok = false;
}

// Mirror the host group list into HBM (packed member pool, slot = index).
// n u32 words of pinned staging + device scratch for a delta (the stream has drained since the last use: a delta goes
// up once per tick, in front of it)
static int32_t delta_stage(pm_engine* e, int which, size_t n) {
  HIPCHK(hipStreamSynchronize(e->stream));  // (nothing in flight may still read the staging: an idle stream answers at once)
  if (e->h_delta_cap[which] < n) {
    if (e->h_delta_pin[which]) (void)hipHostFree(e->h_delta_pin[which]);
    e->h_delta_pin[which] = nullptr;
    e->h_delta_cap[which] = 0;
    const size_t cap = n + n / 2 + 4096;
    HIPCHK(hipHostMalloc((void**)&e->h_delta_pin[which], cap * sizeof(uint32_t)));
    e->h_delta_cap[which] = cap;
  }
  HIPCHK(e->d_delta[which].ensure(n));
  return PM_OK;
}

static int32_t push_groups(pm_engine* e) {
  if (!e->groups_dirty) return PM_OK;
  // ---- the churn path: only dissolutions and new rows since the device last held the list.  The dissolved groups stay
  // where they are, in both lists, as tombstones (no worker names them any more; a slot number — what group_of and the
  // published rows hold — keeps its meaning); their workers and the new rows read "no group" on the device after one
  // scatter and one fill.  Compaction waits until a good part of the list is dead or the arrays run short of room for
  // what the next carve may append.
  const size_t room_g = std::min(e->d_g_cfg.cap, std::min(e->d_g_task.cap, std::min(e->d_g_task_next.cap, e->d_g_id.cap)));
  const size_t room_m = std::min(e->d_members.cap, std::min(e->d_by_rank.cap, e->d_rank_in_group.cap));
  if (e->groups_delta_ok && e->n_dead_groups * 4 <= e->groups.size() + 256 && size_t(e->d_n_groups) + e->W <= room_g &&
      size_t(e->d_n_members) + e->W <= room_m && e->d_n_groups == e->groups.size()) {
    HIPCHK(e->d_group_of.grow_keep(std::max<size_t>(e->W, 1), e->delta_tail_from == PM_NONE ? e->W : e->delta_tail_from, e->stream));
    if (e->delta_tail_from != PM_NONE && e->delta_tail_from < e->W)
      HIPCHK(hipMemsetAsync(e->d_group_of.p + e->delta_tail_from, 0xFF, size_t(e->W - e->delta_tail_from) * 4, e->stream));
    if (!e->delta_free.empty()) {
      const size_t n = e->delta_free.size();
      int32_t rc = delta_stage(e, 0, n);
      if (rc) return rc;
      std::memcpy(e->h_delta_pin[0], e->delta_free.data(), n * 4);
      HIPCHK(hipMemcpyAsync(e->d_delta[0].p, e->h_delta_pin[0], n * 4, hipMemcpyHostToDevice, e->stream));
      launch_scatter_const(reinterpret_cast<uint32_t*>(e->d_group_of.p), e->d_delta[0].p, uint32_t(n), 0xFFFFFFFFu, e->stream);
      HIPCHK(hipGetLastError());
    }
    e->delta_free.clear();
    e->delta_tail_from = PM_NONE;
    e->groups_dirty = false;
    e->delta_pushes++;
    return PM_OK;
  }
  compact_groups(e);
  if (!e->groups_dirty) return PM_OK;
  const size_t G = e->groups.size();
  std::vector<uint32_t> g_cfg(G), g_n(G), g_off(G), g_task(G), members;
  std::vector<uint64_t> g_id(G);
  members.reserve(e->W);
  for (size_t g = 0; g < G; ++g) {
    const Group& gr = e->groups[g];
    g_cfg[g] = gr.cfg;
    g_n[g] = uint32_t(gr.members.size());
    g_off[g] = uint32_t(members.size());
    g_task[g] = gr.task;
    g_id[g] = gr.id;
    members.insert(members.end(), gr.members.begin(), gr.members.end());
  }
  // (twice the table: room for a carve's appends on top of a list that carries tombstones — see the delta path above)
  const size_t capG = std::max<size_t>(size_t(2) * e->W, 1), capM = std::max<size_t>(size_t(2) * e->W, 1);
  HIPCHK(e->d_g_cfg.ensure(capG));
  HIPCHK(e->d_g_n.ensure(capG));
  HIPCHK(e->d_g_off.ensure(capG));
  HIPCHK(e->d_g_task.ensure(capG));
  HIPCHK(e->d_g_task_next.ensure(capG));
  HIPCHK(e->d_g_id.ensure(capG));
  HIPCHK(e->d_members.ensure(capM));
  HIPCHK(e->d_by_rank.ensure(capM));
  HIPCHK(e->d_rank_in_group.ensure(capM));
  HIPCHK(e->d_group_of.ensure(capM));
  if (G) {
    HIPCHK(hipMemcpyAsync(e->d_g_cfg.p, g_cfg.data(), G * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_g_n.p, g_n.data(), G * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_g_off.p, g_off.data(), G * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_g_task.p, g_task.data(), G * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_g_id.p, g_id.data(), G * 8, hipMemcpyHostToDevice, e->stream));
  }
  if (!members.empty())
    HIPCHK(hipMemcpyAsync(e->d_members.p, members.data(), members.size() * 4, hipMemcpyHostToDevice, e->stream));
  if (e->W) {
    if (G == 0)  // no group holds a worker (a cold match behind pm_reset_groups): every row is -1, nothing is staged
      HIPCHK(hipMemsetAsync(e->d_group_of.p, 0xFF, size_t(e->W) * 4, e->stream));
    else
      HIPCHK(hipMemcpyAsync(e->d_group_of.p, e->h_group_of.data(), size_t(e->W) * 4, hipMemcpyHostToDevice, e->stream));
  }
  if (G) HIPCHK(hipStreamSynchronize(e->stream));  // the staging vectors die here
  e->d_n_groups = uint32_t(G);
  e->d_n_members = uint32_t(members.size());
  e->groups_dirty = false;
  e->groups_delta_ok = true;  // the device holds the list: dissolutions and new rows from here on are deltas
  e->delta_free.clear();
  e->delta_tail_from = PM_NONE;
  return PM_OK;
}

// status changes only touch the host copy of the flags column; the column goes up once before its next use
static int32_t sync_flags(pm_engine* e) {
  if (!e->flags_dirty || !e->have_workers) return PM_OK;
  if (e->flags_delta_ok && e->delta_flags.size() * 8 <= size_t(e->W)) {  // a few rows: {index, value} pairs and one scatter
    const size_t n = e->delta_flags.size();
    if (n) {
      int32_t rc = delta_stage(e, 1, 2 * n);
      if (rc) return rc;
      for (size_t k = 0; k < n; ++k) {
        e->h_delta_pin[1][2 * k] = e->delta_flags[k];
        e->h_delta_pin[1][2 * k + 1] = e->h_flags[e->delta_flags[k]];
      }
      HIPCHK(hipMemcpyAsync(e->d_delta[1].p, e->h_delta_pin[1], n * 8, hipMemcpyHostToDevice, e->stream));
      launch_scatter_pairs(e->d_flags.p, e->d_delta[1].p, uint32_t(n), e->stream);
      HIPCHK(hipGetLastError());
    }
  } else {
    HIPCHK(hipMemcpyAsync(e->d_flags.p, e->h_flags.data(), size_t(e->W) * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));  // pageable source
  }
  e->delta_flags.clear();
  e->flags_delta_ok = true;
  e->flags_dirty = false;
  return PM_OK;
}

static int32_t ensure_compat(pm_engine* e) {
  if (!e->have_cfgs || !e->have_workers) return set_error(PM_ESTATE, "configs and workers must be uploaded first");
  {
    int32_t rcf = sync_flags(e);
    if (rcf) return rcf;
  }
  if (!e->compat_dirty) return PM_OK;
  for (const pm_gpu_alt_row& a : e->alts)
    if ((a.flags & PM_G_MODEL) && a.model_row >= e->model_rows)
      return set_error(PM_ESTATE, "a GPU alternative names a model row but pm_set_model_table was not called");
  // (once per change of the column or the table, not once per tick: a pass over every row on the host is 100 us at
  // 100,000 workers, a tenth of a churn tick's match)
  if (e->model_rows && e->cls_check_dirty)
    for (uint32_t w = 0; w < e->W; ++w)
      if ((e->h_flags[w] & PM_W_GPU_MODEL) && e->h_gpu_cls[w] >= e->model_classes)
        return set_error(PM_ERANGE, "worker gpu_model_class outside the model table");
  e->cls_check_dirty = false;
  HIPCHK(e->d_compat.ensure(std::max<size_t>(e->W, 1)));
  CompatArgs a{};
  a.W = e->W;
  a.n_cfgs = uint32_t(e->cfgs.size());
  a.model_words = (e->model_classes + 31u) / 32u;
  a.flags = e->d_flags.p;
  a.gpu_count = e->d_gpu_count.p;
  a.gpu_mem = e->d_gpu_mem.p;
  a.gpu_cls = e->d_gpu_cls.p;
  a.cpu_cores = e->d_cpu_cores.p;
  a.ram = e->d_ram.p;
  a.storage = e->d_storage.p;
  a.cfgs = e->d_cfgs.p;
  a.alts = e->d_alts.p;
  a.model_bits = e->d_model_bits.p;
  a.compat = e->d_compat.p;
  HIPCHK(hipEventRecord(e->kev[0], e->stream));
  launch_compat(a, e->stream);
  HIPCHK(hipEventRecord(e->kev[1], e->stream));
  HIPCHK(hipGetLastError());
  e->k_compat_recorded = true;
  e->compat_dirty = false;
  e->h_compat_valid = false;
  return PM_OK;
}

static int32_t pull_compat(pm_engine* e) {
  if (e->h_compat_valid) return PM_OK;
  e->h_compat.resize(e->W);
  if (e->W) {
    HIPCHK(hipMemcpyAsync(e->h_compat.data(), e->d_compat.p, size_t(e->W) * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  e->h_compat_valid = true;
  return PM_OK;
}

// Haversine exactly as the reference evaluates it (mod.rs:218-231) with the host libm.  Used only by
// the exact-resolve path for steps the GPU certificate could not prove (see carve_kernel).
static double host_distance(double lat1, double lon1, double lat2, double lon2) {
  const double R = 6371.0;
  const double lat1r = lat1 * PM_RAD, lat2r = lat2 * PM_RAD;
  const double dlat = (lat2 - lat1) * PM_RAD, dlon = (lon2 - lon1) * PM_RAD;
  const double s1 = std::sin(dlat / 2.0), s2 = std::sin(dlon / 2.0);
  const double a = s1 * s1 + std::cos(lat1r) * std::cos(lat2r) * (s2 * s2);
  const double c = 2.0 * std::atan2(std::sqrt(a), std::sqrt(1.0 - a));
  return R * c;
}

struct CarvePlan {
  std::vector<uint32_t> avail;  // configuration indices in carve order
};

static size_t carve_lds_bytes(uint32_t /*stride_words*/, bool* in_lds) {
  *in_lds = true;  // the kernel decides per configuration whether its candidate list fits in LDS
  return PM_CARVE_LDS_BYTES;
}

static int32_t fill_carve_args(pm_engine* e, CarveArgs* a, uint32_t mode, uint32_t n_order, bool stream = false) {
  const size_t cap = std::max<size_t>(e->W, 1);
  HIPCHK(e->d_order.ensure(cap));
  HIPCHK(e->d_c_lat.ensure(cap));
  HIPCHK(e->d_c_lon.ensure(cap));
  HIPCHK(e->d_c_cos.ensure(cap));
  HIPCHK(e->d_cc_lat.ensure(cap));
  HIPCHK(e->d_cc_lon.ensure(cap));
  HIPCHK(e->d_cc_cos.ensure(cap));
  for (int k = 0; k < 3; ++k) {
    HIPCHK(e->d_c_u[k].ensure(cap));
    HIPCHK(e->d_cc_u[k].ensure(cap));
  }
  HIPCHK(e->d_c_compat.ensure(cap));
  HIPCHK(e->d_keys.ensure(cap));
  HIPCHK(e->d_slot_pos.ensure(cap));
  HIPCHK(e->d_slot_wid.ensure(cap));
  HIPCHK(e->d_c_site.ensure(cap));
  HIPCHK(e->d_cc_site.ensure(cap));
  {  // proposal rows: at most PM_PROP_MAX_SEEDS + 63 seeds per batch, dealt round-robin over the ranks
    const size_t world = e->dist_world;
    const size_t rows_pr = (size_t(PM_PROP_MAX_SEEDS) + 64 + world - 1) / world;
    HIPCHK(e->d_prop.ensure(rows_pr * world * PM_PROP_ROW));
    if (world > 1) HIPCHK(e->d_prop_send.ensure(rows_pr * PM_PROP_ROW));
    HIPCHK(e->d_seed_map.ensure((cap + 63) / 64 + 64));
    HIPCHK(e->d_seed_prefix.ensure((cap + 63) / 64 + 64));
    HIPCHK(e->d_seed_slots.ensure(size_t(PM_PROP_MAX_SEEDS) + 128));
    HIPCHK(e->d_prep_block_counts.ensure(((cap + 255) / 256 + 1) * PM_MAX_CONFIGS));
    HIPCHK(e->d_prep_counts.ensure(PM_MAX_CONFIGS + 8));
  }
  HIPCHK(e->d_status.ensure(1));
  HIPCHK(e->d_carve_args.ensure(2));
  HIPCHK(e->d_desc.ensure(2));
  const uint32_t stride = uint32_t((cap + 63) / 64);
  HIPCHK(e->d_bits.ensure(size_t(stride) * 4));
  HIPCHK(e->d_snap.ensure(size_t(stride) * 2));
  if (stream) {
    HIPCHK(e->d_c_pack.ensure(cap * 4));
    HIPCHK(e->d_cs_pack.ensure(cap * 4));
    HIPCHK(e->d_cfgbits.ensure(size_t(stride) * PM_MAX_CONFIGS));
    HIPCHK(e->d_stream_ctl.ensure(PM_STREAM_CTL_WORDS));
    if (!e->d_stream_sq.p) {  // (tags never repeat within 127 launches; the rings are cleared when the counter wraps)
      HIPCHK(e->d_stream_sq.ensure(PM_STREAM_SQ));
      HIPCHK(e->d_stream_row_lo.ensure(size_t(PM_STREAM_RQ) * 64));
      HIPCHK(e->d_stream_row_hi.ensure(size_t(PM_STREAM_RQ) * 64));
      e->stream_seq = 0;
    }
  }
  if (mode == CARVE_MODE_FORM && e->prune_mode && e->cfg.proximity_enabled) {
    if (!e->d_cell_cnt.p) {  // (the scan leaves the counts zero behind it: cleared once)
      HIPCHK(e->d_cell_cnt.ensure(PM_CELL_TABLE));
      HIPCHK(hipMemsetAsync(e->d_cell_cnt.p, 0, e->d_cell_cnt.cap * sizeof(uint32_t), e->stream));
    }
    if (!e->d_cell_start.p) {  // + the scan's block sums and its ticket (zero between scans)
      HIPCHK(e->d_cell_start.ensure(PM_CELL_TABLE + 384));
      HIPCHK(hipMemsetAsync(e->d_cell_start.p, 0, e->d_cell_start.cap * sizeof(uint32_t), e->stream));
    }
    HIPCHK(e->d_pos_cell.ensure(cap));
    HIPCHK(e->d_pos_rank.ensure(cap));
    HIPCHK(e->d_cs_of_pos.ensure(cap));
    HIPCHK(e->d_cs_slot.ensure(cap));
    HIPCHK(e->d_cs_site.ensure(cap));
    for (auto& u : e->d_cs_u) HIPCHK(u.ensure(cap));
  }
  std::memset(a, 0, sizeof(*a));
  a->mode = mode;
  a->W = e->W;
  a->proximity = e->cfg.proximity_enabled;
  a->debug_uncertain_every = e->cfg.debug_uncertain_every;
  a->wflags = e->d_flags.p;
  a->lat = e->d_lat.p;
  a->lon = e->d_lon.p;
  a->coslat = e->d_coslat.p;
  a->ux = e->d_ux.p;
  a->uy = e->d_uy.p;
  a->uz = e->d_uz.p;
  a->compat = e->d_compat.p;
  a->group_of = e->d_group_of.p;
  a->order = e->d_order.p;
  a->n_order = n_order;
  a->c_lat = e->d_c_lat.p;
  a->c_lon = e->d_c_lon.p;
  a->c_cos = e->d_c_cos.p;
  a->c_ux = e->d_c_u[0].p;
  a->c_uy = e->d_c_u[1].p;
  a->c_uz = e->d_c_u[2].p;
  a->c_compat = e->d_c_compat.p;
  a->alive_g = e->d_bits.p;
  a->loc_g = e->d_bits.p + stride;
  a->cc_lat = e->d_cc_lat.p;
  a->cc_lon = e->d_cc_lon.p;
  a->cc_cos = e->d_cc_cos.p;
  a->cc_ux = e->d_cc_u[0].p;
  a->cc_uy = e->d_cc_u[1].p;
  a->cc_uz = e->d_cc_u[2].p;
  a->keys = e->d_keys.p;
  a->slot_pos = e->d_slot_pos.p;
  a->slot_wid = e->d_slot_wid.p;
  a->site = e->d_site.p;
  a->c_site = e->d_c_site.p;
  a->cc_site = e->d_cc_site.p;
  a->prop = e->d_prop.p;
  a->prop_send = e->dist_world > 1 ? e->d_prop_send.p : e->d_prop.p;
  a->seed_map = e->d_seed_map.p;
  a->seed_prefix = e->d_seed_prefix.p;
  a->seed_slots = e->d_seed_slots.p;
  a->dist_rank = e->dist_rank;
  a->dist_world = e->dist_world;
  a->count_keys = e->cfg.time_proposer ? 1u : 0u;
  a->prep_block_counts = e->d_prep_block_counts.p;
  a->prep_counts = e->d_prep_counts.p;
  a->bits_scratch = e->d_bits.p + size_t(stride) * 2;
  a->bits_stride = stride;
  a->status = e->d_status.p;
  a->desc = e->d_desc.p;
  a->alive_snap = e->d_snap.p;
  a->debug_mem_above = e->debug_mem_above;
  if (mode == CARVE_MODE_FORM && e->prune_mode && e->cfg.proximity_enabled) {
    a->prune_mode = e->prune_mode;
    a->prune_factor = e->prune_factor;
    a->walk_cap_div = e->walk_cap_div;
    a->cell_cnt = e->d_cell_cnt.p;
    a->cell_start = e->d_cell_start.p;
    a->pos_cell = e->d_pos_cell.p;
    a->pos_rank = e->d_pos_rank.p;
    a->cs_of_pos = e->d_cs_of_pos.p;
    a->cs_slot = e->d_cs_slot.p;
    a->cs_site = e->d_cs_site.p;
    a->cs_ux = e->d_cs_u[0].p;
    a->cs_uy = e->d_cs_u[1].p;
    a->cs_uz = e->d_cs_u[2].p;
  }
  if (stream) {
    // slot == position: the per-slot columns ARE the per-position columns, a slot's worker is the eligible list's
    // entry; d_bits = {published candidate bitmap, loc bitmap, the validator's master bitmap}
    a->stream = 1;
    a->cc_lat = a->c_lat;
    a->cc_lon = a->c_lon;
    a->cc_cos = a->c_cos;
    a->cc_ux = a->c_ux;
    a->cc_uy = a->c_uy;
    a->cc_uz = a->c_uz;
    a->cc_site = a->c_site;
    a->slot_wid = a->order;
    a->bits_scratch = e->d_bits.p;
    a->loc_g = e->d_bits.p + stride;
    a->alive_g = e->d_bits.p + size_t(stride) * 2;
    a->cfgbits = e->d_cfgbits.p;
    a->c_pack = e->d_c_pack.p;
    a->cs_pack = e->d_cs_pack.p;
    a->stream_sq = (unsigned long long*)e->d_stream_sq.p;
    a->stream_row_lo = (unsigned long long*)e->d_stream_row_lo.p;
    a->stream_row_hi = (unsigned long long*)e->d_stream_row_hi.p;
    a->stream_ctl = e->d_stream_ctl.p;
#if defined(PM_CARVE_PROF) || defined(PM_ROW_REC)
    HIPCHK(e->d_stream_trace.ensure(size_t(PM_STREAM_TRACE_CAP) * 2));
    a->stream_trace = (unsigned long long*)e->d_stream_trace.p;
#endif
#ifdef PM_ROW_REC  // (a measuring build: one record of time stamps per ticket, pm_debug_row_records)
    HIPCHK(hipMemsetAsync(e->d_stream_trace.p, 0, size_t(PM_STREAM_TRACE_CAP) * 16, e->stream));
#endif
    a->stream_la = e->stream_la_env;
    a->stream_la_div = e->stream_la_div_env;
    a->stream_row_spins = e->stream_row_spins_env;
    a->debug_abort_after = e->debug_abort_after;
  }
  return PM_OK;
}

// One exact carve step on the host for configuration `cfg` (FORM mode): the same rule as
// mod.rs:511-551 with glibc distances and a stable sort.  Appends the group to the device arrays.
static int32_t host_resolve_form_step(pm_engine* e, uint32_t cfg, CarveStatus* st) {
  int32_t rc = pull_compat(e);
  if (rc) return rc;
  std::vector<int32_t> group_of(e->W);
  if (e->W) {
    HIPCHK(hipMemcpyAsync(group_of.data(), e->d_group_of.p, size_t(e->W) * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  const pm_config_row& c = e->cfgs[cfg];
  std::vector<uint32_t> compat;
  size_t total_available = 0;
  for (uint32_t w = 0; w < e->W; ++w) {
    const uint32_t f = e->h_flags[w];
    if (!((f & PM_W_HEALTHY) && (f & PM_W_HAS_P2P) && group_of[w] < 0)) continue;
    ++total_available;
    if ((e->h_compat[w] >> cfg) & 1ull) compat.push_back(w);
  }
  if (total_available < c.min_group_size || compat.size() < c.min_group_size || compat.empty())
    return set_error(PM_ESTATE, "host resolve: nothing to carve (state mismatch)");
  uint32_t seed = compat[0];
  bool seed_loc = false;
  for (uint32_t w : compat)
    if (e->h_flags[w] & PM_W_HAS_LOC) {
      seed = w;
      seed_loc = true;
      break;
    }
  std::vector<uint32_t> rest;
  for (uint32_t w : compat)
    if (w != seed) rest.push_back(w);
  if (seed_loc && e->cfg.proximity_enabled) {
    std::vector<double> d(e->W, 0.0);
    for (uint32_t w : rest)
      d[w] = (e->h_flags[w] & PM_W_HAS_LOC) ? host_distance(e->h_lat[seed], e->h_lon[seed], e->h_lat[w], e->h_lon[w])
                                            : 1.7976931348623157e308;
    std::stable_sort(rest.begin(), rest.end(), [&](uint32_t a, uint32_t b) { return d[a] < d[b]; });
  }
  std::vector<uint32_t> members{seed};
  for (uint32_t w : rest) {
    if (members.size() >= c.max_group_size) break;
    members.push_back(w);
  }
  if (members.size() < c.min_group_size) return set_error(PM_ESTATE, "host resolve: group below min size");
  const uint32_t g = st->n_groups, off = st->n_members, n = uint32_t(members.size());
  if (g >= e->d_g_cfg.cap || off + n > e->d_members.cap) return set_error(PM_ENOMEM, "group arrays overflow");
  HIPCHK(hipMemcpyAsync(e->d_g_cfg.p + g, &cfg, 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_g_n.p + g, &n, 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_g_off.p + g, &off, 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_members.p + off, members.data(), size_t(n) * 4, hipMemcpyHostToDevice, e->stream));
  for (uint32_t w : members) group_of[w] = int32_t(g);
  HIPCHK(hipMemcpyAsync(e->d_group_of.p, group_of.data(), size_t(e->W) * 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  st->n_groups += 1;
  st->n_members += n;
  st->n_solo += n == 1 ? 1u : 0u;
  st->steps_total += 1;
  return PM_OK;
}

// Run the persistent carve kernel until it reports DONE, settling UNCERTAIN steps on the host.
// Append the group records of the last carve to the host list (ids from the generate_group_id stream).
static int32_t absorb_groups(pm_engine* e) {
  if (!e->absorb_pending) return PM_OK;
  HIPCHK(hipEventSynchronize(e->ev_groups));
  const uint32_t g0 = e->ab_g0, ng = e->ab_g1 - e->ab_g0, m0 = e->ab_m0;
  const uint32_t *g_cfg = e->ab_cfg_p, *g_n = e->ab_n_p, *g_off = e->ab_off_p, *members = e->ab_mem_p;
  // (geometric: an exact reserve moves the whole list — 30,000 records — at every tick that appends to a full vector)
  if (e->groups.capacity() < e->groups.size() + ng) e->groups.reserve(std::max(e->groups.size() + ng, e->groups.capacity() * 2));
  for (uint32_t k = 0; k < ng; ++k) {
    Group gr;
    gr.id = splitmix64_next(&e->id_rng);
    gr.cfg = g_cfg[k];
    gr.task = PM_NONE;
    gr.task_uid = 0;
    gr.members.assign(members + (g_off[k] - m0), members + (g_off[k] - m0) + g_n[k]);
    for (uint32_t w : gr.members) e->h_group_of[w] = int32_t(g0 + k);
    log_group_event(e, PM_GROUP_CREATED, gr);  // mod.rs:612-625
    e->groups.push_back(std::move(gr));
  }
  e->absorb_pending = false;
  return PM_OK;
}

// Every entry point that reads the host group list or h_group_of first takes in the records of a carve whose
// absorption was deferred (pm_tick defers it behind the pair sweep; a tick that failed half-way leaves it pending).
#define ABSORB_PENDING(e)                 \
  do {                                    \
    int32_t rc_abs_ = absorb_groups(e);   \
    if (rc_abs_) return rc_abs_;          \
  } while (0)

// PM_TRACE_HOST=1: host-side timestamps (us since the first mark) on stderr, to find where a match waits
static void host_mark(const char* what) {
  static const bool on = [] { const char* v = getenv("PM_TRACE_HOST"); return v && *v == '1'; }();
  if (!on) return;
  static const auto t0 = std::chrono::steady_clock::now();
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, "[pm host] %10.1f us  %s\n", us, what);
}

// One try_form_new_groups run (mod.rs:478-628) as a resumable sequence, so the same code serves the
// single-GPU tick (launches queued blindly, status read when they are done) and the stepwise multi-GPU tick
// (one status read per proposal batch, the all-gather of the batch's rows issued by the caller in between).
struct FormRun {
  CarveArgs a;
  BatchDesc desc[2] = {};   // the batch descriptors as of the last poll
  CarveStatus st;
  std::vector<uint32_t> avail;
  uint32_t g0 = 0, m0 = 0, start_ci = 0;
  size_t lds = 0;
  bool use_props = false;
  bool nothing = false;  // no configuration / no worker: nothing to carve
  uint32_t n_bound = 0;  // rows outside any group when the carve starts (>= the eligible list): sizes the prep grids
  uint32_t n_elig_hint = 0;  // the eligible ones among them, by the host mirror
  bool stream = false;       // one streaming launch (carve_stream_kernel) instead of the batch pipeline
  bool single_call = true;   // run_form drives the whole carve (the stepwise multi-GPU tick exchanges rows per batch)
  uint32_t stream_wgs = 0;   // proposer workgroups of the streaming launch
  bool local_carve = false;  // the multi-GPU tick: this rank carves the whole pool itself (every rank does, and ends with the same groups)
  bool rearmed = false;      // the carve took more than its first launch sequence (host-resolved step, aborted streaming launch)
  uint32_t stage_cap = 0;    // streaming carve: entries of each staging array carve_finish_kernel fills in pinned host memory
};

static int32_t launch_propose_timed(pm_engine* e, const CarveArgs* d_args, uint32_t n_bound, hipStream_t s);

static int32_t form_setup_args(pm_engine* e, FormRun* r);

// fresh: the first launch sequence of a carve — the status block (RUNNING, the group and member counts so far, zero
// otherwise) is initialised by the first kernel of the sequence instead of by a copy in front of it
static int32_t form_queue_init(pm_engine* e, FormRun* r, bool fresh = false) {
  fresh = fresh && r->use_props;  // (the proposal-free carve's one launch reads the status it is handed)
  if (!fresh) HIPCHK(hipMemcpyAsync(e->d_status.p, &r->st, sizeof(r->st), hipMemcpyHostToDevice, e->stream));
  if (r->use_props) {
    if (!r->stream) HIPCHK(hipMemsetAsync(e->d_desc.p, 0, 2 * sizeof(BatchDesc), e->stream));  // (the batch pipeline's descriptors)
    // the ordered eligible list, and the spatial index of its positions when there are enough of them to matter
    const uint32_t index_min = !r->a.prune_mode ? 0u : r->a.prune_mode >= 2u ? 1u : PM_CELL_AUTO_N;
    if (r->stream) {
      // every launch tags its tickets and rows from a range of its own (a launch re-armed behind a host-resolved step
      // must not take the rows of the one before it for its own); the rings are cleared when the counter wraps
      if (e->stream_seq == 0 || e->stream_seq >= 127) {
        HIPCHK(hipMemsetAsync(e->d_stream_sq.p, 0, size_t(PM_STREAM_SQ) * 8, e->stream));
        HIPCHK(hipMemsetAsync(e->d_stream_row_lo.p, 0, size_t(PM_STREAM_RQ) * 64 * 8, e->stream));
        HIPCHK(hipMemsetAsync(e->d_stream_row_hi.p, 0, size_t(PM_STREAM_RQ) * 64 * 8, e->stream));
        e->stream_seq = 0;
      }
      e->stream_seq += 1;
      const uint32_t tag0 = e->stream_seq << 25;
      if (r->a.stream_tag0 != tag0) {  // (the first sequence of a carve carries its tag in the argument block it uploads)
        r->a.stream_tag0 = tag0;
        HIPCHK(hipMemcpyAsync(&e->d_carve_args.p->stream_tag0, &r->a.stream_tag0, sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
      }
    }
    e->tick_carve_launches += launch_carve_elig(e->d_carve_args.p, e->W, r->n_bound, r->n_elig_hint >= index_min ? index_min : 0u,
                                                r->start_ci, fresh, r->st.n_groups, r->st.n_members, e->stream);
    if (r->stream) {  // everything else in one launch (+ the pass that turns positions into worker ids)
      HIPCHK(launch_carve_stream(e->d_carve_args.p, r->start_ci, r->stream_wgs, e->stream));
      e->tick_carve_launches += 2;
      return PM_OK;
    }
    e->tick_carve_launches += launch_carve_prep(e->d_carve_args.p, r->n_bound, e->stream);  // the first candidate list
    HIPCHK(hipGetLastError());
  } else
    HIPCHK(launch_carve(e->d_carve_args.p, CARVE_F_INIT | CARVE_F_RUN | CARVE_F_ALL, r->start_ci, r->lds, e->stream));
  e->tick_carve_launches++;
  return PM_OK;
}

// local_carve: a rank of the multi-GPU tick — the carve is replicated, not exchanged (see pm_dist_tick_begin): the launches
// are those of one GPU whatever dist_world says
static int32_t form_begin(pm_engine* e, FormRun* r, bool allow_pipeline, bool local_carve = false) {
  int32_t rc = absorb_groups(e);  // a match that failed half-way may have left the last carve unabsorbed
  if (rc) return rc;
  rc = ensure_compat(e);
  if (rc) return rc;
  rc = push_groups(e);
  if (rc) return rc;
  available_order(e->cfgs.data(), uint32_t(e->cfgs.size()), e->enabled, &r->avail);
  r->g0 = e->d_n_groups;
  r->m0 = e->d_n_members;
  r->start_ci = 0;
  r->nothing = r->avail.empty() || e->W == 0;
  if (r->nothing) return PM_OK;
  // the eligible list is a subset of the rows no group holds (host mirror, current after absorb_groups): the
  // per-round preparation kernels are sized by that, not by the table (an incremental tick on a standing swarm
  // prepares lists of a few thousand positions out of a hundred thousand rows)
  // ... and how many of those the kernels will find eligible (mod.rs:492-497), as far as the host mirror knows: only
  // used to decide whether the spatial index is worth its four launches (a standing swarm's tick: a few thousand).
  // (Both counts in ONE branch-free pass the compiler vectorises: at 100,000 workers two passes with branches were
  // 150 - 250 us of a churn tick's 1.3 ms.)
  if (e->h_group_of.size() == e->W && e->h_flags.size() == e->W) {
    const int32_t* const gof = e->h_group_of.data();
    const uint32_t* const fl = e->h_flags.data();
    const uint32_t need = PM_W_HEALTHY | PM_W_HAS_P2P;
    uint32_t nb = 0, ne = 0;
    for (uint32_t w = 0; w < e->W; ++w) {
      const uint32_t un = uint32_t(gof[w]) >> 31;  // 1 where no group holds the row
      nb += un;
      ne += un & uint32_t((fl[w] & need) == need);
    }
    r->n_bound = nb;
    r->n_elig_hint = ne;
  } else {
    r->n_bound = uint32_t(std::count_if(e->h_group_of.begin(), e->h_group_of.end(), [](int32_t g) { return g < 0; }));
    r->n_elig_hint = r->n_bound;
  }
  if (e->h_group_of.size() != e->W) r->n_bound = r->n_elig_hint = e->W;
  if (r->n_bound == 0) r->n_bound = 1;
  r->st = CarveStatus{};
  r->st.state = CARVE_STATE_RUNNING;
  r->st.n_groups = r->g0;
  r->st.n_members = r->m0;
  r->single_call = allow_pipeline;
  r->local_carve = local_carve;
  r->use_props = e->cfg.carve_variant != 1 && e->cfg.proximity_enabled;
  // The streaming carve: one engine, one call, positions that fit the validator's LDS bitmaps.  Everything else (the
  // stepwise multi-GPU tick, swarms beyond 262,144 unassigned rows) goes through the batch pipeline.
  r->stream = r->single_call && r->use_props && e->cfg.carve_variant == 0 && (e->dist_world == 1 || r->local_carve) &&
              r->n_bound <= PM_CARVE_BIG_SLOTS &&
              !e->debug_mem_above;  // (the test hook for the all-in-HBM lists is the batch pipeline's)
  host_mark("form: begin (host mirrors counted)");
  rc = form_setup_args(e, r);
  if (rc) return rc;
  host_mark("form: arguments queued");
  HIPCHK(hipEventRecord(e->kev[2], e->stream));
  return form_queue_init(e, r, /*fresh=*/true);  // prepares the first candidate list (all of it when there are no proposals)
}

// The argument block(s) of a carve, filled and uploaded (again, when a streaming launch gave up and the batch
// pipeline takes over).
static int32_t form_setup_args(pm_engine* e, FormRun* r) {
  CarveArgs& a = r->a;
  int32_t rc = fill_carve_args(e, &a, CARVE_MODE_FORM, 0, r->stream);
  if (rc) return rc;
  if (r->stream) {
    // proposer workgroups: enough waves to cover a row's latency at the chain's pace, by the size of the list
    uint32_t wgs = e->stream_wgs_env ? e->stream_wgs_env : r->n_elig_hint / 64u + 48u;
    const uint32_t max_wgs = e->n_cus > 8u ? e->n_cus - 4u : 4u;
    r->stream_wgs = std::max(1u, std::min(wgs, max_wgs));
    // the tag of the launch sequence form_queue_init is about to queue (it advances the counter: see there)
    a.stream_tag0 = ((e->stream_seq == 0 || e->stream_seq >= 127) ? 1u : e->stream_seq + 1u) << 25;
    // carve_finish_kernel: group ids, empty task words, and the host's copy of records and status
    a.id_state = e->id_rng;
    a.id_g0 = r->g0;
    a.stage_m0 = r->m0;
    a.g_id_out = (unsigned long long*)e->d_g_id.p;
    a.g_task_out = e->d_g_task.p;
    r->stage_cap = std::max<uint32_t>(r->n_bound, 1u);  // (a group holds at least one of the rows no group holds yet)
    const size_t need = size_t(4) * r->stage_cap;
    if (e->h_gstage_cap < need) {
      HIPCHK(hipStreamSynchronize(e->stream));  // (nothing in flight may still write the old buffer)
      if (e->h_gstage) (void)hipHostFree(e->h_gstage);
      e->h_gstage = nullptr;
      e->h_gstage_cap = 0;
      const size_t cap = std::max<size_t>(need, size_t(4) * std::max<uint32_t>(e->W, 1));
      HIPCHK(hipHostMalloc((void**)&e->h_gstage, cap * sizeof(uint32_t)));
      e->h_gstage_cap = cap;
    }
    if (!e->h_status) HIPCHK(hipHostMalloc((void**)&e->h_status, sizeof(CarveStatus)));
    a.stage_cap_g = a.stage_cap_m = r->stage_cap;
    a.stage_cfg = e->h_gstage;
    a.stage_n = e->h_gstage + r->stage_cap;
    a.stage_off = e->h_gstage + size_t(2) * r->stage_cap;
    a.stage_mem = e->h_gstage + size_t(3) * r->stage_cap;
    a.h_status = e->h_status;
    e->h_status->state = 0xFFFFFFFFu;  // (not yet written by this carve)
  }
  if (r->local_carve) {  // (rows are made here for every seed: no segment of another rank's to wait for)
    a.dist_rank = 0;
    a.dist_world = 1;
    a.prop_send = a.prop;
  }
  a.n_avail = uint32_t(r->avail.size());
  for (size_t i = 0; i < r->avail.size(); ++i) {
    a.avail_cfg[i] = r->avail[i];
    a.min_size[i] = e->cfgs[r->avail[i]].min_group_size;
    a.max_size[i] = e->cfgs[r->avail[i]].max_group_size;
  }
  a.g_cfg = e->d_g_cfg.p;
  a.g_n = e->d_g_n.p;
  a.g_off = e->d_g_off.p;
  a.members = e->d_members.p;
  a.cap_groups = uint32_t(std::min<size_t>(e->d_g_cfg.cap, 0xFFFFFFFFu));
  a.cap_members = uint32_t(std::min<size_t>(e->d_members.cap, 0xFFFFFFFFu));
  bool in_lds;
  r->lds = carve_lds_bytes(a.bits_stride, &in_lds);
  HIPCHK(hipMemcpyAsync(e->d_carve_args.p, &a, sizeof(a), hipMemcpyHostToDevice, e->stream));
  return PM_OK;
}

// the proposer launch, bracketed by its own hipEvents when pm_engine_config.time_proposer asks for the split
static int32_t launch_propose_timed(pm_engine* e, const CarveArgs* d_args, uint32_t n_bound, hipStream_t s) {
  if (!e->cfg.time_proposer) {
    launch_carve_propose(d_args, n_bound, s);
    return PM_OK;
  }
  while (e->prop_ev.size() < e->prop_ev_used + 2) {
    hipEvent_t x = nullptr;
    HIPCHK(hipEventCreate(&x));
    e->prop_ev.push_back(x);
  }
  HIPCHK(hipEventRecord(e->prop_ev[e->prop_ev_used], s));
  launch_carve_propose(d_args, n_bound, s);
  HIPCHK(hipEventRecord(e->prop_ev[e->prop_ev_used + 1], s));
  e->prop_ev_used += 2;
  return PM_OK;
}

// (propose, validate) pairs: one per configuration plus one per re-proposal round; launches queued behind a
// finished carve return immediately
static int32_t form_queue_pairs(pm_engine* e, FormRun* r, uint32_t count) {
  for (uint32_t k = 0; k < count; ++k) {
    int32_t rc = launch_propose_timed(e, e->d_carve_args.p, r->n_bound, e->stream);
    if (rc) return rc;
    HIPCHK(launch_carve(e->d_carve_args.p, CARVE_F_RUN | CARVE_F_PROPS | CARVE_F_EXTPREP, 0, r->lds, e->stream));
    e->tick_carve_launches += 2u + launch_carve_prep(e->d_carve_args.p, r->n_bound, e->stream);  // the next candidate list
  }
  return PM_OK;
}

// Wait for everything queued so far and read the carve's status.  An UNCERTAIN step is settled on the host
// (glibc distances) and the carve re-armed from that configuration; the caller sees RUNNING then.
static int32_t form_poll(pm_engine* e, FormRun* r) {
  for (;;) {
    HIPCHK(hipEventRecord(e->kev[3], e->stream));
    HIPCHK(hipMemcpyAsync(&r->st, e->d_status.p, sizeof(r->st), hipMemcpyDeviceToHost, e->stream));
    // (the batch descriptors: on the engine's stream with the status — a blocking copy on the null stream was a second
    // driver round trip per poll; the streaming carve has none)
    const bool want_desc = r->use_props && !r->stream;
    if (want_desc) HIPCHK(hipMemcpyAsync(r->desc, e->d_desc.p, sizeof(r->desc), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, e->kev[2], e->kev[3]));
      e->k_ms_carve += ms;
      HIPCHK(hipEventRecord(e->kev[2], e->stream));
      for (size_t k = 0; k + 1 < e->prop_ev_used; k += 2) {
        HIPCHK(hipEventElapsedTime(&ms, e->prop_ev[k], e->prop_ev[k + 1]));
        e->k_ms_propose += ms;
      }
      e->prop_ev_used = 0;
    }
    if (r->st.state == CARVE_STATE_OVERFLOW) return set_error(PM_ENOMEM, "carve: group arrays overflow");
    if (r->st.state == CARVE_STATE_ABORTED) {
      // A bounded wait inside a launch gave up (a GPU shared with other work can keep the proposers of a streaming
      // launch from running beside its validator).  What was committed stands — every commit is exact — and the carve
      // continues from there on the batch pipeline, whose launches depend on nothing running beside them.
      if (!r->stream) return set_error(PM_ENODEV, "carve: a hand-shake inside the validator timed out");
      e->tick_stream_aborts++;
      r->rearmed = true;
      r->stream = false;
      int32_t rca = form_setup_args(e, r);
      if (rca) return rca;
      r->start_ci = r->st.stop_ci;
      r->st.state = CARVE_STATE_RUNNING;
      rca = form_queue_init(e, r);
      if (rca) return rca;
      return PM_OK;  // (RUNNING: the caller queues propose / validate rounds)
    }
    if (r->st.state != CARVE_STATE_UNCERTAIN) return PM_OK;
    int32_t rc = host_resolve_form_step(e, r->avail[r->st.stop_ci], &r->st);
    if (rc) return rc;
    e->tick_host_resolved++;
    r->rearmed = true;
    r->start_ci = r->st.stop_ci;
    r->st.state = CARVE_STATE_RUNNING;
    rc = form_queue_init(e, r);
    if (rc) return rc;
    // read the status again: the INIT launch prepares the next list (and, without proposals, runs on to the end
    // or the next stop) — the caller sizes the batch's exchange from what it reports
  }
}

// have_event: ev_groups has been recorded behind the carve already (and waited for)
static int32_t form_finish(pm_engine* e, FormRun* r, uint32_t* n_formed, bool defer_absorb, bool have_event = false) {
  if (n_formed) *n_formed = 0;
  if (r->nothing) return PM_OK;
  const CarveStatus& st = r->st;
  e->tick_fast_steps += st.fast_steps;
  e->tick_carve_steps += st.steps_total;
  e->tick_cand_sum += st.cand_sum;
  e->tick_props += r->stream && !st.n_props ? st.stream_tickets : st.n_props;
  e->tick_prop_keys += st.prop_keys;
  e->tick_stream_timeouts += st.stream_timeouts;
  e->tick_stream_tickets += st.stream_tickets;
  std::memcpy(e->carve_prof, st.prof, sizeof(st.prof));
  for (int k = 0; k < 8; ++k) e->carve_why[k] = st.why[k];
  e->carve_why[8] = st.n_batches;
  e->carve_why[9] = st.n_void;
  e->carve_why[10] = st.pruned_batches;
  e->carve_why[11] = st.prune_fallbacks;
  e->carve_why[12] = st.cell_g;
  e->carve_why[13] = st.n_indexed;
  e->carve_why[14] = r->stream ? 1u : 0u;
  e->carve_why[15] = st.stream_tickets;
  e->carve_why[16] = st.stream_timeouts;
  e->carve_why[17] = st.stream_switches;
  e->carve_why[18] = st.stream_listed;
  e->carve_why[19] = r->stream_wgs;
  e->carve_why[20] = e->tick_stream_aborts;
  e->carve_why[21] = st.slow_steps;
  e->carve_why[22] = st.stream_pre_used;
  e->carve_why[23] = st.stream_pre_lost;
#ifdef PM_BATCH_LOG
  e->blog.assign(st.blog, st.blog + 3 * std::min<uint32_t>(st.blog_n, 512u));
#endif

  // The new group records stay in HBM for the match; their ids (generate_group_id stream) and empty task
  // words are filled in on the device, and a copy travels to pinned host memory for absorb_groups().
  const uint32_t g0 = r->g0, m0 = r->m0, g1 = st.n_groups, m1 = st.n_members;
  if (g1 > g0) {
    const uint32_t ng = g1 - g0, nm = m1 - m0;
    // One streaming launch sequence did the whole carve: carve_finish_kernel has written the records' host copy, the
    // ids and the empty task words already (the caller waited for the stream since).  Otherwise — a re-armed carve, the
    // batch pipeline — they are copied and filled in here.
    const bool staged = r->stream && !r->rearmed && ng <= r->stage_cap && nm <= r->stage_cap;
    if (staged) {
      e->ab_cfg_p = e->h_gstage;
      e->ab_n_p = e->h_gstage + r->stage_cap;
      e->ab_off_p = e->h_gstage + size_t(2) * r->stage_cap;
      e->ab_mem_p = e->h_gstage + size_t(3) * r->stage_cap;
      if (!have_event) HIPCHK(hipEventRecord(e->ev_groups, e->stream));
    } else {
      const size_t need = size_t(3) * ng + nm;
      if (e->h_gstage_cap < need) {
        if (e->h_gstage) (void)hipHostFree(e->h_gstage);
        e->h_gstage = nullptr;
        e->h_gstage_cap = 0;
        const size_t cap = std::max<size_t>(need, size_t(4) * std::max<uint32_t>(e->W, 1));
        HIPCHK(hipHostMalloc((void**)&e->h_gstage, cap * sizeof(uint32_t)));
        e->h_gstage_cap = cap;
      }
      uint32_t* st_cfg = e->h_gstage;
      HIPCHK(hipMemcpyAsync(st_cfg, e->d_g_cfg.p + g0, size_t(ng) * 4, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(st_cfg + ng, e->d_g_n.p + g0, size_t(ng) * 4, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(st_cfg + 2 * size_t(ng), e->d_g_off.p + g0, size_t(ng) * 4, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(st_cfg + 3 * size_t(ng), e->d_members.p + m0, size_t(nm) * 4, hipMemcpyDeviceToHost, e->stream));
      e->ab_cfg_p = st_cfg;
      e->ab_n_p = st_cfg + ng;
      e->ab_off_p = st_cfg + 2 * size_t(ng);
      e->ab_mem_p = st_cfg + 3 * size_t(ng);
      HIPCHK(hipEventRecord(e->ev_groups, e->stream));
      host_mark("form: group copies queued");
      launch_group_ids(e->d_g_id.p + g0, e->d_g_task.p + g0, ng, e->id_rng, e->stream);
      HIPCHK(hipGetLastError());
    }
    e->absorb_pending = true;
    e->ab_g0 = g0;
    e->ab_g1 = g1;
    e->ab_m0 = m0;
    e->ab_m1 = m1;
    e->ab_solo = st.n_solo;
    e->d_n_groups = g1;
    e->d_n_members = m1;
    if (!defer_absorb) {
      int32_t rc = absorb_groups(e);
      if (rc) return rc;
    }
  }
  if (n_formed) *n_formed = g1 - g0;
  return PM_OK;
}

// The part of a single-call carve behind form_begin: wait for what was queued, queue more rounds if the carve needs
// them, take the result in.  (Split from the beginning so that pm_tick_many can start the carves of several engines
// before it waits for the first.)
static int32_t run_form_wait(pm_engine* e, FormRun& r) {
  int32_t rc = PM_OK;
  if (!r.nothing) {
    // (propose, validate, prepare) rounds are queued blindly and the ones behind a finished carve return at once —
    // at ~5 us per empty launch.  The first queue is sized by what the previous carve of this engine needed (a
    // periodic match changes little from tick to tick); one configuration = at least one round otherwise.
    uint32_t batch = e->form_rounds_hint ? e->form_rounds_hint + 1u : r.a.n_avail + 3u;
    for (uint32_t spins = 0;; ++spins) {
      // every poll either ends the carve or follows launches that formed at least one group or moved on to the
      // next configuration: far fewer rounds than this, or the device side is stuck — fail instead of hanging
      if (spins > 4096u + e->W / 8u) return set_error(PM_ENODEV, "carve made no progress");
      if (r.use_props && !r.stream) {
        rc = form_queue_pairs(e, &r, batch);
        if (rc) return rc;
      }
      host_mark("form: carve queued");
      rc = form_poll(e, &r);
      if (rc) return rc;
      host_mark("form: status back");
      if (r.st.state == CARVE_STATE_DONE) break;
      if (r.st.state != CARVE_STATE_RUNNING || !r.use_props) return set_error(PM_ENODEV, "carve kernel did not complete");
      if (r.stream) return set_error(PM_ENODEV, "streaming carve did not complete");  // (it ends DONE, or re-armed above)
      batch = 8u;  // more re-proposal rounds than were queued, or re-armed after a host-resolved step
    }
  }
  if (!r.nothing && r.use_props && !r.stream) e->form_rounds_hint = r.st.n_batches + r.st.n_void;
  return PM_OK;
}
static int32_t run_form_rest(pm_engine* e, FormRun& r, uint32_t* n_formed, bool defer_absorb) {
  int32_t rc = run_form_wait(e, r);
  if (rc) return rc;
  return form_finish(e, &r, n_formed, defer_absorb);
}

static int32_t run_form(pm_engine* e, uint32_t* n_formed, bool defer_absorb = false) {
  FormRun r;
  int32_t rc = form_begin(e, &r, /*allow_pipeline=*/true);
  if (rc) return rc;
  return run_form_rest(e, r, n_formed, defer_absorb);
}

// ------------------------------------------------------------------------------------------------
// pair sweep plumbing

static int32_t ensure_task_planes(pm_engine* e) {
  if (!e->have_tasks) return set_error(PM_ESTATE, "tasks must be uploaded first");
  const uint32_t n_planes = uint32_t(e->cfgs.size());
  const uint32_t stride = e->t_cap / 64u;
  if (e->tplanes_dirty) {  // whole table (upload, capacity change, new configurations); deltas patch the planes
    HIPCHK(e->d_tplanes.ensure(std::max<size_t>(size_t(stride) * (n_planes + 1), 1)));  // + the OR plane
    launch_build_planes(e->d_tmask.p, e->t_cap, e->t_lo, e->t_cap, stride, n_planes, e->d_tplanes.p, e->stream);
    HIPCHK(hipGetLastError());
    e->tplanes_dirty = false;
  }
  if (e->tprefix_dirty) {
    launch_task_prefix(e->d_tlive.p, e->t_lo / 64u, stride, e->d_tprefix.p, e->stream);
    HIPCHK(hipGetLastError());
    e->tprefix_dirty = false;
  }
  return PM_OK;
}

// handle -> position in the caller's current get_all_tasks list (host side; the device does the same in
// claim_publish_kernel)
static uint32_t task_position(pm_engine* e, uint32_t u) {
  if (u == PM_NONE || u < e->t_lo || u >= e->t_cap) return PM_NONE;
  if (!e->h_tprefix_valid) {
    const uint32_t stride = e->t_cap / 64u;
    e->h_tprefix.assign(stride, 0);
    uint32_t acc = 0;
    for (uint32_t j = e->t_lo / 64u; j < stride; ++j) {
      e->h_tprefix[j] = acc;
      acc += uint32_t(__builtin_popcountll(e->h_tlive[j]));
    }
    e->h_tprefix_valid = true;
  }
  const uint64_t w = e->h_tlive[u >> 6];
  if (!((w >> (u & 63u)) & 1ull)) return PM_NONE;
  return e->h_tprefix[u >> 6] + uint32_t(__builtin_popcountll(w & ((1ull << (u & 63u)) - 1ull)));
}

static int32_t ensure_sweep_outputs(pm_engine* e, uint32_t R) {
  HIPCHK(e->d_first.ensure(std::max<uint32_t>(R, 1)));
  HIPCHK(e->d_count.ensure(std::max<uint32_t>(R, 1)));
  HIPCHK(e->d_rank.ensure(std::max<uint32_t>(R, 1)));
  HIPCHK(e->d_chosen.ensure(std::max<uint32_t>(R, 1)));
  return PM_OK;
}

// rank-th applicable task for one configuration bit (merge path: find_best_task_for_group,
// mod.rs:1122-1189).  Returns PM_NONE if no task is applicable.
static int32_t pick_task_for_config(pm_engine* e, uint32_t cfg, uint64_t group_id, uint32_t* task_out) {
  *task_out = PM_NONE;
  if (!e->have_tasks || e->T == 0) return PM_OK;
  // The applicable list depends only on the configuration bit.  Per-configuration counts are kept per version
  // of the task table (one pass), so a merge costs one early-exit scan to the chosen task — not two passes over
  // a million rows.
  if (!e->cfg_app_valid) {
    std::memset(e->cfg_app_count, 0, sizeof(e->cfg_app_count));
    for (uint32_t u = e->t_lo; u < e->t_cap; ++u) {
      uint64_t m = e->h_tmask[u];  // 0 for tombstones
      while (m) {
        e->cfg_app_count[__builtin_ctzll(m)]++;
        m &= m - 1;
      }
    }
    e->cfg_app_valid = true;
  }
  const uint32_t n_app = e->cfg_app_count[cfg];
  if (!n_app) return PM_OK;
  const uint64_t bit = 1ull << cfg;
  uint32_t r = 0;
  if (e->cfg.chooser == PM_CHOOSE_SEEDED) r = uint32_t(splitmix64_mix(e->cfg.chooser_seed ^ group_id) % n_app);
  for (uint32_t u = e->t_lo; u < e->t_cap; ++u)
    if (e->h_tmask[u] & bit) {
      if (r == 0) {
        *task_out = u;  // the handle
        return PM_OK;
      }
      --r;
    }
  return PM_OK;
}

// The pair sweep + chooser + claim for every worker — or, in a multi-GPU tick (`dist`), for the workers this
// rank owns, whose rows go packed into this rank's segment of the exchange buffer (pm_dist_match_begin).
// G_ub: the task words of that many groups are carried over (0 = the device's group count: every caller but the tick that
// queues the match before it knows how many groups the carve in front of it formed)
static int32_t run_match(pm_engine* e, bool want_count, std::vector<uint32_t>* count_out, bool dist = false, size_t G_ub = 0) {
  if (!e->have_cfgs || !e->have_workers || !e->have_tasks)
    return set_error(PM_ESTATE, "configs, workers and tasks must be uploaded first");
  int32_t rc = push_groups(e);
  if (rc) return rc;
  const int variant = int(e->cfg.sweep_variant);
  rc = ensure_task_planes(e);  // (also the live prefix the published positions come from)
  if (rc) return rc;
  const uint32_t R = dist ? uint32_t(e->h_own_rows.size()) : e->W;
  const uint32_t* rows = dist ? e->d_own_rows.p : nullptr;
  rc = ensure_sweep_outputs(e, R);
  if (rc) return rc;
  HIPCHK(e->d_sel.ensure(std::max<uint32_t>(e->W, 1)));
  HIPCHK(e->d_table.ensure(std::max<uint32_t>(e->W, 1)));
  HIPCHK(e->d_task_col.ensure(std::max<uint32_t>(e->W, 1)));
  if (e->W == 0) return PM_OK;
  const uint32_t n_planes = uint32_t(e->cfgs.size());

  host_mark("match: selector launch");
  launch_worker_selector(e->d_group_of.p, e->d_g_cfg.p, R, rows, e->d_sel.p, e->stream);
  HIPCHK(hipEventRecord(e->kev[4], e->stream));
  const uint32_t t_stride = e->t_cap / 64u;
  launch_pair_sweep(variant, e->d_sel.p, R, e->d_tmask.p, e->d_tplanes.p, e->t_lo, e->t_cap, t_stride, n_planes,
                    e->d_first.p, e->d_count.p, e->stream);
  HIPCHK(hipEventRecord(e->kev[5], e->stream));
  e->k_sweep_recorded = true;
  const uint32_t* chosen = e->d_first.p;  // PM_CHOOSE_FIRST: the first applicable task
  if (e->cfg.chooser == PM_CHOOSE_SEEDED) {
    launch_chooser_rank(e->d_group_of.p, e->d_g_id.p, e->d_count.p, R, rows, e->cfg.chooser_seed, e->d_rank.p,
                        e->stream);
    launch_pair_select(variant, e->d_sel.p, R, e->d_tmask.p, e->d_tplanes.p, e->t_lo, e->t_cap, t_stride, n_planes,
                       e->d_rank.p, e->d_chosen.p, e->stream);
    chosen = e->d_chosen.p;
  }
  launch_group_rank(e->d_group_of.p, e->d_g_n.p, e->d_g_off.p, e->d_members.p, e->d_addr_rank.p, e->W,
                    e->d_rank_in_group.p, e->d_by_rank.p, e->stream);
  const size_t G = G_ub ? G_ub : e->d_n_groups;  // == groups.size() once the last carve is absorbed
  if (G) HIPCHK(hipMemcpyAsync(e->d_g_task_next.p, e->d_g_task.p, G * 4, hipMemcpyDeviceToDevice, e->stream));
  ClaimArgs c{};
  c.R = R;
  c.rows = rows;
  c.group_of = e->d_group_of.p;
  c.g_n = e->d_g_n.p;
  c.g_off = e->d_g_off.p;
  c.g_task = e->d_g_task.p;
  c.g_id = e->d_g_id.p;
  c.g_task_next = e->d_g_task_next.p;
  c.chosen = chosen;
  c.rank_in_group = e->d_rank_in_group.p;
  c.by_rank = e->d_by_rank.p;
  c.t_live = e->d_tlive.p;
  c.t_prefix = e->d_tprefix.p;
  c.table = dist ? e->d_table_x.p + size_t(e->dist_rank) * e->dist_cap_t : e->d_table.p;
  c.task_col = e->d_task_col.p;
  launch_claim_publish(c, e->stream);
  HIPCHK(hipGetLastError());
  if (want_count && count_out) {
    count_out->resize(R);
    HIPCHK(hipMemcpyAsync(count_out->data(), e->d_count.p, size_t(R) * 4, hipMemcpyDeviceToHost, e->stream));
  }
  return PM_OK;
}

// The rows of a snapshot buffer, grown if need be (buffers only grow; a replaced one is retired, not freed: a reader
// may still hold it)
static int32_t pub_buffer(pm_engine* e, PubTable& t, uint32_t rows, uint64_t** out) {
  uint64_t* words = t.words.load(std::memory_order_relaxed);
  if (t.cap_rows < rows || !words) {
    const size_t cap = std::max<size_t>(size_t(rows) + rows / 8 + 64, 64);
    uint64_t* nw = nullptr;
    if (hipHostMalloc((void**)&nw, cap * 32) != hipSuccess || !nw)
      return set_error(PM_ENOMEM, "out of pinned host memory for the published table");
    std::memset(nw, 0, cap * 32);
    const uint64_t s0 = t.seq.load(std::memory_order_relaxed);
    t.seq.store(s0 + 1, std::memory_order_relaxed);  // odd while the pointer changes
    std::atomic_thread_fence(std::memory_order_release);
    if (words) e->pub_retired.push_back(words);
    words = nw;
    t.cap_rows = cap;
    t.words.store(nw, std::memory_order_relaxed);
    t.seq.store(s0 + 2, std::memory_order_release);
  }
  *out = words;
  return PM_OK;
}

// Between two ticks the published table says what the last tick computed, while a task delta changes the POSITION of
// every task behind it in the caller's list and a deleted task or a dead worker takes a group with it
// (mod.rs:1259-1288, status_update_impl.rs:17-29).  The reference binds a group to its task by id
// (get_current_group_task, scheduler_impl.rs:33): a heartbeat between two management-loop runs sees the same task
// under its new position, and nothing for a worker whose group has just been dissolved.  pub_patch applies exactly
// that to the published rows, in place (the sequence counter of the buffer is odd meanwhile: readers retry), from the
// host's group list: a row of a dissolved group becomes the row of a worker in no group, a row of a standing group gets
// the current position of the task the group holds.  `only` = just these workers (a dissolved group's members);
// nullptr = every row (a task delta).  No device work; the device-side task column waits for the next tick.
static void pub_patch(pm_engine* e, const std::vector<uint32_t>* only) {
  const int cur = e->pub_cur.load(std::memory_order_relaxed);
  if (cur < 0) return;  // nothing published
  // the published rows name their group by the slot it had then; if the list has been compacted since, by its id
  const bool by_id = e->pub_groups_epoch != e->groups_epoch;
  // (id -> slot, built once per numbering of the list — a status storm calls this once per worker: slots move only when
  // the epoch does, a group dissolved since is found and seen dead, and the published rows, written before the epoch
  // moved, name no group appended since)
  std::unordered_map<uint64_t, uint32_t>& slot_of_id = e->slot_of_id;
  if (by_id && e->slot_of_id_epoch != e->groups_epoch) {
    slot_of_id.clear();
    slot_of_id.reserve(e->groups.size() * 2);
    for (size_t g = 0; g < e->groups.size(); ++g) slot_of_id.emplace(e->groups[g].id, uint32_t(g));
    e->slot_of_id_epoch = e->groups_epoch;
  }
  PubTable& t = e->pub[cur];
  if (t.cleared.load(std::memory_order_relaxed)) return;  // (pm_reset_groups: no row of this buffer names a group of this list)
  const uint32_t n = t.n.load(std::memory_order_relaxed);
  pm_assignment* rows = reinterpret_cast<pm_assignment*>(t.words.load(std::memory_order_relaxed));
  if (!rows || !n) return;
  // one row as it reads now; keep_shift: what readers of the buffer the row goes to add to a position
  auto patched = [&](pm_assignment a, uint32_t keep_shift) -> pm_assignment {
    if (a.group_slot == PM_NONE) return a;
    uint32_t slot = a.group_slot;
    if (by_id) {
      const auto it = slot_of_id.find(a.group_id);
      slot = it == slot_of_id.end() ? PM_NONE : it->second;
    }
    const bool gone = slot >= e->groups.size() || e->groups[slot].dead;
    if (gone) {
      a.task = PM_NONE;
      a.group_slot = PM_NONE;
      a.group_index = 0;
      a.group_size = 0;
      a.next_worker = PM_NONE;
      a.group_id = 0;
    } else {
      const uint32_t h = e->groups[slot].task;
      a.task = h == PM_NONE ? PM_NONE : task_position(e, h) - keep_shift;  // (readers add the buffer's shift)
    }
    return a;
  };
  auto store_row = [](pm_assignment* dst, const pm_assignment& a) {
    uint64_t v[4];
    std::memcpy(v, &a, sizeof(a));
    uint64_t* d = reinterpret_cast<uint64_t*>(dst);
    for (int k = 0; k < 4; ++k) __atomic_store_n(&d[k], v[k], __ATOMIC_RELAXED);
  };
  if (!only) {
    // Every row: written to the buffer that is NOT current, which then becomes the current one — exactly what a
    // publish does — so a heartbeat thread never waits for a pass over W rows (in place, the buffer's sequence counter
    // would be odd for the whole pass and every look-up would spin on it).
    const int nx = cur ^ 1;
    PubTable& d = e->pub[nx];
    uint64_t* dw = nullptr;
    if (pub_buffer(e, d, n, &dw) == PM_OK && dw) {
      pm_assignment* drows = reinterpret_cast<pm_assignment*>(dw);
      const uint64_t d0 = d.seq.load(std::memory_order_relaxed);
      d.seq.store(d0 + 1, std::memory_order_relaxed);  // odd: being written
      std::atomic_thread_fence(std::memory_order_release);
      for (uint32_t w = 0; w < n; ++w) store_row(&drows[w], patched(rows[w], 0u));
      d.n.store(n, std::memory_order_relaxed);
      d.task_shift.store(0, std::memory_order_relaxed);  // (the rows hold current positions)
      d.cleared.store(0, std::memory_order_relaxed);
      d.seq.store(d0 + 2, std::memory_order_release);  // even: stable
      e->pub_cur.store(nx, std::memory_order_release);
      e->h_table = drows;
      return;
    }
    // (no pinned memory for the other buffer: in place — look-ups retry until the pass is through)
  }
  const uint64_t s0 = t.seq.load(std::memory_order_relaxed);
  t.seq.store(s0 + 1, std::memory_order_relaxed);  // odd: being written
  std::atomic_thread_fence(std::memory_order_release);
  if (only) {  // the members of groups dissolved just now: rows in place
    // A worker is in one group at a time, and `only` holds members of groups that are gone: a row of theirs that names a
    // group at all names one that is gone (the one just dissolved, or an older one the table still carried) — it reads "no
    // group" from now on, whatever it named.  No look at the group list (by slot, or by id through a hash map once the
    // list has been compacted): for the 5,000 workers a status sweep of a thousand deaths frees, those look-ups — two
    // cache misses a row — were 260 of the call's 300 us (PM_HOST_MARKS_STATUS build, tools/churn_probe.py).
    for (uint32_t w : *only) {
      if (w >= n) continue;
      uint64_t* d = reinterpret_cast<uint64_t*>(&rows[w]);
      if (uint32_t(__atomic_load_n(&d[0], __ATOMIC_RELAXED) >> 32) == PM_NONE) continue;  // (no group: as it is)
      const uint64_t w2 = __atomic_load_n(&d[2], __ATOMIC_RELAXED);                        // next_worker | padding
      __atomic_store_n(&d[0], (uint64_t(PM_NONE) << 32) | PM_NONE, __ATOMIC_RELAXED);      // task, group_slot
      __atomic_store_n(&d[1], 0ull, __ATOMIC_RELAXED);                                      // group_index, group_size
      __atomic_store_n(&d[2], (w2 & 0xFFFFFFFF00000000ull) | PM_NONE, __ATOMIC_RELAXED);
      __atomic_store_n(&d[3], 0ull, __ATOMIC_RELAXED);                                      // group_id
    }
  } else {
    for (uint32_t w = 0; w < n; ++w) store_row(&rows[w], patched(rows[w], 0u));
    t.task_shift.store(0, std::memory_order_relaxed);  // (the rows hold current positions again)
  }
  t.seq.store(s0 + 2, std::memory_order_release);  // even: stable
}

// n tasks were inserted in front of the list: every published position moves back by n — one word instead of a pass
// over the rows (readers add the buffer's shift to the position they read; the seqlock makes row and shift one unit)
static void pub_shift_tasks(pm_engine* e, uint32_t n_new) {
  const int cur = e->pub_cur.load(std::memory_order_relaxed);
  if (cur < 0) return;
  PubTable& t = e->pub[cur];
  const uint64_t s0 = t.seq.load(std::memory_order_relaxed);
  t.seq.store(s0 + 1, std::memory_order_relaxed);
  std::atomic_thread_fence(std::memory_order_release);
  t.task_shift.store(t.task_shift.load(std::memory_order_relaxed) + n_new, std::memory_order_relaxed);
  t.seq.store(s0 + 2, std::memory_order_release);
}

// D2H of the assignment table + the group task words; the table lands directly in the snapshot buffer that is
// not current (pinned host memory, written by the copy engine between the odd and the even mark of its sequence
// counter — see PubTable), which then becomes the published one.  In two halves, so that a tick can queue the copies
// right behind the claim and build its host copy of the new groups (absorb_groups) while they travel: publish_begin
// marks the buffer and queues the copies, publish_end waits for them, flips the buffers and takes the groups' task
// words in (the host list must be complete by then: it has d_n_groups entries once the last carve is absorbed).
struct PubRun {
  int nx = 0;
  uint64_t s0 = 0;
  uint64_t* words = nullptr;
  size_t G = 0;
};

static void publish_abandon(pm_engine* e, const PubRun& pr) {
  (void)hipStreamSynchronize(e->stream);  // (nothing may still be writing the buffer when it reads as stable again)
  PubTable& t = e->pub[pr.nx];
  t.n.store(0, std::memory_order_relaxed);  // contents undefined: nothing to look up in this buffer
  t.seq.store(pr.s0 + 2, std::memory_order_release);
}

// G: the group count the task words are copied for — groups.size(), or, in front of absorb_groups, the device's count
// (d_n_groups == groups.size() once the last carve is absorbed)
static int32_t publish_begin(pm_engine* e, PubRun* pr, size_t G) {
  if (e->h_gtask_cap < G) {
    if (e->h_gtask_pinned) (void)hipHostFree(e->h_gtask_pinned);
    e->h_gtask_pinned = nullptr;
    e->h_gtask_cap = 0;
    const size_t cap = std::max<size_t>(G, std::max<uint32_t>(e->W, 1));
    HIPCHK(hipHostMalloc((void**)&e->h_gtask_pinned, cap * sizeof(uint32_t)));
    e->h_gtask_cap = cap;
  }
  const int cur = e->pub_cur.load(std::memory_order_relaxed);
  const int nx = cur < 0 ? 0 : (cur ^ 1);
  PubTable& t = e->pub[nx];
  uint64_t* words = nullptr;
  {
    int32_t rcb = pub_buffer(e, t, e->W, &words);
    if (rcb) return rcb;
  }
  const uint64_t s0 = t.seq.load(std::memory_order_relaxed);
  t.seq.store(s0 + 1, std::memory_order_relaxed);  // odd: being written
  std::atomic_thread_fence(std::memory_order_release);
  pr->nx = nx;
  pr->s0 = s0;
  pr->words = words;
  pr->G = G;
  hipError_t herr = hipSuccess;
  if (e->W) herr = hipMemcpyAsync(words, e->d_table.p, sizeof(pm_assignment) * e->W, hipMemcpyDeviceToHost, e->stream);
  if (herr == hipSuccess && G)
    herr = hipMemcpyAsync(e->h_gtask_pinned, e->d_g_task_next.p, G * 4, hipMemcpyDeviceToHost, e->stream);
  if (herr != hipSuccess) {
    publish_abandon(e, *pr);
    HIPCHK(herr);
  }
  return PM_OK;
}

static int32_t publish_end(pm_engine* e, const PubRun& pr) {
  PubTable& t = e->pub[pr.nx];
  const hipError_t herr = hipStreamSynchronize(e->stream);
  if (herr != hipSuccess) {
    publish_abandon(e, pr);
    HIPCHK(herr);
  }
  const size_t G = e->groups.size();
  if (G != pr.G) {  // (the copies were sized by the device's group count in front of absorb_groups: see publish_begin)
    publish_abandon(e, pr);
    return set_error(PM_ESTATE, "publish: the host group list and the device group arrays disagree");
  }
  const uint32_t* g_task = e->h_gtask_pinned;
  t.n.store(e->W, std::memory_order_relaxed);
  t.task_shift.store(0, std::memory_order_relaxed);
  t.cleared.store(0, std::memory_order_relaxed);
  t.seq.store(pr.s0 + 2, std::memory_order_release);  // even: stable
  e->pub_cur.store(pr.nx, std::memory_order_release);
  e->h_table = reinterpret_cast<const pm_assignment*>(pr.words);
  e->pub_groups_epoch = e->groups_epoch;
  for (size_t g = 0; g < G; ++g) {
    if (e->groups[g].dead) continue;  // (a tombstone: its device record is stale)
    e->groups[g].task = g_task[g];
    e->groups[g].task_uid = g_task[g] == PM_NONE ? 0 : (e->tasks_have_uid ? e->h_tuid[g_task[g]] : task_position(e, g_task[g]));
  }
  std::swap(e->d_g_task, e->d_g_task_next);
  return PM_OK;
}

static int32_t publish(pm_engine* e) {
  PubRun pr;
  int32_t rc = publish_begin(e, &pr, e->groups.size());
  if (rc) return rc;
  return publish_end(e, pr);
}

// ------------------------------------------------------------------------------------------------
// merge (try_merge_solo_groups, mod.rs:631-971)

// get_all_groups order: by id formatted "{:x}" compared as strings (mod.rs:1040).
static bool hex_id_less(uint64_t a, uint64_t b) {
  auto len = [](uint64_t x) {
    int n = 1;
    while (x >>= 4) ++n;
    return n;
  };
  const int la = len(a), lb = len(b);
  const uint64_t aa = a << (4 * (16 - la)), bb = b << (4 * (16 - lb));  // left-align the digit strings
  if (aa != bb) return aa < bb;
  return la < lb;  // a proper prefix sorts first
}

// Exact host version of one attempt_group_merge selection (mod.rs:752-860) over `rem` (workers of the
// remaining compatible solo groups in get_all_groups order).
static void host_merge_select(pm_engine* e, const std::vector<uint32_t>& rem, const pm_config_row& c,
                              std::vector<uint32_t>* batch) {
  batch->clear();
  if (e->cfg.proximity_enabled) {
    size_t seed_pos = rem.size();
    for (size_t i = 0; i < rem.size(); ++i)
      if (e->h_flags[rem[i]] & PM_W_HAS_LOC) {
        seed_pos = i;
        break;
      }
    if (seed_pos < rem.size()) {
      const uint32_t seed = rem[seed_pos];
      batch->push_back(seed);
      std::vector<std::pair<double, uint32_t>> others;
      for (size_t i = 0; i < rem.size(); ++i)
        if (i != seed_pos && (e->h_flags[rem[i]] & PM_W_HAS_LOC))
          others.emplace_back(host_distance(e->h_lat[seed], e->h_lon[seed], e->h_lat[rem[i]], e->h_lon[rem[i]]), rem[i]);
      std::stable_sort(others.begin(), others.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
      for (const auto& o : others)
        if (batch->size() + 1 <= c.max_group_size) {
          batch->push_back(o.second);
          if (batch->size() >= c.max_group_size) break;
        }
    }
  }
  if (batch->empty() || (batch->size() < c.max_group_size && batch->size() < c.min_group_size)) {
    if (batch->size() < c.min_group_size) batch->clear();
    for (uint32_t w : rem) {
      if (std::find(batch->begin(), batch->end(), w) != batch->end()) continue;
      if (batch->size() + 1 <= c.max_group_size) {
        batch->push_back(w);
        if (batch->size() >= c.max_group_size) break;
      }
    }
  }
}

static int32_t run_merge(pm_engine* e, uint32_t* n_merged) {
  if (n_merged) *n_merged = 0;
  if (e->n_dead_groups) {  // (the records of a carve not yet absorbed are numbered behind the list as it is: take
    int32_t rc0 = absorb_groups(e);  // them in before the list is compacted)
    if (rc0) return rc0;
  }
  compact_groups(e);
  if (!e->have_cfgs || !e->have_workers) return set_error(PM_ESTATE, "configs and workers must be uploaded first");
  size_t solo = e->absorb_pending ? e->ab_solo : 0;  // single-node groups of the carve not yet absorbed
  for (const Group& g : e->groups) solo += g.members.size() == 1;
  e->tick_needs_merge = solo >= 2;  // (pm_tick: whether the next tick may queue its pair sweep behind the carve unseen)
  if (solo < 2) return PM_OK;  // mod.rs:641-644
  {
    int32_t rc0 = absorb_groups(e);
    if (rc0) return rc0;
  }
  int32_t rc = ensure_compat(e);
  if (rc) return rc;
  rc = pull_compat(e);
  if (rc) return rc;
  std::vector<uint32_t> avail;
  available_order(e->cfgs.data(), uint32_t(e->cfgs.size()), e->enabled, &avail);
  uint32_t merged = 0;
  // Linear in the number of solo groups: the list is never erased from in the middle (a merged group's old slots are
  // marked dead and dropped by ONE compaction at the end — the order that leaves is the order erase + push_back leaves,
  // mod.rs:903-942), group_of is patched for the merged workers only, the workers used by earlier batches are a byte
  // per worker, and a relaunch costs one host wait.
  std::vector<uint8_t> used(std::max<size_t>(e->W, 1), 0);
  std::vector<uint32_t> slots, rem, order, b_n, b_off, mem;
  std::vector<std::vector<uint32_t>> batches;
  CarveArgs a;
  CarveStatus st{}, st_in{};
  for (uint32_t cfg : avail) {  // mod.rs:654
    const pm_config_row& c = e->cfgs[cfg];
    compact_groups(e);  // (what the previous configuration merged away)
    // get_all_groups (sorted by id string) -> find_compatible_solo_groups (mod.rs:712-734)
    slots.clear();
    for (uint32_t s = 0; s < e->groups.size(); ++s)
      if (e->groups[s].members.size() == 1 && ((e->h_compat[e->groups[s].members[0]] >> cfg) & 1ull)) slots.push_back(s);
    if (slots.size() < c.min_group_size) continue;  // mod.rs:688-691
    std::sort(slots.begin(), slots.end(),
              [&](uint32_t x, uint32_t y) { return hex_id_less(e->groups[x].id, e->groups[y].id); });
    rem.clear();
    for (uint32_t s : slots) rem.push_back(e->groups[s].members[0]);

    // ---- selection of all batches for this configuration on the GPU (carve kernel, MERGE mode)
    batches.clear();
    {
      rc = push_groups(e);
      if (rc) return rc;
      order = rem;
      auto mark_used = [&](const std::vector<uint32_t>& b) {
        for (uint32_t w : b) used[w] = 1;
      };
      auto drop_used = [&]() {  // the workers of `rem` no batch has taken, in `rem`'s order
        order.clear();
        for (uint32_t w : rem)
          if (!used[w]) order.push_back(w);
      };
      const size_t cap = std::max<size_t>(e->W, 1);
      HIPCHK(e->d_m_cfg.ensure(cap));
      HIPCHK(e->d_m_n.ensure(cap));
      HIPCHK(e->d_m_off.ensure(cap));
      HIPCHK(e->d_m_members.ensure(cap));
      // The selections of a long list go through the streaming carve first (carve_stream_kernel in MERGE mode: the chain
      // commits the groups of a seed and its max - 1 nearest located candidates at 0.3 us each where the single-workgroup
      // kernel below sweeps the whole list for every one of them, 17 us at 5,000 candidates); whatever that launch leaves —
      // nothing, as a rule: it ends with exact steps of its own — is the old kernel's.
      bool try_stream = e->cfg.carve_variant == 0 && e->cfg.proximity_enabled && !e->debug_mem_above &&
                        order.size() >= e->merge_stream_min && order.size() <= PM_CARVE_BIG_SLOTS && c.max_group_size > 1 &&
                        c.max_group_size - 1u < PM_PROP_KMAX && c.min_group_size >= 1;
      for (;;) {
        const bool stream_now = try_stream;
        try_stream = false;
        rc = fill_carve_args(e, &a, CARVE_MODE_MERGE, uint32_t(order.size()), stream_now);
        if (rc) return rc;
        a.n_avail = 1;
        a.avail_cfg[0] = cfg;
        a.min_size[0] = c.min_group_size;
        a.max_size[0] = c.max_group_size;
        a.g_cfg = e->d_m_cfg.p;
        a.g_n = e->d_m_n.p;
        a.g_off = e->d_m_off.p;
        a.members = e->d_m_members.p;
        a.cap_groups = uint32_t(cap);
        a.cap_members = uint32_t(cap);
        bool in_lds;
        const size_t lds = carve_lds_bytes(a.bits_stride, &in_lds);
        st_in = CarveStatus{};
        st_in.state = CARVE_STATE_RUNNING;
        st_in.steps_total = e->tick_carve_steps;
        // a launch makes at most order.size() / 2 batches out of at most order.size() members: the three arrays come
        // back at that bound together with the status — ONE wait per launch
        const size_t n_o = order.size(), nb_max = std::max<size_t>(n_o / 2, 1);
        b_n.resize(nb_max);
        b_off.resize(nb_max);
        mem.resize(std::max<size_t>(n_o, 1));
        if (n_o) HIPCHK(hipMemcpyAsync(e->d_order.p, order.data(), n_o * 4, hipMemcpyHostToDevice, e->stream));
        if (stream_now) {
          if (e->stream_seq == 0 || e->stream_seq >= 127) {  // (see form_queue_init: the rings are cleared when the tags wrap)
            HIPCHK(hipMemsetAsync(e->d_stream_sq.p, 0, size_t(PM_STREAM_SQ) * 8, e->stream));
            HIPCHK(hipMemsetAsync(e->d_stream_row_lo.p, 0, size_t(PM_STREAM_RQ) * 64 * 8, e->stream));
            HIPCHK(hipMemsetAsync(e->d_stream_row_hi.p, 0, size_t(PM_STREAM_RQ) * 64 * 8, e->stream));
            e->stream_seq = 0;
          }
          e->stream_seq += 1;
          a.stream_tag0 = e->stream_seq << 25;
          const uint32_t wgs_want = e->stream_wgs_env ? e->stream_wgs_env : uint32_t(n_o) / 64u + 48u;
          const uint32_t wgs = std::max(1u, std::min(wgs_want, e->n_cus > 8u ? e->n_cus - 4u : 4u));
          HIPCHK(hipMemcpyAsync(e->d_carve_args.p, &a, sizeof(a), hipMemcpyHostToDevice, e->stream));
          launch_merge_place(e->d_carve_args.p, uint32_t(n_o), 0u, 0u, e->tick_carve_steps, e->stream);
          HIPCHK(launch_carve_stream(e->d_carve_args.p, 0u, wgs, e->stream));
          e->tick_carve_launches += 2;
        } else {
          HIPCHK(hipMemcpyAsync(e->d_status.p, &st_in, sizeof(st_in), hipMemcpyHostToDevice, e->stream));
          HIPCHK(hipMemcpyAsync(e->d_carve_args.p, &a, sizeof(a), hipMemcpyHostToDevice, e->stream));
          HIPCHK(launch_carve(e->d_carve_args.p, CARVE_F_INIT | CARVE_F_RUN | CARVE_F_ALL, 0, lds, e->stream));
        }
        HIPCHK(hipMemcpyAsync(&st, e->d_status.p, sizeof(st), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(b_n.data(), e->d_m_n.p, nb_max * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(b_off.data(), e->d_m_off.p, nb_max * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(mem.data(), e->d_m_members.p, mem.size() * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        e->tick_carve_launches++;
        if (st.state == CARVE_STATE_OVERFLOW) return set_error(PM_ENOMEM, "merge: batch arrays overflow");
        const bool stream_gave_up = stream_now && st.state == CARVE_STATE_ABORTED;  // (what it committed stands)
        if (stream_gave_up) e->tick_stream_aborts++;
        if (stream_now) e->merge_streamed++;
        if (st.state != CARVE_STATE_DONE && st.state != CARVE_STATE_UNCERTAIN && !stream_gave_up)
          return set_error(PM_ENODEV, "merge kernel did not complete");
        const uint32_t nb = st.n_groups;
        if (nb > nb_max || st.n_members > mem.size()) return set_error(PM_ENODEV, "merge: more batches than candidates");
        for (uint32_t k = 0; k < nb; ++k) {
          batches.emplace_back(mem.begin() + b_off[k], mem.begin() + b_off[k] + b_n[k]);
          mark_used(batches.back());
        }
        e->tick_carve_steps = st.steps_total;
        drop_used();
        if (st.state == CARVE_STATE_DONE) break;
        if (stream_gave_up) continue;  // the single-workgroup kernel takes the rest of the list
        // UNCERTAIN: settle exactly this selection on the host, then let the kernel continue
        if (order.size() < c.min_group_size) break;
        std::vector<uint32_t> b;
        host_merge_select(e, order, c, &b);
        e->tick_host_resolved++;
        e->tick_carve_steps++;
        if (b.size() < 2) break;
        mark_used(b);
        batches.push_back(std::move(b));
        drop_used();
        if (order.size() < c.min_group_size) break;
      }
      for (uint32_t w : rem) used[w] = 0;
    }

    // ---- apply the batches in order (is_merge_beneficial / should_switch_tasks / execute_group_merge)
    for (const auto& b : batches) {
      if (b.size() < 2) break;                 // mod.rs:868-870
      if (!e->cfg.switching_enabled) break;    // mod.rs:263-265
      bool blocked = false;
      if (!e->cfg.prefer_larger_groups)        // mod.rs:277-287
        for (uint32_t w : b)
          if (e->groups[e->h_group_of[w]].task != PM_NONE) blocked = true;
      if (blocked) break;
      Group gr;
      gr.id = splitmix64_next(&e->id_rng);  // mod.rs:886
      gr.cfg = cfg;
      gr.members = b;
      gr.task = PM_NONE;
      gr.task_uid = 0;
      rc = pick_task_for_config(e, cfg, gr.id, &gr.task);  // find_best_task_for_group, mod.rs:896
      if (rc) return rc;
      if (gr.task != PM_NONE) gr.task_uid = e->tasks_have_uid ? e->h_tuid[gr.task] : task_position(e, gr.task);
      for (uint32_t w : b) log_group_event(e, PM_GROUP_DESTROYED, e->groups[e->h_group_of[w]]);  // send_merge_webhooks,
      log_group_event(e, PM_GROUP_CREATED, gr);                                                   // mod.rs:974-1000
      const int32_t slot_new = int32_t(e->groups.size());
      for (uint32_t w : b) {  // mod.rs:903-921: the solo groups go (dropped from the list by compact_groups)
        e->groups[e->h_group_of[w]].dead = true;
        e->n_dead_groups++;
        e->h_group_of[w] = slot_new;
      }
      e->groups.push_back(std::move(gr));  // mod.rs:924-942
      e->groups_dirty = true, e->groups_delta_ok = false;
      ++merged;
    }
  }
  compact_groups(e);
  if (n_merged) *n_merged = merged;
  return PM_OK;
}

}  // namespace pm

// ================================================================================================
// C ABI

extern "C" {

const char* pm_last_error(void) { return pm::g_last_error.c_str(); }

void pm_engine_config_default(pm_engine_config* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->abi_version = PM_ABI_VERSION;
  c->device = 0;
  c->proximity_enabled = 1;
  c->switching_enabled = 1;
  c->prefer_larger_groups = 1;
  c->chooser = PM_CHOOSE_FIRST;
  c->group_id_seed = 1;
}

int32_t pm_engine_create(const pm_engine_config* cfg, pm_engine** out) {
  if (!cfg || !out) return set_error(PM_EINVAL, "null argument");
  if (cfg->abi_version != PM_ABI_VERSION) return set_error(PM_EINVAL, "ABI version mismatch");
  if (cfg->carve_variant != 0 && cfg->carve_variant != 1 && cfg->carve_variant != 3)
    return set_error(PM_EINVAL, "carve_variant: 0 (streaming carve), 1 (exact sweep only) or 3 (batch pipeline); 2 and 4 were removed");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return set_error(PM_ENODEV, "no HIP device visible: the matching engine needs an MI355X (gfx950); there is no CPU fallback");
  if (cfg->device < 0 || cfg->device >= n_dev) return set_error(PM_EINVAL, "device ordinal out of range");
  HIPCHK(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(PM_ENODEV, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  HIPCHK(carve_kernels_init());  // (per device; idempotent)
  pm_engine* e = new (std::nothrow) pm_engine();
  if (!e) return set_error(PM_ENOMEM, "out of host memory");
  e->cfg = *cfg;
  e->id_rng = cfg->group_id_seed;
  if (const char* v = getenv("PM_PRUNE_MODE")) {  // (experiments: see pm_debug_prune_mode)
    if (v[0] >= '0' && v[0] <= '3' && !v[1]) e->prune_mode = uint32_t(v[0] - '0');
  }
  if (const char* v = getenv("PM_WALK_CAP_DIV")) {
    const long f = atol(v);
    if (f > 0 && f < 1000) e->walk_cap_div = uint32_t(f);
  }
  if (const char* v = getenv("PM_PRUNE_FACTOR")) {
    const long f = atol(v);
    if (f > 0 && f < (1l << 30)) e->prune_factor = uint32_t(f);
  }
  if (const char* v = getenv("PM_STREAM_WGS")) {
    const long f = atol(v);
    if (f > 0 && f < 4096) e->stream_wgs_env = uint32_t(f);
  }
  if (const char* v = getenv("PM_STREAM_ROW_SPINS")) {  // (tests: a validator that hardly waits for its rows)
    const long f = atol(v);
    if (f > 0 && f < (1l << 24)) e->stream_row_spins_env = uint32_t(f);
  }
  if (const char* v = getenv("PM_STREAM_LA_DIV")) {
    const long f = atol(v);
    if (f > 0 && f < 4096) e->stream_la_div_env = uint32_t(f);
  }
  if (const char* v = getenv("PM_MERGE_STREAM_MIN")) {
    const long f = atol(v);
    if (f >= 2 && f <= (1l << 30)) e->merge_stream_min = uint32_t(f);
  }
  if (const char* v = getenv("PM_STREAM_LA")) {
    const long f = atol(v);
    if (f > 0 && f <= long(PM_STREAM_LA_MAX)) e->stream_la_env = uint32_t(f);
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0)
      e->n_cus = uint32_t(cus);
  }
  // (an engine owns ONE stream: K engines in a process take K of the HIP runtime's hardware queues)
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
    delete e;
    return set_error(PM_ENODEV, "hipStreamCreate failed");
  }
  e->stream_owned = e->stream;
  for (auto& ev : e->ev)
    if (hipEventCreate(&ev) != hipSuccess) {
      delete e;
      return set_error(PM_ENODEV, "hipEventCreate failed");
    }
  for (auto& ev : e->kev)
    if (hipEventCreate(&ev) != hipSuccess) {
      delete e;
      return set_error(PM_ENODEV, "hipEventCreate failed");
    }
  if (hipEventCreateWithFlags(&e->ev_groups, hipEventDisableTiming) != hipSuccess) {
    delete e;
    return set_error(PM_ENODEV, "hipEventCreate failed");
  }
  *out = e;
  return PM_OK;
}

void pm_engine_destroy(pm_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  (void)hipStreamSynchronize(e->stream);
  e->d_cfgs.release(); e->d_alts.release(); e->d_model_bits.release();
  e->d_flags.release(); e->d_gpu_count.release(); e->d_gpu_mem.release(); e->d_gpu_cls.release();
  e->d_cpu_cores.release(); e->d_ram.release(); e->d_storage.release(); e->d_addr_rank.release();
  e->d_lat.release(); e->d_lon.release(); e->d_coslat.release(); e->d_compat.release();
  e->d_ux.release(); e->d_uy.release(); e->d_uz.release();
  for (int k = 0; k < 3; ++k) { e->d_c_u[k].release(); e->d_cc_u[k].release(); }
  e->d_tmask.release(); e->d_tplanes.release(); e->d_created.release(); e->d_tlive.release(); e->d_tprefix.release();
  e->d_first_c.release(); e->d_count_c.release(); e->d_tdel.release();
  e->d_group_of.release(); e->d_g_cfg.release(); e->d_g_n.release(); e->d_g_off.release();
  e->d_g_task.release(); e->d_g_task_next.release(); e->d_members.release(); e->d_by_rank.release();
  e->d_rank_in_group.release(); e->d_g_id.release();
  e->d_order.release(); e->d_c_lat.release(); e->d_c_lon.release(); e->d_c_cos.release();
  e->d_cc_lat.release(); e->d_cc_lon.release(); e->d_cc_cos.release(); e->d_slot_pos.release(); e->d_slot_wid.release();
  e->d_site.release(); e->d_c_site.release(); e->d_cc_site.release(); e->d_prop.release(); e->d_prop_send.release(); e->d_seed_map.release(); e->d_seed_prefix.release(); e->d_seed_slots.release(); e->d_prep_block_counts.release(); e->d_prep_counts.release();
  e->d_c_compat.release(); e->d_keys.release(); e->d_bits.release(); e->d_status.release(); e->d_carve_args.release();
  e->d_m_cfg.release(); e->d_m_n.release(); e->d_m_off.release(); e->d_m_members.release();
  e->d_cfgbits.release(); e->d_stream_sq.release(); e->d_stream_row_lo.release(); e->d_stream_row_hi.release();
  e->d_stream_ctl.release(); e->d_stream_trace.release();
  e->d_sel.release(); e->d_wplanes.release(); e->d_sel_perm.release();
  e->d_first.release(); e->d_count.release(); e->d_rank.release(); e->d_chosen.release(); e->d_perm.release();
  e->d_table.release(); e->d_task_col.release(); e->d_shard.release(); e->d_own_rows.release(); e->d_xrow.release();
  e->d_sel_own.release(); e->d_table_x.release(); e->d_row_stage.release(); e->d_nb_idx.release(); e->d_nb_val.release();
  if (e->h_gstage) (void)hipHostFree(e->h_gstage);
  if (e->h_status) (void)hipHostFree(e->h_status);
  for (int k = 0; k < 2; ++k) {
    if (e->h_delta_pin[k]) (void)hipHostFree(e->h_delta_pin[k]);
    e->d_delta[k].release();
  }
  if (e->h_gtask_pinned) (void)hipHostFree(e->h_gtask_pinned);
  if (e->ev_groups) (void)hipEventDestroy(e->ev_groups);
  for (PubTable& t : e->pub)
    if (t.words.load()) (void)hipHostFree(t.words.load());
  for (uint64_t* q : e->pub_retired) (void)hipHostFree(q);
  for (auto& ev : e->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : e->kev)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t x : e->prop_ev) (void)hipEventDestroy(x);
  if (e->stream_owned) (void)hipStreamDestroy(e->stream_owned);
  e->d_cell_cnt.release(); e->d_cell_start.release(); e->d_pos_cell.release(); e->d_pos_rank.release();
  e->d_cs_of_pos.release(); e->d_cs_slot.release(); e->d_cs_site.release();
  for (auto& u : e->d_cs_u) u.release();
  e->d_c_pack.release();
  e->d_cs_pack.release();
  e->d_desc.release(); e->d_snap.release(); e->d_ikeys.release(); e->d_umask.release(); e->d_ivals.release();
  delete e->form;
  delete e;
}

int32_t pm_set_configs(pm_engine* e, const pm_config_row* cfgs, uint32_t n_cfgs, const pm_gpu_alt_row* alts,
                       uint32_t n_alts) {
  if (!e || (n_cfgs && !cfgs) || (n_alts && !alts)) return set_error(PM_EINVAL, "null argument");
  if (n_cfgs > PM_MAX_CONFIGS) return set_error(PM_EINVAL, "more than PM_MAX_CONFIGS configurations");
  for (uint32_t i = 0; i < n_cfgs; ++i) {
    if (cfgs[i].max_group_size < cfgs[i].min_group_size)
      return set_error(PM_EINVAL, "Plugin configuration is invalid (max_group_size < min_group_size)");  // mod.rs:145
    if (cfgs[i].min_group_size == 0)
      return set_error(PM_EINVAL, "min_group_size == 0 is rejected (the reference would form empty groups forever)");
    if (uint64_t(cfgs[i].alt_begin) + cfgs[i].alt_count > n_alts)
      return set_error(PM_ERANGE, "alternative range outside the alt table");
  }
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  HIPCHK(hipSetDevice(e->cfg.device));
  e->cfgs.assign(cfgs, cfgs + n_cfgs);
  e->alts.assign(alts, alts + n_alts);
  int32_t rc = upload(e->d_cfgs, e->cfgs.data(), e->cfgs.size(), e->stream);
  if (rc) return rc;
  rc = upload(e->d_alts, e->alts.data(), e->alts.size(), e->stream);
  if (rc) return rc;
  HIPCHK(e->d_model_bits.ensure(1));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->have_cfgs = true;
  e->compat_dirty = true;
  e->tplanes_dirty = true;
  return PM_OK;
}

int32_t pm_set_model_table(pm_engine* e, const uint32_t* bits, uint32_t n_rows, uint32_t n_classes) {
  if (!e || (n_rows && n_classes && !bits)) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  const size_t words = size_t(n_rows) * ((n_classes + 31u) / 32u);
  e->model_bits.assign(bits, bits + words);
  e->model_rows = n_rows;
  e->model_classes = n_classes;
  e->cls_check_dirty = true;
  int32_t rc = upload(e->d_model_bits, e->model_bits.data(), words, e->stream);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  e->compat_dirty = true;
  return PM_OK;
}

int32_t pm_set_enabled_mask(pm_engine* e, uint64_t enabled) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  e->enabled = enabled;
  return PM_OK;
}

// ---- worker table ingestion.  Coordinates are interned into site ids (the carve certificate needs "same place"
// as an exact integer compare); the map persists so that row deltas intern incrementally.

static uint32_t site_of_row(pm_engine* e, size_t w) {
  uint64_t a, b;
  std::memcpy(&a, &e->h_lat[w], 8);
  std::memcpy(&b, &e->h_lon[w], 8);
  if (e->h_lat[w] == 0.0) a = 0;  // +0.0 and -0.0 compare equal in the reference's f64 arithmetic
  if (e->h_lon[w] == 0.0) b = 0;
  auto it = e->site_map.emplace(std::make_pair(a, b), uint32_t(e->site_map.size())).first;
  if (it->second >= e->site_pop.size()) e->site_pop.push_back(0);
  return it->second;
}

static void refresh_site_bits(pm_engine* e) {
  for (size_t w = 0; w < e->W; ++w) {
    const uint32_t id = e->h_site[w] & 0x7FFFFFFFu;
    e->h_site[w] = id | (e->site_pop[id] >= 2 ? 0x80000000u : 0u);
  }
  e->site_bits_dirty = false;
}

// the whole table: host mirror -> HBM columns
static int32_t upload_worker_columns(pm_engine* e) {
  const size_t W = e->W;
  int32_t rc;
  if ((rc = upload(e->d_flags, e->h_flags.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_gpu_count, e->h_gpu_count.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_gpu_mem, e->h_gpu_mem.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_gpu_cls, e->h_gpu_cls.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_cpu_cores, e->h_cpu_cores.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_ram, e->h_ram.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_storage, e->h_storage.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_addr_rank, e->h_addr_rank.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_lat, e->h_lat.data(), W, e->stream))) return rc;
  if ((rc = upload(e->d_lon, e->h_lon.data(), W, e->stream))) return rc;
  e->site_map.clear();
  e->site_map.reserve(W * 2);
  e->site_pop.clear();
  e->h_site.resize(W);
  for (size_t w = 0; w < W; ++w) {
    e->h_site[w] = site_of_row(e, w);
    e->site_pop[e->h_site[w]] += (e->h_flags[w] & PM_W_HAS_LOC) ? 1u : 0u;
  }
  refresh_site_bits(e);
  if ((rc = upload(e->d_site, e->h_site.data(), W, e->stream))) return rc;
  HIPCHK(e->d_coslat.ensure(W ? W : 1));
  HIPCHK(e->d_ux.ensure(W ? W : 1));
  HIPCHK(e->d_uy.ensure(W ? W : 1));
  HIPCHK(e->d_uz.ensure(W ? W : 1));
  launch_geo(e->d_lat.p, e->d_lon.p, e->d_coslat.p, e->d_ux.p, e->d_uy.p, e->d_uz.p, uint32_t(W), e->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  e->flags_dirty = false;
  e->flags_delta_ok = true;  // (the whole column has just gone up)
  e->delta_flags.clear();
  e->price_dirty = true;
  e->compat_dirty = true;
  return PM_OK;
}

// (price, index) order of the swept axis for pm_match_per_task; rebuilt only when a price changed
static void ensure_price_order(pm_engine* e) {
  if (!e->price_dirty) return;
  e->any_price = std::any_of(e->h_price.begin(), e->h_price.end(), [](uint32_t p) { return p != 0; });
  e->price_perm.clear();
  if (e->any_price) {
    e->price_perm.resize(e->W);
    for (uint32_t i = 0; i < e->W; ++i) e->price_perm[i] = i;
    std::stable_sort(e->price_perm.begin(), e->price_perm.end(),
                     [&](uint32_t a, uint32_t b) { return e->h_price[a] < e->h_price[b]; });
  }
  e->price_dirty = false;
}

static bool worker_soa_complete(const pm_worker_soa* w) {
  return !w->n || (w->flags && w->gpu_count && w->gpu_mem_mb && w->gpu_model_class && w->cpu_cores && w->ram_mb &&
                   w->storage_gb && w->lat && w->lon);
}

int32_t pm_upload_workers(pm_engine* e, const pm_worker_soa* w, uint32_t keep_groups) {
  if (!e || !w) return set_error(PM_EINVAL, "null argument");
  const uint32_t n = w->n;
  if (!worker_soa_complete(w)) return set_error(PM_EINVAL, "null worker column");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  ABSORB_PENDING(e);
  if (keep_groups && e->have_workers && n != e->W)
    return set_error(PM_EINVAL, "keep_groups requires the same rows in the same order (use pm_append_workers / "
                                "pm_update_workers for deltas)");
  if (n != e->W && e->dist_world > 1) {  // ownership is per worker row: pm_dist_configure must follow
    e->h_shard.clear();
    e->h_own_rows.clear();
  }
  e->W = n;
  e->h_flags.assign(w->flags, w->flags + n);
  e->h_gpu_count.assign(w->gpu_count, w->gpu_count + n);
  e->h_gpu_mem.assign(w->gpu_mem_mb, w->gpu_mem_mb + n);
  e->h_gpu_cls.assign(w->gpu_model_class, w->gpu_model_class + n);
  e->cls_check_dirty = true;
  e->h_cpu_cores.assign(w->cpu_cores, w->cpu_cores + n);
  e->h_ram.assign(w->ram_mb, w->ram_mb + n);
  e->h_storage.assign(w->storage_gb, w->storage_gb + n);
  if (w->price) e->h_price.assign(w->price, w->price + n); else e->h_price.assign(n, 0);
  if (w->addr_rank) {
    e->h_addr_rank.assign(w->addr_rank, w->addr_rank + n);
  } else {
    e->h_addr_rank.resize(n);
    for (uint32_t i = 0; i < n; ++i) e->h_addr_rank[i] = i;
  }
  e->h_lat.assign(w->lat, w->lat + n);
  e->h_lon.assign(w->lon, w->lon + n);
  int32_t rc = upload_worker_columns(e);
  if (rc) return rc;
  e->have_workers = true;
  if (!keep_groups || e->h_group_of.size() != n) reset_groups_locked(e);
  e->groups_dirty = true, e->groups_delta_ok = false;
  return PM_OK;
}

// Rows idx[0..n) <- rows: host mirror, incremental site interning, then ONE packed H2D copy and a scatter kernel
// (update_rows_kernel also refreshes cos(lat)).  Shared by pm_update_workers and pm_append_workers.
static int32_t scatter_rows(pm_engine* e, const uint32_t* idx, const pm_worker_soa* rows, uint32_t first_new) {
  const uint32_t n = rows->n;
  if (!n) return PM_OK;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t w = idx[k];
    if (w < first_new) {  // an existing row leaves its old site (rows from first_new on are being appended)
      const uint32_t old = e->h_site[w] & 0x7FFFFFFFu;
      if ((e->h_flags[w] & PM_W_HAS_LOC) && old < e->site_pop.size()) {
        if (e->site_pop[old] == 2) e->site_bits_dirty = true;
        if (e->site_pop[old]) e->site_pop[old]--;
      }
    }
    e->h_flags[w] = rows->flags[k];
    e->h_gpu_count[w] = rows->gpu_count[k];
    e->h_gpu_mem[w] = rows->gpu_mem_mb[k];
    e->h_gpu_cls[w] = rows->gpu_model_class[k];
    e->cls_check_dirty = true;
    e->h_cpu_cores[w] = rows->cpu_cores[k];
    e->h_ram[w] = rows->ram_mb[k];
    e->h_storage[w] = rows->storage_gb[k];
    if (rows->price) {
      if (e->h_price[w] != rows->price[k]) e->price_dirty = true;
      e->h_price[w] = rows->price[k];
    }
    if (rows->addr_rank) e->h_addr_rank[w] = rows->addr_rank[k];
    e->h_lat[w] = rows->lat[k];
    e->h_lon[w] = rows->lon[k];
    const uint32_t id = site_of_row(e, w);
    if (rows->flags[k] & PM_W_HAS_LOC) {
      if (e->site_pop[id] == 1) e->site_bits_dirty = true;
      e->site_pop[id]++;
    }
    e->h_site[w] = id | (e->site_pop[id] >= 2 ? 0x80000000u : 0u);
  }
  // packed staging: [lat n][lon n] f64, then 10 u32 columns: idx, flags, gpu_count, gpu_mem, gpu_cls, cpu_cores,
  // ram, storage, addr_rank, site
  const size_t bytes = size_t(n) * (16 + 10 * 4);
  std::vector<unsigned char> host(bytes);
  double* hd = reinterpret_cast<double*>(host.data());
  uint32_t* hu = reinterpret_cast<uint32_t*>(host.data() + size_t(n) * 16);
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t w = idx[k];
    hd[k] = e->h_lat[w];
    hd[n + k] = e->h_lon[w];
    hu[0 * size_t(n) + k] = w;
    hu[1 * size_t(n) + k] = e->h_flags[w];
    hu[2 * size_t(n) + k] = e->h_gpu_count[w];
    hu[3 * size_t(n) + k] = e->h_gpu_mem[w];
    hu[4 * size_t(n) + k] = e->h_gpu_cls[w];
    hu[5 * size_t(n) + k] = e->h_cpu_cores[w];
    hu[6 * size_t(n) + k] = e->h_ram[w];
    hu[7 * size_t(n) + k] = e->h_storage[w];
    hu[8 * size_t(n) + k] = e->h_addr_rank[w];
    hu[9 * size_t(n) + k] = e->h_site[w];
  }
  HIPCHK(e->d_row_stage.ensure(bytes));
  HIPCHK(hipMemcpyAsync(e->d_row_stage.p, host.data(), bytes, hipMemcpyHostToDevice, e->stream));
  RowUpdateArgs a{};
  a.n = n;
  a.lat_in = reinterpret_cast<const double*>(e->d_row_stage.p);
  a.lon_in = a.lat_in + n;
  a.u32_in = reinterpret_cast<const uint32_t*>(e->d_row_stage.p + size_t(n) * 16);
  a.flags = e->d_flags.p;
  a.gpu_count = e->d_gpu_count.p;
  a.gpu_mem = e->d_gpu_mem.p;
  a.gpu_cls = e->d_gpu_cls.p;
  a.cpu_cores = e->d_cpu_cores.p;
  a.ram = e->d_ram.p;
  a.storage = e->d_storage.p;
  a.addr_rank = e->d_addr_rank.p;
  a.site = e->d_site.p;
  a.lat = e->d_lat.p;
  a.lon = e->d_lon.p;
  a.coslat = e->d_coslat.p;
  a.ux = e->d_ux.p;
  a.uy = e->d_uy.p;
  a.uz = e->d_uz.p;
  launch_update_rows(a, e->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));  // the staging vector dies here
  if (e->site_bits_dirty) {  // a site crossed the one <-> two located workers line: refresh the hint bits
    refresh_site_bits(e);
    int32_t rc = upload(e->d_site, e->h_site.data(), e->W, e->stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  e->compat_dirty = true;
  e->h_compat_valid = false;
  return PM_OK;
}

int32_t pm_update_workers(pm_engine* e, const uint32_t* idx, const pm_worker_soa* rows) {
  if (!e || !rows || (rows->n && !idx)) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->have_workers) return set_error(PM_ESTATE, "workers must be uploaded first");
  if (!worker_soa_complete(rows)) return set_error(PM_EINVAL, "null worker column");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  for (uint32_t k = 0; k < rows->n; ++k)
    if (idx[k] >= e->W) return set_error(PM_ERANGE, "worker index out of range");
  int32_t rc = sync_flags(e);  // status-only changes made since the last upload go up first (whole column)
  if (rc) return rc;
  return scatter_rows(e, idx, rows, e->W);
}

int32_t pm_append_workers(pm_engine* e, const pm_worker_soa* rows, uint32_t* first_index) {
  if (!e || !rows) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->have_workers) return set_error(PM_ESTATE, "workers must be uploaded first (an empty table is fine)");
  if (!worker_soa_complete(rows)) return set_error(PM_EINVAL, "null worker column");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  ABSORB_PENDING(e);
  const uint32_t w0 = e->W, n = rows->n;
  if (first_index) *first_index = w0;
  if (!n) return PM_OK;
  if (uint64_t(w0) + n > 0x7FFFFFFFull) return set_error(PM_ERANGE, "worker table too large");
  int32_t rc = sync_flags(e);
  if (rc) return rc;
  const size_t W1 = size_t(w0) + n;
  HIPCHK(e->d_flags.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_gpu_count.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_gpu_mem.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_gpu_cls.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_cpu_cores.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_ram.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_storage.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_addr_rank.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_site.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_lat.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_lon.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_coslat.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_ux.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_uy.grow_keep(W1, w0, e->stream));
  HIPCHK(e->d_uz.grow_keep(W1, w0, e->stream));
  e->h_flags.resize(W1, 0);
  e->h_gpu_count.resize(W1, 0);
  e->h_gpu_mem.resize(W1, 0);
  e->h_gpu_cls.resize(W1, 0);
  e->h_cpu_cores.resize(W1, 0);
  e->h_ram.resize(W1, 0);
  e->h_storage.resize(W1, 0);
  e->h_price.resize(W1, 0);
  e->h_addr_rank.resize(W1, 0);
  e->h_lat.resize(W1, 0.0);
  e->h_lon.resize(W1, 0.0);
  e->h_site.resize(W1, 0);
  std::vector<uint32_t> idx(n);
  for (uint32_t k = 0; k < n; ++k) {
    idx[k] = w0 + k;
    if (!rows->addr_rank) e->h_addr_rank[w0 + k] = w0 + k;
  }
  // a new row joins no group: the existing groups and their claimed tasks stay as they are (mod.rs:487-497)
  e->h_group_of.resize(W1, -1);
  e->W = uint32_t(W1);
  rc = scatter_rows(e, idx.data(), rows, w0);
  if (rc) return rc;
  if (e->dist_world > 1) {  // ownership is per worker row: pm_dist_configure must follow
    e->h_shard.clear();
    e->h_own_rows.clear();
  }
  e->price_dirty = true;
  if (e->groups_delta_ok && e->delta_tail_from == PM_NONE) e->delta_tail_from = w0;
  e->groups_dirty = true;  // group_of is sized by W (a delta while groups_delta_ok: see push_groups)
  return PM_OK;
}

int32_t pm_set_addr_ranks(pm_engine* e, const uint32_t* addr_rank, uint32_t n) {
  if (!e || (n && !addr_rank)) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->have_workers || n != e->W) return set_error(PM_ERANGE, "one rank per worker row");
  HIPCHK(hipSetDevice(e->cfg.device));
  e->h_addr_rank.assign(addr_rank, addr_rank + n);
  int32_t rc = upload(e->d_addr_rank, e->h_addr_rank.data(), n, e->stream);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  return PM_OK;
}

// A fresh, empty task index space of capacity `cap` (a multiple of 64): host mirror and device columns zeroed.
static int32_t tasks_alloc(pm_engine* e, uint32_t cap) {
  e->h_tmask.assign(cap, 0);
  e->h_created.assign(cap, 0);
  e->h_tlive.assign(cap / 64u, 0);
  if (e->tasks_have_uid) e->h_tuid.assign(cap, 0); else e->h_tuid.clear();
  e->t_cap = cap;
  e->t_lo = cap;
  e->T = e->t_dead = 0;
  e->h_tprefix_valid = false;
  HIPCHK(e->d_tmask.ensure(cap));
  HIPCHK(e->d_created.ensure(cap));
  HIPCHK(e->d_tlive.ensure(cap / 64u));
  HIPCHK(e->d_tprefix.ensure(cap / 64u));
  HIPCHK(hipMemsetAsync(e->d_tmask.p, 0, size_t(cap) * 8, e->stream));
  HIPCHK(hipMemsetAsync(e->d_tlive.p, 0, size_t(cap / 64u) * 8, e->stream));
  e->tplanes_dirty = true;
  e->tprefix_dirty = true;
  return PM_OK;
}

// Host range [u0, u1) of the table -> HBM (masks, created_at, the live words it touches)
static int32_t tasks_push_range(pm_engine* e, uint32_t u0, uint32_t u1) {
  if (u1 <= u0) return PM_OK;
  const uint32_t w0 = u0 / 64u, w1 = (u1 + 63u) / 64u;
  HIPCHK(hipMemcpyAsync(e->d_tmask.p + u0, e->h_tmask.data() + u0, size_t(u1 - u0) * 8, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_created.p + u0, e->h_created.data() + u0, size_t(u1 - u0) * 8, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_tlive.p + w0, e->h_tlive.data() + w0, size_t(w1 - w0) * 8, hipMemcpyHostToDevice, e->stream));
  return PM_OK;
}

// The table outgrew the room in front of it: a larger index space, the used part moves to its top.  Handles
// shift by (new capacity - old capacity): the groups' claims are shifted with them.
static int32_t tasks_grow(pm_engine* e, uint32_t cap) {
  const uint32_t old_cap = e->t_cap, old_lo = e->t_lo, used = old_cap - old_lo, new_lo = cap - used;
  const uint32_t shift = new_lo - old_lo;
  std::vector<uint64_t> tmask(e->h_tmask.begin() + old_lo, e->h_tmask.end());
  std::vector<int64_t> created(e->h_created.begin() + old_lo, e->h_created.end());
  std::vector<uint64_t> tuid;
  if (e->tasks_have_uid) tuid.assign(e->h_tuid.begin() + old_lo, e->h_tuid.end());
  std::vector<uint64_t> live_old = e->h_tlive;
  const uint32_t T = e->T, dead = e->t_dead;
  int32_t rc = tasks_alloc(e, cap);
  if (rc) return rc;
  std::copy(tmask.begin(), tmask.end(), e->h_tmask.begin() + new_lo);
  std::copy(created.begin(), created.end(), e->h_created.begin() + new_lo);
  if (e->tasks_have_uid) std::copy(tuid.begin(), tuid.end(), e->h_tuid.begin() + new_lo);
  for (uint32_t u = old_lo; u < old_cap; ++u)
    if ((live_old[u >> 6] >> (u & 63u)) & 1ull) {
      const uint32_t v = u + shift;
      e->h_tlive[v >> 6] |= 1ull << (v & 63u);
    }
  e->t_lo = new_lo;
  e->T = T;
  e->t_dead = dead;
  for (Group& g : e->groups)
    if (g.task != PM_NONE) g.task += shift;
  e->groups_dirty = true, e->groups_delta_ok = false;
  if (e->uid_map_valid)
    for (auto& kv : e->uid_to_u) kv.second += shift;
  rc = tasks_push_range(e, new_lo, cap);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  return PM_OK;
}

static uint32_t task_capacity_for(uint32_t n) {  // room for about as many insertions as there are tasks
  const uint64_t want = uint64_t(n) * 2 + 65536;
  return uint32_t(std::min<uint64_t>((want + 63) & ~uint64_t(63), 0xFFFFFFC0ull));
}

int32_t pm_upload_tasks(pm_engine* e, const pm_task_soa* t) {
  if (!e || !t) return set_error(PM_EINVAL, "null argument");
  if (t->n && (!t->topo_mask || !t->created_at)) return set_error(PM_EINVAL, "null task column");
  if (t->n > 0x7FFFFFFFu) return set_error(PM_ERANGE, "task table too large");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  ABSORB_PENDING(e);
  const uint32_t n = t->n;
  const bool had_uid = e->tasks_have_uid;
  e->tasks_have_uid = t->uid != nullptr;
  e->uid_to_u.clear();
  e->uid_map_valid = false;
  e->cfg_app_valid = false;
  int32_t rc;
  if (e->t_cap < n + 64u || uint64_t(e->t_cap) > uint64_t(n) * 8 + (1u << 20) || had_uid != e->tasks_have_uid) {
    rc = tasks_alloc(e, task_capacity_for(n));
    if (rc) return rc;
  }
  // the snapshot replaces the used part: position i of the caller's list -> index lo + i; whatever the old table
  // had in front of the new one is cleared (only the rows that changed hands travel)
  const uint32_t cap = e->t_cap, old_lo = e->t_lo, lo = cap - n, clear_from = std::min(old_lo, lo);
  for (uint32_t u = clear_from; u < lo; ++u) e->h_tmask[u] = 0;
  for (uint32_t w = clear_from / 64u; w < cap / 64u; ++w) e->h_tlive[w] = 0;
  std::memcpy(e->h_tmask.data() + lo, t->topo_mask, size_t(n) * 8);
  std::memcpy(e->h_created.data() + lo, t->created_at, size_t(n) * 8);
  if (t->uid) std::memcpy(e->h_tuid.data() + lo, t->uid, size_t(n) * 8);
  for (uint32_t u = lo; u < cap; ++u) e->h_tlive[u >> 6] |= 1ull << (u & 63u);
  e->t_lo = lo;
  e->T = n;
  e->t_dead = 0;
  e->h_tprefix_valid = false;
  rc = tasks_push_range(e, clear_from, cap);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));  // pageable sources
  e->tplanes_dirty = true;
  e->tprefix_dirty = true;
  e->have_tasks = true;
  // re-bind claimed tasks by identity; groups whose task vanished are dissolved (on_task_deleted,
  // mod.rs:1259-1288)
  // Only the claimed tasks have to be found again: an open-addressing table of their ids (a few thousand
  // entries, cache-resident), one pass over the new id column.
  size_t n_claim = 0;
  for (const Group& g : e->groups) n_claim += (!g.dead && g.task != PM_NONE);
  if (n_claim) {
    size_t hcap = 64;
    while (hcap < n_claim * 4) hcap <<= 1;
    const uint64_t EMPTY = ~0ull;  // a task id of all ones simply stays unresolved in the table path below
    std::vector<uint64_t> keys(hcap, EMPTY);
    std::vector<uint32_t> vals(hcap, PM_NONE);
    auto slot_of = [&](uint64_t k) {
      size_t h = size_t(splitmix64_mix(k)) & (hcap - 1);
      while (keys[h] != EMPTY && keys[h] != k) h = (h + 1) & (hcap - 1);
      return h;
    };
    if (e->tasks_have_uid) {
      for (const Group& g : e->groups)
        if (!g.dead && g.task != PM_NONE && g.task_uid != EMPTY) keys[slot_of(g.task_uid)] = g.task_uid;
      for (uint32_t i = 0; i < n; ++i) {
        const uint64_t k = t->uid[i];
        size_t h = size_t(splitmix64_mix(k)) & (hcap - 1);
        while (keys[h] != EMPTY && keys[h] != k) h = (h + 1) & (hcap - 1);
        if (keys[h] == k && vals[h] == PM_NONE) vals[h] = i;  // first occurrence, like a map's emplace
      }
    }
    for (size_t g = 0; g < e->groups.size(); ++g) {  // (creation order: dissolutions are logged in it)
      Group& gr = e->groups[g];
      if (gr.dead || gr.task == PM_NONE) continue;
      uint32_t ni = PM_NONE;  // position in the new list
      if (e->tasks_have_uid) {
        if (gr.task_uid != EMPTY) ni = vals[slot_of(gr.task_uid)];
      } else if (gr.task_uid < n) {
        ni = uint32_t(gr.task_uid);  // without ids the identity of a task is its position
      }
      if (ni == PM_NONE) {
        dissolve_locked(e, uint32_t(g));
      } else if (e->t_lo + ni != gr.task) {
        gr.task = e->t_lo + ni;
        e->groups_dirty = true, e->groups_delta_ok = false;
      }
    }
  }
  pub_patch(e, nullptr);  // the published positions follow the new list
  return PM_OK;
}

// on_task_created (node_groups/mod.rs:1224-1243) + TaskStore::add_task (store/domains/task_store.rs:33-55): new
// tasks are the newest, i.e. they sort in FRONT of get_all_tasks (:79, created_at desc).  Only the new rows
// travel; the bit planes are patched in the words they fall into; nothing else moves, so every claimed task
// keeps its handle.
static int32_t tasks_insert_front_locked(pm_engine* e, const pm_task_soa* t);

int32_t pm_tasks_insert_front(pm_engine* e, const pm_task_soa* t) {
  if (!e || !t) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  return tasks_insert_front_locked(e, t);
}

// pm_tasks_insert_front and, with republish != 0, pm_match's pair sweep + claim + publish on the standing groups under
// the same lock: a group that holds no task is offered the new one before the next tick, as the reference would offer
// it at that group's next heartbeat (scheduler_impl.rs:33-74).  No carve.
int32_t pm_tasks_insert_front_ex(pm_engine* e, const pm_task_soa* t, uint32_t republish) {
  if (!e || !t) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  int32_t rc = tasks_insert_front_locked(e, t);
  if (rc || !republish || !t->n) return rc;
  if (e->dist_world > 1) return set_error(PM_ESTATE, "multi-GPU engine: the republish belongs to the stepwise tick");
  if (!e->have_cfgs || !e->have_workers) return PM_OK;  // (nothing to match yet: the next tick publishes)
  ABSORB_PENDING(e);
  rc = run_match(e, false, nullptr);
  if (rc) return rc;
  return publish(e);
}

static int32_t tasks_insert_front_locked(pm_engine* e, const pm_task_soa* t) {
  if (t->n && (!t->topo_mask || !t->created_at)) return set_error(PM_EINVAL, "null task column");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (!e->have_tasks) return set_error(PM_ESTATE, "tasks must be uploaded first (an empty table is fine)");
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  if ((t->uid != nullptr) != e->tasks_have_uid && (e->T || e->t_dead))
    return set_error(PM_EINVAL, "the table was uploaded with / without task ids: the new rows must match");
  const uint32_t n = t->n;
  if (!n) return PM_OK;
  for (uint32_t i = 1; i < n; ++i)
    if (t->created_at[i] > t->created_at[i - 1])
      return set_error(PM_EINVAL, "rows must be in get_all_tasks order (created_at descending)");
  for (uint32_t u = e->t_lo; u < e->t_cap; ++u)  // the newest live task of the table
    if ((e->h_tlive[u >> 6] >> (u & 63u)) & 1ull) {
      // an equal timestamp would sort BEHIND the older task (stable sort, task_store.rs:79): not a front insertion
      if (t->created_at[n - 1] <= e->h_created[u])
        return set_error(PM_EINVAL, "not newer than the newest task of the table: use pm_upload_tasks");
      break;
    }
  ABSORB_PENDING(e);
  if (!e->T && !e->t_dead) e->tasks_have_uid = t->uid != nullptr;
  if (e->t_lo < n) {  // out of room in front: a larger index space, everything moves to its top
    if (e->tasks_have_uid && e->h_tuid.size() != e->t_cap) e->h_tuid.assign(e->t_cap, 0);
    int32_t rc = tasks_grow(e, task_capacity_for(e->t_cap - e->t_lo + n));
    if (rc) return rc;
  }
  if (e->tasks_have_uid && e->h_tuid.size() != e->t_cap) e->h_tuid.assign(e->t_cap, 0);
  const uint32_t lo = e->t_lo - n;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t u = lo + i;
    e->h_tmask[u] = t->topo_mask[i];
    e->h_created[u] = t->created_at[i];
    if (t->uid) {
      e->h_tuid[u] = t->uid[i];
      if (e->uid_map_valid) e->uid_to_u.emplace(t->uid[i], u);
    }
    e->h_tlive[u >> 6] |= 1ull << (u & 63u);
    if (e->cfg_app_valid) {
      uint64_t m = t->topo_mask[i];
      while (m) {
        e->cfg_app_count[__builtin_ctzll(m)]++;
        m &= m - 1;
      }
    }
  }
  {
    int32_t rcp = tasks_push_range(e, lo, e->t_lo);
    if (rcp) return rcp;
  }
  if (!e->tplanes_dirty && e->d_tplanes.p)  // patch the planes: the touched words are rebuilt from the masks
    launch_build_planes(e->d_tmask.p, e->t_cap, lo, e->t_lo, e->t_cap / 64u, uint32_t(e->cfgs.size()), e->d_tplanes.p,
                        e->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));  // pageable sources
  e->t_lo = lo;
  e->T += n;
  e->tprefix_dirty = true;
  e->h_tprefix_valid = false;
  pub_shift_tasks(e, n);  // n new tasks in front of the list: every published position moves back by n
  return PM_OK;
}

// on_task_deleted (node_groups/mod.rs:1245-1325): the tasks leave the table (tombstones: nothing moves) and every
// group that had claimed one of them is dissolved (:1259-1288).  Unknown ids are ignored.
int32_t pm_tasks_delete(pm_engine* e, const uint64_t* uids, uint32_t n, uint32_t* n_deleted) {
  if (!e || (n && !uids)) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (n_deleted) *n_deleted = 0;
  if (!e->have_tasks) return set_error(PM_ESTATE, "tasks must be uploaded first");
  if (!e->tasks_have_uid) return set_error(PM_ESTATE, "the task table carries no ids (pm_task_soa.uid)");
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  ABSORB_PENDING(e);
  if (!e->uid_map_valid) {
    e->uid_to_u.clear();
    e->uid_to_u.reserve(size_t(e->T) * 2);
    for (uint32_t u = e->t_cap; u-- > e->t_lo;)  // descending: the first occurrence in list order wins
      if ((e->h_tlive[u >> 6] >> (u & 63u)) & 1ull) e->uid_to_u[e->h_tuid[u]] = u;
    e->uid_map_valid = true;
  }
  // what is to go (nothing is changed yet: everything that can fail comes before the first mutation, so that the
  // host mirror and the device table cannot drift apart on an error return)
  std::vector<uint32_t> slots;  // handles of the deleted tasks
  for (uint32_t k = 0; k < n; ++k) {
    auto it = e->uid_to_u.find(uids[k]);
    if (it == e->uid_to_u.end()) continue;
    slots.push_back(it->second);
  }
  if (slots.empty()) return PM_OK;
  std::sort(slots.begin(), slots.end());
  slots.erase(std::unique(slots.begin(), slots.end()), slots.end());  // (an id named twice)
  HIPCHK(e->d_tdel.ensure(slots.size()));
  if (e->tplanes_dirty || !e->d_tplanes.p) {  // no planes yet: only the columns need the update
    int32_t rc = ensure_task_planes(e);
    if (rc) return rc;
  }
  HIPCHK(hipMemcpyAsync(e->d_tdel.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, e->stream));
  launch_task_delete(e->d_tdel.p, uint32_t(slots.size()), e->d_tmask.p, e->d_tlive.p, e->d_tplanes.p, e->t_cap / 64u,
                     uint32_t(e->cfgs.size()), e->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  // ---- the device table is updated: now the host mirror, the groups, the published rows
  for (uint32_t u : slots) {
    e->uid_to_u.erase(e->h_tuid[u]);
    if (e->cfg_app_valid) {
      uint64_t m = e->h_tmask[u];
      while (m) {
        e->cfg_app_count[__builtin_ctzll(m)]--;
        m &= m - 1;
      }
    }
    e->h_tmask[u] = 0;
    e->h_tlive[u >> 6] &= ~(1ull << (u & 63u));
  }
  e->h_tprefix_valid = false;
  for (size_t g = 0; g < e->groups.size(); ++g) {  // (creation order: dissolutions are logged in it)
    const Group& gr = e->groups[g];
    if (!gr.dead && gr.task != PM_NONE && std::binary_search(slots.begin(), slots.end(), gr.task))
      dissolve_locked(e, uint32_t(g));
  }
  e->T -= uint32_t(slots.size());
  e->t_dead += uint32_t(slots.size());
  while (e->t_lo < e->t_cap && !((e->h_tlive[e->t_lo >> 6] >> (e->t_lo & 63u)) & 1ull)) {  // dead rows in front
    e->t_lo++;
    e->t_dead--;
  }
  e->tprefix_dirty = true;
  e->h_tprefix_valid = false;
  if (n_deleted) *n_deleted = uint32_t(slots.size());
  pub_patch(e, nullptr);  // a deleted task (and the group that held it) is gone, the tasks behind it move up
  return PM_OK;
}

int32_t pm_on_worker_status(pm_engine* e, uint32_t worker, uint32_t flags_new, uint32_t dead) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  if (!e->have_workers || worker >= e->W) return set_error(PM_ERANGE, "worker index out of range");
  HIPCHK(hipSetDevice(e->cfg.device));
  ABSORB_PENDING(e);
  e->h_flags[worker] = flags_new;
  if (e->flags_delta_ok) e->delta_flags.push_back(worker);
  e->flags_dirty = true;   // uploaded once before the next kernel that reads the column (sync_flags)
  e->compat_dirty = true;  // HAS_SPECS etc. may have changed with the row
  if (dead && e->h_group_of[worker] >= 0) {  // status_update_impl.rs:17-29
    const uint32_t slot = uint32_t(e->h_group_of[worker]);
    const std::vector<uint32_t> members = e->groups[slot].members;
    dissolve_locked(e, slot);
    pub_patch(e, &members);  // its workers are in no group from now on, also for a look-up before the next tick
  }
  return PM_OK;
}

int32_t pm_on_worker_status_many(pm_engine* e, const uint32_t* workers, const uint32_t* flags_new, const uint32_t* dead,
                                 uint32_t n) {
  if (!e || (n && (!workers || !flags_new))) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  if (!e->have_workers) return set_error(PM_ERANGE, "worker index out of range");
  for (uint32_t k = 0; k < n; ++k)
    if (workers[k] >= e->W) return set_error(PM_ERANGE, "worker index out of range");
  if (!n) return PM_OK;
  HIPCHK(hipSetDevice(e->cfg.device));
#ifdef PM_HOST_MARKS_STATUS  // (a measuring build: where a status sweep's time goes)
  host_mark("status: begin");
#endif
  ABSORB_PENDING(e);
#ifdef PM_HOST_MARKS_STATUS
  host_mark("status: pending groups absorbed");
#endif
  std::vector<uint32_t> freed;  // members of the groups this sweep dissolves
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t w = workers[k];
    e->h_flags[w] = flags_new[k];
    if (e->flags_delta_ok) e->delta_flags.push_back(w);
    if (dead && dead[k] && e->h_group_of[w] >= 0) {  // status_update_impl.rs:17-29
      const uint32_t slot = uint32_t(e->h_group_of[w]);
      freed.insert(freed.end(), e->groups[slot].members.begin(), e->groups[slot].members.end());
      dissolve_locked(e, slot);
    }
  }
#ifdef PM_HOST_MARKS_STATUS
  host_mark("status: groups dissolved");
#endif
  e->flags_dirty = true;
  e->compat_dirty = true;
  if (!freed.empty()) pub_patch(e, &freed);
#ifdef PM_HOST_MARKS_STATUS
  host_mark("status: published rows patched");
#endif
  return PM_OK;
}

int32_t pm_enable_group_events(pm_engine* e, uint32_t on) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  ABSORB_PENDING(e);  // creations of a carve that has not been absorbed yet belong to the old setting
  e->events_on = on != 0;
  if (!e->events_on) {
    e->ev_log.clear();
    e->ev_members.clear();
  }
  return PM_OK;
}

int32_t pm_drain_group_events(pm_engine* e, pm_group_event* events, uint32_t cap_events, uint32_t* members,
                              uint32_t cap_members, uint32_t* n_events, uint32_t* n_members) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  ABSORB_PENDING(e);
  const uint32_t ne = uint32_t(e->ev_log.size()), nm = uint32_t(e->ev_members.size());
  if (n_events) *n_events = ne;
  if (n_members) *n_members = nm;
  if ((ne && (!events || cap_events < ne)) || (nm && (!members || cap_members < nm)))
    return set_error(PM_ERANGE, "event buffers too small");
  std::copy(e->ev_log.begin(), e->ev_log.end(), events);
  std::copy(e->ev_members.begin(), e->ev_members.end(), members);
  e->ev_log.clear();
  e->ev_members.clear();
  return PM_OK;
}

int32_t pm_dissolve_group(pm_engine* e, uint32_t slot) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  ABSORB_PENDING(e);
  compact_groups(e);  // `slot` is a number of the compacted list (what pm_get_groups reports)
  if (slot >= e->groups.size()) return set_error(PM_ERANGE, "group slot out of range");
  const std::vector<uint32_t> members = e->groups[slot].members;
  dissolve_locked(e, slot);
  pub_patch(e, &members);
  compact_groups(e);
  return PM_OK;
}

int32_t pm_reset_groups(pm_engine* e) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  reset_groups_locked(e);
  return PM_OK;
}

int32_t pm_compat_masks(pm_engine* e, uint64_t* mask_out) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  int32_t rc = ensure_compat(e);
  if (rc) return rc;
  if (mask_out && e->W) {
    HIPCHK(hipMemcpyAsync(mask_out, e->d_compat.p, size_t(e->W) * 8, hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return PM_OK;
}

int32_t pm_form_groups(pm_engine* e, uint32_t* n_formed) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  HIPCHK(hipSetDevice(e->cfg.device));
  e->tick_host_resolved = e->tick_carve_launches = e->tick_carve_steps = 0;
  e->tick_fast_steps = 0;
  e->tick_stream_aborts = 0;
  int32_t rc = run_form(e, n_formed);
  e->last_stats.carve_fast_steps = e->tick_fast_steps;
  e->last_stats.host_resolved_steps = e->tick_host_resolved;
  e->last_stats.carve_launches = e->tick_carve_launches;
  e->last_stats.carve_steps = e->tick_carve_steps;
  return rc;
}

int32_t pm_merge_solo_groups(pm_engine* e, uint32_t* n_merged) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  HIPCHK(hipSetDevice(e->cfg.device));
  e->tick_host_resolved = e->tick_carve_launches = e->tick_carve_steps = 0;
  int32_t rc = run_merge(e, n_merged);
  e->last_stats.host_resolved_steps = e->tick_host_resolved;
  e->last_stats.carve_launches = e->tick_carve_launches;
  e->last_stats.carve_steps = e->tick_carve_steps;
  return rc;
}

int32_t pm_get_groups(pm_engine* e, int32_t* group_of_worker, pm_group* groups, uint32_t cap_groups,
                      uint32_t* n_groups, uint32_t* members, uint32_t cap_members, uint32_t* n_members) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  ABSORB_PENDING(e);
  compact_groups(e);
  const uint32_t G = uint32_t(e->groups.size());
  uint32_t M = 0;
  for (const Group& g : e->groups) M += uint32_t(g.members.size());
  if (n_groups) *n_groups = G;
  if (n_members) *n_members = M;
  if (group_of_worker) std::copy(e->h_group_of.begin(), e->h_group_of.end(), group_of_worker);
  if (groups && cap_groups < G) return set_error(PM_ERANGE, "groups buffer too small");
  if (members && cap_members < M) return set_error(PM_ERANGE, "members buffer too small");
  uint32_t off = 0;
  for (uint32_t g = 0; g < G; ++g) {
    const Group& gr = e->groups[g];
    if (groups) {
      groups[g].id = gr.id;
      groups[g].config = gr.cfg;
      groups[g].n_members = uint32_t(gr.members.size());
      groups[g].member_begin = off;
      groups[g].task = task_position(e, gr.task);
    }
    if (members) {
      std::vector<uint32_t> m = gr.members;  // BTreeSet<String> order = address rank
      std::sort(m.begin(), m.end(), [&](uint32_t a, uint32_t b) { return e->h_addr_rank[a] < e->h_addr_rank[b]; });
      std::copy(m.begin(), m.end(), members + off);
    }
    off += uint32_t(gr.members.size());
  }
  return PM_OK;
}

int32_t pm_match(pm_engine* e, uint32_t* task_of_worker, uint32_t* applicable_count) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_world > 1) return set_error(PM_ESTATE, "multi-GPU engine: use the stepwise tick (pm_dist_tick_begin ...)");
  ABSORB_PENDING(e);
  std::vector<uint32_t> cnt;
  int32_t rc = run_match(e, applicable_count != nullptr, &cnt);
  if (rc) return rc;
  rc = publish(e);
  if (rc) return rc;
  if (task_of_worker)
    for (uint32_t w = 0; w < e->W; ++w) task_of_worker[w] = e->h_table[w].task;
  if (applicable_count) std::copy(cnt.begin(), cnt.end(), applicable_count);
  return PM_OK;
}

// north_star orientation: rows = tasks, swept axis = workers (the ones this rank owns in a multi-GPU set-up);
// leaves first / count per task in d_first / d_count (worker indices are global: every rank holds the whole table)
static int32_t run_match_per_task(pm_engine* e) {
  if (!e->have_tasks) return set_error(PM_ESTATE, "tasks must be uploaded first");
  if (e->dist_world > 1 && e->h_shard.size() != e->W)
    return set_error(PM_ESTATE, "the worker table changed size: call pm_dist_configure again");
  ABSORB_PENDING(e);
  int32_t rc = ensure_compat(e);
  if (rc) return rc;
  rc = push_groups(e);
  if (rc) return rc;
  const int variant = int(e->cfg.sweep_variant);
  const uint32_t n_planes = uint32_t(e->cfgs.size());
  rc = ensure_task_planes(e);  // (the live prefix: results are reported by list position)
  if (rc) return rc;
  const uint32_t R = e->t_cap - e->t_lo;  // rows of the sweep: the used part of the table, tombstones included
  rc = ensure_sweep_outputs(e, R);
  if (rc) return rc;
  HIPCHK(e->d_first_c.ensure(std::max<uint32_t>(e->T, 1)));
  HIPCHK(e->d_count_c.ensure(std::max<uint32_t>(e->T, 1)));
  HIPCHK(e->d_sel.ensure(std::max<uint32_t>(e->W, 1)));
  ensure_price_order(e);
  launch_eligible_selector(e->d_flags.p, e->d_group_of.p, e->d_compat.p, e->enabled, e->W,
                           e->dist_world > 1 ? e->d_shard.p : nullptr, e->dist_rank, e->d_sel.p, e->stream);
  const uint64_t* cols = e->d_sel.p;
  if (e->any_price) {  // order the swept axis by (price, index) so "first hit" is the best bid
    HIPCHK(e->d_perm.ensure(e->W));
    HIPCHK(e->d_sel_perm.ensure(e->W));
    std::vector<uint64_t> sel(e->W), selp(e->W);
    HIPCHK(hipMemcpyAsync(sel.data(), e->d_sel.p, size_t(e->W) * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (uint32_t i = 0; i < e->W; ++i) selp[i] = sel[e->price_perm[i]];
    HIPCHK(hipMemcpyAsync(e->d_sel_perm.p, selp.data(), size_t(e->W) * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    cols = e->d_sel_perm.p;
  }
  if (variant != 1) {
    const size_t n_words = (size_t(e->W) + 63) / 64;
    HIPCHK(e->d_wplanes.ensure(std::max<size_t>(n_words * (n_planes + 1), 1)));  // + the OR plane
    launch_build_planes(cols, e->W, 0, e->W, uint32_t(n_words), n_planes, e->d_wplanes.p, e->stream);
  }
  // Tasks that name the same set of configurations have the same bidders: sweep once per DISTINCT topology mask (a
  // few thousand at a million tasks) and let every task read its mask's result.  Worth it from a few hundred thousand
  // rows on (the table costs three small launches and one counter read: 0.19 against 0.13 ms at 100 k tasks, 1.5
  // against 3.2 ms at a million); a table that fills up (more than 2^16 distinct masks) falls back to one row per task.
  constexpr uint32_t H = 1u << 17, CAP_U = 1u << 16;
  bool per_mask = R >= 200000u && n_planes < 64u && variant != 1;
  uint32_t n_u = 0;
  if (per_mask) {
    const uint64_t valid = (1ull << n_planes) - 1ull;
    HIPCHK(e->d_ikeys.ensure(H));
    HIPCHK(e->d_ivals.ensure(H + 2));
    HIPCHK(e->d_umask.ensure(CAP_U));
    uint32_t* counter = e->d_ivals.p + H;  // [0] distinct masks, [1] table overflow
    launch_task_intern(e->d_tmask.p, e->d_tlive.p, e->t_lo, e->t_cap, valid, e->d_ikeys.p, H, e->d_ivals.p, counter,
                       e->d_umask.p, CAP_U, e->stream);
    uint32_t h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, counter, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    n_u = h[0];
    if (h[1] || n_u > CAP_U) per_mask = false;
    if (per_mask) {
      rc = ensure_sweep_outputs(e, std::max<uint32_t>(n_u, 1));
      if (rc) return rc;
      launch_pair_sweep(variant, e->d_umask.p, n_u, cols, e->d_wplanes.p, 0, e->W, uint32_t((size_t(e->W) + 63) / 64),
                        n_planes, e->d_first.p, e->d_count.p, e->stream);
      launch_task_compact_class(e->d_first.p, e->d_count.p, e->d_tmask.p, valid, e->d_ikeys.p, e->d_ivals.p, H, e->t_lo,
                                e->t_cap, e->d_tlive.p, e->d_tprefix.p, e->d_first_c.p, e->d_count_c.p, e->stream);
    }
  }
  if (!per_mask) {
    launch_pair_sweep(variant, e->d_tmask.p + e->t_lo, R, cols, e->d_wplanes.p, 0, e->W, uint32_t((size_t(e->W) + 63) / 64),
                      n_planes, e->d_first.p, e->d_count.p, e->stream);
    launch_task_compact(e->d_first.p, e->d_count.p, e->t_lo, e->t_cap, e->d_tlive.p, e->d_tprefix.p, e->d_first_c.p,
                        e->d_count_c.p, e->stream);
  }
  HIPCHK(hipGetLastError());
  return PM_OK;
}

int32_t pm_match_per_task(pm_engine* e, uint32_t* best_worker, uint32_t* candidate_count) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  HIPCHK(hipSetDevice(e->cfg.device));
  int32_t rc = run_match_per_task(e);
  if (rc) return rc;
  if (best_worker && e->T)
    HIPCHK(hipMemcpyAsync(best_worker, e->d_first_c.p, size_t(e->T) * 4, hipMemcpyDeviceToHost, e->stream));
  if (candidate_count && e->T)
    HIPCHK(hipMemcpyAsync(candidate_count, e->d_count_c.p, size_t(e->T) * 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (best_worker && e->any_price)
    for (uint32_t t = 0; t < e->T; ++t)
      if (best_worker[t] != PM_NONE) best_worker[t] = e->price_perm[best_worker[t]];
  return PM_OK;
}

int32_t pm_match_per_task_device(pm_engine* e, uint64_t* best_ptr, uint64_t* count_ptr, uint32_t* n) {
  if (!e || !best_ptr || !count_ptr || !n) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  HIPCHK(hipSetDevice(e->cfg.device));
  ensure_price_order(e);
  if (e->any_price) return set_error(PM_ESTATE, "device-side bids are index-ordered: not available with a price column");
  int32_t rc = run_match_per_task(e);
  if (rc) return rc;
  if (e->own_stream) HIPCHK(hipStreamSynchronize(e->stream));  // a caller-supplied stream orders the consumer itself
  *best_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->d_first_c.p));
  *count_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->d_count_c.p));
  *n = e->T;
  return PM_OK;
}

int32_t pm_newest_task(pm_engine* e, uint32_t* task_idx) {
  if (!e || !task_idx) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (!e->have_tasks) return set_error(PM_ESTATE, "tasks must be uploaded first");
  *task_idx = PM_NONE;
  if (e->T == 0) return PM_OK;
  const uint32_t nb = std::min<uint32_t>(1024, (e->t_cap - e->t_lo + 255u) / 256u);
  HIPCHK(e->d_nb_idx.ensure(nb));
  HIPCHK(e->d_nb_val.ensure(nb));
  launch_newest(e->d_created.p, e->d_tlive.p, e->t_lo, e->t_cap, e->d_nb_idx.p, e->d_nb_val.p, nb, e->stream);
  HIPCHK(hipGetLastError());
  std::vector<uint32_t> bi(nb);
  std::vector<long long> bv(nb);
  HIPCHK(hipMemcpyAsync(bi.data(), e->d_nb_idx.p, nb * 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(bv.data(), e->d_nb_val.p, nb * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  uint32_t best = PM_NONE;
  long long bval = 0;
  for (uint32_t k = 0; k < nb; ++k) {
    if (bi[k] == PM_NONE) continue;
    if (best == PM_NONE || bv[k] > bval || (bv[k] == bval && bi[k] > best)) {
      best = bi[k];
      bval = bv[k];
    }
  }
  *task_idx = task_position(e, best);
  return PM_OK;
}

static void tick_reset(pm_engine* e) {
  e->tick_host_resolved = e->tick_carve_launches = e->tick_carve_steps = 0;
  e->tick_fast_steps = 0;
  e->tick_cand_sum = 0;
  e->tick_props = 0;
  e->tick_prop_keys = 0;
  e->tick_stream_timeouts = e->tick_stream_tickets = e->tick_stream_aborts = 0;
  e->k_ms_propose = 0;
  e->prop_ev_used = 0;
  e->k_ms_compat = e->k_ms_carve = e->k_ms_sweep = 0;
  e->k_sweep_recorded = e->k_compat_recorded = false;
}

static int32_t tick_stats(pm_engine* e, pm_stats* stats, uint32_t n_formed, uint32_t n_merged) {
  HIPCHK(hipEventRecord(e->ev[5], e->stream));
  HIPCHK(hipEventSynchronize(e->ev[5]));
  pm_stats s{};
  HIPCHK(hipEventElapsedTime(&s.ms_compat, e->ev[0], e->ev[1]));
  HIPCHK(hipEventElapsedTime(&s.ms_carve, e->ev[1], e->ev[2]));
  HIPCHK(hipEventElapsedTime(&s.ms_merge, e->ev[2], e->ev[3]));
  HIPCHK(hipEventElapsedTime(&s.ms_sweep, e->ev[3], e->ev[4]));
  HIPCHK(hipEventElapsedTime(&s.ms_publish, e->ev[4], e->ev[5]));
  HIPCHK(hipEventElapsedTime(&s.ms_total, e->ev[0], e->ev[5]));
  if (e->k_compat_recorded) HIPCHK(hipEventElapsedTime(&s.ms_compat_kernel, e->kev[0], e->kev[1]));
  if (e->k_sweep_recorded) HIPCHK(hipEventElapsedTime(&s.ms_sweep_kernel, e->kev[4], e->kev[5]));
  s.ms_carve_kernel = e->k_ms_carve;
  s.carve_cand_sum = e->tick_cand_sum;
  s.ms_propose_kernel = e->k_ms_propose;
  s.proposals = e->tick_props;
  s.propose_keys = e->tick_prop_keys;
  s.n_groups = uint32_t(e->groups.size() - e->n_dead_groups);
  s.n_formed = n_formed;
  s.n_merged = n_merged;
  s.carve_steps = e->tick_carve_steps;
  s.carve_fast_steps = e->tick_fast_steps;
  s.host_resolved_steps = e->tick_host_resolved;
  s.carve_launches = e->tick_carve_launches;
  s.pair_evals = uint64_t(e->T) * uint64_t(e->W);
  e->last_stats = s;
  if (stats) *stats = s;
  return PM_OK;
}

int32_t pm_tick(pm_engine* e, pm_stats* stats) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (!e->have_cfgs || !e->have_workers || !e->have_tasks)
    return set_error(PM_ESTATE, "configs, workers and tasks must be uploaded first");
  if (e->dist_world > 1) return set_error(PM_ESTATE, "multi-GPU engine: use the stepwise tick (pm_dist_tick_begin ...)");
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  tick_reset(e);
  uint32_t n_formed = 0, n_merged = 0;
  HIPCHK(hipEventRecord(e->ev[0], e->stream));
  e->compat_dirty = true;  // a full-swarm match re-evaluates the W x C predicate, like mod.rs:511-515
  // (the group deltas of a churn tick first: their staging waits for the stream to drain — an idle stream answers at
  // once, one that has just been handed the compat sweep of 100,000 rows answers 100 us later)
  int32_t rc = absorb_groups(e);  // (form_begin's order: a carve left unabsorbed first, then the list goes up)
  if (rc) return rc;
  rc = push_groups(e);
  if (rc) return rc;
  rc = ensure_compat(e);
  if (rc) return rc;
  HIPCHK(hipEventRecord(e->ev[1], e->stream));
  // The host copy of the new groups is built while the pair sweep runs, unless the merge pass needs it.
  host_mark("tick: compat queued");
  FormRun r;
  rc = form_begin(e, &r, /*allow_pipeline=*/true);
  if (rc) return rc;
  // ---- the streaming carve: everything behind it is queued BEFORE the host waits for any of it.  The carve's last
  // kernel has completed the group records on the device (ids, empty task words) and written their host copy and the
  // status into pinned memory itself, so behind it come — without a copy or a host round trip in between — the pair
  // sweep, the claim, and the copies of the table: ONE wait per match instead of two, and nothing on the stream while the
  // host sleeps and wakes (30 us each way) or walks through a dozen enqueue calls.  The bet is that the carve ends DONE in
  // its one launch and leaves fewer than two single-node groups (no merge pass): if not, the queued work was for nothing
  // — it changes no state the engine keeps — and the general path below takes over from the status the launch left.
  bool done = false;
  if (r.stream && !r.nothing && !e->tick_needs_merge) {
    PubRun pr;
    HIPCHK(hipEventRecord(e->kev[3], e->stream));
    HIPCHK(hipEventRecord(e->ev_groups, e->stream));  // the carve is through: records and status are in host memory
    HIPCHK(hipEventRecord(e->ev[2], e->stream));
    HIPCHK(hipEventRecord(e->ev[3], e->stream));
    const size_t G_ub = std::min<size_t>(size_t(r.g0) + r.stage_cap, std::min(e->d_g_task.cap, e->d_g_task_next.cap));
    rc = run_match(e, false, nullptr, false, G_ub);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->ev[4], e->stream));
    rc = publish_begin(e, &pr, G_ub);
    if (rc) return rc;
    host_mark("tick: everything queued");
    // the host's copy of the new groups is built while the pair sweep and the claim run
    if (hipEventSynchronize(e->ev_groups) != hipSuccess) {
      publish_abandon(e, pr);
      return set_error(PM_ENODEV, "tick: the stream failed");
    }
    host_mark("tick: carve through");
    const CarveStatus& hs = *e->h_status;
    size_t solo = hs.n_solo;
    for (const Group& g : e->groups) solo += g.members.size() == 1 && !g.dead;
    const bool fits = hs.n_groups >= r.g0 && hs.n_groups - r.g0 <= r.stage_cap && hs.n_members >= r.m0 &&
                      hs.n_members - r.m0 <= r.stage_cap && hs.n_groups <= G_ub;
    if (hs.state == CARVE_STATE_DONE && fits && solo < 2) {
      r.st = hs;
      rc = form_finish(e, &r, &n_formed, /*defer_absorb=*/false, /*have_event=*/true);
      if (rc) {
        publish_abandon(e, pr);
        return rc;
      }
      host_mark("tick: groups absorbed");
      pr.G = e->groups.size();
      rc = publish_end(e, pr);  // (waits for the stream: the table's copies)
      host_mark("tick: published");
      if (rc) return rc;
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, e->kev[2], e->kev[3]));
      e->k_ms_carve += ms;
      done = true;
    } else {
      publish_abandon(e, pr);
      e->tick_needs_merge = solo >= 2;
    }
  }
  if (!done) {
    rc = run_form_rest(e, r, &n_formed, /*defer_absorb=*/true);
    if (rc) return rc;
    host_mark("tick: form done");
    HIPCHK(hipEventRecord(e->ev[2], e->stream));
    rc = run_merge(e, &n_merged);
    if (rc) return rc;
    host_mark("tick: merge done");
    HIPCHK(hipEventRecord(e->ev[3], e->stream));
    rc = run_match(e, false, nullptr);
    if (rc) return rc;
    host_mark("tick: match queued");
    HIPCHK(hipEventRecord(e->ev[4], e->stream));
    // the table's copies travel while the host builds its copy of the new groups
    PubRun pr;
    rc = publish_begin(e, &pr, e->d_n_groups);
    if (rc) return rc;
    host_mark("tick: table copies queued");
    rc = absorb_groups(e);
    if (rc) {
      publish_abandon(e, pr);
      return rc;
    }
    host_mark("tick: groups absorbed");
    rc = publish_end(e, pr);
    host_mark("tick: published");
    if (rc) return rc;
  }
  return tick_stats(e, stats, n_formed, n_merged);
}

// Several pools, one call.  A match is mostly one long launch the host waits for (the streaming carve), so an
// orchestrator process that serves K pools (K engines) gains nothing from calling pm_tick K times in a row.  Here ONE
// thread walks the engines three times — start every engine's carve on its own stream; as the carves finish, queue each
// engine's group records, pair sweep and claim; take the results in and publish — so the K carve launches are resident
// side by side (give every engine its share of the CUs first: pm_set_carve_workgroups; give the process enough hardware
// queues: GPU_MAX_HW_QUEUES >= 2 K, see include/pm_engine.h — with the runtime's 4 the engines' streams share queues
// and the launches run in turn whoever starts them) and the host is never inside two HIP calls at once.  Per engine
// the sequence of device work is exactly pm_tick's: same kernels, same order, same stream.  With 16 queues: K = 4
// 2.8x, K = 8 3.1x the one-pool rate (K host threads calling pm_tick: 3.3x and 1.6x from Python, 3.0x and 2.4x from
// threads inside the library).
//   PM_TICK_MANY_THREADS: one host thread per engine, each calling pm_tick (kept for that comparison).
int32_t pm_tick_many(pm_engine* const* engines, uint32_t n, pm_stats* stats, uint32_t flags) {
  if (!engines || n == 0) return set_error(PM_EINVAL, "null argument");
  if (n > 1024u) return set_error(PM_EINVAL, "more than 1024 engines in one call");
  if (flags & ~uint32_t(PM_TICK_MANY_THREADS)) return set_error(PM_EINVAL, "unknown flag");
  for (uint32_t i = 0; i < n; ++i) {
    if (!engines[i]) return set_error(PM_EINVAL, "null engine");
    for (uint32_t j = 0; j < i; ++j)
      if (engines[j] == engines[i]) return set_error(PM_EINVAL, "the same engine twice in one call");
  }
  if (flags & PM_TICK_MANY_THREADS) {
    std::vector<int32_t> rcs(n, PM_OK);
    std::vector<std::string> msgs(n);
    std::vector<std::thread> th;
    th.reserve(n);
    for (uint32_t i = 0; i < n; ++i)
      th.emplace_back([&, i] {
        rcs[i] = pm_tick(engines[i], stats ? &stats[i] : nullptr);
        if (rcs[i]) msgs[i] = g_last_error;  // (thread-local: carried to the caller's thread below)
      });
    for (std::thread& t : th) t.join();
    for (uint32_t i = 0; i < n; ++i)
      if (rcs[i]) return set_error(rcs[i], msgs[i]);
    return PM_OK;
  }
  // every engine's lock for the whole call, taken in address order (two overlapping calls cannot cross)
  std::vector<pm_engine*> by_addr(engines, engines + n);
  std::sort(by_addr.begin(), by_addr.end(), [](const pm_engine* a, const pm_engine* b) { return std::less<const pm_engine*>()(a, b); });
  std::vector<std::unique_lock<std::mutex>> locks;
  locks.reserve(n);
  for (pm_engine* e : by_addr) locks.emplace_back(e->mu);
  for (uint32_t i = 0; i < n; ++i) {
    pm_engine* e = engines[i];
    if (!e->have_cfgs || !e->have_workers || !e->have_tasks)
      return set_error(PM_ESTATE, "configs, workers and tasks must be uploaded first");
    if (e->dist_world > 1) return set_error(PM_ESTATE, "multi-GPU engine: use the stepwise tick (pm_dist_tick_begin ...)");
    if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  }
  std::vector<std::unique_ptr<FormRun>> runs(n);
  std::vector<PubRun> pubs(n);
  std::vector<uint32_t> n_formed(n, 0), n_merged(n, 0);
  // An engine that fails leaves the walk (its state is what a failed pm_tick leaves); the others finish their tick.
  // The call reports the first failure.
  std::vector<int32_t> rcs(n, PM_OK);
  std::vector<std::string> msgs(n);
  auto failed = [&](uint32_t i, int32_t rc) {
    rcs[i] = rc;
    msgs[i] = g_last_error;
  };
  const uint32_t i_lo = 0, i_hi = n;
  {
  // ---- 1: compatibility masks, the eligible list, the carve's launch(es) — nothing here waits for a carve
  for (uint32_t i = i_lo; i < i_hi; ++i) {
    pm_engine* e = engines[i];
    auto stage = [&]() -> int32_t {
      HIPCHK(hipSetDevice(e->cfg.device));
      tick_reset(e);
      HIPCHK(hipEventRecord(e->ev[0], e->stream));
      e->compat_dirty = true;
      int32_t rc = ensure_compat(e);
      if (rc) return rc;
      HIPCHK(hipEventRecord(e->ev[1], e->stream));
      runs[i].reset(new (std::nothrow) FormRun());
      if (!runs[i]) return set_error(PM_ENOMEM, "out of host memory");
      return form_begin(e, runs[i].get(), /*allow_pipeline=*/true);
    };
    const int32_t rc = stage();
    if (rc) failed(i, rc);
  }
  // ---- 2: in launch order (the first carve started is the first to end): the carve's result, the merge pass, the
  // pair sweep and the claim, queued behind it on the engine's stream
  for (uint32_t i = i_lo; i < i_hi; ++i) {
    if (rcs[i]) continue;
    pm_engine* e = engines[i];
    auto stage = [&]() -> int32_t {
      HIPCHK(hipSetDevice(e->cfg.device));
      int32_t rc = run_form_rest(e, *runs[i], &n_formed[i], /*defer_absorb=*/true);
      if (rc) return rc;
      HIPCHK(hipEventRecord(e->ev[2], e->stream));
      rc = run_merge(e, &n_merged[i]);
      if (rc) return rc;
      HIPCHK(hipEventRecord(e->ev[3], e->stream));
      rc = run_match(e, false, nullptr);
      if (rc) return rc;
      HIPCHK(hipEventRecord(e->ev[4], e->stream));
      return publish_begin(e, &pubs[i], e->d_n_groups);  // (the table's copies: behind the claim, in front of the host's wait)
    };
    const int32_t rc = stage();
    runs[i].reset();
    if (rc) failed(i, rc);
  }
  // ---- 3: the host copy of the new groups, the published table
  for (uint32_t i = i_lo; i < i_hi; ++i) {
    if (rcs[i]) continue;
    pm_engine* e = engines[i];
    auto stage = [&]() -> int32_t {
      HIPCHK(hipSetDevice(e->cfg.device));
      int32_t rc = absorb_groups(e);
      if (rc) {
        publish_abandon(e, pubs[i]);
        return rc;
      }
      rc = publish_end(e, pubs[i]);
      if (rc) return rc;
      return tick_stats(e, stats ? &stats[i] : nullptr, n_formed[i], n_merged[i]);
    };
    const int32_t rc = stage();
    if (rc) failed(i, rc);
  }
  }  // (the chunk, or the one block that holds all three walks)
  for (uint32_t i = 0; i < n; ++i)
    if (rcs[i]) return set_error(rcs[i], msgs[i]);
  return PM_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU: ownership, the stepwise tick and its two exchanges (see include/pm_engine.h)

// Pools sharing one GPU: the carve's launch keeps one workgroup per CU resident from its first configuration to its
// last (the validator + its row-making workgroups), so K engines matching at the same time fit side by side only if
// each asks for its share of the CUs.  0 = by the size of the eligible list (a pool with the GPU to itself).
int32_t pm_set_carve_workgroups(pm_engine* e, uint32_t n) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  if (n > 4096u) return set_error(PM_EINVAL, "carve workgroups 0..4096");
  std::lock_guard<std::mutex> lk(e->mu);
  e->stream_wgs_env = n;
  return PM_OK;
}

int32_t pm_set_stream(pm_engine* e, void* hip_stream) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  HIPCHK(hipStreamSynchronize(e->stream));
  if (hip_stream) {
    e->stream = static_cast<hipStream_t>(hip_stream);
    e->own_stream = false;
  } else {
    e->stream = e->stream_owned;
    e->own_stream = true;
  }
  return PM_OK;
}

static void dist_abort(pm_engine* e) {
  delete e->form;
  e->form = nullptr;
  e->dist_phase = 0;
}

int32_t pm_dist_configure(pm_engine* e, uint32_t rank, uint32_t world, const uint8_t* shard_of_worker) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  if (world == 0 || world > 64 || rank >= world) return set_error(PM_EINVAL, "rank / world out of range (world <= 64)");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 0) return set_error(PM_ESTATE, "a stepwise tick is in progress");
  if (world > 1 && !e->have_workers) return set_error(PM_ESTATE, "workers must be uploaded first");
  if (world > 1 && e->W && !shard_of_worker) return set_error(PM_EINVAL, "null shard column");
  e->dist_rank = world > 1 ? rank : 0;
  e->dist_world = world;
  e->h_shard.clear();
  e->h_own_rows.clear();
  e->dist_cap_t = 0;
  if (world == 1) return PM_OK;
  const uint32_t W = e->W;
  for (uint32_t w = 0; w < W; ++w)
    if (shard_of_worker[w] >= world) return set_error(PM_ERANGE, "shard index outside the world");
  e->h_shard.assign(shard_of_worker, shard_of_worker + W);
  std::vector<uint32_t> count(world, 0), xrow(W);
  for (uint32_t w = 0; w < W; ++w) xrow[w] = count[e->h_shard[w]]++;  // index within the shard, in worker order
  const uint32_t cap_t = std::max<uint32_t>(*std::max_element(count.begin(), count.end()), 1u);
  for (uint32_t w = 0; w < W; ++w) {
    xrow[w] += uint32_t(e->h_shard[w]) * cap_t;
    if (e->h_shard[w] == rank) e->h_own_rows.push_back(w);
  }
  e->dist_cap_t = cap_t;
  int32_t rc = upload(e->d_shard, e->h_shard.data(), W, e->stream);
  if (rc) return rc;
  rc = upload(e->d_own_rows, e->h_own_rows.data(), e->h_own_rows.size(), e->stream);
  if (rc) return rc;
  rc = upload(e->d_xrow, xrow.data(), W, e->stream);
  if (rc) return rc;
  HIPCHK(e->d_table_x.ensure(size_t(cap_t) * world));
  HIPCHK(hipMemsetAsync(e->d_table_x.p, 0xFF, sizeof(pm_assignment) * size_t(cap_t) * world, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));  // the staging vectors die here
  return PM_OK;
}

int32_t pm_dist_tick_begin(pm_engine* e) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (!e->have_cfgs || !e->have_workers || !e->have_tasks)
    return set_error(PM_ESTATE, "configs, workers and tasks must be uploaded first");
  if (e->dist_phase != 0) dist_abort(e);  // an abandoned stepwise tick
  if (e->dist_world > 1 && e->h_shard.size() != e->W)
    return set_error(PM_ESTATE, "the worker table changed size: call pm_dist_configure again");
  tick_reset(e);
  HIPCHK(hipEventRecord(e->ev[0], e->stream));
  e->compat_dirty = true;
  int32_t rc = ensure_compat(e);
  if (rc) return rc;
  HIPCHK(hipEventRecord(e->ev[1], e->stream));
  e->form = new (std::nothrow) FormRun();
  if (!e->form) return set_error(PM_ENOMEM, "out of host memory");
  // The carve is REPLICATED: every rank runs the whole of it — the streaming launch, as on one GPU — and ends with the
  // identical groups and ids, because the result does not depend on how the launch went (which rows arrived when), only
  // on the reference's rule: nothing is exchanged until the published table.
  rc = form_begin(e, e->form, /*allow_pipeline=*/true, /*local_carve=*/true);
  if (rc) {
    dist_abort(e);
    return rc;
  }
  e->dist_phase = 1;
  return PM_OK;
}

int32_t pm_dist_carve_next(pm_engine* e, pm_dist_xfer* x, uint32_t* more) {
  if (!e || !x || !more) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 1) return set_error(PM_ESTATE, "pm_dist_tick_begin first");
  FormRun* r = e->form;
  *more = 0;
  std::memset(x, 0, sizeof(*x));
  if (!r->nothing) {
    // (kept in the protocol for a local compute that deals a batch's rows over the ranks — tests/dist_model.py does, the
    // engine did until round 5: *more stays 0, there is nothing to exchange for the carve)
    int32_t rc = run_form_wait(e, *r);
    if (rc) {
      dist_abort(e);
      return rc;
    }
    if (r->st.state != CARVE_STATE_DONE) {
      dist_abort(e);
      return set_error(PM_ENODEV, "carve kernel did not complete");
    }
  }
  e->dist_phase = 2;
  return PM_OK;
}

int32_t pm_dist_carve_validate(pm_engine* e) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  (void)e;
  return set_error(PM_ESTATE, "no proposal batch pending (the carve of the multi-GPU tick is replicated: pm_dist_carve_next reports none)");
}

int32_t pm_dist_match_begin(pm_engine* e, pm_dist_xfer* x) {
  if (!e || !x) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 2) return set_error(PM_ESTATE, "the carve is not finished (pm_dist_carve_next until more == 0)");
  std::memset(x, 0, sizeof(*x));
  int32_t rc = form_finish(e, e->form, &e->dist_n_formed, /*defer_absorb=*/true);
  delete e->form;
  e->form = nullptr;
  if (rc == PM_OK) {
    HIPCHK(hipEventRecord(e->ev[2], e->stream));
    rc = run_merge(e, &e->dist_n_merged);
  }
  if (rc == PM_OK) {
    HIPCHK(hipEventRecord(e->ev[3], e->stream));
    rc = run_match(e, false, nullptr, e->dist_world > 1);
  }
  if (rc) {
    e->dist_phase = 0;
    return rc;
  }
  HIPCHK(hipEventRecord(e->ev[4], e->stream));
  if (e->dist_world > 1) {
    x->recv_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->d_table_x.p));
    x->send_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->d_table_x.p + size_t(e->dist_rank) * e->dist_cap_t));
    x->bytes_per_rank = uint64_t(e->dist_cap_t) * sizeof(pm_assignment);
  }
  e->dist_phase = 3;
  return PM_OK;
}

int32_t pm_dist_tick_end(pm_engine* e, pm_stats* stats) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->dist_phase != 3) return set_error(PM_ESTATE, "pm_dist_match_begin first");
  e->dist_phase = 0;
  if (e->dist_world > 1) {
    launch_table_scatter(e->d_table_x.p, e->d_xrow.p, e->W, e->d_table.p, e->d_task_col.p, e->d_g_task_next.p,
                         e->d_tlive.p, e->d_tprefix.p, e->stream);
    HIPCHK(hipGetLastError());
  }
  int32_t rc = absorb_groups(e);
  if (rc) return rc;
  rc = publish(e);
  if (rc) return rc;
  return tick_stats(e, stats, e->dist_n_formed, e->dist_n_merged);
}

int32_t pm_lookup_task_for_worker(pm_engine* e, uint32_t worker, pm_assignment* out) {
  if (!e || !out) return set_error(PM_EINVAL, "null argument");
  for (;;) {
    const int cur = e->pub_cur.load(std::memory_order_acquire);
    if (cur < 0) return set_error(PM_ESTATE, "no assignment table published yet");
    const PubTable& t = e->pub[cur];
    const uint64_t s1 = t.seq.load(std::memory_order_acquire);
    if (s1 & 1u) continue;  // two publishes since `cur` was read: take the newer buffer
    const uint64_t* words = t.words.load(std::memory_order_relaxed);
    const uint32_t n = t.n.load(std::memory_order_relaxed);
    const uint32_t shift = t.task_shift.load(std::memory_order_relaxed);
    const uint32_t cleared = t.cleared.load(std::memory_order_relaxed);
    uint64_t row[4] = {0, 0, 0, 0};
    const bool in_range = worker < n;
    if (in_range)
      for (int k = 0; k < 4; ++k) row[k] = __atomic_load_n(&words[size_t(worker) * 4 + k], __ATOMIC_RELAXED);
    std::atomic_thread_fence(std::memory_order_acquire);
    if (t.seq.load(std::memory_order_relaxed) != s1) continue;
    if (!in_range) return set_error(PM_ERANGE, "worker index out of range");
    std::memcpy(out, row, sizeof(*out));
    if (cleared) {  // (pm_reset_groups since this buffer was written)
      std::memset(out, 0, sizeof(*out));
      out->task = PM_NONE;
      out->group_slot = PM_NONE;
      out->next_worker = PM_NONE;
    }
    if (out->task != PM_NONE) out->task += shift;
    return PM_OK;
  }
}

int32_t pm_device_task_column(pm_engine* e, uint64_t* device_ptr, uint32_t* n) {
  if (!e || !device_ptr || !n) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->d_task_col.p) return set_error(PM_ESTATE, "no assignment table computed yet");
  *device_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->d_task_col.p));
  *n = e->W;
  return PM_OK;
}

// debug (include/pm_engine_debug.h): counters of the last carve; copies min(cap, 72) words
int32_t pm_debug_carve_prof(pm_engine* e, unsigned long long* out, uint32_t cap) {
  if (!e || !out) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  const uint32_t n = std::min<uint32_t>(cap, 32u);
  std::memcpy(out, e->carve_prof, size_t(n) * sizeof(unsigned long long));
  for (uint32_t k = 32; k < cap && k < 56; ++k) out[k] = e->carve_why[k - 32];
  for (uint32_t k = 56; k < cap && k < 88; ++k) out[k] = e->carve_prof[k - 56 + 32];  // (phase counters 32..47)  // (how the validation launches ended; the index)
  return PM_OK;
}

// debug (PM_CARVE_PROF builds): the timeline of the last streaming carve launch — up to cap events of two u64 each
// {s_memtime, type | a << 8 | b << 32}; *n = events recorded
int32_t pm_debug_stream_trace(pm_engine* e, unsigned long long* out, uint32_t cap, uint32_t* n) {
  if (!e || !out || !n) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  *n = 0;
  if (!e->d_stream_trace.p || !e->d_stream_ctl.p) return PM_OK;
  uint32_t cnt = 0;
  HIPCHK(hipMemcpy(&cnt, e->d_stream_ctl.p + SC_TRACE, 4, hipMemcpyDeviceToHost));
#ifdef PM_ROW_REC  // (the validator's events live in the second half of the buffer there)
  cnt = std::min<uint32_t>(cnt, std::min<uint32_t>(cap, PM_STREAM_TRACE_CAP / 2u));
  if (cnt) HIPCHK(hipMemcpy(out, e->d_stream_trace.p + PM_STREAM_TRACE_CAP, size_t(cnt) * 16, hipMemcpyDeviceToHost));
#else
  cnt = std::min<uint32_t>(cnt, std::min<uint32_t>(cap, PM_STREAM_TRACE_CAP));
  if (cnt) HIPCHK(hipMemcpy(out, e->d_stream_trace.p, size_t(cnt) * 16, hipMemcpyDeviceToHost));
#endif
  *n = cnt;
  return PM_OK;
}

// debug (include/pm_engine_debug.h): candidate lists longer than `n` slots take the all-in-HBM carve path (carve_step_mem), which
// otherwise needs more than 262,144 candidates for one configuration; 0 = off
// debug (PM_ROW_REC builds, tools/row_rec.py): the time stamps the row makers of the last streaming launch left, eight words
// per ticket (see stream_proposer); 0 rows from any other build
int32_t pm_debug_row_records(pm_engine* e, unsigned long long* out, uint32_t cap_rows, uint32_t* n_rows) {
  if (!e || !out || !n_rows) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  *n_rows = 0;
#ifdef PM_ROW_REC
  if (!e->d_stream_trace.p) return PM_OK;
  const uint32_t n = std::min<uint32_t>(cap_rows, PM_STREAM_TRACE_CAP / 8u);
  HIPCHK(hipMemcpy(out, e->d_stream_trace.p, size_t(n) * 64, hipMemcpyDeviceToHost));
  *n_rows = n;
#endif
  return PM_OK;
}
int32_t pm_debug_mem_lists_above(pm_engine* e, uint32_t n) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  e->debug_mem_above = n;
  return PM_OK;
}

// debug (include/pm_engine_debug.h): the streaming carve's chain gives its launch up (CARVE_STATE_ABORTED, as a lost
// hand-shake inside the validator would) once n steps of the carve are committed; the engine continues on the batch
// pipeline from there (form_poll).  0 = off
int32_t pm_debug_stream_abort_after(pm_engine* e, uint32_t n) {
  if (!e) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  e->debug_abort_after = n;
  return PM_OK;
}

// debug (include/pm_engine_debug.h): merge configurations (pm_merge_solo_groups, pm_tick) whose selections went through the
// streaming carve since the engine was created
int32_t pm_debug_merge_streamed(pm_engine* e, uint32_t* n) {
  if (!e || !n) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  *n = e->merge_streamed;
  return PM_OK;
}

// debug (include/pm_engine_debug.h): times the device's group state was brought up to date by a delta (dissolved groups'
// workers + new rows) instead of the whole list, since the engine was created
int32_t pm_debug_delta_pushes(pm_engine* e, uint32_t* n) {
  if (!e || !n) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  *n = e->delta_pushes;
  return PM_OK;
}

// debug (include/pm_engine_debug.h): when the proposer walks the spatial index instead of sweeping the whole candidate list —
// 0 never, 1 when it pays (default), 2 whenever the carve has one (built for any swarm of 64+ positions),
// 3 = 2 with every seed sent through the whole-list fallback
int32_t pm_debug_prune_mode(pm_engine* e, uint32_t mode) {
  if (!e || mode > 3u) return set_error(PM_EINVAL, "prune mode 0..3");
  std::lock_guard<std::mutex> lk(e->mu);
  e->prune_mode = mode;
  return PM_OK;
}

#ifdef PM_BATCH_LOG
extern "C" int32_t pm_debug_batch_log(pm_engine* e, uint32_t* out, uint32_t cap_words, uint32_t* n_words) {
  if (!e || !out || !n_words) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  *n_words = uint32_t(e->blog.size());
  std::memcpy(out, e->blog.data(), sizeof(uint32_t) * std::min<size_t>(cap_words, e->blog.size()));
  return PM_OK;
}
#endif

// debug (include/pm_engine_debug.h): stream triad over 3 x n_doubles f64 on the engine's stream, best of `reps` -> GB/s
int32_t pm_debug_hbm_triad(pm_engine* e, uint64_t n_doubles, uint32_t reps, double* gb_per_s) {
  if (!e || !gb_per_s || !n_doubles) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  double *a = nullptr, *b = nullptr, *c = nullptr;
  HIPCHK(hipMalloc((void**)&a, n_doubles * 8));
  if (hipMalloc((void**)&b, n_doubles * 8) != hipSuccess || hipMalloc((void**)&c, n_doubles * 8) != hipSuccess) {
    (void)hipFree(a);
    if (b) (void)hipFree(b);
    return set_error(PM_ENOMEM, "triad buffers");
  }
  (void)hipMemsetAsync(b, 0, n_doubles * 8, e->stream);
  (void)hipMemsetAsync(c, 0, n_doubles * 8, e->stream);
  float best = 1e30f;
  for (uint32_t r = 0; r < reps + 1; ++r) {
    (void)hipEventRecord(e->kev[0], e->stream);
    launch_triad(b, c, a, size_t(n_doubles), e->stream);
    (void)hipEventRecord(e->kev[1], e->stream);
    (void)hipEventSynchronize(e->kev[1]);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e->kev[0], e->kev[1]);
    if (r > 0 && ms < best) best = ms;
  }
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipFree(c);
  *gb_per_s = double(n_doubles) * 24.0 / (double(best) * 1e-3) / 1e9;
  return PM_OK;
}

// debug (include/pm_engine_debug.h): rows built by the insertion and by the networks from the same keys, compared on the
// device (mismatches[0]: 1 = a register differs, 2 = a threshold, 4 = what a tracker answers; [1] = rows that differ); the
// networks' rows come back for the caller's own sort
int32_t pm_debug_row_networks(pm_engine* e, const uint64_t* keys, const uint32_t* sites, uint32_t n_waves, uint32_t n_per_wave,
                              uint32_t slot_bits, uint64_t ulps, uint32_t upto, uint64_t* rows_out, uint32_t* mismatches) {
  if (!e || !keys || !sites || !rows_out || !mismatches || !n_waves || (n_waves & 3u) || !n_per_wave || slot_bits == 0 || slot_bits > 24)
    return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->cfg.device));
  const size_t nk = size_t(n_waves) * n_per_wave, ns = size_t(1) << slot_bits;
  uint64_t *d_k = nullptr, *d_rows = nullptr;
  uint32_t *d_s = nullptr, *d_m = nullptr;
  int32_t rc = PM_OK;
  if (hipMalloc((void**)&d_k, nk * 8) != hipSuccess || hipMalloc((void**)&d_rows, size_t(n_waves) * 64 * 8) != hipSuccess ||
      hipMalloc((void**)&d_s, ns * 4) != hipSuccess || hipMalloc((void**)&d_m, 8) != hipSuccess) {
    rc = set_error(PM_ENOMEM, "row network test buffers");
  } else {
    (void)hipMemcpyAsync(d_k, keys, nk * 8, hipMemcpyHostToDevice, e->stream);
    (void)hipMemcpyAsync(d_s, sites, ns * 4, hipMemcpyHostToDevice, e->stream);
    (void)hipMemsetAsync(d_m, 0, 8, e->stream);
    launch_row_network_test(d_k, d_s, n_waves, n_per_wave, slot_bits, ulps, upto, d_rows, d_m, e->stream);
    (void)hipMemcpyAsync(rows_out, d_rows, size_t(n_waves) * 64 * 8, hipMemcpyDeviceToHost, e->stream);
    (void)hipMemcpyAsync(mismatches, d_m, 8, hipMemcpyDeviceToHost, e->stream);
    if (hipStreamSynchronize(e->stream) != hipSuccess) rc = set_error(PM_ENODEV, "row network test");
  }
  if (d_k) (void)hipFree(d_k);
  if (d_rows) (void)hipFree(d_rows);
  if (d_s) (void)hipFree(d_s);
  if (d_m) (void)hipFree(d_m);
  return rc;
}

int32_t pm_last_stats(pm_engine* e, pm_stats* stats) {
  if (!e || !stats) return set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  *stats = e->last_stats;
  return PM_OK;
}

}  // extern "C"
