/*
 * pm_engine.h — C ABI of the MI355X job-to-worker matching engine (libpm_engine.so).
 *
 * This is the drop-in boundary for the allocation hot path of the PrimeIntellect-ai/protocol
 * orchestrator.  The reference has no FFI today; each entry point below names the Rust item whose
 * work it takes over (paths relative to /root/reference/crates) and INTEGRATION.md shows the
 * `GpuMatchPlugin` variant + `extern "C"` block a maintainer would add next to
 * orchestrator/src/plugins/mod.rs:60-79.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every call returns PM_OK (0) or a negative PM_E* code and never throws/aborts across the ABI;
 *     pm_last_error() returns a thread-local message for the last failing call on this thread;
 *   - inputs are borrowed SoA column views, valid only for the duration of the call (the engine
 *     copies them to HBM); outputs are caller-allocated;
 *   - rows are identified by index: worker index = position in the caller's `get_nodes()` order
 *     (orchestrator/src/store/domains/node_store.rs:163-209 — the order is significant, it is the
 *     reference's tie-break), task index = position in `get_all_tasks()` order
 *     (store/domains/task_store.rs:57-82), config index = position in the configuration list given
 *     to pm_set_configs;
 *   - the engine needs a gfx950 device and fails with PM_ENODEV without one.  There is no CPU
 *     fallback.
 *
 * Threading (orchestrator/src/plugins/mod.rs:66-78 is called concurrently from actix workers):
 *   pm_lookup_task_for_worker may be called from any thread at any time, also while pm_tick runs: the published
 *   table is double-buffered behind a sequence counter (seqlock) — no lock, no HIP call, no allocation; a reader
 *   retries only if two publishes complete during its 32-byte row copy.  Every other call takes the engine's
 *   single-writer mutex.
 */
#ifndef PM_ENGINE_H
#define PM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_ABI_VERSION 3 /* 3 (round 6): + the by-id / by-worker group calls of the plugin's read surface; the stepwise tick's
                            carve_next / carve_validate pair became pm_dist_carve_wait; pm_upload_workers(keep_groups = 1)
                            accepts new rows behind the known ones; + pm_host_to_lowercase (the model rule is Unicode now); carve_variant 2 / 4
                            are PM_EINVAL (since round 5); pm_stats of a pm_tick / pm_tick_many: ms_publish is 0 (the claim publishes),
                            ms_total is the host's clock over the call, ms_sweep_kernel is measured by a time_proposer engine only */

enum {
  PM_OK = 0,
  PM_EINVAL = -1,   /* bad argument / invalid configuration (the reference constructor panics,
                       node_groups/mod.rs:142-147) */
  PM_ENODEV = -2,   /* no usable gfx950 device / HIP runtime error */
  PM_ENOMEM = -3,
  PM_ESTATE = -4,   /* call order violated (e.g. tick before tables were uploaded) */
  PM_ERANGE = -5,   /* index out of range / caller buffer too small */
  PM_EPARSE = -6,   /* ComputeRequirements::from_str returned Err */
  PM_EPANIC = -7    /* the reference would panic on this input (e.g. shared/models/node.rs:251) */
};

#define PM_NONE 0xFFFFFFFFu /* "no task" / "no worker" / "no group" */
#define PM_MAX_CONFIGS 64   /* one bit per NodeGroupConfiguration in the u64 masks */

typedef struct pm_engine pm_engine;

/* ------------------------------------------------------------------ worker table
 * Projection of OrchestratorNode (orchestrator/src/models/node.rs:11-37) + ComputeSpecs
 * (shared/src/models/node.rs:25-35,72-78,153-157) + NodeLocation (:543-550).  Option<T> fields
 * become a flag bit + a value column. */
enum {
  PM_W_HAS_SPECS = 1u << 0,  /* compute_specs.is_some() */
  PM_W_HAS_GPU = 1u << 1,    /* compute_specs.gpu.is_some() */
  PM_W_GPU_COUNT = 1u << 2,  /* gpu.count.is_some() */
  PM_W_GPU_MEM = 1u << 3,    /* gpu.memory_mb.is_some() */
  PM_W_GPU_MODEL = 1u << 4,  /* gpu.model.is_some() -> gpu_model_class valid */
  PM_W_HAS_CPU = 1u << 5,    /* compute_specs.cpu.is_some() */
  PM_W_CPU_CORES = 1u << 6,  /* cpu.cores.is_some() */
  PM_W_RAM = 1u << 7,        /* ram_mb.is_some() */
  PM_W_STORAGE = 1u << 8,    /* storage_gb.is_some() */
  PM_W_HEALTHY = 1u << 9,    /* status == NodeStatus::Healthy (models/node.rs:75-85) */
  PM_W_HAS_P2P = 1u << 10,   /* p2p_id.is_some() */
  PM_W_HAS_LOC = 1u << 11    /* location.is_some() */
};

typedef struct {
  uint32_t n;
  const uint32_t* flags;           /* PM_W_* */
  const uint32_t* gpu_count;
  const uint32_t* gpu_mem_mb;
  const uint32_t* gpu_model_class; /* index of the interned gpu.model string (pm_set_model_table) */
  const uint32_t* cpu_cores;
  const uint32_t* ram_mb;
  const uint32_t* storage_gb;
  const uint32_t* price;           /* extension column; NULL or all-zero in every parity run */
  const uint32_t* addr_rank;       /* rank of address.to_string() in byte order: GROUP_INDEX is the
                                      rank inside the group's BTreeSet<String> (node_groups/mod.rs:
                                      424-434). NULL => worker index order */
  const double* lat;               /* NodeLocation.latitude  (read only when PM_W_HAS_LOC) */
  const double* lon;               /* NodeLocation.longitude */
} pm_worker_soa;

/* ------------------------------------------------------------------ configuration table
 * NodeGroupConfiguration (node_groups/mod.rs:30-37) + ComputeRequirements / GpuRequirements
 * (shared/src/models/node.rs:49-70). */
enum {
  PM_R_HAS_REQ = 1u << 0,   /* compute_requirements.is_some() */
  PM_R_CPU = 1u << 1,       /* requirements.cpu.is_some() */
  PM_R_CPU_CORES = 1u << 2, /* requirements.cpu.cores.is_some() */
  PM_R_RAM = 1u << 3,
  PM_R_STORAGE = 1u << 4
};
typedef struct {
  uint32_t flags; /* PM_R_* */
  uint32_t cpu_cores, ram_mb, storage_gb;
  uint32_t alt_begin, alt_count; /* GPU alternatives (OR) in the pm_gpu_alt_row table */
  uint32_t min_group_size, max_group_size;
} pm_config_row; /* 32 B */

enum {
  PM_G_COUNT = 1u << 0,
  PM_G_MODEL = 1u << 1,
  PM_G_MEM = 1u << 2,
  PM_G_MEM_MIN = 1u << 3,
  PM_G_MEM_MAX = 1u << 4,
  PM_G_TOT_MIN = 1u << 5,
  PM_G_TOT_MAX = 1u << 6
};
typedef struct {
  uint32_t flags; /* PM_G_* */
  uint32_t count, memory_mb, memory_mb_min, memory_mb_max, total_memory_min, total_memory_max;
  uint32_t model_row; /* row of the model bit table (valid when PM_G_MODEL) */
} pm_gpu_alt_row; /* 32 B */

/* ------------------------------------------------------------------ task table
 * Projection of Task (shared/src/models/task.rs:162-184) to what the path reads. */
typedef struct {
  uint32_t n;
  const uint64_t* topo_mask; /* bit c set iff allowed_topologies contains config c's name; ~0ull when
                                any Option on the way is None (scheduler_impl.rs:44-59) */
  const int64_t* created_at;
  const uint64_t* uid;       /* stable task identity across uploads (e.g. low 64 bits of the UUID);
                                NULL => identity = index */
} pm_task_soa;

/* ------------------------------------------------------------------ engine */
enum { PM_CHOOSE_FIRST = 0, PM_CHOOSE_SEEDED = 1 };

typedef struct {
  uint32_t abi_version;          /* PM_ABI_VERSION */
  int32_t device;                /* HIP device ordinal (LOCAL_RANK) */
  uint32_t proximity_enabled;    /* ProximityOptimizationPolicy.enabled (mod.rs:85-88, default 1) */
  uint32_t switching_enabled;    /* TaskSwitchingPolicy.enabled        (mod.rs:90-97, default 1) */
  uint32_t prefer_larger_groups; /* TaskSwitchingPolicy.prefer_larger_groups (default 1) */
  uint32_t chooser;              /* replaces rand::rng().choose (scheduler_impl.rs:66-70, mod.rs:1175) */
  uint64_t chooser_seed;
  uint64_t group_id_seed;        /* replaces generate_group_id (mod.rs:1489-1493): ids are the
                                    splitmix64 stream of this seed */
  uint32_t debug_uncertain_every; /* test hook: treat every n-th carve step as a near-tie so the
                                     exact host resolve path runs (0 = off) */
  uint32_t sweep_variant;        /* pair-sweep kernel: 0 = default (best), 1 = scalar reference kernel */
  uint32_t carve_variant;        /* group formation: 0 = default: the streaming carve — ONE launch per pass; workgroup 0
                                        validates (the in-order chain), every other workgroup computes neighbour rows for
                                        the seeds a bounded look-ahead in front of the chain; slot == position, nothing
                                        is compacted.  (Swarms with more than 262,144 unassigned rows take the batch
                                        pipeline, 3.)
                                    1 = single-workgroup sequential exact sweep only (the straightforward kernel),
                                    3 = the batch pipeline: per batch, full-chip preparation of a compact candidate
                                        list, full-chip neighbour-list proposals, validation by the in-order chain
                                        (the default of rounds 1-3; what a streaming launch that gives up falls back to).
                                    Anything else is PM_EINVAL (2: single-wave validation, 4: two batches in flight —
                                    both measured slower in rounds 2 / 3 and removed in round 5; their numbers are in
                                    profiles/r03_*) */
  uint32_t time_proposer;        /* bench: hipEvents around every proposer launch (pm_stats.ms_propose_kernel) and around the
                                    pair sweep's kernel (pm_stats.ms_sweep_kernel) */
} pm_engine_config;

void pm_engine_config_default(pm_engine_config*);

/* NodeGroupsPlugin::new / Scheduler::new (node_groups/mod.rs:113-175, scheduler/mod.rs:14-24) */
int32_t pm_engine_create(const pm_engine_config* cfg, pm_engine** out);
void pm_engine_destroy(pm_engine*);
const char* pm_last_error(void);

/* Configuration templates in caller order; the engine applies the constructor sort
 * (mod.rs:150-164) and validity checks (:138-147 -> PM_EINVAL) itself.  n_cfgs <= PM_MAX_CONFIGS;
 * min_group_size == 0 is rejected (the reference would create empty groups forever). */
int32_t pm_set_configs(pm_engine*, const pm_config_row* cfgs, uint32_t n_cfgs,
                       const pm_gpu_alt_row* alts, uint32_t n_alts);
/* Host-evaluated model rule (shared/src/models/node.rs:463-484): bits[row * words + (cls >> 5)]
 * bit (cls & 31) = requirement row `row` matches interned spec model string `cls`;
 * words = (n_classes + 31) / 32.  pm_host_build_model_table() produces it. */
int32_t pm_set_model_table(pm_engine*, const uint32_t* bits, uint32_t n_rows, uint32_t n_classes);
/* "available_node_group_configs" (mod.rs:399-418, :1328-1348): bit c = config c enabled. */
int32_t pm_set_enabled_mask(pm_engine*, uint64_t enabled);

/* NodeStore::get_nodes snapshot (order significant). Resets group state iff keep_groups == 0.
 * keep_groups == 1 keeps the groups and their claimed tasks (groups are keyed by row index): the rows the engine already
 * has must come first, in the same order; rows behind them are new ones, exactly as pm_append_workers would have added
 * them (in no group).  It is the re-synchronisation call of a host whose delta calls failed half-way: every row again,
 * nothing dissolved, the id stream not restarted — followed by pm_on_worker_status_many for the rows that died meanwhile. */
int32_t pm_upload_workers(pm_engine*, const pm_worker_soa* workers, uint32_t keep_groups);
/* A node the discovery monitor sees for the first time (orchestrator/src/discovery/monitor.rs:236-420 ->
 * NodeStore::add_node): rows are appended behind the existing ones (*first_index = index of the first new row),
 * existing groups and claims are untouched, and the next pm_tick carves the newcomers together with the
 * leftovers — the reference is incremental in exactly this way (node_groups/mod.rs:487-497, tests.rs:993-1213).
 * The host keeps a stable address -> row index map; a node that leaves is tombstoned (pm_on_worker_status with
 * PM_W_HEALTHY cleared), never removed, so indices stay valid.  Only the new rows travel to the GPU. */
int32_t pm_append_workers(pm_engine*, const pm_worker_soa* rows, uint32_t* first_index);
/* Churn: overwrite rows idx[0..rows->n) (discovery sync: specs / location / status of known nodes changed).
 * Only these rows travel (one packed copy + a scatter kernel); groups are untouched.  A row that loses
 * PM_W_HEALTHY is NOT dissolved here; call pm_on_worker_status for Dead/LowBalance. */
int32_t pm_update_workers(pm_engine*, const uint32_t* idx, const pm_worker_soa* rows);
/* GROUP_INDEX is the rank of address.to_string() inside the group (mod.rs:424-434).  Appending nodes shifts the
 * global ranks of the others: the host re-ranks its address strings and replaces the whole column (4 B/worker). */
int32_t pm_set_addr_ranks(pm_engine*, const uint32_t* addr_rank, uint32_t n_workers);
/* TaskStore::get_all_tasks snapshot. Groups whose claimed task uid disappeared are dissolved
 * (on_task_deleted, mod.rs:1245-1288). */
int32_t pm_upload_tasks(pm_engine*, const pm_task_soa* tasks);
/* Task deltas — the observer-invalidated task cache of SURVEY section 8f row 2, replacing the per-heartbeat
 * get_all_tasks (store/domains/task_store.rs:57-82) AND the re-upload of the snapshot:
 *   pm_tasks_insert_front  on_task_created (mod.rs:1224-1243): rows in get_all_tasks order (created_at
 *                          descending), every one of them strictly newer than the newest task of the table, so they
 *                          sort in front of it (task_store.rs:79); PM_EINVAL otherwise -> fall back to
 *                          pm_upload_tasks.  Only the new rows travel, the bit planes are patched in the words
 *                          they fall into, and every claimed task keeps its binding.
 *   pm_tasks_delete        on_task_deleted (mod.rs:1245-1325) by task id (needs pm_task_soa.uid): the rows leave
 *                          the list, groups that had claimed one of them are dissolved; unknown ids are ignored.
 * Task indices reported by the engine (pm_assignment.task, pm_group.task, pm_match*, pm_newest_task) are always
 * positions in the caller's CURRENT list: after an insertion of n rows every older task's index is n higher,
 * after a deletion the later ones move up — exactly as in the caller's own Vec<Task>.  That includes the table
 * pm_lookup_task_for_worker reads: both calls (and pm_upload_tasks) re-derive the published positions on the host
 * before they return, and clear the rows of the groups a deleted task took with it — no tick needed in between. */
int32_t pm_tasks_insert_front(pm_engine*, const pm_task_soa* rows);
/* pm_tasks_insert_front and, with republish != 0, pm_match's pair sweep + claim + publish on the standing groups in the
 * same call (no carve): a group that holds no task is offered the new one NOW — the reference offers it at that
 * group's next heartbeat (scheduler_impl.rs:33-74), the engine otherwise at the next tick.  ~0.2 ms at 100k tasks x
 * 10k workers.  republish == 0 is pm_tasks_insert_front. */
int32_t pm_tasks_insert_front_ex(pm_engine*, const pm_task_soa* rows, uint32_t republish);
int32_t pm_tasks_delete(pm_engine*, const uint64_t* uids, uint32_t n, uint32_t* n_deleted);

/* StatusUpdatePlugin::handle_status_change (status_update_impl.rs:8-39): dead != 0 means the new
 * status is Dead or LowBalance => dissolve the worker's whole group. */
int32_t pm_on_worker_status(pm_engine*, uint32_t worker, uint32_t flags_new, uint32_t dead);
/* The same for n workers in one call, applied in array order (the status updater's sweep marks many nodes in one
 * pass, status_update/mod.rs; a tombstoning node sync does too).  dead may be NULL (= none of them).  An index out
 * of range fails the call before anything is applied. */
int32_t pm_on_worker_status_many(pm_engine*, const uint32_t* workers, const uint32_t* flags_new, const uint32_t* dead,
                                 uint32_t n);
/* Group life-cycle feed for the webhook / metrics emission that follows every creation and dissolution in the
 * reference (send_group_created after try_form_new_groups, mod.rs:612-625; send_group_destroyed per dissolved solo
 * group then send_group_created for the merged one, mod.rs:974-1000; send_group_destroyed in dissolve_group,
 * mod.rs:1469-1481).  Off by default; once enabled the engine logs one event per creation / dissolution, in the
 * order the reference would emit them, until the caller drains the log:
 *   pm_form_groups / pm_tick      created, in formation order
 *   pm_merge_solo_groups / tick   per merge: destroyed for each solo group of the batch (batch order), then created
 *   pm_on_worker_status(dead), pm_dissolve_group, pm_tasks_delete, pm_upload_tasks (claimed task gone)   destroyed
 *     (several groups hit by one pm_tasks_delete: the reference walks a Redis SCAN there — order unpinned; the engine
 *     goes through its group list in creation order)
 * pm_reset_groups and a pm_upload_workers that drops the groups log nothing (no counterpart in the reference).
 * Members are worker indices in BTreeSet<String> order (address rank), as in pm_get_groups. */
enum { PM_GROUP_CREATED = 1, PM_GROUP_DESTROYED = 2 };
typedef struct pm_group_event {
  uint64_t group_id;     /* generate_group_id value */
  uint32_t kind;         /* PM_GROUP_CREATED / PM_GROUP_DESTROYED */
  uint32_t config;       /* index of the configuration (row of pm_set_configs) */
  uint32_t member_begin; /* into the members array handed to pm_drain_group_events */
  uint32_t n_members;
} pm_group_event;
int32_t pm_enable_group_events(pm_engine*, uint32_t on); /* switching off also clears the log */
/* Copies the logged events (oldest first) and their members out and clears the log.  n_events / n_members always
 * report what the log holds; with a buffer that is too small (or NULL) the call returns PM_ERANGE and drains
 * nothing, so `pm_drain_group_events(e, NULL, 0, NULL, 0, &ne, &nm)` is the size query. */
int32_t pm_drain_group_events(pm_engine*, pm_group_event* events, uint32_t cap_events, uint32_t* members,
                              uint32_t cap_members, uint32_t* n_events, uint32_t* n_members);
/* dissolve_group (mod.rs:1423-1487) by group slot. */
int32_t pm_dissolve_group(pm_engine*, uint32_t group_slot);
/* dissolve_group(&group_id) as the API routes call it (mod.rs:1002-1004 -> :1423-1487; DELETE /groups/{id} and
 * force-regroup, api/routes/groups.rs:126, :349): by the group's id (the generate_group_id value whose "{:x}" text the
 * route carries).  An id that names no group is not an error — the reference logs "No group found" and returns Ok(()):
 * PM_OK with *dissolved = 0.  The group's rows in the published table read "no group" before the call returns; the
 * life-cycle feed gets its PM_GROUP_DESTROYED.  dissolved may be NULL. */
int32_t pm_dissolve_group_by_id(pm_engine*, uint64_t group_id, uint32_t* dissolved);
/* Drop all groups (bench: cold start of a full-swarm match). */
int32_t pm_reset_groups(pm_engine*);

/* Phase A — is_node_compatible_with_config x ComputeSpecs::meets over W x C
 * (mod.rs:206-215, shared/src/models/node.rs:377-541).  mask_out (W entries) may be NULL. */
int32_t pm_compat_masks(pm_engine*, uint64_t* mask_out);

/* try_form_new_groups (mod.rs:478-628) / try_merge_solo_groups (:631-971). */
int32_t pm_form_groups(pm_engine*, uint32_t* n_formed);
int32_t pm_merge_solo_groups(pm_engine*, uint32_t* n_merged);

typedef struct {
  uint64_t id;          /* generate_group_id stream */
  uint32_t config;      /* configuration index */
  uint32_t n_members;
  uint32_t member_begin; /* offset into the members array returned alongside */
  uint32_t task;        /* claimed task index (group_task:<id>) or PM_NONE */
} pm_group;

/* get_all_groups (mod.rs:1006-1044) in slot (creation) order.  group_of_worker (W entries, slot in
 * the returned array or -1), groups (cap_groups) and members (cap_members, BTreeSet order) may be
 * NULL to query counts only.  (The reference sorts the list by id text, mod.rs:1040: the caller's step — the ids are
 * numbers here and "{:x}" strings there.) */
int32_t pm_get_groups(pm_engine*, int32_t* group_of_worker, pm_group* groups, uint32_t cap_groups,
                      uint32_t* n_groups, uint32_t* members, uint32_t cap_members, uint32_t* n_members);
/* ONE group, for the read surface the API routes use between ticks:
 *   pm_get_group_by_id      get_group_by_id (mod.rs:1046-1055; GET /groups/{id}/logs, api/routes/groups.rs:160)
 *   pm_get_group_of_worker  get_node_group (mod.rs:324-337; the storage route's upload name, api/routes/storage.rs:150,
 *                           and per node get_node_groups_batch, mod.rs:339-397, api/routes/nodes.rs:74) — worker = the
 *                           row index of the node's address
 * *slot = the group's slot (as pm_get_groups numbers them NOW) or PM_NONE: no such group / the worker is in none
 * (Ok(None) in the reference) — PM_OK either way.  out->member_begin is 0; members (cap_members entries, BTreeSet order:
 * the position of a worker in it is get_idx_in_group, mod.rs:424-434) may be NULL; with a members buffer that is too small
 * the call returns PM_ERANGE after filling *out (out->n_members says what is needed).  Host-side state only: no HIP call,
 * but the engine's mutex (a tick in flight finishes first). */
int32_t pm_get_group_by_id(pm_engine*, uint64_t group_id, pm_group* out, uint32_t* members, uint32_t cap_members,
                           uint32_t* slot);
int32_t pm_get_group_of_worker(pm_engine*, uint32_t worker, pm_group* out, uint32_t* members, uint32_t cap_members,
                               uint32_t* slot);

/* Phase B, reference orientation — NodeGroupsPlugin::filter_tasks (scheduler_impl.rs:11-110) for
 * EVERY worker at once: the T x W topology sweep, the chooser and the per-group claim (SETNX :74).
 * task_of_worker[w] = task index or PM_NONE; applicable_count[w] = number of applicable tasks the
 * sweep found for w's group (0 for workers outside groups).  Either output may be NULL. */
int32_t pm_match(pm_engine*, uint32_t* task_of_worker, uint32_t* applicable_count);

/* Phase B, north_star orientation — for every task the best bid among eligible compatible
 * workers: eligible = Healthy & p2p & unassigned (mod.rs:492-497), compatible = some config in
 * topo_mask[t] & enabled with compat bit set; best = min (price, worker index).
 * best_worker[t] = index or PM_NONE; candidate_count[t] = number of such workers. */
int32_t pm_match_per_task(pm_engine*, uint32_t* best_worker, uint32_t* candidate_count);

/* NewestTaskPlugin::filter_tasks (newest_task/mod.rs:8-19): argmax created_at, last max wins. */
int32_t pm_newest_task(pm_engine*, uint32_t* task_idx);

typedef struct {
  /* GPU time per phase of the last pm_tick, from hipEvents on the engine's stream (ms).  ms_publish is 0 where the claim
     writes the snapshot buffer itself (pm_tick, pm_tick_many: the sweep phase ends with it); ms_total is the tick as its
     caller sees it: there the host's clock from the tick's begin to its end, elsewhere first event to last */
  float ms_compat, ms_carve, ms_merge, ms_sweep, ms_publish, ms_total;
  /* kernel-only durations (hipEvents recorded immediately around the launches, summed over relaunches); ms_sweep_kernel is
     measured — two more events, each a barrier packet on the stream — only by an engine created with time_proposer, else 0 */
  float ms_compat_kernel, ms_carve_kernel, ms_sweep_kernel;
  uint32_t n_groups, n_formed, n_merged;
  uint32_t carve_steps;         /* groups carved + merge selections done on the GPU */
  uint32_t carve_fast_steps;    /* of which committed without an exact sweep by the whole workgroup: straight from a
                                   neighbour-list proposal, first-come, or (streaming carve) in registers at the end of
                                   a configuration's located phase */
  uint32_t host_resolved_steps; /* carve steps whose near-tie was settled by the exact host path */
  uint32_t carve_launches;
  uint64_t pair_evals;          /* T x W of the sweep */
  uint64_t carve_cand_sum;      /* sum over carve steps of the remaining candidates scanned (roofline bytes) */
  /* proposer (carve_propose_kernel): summed launch durations (hipEvents around every launch; 0 when the launches
   * are not timed individually), neighbour lists computed by this rank, Haversine keys their sweeps evaluated */
  float ms_propose_kernel;
  uint32_t proposals;
  uint64_t propose_keys;
} pm_stats;

/* One full-swarm match on one GPU (a multi-GPU engine uses the stepwise tick below): compat masks -> form groups
 * -> merge solo groups -> pair sweep + claim ->
 * publish the assignment table (run_group_management_loop body, mod.rs:180-203, plus one
 * get_task_for_node per worker, scheduler/mod.rs:26-36). */
int32_t pm_tick(pm_engine*, pm_stats* stats);
/* Stats of the last pm_tick / pm_form_groups / pm_merge_solo_groups call. */
int32_t pm_last_stats(pm_engine*, pm_stats* stats);

typedef struct {
  uint32_t task;        /* task index or PM_NONE */
  uint32_t group_slot;  /* PM_NONE when the worker is in no group */
  uint32_t group_index; /* GROUP_INDEX (mod.rs:424-434) */
  uint32_t group_size;  /* GROUP_SIZE */
  uint32_t next_worker; /* worker whose p2p id is NEXT_P2P_ADDRESS (scheduler_impl.rs:115-128) */
  uint64_t group_id;    /* GROUP_ID */
} pm_assignment;

/* Scheduler::get_task_for_node (scheduler/mod.rs:26-36) served from the table published by the last
 * pm_tick / pm_match: lock-free (seqlock over two buffers), no HIP call. */
int32_t pm_lookup_task_for_worker(pm_engine*, uint32_t worker, pm_assignment* out);

/* Device-resident view of the last published per-worker task column (u32 task index or PM_NONE, W
 * entries) for device-side consumers — e.g. the cross-shard RCCL all-gather of table shards.  The
 * pointer stays valid until the next worker upload; contents are rewritten by pm_tick / pm_match. */
int32_t pm_device_task_column(pm_engine*, uint64_t* device_ptr, uint32_t* n);

/* ------------------------------------------------------------------ multi-GPU
 * One engine per GPU (one process per GPU), every engine holds the WHOLE swarm — 100k worker rows are 6 MB — and
 * worker w is owned by rank shard_of_worker[w] (the caller's hash of the address, e.g. splitmix64(address) %
 * world).  The reference carves from ONE pool (node_groups/mod.rs:492-503), so the carve domain is not split:
 * what is split is the parallel work, and every rank ends a tick with bit-identical groups and tables.
 *   - the carve — try_form_new_groups' chain of dependent steps — runs REPLICATED: every rank runs the whole streaming
 *     launch on identical inputs and ends on the identical groups and ids; nothing is exchanged for it (rounds 2-4 dealt a
 *     batch's neighbour-list proposals over the ranks and all-gathered them ~55 times a tick: slower on 8 GPUs than on one);
 *   - the pair sweep + chooser + claim run for the OWNED workers only; the published rows are all-gathered once
 *     per tick and scattered into every rank's full table (the "cross-shard conflict-resolution all-gather");
 *   - pm_match_per_task bids with the owned workers only; the caller folds the per-task bests (min index / sum).
 * The collectives are the caller's (RCCL ncclAllGather, or torch.distributed): the engine hands out device
 * pointers.  recv = [world][bytes_per_rank]; send = this rank's contribution (bytes_per_rank bytes); issue
 * all-gather(send -> recv) on the engine's stream (pm_set_stream) and call the next step.  bytes_per_rank == 0:
 * nothing to exchange for this step.  Status / task / worker updates are replicated calls: every rank receives
 * every event.  A stepwise tick with world == 1 is the plain tick in steps (no exchange). */
typedef struct {
  uint64_t send_ptr, recv_ptr; /* device pointers */
  uint64_t bytes_per_rank;
} pm_dist_xfer;

/* Several engines (pools) matching on ONE GPU at the same time: the carve's launch keeps a workgroup resident on every
 * CU it uses for as long as it runs — the validator and the workgroups that make its neighbour rows — so K launches run
 * side by side only if each takes its share: n = (CUs of the device - K) / K row-making workgroups per engine.  A
 * 10k-worker swarm loses a few percent with 48 instead of ~200 of them; without the call (n == 0: sized by the swarm,
 * up to every CU) a second engine's launch queues behind the first one's.  tests/test_gpu_parity.py
 * (test_pools_share_one_gpu), bench.py `pools_on_one_gpu`.
 *   The process must also give the HIP runtime enough hardware queues: it multiplexes a process's streams onto
 * GPU_MAX_HW_QUEUES of them (environment, read once when the runtime starts; 4 by default) and two streams on one queue
 * run in turn.  An engine owns one stream (two until round 5), so with the default the fifth engine's carve already waits
 * behind another engine's (round 4, two streams an engine: K = 4 pools 1.4x the one-pool rate with 4 queues, 3.3x with 8
 * or more — profiles/r04_pools_hw_queues.json).  Set GPU_MAX_HW_QUEUES >= K (16 is a good value up to K = 8; 32
 * oversubscribes the device's queue slots) before the first HIP call of the process, from the process entry point —
 * INTEGRATION.md "Several pools on one GPU". */
int32_t pm_set_carve_workgroups(pm_engine*, uint32_t n);
/* pm_tick for n engines (pools) in ONE call from ONE host thread: every engine's carve is started on its own stream
 * before the first is waited for, so the launches are resident side by side (pm_set_carve_workgroups first) and the
 * host is never inside two HIP calls at once; per engine the device work is pm_tick's, in pm_tick's order, and what
 * it publishes is what pm_tick would have published.  stats: n entries or NULL; stats[i].ms_total = device time from
 * the start of engine i's tick to its published table, i.e. the latency pool i sees inside the batch.  An engine that
 * fails leaves the batch (in the state a failed pm_tick leaves it in), the others finish; the call returns the first
 * failure.  The reference runs one pool per orchestrator process (run_group_management_loop, mod.rs:180-203): this is
 * the entry point for a process that serves several.
 * THIS is the supported way to match several pools of one process (K = 4: 3.0x the one-pool rate, every pool's match
 * within 1.1x of the median; K = 8: 3.8x — profiles/r05*_bench.json `pools_on_one_gpu.tick_many`); K application threads
 * that each call pm_tick work too, but what they reach depends on the caller's threads (a Python harness: 1.8x at K = 4
 * with single matches of 7 ms; threads inside the library, below: 3.1x).
 *   flags: 0, or PM_TICK_MANY_THREADS = one host thread per engine inside the library, each calling pm_tick (K = 4: 3.1x,
 *   K = 8: 4.1x, K = 16: 2.5x where the one-thread walk keeps 3.1x). */
#define PM_TICK_MANY_THREADS 1u
int32_t pm_tick_many(pm_engine* const* engines, uint32_t n, pm_stats* stats, uint32_t flags);
/* All work of this engine goes to the caller's HIP stream (hipStream_t), e.g. the stream its RCCL calls use, so
 * kernels and collectives are ordered without host synchronisation.  NULL = back to the engine's own stream. */
int32_t pm_set_stream(pm_engine*, void* hip_stream);
/* After pm_upload_workers (and again whenever the row count changes).  world == 1 switches back. */
int32_t pm_dist_configure(pm_engine*, uint32_t rank, uint32_t world, const uint8_t* shard_of_worker);
/* The tick in steps:
 *   pm_dist_tick_begin         compat sweep; the carve is started — the WHOLE carve, on every rank (the reference carves
 *                              from one pool, node_groups/mod.rs:492-503, and the chain of dependent steps that forms the
 *                              groups does not shard: it is replicated, one streaming launch per rank, and ends on the
 *                              identical groups and ids everywhere)
 *   pm_dist_carve_wait         waits for the carve and settles near-ties on the host (replicated, deterministic)
 *   pm_dist_match_begin(&x)    solo merge, pair sweep + claim of the OWNED workers -> all-gather x (the ONE exchange of a
 *                              tick) ->
 *   pm_dist_tick_end(&stats)   scatter into the full table, publish (pm_lookup_* serve every worker)
 * (ABI 2 had a carve_next(&x, &more) + carve_validate pair in the place of pm_dist_carve_wait: a loop that
 * exchanged nothing since round 5.  Removed with the version bump, so that a driver written as that loop fails at the
 * version check and not at run time.) */
int32_t pm_dist_tick_begin(pm_engine*);
int32_t pm_dist_carve_wait(pm_engine*);
int32_t pm_dist_match_begin(pm_engine*, pm_dist_xfer* x);
int32_t pm_dist_tick_end(pm_engine*, pm_stats* stats);
/* pm_match_per_task with the results left on the device (u32[T] each, worker indices are global) for a
 * device-side fold across ranks; not available with a non-zero price column. */
int32_t pm_match_per_task_device(pm_engine*, uint64_t* best_ptr, uint64_t* count_ptr, uint32_t* n_tasks);

/* ------------------------------------------------------------------ host helpers (no GPU needed)
 * ComputeRequirements::from_str (shared/src/models/node.rs:180-374).  Fills cfg->flags/cpu/ram/
 * storage/alt_count (alt_begin, min/max sizes untouched) and up to alt_cap alternatives; model
 * strings are returned as offsets: alts[i].model_row = byte offset into models_out of a
 * NUL-terminated copy.  Returns PM_OK, PM_EPARSE, PM_EPANIC or PM_ERANGE. */
int32_t pm_host_parse_requirements(const char* s, pm_config_row* cfg, pm_gpu_alt_row* alts,
                                   uint32_t alt_cap, char* models_out, size_t models_cap);
/* GpuSpecs::meets model rule (shared/src/models/node.rs:463-484) for one string pair: both sides through
 * `str::to_lowercase` — the full Unicode mapping (pm_host_to_lowercase), not ASCII — then ' ' -> '_', the requirement split
 * at ',' and every part trimmed of Unicode white space, four `contains` with and without '_'. */
int32_t pm_host_model_matches(const char* spec_model, const char* req_model);
/* str::to_lowercase (UTF-8 in, UTF-8 out): every code point through char::to_lowercase (up to three code points for one:
 * U+0130), a capital sigma that ends a word as the final form.  Tables: Unicode 13.0 (tools/make_unicode_tables.py).
 * *needed = strlen(result) + 1; out may be NULL with cap 0 to size the buffer; PM_ERANGE when cap is too small. */
int32_t pm_host_to_lowercase(const char* in, char* out, size_t cap, size_t* needed);
/* Bit table for pm_set_model_table: req_models[n_rows] x spec_models[n_classes]. */
int32_t pm_host_build_model_table(const char* const* req_models, uint32_t n_rows,
                                  const char* const* spec_models, uint32_t n_classes, uint32_t* bits_out);
/* Constructor sort (mod.rs:150-164) + runtime filter/sort (:399-418): writes the carve order of the
 * enabled configurations, returns their count in *n_out. */
int32_t pm_host_config_order(const pm_config_row* cfgs, uint32_t n_cfgs, uint64_t enabled,
                             uint32_t* order_out, uint32_t* n_out);

/* ---- the step right after the match: group variables in the task that is handed to the worker
 * (SURVEY section 8f row 3).  The engine returns numbers (pm_assignment); these helpers do the string work
 * the reference does with chained `str::replace` calls, in the same order — every pass scans the result of
 * the previous one.  Output convention: *needed = strlen(result) + 1; PM_ERANGE if cap is too small (nothing
 * is written then); out may be NULL with cap 0 to size the buffer.
 *
 * pm_host_group_vars — scheduler_impl.rs:160-183, applied by the reference to every env-var value and every
 *   cmd argument: ${GROUP_INDEX}, ${GROUP_SIZE}, ${NEXT_P2P_ADDRESS}, ${GROUP_ID}, ${TOTAL_UPLOAD_COUNT},
 *   ${LAST_FILE_IDX} (= pm_host_last_file_idx(total_upload_count)).
 * pm_host_volume_vars — scheduler_impl.rs:185-200: ${GROUP_ID} in host_path / container_path.
 * pm_host_upload_name_vars — orchestrator/src/api/routes/storage.rs:150-215: ${NODE_GROUP_ID},
 *   ${NODE_GROUP_SIZE}, ${NODE_GROUP_INDEX} (only when the node is in a group: group_id != NULL), then
 *   ${TOTAL_UPLOAD_COUNT_AFTER} and ${CURRENT_FILE_INDEX} (= upload_count saturating-minus 1).
 * pm_host_last_file_idx — `total_upload_count.parse::<u32>().unwrap_or(0).saturating_sub(1)`
 *   (scheduler_impl.rs:155-158): Rust u32 syntax (optional '+', ASCII digits, no blanks, no overflow). */
typedef struct pm_group_vars {
  uint32_t group_index;            /* pm_assignment.group_index */
  uint32_t group_size;             /* pm_assignment.group_size */
  const char* next_p2p_address;    /* p2p id of pm_assignment.next_worker ("" if unknown) */
  const char* group_id;            /* the group's id string */
  const char* total_upload_count;  /* decimal count of `upload:<node>:<group>:*` keys, as the reference keeps it */
} pm_group_vars;
int32_t pm_host_group_vars(const char* in, const pm_group_vars* v, char* out, size_t cap, size_t* needed);
int32_t pm_host_volume_vars(const char* in, const char* group_id, char* out, size_t cap, size_t* needed);
int32_t pm_host_upload_name_vars(const char* in, const char* group_id, uint32_t group_size, uint32_t group_index,
                                 uint64_t upload_count, char* out, size_t cap, size_t* needed);
uint32_t pm_host_last_file_idx(const char* total_upload_count);

uint32_t pm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
