/*
 * pm_oracle.h — CPU restatement ("oracle") of the PrimeIntellect-ai/protocol
 * orchestrator allocation path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libpm_engine.so) never links, loads or
 * calls anything in oracle/.
 *
 * Every function cites the reference region it restates (paths relative to
 * /root/reference/crates).  The oracle works on string-bearing AoS rows like
 * the reference does (model strings, config names, address strings), NOT on
 * the engine's SoA projection, so the projection itself is under test.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   PINNED   by the reference's own known-answer tests, transcribed in
 *            tests/golden/node_rs_kats.json:
 *              orc_parse_requirements   shared/src/models/node.rs:659-736,1044-1063,1082-1116,1229-1241
 *              orc_meets                shared/src/models/node.rs:740-1042,1065-1080,1118-1227
 *              orc_newest_task          orchestrator/src/plugins/newest_task/mod.rs:29-55
 *            in tests/golden/group_vars_kats.json:
 *              orc_group_vars, orc_last_file_idx   orchestrator/src/plugins/node_groups/tests.rs:565-676
 *              orc_upload_name_vars                orchestrator/src/api/routes/storage.rs:575-760
 *            and structurally (group counts / sizes / membership) by ports of
 *            orchestrator/src/plugins/node_groups/tests.rs.
 *   PARITY UNPINNED (the reference itself is non-deterministic or depends on
 *            an un-vendored crate / libm here):
 *              - which applicable task a group receives (rand 0.9.1 ThreadRng,
 *                scheduler_impl.rs:66-70, mod.rs:1175-1177) -> injected chooser
 *              - group ids (rand u64, mod.rs:1489-1493)      -> injected generator
 *              - input node order (Redis SMEMBERS)           -> explicit order
 *              - f64 Haversine bit patterns (Rust std -> glibc libm): this
 *                oracle calls the same glibc sin/cos/atan2/sqrt, built with
 *                -ffp-contract=off, but no Rust binary exists here to confirm.
 */
#ifndef PM_ORACLE_H
#define PM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_ALTS 16   /* GPU alternatives per requirement (reference: Vec, tests use <= 4) */
#define ORC_MODEL_LEN 64
#define ORC_NAME_LEN 32
#define ORC_ADDR_LEN 48
#define ORC_MAX_TOPO 4

/* ---- shared/src/models/node.rs:59-70 GpuRequirements (Option<u32> -> flag bit + value) */
enum {
  ORC_G_COUNT = 1u << 0,
  ORC_G_MODEL = 1u << 1,
  ORC_G_MEM = 1u << 2,
  ORC_G_MEM_MIN = 1u << 3,
  ORC_G_MEM_MAX = 1u << 4,
  ORC_G_TOT_MIN = 1u << 5,
  ORC_G_TOT_MAX = 1u << 6
};
typedef struct {
  uint32_t flags;
  uint32_t count, memory_mb, memory_mb_min, memory_mb_max, total_memory_min, total_memory_max;
  char model[ORC_MODEL_LEN];
} orc_gpu_req;

/* ---- shared/src/models/node.rs:49-57 ComputeRequirements */
enum {
  ORC_R_CPU = 1u << 0,       /* requirements.cpu is Some */
  ORC_R_CPU_CORES = 1u << 1, /* requirements.cpu.cores is Some */
  ORC_R_RAM = 1u << 2,
  ORC_R_STORAGE = 1u << 3
};
typedef struct {
  uint32_t flags;
  uint32_t cpu_cores, ram_mb, storage_gb;
  uint32_t n_gpu;
  orc_gpu_req gpu[ORC_MAX_ALTS];
} orc_requirements;

/* ---- shared/src/models/node.rs:25-35,72-78,153-157 ComputeSpecs / GpuSpecs / CpuSpecs */
enum {
  ORC_S_GPU = 1u << 0, /* specs.gpu is Some */
  ORC_S_G_COUNT = 1u << 1,
  ORC_S_G_MODEL = 1u << 2,
  ORC_S_G_MEM = 1u << 3,
  ORC_S_CPU = 1u << 4, /* specs.cpu is Some */
  ORC_S_CPU_CORES = 1u << 5,
  ORC_S_RAM = 1u << 6,
  ORC_S_STORAGE = 1u << 7
};
typedef struct {
  uint32_t flags;
  uint32_t gpu_count, gpu_memory_mb, cpu_cores, ram_mb, storage_gb;
  char gpu_model[ORC_MODEL_LEN];
} orc_specs;

/* ---- orchestrator/src/models/node.rs:75-85 NodeStatus (declaration order) */
enum {
  ORC_ST_DISCOVERED = 0,
  ORC_ST_WAITING = 1,
  ORC_ST_HEALTHY = 2,
  ORC_ST_UNHEALTHY = 3,
  ORC_ST_DEAD = 4,
  ORC_ST_EJECTED = 5,
  ORC_ST_BANNED = 6,
  ORC_ST_LOWBALANCE = 7
};

/* ---- orchestrator/src/models/node.rs:11-37 OrchestratorNode (fields the path reads) */
typedef struct {
  char address[ORC_ADDR_LEN]; /* Address::to_string() */
  uint32_t status;
  uint32_t has_p2p;      /* p2p_id.is_some() */
  uint32_t has_specs;    /* compute_specs.is_some() */
  uint32_t has_location; /* location.is_some() */
  orc_specs specs;
  double latitude, longitude;
} orc_node;

/* ---- orchestrator/src/plugins/node_groups/mod.rs:30-37 NodeGroupConfiguration */
typedef struct {
  char name[ORC_NAME_LEN];
  uint64_t min_group_size, max_group_size;
  uint32_t has_requirements; /* compute_requirements.is_some() */
  uint32_t _pad;
  orc_requirements req;
} orc_config;

/* ---- shared/src/models/task.rs:162-184 Task + :58-61 SchedulingConfig, projected to what
 * scheduler_impl.rs:42-61 / mod.rs:1134-1162 read: created_at and
 * scheduling_config.plugins["node_groups"]["allowed_topologies"]. */
typedef struct {
  int64_t created_at;
  uint32_t restricted;   /* 1 iff the allowed_topologies key exists (any None on the way => 0) */
  uint32_t n_topologies; /* may be 0 with restricted=1: then nothing is allowed */
  char topologies[ORC_MAX_TOPO][ORC_NAME_LEN];
} orc_task;

/* ------------------------------------------------------------------ pure functions */

/* shared/src/models/node.rs:180-374  ComputeRequirements::from_str.
 * Returns 0 = Ok, 1 = Err(..), 2 = the reference would panic (the `.unwrap()` on a
 * non-numeric value inside the min/max cross checks, :239,:262,:287,:306).
 * err (optional) receives a short message. */
int orc_parse_requirements(const char* s, orc_requirements* out, char* err, size_t errlen);

/* shared/src/models/node.rs:377-541  ComputeSpecs::meets (+ GpuSpecs::meets, CpuSpecs::meets). */
int orc_meets(const orc_specs* specs, const orc_requirements* req);

/* shared/src/models/node.rs:463-484  the model-string rule alone (Unicode to_lowercase, Unicode trim). */
int orc_model_matches(const char* spec_model, const char* req_model);
/* str::to_lowercase (Unicode: char::to_lowercase per code point + the final-sigma rule), at most cap - 1 bytes + NUL */
void orc_to_lowercase_str(const char* in, char* out, size_t cap);

/* orchestrator/src/plugins/node_groups/mod.rs:206-215 */
int orc_is_node_compatible_with_config(const orc_config* cfg, const orc_node* node);

/* orchestrator/src/plugins/node_groups/mod.rs:218-231  Haversine, glibc libm. */
double orc_calculate_distance(double lat1, double lon1, double lat2, double lon2);
void orc_distance_column(double lat0, double lon0, const double* lat, const double* lon, size_t n, double* out);

/* orchestrator/src/plugins/node_groups/mod.rs:138-164: validity panics + template sort.
 * order_out[i] = index (into cfgs) of the i-th template after the stable sort.
 * Returns 0, or 2 if the reference constructor would panic (duplicate name / max<min). */
int orc_sort_configs(const orc_config* cfgs, size_t n, uint32_t* order_out);

/* orchestrator/src/plugins/node_groups/mod.rs:399-418: filter templates (already in
 * orc_sort_configs order) by the enabled set, then stable re-sort min_group_size desc.
 * enabled[i] refers to cfgs[i].  Returns the number written to order_out. */
size_t orc_available_configs(const orc_config* cfgs, const uint32_t* template_order, size_t n,
                             const uint8_t* enabled, uint32_t* order_out);

/* orchestrator/src/plugins/newest_task/mod.rs:8-19: max_by_key(created_at) — LAST max wins.
 * Returns the task index or -1 for an empty slice. */
int64_t orc_newest_task(const orc_task* tasks, size_t n);

/* scheduler_impl.rs:42-61 / mod.rs:1134-1162 topology predicate for one (task, config name). */
int orc_task_applicable(const orc_task* t, const char* config_name);

/* ------------------------------------------------------------------ swarm state */

typedef struct orc_state orc_state;

enum { ORC_CHOOSE_FIRST = 0, ORC_CHOOSE_SEEDED = 1 };

typedef struct {
  uint32_t proximity_enabled;     /* ProximityOptimizationPolicy.enabled   (mod.rs:85-88: default true) */
  uint32_t switching_enabled;     /* TaskSwitchingPolicy.enabled           (mod.rs:90-97: default true) */
  uint32_t prefer_larger_groups;  /* TaskSwitchingPolicy.prefer_larger_groups (default true) */
  uint32_t chooser;               /* injected replacement for rand::rng().choose */
  uint64_t chooser_seed;
  uint64_t group_id_seed;         /* injected replacement for generate_group_id (mod.rs:1489-1493) */
  uint32_t reference_shaped;      /* 1: re-filter `meets` every carve step and sort with the
                                     2-Haversines-per-comparison comparator like mod.rs:511-542;
                                     0: masks once, distances cached ("best-effort CPU") */
  uint32_t _pad;
} orc_policy;

/* nodes/cfgs/tasks are borrowed; they must outlive the state.  cfgs are the *templates* in
 * caller order; the state applies orc_sort_configs itself.  Returns NULL if the reference
 * constructor would panic. */
orc_state* orc_state_new(const orc_node* nodes, size_t n_nodes, const orc_config* cfgs, size_t n_cfgs,
                         const orc_policy* policy);
void orc_state_free(orc_state*);

/* "available_node_group_configs" Redis set (mod.rs:1328-1348): enabled[i] for cfgs[i]. */
void orc_state_set_enabled(orc_state*, const uint8_t* enabled);
/* TaskStore contents, already in get_all_tasks order (task_store.rs:79). Borrowed. */
void orc_state_set_tasks(orc_state*, const orc_task* tasks, size_t n_tasks);
/* node status flip (the rows are caller-owned; this also runs handle_status_change,
 * status_update_impl.rs:8-39: Dead|LowBalance => dissolve the node's group). */
void orc_state_set_node_status(orc_state*, orc_node* nodes_mut, size_t idx, uint32_t status);

/* Task table replaced (get_all_tasks changed): map[i] = new index of old task i, or -1 if it was deleted.
 * Groups holding a deleted task are dissolved (on_task_deleted, mod.rs:1259-1288). */
void orc_state_remap_tasks(orc_state*, const int64_t* map, size_t n_old);

/* mod.rs:478-628 try_form_new_groups. Returns number of groups formed this call. */
size_t orc_try_form_new_groups(orc_state*);
/* mod.rs:631-971 try_merge_solo_groups (+ find_best_task_for_group :1122-1189 via chooser).
 * Returns number of merged groups created. */
size_t orc_try_merge_solo_groups(orc_state*);
/* mod.rs:1423-1487 dissolve_group by slot. */
void orc_dissolve_group(orc_state*, uint32_t group_slot);

/* scheduler_impl.rs:11-110 NodeGroupsPlugin::filter_tasks for node `idx`, including the SETNX
 * claim (:74).  Returns the task index or -1 (empty Vec).  Optional outs (may be NULL):
 * group_index = GROUP_INDEX (mod.rs:424-434), group_size, next_node = index of the node whose
 * p2p id becomes NEXT_P2P_ADDRESS (scheduler_impl.rs:115-128). */
int64_t orc_filter_tasks_node_groups(orc_state*, size_t node_idx, uint32_t* group_index,
                                     uint32_t* group_size, uint32_t* next_node);

/* scheduler/mod.rs:26-36 Scheduler::get_task_for_node with plugins = [NodeGroupsPlugin] when
 * use_node_groups, else the default [NewestTaskPlugin] (:16-19).  Returns task idx or -1. */
int64_t orc_get_task_for_node(orc_state*, size_t node_idx, int use_node_groups);

/* ---- state read-back */
size_t orc_n_groups(const orc_state*);        /* live groups */
size_t orc_group_slots(const orc_state*);     /* slots ever allocated (live or dissolved) */
/* node -> group slot or -1 ("node_to_group" hash) */
const int32_t* orc_node_to_group(const orc_state*);
/* Per slot: returns 0 if the slot is dissolved. members receives up to cap node indices in
 * BTreeSet<String> (address byte) order. */
int orc_group_info(const orc_state*, uint32_t slot, uint64_t* id, uint32_t* config_idx,
                   uint32_t* n_members, uint32_t* members, size_t cap, int64_t* task_idx);
/* last orc_try_form_new_groups: number of Haversine evaluations and `meets` evaluations done */
void orc_counters(const orc_state*, uint64_t* n_haversine, uint64_t* n_meets);

/* What the webhook plugins would have been called with, in call order: send_group_created for every group of
 * try_form_new_groups (mod.rs:612-625); per merge send_group_destroyed for each dissolved solo group, then
 * send_group_created for the merged one (mod.rs:974-1000); send_group_destroyed in dissolve_group
 * (mod.rs:1469-1481).  Members in group.nodes (BTreeSet<String>) order.  orc_events returns the number of logged
 * events and copies them out when the buffers are large enough. */
enum { ORC_GROUP_CREATED = 1, ORC_GROUP_DESTROYED = 2 };
typedef struct orc_group_event {
  uint64_t group_id;
  uint32_t kind, config, member_begin, n_members;
} orc_group_event;
size_t orc_events(const orc_state*, orc_group_event* out, size_t cap, uint32_t* members, size_t cap_members,
                  size_t* n_members);
void orc_events_clear(orc_state*);

/* splitmix64 — the one PRNG used by the generator, the chooser and the group-id stream. */
uint64_t orc_splitmix64(uint64_t* state);

/* Whole-table helpers used by parity tests and the cpu_baseline timing leg. */
/* W x C compat sweep: mask_out[w] bit i = is_node_compatible_with_config(cfgs[i], nodes[w]). */
void orc_compat_masks(const orc_node* nodes, size_t n_nodes, const orc_config* cfgs, size_t n_cfgs,
                      uint64_t* mask_out);
/* T x W pair sweep in the reference's orientation (scheduler_impl.rs:42-61 run once per node):
 * for node w whose group has configuration cfg_of_node[w] (-1: not in a group), first_out[w] =
 * index of the first applicable task (UINT32_MAX none) and count_out[w] = number applicable. */
void orc_pair_sweep_per_worker(const orc_task* tasks, size_t n_tasks, const orc_config* cfgs,
                               const int32_t* cfg_of_node, size_t n_nodes, uint32_t* first_out,
                               uint32_t* count_out);

/* the same, nodes split over n_threads POSIX threads (returns 1 if a thread could not be started and its share ran
 * on the caller) */
int orc_pair_sweep_per_worker_mt(const orc_task* tasks, size_t n_tasks, const orc_config* cfgs,
                                 const int32_t* cfg_of_node, size_t n_nodes, uint32_t* first_out,
                                 uint32_t* count_out, uint32_t n_threads);

/* ---- group variables of the task handed to a grouped node (scheduler_impl.rs:155-200) and of the upload
 * file name (orchestrator/src/api/routes/storage.rs:150-215).  Each returns a malloc'd string (free with
 * orc_free_string). */
char* orc_group_vars(const char* in, uint32_t group_index, uint32_t group_size, const char* next_p2p_address,
                     const char* group_id, const char* total_upload_count);
char* orc_volume_vars(const char* in, const char* group_id);
/* group_id == NULL: the node is in no group (storage.rs:159-163) */
char* orc_upload_name_vars(const char* in, const char* group_id, uint32_t group_size, uint32_t group_index,
                           uint64_t upload_count);
uint32_t orc_last_file_idx(const char* total_upload_count);
void orc_free_string(char* s);

#ifdef __cplusplus
}
#endif
#endif
