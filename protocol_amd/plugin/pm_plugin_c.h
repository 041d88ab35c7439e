/* pm_plugin_c.h — a flat C face of the compiled host side (gpu_match_plugin.hpp), so that the Python test harness can
 * drive the C++ GpuMatchPlugin + Scheduler with the scenarios it drives tests/shim_replay.py with, against the oracle
 * on a GPU and against tests/cpp/mock_engine.cpp without one.  Not part of the drop-in boundary (that is
 * include/pm_engine.h below the plugin, and the reference's own Rust interface above it): a test driver's handle on a
 * C++ object.  One pmx_plugin = a GpuMatchPlugin, a recording WebhookPlugin, a TaskStore whose list the caller sets,
 * and a Scheduler whose chain is headed by the plugin.  Every call returns 0 or -1 (pmx_last_error says why); strings
 * are NUL-terminated UTF-8; text results follow the two-call convention of the pm_host_* helpers (*needed = strlen + 1;
 * -2 when cap is too small, nothing written). */
#ifndef PM_PLUGIN_C_H
#define PM_PLUGIN_C_H

#include <stddef.h>
#include <stdint.h>

#include "pm_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmx_plugin pmx_plugin;

typedef struct {
  const char* name;
  uint32_t min_group_size, max_group_size;
  const char* compute_requirements; /* NULL = None */
} pmx_config;

/* NodeStatus, in the reference's order (orchestrator/src/models/node.rs:75-85) */
enum { PMX_DISCOVERED = 0, PMX_WAITING_FOR_HEARTBEAT, PMX_HEALTHY, PMX_UNHEALTHY, PMX_DEAD, PMX_EJECTED, PMX_BANNED, PMX_LOW_BALANCE };

/* one OrchestratorNode: `has` says which Options are Some — the PM_W_* bits of include/pm_engine.h minus PM_W_HEALTHY
 * (the status says that) */
typedef struct {
  const char* address; /* address.to_string() */
  uint32_t status;     /* PMX_* */
  uint32_t has;        /* PM_W_HAS_SPECS | PM_W_HAS_GPU | ... | PM_W_HAS_P2P | PM_W_HAS_LOC */
  const char* p2p_id;
  uint32_t gpu_count, gpu_memory_mb;
  const char* gpu_model;
  uint32_t cpu_cores, ram_mb, storage_gb;
  double latitude, longitude;
} pmx_node;

typedef struct {
  const char* id; /* Uuid text */
  const char* name;
  int64_t created_at;
  int32_t n_topologies; /* -1 = no allowed_topologies (any None on the way) */
  const char* const* topologies;
  uint32_t n_env;
  const char* const* env_keys;
  const char* const* env_values;
  int32_t n_cmd; /* -1 = None */
  const char* const* cmd;
  int32_t n_mounts; /* -1 = None */
  const char* const* mount_host;
  const char* const* mount_container;
} pmx_task;

const char* pmx_last_error(void);
int32_t pmx_create(const pmx_config* cfgs, uint32_t n, int32_t device, pmx_plugin** out);
void pmx_destroy(pmx_plugin*);
pm_engine* pmx_engine(pmx_plugin*); /* the plugin's engine (borrowed): pm_get_groups & co. for the parity checks */
void pmx_set_upload_count(pmx_plugin*, uint64_t n); /* what the upload counter answers (the Redis key count) */
void pmx_set_republish_on_insert(pmx_plugin*, uint32_t on);

int32_t pmx_sync_nodes(pmx_plugin*, const pmx_node* nodes, uint32_t n);
/* the store's list := tasks (get_all_tasks order), then GpuMatchPlugin::sync_tasks */
int32_t pmx_sync_tasks(pmx_plugin*, const pmx_task* tasks, uint32_t n);
/* TaskStore::add_task: the store's list gets the task at its created_at-descending place, then the observer runs */
int32_t pmx_on_task_created(pmx_plugin*, const pmx_task* task);
/* TaskStore::delete_task by id, then the observer (unknown id: -1) */
int32_t pmx_on_task_deleted(pmx_plugin*, const char* task_id);
int32_t pmx_handle_status_change(pmx_plugin*, const char* address, uint32_t status);
int32_t pmx_tick(pmx_plugin*, pm_stats* stats);
int32_t pmx_row_of(pmx_plugin*, const char* address, uint32_t* row); /* -1 if the plugin does not know the address */
uint32_t pmx_known_nodes(pmx_plugin*);
uint32_t pmx_store_loads(pmx_plugin*); /* how often the scheduler asked the store for its task list */

/* Scheduler::get_task_for_node.  Text: empty = None; else lines "id\t<id>", "name\t<name>", "env\t<key>\t<value>" (key
 * order), "cmd\t<arg>", "mount\t<host>\t<container>"; backslash, tab and newline inside a field are \\ \t \n. */
int32_t pmx_get_task_for_node(pmx_plugin*, const char* address, int64_t now, char* out, size_t cap, size_t* needed);
/* the webhooks sent since the last successful call, one per line: "created|destroyed\t<group id>\t<configuration
 * name>\t<node>\t<node>..." */
int32_t pmx_take_webhooks(pmx_plugin*, char* out, size_t cap, size_t* needed);

/* ---- the plugin's read surface (node_groups/mod.rs:324-434, :1002-1065: what the API routes call).  A group is one line
 * "<id>\t<configuration name>\t<created_at ms>\t<node>\t<node>..." (nodes in BTreeSet<String> order). */
void pmx_set_clock(pmx_plugin*, int64_t now_ms); /* what the plugin's clock answers from now on (NodeGroup.created_at) */
int32_t pmx_get_all_groups(pmx_plugin*, char* out, size_t cap, size_t* needed);               /* sorted by id text */
int32_t pmx_get_group_by_id(pmx_plugin*, const char* group_id, char* out, size_t cap, size_t* needed); /* empty = None */
/* get_node_group + get_idx_in_group: empty = None, else "<idx>\t" in front of the group line */
int32_t pmx_get_node_group(pmx_plugin*, const char* address, char* out, size_t cap, size_t* needed);
/* one line per asked address, in the order asked: "<address>\t-" (None) or "<address>\t" + the group line */
int32_t pmx_get_node_groups_batch(pmx_plugin*, const char* const* addresses, uint32_t n, char* out, size_t cap, size_t* needed);
int32_t pmx_get_all_node_group_mappings(pmx_plugin*, char* out, size_t cap, size_t* needed); /* "<address>\t<group id>", by address */
/* get_all_configuration_templates (available_only = 0) / get_available_configurations (1): "<name>\t<min>\t<max>\t<requirements or ->" */
int32_t pmx_get_configurations(pmx_plugin*, uint32_t available_only, char* out, size_t cap, size_t* needed);
int32_t pmx_dissolve_group(pmx_plugin*, const char* group_id);
/* the storage route's file name (api/routes/storage.rs:147-207) through get_node_group + get_idx_in_group; the key count is
 * pmx_set_upload_count's value.  Two lines: the name, then the group id the route keys its counter by ("no-group" if none). */
int32_t pmx_upload_file_name(pmx_plugin*, const char* file_name, const char* address, char* out, size_t cap, size_t* needed);

#ifdef __cplusplus
}
#endif
#endif
