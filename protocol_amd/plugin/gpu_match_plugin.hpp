// gpu_match_plugin.hpp — the host side of the drop-in, compiled: the orchestrator's scheduler types and the plugin that
// binds libpm_engine.so (include/pm_engine.h), in C++ because this image has no Rust toolchain.  It is the twin of
// rust/gpu_match_plugin.rs (the binding a maintainer adds to the reference; never compiled here), method for method and
// lock for lock, so that what the Rust source says is also said by something a compiler, the sanitizers and the tests
// have seen.  Names, argument meaning and error behaviour follow the reference (all paths relative to
// /root/reference/crates):
//
//   OrchestratorNode, NodeStatus          orchestrator/src/models/node.rs:11-37, :75-85 (the fields the path reads)
//   ComputeSpecs, GpuSpecs, CpuSpecs      shared/src/models/node.rs (compute_specs of a node)
//   Task, VolumeMount                     shared/src/models/task.rs:163-184, :60-98
//   NodeGroupConfiguration                orchestrator/src/plugins/node_groups/mod.rs:30-37
//   SchedulerPlugin::filter_tasks         orchestrator/src/plugins/mod.rs:60-79  (enum dispatch -> virtual call)
//   NewestTaskPlugin                      orchestrator/src/plugins/newest_task/mod.rs:8-20
//   Scheduler::get_task_for_node          orchestrator/src/scheduler/mod.rs:9-76, with the edit INTEGRATION.md shows
//                                         ("The task list per heartbeat": a chain headed by the engine's plugin does
//                                         not load the store's task list)
//   GpuMatchPlugin                        replaces NodeGroupsPlugin: new (mod.rs:113-175), the management loop's body
//                                         (tick = try_form_new_groups + try_merge_solo_groups, mod.rs:180-203),
//                                         filter_tasks (scheduler_impl.rs:11-205), handle_status_change
//                                         (status_update_impl.rs:8-39), the task observers (mod.rs:1224-1325)
//   WebhookPlugin                         orchestrator/src/plugins/webhook/mod.rs:240-266 (the two calls the path makes)
//   NodeGroup + the plugin's READ SURFACE  node_groups/mod.rs:63-69; get_node_group :324-337, get_node_groups_batch
//                                         :339-397, get_available_configurations :399-418,
//                                         get_all_configuration_templates :420-422, get_idx_in_group :424-434,
//                                         dissolve_group :1002-1004 (-> :1423-1487), get_all_groups :1006-1044,
//                                         get_group_by_id :1046-1055, get_all_node_group_mappings :1057-1065 — what the
//                                         API routes call on AppState.node_groups_plugin (api/routes/groups.rs:34-160,
//                                         :319-360, nodes.rs:68-90, storage.rs:147-156, metrics/sync_service.rs:57-75,
//                                         :273-274); get_task_topologies :1407-1421 (api/routes/task.rs:68-72)
//
// Errors: the Rust returns anyhow::Result and panics in the constructor; here every failed engine call throws
// EngineError (code + pm_last_error text) and the constructor's panics are std::invalid_argument with the reference's
// messages.  Rust's async is not mirrored: every method is a plain blocking call, as the FFI calls underneath are.
#ifndef PM_GPU_MATCH_PLUGIN_HPP
#define PM_GPU_MATCH_PLUGIN_HPP

#include <atomic>
#include <chrono>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "pm_engine.h"

namespace orchestrator {

// alloy::primitives::Address as the path uses it: a key, and `address.to_string()` (GROUP_INDEX is the rank of that
// string inside the group, node_groups/mod.rs:424-434; ${NODE_ADDRESS}).  Held as that string.
struct Address {
  std::string text;
  Address() = default;
  explicit Address(std::string s) : text(std::move(s)) {}
  const std::string& to_string() const { return text; }
  bool operator==(const Address& o) const { return text == o.text; }
  bool operator!=(const Address& o) const { return text != o.text; }
  static Address zero() { return Address("0x0000000000000000000000000000000000000000"); }
};
struct AddressHash {
  size_t operator()(const Address& a) const { return std::hash<std::string>()(a.text); }
};

enum class NodeStatus { Discovered, WaitingForHeartbeat, Healthy, Unhealthy, Dead, Ejected, Banned, LowBalance };

struct GpuSpecs {
  std::optional<uint32_t> count;
  std::optional<std::string> model;
  std::optional<uint32_t> memory_mb;
};
struct CpuSpecs {
  std::optional<uint32_t> cores;
};
struct ComputeSpecs {
  std::optional<GpuSpecs> gpu;
  std::optional<CpuSpecs> cpu;
  std::optional<uint32_t> ram_mb;
  std::optional<uint32_t> storage_gb;
};
struct NodeLocation {
  double latitude = 0.0, longitude = 0.0;
};
struct OrchestratorNode {
  Address address;
  NodeStatus status = NodeStatus::Discovered;
  std::optional<std::string> p2p_id;
  std::optional<ComputeSpecs> compute_specs;
  std::optional<NodeLocation> location;
};

struct VolumeMount {
  std::string host_path, container_path;
  // shared/src/models/task.rs:75-98 (${TIMESTAMP} takes `now`: the caller's clock, seconds)
  VolumeMount replace_labels(const std::string& task_id, const std::optional<std::string>& node_address, int64_t now) const;
};

struct Task {
  std::string id;    // Uuid, hyphenated lower-case hex (task.id.to_string())
  std::string name;
  std::string image;
  int64_t created_at = 0;
  // scheduling_config.plugins["node_groups"]["allowed_topologies"] (scheduler_impl.rs:44-59); nullopt = any None on the way
  std::optional<std::vector<std::string>> allowed_topologies;
  std::optional<std::map<std::string, std::string>> env_vars;
  std::optional<std::vector<std::string>> cmd;
  std::optional<std::vector<VolumeMount>> volume_mounts;
  bool operator==(const Task& o) const;
};
// task.id.as_u64_pair().1: the low 64 bits of the UUID — the identity the engine keeps a claim by
uint64_t task_uid(const Task& t);

struct NodeGroupConfiguration {
  std::string name;
  size_t min_group_size = 0, max_group_size = 0;
  // the reference holds a ComputeRequirements parsed by FromStr at deserialisation (shared/src/models/node.rs:180-374);
  // here the string travels and the library's parser (pm_host_parse_requirements) does that step
  std::optional<std::string> compute_requirements;
};

// NodeGroup (node_groups/mod.rs:63-69)
struct NodeGroup {
  std::string id;                  // generate_group_id: format!("{:x}", u64) (mod.rs:1489-1493)
  std::vector<std::string> nodes;  // BTreeSet<String>: address.to_string() in byte order
  int64_t created_at = 0;          // chrono::DateTime<Utc>, as milliseconds since the epoch: the plugin's clock when the
                                   // creation was reported to it (the reference stamps Utc::now() in the same loop pass, mod.rs:575)
  std::string configuration_name;
  bool operator==(const NodeGroup& o) const {
    return id == o.id && nodes == o.nodes && created_at == o.created_at && configuration_name == o.configuration_name;
  }
};

// the two calls of webhook/mod.rs the path makes; `nodes` in group.nodes (BTreeSet<String>) order
class WebhookPlugin {
 public:
  virtual ~WebhookPlugin() = default;
  virtual void send_group_created(const std::string& group_id, const std::string& configuration_name,
                                  const std::vector<std::string>& nodes) = 0;
  virtual void send_group_destroyed(const std::string& group_id, const std::string& configuration_name,
                                    const std::vector<std::string>& nodes) = 0;
};

class EngineError : public std::runtime_error {
 public:
  EngineError(int32_t code, const std::string& what) : std::runtime_error(what), code_(code) {}
  int32_t code() const { return code_; }

 private:
  int32_t code_;
};

// SchedulerPlugin (plugins/mod.rs:60-79): the reference's enum, as an interface
class SchedulerPlugin {
 public:
  virtual ~SchedulerPlugin() = default;
  virtual std::vector<Task> filter_tasks(const std::vector<Task>& tasks, const Address& node_address) = 0;
  // (true for the plugin that serves from its own task list — the scheduler then loads none)
  virtual bool serves_from_own_task_list() const { return false; }
};

class NewestTaskPlugin : public SchedulerPlugin {
 public:
  std::vector<Task> filter_tasks(const std::vector<Task>& tasks, const Address&) override;
};

// The collective of the multi-GPU tick (include/pm_engine.h "multi-GPU"): one process per GPU, every rank holds
// the whole swarm and runs the whole carve (replicated), the pair sweep + claim run for the OWNED workers, and the published
// rows are all-gathered ONCE per tick.  The plugin is handed the communicator; it owns none.
//   RcclAllGather (rccl_all_gather.hpp)   ncclAllGather over xGMI on the stream the engine's kernels run on
//   LocalAllGather (same header)          ranks that live in one process (tests; several ranks sharing a GPU)
class AllGather {
 public:
  virtual ~AllGather() = default;
  virtual uint32_t rank() const = 0;
  virtual uint32_t world() const = 0;
  // the hipStream_t the engine's kernels and the collective share (ordered without a host wait); nullptr = the engine's own
  virtual void* stream() const = 0;
  // device pointers; recv = [world][bytes_per_rank], send = this rank's segment (it may alias recv's own slot): enqueue on stream()
  virtual void all_gather(const void* send, void* recv, size_t bytes_per_rank) = 0;
};

// owner rank of a node: splitmix64 finaliser of the low 8 bytes of the address, mod world
uint32_t shard_of(const Address& a, uint32_t world);

// The third variant.  Thread-safe like the reference's Arc<NodeGroupsPlugin>: heartbeats (filter_tasks) from any
// thread beside the management loop (sync_nodes / tick), the status updater (handle_status_change) and the task
// store's observers (on_task_created / on_task_deleted).
// LOCK ORDER: `nodes_`, then `tasks_`, then the engine's own mutex (inside every pm_* call but the look-up) — the
// order rust/gpu_match_plugin.rs states; tests/cpp/plugin_test.cpp runs it under ThreadSanitizer.
class GpuMatchPlugin : public SchedulerPlugin {
 public:
  using UploadCounter = std::function<size_t(const Address&, const std::string& group_id)>;

  // NodeGroupsPlugin::new's contract (mod.rs:113-175): duplicate names -> "Configuration names must be unique",
  // max < min (or min == 0, which the engine refuses too) -> "Plugin configuration is invalid"; both
  // std::invalid_argument.  An unparsable requirement string is std::invalid_argument as well (the reference fails
  // earlier, at deserialisation).
  GpuMatchPlugin(std::vector<NodeGroupConfiguration> templates, int32_t device, UploadCounter upload_counter,
                 std::vector<std::shared_ptr<WebhookPlugin>> webhook_plugins = {});
  ~GpuMatchPlugin() override;
  GpuMatchPlugin(const GpuMatchPlugin&) = delete;
  GpuMatchPlugin& operator=(const GpuMatchPlugin&) = delete;

  // every management interval, with node_store.get_nodes() (any order): new nodes are appended, changed rows
  // rewritten, departed nodes tombstoned (their groups dissolve)
  void sync_nodes(const std::vector<OrchestratorNode>& snapshot);
  // task_store.get_all_tasks() (created_at descending): start-up, and the fallback when a delta does not apply
  void sync_tasks(std::vector<Task> tasks);
  // the task store's observers (mod.rs:1224-1243, :1245-1325)
  void on_task_created(const Task& task, const std::function<std::vector<Task>()>& all_tasks);
  void on_task_deleted(const Task& task);
  // one body of run_group_management_loop (mod.rs:180-203) + every worker's filter_tasks, then the webhooks
  pm_stats tick();
  // A process that serves several pools on one GPU (INTEGRATION.md "Several pools on one GPU"): one pm_tick_many call —
  // every pool's carve started before the first is waited for — then each pool's webhooks.  pools[i]->tick() K times
  // in a row matches one pool after the other.
  static std::vector<pm_stats> tick_many(const std::vector<GpuMatchPlugin*>& pools);
  // The management interval of ONE pool matched by several GPUs, one process (or thread) per GPU: this plugin is rank
  // comm.rank() of comm.world().  Every rank is fed every store event (sync_nodes, the task observers, status changes —
  // replicated calls) and calls tick_dist at the same point of its loop; every rank ends with the identical groups and
  // the full published table (any rank answers any heartbeat).  The five calls of INTEGRATION.md "Multi-GPU":
  // pm_dist_tick_begin, pm_dist_carve_wait, pm_dist_match_begin, the ONE all-gather, pm_dist_tick_end.  From its first
  // tick_dist on a plugin of rank > 0 delivers no webhooks (every rank sees every creation and dissolution: rank 0 reports
  // them; the others drain their feed and keep the created_at stamps).  Called by the management loop's thread only
  // (like tick); ownership is recomputed when the node table grew.
  pm_stats tick_dist(AllGather& comm);
  // SchedulerPlugin::filter_tasks: `tasks` is ignored (the plugin serves from its own list)
  std::vector<Task> filter_tasks(const std::vector<Task>& tasks, const Address& node_address) override;
  bool serves_from_own_task_list() const override { return true; }
  // StatusUpdatePlugin::handle_status_change (status_update_impl.rs:8-39)
  void handle_status_change(const OrchestratorNode& node);

  // ---- the read surface the API routes use (AppState.node_groups_plugin in the reference; INTEGRATION.md "The routes").
  // Host-side state of the engine only: no GPU work, any thread, also while a tick runs (it then waits for the tick).
  // get_all_groups (mod.rs:1006-1044): every group, sorted by id text (:1040)
  std::vector<NodeGroup> get_all_groups() const;
  // get_group_by_id (mod.rs:1046-1055): an id that is not the "{:x}" text of a live group's id is nullopt
  std::optional<NodeGroup> get_group_by_id(const std::string& group_id) const;
  // get_all_node_group_mappings (mod.rs:1057-1065): node address text -> group id text (HGETALL node_to_group)
  std::unordered_map<std::string, std::string> get_all_node_group_mappings() const;
  // get_node_group (mod.rs:324-337): the group of the node with this address TEXT (the key of the reference's hash)
  std::optional<NodeGroup> get_node_group(const std::string& node_addr) const;
  // get_node_groups_batch (mod.rs:339-397): every asked address is a key of the result; one snapshot of the engine's list
  std::unordered_map<std::string, std::optional<NodeGroup>> get_node_groups_batch(const std::vector<std::string>& node_addresses) const;
  // get_idx_in_group (mod.rs:424-434): position in group.nodes; "Node {} not found in group" -> std::out_of_range
  size_t get_idx_in_group(const NodeGroup& node_group, const std::string& node_addr) const;
  // get_available_configurations (mod.rs:399-418): the templates some task names, min_group_size descending (stable)
  std::vector<NodeGroupConfiguration> get_available_configurations() const;
  // get_all_configuration_templates (mod.rs:420-422): in the constructor's order (mod.rs:150-164)
  std::vector<NodeGroupConfiguration> get_all_configuration_templates() const;
  // dissolve_group (mod.rs:1002-1004 -> :1423-1487): by id text; an unknown id is not an error; sends the webhook
  void dissolve_group(const std::string& group_id);

  // chrono::Utc::now() for NodeGroup.created_at, milliseconds since the epoch (tests inject their own)
  std::function<int64_t()> clock = [] {
    return int64_t(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count());
  };

  // on_task_created also re-matches the standing groups (pm_tasks_insert_front_ex, republish = 1)
  std::atomic<bool> republish_on_insert{false};

  pm_engine* engine_ptr() const { return engine_; }   // (a process that serves several pools: pm_tick_many)
  // what the tests look at
  size_t known_nodes() const;
  std::optional<uint32_t> row_of(const Address& a) const;

 private:
  struct Row {
    uint32_t flags = 0, gpu_count = 0, gpu_mem = 0, gpu_class = 0, cpu_cores = 0, ram = 0, storage = 0;
    double lat = 0.0, lon = 0.0;
    bool operator==(const Row& o) const;
    bool operator!=(const Row& o) const { return !(*this == o); }
  };
  struct RowColumns {
    std::vector<uint32_t> flags, gpu_count, gpu_mem, gpu_class, cpu_cores, ram, storage, addr_rank;
    std::vector<double> lat, lon;
    void push(const Row& r, uint32_t rank);
    pm_worker_soa soa() const;
  };
  struct NodeTable {
    std::unordered_map<Address, uint32_t, AddressHash> index;
    std::vector<Address> addresses;
    std::vector<std::string> address_strings, p2p_ids;
    std::vector<Row> rows;
    std::vector<bool> present;
    std::vector<uint32_t> by_address;   // rows in address-string order (kept sorted: a new node is one binary search)
    std::vector<std::string> spec_models;
    std::unordered_map<std::string, uint32_t> spec_model_index;
  };

  void check(int32_t rc) const;
  void set_configs(const std::vector<NodeGroupConfiguration>& templates);
  void push_model_table(const NodeTable& nodes) const;
  static Row project(const OrchestratorNode& node, NodeTable& table, bool* new_model);
  static std::vector<uint32_t> address_ranks(const std::vector<uint32_t>& by_address, size_t known);
  uint64_t topology_mask(const Task& t) const;
  void push_enabled(const std::vector<Task>& tasks);
  void sync_tasks_locked(std::vector<Task>& guard, std::vector<Task> tasks);
  void emit_group_webhooks();
  struct GroupSnapshot {
    std::vector<int32_t> group_of;   // per row: index into `groups` or -1
    std::vector<pm_group> groups;
    std::vector<uint32_t> members;
  };
  GroupSnapshot snapshot_groups(bool want_group_of) const;
  NodeGroup make_group(const NodeTable& t, const pm_group& g, const uint32_t* members) const;
  std::optional<uint32_t> row_of_address_text(const NodeTable& t, const std::string& text) const;

  pm_engine* engine_ = nullptr;
  std::vector<NodeGroupConfiguration> templates_;   // caller order: the engine's configuration index
  std::vector<pm_config_row> config_rows_;          // what pm_set_configs was given (pm_host_config_order reads sizes + PM_R_HAS_REQ)
  std::atomic<uint64_t> enabled_mask_{0};           // "available_node_group_configs" as last pushed (push_enabled)
  mutable std::mutex group_meta_mu_;                // LEAF lock: never held across an engine call or another lock
  mutable std::unordered_map<uint64_t, int64_t> group_created_at_;
  std::vector<std::string> config_names_;
  std::vector<std::string> req_models_;   // requirement model strings, one per pm_gpu_alt_row.model_row
  mutable std::shared_mutex nodes_mu_;
  NodeTable nodes_;
  bool engine_rows_stale_ = false;   // an engine call of sync_nodes failed half-way: the next one re-sends every row
  // tick_dist's (the management loop's thread only): what pm_dist_configure was last told
  std::atomic<uint32_t> dist_rank_{0};   // (read by emit_group_webhooks on any thread)
  uint32_t dist_world_ = 1;
  size_t dist_rows_ = size_t(-1);
  void* dist_stream_ = nullptr;
  mutable std::shared_mutex tasks_mu_;
  std::vector<Task> tasks_;               // get_all_tasks order: the engine reports positions in this list
  UploadCounter upload_counter_;
  std::vector<std::shared_ptr<WebhookPlugin>> webhook_plugins_;
};

// store_context.task_store, as far as the scheduler uses it (task_store.rs:57-82)
class TaskStore {
 public:
  virtual ~TaskStore() = default;
  virtual std::vector<Task> get_all_tasks() = 0;
};

// get_task_topologies (node_groups/mod.rs:1407-1421): what create_task checks when the grouping plugin is active
// ("No topology found for task but grouping plugin is active", api/routes/task.rs:68-75): the task's allowed_topologies,
// empty when any Option on the way is None
std::vector<std::string> get_task_topologies(const Task& task);

// The storage route's file name (api/routes/storage.rs:147-185, after generate_file_name): ${NODE_GROUP_ID},
// ${NODE_GROUP_SIZE}, ${NODE_GROUP_INDEX} through get_node_group + get_idx_in_group when the node is in a group, then
// ${TOTAL_UPLOAD_COUNT_AFTER} / ${CURRENT_FILE_INDEX} from count_uploads(address, group id or "no-group") — the number of
// `upload:<address>:<group|no-group>:*` keys, which stays with the store (storage.rs:170-199).
// *group_id_out (may be null) = the group id the route keys its upload counter by, empty when the node is in no group.
std::string upload_file_name(const GpuMatchPlugin& plugin, const std::string& file_name, const std::string& address,
                             const std::function<uint64_t(const std::string& address, const std::string& group_or_no_group)>& count_uploads,
                             std::string* group_id_out = nullptr);

class Scheduler {
 public:
  // Scheduler::new (scheduler/mod.rs:14-24): an empty chain gets the NewestTaskPlugin
  Scheduler(std::shared_ptr<TaskStore> task_store, std::vector<std::shared_ptr<SchedulerPlugin>> plugins);
  // scheduler/mod.rs:26-76; `now` feeds ${TIMESTAMP} in volume mounts (chrono::Utc::now() in the reference)
  std::optional<Task> get_task_for_node(const Address& node_address, int64_t now = 0);

 private:
  std::shared_ptr<TaskStore> task_store_;
  std::vector<std::shared_ptr<SchedulerPlugin>> plugins_;
};

}  // namespace orchestrator
#endif
