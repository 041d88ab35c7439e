// gpu_match_plugin.cpp — see gpu_match_plugin.hpp.  Statement for statement the sequence of rust/gpu_match_plugin.rs
// over the C ABI (include/pm_engine.h); comments name the Rust function or the reference lines a block follows.
#include "gpu_match_plugin.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

namespace orchestrator {

namespace {

// str::replace: every non-overlapping match, left to right
std::string replace_all(std::string s, const std::string& from, const std::string& to) {
  if (from.empty()) return s;
  size_t at = 0;
  while ((at = s.find(from, at)) != std::string::npos) {
    s.replace(at, from.size(), to);
    at += to.size();
  }
  return s;
}

std::string hex_lower(uint64_t v) {  // format!("{:x}", v): generate_group_id, mod.rs:1489-1493
  char buf[24];
  std::snprintf(buf, sizeof(buf), "%llx", (unsigned long long)v);
  return buf;
}

// the inverse of format!("{:x}", u64): lower-case hex digits, no sign, no prefix, no leading zero (but "0"), <= 16 of them.
// Anything else is the text of no group id (a Redis key that does not exist in the reference).
bool parse_group_id(const std::string& s, uint64_t* out) {
  if (s.empty() || s.size() > 16 || (s.size() > 1 && s[0] == '0')) return false;
  uint64_t v = 0;
  for (char c : s) {
    uint64_t d;
    if (c >= '0' && c <= '9') d = uint64_t(c - '0');
    else if (c >= 'a' && c <= 'f') d = uint64_t(c - 'a' + 10);
    else return false;
    v = (v << 4) | d;
  }
  *out = v;
  return true;
}

// the two-call convention of the pm_host_* string helpers: size, then fill
template <typename F>
std::string render(F&& f, const std::function<void(int32_t)>& check) {
  size_t need = 0;
  check(f(nullptr, 0, &need));
  std::string buf(need, '\0');
  check(f(buf.data(), need, &need));
  if (!buf.empty()) buf.pop_back();  // the terminating NUL
  return buf;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ model types

VolumeMount VolumeMount::replace_labels(const std::string& task_id, const std::optional<std::string>& node_address,
                                        int64_t now) const {
  VolumeMount out = *this;
  for (std::string* p : {&out.host_path, &out.container_path}) {
    *p = replace_all(*p, "${TASK_ID}", task_id);
    if (node_address) *p = replace_all(*p, "${NODE_ADDRESS}", *node_address);
    *p = replace_all(*p, "${TIMESTAMP}", std::to_string(now));
  }
  return out;
}

bool Task::operator==(const Task& o) const {
  auto mounts_eq = [](const std::optional<std::vector<VolumeMount>>& a, const std::optional<std::vector<VolumeMount>>& b) {
    if (a.has_value() != b.has_value()) return false;
    if (!a) return true;
    if (a->size() != b->size()) return false;
    for (size_t i = 0; i < a->size(); ++i)
      if ((*a)[i].host_path != (*b)[i].host_path || (*a)[i].container_path != (*b)[i].container_path) return false;
    return true;
  };
  return id == o.id && name == o.name && image == o.image && created_at == o.created_at &&
         allowed_topologies == o.allowed_topologies && env_vars == o.env_vars && cmd == o.cmd &&
         mounts_eq(volume_mounts, o.volume_mounts);
}

uint64_t task_uid(const Task& t) {
  // Uuid::as_u64_pair().1: the second eight bytes, big-endian = the last sixteen hex digits of the text form
  uint64_t v = 0;
  int digits = 0;
  for (size_t i = t.id.size(); i-- > 0 && digits < 16;) {
    const char c = t.id[i];
    uint64_t d;
    if (c >= '0' && c <= '9') d = uint64_t(c - '0');
    else if (c >= 'a' && c <= 'f') d = uint64_t(c - 'a' + 10);
    else if (c >= 'A' && c <= 'F') d = uint64_t(c - 'A' + 10);
    else continue;  // '-'
    v |= d << (4 * digits);
    ++digits;
  }
  return v;
}

std::vector<Task> NewestTaskPlugin::filter_tasks(const std::vector<Task>& tasks, const Address&) {
  if (tasks.empty()) return {};
  // Iterator::max_by_key returns the LAST of several maxima
  const Task* best = &tasks[0];
  for (const Task& t : tasks)
    if (t.created_at >= best->created_at) best = &t;
  return {*best};
}

// ------------------------------------------------------------------------------------------------ the plugin

static std::atomic<int>& pools_in_process() {  // live GpuMatchPlugin objects
  static std::atomic<int> n{0};
  return n;
}

bool GpuMatchPlugin::Row::operator==(const Row& o) const {
  // (f64 compared like Rust's derived PartialEq: by value)
  return flags == o.flags && gpu_count == o.gpu_count && gpu_mem == o.gpu_mem && gpu_class == o.gpu_class &&
         cpu_cores == o.cpu_cores && ram == o.ram && storage == o.storage && lat == o.lat && lon == o.lon;
}

void GpuMatchPlugin::RowColumns::push(const Row& r, uint32_t rank) {
  flags.push_back(r.flags);
  gpu_count.push_back(r.gpu_count);
  gpu_mem.push_back(r.gpu_mem);
  gpu_class.push_back(r.gpu_class);
  cpu_cores.push_back(r.cpu_cores);
  ram.push_back(r.ram);
  storage.push_back(r.storage);
  addr_rank.push_back(rank);
  lat.push_back(r.lat);
  lon.push_back(r.lon);
}

pm_worker_soa GpuMatchPlugin::RowColumns::soa() const {
  pm_worker_soa w{};
  w.n = uint32_t(flags.size());
  w.flags = flags.data();
  w.gpu_count = gpu_count.data();
  w.gpu_mem_mb = gpu_mem.data();
  w.gpu_model_class = gpu_class.data();
  w.cpu_cores = cpu_cores.data();
  w.ram_mb = ram.data();
  w.storage_gb = storage.data();
  w.price = nullptr;
  w.addr_rank = addr_rank.data();
  w.lat = lat.data();
  w.lon = lon.data();
  return w;
}

void GpuMatchPlugin::check(int32_t rc) const {
  if (rc == PM_OK) return;
  const char* msg = pm_last_error();
  throw EngineError(rc, "pm_engine error " + std::to_string(rc) + ": " + (msg ? msg : ""));
}

GpuMatchPlugin::GpuMatchPlugin(std::vector<NodeGroupConfiguration> templates, int32_t device, UploadCounter upload_counter,
                               std::vector<std::shared_ptr<WebhookPlugin>> webhook_plugins)
    : upload_counter_(std::move(upload_counter)), webhook_plugins_(std::move(webhook_plugins)) {
  // Hardware queues for the HIP runtime (GPU_MAX_HW_QUEUES, read once at the first HIP call of the process): the
  // PROCESS ENTRY POINT sets it, before any thread exists — a library constructor that writes the environment races
  // with every getenv in a threaded host ("main.rs").  Here it is only looked at: one pool is
  // indifferent to the value, several pools in one process want >= 1 per pool (include/pm_engine.h, pm_set_carve_workgroups).
  {
    std::set<std::string> seen;
    for (const NodeGroupConfiguration& t : templates)
      if (!seen.insert(t.name).second) throw std::invalid_argument("Configuration names must be unique");  // mod.rs:142-144
  }
  pm_engine_config cfg;
  pm_engine_config_default(&cfg);
  cfg.device = device;
  check(pm_engine_create(&cfg, &engine_));
  try {
    for (const NodeGroupConfiguration& t : templates) config_names_.push_back(t.name);
    set_configs(templates);
    templates_ = std::move(templates);
    // an empty worker / task table, so that the delta calls have something to extend
    RowColumns empty;
    const pm_worker_soa w = empty.soa();
    check(pm_upload_workers(engine_, &w, 0));
    check(pm_enable_group_events(engine_, 1));  // the webhook feed
    const uint64_t no_uid = 0;
    pm_task_soa t{};
    t.n = 0;
    t.uid = &no_uid;  // (non-null: "the tasks have ids")
    check(pm_upload_tasks(engine_, &t));
  } catch (...) {
    pm_engine_destroy(engine_);
    engine_ = nullptr;
    throw;
  }
  if (pools_in_process().fetch_add(1) == 1)  // (a SECOND live pool in the process: one pool is indifferent to the value)
    if (const char* q = getenv("GPU_MAX_HW_QUEUES"); !q || atoi(q) < 8)
      fprintf(stderr, "GpuMatchPlugin: a second pool in this process and GPU_MAX_HW_QUEUES is %s; set it to 16 in the launcher "
                      "before the first HIP call\n", q ? q : "unset (runtime default 4)");
}

GpuMatchPlugin::~GpuMatchPlugin() {
  pools_in_process().fetch_sub(1);
  if (engine_) pm_engine_destroy(engine_);
}

// NodeGroupConfiguration + its requirement string -> pm_config_row / pm_gpu_alt_row (set_configs of the Rust shim;
// the field-by-field projection there is pm_host_parse_requirements here, the same product parser the tests pin to
// the reference's vectors)
void GpuMatchPlugin::set_configs(const std::vector<NodeGroupConfiguration>& templates) {
  std::vector<pm_config_row> rows;
  std::vector<pm_gpu_alt_row> alts;
  for (const NodeGroupConfiguration& t : templates) {
    pm_config_row row{};
    if (t.compute_requirements) {
      std::vector<pm_gpu_alt_row> a(64);
      std::string models(t.compute_requirements->size() + 64 + 1, '\0');
      int32_t rc = pm_host_parse_requirements(t.compute_requirements->c_str(), &row, a.data(), uint32_t(a.size()),
                                              models.data(), models.size());
      if (rc == PM_ERANGE) {  // more alternatives than the first guess
        a.resize(std::max<size_t>(row.alt_count, 1024));
        rc = pm_host_parse_requirements(t.compute_requirements->c_str(), &row, a.data(), uint32_t(a.size()), models.data(),
                                        models.size());
      }
      if (rc != PM_OK)
        throw std::invalid_argument("compute_requirements of '" + t.name + "': " + (pm_last_error() ? pm_last_error() : ""));
      row.flags |= PM_R_HAS_REQ;
      row.alt_begin = uint32_t(alts.size());
      for (uint32_t k = 0; k < row.alt_count; ++k) {
        pm_gpu_alt_row g = a[k];
        if (g.flags & PM_G_MODEL) {  // (model_row comes back as a byte offset into `models`)
          const std::string m(models.c_str() + g.model_row);
          g.model_row = uint32_t(req_models_.size());
          req_models_.push_back(m);
        }
        alts.push_back(g);
      }
    }
    row.min_group_size = uint32_t(t.min_group_size);
    row.max_group_size = uint32_t(t.max_group_size);
    rows.push_back(row);
  }
  const int32_t rc = pm_set_configs(engine_, rows.data(), uint32_t(rows.size()), alts.data(), uint32_t(alts.size()));
  if (rc == PM_EINVAL) throw std::invalid_argument("Plugin configuration is invalid");  // mod.rs:145-147
  check(rc);
  config_rows_ = rows;
  push_model_table(NodeTable{});
}

// The substring rule of GpuSpecs::meets (shared/models/node.rs:463-484), evaluated once per (requirement model,
// interned spec model) pair by the library; re-sent whenever a new spec model shows up.
void GpuMatchPlugin::push_model_table(const NodeTable& nodes) const {
  std::vector<const char*> spec_p, req_p;
  for (const std::string& s : nodes.spec_models) spec_p.push_back(s.c_str());
  for (const std::string& s : req_models_) req_p.push_back(s.c_str());
  const size_t words = (spec_p.size() + 31) / 32;
  std::vector<uint32_t> bits(std::max<size_t>(req_p.size() * words, 1), 0u);
  check(pm_host_build_model_table(req_p.data(), uint32_t(req_p.size()), spec_p.data(), uint32_t(spec_p.size()), bits.data()));
  check(pm_set_model_table(engine_, bits.data(), uint32_t(req_p.size()), uint32_t(spec_p.size())));
}

// Projection of one OrchestratorNode (orchestrator/src/models/node.rs:11-37) into the SoA row.
GpuMatchPlugin::Row GpuMatchPlugin::project(const OrchestratorNode& node, NodeTable& table, bool* new_model) {
  Row r;
  if (node.status == NodeStatus::Healthy) r.flags |= PM_W_HEALTHY;
  if (node.p2p_id) r.flags |= PM_W_HAS_P2P;
  if (node.location) {
    r.flags |= PM_W_HAS_LOC;
    r.lat = node.location->latitude;
    r.lon = node.location->longitude;
  }
  if (node.compute_specs) {
    const ComputeSpecs& s = *node.compute_specs;
    r.flags |= PM_W_HAS_SPECS;
    if (s.gpu) {
      r.flags |= PM_W_HAS_GPU;
      if (s.gpu->count) { r.flags |= PM_W_GPU_COUNT; r.gpu_count = *s.gpu->count; }
      if (s.gpu->memory_mb) { r.flags |= PM_W_GPU_MEM; r.gpu_mem = *s.gpu->memory_mb; }
      if (s.gpu->model) {
        r.flags |= PM_W_GPU_MODEL;
        auto it = table.spec_model_index.find(*s.gpu->model);
        if (it == table.spec_model_index.end()) {
          *new_model = true;
          table.spec_models.push_back(*s.gpu->model);
          it = table.spec_model_index.emplace(*s.gpu->model, uint32_t(table.spec_models.size() - 1)).first;
        }
        r.gpu_class = it->second;
      }
    }
    if (s.cpu) {
      r.flags |= PM_W_HAS_CPU;
      if (s.cpu->cores) { r.flags |= PM_W_CPU_CORES; r.cpu_cores = *s.cpu->cores; }
    }
    if (s.ram_mb) { r.flags |= PM_W_RAM; r.ram = *s.ram_mb; }
    if (s.storage_gb) { r.flags |= PM_W_STORAGE; r.storage = *s.storage_gb; }
  }
  return r;
}

// rank of address.to_string() in byte order (BTreeSet<String>, mod.rs:424-434) among the first `known` rows: one
// pass over the sorted row list
std::vector<uint32_t> GpuMatchPlugin::address_ranks(const std::vector<uint32_t>& by_address, size_t known) {
  std::vector<uint32_t> rank(known, 0u);
  uint32_t r = 0;
  for (uint32_t i : by_address)
    if (i < known) rank[i] = r++;
  return rank;
}

void GpuMatchPlugin::sync_nodes(const std::vector<OrchestratorNode>& snapshot) {
  {
    std::unique_lock<std::shared_mutex> lk(nodes_mu_);
    NodeTable& t = nodes_;
    bool new_model = false;
    std::vector<bool> seen(t.rows.size(), false);
    RowColumns appended, updated;
    std::vector<uint32_t> upd_idx;
    for (const OrchestratorNode& node : snapshot) {
      const Row row = project(node, t, &new_model);
      const auto it = t.index.find(node.address);
      if (it != t.index.end()) {
        const size_t i = it->second;
        if (i < seen.size()) seen[i] = true;   // (an address twice in one snapshot: the second one finds the row just appended)
        t.p2p_ids[i] = node.p2p_id.value_or(std::string());
        if (t.rows[i] != row || !t.present[i]) {
          t.rows[i] = row;
          t.present[i] = true;
          if (i < seen.size()) {
            upd_idx.push_back(uint32_t(i));
            updated.push(row, 0);  // ranks are replaced below
          }
        }
      } else {
        const uint32_t i = uint32_t(t.rows.size());
        t.index.emplace(node.address, i);
        t.addresses.push_back(node.address);
        t.address_strings.push_back(node.address.to_string());
        t.p2p_ids.push_back(node.p2p_id.value_or(std::string()));
        t.rows.push_back(row);
        t.present.push_back(true);
        const std::string& key = t.address_strings[i];
        const auto at = std::partition_point(t.by_address.begin(), t.by_address.end(),
                                             [&](uint32_t j) { return t.address_strings[j] < key; });
        t.by_address.insert(at, i);
        appended.push(row, i);
      }
    }
    // The row map above is the plugin's truth from here on; the engine calls below bring the engine's worker table to
    // it.  If one of them fails the two have diverged (tombstones recorded here and never sent, rows appended here the
    // engine does not have): the next interval then re-sends the whole table instead of deltas.
    if (engine_rows_stale_) {
      // Every row again, in row order — which is the order the engine has its own in, with the rows it never received
      // behind them: pm_upload_workers(keep_groups = 1) keeps the standing groups, their claims and the id stream (rows
      // never move, so a group's row indices are as valid as before).  Then the deaths the engine may have missed: every
      // row that is not in the store, as dead — a no-op for a row in no group, the reference's dissolution (and its
      // send_group_destroyed) for the others.
      RowColumns all;
      const std::vector<uint32_t> ranks = address_ranks(t.by_address, t.rows.size());
      for (size_t i = 0; i < seen.size(); ++i)
        if (!seen[i]) t.present[i] = false;
      std::vector<uint32_t> gone, gone_flags;
      for (size_t i = 0; i < t.rows.size(); ++i) {
        if (!t.present[i]) {
          t.rows[i].flags &= ~uint32_t(PM_W_HEALTHY);
          gone.push_back(uint32_t(i));
          gone_flags.push_back(t.rows[i].flags);
        }
        all.push(t.rows[i], ranks[i]);
      }
      push_model_table(t);
      const pm_worker_soa w = all.soa();
      check(pm_upload_workers(engine_, &w, 1));
      if (!gone.empty()) {
        const std::vector<uint32_t> dead(gone.size(), 1u);
        check(pm_on_worker_status_many(engine_, gone.data(), gone_flags.data(), dead.data(), uint32_t(gone.size())));
      }
      engine_rows_stale_ = false;
      lk.unlock();
      emit_group_webhooks();
      return;
    }
    engine_rows_stale_ = true;  // (cleared behind the last engine call below)
    if (new_model) push_model_table(t);
    // nodes that left the store: tombstone (their group dissolves, like a death; status_update_impl.rs:17-29)
    std::vector<uint32_t> gone, gone_flags;
    for (size_t i = 0; i < seen.size(); ++i)
      if (!seen[i] && t.present[i]) {
        t.present[i] = false;
        t.rows[i].flags &= ~uint32_t(PM_W_HEALTHY);
        gone.push_back(uint32_t(i));
        gone_flags.push_back(t.rows[i].flags);
      }
    if (!gone.empty()) {
      const std::vector<uint32_t> dead(gone.size(), 1u);
      check(pm_on_worker_status_many(engine_, gone.data(), gone_flags.data(), dead.data(), uint32_t(gone.size())));
    }
    if (!upd_idx.empty()) {
      // keep the ranks the engine already has for rewritten rows (ranks among the rows it knows: the new ones of
      // this snapshot are sent behind this call)
      const std::vector<uint32_t> ranks = address_ranks(t.by_address, seen.size());
      for (size_t k = 0; k < upd_idx.size(); ++k) updated.addr_rank[k] = ranks[upd_idx[k]];
      const pm_worker_soa w = updated.soa();
      check(pm_update_workers(engine_, upd_idx.data(), &w));
    }
    if (!appended.flags.empty()) {
      // (a row appended by this very snapshot may have been rewritten by a later entry of it: send what it is now)
      for (size_t k = 0; k < appended.flags.size(); ++k) {
        const Row& r = t.rows[seen.size() + k];
        appended.flags[k] = r.flags;
        appended.gpu_count[k] = r.gpu_count;
        appended.gpu_mem[k] = r.gpu_mem;
        appended.gpu_class[k] = r.gpu_class;
        appended.cpu_cores[k] = r.cpu_cores;
        appended.ram[k] = r.ram;
        appended.storage[k] = r.storage;
        appended.lat[k] = r.lat;
        appended.lon[k] = r.lon;
      }
      uint32_t first = 0;
      const pm_worker_soa w = appended.soa();
      check(pm_append_workers(engine_, &w, &first));
      if (first != seen.size()) throw EngineError(PM_ESTATE, "the engine's worker table and the plugin's row map disagree");
      // a new address shifts the global ranks of the others: GROUP_INDEX only needs the relative order
      const std::vector<uint32_t> ranks = address_ranks(t.by_address, t.rows.size());
      check(pm_set_addr_ranks(engine_, ranks.data(), uint32_t(ranks.size())));
    }
    engine_rows_stale_ = false;
  }
  emit_group_webhooks();  // tombstoned nodes dissolved their groups
}

// scheduler_impl.rs:44-59: any None on the way => every configuration allowed
uint64_t GpuMatchPlugin::topology_mask(const Task& t) const {
  if (!t.allowed_topologies) return ~0ull;
  uint64_t m = 0;
  for (const std::string& name : *t.allowed_topologies) {
    const auto it = std::find(config_names_.begin(), config_names_.end(), name);
    if (it != config_names_.end()) m |= 1ull << size_t(it - config_names_.begin());
  }
  return m;
}

void GpuMatchPlugin::push_enabled(const std::vector<Task>& tasks) {
  // available_node_group_configs: every topology some task names (on_task_created, mod.rs:1224-1243)
  uint64_t enabled = 0;
  for (const Task& t : tasks) {
    const uint64_t m = topology_mask(t);
    if (m != ~0ull) enabled |= m;
  }
  check(pm_set_enabled_mask(engine_, enabled));
  enabled_mask_.store(enabled);
}

// The engine reports a task as a POSITION in `tasks_`, and it re-derives the published positions inside
// pm_tasks_insert_front / pm_tasks_delete / pm_upload_tasks — so the list and the engine's table change under ONE
// write lock, and filter_tasks holds the read lock from the look-up to the index: a heartbeat never pairs a position
// of the new table with the old list (or the other way round).

void GpuMatchPlugin::sync_tasks_locked(std::vector<Task>& guard, std::vector<Task> tasks) {
  std::vector<uint64_t> masks, uid;
  std::vector<int64_t> created;
  for (const Task& t : tasks) {
    masks.push_back(topology_mask(t));
    created.push_back(t.created_at);
    uid.push_back(task_uid(t));
  }
  const uint64_t no_uid = 0;
  pm_task_soa soa{};
  soa.n = uint32_t(tasks.size());
  soa.topo_mask = masks.data();
  soa.created_at = created.data();
  soa.uid = tasks.empty() ? &no_uid : uid.data();
  check(pm_upload_tasks(engine_, &soa));
  push_enabled(tasks);
  guard = std::move(tasks);
}

void GpuMatchPlugin::sync_tasks(std::vector<Task> tasks) {
  std::unique_lock<std::shared_mutex> guard(tasks_mu_);
  sync_tasks_locked(tasks_, std::move(tasks));
}

void GpuMatchPlugin::on_task_created(const Task& task, const std::function<std::vector<Task>()>& all_tasks) {
  const uint64_t mask = topology_mask(task), uid = task_uid(task);
  const int64_t created = task.created_at;
  pm_task_soa soa{};
  soa.n = 1;
  soa.topo_mask = &mask;
  soa.created_at = &created;
  soa.uid = &uid;
#ifndef PM_PLUGIN_TEST_ROUND3_LOCK_ORDER
  std::unique_lock<std::shared_mutex> guard(tasks_mu_);  // (before the engine call: see LOCK ORDER)
  const int32_t rc = republish_on_insert.load() ? pm_tasks_insert_front_ex(engine_, &soa, 1) : pm_tasks_insert_front(engine_, &soa);
#else  // the order the round-3 review found in the Rust shim (engine first, lock second): built only by the test that
       // shows tests/cpp/plugin_test.cpp sees it
  const int32_t rc = republish_on_insert.load() ? pm_tasks_insert_front_ex(engine_, &soa, 1) : pm_tasks_insert_front(engine_, &soa);
  std::unique_lock<std::shared_mutex> guard(tasks_mu_);
#endif
  if (rc != PM_OK) {  // equal or older timestamps: the snapshot
    sync_tasks_locked(tasks_, all_tasks());
    return;
  }
  tasks_.insert(tasks_.begin(), task);
  push_enabled(tasks_);
}

void GpuMatchPlugin::on_task_deleted(const Task& task) {
  const uint64_t uid = task_uid(task);
  uint32_t n = 0;
  {
    std::unique_lock<std::shared_mutex> guard(tasks_mu_);  // (before the engine call: see LOCK ORDER)
    check(pm_tasks_delete(engine_, &uid, 1, &n));
    tasks_.erase(std::remove_if(tasks_.begin(), tasks_.end(), [&](const Task& t) { return t.id == task.id; }), tasks_.end());
    push_enabled(tasks_);
  }
  emit_group_webhooks();  // dissolve_group's send_group_destroyed, mod.rs:1469-1481
}

pm_stats GpuMatchPlugin::tick() {
  pm_stats s{};
  check(pm_tick(engine_, &s));
  emit_group_webhooks();
  return s;
}

std::vector<pm_stats> GpuMatchPlugin::tick_many(const std::vector<GpuMatchPlugin*>& pools) {
  std::vector<pm_engine*> engines;
  for (GpuMatchPlugin* p : pools) engines.push_back(p->engine_);
  std::vector<pm_stats> stats(pools.size());
  if (pools.empty()) return stats;
  pools[0]->check(pm_tick_many(engines.data(), uint32_t(engines.size()), stats.data(), 0));
  for (GpuMatchPlugin* p : pools) p->emit_group_webhooks();  // each pool drains its own life-cycle feed
  return stats;
}

uint32_t shard_of(const Address& a, uint32_t world) {
  // the low 8 bytes of the 20-byte address = the last sixteen hex digits of its text, big-endian
  uint64_t v = 0;
  int digits = 0;
  for (size_t i = a.text.size(); i-- > 0 && digits < 16;) {
    const char c = a.text[i];
    uint64_t d;
    if (c >= '0' && c <= '9') d = uint64_t(c - '0');
    else if (c >= 'a' && c <= 'f') d = uint64_t(c - 'a' + 10);
    else if (c >= 'A' && c <= 'F') d = uint64_t(c - 'A' + 10);
    else break;  // 'x'
    v |= d << (4 * digits);
    ++digits;
  }
  uint64_t z = v + 0x9E3779B97F4A7C15ull;  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return world ? uint32_t(z % world) : 0u;
}

pm_stats GpuMatchPlugin::tick_dist(AllGather& comm) {
  const uint32_t rank = comm.rank(), world = comm.world();
  if (world == 0 || rank >= world) throw std::invalid_argument("tick_dist: rank outside the communicator");
  {
    std::shared_lock<std::shared_mutex> lk(nodes_mu_);
    if (dist_stream_ != comm.stream()) {
      check(pm_set_stream(engine_, comm.stream()));  // the engine's kernels and the collective on one stream: no host wait between them
      dist_stream_ = comm.stream();
    }
    if (rank != dist_rank_.load() || world != dist_world_ || nodes_.rows.size() != dist_rows_) {
      // ownership is per row and every rank computes the same: a hash of the address, nothing is negotiated
      std::vector<uint8_t> shard(nodes_.rows.size());
      for (size_t i = 0; i < shard.size(); ++i) shard[i] = uint8_t(shard_of(nodes_.addresses[i], world));
      check(pm_dist_configure(engine_, rank, world, shard.empty() ? nullptr : shard.data()));
      dist_rank_.store(rank), dist_world_ = world;
      dist_rows_ = engine_rows_stale_ ? size_t(-1) : nodes_.rows.size();  // (the engine is behind the row map: configure again next time)
    }
  }
  pm_stats s{};
  pm_dist_xfer x{};
  check(pm_dist_tick_begin(engine_));   // compat sweep; the whole carve is started (replicated: every rank runs it)
  check(pm_dist_carve_wait(engine_));   // waits for it; near-ties settled on the host, identically everywhere
  check(pm_dist_match_begin(engine_, &x));  // solo merge, pair sweep + claim of the OWNED workers
  if (x.bytes_per_rank)                 // the ONE exchange of a tick: the published rows
    comm.all_gather(reinterpret_cast<const void*>(uintptr_t(x.send_ptr)), reinterpret_cast<void*>(uintptr_t(x.recv_ptr)), size_t(x.bytes_per_rank));
  check(pm_dist_tick_end(engine_, &s));  // scatter into the full table, publish
  emit_group_webhooks();
  return s;
}

// Drains the engine's group life-cycle feed into send_group_created / send_group_destroyed, in the order the
// reference emits them.  Runs after everything that can create or dissolve groups.
void GpuMatchPlugin::emit_group_webhooks() {
  uint32_t ne = 0, nm = 0;
  int32_t rc = pm_drain_group_events(engine_, nullptr, 0, nullptr, 0, &ne, &nm);
  if (rc == PM_OK) return;  // empty log
  // (another thread may log events between the size query and the drain: grow and try again; a drain that still
  // fails is left for the next call — it is not the tick that failed)
  std::vector<pm_group_event> events;
  std::vector<uint32_t> members;
  bool drained = false;
  for (int attempt = 0; attempt < 8 && !drained; ++attempt) {
    events.resize(ne);
    members.resize(nm);
    const uint32_t cap_e = ne, cap_m = nm;
    rc = pm_drain_group_events(engine_, events.data(), cap_e, members.data(), cap_m, &ne, &nm);
    if (rc == PM_OK) drained = true;
    else if (rc != PM_ERANGE) return;
  }
  if (!drained) return;
  {  // NodeGroup.created_at (mod.rs:575: Utc::now() when the group is formed) — the clock at the report of the creation
    const int64_t now = clock ? clock() : 0;
    std::lock_guard<std::mutex> lk(group_meta_mu_);
    for (uint32_t k = 0; k < ne; ++k) {
      if (events[k].kind == PM_GROUP_CREATED) group_created_at_[events[k].group_id] = now;
      else group_created_at_.erase(events[k].group_id);
    }
  }
  if (webhook_plugins_.empty() || dist_rank_.load() != 0) return;  // (a rank > 0 of a multi-GPU pool: rank 0 reports)
  // (the address strings are copied under the lock and the deliveries made without it: a slow webhook endpoint must not
  // hold handle_status_change and sync_nodes up)
  std::vector<std::vector<std::string>> nodes_of(ne);
  {
    std::shared_lock<std::shared_mutex> lk(nodes_mu_);
    for (uint32_t k = 0; k < ne; ++k)
      for (uint32_t j = 0; j < events[k].n_members; ++j)
        nodes_of[k].push_back(nodes_.address_strings[members[events[k].member_begin + j]]);  // group.nodes order
  }
  for (uint32_t k = 0; k < ne; ++k) {
    const pm_group_event& ev = events[k];
    const std::string id = hex_lower(ev.group_id);
    const std::string& name = config_names_[ev.config];
    const std::vector<std::string>& nodes = nodes_of[k];
    for (const auto& p : webhook_plugins_) {
      try {
        if (ev.kind == PM_GROUP_CREATED) p->send_group_created(id, name, nodes);
        else p->send_group_destroyed(id, name, nodes);
      } catch (const std::exception& e) {  // as in the reference: logged, not fatal
        std::fprintf(stderr, "Failed to send group webhook: %s\n", e.what());
      }
    }
  }
}

// SchedulerPlugin::filter_tasks (plugins/mod.rs:66-78): lock-free look-up + the `${...}` templating the reference does
// at scheduler_impl.rs:112-205 (GROUP_INDEX, GROUP_SIZE, NEXT_P2P_ADDRESS, GROUP_ID, upload count).
std::vector<Task> GpuMatchPlugin::filter_tasks(const std::vector<Task>&, const Address& node_address) {
  std::shared_lock<std::shared_mutex> nodes(nodes_mu_);
  const auto it = nodes_.index.find(node_address);
  if (it == nodes_.index.end()) return {};
  const uint32_t w = it->second;
  pm_assignment a{};
  Task task;
  {
    std::shared_lock<std::shared_mutex> tasks(tasks_mu_);  // held from the look-up to the index (see LOCK ORDER)
    if (pm_lookup_task_for_worker(engine_, w, &a) != PM_OK || a.task == PM_NONE) return {};
    if (a.task >= tasks_.size()) return {};
    task = tasks_[a.task];
  }
  const std::string group_id = hex_lower(a.group_id);
  const std::string next = a.next_worker < nodes_.p2p_ids.size() ? nodes_.p2p_ids[a.next_worker] : std::string();
  const std::string count = std::to_string(upload_counter_ ? upload_counter_(node_address, group_id) : 0);
  pm_group_vars vars{};
  vars.group_index = a.group_index;
  vars.group_size = a.group_size;
  vars.next_p2p_address = next.c_str();
  vars.group_id = group_id.c_str();
  vars.total_upload_count = count.c_str();
  const auto chk = [this](int32_t rc) { check(rc); };
  const auto group_vars = [&](const std::string& s) {
    return render([&](char* o, size_t c, size_t* n) { return pm_host_group_vars(s.c_str(), &vars, o, c, n); }, chk);
  };
  if (!task.env_vars) task.env_vars.emplace();
  (*task.env_vars)["GROUP_INDEX"] = std::to_string(a.group_index);  // scheduler_impl.rs:161
  for (auto& kv : *task.env_vars) kv.second = group_vars(kv.second);
  if (task.cmd)
    for (std::string& arg : *task.cmd) arg = group_vars(arg);
  if (task.volume_mounts)  // scheduler_impl.rs:185-200
    for (VolumeMount& m : *task.volume_mounts)
      for (std::string* path : {&m.host_path, &m.container_path})
        *path = render([&](char* o, size_t c, size_t* n) { return pm_host_volume_vars(path->c_str(), group_id.c_str(), o, c, n); }, chk);
  return {task};
}

// ------------------------------------------------------------------------------------------------ the read surface
// What the API routes call on AppState.node_groups_plugin (node_groups/mod.rs:324-434, :1002-1065).  The reference reads
// Redis; here the engine's host-side group list is the store (pm_get_groups / pm_get_group_by_id / pm_get_group_of_worker /
// pm_dissolve_group_by_id) and the plugin's node table turns row indices back into address strings.

GpuMatchPlugin::GroupSnapshot GpuMatchPlugin::snapshot_groups(bool want_group_of) const {
  GroupSnapshot snap;
  for (int attempt = 0; attempt < 8; ++attempt) {  // (a tick between the size query and the copy: ask again)
    uint32_t ng = 0, nm = 0;
    check(pm_get_groups(engine_, nullptr, nullptr, 0, &ng, nullptr, 0, &nm));
    snap.groups.resize(ng);
    snap.members.resize(nm);
    // (group_of_worker gets W entries, W the ENGINE's row count: never more than the plugin's, whose lock the caller holds)
    if (want_group_of) snap.group_of.assign(nodes_.rows.size(), -1);
    const int32_t rc = pm_get_groups(engine_, want_group_of && !snap.group_of.empty() ? snap.group_of.data() : nullptr,
                                     ng ? snap.groups.data() : nullptr, ng, &ng, nm ? snap.members.data() : nullptr, nm, &nm);
    if (rc == PM_OK) {
      snap.groups.resize(ng);
      snap.members.resize(nm);
      return snap;
    }
    if (rc != PM_ERANGE) check(rc);
  }
  throw EngineError(PM_ESTATE, "the group list kept changing under pm_get_groups");
}

NodeGroup GpuMatchPlugin::make_group(const NodeTable& t, const pm_group& g, const uint32_t* members) const {
  NodeGroup out;
  out.id = hex_lower(g.id);
  for (uint32_t j = 0; j < g.n_members; ++j) out.nodes.push_back(t.address_strings[members[j]]);  // BTreeSet order already
  out.configuration_name = config_names_[g.config];
  {
    std::lock_guard<std::mutex> lk(group_meta_mu_);
    const auto it = group_created_at_.find(g.id);
    out.created_at = it != group_created_at_.end() ? it->second : (clock ? clock() : 0);  // (formed, creation not reported yet)
  }
  return out;
}

// the reference keys node_to_group by address TEXT: exact string match (binary search in the address-ordered row list)
std::optional<uint32_t> GpuMatchPlugin::row_of_address_text(const NodeTable& t, const std::string& text) const {
  const auto at = std::partition_point(t.by_address.begin(), t.by_address.end(), [&](uint32_t j) { return t.address_strings[j] < text; });
  if (at == t.by_address.end() || t.address_strings[*at] != text) return std::nullopt;
  return *at;
}

std::vector<NodeGroup> GpuMatchPlugin::get_all_groups() const {
  std::shared_lock<std::shared_mutex> lk(nodes_mu_);
  const GroupSnapshot snap = snapshot_groups(false);
  std::vector<NodeGroup> out;
  out.reserve(snap.groups.size());
  for (const pm_group& g : snap.groups) out.push_back(make_group(nodes_, g, snap.members.data() + g.member_begin));
  std::sort(out.begin(), out.end(), [](const NodeGroup& a, const NodeGroup& b) { return a.id < b.id; });  // mod.rs:1040
  return out;
}

std::optional<NodeGroup> GpuMatchPlugin::get_group_by_id(const std::string& group_id) const {
  uint64_t id = 0;
  if (!parse_group_id(group_id, &id)) return std::nullopt;
  std::shared_lock<std::shared_mutex> lk(nodes_mu_);
  pm_group g{};
  uint32_t slot = PM_NONE;
  std::vector<uint32_t> members(64);
  int32_t rc = pm_get_group_by_id(engine_, id, &g, members.data(), uint32_t(members.size()), &slot);
  if (rc == PM_ERANGE && slot != PM_NONE) {  // (a group of more than 64 nodes)
    members.resize(g.n_members);
    rc = pm_get_group_by_id(engine_, id, &g, members.data(), uint32_t(members.size()), &slot);
  }
  check(rc);
  if (slot == PM_NONE) return std::nullopt;
  return make_group(nodes_, g, members.data());
}

std::unordered_map<std::string, std::string> GpuMatchPlugin::get_all_node_group_mappings() const {
  std::shared_lock<std::shared_mutex> lk(nodes_mu_);
  const GroupSnapshot snap = snapshot_groups(false);
  std::unordered_map<std::string, std::string> out;
  for (const pm_group& g : snap.groups) {
    const std::string id = hex_lower(g.id);
    for (uint32_t j = 0; j < g.n_members; ++j) out.emplace(nodes_.address_strings[snap.members[g.member_begin + j]], id);
  }
  return out;
}

std::optional<NodeGroup> GpuMatchPlugin::get_node_group(const std::string& node_addr) const {
  std::shared_lock<std::shared_mutex> lk(nodes_mu_);
  const std::optional<uint32_t> row = row_of_address_text(nodes_, node_addr);
  if (!row) return std::nullopt;
  pm_group g{};
  uint32_t slot = PM_NONE;
  std::vector<uint32_t> members(64);
  int32_t rc = pm_get_group_of_worker(engine_, *row, &g, members.data(), uint32_t(members.size()), &slot);
  if (rc == PM_ERANGE && slot != PM_NONE) {
    members.resize(g.n_members);
    rc = pm_get_group_of_worker(engine_, *row, &g, members.data(), uint32_t(members.size()), &slot);
  }
  if (rc == PM_ERANGE && slot == PM_NONE) return std::nullopt;  // (a row the engine has not been sent yet: in no group)
  check(rc);
  if (slot == PM_NONE) return std::nullopt;
  return make_group(nodes_, g, members.data());
}

std::unordered_map<std::string, std::optional<NodeGroup>> GpuMatchPlugin::get_node_groups_batch(
    const std::vector<std::string>& node_addresses) const {
  std::unordered_map<std::string, std::optional<NodeGroup>> out;
  if (node_addresses.empty()) return out;  // mod.rs:346-348
  std::shared_lock<std::shared_mutex> lk(nodes_mu_);
  const GroupSnapshot snap = snapshot_groups(true);
  std::unordered_map<int32_t, NodeGroup> made;  // (every group is built once, like the reference's MGET of the unique ids)
  for (const std::string& a : node_addresses) {
    std::optional<NodeGroup> group;
    const std::optional<uint32_t> row = row_of_address_text(nodes_, a);
    if (row && *row < snap.group_of.size() && snap.group_of[*row] >= 0) {
      const int32_t gi = snap.group_of[*row];
      auto it = made.find(gi);
      if (it == made.end()) it = made.emplace(gi, make_group(nodes_, snap.groups[size_t(gi)], snap.members.data() + snap.groups[size_t(gi)].member_begin)).first;
      group = it->second;
    }
    out[a] = group;
  }
  return out;
}

size_t GpuMatchPlugin::get_idx_in_group(const NodeGroup& node_group, const std::string& node_addr) const {
  const auto it = std::find(node_group.nodes.begin(), node_group.nodes.end(), node_addr);
  if (it == node_group.nodes.end()) throw std::out_of_range("Node " + node_addr + " not found in group");
  return size_t(it - node_group.nodes.begin());
}

std::vector<NodeGroupConfiguration> GpuMatchPlugin::get_available_configurations() const {
  std::vector<uint32_t> order(config_rows_.size() + 1);
  uint32_t n = 0;
  check(pm_host_config_order(config_rows_.data(), uint32_t(config_rows_.size()), enabled_mask_.load(), order.data(), &n));
  std::vector<NodeGroupConfiguration> out;
  for (uint32_t k = 0; k < n; ++k) out.push_back(templates_[order[k]]);
  return out;
}

std::vector<NodeGroupConfiguration> GpuMatchPlugin::get_all_configuration_templates() const {
  // every template, in the order the constructor's sort leaves them in (mod.rs:150-164): the carve order with nothing disabled
  const size_t C = config_rows_.size();
  std::vector<uint32_t> order(C + 1);
  uint32_t n = 0;
  check(pm_host_config_order(config_rows_.data(), uint32_t(C), C >= 64 ? ~0ull : ((1ull << C) - 1ull), order.data(), &n));
  std::vector<NodeGroupConfiguration> out;
  for (uint32_t k = 0; k < n; ++k) out.push_back(templates_[order[k]]);
  return out;
}

void GpuMatchPlugin::dissolve_group(const std::string& group_id) {
  uint64_t id = 0;
  if (!parse_group_id(group_id, &id)) return;  // "No group found with ID" (mod.rs:1483-1484): Ok(())
  uint32_t dissolved = 0;
  check(pm_dissolve_group_by_id(engine_, id, &dissolved));
  if (dissolved) emit_group_webhooks();  // send_group_destroyed, mod.rs:1469-1481
}

std::vector<std::string> get_task_topologies(const Task& task) {
  return task.allowed_topologies ? *task.allowed_topologies : std::vector<std::string>{};
}

std::string upload_file_name(const GpuMatchPlugin& plugin, const std::string& file_name_in, const std::string& address,
                             const std::function<uint64_t(const std::string&, const std::string&)>& count_uploads,
                             std::string* group_id_out) {
  // storage.rs:147-166: the group variables only when the node is in a group ...
  const std::optional<NodeGroup> group = plugin.get_node_group(address);
  size_t idx = 0;
  if (group) idx = plugin.get_idx_in_group(*group, address);
  if (group_id_out) *group_id_out = group ? group->id : std::string();
  // ... :170-199: the count of this node's uploads under the group's key (or "no-group"), :202-207 the two count variables
  const uint64_t count = count_uploads ? count_uploads(address, group ? group->id : std::string("no-group")) : 0;
  const auto chk = [](int32_t rc) {
    if (rc != PM_OK) throw EngineError(rc, std::string("pm_host_upload_name_vars: ") + (pm_last_error() ? pm_last_error() : ""));
  };
  return render(
      [&](char* o, size_t c, size_t* n) {
        return pm_host_upload_name_vars(file_name_in.c_str(), group ? group->id.c_str() : nullptr, group ? uint32_t(group->nodes.size()) : 0u,
                                        uint32_t(idx), count, o, c, n);
      },
      chk);
}

void GpuMatchPlugin::handle_status_change(const OrchestratorNode& node) {
  uint32_t dead = 0;
  {
    std::unique_lock<std::shared_mutex> lk(nodes_mu_);
    const auto it = nodes_.index.find(node.address);
    if (it == nodes_.index.end()) return;
    const uint32_t w = it->second;
    uint32_t flags = nodes_.rows[w].flags & ~uint32_t(PM_W_HEALTHY);
    if (node.status == NodeStatus::Healthy) flags |= PM_W_HEALTHY;
    nodes_.rows[w].flags = flags;
    dead = (node.status == NodeStatus::Dead || node.status == NodeStatus::LowBalance) ? 1u : 0u;
    check(pm_on_worker_status(engine_, w, flags, dead));
  }
  if (dead) emit_group_webhooks();  // the whole group was dissolved (status_update_impl.rs:17-29)
}

size_t GpuMatchPlugin::known_nodes() const {
  std::shared_lock<std::shared_mutex> lk(nodes_mu_);
  return nodes_.rows.size();
}

std::optional<uint32_t> GpuMatchPlugin::row_of(const Address& a) const {
  std::shared_lock<std::shared_mutex> lk(nodes_mu_);
  const auto it = nodes_.index.find(a);
  if (it == nodes_.index.end()) return std::nullopt;
  return it->second;
}

// ------------------------------------------------------------------------------------------------ the scheduler

Scheduler::Scheduler(std::shared_ptr<TaskStore> task_store, std::vector<std::shared_ptr<SchedulerPlugin>> plugins)
    : task_store_(std::move(task_store)), plugins_(std::move(plugins)) {
  if (plugins_.empty()) plugins_.push_back(std::make_shared<NewestTaskPlugin>());
}

std::optional<Task> Scheduler::get_task_for_node(const Address& node_address, int64_t now) {
  // INTEGRATION.md "The task list per heartbeat": LRANGE + T x GET + T x serde per heartbeat (task_store.rs:57-82) is
  // what the reference pays here; a chain headed by the engine's plugin is served from the plugin's own list
  std::vector<Task> all_tasks;
  if (!plugins_.front()->serves_from_own_task_list()) all_tasks = task_store_->get_all_tasks();
  for (const auto& plugin : plugins_) all_tasks = plugin->filter_tasks(all_tasks, node_address);
  if (all_tasks.empty()) return std::nullopt;
  Task task = all_tasks[0];
  const std::string addr = node_address.to_string();
  const auto vars = [&](const std::string& s) { return replace_all(replace_all(s, "${TASK_ID}", task.id), "${NODE_ADDRESS}", addr); };
  if (task.env_vars)
    for (auto& kv : *task.env_vars) kv.second = vars(kv.second);
  if (task.cmd)
    for (std::string& arg : *task.cmd) arg = vars(arg);
  if (task.volume_mounts)
    for (VolumeMount& m : *task.volume_mounts) m = m.replace_labels(task.id, addr, now);
  return task;
}

}  // namespace orchestrator
