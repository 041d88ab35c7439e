// pm_plugin_c.cpp — see pm_plugin_c.h: a test driver's handle on GpuMatchPlugin + Scheduler.
#include "pm_plugin_c.h"

#include <algorithm>
#include <mutex>
#include <string>

#include "gpu_match_plugin.hpp"
#include "pm_plugin_c_internal.hpp"

using namespace orchestrator;
using pmx_detail::ListStore;
using pmx_detail::RecordingWebhook;

namespace {

thread_local std::string g_error;

std::string esc(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '\\') o += "\\\\";
    else if (c == '\t') o += "\\t";
    else if (c == '\n') o += "\\n";
    else o += c;
  }
  return o;
}

int32_t give(const std::string& text, char* out, size_t cap, size_t* needed) {
  if (!needed) {
    g_error = "null argument";
    return -1;
  }
  *needed = text.size() + 1;
  if (!out || cap < text.size() + 1) return out || cap ? -2 : 0;  // (NULL / 0 = the size query)
  std::copy(text.begin(), text.end(), out);
  out[text.size()] = '\0';
  return 0;
}

Task to_task(const pmx_task& t) {
  Task r;
  r.id = t.id ? t.id : "";
  r.name = t.name ? t.name : "";
  r.image = "image";
  r.created_at = t.created_at;
  if (t.n_topologies >= 0) {
    r.allowed_topologies.emplace();
    for (int32_t i = 0; i < t.n_topologies; ++i) r.allowed_topologies->push_back(t.topologies[i]);
  }
  if (t.n_env) {
    r.env_vars.emplace();
    for (uint32_t i = 0; i < t.n_env; ++i) (*r.env_vars)[t.env_keys[i]] = t.env_values[i];
  }
  if (t.n_cmd >= 0) {
    r.cmd.emplace();
    for (int32_t i = 0; i < t.n_cmd; ++i) r.cmd->push_back(t.cmd[i]);
  }
  if (t.n_mounts >= 0) {
    r.volume_mounts.emplace();
    for (int32_t i = 0; i < t.n_mounts; ++i) r.volume_mounts->push_back(VolumeMount{t.mount_host[i], t.mount_container[i]});
  }
  return r;
}

OrchestratorNode to_node(const pmx_node& n) {
  OrchestratorNode r;
  r.address = Address(n.address ? n.address : "");
  r.status = NodeStatus(n.status);
  if (n.has & PM_W_HAS_P2P) r.p2p_id = std::string(n.p2p_id ? n.p2p_id : "");
  if (n.has & PM_W_HAS_LOC) r.location = NodeLocation{n.latitude, n.longitude};
  if (n.has & PM_W_HAS_SPECS) {
    ComputeSpecs s;
    if (n.has & PM_W_HAS_GPU) {
      GpuSpecs g;
      if (n.has & PM_W_GPU_COUNT) g.count = n.gpu_count;
      if (n.has & PM_W_GPU_MEM) g.memory_mb = n.gpu_memory_mb;
      if (n.has & PM_W_GPU_MODEL) g.model = std::string(n.gpu_model ? n.gpu_model : "");
      s.gpu = g;
    }
    if (n.has & PM_W_HAS_CPU) {
      CpuSpecs c;
      if (n.has & PM_W_CPU_CORES) c.cores = n.cpu_cores;
      s.cpu = c;
    }
    if (n.has & PM_W_RAM) s.ram_mb = n.ram_mb;
    if (n.has & PM_W_STORAGE) s.storage_gb = n.storage_gb;
    r.compute_specs = s;
  }
  return r;
}

std::string group_line(const NodeGroup& g) {
  std::string s = esc(g.id) + "\t" + esc(g.configuration_name) + "\t" + std::to_string(g.created_at);
  for (const std::string& n : g.nodes) s += "\t" + esc(n);
  return s;
}

}  // namespace

#define PMX_TRY(...)                  \
  try {                               \
    __VA_ARGS__;                      \
    return 0;                         \
  } catch (const std::exception& e) { \
    g_error = e.what();               \
    return -1;                        \
  }

extern "C" {

const char* pmx_last_error(void) { return g_error.c_str(); }

int32_t pmx_create(const pmx_config* cfgs, uint32_t n, int32_t device, pmx_plugin** out) {
  if (!out || (n && !cfgs)) {
    g_error = "null argument";
    return -1;
  }
  PMX_TRY({
    std::vector<NodeGroupConfiguration> t;
    for (uint32_t i = 0; i < n; ++i) {
      NodeGroupConfiguration c{cfgs[i].name, cfgs[i].min_group_size, cfgs[i].max_group_size, std::nullopt};
      if (cfgs[i].compute_requirements) c.compute_requirements = std::string(cfgs[i].compute_requirements);
      t.push_back(c);
    }
    std::unique_ptr<pmx_plugin> p(new pmx_plugin());
    pmx_plugin* raw = p.get();
    p->plugin = std::make_shared<GpuMatchPlugin>(
        std::move(t), device, [raw](const Address&, const std::string&) { return size_t(raw->upload_count.load()); },
        std::vector<std::shared_ptr<WebhookPlugin>>{p->hook});
    p->scheduler.reset(new Scheduler(p->store, {p->plugin}));
    p->plugin->clock = [raw] { return raw->now_ms.load(); };
    *out = p.release();
  })
}

void pmx_destroy(pmx_plugin* p) { delete p; }
pm_engine* pmx_engine(pmx_plugin* p) { return p ? p->plugin->engine_ptr() : nullptr; }
void pmx_set_upload_count(pmx_plugin* p, uint64_t n) { p->upload_count = n; }
void pmx_set_republish_on_insert(pmx_plugin* p, uint32_t on) { p->plugin->republish_on_insert = on != 0; }

int32_t pmx_sync_nodes(pmx_plugin* p, const pmx_node* nodes, uint32_t n) {
  PMX_TRY({
    std::vector<OrchestratorNode> snap;
    for (uint32_t i = 0; i < n; ++i) snap.push_back(to_node(nodes[i]));
    p->plugin->sync_nodes(snap);
  })
}

int32_t pmx_sync_tasks(pmx_plugin* p, const pmx_task* tasks, uint32_t n) {
  PMX_TRY({
    std::vector<Task> list;
    for (uint32_t i = 0; i < n; ++i) list.push_back(to_task(tasks[i]));
    {
      std::lock_guard<std::mutex> lk(p->store->mu);
      p->store->tasks = list;
    }
    p->plugin->sync_tasks(std::move(list));
  })
}

int32_t pmx_on_task_created(pmx_plugin* p, const pmx_task* task) {
  PMX_TRY({
    const Task t = to_task(*task);
    {
      std::lock_guard<std::mutex> lk(p->store->mu);  // get_all_tasks order: created_at descending (task_store.rs:79), stable
      auto at = std::find_if(p->store->tasks.begin(), p->store->tasks.end(), [&](const Task& o) { return o.created_at < t.created_at; });
      p->store->tasks.insert(at, t);
    }
    p->plugin->on_task_created(t, [&] { return p->store->snapshot(); });
  })
}

int32_t pmx_on_task_deleted(pmx_plugin* p, const char* task_id) {
  PMX_TRY({
    Task gone;
    {
      std::lock_guard<std::mutex> lk(p->store->mu);
      auto it = std::find_if(p->store->tasks.begin(), p->store->tasks.end(), [&](const Task& o) { return o.id == task_id; });
      if (it == p->store->tasks.end()) throw std::invalid_argument("no such task in the store");
      gone = *it;
      p->store->tasks.erase(it);
    }
    p->plugin->on_task_deleted(gone);
  })
}

int32_t pmx_handle_status_change(pmx_plugin* p, const char* address, uint32_t status) {
  PMX_TRY({
    OrchestratorNode n;
    n.address = Address(address);
    n.status = NodeStatus(status);
    p->plugin->handle_status_change(n);
  })
}

int32_t pmx_tick(pmx_plugin* p, pm_stats* stats) {
  PMX_TRY({
    const pm_stats s = p->plugin->tick();
    if (stats) *stats = s;
  })
}

int32_t pmx_row_of(pmx_plugin* p, const char* address, uint32_t* row) {
  const std::optional<uint32_t> r = p->plugin->row_of(Address(address));
  if (!r) {
    g_error = "unknown address";
    return -1;
  }
  if (row) *row = *r;
  return 0;
}
uint32_t pmx_known_nodes(pmx_plugin* p) { return uint32_t(p->plugin->known_nodes()); }
uint32_t pmx_store_loads(pmx_plugin* p) {
  std::lock_guard<std::mutex> lk(p->store->mu);
  return p->store->loads;
}

int32_t pmx_get_task_for_node(pmx_plugin* p, const char* address, int64_t now, char* out, size_t cap, size_t* needed) {
  try {
    const std::optional<Task> t = p->scheduler->get_task_for_node(Address(address), now);
    std::string text;
    if (t) {
      text += "id\t" + esc(t->id) + "\nname\t" + esc(t->name) + "\n";
      if (t->env_vars)
        for (const auto& kv : *t->env_vars) text += "env\t" + esc(kv.first) + "\t" + esc(kv.second) + "\n";
      if (t->cmd)
        for (const std::string& a : *t->cmd) text += "cmd\t" + esc(a) + "\n";
      if (t->volume_mounts)
        for (const VolumeMount& m : *t->volume_mounts) text += "mount\t" + esc(m.host_path) + "\t" + esc(m.container_path) + "\n";
    }
    return give(text, out, cap, needed);
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1;
  }
}

#define PMX_TEXT(...)                 \
  try {                               \
    std::string text;                 \
    __VA_ARGS__;                      \
    return give(text, out, cap, needed); \
  } catch (const std::exception& e) { \
    g_error = e.what();               \
    return -1;                        \
  }

void pmx_set_clock(pmx_plugin* p, int64_t now_ms) { p->now_ms = now_ms; }

int32_t pmx_get_all_groups(pmx_plugin* p, char* out, size_t cap, size_t* needed) {
  PMX_TEXT({
    for (const NodeGroup& g : p->plugin->get_all_groups()) text += group_line(g) + "\n";
  })
}

int32_t pmx_get_group_by_id(pmx_plugin* p, const char* group_id, char* out, size_t cap, size_t* needed) {
  PMX_TEXT({
    const std::optional<NodeGroup> g = p->plugin->get_group_by_id(group_id ? group_id : "");
    if (g) text = group_line(*g) + "\n";
  })
}

int32_t pmx_get_node_group(pmx_plugin* p, const char* address, char* out, size_t cap, size_t* needed) {
  PMX_TEXT({
    const std::string a = address ? address : "";
    const std::optional<NodeGroup> g = p->plugin->get_node_group(a);
    if (g) text = std::to_string(p->plugin->get_idx_in_group(*g, a)) + "\t" + group_line(*g) + "\n";
  })
}

int32_t pmx_get_node_groups_batch(pmx_plugin* p, const char* const* addresses, uint32_t n, char* out, size_t cap, size_t* needed) {
  PMX_TEXT({
    std::vector<std::string> asked;
    for (uint32_t i = 0; i < n; ++i) asked.push_back(addresses[i]);
    const auto got = p->plugin->get_node_groups_batch(asked);
    for (const std::string& a : asked) {
      const auto it = got.find(a);
      if (it == got.end()) throw std::logic_error("an asked address is missing from the batch result");
      text += esc(a) + "\t" + (it->second ? group_line(*it->second) : std::string("-")) + "\n";
    }
  })
}

int32_t pmx_get_all_node_group_mappings(pmx_plugin* p, char* out, size_t cap, size_t* needed) {
  PMX_TEXT({
    const auto m = p->plugin->get_all_node_group_mappings();
    std::vector<std::pair<std::string, std::string>> v(m.begin(), m.end());
    std::sort(v.begin(), v.end());
    for (const auto& kv : v) text += esc(kv.first) + "\t" + esc(kv.second) + "\n";
  })
}

int32_t pmx_get_configurations(pmx_plugin* p, uint32_t available_only, char* out, size_t cap, size_t* needed) {
  PMX_TEXT({
    const std::vector<NodeGroupConfiguration> v =
        available_only ? p->plugin->get_available_configurations() : p->plugin->get_all_configuration_templates();
    for (const NodeGroupConfiguration& c : v)
      text += esc(c.name) + "\t" + std::to_string(c.min_group_size) + "\t" + std::to_string(c.max_group_size) + "\t" +
              (c.compute_requirements ? esc(*c.compute_requirements) : std::string("-")) + "\n";
  })
}

int32_t pmx_dissolve_group(pmx_plugin* p, const char* group_id) { PMX_TRY(p->plugin->dissolve_group(group_id ? group_id : "")) }

int32_t pmx_upload_file_name(pmx_plugin* p, const char* file_name, const char* address, char* out, size_t cap, size_t* needed) {
  PMX_TEXT({
    std::string key_group;
    const std::string name = upload_file_name(
        *p->plugin, file_name ? file_name : "", address ? address : "",
        [&](const std::string&, const std::string& g) {
          key_group = g;
          return uint64_t(p->upload_count.load());
        });
    text = esc(name) + "\n" + esc(key_group) + "\n";
  })
}

int32_t pmx_take_webhooks(pmx_plugin* p, char* out, size_t cap, size_t* needed) {
  std::lock_guard<std::mutex> lk(p->hook->mu);
  const int32_t rc = give(p->hook->text, out, cap, needed);
  if (rc == 0 && out) p->hook->text.clear();
  return rc;
}

}  // extern "C"
