// mock_engine.cpp — TEST INFRASTRUCTURE: a CPU stand-in for libpm_engine.so behind the same C ABI (include/pm_engine.h),
// so that the compiled host side (protocol_amd/plugin/gpu_match_plugin.cpp) can be run, sanitised and raced in a
// container without a GPU.  It is NOT the engine and NOT the oracle: the matching here is a toy (every healthy worker
// with a p2p id is compatible with every configuration; groups are cut from the free workers in index order, one per
// configuration and pass), just enough for the plugin's own bookkeeping to have something to keep: row indices, address ranks,
// positions in the task list that move under deltas, groups that dissolve, the life-cycle feed.  What it keeps of the
// real contract, because the plugin relies on it:
//   * rows are appended, never moved; indices out of range fail a call before anything is applied;
//   * pm_tasks_insert_front refuses rows that are not strictly newer than the newest task (PM_EINVAL);
//   * a group is bound to its task by uid: the position pm_lookup_task_for_worker reports is the task's position in the
//     CURRENT list; a deleted task or a dead member dissolves the group and its rows read "no group" at once;
//   * a task inserted without `republish` reaches an idle group only at the next tick;
//   * the life-cycle feed (created at the tick, destroyed at the dissolution), members in address-rank order, and the
//     two-call drain (PM_ERANGE + sizes when the buffer is too small, nothing drained).
// Every call is logged (pm_mock_calls) so a test can check WHAT the plugin sent, in which order.  The pm_host_* helpers
// are the product's own (pm_host.cpp is compiled into this library unchanged).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pm_engine.h"
#include "pm_internal.h"

namespace pm {
static thread_local std::string g_last_error;
int32_t set_error(int32_t code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
}  // namespace pm

struct MockGroup {
  uint64_t id;
  uint32_t cfg;
  uint64_t task_uid;  // 0 = holds no task
  std::vector<uint32_t> members;  // address-rank order
};

struct pm_engine {
  std::mutex mu;
  std::vector<pm_config_row> cfgs;
  uint64_t enabled = 0;
  std::vector<uint32_t> flags, addr_rank;
  struct T { uint64_t mask; int64_t created; uint64_t uid; };
  std::vector<T> tasks;
  std::vector<MockGroup> groups;
  std::vector<int32_t> group_of;
  bool published = false;
  uint64_t id_rng = 1, id_seed = 1;
  bool events_on = false;
  std::vector<pm_group_event> ev;
  std::vector<uint32_t> ev_members;
  // the stepwise multi-GPU tick: ownership, the exchange buffer ("device" memory is host memory here), the gathered table
  uint32_t dist_rank = 0, dist_world = 1, dist_cap = 0;
  int dist_phase = 0;
  uint32_t dist_formed = 0;
  std::vector<uint8_t> shard;
  std::vector<uint32_t> xrow;              // shard * cap + index within the shard
  std::vector<pm_assignment> xbuf, table;  // [world][cap] / per worker, as gathered
  bool table_valid = false;
};

static std::mutex g_log_mu;
static std::string g_log;
static std::string g_log_copy;
static int g_delay_us = 0;

static void logf(const std::string& s) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  g_log += s;
  g_log += '\n';
}
static std::string list(const uint32_t* p, uint32_t n) {
  std::string s = "[";
  for (uint32_t i = 0; i < n; ++i) s += (i ? "," : "") + std::to_string(p[i]);
  return s + "]";
}
static void delay() {
  if (g_delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(g_delay_us));
}

static void log_event(pm_engine* e, uint32_t kind, const MockGroup& g) {
  if (!e->events_on) return;
  pm_group_event ev{};
  ev.group_id = g.id;
  ev.kind = kind;
  ev.config = g.cfg;
  ev.member_begin = uint32_t(e->ev_members.size());
  ev.n_members = uint32_t(g.members.size());
  e->ev_members.insert(e->ev_members.end(), g.members.begin(), g.members.end());
  e->ev.push_back(ev);
}

static void dissolve(pm_engine* e, size_t slot) {
  e->table_valid = false;  // (the real engine patches the published rows in place; the toy falls back to computing them)
  log_event(e, PM_GROUP_DESTROYED, e->groups[slot]);
  e->groups.erase(e->groups.begin() + long(slot));
  std::fill(e->group_of.begin(), e->group_of.end(), -1);
  for (size_t g = 0; g < e->groups.size(); ++g)
    for (uint32_t w : e->groups[g].members) e->group_of[w] = int32_t(g);
}

static void offer_tasks(pm_engine* e) {  // idle groups take the first task that names their configuration
  for (MockGroup& g : e->groups) {
    if (g.task_uid) continue;
    for (const auto& t : e->tasks)
      if ((t.mask >> g.cfg) & 1ull) {
        g.task_uid = t.uid;
        break;
      }
  }
}

extern "C" {

const char* pm_last_error(void) { return pm::g_last_error.c_str(); }

void pm_engine_config_default(pm_engine_config* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->abi_version = PM_ABI_VERSION;
  c->proximity_enabled = c->switching_enabled = c->prefer_larger_groups = 1;
  c->group_id_seed = 1;
}

int32_t pm_engine_create(const pm_engine_config* cfg, pm_engine** out) {
  if (!cfg || !out) return pm::set_error(PM_EINVAL, "null argument");
  if (cfg->abi_version != PM_ABI_VERSION) return pm::set_error(PM_EINVAL, "ABI version mismatch");
  pm_engine* e = new pm_engine();
  e->id_rng = e->id_seed = cfg->group_id_seed;
  *out = e;
  logf("create device=" + std::to_string(cfg->device));
  return PM_OK;
}
void pm_engine_destroy(pm_engine* e) {
  logf("destroy");
  delete e;
}

int32_t pm_set_configs(pm_engine* e, const pm_config_row* cfgs, uint32_t n, const pm_gpu_alt_row* alts, uint32_t n_alts) {
  if (!e || (n && !cfgs) || (n_alts && !alts)) return pm::set_error(PM_EINVAL, "null argument");
  for (uint32_t i = 0; i < n; ++i) {
    if (cfgs[i].min_group_size == 0 || cfgs[i].max_group_size < cfgs[i].min_group_size)
      return pm::set_error(PM_EINVAL, "Plugin configuration is invalid");
    if (cfgs[i].alt_begin + cfgs[i].alt_count > n_alts) return pm::set_error(PM_EINVAL, "alternative range outside the table");
  }
  std::lock_guard<std::mutex> lk(e->mu);
  e->cfgs.assign(cfgs, cfgs + n);
  std::string s = "set_configs n=" + std::to_string(n) + " alts=" + std::to_string(n_alts) + " sizes=";
  for (uint32_t i = 0; i < n; ++i) s += std::to_string(cfgs[i].min_group_size) + "-" + std::to_string(cfgs[i].max_group_size) + " ";
  for (uint32_t i = 0; i < n_alts; ++i)
    if (alts[i].flags & PM_G_MODEL) s += "model_row=" + std::to_string(alts[i].model_row) + " ";
  logf(s);
  return PM_OK;
}
int32_t pm_set_model_table(pm_engine* e, const uint32_t* bits, uint32_t n_rows, uint32_t n_classes) {
  const uint32_t words = (n_classes + 31u) / 32u;
  if (!e || (!bits && n_rows * words != 0u)) return pm::set_error(PM_EINVAL, "null argument");
  std::string s = "set_model_table rows=" + std::to_string(n_rows) + " classes=" + std::to_string(n_classes) + " bits=";
  for (uint32_t i = 0; i < n_rows * words; ++i) s += std::to_string(bits[i]) + " ";
  logf(s);
  return PM_OK;
}
int32_t pm_set_enabled_mask(pm_engine* e, uint64_t enabled) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  e->enabled = enabled;
  logf("set_enabled_mask " + std::to_string(enabled));
  return PM_OK;
}

int32_t pm_upload_workers(pm_engine* e, const pm_worker_soa* w, uint32_t keep_groups) {
  if (!e || !w) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (keep_groups && w->n < e->flags.size()) return pm::set_error(PM_EINVAL, "keep_groups requires the rows the engine has in front");
  e->flags.assign(w->flags, w->flags + w->n);
  e->addr_rank.assign(w->addr_rank, w->addr_rank + w->n);
  if (!keep_groups) {
    e->groups.clear();
    e->group_of.assign(w->n, -1);
    e->id_rng = e->id_seed;  // (the real engine restarts its id stream with the groups)
  } else {
    e->group_of.resize(w->n, -1);
  }
  logf("upload_workers n=" + std::to_string(w->n) + " keep=" + std::to_string(keep_groups));
  return PM_OK;
}
static std::atomic<int> g_fail_appends{0};  // pm_mock_fail_appends: that many pm_append_workers calls fail before anything is applied
int32_t pm_append_workers(pm_engine* e, const pm_worker_soa* rows, uint32_t* first_index) {
  if (!e || !rows || !first_index) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (g_fail_appends.load() > 0) {
    g_fail_appends.fetch_sub(1);
    logf("append_workers FAILED (injected)");
    return pm::set_error(PM_ENOMEM, "injected failure");
  }
  *first_index = uint32_t(e->flags.size());
  e->flags.insert(e->flags.end(), rows->flags, rows->flags + rows->n);
  e->addr_rank.insert(e->addr_rank.end(), rows->addr_rank, rows->addr_rank + rows->n);
  e->group_of.resize(e->flags.size(), -1);
  std::string cls = "[";  // (the class only where the row says it has a model: what the engine reads)
  for (uint32_t i = 0; i < rows->n; ++i)
    cls += (i ? "," : "") + ((rows->flags[i] & PM_W_GPU_MODEL) ? std::to_string(rows->gpu_model_class[i]) : std::string("-"));
  logf("append_workers n=" + std::to_string(rows->n) + " first=" + std::to_string(*first_index) + " flags=" + list(rows->flags, rows->n) +
       " gpu_class=" + cls + "]");
  return PM_OK;
}
int32_t pm_update_workers(pm_engine* e, const uint32_t* idx, const pm_worker_soa* rows) {
  if (!e || !idx || !rows) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  for (uint32_t k = 0; k < rows->n; ++k)
    if (idx[k] >= e->flags.size()) return pm::set_error(PM_ERANGE, "worker index out of range");
  for (uint32_t k = 0; k < rows->n; ++k) {
    e->flags[idx[k]] = rows->flags[k];
    e->addr_rank[idx[k]] = rows->addr_rank[k];
  }
  logf("update_workers idx=" + list(idx, rows->n) + " flags=" + list(rows->flags, rows->n) + " ranks=" + list(rows->addr_rank, rows->n));
  return PM_OK;
}
int32_t pm_set_addr_ranks(pm_engine* e, const uint32_t* r, uint32_t n) {
  if (!e || !r) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (n != e->flags.size()) return pm::set_error(PM_EINVAL, "addr_rank column of the wrong length");
  e->addr_rank.assign(r, r + n);
  logf("set_addr_ranks " + list(r, n));
  return PM_OK;
}

static void drop_groups_without_task(pm_engine* e) {  // a claimed task that is gone takes its group with it
  for (size_t g = 0; g < e->groups.size();) {
    const uint64_t u = e->groups[g].task_uid;
    const bool gone = u && std::none_of(e->tasks.begin(), e->tasks.end(), [&](const pm_engine::T& t) { return t.uid == u; });
    if (gone) dissolve(e, g);
    else ++g;
  }
}

int32_t pm_upload_tasks(pm_engine* e, const pm_task_soa* t) {
  if (!e || !t) return pm::set_error(PM_EINVAL, "null argument");
  {
    std::lock_guard<std::mutex> lk(e->mu);
    e->tasks.clear();
    for (uint32_t i = 0; i < t->n; ++i) e->tasks.push_back({t->topo_mask[i], t->created_at[i], t->uid[i]});
    drop_groups_without_task(e);
    logf("upload_tasks n=" + std::to_string(t->n));
  }
  delay();  // (the new positions are visible to look-ups from here; the caller has not changed its list yet)
  return PM_OK;
}
int32_t pm_tasks_insert_front_ex(pm_engine* e, const pm_task_soa* t, uint32_t republish) {
  if (!e || !t) return pm::set_error(PM_EINVAL, "null argument");
  {
  std::lock_guard<std::mutex> lk(e->mu);
  for (uint32_t i = 0; i < t->n; ++i) {
    const int64_t newest_behind = i + 1 < t->n ? t->created_at[i + 1] : (e->tasks.empty() ? INT64_MIN : e->tasks[0].created);
    if (!(t->created_at[i] > newest_behind)) return pm::set_error(PM_EINVAL, "inserted tasks must be newer than the table's newest");
  }
  std::vector<pm_engine::T> front;
  for (uint32_t i = 0; i < t->n; ++i) front.push_back({t->topo_mask[i], t->created_at[i], t->uid[i]});
  e->tasks.insert(e->tasks.begin(), front.begin(), front.end());
  if (republish) offer_tasks(e);
  logf("tasks_insert_front n=" + std::to_string(t->n) + " republish=" + std::to_string(republish));
  }
  delay();
  return PM_OK;
}
int32_t pm_tasks_insert_front(pm_engine* e, const pm_task_soa* t) { return pm_tasks_insert_front_ex(e, t, 0); }
int32_t pm_tasks_delete(pm_engine* e, const uint64_t* uids, uint32_t n, uint32_t* n_deleted) {
  if (!e || (n && !uids)) return pm::set_error(PM_EINVAL, "null argument");
  {
  std::lock_guard<std::mutex> lk(e->mu);
  uint32_t d = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const auto it = std::find_if(e->tasks.begin(), e->tasks.end(), [&](const pm_engine::T& t) { return t.uid == uids[i]; });
    if (it != e->tasks.end()) {
      e->tasks.erase(it);
      ++d;
    }
  }
  drop_groups_without_task(e);
  if (n_deleted) *n_deleted = d;
  logf("tasks_delete n=" + std::to_string(n) + " deleted=" + std::to_string(d));
  }
  delay();
  return PM_OK;
}

static void status_locked(pm_engine* e, uint32_t w, uint32_t flags_new, uint32_t dead) {
  e->flags[w] = flags_new;
  if (dead && e->group_of[w] >= 0) dissolve(e, size_t(e->group_of[w]));
}
int32_t pm_on_worker_status(pm_engine* e, uint32_t w, uint32_t flags_new, uint32_t dead) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (w >= e->flags.size()) return pm::set_error(PM_ERANGE, "worker index out of range");
  status_locked(e, w, flags_new, dead);
  logf("on_worker_status w=" + std::to_string(w) + " flags=" + std::to_string(flags_new) + " dead=" + std::to_string(dead));
  return PM_OK;
}
int32_t pm_on_worker_status_many(pm_engine* e, const uint32_t* ws, const uint32_t* fl, const uint32_t* dead, uint32_t n) {
  if (!e || (n && (!ws || !fl))) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  for (uint32_t i = 0; i < n; ++i)
    if (ws[i] >= e->flags.size()) return pm::set_error(PM_ERANGE, "worker index out of range");
  for (uint32_t i = 0; i < n; ++i) status_locked(e, ws[i], fl[i], dead ? dead[i] : 0u);
  logf("on_worker_status_many w=" + list(ws, n) + " flags=" + list(fl, n));
  return PM_OK;
}

int32_t pm_enable_group_events(pm_engine* e, uint32_t on) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  e->events_on = on != 0;
  if (!on) {
    e->ev.clear();
    e->ev_members.clear();
  }
  logf("enable_group_events " + std::to_string(on));
  return PM_OK;
}
int32_t pm_drain_group_events(pm_engine* e, pm_group_event* events, uint32_t cap_e, uint32_t* members, uint32_t cap_m,
                              uint32_t* n_events, uint32_t* n_members) {
  if (!e || !n_events || !n_members) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  const uint32_t ne = uint32_t(e->ev.size()), nm = uint32_t(e->ev_members.size());
  *n_events = ne;
  *n_members = nm;
  if (ne == 0) return PM_OK;
  if (!events || cap_e < ne || (nm && (!members || cap_m < nm))) return pm::set_error(PM_ERANGE, "event buffers too small");
  std::copy(e->ev.begin(), e->ev.end(), events);
  std::copy(e->ev_members.begin(), e->ev_members.end(), members);
  e->ev.clear();
  e->ev_members.clear();
  return PM_OK;
}

static void fill_group(const MockGroup& g, pm_group* out, const pm_engine* e) {
  out->id = g.id;
  out->config = g.cfg;
  out->n_members = uint32_t(g.members.size());
  out->member_begin = 0;
  out->task = PM_NONE;
  if (g.task_uid)
    for (size_t i = 0; i < e->tasks.size(); ++i)
      if (e->tasks[i].uid == g.task_uid) out->task = uint32_t(i);
}
int32_t pm_get_groups(pm_engine* e, int32_t* group_of_worker, pm_group* groups, uint32_t cap_groups, uint32_t* n_groups,
                      uint32_t* members, uint32_t cap_members, uint32_t* n_members) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  uint32_t M = 0;
  for (const MockGroup& g : e->groups) M += uint32_t(g.members.size());
  if (n_groups) *n_groups = uint32_t(e->groups.size());
  if (n_members) *n_members = M;
  if (group_of_worker) std::copy(e->group_of.begin(), e->group_of.end(), group_of_worker);
  if (groups && cap_groups < e->groups.size()) return pm::set_error(PM_ERANGE, "groups buffer too small");
  if (members && cap_members < M) return pm::set_error(PM_ERANGE, "members buffer too small");
  uint32_t off = 0;
  for (size_t g = 0; g < e->groups.size(); ++g) {
    if (groups) {
      fill_group(e->groups[g], &groups[g], e);
      groups[g].member_begin = off;
    }
    if (members) std::copy(e->groups[g].members.begin(), e->groups[g].members.end(), members + off);
    off += uint32_t(e->groups[g].members.size());
  }
  return PM_OK;
}
static int32_t give_one(pm_engine* e, size_t g, pm_group* out, uint32_t* members, uint32_t cap) {
  fill_group(e->groups[g], out, e);
  if (!members) return PM_OK;
  if (cap < e->groups[g].members.size()) return pm::set_error(PM_ERANGE, "members buffer too small");
  std::copy(e->groups[g].members.begin(), e->groups[g].members.end(), members);
  return PM_OK;
}
int32_t pm_get_group_by_id(pm_engine* e, uint64_t id, pm_group* out, uint32_t* members, uint32_t cap, uint32_t* slot) {
  if (!e || !out || !slot) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  std::memset(out, 0, sizeof(*out));
  *slot = PM_NONE;
  for (size_t g = 0; g < e->groups.size(); ++g)
    if (e->groups[g].id == id) {
      *slot = uint32_t(g);
      return give_one(e, g, out, members, cap);
    }
  return PM_OK;
}
int32_t pm_get_group_of_worker(pm_engine* e, uint32_t w, pm_group* out, uint32_t* members, uint32_t cap, uint32_t* slot) {
  if (!e || !out || !slot) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (w >= e->flags.size()) return pm::set_error(PM_ERANGE, "worker index out of range");
  std::memset(out, 0, sizeof(*out));
  *slot = e->group_of[w] < 0 ? PM_NONE : uint32_t(e->group_of[w]);
  if (e->group_of[w] < 0) return PM_OK;
  return give_one(e, size_t(e->group_of[w]), out, members, cap);
}
int32_t pm_dissolve_group_by_id(pm_engine* e, uint64_t id, uint32_t* dissolved) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (dissolved) *dissolved = 0;
  for (size_t g = 0; g < e->groups.size(); ++g)
    if (e->groups[g].id == id) {
      dissolve(e, g);
      if (dissolved) *dissolved = 1;
      logf("dissolve_group_by_id found");
      return PM_OK;
    }
  logf("dissolve_group_by_id unknown");
  return PM_OK;
}

static uint32_t form_groups_locked(pm_engine* e);
static void row_of_worker(pm_engine* e, uint32_t w, pm_assignment* out);

int32_t pm_tick(pm_engine* e, pm_stats* stats) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  const uint32_t formed = form_groups_locked(e);
  offer_tasks(e);
  e->published = true;
  e->table_valid = false;
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->n_groups = uint32_t(e->groups.size());
    stats->n_formed = formed;
    stats->pair_evals = uint64_t(e->tasks.size()) * e->flags.size();
  }
  logf("tick formed=" + std::to_string(formed));
  return PM_OK;
}

static uint32_t form_groups_locked(pm_engine* e) {
  std::vector<uint32_t> order;
  pm::available_order(e->cfgs.data(), uint32_t(e->cfgs.size()), e->enabled, &order);
  uint32_t formed = 0;
  // one group per configuration and pass, in carve order, until a pass forms none (so that every configuration gets
  // groups: a toy, see the head of the file)
  for (bool progress = true; progress;) {
    progress = false;
    for (uint32_t c : order) {
      const pm_config_row& cfg = e->cfgs[c];
      std::vector<uint32_t> free_w;
      for (uint32_t w = 0; w < e->flags.size(); ++w)
        if (e->group_of[w] < 0 && (e->flags[w] & PM_W_HEALTHY) && (e->flags[w] & PM_W_HAS_P2P)) free_w.push_back(w);
      if (free_w.size() < cfg.min_group_size) continue;
      free_w.resize(std::min<size_t>(free_w.size(), cfg.max_group_size));
      MockGroup g;
      g.id = pm::splitmix64_next(&e->id_rng);
      g.cfg = c;
      g.task_uid = 0;
      g.members = free_w;
      std::sort(g.members.begin(), g.members.end(), [&](uint32_t a, uint32_t b) { return e->addr_rank[a] < e->addr_rank[b]; });
      for (uint32_t w : g.members) e->group_of[w] = int32_t(e->groups.size());
      log_event(e, PM_GROUP_CREATED, g);
      e->groups.push_back(g);
      ++formed;
      progress = true;
    }
  }
  return formed;
}

// ---- the stepwise multi-GPU tick: the carve replicated, the published rows of the OWNED workers exchanged.  Look-ups of a
// rank in a world > 1 are served from the GATHERED table only: rows another rank never sent read as garbage (0xFF).
int32_t pm_set_stream(pm_engine* e, void* s) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  logf(std::string("set_stream ") + (s ? "caller's" : "own"));
  return PM_OK;
}
int32_t pm_dist_configure(pm_engine* e, uint32_t rank, uint32_t world, const uint8_t* shard) {
  if (!e || world == 0 || rank >= world) return pm::set_error(PM_EINVAL, "rank / world out of range");
  std::lock_guard<std::mutex> lk(e->mu);
  const uint32_t W = uint32_t(e->flags.size());
  if (world > 1 && W && !shard) return pm::set_error(PM_EINVAL, "null shard column");
  e->dist_rank = rank;
  e->dist_world = world;
  e->shard.assign(shard ? shard : nullptr, shard ? shard + W : nullptr);
  std::vector<uint32_t> count(world, 0);
  e->xrow.assign(W, 0);
  for (uint32_t w = 0; w < W && world > 1; ++w) {
    if (shard[w] >= world) return pm::set_error(PM_ERANGE, "shard index outside the world");
    e->xrow[w] = count[shard[w]]++;
  }
  e->dist_cap = std::max<uint32_t>(*std::max_element(count.begin(), count.end()), 1u);
  for (uint32_t w = 0; w < W && world > 1; ++w) e->xrow[w] += uint32_t(shard[w]) * e->dist_cap;
  e->xbuf.assign(size_t(world) * e->dist_cap, pm_assignment{});
  std::memset(e->xbuf.data(), 0xFF, e->xbuf.size() * sizeof(pm_assignment));
  logf("dist_configure rank=" + std::to_string(rank) + " world=" + std::to_string(world) + " rows=" + std::to_string(W));
  return PM_OK;
}
int32_t pm_dist_tick_begin(pm_engine* e) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_world > 1 && e->shard.size() != e->flags.size())
    return pm::set_error(PM_ESTATE, "the worker table changed size: call pm_dist_configure again");
  e->dist_phase = 1;
  logf("dist_tick_begin");
  return PM_OK;
}
int32_t pm_dist_carve_wait(pm_engine* e) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 1) return pm::set_error(PM_ESTATE, "pm_dist_tick_begin first");
  e->dist_formed = form_groups_locked(e);  // replicated: every rank forms the same groups with the same ids
  e->dist_phase = 2;
  logf("dist_carve_wait");
  return PM_OK;
}
int32_t pm_dist_match_begin(pm_engine* e, pm_dist_xfer* x) {
  if (!e || !x) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 2) return pm::set_error(PM_ESTATE, "the carve is not finished (pm_dist_carve_wait first)");
  std::memset(x, 0, sizeof(*x));
  offer_tasks(e);
  e->published = true;
  if (e->dist_world > 1) {
    std::memset(e->xbuf.data(), 0xFF, e->xbuf.size() * sizeof(pm_assignment));
    for (uint32_t w = 0; w < e->flags.size(); ++w)
      if (e->shard[w] == e->dist_rank) row_of_worker(e, w, &e->xbuf[e->xrow[w]]);  // the OWNED workers only
    x->recv_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->xbuf.data()));
    x->send_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->xbuf.data() + size_t(e->dist_rank) * e->dist_cap));
    x->bytes_per_rank = uint64_t(e->dist_cap) * sizeof(pm_assignment);
  }
  e->dist_phase = 3;
  logf("dist_match_begin bytes_per_rank=" + std::to_string(x->bytes_per_rank));
  return PM_OK;
}
int32_t pm_dist_tick_end(pm_engine* e, pm_stats* stats) {
  if (!e) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->dist_phase != 3) return pm::set_error(PM_ESTATE, "pm_dist_match_begin first");
  e->dist_phase = 0;
  if (e->dist_world > 1) {
    e->table.resize(e->flags.size());
    for (uint32_t w = 0; w < e->flags.size(); ++w) e->table[w] = e->xbuf[e->xrow[w]];
    e->table_valid = true;
  }
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->n_groups = uint32_t(e->groups.size());
    stats->n_formed = e->dist_formed;
  }
  logf("dist_tick_end formed=" + std::to_string(e->dist_formed));
  return PM_OK;
}

int32_t pm_tick_many(pm_engine* const* engines, uint32_t n, pm_stats* stats, uint32_t flags) {
  if (!engines || n == 0 || flags > 1u) return pm::set_error(PM_EINVAL, "null argument");
  logf("tick_many n=" + std::to_string(n));
  for (uint32_t i = 0; i < n; ++i) {
    const int32_t rc = pm_tick(engines[i], stats ? &stats[i] : nullptr);
    if (rc) return rc;
  }
  return PM_OK;
}

int32_t pm_lookup_task_for_worker(pm_engine* e, uint32_t w, pm_assignment* out) {
  if (!e || !out) return pm::set_error(PM_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->published) return pm::set_error(PM_ESTATE, "no assignment table published yet");
  if (w >= e->flags.size()) return pm::set_error(PM_ERANGE, "worker index out of range");
  if (e->table_valid && w < e->table.size()) {  // a rank of a world > 1: what the all-gather brought
    *out = e->table[w];
    return PM_OK;
  }
  row_of_worker(e, w, out);
  return PM_OK;
}
static void row_of_worker(pm_engine* e, uint32_t w, pm_assignment* out) {
  std::memset(out, 0, sizeof(*out));
  out->task = out->group_slot = out->next_worker = PM_NONE;
  if (e->group_of[w] < 0) return;
  const MockGroup& g = e->groups[size_t(e->group_of[w])];
  const size_t k = size_t(std::find(g.members.begin(), g.members.end(), w) - g.members.begin());
  out->group_slot = uint32_t(e->group_of[w]);
  out->group_index = uint32_t(k);
  out->group_size = uint32_t(g.members.size());
  out->next_worker = g.members[(k + 1) % g.members.size()];
  out->group_id = g.id;
  if (g.task_uid)
    for (size_t i = 0; i < e->tasks.size(); ++i)
      if (e->tasks[i].uid == g.task_uid) out->task = uint32_t(i);
}

// ---- what only the mock has
const char* pm_mock_calls(void) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  g_log_copy = g_log;
  return g_log_copy.c_str();
}
void pm_mock_reset_calls(void) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  g_log.clear();
}
void pm_mock_fail_appends(int n) { g_fail_appends.store(n); }
void pm_mock_set_delay_us(int us) { g_delay_us = us; }  // a pause at the END of every task-table call, behind the mock's own lock (race windows)

}  // extern "C"
