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

#include "pm_engine_types.inc"
}  // namespace pm

using namespace pm;

namespace pm {
struct FormRun;
}

#include "pm_engine_state.inc"

namespace pm {

#include "pm_engine_groups.inc"
#include "pm_engine_carve.inc"
#include "pm_engine_match.inc"
#include "pm_engine_merge.inc"
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
  e->d_sel_own.release(); e->d_table_x.release(); e->d_row_stage.release(); if (e->h_row_pin) (void)hipHostFree(e->h_row_pin); if (e->ev_row_stage) (void)hipEventDestroy(e->ev_row_stage); e->d_nb_idx.release(); e->d_nb_val.release();
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

#include "pm_engine_workers.inc"
#include "pm_engine_tasks.inc"
#include "pm_engine_api.inc"
#include "pm_engine_tick.inc"
#include "pm_engine_dist.inc"
#include "pm_engine_debug.inc"
}  // extern "C"
