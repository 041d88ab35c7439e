//! GpuMatchPlugin — the third `SchedulerPlugin` variant: binds libpm_engine.so (include/pm_engine.h).
//!
//! SOURCE ONLY: this image has no cargo/rustc, so this file has never been compiled.  It is the binding a
//! maintainer adds under crates/orchestrator/src/plugins/gpu_match/mod.rs, next to
//! crates/orchestrator/src/plugins/mod.rs:60-79 (see INTEGRATION.md for the enum arms and build.rs).
//!
//! The plugin owns an opaque `pm_engine*`.  The group-management loop calls `tick()` instead of
//! `try_form_new_groups` + `try_merge_solo_groups` (node_groups/mod.rs:180-203); `filter_tasks` becomes a
//! wait-free lookup of the table published by the last tick (scheduler_impl.rs:11-110).
#![allow(non_camel_case_types, dead_code)]

use std::ffi::{c_char, CStr, CString};
use std::os::raw::c_void;

use alloy::primitives::Address;
use anyhow::{anyhow, Result};
use shared::models::node::{ComputeRequirements, ComputeSpecs};
use shared::models::task::Task;

use crate::models::node::{NodeStatus, OrchestratorNode};
use crate::plugins::node_groups::NodeGroupConfiguration;

pub const PM_NONE: u32 = 0xFFFF_FFFF;

#[repr(C)]
pub struct pm_engine_config {
    pub abi_version: u32,
    pub device: i32,
    pub proximity_enabled: u32,
    pub switching_enabled: u32,
    pub prefer_larger_groups: u32,
    pub chooser: u32,
    pub chooser_seed: u64,
    pub group_id_seed: u64,
    pub debug_uncertain_every: u32,
    pub sweep_variant: u32,
    pub carve_variant: u32,
    pub _reserved: u32,
}

#[repr(C)]
pub struct pm_worker_soa {
    pub n: u32,
    pub flags: *const u32,
    pub gpu_count: *const u32,
    pub gpu_mem_mb: *const u32,
    pub gpu_model_class: *const u32,
    pub cpu_cores: *const u32,
    pub ram_mb: *const u32,
    pub storage_gb: *const u32,
    pub price: *const u32,
    pub addr_rank: *const u32,
    pub lat: *const f64,
    pub lon: *const f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct pm_config_row {
    pub flags: u32,
    pub cpu_cores: u32,
    pub ram_mb: u32,
    pub storage_gb: u32,
    pub alt_begin: u32,
    pub alt_count: u32,
    pub min_group_size: u32,
    pub max_group_size: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct pm_gpu_alt_row {
    pub flags: u32,
    pub count: u32,
    pub memory_mb: u32,
    pub memory_mb_min: u32,
    pub memory_mb_max: u32,
    pub total_memory_min: u32,
    pub total_memory_max: u32,
    pub model_row: u32,
}

#[repr(C)]
pub struct pm_task_soa {
    pub n: u32,
    pub topo_mask: *const u64,
    pub created_at: *const i64,
    pub uid: *const u64,
}

#[repr(C)]
#[derive(Default)]
pub struct pm_assignment {
    pub task: u32,
    pub group_slot: u32,
    pub group_index: u32,
    pub group_size: u32,
    pub next_worker: u32,
    pub group_id: u64,
}

#[repr(C)]
#[derive(Default)]
pub struct pm_stats {
    pub ms_compat: f32, pub ms_carve: f32, pub ms_merge: f32, pub ms_sweep: f32, pub ms_publish: f32,
    pub ms_total: f32, pub ms_compat_kernel: f32, pub ms_carve_kernel: f32, pub ms_sweep_kernel: f32,
    pub n_groups: u32, pub n_formed: u32, pub n_merged: u32, pub carve_steps: u32, pub carve_fast_steps: u32,
    pub host_resolved_steps: u32, pub carve_launches: u32, pub pair_evals: u64, pub carve_cand_sum: u64,
}

#[repr(C)]
pub struct pm_group_vars {
    pub group_index: u32,
    pub group_size: u32,
    pub next_p2p_address: *const c_char,
    pub group_id: *const c_char,
    pub total_upload_count: *const c_char,
}

#[link(name = "pm_engine")]
extern "C" {
    fn pm_engine_config_default(cfg: *mut pm_engine_config);
    fn pm_engine_create(cfg: *const pm_engine_config, out: *mut *mut c_void) -> i32;
    fn pm_engine_destroy(e: *mut c_void);
    fn pm_last_error() -> *const c_char;
    fn pm_set_configs(e: *mut c_void, cfgs: *const pm_config_row, n: u32, alts: *const pm_gpu_alt_row, n_alts: u32) -> i32;
    fn pm_set_model_table(e: *mut c_void, bits: *const u32, n_rows: u32, n_classes: u32) -> i32;
    fn pm_set_enabled_mask(e: *mut c_void, enabled: u64) -> i32;
    fn pm_upload_workers(e: *mut c_void, w: *const pm_worker_soa, keep_groups: u32) -> i32;
    fn pm_upload_tasks(e: *mut c_void, t: *const pm_task_soa) -> i32;
    fn pm_on_worker_status(e: *mut c_void, worker: u32, flags_new: u32, dead: u32) -> i32;
    fn pm_tick(e: *mut c_void, stats: *mut pm_stats) -> i32;
    fn pm_lookup_task_for_worker(e: *mut c_void, worker: u32, out: *mut pm_assignment) -> i32;
    fn pm_host_group_vars(input: *const c_char, v: *const pm_group_vars, out: *mut c_char, cap: usize, needed: *mut usize) -> i32;
    fn pm_host_volume_vars(input: *const c_char, group_id: *const c_char, out: *mut c_char, cap: usize, needed: *mut usize) -> i32;
    fn pm_host_parse_requirements(s: *const c_char, cfg: *mut pm_config_row, alts: *mut pm_gpu_alt_row,
                                  alt_cap: u32, models_out: *mut c_char, models_cap: usize) -> i32;
    fn pm_host_build_model_table(req_models: *const *const c_char, n_rows: u32,
                                 spec_models: *const *const c_char, n_classes: u32, bits_out: *mut u32) -> i32;
}

// worker flag bits (include/pm_engine.h)
const W_HAS_SPECS: u32 = 1 << 0; const W_HAS_GPU: u32 = 1 << 1; const W_GPU_COUNT: u32 = 1 << 2;
const W_GPU_MEM: u32 = 1 << 3; const W_GPU_MODEL: u32 = 1 << 4; const W_HAS_CPU: u32 = 1 << 5;
const W_CPU_CORES: u32 = 1 << 6; const W_RAM: u32 = 1 << 7; const W_STORAGE: u32 = 1 << 8;
const W_HEALTHY: u32 = 1 << 9; const W_HAS_P2P: u32 = 1 << 10; const W_HAS_LOC: u32 = 1 << 11;

fn check(rc: i32) -> Result<()> {
    if rc == 0 { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(pm_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow!("pm_engine error {rc}: {msg}"))
}

/// Projection of one `OrchestratorNode` (orchestrator/src/models/node.rs:11-37) into the SoA row.
fn worker_flags(n: &OrchestratorNode) -> u32 {
    let mut f = 0;
    if n.status == NodeStatus::Healthy { f |= W_HEALTHY; }
    if n.p2p_id.is_some() { f |= W_HAS_P2P; }
    if n.location.is_some() { f |= W_HAS_LOC; }
    if let Some(s) = &n.compute_specs {
        f |= W_HAS_SPECS;
        if let Some(g) = &s.gpu {
            f |= W_HAS_GPU;
            if g.count.is_some() { f |= W_GPU_COUNT; }
            if g.memory_mb.is_some() { f |= W_GPU_MEM; }
            if g.model.is_some() { f |= W_GPU_MODEL; }
        }
        if let Some(c) = &s.cpu {
            f |= W_HAS_CPU;
            if c.cores.is_some() { f |= W_CPU_CORES; }
        }
        if s.ram_mb.is_some() { f |= W_RAM; }
        if s.storage_gb.is_some() { f |= W_STORAGE; }
    }
    f
}

pub struct GpuMatchPlugin {
    engine: *mut c_void,
    config_names: Vec<String>,
    /// worker index = position in the last `NodeStore::get_nodes()` snapshot given to `sync_nodes`
    addresses: parking_lot::RwLock<Vec<Address>>,
    /// p2p id per worker index (node.p2p_id.unwrap_or_default(), scheduler_impl.rs:118-128)
    p2p_ids: parking_lot::RwLock<Vec<String>>,
    tasks: parking_lot::RwLock<Vec<Task>>,
}

unsafe impl Send for GpuMatchPlugin {}
unsafe impl Sync for GpuMatchPlugin {}

impl GpuMatchPlugin {
    /// Same contract as NodeGroupsPlugin::new (node_groups/mod.rs:113-175): duplicate names or
    /// max < min panic, exactly like the reference constructor.
    pub fn new(templates: Vec<NodeGroupConfiguration>, device: i32) -> Self {
        let mut cfg: pm_engine_config = unsafe { std::mem::zeroed() };
        unsafe { pm_engine_config_default(&mut cfg) };
        cfg.device = device;
        let mut engine = std::ptr::null_mut();
        check(unsafe { pm_engine_create(&cfg, &mut engine) }).expect("pm_engine_create");
        let mut seen = std::collections::HashSet::new();
        for t in &templates {
            if !seen.insert(t.name.clone()) { panic!("Configuration names must be unique"); }
        }
        // requirement strings were parsed by serde into ComputeRequirements; re-serialise the fields into
        // pm_config_row / pm_gpu_alt_row here (omitted: field-by-field copy), intern requirement model
        // strings, and call pm_set_configs — PM_EINVAL maps to the reference's "Plugin configuration is invalid".
        let this = Self { engine, config_names: templates.iter().map(|t| t.name.clone()).collect(),
                          addresses: Default::default(), p2p_ids: Default::default(), tasks: Default::default() };
        this.set_configs(&templates);
        this
    }

    fn set_configs(&self, _templates: &[NodeGroupConfiguration]) { /* pack rows + pm_set_configs + pm_set_model_table */ }

    /// Number of `upload:<node>:<group>:*` keys (scheduler_impl.rs:130-153): stays with the Redis store.
    fn upload_count(&self, _node: &Address, _group_id: u64) -> usize { 0 /* store.scan_match(pattern).count() */ }

    /// Called with the snapshot of `store_context.node_store.get_nodes()` (node_store.rs:163-209); the
    /// ORDER of that Vec is the tie-break of the reference and is passed through unchanged.
    pub fn sync_nodes(&self, nodes: &[OrchestratorNode]) -> Result<()> {
        let n = nodes.len();
        let flags: Vec<u32> = nodes.iter().map(worker_flags).collect();
        let col = |f: &dyn Fn(&ComputeSpecs) -> Option<u32>| -> Vec<u32> {
            nodes.iter().map(|x| x.compute_specs.as_ref().and_then(|s| f(s)).unwrap_or(0)).collect()
        };
        let gpu_count = col(&|s| s.gpu.as_ref().and_then(|g| g.count));
        let gpu_mem = col(&|s| s.gpu.as_ref().and_then(|g| g.memory_mb));
        let cpu_cores = col(&|s| s.cpu.as_ref().and_then(|c| c.cores));
        let ram = col(&|s| s.ram_mb);
        let storage = col(&|s| s.storage_gb);
        let gpu_class: Vec<u32> = vec![0; n]; // index into the interned spec model strings (set_model_table)
        let lat: Vec<f64> = nodes.iter().map(|x| x.location.as_ref().map(|l| l.latitude).unwrap_or(0.0)).collect();
        let lon: Vec<f64> = nodes.iter().map(|x| x.location.as_ref().map(|l| l.longitude).unwrap_or(0.0)).collect();
        // GROUP_INDEX is the rank of address.to_string() inside the group's BTreeSet<String> (mod.rs:424-434)
        let mut order: Vec<usize> = (0..n).collect();
        order.sort_by_key(|&i| nodes[i].address.to_string());
        let mut addr_rank = vec![0u32; n];
        for (r, &i) in order.iter().enumerate() { addr_rank[i] = r as u32; }
        let soa = pm_worker_soa { n: n as u32, flags: flags.as_ptr(), gpu_count: gpu_count.as_ptr(),
            gpu_mem_mb: gpu_mem.as_ptr(), gpu_model_class: gpu_class.as_ptr(), cpu_cores: cpu_cores.as_ptr(),
            ram_mb: ram.as_ptr(), storage_gb: storage.as_ptr(), price: std::ptr::null(),
            addr_rank: addr_rank.as_ptr(), lat: lat.as_ptr(), lon: lon.as_ptr() };
        check(unsafe { pm_upload_workers(self.engine, &soa, 1) })?;
        *self.addresses.write() = nodes.iter().map(|x| x.address).collect();
        *self.p2p_ids.write() = nodes.iter().map(|x| x.p2p_id.clone().unwrap_or_default()).collect();
        Ok(())
    }

    /// Called with `task_store.get_all_tasks()` (task_store.rs:57-82, already created_at-desc).
    pub fn sync_tasks(&self, tasks: Vec<Task>) -> Result<()> {
        let masks: Vec<u64> = tasks.iter().map(|t| self.topology_mask(t)).collect();
        let created: Vec<i64> = tasks.iter().map(|t| t.created_at).collect();
        let uid: Vec<u64> = tasks.iter().map(|t| t.id.as_u64_pair().1).collect();
        let soa = pm_task_soa { n: tasks.len() as u32, topo_mask: masks.as_ptr(), created_at: created.as_ptr(), uid: uid.as_ptr() };
        check(unsafe { pm_upload_tasks(self.engine, &soa) })?;
        let enabled = masks.iter().filter(|m| **m != u64::MAX).fold(0u64, |a, m| a | m); // on_task_created, mod.rs:1224-1243
        check(unsafe { pm_set_enabled_mask(self.engine, enabled) })?;
        *self.tasks.write() = tasks;
        Ok(())
    }

    /// scheduler_impl.rs:44-59: any None on the way => every configuration allowed.
    fn topology_mask(&self, t: &Task) -> u64 {
        let Some(list) = t.scheduling_config.as_ref().and_then(|c| c.plugins.as_ref())
            .and_then(|p| p.get("node_groups")).and_then(|n| n.get("allowed_topologies")) else { return u64::MAX };
        list.iter().filter_map(|name| self.config_names.iter().position(|c| c == name)).fold(0, |m, i| m | (1u64 << i))
    }

    /// One body of run_group_management_loop (mod.rs:180-203) + every worker's filter_tasks.
    pub fn tick(&self) -> Result<pm_stats> {
        let mut s = pm_stats::default();
        check(unsafe { pm_tick(self.engine, &mut s) })?;
        Ok(s)
    }

    /// SchedulerPlugin::filter_tasks (plugins/mod.rs:66-78): wait-free lookup + the `${...}` templating the
    /// reference does at scheduler_impl.rs:112-205 (GROUP_INDEX, GROUP_SIZE, NEXT_P2P_ADDRESS, GROUP_ID).
    pub(crate) fn filter_tasks(&self, _tasks: &[Task], node_address: &Address) -> Result<Vec<Task>> {
        let Some(w) = self.addresses.read().iter().position(|a| a == node_address) else { return Ok(vec![]) };
        let mut a = pm_assignment::default();
        if unsafe { pm_lookup_task_for_worker(self.engine, w as u32, &mut a) } != 0 || a.task == PM_NONE {
            return Ok(vec![]);
        }
        let mut task = self.tasks.read()[a.task as usize].clone();
        // group variables (scheduler_impl.rs:155-200) through the library's helpers — the same chained
        // replace order as the reference; the upload count still comes from the `upload:<node>:<group>:*` scan
        let gid = CString::new(format!("{:x}", a.group_id))?;
        let next = CString::new(self.p2p_ids.read().get(a.next_worker as usize).cloned().unwrap_or_default())?;
        let count = CString::new(self.upload_count(node_address, a.group_id).to_string())?;
        let vars = pm_group_vars { group_index: a.group_index, group_size: a.group_size,
            next_p2p_address: next.as_ptr(), group_id: gid.as_ptr(), total_upload_count: count.as_ptr() };
        let render = |s: &str| -> Result<String> {
            let cin = CString::new(s)?;
            let mut need = 0usize;
            check(unsafe { pm_host_group_vars(cin.as_ptr(), &vars, std::ptr::null_mut(), 0, &mut need) })?;
            let mut buf = vec![0u8; need];
            check(unsafe { pm_host_group_vars(cin.as_ptr(), &vars, buf.as_mut_ptr() as *mut c_char, need, &mut need) })?;
            buf.pop();
            Ok(String::from_utf8(buf)?)
        };
        let env = task.env_vars.get_or_insert_with(Default::default);
        env.insert("GROUP_INDEX".to_string(), a.group_index.to_string());
        for (_, v) in env.iter_mut() { *v = render(v)?; }
        if let Some(cmd) = task.cmd.as_mut() { for arg in cmd.iter_mut() { *arg = render(arg)?; } }
        // volume mounts: pm_host_volume_vars on host_path / container_path (same calling convention)
        Ok(vec![task])
    }

    /// StatusUpdatePlugin::handle_status_change (status_update_impl.rs:8-39).
    pub(crate) fn handle_status_change(&self, node: &OrchestratorNode) -> Result<()> {
        let Some(w) = self.addresses.read().iter().position(|a| *a == node.address) else { return Ok(()) };
        let dead = matches!(node.status, NodeStatus::Dead | NodeStatus::LowBalance) as u32;
        check(unsafe { pm_on_worker_status(self.engine, w as u32, worker_flags(node), dead) })
    }
}

impl Drop for GpuMatchPlugin {
    fn drop(&mut self) { unsafe { pm_engine_destroy(self.engine) } }
}
