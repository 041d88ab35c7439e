//! GpuMatchPlugin — the third `SchedulerPlugin` variant: binds libpm_engine.so (include/pm_engine.h, ABI v3).
//!
//! SOURCE ONLY: this image has no cargo/rustc, so this file has never been compiled.  It is the binding a
//! maintainer adds under crates/orchestrator/src/plugins/gpu_match/mod.rs, next to
//! crates/orchestrator/src/plugins/mod.rs:60-79 (see INTEGRATION.md for the enum arms and build.rs).
//!
//! The plugin owns an opaque `pm_engine*`.  The group-management loop calls `tick()` instead of
//! `try_form_new_groups` + `try_merge_solo_groups` (node_groups/mod.rs:180-203); `filter_tasks` becomes a
//! lock-free lookup of the table published by the last tick (scheduler_impl.rs:11-110).
//!
//! The READ SURFACE the API routes call on `AppState.node_groups_plugin` (node_groups/mod.rs:324-434, :1002-1065) is
//! here with the reference's names, signatures and result types, so the routes compile unchanged against
//! `Option<Arc<GpuMatchPlugin>>` (INTEGRATION.md "The routes"): get_all_groups, get_group_by_id,
//! get_all_node_group_mappings, get_node_group, get_node_groups_batch, get_idx_in_group,
//! get_available_configurations, get_all_configuration_templates, dissolve_group.  The engine's host-side group list
//! is the store behind them (the reference reads Redis).
//!
//! Row identity.  The engine keys workers, groups and claims by ROW INDEX.  `NodeStore::get_nodes` is Redis
//! SMEMBERS order followed by a status sort (node_store.rs:195-206), so the position of a node in that Vec moves
//! whenever anybody's status changes.  The plugin therefore keeps a stable `address -> row` map: a node seen for the
//! first time is APPENDED (pm_append_workers), a known node whose projection changed is rewritten in place
//! (pm_update_workers), a node that left the store is tombstoned (PM_W_HEALTHY cleared + dissolve, never removed).
//! The order in which nodes first appear is the engine's input order (the reference's own tie-break is the
//! unspecified SMEMBERS order).
#![allow(non_camel_case_types, dead_code)]

use std::collections::{BTreeSet, HashMap};
use std::ffi::{c_char, CStr, CString};
use std::os::raw::c_void;
use std::sync::atomic::{AtomicU32, AtomicU64, Ordering};

use alloy::primitives::Address;
use anyhow::{anyhow, Error, Result};
use shared::models::node::{ComputeRequirements, ComputeSpecs};
use shared::models::task::Task;

use crate::models::node::{NodeStatus, OrchestratorNode};
use crate::plugins::node_groups::{NodeGroup, NodeGroupConfiguration};
use crate::plugins::webhook::WebhookPlugin;

pub const PM_NONE: u32 = 0xFFFF_FFFF;
pub const PM_ABI_VERSION: u32 = 3;
const PM_EINVAL: i32 = -1;
const PM_ESTATE: i32 = -4;
const PM_ERANGE: i32 = -5;

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
    pub time_proposer: u32,
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

/// include/pm_engine.h: one entry of the group life-cycle feed (kind 1 = created, 2 = destroyed)
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct pm_group_event {
    pub group_id: u64,
    pub kind: u32,
    pub config: u32,
    pub member_begin: u32,
    pub n_members: u32,
}

/// include/pm_engine.h: one group of pm_get_groups / pm_get_group_by_id / pm_get_group_of_worker
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct pm_group {
    pub id: u64,
    pub config: u32,
    pub n_members: u32,
    pub member_begin: u32,
    pub task: u32,
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
    pub ms_propose_kernel: f32, pub proposals: u32, pub propose_keys: u64,
}

#[repr(C)]
pub struct pm_group_vars {
    pub group_index: u32,
    pub group_size: u32,
    pub next_p2p_address: *const c_char,
    pub group_id: *const c_char,
    pub total_upload_count: *const c_char,
}

/// One all-gather of the multi-GPU tick (device pointers; recv = [world][bytes_per_rank]).
#[repr(C)]
#[derive(Default)]
pub struct pm_dist_xfer {
    pub send_ptr: u64,
    pub recv_ptr: u64,
    pub bytes_per_rank: u64,
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
    fn pm_append_workers(e: *mut c_void, rows: *const pm_worker_soa, first_index: *mut u32) -> i32;
    fn pm_update_workers(e: *mut c_void, idx: *const u32, rows: *const pm_worker_soa) -> i32;
    fn pm_set_addr_ranks(e: *mut c_void, ranks: *const u32, n: u32) -> i32;
    fn pm_upload_tasks(e: *mut c_void, t: *const pm_task_soa) -> i32;
    fn pm_tasks_insert_front(e: *mut c_void, rows: *const pm_task_soa) -> i32;
    fn pm_tasks_insert_front_ex(e: *mut c_void, rows: *const pm_task_soa, republish: u32) -> i32;
    fn pm_tasks_delete(e: *mut c_void, uids: *const u64, n: u32, n_deleted: *mut u32) -> i32;
    fn pm_on_worker_status(e: *mut c_void, worker: u32, flags_new: u32, dead: u32) -> i32;
    fn pm_on_worker_status_many(e: *mut c_void, workers: *const u32, flags_new: *const u32, dead: *const u32, n: u32) -> i32;
    fn pm_enable_group_events(e: *mut c_void, on: u32) -> i32;
    fn pm_drain_group_events(e: *mut c_void, events: *mut pm_group_event, cap_events: u32, members: *mut u32,
                             cap_members: u32, n_events: *mut u32, n_members: *mut u32) -> i32;
    fn pm_dissolve_group(e: *mut c_void, group_slot: u32) -> i32;
    fn pm_dissolve_group_by_id(e: *mut c_void, group_id: u64, dissolved: *mut u32) -> i32;
    fn pm_get_groups(e: *mut c_void, group_of_worker: *mut i32, groups: *mut pm_group, cap_groups: u32, n_groups: *mut u32,
                     members: *mut u32, cap_members: u32, n_members: *mut u32) -> i32;
    fn pm_get_group_by_id(e: *mut c_void, group_id: u64, out: *mut pm_group, members: *mut u32, cap_members: u32, slot: *mut u32) -> i32;
    fn pm_get_group_of_worker(e: *mut c_void, worker: u32, out: *mut pm_group, members: *mut u32, cap_members: u32, slot: *mut u32) -> i32;
    fn pm_host_config_order(cfgs: *const pm_config_row, n_cfgs: u32, enabled: u64, order_out: *mut u32, n_out: *mut u32) -> i32;
    fn pm_tick(e: *mut c_void, stats: *mut pm_stats) -> i32;
    fn pm_lookup_task_for_worker(e: *mut c_void, worker: u32, out: *mut pm_assignment) -> i32;
    fn pm_host_group_vars(input: *const c_char, v: *const pm_group_vars, out: *mut c_char, cap: usize, needed: *mut usize) -> i32;
    fn pm_host_volume_vars(input: *const c_char, group_id: *const c_char, out: *mut c_char, cap: usize, needed: *mut usize) -> i32;
    fn pm_host_build_model_table(req_models: *const *const c_char, n_rows: u32,
                                 spec_models: *const *const c_char, n_classes: u32, bits_out: *mut u32) -> i32;
    // multi-GPU (one process per GPU; the all-gathers are ncclAllGather on the stream handed to pm_set_stream)
    fn pm_set_stream(e: *mut c_void, hip_stream: *mut c_void) -> i32;
    fn pm_set_carve_workgroups(e: *mut c_void, n: u32) -> i32;
    // (several pools served by one orchestrator process: one match per engine in one call — not used by the one-pool plugin below)
    fn pm_tick_many(engines: *const *mut c_void, n: u32, stats: *mut pm_stats, flags: u32) -> i32;
    fn pm_dist_configure(e: *mut c_void, rank: u32, world: u32, shard_of_worker: *const u8) -> i32;
    fn pm_dist_tick_begin(e: *mut c_void) -> i32;
    fn pm_dist_carve_wait(e: *mut c_void) -> i32;
    fn pm_dist_match_begin(e: *mut c_void, x: *mut pm_dist_xfer) -> i32;
    fn pm_dist_tick_end(e: *mut c_void, stats: *mut pm_stats) -> i32;
}

/// RCCL (librccl.so, rccl/rccl.h): the one collective the multi-GPU tick issues
#[link(name = "rccl")]
extern "C" {
    fn ncclAllGather(sendbuff: *const c_void, recvbuff: *mut c_void, sendcount: usize, datatype: i32,
                     comm: *mut c_void, stream: *mut c_void) -> i32;
    fn ncclGetErrorString(result: i32) -> *const c_char;
}
const NCCL_UINT8: i32 = 1;      // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

/// The collective of the multi-GPU tick (include/pm_engine.h "multi-GPU"): one process per GPU, every rank holds
/// the whole swarm and runs the whole carve (replicated), the pair sweep + claim run for the OWNED workers, and the
/// published rows are all-gathered ONCE per tick.  The plugin is handed the communicator; it owns none.
pub trait AllGather: Send + Sync {
    fn rank(&self) -> u32;
    fn world(&self) -> u32;
    /// the hipStream_t the engine's kernels and the collective share (ordered without a host wait); null = the engine's own
    fn stream(&self) -> *mut c_void;
    /// device pointers; recv = [world][bytes_per_rank], send = this rank's slot of it: enqueue on stream()
    fn all_gather(&self, send: *const c_void, recv: *mut c_void, bytes_per_rank: usize) -> Result<()>;
}

/// ncclAllGather over xGMI on a communicator and a stream the host owns (`ncclComm_t`, `hipStream_t`: created by
/// `main()` with ncclCommInitRank after the launcher's rendezvous — protocol_amd/plugin/rccl_all_gather.cpp shows one
/// through an id file).  The tick's one collective moves world x cap x 32 bytes (8 ranks x 100k workers: 3.2 MB landed
/// per GPU): latency-bound, a single call on the compute stream.
pub struct RcclAllGather { pub comm: *mut c_void, pub stream: *mut c_void, pub rank: u32, pub world: u32 }
unsafe impl Send for RcclAllGather {}
unsafe impl Sync for RcclAllGather {}
impl AllGather for RcclAllGather {
    fn rank(&self) -> u32 { self.rank }
    fn world(&self) -> u32 { self.world }
    fn stream(&self) -> *mut c_void { self.stream }
    fn all_gather(&self, send: *const c_void, recv: *mut c_void, bytes_per_rank: usize) -> Result<()> {
        // (send is this rank's slot of recv: RCCL's in-place form, no staging copy)
        let rc = unsafe { ncclAllGather(send, recv, bytes_per_rank, NCCL_UINT8, self.comm, self.stream) };
        if rc == 0 { return Ok(()); }
        Err(anyhow!("ncclAllGather: {}", unsafe { CStr::from_ptr(ncclGetErrorString(rc)) }.to_string_lossy()))
    }
}

/// owner rank of a node: splitmix64 finaliser of the low 8 bytes of the address, mod world
pub fn shard_of(a: &Address, world: u32) -> u32 {
    let mut low = [0u8; 8];
    low.copy_from_slice(&a.as_slice()[12..20]);
    let mut z = u64::from_be_bytes(low).wrapping_add(0x9E37_79B9_7F4A_7C15);
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    z ^= z >> 31;
    if world == 0 { 0 } else { (z % world as u64) as u32 }
}

/// what pm_dist_configure was last told (tick_dist's; the management loop's thread only)
struct DistState { world: u32, rows: usize, stream: *mut c_void }

// worker flag bits (include/pm_engine.h)
const W_HAS_SPECS: u32 = 1 << 0; const W_HAS_GPU: u32 = 1 << 1; const W_GPU_COUNT: u32 = 1 << 2;
const W_GPU_MEM: u32 = 1 << 3; const W_GPU_MODEL: u32 = 1 << 4; const W_HAS_CPU: u32 = 1 << 5;
const W_CPU_CORES: u32 = 1 << 6; const W_RAM: u32 = 1 << 7; const W_STORAGE: u32 = 1 << 8;
const W_HEALTHY: u32 = 1 << 9; const W_HAS_P2P: u32 = 1 << 10; const W_HAS_LOC: u32 = 1 << 11;
// requirement flag bits
const R_HAS_REQ: u32 = 1 << 0; const R_CPU: u32 = 1 << 1; const R_CPU_CORES: u32 = 1 << 2;
const R_RAM: u32 = 1 << 3; const R_STORAGE: u32 = 1 << 4;
const G_COUNT: u32 = 1 << 0; const G_MODEL: u32 = 1 << 1; const G_MEM: u32 = 1 << 2; const G_MEM_MIN: u32 = 1 << 3;
const G_MEM_MAX: u32 = 1 << 4; const G_TOT_MIN: u32 = 1 << 5; const G_TOT_MAX: u32 = 1 << 6;

fn check(rc: i32) -> Result<()> {
    if rc == 0 { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(pm_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow!("pm_engine error {rc}: {msg}"))
}

/// One worker row as the engine sees it; `PartialEq` is what decides whether a known node is rewritten.
#[derive(Clone, PartialEq, Default)]
struct Row {
    flags: u32, gpu_count: u32, gpu_mem: u32, gpu_class: u32, cpu_cores: u32, ram: u32, storage: u32,
    lat: f64, lon: f64,
}

/// Columns of a batch of rows, kept alive for the duration of one FFI call.
#[derive(Default)]
struct RowColumns {
    flags: Vec<u32>, gpu_count: Vec<u32>, gpu_mem: Vec<u32>, gpu_class: Vec<u32>, cpu_cores: Vec<u32>,
    ram: Vec<u32>, storage: Vec<u32>, addr_rank: Vec<u32>, lat: Vec<f64>, lon: Vec<f64>,
}
impl RowColumns {
    fn push(&mut self, r: &Row, rank: u32) {
        self.flags.push(r.flags); self.gpu_count.push(r.gpu_count); self.gpu_mem.push(r.gpu_mem);
        self.gpu_class.push(r.gpu_class); self.cpu_cores.push(r.cpu_cores); self.ram.push(r.ram);
        self.storage.push(r.storage); self.addr_rank.push(rank); self.lat.push(r.lat); self.lon.push(r.lon);
    }
    fn soa(&self) -> pm_worker_soa {
        pm_worker_soa { n: self.flags.len() as u32, flags: self.flags.as_ptr(), gpu_count: self.gpu_count.as_ptr(),
            gpu_mem_mb: self.gpu_mem.as_ptr(), gpu_model_class: self.gpu_class.as_ptr(),
            cpu_cores: self.cpu_cores.as_ptr(), ram_mb: self.ram.as_ptr(), storage_gb: self.storage.as_ptr(),
            price: std::ptr::null(), addr_rank: self.addr_rank.as_ptr(), lat: self.lat.as_ptr(), lon: self.lon.as_ptr() }
    }
}

/// The plugin's mirror of the engine's worker table (rows never move, never disappear).
#[derive(Default)]
struct NodeTable {
    index: HashMap<Address, u32>,
    addresses: Vec<Address>,
    address_strings: Vec<String>,   // address.to_string(): GROUP_INDEX is the rank of this string (mod.rs:424-434)
    p2p_ids: Vec<String>,           // node.p2p_id.unwrap_or_default() (scheduler_impl.rs:118-128)
    rows: Vec<Row>,
    present: Vec<bool>,             // still in the node store
    by_address: Vec<u32>,           // rows in address-string order (kept sorted: a new node is one binary search)
    spec_models: Vec<String>,       // interned gpu.model strings; the index is gpu_model_class
    spec_model_index: HashMap<String, u32>,
    /// an engine call of sync_nodes failed half-way: the row map above is ahead of the engine's worker table (tombstones
    /// recorded here and never sent, rows appended here the engine does not have) — the next interval re-sends every row
    engine_rows_stale: bool,
}

pub struct GpuMatchPlugin {
    engine: *mut c_void,
    templates: Vec<NodeGroupConfiguration>,     // caller order: the engine's configuration index
    config_rows: Vec<pm_config_row>,            // what pm_set_configs was given (pm_host_config_order reads sizes + R_HAS_REQ)
    enabled_mask: AtomicU64,                    // "available_node_group_configs" as last pushed (push_enabled)
    /// NodeGroup.created_at (mod.rs:575: Utc::now() when the group forms): stamped when the creation is reported by the
    /// life-cycle feed, dropped with the group.  A LEAF lock: never held across an engine call or another lock.
    group_created_at: parking_lot::Mutex<HashMap<u64, chrono::DateTime<chrono::Utc>>>,
    dist: parking_lot::Mutex<DistState>,
    dist_rank: AtomicU32,                       // (read by emit_group_webhooks on any thread: rank 0 reports)
    config_names: Vec<String>,
    req_models: Vec<CString>,       // requirement model strings, one per pm_gpu_alt_row.model_row
    nodes: parking_lot::RwLock<NodeTable>,
    tasks: parking_lot::RwLock<Vec<Task>>,   // get_all_tasks order: the engine reports positions in this Vec
    /// `upload:<node>:<group>:*` key count (scheduler_impl.rs:130-153): stays with the Redis store
    upload_counter: Box<dyn Fn(&Address, &str) -> usize + Send + Sync>,
    /// as in NodeGroupsPlugin (mod.rs:107, 124): the plugins told about every group created / destroyed
    webhook_plugins: Option<Vec<WebhookPlugin>>,
    /// on_task_created also re-matches the standing groups (pm_tasks_insert_front_ex, republish = 1): a group that
    /// holds no task is served the new one at its next heartbeat, as in the reference (scheduler_impl.rs:33-74),
    /// instead of after the next tick.  Costs one pair sweep + publish (~0.2 ms at 100k x 10k) per created task.
    pub republish_on_insert: bool,
}

unsafe impl Send for DistState {}
unsafe impl Send for GpuMatchPlugin {}
unsafe impl Sync for GpuMatchPlugin {}

fn task_uid(t: &Task) -> u64 { t.id.as_u64_pair().1 }

impl GpuMatchPlugin {
    /// Same contract as NodeGroupsPlugin::new (node_groups/mod.rs:113-175): duplicate names or max < min panic,
    /// exactly like the reference constructor (the engine answers PM_EINVAL for the latter).
    pub fn new(templates: Vec<NodeGroupConfiguration>, device: i32,
               upload_counter: Box<dyn Fn(&Address, &str) -> usize + Send + Sync>,
               webhook_plugins: Option<Vec<WebhookPlugin>>) -> Self {
        // Hardware queues for the HIP runtime (GPU_MAX_HW_QUEUES, read once at the first HIP call of the process — the
        // pm_engine_create below; the orchestrator uses HIP for nothing else): the runtime maps a process's streams
        // onto 4 of them by default and runs two streams that share one in turn; an engine owns one stream, so a
        // process that serves several pools on one GPU needs >= 1 per pool (include/pm_engine.h,
        // pm_set_carve_workgroups).  `main()` sets it BEFORE the tokio runtime starts: writing the
        // environment from here would race with every getenv of a threaded process (and is `unsafe` in Rust 2024).
        // One pool is indifferent to the value; here it is only looked at.
        match std::env::var("GPU_MAX_HW_QUEUES").ok().and_then(|v| v.parse::<u32>().ok()) {
            Some(q) if q >= 8 => {}
            other => log::warn!("GPU_MAX_HW_QUEUES is {:?}: set it to 16 in main() before the runtime starts if this \
                                 process serves more than one pool on the GPU", other),
        }
        let mut cfg: pm_engine_config = unsafe { std::mem::zeroed() };
        unsafe { pm_engine_config_default(&mut cfg) };
        cfg.device = device;
        let mut engine = std::ptr::null_mut();
        check(unsafe { pm_engine_create(&cfg, &mut engine) }).expect("pm_engine_create");
        let mut seen = std::collections::HashSet::new();
        for t in &templates {
            if !seen.insert(t.name.clone()) { panic!("Configuration names must be unique"); }   // mod.rs:142-144
        }
        let mut this = Self { engine, templates: templates.clone(), config_rows: Vec::new(), enabled_mask: AtomicU64::new(0),
                              group_created_at: Default::default(),
                              dist: parking_lot::Mutex::new(DistState { world: 1, rows: usize::MAX, stream: std::ptr::null_mut() }),
                              dist_rank: AtomicU32::new(0),
                              config_names: templates.iter().map(|t| t.name.clone()).collect(),
                              req_models: Vec::new(), nodes: Default::default(), tasks: Default::default(),
                              upload_counter, webhook_plugins, republish_on_insert: false };
        this.set_configs(&templates);
        // an empty worker / task table, so that the delta calls have something to extend
        let empty = RowColumns::default();
        check(unsafe { pm_upload_workers(this.engine, &empty.soa(), 0) }).expect("pm_upload_workers");
        check(unsafe { pm_enable_group_events(this.engine, 1) }).expect("pm_enable_group_events");   // webhook feed
        let t = pm_task_soa { n: 0, topo_mask: std::ptr::null(), created_at: std::ptr::null(), uid: [0u64; 0].as_ptr() };
        check(unsafe { pm_upload_tasks(this.engine, &t) }).expect("pm_upload_tasks");
        this
    }

    /// NodeGroupConfiguration (mod.rs:30-37) + ComputeRequirements / GpuRequirements (shared/models/node.rs:49-70)
    /// -> pm_config_row / pm_gpu_alt_row.  The requirement strings were already parsed by
    /// `ComputeRequirements::from_str` (serde), so this is a field-by-field projection.
    fn set_configs(&mut self, templates: &[NodeGroupConfiguration]) {
        let mut rows = Vec::with_capacity(templates.len());
        let mut alts: Vec<pm_gpu_alt_row> = Vec::new();
        let opt = |v: Option<u32>, bit: u32, flags: &mut u32| -> u32 { if v.is_some() { *flags |= bit; } v.unwrap_or(0) };
        for t in templates {
            let mut row = pm_config_row { min_group_size: t.min_group_size as u32, max_group_size: t.max_group_size as u32,
                                          ..Default::default() };
            if let Some(req) = &t.compute_requirements {
                row.flags |= R_HAS_REQ;
                if let Some(cpu) = &req.cpu {
                    row.flags |= R_CPU;
                    row.cpu_cores = opt(cpu.cores, R_CPU_CORES, &mut row.flags);
                }
                row.ram_mb = opt(req.ram_mb, R_RAM, &mut row.flags);
                row.storage_gb = opt(req.storage_gb, R_STORAGE, &mut row.flags);
                row.alt_begin = alts.len() as u32;
                row.alt_count = req.gpu.len() as u32;
                for g in &req.gpu {
                    let mut a = pm_gpu_alt_row::default();
                    a.count = opt(g.count, G_COUNT, &mut a.flags);
                    a.memory_mb = opt(g.memory_mb, G_MEM, &mut a.flags);
                    a.memory_mb_min = opt(g.memory_mb_min, G_MEM_MIN, &mut a.flags);
                    a.memory_mb_max = opt(g.memory_mb_max, G_MEM_MAX, &mut a.flags);
                    a.total_memory_min = opt(g.total_memory_min, G_TOT_MIN, &mut a.flags);
                    a.total_memory_max = opt(g.total_memory_max, G_TOT_MAX, &mut a.flags);
                    if let Some(m) = &g.model {
                        a.flags |= G_MODEL;
                        a.model_row = self.req_models.len() as u32;
                        self.req_models.push(CString::new(m.as_str()).expect("model string"));
                    }
                    alts.push(a);
                }
            }
            rows.push(row);
        }
        let rc = unsafe { pm_set_configs(self.engine, rows.as_ptr(), rows.len() as u32, alts.as_ptr(), alts.len() as u32) };
        if rc == PM_EINVAL { panic!("Plugin configuration is invalid"); }                                   // mod.rs:145-147
        check(rc).expect("pm_set_configs");
        self.config_rows = rows;
        self.push_model_table(&NodeTable::default());
    }

    /// The substring rule of GpuSpecs::meets (shared/models/node.rs:463-484), evaluated once per
    /// (requirement model, interned spec model) pair by the library; re-sent whenever a new spec model shows up.
    fn push_model_table(&self, nodes: &NodeTable) {
        let spec: Vec<CString> = nodes.spec_models.iter().map(|s| CString::new(s.as_str()).unwrap()).collect();
        let spec_p: Vec<*const c_char> = spec.iter().map(|s| s.as_ptr()).collect();
        let req_p: Vec<*const c_char> = self.req_models.iter().map(|s| s.as_ptr()).collect();
        let words = (spec.len() + 31) / 32;
        let mut bits = vec![0u32; (self.req_models.len() * words).max(1)];
        check(unsafe { pm_host_build_model_table(req_p.as_ptr(), req_p.len() as u32, spec_p.as_ptr(), spec_p.len() as u32,
                                                 bits.as_mut_ptr()) }).expect("pm_host_build_model_table");
        check(unsafe { pm_set_model_table(self.engine, bits.as_ptr(), req_p.len() as u32, spec_p.len() as u32) })
            .expect("pm_set_model_table");
    }

    /// Projection of one `OrchestratorNode` (orchestrator/src/models/node.rs:11-37) into the SoA row.
    fn project(node: &OrchestratorNode, table: &mut NodeTable, new_model: &mut bool) -> Row {
        let mut r = Row::default();
        if node.status == NodeStatus::Healthy { r.flags |= W_HEALTHY; }
        if node.p2p_id.is_some() { r.flags |= W_HAS_P2P; }
        if let Some(l) = &node.location { r.flags |= W_HAS_LOC; r.lat = l.latitude; r.lon = l.longitude; }
        if let Some(s) = &node.compute_specs {
            let s: &ComputeSpecs = s;
            r.flags |= W_HAS_SPECS;
            if let Some(g) = &s.gpu {
                r.flags |= W_HAS_GPU;
                if let Some(c) = g.count { r.flags |= W_GPU_COUNT; r.gpu_count = c; }
                if let Some(m) = g.memory_mb { r.flags |= W_GPU_MEM; r.gpu_mem = m; }
                if let Some(m) = &g.model {
                    r.flags |= W_GPU_MODEL;
                    r.gpu_class = *table.spec_model_index.entry(m.clone()).or_insert_with(|| {
                        *new_model = true;
                        table.spec_models.push(m.clone());
                        (table.spec_models.len() - 1) as u32
                    });
                }
            }
            if let Some(c) = &s.cpu {
                r.flags |= W_HAS_CPU;
                if let Some(n) = c.cores { r.flags |= W_CPU_CORES; r.cpu_cores = n; }
            }
            if let Some(v) = s.ram_mb { r.flags |= W_RAM; r.ram = v; }
            if let Some(v) = s.storage_gb { r.flags |= W_STORAGE; r.storage = v; }
        }
        r
    }

    /// Called every management interval with `store_context.node_store.get_nodes()` (node_store.rs:163-209).
    /// Sends only what changed: new nodes are appended, changed rows rewritten, departed nodes tombstoned.
    pub fn sync_nodes(&self, snapshot: &[OrchestratorNode]) -> Result<()> {
        let mut t = self.nodes.write();
        let mut new_model = false;
        let mut seen = vec![false; t.rows.len()];
        let mut appended = RowColumns::default();
        let (mut upd_idx, mut updated) = (Vec::<u32>::new(), RowColumns::default());
        for node in snapshot {
            let row = Self::project(node, &mut t, &mut new_model);
            match t.index.get(&node.address).copied() {
                Some(i) => {
                    let i = i as usize;
                    if i < seen.len() { seen[i] = true; }   // (an address twice in one snapshot: the second one finds the row just appended)
                    t.p2p_ids[i] = node.p2p_id.clone().unwrap_or_default();
                    if t.rows[i] != row || !t.present[i] {
                        t.rows[i] = row.clone();
                        t.present[i] = true;
                        if i < seen.len() {
                            upd_idx.push(i as u32);
                            updated.push(&row, 0);      // ranks are replaced wholesale below when they change
                        }
                    }
                }
                None => {
                    let i = t.rows.len() as u32;
                    t.index.insert(node.address, i);
                    t.addresses.push(node.address);
                    t.address_strings.push(node.address.to_string());
                    t.p2p_ids.push(node.p2p_id.clone().unwrap_or_default());
                    t.rows.push(row.clone());
                    t.present.push(true);
                    let key = &t.address_strings[i as usize];
                    let at = t.by_address.partition_point(|&j| t.address_strings[j as usize] < *key);
                    t.by_address.insert(at, i);
                    appended.push(&row, i);
                }
            }
        }
        // The row map above is the plugin's truth from here on; the engine calls below bring the engine's worker table
        // to it.  If one of them fails the two have diverged: the next interval then re-sends the whole table.
        if t.engine_rows_stale {
            // Every row again, in row order — the order the engine has its own in, with the rows it never received
            // behind them: pm_upload_workers(keep_groups = 1) keeps the standing groups, their claims and the id stream
            // (rows never move, so a group's row indices are as valid as before).  Then the deaths the engine may have
            // missed: every row that is not in the store, as dead — a no-op for a row in no group, the reference's
            // dissolution (and its send_group_destroyed) for the others.
            let ranks = Self::address_ranks(&t.by_address, t.rows.len());
            for i in 0..seen.len() { if !seen[i] { t.present[i] = false; } }
            let mut all = RowColumns::default();
            let (mut gone, mut gone_flags) = (Vec::<u32>::new(), Vec::<u32>::new());
            for i in 0..t.rows.len() {
                if !t.present[i] {
                    t.rows[i].flags &= !W_HEALTHY;
                    gone.push(i as u32);
                    gone_flags.push(t.rows[i].flags);
                }
                let row = t.rows[i].clone();
                all.push(&row, ranks[i]);
            }
            self.push_model_table(&t);
            check(unsafe { pm_upload_workers(self.engine, &all.soa(), 1) })?;
            if !gone.is_empty() {
                let dead = vec![1u32; gone.len()];
                check(unsafe { pm_on_worker_status_many(self.engine, gone.as_ptr(), gone_flags.as_ptr(), dead.as_ptr(), gone.len() as u32) })?;
            }
            t.engine_rows_stale = false;
            drop(t);
            return self.emit_group_webhooks();
        }
        t.engine_rows_stale = true;     // (cleared behind the last engine call below: every `?` in between leaves it set)
        if new_model { self.push_model_table(&t); }
        // nodes that left the store: tombstone (their group dissolves, like a death; status_update_impl.rs:17-29)
        let (mut gone, mut gone_flags) = (Vec::<u32>::new(), Vec::<u32>::new());
        for i in 0..seen.len() {
            if !seen[i] && t.present[i] {
                t.present[i] = false;
                t.rows[i].flags &= !W_HEALTHY;
                gone.push(i as u32);
                gone_flags.push(t.rows[i].flags);
            }
        }
        if !gone.is_empty() {
            let dead = vec![1u32; gone.len()];
            check(unsafe { pm_on_worker_status_many(self.engine, gone.as_ptr(), gone_flags.as_ptr(), dead.as_ptr(), gone.len() as u32) })?;
        }
        if !upd_idx.is_empty() {
            // keep the ranks the engine already has for rewritten rows (ranks among the rows it knows: the new
            // ones of this snapshot are sent behind this call)
            let ranks = Self::address_ranks(&t.by_address, seen.len());
            for (k, &i) in upd_idx.iter().enumerate() { updated.addr_rank[k] = ranks[i as usize]; }
            check(unsafe { pm_update_workers(self.engine, upd_idx.as_ptr(), &updated.soa()) })?;
        }
        if !appended.flags.is_empty() {
            // (a row appended by this very snapshot may have been rewritten by a later entry of it: send what it is now)
            for k in 0..appended.flags.len() {
                let r = t.rows[seen.len() + k].clone();
                appended.flags[k] = r.flags; appended.gpu_count[k] = r.gpu_count; appended.gpu_mem[k] = r.gpu_mem;
                appended.gpu_class[k] = r.gpu_class; appended.cpu_cores[k] = r.cpu_cores; appended.ram[k] = r.ram;
                appended.storage[k] = r.storage; appended.lat[k] = r.lat; appended.lon[k] = r.lon;
            }
            let mut first = 0u32;
            check(unsafe { pm_append_workers(self.engine, &appended.soa(), &mut first) })?;
            if first as usize != seen.len() {
                return Err(anyhow!("pm_engine error {PM_ESTATE}: the engine's worker table and the plugin's row map disagree"));
            }
            // a new address shifts the global ranks of the others: GROUP_INDEX only needs the relative order
            let ranks = Self::address_ranks(&t.by_address, t.rows.len());
            check(unsafe { pm_set_addr_ranks(self.engine, ranks.as_ptr(), ranks.len() as u32) })?;
        }
        t.engine_rows_stale = false;
        drop(t);
        self.emit_group_webhooks()      // tombstoned nodes dissolved their groups
    }

    /// rank of address.to_string() in byte order (BTreeSet<String>, mod.rs:424-434) among the first `known` rows:
    /// one pass over the sorted row list (no string is compared here; the list is kept sorted as nodes arrive)
    fn address_ranks(by_address: &[u32], known: usize) -> Vec<u32> {
        let mut rank = vec![0u32; known];
        let mut r = 0u32;
        for &i in by_address {
            if (i as usize) < known { rank[i as usize] = r; r += 1; }
        }
        rank
    }

    /// scheduler_impl.rs:44-59: any None on the way => every configuration allowed.
    fn topology_mask(&self, t: &Task) -> u64 {
        let Some(list) = t.scheduling_config.as_ref().and_then(|c| c.plugins.as_ref())
            .and_then(|p| p.get("node_groups")).and_then(|n| n.get("allowed_topologies")) else { return u64::MAX };
        list.iter().filter_map(|name| self.config_names.iter().position(|c| c == name)).fold(0, |m, i| m | (1u64 << i))
    }

    fn push_enabled(&self, tasks: &[Task]) -> Result<()> {
        // available_node_group_configs: every topology some task names (on_task_created, mod.rs:1224-1243)
        let enabled = tasks.iter().map(|t| self.topology_mask(t)).filter(|m| *m != u64::MAX).fold(0u64, |a, m| a | m);
        check(unsafe { pm_set_enabled_mask(self.engine, enabled) })?;
        self.enabled_mask.store(enabled, Ordering::Release);
        Ok(())
    }

    // LOCK ORDER: `nodes`, then `tasks`, then the engine's own mutex (taken inside every pm_* call but the look-up).
    // The engine reports a task as a POSITION in `tasks`, and it re-derives the published positions inside
    // pm_tasks_insert_front / pm_tasks_delete / pm_upload_tasks — so the Vec and the engine's table change under ONE
    // write lock, and filter_tasks holds the read lock from the look-up to the index: a heartbeat never pairs a
    // position of the new table with the old Vec (or the other way round).  (The reference binds group -> task by id,
    // scheduler_impl.rs:62-82, and has no such window; this is what keeps the shim from having one.)

    /// the snapshot upload with `tasks` already locked for writing
    fn sync_tasks_locked(&self, guard: &mut Vec<Task>, tasks: Vec<Task>) -> Result<()> {
        let masks: Vec<u64> = tasks.iter().map(|t| self.topology_mask(t)).collect();
        let created: Vec<i64> = tasks.iter().map(|t| t.created_at).collect();
        let uid: Vec<u64> = tasks.iter().map(task_uid).collect();
        let soa = pm_task_soa { n: tasks.len() as u32, topo_mask: masks.as_ptr(), created_at: created.as_ptr(), uid: uid.as_ptr() };
        check(unsafe { pm_upload_tasks(self.engine, &soa) })?;
        self.push_enabled(&tasks)?;
        *guard = tasks;
        Ok(())
    }

    /// Full snapshot: `task_store.get_all_tasks()` (task_store.rs:57-82, already created_at-desc).  Start-up and
    /// the fallback when a delta does not apply.
    pub fn sync_tasks(&self, tasks: Vec<Task>) -> Result<()> {
        let mut guard = self.tasks.write();
        self.sync_tasks_locked(&mut guard, tasks)
    }

    /// TaskStore observer (task_store.rs:46-52 -> on_task_created, mod.rs:1224-1243): the new task is the newest, so
    /// it goes in front; only this row travels to the GPU.  Equal or older timestamps fall back to the snapshot.
    pub fn on_task_created(&self, task: &Task, all_tasks: impl FnOnce() -> Vec<Task>) -> Result<()> {
        let (mask, created, uid) = (self.topology_mask(task), task.created_at, task_uid(task));
        let soa = pm_task_soa { n: 1, topo_mask: &mask, created_at: &created, uid: &uid };
        let mut tasks = self.tasks.write();                 // (before the engine call: see LOCK ORDER)
        let rc = if self.republish_on_insert { unsafe { pm_tasks_insert_front_ex(self.engine, &soa, 1) } }
                 else { unsafe { pm_tasks_insert_front(self.engine, &soa) } };
        if rc != 0 { return self.sync_tasks_locked(&mut tasks, all_tasks()); }
        tasks.insert(0, task.clone());
        self.push_enabled(&tasks)
    }

    /// on_task_deleted (mod.rs:1245-1325): the engine dissolves every group that had claimed the task.
    pub fn on_task_deleted(&self, task: &Task) -> Result<()> {
        let uid = task_uid(task);
        let mut n = 0u32;
        let mut tasks = self.tasks.write();                 // (before the engine call: see LOCK ORDER)
        check(unsafe { pm_tasks_delete(self.engine, &uid, 1, &mut n) })?;
        tasks.retain(|t| t.id != task.id);
        self.push_enabled(&tasks)?;
        drop(tasks);
        self.emit_group_webhooks()      // dissolve_group's send_group_destroyed, mod.rs:1469-1481
    }

    /// One body of run_group_management_loop (mod.rs:180-203) + every worker's filter_tasks, then the webhooks the
    /// reference sends from inside try_form_new_groups / execute_group_merge (mod.rs:612-625, 974-1000).
    pub fn tick(&self) -> Result<pm_stats> {
        let mut s = pm_stats::default();
        check(unsafe { pm_tick(self.engine, &mut s) })?;
        self.emit_group_webhooks()?;
        Ok(s)
    }

    /// The management interval of ONE pool matched by several GPUs, one process per GPU: this plugin is rank comm.rank()
    /// of comm.world().  Every rank is fed every store event (sync_nodes, the task observers, status changes — replicated
    /// calls) and calls tick_dist at the same point of its loop; every rank ends with the identical groups and the full
    /// published table (any rank answers any heartbeat).  The five calls of INTEGRATION.md "Multi-GPU" around ONE
    /// all-gather.  From its first tick_dist on a plugin of rank > 0 delivers no webhooks (rank 0 reports them).
    pub fn tick_dist(&self, comm: &dyn AllGather) -> Result<pm_stats> {
        let (rank, world) = (comm.rank(), comm.world());
        if world == 0 || rank >= world { return Err(anyhow!("tick_dist: rank outside the communicator")); }
        {
            let t = self.nodes.read();
            let mut d = self.dist.lock();
            if d.stream != comm.stream() {
                // the engine's kernels and the collective on one stream: no host wait between them
                check(unsafe { pm_set_stream(self.engine, comm.stream()) })?;
                d.stream = comm.stream();
            }
            if rank != self.dist_rank.load(Ordering::Acquire) || world != d.world || t.rows.len() != d.rows {
                // ownership is per row and every rank computes the same: a hash of the address, nothing is negotiated
                let shard: Vec<u8> = t.addresses.iter().map(|a| shard_of(a, world) as u8).collect();
                check(unsafe { pm_dist_configure(self.engine, rank, world, if shard.is_empty() { std::ptr::null() } else { shard.as_ptr() }) })?;
                self.dist_rank.store(rank, Ordering::Release);
                d.world = world;
                d.rows = if t.engine_rows_stale { usize::MAX } else { t.rows.len() };  // (the engine is behind the row map: configure again next time)
            }
        }
        let (mut s, mut x) = (pm_stats::default(), pm_dist_xfer::default());
        check(unsafe { pm_dist_tick_begin(self.engine) })?;      // compat sweep; the whole carve is started (replicated)
        check(unsafe { pm_dist_carve_wait(self.engine) })?;      // waits for it; near-ties settled on the host, identically everywhere
        check(unsafe { pm_dist_match_begin(self.engine, &mut x) })?;   // solo merge, pair sweep + claim of the OWNED workers
        if x.bytes_per_rank != 0 {                               // the ONE exchange of a tick: the published rows
            comm.all_gather(x.send_ptr as usize as *const c_void, x.recv_ptr as usize as *mut c_void, x.bytes_per_rank as usize)?;
        }
        check(unsafe { pm_dist_tick_end(self.engine, &mut s) })?;   // scatter into the full table, publish
        self.emit_group_webhooks()?;
        Ok(s)
    }

    /// Drains the engine's group life-cycle feed into send_group_created / send_group_destroyed, in the order the
    /// reference emits them.  Runs after everything that can create or dissolve groups: tick, handle_status_change,
    /// on_task_deleted, sync_nodes (tombstones).  (pm_enable_group_events(1) in `new`.)
    fn emit_group_webhooks(&self) -> Result<()> {
        let (mut ne, mut nm) = (0u32, 0u32);
        let rc = unsafe { pm_drain_group_events(self.engine, std::ptr::null_mut(), 0, std::ptr::null_mut(), 0, &mut ne, &mut nm) };
        if rc == 0 { return Ok(()); }                       // empty log
        // (another thread may log events between the size query and the drain: grow and try again; a drain that
        // still fails is logged and left for the next call — it is not the tick that failed)
        let (mut events, mut members) = (Vec::new(), Vec::new());
        for _ in 0..8 {
            events.resize(ne as usize, pm_group_event::default());
            members.resize(nm as usize, 0u32);
            let (cap_e, cap_m) = (ne, nm);
            let rc = unsafe { pm_drain_group_events(self.engine, events.as_mut_ptr(), cap_e, members.as_mut_ptr(), cap_m, &mut ne, &mut nm) };
            if rc == 0 { break; }
            if rc != PM_ERANGE { log::error!("pm_drain_group_events: {rc}"); return Ok(()); }
        }
        if events.len() < ne as usize { log::error!("group events kept for the next drain"); return Ok(()); }
        {   // NodeGroup.created_at (mod.rs:575): the clock at the report of the creation
            let now = chrono::Utc::now();
            let mut stamps = self.group_created_at.lock();
            for ev in &events[..ne as usize] {
                if ev.kind == 1 { stamps.insert(ev.group_id, now); } else { stamps.remove(&ev.group_id); }
            }
        }
        let Some(plugins) = &self.webhook_plugins else { return Ok(()) };
        if self.dist_rank.load(Ordering::Acquire) != 0 { return Ok(()); }     // a rank > 0 of a multi-GPU pool: rank 0 reports
        let t = self.nodes.read();
        for ev in &events[..ne as usize] {
            let id = format!("{:x}", ev.group_id);                                   // generate_group_id, mod.rs:1489-1493
            let name = self.config_names[ev.config as usize].clone();
            let nodes: Vec<String> = members[ev.member_begin as usize..(ev.member_begin + ev.n_members) as usize]
                .iter().map(|&w| t.address_strings[w as usize].clone()).collect();  // group.nodes order
            for p in plugins {
                let r = if ev.kind == 1 { p.send_group_created(id.clone(), name.clone(), nodes.clone()) }
                        else { p.send_group_destroyed(id.clone(), name.clone(), nodes.clone()) };
                if let Err(e) = r { log::error!("Failed to send group webhook: {e}"); }   // as in the reference: logged, not fatal
            }
        }
        Ok(())
    }

    fn render(f: impl Fn(*mut c_char, usize, *mut usize) -> i32) -> Result<String> {
        let mut need = 0usize;
        check(f(std::ptr::null_mut(), 0, &mut need))?;
        let mut buf = vec![0u8; need];
        check(f(buf.as_mut_ptr() as *mut c_char, need, &mut need))?;
        buf.pop();
        Ok(String::from_utf8(buf)?)
    }

    /// (The positions pm_lookup_task_for_worker reports index `tasks` as it is NOW: the engine re-derives the published
    /// positions inside pm_tasks_insert_front / pm_tasks_delete / pm_upload_tasks, and clears the rows of a group that a
    /// deleted task or a dead node dissolved — no tick needed in between.)
    /// SchedulerPlugin::filter_tasks (plugins/mod.rs:66-78): lock-free lookup + the `${...}` templating the
    /// reference does at scheduler_impl.rs:112-205 (GROUP_INDEX, GROUP_SIZE, NEXT_P2P_ADDRESS, GROUP_ID, upload count).
    /// `_tasks` is ignored — and with the edit of `Scheduler::get_task_for_node` shown in INTEGRATION.md ("The task list
    /// per heartbeat") the scheduler does not even load it: the plugin serves from its own copy, kept current by the
    /// task observers.
    pub(crate) fn filter_tasks(&self, _tasks: &[Task], node_address: &Address) -> Result<Vec<Task>> {
        let nodes = self.nodes.read();
        let Some(&w) = nodes.index.get(node_address) else { return Ok(vec![]) };
        let mut a = pm_assignment::default();
        let tasks = self.tasks.read();                      // held from the look-up to the index (see LOCK ORDER)
        if unsafe { pm_lookup_task_for_worker(self.engine, w, &mut a) } != 0 || a.task == PM_NONE {
            return Ok(vec![]);
        }
        let Some(mut task) = tasks.get(a.task as usize).cloned() else { return Ok(vec![]) };
        drop(tasks);
        let group_id = format!("{:x}", a.group_id);                                   // generate_group_id, mod.rs:1489-1493
        let gid = CString::new(group_id.as_str())?;
        let next = CString::new(nodes.p2p_ids.get(a.next_worker as usize).cloned().unwrap_or_default())?;
        let count = CString::new((self.upload_counter)(node_address, &group_id).to_string())?;
        let vars = pm_group_vars { group_index: a.group_index, group_size: a.group_size,
            next_p2p_address: next.as_ptr(), group_id: gid.as_ptr(), total_upload_count: count.as_ptr() };
        let group_vars = |s: &str| -> Result<String> {
            let cin = CString::new(s)?;
            Self::render(|o, c, n| unsafe { pm_host_group_vars(cin.as_ptr(), &vars, o, c, n) })
        };
        let env = task.env_vars.get_or_insert_with(Default::default);
        env.insert("GROUP_INDEX".to_string(), a.group_index.to_string());             // scheduler_impl.rs:161
        for (_, v) in env.iter_mut() { *v = group_vars(v)?; }
        if let Some(cmd) = task.cmd.as_mut() { for arg in cmd.iter_mut() { *arg = group_vars(arg)?; } }
        if let Some(mounts) = task.volume_mounts.as_mut() {                           // scheduler_impl.rs:185-200
            for m in mounts.iter_mut() {
                for path in [&mut m.host_path, &mut m.container_path] {
                    let cin = CString::new(path.as_str())?;
                    *path = Self::render(|o, c, n| unsafe { pm_host_volume_vars(cin.as_ptr(), gid.as_ptr(), o, c, n) })?;
                }
            }
        }
        Ok(vec![task])
    }

    // ------------------------------------------------------------------------------------------------ the read surface
    // What the API routes call on AppState.node_groups_plugin, with the reference's names and result types
    // (node_groups/mod.rs:324-434, :1002-1065).  `async` only because the reference's are (the routes `.await` them):
    // nothing here waits for anything but the engine's mutex.

    /// one snapshot of the engine's group list: (group_of per row or empty, groups, members)
    fn snapshot_groups(&self, rows: usize, want_group_of: bool) -> Result<(Vec<i32>, Vec<pm_group>, Vec<u32>)> {
        for _ in 0..8 {      // (a tick between the size query and the copy: ask again)
            let (mut ng, mut nm) = (0u32, 0u32);
            check(unsafe { pm_get_groups(self.engine, std::ptr::null_mut(), std::ptr::null_mut(), 0, &mut ng, std::ptr::null_mut(), 0, &mut nm) })?;
            let mut groups = vec![pm_group::default(); ng as usize];
            let mut members = vec![0u32; nm as usize];
            // (group_of_worker gets W entries, W the ENGINE's row count: never more than the plugin's, whose lock the caller holds)
            let mut group_of = if want_group_of { vec![-1i32; rows] } else { Vec::new() };
            let gp = if want_group_of && rows > 0 { group_of.as_mut_ptr() } else { std::ptr::null_mut() };
            let rc = unsafe { pm_get_groups(self.engine, gp, groups.as_mut_ptr(), ng, &mut ng, members.as_mut_ptr(), nm, &mut nm) };
            if rc == 0 {
                groups.truncate(ng as usize);
                members.truncate(nm as usize);
                return Ok((group_of, groups, members));
            }
            if rc != PM_ERANGE { check(rc)?; }
        }
        Err(anyhow!("the group list kept changing under pm_get_groups"))
    }

    fn make_group(&self, t: &NodeTable, g: &pm_group, members: &[u32]) -> NodeGroup {
        let created_at = self.group_created_at.lock().get(&g.id).copied().unwrap_or_else(chrono::Utc::now);
        NodeGroup {
            id: format!("{:x}", g.id),                                                  // generate_group_id, mod.rs:1489-1493
            nodes: members.iter().map(|&w| t.address_strings[w as usize].clone()).collect::<BTreeSet<String>>(),
            created_at,
            configuration_name: self.config_names[g.config as usize].clone(),
        }
    }

    /// the inverse of format!("{:x}", u64): lower-case hex, no sign / prefix / leading zero (but "0"), <= 16 digits.
    /// Anything else is the text of no group id (a Redis key that does not exist in the reference).
    fn parse_group_id(s: &str) -> Option<u64> {
        if s.is_empty() || s.len() > 16 || (s.len() > 1 && s.starts_with('0')) { return None; }
        if !s.bytes().all(|c| c.is_ascii_digit() || (b'a'..=b'f').contains(&c)) { return None; }
        u64::from_str_radix(s, 16).ok()
    }

    /// the reference keys node_to_group by address TEXT: exact string match (binary search in the address-ordered rows)
    fn row_of_address_text(t: &NodeTable, text: &str) -> Option<u32> {
        let at = t.by_address.partition_point(|&j| t.address_strings[j as usize].as_str() < text);
        t.by_address.get(at).copied().filter(|&j| t.address_strings[j as usize] == text)
    }

    fn one_group(&self, t: &NodeTable, call: impl Fn(*mut pm_group, *mut u32, u32, *mut u32) -> i32) -> Result<Option<NodeGroup>> {
        let (mut g, mut slot) = (pm_group::default(), PM_NONE);
        let mut members = vec![0u32; 64];
        let mut rc = call(&mut g, members.as_mut_ptr(), members.len() as u32, &mut slot);
        if rc == PM_ERANGE && slot != PM_NONE {          // a group of more than 64 nodes: its size is in the record
            members.resize(g.n_members as usize, 0);
            rc = call(&mut g, members.as_mut_ptr(), members.len() as u32, &mut slot);
        }
        if rc == PM_ERANGE && slot == PM_NONE { return Ok(None); }   // (a row the engine has not been sent yet: in no group)
        check(rc)?;
        if slot == PM_NONE { return Ok(None); }
        Ok(Some(self.make_group(t, &g, &members[..g.n_members as usize])))
    }

    /// get_all_groups (mod.rs:1006-1044): every group, sorted by id text (:1040)
    pub(crate) async fn get_all_groups(&self) -> Result<Vec<NodeGroup>, Error> {
        let t = self.nodes.read();
        let (_, groups, members) = self.snapshot_groups(t.rows.len(), false)?;
        let mut out: Vec<NodeGroup> = groups.iter()
            .map(|g| self.make_group(&t, g, &members[g.member_begin as usize..(g.member_begin + g.n_members) as usize])).collect();
        out.sort_by(|a, b| a.id.cmp(&b.id));
        Ok(out)
    }

    /// get_group_by_id (mod.rs:1046-1055)
    pub(crate) async fn get_group_by_id(&self, group_id: &str) -> Result<Option<NodeGroup>, Error> {
        let Some(id) = Self::parse_group_id(group_id) else { return Ok(None) };
        let t = self.nodes.read();
        self.one_group(&t, |g, m, cap, slot| unsafe { pm_get_group_by_id(self.engine, id, g, m, cap, slot) })
    }

    /// get_all_node_group_mappings (mod.rs:1057-1065): node address text -> group id text
    pub(crate) async fn get_all_node_group_mappings(&self) -> Result<HashMap<String, String>, Error> {
        let t = self.nodes.read();
        let (_, groups, members) = self.snapshot_groups(t.rows.len(), false)?;
        let mut out = HashMap::new();
        for g in &groups {
            let id = format!("{:x}", g.id);
            for &w in &members[g.member_begin as usize..(g.member_begin + g.n_members) as usize] {
                out.insert(t.address_strings[w as usize].clone(), id.clone());
            }
        }
        Ok(out)
    }

    /// get_node_group (mod.rs:324-337)
    pub async fn get_node_group(&self, node_addr: &str) -> Result<Option<NodeGroup>, Error> {
        let t = self.nodes.read();
        let Some(row) = Self::row_of_address_text(&t, node_addr) else { return Ok(None) };
        self.one_group(&t, |g, m, cap, slot| unsafe { pm_get_group_of_worker(self.engine, row, g, m, cap, slot) })
    }

    /// get_node_groups_batch (mod.rs:339-397): every asked address is a key of the result; one snapshot of the list
    pub async fn get_node_groups_batch(&self, node_addresses: &[String]) -> Result<HashMap<String, Option<NodeGroup>>, Error> {
        let mut result = HashMap::new();
        if node_addresses.is_empty() { return Ok(result); }                             // mod.rs:346-348
        let t = self.nodes.read();
        let (group_of, groups, members) = self.snapshot_groups(t.rows.len(), true)?;
        let mut made: HashMap<i32, NodeGroup> = HashMap::new();      // every group is built once (the reference MGETs the unique ids)
        for a in node_addresses {
            let gi = Self::row_of_address_text(&t, a).and_then(|r| group_of.get(r as usize).copied()).filter(|&g| g >= 0);
            let group = gi.map(|gi| made.entry(gi).or_insert_with(|| {
                let g = &groups[gi as usize];
                self.make_group(&t, g, &members[g.member_begin as usize..(g.member_begin + g.n_members) as usize])
            }).clone());
            result.insert(a.clone(), group);
        }
        Ok(result)
    }

    /// get_idx_in_group (mod.rs:424-434)
    pub fn get_idx_in_group(&self, node_group: &NodeGroup, node_addr: &str) -> Result<usize, Error> {
        node_group.nodes.iter().position(|n| n == node_addr).ok_or_else(|| anyhow!("Node {} not found in group", node_addr))
    }

    fn ordered_templates(&self, enabled: u64) -> Vec<NodeGroupConfiguration> {
        let mut order = vec![0u32; self.config_rows.len() + 1];
        let mut n = 0u32;
        let rc = unsafe { pm_host_config_order(self.config_rows.as_ptr(), self.config_rows.len() as u32, enabled, order.as_mut_ptr(), &mut n) };
        if rc != 0 { return vec![]; }                                                   // (the reference answers vec![] on a store error, mod.rs:400-402)
        order[..n as usize].iter().map(|&c| self.templates[c as usize].clone()).collect()
    }

    /// get_available_configurations (mod.rs:399-418): the templates some task names, min_group_size descending (stable)
    pub async fn get_available_configurations(&self) -> Vec<NodeGroupConfiguration> {
        self.ordered_templates(self.enabled_mask.load(Ordering::Acquire))
    }

    /// get_all_configuration_templates (mod.rs:420-422): in the constructor's order (mod.rs:150-164) = the carve order
    /// with nothing disabled
    pub fn get_all_configuration_templates(&self) -> Vec<NodeGroupConfiguration> {
        let c = self.config_rows.len();
        self.ordered_templates(if c >= 64 { u64::MAX } else { (1u64 << c) - 1 })
    }

    /// dissolve_group (mod.rs:1002-1004 -> :1423-1487): by id text; an id that names no group is Ok(()) ("No group found")
    pub(crate) async fn dissolve_group(&self, group_id: &str) -> Result<(), Error> {
        let Some(id) = Self::parse_group_id(group_id) else { return Ok(()) };
        let mut dissolved = 0u32;
        check(unsafe { pm_dissolve_group_by_id(self.engine, id, &mut dissolved) })?;
        if dissolved != 0 { self.emit_group_webhooks()?; }                              // send_group_destroyed, mod.rs:1469-1481
        Ok(())
    }

    /// StatusUpdatePlugin::handle_status_change (status_update_impl.rs:8-39).
    pub(crate) fn handle_status_change(&self, node: &OrchestratorNode) -> Result<()> {
        let mut t = self.nodes.write();
        let Some(&w) = t.index.get(&node.address) else { return Ok(()) };
        let mut flags = t.rows[w as usize].flags & !W_HEALTHY;
        if node.status == NodeStatus::Healthy { flags |= W_HEALTHY; }
        t.rows[w as usize].flags = flags;
        let dead = matches!(node.status, NodeStatus::Dead | NodeStatus::LowBalance) as u32;
        check(unsafe { pm_on_worker_status(self.engine, w, flags, dead) })?;
        drop(t);
        if dead != 0 { self.emit_group_webhooks()?; }      // the whole group was dissolved (status_update_impl.rs:17-29)
        Ok(())
    }
}

impl Drop for GpuMatchPlugin {
    fn drop(&mut self) { unsafe { pm_engine_destroy(self.engine) } }
}
