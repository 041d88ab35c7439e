// pm_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the matching engine.
//
// Reference regions each kernel takes over (paths relative to /root/reference/crates):
//   compat_kernel        shared/src/models/node.rs:377-541 (ComputeSpecs::meets, GpuSpecs::meets,
//                        CpuSpecs::meets) under orchestrator/src/plugins/node_groups/mod.rs:206-215
//   pair_sweep_*         orchestrator/src/plugins/node_groups/scheduler_impl.rs:42-61 /
//                        mod.rs:1134-1162 (topology filter), one row per heartbeat
//   newest_kernel        orchestrator/src/plugins/newest_task/mod.rs:8-19
//   carve_kernel         orchestrator/src/plugins/node_groups/mod.rs:478-628 (try_form_new_groups)
//                        with :218-255 (Haversine proximity) and, in MERGE mode, the selection half
//                        of :752-860 (attempt_group_merge)
//
// Wave = 64 lanes everywhere.  No MFMA: this is integer scan / compare / reduce work plus one f64
// Haversine term on the VALU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pm_device.h"

namespace pm {

// ------------------------------------------------------------------------------------------------
// wave / block reduction helpers (64-lane butterflies; results valid in every lane)

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor(v, o, 64));
  return v;
}
struct KeyIdx {
  uint64_t k;
  uint32_t i;
};
__device__ __forceinline__ bool ki_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
  return ka < kb || (ka == kb && ia < ib);
}
__device__ __forceinline__ KeyIdx wave_min_ki(KeyIdx v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t lo = __shfl_xor((uint32_t)v.k, o, 64);
    uint32_t hi = __shfl_xor((uint32_t)(v.k >> 32), o, 64);
    uint32_t oi = __shfl_xor(v.i, o, 64);
    uint64_t ok = ((uint64_t)hi << 32) | lo;
    if (ki_less(ok, oi, v.k, v.i)) {
      v.k = ok;
      v.i = oi;
    }
  }
  return v;
}

// ------------------------------------------------------------------------------------------------
// Phase A: W x C compat sweep.  One worker per lane, eight coalesced u32 column loads, config and
// alternative rows are wave-uniform (scalar loads), the model rule is one bit of a host-built table.

__device__ __forceinline__ bool gpu_alt_meets(uint32_t wf, uint32_t wcount, uint32_t wmem, uint32_t wcls,
                                              const pm_gpu_alt_row& a, const uint32_t* __restrict__ model_bits,
                                              uint32_t words) {
  // GpuSpecs::meets, shared/src/models/node.rs:445-526
  if (a.flags & PM_G_COUNT) {  // :447-461 — equality; spec None passes only for a required 0
    if (!(wf & PM_W_GPU_COUNT)) {
      if (a.count > 0) return false;
    } else if (wcount != a.count) {
      return false;
    }
  }
  if (a.flags & PM_G_MODEL) {  // :463-484, evaluated on the host into model_bits
    if (!(wf & PM_W_GPU_MODEL)) return false;
    uint32_t word = model_bits[a.model_row * words + (wcls >> 5)];
    if (!((word >> (wcls & 31)) & 1u)) return false;
  }
  const bool mem_some = (wf & PM_W_GPU_MEM) != 0;
  if ((a.flags & PM_G_MEM) && (!mem_some || wmem < a.memory_mb)) return false;          // :487-491
  if ((a.flags & PM_G_MEM_MIN) && (!mem_some || wmem < a.memory_mb_min)) return false;  // :494-498
  if ((a.flags & PM_G_MEM_MAX) && (!mem_some || wmem > a.memory_mb_max)) return false;  // :499-503
  if ((wf & PM_W_GPU_COUNT) && mem_some) {  // :506-522 — skipped when count or memory is None
    const uint32_t total = wcount * wmem;   // u32 wrapping multiply, as in a release build
    if ((a.flags & PM_G_TOT_MIN) && total < a.total_memory_min) return false;
    if ((a.flags & PM_G_TOT_MAX) && total > a.total_memory_max) return false;
  }
  return true;
}

__global__ __launch_bounds__(256) void compat_kernel(CompatArgs p) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w >= p.W) return;
  const uint32_t wf = p.flags[w];
  const uint32_t wcount = p.gpu_count[w], wmem = p.gpu_mem[w], wcls = p.gpu_cls[w];
  const uint32_t wcores = p.cpu_cores[w], wram = p.ram[w], wsto = p.storage[w];
  uint64_t mask = 0;
  for (uint32_t c = 0; c < p.n_cfgs; ++c) {
    const pm_config_row cfg = p.cfgs[c];  // uniform -> SGPRs
    bool ok;
    if (!(cfg.flags & PM_R_HAS_REQ)) {
      ok = true;  // (None, _) => true, mod.rs:211
    } else if (!(wf & PM_W_HAS_SPECS)) {
      ok = false;  // (Some, None) => false, mod.rs:212
    } else {
      ok = true;  // ComputeSpecs::meets, node.rs:379-440
      if (cfg.flags & PM_R_CPU) {  // :381-393 + CpuSpecs::meets :531-540
        if (!(wf & PM_W_HAS_CPU)) ok = false;
        if ((cfg.flags & PM_R_CPU_CORES) && (!(wf & PM_W_CPU_CORES) || wcores < cfg.cpu_cores)) ok = false;
      }
      if ((cfg.flags & PM_R_RAM) && (!(wf & PM_W_RAM) || wram < cfg.ram_mb)) ok = false;              // :396-404
      if ((cfg.flags & PM_R_STORAGE) && (!(wf & PM_W_STORAGE) || wsto < cfg.storage_gb)) ok = false;  // :407-418
      if (cfg.alt_count) {  // :420-435 — OR over alternatives
        bool any = false;
        if (wf & PM_W_HAS_GPU)
          for (uint32_t k = 0; k < cfg.alt_count; ++k)
            any |= gpu_alt_meets(wf, wcount, wmem, wcls, p.alts[cfg.alt_begin + k], p.model_bits, p.model_words);
        ok = ok && any;
      }
    }
    mask |= (uint64_t)ok << c;
  }
  p.compat[w] = mask;
}

// cos(lat * pi/180) per worker, consumed by the Haversine term of the carve kernel.
__global__ __launch_bounds__(256) void coslat_kernel(const double* __restrict__ lat, double* __restrict__ coslat,
                                                     uint32_t W) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w < W) coslat[w] = cos(lat[w] * PM_RAD);
}

// ------------------------------------------------------------------------------------------------
// Phase B, scalar kernel (sweep_variant 1): rows live in lanes, the swept axis is wave-uniform.
// Every (row, col) pair is evaluated: hit = (row_sel & col_mask) != 0; folds count and first hit.
// grid = (ceil(R/256), n_chunks); partials are combined by pair_finalize_kernel.

__global__ __launch_bounds__(256) void pair_sweep_scalar_kernel(const uint64_t* __restrict__ row_sel, uint32_t R,
                                                                const uint64_t* __restrict__ col_mask,
                                                                uint32_t n_cols, uint32_t chunk,
                                                                uint32_t* __restrict__ part_first,
                                                                uint32_t* __restrict__ part_count) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  const uint32_t c0 = blockIdx.y * chunk;
  const uint32_t c1 = min(n_cols, c0 + chunk);
  const uint64_t sel = r < R ? row_sel[r] : 0ull;
  uint32_t first = PM_NONE, cnt = 0;
  for (uint32_t c = c0; c < c1; ++c) {
    const uint64_t m = col_mask[c];  // uniform address -> s_load
    const bool hit = (m & sel) != 0ull;
    cnt += hit;
    first = hit ? min(first, c) : first;
  }
  if (r < R) {
    part_first[(size_t)blockIdx.y * R + r] = first;
    part_count[(size_t)blockIdx.y * R + r] = cnt;
  }
}

// rank-th hit of a row (seeded chooser): second pass over the same pairs.
__global__ __launch_bounds__(256) void pair_select_scalar_kernel(const uint64_t* __restrict__ row_sel, uint32_t R,
                                                                 const uint64_t* __restrict__ col_mask,
                                                                 uint32_t n_cols, const uint32_t* __restrict__ rank,
                                                                 uint32_t* __restrict__ out) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  const uint64_t sel = r < R ? row_sel[r] : 0ull;
  uint32_t want = r < R ? rank[r] : PM_NONE;
  uint32_t res = PM_NONE, seen = 0;
  for (uint32_t c = 0; c < n_cols; ++c) {
    const bool hit = (col_mask[c] & sel) != 0ull;
    res = (hit && seen == want) ? c : res;
    seen += hit;
  }
  if (r < R) out[r] = res;
}

// ------------------------------------------------------------------------------------------------
// Phase B, bit-sliced kernel (default): the swept axis is stored as bit planes, plane[c][j] holds
// bit c of the masks of columns 64j..64j+63, so one 64-bit AND/OR evaluates 64 (row, col) pairs.
// rows live in lanes; the plane words of a chunk are staged in LDS and read as broadcasts.

__global__ __launch_bounds__(256) void build_planes_kernel(const uint64_t* __restrict__ col_mask, uint32_t n_cols,
                                                           uint32_t n_words, uint32_t n_planes,
                                                           uint64_t* __restrict__ planes) {
  // one wave per 64-column word; lane l owns column 64*j + l; __ballot gives the plane word.
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t j = (blockIdx.x * 256u + threadIdx.x) >> 6;
  if (j >= n_words) return;
  const uint32_t c = j * 64u + lane;
  const uint64_t m = c < n_cols ? col_mask[c] : 0ull;
  for (uint32_t b = 0; b < n_planes; ++b) {
    const uint64_t word = __ballot((m >> b) & 1ull);
    if (lane == 0) planes[(size_t)b * n_words + j] = word;
  }
}

template <int MAXP>
__global__ __launch_bounds__(256) void pair_sweep_planes_kernel(const uint64_t* __restrict__ row_sel, uint32_t R,
                                                                const uint64_t* __restrict__ planes,
                                                                uint32_t n_words, uint32_t n_planes,
                                                                uint32_t words_per_chunk,
                                                                uint32_t* __restrict__ part_first,
                                                                uint32_t* __restrict__ part_count) {
  extern __shared__ uint64_t s_pl[];  // [n_planes][words_per_chunk]
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  const uint32_t j0 = blockIdx.y * words_per_chunk;
  const uint32_t nj = min(words_per_chunk, n_words - j0);
  for (uint32_t i = threadIdx.x; i < n_planes * nj; i += 256u) {
    const uint32_t b = i / nj, j = i - b * nj;
    s_pl[b * words_per_chunk + j] = planes[(size_t)b * n_words + j0 + j];
  }
  __syncthreads();
  const uint64_t sel = r < R ? row_sel[r] : 0ull;
  uint32_t first = PM_NONE, cnt = 0;
  for (uint32_t j = 0; j < nj; ++j) {
    uint64_t hits = 0;
    uint64_t s = sel;
    while (s) {  // OR the planes this row selects; rows of one group share the selector
      const uint32_t b = __builtin_ctzll(s);
      s &= s - 1;
      if (b < n_planes) hits |= s_pl[b * words_per_chunk + j];
    }
    cnt += __popcll(hits);
    if (hits && first == PM_NONE) first = (j0 + j) * 64u + __builtin_ctzll(hits);
  }
  if (r < R) {
    part_first[(size_t)blockIdx.y * R + r] = first;
    part_count[(size_t)blockIdx.y * R + r] = cnt;
  }
}

template <int MAXP>
__global__ __launch_bounds__(256) void pair_select_planes_kernel(const uint64_t* __restrict__ row_sel, uint32_t R,
                                                                 const uint64_t* __restrict__ planes,
                                                                 uint32_t n_words, uint32_t n_planes,
                                                                 const uint32_t* __restrict__ rank,
                                                                 uint32_t* __restrict__ out) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= R) return;
  const uint64_t sel = row_sel[r];
  uint32_t want = rank[r];
  uint32_t res = PM_NONE;
  if (want != PM_NONE) {
    for (uint32_t j = 0; j < n_words; ++j) {
      uint64_t hits = 0, s = sel;
      while (s) {
        const uint32_t b = __builtin_ctzll(s);
        s &= s - 1;
        if (b < n_planes) hits |= planes[(size_t)b * n_words + j];
      }
      const uint32_t pc = __popcll(hits);
      if (want < pc) {
        for (uint32_t k = 0; k < want; ++k) hits &= hits - 1;  // drop `want` lowest set bits
        res = j * 64u + __builtin_ctzll(hits);
        break;
      }
      want -= pc;
    }
  }
  out[r] = res;
}

// Combine chunk partials: first = min over chunks, count = sum.
__global__ __launch_bounds__(256) void pair_combine_kernel(const uint32_t* __restrict__ part_first,
                                                           const uint32_t* __restrict__ part_count, uint32_t R,
                                                           uint32_t n_chunks, uint32_t* __restrict__ first,
                                                           uint32_t* __restrict__ count) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= R) return;
  uint32_t f = PM_NONE, c = 0;
  for (uint32_t k = 0; k < n_chunks; ++k) {
    f = min(f, part_first[(size_t)k * R + r]);
    c += part_count[(size_t)k * R + r];
  }
  first[r] = f;
  count[r] = c;
}

// ------------------------------------------------------------------------------------------------
// Selector / claim kernels around the reference-orientation sweep (scheduler_impl.rs:11-110).

__device__ __forceinline__ uint64_t splitmix64_mix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// row selector of worker w = the configuration bit of its group (0 when not in a group).
__global__ __launch_bounds__(256) void worker_selector_kernel(const int32_t* __restrict__ group_of,
                                                              const uint32_t* __restrict__ g_cfg, uint32_t W,
                                                              uint64_t* __restrict__ sel) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w >= W) return;
  const int32_t g = group_of[w];
  sel[w] = g >= 0 ? (1ull << g_cfg[g]) : 0ull;
}

// rank of the chosen task inside the applicable list (PM_CHOOSE_SEEDED): mix(seed ^ group id) % n.
__global__ __launch_bounds__(256) void chooser_rank_kernel(const int32_t* __restrict__ group_of,
                                                           const uint64_t* __restrict__ g_id,
                                                           const uint32_t* __restrict__ count, uint32_t W,
                                                           uint64_t seed, uint32_t* __restrict__ rank) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w >= W) return;
  const int32_t g = group_of[w];
  const uint32_t n = count[w];
  rank[w] = (g >= 0 && n) ? (uint32_t)(splitmix64_mix(seed ^ g_id[g]) % n) : PM_NONE;
}

// Column mask of worker w for the per-task orientation: eligible (Healthy & p2p & unassigned,
// mod.rs:492-497) ? compat & enabled : 0.
__global__ __launch_bounds__(256) void eligible_selector_kernel(const uint32_t* __restrict__ wflags,
                                                                const int32_t* __restrict__ group_of,
                                                                const uint64_t* __restrict__ compat, uint64_t enabled,
                                                                uint32_t W, uint64_t* __restrict__ sel) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w >= W) return;
  const uint32_t f = wflags[w];
  const bool e = (f & PM_W_HEALTHY) && (f & PM_W_HAS_P2P) && group_of[w] < 0;
  sel[w] = e ? (compat[w] & enabled) : 0ull;
}

// GROUP_INDEX = rank of the worker's address inside the group's BTreeSet<String> (mod.rs:424-434);
// by_rank[off + r] = the member of rank r, used for NEXT_P2P_ADDRESS (scheduler_impl.rs:115-128).
__global__ __launch_bounds__(256) void group_rank_kernel(const int32_t* __restrict__ group_of,
                                                         const uint32_t* __restrict__ g_n,
                                                         const uint32_t* __restrict__ g_off,
                                                         const uint32_t* __restrict__ members,
                                                         const uint32_t* __restrict__ addr_rank, uint32_t W,
                                                         uint32_t* __restrict__ rank_in_group,
                                                         uint32_t* __restrict__ by_rank) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w >= W) return;
  const int32_t g = group_of[w];
  if (g < 0) {
    rank_in_group[w] = 0;
    return;
  }
  const uint32_t n = g_n[g], off = g_off[g], my = addr_rank[w];
  uint32_t idx = 0;
  for (uint32_t k = 0; k < n; ++k) idx += addr_rank[members[off + k]] < my;
  rank_in_group[w] = idx;
  by_rank[off + idx] = w;
}

// Claim (SETNX, scheduler_impl.rs:74 / mod.rs:471-476) + publish row.  Every member of a group
// computed the same choice, so the group's task word is written with the same value by all.
__global__ __launch_bounds__(256) void claim_publish_kernel(ClaimArgs p) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w >= p.W) return;
  const int32_t g = p.group_of[w];
  pm_assignment a;
  a.task = PM_NONE;
  a.group_slot = PM_NONE;
  a.group_index = 0;
  a.group_size = 0;
  a.next_worker = PM_NONE;
  a.group_id = 0;
  if (g >= 0) {
    uint32_t t = p.g_task[g];  // get_current_group_task (scheduler_impl.rs:33)
    if (t == PM_NONE) {
      t = p.chosen[w];
      if (t != PM_NONE) p.g_task_next[g] = t;  // same value from every member
    }
    const uint32_t n = p.g_n[g], off = p.g_off[g];
    const uint32_t idx = p.rank_in_group[w];
    a.task = t;
    a.group_slot = (uint32_t)g;
    a.group_index = idx;
    a.group_size = n;
    a.next_worker = p.by_rank[off + ((idx + 1u == n) ? 0u : idx + 1u)];  // (idx + 1) % n
    a.group_id = p.g_id[g];
  }
  p.table[w] = a;
  p.task_col[w] = a.task;
}

// ------------------------------------------------------------------------------------------------
// NewestTaskPlugin: argmax (created_at, index) — LDS-staged wavefront argmax, last max wins.

__global__ __launch_bounds__(256) void newest_kernel(const int64_t* __restrict__ created_at, uint32_t T,
                                                     unsigned long long* __restrict__ best_key,
                                                     uint32_t* __restrict__ best_idx_by_block,
                                                     long long* __restrict__ best_val_by_block) {
  __shared__ long long s_v[4];
  __shared__ uint32_t s_i[4];
  long long bv = INT64_MIN;
  uint32_t bi = PM_NONE;
  for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < T; t += gridDim.x * 256u) {
    const long long v = created_at[t];
    if (bi == PM_NONE || v >= bv) {  // ascending t within a thread: >= keeps the last max
      bv = v;
      bi = t;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)bv, o, 64), hi = __shfl_xor((uint32_t)((uint64_t)bv >> 32), o, 64);
    const long long ov = (long long)(((uint64_t)hi << 32) | lo);
    const uint32_t oi = __shfl_xor(bi, o, 64);
    if (oi != PM_NONE && (bi == PM_NONE || ov > bv || (ov == bv && oi > bi))) {
      bv = ov;
      bi = oi;
    }
  }
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) {
    s_v[wave] = bv;
    s_i[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k)
      if (s_i[k] != PM_NONE && (bi == PM_NONE || s_v[k] > bv || (s_v[k] == bv && s_i[k] > bi))) {
        bv = s_v[k];
        bi = s_i[k];
      }
    best_idx_by_block[blockIdx.x] = bi;
    best_val_by_block[blockIdx.x] = bv;
  }
  (void)best_key;
}

// ------------------------------------------------------------------------------------------------
// Carve kernel: the whole greedy group formation of one tick inside ONE persistent workgroup
// (16 waves).  The greedy is a chain of dependent steps (group g+1's seed depends on what group g
// removed), so there is no cross-workgroup traffic to pay for: candidates are position-compacted,
// the alive/candidate bitmaps live in LDS, and every step is
//   seed search (bitmap scan) -> Haversine term for every remaining candidate -> top-(max-1)
//   selection by (key, position) with a wavefront argmin staged through LDS -> commit.
//
// Ordering key.  The reference sorts by d = 6371 * 2 * atan2(sqrt(a), sqrt(1-a)) computed with glibc
// libm (mod.rs:218-231).  d is a monotone function of a, so the kernel orders by a (f64, OCML sin)
// and proves the selection equal to the reference's: if every candidate whose a lies within a
// relative 2^-36 band around the last selected one has bit-identical coordinates (then the
// reference's distances tie exactly and the stable sort falls back to input order, like the
// kernel's (key, position) order), the selected SET is the reference's.  Otherwise the step is
// reported as UNCERTAIN and the engine settles exactly that step on the host with glibc.

#define CARVE_THREADS 1024
#define CARVE_WAVES 16

// ---- wave-wide unsigned min via DPP (no LDS traffic): row_shr 1,2,4,8 -> row_bcast15 -> row_bcast31,
// result broadcast from lane 63 with readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_min_step(uint64_t v) {
  const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(uint32_t)v, CTRL, ROW_MASK, 0xF, false);
  const uint32_t ohi =
      (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xF, false);
  const uint64_t o = ((uint64_t)ohi << 32) | olo;
  return o < v ? o : v;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
  v = dpp_min_step<0x111, 0xF>(v);  // row_shr:1
  v = dpp_min_step<0x112, 0xF>(v);  // row_shr:2
  v = dpp_min_step<0x114, 0xF>(v);  // row_shr:4
  v = dpp_min_step<0x118, 0xF>(v);  // row_shr:8
  v = dpp_min_step<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_min_step<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
  const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((uint64_t)hi << 32) | lo;
}

// barrier for exchanges that go through LDS only (no global-memory drain)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BlockRed {
  uint32_t a[CARVE_WAVES];
  uint32_t b[CARVE_WAVES];
  uint32_t part_n[CARVE_WAVES];
  uint32_t flag[CARVE_WAVES];
};

// sin on [-pi/2, pi/2] as an odd Taylor polynomial to x^19 (|rel err| < 1e-15 there); larger arguments
// (longitude differences beyond 180 degrees) take the OCML path.  The certificate band (2^-35) is four
// orders of magnitude wider than this error.
__device__ __forceinline__ double sin_band(double x) {
  if (fabs(x) > 1.5707963267948966) return sin(x);
  const double z = x * x;
  double p = -8.2206352466243297e-18;               // -1/19!
  p = fma(p, z, 2.8114572543455206e-15);            //  1/17!
  p = fma(p, z, -7.6471637318198164e-13);           // -1/15!
  p = fma(p, z, 1.6059043836821613e-10);            //  1/13!
  p = fma(p, z, -2.5052108385441720e-08);           // -1/11!
  p = fma(p, z, 2.7557319223985893e-06);            //  1/9!
  p = fma(p, z, -1.9841269841269841e-04);           // -1/7!
  p = fma(p, z, 8.3333333333333332e-03);            //  1/5!
  p = fma(p, z, -1.6666666666666666e-01);           // -1/3!
  return fma(x * z, p, x);
}

__device__ __forceinline__ double hav_a(double lat1, double lon1, double cos1, double lat2, double lon2,
                                        double cos2) {
  const double dlat = (lat2 - lat1) * PM_RAD;
  const double dlon = (lon2 - lon1) * PM_RAD;
  const double s1 = sin_band(dlat * 0.5);
  const double s2 = sin_band(dlon * 0.5);
  return s1 * s1 + cos1 * cos2 * (s2 * s2);
}

__device__ __forceinline__ bool bit_at(const uint64_t* b, uint32_t i) { return (b[i >> 6] >> (i & 63u)) & 1ull; }

// Ordering key of a candidate: the f64 bits of its Haversine term `a` with the low SLOT_BITS replaced by
// the slot number (slot order == input order), so one u64 compare is the whole (distance, input order)
// comparison.  Dropping SLOT_BITS mantissa bits is covered by the certificate band.
__device__ __forceinline__ uint64_t pack_key(uint64_t key_bits, uint32_t slot, uint32_t slot_bits) {
  return ((key_bits >> slot_bits) << slot_bits) | slot;
}

enum { STEP_CONTINUE = 0, STEP_BREAK = 1, STEP_UNCERTAIN = 2, STEP_OVERFLOW = 3 };

#ifdef PM_CARVE_PROF
#define PROF_DECL uint64_t prof_t0 = __builtin_amdgcn_s_memtime()
#define PROF_MARK(slot)                                          \
  do {                                                           \
    const uint64_t t_ = __builtin_amdgcn_s_memtime();            \
    if (threadIdx.x == 0) p.status->prof[slot] += t_ - prof_t0;  \
    prof_t0 = t_;                                                \
  } while (0)
#else
#define PROF_DECL
#define PROF_MARK(slot)
#endif

struct StepCtx {
  uint32_t mode, proximity, min_s, max_s, cfg;
  uint32_t n_list;       // slots of the current list
  uint32_t n_cand;       // live slots
  uint32_t n_groups, mem_off;
  uint32_t total_available;
  uint32_t steps;
  unsigned long long cand_sum;
};

// LDS carve of one candidate list of at most E*1024 slots.  Per-lane state lives in registers for the
// whole run: coordinates of the lane's E slots (slot = tid + j*1024), loaded once after a compaction;
// packed keys are recomputed per step.  LDS holds the worker ids, the alive / loc bitmaps and the
// per-wave partial selections.  Runs steps until the configuration is exhausted, a recompaction is due,
// or a step cannot be certified.  Three LDS-only barriers per step.
template <int E>
__device__ int carve_run_lds(const CarveArgs& p, BlockRed& red, StepCtx& c, const uint32_t* l_wid, uint64_t* l_alive,
                             const uint64_t* l_loc, uint64_t* part, uint32_t* sel_out, uint32_t steps_before) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  constexpr uint32_t SB = PM_CARVE_SLOT_BITS;
  const uint32_t lw = (c.n_list + 63u) >> 6;
  double clat[E], clon[E], ccos[E];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const uint32_t s = tid + (uint32_t)j * CARVE_THREADS;
    const bool in = s < c.n_list;
    clat[j] = in ? p.cc_lat[s] : 0.0;
    clon[j] = in ? p.cc_lon[s] : 0.0;
    ccos[j] = in ? p.cc_cos[s] : 0.0;
  }

  for (;;) {
    // FORM: `while total_available >= min` (mod.rs:507) with `compatible < min => break` (:517-519).
    // MERGE: `while remaining_groups.len() >= min` (mod.rs:695).
    if (!((c.mode == CARVE_MODE_MERGE || c.total_available >= c.min_s) && c.n_cand >= c.min_s && c.n_cand > 0))
      return STEP_BREAK;
    PROF_DECL;
    // ---- seed: first live slot with a location, else first live slot (mod.rs:526-530); every wave finds
    // it redundantly from the bitmaps with one ballot per 64 words (no barrier, no shuffle tree)
    uint32_t f_loc = PM_NONE, f_any = PM_NONE;
    for (uint32_t j0 = 0; j0 < lw && (f_loc == PM_NONE || f_any == PM_NONE); j0 += 64u) {
      const uint32_t j = j0 + lane;
      const uint64_t al = j < lw ? l_alive[j] : 0ull;
      const uint64_t ll = j < lw ? (al & l_loc[j]) : 0ull;
      if (f_any == PM_NONE) {
        const uint64_t nz = __ballot(al != 0ull);
        if (nz) {
          const int src = __builtin_ctzll(nz);
          const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(al >> 32), src) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)al, src);
          f_any = (j0 + src) * 64u + __builtin_ctzll(w);
        }
      }
      if (f_loc == PM_NONE) {
        const uint64_t nz = __ballot(ll != 0ull);
        if (nz) {
          const int src = __builtin_ctzll(nz);
          const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ll >> 32), src) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ll, src);
          f_loc = (j0 + src) * 64u + __builtin_ctzll(w);
        }
      }
    }
    PROF_MARK(0);

    const uint32_t want = c.max_s - 1u < c.n_cand - 1u ? c.max_s - 1u : c.n_cand - 1u;  // fill to max (mod.rs:545-551)
    uint64_t rk[E];
    uint32_t seed = f_any;
    bool use_dist = false, located_only = false;
    uint64_t last = 0;
    uint32_t n_sel = 0, total = 0;

    // attempt 0: FORM, or MERGE with proximity (mod.rs:762-821); attempt 1: MERGE first-come (:824-848)
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (c.mode == CARVE_MODE_FORM) {
        if (attempt == 1) break;
        seed = f_any;
        use_dist = false;
        if (c.proximity && f_loc != PM_NONE) {  // seed = first WITH a location
          seed = f_loc;
          use_dist = true;
        }  // else first-come (:553-561), or a seed without location makes the sort a no-op (:238)
        located_only = false;
      } else if (attempt == 0) {
        if (!(c.proximity && f_loc != PM_NONE)) continue;
        seed = f_loc;
        use_dist = true;
        located_only = true;
      } else {
        if (!(total == 0 || (total < c.max_s && total < c.min_s))) break;
        seed = f_any;
        use_dist = false;
        located_only = false;
      }

      // ---- keys (registers only; the seed's coordinates are one uniform load each)
      const double slat = p.cc_lat[seed], slon = p.cc_lon[seed], scos = p.cc_cos[seed];
      uint64_t lmin = ~0ull;
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const uint32_t s = tid + (uint32_t)j * CARVE_THREADS;
        uint64_t k = ~0ull;
        if (s < c.n_list && s != seed && bit_at(l_alive, s)) {
          if (!use_dist) {
            k = s;
          } else if (bit_at(l_loc, s)) {
            k = pack_key((uint64_t)__double_as_longlong(hav_a(slat, slon, scos, clat[j], clon[j], ccos[j])), s, SB);
          } else if (!located_only) {
            k = pack_key(PM_KEY_NOLOC, s, SB);
          }
        }
        rk[j] = k;
        lmin = k < lmin ? k : lmin;
      }
      PROF_MARK(1);
      n_sel = 0;
      last = 0;
      if (want > 0) {
        if (want <= PM_CARVE_PART) {
          // ---- level 1: this wave's `want` smallest, DPP argmin rounds, no barrier
          uint32_t cnt = 0;
          while (cnt < want) {
            const uint64_t v = wave_min_u64(lmin);
            if (v == ~0ull) break;
            if (lane == 0) part[wave * PM_CARVE_PART + cnt] = v;
            ++cnt;
            if (lmin == v) {  // the owning lane advances to its next element (keys are unique)
              uint64_t m = ~0ull;
#pragma unroll
              for (int j = 0; j < E; ++j) m = (rk[j] > v && rk[j] < m) ? rk[j] : m;
              lmin = m;
            }
          }
          if (lane == 0) red.part_n[wave] = cnt;
          PROF_MARK(2);
          lds_barrier();
          PROF_MARK(3);
          // ---- level 2: 16-way merge of the sorted partial lists, redundantly in every wave
          uint32_t ptr = 0;
          const uint32_t my_n = lane < CARVE_WAVES ? red.part_n[lane] : 0u;
          uint64_t head = my_n ? part[lane * PM_CARVE_PART] : ~0ull;
          uint64_t mine = ~0ull;
          while (n_sel < want) {
            const uint64_t v = wave_min_u64(head);
            if (v == ~0ull) break;
            if (lane == n_sel) mine = v;
            last = v;
            ++n_sel;
            if (head == v) {
              ++ptr;
              head = ptr < my_n ? part[lane * PM_CARVE_PART + ptr] : ~0ull;
            }
          }
          if (wave == 0 && lane < n_sel) sel_out[lane] = (uint32_t)(mine & ((1ull << SB) - 1ull));
          PROF_MARK(4);
        } else {
          // ---- wide groups: one workgroup-wide round per member (wave argmin -> LDS -> fold)
          while (n_sel < want) {
            const uint64_t v = wave_min_u64(lmin);
            if (lane == 0) part[wave] = v;
            lds_barrier();
            uint64_t b = ~0ull;
#pragma unroll
            for (uint32_t k = 0; k < CARVE_WAVES; ++k) b = part[k] < b ? part[k] : b;
            lds_barrier();
            if (b == ~0ull) break;
            const uint32_t bs = (uint32_t)(b & ((1ull << SB) - 1ull));
            if (tid == 0) {
              if (n_sel < PM_CARVE_SEL_CAP)
                sel_out[n_sel] = bs;
              else if (c.mem_off + 1u + n_sel < p.cap_members)
                p.members[c.mem_off + 1u + n_sel] = l_wid[bs];
            }
            last = b;
            ++n_sel;
            if (lmin == b) {
              uint64_t m = ~0ull;
#pragma unroll
              for (int j = 0; j < E; ++j) m = (rk[j] > b && rk[j] < m) ? rk[j] : m;
              lmin = m;
            }
          }
        }
      }
      total = 1u + n_sel;
      if (c.mode == CARVE_MODE_FORM) break;
      if (attempt == 0 && want > 0 && want <= PM_CARVE_PART) lds_barrier();  // part/part_n reused by attempt 1
    }
    if (total == 0) return STEP_BREAK;                                   // MERGE: nothing selectable
    if (c.mode == CARVE_MODE_FORM && total < c.min_s) return STEP_BREAK;  // mod.rs:564-566
    if (c.mode == CARVE_MODE_MERGE && total < 2u) return STEP_BREAK;      // is_merge_beneficial (mod.rs:868-870)

    // ---- exactness certificate for a distance-ordered selection (see the comment above carve_kernel)
    int uncertain = p.debug_uncertain_every && use_dist &&
                    ((steps_before + c.steps + 1u) % p.debug_uncertain_every) == 0u;
    const uint64_t noloc_key = (PM_KEY_NOLOC >> SB) << SB;
    const uint64_t last_key = (last >> SB) << SB;
    if (use_dist && n_sel > 0 && last_key != noloc_key) {
      const uint32_t ls = (uint32_t)(last & ((1ull << SB) - 1ull));
      const double a_m = __longlong_as_double((long long)last_key);
      const double band = a_m * PM_TIE_BAND + 1e-300;
      const double mlat = p.cc_lat[ls], mlon = p.cc_lon[ls];  // uniform loads
      if (a_m > PM_A_MAX_SAFE) uncertain = 1;
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const uint64_t kb = (rk[j] >> SB) << SB;
        const double a = __longlong_as_double((long long)kb);
        const bool near = rk[j] != ~0ull && kb != noloc_key && fabs(a - a_m) <= band;
        if (near && (clat[j] != mlat || clon[j] != mlon)) uncertain = 1;
      }
    }
    const uint64_t ub = __ballot(uncertain != 0);
    if (lane == 0) red.flag[wave] = ub != 0ull;
    PROF_MARK(5);
    lds_barrier();
    PROF_MARK(6);
    uint32_t any = 0;
#pragma unroll
    for (uint32_t k = 0; k < CARVE_WAVES; ++k) any |= red.flag[k];
    if (any) {
      if (tid == 0) p.status->stop_seed = l_wid[seed];
      return STEP_UNCERTAIN;
    }
    if (c.n_groups >= p.cap_groups || c.mem_off + total > p.cap_members) return STEP_OVERFLOW;

    // ---- commit (create_group_atomically mod.rs:299-322; healthy_nodes.retain :585): selected slots =
    // seed + every key <= last.  Each wave owns whole bitmap words (slot>>6 == j*16 + wave): ballot writes them.
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const uint32_t s = tid + (uint32_t)j * CARVE_THREADS;
      const uint32_t wj = (uint32_t)j * CARVE_WAVES + wave;
      if (wj < lw) {  // wave-uniform
        const bool was = bit_at(l_alive, s);
        const bool sel = was && (s == seed || (n_sel > 0 && rk[j] <= last));
        const uint64_t nw = __ballot(was && !sel);
        if (lane == 0) l_alive[wj] = nw;
      }
    }
    if (wave == 0) {  // group record + members: LDS -> fire-and-forget global stores
      if (lane == 0) {
        p.members[c.mem_off] = l_wid[seed];
        p.g_cfg[c.n_groups] = c.cfg;
        p.g_n[c.n_groups] = total;
        p.g_off[c.n_groups] = c.mem_off;
      }
      const uint32_t lim = n_sel < PM_CARVE_SEL_CAP ? n_sel : PM_CARVE_SEL_CAP;
      for (uint32_t r = lane; r < lim; r += 64u) p.members[c.mem_off + 1u + r] = l_wid[sel_out[r]];
    }
    PROF_MARK(7);
    lds_barrier();
    PROF_MARK(8);
    c.n_groups += 1;
    c.mem_off += total;
    c.cand_sum += c.n_cand;
    c.n_cand -= total;
    c.total_available -= total;  // mod.rs:586
    c.steps += 1;
    // drop dead slots once more than half of the list is gone
    if (c.n_cand * 2u < c.n_list && c.n_list > CARVE_THREADS) return STEP_CONTINUE;
  }
}

// ---- generic path for candidate lists that do not fit the LDS/register scheme (> PM_CARVE_SLOTS):
// packed keys, positions and bitmaps live in HBM/L2; one workgroup-wide argmin round per member.
__device__ int carve_step_mem(const CarveArgs& p, BlockRed& red, StepCtx& c, uint64_t* part, uint64_t* key,
                              const uint32_t* wid, uint64_t* alive, const uint64_t* loc, uint32_t steps_before) {
  if (!((c.mode == CARVE_MODE_MERGE || c.total_available >= c.min_s) && c.n_cand >= c.min_s && c.n_cand > 0))
    return STEP_BREAK;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  constexpr uint32_t SB = PM_CARVE_SLOT_BITS_MEM;
  const uint32_t lw = (c.n_list + 63u) >> 6;
  uint32_t f_loc = PM_NONE, f_any = PM_NONE;
  for (uint32_t j = tid; j < lw; j += CARVE_THREADS) {
    const uint64_t al = alive[j];
    if (al && f_any == PM_NONE) f_any = j * 64u + __builtin_ctzll(al);
    const uint64_t ll = al & loc[j];
    if (ll && f_loc == PM_NONE) f_loc = j * 64u + __builtin_ctzll(ll);
  }
  f_loc = wave_min(f_loc);
  f_any = wave_min(f_any);
  if (lane == 0) {
    red.a[wave] = f_loc;
    red.b[wave] = f_any;
  }
  __syncthreads();
  f_loc = PM_NONE;
  f_any = PM_NONE;
  for (uint32_t k = 0; k < CARVE_WAVES; ++k) {
    f_loc = min(f_loc, red.a[k]);
    f_any = min(f_any, red.b[k]);
  }
  __syncthreads();
  const uint32_t want = c.max_s - 1u < c.n_cand - 1u ? c.max_s - 1u : c.n_cand - 1u;
  uint32_t seed = f_any;
  bool use_dist = false, located_only = false;
  uint64_t last = 0;
  uint32_t n_sel = 0, total = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (c.mode == CARVE_MODE_FORM) {
      if (attempt == 1) break;
      seed = f_any;
      use_dist = false;
      if (c.proximity && f_loc != PM_NONE) {
        seed = f_loc;
        use_dist = true;
      }
      located_only = false;
    } else if (attempt == 0) {
      if (!(c.proximity && f_loc != PM_NONE)) continue;
      seed = f_loc;
      use_dist = true;
      located_only = true;
    } else {
      if (!(total == 0 || (total < c.max_s && total < c.min_s))) break;
      seed = f_any;
      use_dist = false;
      located_only = false;
    }
    const double slat = p.cc_lat[seed], slon = p.cc_lon[seed], scos = p.cc_cos[seed];
    uint64_t lmin = ~0ull;
    for (uint32_t s = tid; s < c.n_list; s += CARVE_THREADS) {
      uint64_t k = ~0ull;
      if (s != seed && bit_at(alive, s)) {
        if (!use_dist) {
          k = s;
        } else if (bit_at(loc, s)) {
          k = pack_key((uint64_t)__double_as_longlong(hav_a(slat, slon, scos, p.cc_lat[s], p.cc_lon[s], p.cc_cos[s])), s, SB);
        } else if (!located_only) {
          k = pack_key(PM_KEY_NOLOC, s, SB);
        }
      }
      key[s] = k;
      lmin = k < lmin ? k : lmin;
    }
    n_sel = 0;
    last = 0;
    while (n_sel < want) {
      const uint64_t v = wave_min_u64(lmin);
      if (lane == 0) part[wave] = v;
      __syncthreads();
      uint64_t b = ~0ull;
      for (uint32_t k = 0; k < CARVE_WAVES; ++k) b = part[k] < b ? part[k] : b;
      __syncthreads();
      if (b == ~0ull) break;
      if (tid == 0 && c.mem_off + 1u + n_sel < p.cap_members)
        p.members[c.mem_off + 1u + n_sel] = wid[(uint32_t)(b & ((1ull << SB) - 1ull))];
      last = b;
      ++n_sel;
      if (lmin == b) {
        uint64_t m = ~0ull;
        for (uint32_t s = tid; s < c.n_list; s += CARVE_THREADS) {
          const uint64_t k = key[s];
          m = (k > b && k < m) ? k : m;
        }
        lmin = m;
      }
    }
    total = 1u + n_sel;
    if (c.mode == CARVE_MODE_FORM) break;
  }
  if (total == 0) return STEP_BREAK;
  if (c.mode == CARVE_MODE_FORM && total < c.min_s) return STEP_BREAK;
  if (c.mode == CARVE_MODE_MERGE && total < 2u) return STEP_BREAK;

  int uncertain = p.debug_uncertain_every && use_dist &&
                  ((steps_before + c.steps + 1u) % p.debug_uncertain_every) == 0u;
  const uint64_t last_key = (last >> SB) << SB;
  if (use_dist && n_sel > 0 && last_key != ((PM_KEY_NOLOC >> SB) << SB)) {
    const uint32_t ls = (uint32_t)(last & ((1ull << SB) - 1ull));
    const double a_m = __longlong_as_double((long long)last_key);
    const double band = a_m * PM_TIE_BAND_MEM + 1e-300;
    const double mlat = p.cc_lat[ls], mlon = p.cc_lon[ls];
    if (a_m > PM_A_MAX_SAFE) uncertain = 1;
    for (uint32_t s = tid; s < c.n_list; s += CARVE_THREADS) {
      const uint64_t k = key[s];
      if (k == ~0ull || s == ls) continue;
      const uint64_t kb = (k >> SB) << SB;
      if (kb == ((PM_KEY_NOLOC >> SB) << SB)) continue;
      const double a = __longlong_as_double((long long)kb);
      if (fabs(a - a_m) <= band && (p.cc_lat[s] != mlat || p.cc_lon[s] != mlon)) uncertain = 1;
    }
  }
  if (__syncthreads_or(uncertain)) {
    if (tid == 0) p.status->stop_seed = wid[seed];
    return STEP_UNCERTAIN;
  }
  if (c.n_groups >= p.cap_groups || c.mem_off + total > p.cap_members) return STEP_OVERFLOW;
  for (uint32_t s = tid; s < c.n_list; s += CARVE_THREADS) {
    if (!bit_at(alive, s)) continue;
    if (s == seed || (n_sel > 0 && key[s] <= last))
      atomicAnd((unsigned long long*)&alive[s >> 6], ~(1ull << (s & 63u)));
  }
  if (tid == 0) {
    p.members[c.mem_off] = wid[seed];
    p.g_cfg[c.n_groups] = c.cfg;
    p.g_n[c.n_groups] = total;
    p.g_off[c.n_groups] = c.mem_off;
  }
  __syncthreads();
  c.n_groups += 1;
  c.mem_off += total;
  c.cand_sum += c.n_cand;
  c.n_cand -= total;
  c.total_available -= total;
  c.steps += 1;
  return STEP_CONTINUE;
}

// Stable compaction of the live positions of this configuration into list slots: two passes over
// contiguous per-wave ranges.  Returns the list length; red.a keeps the per-wave counts for the placement.
__device__ uint32_t carve_compact_count(const CarveArgs& p, BlockRed& red, uint32_t n, uint64_t cbit) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n_words = (n + 63u) >> 6;
  const uint32_t wpw = (n_words + CARVE_WAVES - 1u) / CARVE_WAVES;
  const uint32_t j0 = wave * wpw, j1 = min(n_words, j0 + wpw);
  uint32_t cnt = 0;
  for (uint32_t j = j0; j < j1; ++j) {
    const uint32_t i = j * 64u + lane;
    const bool c = i < n && bit_at(p.alive_g, i) && (p.mode == CARVE_MODE_MERGE || (p.c_compat[i] & cbit) != 0ull);
    cnt += __popcll(__ballot(c));
  }
  __syncthreads();  // previous users of red.a are done
  if (lane == 0) red.a[wave] = cnt;
  __syncthreads();
  uint32_t total = 0;
  for (uint32_t k = 0; k < CARVE_WAVES; ++k) total += red.a[k];
  return total;
}

__device__ void carve_compact_place(const CarveArgs& p, BlockRed& red, uint32_t n, uint64_t cbit, uint32_t n_list,
                                    uint32_t* wid, uint64_t* alive, uint64_t* loc) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_words = (n + 63u) >> 6;
  const uint32_t wpw = (n_words + CARVE_WAVES - 1u) / CARVE_WAVES;
  const uint32_t j0 = wave * wpw, j1 = min(n_words, j0 + wpw);
  uint32_t off = 0;
  for (uint32_t k = 0; k < wave; ++k) off += red.a[k];
  for (uint32_t j = j0; j < j1; ++j) {
    const uint32_t i = j * 64u + lane;
    const bool c = i < n && bit_at(p.alive_g, i) && (p.mode == CARVE_MODE_MERGE || (p.c_compat[i] & cbit) != 0ull);
    const uint64_t bal = __ballot(c);
    if (c) {
      const uint32_t s = off + __popcll(bal & ((1ull << lane) - 1ull));
      p.slot_pos[s] = i;
      wid[s] = p.order[i];
      p.cc_lat[s] = p.c_lat[i];
      p.cc_lon[s] = p.c_lon[i];
      p.cc_cos[s] = p.c_cos[i];
    }
    off += __popcll(bal);
  }
  __syncthreads();
  const uint32_t lw = (n_list + 63u) >> 6;
  for (uint32_t base = 0; base < lw * 64u; base += CARVE_THREADS) {
    const uint32_t s = base + tid;
    const bool in = s < n_list;
    const bool hl = in && bit_at(p.loc_g, p.slot_pos[s]);
    const uint64_t ba = __ballot(in), bl = __ballot(hl);
    if (lane == 0 && (s >> 6) < lw) {
      alive[s >> 6] = ba;
      loc[s >> 6] = bl;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(CARVE_THREADS) void carve_kernel(CarveArgs p) {
  // All LDS lives in the dynamic region, every carve offset a multiple of 16 B (a static __shared__ in
  // front of it would shift the base and put every 64-bit DS access on the 64-cycle misaligned path).
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  uint64_t* part = reinterpret_cast<uint64_t*>(s_raw);                       // [16 * PART]
  uint64_t* lds_alive = part + CARVE_WAVES * PM_CARVE_PART;                  // [SLOTS / 64]
  uint64_t* lds_loc = lds_alive + PM_CARVE_SLOTS / 64;                       // [SLOTS / 64]
  uint32_t* lds_wid = reinterpret_cast<uint32_t*>(lds_loc + PM_CARVE_SLOTS / 64);  // [SLOTS]
  uint32_t* sel_out = lds_wid + PM_CARVE_SLOTS;                              // [SEL_CAP]
  BlockRed& red = *reinterpret_cast<BlockRed*>(sel_out + PM_CARVE_SEL_CAP);
  uint32_t& s_n = *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(&red) + sizeof(BlockRed));

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  CarveStatus* st = p.status;

  // ---- ordered eligible list.  FORM: compact the eligible rows (Healthy & p2p & unassigned,
  // mod.rs:492-497) in input order.  MERGE: supplied by the engine.
  uint32_t n;
  if (p.mode == CARVE_MODE_FORM) {
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (uint32_t base = 0; base < p.W; base += CARVE_THREADS) {
      const uint32_t w = base + tid;
      bool e = false;
      if (w < p.W) {
        const uint32_t f = p.wflags[w];
        e = (f & PM_W_HEALTHY) && (f & PM_W_HAS_P2P) && p.group_of[w] < 0;
      }
      const uint64_t bal = __ballot(e);
      if (lane == 0) red.a[wave] = __popcll(bal);
      __syncthreads();
      uint32_t off = s_n;
      for (uint32_t k = 0; k < wave; ++k) off += red.a[k];
      if (e) p.order[off + __popcll(bal & ((1ull << lane) - 1ull))] = w;
      __syncthreads();
      if (tid == 0) {
        uint32_t tot = 0;
        for (uint32_t k = 0; k < CARVE_WAVES; ++k) tot += red.a[k];
        s_n += tot;
      }
      __syncthreads();
    }
    n = s_n;
  } else {
    n = p.n_order;
  }
  const uint32_t n_words = (n + 63u) >> 6;

  // position-indexed columns + alive / loc bitmaps (L2 resident)
  for (uint32_t base = 0; base < n_words * 64u; base += CARVE_THREADS) {
    const uint32_t i = base + tid;
    bool has_loc = false;
    if (i < n) {
      const uint32_t w = p.order[i];
      has_loc = (p.wflags[w] & PM_W_HAS_LOC) != 0;
      p.c_lat[i] = p.lat[w];
      p.c_lon[i] = p.lon[w];
      p.c_cos[i] = p.coslat[w];
      p.c_compat[i] = p.compat[w];
    }
    const uint64_t bl = __ballot(has_loc);
    const uint64_t ba = __ballot(i < n);
    if (lane == 0 && (i >> 6) < n_words) {
      p.loc_g[i >> 6] = bl;
      p.alive_g[i >> 6] = ba;
    }
  }
  __syncthreads();

  const uint32_t steps_before = st->steps_total;
  StepCtx c;
  c.mode = p.mode;
  c.proximity = p.proximity;
  c.n_groups = st->n_groups;
  c.mem_off = st->n_members;
  c.total_available = n;  // mod.rs:503
  c.steps = 0;
  c.cand_sum = 0;
  uint32_t exit_state = CARVE_STATE_DONE, stop_ci = p.n_avail;

  for (uint32_t ci = p.start_ci; ci < p.n_avail && exit_state == CARVE_STATE_DONE; ++ci) {  // mod.rs:505
    c.cfg = p.avail_cfg[ci];
    c.min_s = p.min_size[ci];
    c.max_s = p.max_size[ci];
    const uint64_t cbit = 1ull << c.cfg;
    if (p.mode == CARVE_MODE_FORM && c.total_available < c.min_s) continue;  // `while` of mod.rs:507 never entered

    // candidate list of this configuration (mod.rs:511-515 evaluated once; removals are applied to the
    // bitmaps instead of re-filtering).  MERGE: the list is already filtered by the engine.
    for (;;) {
      PROF_DECL;
      c.n_list = carve_compact_count(p, red, n, cbit);
      c.n_cand = c.n_list;
      const bool in_lds = c.n_list <= PM_CARVE_SLOTS;
      uint32_t* wid = in_lds ? lds_wid : p.slot_wid;
      uint64_t* alive = in_lds ? lds_alive : p.bits_scratch;
      uint64_t* loc = in_lds ? lds_loc : p.bits_scratch + p.bits_stride;
      carve_compact_place(p, red, n, cbit, c.n_list, wid, alive, loc);
      PROF_MARK(9);

      int rc;
      if (!in_lds) {
        do {
          rc = carve_step_mem(p, red, c, part, p.keys, wid, alive, loc, steps_before);
        } while (rc == STEP_CONTINUE && !(c.n_cand * 2u < c.n_list));
      } else if (c.n_list <= 1u * CARVE_THREADS) {
        rc = carve_run_lds<1>(p, red, c, wid, alive, loc, part, sel_out, steps_before);
      } else if (c.n_list <= 2u * CARVE_THREADS) {
        rc = carve_run_lds<2>(p, red, c, wid, alive, loc, part, sel_out, steps_before);
      } else if (c.n_list <= 4u * CARVE_THREADS) {
        rc = carve_run_lds<4>(p, red, c, wid, alive, loc, part, sel_out, steps_before);
      } else {
        rc = carve_run_lds<8>(p, red, c, wid, alive, loc, part, sel_out, steps_before);
      }
      __syncthreads();
#ifdef PM_CARVE_PROF
      prof_t0 = __builtin_amdgcn_s_memtime();
#endif
      // dead slots -> position bitmap, so the next compaction / configuration sees the removals
      for (uint32_t s = tid; s < c.n_list; s += CARVE_THREADS)
        if (!bit_at(alive, s)) {
          const uint32_t i = p.slot_pos[s];
          atomicAnd((unsigned long long*)&p.alive_g[i >> 6], ~(1ull << (i & 63u)));
        }
      __syncthreads();
      PROF_MARK(10);
      if (rc == STEP_UNCERTAIN || rc == STEP_OVERFLOW) {
        exit_state = rc == STEP_UNCERTAIN ? CARVE_STATE_UNCERTAIN : CARVE_STATE_OVERFLOW;
        stop_ci = ci;
        break;
      }
      if (rc == STEP_BREAK) break;  // configuration exhausted; STEP_CONTINUE => recompact and go on
    }
  }

  // group_of for everything carved by this launch (FORM), one parallel pass at the end
  if (p.mode == CARVE_MODE_FORM) {
    __syncthreads();
    for (uint32_t g = st->n_groups + wave; g < c.n_groups; g += CARVE_WAVES) {
      const uint32_t off = p.g_off[g], gn = p.g_n[g];
      for (uint32_t k = lane; k < gn; k += 64u) p.group_of[p.members[off + k]] = (int32_t)g;
    }
  }
  __syncthreads();
  if (tid == 0) {
    st->state = exit_state;
    st->n_groups = c.n_groups;
    st->n_members = c.mem_off;
    st->steps_total = steps_before + c.steps;
    st->stop_ci = stop_ci;
    st->n_eligible = n;
    st->cand_sum += c.cand_sum;
  }
}

// ------------------------------------------------------------------------------------------------
// launchers (called from pm_engine.cpp)

void launch_compat(const CompatArgs& a, hipStream_t s) {
  if (a.W == 0) return;
  hipLaunchKernelGGL(compat_kernel, dim3((a.W + 255u) / 256u), dim3(256), 0, s, a);
}
void launch_coslat(const double* lat, double* coslat, uint32_t W, hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(coslat_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, lat, coslat, W);
}
void launch_worker_selector(const int32_t* group_of, const uint32_t* g_cfg, uint32_t W, uint64_t* sel,
                            hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(worker_selector_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, group_of, g_cfg, W, sel);
}
void launch_chooser_rank(const int32_t* group_of, const uint64_t* g_id, const uint32_t* count, uint32_t W,
                         uint64_t seed, uint32_t* rank, hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(chooser_rank_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, group_of, g_id, count, W, seed,
                     rank);
}
void launch_eligible_selector(const uint32_t* wflags, const int32_t* group_of, const uint64_t* compat,
                              uint64_t enabled, uint32_t W, uint64_t* sel, hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(eligible_selector_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, wflags, group_of, compat,
                     enabled, W, sel);
}
void launch_group_rank(const int32_t* group_of, const uint32_t* g_n, const uint32_t* g_off, const uint32_t* members,
                       const uint32_t* addr_rank, uint32_t W, uint32_t* rank_in_group, uint32_t* by_rank,
                       hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(group_rank_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, group_of, g_n, g_off, members,
                     addr_rank, W, rank_in_group, by_rank);
}
void launch_claim_publish(const ClaimArgs& a, hipStream_t s) {
  if (a.W == 0) return;
  hipLaunchKernelGGL(claim_publish_kernel, dim3((a.W + 255u) / 256u), dim3(256), 0, s, a);
}

void launch_build_planes(const uint64_t* col_mask, uint32_t n_cols, uint32_t n_planes, uint64_t* planes,
                         hipStream_t s) {
  const uint32_t n_words = (n_cols + 63u) / 64u;
  if (n_words == 0) return;
  hipLaunchKernelGGL(build_planes_kernel, dim3((n_words * 64u + 255u) / 256u), dim3(256), 0, s, col_mask, n_cols,
                     n_words, n_planes, planes);
}

// Pair sweep: rows x cols -> first hit + hit count per row.  scratch holds 2 * n_chunks * R u32.
void launch_pair_sweep(int variant, const uint64_t* row_sel, uint32_t R, const uint64_t* col_mask,
                       const uint64_t* planes, uint32_t n_cols, uint32_t n_planes, uint32_t* scratch,
                       uint32_t max_chunks, uint32_t* first, uint32_t* count, hipStream_t s) {
  if (R == 0) return;
  const uint32_t rb = (R + 255u) / 256u;
  uint32_t n_chunks;
  uint32_t* part_first = scratch;
  if (variant == 1) {
    // enough workgroups to cover 256 CUs several times over
    n_chunks = (4096u + rb - 1u) / rb;
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    uint32_t chunk = (n_cols + n_chunks - 1u) / (n_chunks ? n_chunks : 1u);
    if (chunk == 0) chunk = 1;
    n_chunks = n_cols ? (n_cols + chunk - 1u) / chunk : 1u;
    uint32_t* part_count = scratch + (size_t)n_chunks * R;
    hipLaunchKernelGGL(pair_sweep_scalar_kernel, dim3(rb, n_chunks), dim3(256), 0, s, row_sel, R, col_mask, n_cols,
                       chunk, part_first, part_count);
    hipLaunchKernelGGL(pair_combine_kernel, dim3(rb), dim3(256), 0, s, part_first, part_count, R, n_chunks, first,
                       count);
  } else {
    const uint32_t n_words = (n_cols + 63u) / 64u;
    uint32_t wpc = 0;
    if (n_words) {
      n_chunks = (2048u + rb - 1u) / rb;
      if (n_chunks > max_chunks) n_chunks = max_chunks;
      if (n_chunks > n_words) n_chunks = n_words;
      wpc = (n_words + n_chunks - 1u) / n_chunks;
      const uint32_t lds_cap_words = (48u * 1024u / 8u) / (n_planes ? n_planes : 1u);
      if (wpc > lds_cap_words) wpc = lds_cap_words;
      n_chunks = (n_words + wpc - 1u) / wpc;
    } else {
      n_chunks = 1;
      wpc = 1;
    }
    uint32_t* part_count = scratch + (size_t)n_chunks * R;
    if (n_words == 0) {
      hipMemsetAsync(part_first, 0xFF, sizeof(uint32_t) * R, s);
      hipMemsetAsync(part_count, 0, sizeof(uint32_t) * R, s);
    } else {
      const size_t lds = (size_t)n_planes * wpc * sizeof(uint64_t);
      hipLaunchKernelGGL(pair_sweep_planes_kernel<64>, dim3(rb, n_chunks), dim3(256), lds, s, row_sel, R, planes,
                         n_words, n_planes, wpc, part_first, part_count);
    }
    hipLaunchKernelGGL(pair_combine_kernel, dim3(rb), dim3(256), 0, s, part_first, part_count, R, n_chunks, first,
                       count);
  }
}

uint32_t pair_sweep_scratch_chunks(int variant, uint32_t R, uint32_t n_cols, uint32_t n_planes) {
  const uint32_t rb = (R + 255u) / 256u;
  if (rb == 0) return 1;
  if (variant == 1) {
    uint32_t n_chunks = (4096u + rb - 1u) / rb;
    return n_chunks + 1u;
  }
  const uint32_t n_words = (n_cols + 63u) / 64u;
  const uint32_t lds_cap_words = (48u * 1024u / 8u) / (n_planes ? n_planes : 1u);
  uint32_t by_lds = n_words / (lds_cap_words ? lds_cap_words : 1u) + 2u;
  uint32_t by_fill = (2048u + rb - 1u) / rb + 1u;
  return by_lds > by_fill ? by_lds : by_fill;
}

void launch_pair_select(int variant, const uint64_t* row_sel, uint32_t R, const uint64_t* col_mask,
                        const uint64_t* planes, uint32_t n_cols, uint32_t n_planes, const uint32_t* rank,
                        uint32_t* out, hipStream_t s) {
  if (R == 0) return;
  const uint32_t rb = (R + 255u) / 256u;
  if (variant == 1) {
    hipLaunchKernelGGL(pair_select_scalar_kernel, dim3(rb), dim3(256), 0, s, row_sel, R, col_mask, n_cols, rank, out);
  } else {
    const uint32_t n_words = (n_cols + 63u) / 64u;
    hipLaunchKernelGGL(pair_select_planes_kernel<64>, dim3(rb), dim3(256), 0, s, row_sel, R, planes, n_words,
                       n_planes, rank, out);
  }
}

void launch_newest(const int64_t* created_at, uint32_t T, uint32_t* idx_by_block, long long* val_by_block,
                   uint32_t n_blocks, hipStream_t s) {
  hipLaunchKernelGGL(newest_kernel, dim3(n_blocks), dim3(256), 0, s, created_at, T, (unsigned long long*)nullptr,
                     idx_by_block, val_by_block);
}

hipError_t launch_carve(const CarveArgs& a, size_t lds_bytes, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)carve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)PM_CARVE_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(carve_kernel, dim3(1), dim3(CARVE_THREADS), lds_bytes, s, a);
  return hipGetLastError();
}

}  // namespace pm
