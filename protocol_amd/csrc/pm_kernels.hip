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

// gfx950 only, and not as a formality: beside the CDNA4 instruction forms (DPP row shifts, ds_bpermute, s_memtime) the
// placement kernels' last-block hand-over (carve_prep_place_kernel, carve_elig_place_kernel) pairs a WORKGROUP-scope
// release with an agent-scope acquire.  That is less than the HSA memory model asks for; it holds here because what the
// last block reads of the others was written by agent-scope atomics, which on this target are performed at the shared
// L2 side once the issuing wave's vmcnt has drained (no-return atomics count under vmcnt, there is no vscnt), and
// because the engine never runs in tgsplit mode.  On a target without those properties the release must be agent-scope.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "pm_kernels.hip is written for gfx950 (MI355X): see the note on the placement kernels' fences above"
#endif

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

// stream triad a = b + 3 c over f64 (3 x 8 bytes per element): the measured HBM rate bench.py cites next to the
// nominal 8 TB/s
__global__ __launch_bounds__(256) void triad_kernel(const double* __restrict__ b, const double* __restrict__ c,
                                                    double* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) a[i] = b[i] + 3.0 * c[i];
}

// Per worker: cos(lat * pi/180) for the Haversine term, and the unit vector of the location for its chord form.
__device__ __forceinline__ void geo_of(double la, double lo, double* coslat, double* ux, double* uy, double* uz) {
  const double phi = la * PM_RAD, lam = lo * PM_RAD;
  const double c = cos(phi);
  *coslat = c;
  *ux = c * cos(lam);
  *uy = c * sin(lam);
  *uz = sin(phi);
}
__global__ __launch_bounds__(256) void geo_kernel(const double* __restrict__ lat, const double* __restrict__ lon,
                                                  double* __restrict__ coslat, double* __restrict__ ux,
                                                  double* __restrict__ uy, double* __restrict__ uz, uint32_t W) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w < W) geo_of(lat[w], lon[w], &coslat[w], &ux[w], &uy[w], &uz[w]);
}

// Row deltas of the worker table (discovery sync / status updater, orchestrator/src/discovery/monitor.rs:236-420,
// plugins/node_groups/status_update_impl.rs:8-39): n packed rows -> the SoA columns, cos(lat) refreshed.
__global__ __launch_bounds__(256) void update_rows_kernel(RowUpdateArgs p) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= p.n) return;
  const size_t n = p.n;
  const uint32_t w = p.u32_in[k];
  p.flags[w] = p.u32_in[1 * n + k];
  p.gpu_count[w] = p.u32_in[2 * n + k];
  p.gpu_mem[w] = p.u32_in[3 * n + k];
  p.gpu_cls[w] = p.u32_in[4 * n + k];
  p.cpu_cores[w] = p.u32_in[5 * n + k];
  p.ram[w] = p.u32_in[6 * n + k];
  p.storage[w] = p.u32_in[7 * n + k];
  p.addr_rank[w] = p.u32_in[8 * n + k];
  p.site[w] = p.u32_in[9 * n + k];
  const double la = p.lat_in[k], lo = p.lon_in[k];
  p.lat[w] = la;
  p.lon[w] = lo;
  geo_of(la, lo, &p.coslat[w], &p.ux[w], &p.uy[w], &p.uz[w]);
}

// ------------------------------------------------------------------------------------------------
// Phase B.  hit(row, col) = (row_sel[row] & col_mask[col]) != 0; per row: first hit and hit count.
// Both kernels split the swept axis over blockIdx.y so that a few thousand workgroups fill the chip, keep a
// row's partial result in registers and fold it into first[] / count[] with one atomicMin + atomicAdd per
// (row, split) that saw a hit (pair_init_kernel sets first = PM_NONE, count = 0 beforehand; min and sum are
// order-independent, so the result is deterministic).  With a single split the results are stored directly.

__global__ __launch_bounds__(256) void pair_init_kernel(uint32_t* __restrict__ first, uint32_t* __restrict__ count,
                                                        uint32_t R) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r < R) {
    first[r] = PM_NONE;
    count[r] = 0u;
  }
}

__device__ __forceinline__ void pair_fold(uint32_t* __restrict__ first, uint32_t* __restrict__ count, uint32_t r,
                                          uint32_t f, uint32_t cnt, bool atomic) {
  if (!atomic) {
    first[r] = f;
    count[r] = cnt;
  } else if (cnt) {
    atomicMin(&first[r], f);
    atomicAdd(&count[r], cnt);
  }
}

// Scalar kernel (sweep_variant 1): rows live in lanes, the swept axis is wave-uniform (s_load), one pair per
// compare.  Kept as the straightforward kernel the bit-sliced one is checked against.
__global__ __launch_bounds__(256) void pair_sweep_scalar_kernel(const uint64_t* __restrict__ row_sel, uint32_t R,
                                                                const uint64_t* __restrict__ col_mask,
                                                                uint32_t c_begin, uint32_t c_end, uint32_t chunk,
                                                                uint32_t* __restrict__ first_out,
                                                                uint32_t* __restrict__ count_out, uint32_t atomic) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  const uint32_t c0 = c_begin + blockIdx.y * chunk;
  const uint32_t c1 = min(c_end, c0 + chunk);
  const uint64_t sel = r < R ? row_sel[r] : 0ull;
  uint32_t first = PM_NONE, cnt = 0;
  for (uint32_t c = c0; c < c1; ++c) {
    const uint64_t m = col_mask[c];  // uniform address -> s_load
    const bool hit = (m & sel) != 0ull;
    cnt += hit;
    first = hit ? min(first, c) : first;
  }
  if (r < R) pair_fold(first_out, count_out, r, first, cnt, atomic != 0u);
}

// rank-th hit of a row (seeded chooser): second pass over the same pairs.
__global__ __launch_bounds__(256) void pair_select_scalar_kernel(const uint64_t* __restrict__ row_sel, uint32_t R,
                                                                 const uint64_t* __restrict__ col_mask,
                                                                 uint32_t c_begin, uint32_t c_end,
                                                                 const uint32_t* __restrict__ rank,
                                                                 uint32_t* __restrict__ out) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  const uint64_t sel = r < R ? row_sel[r] : 0ull;
  uint32_t want = r < R ? rank[r] : PM_NONE;
  uint32_t res = PM_NONE, seen = 0;
  for (uint32_t c = c_begin; c < c_end; ++c) {
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

// The swept axis lives in a fixed-capacity index space (the task table grows downwards from its capacity, see
// pm_engine.cpp): plane b is planes[b * stride .. ), word j holds columns 64j..64j+63; kernels work on a word
// range [w_begin, w_end) and report absolute column indices.
__global__ __launch_bounds__(256) void build_planes_kernel(const uint64_t* __restrict__ col_mask, uint32_t n_cols,
                                                           uint32_t w_begin, uint32_t w_end, uint32_t stride,
                                                           uint32_t n_planes, uint64_t* __restrict__ planes) {
  // one wave per 64-column word; lane l owns column 64*j + l; __ballot gives the plane word.
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t j = w_begin + ((blockIdx.x * 256u + threadIdx.x) >> 6);
  if (j >= w_end) return;
  const uint32_t c = j * 64u + lane;
  const uint64_t m = c < n_cols ? col_mask[c] : 0ull;
  for (uint32_t b = 0; b < n_planes; ++b) {
    const uint64_t word = __ballot((m >> b) & 1ull);
    if (lane == 0) planes[(size_t)b * stride + j] = word;
  }
  // plane n_planes: the OR of all of them — what a row that selects every plane (a task without topology
  // restriction: mask ~0) hits, in one read instead of n_planes
  const uint64_t valid = n_planes >= 64u ? ~0ull : ((1ull << n_planes) - 1ull);
  const uint64_t any = __ballot((m & valid) != 0ull);
  if (lane == 0) planes[(size_t)n_planes * stride + j] = any;
}

// grid = (ceil(R / 256), n_split): workgroup (x, y) sweeps its 256 rows over the plane words
// [y * words_per_split, (y + 1) * words_per_split), staged through LDS in pieces of words_per_piece.
template <uint32_t RPT>
__global__ __launch_bounds__(256) void pair_sweep_planes_kernel(const uint64_t* __restrict__ row_sel, uint32_t R,
                                                                const uint64_t* __restrict__ planes,
                                                                uint32_t stride, uint32_t w_begin, uint32_t w_end,
                                                                uint32_t n_planes, uint32_t words_per_split,
                                                                uint32_t words_per_piece,
                                                                uint32_t* __restrict__ first_out,
                                                                uint32_t* __restrict__ count_out, uint32_t atomic) {
  extern __shared__ uint64_t s_pl[];  // [n_planes + 1][lds_stride]: the planes of a piece and their OR
  // The lanes of a wave read DIFFERENT planes at the same word j: with a plane stride that is a multiple of the 64
  // LDS banks (x 4 B) all of those reads would land in one bank.  An odd stride (in 8-byte words) spreads them.
  const uint32_t lds_stride = words_per_piece | 1u;
  // RPT rows per thread (rows r0, r0 + 256, ...): a workgroup stages every plane of its word range into LDS once, so
  // with a million rows (the per-task orientation) four rows per thread quarter that traffic — 1.2 GB of L2 reads
  // at 1M x 100k otherwise
  const uint32_t r0 = blockIdx.x * (256u * RPT) + threadIdx.x;
  const uint32_t w0 = w_begin + blockIdx.y * words_per_split;
  const uint32_t w1 = min(w_end, w0 + words_per_split);
  const uint64_t valid = n_planes >= 64u ? ~0ull : ((1ull << n_planes) - 1ull);
  uint64_t sel[RPT], keep[RPT];
  uint32_t first[RPT], cnt[RPT], my_plane[RPT];
  bool single[RPT];
#pragma unroll
  for (uint32_t k = 0; k < RPT; ++k) {
    const uint32_t r = r0 + k * 256u;
    sel[k] = r < R ? (row_sel[r] & valid) : 0ull;
    if (sel[k] == valid && n_planes > 1u && n_planes < 64u) sel[k] = 1ull << n_planes;  // every plane selected: read their OR
    first[k] = PM_NONE;
    cnt[k] = 0;
    // wave-uniform: does any row of this wave select more than one plane?
    single[k] = __ballot((sel[k] & (sel[k] - 1ull)) != 0ull) == 0ull;
    my_plane[k] = sel[k] ? (uint32_t)__builtin_ctzll(sel[k]) : 0u;
    keep[k] = sel[k] ? ~0ull : 0ull;  // rows without a selector (workers outside groups) hit nothing
  }
  for (uint32_t j0 = w0; j0 < w1; j0 += words_per_piece) {
    const uint32_t nj = min(words_per_piece, w1 - j0);
    __syncthreads();  // the previous piece has been consumed
    for (uint32_t b = 0; b <= n_planes; ++b)
      for (uint32_t j = threadIdx.x; j < nj; j += 256u) s_pl[b * lds_stride + j] = planes[(size_t)b * stride + j0 + j];
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < RPT; ++k) {
      if (single[k]) {
        // every row of this wave selects at most one plane (the reference orientation: a group's configuration
        // bit; a task of one topology, or of all of them): one LDS read per word, no selector loop; four words per
        // step keep the reads in flight
        const uint64_t* pl = s_pl + my_plane[k] * lds_stride;
        uint32_t j = 0;
        for (; j + 4u <= nj; j += 4u) {
          const uint64_t h0 = pl[j] & keep[k], h1 = pl[j + 1u] & keep[k], h2 = pl[j + 2u] & keep[k], h3 = pl[j + 3u] & keep[k];
          cnt[k] += __popcll(h0) + __popcll(h1) + __popcll(h2) + __popcll(h3);
          if (first[k] == PM_NONE && (h0 | h1 | h2 | h3)) {
            const uint32_t u = h0 ? 0u : (h1 ? 1u : (h2 ? 2u : 3u));
            const uint64_t h = h0 ? h0 : (h1 ? h1 : (h2 ? h2 : h3));
            first[k] = (j0 + j + u) * 64u + __builtin_ctzll(h);
          }
        }
        for (; j < nj; ++j) {
          const uint64_t h = pl[j] & keep[k];
          cnt[k] += __popcll(h);
          if (h && first[k] == PM_NONE) first[k] = (j0 + j) * 64u + __builtin_ctzll(h);
        }
      } else {
        for (uint32_t j = 0; j < nj; ++j) {
          uint64_t hits = 0;
          uint64_t s = sel[k];
          while (s) {  // OR the planes this row selects
            const uint32_t b = __builtin_ctzll(s);
            s &= s - 1;
            hits |= s_pl[b * lds_stride + j];
          }
          cnt[k] += __popcll(hits);
          if (hits && first[k] == PM_NONE) first[k] = (j0 + j) * 64u + __builtin_ctzll(hits);
        }
      }
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < RPT; ++k) {
    const uint32_t r = r0 + k * 256u;
    if (r < R) pair_fold(first_out, count_out, r, first[k], cnt[k], atomic != 0u);
  }
}

__global__ __launch_bounds__(256) void pair_select_planes_kernel(const uint64_t* __restrict__ row_sel, uint32_t R,
                                                                 const uint64_t* __restrict__ planes,
                                                                 uint32_t stride, uint32_t w_begin, uint32_t w_end,
                                                                 uint32_t n_planes, const uint32_t* __restrict__ rank,
                                                                 uint32_t* __restrict__ out) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= R) return;
  const uint64_t valid = n_planes >= 64u ? ~0ull : ((1ull << n_planes) - 1ull);
  const uint64_t sel = row_sel[r] & valid;
  uint32_t want = rank[r];
  uint32_t res = PM_NONE;
  if (want != PM_NONE) {
    for (uint32_t j = w_begin; j < w_end; ++j) {
      uint64_t hits = 0, s = sel;
      while (s) {
        const uint32_t b = __builtin_ctzll(s);
        s &= s - 1;
        hits |= planes[(size_t)b * stride + j];
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

// wave-wide inclusive prefix sum in six DPP adds (row_shr 1, 2, 4, 8 inside the rows of 16, then the row ends carried
// over with row_bcast 15 / 31): no LDS traffic.  (Six __shfl_up are six ds_bpermute round trips, ~130 cycles each — a
// third of a pass of the bitmap sweep and of the ticketer's look.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_step(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v = dpp_add_step<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add_step<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add_step<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add_step<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add_step<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_add_step<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

// ---- task table deltas (the table grows downwards: new tasks sit in front of the old ones)
// live prefix: prefix[j] = live tasks in words [w_begin, j); one workgroup, 64-word passes
__global__ __launch_bounds__(64) void task_prefix_kernel(const uint64_t* __restrict__ live, uint32_t w_begin,
                                                         uint32_t w_end, uint32_t* __restrict__ prefix) {
  const uint32_t lane = threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t j0 = w_begin; j0 < w_end; j0 += 64u) {
    const uint32_t j = j0 + lane;
    const uint32_t cnt = j < w_end ? (uint32_t)__popcll(live[j]) : 0u;
    const uint32_t incl = wave_incl_scan_u32(cnt);
    if (j < w_end) prefix[j] = acc + incl - cnt;
    acc += __shfl(incl, 63, 64);
  }
}
// deleted tasks: clear the mask, the live bit and the task's bit in every plane
__global__ __launch_bounds__(256) void task_delete_kernel(const uint32_t* __restrict__ slots, uint32_t n,
                                                          uint64_t* __restrict__ tmask,
                                                          uint64_t* __restrict__ live, uint64_t* __restrict__ planes,
                                                          uint32_t stride, uint32_t n_planes) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  const uint32_t u = slots[k];
  const uint64_t bit = 1ull << (u & 63u);
  tmask[u] = 0ull;
  atomicAnd((unsigned long long*)&live[u >> 6], ~bit);
  for (uint32_t b = 0; b <= n_planes; ++b)  // (plane n_planes is the OR of the others)
    atomicAnd((unsigned long long*)&planes[(size_t)b * stride + (u >> 6)], ~bit);
}
// per-task results of the north_star orientation, from table slots to positions in get_all_tasks order
__global__ __launch_bounds__(256) void task_compact_kernel(const uint32_t* __restrict__ first_u,
                                                           const uint32_t* __restrict__ count_u, uint32_t u_begin,
                                                           uint32_t u_end, const uint64_t* __restrict__ live,
                                                           const uint32_t* __restrict__ prefix,
                                                           uint32_t* __restrict__ first_out,
                                                           uint32_t* __restrict__ count_out) {
  const uint32_t u = u_begin + blockIdx.x * 256u + threadIdx.x;
  if (u >= u_end) return;
  const uint64_t w = live[u >> 6];
  if (!((w >> (u & 63u)) & 1ull)) return;
  const uint32_t pos = prefix[u >> 6] + (uint32_t)__popcll(w & ((1ull << (u & 63u)) - 1ull));
  first_out[pos] = first_u[u - u_begin];
  count_out[pos] = count_u[u - u_begin];
}

// ------------------------------------------------------------------------------------------------
// Selector / claim kernels around the reference-orientation sweep (scheduler_impl.rs:11-110).

__device__ __forceinline__ uint64_t splitmix64_mix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- per-task orientation: tasks that name the same set of configurations have the same bidders, so the sweep runs
// once per DISTINCT topology mask (a few thousand at a million tasks) and every task reads its mask's result.
// An open-addressing table of the masks (key = mask, ~0 = empty): insert, number the used slots densely, look up.
#define PM_INTERN_EMPTY (~0ull)
__global__ __launch_bounds__(256) void task_intern_insert_kernel(const uint64_t* __restrict__ tmask,
                                                                 const uint64_t* __restrict__ live, uint32_t u_begin,
                                                                 uint32_t u_end, uint64_t valid,
                                                                 unsigned long long* __restrict__ keys, uint32_t h_mask,
                                                                 uint32_t* __restrict__ overflow) {
  const uint32_t u = u_begin + blockIdx.x * 256u + threadIdx.x;
  if (u >= u_end || !((live[u >> 6] >> (u & 63u)) & 1ull)) return;
  const uint64_t m = tmask[u] & valid;
  if (m == 0ull) return;  // (names no configuration that exists: no bidder)
  uint32_t h = (uint32_t)splitmix64_mix(m) & h_mask;
  for (uint32_t probe = 0; probe <= h_mask; ++probe, h = (h + 1u) & h_mask) {
    const unsigned long long k = keys[h];
    if (k == m) return;
    if (k == PM_INTERN_EMPTY) {
      const unsigned long long old = atomicCAS(&keys[h], PM_INTERN_EMPTY, (unsigned long long)m);
      if (old == PM_INTERN_EMPTY || old == m) return;
    }
  }
  *overflow = 1u;
}
__global__ __launch_bounds__(256) void task_intern_number_kernel(const unsigned long long* __restrict__ keys,
                                                                 uint32_t n_slots, uint32_t* __restrict__ vals,
                                                                 uint32_t* __restrict__ counter,
                                                                 uint64_t* __restrict__ umask, uint32_t cap_u) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= n_slots) return;
  const unsigned long long k = keys[s];
  if (k == PM_INTERN_EMPTY) return;
  const uint32_t id = atomicAdd(counter, 1u);
  vals[s] = id;
  if (id < cap_u) umask[id] = k;
}
// results per distinct mask -> per task, at the task's position in the caller's list
__global__ __launch_bounds__(256) void task_compact_class_kernel(const uint32_t* __restrict__ first_c,
                                                                 const uint32_t* __restrict__ count_c,
                                                                 const uint64_t* __restrict__ tmask, uint64_t valid,
                                                                 const unsigned long long* __restrict__ keys,
                                                                 const uint32_t* __restrict__ vals, uint32_t h_mask,
                                                                 uint32_t u_begin, uint32_t u_end,
                                                                 const uint64_t* __restrict__ live,
                                                                 const uint32_t* __restrict__ prefix,
                                                                 uint32_t* __restrict__ first_out,
                                                                 uint32_t* __restrict__ count_out) {
  const uint32_t u = u_begin + blockIdx.x * 256u + threadIdx.x;
  if (u >= u_end) return;
  const uint64_t w = live[u >> 6];
  if (!((w >> (u & 63u)) & 1ull)) return;
  const uint32_t pos = prefix[u >> 6] + (uint32_t)__popcll(w & ((1ull << (u & 63u)) - 1ull));
  const uint64_t m = tmask[u] & valid;
  uint32_t first = PM_NONE, count = 0u;
  if (m != 0ull) {
    uint32_t h = (uint32_t)splitmix64_mix(m) & h_mask;
    while (keys[h] != m) h = (h + 1u) & h_mask;  // (every live mask was inserted)
    const uint32_t id = vals[h];
    first = first_c[id];
    count = count_c[id];
  }
  first_out[pos] = first;
  count_out[pos] = count;
}

// row selector of worker w = the configuration bit of its group (0 when not in a group).  `rows` (optional):
// the sweep's rows are the workers rows[0..R) — the ones this rank owns in a multi-GPU tick — instead of 0..R).
__global__ __launch_bounds__(256) void worker_selector_kernel(const int32_t* __restrict__ group_of,
                                                              const uint32_t* __restrict__ g_cfg, uint32_t R,
                                                              const uint32_t* __restrict__ rows,
                                                              uint64_t* __restrict__ sel) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= R) return;
  const int32_t g = group_of[rows ? rows[r] : r];
  sel[r] = g >= 0 ? (1ull << g_cfg[g]) : 0ull;
}

// rank of the chosen task inside the applicable list (PM_CHOOSE_SEEDED): mix(seed ^ group id) % n.
__global__ __launch_bounds__(256) void chooser_rank_kernel(const int32_t* __restrict__ group_of,
                                                           const uint64_t* __restrict__ g_id,
                                                           const uint32_t* __restrict__ count, uint32_t R,
                                                           const uint32_t* __restrict__ rows, uint64_t seed,
                                                           uint32_t* __restrict__ rank) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= R) return;
  const int32_t g = group_of[rows ? rows[r] : r];
  const uint32_t n = count[r];
  rank[r] = (g >= 0 && n) ? (uint32_t)(splitmix64_mix(seed ^ g_id[g]) % n) : PM_NONE;
}

// Column mask of worker w for the per-task orientation: eligible (Healthy & p2p & unassigned,
// mod.rs:492-497) ? compat & enabled : 0.
// `shard` (optional): only the workers this rank owns bid (multi-GPU: the per-task bests are folded across ranks).
__global__ __launch_bounds__(256) void eligible_selector_kernel(const uint32_t* __restrict__ wflags,
                                                                const int32_t* __restrict__ group_of,
                                                                const uint64_t* __restrict__ compat, uint64_t enabled,
                                                                uint32_t W, const uint8_t* __restrict__ shard,
                                                                uint32_t my_rank, uint64_t* __restrict__ sel) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w >= W) return;
  const uint32_t f = wflags[w];
  const bool e = (f & PM_W_HEALTHY) && (f & PM_W_HAS_P2P) && group_of[w] < 0 && (!shard || shard[w] == my_rank);
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

__device__ __forceinline__ uint32_t task_position(const uint64_t* __restrict__ live, const uint32_t* __restrict__ prefix,
                                                  uint32_t u) {
  return prefix[u >> 6] + (uint32_t)__popcll(live[u >> 6] & ((1ull << (u & 63u)) - 1ull));
}

// Claim (SETNX, scheduler_impl.rs:74 / mod.rs:471-476) + publish row.  Every member of a group
// computed the same choice, so the group's task word is written with the same value by all.
__global__ __launch_bounds__(256) void claim_publish_kernel(ClaimArgs p) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= p.R) return;
  const uint32_t w = p.rows ? p.rows[r] : r;
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
      t = p.chosen[r];
      if (t != PM_NONE) p.g_task_next[g] = t;  // same value from every member
    }
    const uint32_t n = p.g_n[g], off = p.g_off[g];
    const uint32_t idx = p.rank_in_group[w];
    // published: the task's position in get_all_tasks order (live tasks in front of it in the table); a row that
    // goes into the multi-GPU exchange keeps the handle — table_scatter_kernel needs it for the group's claim
    a.task = (t == PM_NONE || p.rows) ? t : task_position(p.t_live, p.t_prefix, t);
    a.group_slot = (uint32_t)g;
    a.group_index = idx;
    a.group_size = n;
    a.next_worker = p.by_rank[off + ((idx + 1u == n) ? 0u : idx + 1u)];  // (idx + 1) % n
    a.group_id = p.g_id[g];
  }
  if (p.rows) {  // multi-GPU: this rank's rows, packed, into its segment of the exchange buffer
    p.table[r] = a;
  } else {
    p.table[w] = a;
    p.task_col[w] = a.task;
  }
}

// Multi-GPU: the all-gathered segments ([world][cap_t] rows, xrow[w] = where worker w's row landed) -> the full
// per-worker table, the compact task column and the groups' claimed-task words (every member of a group
// carries the same task, whichever rank computed its row).
__global__ __launch_bounds__(256) void table_scatter_kernel(const pm_assignment* __restrict__ x,
                                                            const uint32_t* __restrict__ xrow, uint32_t W,
                                                            pm_assignment* __restrict__ table,
                                                            uint32_t* __restrict__ task_col,
                                                            uint32_t* __restrict__ g_task_next,
                                                            const uint64_t* __restrict__ t_live,
                                                            const uint32_t* __restrict__ t_prefix) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w >= W) return;
  pm_assignment a = x[xrow[w]];
  if (a.group_slot != PM_NONE && a.task != PM_NONE) g_task_next[a.group_slot] = a.task;  // exchanged rows carry handles
  if (a.task != PM_NONE) a.task = task_position(t_live, t_prefix, a.task);
  table[w] = a;
  task_col[w] = a.task;
}

// ------------------------------------------------------------------------------------------------
// NewestTaskPlugin: argmax (created_at, index) — LDS-staged wavefront argmax, last max wins.

__global__ __launch_bounds__(256) void newest_kernel(const int64_t* __restrict__ created_at,
                                                     const uint64_t* __restrict__ live, uint32_t t_begin, uint32_t T,
                                                     unsigned long long* __restrict__ best_key,
                                                     uint32_t* __restrict__ best_idx_by_block,
                                                     long long* __restrict__ best_val_by_block) {
  __shared__ long long s_v[4];
  __shared__ uint32_t s_i[4];
  long long bv = INT64_MIN;
  uint32_t bi = PM_NONE;
  for (uint32_t t = t_begin + blockIdx.x * 256u + threadIdx.x; t < T; t += gridDim.x * 256u) {
    if (!((live[t >> 6] >> (t & 63u)) & 1ull)) continue;  // a deleted task's slot
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
// Carve kernel: the sequential part of the greedy group formation inside ONE workgroup (8 waves).  The
// greedy is a chain of dependent steps (group g+1's seed depends on what group g removed), so there is no
// cross-workgroup traffic to pay for: candidates are position-compacted, the alive/candidate bitmaps live
// in LDS.  The exact form of a step is
//   seed search (bitmap scan) -> Haversine term for every remaining candidate -> top-(max-1)
//   selection by (key, position) with a wavefront argmin staged through LDS -> commit;
// almost every step is instead served from the neighbour lists carve_propose_kernel computed on the whole
// chip (carve_chain / carve_fast_steps below), which only have to be filtered against the bitmap.
//
// Ordering key.  The reference sorts by d = 6371 * 2 * atan2(sqrt(a), sqrt(1-a)) computed with glibc
// libm (mod.rs:218-231).  d is a monotone function of a, so the kernel orders by a (f64, polynomial sin: sin_band)
// and proves the selection equal to the reference's: if every candidate whose a lies within a
// relative 2^-36 band around the last selected one has bit-identical coordinates (then the
// reference's distances tie exactly and the stable sort falls back to input order, like the
// kernel's (key, position) order), the selected SET is the reference's.  Otherwise the step is
// reported as UNCERTAIN and the engine settles exactly that step on the host with glibc.

// (16 waves compile — the whole kernel then has to fit 128 VGPRs: 103 spilled — and were measured in round 2, with the
// speculative rounds of that round: carve 3.57 ms instead of 2.47 at 100k x 10k, 24.3 instead of 19.2 ms at 1M x 100k)
#ifndef CARVE_WAVES
#define CARVE_WAVES 8
#endif
#define CARVE_THREADS (CARVE_WAVES * 64)
// a value that is the same in every lane, moved to an SGPR
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
// an LDS address as an opaque SGPR value (wave-uniform by construction): the compiler can neither re-derive it nor
// treat it as a vector value
template <typename LP, typename GP>
__device__ __forceinline__ LP lds_pin(GP* g) {
  LP p = (LP)g;
  uint32_t a = (uint32_t)(uintptr_t)p;
  a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
  asm volatile("" : "+s"(a));
  return (LP)(uintptr_t)a;
}
// The carve kernels take their argument block through a pointer (see carve_kernel), so the compiler cannot
// see that the pointers inside it are global memory and would emit FLAT accesses — which count against the
// LDS counter as well and serialise every LDS wait behind the outstanding HBM traffic.  G() restores the
// address space at the point of use.
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T* G(T* q) {
  return (__attribute__((address_space(1))) T*)q;
}

// ---- wave-wide unsigned min via DPP (no LDS traffic): row_shr 1,2,4,8 -> row_bcast15 -> row_bcast31,
// result broadcast from lane 63 with readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min32_step(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, CTRL, ROW_MASK, 0xF, false);
  return o < v ? o : v;  // folds into one v_min_u32 with a DPP operand
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = dpp_min32_step<0x111, 0xF>(v);  // row_shr:1
  v = dpp_min32_step<0x112, 0xF>(v);  // row_shr:2
  v = dpp_min32_step<0x114, 0xF>(v);  // row_shr:4
  v = dpp_min32_step<0x118, 0xF>(v);  // row_shr:8
  v = dpp_min32_step<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_min32_step<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit minimum as two 32-bit reductions (high words, then the low words of the lanes that hold the minimal
// high word): a 64-bit compare-and-select per DPP step costs about three times as many instructions
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
  const uint32_t hi = (uint32_t)(v >> 32);
  const uint32_t mh = wave_min_u32(hi);
  const uint32_t ml = wave_min_u32(hi == mh ? (uint32_t)v : 0xFFFFFFFFu);
  return ((uint64_t)mh << 32) | ml;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t l) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
}

// barrier for exchanges that go through LDS only (no global-memory drain)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BlockRed {
  uint32_t a[CARVE_WAVES];
  uint32_t b[CARVE_WAVES];
  uint32_t part_n[CARVE_WAVES];
  uint32_t flag[CARVE_WAVES];
  // mailbox: wave 0 -> workgroup after a run of fast (proposal) steps
  uint32_t f_action, f_n_cand, f_total_available, f_n_groups, f_mem_off, f_steps, f_fast, f_pad;
  unsigned long long f_cand_sum;
  unsigned long long _spare0;
  uint32_t _spare1, _spare2;
};

// sin on [-pi/2, pi/2] as an odd Taylor polynomial to x^19 (|rel err| < 1e-15 there).  Half a longitude difference
// beyond 180 degrees lies in (pi/2, pi]: reflected, sin(x) = sin(pi - x) with pi in two pieces (the reflection is
// exact to 1e-32) — those are two of every five pairs of a world-wide swarm, and the OCML path they used to take
// (argument reduction with a table in memory) cost a wave more than everything else in a step of stream_small or a
// sweep of carve_exact_step.  Only what is no difference of two longitudes still goes there.  The certificate band
// (2^-35) is four orders of magnitude wider than this error.
__device__ __forceinline__ double sin_band(double x) {
  double ax = fabs(x);
  if (ax > 1.5707963267948966) {
    if (ax > 3.2) return sin(x);
    ax = (3.141592653589793116 - ax) + 1.2246467991473532e-16;
    x = x < 0.0 ? -ax : ax;
  }
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

// The proposer's Haversine term.  a = sin^2(dphi/2) + cos cos sin^2(dlam/2) is, exactly, a quarter of the squared
// chord between the two unit vectors: a = |u1 - u2|^2 / 4 — three subtractions, a multiply and two fma instead of
// two sine polynomials.  The unit vectors carry an absolute error of ~2e-16 per component, so the chord form has a
// relative error of ~7e-16 / sqrt(a): under 1e-12 — a fifteenth of the certificate band (2^-36) — for a >=
// PM_A_CHORD_MIN (about 10 km), and that is where it is used; nearer candidates (rare: a handful per seed) take the
// sine form, whose error is independent of the distance.  Every path of the proposer goes through this one
// function, so a candidate's key is the same bit pattern wherever it is computed.
struct SeedGeo {
  double lat, lon, cos, ux, uy, uz;
};
template <typename DP>
__device__ __forceinline__ double prox_a(const SeedGeo& s, double ux, double uy, double uz, DP lat, DP lon, DP cs, uint32_t t) {
  const double dx = ux - s.ux, dy = uy - s.uy, dz = uz - s.uz;
  const double a = 0.25 * fma(dx, dx, fma(dy, dy, dz * dz));
  if (a >= PM_A_CHORD_MIN) return a;
  return hav_a(s.lat, s.lon, s.cos, lat[t], lon[t], cs[t]);
}

template <typename P>
__device__ __forceinline__ bool bit_at(P b, uint32_t i) { return (b[i >> 6] >> (i & 63u)) & 1ull; }

// Ordering key of a candidate: the f64 bits of its Haversine term `a` with the low SLOT_BITS replaced by
// the slot number (slot order == input order), so one u64 compare is the whole (distance, input order)
// comparison.  Dropping SLOT_BITS mantissa bits is covered by the certificate band.
__device__ __forceinline__ uint64_t pack_key(uint64_t key_bits, uint32_t slot, uint32_t slot_bits) {
  return ((key_bits >> slot_bits) << slot_bits) | slot;
}

enum { STEP_CONTINUE = 0, STEP_BREAK = 1, STEP_UNCERTAIN = 2, STEP_OVERFLOW = 3, STEP_ABORT = 4 };

#ifdef PM_CARVE_PROF
#define PROF_DECL uint64_t prof_t0 = __builtin_amdgcn_s_memtime()
#define PROF_MARK(slot)                                                     \
  do {                                                                      \
    const uint64_t t_ = __builtin_amdgcn_s_memtime();                       \
    if (threadIdx.x == 0) G(p.status)->prof[slot] += t_ - prof_t0;  \
    prof_t0 = t_;                                                           \
  } while (0)
#else
#define PROF_DECL
#define PROF_MARK(slot)
#endif

struct StepCtx {
  uint32_t mode, proximity, min_s, max_s, cfg;
  uint32_t n_list;       // slots of the current list
  uint32_t n_cand;       // live slots
  uint32_t n_start;      // live slots when the validation of this batch started
  uint32_t n_groups, mem_off;
  uint32_t total_available;
  uint32_t steps, fast_steps;
  unsigned long long cand_sum;
  // proposals (0 = none)
  uint32_t prop_k, prop_limit;
  uint32_t rows_pr;  // rows per rank in the proposal buffer for this batch: ceil(seeds / world)
  uint32_t n_seeds;  // seeds of the batch
  bool use_props;
  // packed-key geometry of the current list: low slot_bits of a key hold the slot; certificate band
  uint32_t slot_bits;
  double band;
  bool big;  // per-slot arrays live in HBM/L2 (list above PM_CARVE_SLOTS), bitmaps + staged rows in LDS
};

__device__ __forceinline__ void ctx_set_geometry(StepCtx& c) {
  c.big = c.n_list > PM_CARVE_SLOTS;
  c.slot_bits = c.big ? PM_CARVE_SLOT_BITS_BIG : PM_CARVE_SLOT_BITS;
  c.band = c.big ? PM_TIE_BAND_BIG : PM_TIE_BAND;
}

enum { FAST_DONE = 0, FAST_SLOW = 1, FAST_OVERFLOW = 2, FAST_AGAIN = 3, FAST_REPROPOSE = 4, FAST_SEQ = 5, FAST_WIDEN = 6,
       FAST_ABORT = 7, FAST_TAIL = 8, FAST_TINY = 9, FAST_RANOUT = 10 };

#ifdef PM_CARVE_PROF_FINE
#define PROF_COUNT(slot) do { if (lane == 0) G(p.status)->prof[slot] += 1; } while (0)
#else
#define PROF_COUNT(slot)
#endif
#define FAST_RETURN(code) do { c_ref = c; seed_cur = cur; return (code); } while (0)
#ifdef PM_CARVE_PROF  // why a step went to the exact sweep: 20 no proposal, 21 debug hook, 25 row exhausted, 31 certificate
#define SLOW_RETURN(why) do { if (lane == 0) G(p.status)->prof[why] += 1; FAST_RETURN(FAST_SLOW); } while (0)
#else
#define SLOW_RETURN(why) FAST_RETURN(FAST_SLOW)
#endif

// The proposals of a batch as the validating wave sees them: seed number i (rank among the live located slots below
// prop_limit at preparation time, ascending slot order) -> its slot (seed_slots, dense) and its row.
__device__ __forceinline__ uint32_t prop_row_of(uint32_t i, uint32_t world, uint32_t rows_pr) {
  return world > 1u ? (i % world) * rows_pr + i / world : i;
}

// Fast steps one at a time, executed by wave 0 alone while the other waves are parked at a barrier: the expensive
// part of a step (keys for every live candidate + top-k) was done for every possible seed by
// carve_propose_kernel against the live set at the start of the batch.  Because candidates are
// only ever REMOVED, the reference's sorted remaining list is the proposal row minus the dead entries,
// as long as the row still holds enough live entries and the boundary can be certified; otherwise
// the step is handed to the exact full sweep (FAST_SLOW).  The whole wave looks at one row (lane = entry), so this
// path sees all 63 entries and re-derives the certificate from the keys; it takes what the in-order chain
// (carve_chain) hands over — rows that are not certified wholesale, thinned-out rows, wide groups, the last
// partial group of a configuration — and the first-come tail.  At most max_steps steps (FAST_AGAIN when reached).
// STREAM (carve_stream_kernel): the rows arrive per TICKET — seed_cur is the ticket the chain stopped at (PM_NONE =
// none), sl_words the validator's ticket table in LDS (StreamLds), a row lives in the granule rings stream_row_lo / _hi
// under the ticket's tag.
template <bool BIG, bool STREAM = false>
__device__ __noinline__ int carve_fast_steps(const CarveArgs& p, StepCtx& c_ref, const uint32_t* l_site,
                                             uint64_t* l_alive, const uint64_t* l_loc, uint32_t steps_before,
                                             uint32_t& seed_cur, uint32_t max_steps, uint32_t* sl_words = nullptr) {
  StepCtx c = c_ref;  // registers for the whole run (the reference lives in the caller's scratch frame)
  const uint32_t lane = threadIdx.x & 63u;
  // argument-block fields used per step, loaded once: the stores below go through flat pointers the compiler
  // must assume may alias the block itself
  const auto members = G(p.members);
  const auto g_cfg = G(p.g_cfg);
  const auto g_n = G(p.g_n);
  const auto g_off = G(p.g_off);
  const auto prop = G((const uint64_t*)p.prop);
  const auto seed_slots = G((const uint32_t*)p.seed_slots);
  const uint32_t cap_groups = p.cap_groups, cap_members = p.cap_members;
  const uint32_t dbg_every = p.debug_uncertain_every;
  const uint32_t world = UNI(p.dist_world), rows_pr = UNI(c.rows_pr), n_seeds = UNI(c.n_seeds);
  constexpr uint32_t SB = BIG ? PM_CARVE_SLOT_BITS_BIG : PM_CARVE_SLOT_BITS;
  constexpr uint64_t SLOT_MASK = (1ull << SB) - 1ull;
  constexpr uint64_t noloc_key = (PM_KEY_NOLOC >> SB) << SB;
  constexpr double band_rel = BIG ? PM_TIE_BAND_BIG : PM_TIE_BAND;
  typedef __attribute__((address_space(3))) unsigned long long lds_u64;
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  lds_u64* const A = (lds_u64*)l_alive;
  const lds_u64* const LOC = (const lds_u64*)l_loc;
  const lds_u32* const SITE3 = (const lds_u32*)l_site;
  auto alive_at = [A](uint32_t i) -> bool { return (A[i >> 6] >> (i & 63u)) & 1ull; };
  auto kill = [A](uint32_t i) {
    __hip_atomic_fetch_and(&A[i >> 6], ~(1ull << (i & 63u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // (streaming carve: the proposers' copy of the candidate bitmap follows every removal)
  const auto candg = G((unsigned long long*)p.bits_scratch);
  auto mirror = [candg](uint32_t i) {
    if (STREAM) __hip_atomic_fetch_and(&candg[i >> 6], ~(1ull << (i & 63u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // (members are recorded as SLOTS and translated to worker ids by one parallel pass after the run)
  auto site_of = [SITE3, l_site](uint32_t sl) -> uint32_t { return BIG ? l_site[sl] : SITE3[sl]; };
  const uint32_t lw = (c.n_list + 63u) >> 6;
  uint32_t cur = UNI(seed_cur);  // seeds in front of it are dead for good (they are consumed in ascending order)
  uint32_t fc_j = 0;             // first bitmap word that may still hold a live slot (first-come steps)
  uint32_t done = 0;
  PROF_COUNT(20);  // calls
  for (;;) {
    if (done >= max_steps) FAST_RETURN(FAST_AGAIN);
    if (!(c.total_available >= c.min_s && c.n_cand >= c.min_s && c.n_cand > 0)) FAST_RETURN(FAST_DONE);
    // ---- seed (mod.rs:526-530): the first live located slot = the first live entry of the batch's seed list
    uint32_t f_loc = PM_NONE;
    if (c.proximity) {
      if (STREAM) {
        // Tickets are issued in ascending position order to the located candidates alive at that moment, and
        // consumed in order: the first ticket from `cur` on whose seed is still alive names the first live located
        // candidate.  Behind the last ticket issued, the bitmaps say.
        typedef __attribute__((address_space(3))) uint32_t sl_u32;
        const sl_u32* const TP = (const sl_u32*)sl_words;
        const uint32_t t_req = UNI(TP[PM_STREAM_SLW_TREQ]);
        const uint32_t ci_now = (UNI(TP[PM_STREAM_SLW_PAY]) >> 18) & 63u;  // (tickets of other configurations are skipped)
        while (cur != PM_NONE && cur < t_req) {
          const uint32_t tp = UNI(TP[cur & (PM_STREAM_TP - 1u)]);
          const uint32_t pos = tp & 0x3FFFFu;
          if ((tp >> 18) == ci_now && alive_at(pos)) {
            f_loc = pos;
            break;
          }
          ++cur;
        }
        if (f_loc == PM_NONE) {
          cur = PM_NONE;
          for (uint32_t j0 = 0; j0 < lw; j0 += 64u) {
            const uint32_t j = j0 + lane;
            const uint64_t ll = j < lw ? (A[j] & LOC[j]) : 0ull;
            const uint64_t nz = __ballot(ll != 0ull);
            if (nz) {
              const int src = __builtin_ctzll(nz);
              const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ll >> 32), src) << 32) |
                                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ll, src);
              f_loc = (j0 + src) * 64u + __builtin_ctzll(w);
              break;
            }
          }
        }
      } else if (c.prop_k) {
        while (cur < n_seeds) {
          const uint32_t i = cur + lane;
          const uint32_t sl = seed_slots[i < n_seeds ? i : n_seeds - 1u];
          const bool al = i < n_seeds && alive_at(sl);
          const uint64_t m = __ballot(al);
          if (m) {
            const int l = __builtin_ctzll(m);
            cur += (uint32_t)l;
            f_loc = (uint32_t)__builtin_amdgcn_readlane((int)sl, l);
            break;
          }
          cur += 64u;
        }
        if (cur > n_seeds) cur = n_seeds;
        if (f_loc == PM_NONE && c.prop_limit < c.n_list) {
          // every proposed slot is used up; located candidates beyond the proposal batch need a new round
          bool more = false;
          for (uint32_t j0 = c.prop_limit >> 6; j0 < lw && !more; j0 += 64u) {
            const uint32_t j = j0 + lane;
            uint64_t ll = j < lw ? (A[j] & LOC[j]) : 0ull;
            if (j == (c.prop_limit >> 6)) ll &= ~((1ull << (c.prop_limit & 63u)) - 1ull);
            more = __ballot(ll != 0ull) != 0ull;
          }
          if (more) {
            if (lane == 0) G(p.status)->why[1] += 1u;
            FAST_RETURN(FAST_REPROPOSE);
          }
        }
      } else {
        for (uint32_t j0 = 0; j0 < lw; j0 += 64u) {
          const uint32_t j = j0 + lane;
          const uint64_t ll = j < lw ? (A[j] & LOC[j]) : 0ull;
          const uint64_t nz = __ballot(ll != 0ull);
          if (nz) {
            const int src = __builtin_ctzll(nz);
            const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ll >> 32), src) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ll, src);
            f_loc = (j0 + src) * 64u + __builtin_ctzll(w);
            break;
          }
        }
      }
    }
    const uint32_t want = c.max_s - 1u < c.n_cand - 1u ? c.max_s - 1u : c.n_cand - 1u;  // mod.rs:545-551
    if (c.n_groups >= cap_groups || c.mem_off + want + 1u > cap_members) FAST_RETURN(FAST_OVERFLOW);

    if (STREAM && f_loc == PM_NONE) FAST_RETURN(FAST_TAIL);  // (the whole workgroup drains the tail: stream_first_come)
    if (f_loc == PM_NONE) {
      // no located candidate (or proximity off): the group is the first `want + 1` live slots in input
      // order (mod.rs:553-561; a seed without location makes the sort a no-op, :238).
      // Once there is no located candidate there never will be one again, so the rest of the configuration
      // is drained right here; these steps always take the lowest live slots, every word below fc_j is
      // empty for good, and the scan resumes where it stopped.
      uint32_t n_cand = UNI(c.n_cand), total_av = UNI(c.total_available), n_groups = UNI(c.n_groups),
               mem_off = UNI(c.mem_off), steps = 0;
      const uint32_t min_s = UNI(c.min_s), max_s = UNI(c.max_s), cfg = UNI(c.cfg);
      unsigned long long cand_sum = 0;
      int ret = FAST_DONE;
      for (;;) {
        if (!(total_av >= min_s && n_cand >= min_s && n_cand > 0u)) break;
        const uint32_t need = max_s < n_cand ? max_s : n_cand;  // want + 1 (mod.rs:545-551)
        if (n_groups >= cap_groups || mem_off + need > cap_members) {
          ret = FAST_OVERFLOW;
          break;
        }
        uint32_t cnt = 0;
        for (uint32_t j = fc_j; j < lw && cnt < need; ++j) {
          const uint64_t w = A[j];
          const uint32_t w_lo = UNI((uint32_t)w), w_hi = UNI((uint32_t)(w >> 32));
          if (!(w_lo | w_hi)) continue;
          fc_j = j;
          // the lowest (need - cnt) set bits of w: rank of each set bit within the word, one ballot
          const uint32_t rk = __builtin_amdgcn_mbcnt_hi(w_hi, __builtin_amdgcn_mbcnt_lo(w_lo, 0u));
          const bool mine = ((w >> lane) & 1ull) && rk < need - cnt;
          const uint64_t take = __ballot(mine);
          if (mine) members[mem_off + cnt + rk] = j * 64u + lane;
          if (lane == 0) {
            A[j] = w & ~take;
            if (STREAM && take) __hip_atomic_fetch_and(&candg[j], ~take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          cnt += __popcll(take);
        }
        if (lane == 0) {
          g_cfg[n_groups] = cfg;
          g_n[n_groups] = cnt;
          g_off[n_groups] = mem_off;
        }
        n_groups += 1;
        mem_off += cnt;
        cand_sum += n_cand;
        n_cand -= cnt;
        total_av -= cnt;
        steps += 1;
        PROF_COUNT(21);  // first-come steps
      }
      c.n_groups = n_groups;
      c.mem_off = mem_off;
      c.cand_sum += cand_sum;
      c.n_cand = n_cand;
      c.total_available = total_av;
      c.steps += steps;
      c.fast_steps += steps;
      FAST_RETURN(ret);
    }

    PROF_COUNT(22);  // located sequential steps (attempts)
    const uint32_t seed = f_loc;
    if (c.prop_k == 0 || seed >= c.prop_limit || (STREAM && cur == PM_NONE)) {  SLOW_RETURN(20); }
    if (dbg_every && ((steps_before + c.steps + 1u) % dbg_every) == 0u) SLOW_RETURN(21);

    // ---- the seed's neighbour row: one packed key per lane, ascending.  Candidates with the seed's exact
    // coordinates are at distance 0 and head the row in slot order (only those behind the seed are listed: a live one
    // in front of it would have been the seed).
    const size_t rbase = STREAM ? (size_t)(cur & (PM_STREAM_RQ - 1u)) * 64u : (size_t)prop_row_of(cur, world, rows_pr) * PM_PROP_ROW;
    uint32_t nk_word;
    if (STREAM) {  // (granule 0: the flags word under the ticket's tag — a row that never arrived does not carry it)
      const unsigned long long g0 = __hip_atomic_load(&G(p.stream_row_lo)[rbase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (UNI((uint32_t)(g0 >> 32)) != UNI(p.stream_tag0) + cur) SLOW_RETURN(20);
      nk_word = UNI((uint32_t)g0);
    } else {
      nk_word = UNI((uint32_t)prop[rbase]);  // the row's flags word
    }
    const uint32_t n_k = nk_word & 0xFFu;
    const bool complete = (nk_word & PM_ROW_COMPLETE) != 0u;
    const bool tail_ok = (nk_word & PM_ROW_TAIL_OK) != 0u;
    const bool row_clean = (nk_word & PM_ROW_CLEAN) != 0u;
    const bool tail_clear = (nk_word & PM_ROW_TAIL_CLEAR) != 0u;
    const bool row_safe = (nk_word & PM_ROW_SAFE) != 0u;
    uint64_t e;
    if (STREAM) {  // entry `lane` = granules lane + 1 of the two rings (low and high half of the packed key)
      const uint32_t gi = (lane + 1u) & 63u;
      const unsigned long long lo = __hip_atomic_load(&G(p.stream_row_lo)[rbase + gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long hi = __hip_atomic_load(&G(p.stream_row_hi)[rbase + gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t tag = UNI(p.stream_tag0) + cur;
      const bool ok = lane >= n_k || ((uint32_t)(lo >> 32) == tag && (uint32_t)(hi >> 32) == tag);
      if (__ballot(!ok)) SLOW_RETURN(20);
      e = lane < n_k ? ((hi << 32) | (lo & 0xFFFFFFFFull)) : ~0ull;
    } else {
      e = lane < n_k ? prop[rbase + 1u + lane] : ~0ull;
    }
    const uint32_t slot = (uint32_t)(e & SLOT_MASK);
    const bool alive = lane < n_k && alive_at(slot);
    const uint64_t am = __ballot(alive);
    const uint32_t rank = __popcll(am & ((1ull << lane) - 1ull));
#ifdef PM_CARVE_PROF  // (timeline of the streaming carve: a row that ran out — entries, live ones, wanted, flags)
    if (STREAM && (uint32_t)__popcll(am) < want && lane == 0 && p.stream_trace) {
      const uint32_t ti_ = atomicAdd(&p.stream_ctl[SC_TRACE], 1u);
      if (ti_ < PM_STREAM_TRACE_CAP) {
        p.stream_trace[2u * ti_] = __builtin_amdgcn_s_memtime();
        p.stream_trace[2u * ti_ + 1u] = 14ull | ((unsigned long long)((n_k | ((uint32_t)__popcll(am) << 8) | (want << 16)) & 0xFFFFFFu) << 8) |
                                        ((unsigned long long)(nk_word >> 16) << 32);
      }
    }
#endif
    if ((uint32_t)__popcll(am) < want) {  // row exhausted by earlier groups
      if (STREAM) {  // (the streaming carve wants to know: the rows requested along with this one are as old)
#ifdef PM_CARVE_PROF
        if (lane == 0) G(p.status)->prof[25] += 1;
#endif
        FAST_RETURN(FAST_RANOUT);
      }
      SLOW_RETURN(25);
    }
    const bool sel = alive && rank < want;
    // the proposer certified the whole row (clean, safe) and its tail (complete / tail_clear): nothing left to prove
    if (want > 0 && !(row_clean && row_safe && (complete || tail_clear))) {
      const uint64_t lm = __ballot(sel && rank == want - 1u);
      const int lane_m = __builtin_ctzll(lm);
      const uint64_t e_m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e >> 32), lane_m) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e, lane_m);
      const uint64_t kb_m = (e_m >> SB) << SB;
      if (kb_m != noloc_key) {
        // exactness certificate: every live candidate of the row within the band AROUND the last selected one
        // — selected or not — must sit at its site (then the reference's distances tie exactly there and slot
        // order decides).  The test is symmetric: a selected entry of another site just below the boundary and
        // an unselected one of the boundary's site just above it may be ordered either way by the reference.
        const double a_m = __longlong_as_double((long long)kb_m);
        const double band = a_m * band_rel + 1e-300;
        if (a_m > PM_A_MAX_SAFE) SLOW_RETURN(31);
        const uint32_t site_m = site_of((uint32_t)(e_m & SLOT_MASK));
        const uint64_t kb = (e >> SB) << SB;
        const bool near = alive && kb != noloc_key && fabs(__longlong_as_double((long long)kb) - a_m) <= band;
        if (__ballot(near && site_of(slot) != site_m)) {  SLOW_RETURN(31); }
        if (!complete) {
          // candidates beyond the list are >= its last entry: either that entry clears the band, or it sits
          // at e_m's site and the proposer verified (tail_ok) that everything unlisted within the band of
          // the last entry is at that site too (then those tie exactly and have larger slots)
          const int last_l = (int)n_k - 1;
          const uint64_t e_l = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e >> 32), last_l) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e, last_l);
          const uint64_t kb_l = (e_l >> SB) << SB;
          if (kb_l != noloc_key && (__longlong_as_double((long long)kb_l) - a_m) <= band) {
            if (!(tail_ok && site_of((uint32_t)(e_l & SLOT_MASK)) == site_m)) SLOW_RETURN(31);
          }
        }
      }
    }
    // ---- commit (create_group_atomically mod.rs:299-322; healthy_nodes.retain :585): the seed, then the row's
    // first `want` live entries in key order
    if (sel) {
      kill(slot);
      mirror(slot);
      members[c.mem_off + 1u + rank] = slot;
    }
    if (lane == 0) {
      kill(seed);
      mirror(seed);
      members[c.mem_off] = seed;
      g_cfg[c.n_groups] = c.cfg;
      g_n[c.n_groups] = want + 1u;
      g_off[c.n_groups] = c.mem_off;
    }
    c.n_groups += 1;
    c.mem_off += want + 1u;
    c.cand_sum += c.n_cand;
    c.n_cand -= want + 1u;
    c.total_available -= want + 1u;
    c.steps += 1;
    c.fast_steps += 1;
    cur += 1u;  // this seed is dead now
    ++done;
  }
}

// The chain of located steps of a proposal batch, one seed after the other — no speculation: what makes a step cheap
// is that nothing but the step's own dependency is on its critical path.  A step depends on its predecessors through
// the alive bitmap only (row -> live bits -> kill: one LDS read, one ballot, one LDS atomic), and LDS operations of
// one wave execute in order, so the next seed's read is issued right behind this seed's kill without waiting for
// it.  A lone wave issues an instruction every eight or nine cycles, so the step is cut down to that dependency and
// everything else is done by two other waves of the workgroup, a pipeline through LDS rings (no barrier; the waves
// poll a few control words):
//   wave 1, the producer:  walks the batch's seed list, takes the entries whose seed is alive at that moment,
//                          requests their proposal rows (lane = row entry, lane 0 = the seed itself; 256 coalesced
//                          bytes per row, two blocks of CHAIN_BLOCK rows in flight) and parks them, digested — per
//                          lane the LDS address of the slot's bitmap word and its bit — in a ring;
//   wave 0, the chain:     per ring entry: the live lanes of the row (one read, one ballot); a dead seed (absorbed by
//                          a group since its row was requested) is no step; otherwise the seed and its first `want`
//                          live entries are killed (one atomic) and the live mask is passed on;
//   wave 2, the collector: re-derives the selection from the live mask, collects the members in LDS (the key array is
//                          idle while the chain runs) and writes them out with the group records, then hands the ring
//                          entry back to the producer.
// Because candidates are only ever REMOVED, the reference's sorted remaining list (mod.rs:234-255) is the proposal
// row minus its dead entries; the group is the seed plus the row's first `want` live entries (mod.rs:545-561).
// Full groups only (want = max_s - 1), and only rows whose flags word settles the certificate wholesale.  Everything
// else is handed to carve_fast_steps, which looks at the seed at seed_cur with the row's keys: other rows, exhausted
// rows (-> exact sweep), the debug hook (FAST_SLOW: exactly one step), the last partial group, the first-come tail,
// the end of the batch (FAST_SEQ).
// a proposal batch ends (the list is compacted and re-proposed) once fewer than 1 / PM_THIN_DIV of the slots that were
// alive when its validation started are left; a batch prepared beside the one in front of it is not worth validating
// when less than 1 / PM_STALE_DIV of its list is still alive (the next one is prepared from the state as it is then)
#ifndef PM_THIN_DIV
#define PM_THIN_DIV 3u
#endif
#ifndef PM_STALE_DIV
#define PM_STALE_DIV 3u
#endif
#ifndef CHAIN_BLOCK
#define CHAIN_BLOCK 8u
#endif
#ifndef CHAIN_RING
#define CHAIN_RING 64u                                 // rows parked in LDS (a power of two, >= 2 blocks; 32 until the
#endif                                                 // streaming carve's four parkers filled it faster than it drained)
// LDS layout of the chain inside the (idle) key array, in 32-bit words: per ring entry and lane (bitmap word address,
// bit) as one 64-bit word and the slot; per ring entry the seed number, the digested flags and the live mask; the
// control block; then the member buffer
#define CHAIN_RING_WORDS (3u * CHAIN_RING * 64u + 5u * CHAIN_RING + 16u)
#define CHAIN_STAGE_WORDS (PM_CARVE_SLOTS * 2u - CHAIN_RING_WORDS)
// control words (u32 index into the control block)
enum { CC_HEAD = 0, CC_CRIT = 1, CC_TAIL = 2, CC_CMD = 3, CC_ACK1 = 4, CC_ACK2 = 5, CC_START = 6, CC_DONE = 7,
       CC_ABORT = 8, CC_G0 = 9, CC_M0 = 10 };
enum { CH_RUN = 1u, CH_STOP = 2u, CH_QUIT = 3u };  // low two bits of a command word (the rest: sequence number)
#define CHAIN_SPIN_LIMIT (1u << 22)  // polls before a wait gives up (a lost hand-shake must never hang the GPU)
typedef __attribute__((address_space(3))) uint32_t chain_lds_u32;
typedef __attribute__((address_space(3))) unsigned long long chain_lds_u64;
__device__ __forceinline__ uint32_t cc_ld(chain_lds_u32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void cc_st(chain_lds_u32* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
struct ChainLds {
  chain_lds_u64 *RAB, *Q;
  chain_lds_u32 *RE, *RI, *RM, *RS, *CC, *STAGE;  // (RS: streaming carve — entry q of a run is parked when RS[q % R] == q + 1)
};
__device__ __forceinline__ ChainLds chain_lds(uint32_t* l_buf) {
  ChainLds L;
  L.RAB = lds_pin<chain_lds_u64*>(l_buf);
  L.Q = L.RAB + CHAIN_RING * 64u;
  L.RE = (chain_lds_u32*)(L.Q + CHAIN_RING);
  L.RI = L.RE + CHAIN_RING * 64u;
  L.RM = L.RI + CHAIN_RING;
  L.RS = L.RM + CHAIN_RING;
  L.CC = L.RS + CHAIN_RING;
  L.STAGE = L.CC + 16u;
  return L;
}
// wait for a command word other than `seen`; 0 = gave up (abort)
__device__ __forceinline__ uint32_t chain_wait_cmd(const ChainLds& L, uint32_t seen) {
  uint32_t cmd, spins = 0u;
  while ((cmd = cc_ld(&L.CC[CC_CMD])) == seen) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > CHAIN_SPIN_LIMIT || cc_ld(&L.CC[CC_ABORT])) {
      cc_st(&L.CC[CC_ABORT], 1u);
      return 0u;
    }
  }
  return cmd;
}

// ---- producer (wave 1): serves RUN commands until QUIT
template <bool BIG>
__device__ __noinline__ void carve_chain_produce(const CarveArgs& p, const StepCtx& c, uint64_t* l_alive, uint32_t* l_buf) {
  constexpr uint32_t NB = CHAIN_BLOCK, R = CHAIN_RING;
  static_assert(R >= 2u * NB && (R & (R - 1u)) == 0u, "ring geometry");
  const uint32_t lane = threadIdx.x & 63u;
  chain_lds_u32* const A = lds_pin<chain_lds_u32*>(l_alive);
  const ChainLds L = chain_lds(l_buf);
  const uint32_t a_base = (uint32_t)(uintptr_t)A;
  const auto rows32 = G((const uint32_t*)p.prop);  // row r: 2 * PM_PROP_ROW words; its compact slot list (flags word,
                                                   // slot of entry 0, 1, ...) starts at word 2 * PM_PROP_SLOTS
  const auto seed_slots = G((const uint32_t*)p.seed_slots);
  const uint32_t world = UNI(p.dist_world), rows_pr = UNI(c.rows_pr), n_seeds = UNI(c.n_seeds);
  uint32_t seen = 0u;  // last command word acted on
  for (;;) {
    uint32_t cmd = chain_wait_cmd(L, seen);
    if (cmd == 0u) return;
    seen = cmd;
    if ((cmd & 3u) == CH_QUIT) return;
    if ((cmd & 3u) != CH_RUN) {  // (a STOP without a RUN in between)
      cc_st(&L.CC[CC_ACK1], cmd);
      continue;
    }
    // ---- RUN: the seed list from CC_START on, 64 entries at a time (q: slots of entries cbase .. cbase + 63; qn: the
    // next 64, on their way)
    const uint32_t start = UNI(cc_ld(&L.CC[CC_START]));
    uint32_t cbase = start & ~63u, cpos = start, head = 0u;  // cpos: first entry not yet handed to a block
    auto load_chunk = [&](uint32_t b) -> uint32_t {
      const uint32_t i = b + lane;
      return seed_slots[i < n_seeds ? i : (n_seeds ? n_seeds - 1u : 0u)];
    };
    uint32_t q = load_chunk(cbase), qn = load_chunk(cbase + 64u);
    uint32_t rowA[NB], rowB[NB], idxA = 0u, idxB = 0u, posA = 0u, posB = 0u, nA = 0u, nB = 0u;
#pragma unroll
    for (uint32_t k = 0; k < NB; ++k) rowA[k] = rowB[k] = 0u;
    // request the rows of the next (up to) NB seeds of the list that are alive right now
    auto request = [&](uint32_t (&prow)[NB], uint32_t& pidx_v, uint32_t& ppos_v, uint32_t& pend_n) {
      pend_n = 0u;
      while (pend_n == 0u && cpos < n_seeds) {
        const uint32_t i = cbase + lane;
        const uint32_t w = A[q >> 5];
        uint64_t m = __ballot(i >= cpos && i < n_seeds && ((w >> (q & 31u)) & 1u) != 0u);
        const uint32_t n_m = (uint32_t)__popcll(m);
        pend_n = n_m < NB ? n_m : NB;
#pragma unroll
        for (uint32_t k = 0; k < NB; ++k) {
          if (k < pend_n) {  // the k-th live entry of the chunk -> lane k of the block registers
            const uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            const uint32_t si = cbase + l;
            const uint32_t sp = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)l);
            pidx_v = lane == k ? si : pidx_v;
            ppos_v = lane == k ? sp : ppos_v;
            prow[k] = rows32[(size_t)prop_row_of(si, world, rows_pr) * (2u * PM_PROP_ROW) + 2u * PM_PROP_SLOTS + lane];
            cpos = si + 1u;
          }
        }
        if (m == 0ull) cpos = cbase + 64u;  // nothing alive behind them in this chunk
        if (cpos >= cbase + 64u) {          // the chunk is used up: on to the next one
          cbase += 64u;
          q = qn;
          qn = load_chunk(cbase + 64u);
        }
      }
    };
    // park a block that has arrived: wait for room, write the digested rows, publish the new head.
    // Returns false when a new command came in while waiting.
    auto park = [&](uint32_t (&prow)[NB], uint32_t pidx_v, uint32_t ppos_v, uint32_t& pend_n) -> bool {
      if (pend_n == 0u) return true;
      uint32_t sp_n = 0u;
      while (head - cc_ld(&L.CC[CC_TAIL]) + pend_n > R) {
        __builtin_amdgcn_s_sleep(1);
        if (cc_ld(&L.CC[CC_CMD]) != seen) return false;
        if (++sp_n > CHAIN_SPIN_LIMIT) {
          cc_st(&L.CC[CC_ABORT], 1u);
          return false;
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < NB; ++k) {
        if (k < pend_n) {
          const uint32_t meta = UNI(prow[k]);
          const uint32_t sp = (uint32_t)__builtin_amdgcn_readlane((int)ppos_v, (int)k);
          const uint32_t si = (uint32_t)__builtin_amdgcn_readlane((int)pidx_v, (int)k);
          const uint32_t e = lane == 0u ? sp : prow[k];
          const uint32_t r = (head + k) & (R - 1u);
          const uint32_t o = r * 64u + lane;
          const uint32_t bit = lane <= (meta & 0xFFu) ? 1u << (e & 31u) : 0u;
          L.RAB[o] = ((unsigned long long)bit << 32) | (a_base + ((e >> 5) << 2));
          L.RE[o] = e;
          // what a step needs of the flags word: bit 0 clean and safe, bit 1 the tail is settled, bits 8..15 the
          // first entry within the band of the last one
          const uint32_t m2 = (((meta & PM_ROW_CLEAN) && (meta & PM_ROW_SAFE)) ? 1u : 0u) |
                              ((meta & (PM_ROW_COMPLETE | PM_ROW_TAIL_CLEAR | PM_ROW_TAIL_OK)) ? 2u : 0u) | (meta & 0xFF00u);
          if (lane == 0u) {
            L.RI[r] = si;
            L.RM[r] = m2;
          }
        }
      }
      head += pend_n;
      pend_n = 0u;
      cc_st(&L.CC[CC_HEAD], head);  // (LDS operations of a wave execute in order: the rows are there before the head)
      return true;
    };
    bool running = true;
    request(rowA, idxA, posA, nA);
    while (running) {
      request(rowB, idxB, posB, nB);
      if (!park(rowA, idxA, posA, nA)) break;
      if (nB == 0u) running = false;
      if (running) {
        request(rowA, idxA, posA, nA);
        if (!park(rowB, idxB, posB, nB)) break;
        if (nA == 0u) running = false;
      }
      if (cc_ld(&L.CC[CC_CMD]) != seen) break;
    }
    if (!running) cc_st(&L.CC[CC_DONE], 1u);  // the list is used up: CC_HEAD is final
    cmd = chain_wait_cmd(L, seen);
    if (cmd == 0u) return;
    seen = cmd;
    if ((cmd & 3u) == CH_QUIT) return;
    cc_st(&L.CC[CC_ACK1], cmd);  // STOP
  }
}

// ---- collector (wave 2): serves RUN commands until QUIT
template <bool BIG, bool STREAM = false>
__device__ __noinline__ void carve_chain_collect(const CarveArgs& p, const StepCtx& c, uint32_t* l_buf) {
  constexpr uint32_t R = CHAIN_RING;
  const uint32_t lane = threadIdx.x & 63u;
  const ChainLds L = chain_lds(l_buf);
  const auto members = G(p.members);
  const auto g_cfg = G(p.g_cfg);
  const auto g_n = G(p.g_n);
  const auto g_off = G(p.g_off);
  const uint32_t group_n = UNI(c.max_s), want = group_n - 1u, cfg = UNI(c.cfg);
  // (read HERE, once: inside the loop the compiler re-reads the argument block behind every store it cannot prove
  // unaliased — a round trip to L2 per collected entry, four times what the chain takes to commit one: the collector
  // fell behind, the ring looked full to the parkers, and the chain ran dry in front of a full ring)
  const auto free32 = G((uint32_t*)p.bits_scratch);
  uint32_t seen = 0u;
  for (;;) {
    uint32_t cmd = chain_wait_cmd(L, seen);
    if (cmd == 0u) return;
    seen = cmd;
    if ((cmd & 3u) == CH_QUIT) return;
    if ((cmd & 3u) != CH_RUN) {
      cc_st(&L.CC[CC_ACK2], cmd);
      continue;
    }
    uint32_t n_groups = UNI(cc_ld(&L.CC[CC_G0])), mem_off = UNI(cc_ld(&L.CC[CC_M0]));  // as of the last write-out
    uint32_t staged = 0u, t2 = 0u;
    // groups collected in LDS -> group records + members (all of them full groups, back to back)
    auto write_out = [&]() {
      for (uint32_t g = lane; g < staged; g += 64u) {
        g_cfg[n_groups + g] = cfg;
        g_n[n_groups + g] = group_n;
        g_off[n_groups + g] = mem_off + g * group_n;
      }
      const uint32_t nm = staged * group_n;
      for (uint32_t k = lane; k < nm; k += 64u) members[mem_off + k] = L.STAGE[k];
      n_groups += staged;
      mem_off += nm;
      staged = 0u;
    };
    uint32_t idle = 0u;
    bool stopping = false;
    for (;;) {
      const uint32_t crit = UNI(cc_ld(&L.CC[CC_CRIT]));
      if (t2 != crit) {
        idle = 0u;
        while (t2 != crit) {
          t2 = UNI(t2);
          staged = UNI(staged);
          const uint32_t r = t2 & (R - 1u);
          const unsigned long long av = L.Q[r];
          const uint64_t a = ((uint64_t)UNI((uint32_t)(av >> 32)) << 32) | UNI((uint32_t)av);
          if (a) {  // a committed step: the seed and its `want` nearest live neighbours, in key order
            const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a, 0u));
            if (((a >> lane) & 1ull) && rk <= want) {
              const uint32_t sl = L.RE[r * 64u + lane];
              L.STAGE[staged * group_n + rk] = sl;  // slots (translated to worker ids after the run)
              // (streaming carve: the proposers' copy of the candidate bitmap follows the chain, a few steps behind)
              if (STREAM)
                __hip_atomic_fetch_and(&free32[sl >> 5], ~(1u << (sl & 31u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            staged += 1u;
            if ((staged + 1u) * group_n > CHAIN_STAGE_WORDS) write_out();
          }
          ++t2;
        }
        cc_st(&L.CC[CC_TAIL], t2);  // room for the producer
        continue;
      }
      if (stopping) break;  // (CC_CRIT was final when the STOP was seen, and everything up to it is collected)
      if (cc_ld(&L.CC[CC_CMD]) != seen) {
        stopping = true;  // one more look at CC_CRIT: the chain publishes its last entries before the STOP
        continue;
      }
      __builtin_amdgcn_s_sleep(1);
      if (++idle > CHAIN_SPIN_LIMIT || cc_ld(&L.CC[CC_ABORT])) {
        cc_st(&L.CC[CC_ABORT], 1u);
        return;
      }
    }
    write_out();
    cmd = chain_wait_cmd(L, seen);
    if (cmd == 0u) return;
    seen = cmd;
    if ((cmd & 3u) == CH_QUIT) return;
    cc_st(&L.CC[CC_ACK2], cmd);  // STOP: the members are written
  }
}

// ---- the chain itself (wave 0)
template <bool BIG>
__device__ __noinline__ int carve_chain(const CarveArgs& p, StepCtx& c_ref, uint32_t* l_buf, uint32_t steps_before,
                                        uint32_t& seed_cur, uint32_t& cmd_seq) {
  StepCtx c = c_ref;
  constexpr uint32_t R = CHAIN_RING;
  const uint32_t lane = threadIdx.x & 63u;
  typedef chain_lds_u32 lds_u32;
  const ChainLds L = chain_lds(l_buf);
  const uint32_t dbg_every = UNI(p.debug_uncertain_every);
  const uint32_t n_seeds = UNI(c.n_seeds);
  const uint32_t n_list_v = UNI(c.n_list), n_start = UNI(c.n_start);
  const uint32_t group_n = UNI(c.max_s), want = group_n - 1u;
  const uint32_t cap_g = UNI(p.cap_groups), cap_m = UNI(p.cap_members);
  const uint32_t step0 = UNI(steps_before) + UNI(c.steps);
  const uint32_t base_cand = UNI(c.n_cand), base_groups = UNI(c.n_groups), base_mem = UNI(c.mem_off);
  uint32_t n_cand = base_cand, commits = 0;
  int action = FAST_SEQ;
  uint32_t exit_cur = n_seeds;  // where carve_fast_steps resumes its search for the first live seed
  // Commits that can follow one another before any of the conditions that end the chain can come true (they are
  // looked at again when the budget is used up): candidates for full groups, room in the output arrays, the
  // re-proposal threshold, the debug hook.  >= 1 whenever none of those conditions holds.
  auto budget_now = [&]() -> uint32_t {
    uint32_t b = n_cand / group_n;
    const uint32_t room_g = cap_g - (base_groups + commits), room_m = (cap_m - base_mem) / group_n - commits;
    b = b < room_g ? b : room_g;
    b = b < room_m ? b : room_m;
    if (n_list_v > 256u) {
      const uint32_t t = n_cand * PM_THIN_DIV >= n_start ? (n_cand * PM_THIN_DIV - n_start) / (PM_THIN_DIV * group_n) + 1u : 1u;
      b = b < t ? b : t;
    }
    if (dbg_every) {
      const uint32_t r = (step0 + commits + 1u) % dbg_every;
      b = b < dbg_every - r ? b : dbg_every - r;
    }
    return b;
  };
#ifdef PM_CARVE_PROF
  uint64_t ct = __builtin_amdgcn_s_memtime(), ct_wait = 0, ct_steps = 0, ct_stop = 0;
  uint32_t cn_outer = 0, cn_dead = 0, cn_wait = 0;
#define CH_MARK(var) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); var += t_ - ct; ct = t_; } while (0)
#define CH_COUNT(var) (++var)
#else
#define CH_MARK(var)
#define CH_COUNT(var)
#endif

  // ---- start the producer and the collector
  if (lane == 0u) {
    cc_st(&L.CC[CC_HEAD], 0u);
    cc_st(&L.CC[CC_CRIT], 0u);
    cc_st(&L.CC[CC_TAIL], 0u);
    cc_st(&L.CC[CC_DONE], 0u);
    cc_st(&L.CC[CC_START], UNI(seed_cur));
    cc_st(&L.CC[CC_G0], base_groups);
    cc_st(&L.CC[CC_M0], base_mem);
  }
  cmd_seq += 4u;
  if (lane == 0u) cc_st(&L.CC[CC_CMD], cmd_seq | CH_RUN);
  uint32_t tail = 0u, budget = 0u;
  bool aborted = false;
#ifdef PM_CHAIN_PRIO
  __builtin_amdgcn_s_setprio(3);  // (the chain is the critical path; the waves beside it only feed it)
#endif
  if (n_seeds > 0u) {
    for (;;) {
      CH_COUNT(cn_outer);
      tail = UNI(tail);
      // ---- rows parked and not yet looked at
      uint32_t head = UNI(cc_ld(&L.CC[CC_HEAD]));
      if (head == tail) {
        uint32_t sp_n = 0u;
        bool used_up = false;
        for (;;) {
          const uint32_t done = UNI(cc_ld(&L.CC[CC_DONE]));
          head = UNI(cc_ld(&L.CC[CC_HEAD]));  // (read behind the flag: a set flag means this head is final)
          if (head != tail) break;
          if (done) {
            used_up = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          if (++sp_n > CHAIN_SPIN_LIMIT || UNI(cc_ld(&L.CC[CC_ABORT]))) {
            aborted = true;
            break;
          }
        }
        CH_COUNT(cn_wait);
        if (aborted) break;
        if (used_up) {  // the list is used up
          action = FAST_SEQ;
          exit_cur = n_seeds;
          break;
        }
      }
      CH_MARK(ct_wait);
      uint32_t n_steps = head - tail;
      n_steps = n_steps < 16u ? n_steps : 16u;
      bool stop = false;
      unsigned long long rab_n = L.RAB[(tail & (R - 1u)) * 64u + lane];
      uint32_t rm_n = L.RM[tail & (R - 1u)];
      const uint32_t flags_needed = want != 0u ? 3u : 0u;
      uint32_t s = 0u;
      while (s < n_steps) {
        // ---- the steps that need no second look, as straight-line code with ONE way out: a dead seed (absorbed by a
        // group since its row was requested) is a step that selects nothing
        bool good = true;
        uint64_t a = 0ull;
        uint32_t r = 0u, ra = 0u, rb = 0u, m2 = 0u;
        do {
          r = tail & (R - 1u);
          ra = (uint32_t)rab_n;
          rb = (uint32_t)(rab_n >> 32);
          m2 = UNI(rm_n);
          const uint32_t w = *(lds_u32*)(uintptr_t)ra;
          rab_n = L.RAB[((tail + 1u) & (R - 1u)) * 64u + lane];  // (the next step's row, one step early)
          rm_n = L.RM[(tail + 1u) & (R - 1u)];
          a = __ballot((w & rb) != 0u);
          const bool live = (a & 1ull) != 0ull;
          // my rank among the live lanes: 0 for the seed, e + 1 for the e-th live entry
          const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a, 0u));
          const uint64_t selm = live ? a & __ballot(rk <= want) : 0ull;  // the seed and its `want` nearest live neighbours
          // enough live entries, a row whose flags settle the certificate wholesale, and no end of the chain in sight
          good = !live || ((uint32_t)__popcll(selm) == group_n && (m2 & flags_needed) == flags_needed && budget != 0u);
          if (__builtin_expect(!good, 0)) break;
          // commit (create_group_atomically mod.rs:299-322; healthy_nodes.retain :585)
          if ((selm >> lane) & 1ull)
            __hip_atomic_fetch_and((lds_u32*)(uintptr_t)ra, ~rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          L.Q[r] = live ? a : 0ull;  // (every lane the same word: the collector re-derives the selection from it)
          ++tail;
          ++s;
          n_cand -= live ? group_n : 0u;
          commits += live ? 1u : 0u;
          budget -= live ? 1u : 0u;
#ifdef PM_CARVE_PROF
          cn_dead += live ? 0u : 1u;
#endif
        } while (s < n_steps);
        if (good) break;
        // ---- a live seed that needs a second look (its row is loaded: ra, rb, m2, a)
        if (budget == 0u) {
          exit_cur = UNI(L.RI[r]);
          // `while total_available >= min` with `compatible < min => break` (mod.rs:507,517-519) hold while full
          // groups fit (max_s >= min_s); the last, partial group is carve_fast_steps' business
          if (n_cand < group_n) {
            action = FAST_SEQ;
            stop = true;
            break;
          }
          if (base_groups + commits >= cap_g || base_mem + (commits + 1u) * group_n > cap_m) {
            action = FAST_OVERFLOW;
            stop = true;
            break;
          }
          // A good part of what was alive when the batch started is gone: the neighbour rows are thinning out.
          // Re-prepare (compact) and re-propose now, before rows start running out of live entries.
          if (n_cand * PM_THIN_DIV < n_start && n_list_v > 256u && commits > 0u) {
            if (lane == 0u) p.status->why[0] += 1u;
            action = FAST_REPROPOSE;
            stop = true;
            break;
          }
          if (dbg_every && ((step0 + commits + 1u) % dbg_every) == 0u) {
            action = FAST_SLOW;
            stop = true;
            break;
          }
          budget = budget_now();
        }
        {
          const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)a, 0u));
          const uint64_t selm = a & __ballot(rk <= want);
          // enough live entries, and a row whose flags settle the certificate: no two entries near each other at
          // different sites, nothing near the antipode, and a tail that is complete / clear / at one site — or a
          // selection that ends in front of the tail's band
          bool ok = (uint32_t)__popcll(selm) == group_n;
          if (want != 0u && (m2 & 3u) != 3u)
            ok = ok && (m2 & 1u) && 62u - (uint32_t)__builtin_clzll(selm | 2ull) < ((m2 >> 8) & 0xFFu);
          if (!ok) {
            exit_cur = UNI(L.RI[r]);
            action = FAST_SLOW;
            stop = true;
            break;
          }
          if ((selm >> lane) & 1ull)
            __hip_atomic_fetch_and((lds_u32*)(uintptr_t)ra, ~rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          L.Q[r] = a;
          ++tail;
          ++s;
          n_cand -= group_n;
          commits += 1u;
          budget -= 1u;
        }
      }
      if (lane == 0u) cc_st(&L.CC[CC_CRIT], tail);  // for the collector
      CH_MARK(ct_steps);
      if (stop) break;
    }
  }
#ifdef PM_CHAIN_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  // ---- stop the other two (they must be off the key array before it is used again; the collector writes out first)
  cmd_seq += 4u;
  if (lane == 0u) cc_st(&L.CC[CC_CMD], cmd_seq | CH_STOP);
  {
    uint32_t sp_n = 0u;
    while (UNI(cc_ld(&L.CC[CC_ACK1])) != (cmd_seq | CH_STOP) || UNI(cc_ld(&L.CC[CC_ACK2])) != (cmd_seq | CH_STOP)) {
      __builtin_amdgcn_s_sleep(1);
      if (++sp_n > CHAIN_SPIN_LIMIT || UNI(cc_ld(&L.CC[CC_ABORT]))) {
        aborted = true;
        break;
      }
    }
  }
  CH_MARK(ct_stop);
  if (aborted) {
    if (lane == 0u) cc_st(&L.CC[CC_ABORT], 1u);
    action = FAST_ABORT;  // a hand-shake inside the workgroup timed out: reported as CARVE_STATE_ABORTED (never seen)
  }
  c.n_groups = base_groups + commits;
  c.mem_off = base_mem + commits * group_n;
  c.n_cand = n_cand;
  c.total_available -= commits * group_n;
  c.steps += commits;
  c.fast_steps += commits;
  // sum over the commits of the live candidates before each: base, base - g, base - 2g, ...
  c.cand_sum += (unsigned long long)commits * base_cand -
                (unsigned long long)group_n * ((unsigned long long)commits * (commits ? commits - 1u : 0u) / 2ull);
#ifdef PM_CARVE_PROF
  if (lane == 0u) {
    unsigned long long* pr = (unsigned long long*)p.status->prof;
    pr[1] += 1u;        // calls
    pr[2] += commits;
    pr[4] += action == FAST_SLOW ? 1u : 0u;
    pr[16] += ct_wait;
    pr[17] += ct_stop;
    pr[18] += ct_steps;
    pr[19] += cn_outer;
    pr[23] += cn_dead;
    pr[24] += cn_wait;
  }
#endif
  c_ref = c;
  seed_cur = exit_cur;
  return action;
}

// One exact step of a configuration (the reference's filter + sort + take, mod.rs:511-561, evaluated as it stands):
// keys for every live candidate, two-level selection (DPP argmin rounds per wave, 8-way merge), certificate, commit.
// The whole workgroup; three LDS-only barriers.  STEP_CONTINUE = one group committed.
// SPARSE (streaming carve, where slot == position and a list is as long as the eligible list whatever is left of it):
// the sweeps run over the live candidates only — `cl` lists their slots, ascending, n_cl of them, keys are indexed
// like `cl` — so a step costs what is left, not what there was.
template <bool BIG, bool SPARSE = false>
__device__ __noinline__ int carve_exact_step(const CarveArgs& p, BlockRed& red, StepCtx& c, const uint32_t* l_wid,
                                             uint64_t* l_key, uint64_t* l_alive, const uint64_t* l_loc, uint64_t* part,
                                             uint32_t* sel_out, uint32_t steps_before, unsigned long long* mirror,
                                             const uint32_t* cl = nullptr, uint32_t n_cl = 0u) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_it = SPARSE ? n_cl : c.n_list;  // entries the sweeps run over
  auto slot_of = [cl](uint32_t i) -> uint32_t { return SPARSE ? cl[i] : i; };
  constexpr uint32_t SB = BIG ? PM_CARVE_SLOT_BITS_BIG : PM_CARVE_SLOT_BITS;
  constexpr double band_rel = BIG ? PM_TIE_BAND_BIG : PM_TIE_BAND;
  const uint32_t lw = (c.n_list + 63u) >> 6;
  auto wid_of = [](uint32_t sl) -> uint32_t { return sl; };  // members are recorded as SLOTS (translated after the run)
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

  const uint32_t want = c.max_s - 1u < c.n_cand - 1u ? c.max_s - 1u : c.n_cand - 1u;  // fill to max (mod.rs:545-551)
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
    const double slat = G(p.cc_lat)[seed], slon = G(p.cc_lon)[seed], scos = G(p.cc_cos)[seed];
    uint64_t lmin = ~0ull;
    for (uint32_t i = tid; i < n_it; i += CARVE_THREADS) {
      const uint32_t s = slot_of(i);
      uint64_t k = ~0ull;
      if (s != seed && (SPARSE || bit_at(l_alive, s))) {
        if (!use_dist) {
          k = s;
        } else if (bit_at(l_loc, s)) {
          k = pack_key((uint64_t)__double_as_longlong(
                           hav_a(slat, slon, scos, G(p.cc_lat)[s], G(p.cc_lon)[s], G(p.cc_cos)[s])), s, SB);
        } else if (!located_only) {
          k = pack_key(PM_KEY_NOLOC, s, SB);
        }
      }
      l_key[i] = k;
      lmin = k < lmin ? k : lmin;
    }
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
            for (uint32_t i = tid; i < n_it; i += CARVE_THREADS) {
              const uint64_t k = l_key[i];
              m = (k > v && k < m) ? k : m;
            }
            lmin = m;
          }
        }
        if (lane == 0) red.part_n[wave] = cnt;
              lds_barrier();
              // ---- level 2: merge of the waves' sorted partial lists, redundantly in every wave
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
              G(p.members)[c.mem_off + 1u + n_sel] = wid_of(bs);
          }
          last = b;
          ++n_sel;
          if (lmin == b) {
            uint64_t m = ~0ull;
            for (uint32_t i = tid; i < n_it; i += CARVE_THREADS) {
              const uint64_t k = l_key[i];
              m = (k > b && k < m) ? k : m;
            }
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
    const double band = a_m * band_rel + 1e-300;
    const double mlat = G(p.cc_lat)[ls], mlon = G(p.cc_lon)[ls];  // uniform loads
    if (a_m > PM_A_MAX_SAFE) uncertain = 1;
    for (uint32_t i = tid; i < n_it; i += CARVE_THREADS) {
      const uint32_t s = slot_of(i);
      const uint64_t k = l_key[i];
      const uint64_t kb = (k >> SB) << SB;
      const double a = __longlong_as_double((long long)kb);
      const bool near = k != ~0ull && kb != noloc_key && fabs(a - a_m) <= band;
      if (near && (G(p.cc_lat)[s] != mlat || G(p.cc_lon)[s] != mlon)) uncertain = 1;
    }
  }
  const uint64_t ub = __ballot(uncertain != 0);
  if (lane == 0) red.flag[wave] = ub != 0ull;
  lds_barrier();
  uint32_t any = 0;
#pragma unroll
  for (uint32_t k = 0; k < CARVE_WAVES; ++k) any |= red.flag[k];
  if (any) {
    if (tid == 0) G(p.status)->stop_seed = l_wid[seed];
    return STEP_UNCERTAIN;
  }
  if (c.n_groups >= p.cap_groups || c.mem_off + total > p.cap_members) return STEP_OVERFLOW;

  // ---- commit (create_group_atomically mod.rs:299-322; healthy_nodes.retain :585): selected slots =
  // seed + every key <= last.  Each wave owns whole bitmap words (slot>>6 == j*16 + wave): ballot writes them.
  if (SPARSE) {  // (a handful of bits among the live candidates: one atomic each, here and in the published copy)
    for (uint32_t i = tid; i < n_it; i += CARVE_THREADS) {
      const uint32_t s = slot_of(i);
      if (s == seed || (n_sel > 0 && l_key[i] <= last)) {
        atomicAnd((unsigned long long*)&l_alive[s >> 6], ~(1ull << (s & 63u)));
        if (mirror) __hip_atomic_fetch_and(&mirror[s >> 6], ~(1ull << (s & 63u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else {
    for (uint32_t wj = wave; wj < lw; wj += CARVE_WAVES) {  // wave-uniform
      const uint32_t s = wj * 64u + lane;
      const bool was = bit_at(l_alive, s);
      const bool sel = was && (s == seed || (n_sel > 0 && l_key[s] <= last));
      const uint64_t nw = __ballot(was && !sel);
      const uint64_t ow = __ballot(was);
      if (lane == 0) {
        l_alive[wj] = nw;
        // (streaming carve: the published bitmap of free positions follows every removal — it holds every
        // configuration's candidates, so only the bits this step took are cleared)
        if (mirror && ow != nw) __hip_atomic_fetch_and(&mirror[wj], ~(ow ^ nw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (wave == 0) {  // group record + members: LDS -> fire-and-forget global stores
    if (lane == 0) {
      G(p.members)[c.mem_off] = wid_of(seed);
      G(p.g_cfg)[c.n_groups] = c.cfg;
      G(p.g_n)[c.n_groups] = total;
      G(p.g_off)[c.n_groups] = c.mem_off;
    }
    const uint32_t lim = n_sel < PM_CARVE_SEL_CAP ? n_sel : PM_CARVE_SEL_CAP;
    for (uint32_t r = lane; r < lim; r += 64u) G(p.members)[c.mem_off + 1u + r] = wid_of(sel_out[r]);
  }
  lds_barrier();
  PROF_MARK(22);  // one exact step
  c.n_groups += 1;
  c.mem_off += total;
  c.cand_sum += c.n_cand;
  c.n_cand -= total;
  c.total_available -= total;  // mod.rs:586
  c.steps += 1;
  return STEP_CONTINUE;
}

// LDS carve of one candidate list of at most PM_CARVE_SLOTS slots: worker ids, site ids, packed keys,
// the alive / loc bitmaps and the per-wave partial selections all live in LDS (slot s is owned by thread
// s % CARVE_THREADS).  Fast steps come from the proposals; a slow step is the exact full sweep: keys for every
// live candidate, two-level selection (DPP argmin rounds per wave, 8-way merge), certificate, commit —
// three LDS-only barriers.  Runs until the configuration is exhausted, a recompaction is due, or a step
// cannot be certified.
template <bool BIG>
__device__ __noinline__ int carve_run_lds(const CarveArgs& p, BlockRed& red, StepCtx& c, const uint32_t* l_wid,
                                          const uint32_t* l_site, uint64_t* l_key, uint64_t* l_alive,
                                          const uint64_t* l_loc, uint64_t* part, uint32_t* sel_out,
                                          uint32_t* l_stage, uint32_t steps_before) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool have_props = c.mode == CARVE_MODE_FORM && c.use_props;
  uint32_t seed_cur = 0;  // wave 0: how far into the batch's seed list the carve has come
  const ChainLds CL = chain_lds(l_stage);
  for (;;) {
    if (have_props) {
      // Everything the proposals can serve is done by wave 0 alone (everyone else waits at the barrier): the chain
      // of located steps (carve_chain), and — with the row's keys at hand — whatever it hands over: rows whose
      // flags do not settle the certificate, exhausted rows, the last partial group, the first-come tail.
      const bool chain = c.prop_k && c.proximity && p.rounds_enabled && c.max_s - 1u < PM_PROP_KMAX;
      if (chain) {  // the control block of the chain (an exact step in between has used the key array it lives in)
        if (tid < 16u) cc_st(&CL.CC[tid], 0u);
        lds_barrier();
      }
      if (wave == 0) {
        PROF_DECL;
        int act = FAST_SEQ;
        uint32_t cmd_seq = 0u;  // commands to the producer are numbered
        for (;;) {
          if (chain) {
            act = carve_chain<BIG>(p, c, l_stage, steps_before, seed_cur, cmd_seq);
            PROF_MARK(0);
            if (act == FAST_OVERFLOW || act == FAST_REPROPOSE || act == FAST_ABORT) break;
          }
          act = carve_fast_steps<BIG>(p, c, l_site, l_alive, l_loc, steps_before, seed_cur,
                                      (chain && act == FAST_SLOW) ? 1u : 0xFFFFFFFFu);
          if (act != FAST_AGAIN) break;
        }
        if (chain && lane == 0) cc_st(&CL.CC[CC_CMD], (cmd_seq + 4u) | CH_QUIT);  // waves 1 and 2 come to the barrier
        if (lane == 0) {
          red.f_action = (uint32_t)act;
          red.f_n_cand = c.n_cand;
          red.f_total_available = c.total_available;
          red.f_n_groups = c.n_groups;
          red.f_mem_off = c.mem_off;
          red.f_steps = c.steps;
          red.f_fast = c.fast_steps;
          red.f_cand_sum = c.cand_sum;
        }
        PROF_MARK(11);
      } else if (wave == 1 && chain) {
        carve_chain_produce<BIG>(p, c, l_alive, l_stage);
      } else if (wave == 2 && chain) {
        carve_chain_collect<BIG>(p, c, l_stage);
      }
      lds_barrier();
      const uint32_t act = red.f_action;
      c.n_cand = red.f_n_cand;
      c.total_available = red.f_total_available;
      c.n_groups = red.f_n_groups;
      c.mem_off = red.f_mem_off;
      c.steps = red.f_steps;
      c.fast_steps = red.f_fast;
      c.cand_sum = red.f_cand_sum;
      lds_barrier();  // the mailbox is rewritten after the next slow step
      if (act == FAST_DONE) return STEP_BREAK;
      if (act == FAST_OVERFLOW) return STEP_OVERFLOW;
      if (act == FAST_ABORT) return STEP_ABORT;
      if (act == FAST_REPROPOSE) return STEP_CONTINUE;  // re-prepare: next proposal batch
    }
    {
      const int rc = carve_exact_step<BIG>(p, red, c, l_wid, l_key, l_alive, l_loc, part, sel_out, steps_before, nullptr);
      if (rc != STEP_CONTINUE) return rc;
    }
    // drop dead slots once more than half of the list is gone.  With proposals this also ends the launch:
    // the list is re-prepared and the next propose / validate pair continues with fresh neighbour lists.
    if (c.n_cand * (have_props ? PM_THIN_DIV : 2u) < (have_props ? c.n_start : c.n_list) && c.n_list > (have_props ? 256u : CARVE_THREADS)) {
      if (tid == 0) p.status->why[2] += 1u;
      return STEP_CONTINUE;
    }
  }
}

// ---- generic path for candidate lists that do not fit the LDS/register scheme (> PM_CARVE_SLOTS):
// packed keys, positions and bitmaps live in HBM/L2; one workgroup-wide argmin round per member.
__device__ __noinline__ int carve_step_mem(const CarveArgs& p, BlockRed& red, StepCtx& c, uint64_t* part, uint64_t* key,
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
    const double slat = G(p.cc_lat)[seed], slon = G(p.cc_lon)[seed], scos = G(p.cc_cos)[seed];
    uint64_t lmin = ~0ull;
    for (uint32_t s = tid; s < c.n_list; s += CARVE_THREADS) {
      uint64_t k = ~0ull;
      if (s != seed && bit_at(alive, s)) {
        if (!use_dist) {
          k = s;
        } else if (bit_at(loc, s)) {
          k = pack_key((uint64_t)__double_as_longlong(hav_a(slat, slon, scos, G(p.cc_lat)[s], G(p.cc_lon)[s], G(p.cc_cos)[s])), s, SB);
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
        G(p.members)[c.mem_off + 1u + n_sel] = wid[(uint32_t)(b & ((1ull << SB) - 1ull))];
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
    const double mlat = G(p.cc_lat)[ls], mlon = G(p.cc_lon)[ls];
    if (a_m > PM_A_MAX_SAFE) uncertain = 1;
    for (uint32_t s = tid; s < c.n_list; s += CARVE_THREADS) {
      const uint64_t k = key[s];
      if (k == ~0ull || s == ls) continue;
      const uint64_t kb = (k >> SB) << SB;
      if (kb == ((PM_KEY_NOLOC >> SB) << SB)) continue;
      const double a = __longlong_as_double((long long)kb);
      if (fabs(a - a_m) <= band && (G(p.cc_lat)[s] != mlat || G(p.cc_lon)[s] != mlon)) uncertain = 1;
    }
  }
  if (__syncthreads_or(uncertain)) {
    if (tid == 0) G(p.status)->stop_seed = wid[seed];
    return STEP_UNCERTAIN;
  }
  if (c.n_groups >= p.cap_groups || c.mem_off + total > p.cap_members) return STEP_OVERFLOW;
  for (uint32_t s = tid; s < c.n_list; s += CARVE_THREADS) {
    if (!bit_at(alive, s)) continue;
    if (s == seed || (n_sel > 0 && key[s] <= last))
      atomicAnd((unsigned long long*)&alive[s >> 6], ~(1ull << (s & 63u)));
  }
  if (tid == 0) {
    G(p.members)[c.mem_off] = wid[seed];
    G(p.g_cfg)[c.n_groups] = c.cfg;
    G(p.g_n)[c.n_groups] = total;
    G(p.g_off)[c.n_groups] = c.mem_off;
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
__device__ __noinline__ uint32_t carve_compact_count(const CarveArgs& p, BlockRed& red, uint32_t n, uint64_t cbit,
                                                      const uint64_t* alive_bits) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n_words = (n + 63u) >> 6;
  const uint32_t wpw = (n_words + CARVE_WAVES - 1u) / CARVE_WAVES;
  const uint32_t j0 = wave * wpw, j1 = min(n_words, j0 + wpw);
  uint32_t cnt = 0;
  const bool merge = p.mode == CARVE_MODE_MERGE;
  const auto alive_g = G(alive_bits);
  const auto c_compat = G((const uint64_t*)p.c_compat);
  for (uint32_t jb = j0; jb < j1; jb += 8u) {  // eight words per batch; every load unconditional (clamped
    uint64_t aw[8], cm[8];                     // index) and independent, so the whole batch is in flight at once
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t j = jb + (uint32_t)u;
      const uint32_t i = j * 64u + lane;
      aw[u] = alive_g[j < n_words ? j : n_words - 1u];
      cm[u] = merge ? ~0ull : c_compat[i < n ? i : n - 1u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t j = jb + (uint32_t)u;
      const uint32_t i = j * 64u + lane;
      const bool c = j < j1 && i < n && ((aw[u] >> lane) & 1ull) && (cm[u] & cbit) != 0ull;
      cnt += __popcll(__ballot(c));
    }
  }
  __syncthreads();  // previous users of red.a are done
  if (lane == 0) red.a[wave] = cnt;
  __syncthreads();
  uint32_t total = 0;
  for (uint32_t k = 0; k < CARVE_WAVES; ++k) total += red.a[k];
  return total;
}

__device__ __noinline__ void carve_compact_place(const CarveArgs& p, BlockRed& red, uint32_t n, uint64_t cbit, uint32_t n_list,
                                                  const uint64_t* alive_bits) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_words = (n + 63u) >> 6;
  const uint32_t wpw = (n_words + CARVE_WAVES - 1u) / CARVE_WAVES;
  const uint32_t j0 = wave * wpw, j1 = min(n_words, j0 + wpw);
  const auto alive = G(p.bits_scratch);
  const auto loc = G(p.bits_scratch) + p.bits_stride;
  uint32_t off = 0;
  for (uint32_t k = 0; k < wave; ++k) off += red.a[k];
  const bool merge = p.mode == CARVE_MODE_MERGE;
  const auto alive_g = G(alive_bits);
  const auto c_compat = G((const uint64_t*)p.c_compat);
  for (uint32_t jb = j0; jb < j1; jb += 4u) {
    uint64_t aw[4], cm[4], lg[4];
    uint32_t ow[4], os[4];
    double la[4], lo[4], co[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // every load of the batch first, unconditional (clamped index)
      const uint32_t j = jb + (uint32_t)u;
      const uint32_t i = j * 64u + lane;
      const uint32_t ic = i < n ? i : n - 1u;
      aw[u] = alive_g[j < n_words ? j : n_words - 1u];
      lg[u] = G(p.loc_g)[j < n_words ? j : n_words - 1u];
      cm[u] = merge ? ~0ull : c_compat[ic];
      ow[u] = G(p.order)[ic];
      os[u] = G(p.c_site)[ic];
      la[u] = G(p.c_lat)[ic];
      lo[u] = G(p.c_lon)[ic];
      co[u] = G(p.c_cos)[ic];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = jb + (uint32_t)u;
      const uint32_t i = j * 64u + lane;
      const bool c = j < j1 && i < n && ((aw[u] >> lane) & 1ull) && (cm[u] & cbit) != 0ull;
      const uint64_t bal = __ballot(c);
      if (c) {
        const uint32_t s = off + __popcll(bal & ((1ull << lane) - 1ull));
        G(p.slot_pos)[s] = i | (((uint32_t)(lg[u] >> lane) & 1u) << 31);  // bit 31: has a location
        G(p.slot_wid)[s] = ow[u];
        G(p.cc_lat)[s] = la[u];
        G(p.cc_lon)[s] = lo[u];
        G(p.cc_cos)[s] = co[u];
        G(p.cc_site)[s] = os[u];
      }
      off += __popcll(bal);
    }
  }
  __syncthreads();
  const uint32_t lw = (n_list + 63u) >> 6;
  // slot bitmaps: alive = all ones, loc from bit 31 of slot_pos (four independent coalesced loads per step)
  for (uint32_t base = 0; base < lw * 64u; base += 4u * CARVE_THREADS) {
    uint32_t sp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t s = base + (uint32_t)u * CARVE_THREADS + tid;
      sp[u] = G(p.slot_pos)[s < n_list ? s : n_list - 1u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t s = base + (uint32_t)u * CARVE_THREADS + tid;
      const bool in = s < n_list;
      const uint64_t ba = __ballot(in), bl = __ballot(in && (sp[u] >> 31));
      if (lane == 0 && (s >> 6) < lw) {
        alive[s >> 6] = ba;
        loc[s >> 6] = bl;
      }
    }
  }
  __syncthreads();
}

// Proposal generator: one wave per located live slot of the prepared configuration.  The wave sweeps the whole
// candidate list and keeps its 64 smallest keys SORTED ACROSS ITS LANES (lane i = the i-th nearest so far).  A
// candidate enters only if it beats lane 63 (one compare against a wave-uniform threshold; after the first few
// hundred slots almost nothing does: 64 (1 + ln(n / 64)) insertions over a list of n), and an insertion is a
// wave-wide shift by one lane (DPP wave_shr) from the insertion point — so the sweep's inner loop is the key
// arithmetic alone, and the finished register IS the row: no per-lane queues, no pop rounds, no re-sweeps.
struct NearRow {
  uint64_t key;     // ascending over the lanes; ~0 = empty
  uint64_t tau;     // lane 63's key (wave-uniform): what a candidate has to beat
  uint64_t tau_hi;  // keys in (tau, tau_hi) may end up within the certificate band of the row's last entry
  // the unlisted candidates that came that close (rejected at the threshold, or pushed out of lane 63 later):
  // the smallest key and its site, and the smallest key at any OTHER site — enough to answer, at the end, "is
  // there an unlisted candidate within the band of the last entry that does not sit at that entry's site?"
  uint64_t m1, m2;
  uint32_t s1;
};
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v) {  // lane i <- lane i - 1, lane 0 <- 0
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
  return ((uint64_t)hi << 32) | lo;
}
// Upper end of the window behind the threshold.  The row's last entry (the K-th nearest, K <= 63) never lies
// beyond lane 63, and lane 63 only ever moves inwards, so a candidate that finishes within the band
// (a <= a_last (1 + 4 band) + 1e-300) of the last entry is, whenever it is looked at, fewer than 4 band 2^53 ulps
// above the threshold's `a` (or below 1e-290): in integer terms — keys are the bit patterns of positive doubles —
// below tau + ulps.  `ulps` = 2^20 / 2^25 (twice what the band needs; the key truncation is 2^13 / 2^18).
// Location-less thresholds, and a row that is not full yet, have no window.
__device__ __forceinline__ uint64_t near_window(uint64_t tau, uint32_t SB, uint64_t ulps) {
  if (tau >= ((PM_KEY_NOLOC >> SB) << SB)) return tau;
  const uint64_t hi = tau + ulps, floor_bits = 0x03B8F2B061AEA073ull;  // bits of 1e-290, rounded up
  return (hi > floor_bits ? hi : floor_bits) | ((1ull << SB) - 1ull);
}
__device__ __forceinline__ void near_track(NearRow& r, uint64_t k, uint32_t site, bool cand) {
  uint64_t chg = __ballot(cand && (k < r.m1 || (site != r.s1 && k < r.m2)));
  while (chg) {  // (a city of co-located workers: the first one sets m1 / s1, the others change nothing)
    const uint32_t l = (uint32_t)__builtin_ctzll(chg);
    chg &= chg - 1ull;
    const uint64_t kk = readlane_u64(k, l);
    const uint32_t ss = (uint32_t)__builtin_amdgcn_readlane((int)site, (int)l);
    if (kk < r.m1) {
      if (ss != r.s1) r.m2 = r.m1;
      r.m1 = kk;
      r.s1 = ss;
    } else if (ss != r.s1 && kk < r.m2) {
      r.m2 = kk;
    }
  }
}
// the candidates of one 64-slot stride (k = ~0 where the lane has none; site = the candidate's site) against the row
template <typename SP>
__device__ __forceinline__ void near_row_offer(NearRow& r, uint64_t k, uint32_t site, SP cc_site, uint32_t SB, uint64_t ulps) {
  if (!__ballot(k < r.tau_hi)) return;
  uint64_t m = __ballot(k < r.tau);
  if (m) {
    const uint64_t old = r.key;
    do {
      const uint32_t l = (uint32_t)__builtin_ctzll(m);
      m &= m - 1ull;
      const uint64_t kk = readlane_u64(k, l);  // wave-uniform
      const uint64_t prev = wave_shr1_u64(r.key);
      const bool gt = r.key > kk, pgt = prev > kk;  // keys are distinct (the slot is in the low bits)
      r.key = gt ? (pgt ? prev : kk) : r.key;     // (a key that no longer beats lane 63 changes nothing)
    } while (m);
    r.tau = readlane_u64(r.key, 63u);
    r.tau_hi = near_window(r.tau, SB, ulps);
    const bool ev = old > r.tau && old < r.tau_hi;  // pushed out of the row, still near
    if (__ballot(ev)) {
      const uint32_t es = ev ? cc_site[(uint32_t)(old & ((1ull << SB) - 1ull))] : 0u;
      near_track(r, old, es, ev);
    }
  }
  const bool nm = k > r.tau && k < r.tau_hi;  // not (or no longer) in the row, but near
  if (__ballot(nm)) near_track(r, k, site, nm);
}

// The candidate columns of a tile of PROP_TILE slots, staged in LDS and shared by the four waves (= four seeds) of a
// workgroup: the sweep of a seed is then a chain of LDS reads (~64 cycles) instead of L2 round trips (~500), the
// next tile's global loads are in flight while the current one is consumed (two buffers), and every candidate
// row is fetched once per workgroup instead of once per seed.
#ifndef PROP_TILE
#define PROP_TILE 512u  // (256 / 512 / 1024 measure the same at 10 k and 100 k candidates)
#endif
#define PROP_TILE_PER_THREAD (PROP_TILE / 256u)
#define PROP_TILE_WORDS (PROP_TILE / 64u)
struct TileBuf {
  double x[PROP_TILE], y[PROP_TILE], z[PROP_TILE];  // unit vectors
  uint32_t site[PROP_TILE];
  uint64_t alive[PROP_TILE_WORDS], loc[PROP_TILE_WORDS];
};
struct TileRegs {  // one thread's share of a tile on its way from HBM/L2 to LDS: slots tid, tid + 256, ...
  double x[PROP_TILE_PER_THREAD], y[PROP_TILE_PER_THREAD], z[PROP_TILE_PER_THREAD];
  uint32_t site[PROP_TILE_PER_THREAD];
  uint64_t bm;
};
template <typename BP>
__device__ __forceinline__ void tile_fetch(const CarveArgs& p, BP alive, BP loc, uint32_t lw, uint32_t n_list,
                                           uint32_t tile, uint32_t tid, TileRegs& r) {
#pragma unroll
  for (uint32_t h = 0; h < PROP_TILE_PER_THREAD; ++h) {
    const uint32_t t = tile * PROP_TILE + h * 256u + tid;
    const uint32_t tc = t < n_list ? t : n_list - 1u;  // unconditional loads; the bitmaps are zero beyond the list
    r.x[h] = G(p.cc_ux)[tc];
    r.y[h] = G(p.cc_uy)[tc];
    r.z[h] = G(p.cc_uz)[tc];
    r.site[h] = G(p.cc_site)[tc];
  }
  const uint32_t j = tile * PROP_TILE_WORDS + (tid % PROP_TILE_WORDS);
  r.bm = (tid < 2u * PROP_TILE_WORDS && j < lw) ? (tid < PROP_TILE_WORDS ? alive[j] : loc[j]) : 0ull;
}
__device__ __forceinline__ void tile_store(TileBuf& tb, uint32_t tid, const TileRegs& r) {
#pragma unroll
  for (uint32_t h = 0; h < PROP_TILE_PER_THREAD; ++h) {
    const uint32_t o = h * 256u + tid;
    tb.x[o] = r.x[h];
    tb.y[o] = r.y[h];
    tb.z[o] = r.z[h];
    tb.site[o] = r.site[h];
  }
  if (tid < PROP_TILE_WORDS) tb.alive[tid] = r.bm;
  else if (tid < 2u * PROP_TILE_WORDS) tb.loc[tid - PROP_TILE_WORDS] = r.bm;
}
// keys of one seed against a staged tile: lane l owns the slots t = l (mod 64), as everywhere in this kernel.
// Straight-line code: every LDS read of four strides is issued up front, the chord key is computed for every lane,
// and a slot that does not count (dead, the seed itself, same shared site) becomes the key ~0, which never beats
// the row's threshold — a `continue` per condition would put an LDS round trip and a branch between each of them.
__device__ __forceinline__ void tile_keys(const CarveArgs& p, const TileBuf& tb, uint32_t tile, uint32_t lane, uint32_t s,
                                          bool shared, uint32_t ssite, const SeedGeo& sg, uint32_t SB, uint64_t ulps,
                                          NearRow& q, uint32_t& n_mine) {
#pragma unroll 1
  for (uint32_t h = 0; h < PROP_TILE / 256u; ++h) {
    double x[4], y[4], z[4];
    uint32_t si[4];
    uint64_t aw[4], lwd[4];
#pragma unroll
    for (uint32_t v = 0; v < 4u; ++v) {
      const uint32_t u = h * 4u + v, o = u * 64u + lane;
      aw[v] = tb.alive[u];
      lwd[v] = tb.loc[u];
      x[v] = tb.x[o];
      y[v] = tb.y[o];
      z[v] = tb.z[o];
      si[v] = tb.site[o];
    }
#pragma unroll
    for (uint32_t v = 0; v < 4u; ++v) {
      const uint32_t u = h * 4u + v, t = tile * PROP_TILE + u * 64u + lane;
      const bool located = (lwd[v] >> lane) & 1ull;
      // candidates at the seed's own (shared) site are at distance 0 (key = their slot): the ones behind the seed
      // head its row in slot order; the ones in front of it are dead by the time it is a seed (a live located slot
      // in front of it would be the seed instead) and would only fill the row
      const bool counts = ((aw[v] >> lane) & 1ull) && t != s && !(shared && located && si[v] == ssite && t < s);
      const double dx = x[v] - sg.ux, dy = y[v] - sg.uy, dz = z[v] - sg.uz;
      double a = 0.25 * fma(dx, dx, fma(dy, dy, dz * dz));
      // (see prox_a: the sine form below ~10 km.  A candidate at the seed's own site — identical coordinates — has the
      // Haversine term 0 in either form, exactly: sin(0) = 0; a city of co-located workers would otherwise send
      // nearly every stride of its seeds' sweeps through the sine polynomials)
      const bool same_site = located && si[v] == ssite;
      const bool near = counts && located && a < PM_A_CHORD_MIN && !same_site;
      a = same_site ? 0.0 : a;
      if (__ballot(near)) {
        if (near) a = hav_a(sg.lat, sg.lon, sg.cos, G(p.cc_lat)[t], G(p.cc_lon)[t], G(p.cc_cos)[t]);
      }
      const uint64_t kl = pack_key(located ? (uint64_t)__double_as_longlong(a) : PM_KEY_NOLOC, t, SB);
      near_row_offer(q, counts ? kl : ~0ull, si[v], G(p.cc_site), SB, ulps);
      n_mine += counts ? 1u : 0u;
    }
  }
}

// ---- Proposals from the spatial index (cell_*_kernel below).  The whole-list sweep above evaluates every candidate
// for every seed: 13,500 keys to pick 63 at 1M x 100k.  When the list is long and most of the indexed positions are
// still candidates, a seed instead walks the grid cells around its own cell, ring by ring (Chebyshev distance r in
// cell coordinates), and stops in front of the first ring that cannot hold anything of interest: every point of a
// cell at distance r is at least (r - 1) h away in one coordinate, hence in chord length, and what matters to a row
// is only what lies below its window (near_window: the threshold plus the certificate band).  Inside a ring each lane
// owns one run of cells along x — a contiguous range of the cell-sorted entries — and skips it when the box of the
// run is already out of reach of the seed's exact coordinates.  The row, its flags and the near-miss tracker come
// out bit for bit as the whole-list sweep produces them: both are functions of the SET of candidates below the final
// window, and the walk sees all of those.
__device__ __forceinline__ uint32_t cell_g_for(uint32_t n) {
  return n >= PM_CELL_BIG_N ? PM_CELL_G_MAX : n >= PM_CELL_MIN_N ? PM_CELL_G_MAX / 2u : 0u;
}
__device__ __forceinline__ uint32_t cell_coord(double v, uint32_t g) {  // v in [-1, 1]
  const double t = fmax((v + 1.0) * (double)(g >> 1), 0.0);  // (g / 2 is a power of two: the product is exact)
  const uint32_t c = (uint32_t)t;
  return c < g ? c : g - 1u;
}
// Lower bound of the key of anything at least `gx, gy, gz` away from the seed along the axes.  The gaps come from cell
// boundaries, which hold for the stored coordinates up to one rounding of (v + 1); 1e-9 (6 mm) per axis and 1e-6 of
// the result are far beyond that and beyond the difference between the chord form and the sine form of the key
// (relative 2^-31 at worst, the certificate band), and nothing next to a cell of 200 km.
__device__ __forceinline__ uint64_t cell_bound_key(double gx, double gy, double gz, uint32_t SB) {
  gx = fmax(gx - 1e-9, 0.0);
  gy = fmax(gy - 1e-9, 0.0);
  gz = fmax(gz - 1e-9, 0.0);
  const double a = 0.25 * (gx * gx + gy * gy + gz * gz) * (1.0 - 1e-6);
  return pack_key((uint64_t)__double_as_longlong(a), 0u, SB);
}
// one candidate per lane against the row: the key arithmetic of tile_keys, operands in registers
__device__ __forceinline__ void offer_candidate(const CarveArgs& p, const SeedGeo& sg, uint32_t ssite, double x, double y,
                                                double z, uint32_t si, uint32_t t, bool located, bool counts, uint32_t SB,
                                                uint64_t ulps, NearRow& q, uint32_t& n_mine) {
  const double dx = x - sg.ux, dy = y - sg.uy, dz = z - sg.uz;
  double a = 0.25 * fma(dx, dx, fma(dy, dy, dz * dz));
  const bool same_site = located && si == ssite;
  const bool near = counts && located && a < PM_A_CHORD_MIN && !same_site;
  a = same_site ? 0.0 : a;
  if (__ballot(near)) {
    if (near) a = hav_a(sg.lat, sg.lon, sg.cos, G(p.cc_lat)[t], G(p.cc_lon)[t], G(p.cc_cos)[t]);
  }
  const uint64_t kl = pack_key(located ? (uint64_t)__double_as_longlong(a) : PM_KEY_NOLOC, t, SB);
  near_row_offer(q, counts ? kl : ~0ull, si, G(p.cc_site), SB, ulps);
  n_mine += counts ? 1u : 0u;
}
// The entries of up to 64 runs — lane l holds run l: first entry b, length len (0 = none) — against the row.  The runs
// are short (a few cells of a few dozen positions), so they are laid end to end and cut into strides of 64: lane l of a
// stride finds its run by bisecting the running totals held across the lanes (6 ds_bpermute).  Most entries are not
// candidates of THIS list (another configuration's, or gone), and the walk of one seed is a chain of memory round trips,
// not a stream: so the first pass reads only the entries' slot words, four strides per trip, and packs the candidates
// it finds into the wave's LDS buffer; their coordinates are fetched in a second pass, over full strides.
#define CELL_BUF 768u  // candidates a wave collects before it drains them (2 words each)
__device__ __forceinline__ void cell_drain(const CarveArgs& p, uint32_t* wl, uint32_t& nb, uint32_t lane, uint32_t s, bool shared,
                                           uint32_t ssite, const SeedGeo& sg, uint32_t SB, uint64_t ulps, NearRow& q,
                                           uint32_t& n_mine) {
  const auto cs_site = G((const uint32_t*)p.cs_site);
  const auto cs_x = G((const double*)p.cs_ux);
  const auto cs_y = G((const double*)p.cs_uy);
  const auto cs_z = G((const double*)p.cs_uz);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the lanes read what other lanes of the wave packed)
  for (uint32_t k0 = 0; k0 < nb; k0 += 128u) {
    uint32_t t[2], si[2];
    double x[2], y[2], z[2];
    bool in[2];
#pragma unroll
    for (uint32_t v = 0; v < 2u; ++v) {
      const uint32_t k = k0 + v * 64u + lane;
      in[v] = k < nb;
      const uint32_t kc = in[v] ? k : 0u;
      const uint32_t ic = wl[kc];
      t[v] = wl[CELL_BUF + kc];
      x[v] = cs_x[ic];
      y[v] = cs_y[ic];
      z[v] = cs_z[ic];
      si[v] = cs_site[ic];
    }
#pragma unroll
    for (uint32_t v = 0; v < 2u; ++v) {
      if (!__ballot(in[v])) continue;
      // (every indexed position has a location; the slots in front of the seed at its own shared site: see tile_keys)
      const bool counts = in[v] && t[v] != s && !(shared && si[v] == ssite && t[v] < s);
      offer_candidate(p, sg, ssite, x[v], y[v], z[v], si[v], in[v] ? t[v] : 0u, true, counts, SB, ulps, q, n_mine);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  nb = 0;
}
template <bool STREAM = false>
__device__ __forceinline__ void cell_offer_runs(const CarveArgs& p, uint32_t* wl, uint32_t b, uint32_t len, uint32_t lane, uint32_t s,
                                                bool shared, uint32_t ssite, const SeedGeo& sg, uint32_t SB, uint64_t ulps,
                                                NearRow& q, uint32_t& n_mine,
                                                const uint32_t* cfg32 = nullptr) {
  const uint32_t incl = wave_incl_scan_u32(len);
  const uint32_t excl = incl - len;
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const auto cs_slot = G((const uint32_t*)p.cs_slot);
  uint32_t nb = 0;
  for (uint32_t f0 = 0; f0 < total; f0 += 256u) {
    uint32_t t[4], ic[4];
    bool in[4];
#pragma unroll
    for (uint32_t v = 0; v < 4u; ++v) {
      const uint32_t f = f0 + v * 64u + lane;
      in[v] = f < total;
      uint32_t j = 0;
#pragma unroll
      for (uint32_t step = 32u; step; step >>= 1) {
        const uint32_t pv = __shfl(incl, (int)(j + step - 1u), 64);
        j += pv <= f ? step : 0u;
      }
      j = j > 63u ? 63u : j;
      const uint32_t idx = __shfl(b, (int)j, 64) + (f - __shfl(excl, (int)j, 64));
      ic[v] = in[v] ? idx : 0u;
      t[v] = cs_slot[ic[v]];
    }
    // (streaming carve: an entry's slot is its position for the whole carve; whether it is a candidate of the
    // configuration being carved, and still free, is one bit of the validator's published bitmap)
    uint32_t cw[4] = {~0u, ~0u, ~0u, ~0u};
    if (STREAM) {  // free (the validator's published bitmap, a little behind) and compatible with the configuration
      const auto freeg = G((const uint32_t*)p.bits_scratch);
      const auto cfgb = G(cfg32);
#pragma unroll
      for (uint32_t v = 0; v < 4u; ++v) {
        const uint32_t wi = (in[v] ? t[v] : 0u) >> 5;
        cw[v] = __hip_atomic_load(&freeg[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & cfgb[wi];
      }
    }
#pragma unroll
    for (uint32_t v = 0; v < 4u; ++v) {
      const bool cand = in[v] && t[v] != 0xFFFFFFFFu && (!STREAM || ((cw[v] >> (t[v] & 31u)) & 1u) != 0u);
      const uint64_t m = __ballot(cand);
      if (cand) {
        const uint32_t at = nb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        wl[at] = ic[v];
        wl[CELL_BUF + at] = t[v];
      }
      nb += (uint32_t)__popcll(m);
    }
    if (nb > CELL_BUF - 256u) cell_drain(p, wl, nb, lane, s, shared, ssite, sg, SB, ulps, q, n_mine);
  }
  if (nb) cell_drain(p, wl, nb, lane, s, shared, ssite, sg, SB, ulps, q, n_mine);
}
// run qi of ring r around cell (cx, cy, cz): where its entries begin and end in cell_start, and the lower bound of
// the keys in its box.  false = the run lies outside the grid.
__device__ __forceinline__ bool cell_run(uint32_t r, uint32_t qi, int cx, int cy, int cz, uint32_t g, double h, const SeedGeo& sg,
                                         uint32_t SB, uint32_t* lin_b, uint32_t* lin_e, uint64_t* lbk) {
  const int G1 = (int)g;
  int dy = 0, dz = 0, x0 = cx, x1 = cx;
  if (r > 0u) {
    const uint32_t side = 2u * r, n_full = 8u * r, inner = 2u * r - 1u;
    if (qi < n_full) {  // the rim of the (2r + 1)^2 square of (dy, dz): whole runs -r .. r along x
      const uint32_t sd = qi / side, t = qi - sd * side;
      const int R = (int)r, T = (int)t;
      dy = sd == 0u ? -R + T : sd == 1u ? R : sd == 2u ? R - T : -R;
      dz = sd == 0u ? -R : sd == 1u ? -R + T : sd == 2u ? R : R - T;
      x0 = cx - R;
      x1 = cx + R;
    } else {  // the inside of the square: the two end cells x = -r and x = r
      const uint32_t m = (qi - n_full) >> 1;
      dy = (int)(m % inner) - (int)(r - 1u);
      dz = (int)(m / inner) - (int)(r - 1u);
      x0 = x1 = ((qi - n_full) & 1u) ? cx + (int)r : cx - (int)r;
    }
  }
  const int y = cy + dy, z = cz + dz;
  const bool ok = y >= 0 && y < G1 && z >= 0 && z < G1 && x1 >= 0 && x0 < G1;
  x0 = x0 < 0 ? 0 : x0;
  x1 = x1 >= G1 ? G1 - 1 : x1;
  // the box of the run against the seed's own coordinates
  const double xlo = (double)x0 * h - 1.0, xhi = (double)(x1 + 1) * h - 1.0;
  const double ylo = (double)y * h - 1.0, yhi = ylo + h, zlo = (double)z * h - 1.0, zhi = zlo + h;
  const double gx = fmax(fmax(xlo - sg.ux, sg.ux - xhi), 0.0), gy = fmax(fmax(ylo - sg.uy, sg.uy - yhi), 0.0),
               gz = fmax(fmax(zlo - sg.uz, sg.uz - zhi), 0.0);
  *lbk = cell_bound_key(gx, gy, gz, SB);
  const uint32_t row = ok ? ((uint32_t)z * g + (uint32_t)y) * g : 0u;
  *lin_b = row + (uint32_t)(ok ? x0 : 0);
  *lin_e = row + (uint32_t)(ok ? x1 : 0) + 1u;
  return ok;
}
__device__ __forceinline__ uint32_t ring_runs(uint32_t r) { return r ? 8u * r + 2u * (2u * r - 1u) * (2u * r - 1u) : 1u; }
// The walk.  Returns the ring in front of which it stopped (>= 2), or 0 = the rings ran out before the row's window
// closed (a seed far from everything else, or fewer located candidates than a row holds): the caller starts over on
// the whole list.
template <bool STREAM = false>
__device__ __forceinline__ uint32_t cell_walk(const CarveArgs& p, uint32_t* wl, uint32_t g, uint32_t r_max, uint32_t lane, uint32_t s,
                                          bool shared, uint32_t ssite, const SeedGeo& sg, uint32_t SB, uint64_t ulps, NearRow& q,
                                          uint32_t& n_mine,
                                          const uint32_t* cfg32 = nullptr) {
  const double h = 2.0 / (double)g;
  const int cx = (int)cell_coord(sg.ux, g), cy = (int)cell_coord(sg.uy, g), cz = (int)cell_coord(sg.uz, g);
  const auto cstart = G((const uint32_t*)p.cell_start);
  {
    // rings 0, 1 and 2 are 1 + 10 + 34 runs: one trip to cell_start for all of them.  Ring 1 always counts (a cell
    // next door can hold a point a hair away); ring 2 only if the window is still open behind ring 1.
    static_assert(1u + 10u + 34u <= 64u, "three rings in one wave");
    const uint32_t r = lane == 0u ? 0u : lane <= 10u ? 1u : 2u, qi = lane == 0u ? 0u : lane <= 10u ? lane - 1u : lane - 11u;
    uint32_t lb, le;
    uint64_t lbk;
    const bool ok = lane < 45u && (r < g) && cell_run(r, qi, cx, cy, cz, g, h, sg, SB, &lb, &le, &lbk);
    uint32_t b = 0, e = 0;
    if (ok) {
      b = cstart[lb];
      e = cstart[le];
    }
    cell_offer_runs<STREAM>(p, wl, b, (ok && lane <= 10u) ? e - b : 0u, lane, s, shared, ssite, sg, SB, ulps, q, n_mine, cfg32);
    if (cell_bound_key(h, 0.0, 0.0, SB) > q.tau_hi) return 2u;
    if (r_max < 2u || g <= 2u) return 0u;
    cell_offer_runs<STREAM>(p, wl, b, (ok && lane > 10u && !(lbk > q.tau_hi)) ? e - b : 0u, lane, s, shared, ssite, sg, SB, ulps, q, n_mine, cfg32);
  }
  for (uint32_t r = 3u;; ++r) {
    if (cell_bound_key((double)(r - 1u) * h, 0.0, 0.0, SB) > q.tau_hi) return r;  // nothing of interest from this ring on
    if (r > r_max || r >= g) return 0u;
    const uint32_t n_runs = ring_runs(r);
    for (uint32_t q0 = 0; q0 < n_runs; q0 += 64u) {
      const uint32_t qi = q0 + lane;
      uint32_t lb, le;
      uint64_t lbk;
      bool ok = qi < n_runs && cell_run(r, qi < n_runs ? qi : 0u, cx, cy, cz, g, h, sg, SB, &lb, &le, &lbk);
      ok = ok && !(lbk > q.tau_hi);
      uint32_t b = 0, e = 0;
      if (ok) {
        b = cstart[lb];
        e = cstart[le];
      }
      if (!__ballot(ok && e > b)) continue;
      cell_offer_runs<STREAM>(p, wl, b, ok ? e - b : 0u, lane, s, shared, ssite, sg, SB, ulps, q, n_mine, cfg32);
    }
  }
}
// the whole list by one wave, from L2/HBM (the fallback of cell_walk; four strides per trip for the loads to overlap)
template <bool STREAM = false>
__device__ __forceinline__ void list_sweep_solo(const CarveArgs& p, uint32_t n_list, uint32_t lane, uint32_t s, bool shared,
                                             uint32_t ssite, const SeedGeo& sg, uint32_t SB, uint64_t ulps, NearRow& q,
                                             uint32_t& n_mine,
                                             const uint64_t* cfg64 = nullptr) {
  const auto alive = G((const uint64_t*)p.bits_scratch);
  const auto loc = G((const uint64_t*)p.bits_scratch) + p.bits_stride;
  for (uint32_t t0 = 0; t0 < n_list; t0 += 256u) {
    double x[4], y[4], z[4];
    uint32_t si[4];
    uint64_t aw[4], lw[4];
#pragma unroll
    for (uint32_t v = 0; v < 4u; ++v) {
      const uint32_t tb = t0 + v * 64u, t = tb + lane, tc = t < n_list ? t : n_list - 1u;
      const bool w_in = tb < n_list;
      // (the bitmaps are zero beyond the list; streaming carve: the validator clears bits while this runs)
      aw[v] = !w_in ? 0ull : STREAM ? (__hip_atomic_load(&alive[tb >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & G(cfg64)[tb >> 6]) : alive[tb >> 6];
      lw[v] = w_in ? loc[tb >> 6] : 0ull;
      x[v] = G(p.cc_ux)[tc];
      y[v] = G(p.cc_uy)[tc];
      z[v] = G(p.cc_uz)[tc];
      si[v] = G(p.cc_site)[tc];
    }
#pragma unroll
    for (uint32_t v = 0; v < 4u; ++v) {
      const uint32_t t = t0 + v * 64u + lane;
      const bool located = (lw[v] >> lane) & 1ull;
      const bool counts = ((aw[v] >> lane) & 1ull) && t != s && !(shared && located && si[v] == ssite && t < s);
      offer_candidate(p, sg, ssite, x[v], y[v], z[v], si[v], counts ? t : 0u, located, counts, SB, ulps, q, n_mine);
    }
  }
}

// The finished register of a seed's sweep -> its row: *mine_out = this lane's entry (~0 beyond the row's K entries);
// returns the flags word (PM_ROW_*, entries in bits 0..7, first entry within the band of the last one in bits 8..15).
__device__ __forceinline__ uint32_t near_row_finish(const CarveArgs& p, const NearRow& q, bool valid, uint32_t K, uint32_t SB,
                                                    double TIE_BAND, uint32_t lane, uint64_t* mine_out) {
  // ---- the K nearest in (key, slot) order are lanes 0 .. K-1 of the row; lane K holds the first unlisted one
  const uint32_t n_tot = valid ? (uint32_t)__popcll(__ballot(q.key != ~0ull)) : 0u;
  const uint32_t n_k = n_tot < K ? n_tot : K;
  const uint64_t beyond = n_tot > K ? readlane_u64(q.key, K) : ~0ull;
  const uint64_t mine = lane < n_k ? q.key : ~0ull;
  const uint64_t noloc_kb = (PM_KEY_NOLOC >> SB) << SB;
  // row certificates the validator can rely on instead of re-deriving them at every step:
  //  clean      — no two neighbouring entries within the band of each other sit at different sites (entries in
  //               between are within the band too, so this covers every pair of the row)
  //  tail_clear — the first candidate NOT in the row is further than the band from the last entry
  //  tail_ok    — otherwise: everything unlisted within the band of the last entry sits at that entry's site
  uint32_t clean = 1, tail_clear = 0, tail_ok = 0;
  int tail_bad = 0;
  uint64_t e_last = 0;
  double a_last = 0.0, band2 = 0.0;
  uint32_t site_last = 0;
  if (valid) {
    const uint64_t kb = (mine >> SB) << SB;
    const uint32_t my_site = (lane < n_k && kb != noloc_kb) ? G(p.cc_site)[(uint32_t)(mine & ((1ull << SB) - 1ull))] : 0u;
    const uint64_t nkb_lo = __shfl_down((uint32_t)kb, 1, 64), nkb_hi = __shfl_down((uint32_t)(kb >> 32), 1, 64);
    const uint64_t nkb = (nkb_hi << 32) | nkb_lo;
    const uint32_t nsite = __shfl_down(my_site, 1, 64);
    int bad = 0;
    if (lane + 1u < n_k && kb != noloc_kb && nkb != noloc_kb) {
      const double a0 = __longlong_as_double((long long)kb), a1 = __longlong_as_double((long long)nkb);
      if (a1 - a0 <= a1 * (4.0 * TIE_BAND) + 1e-300 && nsite != my_site) bad = 1;
    }
    clean = __ballot(bad) == 0ull;
  }
  if (valid && n_k == K) {
    e_last = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), (int)K - 1) << 32) |
             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, (int)K - 1);
    const uint64_t kb_last = (e_last >> SB) << SB;
    const uint64_t kb_beyond = (beyond >> SB) << SB;
    if (beyond == ~0ull || kb_last == noloc_kb) {
      tail_clear = 1;  // nothing unlisted, or only location-less candidates (exact ties, larger slots)
    } else if (kb_beyond == noloc_kb) {
      tail_clear = 1;
    } else {
      a_last = __longlong_as_double((long long)kb_last);
      const double a_b = __longlong_as_double((long long)kb_beyond);
      if (a_b - a_last > a_b * (4.0 * TIE_BAND) + 1e-300) {
        tail_clear = 1;
      } else {
        site_last = G(p.cc_site)[(uint32_t)(e_last & ((1ull << SB) - 1ull))];
        band2 = a_last * (4.0 * TIE_BAND) + 1e-300;
        // The unlisted candidates closest to the last entry are lanes K .. 63 of the sorted register; whatever
        // else came near the row during the sweep is summarised in the tracker (see NearRow): of those, only
        // the nearest one that does NOT sit at the last entry's site can break the certificate.
        if (lane >= K && lane < n_tot) {
          const uint64_t kb2 = (q.key >> SB) << SB;
          if (kb2 != noloc_kb && __longlong_as_double((long long)kb2) - a_last <= band2 &&
              G(p.cc_site)[(uint32_t)(q.key & ((1ull << SB) - 1ull))] != site_last)
            tail_bad = 1;
        }
        const uint64_t other = q.s1 != site_last ? q.m1 : q.m2;
        const uint64_t kb_o = (other >> SB) << SB;
        if (other != ~0ull && kb_o != noloc_kb && __longlong_as_double((long long)kb_o) - a_last <= band2) tail_bad = 1;
        tail_ok = __ballot(tail_bad) == 0ull;
      }
    }
  }
  // what the validator's chain needs to settle a step from the flags alone: is any listed term near the antipode,
  // and from which entry on does the row lie within the certificate band of its LAST entry (a selection that ends
  // in front of that entry has nothing to do with the row's tail) — the validator's own band expression
  uint32_t safe, j_tail = n_k;
  {
    const uint64_t kb = (mine >> SB) << SB;
    const bool located = lane < n_k && kb != noloc_kb;
    const double a_l = __longlong_as_double((long long)kb);
    safe = __ballot(located && a_l > PM_A_MAX_SAFE) == 0ull;
    if (n_k > 0u) {
      const uint64_t kb_le = (readlane_u64(mine, n_k - 1u) >> SB) << SB;
      if (kb_le != noloc_kb) {
        const double a_le = __longlong_as_double((long long)kb_le);
        const uint64_t within = __ballot(located && (a_le - a_l) <= a_l * TIE_BAND + 1e-300);
        j_tail = within ? (uint32_t)__builtin_ctzll(within) : n_k;
      }
    }
  }
  const uint32_t meta = n_k | (j_tail << 8) | (safe ? PM_ROW_SAFE : 0u) | ((n_k < K) ? PM_ROW_COMPLETE : 0u) |
                        (tail_ok ? PM_ROW_TAIL_OK : 0u) | (clean ? PM_ROW_CLEAN : 0u) | (tail_clear ? PM_ROW_TAIL_CLEAR : 0u);
  *mine_out = mine;
  return meta;
}

__global__ __launch_bounds__(256) void carve_propose_kernel(const CarveArgs* __restrict__ pa) {
  const CarveArgs& p = *pa;  // argument block in device memory: read through the scalar cache, never copied
  const auto st = G((const CarveStatus*)p.status);
  if (st->state != CARVE_STATE_RUNNING) return;
  const auto D = G((const BatchDesc*)p.desc);
  if (!D->planned || !D->valid) return;
  const uint32_t K = D->prop_k, n_list = D->n_list;
  const uint32_t world = p.dist_world, my_rank = p.dist_rank;
  const auto prop_out = G(p.prop_send);
  if (K == 0 || n_list > PM_CARVE_BIG_SLOTS) return;
  const uint32_t SB = n_list > PM_CARVE_SLOTS ? PM_CARVE_SLOT_BITS_BIG : PM_CARVE_SLOT_BITS;
  const double TIE_BAND = n_list > PM_CARVE_SLOTS ? PM_TIE_BAND_BIG : PM_TIE_BAND;
  const uint64_t WINDOW_ULPS = n_list > PM_CARVE_SLOTS ? (1ull << 25) : (1ull << 20);  // 8 band 2^53 (see near_window)
  const uint32_t lane = threadIdx.x & 63u;
  const auto alive = G((const uint64_t*)p.bits_scratch);
  const auto loc = G((const uint64_t*)p.bits_scratch) + p.bits_stride;
  const uint32_t lw = (n_list + 63u) >> 6;
  // ---- the neighbour rows.  Four seeds per workgroup (one per wave) sweep the candidate list together,
  // tile by tile through LDS.  The seeds of the batch are dealt round-robin over the ranks: this rank computes
  // seed numbers my_rank, my_rank + world, ...
  __shared__ TileBuf tiles[2];
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  const uint32_t n_seeds = D->n_seeds, cell_g = D->cell_g;
  const uint32_t n_my = world > 1u ? (n_seeds > my_rank ? (n_seeds - my_rank + world - 1u) / world : 0u) : n_seeds;
  const uint32_t n_tiles = (n_list + PROP_TILE - 1u) / PROP_TILE;
  const auto seed_slots = G((const uint32_t*)p.seed_slots);
  for (uint32_t k0 = blockIdx.x * 4u; k0 < n_my; k0 += gridDim.x * 4u) {
    const uint32_t out_row = k0 + wave;  // row in this rank's send segment
    const bool valid = out_row < n_my;
    const uint32_t s = valid ? seed_slots[world > 1u ? my_rank + world * out_row : out_row] : 0u;
    const uint32_t ssite = G(p.cc_site)[s];
#ifdef PM_PROP_PROF
    uint64_t pt = __builtin_amdgcn_s_memtime(), pt_same = 0, pt_sweep = 0, pt_pop = 0, pt_flags = 0;
    (void)pt_same; (void)pt_sweep; (void)pt_pop; (void)pt_flags;
#define PP_MARK(var) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); var += t_ - pt; pt = t_; } while (0)
#else
#define PP_MARK(var)
#endif
    const SeedGeo sg = {G(p.cc_lat)[s], G(p.cc_lon)[s], G(p.cc_cos)[s], G(p.cc_ux)[s], G(p.cc_uy)[s], G(p.cc_uz)[s]};
    const bool shared = (ssite & 0x80000000u) != 0u;
    NearRow q = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, 0xFFFFFFFFu};
    uint32_t n_mine = 0;
    if (cell_g) {  // (the same for every workgroup of the launch: no barrier on this side)
      if (valid) {
        // (the tiles' LDS is free on this side: a candidate buffer per wave)
        static_assert(4u * 2u * CELL_BUF * sizeof(uint32_t) <= sizeof(tiles), "four candidate buffers in the tiles' LDS");
        uint32_t* wl = reinterpret_cast<uint32_t*>(tiles) + wave * (2u * CELL_BUF);
#ifdef PM_BATCH_LOG
        const uint64_t wt0 = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t stop_r = p.prune_mode != 3u ? cell_walk(p, wl, cell_g, PM_CELL_RMAX, lane, s, shared, ssite, sg, SB, WINDOW_ULPS, q, n_mine) : 0u;
#ifdef PM_BATCH_LOG
        if (lane == 0) {  // (experiment builds: where the walks stopped, what they cost)
          const uint64_t dt = __builtin_amdgcn_s_memtime() - wt0;
          unsigned long long* pr = (unsigned long long*)p.status->prof;
          atomicAdd(&pr[0], (unsigned long long)dt);
          atomicMax(&pr[1], (unsigned long long)dt);
          atomicAdd(&pr[2], 1ull);
          atomicAdd(&pr[3], (unsigned long long)n_mine);
          atomicAdd(&pr[16u + (stop_r < 15u ? stop_r : 15u)], 1ull);
        }
#endif
        if (!stop_r) {
          q = NearRow{~0ull, ~0ull, ~0ull, ~0ull, ~0ull, 0xFFFFFFFFu};
          n_mine = 0;
          list_sweep_solo(p, n_list, lane, s, shared, ssite, sg, SB, WINDOW_ULPS, q, n_mine);
          if (lane == 0) atomicAdd(&p.status->prune_fallbacks, 1u);
        }
      }
    } else {
      TileRegs tr;
      tile_fetch(p, alive, loc, lw, n_list, 0u, tid, tr);
      tile_store(tiles[0], tid, tr);
      __syncthreads();
      for (uint32_t t = 0; t < n_tiles; ++t) {
        const bool more = t + 1u < n_tiles;
        if (more) tile_fetch(p, alive, loc, lw, n_list, t + 1u, tid, tr);  // in flight while this tile is consumed
        if (valid) tile_keys(p, tiles[t & 1u], t, lane, s, shared, ssite, sg, SB, WINDOW_ULPS, q, n_mine);
        if (more) tile_store(tiles[(t + 1u) & 1u], tid, tr);
        __syncthreads();
      }
    }
    PP_MARK(pt_sweep);
    uint64_t mine;
    const uint32_t meta = near_row_finish(p, q, valid, K, SB, TIE_BAND, lane, &mine);
    if (!valid) continue;  // (the workgroup's last seeds may be fewer than four)
    PP_MARK(pt_flags);
#ifdef PM_PROP_PROF
    {
      if (lane == 0) {
        unsigned long long* pr = (unsigned long long*)p.status->prof;
#ifndef PM_CARVE_PROF_FINE  // (the fine build uses these slots for the validator's round phases)
        atomicAdd(&pr[5], (unsigned long long)pt_same);
        atomicAdd(&pr[6], (unsigned long long)pt_sweep);
        atomicAdd(&pr[7], (unsigned long long)pt_pop);
        atomicAdd(&pr[8], (unsigned long long)pt_flags);
        atomicMax(&pr[23], (unsigned long long)(pt_same + pt_sweep + pt_pop + pt_flags));
#endif
        atomicAdd(&pr[24], 1ull);
        atomicAdd(&pr[25], (q.m1 != ~0ull) ? 1ull : 0ull);
      }
    }
#endif
    if (p.count_keys) {  // bookkeeping for the roofline of this kernel (bench only): keys this sweep evaluated
      const uint32_t swept = wave_sum(n_mine);
      if (lane == 0) {
        atomicAdd((unsigned long long*)&p.status->prop_keys, (unsigned long long)swept);
        atomicAdd(&p.status->n_props, 1u);
      }
    }
    // the row: the flags word, then the K sorted entries
    prop_out[(size_t)out_row * PM_PROP_ROW + ((lane + 1u) & 63u)] = lane == 63u ? (uint64_t)meta : mine;
    // ... and the same once more as 32-bit words (flags, slot of entry 0, slot of entry 1, ...): what a lane of
    // the chain's producer reads — 256 coalesced bytes per row
    reinterpret_cast<__attribute__((address_space(1))) uint32_t*>(prop_out + (size_t)out_row * PM_PROP_ROW + PM_PROP_SLOTS)[(lane + 1u) & 63u] =
        lane == 63u ? meta : (uint32_t)(mine & ((1ull << SB) - 1ull));
  }
}

#ifndef PM_PROP_CAP_DIV_WALK
#define PM_PROP_CAP_DIV_WALK 10u
#endif
#ifndef PM_PROP_CAP_DIV_BIG
#define PM_PROP_CAP_DIV_BIG 10u
#endif
#ifndef PM_PROP_CAP_DIV
#define PM_PROP_CAP_DIV 5u
#endif
// One proposal per located slot, at most PM_PROP_MAX_SEEDS per batch.  Returns the slot after the word in which
// the cap-th located slot falls (a later batch covers the rest), or n_list; *n_seeds = the located live slots
// below it (the batch's seeds).  Also records, per bitmap word below the limit, the seed bitmap and the number of
// seeds in front of the word (seed_map / seed_prefix): the rank of a slot among the seeds is the row its
// proposal is stored in (proposer and validator derive it from these two words).
// (one wave; the results are valid in every lane)
__device__ __forceinline__ void prop_limit_scan(const CarveArgs& p, uint32_t n_list, uint32_t lane, uint32_t* limit_out,
                                                uint32_t* n_seeds_out, bool walk = false) {
  // The configuration is re-prepared (and re-proposed) once half of its list is dead; by then the seed
  // pointer has advanced through roughly the first eighth of the slots (each group removes max_s slots
  // spread over the whole list), so later slots never consume this round's proposals: cap the batch.
  // (big lists: a tenth; measured again with the sorted-lane proposer at 1M x 100k, carve through the stepwise
  // tick: 1/6 26.3 ms, 1/8 22.5, 1/10 22.0, 1/14 22.7; small lists 1/3 .. 1/7 all within 1 %)
  // (a batch whose seeds walk the spatial index pays per seed, not per seed and candidate: it can afford more of them)
  uint32_t cap = walk ? n_list / (p.walk_cap_div ? p.walk_cap_div : PM_PROP_CAP_DIV_WALK) : n_list > PM_CARVE_SLOTS ? n_list / PM_PROP_CAP_DIV_BIG : n_list / PM_PROP_CAP_DIV;
  if (cap < 512u) cap = 512u;
  if (cap > PM_PROP_MAX_SEEDS) cap = PM_PROP_MAX_SEEDS;
  const auto g_al = G((const uint64_t*)p.bits_scratch);
  const auto g_lc = G((const uint64_t*)p.bits_scratch) + p.bits_stride;
  const auto seed_map = G(p.seed_map);
  const auto seed_prefix = G(p.seed_prefix);
  const uint32_t lwp = (n_list + 63u) >> 6;
  uint32_t acc = 0, limit = n_list;
  for (uint32_t j0 = 0; j0 < lwp; j0 += 64u) {
    const uint32_t j = j0 + lane;
    const uint64_t m = j < lwp ? (g_al[j] & g_lc[j]) : 0ull;  // bits beyond n_list are zero in both bitmaps
    const uint32_t cnt = (uint32_t)__popcll(m);
    const uint32_t incl = wave_incl_scan_u32(cnt);
    const uint64_t over = __ballot(acc + incl >= cap);
    const uint32_t last = over ? (uint32_t)__builtin_ctzll(over) : 63u;  // last word of this pass inside the batch
    if (j < lwp && lane <= last) {
      seed_map[j] = m;
      seed_prefix[j] = acc + incl - cnt;
    }
    acc += __shfl(incl, (int)last, 64);
    if (over) {
      limit = (j0 + last + 1u) * 64u;
      break;
    }
  }
  *limit_out = limit < n_list ? limit : n_list;
  *n_seeds_out = acc;
}

// seed number -> slot (the proposer takes its seeds from this dense list): every thread of the workgroup expands
// the bitmap words it owns, after prop_limit_scan (and a barrier) have produced seed_map / seed_prefix
__device__ __forceinline__ void prop_seed_slots(const CarveArgs& p, uint32_t limit, uint32_t tid, uint32_t n_threads) {
  const auto seed_map = G((const uint64_t*)p.seed_map);
  const auto seed_prefix = G((const uint32_t*)p.seed_prefix);
  const auto seed_slots = G(p.seed_slots);
  const uint32_t words = (limit + 63u) >> 6;
  for (uint32_t j = tid; j < words; j += n_threads) {
    uint32_t o = seed_prefix[j];
    for (uint64_t mm = seed_map[j]; mm; mm &= mm - 1ull) seed_slots[o++] = j * 64u + (uint32_t)__builtin_ctzll(mm);
  }
}

__device__ __noinline__ uint32_t carve_prop_limit(const CarveArgs& p, BlockRed& red, uint32_t n_list, uint32_t* n_seeds) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (wave == 0) {
    uint32_t limit, acc;
    prop_limit_scan(p, n_list, lane, &limit, &acc);
    if (lane == 0) {
      red.b[0] = limit;
      red.b[1] = acc;
    }
  }
  __syncthreads();
  const uint32_t r = red.b[0];
  *n_seeds = red.b[1];
  prop_seed_slots(p, r, threadIdx.x, CARVE_THREADS);
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------------------------------------
// Full-chip preparation of the next candidate list (proposal-driven FORM carve).  The validator is one
// workgroup; compacting a 100 k-entry eligible list (and scattering group_of for the groups it just formed) on
// one CU took a third of its time.  Between two validation launches run, on every CU:
//   carve_prep_count_kernel   group_of for the groups of the last launch; per block, per remaining configuration,
//                             the number of live compatible positions (+ global totals); clears the slot loc bitmap
//   carve_prep_place_kernel   every block picks the same next configuration from the totals (mod.rs:505-519: the
//                             first whose loop would be entered), derives its offset from the per-block counts
//                             and places its candidates (stable: slot order = input order); the block that
//                             finishes last computes the proposal batch (prop_limit_scan) and publishes the status
// One 64-position word per wave, four waves per block.

#define PREP_WAVES 4

// The plan of a preparation, decided once (one thread) so that every block of the two kernels behind it works on the
// same configuration: where the search for the next configuration starts.  Beside a validation in flight (speculative
// mode) that is a guess: the batch in front works on configuration ci of a list of n_list slots; it ends with the
// list thinned out — the same configuration again — unless the list is small enough to be finished in one go.
__device__ __forceinline__ void plan_batch(const CarveArgs& p) {  // (one thread)
  const uint32_t cur = p.status->cur_ci;
  uint32_t ci0 = cur;
  const BatchDesc dp = *p.desc_prev;
  if (p.speculative && p.desc_prev != p.desc && dp.planned && dp.valid && dp.ci0 == cur)
    ci0 = dp.n_list > 256u ? dp.ci : dp.ci + 1u;
  BatchDesc d = {};
  d.planned = 1u;
  d.ci0 = ci0;
  d.total_available = p.status->total_available;
  *p.desc = d;
}
__global__ __launch_bounds__(128) void carve_plan_kernel(const CarveArgs* __restrict__ pa) {
  static_assert(PM_MAX_CONFIGS + 2u <= 128u, "one thread per counter");
  const CarveArgs& p = *pa;
  const auto st = G(p.status);
  const uint32_t tid = threadIdx.x;
  if (st->state != CARVE_STATE_RUNNING) {
    if (tid == 0) p.desc->planned = 0u;
    return;
  }
  if (tid <= PM_MAX_CONFIGS + 1u) p.prep_counts[tid] = 0u;  // totals + ticket of this preparation
  if (tid == 0) plan_batch(p);
}

__global__ __launch_bounds__(256) void carve_prep_count_kernel(const CarveArgs* __restrict__ pa) {
  const CarveArgs& p = *pa;
  const auto st = G(p.status);
  const auto D = G((const BatchDesc*)p.desc);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t wave_g = blockIdx.x * PREP_WAVES + wave;
  if (!p.speculative) {
    // group_of for the groups the last validation launch appended (idempotent; also runs after the carve ended).
    // (Beside a validation in flight the validator does it itself.)
    const uint32_t g_lo = st->g_lo, g_hi = st->g_hi, n_waves = gridDim.x * PREP_WAVES;
    for (uint32_t g = g_lo + wave_g; g < g_hi; g += n_waves) {
      const uint32_t off = G(p.g_off)[g], gn = G(p.g_n)[g];
      for (uint32_t k = lane; k < gn; k += 64u) G(p.group_of)[G(p.members)[off + k]] = (int32_t)g;
    }
  }
  if (st->state != CARVE_STATE_RUNNING) {
    if (!p.speculative && blockIdx.x == 0 && tid == 0) p.desc->planned = 0u;
    return;
  }
#ifdef PM_BATCH_LOG
  if (blockIdx.x == 0 && tid == 0) p.status->prof[5] = ~0ull;  // (earliest block start of the placement behind this)
#endif
  uint32_t ci0;
  if (!p.speculative) {  // one batch at a time: the plan is the carve's own state, the same in every block
    ci0 = st->cur_ci;
    if (blockIdx.x == 0 && tid == 0) plan_batch(p);
  } else {
    if (!D->planned) return;
    ci0 = D->ci0;
  }
  if (ci0 >= p.n_avail) return;
  __shared__ uint32_t s_cnt[PREP_WAVES][PM_MAX_CONFIGS];
  const uint32_t n = st->n_eligible, n_words = (n + 63u) >> 6;
  const uint32_t j = wave_g;  // this wave's word of the position space
  uint64_t m = 0;
  if (j < n_words) {
    const uint32_t i = j * 64u + lane;
    const uint64_t aw = G(p.alive_g)[j];  // (a validation in flight may be clearing bits: the snapshot is what counts)
    const bool alive = i < n && ((aw >> lane) & 1ull);
    m = alive ? G((const uint64_t*)p.c_compat)[i] : 0ull;
    if (lane == 0) {
      G(p.alive_snap)[j] = aw;
      G(p.bits_scratch)[p.bits_stride + j] = 0ull;  // slot loc bitmap: the placement ORs its bits in
    }
  }
  for (uint32_t ci = ci0; ci < p.n_avail; ++ci) {
    const uint32_t cnt = (uint32_t)__popcll(__ballot((m >> p.avail_cfg[ci]) & 1ull));
    if (lane == 0) s_cnt[wave][ci] = cnt;
  }
  __syncthreads();
  if (tid >= ci0 && tid < p.n_avail) {
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < PREP_WAVES; ++w) sum += s_cnt[w][tid];
    G(p.prep_block_counts)[(size_t)blockIdx.x * PM_MAX_CONFIGS + tid] = sum;
    if (sum) atomicAdd(&p.prep_counts[tid], sum);
  }
}

__global__ __launch_bounds__(256) void carve_prep_place_kernel(const CarveArgs* __restrict__ pa) {
  const CarveArgs& p = *pa;
  const auto st = G(p.status);
  const auto D = G((const BatchDesc*)p.desc);
  if (st->state != CARVE_STATE_RUNNING || !D->planned) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  __shared__ uint32_t s_red[PREP_WAVES + 4];
  __shared__ uint32_t s_bits[PREP_WAVES][64];
#ifdef PM_BATCH_LOG
  const uint64_t pl_t0 = __builtin_amdgcn_s_memtime();
  if (tid == 0) atomicMin((unsigned long long*)&p.status->prof[5], (unsigned long long)pl_t0);
#endif
  const uint32_t n = st->n_eligible, n_words = (n + 63u) >> 6;
  const uint32_t total_available = D->total_available;
  // ---- the next configuration whose loop would be entered (mod.rs:505-519), the same in every block.  (Counts and
  // total_available only ever shrink: a configuration that cannot be entered by these numbers cannot be entered by
  // the validator's either, so skipping it is final.)
  uint32_t ci = D->ci0, n_list = 0;
  for (; ci < p.n_avail; ++ci) {
    const uint32_t min_s = p.min_size[ci];
    if (total_available < min_s) continue;             // `while` never entered (:507)
    n_list = G((const uint32_t*)p.prep_counts)[ci];
    if (n_list < min_s || n_list == 0) continue;       // :517-519
    break;
  }
  const bool none = ci >= p.n_avail;
  // proposals for this list?  (the configuration and the list length decide: the same answer in every block) ...
  const bool props = !none && p.proximity && n_list <= PM_CARVE_BIG_SLOTS && p.max_size[ci] - 1u < PM_PROP_KMAX &&
                     !(p.debug_mem_above && n_list > p.debug_mem_above);
  // ... and do they walk the spatial index?  A walk visits the indexed positions around the seed whether they are
  // still candidates or not, the sweep visits the n_list candidates: the walk pays while the list is long and a good
  // part of the index is still in it.
  uint32_t cell_g = props ? st->cell_g : 0u;
  if (cell_g && p.prune_mode == 1u && (uint64_t)n_list * n_list < (uint64_t)p.prune_factor * st->n_indexed) cell_g = 0u;
  if (!none) {
    const uint64_t cbit = 1ull << p.avail_cfg[ci];
    // ---- this block's first slot: candidates of the blocks in front of it
    uint32_t part = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += 256u) part += G((const uint32_t*)p.prep_block_counts)[(size_t)b * PM_MAX_CONFIGS + ci];
    part = wave_sum(part);
    if (lane == 0) s_red[wave] = part;
    // ---- this wave's word: candidates, their ranks
    const uint32_t j = blockIdx.x * PREP_WAVES + wave;
    const uint32_t i = j * 64u + lane;
    const uint32_t ic = i < n ? i : (n ? n - 1u : 0u);
    const uint64_t aw = j < n_words ? G((const uint64_t*)p.alive_snap)[j] : 0ull;
    const uint64_t lg = j < n_words ? G(p.loc_g)[j] : 0ull;
    const uint64_t cm = n ? G((const uint64_t*)p.c_compat)[ic] : 0ull;
    const uint32_t ow = n ? G(p.order)[ic] : 0u;
    const uint32_t os = n ? G(p.c_site)[ic] : 0u;
    const double la = n ? G(p.c_lat)[ic] : 0.0, lo = n ? G(p.c_lon)[ic] : 0.0, co = n ? G(p.c_cos)[ic] : 0.0;
    const double vx = n ? G(p.c_ux)[ic] : 0.0, vy = n ? G(p.c_uy)[ic] : 0.0, vz = n ? G(p.c_uz)[ic] : 0.0;
    const bool c = i < n && ((aw >> lane) & 1ull) && (cm & cbit) != 0ull;
    const uint64_t bal = __ballot(c);
    const uint32_t cnt = (uint32_t)__popcll(bal);
    if (lane == 0) s_red[PREP_WAVES + wave] = cnt;
    __syncthreads();
#ifdef PM_BATCH_LOG
    if (tid == 0) atomicAdd((unsigned long long*)&p.status->prof[14], (unsigned long long)(__builtin_amdgcn_s_memtime() - pl_t0));  // loads in
#endif
    uint32_t off = 0;
#pragma unroll
    for (uint32_t w = 0; w < PREP_WAVES; ++w) off += s_red[w];
    for (uint32_t w = 0; w < wave; ++w) off += s_red[PREP_WAVES + w];
    const uint32_t rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    const uint32_t has_loc = (uint32_t)(lg >> lane) & 1u;
    if (c) {
      const uint32_t s = off + rank;
      G(p.slot_pos)[s] = i | (has_loc << 31);  // bit 31: has a location
      G(p.slot_wid)[s] = ow;
      G(p.cc_lat)[s] = la;
      G(p.cc_lon)[s] = lo;
      G(p.cc_cos)[s] = co;
      G(p.cc_ux)[s] = vx;
      G(p.cc_uy)[s] = vy;
      G(p.cc_uz)[s] = vz;
      G(p.cc_site)[s] = os;
      s_bits[wave][rank] = has_loc;
    }
    if (cell_g && i < n && has_loc) G(p.cs_slot)[G((const uint32_t*)p.cs_of_pos)[i]] = c ? off + rank : 0xFFFFFFFFu;
    // the located bits of this wave's slots [off, off + cnt): compacted by rank, ORed into the slot loc bitmap
    // (cleared by carve_prep_count_kernel); at most two words
    __syncthreads();
    const uint64_t locm = __ballot(lane < cnt && s_bits[wave][lane] != 0u);
    if (lane == 0 && locm) {
      const auto loc = (unsigned long long*)(p.bits_scratch + p.bits_stride);
      const uint32_t sh = off & 63u;
      atomicOr(&loc[off >> 6], (unsigned long long)(locm << sh));
      if (sh && (locm >> (64u - sh))) atomicOr(&loc[(off >> 6) + 1u], (unsigned long long)(locm >> (64u - sh)));
    }
  }
  // ---- the block that finishes last completes the list and publishes it
#ifdef PM_BATCH_LOG
  if (tid == 0) atomicAdd((unsigned long long*)&p.status->prof[15], (unsigned long long)(__builtin_amdgcn_s_memtime() - pl_t0));  // stores issued
#endif
  // What the last block reads of the others is the slot loc bitmap, and that is written with device-scope atomics
  // only: they need no release, just to have been performed before this block's ticket is taken — which a wait for
  // the wave's outstanding memory operations gives (workgroup-scope fence: s_waitcnt, no cache maintenance).  The
  // column stores are for the kernels behind this one; the end of the kernel releases them.  (A device-scope fence
  // here writes the XCD's L2 back once per BLOCK: 25 of the 42 us this kernel took at 100 k positions.)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
#ifdef PM_BATCH_LOG
  if (tid == 0) {  // this block, start to ticket: sum, max, count
    const uint64_t dt = __builtin_amdgcn_s_memtime() - pl_t0;
    atomicAdd((unsigned long long*)&p.status->prof[8], (unsigned long long)dt);
    atomicMax((unsigned long long*)&p.status->prof[9], (unsigned long long)dt);
    atomicAdd((unsigned long long*)&p.status->prof[10], 1ull);
  }
#endif
  if (tid == 0) s_red[0] = atomicAdd(&p.prep_counts[PM_MAX_CONFIGS], 1u);
  __syncthreads();
  if (s_red[0] != gridDim.x - 1u) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (see what the other blocks' atomics wrote: invalidate, nothing to write back)
#ifdef PM_BATCH_LOG
  const uint64_t pl_t1 = __builtin_amdgcn_s_memtime();  // every block is through: the tail begins
#endif
  if (none) {
    if (tid == 0) {
      p.desc->ci = p.n_avail;
      p.desc->none = 1u;
#ifdef PM_BATCH_LOG
      const uint32_t k = p.status->blog_n++;
      if (k < 512u) p.status->blog[3u * k] = p.status->blog[3u * k + 1u] = p.status->blog[3u * k + 2u] = 0u;
#endif
    }
    return;
  }
  const uint32_t lw = (n_list + 63u) >> 6;
  const auto alive = G(p.bits_scratch);
  for (uint32_t w = tid; w < lw; w += 256u)  // slot alive bitmap: every slot of the fresh list
    alive[w] = (w + 1u < lw || (n_list & 63u) == 0u) ? ~0ull : ((1ull << (n_list & 63u)) - 1ull);
  __syncthreads();  // (the scan below reads them back: same workgroup, the barrier's own fence is enough)
  uint32_t prop_k = 0, limit = 0, n_seeds = 0;
  const uint32_t max_s = p.max_size[ci];
  if (props) {
    const uint32_t k = max_s - 1u + PM_PROP_RESERVE;
    prop_k = k < PM_PROP_KMAX ? k : PM_PROP_KMAX;
    if (wave == 0) {
      prop_limit_scan(p, n_list, lane, &limit, &n_seeds, cell_g != 0u);
      if (lane == 0) {
        s_red[1] = limit;
        s_red[2] = n_seeds;
      }
    }
    __syncthreads();  // (uniform: prop_k depends on the configuration only)
    limit = s_red[1];
    n_seeds = s_red[2];
    prop_seed_slots(p, limit, tid, 256u);
  }
  if (tid == 0) {
    BatchDesc* d = p.desc;
    d->ci = ci;
    d->n_list = n_list;
    d->prop_k = prop_k;
    d->prop_limit = limit;
    d->rows_pr = (n_seeds + p.dist_world - 1u) / p.dist_world;
    d->n_seeds = n_seeds;
    d->cell_g = cell_g;
    d->valid = 1u;
    if (cell_g) p.status->pruned_batches += 1u;
#ifdef PM_BATCH_LOG
    {
      const uint64_t pl_t2 = __builtin_amdgcn_s_memtime();
      p.status->prof[11] += pl_t1 - p.status->prof[5];  // first block start -> last block through
      p.status->prof[12] += pl_t2 - pl_t1;              // the tail
      p.status->prof[13] += 1ull;
    }
    const uint32_t k = p.status->blog_n++;
    if (k < 512u) {
      p.status->blog[3u * k] = n_list;
      p.status->blog[3u * k + 1u] = n_seeds;
      p.status->blog[3u * k + 2u] = cell_g;
    }
#endif
  }
}

// ------------------------------------------------------------------------------------------------
// Full-chip construction of the ordered eligible list at the start of a proposal-driven FORM carve (the INIT launch
// of the validator did this on one CU: 0.5 ms at 100 k rows, and a tenth of an incremental tick that adds a few
// hundred workers to a standing swarm).  Same two-step shape as the list preparation above:
//   carve_elig_count_kernel   eligible rows (Healthy & p2p & unassigned, mod.rs:492-497) per block; clears loc_g
//   carve_elig_place_kernel   stable placement (position order = row order), the position-indexed columns, the
//                             loc bitmap; the block that finishes last writes the alive bitmap and the status
// One 64-row word per wave, four waves per block.

__device__ __forceinline__ bool row_eligible(const CarveArgs& p, uint32_t w) {
  if (w >= p.W) return false;
  const uint32_t f = G(p.wflags)[w];
  return (f & PM_W_HEALTHY) && (f & PM_W_HAS_P2P) && G(p.group_of)[w] < 0;
}

__global__ __launch_bounds__(256) void carve_elig_count_kernel(const CarveArgs* __restrict__ pa) {
  const CarveArgs& p = *pa;
  if (G(p.status)->state != CARVE_STATE_RUNNING) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  __shared__ uint32_t s_c[PREP_WAVES];
  const uint32_t j = blockIdx.x * PREP_WAVES + wave;
  const uint64_t bal = __ballot(row_eligible(p, j * 64u + lane));
  if (lane == 0) {
    s_c[wave] = (uint32_t)__popcll(bal);
    if (j < ((p.W + 63u) >> 6)) G(p.loc_g)[j] = 0ull;  // (the positions are a subset of the rows)
  }
  // (streaming carve: the per-configuration bitmaps of compatible positions, ORed in by the placement)
  if (p.stream && lane < p.n_avail && j < ((p.W + 63u) >> 6)) G(p.cfgbits)[(size_t)lane * p.bits_stride + j] = 0ull;
  if (blockIdx.x == 0 && tid <= PM_MAX_CONFIGS + 1u) p.prep_counts[tid] = 0u;  // totals and both tickets
  __syncthreads();
  if (tid == 0) {
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < PREP_WAVES; ++w) sum += s_c[w];
    G(p.prep_block_counts)[(size_t)blockIdx.x * PM_MAX_CONFIGS] = sum;
  }
}

__global__ __launch_bounds__(256) void carve_elig_place_kernel(const CarveArgs* __restrict__ pa, uint32_t start_ci) {
  const CarveArgs& p = *pa;
  const auto st = G(p.status);
  if (st->state != CARVE_STATE_RUNNING) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  __shared__ uint32_t s_red[2 * PREP_WAVES + 2];
  __shared__ uint32_t s_bits[PREP_WAVES][64];
  __shared__ uint64_t s_cm[PREP_WAVES][64];
  // ---- this block's first position: eligible rows of the blocks in front of it
  uint32_t part = 0;
  for (uint32_t b = tid; b < blockIdx.x; b += 256u) part += G((const uint32_t*)p.prep_block_counts)[(size_t)b * PM_MAX_CONFIGS];
  part = wave_sum(part);
  if (lane == 0) s_red[wave] = part;
  const uint32_t w = (blockIdx.x * PREP_WAVES + wave) * 64u + lane;
  const bool e = row_eligible(p, w);
  const uint64_t bal = __ballot(e);
  const uint32_t cnt = (uint32_t)__popcll(bal);
  if (lane == 0) s_red[PREP_WAVES + wave] = cnt;
  __syncthreads();
  uint32_t off = 0;
#pragma unroll
  for (uint32_t k = 0; k < PREP_WAVES; ++k) off += s_red[k];
  for (uint32_t k = 0; k < wave; ++k) off += s_red[PREP_WAVES + k];
  const uint32_t rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
  if (e) {
    const uint32_t i = off + rank;
    const uint32_t has_loc = (G(p.wflags)[w] & PM_W_HAS_LOC) ? 1u : 0u;
    G(p.order)[i] = w;
    G(p.c_lat)[i] = G(p.lat)[w];
    G(p.c_lon)[i] = G(p.lon)[w];
    G(p.c_cos)[i] = G(p.coslat)[w];
    G(p.c_ux)[i] = G(p.ux)[w];
    G(p.c_uy)[i] = G(p.uy)[w];
    G(p.c_uz)[i] = G(p.uz)[w];
    G(p.c_site)[i] = G(p.site)[w];
    const uint64_t cmw = G(p.compat)[w];
    G(p.c_compat)[i] = cmw;
    s_bits[wave][rank] = has_loc;
    s_cm[wave][rank] = cmw;
  }
  // the located bits of this wave's positions [off, off + cnt): compacted by rank, ORed into loc_g (at most two words)
  __syncthreads();
  const uint64_t locm = __ballot(lane < cnt && s_bits[wave][lane] != 0u);
  if (lane == 0 && locm) {
    const auto loc = (unsigned long long*)p.loc_g;
    const uint32_t sh = off & 63u;
    atomicOr(&loc[off >> 6], (unsigned long long)(locm << sh));
    if (sh && (locm >> (64u - sh))) atomicOr(&loc[(off >> 6) + 1u], (unsigned long long)(locm >> (64u - sh)));
  }
  if (p.stream) {  // the same for every configuration of the carve order: which of these positions are compatible
    const uint64_t cmr = lane < cnt ? s_cm[wave][lane] : 0ull;
    const uint32_t sh = off & 63u;
    for (uint32_t ci = 0; ci < p.n_avail; ++ci) {
      const uint64_t bits = __ballot((cmr >> p.avail_cfg[ci]) & 1ull);
      if (lane == 0 && bits) {
        const auto cb = (unsigned long long*)(p.cfgbits + (size_t)ci * p.bits_stride);
        atomicOr(&cb[off >> 6], (unsigned long long)(bits << sh));
        if (sh && (bits >> (64u - sh))) atomicOr(&cb[(off >> 6) + 1u], (unsigned long long)(bits >> (64u - sh)));
      }
    }
  }
  // ---- the block that finishes last completes the list and publishes it (it reads nothing the other blocks wrote
  // but the ticket: see carve_prep_place_kernel for why this is not a device-scope fence)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (tid == 0) s_red[0] = atomicAdd(&p.prep_counts[PM_MAX_CONFIGS + 1u], 1u);
  __syncthreads();
  if (s_red[0] != gridDim.x - 1u) return;
  __threadfence();
  uint32_t total = 0;
  for (uint32_t b = tid; b < gridDim.x; b += 256u) total += G((const uint32_t*)p.prep_block_counts)[(size_t)b * PM_MAX_CONFIGS];
  total = wave_sum(total);
  __syncthreads();
  if (lane == 0) s_red[1u + wave] = total;
  __syncthreads();
  const uint32_t n = s_red[1] + s_red[2] + s_red[3] + s_red[4];
  const uint32_t n_words = (n + 63u) >> 6;
  for (uint32_t j = tid; j < n_words; j += 256u) {
    const uint64_t all = (j + 1u < n_words || (n & 63u) == 0u) ? ~0ull : ((1ull << (n & 63u)) - 1ull);
    G(p.alive_g)[j] = all;
    if (p.stream) G(p.bits_scratch)[j] = all;  // (streaming carve: the published copy of "what no group holds yet")
  }
  if (tid <= PM_MAX_CONFIGS + 1u) p.prep_counts[tid] = 0u;  // totals + tickets of the first list preparation
  if (tid == 0) {
    st->n_eligible = n;
    st->total_available = n;  // mod.rs:503
    st->cur_ci = start_ci;
    st->need_prep = 1u;
    st->g_lo = st->g_hi = st->n_groups;
  }
}

// ------------------------------------------------------------------------------------------------
// Spatial index of the carve's located positions (see cell_walk).  Positions never move during a carve — only the
// candidate lists drawn from them do — so the index is built once, behind the eligible list: a counting sort by grid
// cell of the unit vectors.
//   cell_count_kernel   every located position: its cell, and its rank among the cell's members (one atomic)
//   cell_scan_*_kernel  exclusive scan of the G^3 counts -> cell_start; leaves the counts zero
//   cell_place_kernel   position -> entry; the entry's unit vector and site, in cell order
// Per batch, carve_prep_place_kernel then writes the slot every entry has in the prepared list (cs_slot, ~0 = none).
__global__ __launch_bounds__(256) void cell_count_kernel(const CarveArgs* __restrict__ pa) {
  const CarveArgs& p = *pa;
  const auto st = G(p.status);
  if (st->state != CARVE_STATE_RUNNING) return;
  const uint32_t n = st->n_eligible;
  const uint32_t g = (p.prune_mode && p.proximity) ? cell_g_for(p.prune_mode >= 2u && n >= 64u && n < PM_CELL_MIN_N ? PM_CELL_MIN_N : n) : 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->cell_g = g;
    st->n_indexed = 0u;
  }
  if (!g) return;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  if (!bit_at(G((const uint64_t*)p.loc_g), i)) {
    G(p.pos_cell)[i] = 0xFFFFFFFFu;
    return;
  }
  const uint32_t c = (cell_coord(G(p.c_uz)[i], g) * g + cell_coord(G(p.c_uy)[i], g)) * g + cell_coord(G(p.c_ux)[i], g);
  G(p.pos_cell)[i] = c;
  G(p.pos_rank)[i] = atomicAdd(&p.cell_cnt[c], 1u);
}

// the scan, on every CU: 1024 counts per block.  (A single workgroup took 230 us for the 262,144 cells of the 64^3 grid.)
//   cell_scan_sums_kernel   the sum of every block's counts; the block that finishes last turns the sums into offsets
//   cell_scan_apply_kernel  every block scans its own counts from its offset, and leaves the counts zero behind it
#define CELL_SCAN_PER_BLOCK 1024u
static_assert((PM_CELL_TABLE + CELL_SCAN_PER_BLOCK - 1u) / CELL_SCAN_PER_BLOCK <= 320u, "block sums fit one look of 256 + 64 threads");
__global__ __launch_bounds__(256) void cell_scan_sums_kernel(const CarveArgs* __restrict__ pa) {
  const CarveArgs& p = *pa;
  const auto st = G(p.status);
  if (st->state != CARVE_STATE_RUNNING) return;
  const uint32_t g = st->cell_g;
  if (!g) return;
  const uint32_t total = g * g * g, n_blocks = (total + CELL_SCAN_PER_BLOCK - 1u) / CELL_SCAN_PER_BLOCK;
  if (blockIdx.x >= n_blocks) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_last;
  const auto cnt = G((const uint32_t*)p.cell_cnt);
  const uint32_t i0 = blockIdx.x * CELL_SCAN_PER_BLOCK + tid * 4u;
  uint32_t v = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) v += i0 + k < total ? cnt[i0 + k] : 0u;
  v = wave_sum(v);
  if (lane == 0) s_w[wave] = v;
  __syncthreads();
  // the sums and the ticket live behind the starts (cell_start has PM_CELL_TABLE words, the table needs total + 1)
  const auto blk = p.cell_start + PM_CELL_TABLE;  // [320] sums -> offsets, [320] ticket
  if (tid == 0) {
    blk[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __threadfence();
    s_last = atomicAdd(&blk[320], 1u) == n_blocks - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // exclusive scan of up to 320 block sums by wave 0: five strides of 64
  if (wave == 0) {
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += 64u) {
      const uint32_t c = c0 + lane;
      const uint32_t x = c < n_blocks ? G((const uint32_t*)blk)[c] : 0u;
      const uint32_t incl = wave_incl_scan_u32(x);
      if (c < n_blocks) blk[c] = carry + incl - x;
      carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) {
      blk[320] = 0u;  // the ticket, for the next carve
      p.cell_start[total] = carry;
      st->n_indexed = carry;
    }
  }
}
__global__ __launch_bounds__(256) void cell_scan_apply_kernel(const CarveArgs* __restrict__ pa) {
  const CarveArgs& p = *pa;
  const auto st = G((const CarveStatus*)p.status);
  if (st->state != CARVE_STATE_RUNNING) return;
  const uint32_t g = st->cell_g;
  if (!g) return;
  const uint32_t total = g * g * g, n_blocks = (total + CELL_SCAN_PER_BLOCK - 1u) / CELL_SCAN_PER_BLOCK;
  if (blockIdx.x >= n_blocks) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  __shared__ uint32_t s_w[4];
  const auto cnt = G(p.cell_cnt);
  const auto start = G(p.cell_start);
  const uint32_t i0 = blockIdx.x * CELL_SCAN_PER_BLOCK + tid * 4u;
  uint32_t v[4];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) v[k] = i0 + k < total ? cnt[i0 + k] : 0u;
  const uint32_t mine = v[0] + v[1] + v[2] + v[3];
  const uint32_t incl = wave_incl_scan_u32(mine);
  if (lane == 63u) s_w[wave] = incl;
  __syncthreads();
  uint32_t run = G((const uint32_t*)(p.cell_start + PM_CELL_TABLE))[blockIdx.x] + incl - mine;
  for (uint32_t w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    if (i0 + k < total) {
      start[i0 + k] = run;
      cnt[i0 + k] = 0u;
    }
    run += v[k];
  }
}

__global__ __launch_bounds__(256) void cell_place_kernel(const CarveArgs* __restrict__ pa) {
  const CarveArgs& p = *pa;
  const auto st = G((const CarveStatus*)p.status);
  if (st->state != CARVE_STATE_RUNNING || !st->cell_g) return;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= st->n_eligible) return;
  const uint32_t c = G((const uint32_t*)p.pos_cell)[i];
  if (c == 0xFFFFFFFFu) {
    G(p.cs_of_pos)[i] = 0xFFFFFFFFu;
    return;
  }
  const uint32_t e = G((const uint32_t*)p.cell_start)[c] + G((const uint32_t*)p.pos_rank)[i];
  G(p.cs_of_pos)[i] = e;
  if (p.stream) G(p.cs_slot)[e] = i;  // (streaming carve: slot == position, for the whole carve)
  G(p.cs_ux)[e] = G((const double*)p.c_ux)[i];
  G(p.cs_uy)[e] = G((const double*)p.c_uy)[i];
  G(p.cs_uz)[e] = G((const double*)p.c_uz)[i];
  G(p.cs_site)[e] = G((const uint32_t*)p.c_site)[i];
}

__global__ __launch_bounds__(CARVE_THREADS) void carve_kernel(const CarveArgs* __restrict__ pa, uint32_t flags_in,
                                                              uint32_t start_ci) {
  const CarveArgs& p = *pa;  // argument block in device memory (a by-value struct this large would be
                             // copied to scratch as soon as a callee takes its address)
  // All LDS lives in the dynamic region, every carve offset a multiple of 16 B (a static __shared__ in
  // front of it would shift the base and put every 64-bit DS access on the 64-cycle misaligned path).
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  uint64_t* part = reinterpret_cast<uint64_t*>(s_raw);                       // [16 * PART]
  uint64_t* lds_alive = part + CARVE_WAVES * PM_CARVE_PART;                  // [SLOTS / 64]
  uint64_t* lds_loc = lds_alive + PM_CARVE_SLOTS / 64;                       // [SLOTS / 64]
  uint32_t* lds_wid = reinterpret_cast<uint32_t*>(lds_loc + PM_CARVE_SLOTS / 64);  // [SLOTS] big lists: the alive bitmap (small lists: unused)
  uint32_t* lds_site = lds_wid + PM_CARVE_SLOTS;                             // [SLOTS]
  uint64_t* lds_key = reinterpret_cast<uint64_t*>(lds_site + PM_CARVE_SLOTS);  // [SLOTS]
  uint32_t* sel_out = reinterpret_cast<uint32_t*>(lds_key + PM_CARVE_SLOTS);  // [SEL_CAP]
  BlockRed& red = *reinterpret_cast<BlockRed*>(sel_out + PM_CARVE_SEL_CAP);
  uint32_t& s_n = *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(&red) + sizeof(BlockRed));

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const auto st = G(p.status);
  uint32_t flags = flags_in;
  if (!(flags & CARVE_F_INIT) && st->state != CARVE_STATE_RUNNING) return;  // queued behind a finished carve
  // (one batch at a time: the totals + ticket of the preparation behind this launch; beside a validation in flight
  // the plan kernel clears them)
  if ((flags & CARVE_F_EXTPREP) && !p.speculative && threadIdx.x <= PM_MAX_CONFIGS) p.prep_counts[threadIdx.x] = 0u;
  PROF_DECL;

  uint32_t n;
  StepCtx c;
  c.mode = p.mode;
  c.proximity = p.proximity;
  c.use_props = (flags_in & CARVE_F_PROPS) != 0u;
  c.steps = 0;
  c.fast_steps = 0;
  c.cand_sum = 0;
  uint32_t ci;          // configuration being prepared / run
  bool prepared;
  if (flags & CARVE_F_INIT) {
    // ---- ordered eligible list.  FORM: compact the eligible rows (Healthy & p2p & unassigned,
    // mod.rs:492-497) in input order.  MERGE: supplied by the engine.
    if (p.mode == CARVE_MODE_FORM) {
      if (tid == 0) s_n = 0;
      __syncthreads();
      for (uint32_t base = 0; base < p.W; base += CARVE_THREADS) {
        const uint32_t w = base + tid;
        bool e = false;
        if (w < p.W) {
          const uint32_t f = G(p.wflags)[w];
          e = (f & PM_W_HEALTHY) && (f & PM_W_HAS_P2P) && G(p.group_of)[w] < 0;
        }
        const uint64_t bal = __ballot(e);
        if (lane == 0) red.a[wave] = __popcll(bal);
        __syncthreads();
        uint32_t off = s_n;
        for (uint32_t k = 0; k < wave; ++k) off += red.a[k];
        if (e) G(p.order)[off + __popcll(bal & ((1ull << lane) - 1ull))] = w;
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
        const uint32_t w = G(p.order)[i];
        has_loc = (G(p.wflags)[w] & PM_W_HAS_LOC) != 0;
        G(p.c_lat)[i] = G(p.lat)[w];
        G(p.c_lon)[i] = G(p.lon)[w];
        G(p.c_cos)[i] = G(p.coslat)[w];
        G(p.c_ux)[i] = G(p.ux)[w];
        G(p.c_uy)[i] = G(p.uy)[w];
        G(p.c_uz)[i] = G(p.uz)[w];
        G(p.c_site)[i] = G(p.site)[w];
        G(p.c_compat)[i] = G(p.compat)[w];
      }
      const uint64_t bl = __ballot(has_loc);
      const uint64_t ba = __ballot(i < n);
      if (lane == 0 && (i >> 6) < n_words) {
        G(p.loc_g)[i >> 6] = bl;
        G(p.alive_g)[i >> 6] = ba;
      }
    }
    __syncthreads();
    c.total_available = n;  // mod.rs:503
    ci = start_ci;
    prepared = false;
    if (flags & CARVE_F_EXTPREP) {  // the lists are prepared by carve_prep_*_kernel on the whole chip
      if (tid <= PM_MAX_CONFIGS) p.prep_counts[tid] = 0u;
      if (tid == 0) {
        st->n_eligible = n;
        st->total_available = n;
        st->cur_ci = start_ci;
        st->need_prep = 1u;
        st->g_lo = st->g_hi = st->n_groups;
      }
      return;
    }
  } else {
    n = st->n_eligible;
    c.total_available = st->total_available;
    ci = st->cur_ci;
    prepared = true;
    if (flags & CARVE_F_EXTPREP) {
      // the batch the preparation kernels describe: acceptable if it started from the configuration the carve is at
      const BatchDesc d = *p.desc;
      if (!d.planned || d.ci0 != ci || !(d.valid || d.none)) {
        if (tid == 0) {  // (prepared beside the batch in front, for the configuration it did not end at)
          st->n_void += 1u;
          st->why[4] += 1u;
        }
        return;
      }
      if (d.none) {  // no configuration left that can be entered
        if (tid == 0) {
          st->state = CARVE_STATE_DONE;
          st->cur_ci = p.n_avail;
        }
        return;
      }
      ci = d.ci;  // (the ones in between cannot be entered)
    }
    if (ci >= p.n_avail) return;
  }
  c.n_groups = st->n_groups;
  c.mem_off = st->n_members;
  const uint32_t groups_at_entry = c.n_groups;
  const uint32_t steps_before = st->steps_total;
  uint32_t exit_state = CARVE_STATE_RUNNING, stop_ci = p.n_avail;
  uint32_t slow_before_cfg = 0;

  PROF_MARK(15);  // init / status load
  for (;;) {
    // ---- prepare: candidate list of the next configuration whose loop would be entered
    // (mod.rs:505-519; the list is mod.rs:511-515 evaluated once, removals are applied to bitmaps)
    if (!prepared) {
      c.n_list = 0;
      c.prop_k = 0;
      c.prop_limit = 0;
      c.rows_pr = 0;
      c.n_seeds = 0;
      for (; ci < p.n_avail; ++ci) {
        c.min_s = p.min_size[ci];
        c.max_s = p.max_size[ci];
        if (p.mode == CARVE_MODE_FORM && c.total_available < c.min_s) continue;  // `while` never entered (:507)
        const uint64_t cbit = 1ull << p.avail_cfg[ci];
        PROF_MARK(29);  // loop overhead
        c.n_list = carve_compact_count(p, red, n, cbit, p.alive_g);
        PROF_MARK(26);
#ifdef PM_CARVE_PROF
        if (tid == 0) G(p.status)->prof[30] += 1;
#endif
        if (c.n_list < c.min_s || c.n_list == 0) continue;  // mod.rs:517-519
        carve_compact_place(p, red, n, cbit, c.n_list, p.alive_g);
        PROF_MARK(27);
        break;
      }
      if (ci >= p.n_avail) {
        exit_state = CARVE_STATE_DONE;
        break;
      }
      // proposals: one neighbour list per located slot, K = (max - 1) + reserve entries
      if ((flags & CARVE_F_PROPS) && p.proximity && c.n_list <= PM_CARVE_BIG_SLOTS && c.max_s - 1u < PM_PROP_KMAX) {
        const uint32_t k = c.max_s - 1u + PM_PROP_RESERVE;
        c.prop_k = k < PM_PROP_KMAX ? k : PM_PROP_KMAX;  // the last entry of a row carries its flags word
        // one proposal per located slot, at most PM_PROP_MAX_SEEDS per round: prop_limit = the slot after the
        // PM_PROP_MAX_SEEDS-th located one (a later round covers the rest)
        uint32_t n_seeds = 0;
        c.prop_limit = carve_prop_limit(p, red, c.n_list, &n_seeds);
        c.rows_pr = (n_seeds + p.dist_world - 1u) / p.dist_world;
        c.n_seeds = n_seeds;
      }
      prepared = true;
      PROF_MARK(28);
      if (!(flags & CARVE_F_RUN)) break;  // prepare-only launch
    } else if (flags & CARVE_F_EXTPREP) {
      const BatchDesc d = *p.desc;
      c.n_list = d.n_list;
      c.prop_k = d.prop_k;
      c.prop_limit = d.prop_limit;
      c.rows_pr = d.rows_pr;
      c.n_seeds = d.n_seeds;
      c.min_s = p.min_size[ci];
      c.max_s = p.max_size[ci];
    } else {
      c.n_list = st->n_list;
      c.prop_k = st->prop_k;
      c.prop_limit = st->prop_limit;
      c.rows_pr = st->rows_pr;
      c.n_seeds = st->n_seeds;
      c.min_s = p.min_size[ci];
      c.max_s = p.max_size[ci];
    }
    c.cfg = p.avail_cfg[ci];
    c.n_cand = c.n_list;
    c.n_start = c.n_list;

    // ---- run the prepared configuration.  Three storage modes:
    //   small (<= PM_CARVE_SLOTS slots):  every per-slot array in LDS
    //   big   (<= PM_CARVE_BIG_SLOTS):    bitmaps + staged proposal rows in LDS, per-slot arrays in HBM/L2
    //   mem   (larger):                   everything in HBM/L2, exact sweep only
    ctx_set_geometry(c);
    const bool force_mem = p.debug_mem_above && c.n_list > p.debug_mem_above;  // (test hook)
    const bool in_lds = c.n_list <= PM_CARVE_SLOTS && !force_mem;
    const bool big = !in_lds && c.n_list <= PM_CARVE_BIG_SLOTS && !force_mem;
    const uint32_t lw = (c.n_list + 63u) >> 6;
    uint64_t* g_alive = p.bits_scratch;
    uint64_t* g_loc = p.bits_scratch + p.bits_stride;
    // big mode: the worker-id / site-id regions (32 KiB each) hold the bitmaps instead
    uint64_t* r_alive = in_lds ? lds_alive : (big ? reinterpret_cast<uint64_t*>(lds_wid) : g_alive);
    uint64_t* r_loc = in_lds ? lds_loc : (big ? reinterpret_cast<uint64_t*>(lds_site) : g_loc);
    if (in_lds || big) {
      for (uint32_t j = tid; j < lw; j += CARVE_THREADS) {
        r_alive[j] = g_alive[j];
        r_loc[j] = g_loc[j];
      }
    }
    if (in_lds)
      for (uint32_t sl = tid; sl < c.n_list; sl += CARVE_THREADS) lds_site[sl] = G(p.cc_site)[sl];
    __syncthreads();
    if ((flags_in & CARVE_F_EXTPREP) && p.mode == CARVE_MODE_FORM && p.speculative) {
      // The list may have been prepared before the batch in front of it was validated: whatever has left the
      // position bitmap since then is a dead slot.  (No-op for a list prepared from the current state.)
      uint32_t live = 0;
      for (uint32_t base = 0; base < lw * 64u; base += CARVE_THREADS) {  // (a whole number of words per pass)
        const uint32_t sl = base + tid;
        bool al = false;
        if (sl < c.n_list) {
          const uint32_t i = G(p.slot_pos)[sl] & 0x7FFFFFFFu;
          al = (G(p.alive_g)[i >> 6] >> (i & 63u)) & 1ull;
        }
        const uint64_t bal = __ballot(al);
        if (lane == 0 && (sl >> 6) < lw) r_alive[sl >> 6] = bal;
        live += (uint32_t)__popcll(bal);
      }
      if (lane == 0) red.a[wave] = live;
      __syncthreads();
      uint32_t tot = 0;
#pragma unroll
      for (uint32_t k = 0; k < CARVE_WAVES; ++k) tot += red.a[k];
      c.n_cand = tot;
      c.n_start = tot;
      __syncthreads();
      // the loop of this configuration is entered only if ... (mod.rs:507, 517-519) — by the numbers as they are now
      if (c.total_available < c.min_s || c.n_cand < c.min_s || c.n_cand == 0u) {
        ++ci;
        c.n_list = 0;
        if (tid == 0) st->why[6] += 1u;
        break;
      }
      // too little of the list is left for its neighbour rows to be of use (it was prepared before the batch in front
      // of it took its share): leave it; the batch behind it was prepared from the state as it is now
      if (c.n_list > 256u && c.n_cand * PM_STALE_DIV < c.n_list) {
        if (tid == 0) {
          st->n_void += 1u;
          st->why[5] += 1u;
        }
        c.n_list = 0;
        break;
      }
    }
    PROF_MARK(10);
    const uint32_t slow0 = c.steps - c.fast_steps;
    const uint32_t mem_before_run = c.mem_off;
    int rc;
    if (in_lds) {
      rc = carve_run_lds<false>(p, red, c, p.slot_wid, lds_site, lds_key, r_alive, r_loc, part, sel_out,
                                reinterpret_cast<uint32_t*>(lds_key), steps_before);
    } else if (big) {
      rc = carve_run_lds<true>(p, red, c, p.slot_wid, p.cc_site, p.keys, r_alive, r_loc, part, sel_out,
                               reinterpret_cast<uint32_t*>(lds_key), steps_before);
    } else {
      do {
        rc = carve_step_mem(p, red, c, part, p.keys, p.slot_wid, g_alive, g_loc, steps_before);
      } while (rc == STEP_CONTINUE && !(c.n_cand * 2u < c.n_list));
    }
    (void)slow0;
    (void)slow_before_cfg;
    __syncthreads();
    PROF_MARK(13);
    if (in_lds || big) {
      // Everything this run took is a member it appended (recorded as a SLOT): clear its position in the eligible
      // bitmap, so the next compaction / configuration sees the removal, and replace the slot by the worker id.
      // (Proportional to the groups formed, not to the length of the list.)
      for (uint32_t k = mem_before_run + tid; k < c.mem_off; k += CARVE_THREADS) {
        const uint32_t sl = G(p.members)[k];
        const uint32_t i = G(p.slot_pos)[sl] & 0x7FFFFFFFu;  // bit 31: the slot has a location
        atomicAnd((unsigned long long*)&G(p.alive_g)[i >> 6], ~(1ull << (i & 63u)));
        G(p.members)[k] = G(p.slot_wid)[sl];
      }
    } else {  // lists in HBM: carve_step_mem records worker ids; dead slots -> position bitmap
      const uint64_t* alive = r_alive;
      for (uint32_t sl = tid; sl < c.n_list; sl += CARVE_THREADS)
        if (!bit_at(alive, sl)) {
          const uint32_t i = G(p.slot_pos)[sl] & 0x7FFFFFFFu;
          atomicAnd((unsigned long long*)&G(p.alive_g)[i >> 6], ~(1ull << (i & 63u)));
        }
    }
    __syncthreads();
    PROF_MARK(12);
    if (rc == STEP_UNCERTAIN || rc == STEP_OVERFLOW || rc == STEP_ABORT) {
      exit_state = rc == STEP_UNCERTAIN ? CARVE_STATE_UNCERTAIN : rc == STEP_ABORT ? CARVE_STATE_ABORTED : CARVE_STATE_OVERFLOW;
      stop_ci = ci;
      break;
    }
    prepared = false;
    if (rc == STEP_BREAK) {  // configuration exhausted; STEP_CONTINUE => re-prepare the same configuration
      ++ci;
      if (tid == 0) st->why[3] += 1u;
    }
    if (flags & CARVE_F_EXTPREP) break;  // the next list is prepared on the whole chip (carve_prep_*_kernel)
    if (!(flags & CARVE_F_ALL)) flags &= ~CARVE_F_RUN;  // per-configuration launch: prepare the next list, leave
  }

  // group_of for everything carved by this launch (FORM), one parallel pass at the end — or, with the external
  // preparation, left to carve_prep_count_kernel (every CU); single-node groups are counted here either way
  const bool ext = (flags_in & CARVE_F_EXTPREP) != 0u;
  // (a launch with the external preparation runs ONE configuration, c.min_s is its minimum: groups of one node can
  // only come from a configuration that allows them — otherwise there is nothing to do here, and a loop with a
  // dependent load per group is a tenth of the launch at 100 k workers)
  const bool need_pass = !ext || p.speculative || c.min_s <= 1u;
  if (p.mode == CARVE_MODE_FORM && need_pass) {
    __syncthreads();
    for (uint32_t g = groups_at_entry + wave; g < c.n_groups; g += CARVE_WAVES) {
      const uint32_t off = G(p.g_off)[g], gn = G(p.g_n)[g];
      if (!ext || p.speculative)  // (one batch at a time: the preparation behind this launch does it on every CU)
        for (uint32_t k = lane; k < gn; k += 64u) G(p.group_of)[G(p.members)[off + k]] = (int32_t)g;
      if (gn == 1u && lane == 0) atomicAdd(&p.status->n_solo, 1u);  // rare
    }
  }
  __syncthreads();
  PROF_MARK(14);
  if (tid == 0) {
    st->state = exit_state;
    st->n_groups = c.n_groups;
    st->n_members = c.mem_off;
    st->steps_total = steps_before + c.steps;
    st->stop_ci = stop_ci;
    st->n_eligible = n;
    st->cand_sum += c.cand_sum;
    st->cur_ci = exit_state == CARVE_STATE_DONE ? p.n_avail : ci;
    st->n_list = c.n_list;
    st->prop_k = c.prop_k;
    st->prop_limit = c.prop_limit;
    st->rows_pr = c.rows_pr;
    st->n_seeds = c.n_seeds;
    st->total_available = c.total_available;
    if (ext) {
      st->need_prep = exit_state == CARVE_STATE_RUNNING ? 1u : 0u;
      st->g_lo = groups_at_entry;
      st->g_hi = c.n_groups;
    }
    st->fast_steps += c.fast_steps;
    st->slow_steps += c.steps - c.fast_steps;
    if (!(flags_in & CARVE_F_INIT)) st->n_batches += 1;
  }
}

#include "pm_stream.inc"

// ------------------------------------------------------------------------------------------------
// launchers (called from pm_engine.cpp)

void launch_compat(const CompatArgs& a, hipStream_t s) {
  if (a.W == 0) return;
  hipLaunchKernelGGL(compat_kernel, dim3((a.W + 255u) / 256u), dim3(256), 0, s, a);
}
void launch_triad(const double* b, const double* c, double* a, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(triad_kernel, dim3(256 * 32), dim3(256), 0, s, b, c, a, n);
}
void launch_geo(const double* lat, const double* lon, double* coslat, double* ux, double* uy, double* uz, uint32_t W,
                hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(geo_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, lat, lon, coslat, ux, uy, uz, W);
}
void launch_update_rows(const RowUpdateArgs& a, hipStream_t s) {
  if (a.n == 0) return;
  hipLaunchKernelGGL(update_rows_kernel, dim3((a.n + 255u) / 256u), dim3(256), 0, s, a);
}
void launch_worker_selector(const int32_t* group_of, const uint32_t* g_cfg, uint32_t R, const uint32_t* rows,
                            uint64_t* sel, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL(worker_selector_kernel, dim3((R + 255u) / 256u), dim3(256), 0, s, group_of, g_cfg, R, rows, sel);
}
void launch_chooser_rank(const int32_t* group_of, const uint64_t* g_id, const uint32_t* count, uint32_t R,
                         const uint32_t* rows, uint64_t seed, uint32_t* rank, hipStream_t s) {
  if (R == 0) return;
  hipLaunchKernelGGL(chooser_rank_kernel, dim3((R + 255u) / 256u), dim3(256), 0, s, group_of, g_id, count, R, rows,
                     seed, rank);
}
void launch_eligible_selector(const uint32_t* wflags, const int32_t* group_of, const uint64_t* compat,
                              uint64_t enabled, uint32_t W, const uint8_t* shard, uint32_t my_rank, uint64_t* sel,
                              hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(eligible_selector_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, wflags, group_of, compat,
                     enabled, W, shard, my_rank, sel);
}
void launch_table_scatter(const pm_assignment* x, const uint32_t* xrow, uint32_t W, pm_assignment* table,
                          uint32_t* task_col, uint32_t* g_task_next, const uint64_t* t_live, const uint32_t* t_prefix,
                          hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(table_scatter_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, x, xrow, W, table, task_col,
                     g_task_next, t_live, t_prefix);
}
void launch_group_rank(const int32_t* group_of, const uint32_t* g_n, const uint32_t* g_off, const uint32_t* members,
                       const uint32_t* addr_rank, uint32_t W, uint32_t* rank_in_group, uint32_t* by_rank,
                       hipStream_t s) {
  if (W == 0) return;
  hipLaunchKernelGGL(group_rank_kernel, dim3((W + 255u) / 256u), dim3(256), 0, s, group_of, g_n, g_off, members,
                     addr_rank, W, rank_in_group, by_rank);
}
void launch_claim_publish(const ClaimArgs& a, hipStream_t s) {
  if (a.R == 0) return;
  hipLaunchKernelGGL(claim_publish_kernel, dim3((a.R + 255u) / 256u), dim3(256), 0, s, a);
}

// A swept axis is the column range [c_begin, c_end) of an index space of n_cols columns whose bit planes have a
// stride of `stride` words (c_begin need not be word-aligned: the bits below it are zero in every plane).
void launch_build_planes(const uint64_t* col_mask, uint32_t n_cols, uint32_t c_begin, uint32_t c_end, uint32_t stride,
                         uint32_t n_planes, uint64_t* planes, hipStream_t s) {
  const uint32_t w0 = c_begin / 64u, w1 = (c_end + 63u) / 64u;
  if (w1 <= w0) return;
  hipLaunchKernelGGL(build_planes_kernel, dim3(((w1 - w0) * 64u + 255u) / 256u), dim3(256), 0, s, col_mask, n_cols, w0,
                     w1, stride, n_planes, planes);
}

// Pair sweep: rows x cols -> first hit (absolute column index) + hit count per row.
void launch_pair_sweep(int variant, const uint64_t* row_sel, uint32_t R, const uint64_t* col_mask,
                       const uint64_t* planes, uint32_t c_begin, uint32_t c_end, uint32_t stride, uint32_t n_planes,
                       uint32_t* first, uint32_t* count, hipStream_t s) {
  if (R == 0) return;
  const uint32_t rb = (R + 255u) / 256u;
  // enough workgroups to cover 256 CUs several times over
  const uint32_t want_split = (2048u + rb - 1u) / rb;
  const uint32_t n_cols = c_end > c_begin ? c_end - c_begin : 0u;
  if (n_cols == 0) {
    hipLaunchKernelGGL(pair_init_kernel, dim3(rb), dim3(256), 0, s, first, count, R);
    return;
  }
  if (variant == 1) {
    uint32_t n_split = want_split < n_cols ? want_split : n_cols;
    const uint32_t chunk = (n_cols + n_split - 1u) / n_split;
    n_split = (n_cols + chunk - 1u) / chunk;
    if (n_split > 1u) hipLaunchKernelGGL(pair_init_kernel, dim3(rb), dim3(256), 0, s, first, count, R);
    hipLaunchKernelGGL(pair_sweep_scalar_kernel, dim3(rb, n_split), dim3(256), 0, s, row_sel, R, col_mask, c_begin,
                       c_end, chunk, first, count, n_split > 1u ? 1u : 0u);
    return;
  }
  const uint32_t w0 = c_begin / 64u, w1 = (c_end + 63u) / 64u, n_words = w1 - w0;
  uint32_t n_split = want_split < n_words ? want_split : n_words;
  const uint32_t wps = (n_words + n_split - 1u) / n_split;
  n_split = (n_words + wps - 1u) / wps;
  const uint32_t lds_cap_words = ((48u * 1024u / 8u) / (n_planes + 1u)) - 1u;  // n_planes + 1 planes, padded stride
  const uint32_t wpp = wps < lds_cap_words ? wps : lds_cap_words;
  const size_t lds = (size_t)(n_planes + 1u) * (wpp | 1u) * sizeof(uint64_t);
  if (n_split > 1u) hipLaunchKernelGGL(pair_init_kernel, dim3(rb), dim3(256), 0, s, first, count, R);
  if (rb >= 1024u)  // very many rows (a million tasks): four per thread, a quarter of the plane staging
    hipLaunchKernelGGL(pair_sweep_planes_kernel<4u>, dim3((rb + 3u) / 4u, n_split), dim3(256), lds, s, row_sel, R, planes,
                       stride, w0, w1, n_planes, wps, wpp, first, count, n_split > 1u ? 1u : 0u);
  else
    hipLaunchKernelGGL(pair_sweep_planes_kernel<1u>, dim3(rb, n_split), dim3(256), lds, s, row_sel, R, planes, stride, w0,
                       w1, n_planes, wps, wpp, first, count, n_split > 1u ? 1u : 0u);
}

void launch_pair_select(int variant, const uint64_t* row_sel, uint32_t R, const uint64_t* col_mask,
                        const uint64_t* planes, uint32_t c_begin, uint32_t c_end, uint32_t stride, uint32_t n_planes,
                        const uint32_t* rank, uint32_t* out, hipStream_t s) {
  if (R == 0) return;
  const uint32_t rb = (R + 255u) / 256u;
  if (variant == 1) {
    hipLaunchKernelGGL(pair_select_scalar_kernel, dim3(rb), dim3(256), 0, s, row_sel, R, col_mask, c_begin, c_end,
                       rank, out);
  } else {
    hipLaunchKernelGGL(pair_select_planes_kernel, dim3(rb), dim3(256), 0, s, row_sel, R, planes, stride,
                       c_begin / 64u, (c_end + 63u) / 64u, n_planes, rank, out);
  }
}

void launch_task_prefix(const uint64_t* live, uint32_t w_begin, uint32_t w_end, uint32_t* prefix, hipStream_t s) {
  if (w_end <= w_begin) return;
  hipLaunchKernelGGL(task_prefix_kernel, dim3(1), dim3(64), 0, s, live, w_begin, w_end, prefix);
}
void launch_task_delete(const uint32_t* slots, uint32_t n, uint64_t* tmask, uint64_t* live, uint64_t* planes,
                        uint32_t stride, uint32_t n_planes, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(task_delete_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, slots, n, tmask, live, planes,
                     stride, n_planes);
}
void launch_task_intern(const uint64_t* tmask, const uint64_t* live, uint32_t u_begin, uint32_t u_end, uint64_t valid,
                        uint64_t* keys, uint32_t n_slots, uint32_t* vals, uint32_t* counter_and_overflow, uint64_t* umask,
                        uint32_t cap_u, hipStream_t s) {
  (void)hipMemsetAsync(keys, 0xFF, size_t(n_slots) * 8, s);
  (void)hipMemsetAsync(counter_and_overflow, 0, 8, s);
  if (u_end > u_begin)
    hipLaunchKernelGGL(task_intern_insert_kernel, dim3((u_end - u_begin + 255u) / 256u), dim3(256), 0, s, tmask, live, u_begin,
                       u_end, valid, (unsigned long long*)keys, n_slots - 1u, counter_and_overflow + 1);
  hipLaunchKernelGGL(task_intern_number_kernel, dim3((n_slots + 255u) / 256u), dim3(256), 0, s,
                     (const unsigned long long*)keys, n_slots, vals, counter_and_overflow, umask, cap_u);
}
void launch_task_compact_class(const uint32_t* first_c, const uint32_t* count_c, const uint64_t* tmask, uint64_t valid,
                               const uint64_t* keys, const uint32_t* vals, uint32_t n_slots, uint32_t u_begin,
                               uint32_t u_end, const uint64_t* live, const uint32_t* prefix, uint32_t* first_out,
                               uint32_t* count_out, hipStream_t s) {
  if (u_end <= u_begin) return;
  hipLaunchKernelGGL(task_compact_class_kernel, dim3((u_end - u_begin + 255u) / 256u), dim3(256), 0, s, first_c, count_c,
                     tmask, valid, (const unsigned long long*)keys, vals, n_slots - 1u, u_begin, u_end, live, prefix,
                     first_out, count_out);
}
void launch_task_compact(const uint32_t* first_u, const uint32_t* count_u, uint32_t u_begin, uint32_t u_end,
                         const uint64_t* live, const uint32_t* prefix, uint32_t* first_out, uint32_t* count_out,
                         hipStream_t s) {
  if (u_end <= u_begin) return;
  hipLaunchKernelGGL(task_compact_kernel, dim3((u_end - u_begin + 255u) / 256u), dim3(256), 0, s, first_u, count_u,
                     u_begin, u_end, live, prefix, first_out, count_out);
}

void launch_newest(const int64_t* created_at, const uint64_t* live, uint32_t t_begin, uint32_t t_end,
                   uint32_t* idx_by_block, long long* val_by_block, uint32_t n_blocks, hipStream_t s) {
  hipLaunchKernelGGL(newest_kernel, dim3(n_blocks), dim3(256), 0, s, created_at, live, t_begin, t_end,
                     (unsigned long long*)nullptr, idx_by_block, val_by_block);
}

void launch_carve_propose(const CarveArgs* d_args, uint32_t W, hipStream_t s) {
  // one wave per located slot, grid-stride; 2048 workgroups x 4 waves keep all 256 CUs busy
  uint32_t blocks = (W + 3u) / 4u;
  if (blocks > 2048u) blocks = 2048u;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(carve_propose_kernel, dim3(blocks), dim3(256), 0, s, d_args);
}

// ids of freshly carved groups: outputs k+1 .. of the splitmix64 stream whose state is `state`
// (generate_group_id, injected — SURVEY section 8c), and an empty task word for each
__global__ __launch_bounds__(256) void group_ids_kernel(uint64_t* __restrict__ g_id, uint32_t* __restrict__ g_task,
                                                        uint32_t n, uint64_t state) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  uint64_t z = state + (uint64_t)(k + 1u) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  g_id[k] = z ^ (z >> 31);
  g_task[k] = PM_NONE;
}
void launch_group_ids(uint64_t* g_id, uint32_t* g_task, uint32_t n, uint64_t rng_state, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(group_ids_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, g_id, g_task, n, rng_state);
}

// the two full-chip kernels that prepare the next candidate list: one 64-position word per wave
uint32_t launch_carve_prep(const CarveArgs* d_args, uint32_t W, bool speculative, hipStream_t s) {  // returns the launches it made
  uint32_t blocks = ((W + 63u) / 64u + PREP_WAVES - 1u) / PREP_WAVES;
  if (blocks == 0) blocks = 1;
  // (beside a validation in flight the plan has to be decided once, ahead of the blocks that act on it; otherwise
  // every block of the count kernel derives the same plan from a status nobody is writing)
  if (speculative) hipLaunchKernelGGL(carve_plan_kernel, dim3(1), dim3(128), 0, s, d_args);
  hipLaunchKernelGGL(carve_prep_count_kernel, dim3(blocks), dim3(256), 0, s, d_args);
  hipLaunchKernelGGL(carve_prep_place_kernel, dim3(blocks), dim3(256), 0, s, d_args);
  return speculative ? 3u : 2u;
}
// n_bound: rows outside any group (an upper bound of the eligible list); index_min: build the spatial index when
// n_bound reaches it (0 = never; the kernels decide the grid from the real length and may still decline)
uint32_t launch_carve_elig(const CarveArgs* d_args, uint32_t W, uint32_t n_bound, uint32_t index_min, uint32_t start_ci, hipStream_t s) {
  uint32_t blocks = ((W + 63u) / 64u + PREP_WAVES - 1u) / PREP_WAVES;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(carve_elig_count_kernel, dim3(blocks), dim3(256), 0, s, d_args);
  hipLaunchKernelGGL(carve_elig_place_kernel, dim3(blocks), dim3(256), 0, s, d_args, start_ci);
  if (!index_min || n_bound < index_min) return 2u;  // (the status of a fresh carve says cell_g = 0)
  const uint32_t pb = (n_bound + 255u) / 256u;
  hipLaunchKernelGGL(cell_count_kernel, dim3(pb), dim3(256), 0, s, d_args);
  const uint32_t sb = (PM_CELL_TABLE + CELL_SCAN_PER_BLOCK - 1u) / CELL_SCAN_PER_BLOCK;  // (blocks beyond the grid in use return)
  hipLaunchKernelGGL(cell_scan_sums_kernel, dim3(sb), dim3(256), 0, s, d_args);
  hipLaunchKernelGGL(cell_scan_apply_kernel, dim3(sb), dim3(256), 0, s, d_args);
  hipLaunchKernelGGL(cell_place_kernel, dim3(pb), dim3(256), 0, s, d_args);
  return 6u;
}
// group_of for the groups of the last validation launch (the count kernel's first half), e.g. after the carve ended
void launch_carve_apply(const CarveArgs* d_args, uint32_t W, hipStream_t s) {
  uint32_t blocks = ((W + 63u) / 64u + PREP_WAVES - 1u) / PREP_WAVES;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(carve_prep_count_kernel, dim3(blocks), dim3(256), 0, s, d_args);
}

hipError_t launch_carve(const CarveArgs* d_args, uint32_t flags, uint32_t start_ci, size_t lds_bytes, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)carve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)PM_CARVE_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(carve_kernel, dim3(1), dim3(CARVE_THREADS), lds_bytes, s, d_args, flags, start_ci);
  return hipGetLastError();
}

}  // namespace pm
