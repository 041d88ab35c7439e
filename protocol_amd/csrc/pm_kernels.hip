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
// the wave's sum: the scan's last lane, read into a scalar register (a __shfl_xor butterfly is three to six
// ds_bpermute round trips; every count at a configuration boundary of the streaming carve ends in one of these)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#ifdef PM_WAVE_SUM_SHFL
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
#endif
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

// One worker per lane and the configurations one after the other: the sweep of one worker x one configuration is a hundred
// dependent instructions, and a wave that has its SIMD to itself gets through one every five cycles or so — at 10,000 workers
// (160 waves on 1,024 SIMDs) the kernel is that chain, 14 us.  While the workers are few a workgroup takes 64 of them and its
// four waves a QUARTER of the configurations each, ORed through LDS (compat_sliced_kernel); from 65,536 workers on every
// SIMD has its waves and the plain form does the same work without the barrier.
template <bool SLICED>
__device__ __forceinline__ void compat_body(const CompatArgs& p) {
  __shared__ uint64_t s_part[SLICED ? 256 : 1];
  const uint32_t lane = threadIdx.x & 63u, slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform: the rows stay scalar loads)
  const uint32_t w = SLICED ? blockIdx.x * 64u + lane : blockIdx.x * 256u + threadIdx.x;
  const bool in = w < p.W;
  if (!SLICED && !in) return;
  const uint32_t per = SLICED ? (p.n_cfgs + 3u) / 4u : p.n_cfgs;
  const uint32_t c_begin = SLICED ? (slice * per < p.n_cfgs ? slice * per : p.n_cfgs) : 0u;
  const uint32_t c_end = SLICED ? (c_begin + per < p.n_cfgs ? c_begin + per : p.n_cfgs) : p.n_cfgs;
  const uint32_t wr = in ? w : 0u;
  const uint32_t wf = p.flags[wr];
  const uint32_t wcount = p.gpu_count[wr], wmem = p.gpu_mem[wr], wcls = p.gpu_cls[wr];
  const uint32_t wcores = p.cpu_cores[wr], wram = p.ram[wr], wsto = p.storage[wr];
  uint64_t mask = 0;
  for (uint32_t c = c_begin; c < c_end; ++c) {
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
  if (!SLICED) {
    p.compat[w] = mask;
    return;
  }
  s_part[threadIdx.x] = mask;
  __syncthreads();
  if (slice == 0u && in) p.compat[w] = s_part[lane] | s_part[64u + lane] | s_part[128u + lane] | s_part[192u + lane];
}
__global__ __launch_bounds__(256) void compat_kernel(CompatArgs p) { compat_body<false>(p); }
__global__ __launch_bounds__(256) void compat_sliced_kernel(CompatArgs p) { compat_body<true>(p); }

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
        // (a word costs its read and two popcounts: the first hit is where the count leaves zero — looked for once a row, not
        // asked about every word — and a row without a selector counts like the others and is zeroed behind the sweep)
        const uint64_t* pl = s_pl + my_plane[k] * lds_stride;
        uint32_t j = 0;
        for (; j + 4u <= nj; j += 4u) {
          const uint64_t h0 = pl[j], h1 = pl[j + 1u], h2 = pl[j + 2u], h3 = pl[j + 3u];
          const uint32_t before = cnt[k];
          cnt[k] += __popcll(h0) + __popcll(h1) + __popcll(h2) + __popcll(h3);
          if (before == 0u && cnt[k] != 0u) {
            const uint32_t u = h0 ? 0u : (h1 ? 1u : (h2 ? 2u : 3u));
            const uint64_t h = h0 ? h0 : (h1 ? h1 : (h2 ? h2 : h3));
            first[k] = (j0 + j + u) * 64u + __builtin_ctzll(h);
          }
        }
        for (; j < nj; ++j) {
          const uint64_t h = pl[j];
          if (h && cnt[k] == 0u) first[k] = (j0 + j) * 64u + __builtin_ctzll(h);
          cnt[k] += __popcll(h);
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
    if (!keep[k]) {  // (rows without a selector hit nothing)
      first[k] = PM_NONE;
      cnt[k] = 0u;
    }
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

// worker_selector_kernel + pair_init_kernel + group_rank_kernel + the copy of the groups' task words, for the match of one
// engine over all its workers (MatchPrepArgs)
__global__ __launch_bounds__(256) void match_prep_kernel(MatchPrepArgs p) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w < p.G) p.g_task_next[w] = p.g_task[w];
  if (w >= p.W) return;
  const int32_t g = p.group_of[w];
  p.first[w] = PM_NONE;
  p.count[w] = 0u;
  if (g < 0) {
    p.sel[w] = 0ull;
    p.rank_in_group[w] = 0;
    return;
  }
  p.sel[w] = 1ull << p.g_cfg[g];
  const uint32_t n = p.g_n[g], off = p.g_off[g], my = p.addr_rank[w];
  uint32_t idx = 0;
  for (uint32_t k = 0; k < n; ++k) idx += p.addr_rank[p.members[off + k]] < my;
  p.rank_in_group[w] = idx;
  p.by_rank[off + idx] = w;
}

__device__ __forceinline__ uint32_t task_position(const uint64_t* __restrict__ live, const uint32_t* __restrict__ prefix,
                                                  uint32_t u) {
  return prefix[u >> 6] + (uint32_t)__popcll(live[u >> 6] & ((1ull << (u & 63u)) - 1ull));
}

// Claim (SETNX, scheduler_impl.rs:74 / mod.rs:471-476) + publish row.  Every member of a group
// computed the same choice, so the group's task word is written with the same value by all.
__global__ __launch_bounds__(256) void claim_publish_kernel(ClaimArgs p) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  const bool in = r < p.R;
  if (!in && !p.h_table) return;  // (with the host's buffer to write, every thread of the workgroup goes to the barrier below)
  const uint32_t w = !in ? 0u : p.rows ? p.rows[r] : r;
  const int32_t g = in ? p.group_of[w] : -1;
  pm_assignment a;
  a.task = PM_NONE;
  a.group_slot = PM_NONE;
  a.group_index = 0;
  a.group_size = 0;
  a.next_worker = PM_NONE;
  a.group_id = 0;
  uint32_t t = PM_NONE;
  if (g >= 0) {
    t = p.g_task[g];  // get_current_group_task (scheduler_impl.rs:33)
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
    return;
  }
  if (in) {
    p.table[w] = a;
    p.task_col[w] = a.task;
  }
  if (p.h_table) {
    // The host's snapshot buffer and the groups' task words, straight from here.  The rows cross PCIe: a thread that stores
    // its own 32-byte row leaves every 64-byte line to be completed by a second instruction (19 GB/s measured); staged in
    // LDS and stored 16 contiguous bytes a lane, an instruction writes whole lines.
    static_assert(sizeof(pm_assignment) == 32, "a row is two 16-byte pieces");
    __shared__ uint4 s_rows[512];
    s_rows[2u * threadIdx.x] = make_uint4(a.task, a.group_slot, a.group_index, a.group_size);
    s_rows[2u * threadIdx.x + 1u] = make_uint4(a.next_worker, 0u, (uint32_t)a.group_id, (uint32_t)(a.group_id >> 32));
    if (g >= 0 && a.group_index == 0u) p.h_gtask[g] = t;  // (one member of every group has rank 0)
    __syncthreads();
    const uint32_t row0 = blockIdx.x * 256u;
    const uint32_t n16 = 2u * (p.R - row0 < 256u ? p.R - row0 : 256u);
    uint4* const dst = reinterpret_cast<uint4*>(p.h_table + row0);
    for (uint32_t k = threadIdx.x; k < n16; k += 256u) dst[k] = s_rows[k];
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

#include "pm_validate.inc"      // the validator's building blocks: LDS layout, the three-wave chain, exact steps
#include "pm_propose.inc"       // neighbour rows: NearRow, the walk over the spatial index, carve_propose_kernel
#include "pm_prep.inc"          // list preparation on the whole chip: candidate lists, the eligible list, the spatial index
#include "pm_carve_kernel.inc"  // carve_kernel: the batch pipeline's validator (and the merge pass)
#include "pm_stream.inc"        // the streaming carve (carve_variant 0): one launch per pass

#include "pm_launch.inc"        // the launchers pm_engine.cpp calls
}  // namespace pm
