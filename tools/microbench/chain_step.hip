// The plain step of the streaming carve's chain (pm_stream.inc, stream_chain) alone on a CU: ONE wave, a ring of 64 rows in LDS, every
// seed alive — what the 37 instructions cost when nothing else runs beside them, and what they cost with pieces taken out.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/microbench/libchain_step.so tools/microbench/chain_step.hip
//   python tools/microbench/chain_step.py          (on the GPU box)
// VARIANT (kernel argument): 0 the product's loop; 1 without the collector's word (ds_write_b64); 2 without the rows after next
// (no prefetch: ran / nrbn stay); 3 without the kill (ds_and); 4 the look alone (ds_read + wait + branch); bit 8: seven more waves of
// the workgroup poll an LDS word the way the parkers wait for room (s_sleep 1 between two looks).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define R 64u
typedef __attribute__((address_space(3))) uint32_t lds_u32;

template <int VARIANT>
__device__ __forceinline__ void run_batch(uint32_t rab0, uint32_t tail0, uint32_t n_steps, uint32_t group_n, uint32_t& commits) {
  const uint32_t lane_ = threadIdx.x & 63u;
  const uint32_t want = group_n - 1u;
  uint32_t lm = n_steps >= 32u ? 0xFFFFFFFFu : ((1u << n_steps) - 1u);
  const uint32_t guard = 1u << n_steps;
  uint32_t ra, nrb, adr, ra_n, nrb_n, adr_n, s_n, s, status;
  uint64_t a;
  s = (uint32_t)__builtin_ctz(lm);
  lm &= lm - 1u;
  adr = rab0 + (((tail0 + s) & (R - 1u)) << 9);
  s_n = (uint32_t)__builtin_ctz(lm | guard);
  adr_n = rab0 + (((tail0 + s_n) & (R - 1u)) << 9);
  ra = *(lds_u32*)(uintptr_t)adr;
  nrb = *(lds_u32*)(uintptr_t)(adr + 4u);
  ra_n = *(lds_u32*)(uintptr_t)adr_n;
  nrb_n = *(lds_u32*)(uintptr_t)(adr_n + 4u);
  uint32_t w, vt, st, su;
  uint64_t va, sx;
  if constexpr (VARIANT == 0) {
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfi_b32 %[vt], %[nrb], 0, %[w]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_bcnt1_i32_b64 %[st], vcc\n\t"
        "s_andn2_b32 %[su], 1, vcc_lo\n\t"
        "s_lshl_b32 %[su], %[su], 6\n\t"
        "s_add_u32 %[su], %[su], %[gn]\n\t"
        "s_cmp_lt_u32 %[st], %[su]\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"
        "v_cmp_ge_u32_e32 vcc, %[want], %[vt]\n\t"
        "s_and_saveexec_b64 %[sx], vcc\n\t"
        "ds_and_b32 %[ra], %[nrb]\n\t"
        "s_mov_b64 exec, %[sx]\n\t"
        "s_cmp_eq_u32 %[lm], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "ds_read_b32 %[w], %[ran]\n\t"
        "v_mov_b64 %[va], %[a]\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[s], %[sn]\n\t"
        "v_mov_b32 %[ra], %[ran]\n\t"
        "v_mov_b32 %[nrb], %[nrbn]\n\t"
        "v_mov_b32 %[adr], %[adrn]\n\t"
        "s_add_u32 %[st], %[lm], -1\n\t"
        "s_and_b32 %[lm], %[lm], %[st]\n\t"
        "s_or_b32 %[st], %[lm], %[guard]\n\t"
        "s_ff1_i32_b32 %[sn], %[st]\n\t"
        "s_add_u32 %[st], %[sn], %[tail0]\n\t"
        "s_and_b32 %[st], %[st], 63\n\t"
        "v_lshl_add_u32 %[adrn], %[st], 9, %[rab0]\n\t"
        "ds_read_b32 %[ran], %[adrn]\n\t"
        "ds_read_b32 %[nrbn], %[adrn] offset:4\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], %[a]\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "4:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx), [a] "=&s"(a),
          [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n), [nrbn] "+v"(nrb_n),
          [adrn] "+v"(adr_n), [lm] "+s"(lm), [s] "+s"(s), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [guard] "s"(guard), [tail0] "s"(tail0), [rab0] "v"(rab0)
        : "vcc", "scc", "memory");
  } else if constexpr (VARIANT == 1) {  // no word for the collector
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfi_b32 %[vt], %[nrb], 0, %[w]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_bcnt1_i32_b64 %[st], vcc\n\t"
        "s_andn2_b32 %[su], 1, vcc_lo\n\t"
        "s_lshl_b32 %[su], %[su], 6\n\t"
        "s_add_u32 %[su], %[su], %[gn]\n\t"
        "s_cmp_lt_u32 %[st], %[su]\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"
        "v_cmp_ge_u32_e32 vcc, %[want], %[vt]\n\t"
        "s_and_saveexec_b64 %[sx], vcc\n\t"
        "ds_and_b32 %[ra], %[nrb]\n\t"
        "s_mov_b64 exec, %[sx]\n\t"
        "s_cmp_eq_u32 %[lm], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "ds_read_b32 %[w], %[ran]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[s], %[sn]\n\t"
        "v_mov_b32 %[ra], %[ran]\n\t"
        "v_mov_b32 %[nrb], %[nrbn]\n\t"
        "v_mov_b32 %[adr], %[adrn]\n\t"
        "s_add_u32 %[st], %[lm], -1\n\t"
        "s_and_b32 %[lm], %[lm], %[st]\n\t"
        "s_or_b32 %[st], %[lm], %[guard]\n\t"
        "s_ff1_i32_b32 %[sn], %[st]\n\t"
        "s_add_u32 %[st], %[sn], %[tail0]\n\t"
        "s_and_b32 %[st], %[st], 63\n\t"
        "v_lshl_add_u32 %[adrn], %[st], 9, %[rab0]\n\t"
        "ds_read_b32 %[ran], %[adrn]\n\t"
        "ds_read_b32 %[nrbn], %[adrn] offset:4\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "4:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx), [a] "=&s"(a),
          [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n), [nrbn] "+v"(nrb_n),
          [adrn] "+v"(adr_n), [lm] "+s"(lm), [s] "+s"(s), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [guard] "s"(guard), [tail0] "s"(tail0), [rab0] "v"(rab0)
        : "vcc", "scc", "memory");
  } else if constexpr (VARIANT == 3) {  // no kill
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfi_b32 %[vt], %[nrb], 0, %[w]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_bcnt1_i32_b64 %[st], vcc\n\t"
        "s_andn2_b32 %[su], 1, vcc_lo\n\t"
        "s_lshl_b32 %[su], %[su], 6\n\t"
        "s_add_u32 %[su], %[su], %[gn]\n\t"
        "s_cmp_lt_u32 %[st], %[su]\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"
        "v_cmp_ge_u32_e32 vcc, %[want], %[vt]\n\t"
        "s_and_saveexec_b64 %[sx], vcc\n\t"
        "s_mov_b64 exec, %[sx]\n\t"
        "s_cmp_eq_u32 %[lm], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "ds_read_b32 %[w], %[ran]\n\t"
        "v_mov_b64 %[va], %[a]\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[s], %[sn]\n\t"
        "v_mov_b32 %[ra], %[ran]\n\t"
        "v_mov_b32 %[nrb], %[nrbn]\n\t"
        "v_mov_b32 %[adr], %[adrn]\n\t"
        "s_add_u32 %[st], %[lm], -1\n\t"
        "s_and_b32 %[lm], %[lm], %[st]\n\t"
        "s_or_b32 %[st], %[lm], %[guard]\n\t"
        "s_ff1_i32_b32 %[sn], %[st]\n\t"
        "s_add_u32 %[st], %[sn], %[tail0]\n\t"
        "s_and_b32 %[st], %[st], 63\n\t"
        "v_lshl_add_u32 %[adrn], %[st], 9, %[rab0]\n\t"
        "ds_read_b32 %[ran], %[adrn]\n\t"
        "ds_read_b32 %[nrbn], %[adrn] offset:4\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], %[a]\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "4:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx), [a] "=&s"(a),
          [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n), [nrbn] "+v"(nrb_n),
          [adrn] "+v"(adr_n), [lm] "+s"(lm), [s] "+s"(s), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [guard] "s"(guard), [tail0] "s"(tail0), [rab0] "v"(rab0)
        : "vcc", "scc", "memory");
  } else if constexpr (VARIANT == 5) {  // two steps a trip, no exec games, no register rotation
    // sets X = (ra, nrb, adr) and Y = (ra_n, nrb_n, adr_n); sc / sn: offsets of the current / the next live entry; lm: the live entries behind
    // the current one.  (Entries of a batch do not wrap around the ring here: rabT is the row of offset 0.)
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
#define STEP_BODY(XRA, XNRB, XADR, YRA, LBL_ATT, LBL_LAST)                                                             \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[lm], 0\n\t"                                                                                  \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "v_mov_b64 %[va], vcc\n\t"                                                                                   \
        "ds_write_b64 %[" XADR "], %[va]\n\t"                                                                        \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_bitset0_b32 %[lm], %[sn]\n\t"                                                                             \
        "s_ff1_i32_b32 %[sn], %[lm]\n\t"                                                                             \
        "v_lshl_add_u32 %[" XADR "], %[sn], 9, %[rabT]\n\t"                                                          \
        "ds_read_b32 %[" XRA "], %[" XADR "]\n\t"                                                                    \
        "ds_read_b32 %[" XNRB "], %[" XADR "] offset:4\n\t"
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "1:\n\t"
        STEP_BODY("ra", "nrb", "adr", "ran", "3f", "2f")
        STEP_BODY("ran", "nrbn", "adrn", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx), [a] "=&s"(a),
          [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n), [nrbn] "+v"(nrb_n),
          [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY
  } else if constexpr (VARIANT == 7) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(3)\n\t"                 /* the look is back; the three behind it stay on their way */      \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_mov_b64 %[va], vcc\n\t"                                                                                   \
        "ds_write_b64 %[" XADR "], %[va]\n\t"                                                                        \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY7
  } else if constexpr (VARIANT == 12) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(3)\n\t"                 /* the look is back; the three behind it stay on their way */      \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_mov_b64 %[va], vcc\n\t"                                                                                   \
        "ds_write_b64 %[" XADR "], %[va]\n\t"                                                                        \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY7
  } else if constexpr (VARIANT == 13) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(3)\n\t"                 /* the look is back; the three behind it stay on their way */      \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_mov_b64 %[va], vcc\n\t"                                                                                   \
        "ds_write_b64 %[" XADR "], %[va]\n\t"                                                                        \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY7
  } else if constexpr (VARIANT == 14) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(3)\n\t"                 /* the look is back; the three behind it stay on their way */      \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "s_nop 0\n\t"                                                                                                \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_mov_b64 %[va], vcc\n\t"                                                                                   \
        "ds_write_b64 %[" XADR "], %[va]\n\t"                                                                        \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY7
  } else if constexpr (VARIANT == 15) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_mov_b32 m0, %[sc]\n\t"                                                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(2)\n\t"                 /* the look is back; the two behind it stay on their way */          \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "s_nop 0\n\t"                                                                                                \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_writelane_b32 %[acl], vcc_lo, m0\n\t"                                                                  \
        "v_writelane_b32 %[ach], vcc_hi, m0\n\t"                                                                  \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2, acl = 0u, ach = 0u;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_writelane_b32 %[acl], vcc_lo, m0\n\t"
        "v_writelane_b32 %[ach], vcc_hi, m0\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_writelane_b32 %[acl], vcc_lo, m0\n\t"
        "v_writelane_b32 %[ach], vcc_hi, m0\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [acl] "+v"(acl), [ach] "+v"(ach), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "m0", "memory");
#undef STEP_BODY7
    if (lane_ < n_steps) *(__attribute__((address_space(3))) unsigned long long*)(uintptr_t)(rabT - lane_ * 8u + (lane_ << 9)) = ((unsigned long long)ach << 32) | acl;  // (lane j: entry j's word into its lane-0 slot)
  } else if constexpr (VARIANT == 16) {  // 15 with a row in ONE read (address and bit: a register pair by its numbers)
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2, adr2, acl = 0u, ach = 0u;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
#define STEP_BODY16(XRA, XNRB, XPAIR, YRA, LBL_ATT, LBL_LAST)                                                         \
        "s_mov_b32 m0, %[sc]\n\t"                                                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"                                                                             \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"                                                                            \
        "v_lshl_add_u32 %[adr2], %[sn2], 9, %[rabT]\n\t"                                                             \
        "s_waitcnt lgkmcnt(1)\n\t"                                                                                   \
        "v_bfi_b32 %[vt], " XNRB ", 0, %[w]\n\t"                                                                     \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "s_nop 0\n\t"                                                                                                \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "v_cndmask_b32_e64 %[vk], -1, " XNRB ", %[sx]\n\t"                                                           \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "ds_and_b32 " XRA ", %[vk]\n\t"                                                                              \
        "s_cmp_eq_u32 %[sn], -1\n\t"                                                                                 \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                                                                                   \
        "ds_read_b32 %[w], " YRA "\n\t"                                                                              \
        "ds_read_b64 " XPAIR ", %[adr2]\n\t"                                                                         \
        "v_writelane_b32 %[acl], vcc_lo, m0\n\t"                                                                     \
        "v_writelane_b32 %[ach], vcc_hi, m0\n\t"                                                                     \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    asm volatile(
        "ds_read_b32 %[w], v20\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY16("v20", "v21", "v[20:21]", "v22", "3f", "2f")
        STEP_BODY16("v22", "v23", "v[22:23]", "v20", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "6:\n\t"
        "v_writelane_b32 %[acl], vcc_lo, m0\n\t"
        "v_writelane_b32 %[ach], vcc_hi, m0\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [acl] "+v"(acl), [ach] "+v"(ach), "+{v20}"(ra), "+{v21}"(nrb),
          "+{v22}"(ra_n), "+{v23}"(nrb_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "m0", "memory");
#undef STEP_BODY16
    if (lane_ < n_steps) *(__attribute__((address_space(3))) unsigned long long*)(uintptr_t)(rabT - lane_ * 8u + (lane_ << 9)) = ((unsigned long long)ach << 32) | acl;
  } else if constexpr (VARIANT == 10) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(3)\n\t"                 /* the look is back; the three behind it stay on their way */      \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_mov_b64 %[va], vcc\n\t"                                                                                   \
        "ds_write_b64 %[" XADR "], %[va]\n\t"                                                                        \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY7
  } else if constexpr (VARIANT == 11) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(3)\n\t"                 /* the look is back; the three behind it stay on their way */      \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_cmp_ge_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cselect_b64 %[okm], -1, 0\n\t"                                                                            \
        "s_and_b64 %[sx], %[sx], %[okm]\n\t"                                                                         \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "s_cmp_eq_u64 %[okm], 0\n\t"                                                                                 \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_mov_b64 %[va], vcc\n\t"                                                                                   \
        "ds_write_b64 %[" XADR "], %[va]\n\t"                                                                        \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2; uint64_t okm;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [okm] "=&s"(okm), [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY7
  } else if constexpr (VARIANT == 8) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(2)\n\t"                 /* the look is back; the two behind it stay on their way */          \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "s_bcnt1_i32_b64 %[st], vcc\n\t"                                                                             \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "s_bfe_i32 %[su], vcc_lo, 0x10000\n\t"                                                                       \
        "s_and_b32 %[st], %[st], %[su]\n\t"                                                                          \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "s_cmp_lt_u32 %[st], %[gn]\n\t"                                                                              \
        "s_cbranch_scc1 " LBL_ATT "\n\t"                                                                             \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY7
  } else if constexpr (VARIANT == 9) {  // variant 5 with the row after next asked for right behind the look, and a wait for the look only
    const uint32_t rabT = rab0 + ((tail0 & (R - 1u)) << 9);
    uint32_t sc = s, vk, sn2;
    lm = n_steps >= 32u ? 0xFFFFFFFEu : (((1u << n_steps) - 1u) & ~1u);  // behind the current one (offset 0)
    s_n = lm ? (uint32_t)__builtin_ctz(lm) : 0xFFFFFFFFu;
    // per step, in the shadow of the look that is on its way: the offset and the address of the entry after next (they depend on the live
    // mask only); behind the kill: the next look, at once the row after next into the set the kill has freed, then the collector's word
#define STEP_BODY7(XRA, XNRB, XADR, XADR2, YRA, LBL_ATT, LBL_LAST)                                                    \
        "s_bitset0_b32 %[lm], %[sn]\n\t"           /* lm: the live entries behind the NEXT one */                     \
        "s_ff1_i32_b32 %[sn2], %[lm]\n\t"          /* the entry after next (-1: none) */                              \
        "v_lshl_add_u32 %[" XADR2 "], %[sn2], 9, %[rabT]\n\t"                                                        \
        "s_waitcnt lgkmcnt(3)\n\t"                 /* the look is back; the three behind it stay on their way */      \
        "v_bfi_b32 %[vt], %[" XNRB "], 0, %[w]\n\t"                                                                  \
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"                                                                         \
        "v_mbcnt_lo_u32_b32 %[vt], vcc_lo, 0\n\t"                                                                    \
        "v_mbcnt_hi_u32_b32 %[vt], vcc_hi, %[vt]\n\t"                                                                \
        "v_cmp_ge_u32_e64 %[sx], %[want], %[vt]\n\t"                                                                 \
        "v_cndmask_b32_e64 %[vk], -1, %[" XNRB "], %[sx]\n\t"                                                        \
        "ds_and_b32 %[" XRA "], %[vk]\n\t"                                                                           \
        "s_cmp_eq_u32 %[sn], -1\n\t"               /* no next entry: this was the batch's last live one */            \
        "s_cbranch_scc1 " LBL_LAST "\n\t"                                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                 /* (the next row's address: asked for a step ago) */               \
        "ds_read_b32 %[w], %[" YRA "]\n\t"                                                                           \
        "ds_read_b32 %[" XRA "], %[" XADR2 "]\n\t"                                                                   \
        "ds_read_b32 %[" XNRB "], %[" XADR2 "] offset:4\n\t"                                                         \
        "v_mov_b64 %[va], vcc\n\t"                                                                                   \
        "ds_write_b64 %[" XADR "], %[va]\n\t"                                                                        \
        "v_mov_b32 %[" XADR "], %[" XADR2 "]\n\t"                                                                    \
        "s_add_u32 %[cm], %[cm], 1\n\t"                                                                              \
        "s_mov_b32 %[sc], %[sn]\n\t"                                                                                 \
        "s_mov_b32 %[sn], %[sn2]\n\t"
    uint32_t adr2;
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_nop 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEP_BODY7("ra", "nrb", "adr", "adr2", "ran", "3f", "2f")
        STEP_BODY7("ran", "nrbn", "adrn", "adr2", "ra", "5f", "6f")
        "s_branch 1b\n\t"
        "2:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adr], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "6:\n\t"
        "v_mov_b64 %[va], vcc\n\t"
        "ds_write_b64 %[adrn], %[va]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_branch 4f\n\t"
        "3:\n\t"
        "s_mov_b32 %[status], 1\n\t"
        "s_branch 4f\n\t"
        "5:\n\t"
        "s_mov_b32 %[status], 3\n\t"
        "4:\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [vk] "=&v"(vk), [adr2] "=&v"(adr2), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx),
          [a] "=&s"(a), [sn2] "=&s"(sn2), [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n),
          [nrbn] "+v"(nrb_n), [adrn] "+v"(adr_n), [lm] "+s"(lm), [sc] "+s"(sc), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [rabT] "v"(rabT)
        : "vcc", "scc", "memory");
#undef STEP_BODY7
  } else {  // VARIANT 4: the look alone — read, wait, count, loop
    asm volatile(
        "ds_read_b32 %[w], %[ra]\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfi_b32 %[vt], %[nrb], 0, %[w]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[vt]\n\t"
        "s_mov_b64 %[a], vcc\n\t"
        "s_cmp_eq_u32 %[lm], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "ds_read_b32 %[w], %[ra]\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_add_u32 %[st], %[lm], -1\n\t"
        "s_and_b32 %[lm], %[lm], %[st]\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        "s_add_u32 %[cm], %[cm], 1\n\t"
        "s_mov_b32 %[status], 0\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [w] "=&v"(w), [vt] "=&v"(vt), [va] "=&v"(va), [st] "=&s"(st), [su] "=&s"(su), [sx] "=&s"(sx), [a] "=&s"(a),
          [status] "=&s"(status), [ra] "+v"(ra), [nrb] "+v"(nrb), [adr] "+v"(adr), [ran] "+v"(ra_n), [nrbn] "+v"(nrb_n),
          [adrn] "+v"(adr_n), [lm] "+s"(lm), [s] "+s"(s), [sn] "+s"(s_n), [cm] "+s"(commits)
        : [gn] "s"(group_n), [want] "s"(want), [guard] "s"(guard), [tail0] "s"(tail0), [rab0] "v"(rab0)
        : "vcc", "scc", "memory");
  }
  (void)status; (void)a;
}

// out[0] = ticks inside the batches, out[1] = commits, out[2] = batches
template <int VARIANT>
__global__ __launch_bounds__(512) void chain_step_kernel(unsigned long long* out, uint32_t n_batches, uint32_t batch_n, uint32_t group_n, uint32_t pollers, uint32_t thin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* RAB = reinterpret_cast<unsigned long long*>(smem);                    // [R][64] {address of the word, ~bit}
  uint32_t* A = reinterpret_cast<uint32_t*>(smem + size_t(R) * 64u * 8u);                    // [256] the bitmap: 8192 positions
  volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(A + 256u);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t a_base = (uint32_t)(uintptr_t)(lds_u32*)A;
  for (uint32_t i = tid; i < R * 64u; i += blockDim.x) {
    const uint32_t r = i >> 6, l = i & 63u;
    const uint32_t pos = ((r * 64u + l) * 37u) & 8191u;  // distinct positions, a row's neighbours in different words of the bitmap
    RAB[i] = ((unsigned long long)(~(1u << (pos & 31u))) << 32) | (a_base + ((pos >> 5) << 2));
  }
  if (tid == 0) *flag = 0u;
  __syncthreads();
  if (wave != 0u) {
    if (wave <= pollers) {  // the way a parker waits for room: a look at an LDS word, a nap, again
      uint32_t n = 0;
      while (*flag == 0u && n < (1u << 26)) { __builtin_amdgcn_s_sleep(1); ++n; }
    }
    return;
  }
  const uint32_t rab0 = (uint32_t)(uintptr_t)(lds_u32*)RAB + lane * 8u;
  unsigned long long ticks = 0, check = 0;
  uint32_t commits = 0u, tail = 0u;
  for (uint32_t b = 0; b < n_batches; ++b) {
    for (uint32_t j = lane; j < 256u; j += 64u) {   // everybody alive again — or, thinned: a pseudo-random five eighths of them
      uint32_t h = (j + 1u) * 2654435761u + b * 40503u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      uint32_t h2 = h * 3266489917u; h2 ^= h2 >> 16;
      A[j] = thin ? (h | (h2 & (h >> 7))) : 0xFFFFFFFFu;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (thin && lane < batch_n) {   // the seeds alive (lane 0 of every row of the batch)
      const uint32_t i0 = ((tail + lane) & (R - 1u)) * 64u, pos = (i0 * 37u) & 8191u;
      atomicOr(&A[pos >> 5], 1u << (pos & 31u));
    }
    for (uint32_t q = 0; q < batch_n; ++q) {  // (the collector's word overwrote the row's)
      const uint32_t i = ((tail + q) & (R - 1u)) * 64u + lane, pos = (i * 37u) & 8191u;
      RAB[i] = ((unsigned long long)(~(1u << (pos & 31u))) << 32) | (a_base + ((pos >> 5) << 2));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    run_batch<VARIANT>(rab0, tail, batch_n, group_n, commits);
    ticks += __builtin_amdgcn_s_memtime() - t0;
    for (uint32_t j = lane; j < 256u; j += 64u) check += (unsigned long long)A[j] * (j + 1u + b);  // what was killed, for comparing the variants
    for (uint32_t q = 0; q < batch_n; ++q) check += RAB[((tail + q) & (R - 1u)) * 64u] * (q + 3u);        // ... and the collector's words
    tail += batch_n;
  }
  if (lane == 0u) {
    *flag = 1u;
    out[0] = ticks;
    out[1] = commits;
    out[2] = n_batches;
  }
  atomicAdd(&out[3], check);
}

extern "C" int chain_step_run(int variant, uint32_t n_batches, uint32_t batch_n, uint32_t group_n, uint32_t pollers, uint32_t thin, unsigned long long* out3, double* ms) {
  unsigned long long* d = nullptr;
  if (hipMalloc(&d, 64) != hipSuccess) return 1;
  hipMemset(d, 0, 64);
  const size_t lds = size_t(R) * 64u * 8u + 1024u + 64u;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  switch (variant) {
    case 0: chain_step_kernel<0><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 1: chain_step_kernel<1><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 3: chain_step_kernel<3><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 8: chain_step_kernel<8><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 9: chain_step_kernel<9><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 10: chain_step_kernel<10><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 11: chain_step_kernel<11><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 12: chain_step_kernel<12><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 13: chain_step_kernel<13><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 14: chain_step_kernel<14><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 15: chain_step_kernel<15><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 16: chain_step_kernel<16><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 7: chain_step_kernel<7><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    case 5: chain_step_kernel<5><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
    default: chain_step_kernel<4><<<1, 512, lds, 0>>>(d, n_batches, batch_n, group_n, pollers, thin); break;
  }
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return 2;
  float f = 0;
  hipEventElapsedTime(&f, e0, e1);
  *ms = f;
  hipMemcpy(out3, d, 32, hipMemcpyDeviceToHost);
  hipFree(d);
  return 0;
}
