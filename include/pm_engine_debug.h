/* pm_engine_debug.h — test hooks and instrumentation of libpm_engine.so.
 *
 * NOT part of the drop-in boundary (include/pm_engine.h is; the Rust shim binds nothing from here).  These entry
 * points exist for tests/, tools/ and bench.py: they force code paths a test could not otherwise reach, or read
 * counters back.  They are exported by every build of the library so that the tests run against the shipped binary;
 * tests/test_host_helpers.py checks that each one declared here is exported.
 */
#ifndef PM_ENGINE_DEBUG_H
#define PM_ENGINE_DEBUG_H

#include "pm_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Counters of the last carve.  out[0..32) and out[56..88): s_memtime phase counters (all zero unless the library was
 * built with -DPM_CARVE_PROF: tools/stream_prof.py, tools/carve_prof.py); out[32..56): how the carve went — reasons
 * its launches ended, index geometry, streaming-carve tickets / timeouts / switches (protocol_amd/engine.py
 * debug_carve_counters names them).  Copies min(cap, 88) words (out[72..88): the row makers' anatomy, tools/stream_prof.py). */
int32_t pm_debug_carve_prof(pm_engine* e, unsigned long long* out, uint32_t cap);

/* Timeline of the last streaming carve launch (PM_CARVE_PROF builds; otherwise *n = 0): up to cap events of two words
 * {s_memtime, type | a << 8 | b << 32}; tools/stream_trace.py decodes them. */
int32_t pm_debug_stream_trace(pm_engine* e, unsigned long long* out, uint32_t cap, uint32_t* n);

/* Test hook: candidate lists longer than n slots take the all-in-HBM carve path (carve_step_mem), which otherwise
 * needs more than 262,144 candidates of one configuration; 0 = off. */
int32_t pm_debug_mem_lists_above(pm_engine* e, uint32_t n);

/* Test hook: the streaming carve (carve_variant 0) gives its launch up — CARVE_STATE_ABORTED, what a lost hand-shake
 * inside the validator workgroup ends in — as soon as n steps of the carve are committed (and the chain's budget runs
 * out there); the engine then continues the carve on the batch pipeline from the configuration it stopped in
 * (pm_stats / debug_carve_counters: stream_aborts).  0 = off. */
int32_t pm_debug_stream_abort_after(pm_engine* e, uint32_t n);

/* Counter: merge configurations (try_merge_groups_for_config, mod.rs:676-709) whose selections went through the streaming
 * carve (lists of PM_MERGE_STREAM_MIN = 512 compatible solo groups or more; the environment variable lowers it for tests)
 * since the engine was created. */
int32_t pm_debug_merge_streamed(pm_engine* e, uint32_t* n);

/* Counter: how often the device's copy of the group state was brought up to date by a DELTA — the workers of the groups
 * dissolved since and the rows appended since, a scatter and a fill — instead of compacting and uploading the whole list
 * (pm_engine.cpp push_groups: the path of a tick that follows status changes and new workers), since the engine was created.
 * pm_get_groups, pm_dissolve_group and the merge pass compact the list: the tick after them uploads it whole. */
int32_t pm_debug_delta_pushes(pm_engine* e, uint32_t* n);

/* Test hook / experiment: when the proposers walk the spatial index instead of sweeping the whole candidate list —
 * 0 never, 1 when it pays (default), 2 whenever the carve has an index, 3 = 2 with every seed forced through the
 * whole-list fallback.  (PM_PRUNE_MODE in the environment sets the default of new engines.) */
int32_t pm_debug_prune_mode(pm_engine* e, uint32_t mode);

/* Stream triad (a = b + 3c, f64) over 3 x n_doubles on the engine's stream, best of reps: the measured HBM rate
 * bench.py cites beside the roofline's 8 TB/s. */
int32_t pm_debug_hbm_triad(pm_engine* e, uint64_t n_doubles, uint32_t reps, double* gb_per_s);

/* A measuring build's (-DPM_ROW_REC) record of the last streaming launch's rows: eight words per ticket — the real-time
 * counter (100 MHz, one clock for all CUs; the validator's events of such a build carry the same) when the ticket was seen, when the first pass of the sweep was packed, when the first batch's keys were there, when the candidates
 * were through, when the row was finished, when it was stored; candidates evaluated | mode << 32; hardware id.  n_rows = 0 from
 * a product build. */
int32_t pm_debug_row_records(pm_engine* e, unsigned long long* out, uint32_t cap_rows, uint32_t* n_rows);
/* ... and the batches of steps its chain took up: four words each — the clock at the top of the chain's loop; first entry |
 * entries << 24 | live ones << 32 | commits so far << 40; the clock when the entries had been looked at; the clock behind the steps.
 * n = 0 from a product build. */
int32_t pm_debug_chain_batches(pm_engine* e, unsigned long long* out, uint32_t cap, uint32_t* n);
/* ... and the blocks of sixteen tickets its parkers took up, eight words each, indexed by first ticket / 16: first ticket | tickets
 * whose rows were asked for << 32 | tickets in the block << 48 | 1 << 63; clock at that moment; when the rows were all there; when the
 * block had its turn; when it had room in the ring; when it was parked; first entry | tickets parked << 32 | block of the run << 48. */
int32_t pm_debug_park_records(pm_engine* e, unsigned long long* out, uint32_t cap_blocks, uint32_t* n_blocks);

/* Neighbour rows two ways from the same keys (keys[n_waves * n_per_wave], ~0 = no candidate; the low slot_bits of a key index
 * sites[1 << slot_bits]; n_waves a multiple of four): the serial insertion, and four strides at a time through the sorting networks for the first `upto`
 * keys of a row (what the streaming carve's rows do).  Compared on the device: mismatches[0] = 1 a register differs | 2 a
 * threshold | 4 what a tracker answers, mismatches[1] = rows that differ; rows_out[n_waves * 64] = the networks' rows. */
int32_t pm_debug_row_networks(pm_engine* e, const uint64_t* keys, const uint32_t* sites, uint32_t n_waves, uint32_t n_per_wave,
                              uint32_t slot_bits, uint64_t ulps, uint32_t upto, uint64_t* rows_out, uint32_t* mismatches);

#ifdef __cplusplus
}
#endif
#endif
