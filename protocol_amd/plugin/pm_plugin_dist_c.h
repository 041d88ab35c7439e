/* pm_plugin_dist_c.h — the C face of GpuMatchPlugin::tick_dist and its communicators (rccl_all_gather.hpp) for the Python
 * test harness; libpm_plugin_dist.so.  Same conventions as pm_plugin_c.h (0 / -1, pmx_last_error_dist). */
#ifndef PM_PLUGIN_DIST_C_H
#define PM_PLUGIN_DIST_C_H

#include <stdint.h>

#include "pm_plugin_c.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmx_comm pmx_comm;

const char* pmx_last_error_dist(void);
/* one process per GPU over RCCL: rank 0 publishes the ncclUniqueId in id_file, the others wait for it */
int32_t pmx_rccl_create(uint32_t rank, uint32_t world, int32_t device, const char* id_file, pmx_comm** out);
/* n ranks of THIS process (one thread each), every one's engine on `device`: out[0..n) */
int32_t pmx_local_world_create(uint32_t n, int32_t device, pmx_comm** out);
void pmx_comm_destroy(pmx_comm*);
/* GpuMatchPlugin::tick_dist(comm): called by every rank (its own thread / process) at the same point */
int32_t pmx_tick_dist(pmx_plugin*, pmx_comm*, pm_stats* stats);
/* A world-of-one RCCL communicator on `device`: ncclAllGather through RcclAllGather::all_gather really runs on the
 * stream (bytes bytes of a pattern, in place) and the data is checked.  What a one-GPU box can verify of the RCCL binding:
 * the library loads, the communicator initialises, the collective is accepted on the tick's stream and completes. */
int32_t pmx_rccl_self_test(int32_t device, uint32_t bytes, const char* id_file);

#ifdef __cplusplus
}
#endif
#endif
