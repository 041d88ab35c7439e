// pm_plugin_dist_c.cpp — see pm_plugin_dist_c.h.
#include "pm_plugin_dist_c.h"

#include <hip/hip_runtime_api.h>

#include <memory>
#include <string>
#include <vector>

#include "pm_plugin_c_internal.hpp"
#include "rccl_all_gather.hpp"

using namespace orchestrator;

struct pmx_comm {
  std::unique_ptr<AllGather> comm;
  LocalAllGather* local = nullptr;  // (comm.get() when the ranks share the process)
};

namespace {
thread_local std::string g_error;
}

extern "C" {

const char* pmx_last_error_dist(void) { return g_error.c_str(); }

int32_t pmx_rccl_create(uint32_t rank, uint32_t world, int32_t device, const char* id_file, pmx_comm** out) {
  try {
    std::unique_ptr<pmx_comm> c(new pmx_comm());
    c->comm.reset(new RcclAllGather(rank, world, device, id_file ? id_file : ""));
    *out = c.release();
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1;
  }
}

int32_t pmx_local_world_create(uint32_t n, int32_t device, pmx_comm** out) {
  try {
    auto world = std::make_shared<LocalWorld>(n);
    std::vector<std::unique_ptr<pmx_comm>> made;
    for (uint32_t r = 0; r < n; ++r) {
      std::unique_ptr<pmx_comm> c(new pmx_comm());
      c->local = new LocalAllGather(world, r, device);
      c->comm.reset(c->local);
      made.push_back(std::move(c));
    }
    for (uint32_t r = 0; r < n; ++r) out[r] = made[r].release();
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1;
  }
}

void pmx_comm_destroy(pmx_comm* c) { delete c; }

int32_t pmx_tick_dist(pmx_plugin* p, pmx_comm* c, pm_stats* stats) {
  try {
    const pm_stats s = p->plugin->tick_dist(*c->comm);
    if (stats) *stats = s;
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    if (c->local) c->local->abandon();  // (the other ranks of the process must not wait for this one)
    return -1;
  }
}

int32_t pmx_rccl_self_test(int32_t device, uint32_t bytes, const char* id_file) {
  try {
    RcclAllGather comm(0, 1, device, id_file ? id_file : "/tmp/pm_rccl_self_test.id");
    unsigned char* buf = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), bytes) != hipSuccess) throw CommError("hipMalloc");
    std::vector<unsigned char> pattern(bytes), back(bytes, 0);
    for (uint32_t i = 0; i < bytes; ++i) pattern[i] = static_cast<unsigned char>(i * 131u + 7u);
    hipStream_t s = static_cast<hipStream_t>(comm.stream());
    bool ok = hipMemcpyAsync(buf, pattern.data(), bytes, hipMemcpyHostToDevice, s) == hipSuccess;
    comm.all_gather(buf, buf, bytes);  // in place, on the stream the copy is on
    ok = ok && hipMemcpyAsync(back.data(), buf, bytes, hipMemcpyDeviceToHost, s) == hipSuccess;
    comm.synchronize();
    (void)hipFree(buf);
    if (!ok || back != pattern) throw CommError("the all-gather of a world of one did not leave the data in place");
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1;
  }
}

}  // extern "C"
