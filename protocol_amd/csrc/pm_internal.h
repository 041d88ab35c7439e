// pm_internal.h — declarations shared by pm_host.cpp and pm_engine.cpp (not part of the ABI).
#ifndef PM_INTERNAL_H
#define PM_INTERNAL_H

#include <stdint.h>

#include <string>
#include <string_view>
#include <vector>

#include "pm_engine.h"

namespace pm {

// Records a thread-local message for pm_last_error() and returns `code`.
int32_t set_error(int32_t code, const std::string& msg);

struct ParsedRequirements {
  uint32_t flags = 0;  // PM_R_CPU | PM_R_CPU_CORES | PM_R_RAM | PM_R_STORAGE
  uint32_t cpu_cores = 0, ram_mb = 0, storage_gb = 0;
  std::vector<pm_gpu_alt_row> alts;
  std::vector<std::string> models;  // parallel to alts (valid where PM_G_MODEL)
};

int32_t parse_requirements(std::string_view s, ParsedRequirements* out);
bool model_matches(std::string_view spec_model, std::string_view req_model);
std::string to_lowercase(std::string_view s);  // str::to_lowercase (Unicode; pm_host.cpp)
void template_order(const pm_config_row* cfgs, uint32_t n, std::vector<uint32_t>* order);
void available_order(const pm_config_row* cfgs, uint32_t n, uint64_t enabled, std::vector<uint32_t>* out);

inline uint64_t splitmix64_next(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t splitmix64_mix(uint64_t x) {
  uint64_t s = x;
  return splitmix64_next(&s);
}

}  // namespace pm

// test hooks and instrumentation: declared in include/pm_engine_debug.h (not part of the drop-in boundary)
#include "pm_engine_debug.h"
#endif
