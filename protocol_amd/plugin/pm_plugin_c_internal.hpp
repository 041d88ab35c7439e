// pm_plugin_c_internal.hpp — what a pmx_plugin handle holds (shared by pm_plugin_c.cpp and pm_plugin_dist_c.cpp; test driver's
// plumbing, see pm_plugin_c.h).
#ifndef PM_PLUGIN_C_INTERNAL_HPP
#define PM_PLUGIN_C_INTERNAL_HPP
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gpu_match_plugin.hpp"

namespace pmx_detail {
using namespace orchestrator;

struct RecordingWebhook : WebhookPlugin {
  std::mutex mu;
  std::string text;
  void line(const char* what, const std::string& id, const std::string& name, const std::vector<std::string>& nodes) {
    std::lock_guard<std::mutex> lk(mu);
    text += what;
    text += '\t' + id + '\t' + name;
    for (const std::string& n : nodes) text += '\t' + n;
    text += '\n';
  }
  void send_group_created(const std::string& id, const std::string& name, const std::vector<std::string>& nodes) override {
    line("created", id, name, nodes);
  }
  void send_group_destroyed(const std::string& id, const std::string& name, const std::vector<std::string>& nodes) override {
    line("destroyed", id, name, nodes);
  }
};

struct ListStore : TaskStore {
  std::mutex mu;
  std::vector<Task> tasks;
  uint32_t loads = 0;
  std::vector<Task> get_all_tasks() override {
    std::lock_guard<std::mutex> lk(mu);
    ++loads;
    return tasks;
  }
  std::vector<Task> snapshot() {
    std::lock_guard<std::mutex> lk(mu);
    return tasks;
  }
};

}  // namespace pmx_detail

struct pmx_plugin {
  std::atomic<int64_t> now_ms{0};
  std::shared_ptr<pmx_detail::RecordingWebhook> hook = std::make_shared<pmx_detail::RecordingWebhook>();
  std::shared_ptr<pmx_detail::ListStore> store = std::make_shared<pmx_detail::ListStore>();
  std::shared_ptr<orchestrator::GpuMatchPlugin> plugin;
  std::unique_ptr<orchestrator::Scheduler> scheduler;
  std::atomic<uint64_t> upload_count{0};
};

#endif
