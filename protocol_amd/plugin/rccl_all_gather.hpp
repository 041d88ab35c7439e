// rccl_all_gather.hpp — the communicators GpuMatchPlugin::tick_dist is handed (gpu_match_plugin.hpp, class AllGather):
//
//   RcclAllGather    one process per GPU, ncclAllGather (RCCL) over xGMI on the stream the engine's kernels run on — the
//                    production transport of the multi-GPU tick ("only for the cross-shard conflict-resolution
//                    all-gather").  The tick's ONE collective moves world x cap x 32 bytes (8 ranks x 100k workers: 3.2 MB
//                    landed per GPU), far below the per-link bandwidth: latency-bound, so it is a single call on the
//                    compute stream — no bucketing, no second stream to overlap with (nothing runs beside it: the next step
//                    is the scatter of what it brings).
//   LocalAllGather   the ranks live in ONE process (one thread each), their engines on the same GPU or on several of the
//                    node: device-to-device copies ordered by events, no host wait inside the exchange.  What the tests run
//                    on a one-GPU box, and a way to match one pool with several engines without RCCL.
//
// Built into libpm_plugin_dist.so (g++; links librccl and libamdhip64) — libpm_plugin.so itself needs neither.
#ifndef PM_RCCL_ALL_GATHER_HPP
#define PM_RCCL_ALL_GATHER_HPP

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gpu_match_plugin.hpp"

namespace orchestrator {

class CommError : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

class RcclAllGather : public AllGather {
 public:
  // Rendezvous of a one-process-per-GPU launch without a network service: rank 0 creates the ncclUniqueId and writes it
  // to `id_file` (written under a temporary name, then renamed: a reader sees all of it or nothing); the other ranks wait
  // for the file (timeout_s).  Then ncclCommInitRank on `device` and a non-blocking stream for the tick.
  RcclAllGather(uint32_t rank, uint32_t world, int32_t device, const std::string& id_file, double timeout_s = 120.0);
  // a communicator and a stream the host already owns (ncclComm_t, hipStream_t); nothing is destroyed with this object
  RcclAllGather(void* nccl_comm, void* hip_stream, uint32_t rank, uint32_t world);
  ~RcclAllGather() override;
  RcclAllGather(const RcclAllGather&) = delete;
  RcclAllGather& operator=(const RcclAllGather&) = delete;

  uint32_t rank() const override { return rank_; }
  uint32_t world() const override { return world_; }
  void* stream() const override { return stream_; }
  void all_gather(const void* send, void* recv, size_t bytes_per_rank) override;
  void synchronize() const;  // host wait for the stream (tests)

 private:
  uint32_t rank_, world_;
  int32_t device_ = -1;
  void* comm_ = nullptr;    // ncclComm_t
  void* stream_ = nullptr;  // hipStream_t
  bool owned_ = false;
};

// the shared state of n in-process ranks
class LocalWorld {
 public:
  explicit LocalWorld(uint32_t n);
  ~LocalWorld();
  uint32_t size() const { return n_; }

 private:
  friend class LocalAllGather;
  void barrier();
  const uint32_t n_;
  std::mutex mu_;
  std::condition_variable cv_;
  uint32_t waiting_ = 0, generation_ = 0;
  bool broken_ = false;                 // a rank failed inside an exchange: everybody leaves with an error
  std::vector<const void*> send_;
  std::vector<void*> ready_, done_;     // hipEvent_t per rank: "my segment is written" / "I have read everybody's"
};

class LocalAllGather : public AllGather {
 public:
  LocalAllGather(std::shared_ptr<LocalWorld> world, uint32_t rank, int32_t device);
  ~LocalAllGather() override;
  uint32_t rank() const override { return rank_; }
  uint32_t world() const override { return world_->size(); }
  void* stream() const override { return stream_; }
  void all_gather(const void* send, void* recv, size_t bytes_per_rank) override;
  void abandon();  // this rank will not reach its next exchange (its tick failed): release the others

 private:
  std::shared_ptr<LocalWorld> world_;
  uint32_t rank_;
  int32_t device_;
  void* stream_ = nullptr;
};

}  // namespace orchestrator
#endif
