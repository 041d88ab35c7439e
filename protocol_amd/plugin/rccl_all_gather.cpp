// rccl_all_gather.cpp — see rccl_all_gather.hpp.  Host code only (g++): the HIP runtime and RCCL through their C APIs.
#include "rccl_all_gather.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>

namespace orchestrator {

namespace {
void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) throw CommError(std::string(what) + ": " + hipGetErrorString(e));
}
void nccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw CommError(std::string(what) + ": " + ncclGetErrorString(r));
}
}  // namespace

RcclAllGather::RcclAllGather(uint32_t rank, uint32_t world, int32_t device, const std::string& id_file, double timeout_s)
    : rank_(rank), world_(world), device_(device), owned_(true) {
  if (world == 0 || rank >= world) throw CommError("RcclAllGather: rank outside the world");
  hip_ok(hipSetDevice(device), "hipSetDevice");
  ncclUniqueId id;
  if (rank == 0) {
    nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId");
    const std::string tmp = id_file + ".tmp";
    {
      std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
      f.write(reinterpret_cast<const char*>(&id), sizeof(id));
      if (!f) throw CommError("RcclAllGather: cannot write " + tmp);
    }
    if (std::rename(tmp.c_str(), id_file.c_str()) != 0) throw CommError("RcclAllGather: cannot publish " + id_file);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      std::ifstream f(id_file, std::ios::binary);
      if (f && f.read(reinterpret_cast<char*>(&id), sizeof(id)) && size_t(f.gcount()) == sizeof(id)) break;
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
        throw CommError("RcclAllGather: rank 0 never published " + id_file);
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  ncclComm_t comm = nullptr;
  nccl_ok(ncclCommInitRank(&comm, int(world), id, int(rank)), "ncclCommInitRank");
  comm_ = comm;
  hipStream_t s = nullptr;
  hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  stream_ = s;
}

RcclAllGather::RcclAllGather(void* nccl_comm, void* hip_stream, uint32_t rank, uint32_t world)
    : rank_(rank), world_(world), comm_(nccl_comm), stream_(hip_stream), owned_(false) {}

RcclAllGather::~RcclAllGather() {
  if (!owned_) return;
  if (device_ >= 0) (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(static_cast<hipStream_t>(stream_));
  if (comm_) (void)ncclCommDestroy(static_cast<ncclComm_t>(comm_));
  if (stream_) (void)hipStreamDestroy(static_cast<hipStream_t>(stream_));
}

void RcclAllGather::all_gather(const void* send, void* recv, size_t bytes_per_rank) {
  // (send is this rank's slot of recv: RCCL's in-place form, no staging copy)
  nccl_ok(ncclAllGather(send, recv, bytes_per_rank, ncclUint8, static_cast<ncclComm_t>(comm_), static_cast<hipStream_t>(stream_)),
          "ncclAllGather");
}

void RcclAllGather::synchronize() const { hip_ok(hipStreamSynchronize(static_cast<hipStream_t>(stream_)), "hipStreamSynchronize"); }

// ------------------------------------------------------------------------------------------------ in-process ranks

LocalWorld::LocalWorld(uint32_t n) : n_(n), send_(n, nullptr), ready_(n, nullptr), done_(n, nullptr) {
  if (n == 0) throw CommError("LocalWorld: no ranks");
}

LocalWorld::~LocalWorld() {
  for (void* e : ready_)
    if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  for (void* e : done_)
    if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
}

void LocalWorld::barrier() {
  std::unique_lock<std::mutex> lk(mu_);
  if (broken_) throw CommError("LocalAllGather: another rank left the exchange");
  const uint32_t gen = generation_;
  if (++waiting_ == n_) {
    waiting_ = 0;
    ++generation_;
    cv_.notify_all();
  } else {
    cv_.wait(lk, [&] { return generation_ != gen || broken_; });
    if (generation_ == gen) throw CommError("LocalAllGather: another rank left the exchange");
  }
}

LocalAllGather::LocalAllGather(std::shared_ptr<LocalWorld> world, uint32_t rank, int32_t device)
    : world_(std::move(world)), rank_(rank), device_(device) {
  if (rank >= world_->size()) throw CommError("LocalAllGather: rank outside the world");
  hip_ok(hipSetDevice(device), "hipSetDevice");
  hipStream_t s = nullptr;
  hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  stream_ = s;
  hipEvent_t a = nullptr, b = nullptr;
  hip_ok(hipEventCreateWithFlags(&a, hipEventDisableTiming), "hipEventCreateWithFlags");
  hip_ok(hipEventCreateWithFlags(&b, hipEventDisableTiming), "hipEventCreateWithFlags");
  std::lock_guard<std::mutex> lk(world_->mu_);
  world_->ready_[rank] = a;
  world_->done_[rank] = b;
}

LocalAllGather::~LocalAllGather() {
  (void)hipSetDevice(device_);
  if (stream_) {
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream_));
    (void)hipStreamDestroy(static_cast<hipStream_t>(stream_));
  }
}

void LocalAllGather::abandon() {
  std::lock_guard<std::mutex> lk(world_->mu_);
  world_->broken_ = true;
  world_->cv_.notify_all();
}

void LocalAllGather::all_gather(const void* send, void* recv, size_t bytes) {
  LocalWorld& w = *world_;
  const uint32_t n = w.size();
  if (n == 1) return;  // (send is recv's own slot)
  hipStream_t s = static_cast<hipStream_t>(stream_);
  try {
    hip_ok(hipSetDevice(device_), "hipSetDevice");
    // my segment is complete when everything queued on my stream so far has run
    hip_ok(hipEventRecord(static_cast<hipEvent_t>(w.ready_[rank_]), s), "hipEventRecord");
    {
      std::lock_guard<std::mutex> lk(w.mu_);
      w.send_[rank_] = send;
    }
    w.barrier();  // every rank has recorded its event and posted its pointer
    for (uint32_t k = 0; k < n; ++k) {
      if (k == rank_) continue;
      hip_ok(hipStreamWaitEvent(s, static_cast<hipEvent_t>(w.ready_[k]), 0), "hipStreamWaitEvent");
      hip_ok(hipMemcpyAsync(static_cast<char*>(recv) + size_t(k) * bytes, w.send_[k], bytes, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    }
    hip_ok(hipEventRecord(static_cast<hipEvent_t>(w.done_[rank_]), s), "hipEventRecord");
    w.barrier();  // every rank has queued its reads
    // nothing I queue from here on (the next tick rewrites my segment) may pass the others' reads of it
    for (uint32_t k = 0; k < n; ++k)
      if (k != rank_) hip_ok(hipStreamWaitEvent(s, static_cast<hipEvent_t>(w.done_[k]), 0), "hipStreamWaitEvent");
  } catch (...) {
    abandon();
    throw;
  }
}

}  // namespace orchestrator
