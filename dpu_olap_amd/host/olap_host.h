// olap_host.h -- everything the C++ host side shares: run-time configuration, the named
// timers the operators export, the RAII device context over the C ABI (include/olap_gpu.h)
// and the Arrow <-> pointer-table glue.
//
// Roles in the reference: host/system/system.h (env configuration), host/timer/timer.h
// (timer::Timers, the return type of every operator's Timers()), host/dpuext/dpuext.hpp
// (dpu::DpuSet), host/dpuext/arrow_utils.{h,cc} (Arrow buffers <-> device) and
// host/dpuext/status.h (error -> arrow::Status).
#pragma once

#include <arrow/api.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "olap_gpu.h"

// ---------------------------------------------------------------- configuration
namespace variables {

inline int from_env(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : fallback;
}
// NR_GPUS plays the part of the reference's NR_DPUS; SF is the scale factor of the BM_*
// argument shapes; MAX_THREADS caps the Arrow CPU pool of the *Native operators.
inline int max_gpus() { return from_env("NR_GPUS", 1); }
inline int scale_factor() { return from_env("SF", 4); }
inline int max_threads() { return from_env("MAX_THREADS", (int)std::thread::hardware_concurrency()); }
inline int max_threads_no_ht() { return from_env("MAX_THREADS", std::max(1, (int)std::thread::hardware_concurrency() / 2)); }

}  // namespace variables

// ------------------------------------------------------------------------ timers
namespace timer {

// One accumulating stopwatch per rank (= GPU).  Device phases come from the library's
// HIP-event timers and are added with Add().
class Timer {
  using clock = std::chrono::steady_clock;

 public:
  explicit Timer(uint32_t ranks) : acc_(std::max(1u, ranks)), t0_(std::max(1u, ranks)) {}
  void Start(uint32_t rank = 0) { t0_.at(rank) = clock::now(); }
  void Stop(uint32_t rank = 0) { acc_.at(rank) += std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - t0_.at(rank)); }
  void Add(std::chrono::nanoseconds d, uint32_t rank = 0) { acc_.at(rank) += d; }
  std::chrono::nanoseconds Result() const {
    std::chrono::nanoseconds sum{0};
    for (const auto& d : acc_) sum += d;
    return sum;
  }

 private:
  std::vector<std::chrono::nanoseconds> acc_;
  std::vector<clock::time_point> t0_;
};

class Timers {
 public:
  explicit Timers(uint32_t ranks) : ranks_(ranks) {}
  std::shared_ptr<Timer> New(const std::string& name) { return by_name_[name] = std::make_shared<Timer>(ranks_); }
  std::shared_ptr<Timer> Get(const std::string& name) {
    auto it = by_name_.find(name);
    return it != by_name_.end() ? it->second : New(name);
  }
  const std::map<std::string, std::shared_ptr<Timer>>& get() const { return by_name_; }

 private:
  uint32_t ranks_;
  std::map<std::string, std::shared_ptr<Timer>> by_name_;
};

}  // namespace timer

// ------------------------------------------------------------------------ device
namespace olap {

// The device set an operator runs on: the role of dpu::DpuSet (host/dpuext/dpuext.hpp:687-715).  One rank = one
// GPU (olap_group over the C ABI); filter / take / sum deal their batches to the ranks, the join spans them all.
class Device {
 public:
  // `n_ranks` ranks on HIP devices 0 .. n_ranks - 1.  A host with fewer devices can run them as logical ranks of
  // device 0 (OLAP_GROUP_LOOPBACK=1: every step of the multi-device path on one GPU); without that it is an error --
  // a result labelled GPUs:N must have used N.
  static arrow::Result<std::shared_ptr<Device>> Allocate(int n_ranks = 1) {
    if (n_ranks < 1 || (n_ranks & (n_ranks - 1))) return arrow::Status::Invalid("the number of GPUs must be a power of two, got ", n_ranks);
    std::vector<int> ids(n_ranks);
    for (int r = 0; r < n_ranks; ++r) ids[r] = r;
    olap_group* group = nullptr;
    int rc = olap_group_create(n_ranks, ids.data(), &group);
    bool loopback = false;
    if (rc != OLAP_OK && n_ranks > 1 && variables::from_env("OLAP_GROUP_LOOPBACK", 0)) {
      std::fill(ids.begin(), ids.end(), 0);
      rc = olap_group_create(n_ranks, ids.data(), &group);
      loopback = true;
    }
    if (rc != OLAP_OK)
      return arrow::Status::UnknownError("olap_group_create(", n_ranks, " devices) failed with code ", rc,
                                         n_ranks > 1 ? " (fewer HIP devices than NR_GPUS? OLAP_GROUP_LOOPBACK=1 runs the ranks on device 0)"
                                                     : " (no usable HIP device?)");
    return std::shared_ptr<Device>(new Device(group, loopback));
  }
  ~Device() { olap_group_destroy(group_); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;

  olap_group* group() const { return group_; }
  olap_ctx* ctx(uint32_t rank = 0) const { return olap_group_ctx(group_, (int)rank); }
  uint32_t nr_ranks() const { return (uint32_t)olap_group_size(group_); }
  bool loopback() const { return loopback_; }

  arrow::Status Check(int rc, uint32_t rank = 0) const {
    return rc == OLAP_OK ? arrow::Status::OK() : arrow::Status::UnknownError("olap error ", rc, ": ", olap_last_error(ctx(rank)));
  }
  arrow::Status CheckGroup(int rc) const {
    return rc == OLAP_OK ? arrow::Status::OK() : arrow::Status::UnknownError("olap error ", rc, ": ", olap_group_last_error(group_));
  }

  // fn(rank) on every rank, one host thread per rank (a context is single-caller; distinct contexts are independent)
  template <class Fn>
  arrow::Status OnRanks(Fn&& fn) const {
    const uint32_t n = nr_ranks();
    if (n == 1) return fn(0u);
    std::vector<arrow::Status> st(n);
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < n; ++r) th.emplace_back([&, r] { st[r] = fn(r); });
    for (auto& t : th) t.join();
    for (auto& s : st) ARROW_RETURN_NOT_OK(s);
    return arrow::Status::OK();
  }

  // Prepare() of every operator: HIP-event timers on, counters cleared.
  arrow::Status BeginTimers() const {
    for (uint32_t r = 0; r < nr_ranks(); ++r) {
      ARROW_RETURN_NOT_OK(Check(olap_set_profiling(ctx(r), 1), r));
      ARROW_RETURN_NOT_OK(Check(olap_reset_timers(ctx(r)), r));
    }
    return arrow::Status::OK();
  }

  // Library phases -> timers (per rank, like the reference's per-rank timers): one entry per phase, plus
  // "gpu-work" = everything but the copies (the reference's dpu-work next to copy-to-dpu / copy-from-dpu).
  arrow::Status FoldTimers(timer::Timers* timers) const {
    for (uint32_t r = 0; r < nr_ranks(); ++r) {
      const char* names[32];
      double ms[32];
      uint64_t launches[32];
      int n = 0;
      ARROW_RETURN_NOT_OK(Check(olap_get_timers(ctx(r), names, ms, launches, 32, &n), r));
      double work_ms = 0;
      for (int i = 0; i < std::min(n, 32); ++i) {
        const std::string name = names[i];
        timers->Get(name)->Add(std::chrono::nanoseconds((int64_t)(ms[i] * 1e6)), r);
        if (name != "copy-to-gpu" && name != "copy-from-gpu") work_ms += ms[i];
      }
      timers->Get("gpu-work")->Add(std::chrono::nanoseconds((int64_t)(work_ms * 1e6)), r);
    }
    return arrow::Status::OK();
  }

 private:
  Device(olap_group* group, bool loopback) : group_(group), loopback_(loopback) {}
  olap_group* group_;
  bool loopback_;
};

// (pointer, rows) tables of one column across batches -- the ABI's view of a
// RecordBatchVector.  Data buffer only (buffers[1]), slice offsets honoured, no nulls.
struct ColumnTable {
  std::vector<const void*> ptrs;
  std::vector<uint64_t> rows;
  uint64_t total_rows() const {
    uint64_t n = 0;
    for (auto r : rows) n += r;
    return n;
  }
};

inline arrow::Result<ColumnTable> MakeColumnTable(const arrow::RecordBatchVector& batches, int column) {
  ColumnTable t;
  for (const auto& batch : batches) {
    if (column < 0 || column >= batch->num_columns()) return arrow::Status::Invalid("column index out of range");
    const auto& d = batch->column_data(column);
    const auto* fixed = dynamic_cast<const arrow::FixedWidthType*>(d->type.get());
    if (!fixed || d->buffers.size() < 2) return arrow::Status::TypeError("fixed-width column expected");
    if (d->GetNullCount() != 0) return arrow::Status::NotImplemented("nullable columns are outside the hot path");
    const int64_t width = fixed->bit_width() / 8;
    const uint8_t* base = d->buffers[1] ? d->buffers[1]->data() : nullptr;
    t.ptrs.push_back(base ? base + d->offset * width : nullptr);
    t.rows.push_back((uint64_t)d->length);
  }
  return t;
}

inline arrow::Result<int> DtypeOf(const arrow::DataType& type) {
  switch (type.id()) {
    case arrow::Type::UINT32: return (int)OLAP_U32;
    case arrow::Type::INT32: return (int)OLAP_I32;
    case arrow::Type::UINT64: return (int)OLAP_U64;
    case arrow::Type::INT64: return (int)OLAP_I64;
    default: return arrow::Status::TypeError("the GPU hot path covers uint32/int32/uint64/int64, got ", type.ToString());
  }
}

inline int WidthOf(int dtype) { return (dtype == OLAP_U32 || dtype == OLAP_I32) ? 4 : 8; }

// A filled host buffer as an arrow::Array (no validity bitmap).
inline std::shared_ptr<arrow::Array> WrapBuffer(const std::shared_ptr<arrow::DataType>& type, int64_t length,
                                                std::shared_ptr<arrow::Buffer> buffer) {
  return arrow::MakeArray(arrow::ArrayData::Make(type, length, {nullptr, std::move(buffer)}));
}

// n output buffers sized for rows[b] elements each, plus the raw pointer table.
struct OutputBuffers {
  std::vector<std::shared_ptr<arrow::Buffer>> buffers;
  std::vector<void*> ptrs;
};

// Room for `rows[b]` elements per batch, carved out of ONE allocation (64-byte aligned slices that keep
// the parent alive): thousands of 256 KiB allocations cost more than the GPU work, and pages a filter
// never writes to are never touched.
inline arrow::Result<OutputBuffers> AllocateOutputs(const std::vector<uint64_t>& rows, int width) {
  OutputBuffers out;
  std::vector<int64_t> offset(rows.size() + 1, 0);
  for (size_t b = 0; b < rows.size(); ++b) offset[b + 1] = offset[b] + (((int64_t)std::max<uint64_t>(rows[b], 1) * width + 63) & ~int64_t(63));
  ARROW_ASSIGN_OR_RAISE(std::shared_ptr<arrow::Buffer> parent, arrow::AllocateBuffer(std::max<int64_t>(offset.back(), 64)));
  for (size_t b = 0; b < rows.size(); ++b) {
    auto slice = arrow::SliceMutableBuffer(parent, offset[b], (int64_t)std::max<uint64_t>(rows[b], 1) * width);
    out.ptrs.push_back(slice->mutable_data());
    out.buffers.push_back(std::move(slice));
  }
  return out;
}

}  // namespace olap
