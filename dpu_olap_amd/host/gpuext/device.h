// device.h -- C++ face of the C ABI (include/olap_gpu.h): what dpu::DpuSet is to the
// reference's operators (host/dpuext/dpuext.hpp:148-927).  RAII context, status mapping to
// arrow::Status (the reference maps dpu_error_t through DPU_RETURN_NOT_OK,
// host/dpuext/status.h:7-11), device buffers wrapped as arrow::Buffer so results stay
// arrow::Array objects, and the fold of the library's HIP-event timers into timer::Timers.
#pragma once

#include <arrow/api.h>

#include <memory>
#include <string>
#include <vector>

#include "olap_gpu.h"
#include "timer/timer.h"

namespace olap {

class Device {
 public:
  // Mirrors `dpu::DpuSet::allocate(n)`: n GPUs (this build drives device 0..n-1 one context
  // each; multi-GPU joins go through the one-process-per-GPU launcher).
  static arrow::Result<std::shared_ptr<Device>> Allocate(int device_id = 0) {
    olap_ctx* ctx = nullptr;
    int rc = olap_create(device_id, &ctx);
    if (rc != OLAP_OK) return arrow::Status::UnknownError("olap_create failed with code ", rc, " (no usable HIP device?)");
    return std::shared_ptr<Device>(new Device(ctx));
  }
  ~Device() { olap_destroy(ctx_); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;

  olap_ctx* ctx() const { return ctx_; }
  size_t nr_ranks() const { return 1; }

  arrow::Status Check(int rc) const {
    if (rc == OLAP_OK) return arrow::Status::OK();
    return arrow::Status::UnknownError("olap error ", rc, ": ", olap_last_error(ctx_));
  }

  std::string Name() const {
    char name[256] = "";
    olap_device_info(ctx_, name, sizeof name, nullptr, nullptr, nullptr);
    return name;
  }

  // Start of an operator's Prepare(): HIP-event timers on, counters cleared.
  arrow::Status BeginTimers() const {
    ARROW_RETURN_NOT_OK(Check(olap_set_profiling(ctx_, 1)));
    return Check(olap_reset_timers(ctx_));
  }

  // Adds the library's per-phase device times to `timers` under the reference's names
  // ("copy-to-gpu" ~ copy-to-dpu, "gpu-work" ~ dpu-work, "copy-from-gpu" ~ copy-from-dpu)
  // plus one entry per kernel phase.
  arrow::Status FoldTimers(timer::Timers* timers) const {
    const char* names[32];
    double ms[32];
    uint64_t launches[32];
    int n = 0;
    ARROW_RETURN_NOT_OK(Check(olap_get_timers(ctx_, names, ms, launches, 32, &n)));
    double work = 0;
    for (int i = 0; i < n && i < 32; ++i) {
      const std::string name = names[i];
      const auto d = std::chrono::nanoseconds((int64_t)(ms[i] * 1e6));
      timers->Get(name)->Add(d);
      if (name != "copy-to-gpu" && name != "copy-from-gpu") work += ms[i];
    }
    timers->Get("gpu-work")->Add(std::chrono::nanoseconds((int64_t)(work * 1e6)));
    return arrow::Status::OK();
  }

 private:
  explicit Device(olap_ctx* ctx) : ctx_(ctx) {}
  olap_ctx* ctx_;
};

// Column `column_index` of every batch as the (pointer, rows) tables the C ABI takes.
// Like the reference (host/dpuext/arrow_utils.cc:25-39,64-66) only the data buffer
// (buffers[1]) is used; the offset of a sliced array is honoured.
struct ColumnTable {
  std::vector<const void*> ptrs;
  std::vector<uint64_t> rows;
};

inline arrow::Result<ColumnTable> MakeColumnTable(const arrow::RecordBatchVector& batches, int column_index) {
  ColumnTable t;
  t.ptrs.reserve(batches.size());
  t.rows.reserve(batches.size());
  for (const auto& b : batches) {
    if (column_index < 0 || column_index >= b->num_columns()) return arrow::Status::Invalid("column index out of range");
    const auto& data = b->column_data(column_index);
    const auto* fw = dynamic_cast<const arrow::FixedWidthType*>(data->type.get());
    if (!fw || data->buffers.size() < 2) return arrow::Status::TypeError("fixed-width column expected");
    if (data->GetNullCount() != 0) return arrow::Status::NotImplemented("nullable columns are outside the hot path");
    const int width = fw->bit_width() / 8;
    const uint8_t* base = data->buffers[1] ? data->buffers[1]->data() : nullptr;
    t.ptrs.push_back(base ? base + (int64_t)data->offset * width : nullptr);
    t.rows.push_back((uint64_t)data->length);
  }
  return t;
}

inline arrow::Result<int> DtypeOf(const arrow::DataType& t) {
  switch (t.id()) {
    case arrow::Type::UINT32: return (int)OLAP_U32;
    case arrow::Type::INT32: return (int)OLAP_I32;
    case arrow::Type::UINT64: return (int)OLAP_U64;
    case arrow::Type::INT64: return (int)OLAP_I64;
    default: return arrow::Status::TypeError("the GPU hot path covers uint32/int32/uint64/int64, got ", t.ToString());
  }
}

// arrow::Array over a freshly filled host buffer: ArrayData{nullptr, buffer}, exactly how the
// reference wraps what it pulls from the DPUs (host/filter/filter_dpu.cc:91-92).
inline std::shared_ptr<arrow::Array> WrapBuffer(const std::shared_ptr<arrow::DataType>& type, int64_t length,
                                                std::shared_ptr<arrow::Buffer> buffer) {
  return arrow::MakeArray(arrow::ArrayData::Make(type, length, {nullptr, std::move(buffer)}));
}

}  // namespace olap
