// FilterGpu -- the GPU counterpart of the reference's FilterDpu
// (host/filter/filter_dpu.h:14-29): same constructor shape (device, batches), same
// Prepare() / GetResult() / Run() / Timers().
#pragma once

#include <arrow/api.h>

#include "gpuext/device.h"
#include "timer/timer.h"

namespace upmemeval {
namespace filter {

class FilterGpu {
 public:
  FilterGpu(olap::Device& system, arrow::RecordBatchVector batches) : system_(system), batches_(std::move(batches)) {}

  arrow::Status Prepare();
  // one chunk per input batch, in batch order (filter_dpu.cc:95,162-166)
  arrow::Result<std::shared_ptr<arrow::ChunkedArray>> GetResult();
  arrow::Result<uint64_t> Run();
  std::shared_ptr<timer::Timers> Timers() { return timers_; }

 private:
  olap::Device& system_;
  arrow::RecordBatchVector batches_;
  std::shared_ptr<timer::Timers> timers_;
};

}  // namespace filter
}  // namespace upmemeval
