// SumGpu -- counterpart of the reference's SumDpu (host/aggr/aggr_dpu.h:14-28).
#pragma once

#include <arrow/api.h>

#include "gpuext/device.h"
#include "timer/timer.h"

namespace upmemeval {
namespace aggr {

class SumGpu {
 public:
  SumGpu(olap::Device& system, arrow::RecordBatchVector batches) : system_(system), batches_(std::move(batches)) {}

  arrow::Status Prepare();
  arrow::Result<uint64_t> Run();
  std::shared_ptr<timer::Timers> Timers() { return timers_; }

 private:
  olap::Device& system_;
  arrow::RecordBatchVector batches_;
  std::shared_ptr<timer::Timers> timers_;
};

}  // namespace aggr
}  // namespace upmemeval
