// TakeGpu -- counterpart of the reference's TakeDpu (host/take/take_dpu.h:14-30).
#pragma once

#include <arrow/api.h>

#include "gpuext/device.h"
#include "timer/timer.h"

namespace upmemeval {
namespace take {

class TakeGpu {
 public:
  TakeGpu(olap::Device& system, arrow::RecordBatchVector batches, arrow::RecordBatchVector indices_batches)
      : system_(system), batches_(std::move(batches)), indices_batches_(std::move(indices_batches)) {}

  arrow::Status Prepare();
  // one record batch per input batch, schema of the data batches (take_dpu.cc:94-99)
  arrow::Result<std::shared_ptr<::arrow::Table>> Run();
  std::shared_ptr<timer::Timers> Timers() { return timers_; }

 private:
  olap::Device& system_;
  arrow::RecordBatchVector batches_;
  arrow::RecordBatchVector indices_batches_;
  std::shared_ptr<timer::Timers> timers_;
};

}  // namespace take
}  // namespace upmemeval
