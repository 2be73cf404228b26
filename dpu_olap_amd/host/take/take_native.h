// TakeNative -- Arrow CPU counterpart (reference host/take/take_native.{h,cc}): one
// compute::Take(data, indices, NoBoundsCheck) task per batch on the CPU thread pool.
#pragma once

#include <arrow/api.h>
#include <arrow/compute/api.h>

#include "timer/timer.h"

namespace upmemeval {
namespace take {

class TakeNative {
 public:
  TakeNative(std::shared_ptr<arrow::Schema> schema, arrow::RecordBatchVector batches,
             arrow::RecordBatchVector indices_batches)
      : schema_(std::move(schema)), batches_(std::move(batches)), indices_batches_(std::move(indices_batches)) {}

  arrow::Status Prepare();
  arrow::Result<std::shared_ptr<::arrow::Table>> Run();
  std::shared_ptr<timer::Timers> Timers() { return nullptr; }

 private:
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_;
  arrow::RecordBatchVector indices_batches_;
};

}  // namespace take
}  // namespace upmemeval
