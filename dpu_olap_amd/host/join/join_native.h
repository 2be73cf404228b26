// JoinNative -- Arrow CPU counterpart (reference host/join/join_native.{h,cc}): two sources
// -> hashjoin{INNER, fk = pk} -> sink, "pk" dropped from the result.  partitioned = true
// joins aligned batch pairs one plan at a time and concatenates (join_native.cc:94-111).
#pragma once

#include <arrow/acero/api.h>
#include <arrow/api.h>
#include <arrow/compute/api.h>

#include "generator/generator.h"
#include "timer/timer.h"

namespace upmemeval {
namespace join {

class JoinNative {
 public:
  JoinNative(std::shared_ptr<arrow::Schema> left_schema, std::shared_ptr<arrow::Schema> right_schema,
             arrow::RecordBatchVector left_batches, arrow::RecordBatchVector right_batches, bool partitioned = false)
      : left_schema_(std::move(left_schema)),
        right_schema_(std::move(right_schema)),
        left_batches_(generator::ToExecBatches(left_batches)),
        right_batches_(generator::ToExecBatches(right_batches)),
        partitioned_(partitioned) {}

  arrow::Status Prepare();
  arrow::Result<std::shared_ptr<arrow::Table>> Run();
  std::shared_ptr<timer::Timers> Timers() { return nullptr; }

 private:
  arrow::Result<std::shared_ptr<arrow::Table>> JoinOnce(std::vector<arrow::compute::ExecBatch> left,
                                                        std::vector<arrow::compute::ExecBatch> right);
  std::shared_ptr<arrow::Schema> left_schema_, right_schema_;
  std::vector<arrow::compute::ExecBatch> left_batches_, right_batches_;
  bool partitioned_;
};

}  // namespace join
}  // namespace upmemeval
