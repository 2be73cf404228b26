// AggrNative -- Arrow CPU counterpart (reference host/aggr/aggr_native.{h,cc}):
// source -> aggregate{fn on "v"} -> sink; Run() returns element 0 of the single result column.
#pragma once

#include <arrow/acero/api.h>
#include <arrow/api.h>
#include <arrow/compute/api.h>

#include "generator/generator.h"
#include "timer/timer.h"

namespace upmemeval {
namespace aggr {

template <class ArrayType>
class AggrNative {
 public:
  AggrNative(std::shared_ptr<arrow::Schema> schema, arrow::RecordBatchVector batches, std::string fn)
      : schema_(std::move(schema)), batches_(generator::ToExecBatches(batches)), fn_(std::move(fn)) {}

  arrow::Status Prepare();
  arrow::Result<typename ArrayType::value_type> Run();
  std::shared_ptr<timer::Timers> Timers() { return nullptr; }

 private:
  std::shared_ptr<arrow::Schema> schema_;
  std::vector<arrow::compute::ExecBatch> batches_;
  std::string fn_;
  std::shared_ptr<arrow::acero::ExecPlan> plan_;
  std::shared_ptr<arrow::Schema> out_schema_;
  arrow::AsyncGenerator<std::optional<arrow::compute::ExecBatch>> sink_;
};

}  // namespace aggr
}  // namespace upmemeval
