// FilterNative -- the Arrow CPU operator the reference compares against
// (host/filter/filter_native.{h,cc}), re-stated for Arrow 25 (arrow::acero; the reference
// targets Arrow 8.0's arrow::compute exec plan): source -> filter(v < 2^30) -> sink on the
// global CPU thread pool.
#pragma once

#include <arrow/acero/api.h>
#include <arrow/api.h>
#include <arrow/compute/api.h>

#include "generator/generator.h"
#include "timer/timer.h"

namespace upmemeval {
namespace filter {

class FilterNative {
 public:
  FilterNative(std::shared_ptr<arrow::Schema> schema, arrow::RecordBatchVector batches)
      : schema_(std::move(schema)), batches_(generator::ToExecBatches(batches)) {}

  arrow::Status Prepare();
  arrow::Result<std::shared_ptr<arrow::Table>> GetResult();
  arrow::Result<uint64_t> Run();
  std::shared_ptr<timer::Timers> Timers() { return nullptr; }

 private:
  std::shared_ptr<arrow::Schema> schema_;
  std::vector<arrow::compute::ExecBatch> batches_;
  std::shared_ptr<arrow::acero::ExecPlan> plan_;
  arrow::AsyncGenerator<std::optional<arrow::compute::ExecBatch>> sink_;
};

// Starts `plan`, drains the sink into a Table, waits for the plan to finish.
arrow::Result<std::shared_ptr<arrow::Table>> RunPlanToTable(
    const std::shared_ptr<arrow::acero::ExecPlan>& plan, const std::shared_ptr<arrow::Schema>& schema,
    arrow::AsyncGenerator<std::optional<arrow::compute::ExecBatch>> sink);

}  // namespace filter
}  // namespace upmemeval
