#include "filter/filter_native.h"

#include <arrow/util/thread_pool.h>

namespace upmemeval {
namespace filter {

namespace ac = ::arrow::acero;
namespace cp = ::arrow::compute;

arrow::Result<std::shared_ptr<arrow::Table>> RunPlanToTable(
    const std::shared_ptr<ac::ExecPlan>& plan, const std::shared_ptr<arrow::Schema>& schema,
    arrow::AsyncGenerator<std::optional<cp::ExecBatch>> sink) {
  auto reader = ac::MakeGeneratorReader(schema, std::move(sink), arrow::default_memory_pool());
  plan->StartProducing();
  ARROW_ASSIGN_OR_RAISE(auto table, arrow::Table::FromRecordBatchReader(reader.get()));
  // The reference (Arrow 8) calls StopProducing() here; on Arrow 25 that races with the
  // plan's own completion and reports "cancelled", so only wait for the finished future.
  ARROW_RETURN_NOT_OK(plan->finished().status());
  return table;
}

arrow::Status FilterNative::Prepare() {
  ARROW_RETURN_NOT_OK(cp::Initialize());
  auto data = generator::MakeAsyncGenerator(batches_, /*parallel=*/true);
  ARROW_ASSIGN_OR_RAISE(plan_, ac::ExecPlan::Make(*cp::threaded_exec_context()));
  ARROW_ASSIGN_OR_RAISE(ac::ExecNode * source,
                        ac::MakeExecNode("source", plan_.get(), {}, ac::SourceNodeOptions{schema_, data}));
  // keep rows with v < 2^30 (reference filter_native.cc:59; an int32 literal against a uint32 column)
  cp::Expression keep = cp::less(cp::field_ref("v"), cp::literal(1 << 30));
  ARROW_ASSIGN_OR_RAISE(ac::ExecNode * filt,
                        ac::MakeExecNode("filter", plan_.get(), {source}, ac::FilterNodeOptions{keep}));
  ARROW_RETURN_NOT_OK(ac::MakeExecNode("sink", plan_.get(), {filt}, ac::SinkNodeOptions{&sink_}));
  return plan_->Validate();
}

arrow::Result<std::shared_ptr<arrow::Table>> FilterNative::GetResult() {
  return RunPlanToTable(plan_, schema_, std::move(sink_));
}

arrow::Result<uint64_t> FilterNative::Run() {
  ARROW_ASSIGN_OR_RAISE(auto t, GetResult());
  return (uint64_t)t->num_rows();
}

}  // namespace filter
}  // namespace upmemeval
