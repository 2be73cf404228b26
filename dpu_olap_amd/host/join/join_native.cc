#include "join/join_native.h"

#include "filter/filter_native.h"  // RunPlanToTable

namespace upmemeval {
namespace join {

namespace ac = ::arrow::acero;
namespace cp = ::arrow::compute;

arrow::Status JoinNative::Prepare() { return cp::Initialize(); }

arrow::Result<std::shared_ptr<arrow::Table>> JoinNative::JoinOnce(std::vector<cp::ExecBatch> left,
                                                                 std::vector<cp::ExecBatch> right) {
  ARROW_ASSIGN_OR_RAISE(auto plan, ac::ExecPlan::Make(*cp::threaded_exec_context()));
  auto lgen = generator::MakeAsyncGenerator(std::move(left), true);
  auto rgen = generator::MakeAsyncGenerator(std::move(right), true);
  ARROW_ASSIGN_OR_RAISE(ac::ExecNode * lsrc,
                        ac::MakeExecNode("source", plan.get(), {}, ac::SourceNodeOptions{left_schema_, lgen}));
  ARROW_ASSIGN_OR_RAISE(ac::ExecNode * rsrc,
                        ac::MakeExecNode("source", plan.get(), {}, ac::SourceNodeOptions{right_schema_, rgen}));
  ac::HashJoinNodeOptions opts{ac::JoinType::INNER, {"fk"}, {"pk"}, cp::literal(true), "_l", "_r"};
  ARROW_ASSIGN_OR_RAISE(ac::ExecNode * hj, ac::MakeExecNode("hashjoin", plan.get(), {lsrc, rsrc}, opts));
  auto out_schema = hj->output_schema();
  arrow::AsyncGenerator<std::optional<cp::ExecBatch>> sink;
  ARROW_RETURN_NOT_OK(ac::MakeExecNode("sink", plan.get(), {hj}, ac::SinkNodeOptions{&sink}));
  ARROW_RETURN_NOT_OK(plan->Validate());
  ARROW_ASSIGN_OR_RAISE(auto table, filter::RunPlanToTable(plan, out_schema, std::move(sink)));
  return table->RemoveColumn(table->schema()->GetFieldIndex("pk"));  // reference join_native.cc:75
}

arrow::Result<std::shared_ptr<arrow::Table>> JoinNative::Run() {
  if (!partitioned_) return JoinOnce(left_batches_, right_batches_);
  if (left_batches_.size() != right_batches_.size()) return arrow::Status::Invalid("partitioned join needs aligned batches");
  std::vector<std::shared_ptr<arrow::Table>> parts;
  for (size_t i = 0; i < left_batches_.size(); ++i) {
    ARROW_ASSIGN_OR_RAISE(auto t, JoinOnce({left_batches_[i]}, {right_batches_[i]}));
    parts.push_back(std::move(t));
  }
  return arrow::ConcatenateTables(parts);
}

}  // namespace join
}  // namespace upmemeval
