#include "operators_native.h"

#include <arrow/util/thread_pool.h>

namespace upmemeval {

namespace ac = arrow::acero;
namespace cp = arrow::compute;

// ---------------------------------------------------------------- SinkPlan
arrow::Status SinkPlan::Open() {
  ARROW_RETURN_NOT_OK(cp::Initialize());
  ARROW_ASSIGN_OR_RAISE(plan_, ac::ExecPlan::Make(*cp::threaded_exec_context()));
  return arrow::Status::OK();
}

arrow::Result<ac::ExecNode*> SinkPlan::AddSource(const std::shared_ptr<arrow::Schema>& schema, std::vector<cp::ExecBatch> batches) {
  return ac::MakeExecNode("source", plan_.get(), {},
                          ac::SourceNodeOptions{schema, generator::MakeAsyncGenerator(std::move(batches), /*parallel=*/true)});
}

arrow::Result<ac::ExecNode*> SinkPlan::Add(const std::string& factory, std::vector<ac::ExecNode*> inputs,
                                           const ac::ExecNodeOptions& options) {
  return ac::MakeExecNode(factory, plan_.get(), std::move(inputs), options);
}

arrow::Status SinkPlan::Close(ac::ExecNode* last) {
  out_schema_ = last->output_schema();
  ARROW_RETURN_NOT_OK(ac::MakeExecNode("sink", plan_.get(), {last}, ac::SinkNodeOptions{&sink_}));
  return plan_->Validate();
}

arrow::Result<std::shared_ptr<arrow::Table>> SinkPlan::Collect() {
  auto reader = ac::MakeGeneratorReader(out_schema_, std::move(sink_), arrow::default_memory_pool());
  plan_->StartProducing();
  ARROW_ASSIGN_OR_RAISE(auto table, arrow::Table::FromRecordBatchReader(reader.get()));
  // no StopProducing(): on Arrow 25 it races with the plan's own completion ("cancelled")
  ARROW_RETURN_NOT_OK(plan_->finished().status());
  return table;
}

// ------------------------------------------------------------------ filter
namespace filter {

arrow::Status FilterNative::Prepare() {
  ARROW_RETURN_NOT_OK(plan_.Open());
  ARROW_ASSIGN_OR_RAISE(auto* source, plan_.AddSource(schema_, batches_));
  // an int32 literal against the uint32 column, as in the reference (filter_native.cc:59)
  ARROW_ASSIGN_OR_RAISE(auto* node, plan_.Add("filter", {source}, ac::FilterNodeOptions{cp::less(cp::field_ref("v"), cp::literal(1 << 30))}));
  return plan_.Close(node);
}

arrow::Result<uint64_t> FilterNative::Run() {
  ARROW_ASSIGN_OR_RAISE(auto table, GetResult());
  return (uint64_t)table->num_rows();
}

}  // namespace filter

// --------------------------------------------------------------------- sum
namespace aggr {

template <class ArrayType>
arrow::Status AggrNative<ArrayType>::Prepare() {
  ARROW_RETURN_NOT_OK(plan_.Open());
  ARROW_ASSIGN_OR_RAISE(auto* source, plan_.AddSource(schema_, batches_));
  ac::AggregateNodeOptions options{{cp::Aggregate(fn_, nullptr, arrow::FieldRef("v"), fn_ + "(v)")}};
  ARROW_ASSIGN_OR_RAISE(auto* node, plan_.Add("aggregate", {source}, options));
  return plan_.Close(node);
}

template <class ArrayType>
arrow::Result<typename ArrayType::value_type> AggrNative<ArrayType>::Run() {
  ARROW_ASSIGN_OR_RAISE(auto table, plan_.Collect());
  auto result = std::dynamic_pointer_cast<ArrayType>(table->column(0)->chunk(0));
  if (!result) return arrow::Status::TypeError("unexpected aggregate result type ", table->column(0)->type()->ToString());
  return result->Value(0);
}

template class AggrNative<arrow::UInt64Array>;
template class AggrNative<arrow::Int64Array>;
template class AggrNative<arrow::DoubleArray>;

}  // namespace aggr

// -------------------------------------------------------------------- take
namespace take {

arrow::Result<std::shared_ptr<arrow::Table>> TakeNative::Run() {
  auto* pool = arrow::internal::GetCpuThreadPool();
  arrow::RecordBatchVector results(batches_.size());
  std::vector<arrow::Future<>> tasks;
  for (size_t b = 0; b < batches_.size(); ++b) {
    ARROW_ASSIGN_OR_RAISE(auto task, pool->Submit([this, b, &results]() -> arrow::Status {
      ARROW_ASSIGN_OR_RAISE(auto taken, cp::Take(batches_[b]->column(0), indices_batches_[b]->column(0), cp::TakeOptions::NoBoundsCheck()));
      auto column = taken.make_array();
      results[b] = arrow::RecordBatch::Make(schema_, column->length(), {column});
      return arrow::Status::OK();
    }));
    tasks.push_back(std::move(task));
  }
  for (auto& t : tasks) ARROW_RETURN_NOT_OK(t.status());
  return arrow::Table::FromRecordBatches(results);
}

}  // namespace take

// -------------------------------------------------------------------- join
namespace join {

arrow::Result<std::shared_ptr<arrow::Table>> JoinNative::HashJoin(std::vector<cp::ExecBatch> left, std::vector<cp::ExecBatch> right) {
  SinkPlan plan;
  ARROW_RETURN_NOT_OK(plan.Open());
  ARROW_ASSIGN_OR_RAISE(auto* lsrc, plan.AddSource(left_schema_, std::move(left)));
  ARROW_ASSIGN_OR_RAISE(auto* rsrc, plan.AddSource(right_schema_, std::move(right)));
  ac::HashJoinNodeOptions options{ac::JoinType::INNER, {"fk"}, {"pk"}, cp::literal(true), "_l", "_r"};
  ARROW_ASSIGN_OR_RAISE(auto* node, plan.Add("hashjoin", {lsrc, rsrc}, options));
  ARROW_RETURN_NOT_OK(plan.Close(node));
  ARROW_ASSIGN_OR_RAISE(auto table, plan.Collect());
  return table->RemoveColumn(table->schema()->GetFieldIndex("pk"));
}

arrow::Result<std::shared_ptr<arrow::Table>> JoinNative::Run() {
  if (!partitioned_) return HashJoin(left_batches_, right_batches_);
  if (left_batches_.size() != right_batches_.size()) return arrow::Status::Invalid("partitioned join needs aligned batches");
  std::vector<std::shared_ptr<arrow::Table>> pieces;
  for (size_t i = 0; i < left_batches_.size(); ++i) {
    ARROW_ASSIGN_OR_RAISE(auto piece, HashJoin({left_batches_[i]}, {right_batches_[i]}));
    pieces.push_back(std::move(piece));
  }
  return arrow::ConcatenateTables(pieces);
}

}  // namespace join

}  // namespace upmemeval
