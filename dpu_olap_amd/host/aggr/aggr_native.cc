#include "aggr/aggr_native.h"

#include "filter/filter_native.h"  // RunPlanToTable

namespace upmemeval {
namespace aggr {

namespace ac = ::arrow::acero;
namespace cp = ::arrow::compute;

template <class ArrayType>
arrow::Status AggrNative<ArrayType>::Prepare() {
  ARROW_RETURN_NOT_OK(cp::Initialize());
  auto data = generator::MakeAsyncGenerator(batches_, /*parallel=*/true);
  ARROW_ASSIGN_OR_RAISE(plan_, ac::ExecPlan::Make(*cp::threaded_exec_context()));
  ARROW_ASSIGN_OR_RAISE(ac::ExecNode * source,
                        ac::MakeExecNode("source", plan_.get(), {}, ac::SourceNodeOptions{schema_, data}));
  // scalar aggregate fn(v) named "fn(v)" (reference aggr_native.cc:68-73)
  ac::AggregateNodeOptions opts{{cp::Aggregate(fn_, nullptr, arrow::FieldRef("v"), fn_ + "(v)")}};
  ARROW_ASSIGN_OR_RAISE(ac::ExecNode * agg, ac::MakeExecNode("aggregate", plan_.get(), {source}, opts));
  out_schema_ = agg->output_schema();
  ARROW_RETURN_NOT_OK(ac::MakeExecNode("sink", plan_.get(), {agg}, ac::SinkNodeOptions{&sink_}));
  return plan_->Validate();
}

template <class ArrayType>
arrow::Result<typename ArrayType::value_type> AggrNative<ArrayType>::Run() {
  ARROW_ASSIGN_OR_RAISE(auto table, filter::RunPlanToTable(plan_, out_schema_, std::move(sink_)));
  auto arr = std::dynamic_pointer_cast<ArrayType>(table->column(0)->chunk(0));
  if (!arr) return arrow::Status::TypeError("unexpected aggregate result type ", table->column(0)->type()->ToString());
  return arr->Value(0);
}

template class AggrNative<::arrow::UInt64Array>;
template class AggrNative<::arrow::Int64Array>;
template class AggrNative<::arrow::DoubleArray>;

}  // namespace aggr
}  // namespace upmemeval
