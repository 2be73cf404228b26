#include "aggr/aggr_gpu.h"

namespace upmemeval {
namespace aggr {

arrow::Status SumGpu::Prepare() {
  timers_ = std::make_shared<timer::Timers>(system_.nr_ranks());
  return system_.BeginTimers();
}

arrow::Result<uint64_t> SumGpu::Run() {
  if (!timers_) return arrow::Status::Invalid("Prepare() was not called");
  if (batches_.empty()) return (uint64_t)0;
  ARROW_ASSIGN_OR_RAISE(int dtype, olap::DtypeOf(*batches_[0]->column(0)->type()));
  ARROW_ASSIGN_OR_RAISE(auto in, olap::MakeColumnTable(batches_, 0));
  uint64_t sum = 0;  // the per-round partials are added inside the library (aggr_dpu.cc:82-84)
  ARROW_RETURN_NOT_OK(system_.Check(
      olap_sum(system_.ctx(), dtype, in.ptrs.data(), in.rows.data(), in.ptrs.size(), OLAP_MEM_HOST, &sum)));
  ARROW_RETURN_NOT_OK(system_.FoldTimers(timers_.get()));
  return sum;
}

}  // namespace aggr
}  // namespace upmemeval
