#include "take/take_gpu.h"

namespace upmemeval {
namespace take {

arrow::Status TakeGpu::Prepare() {
  timers_ = std::make_shared<timer::Timers>(system_.nr_ranks());
  return system_.BeginTimers();
}

arrow::Result<std::shared_ptr<::arrow::Table>> TakeGpu::Run() {
  if (!timers_) return arrow::Status::Invalid("Prepare() was not called");
  if (batches_.size() != indices_batches_.size()) return arrow::Status::Invalid("one index batch per data batch expected");
  if (batches_.empty()) return arrow::Status::Invalid("no batches");
  auto build = timers_->New("build-result");
  const auto type = batches_[0]->column(0)->type();
  ARROW_ASSIGN_OR_RAISE(int dtype, olap::DtypeOf(*type));
  if (indices_batches_[0]->column(0)->type_id() != arrow::Type::UINT32)
    return arrow::Status::TypeError("uint32 selection indices expected (kernel_take, dpu/shared/kernels/take.c:12-15)");
  const int width = (dtype == OLAP_U32 || dtype == OLAP_I32) ? 4 : 8;
  ARROW_ASSIGN_OR_RAISE(auto data, olap::MakeColumnTable(batches_, 0));
  ARROW_ASSIGN_OR_RAISE(auto idx, olap::MakeColumnTable(indices_batches_, 0));
  std::vector<std::shared_ptr<arrow::Buffer>> bufs(batches_.size());
  std::vector<void*> out_ptrs(batches_.size());
  for (size_t b = 0; b < batches_.size(); ++b) {
    ARROW_ASSIGN_OR_RAISE(auto buf, arrow::AllocateBuffer((int64_t)idx.rows[b] * width));
    out_ptrs[b] = buf->mutable_data();
    bufs[b] = std::move(buf);
  }
  ARROW_RETURN_NOT_OK(system_.Check(olap_take(system_.ctx(), dtype, data.ptrs.data(), data.rows.data(),
                                              reinterpret_cast<const uint32_t* const*>(idx.ptrs.data()), idx.rows.data(),
                                              data.ptrs.size(), OLAP_MEM_HOST, out_ptrs.data(), OLAP_MEM_HOST)));
  build->Start();
  arrow::RecordBatchVector results;
  for (size_t b = 0; b < batches_.size(); ++b) {
    auto arr = olap::WrapBuffer(type, (int64_t)idx.rows[b], bufs[b]);
    results.push_back(arrow::RecordBatch::Make(batches_[0]->schema(), arr->length(), {arr}));
  }
  build->Stop();
  ARROW_RETURN_NOT_OK(system_.FoldTimers(timers_.get()));
  return arrow::Table::FromRecordBatches(results);
}

}  // namespace take
}  // namespace upmemeval
