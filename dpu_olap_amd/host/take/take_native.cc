#include "take/take_native.h"

#include <arrow/util/thread_pool.h>

namespace upmemeval {
namespace take {

arrow::Status TakeNative::Prepare() { return arrow::compute::Initialize(); }

arrow::Result<std::shared_ptr<::arrow::Table>> TakeNative::Run() {
  auto* pool = arrow::internal::GetCpuThreadPool();
  arrow::RecordBatchVector out(batches_.size());
  std::vector<arrow::Future<>> pending;
  pending.reserve(batches_.size());
  for (size_t i = 0; i < batches_.size(); ++i) {
    ARROW_ASSIGN_OR_RAISE(auto fut, pool->Submit([this, i, &out]() -> arrow::Status {
      ARROW_ASSIGN_OR_RAISE(auto taken, arrow::compute::Take(batches_[i]->column(0), indices_batches_[i]->column(0),
                                                             arrow::compute::TakeOptions::NoBoundsCheck()));
      auto arr = taken.make_array();
      out[i] = arrow::RecordBatch::Make(schema_, arr->length(), {arr});
      return arrow::Status::OK();
    }));
    pending.push_back(std::move(fut));
  }
  for (auto& f : pending) ARROW_RETURN_NOT_OK(f.status());
  return arrow::Table::FromRecordBatches(out);
}

}  // namespace take
}  // namespace upmemeval
