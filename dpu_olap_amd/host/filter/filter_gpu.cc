#include "filter/filter_gpu.h"

namespace upmemeval {
namespace filter {

// The reference loads the DPU program here, inside the timed region (filter_dpu.cc:23-32);
// the HIP kernels are linked in, so Prepare only arms the timers.
arrow::Status FilterGpu::Prepare() {
  timers_ = std::make_shared<timer::Timers>(system_.nr_ranks());
  return system_.BeginTimers();
}

arrow::Result<std::shared_ptr<arrow::ChunkedArray>> FilterGpu::GetResult() {
  if (!timers_) return arrow::Status::Invalid("Prepare() was not called");
  auto build = timers_->New("build-result");
  if (batches_.empty()) return arrow::ChunkedArray::Make({}, arrow::uint32());
  const auto type = batches_[0]->column(0)->type();
  ARROW_ASSIGN_OR_RAISE(int dtype, olap::DtypeOf(*type));
  const int width = (dtype == OLAP_U32 || dtype == OLAP_I32) ? 4 : 8;
  ARROW_ASSIGN_OR_RAISE(auto in, olap::MakeColumnTable(batches_, 0));
  // worst-case output buffer per batch, as the reference sizes its per-DPU pull (filter_dpu.cc:67-83)
  std::vector<std::shared_ptr<arrow::Buffer>> bufs(batches_.size());
  std::vector<void*> out_ptrs(batches_.size());
  for (size_t b = 0; b < batches_.size(); ++b) {
    ARROW_ASSIGN_OR_RAISE(auto buf, arrow::AllocateBuffer((int64_t)in.rows[b] * width));
    out_ptrs[b] = buf->mutable_data();
    bufs[b] = std::move(buf);
  }
  std::vector<uint64_t> out_rows(batches_.size());
  uint64_t total = 0;
  const uint64_t threshold = 1ull << 30;  // predicate v < 2^30, dpu/shared/kernels/filter.c:25
  ARROW_RETURN_NOT_OK(system_.Check(olap_filter_lt(system_.ctx(), dtype, in.ptrs.data(), in.rows.data(), in.ptrs.size(),
                                                   OLAP_MEM_HOST, &threshold, out_ptrs.data(), OLAP_MEM_HOST,
                                                   out_rows.data(), &total)));
  build->Start();
  arrow::ArrayVector chunks;
  chunks.reserve(batches_.size());
  for (size_t b = 0; b < batches_.size(); ++b) chunks.push_back(olap::WrapBuffer(type, (int64_t)out_rows[b], bufs[b]));
  build->Stop();
  ARROW_RETURN_NOT_OK(system_.FoldTimers(timers_.get()));
  return arrow::ChunkedArray::Make(chunks, type);
}

arrow::Result<uint64_t> FilterGpu::Run() {
  ARROW_ASSIGN_OR_RAISE(auto r, GetResult());
  return (uint64_t)r->length();
}

}  // namespace filter
}  // namespace upmemeval
