#include "join/join_gpu.h"

namespace upmemeval {
namespace join {

// Phases reported: the reference's join timers are outer/full/build/probe/take/partition*
// (join_dpu.cc:146-148); here the library's phases (partition-hist, partition-scan,
// partition-scatter, join-probe, copy-to-gpu, copy-from-gpu) are folded in by FoldTimers.
arrow::Status JoinGpu::Prepare() {
  timers_ = std::make_shared<timer::Timers>(system_.nr_ranks());
  return system_.BeginTimers();
}

arrow::Result<std::shared_ptr<arrow::Table>> JoinGpu::Run() {
  if (!timers_) return arrow::Status::Invalid("Prepare() was not called");
  auto outer = timers_->New("outer");
  outer->Start();
  const int fk = left_schema_->GetFieldIndex("fk");
  const int pk = right_schema_->GetFieldIndex("pk");
  if (fk < 0 || pk < 0) return arrow::Status::Invalid("left table needs a column 'fk', right table a column 'pk'");
  // The kernels carry one 32-bit payload per side with the key (the benchmark schema
  // {fk, y} x {pk, x}); further payload columns would go through row ids + take.
  if (left_schema_->num_fields() != 2 || right_schema_->num_fields() != 2)
    return arrow::Status::NotImplemented("exactly one payload column per side (schema {fk, y} / {pk, x})");
  const int ly = 1 - fk, rx = 1 - pk;
  for (const auto& f : {left_schema_->field(fk), left_schema_->field(ly), right_schema_->field(pk), right_schema_->field(rx)})
    if (f->type()->id() != arrow::Type::UINT32 && f->type()->id() != arrow::Type::INT32)
      return arrow::Status::TypeError("32-bit integer join columns expected, got ", f->ToString());
  ARROW_ASSIGN_OR_RAISE(auto lk, olap::MakeColumnTable(left_batches_, fk));
  ARROW_ASSIGN_OR_RAISE(auto lp, olap::MakeColumnTable(left_batches_, ly));
  ARROW_ASSIGN_OR_RAISE(auto rk, olap::MakeColumnTable(right_batches_, pk));
  ARROW_ASSIGN_OR_RAISE(auto rp, olap::MakeColumnTable(right_batches_, rx));
  uint64_t n_left = 0;
  for (auto r : lk.rows) n_left += r;
  std::shared_ptr<arrow::Buffer> bufs[3];
  for (auto& b : bufs) {
    ARROW_ASSIGN_OR_RAISE(auto nb, arrow::AllocateBuffer((int64_t)(n_left ? n_left : 1) * 4));
    b = std::move(nb);
  }
  uint64_t out_rows = 0;
  ARROW_RETURN_NOT_OK(system_.Check(olap_join_u32(
      system_.ctx(), reinterpret_cast<const uint32_t* const*>(lk.ptrs.data()),
      reinterpret_cast<const uint32_t* const*>(lp.ptrs.data()), lk.rows.data(), lk.ptrs.size(),
      reinterpret_cast<const uint32_t* const*>(rk.ptrs.data()), reinterpret_cast<const uint32_t* const*>(rp.ptrs.data()),
      rk.rows.data(), rk.ptrs.size(), OLAP_MEM_HOST, reinterpret_cast<uint32_t*>(bufs[0]->mutable_data()),
      reinterpret_cast<uint32_t*>(bufs[1]->mutable_data()), reinterpret_cast<uint32_t*>(bufs[2]->mutable_data()), n_left,
      OLAP_MEM_HOST, &out_rows, &stats_)));
  // left columns in their schema order, then the right payload (join_dpu.cc:314-363)
  arrow::FieldVector fields = left_schema_->fields();
  fields.push_back(right_schema_->field(rx));
  std::vector<std::shared_ptr<arrow::Array>> cols(3);
  cols[fk] = olap::WrapBuffer(left_schema_->field(fk)->type(), (int64_t)out_rows, bufs[0]);
  cols[ly] = olap::WrapBuffer(left_schema_->field(ly)->type(), (int64_t)out_rows, bufs[1]);
  cols[2] = olap::WrapBuffer(right_schema_->field(rx)->type(), (int64_t)out_rows, bufs[2]);
  outer->Stop();
  ARROW_RETURN_NOT_OK(system_.FoldTimers(timers_.get()));
  return arrow::Table::Make(arrow::schema(fields), cols);
}

}  // namespace join
}  // namespace upmemeval
