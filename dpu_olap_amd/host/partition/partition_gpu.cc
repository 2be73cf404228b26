#include "partition/partition_gpu.h"


namespace upmemeval {
namespace partition {

arrow::Status PartitionGpu::Prepare() {
  timers_ = std::make_shared<timer::Timers>(system_.nr_ranks());
  return system_.BeginTimers();
}

// Keys travel with the global row index as payload (the reference's
// selection_indices_vector, dpu/shared/kernels/partition.c:233-294); every other column is
// then gathered by those row ids, which is what Partitioner::TakeKernel does on the DPUs
// (host/partition/partitioner.cc:185-207).
arrow::Result<arrow::RecordBatchVector> PartitionGpu::Run() {
  if (!timers_) return arrow::Status::Invalid("Prepare() was not called");
  const int key = schema_->GetFieldIndex(key_column_);
  if (key < 0) return arrow::Status::Invalid("no column named ", key_column_);
  ARROW_ASSIGN_OR_RAISE(auto keys, olap::MakeColumnTable(batches_, key));
  uint64_t n = 0;
  for (auto r : keys.rows) n += r;
  ARROW_ASSIGN_OR_RAISE(auto kbuf, arrow::AllocateBuffer((int64_t)(n ? n : 1) * 4));
  ARROW_ASSIGN_OR_RAISE(auto rbuf, arrow::AllocateBuffer((int64_t)(n ? n : 1) * 4));
  std::vector<uint64_t> offsets(nr_partitions_ + 1);
  ARROW_RETURN_NOT_OK(system_.Check(olap_partition_u32(
      system_.ctx(), reinterpret_cast<const uint32_t* const*>(keys.ptrs.data()), nullptr, keys.rows.data(),
      keys.ptrs.size(), OLAP_MEM_HOST, nr_partitions_, reinterpret_cast<uint32_t*>(kbuf->mutable_data()),
      reinterpret_cast<uint32_t*>(rbuf->mutable_data()), n, OLAP_MEM_HOST, offsets.data())));
  std::shared_ptr<arrow::Buffer> kshared = std::move(kbuf), rshared = std::move(rbuf);
  auto all_keys = olap::WrapBuffer(arrow::uint32(), (int64_t)n, kshared);
  // gather every other column by the row ids on the GPU (olap_take over the column laid out
  // as one contiguous batch; the concatenation is a host copy, not a computation)
  ARROW_ASSIGN_OR_RAISE(auto table, arrow::Table::FromRecordBatches(schema_, batches_));
  arrow::ArrayVector gathered(schema_->num_fields());
  for (int c = 0; c < schema_->num_fields(); ++c) {
    if (c == key) {
      gathered[c] = all_keys;
      continue;
    }
    ARROW_ASSIGN_OR_RAISE(int dtype, olap::DtypeOf(*schema_->field(c)->type()));
    const int width = (dtype == OLAP_U32 || dtype == OLAP_I32) ? 4 : 8;
    std::shared_ptr<arrow::Array> flat;
    if (table->column(c)->num_chunks() == 1) {
      flat = table->column(c)->chunk(0);
    } else if (table->column(c)->num_chunks() == 0) {
      ARROW_ASSIGN_OR_RAISE(flat, arrow::MakeEmptyArray(schema_->field(c)->type()));
    } else {
      ARROW_ASSIGN_OR_RAISE(flat, arrow::Concatenate(table->column(c)->chunks()));
    }
    ARROW_ASSIGN_OR_RAISE(auto obuf, arrow::AllocateBuffer((int64_t)(n ? n : 1) * width));
    const void* dptr = flat->data()->buffers[1] ? flat->data()->buffers[1]->data() + flat->data()->offset * width : nullptr;
    const uint32_t* iptr = reinterpret_cast<const uint32_t*>(rshared->data());
    void* optr = obuf->mutable_data();
    const uint64_t drows = (uint64_t)flat->length(), irows = n;
    if (n) {
      ARROW_RETURN_NOT_OK(system_.Check(olap_take(system_.ctx(), dtype, &dptr, &drows, &iptr, &irows, 1, OLAP_MEM_HOST,
                                                  &optr, OLAP_MEM_HOST)));
    }
    gathered[c] = olap::WrapBuffer(schema_->field(c)->type(), (int64_t)n, std::shared_ptr<arrow::Buffer>(std::move(obuf)));
  }
  arrow::RecordBatchVector out;
  for (uint32_t p = 0; p < nr_partitions_; ++p) {
    const int64_t lo = (int64_t)offsets[p], len = (int64_t)(offsets[p + 1] - offsets[p]);
    arrow::ArrayVector cols;
    for (int c = 0; c < schema_->num_fields(); ++c) cols.push_back(gathered[c]->Slice(lo, len));
    out.push_back(arrow::RecordBatch::Make(schema_, len, cols));
  }
  ARROW_RETURN_NOT_OK(system_.FoldTimers(timers_.get()));
  return out;
}

}  // namespace partition
}  // namespace upmemeval
