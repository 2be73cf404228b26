#include "operators_gpu.h"

namespace upmemeval {

using olap::ColumnTable;

// ------------------------------------------------------------------ filter
namespace filter {

arrow::Result<std::shared_ptr<arrow::ChunkedArray>> FilterGpu::GetResult() {
  ARROW_RETURN_NOT_OK(Prepared());
  if (batches_.empty()) return arrow::ChunkedArray::Make({}, arrow::uint32());
  const auto type = batches_.front()->column(0)->type();
  ARROW_ASSIGN_OR_RAISE(const int dtype, olap::DtypeOf(*type));
  ARROW_ASSIGN_OR_RAISE(ColumnTable in, olap::MakeColumnTable(batches_, 0));
  // worst-case room per batch: a batch may pass entirely
  ARROW_ASSIGN_OR_RAISE(auto out, olap::AllocateOutputs(in.rows, olap::WidthOf(dtype)));
  std::vector<uint64_t> kept(in.rows.size());
  const uint64_t threshold = 1ull << 30;  // the low bytes serve the 32-bit types too
  // batch b runs on rank b mod G (FilterDpu deals its batches to the DPUs in rounds, filter_dpu.cc:104-169); no communication
  const uint32_t G = system_.nr_ranks();
  ARROW_RETURN_NOT_OK(system_.OnRanks([&](uint32_t r) {
    std::vector<const void*> ptrs;
    std::vector<uint64_t> rows, k;
    std::vector<void*> outs;
    for (size_t b = r; b < in.ptrs.size(); b += G) {
      ptrs.push_back(in.ptrs[b]);
      rows.push_back(in.rows[b]);
      outs.push_back(out.ptrs[b]);
    }
    k.resize(ptrs.size());
    ARROW_RETURN_NOT_OK(system_.Check(olap_filter_lt(system_.ctx(r), dtype, ptrs.data(), rows.data(), ptrs.size(), OLAP_MEM_HOST, &threshold,
                                                     outs.data(), OLAP_MEM_HOST, k.data(), nullptr), r));
    for (size_t i = 0; i < k.size(); ++i) kept[r + i * G] = k[i];
    return arrow::Status::OK();
  }));
  auto build = timers_->New("build-result");
  build->Start();
  arrow::ArrayVector chunks;
  for (size_t b = 0; b < kept.size(); ++b) chunks.push_back(olap::WrapBuffer(type, (int64_t)kept[b], out.buffers[b]));
  build->Stop();
  ARROW_RETURN_NOT_OK(Finish());
  return arrow::ChunkedArray::Make(chunks, type);
}

arrow::Result<uint64_t> FilterGpu::Run() {
  ARROW_ASSIGN_OR_RAISE(auto chunks, GetResult());
  return (uint64_t)chunks->length();
}

}  // namespace filter

// -------------------------------------------------------------------- take
namespace take {

arrow::Result<std::shared_ptr<arrow::Table>> TakeGpu::Run() {
  ARROW_RETURN_NOT_OK(Prepared());
  if (batches_.empty() || batches_.size() != indices_batches_.size())
    return arrow::Status::Invalid("one index batch per data batch expected");
  const auto type = batches_.front()->column(0)->type();
  ARROW_ASSIGN_OR_RAISE(const int dtype, olap::DtypeOf(*type));
  if (indices_batches_.front()->column(0)->type_id() != arrow::Type::UINT32)
    return arrow::Status::TypeError("uint32 selection indices expected");
  ARROW_ASSIGN_OR_RAISE(ColumnTable data, olap::MakeColumnTable(batches_, 0));
  ARROW_ASSIGN_OR_RAISE(ColumnTable idx, olap::MakeColumnTable(indices_batches_, 0));
  ARROW_ASSIGN_OR_RAISE(auto out, olap::AllocateOutputs(idx.rows, olap::WidthOf(dtype)));
  const uint32_t G = system_.nr_ranks();  // batch b on rank b mod G, no communication
  ARROW_RETURN_NOT_OK(system_.OnRanks([&](uint32_t r) {
    std::vector<const void*> dp;
    std::vector<const uint32_t*> ip;
    std::vector<uint64_t> dr, ir;
    std::vector<void*> op;
    for (size_t b = r; b < data.ptrs.size(); b += G) {
      dp.push_back(data.ptrs[b]);
      dr.push_back(data.rows[b]);
      ip.push_back(static_cast<const uint32_t*>(idx.ptrs[b]));
      ir.push_back(idx.rows[b]);
      op.push_back(out.ptrs[b]);
    }
    return system_.Check(olap_take(system_.ctx(r), dtype, dp.data(), dr.data(), ip.data(), ir.data(), dp.size(), OLAP_MEM_HOST, op.data(),
                                   OLAP_MEM_HOST), r);
  }));
  arrow::RecordBatchVector result;
  for (size_t b = 0; b < idx.rows.size(); ++b) {
    auto column = olap::WrapBuffer(type, (int64_t)idx.rows[b], out.buffers[b]);
    result.push_back(arrow::RecordBatch::Make(batches_.front()->schema(), column->length(), {column}));
  }
  ARROW_RETURN_NOT_OK(Finish());
  return arrow::Table::FromRecordBatches(result);
}

}  // namespace take

// --------------------------------------------------------------------- sum
namespace aggr {

arrow::Result<uint64_t> SumGpu::Run() {
  ARROW_RETURN_NOT_OK(Prepared());
  uint64_t sum = 0;
  if (!batches_.empty()) {
    ARROW_ASSIGN_OR_RAISE(const int dtype, olap::DtypeOf(*batches_.front()->column(0)->type()));
    ARROW_ASSIGN_OR_RAISE(ColumnTable in, olap::MakeColumnTable(batches_, 0));
    const uint32_t G = system_.nr_ranks();  // batch b on rank b mod G; the partial sums are added on the host (aggr_dpu.cc:82-84)
    std::vector<uint64_t> part(G, 0);
    ARROW_RETURN_NOT_OK(system_.OnRanks([&](uint32_t r) {
      std::vector<const void*> ptrs;
      std::vector<uint64_t> rows;
      for (size_t b = r; b < in.ptrs.size(); b += G) {
        ptrs.push_back(in.ptrs[b]);
        rows.push_back(in.rows[b]);
      }
      return system_.Check(olap_sum(system_.ctx(r), dtype, ptrs.data(), rows.data(), ptrs.size(), OLAP_MEM_HOST, &part[r]), r);
    }));
    for (uint64_t p : part) sum += p;
  }
  ARROW_RETURN_NOT_OK(Finish());
  return sum;
}

}  // namespace aggr

// --------------------------------------------------------------- partition
namespace partition {

// Keys are partitioned together with their global row index (the reference's
// selection_indices_vector); the other columns follow by a GPU gather with those row ids,
// which is what Partitioner::TakeKernel does on the DPUs.
arrow::Result<arrow::RecordBatchVector> PartitionGpu::Run() {
  ARROW_RETURN_NOT_OK(Prepared());
  const int key = schema_->GetFieldIndex(key_column_);
  if (key < 0) return arrow::Status::Invalid("no column named ", key_column_);
  ARROW_ASSIGN_OR_RAISE(ColumnTable keys, olap::MakeColumnTable(batches_, key));
  const uint64_t n = keys.total_rows();
  ARROW_ASSIGN_OR_RAISE(auto part, olap::AllocateOutputs({n, n}, 4));  // [0] keys, [1] row ids
  std::vector<uint64_t> offsets(nr_partitions_ + 1);
  ARROW_RETURN_NOT_OK(system_.Check(olap_partition_u32(
      system_.ctx(), reinterpret_cast<const uint32_t* const*>(keys.ptrs.data()), nullptr, keys.rows.data(),
      keys.ptrs.size(), OLAP_MEM_HOST, nr_partitions_, static_cast<uint32_t*>(part.ptrs[0]),
      static_cast<uint32_t*>(part.ptrs[1]), n, OLAP_MEM_HOST, offsets.data())));
  ARROW_ASSIGN_OR_RAISE(auto table, arrow::Table::FromRecordBatches(schema_, batches_));
  arrow::ArrayVector columns(schema_->num_fields());
  for (int c = 0; c < schema_->num_fields(); ++c) {
    if (c == key) {
      columns[c] = olap::WrapBuffer(arrow::uint32(), (int64_t)n, part.buffers[0]);
      continue;
    }
    ARROW_ASSIGN_OR_RAISE(const int dtype, olap::DtypeOf(*schema_->field(c)->type()));
    // the column as one contiguous buffer (a host copy, not a computation), gathered on the GPU
    std::shared_ptr<arrow::Array> flat;
    if (table->column(c)->num_chunks() == 1) {
      flat = table->column(c)->chunk(0);
    } else if (table->column(c)->num_chunks() == 0) {
      ARROW_ASSIGN_OR_RAISE(flat, arrow::MakeEmptyArray(schema_->field(c)->type()));
    } else {
      ARROW_ASSIGN_OR_RAISE(flat, arrow::Concatenate(table->column(c)->chunks()));
    }
    ARROW_ASSIGN_OR_RAISE(auto gathered, olap::AllocateOutputs({n}, olap::WidthOf(dtype)));
    if (n) {
      const void* data = flat->data()->buffers[1]->data() + flat->data()->offset * olap::WidthOf(dtype);
      const uint32_t* ids = static_cast<const uint32_t*>(part.ptrs[1]);
      const uint64_t data_rows = (uint64_t)flat->length();
      ARROW_RETURN_NOT_OK(system_.Check(olap_take(system_.ctx(), dtype, &data, &data_rows, &ids, &n, 1, OLAP_MEM_HOST,
                                                  gathered.ptrs.data(), OLAP_MEM_HOST)));
    }
    columns[c] = olap::WrapBuffer(schema_->field(c)->type(), (int64_t)n, gathered.buffers[0]);
  }
  arrow::RecordBatchVector result;
  for (uint32_t p = 0; p < nr_partitions_; ++p) {
    const int64_t begin = (int64_t)offsets[p], length = (int64_t)(offsets[p + 1] - offsets[p]);
    arrow::ArrayVector slice;
    for (const auto& column : columns) slice.push_back(column->Slice(begin, length));
    result.push_back(arrow::RecordBatch::Make(schema_, length, slice));
  }
  ARROW_RETURN_NOT_OK(Finish());
  return result;
}

}  // namespace partition

// -------------------------------------------------------------------- join
namespace join {

namespace {

// One column of a table as a single contiguous array (a host copy, not a computation).
arrow::Result<std::shared_ptr<arrow::Array>> Flatten(const arrow::RecordBatchVector& batches, int column,
                                                     const std::shared_ptr<arrow::DataType>& type) {
  arrow::ArrayVector chunks;
  for (const auto& b : batches) chunks.push_back(b->column(column));
  if (chunks.empty()) return arrow::MakeEmptyArray(type);
  if (chunks.size() == 1) return chunks.front();
  return arrow::Concatenate(chunks);
}

}  // namespace

// Gathers column `column` of `batches` at the global row ids `ids` on the GPU (olap_take).
arrow::Result<std::shared_ptr<arrow::Array>> JoinGpu::Gather(const arrow::RecordBatchVector& batches, int column,
                                                             const std::shared_ptr<arrow::Field>& field,
                                                             const uint32_t* ids, uint64_t n) {
  ARROW_ASSIGN_OR_RAISE(const int dtype, olap::DtypeOf(*field->type()));
  ARROW_ASSIGN_OR_RAISE(auto flat, Flatten(batches, column, field->type()));
  ARROW_ASSIGN_OR_RAISE(auto out, olap::AllocateOutputs({n}, olap::WidthOf(dtype)));
  if (n) {
    const void* data = flat->data()->buffers[1]->data() + flat->data()->offset * olap::WidthOf(dtype);
    const uint64_t data_rows = (uint64_t)flat->length();
    ARROW_RETURN_NOT_OK(system_.Check(olap_take(system_.ctx(), dtype, &data, &data_rows, &ids, &n, 1, OLAP_MEM_HOST,
                                                out.ptrs.data(), OLAP_MEM_HOST)));
  }
  return olap::WrapBuffer(field->type(), (int64_t)n, out.buffers[0]);
}

// 64-bit keys (the reference's HT_64BIT_KEYS table, dpu/shared/hashtable/hashtable.h:13-18) through olap_join_u64 on rank 0.
// {fk: 64, y: 32} x {pk: 64, x: 32}: the payloads travel with their keys.  Any other schema (64-bit payload columns, several
// columns per side): the join returns selection vectors and every non-key column is gathered through them, as the 32-bit path
// does -- the reference's hash_probe + take per column (host/join/join_dpu.cc:325-363).
arrow::Result<std::shared_ptr<arrow::Table>> JoinGpu::Run64(int fk, int pk) {
  auto is32 = [](const arrow::Field& f) { return f.type()->id() == arrow::Type::UINT32 || f.type()->id() == arrow::Type::INT32; };
  auto outer = timers_->Get("outer");
  const bool inline_payloads = left_schema_->num_fields() == 2 && right_schema_->num_fields() == 2 && is32(*left_schema_->field(1 - fk)) &&
                               is32(*right_schema_->field(1 - pk));
  ARROW_ASSIGN_OR_RAISE(ColumnTable lkey, olap::MakeColumnTable(left_batches_, fk));
  ARROW_ASSIGN_OR_RAISE(ColumnTable rkey, olap::MakeColumnTable(right_batches_, pk));
  ColumnTable lpay, rpay;
  if (inline_payloads) {
    ARROW_ASSIGN_OR_RAISE(lpay, olap::MakeColumnTable(left_batches_, 1 - fk));
    ARROW_ASSIGN_OR_RAISE(rpay, olap::MakeColumnTable(right_batches_, 1 - pk));
  }
  const uint64_t n_left = lkey.total_rows();
  ARROW_ASSIGN_OR_RAISE(auto okey, olap::AllocateOutputs({n_left}, 8));
  ARROW_ASSIGN_OR_RAISE(auto opay, olap::AllocateOutputs({n_left, n_left}, 4));
  ARROW_RETURN_NOT_OK(system_.Check(olap_join_set_unmatched(system_.ctx(), drop_unmatched_ ? OLAP_UNMATCHED_DROP : OLAP_UNMATCHED_ERROR)));
  uint64_t out_rows = 0;
  const int rc = olap_join_u64(system_.ctx(), reinterpret_cast<const uint64_t* const*>(lkey.ptrs.data()),
                               inline_payloads ? reinterpret_cast<const uint32_t* const*>(lpay.ptrs.data()) : nullptr, lkey.rows.data(),
                               lkey.ptrs.size(), reinterpret_cast<const uint64_t* const*>(rkey.ptrs.data()),
                               inline_payloads ? reinterpret_cast<const uint32_t* const*>(rpay.ptrs.data()) : nullptr, rkey.rows.data(),
                               rkey.ptrs.size(), OLAP_MEM_HOST, static_cast<uint64_t*>(okey.ptrs[0]), static_cast<uint32_t*>(opay.ptrs[0]),
                               static_cast<uint32_t*>(opay.ptrs[1]), n_left, OLAP_MEM_HOST, &out_rows, &stats_);
  (void)olap_join_set_unmatched(system_.ctx(), OLAP_UNMATCHED_ERROR);
  ARROW_RETURN_NOT_OK(system_.Check(rc));
  // result schema: the left columns, then every right column but the key (join_dpu.cc:314-363)
  arrow::FieldVector fields = left_schema_->fields();
  for (int c = 0; c < right_schema_->num_fields(); ++c)
    if (c != pk) fields.push_back(right_schema_->field(c));
  std::vector<std::shared_ptr<arrow::Array>> columns;
  if (inline_payloads) {
    columns.resize(3);
    columns[fk] = olap::WrapBuffer(left_schema_->field(fk)->type(), (int64_t)out_rows, okey.buffers[0]);
    columns[1 - fk] = olap::WrapBuffer(left_schema_->field(1 - fk)->type(), (int64_t)out_rows, opay.buffers[0]);
    columns[2] = olap::WrapBuffer(right_schema_->field(1 - pk)->type(), (int64_t)out_rows, opay.buffers[1]);
  } else {
    const auto* left_ids = static_cast<const uint32_t*>(opay.ptrs[0]);
    const auto* right_ids = static_cast<const uint32_t*>(opay.ptrs[1]);
    for (int c = 0; c < left_schema_->num_fields(); ++c) {
      if (c == fk) {
        columns.push_back(olap::WrapBuffer(left_schema_->field(fk)->type(), (int64_t)out_rows, okey.buffers[0]));
      } else {
        ARROW_ASSIGN_OR_RAISE(auto col, Gather(left_batches_, c, left_schema_->field(c), left_ids, out_rows));
        columns.push_back(std::move(col));
      }
    }
    for (int c = 0; c < right_schema_->num_fields(); ++c) {
      if (c == pk) continue;
      ARROW_ASSIGN_OR_RAISE(auto col, Gather(right_batches_, c, right_schema_->field(c), right_ids, out_rows));
      columns.push_back(std::move(col));
    }
  }
  outer->Stop();
  ARROW_RETURN_NOT_OK(Finish());
  return arrow::Table::Make(arrow::schema(fields), columns);
}

arrow::Result<std::shared_ptr<arrow::Table>> JoinGpu::Run() {
  ARROW_RETURN_NOT_OK(Prepared());
  auto outer = timers_->New("outer");
  outer->Start();
  const int fk = left_schema_->GetFieldIndex("fk"), pk = right_schema_->GetFieldIndex("pk");
  if (fk < 0 || pk < 0) return arrow::Status::Invalid("left table needs a column 'fk', right table a column 'pk'");
  auto is32 = [](const arrow::Field& f) { return f.type()->id() == arrow::Type::UINT32 || f.type()->id() == arrow::Type::INT32; };
  auto is64 = [](const arrow::Field& f) { return f.type()->id() == arrow::Type::UINT64 || f.type()->id() == arrow::Type::INT64; };
  const bool keys64 = is64(*left_schema_->field(fk)) && is64(*right_schema_->field(pk));
  if (!keys64)
    for (const auto& field : {left_schema_->field(fk), right_schema_->field(pk)})
      if (!is32(*field)) return arrow::Status::TypeError("integer join keys of one width (32 or 64 bits) expected, got ", field->ToString());
  if (keys64) return Run64(fk, pk);
  ARROW_ASSIGN_OR_RAISE(ColumnTable lkey, olap::MakeColumnTable(left_batches_, fk));
  ARROW_ASSIGN_OR_RAISE(ColumnTable rkey, olap::MakeColumnTable(right_batches_, pk));
  const uint64_t n_left = lkey.total_rows();
  auto u32pp = [](const ColumnTable& t) { return reinterpret_cast<const uint32_t* const*>(t.ptrs.data()); };
  // result schema: the left columns, then every right column but the key (join_dpu.cc:314-363)
  arrow::FieldVector fields = left_schema_->fields();
  for (int c = 0; c < right_schema_->num_fields(); ++c)
    if (c != pk) fields.push_back(right_schema_->field(c));
  std::vector<std::shared_ptr<arrow::Array>> columns;
  uint64_t out_rows = 0;
  for (uint32_t r = 0; r < system_.nr_ranks(); ++r)
    ARROW_RETURN_NOT_OK(system_.Check(olap_join_set_unmatched(system_.ctx(r), drop_unmatched_ ? OLAP_UNMATCHED_DROP : OLAP_UNMATCHED_ERROR), r));
  struct Restore {
    olap::Device& dev;
    ~Restore() {
      for (uint32_t r = 0; r < dev.nr_ranks(); ++r) (void)olap_join_set_unmatched(dev.ctx(r), OLAP_UNMATCHED_ERROR);
    }
  } restore{system_};
  ARROW_ASSIGN_OR_RAISE(auto out, olap::AllocateOutputs({n_left, n_left, n_left}, 4));
  auto* o0 = static_cast<uint32_t*>(out.ptrs[0]);
  auto* o1 = static_cast<uint32_t*>(out.ptrs[1]);
  auto* o2 = static_cast<uint32_t*>(out.ptrs[2]);
  const bool inline_payloads = left_schema_->num_fields() == 2 && right_schema_->num_fields() == 2 &&
                               is32(*left_schema_->field(1 - fk)) && is32(*right_schema_->field(1 - pk));
  if (inline_payloads) {
    // the benchmark schema {fk, y} x {pk, x}: one 32-bit payload per side travels with its key
    const int ly = 1 - fk, rx = 1 - pk;
    ARROW_ASSIGN_OR_RAISE(ColumnTable lpay, olap::MakeColumnTable(left_batches_, ly));
    ARROW_ASSIGN_OR_RAISE(ColumnTable rpay, olap::MakeColumnTable(right_batches_, rx));
    // across all ranks of the device set (one rank: the plain olap_join_u32): owner split, exchange, per-owner join
    ARROW_RETURN_NOT_OK(system_.CheckGroup(olap_group_join_u32(system_.group(), u32pp(lkey), u32pp(lpay), lkey.rows.data(), lkey.ptrs.size(),
                                                               u32pp(rkey), u32pp(rpay), rkey.rows.data(), rkey.ptrs.size(), OLAP_MEM_HOST,
                                                               o0, o1, o2, n_left, &out_rows, nullptr, &stats_)));
    columns.resize(3);
    columns[fk] = olap::WrapBuffer(left_schema_->field(fk)->type(), (int64_t)out_rows, out.buffers[0]);
    columns[ly] = olap::WrapBuffer(left_schema_->field(ly)->type(), (int64_t)out_rows, out.buffers[1]);
    columns[2] = olap::WrapBuffer(right_schema_->field(rx)->type(), (int64_t)out_rows, out.buffers[2]);
  } else {
    // any other schema: the join returns selection vectors (left row id, right row id of the
    // match) and every non-key column is gathered through them -- the reference's
    // hash_probe + take per column (join_dpu.cc:325-363)
    // (this form runs on rank 0 alone: the multi-device join carries one 32-bit payload per side, and selection
    // vectors are row numbers of the WHOLE table, which the per-rank shares do not keep)
    ARROW_RETURN_NOT_OK(system_.Check(olap_join_u32(system_.ctx(), u32pp(lkey), nullptr, lkey.rows.data(), lkey.ptrs.size(),
                                                    u32pp(rkey), nullptr, rkey.rows.data(), rkey.ptrs.size(), OLAP_MEM_HOST, o0,
                                                    o1, o2, n_left, OLAP_MEM_HOST, &out_rows, &stats_)));
    for (int c = 0; c < left_schema_->num_fields(); ++c) {
      if (c == fk) {
        columns.push_back(olap::WrapBuffer(left_schema_->field(fk)->type(), (int64_t)out_rows, out.buffers[0]));
      } else {
        ARROW_ASSIGN_OR_RAISE(auto col, Gather(left_batches_, c, left_schema_->field(c), o1, out_rows));
        columns.push_back(std::move(col));
      }
    }
    for (int c = 0; c < right_schema_->num_fields(); ++c) {
      if (c == pk) continue;
      ARROW_ASSIGN_OR_RAISE(auto col, Gather(right_batches_, c, right_schema_->field(c), o2, out_rows));
      columns.push_back(std::move(col));
    }
  }
  outer->Stop();
  ARROW_RETURN_NOT_OK(Finish());
  return arrow::Table::Make(arrow::schema(fields), columns);
}

}  // namespace join

}  // namespace upmemeval
