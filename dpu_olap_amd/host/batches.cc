#include "batches.h"

#include <arrow/util/thread_pool.h>

namespace {

uint64_t mix64(uint64_t z) {  // splitmix64 finaliser -- the same function as csrc/common.h
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Bounds {
  bool set = false;
  uint64_t lo = 0, span = 0;
};

template <typename C>
std::shared_ptr<arrow::Array> Uniform(const std::shared_ptr<arrow::DataType>& type, int64_t n, uint64_t seed, uint64_t stream,
                                      const Bounds& b) {
  auto buf = arrow::AllocateBuffer(n * (int64_t)sizeof(C)).ValueOrDie();
  C* v = reinterpret_cast<C*>(buf->mutable_data());
  const uint64_t base = mix64(seed ^ (0xD1B54A32D192ED03ull * (stream + 1)));
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t r = mix64(base + (uint64_t)i);
    v[i] = b.set ? (C)(b.lo + (((r >> 32) * b.span) >> 32))      // multiply-high range reduction (span < 2^32)
                 : (C)(sizeof(C) == 8 ? r : (r >> 32));
  }
  return arrow::MakeArray(arrow::ArrayData::Make(type, n, {nullptr, std::shared_ptr<arrow::Buffer>(std::move(buf))}));
}

}  // namespace

namespace arrow {
namespace random {

std::shared_ptr<Array> RandomArrayGenerator::ArrayOf(const Field& field, int64_t length) {
  Bounds b;
  if (auto md = field.metadata()) {
    auto lo = md->Get("min"), hi = md->Get("max");
    if (lo.ok() && hi.ok()) b = {true, std::stoull(*lo), std::stoull(*hi) - std::stoull(*lo) + 1};
  }
  const uint64_t stream = next_stream_++;
  switch (field.type()->id()) {
    case Type::UINT32: return Uniform<uint32_t>(field.type(), length, seed_, stream, b);
    case Type::INT32: return Uniform<int32_t>(field.type(), length, seed_, stream, b);
    case Type::UINT64: return Uniform<uint64_t>(field.type(), length, seed_, stream, b);
    case Type::INT64: return Uniform<int64_t>(field.type(), length, seed_, stream, b);
    default: std::abort();  // the hot path only has fixed-width integers
  }
}

std::shared_ptr<RecordBatch> RandomArrayGenerator::BatchOf(const FieldVector& fields, int64_t length) {
  ArrayVector columns;
  for (const auto& f : fields) columns.push_back(ArrayOf(*f, length));
  return RecordBatch::Make(schema(fields), length, columns);
}

}  // namespace random
}  // namespace arrow

namespace upmemeval {
namespace generator {

arrow::RecordBatchVector MakeRandomRecordBatches(arrow::random::RandomArrayGenerator& g,
                                                 const std::shared_ptr<arrow::Schema>& schema, int num_batches, int batch_size) {
  arrow::RecordBatchVector batches;
  while ((int)batches.size() < num_batches) batches.push_back(g.BatchOf(schema->fields(), batch_size));
  return batches;
}

arrow::Result<arrow::ArrayVector> MakeIndexColumn(int num_batches, int batch_size) {
  arrow::ArrayVector columns;
  uint32_t next = 0;
  for (int b = 0; b < num_batches; ++b) {
    ARROW_ASSIGN_OR_RAISE(auto buf, arrow::AllocateBuffer((int64_t)batch_size * 4));
    for (uint32_t *p = reinterpret_cast<uint32_t*>(buf->mutable_data()), *e = p + batch_size; p != e; ++p) *p = next++;
    columns.push_back(arrow::MakeArray(
        arrow::ArrayData::Make(arrow::uint32(), batch_size, {nullptr, std::shared_ptr<arrow::Buffer>(std::move(buf))})));
  }
  return columns;
}

arrow::Result<arrow::ArrayVector> MakeForeignKeyColumn(arrow::random::RandomArrayGenerator& g, uint32_t pk_batch_size,
                                                       int32_t num_batches, int32_t batch_size) {
  arrow::ArrayVector columns;
  for (uint64_t i = 0; i < (uint64_t)num_batches; ++i) {
    auto range = arrow::key_value_metadata({"min", "max"}, {std::to_string(i * pk_batch_size), std::to_string((i + 1) * pk_batch_size - 1)});
    columns.push_back(g.ArrayOf(*arrow::field("fk", arrow::uint32(), /*nullable=*/false, range), batch_size));
  }
  return columns;
}

arrow::RecordBatchVector AddColumn(const std::string& name, const arrow::RecordBatchVector& batches, arrow::ArrayVector column) {
  arrow::RecordBatchVector out;
  for (size_t i = 0; i < batches.size(); ++i) out.push_back(batches[i]->AddColumn(0, name, column[i]).ValueOrDie());
  return out;
}

std::vector<arrow::compute::ExecBatch> ToExecBatches(const arrow::RecordBatchVector& batches) {
  std::vector<arrow::compute::ExecBatch> out;
  for (const auto& b : batches) out.emplace_back(*b);
  return out;
}

ExecBatchGenerator MakeAsyncGenerator(std::vector<arrow::compute::ExecBatch> batches, bool parallel) {
  std::vector<std::optional<arrow::compute::ExecBatch>> items(std::make_move_iterator(batches.begin()),
                                                              std::make_move_iterator(batches.end()));
  if (!parallel) return arrow::MakeVectorGenerator(std::move(items));
  // batches "arrive" from a background thread and continue on the CPU pool, like a scan would deliver them
  auto* pool = arrow::internal::GetCpuThreadPool();
  auto background = arrow::MakeBackgroundGenerator(arrow::MakeVectorIterator(std::move(items)), pool).ValueOrDie();
  return arrow::MakeTransferredGenerator(std::move(background), pool);
}

std::shared_ptr<arrow::RecordBatch> RecordBatchOf(std::vector<std::string> names, std::vector<std::shared_ptr<arrow::Array>> data) {
  arrow::FieldVector fields;
  for (size_t i = 0; i < names.size(); ++i) fields.push_back(arrow::field(names[i], data[i]->type()));
  return arrow::RecordBatch::Make(arrow::schema(fields), data.front()->length(), data);
}

}  // namespace generator
}  // namespace upmemeval
