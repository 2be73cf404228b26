#include "generator/generator.h"

#include <arrow/util/thread_pool.h>

#include <cstring>

namespace {

inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser, the same as csrc/common.h
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

inline uint64_t gen64(uint64_t seed, uint64_t stream, uint64_t row) {
  return mix64(mix64(seed ^ (0xD1B54A32D192ED03ull * (stream + 1))) + row);
}

template <typename C>
std::shared_ptr<arrow::Array> FillUniform(const std::shared_ptr<arrow::DataType>& type, int64_t length, uint64_t seed,
                                          uint64_t stream, bool bounded, uint64_t lo, uint64_t span) {
  auto buf = arrow::AllocateBuffer(length * (int64_t)sizeof(C)).ValueOrDie();
  C* out = reinterpret_cast<C*>(buf->mutable_data());
  for (int64_t i = 0; i < length; ++i) {
    const uint64_t r = gen64(seed, stream, (uint64_t)i);
    if (!bounded) {
      out[i] = (C)(sizeof(C) == 8 ? r : (r >> 32));
    } else {  // uniform in [lo, lo + span - 1] by multiply-high (span < 2^32 here)
      out[i] = (C)(lo + (((r >> 32) * span) >> 32));
    }
  }
  return arrow::MakeArray(arrow::ArrayData::Make(type, length, {nullptr, std::shared_ptr<arrow::Buffer>(std::move(buf))}));
}

}  // namespace

namespace arrow {
namespace random {

std::shared_ptr<Array> RandomArrayGenerator::ArrayOf(const Field& field, int64_t length) {
  bool bounded = false;
  uint64_t lo = 0, span = 0;
  if (field.metadata()) {  // "min"/"max" metadata bound the values (arrow/testing/random.cc semantics)
    auto mn = field.metadata()->Get("min");
    auto mx = field.metadata()->Get("max");
    if (mn.ok() && mx.ok()) {
      lo = std::stoull(*mn);
      span = std::stoull(*mx) - lo + 1;
      bounded = true;
    }
  }
  const uint64_t stream = stream_++;
  switch (field.type()->id()) {
    case Type::UINT32: return FillUniform<uint32_t>(field.type(), length, seed_, stream, bounded, lo, span);
    case Type::INT32: return FillUniform<int32_t>(field.type(), length, seed_, stream, bounded, lo, span);
    case Type::UINT64: return FillUniform<uint64_t>(field.type(), length, seed_, stream, bounded, lo, span);
    case Type::INT64: return FillUniform<int64_t>(field.type(), length, seed_, stream, bounded, lo, span);
    default: std::abort();
  }
}

std::shared_ptr<RecordBatch> RandomArrayGenerator::BatchOf(const FieldVector& fields, int64_t length) {
  ArrayVector cols;
  for (const auto& f : fields) cols.push_back(ArrayOf(*f, length));
  return RecordBatch::Make(schema(fields), length, cols);
}

}  // namespace random
}  // namespace arrow

namespace upmemeval {
namespace generator {

arrow::RecordBatchVector MakeRandomRecordBatches(::arrow::random::RandomArrayGenerator& g,
                                                 const std::shared_ptr<arrow::Schema>& schema, int num_batches,
                                                 int batch_size) {
  arrow::RecordBatchVector out;
  out.reserve(num_batches);
  for (int i = 0; i < num_batches; ++i) out.push_back(g.BatchOf(schema->fields(), batch_size));
  return out;
}

// pk = 0, 1, 2, ... across batches (reference generator.cc:59-71)
arrow::Result<arrow::ArrayVector> MakeIndexColumn(int num_batches, int batch_size) {
  arrow::ArrayVector out;
  uint32_t next = 0;
  for (int b = 0; b < num_batches; ++b) {
    ARROW_ASSIGN_OR_RAISE(auto buf, arrow::AllocateBuffer((int64_t)batch_size * 4));
    uint32_t* p = reinterpret_cast<uint32_t*>(buf->mutable_data());
    for (int i = 0; i < batch_size; ++i) p[i] = next++;
    out.push_back(arrow::MakeArray(
        arrow::ArrayData::Make(arrow::uint32(), batch_size, {nullptr, std::shared_ptr<arrow::Buffer>(std::move(buf))})));
  }
  return out;
}

// the new column becomes column 0 (reference generator.cc:32-44)
arrow::RecordBatchVector AddColumn(const std::string& name, const arrow::RecordBatchVector& batches,
                                   arrow::ArrayVector column) {
  arrow::RecordBatchVector out;
  for (size_t i = 0; i < batches.size(); ++i) out.push_back(batches[i]->AddColumn(0, name, column[i]).ValueOrDie());
  return out;
}

// fk of batch i uniform in [i * R, (i + 1) * R - 1] (reference generator.cc:46-57)
arrow::Result<arrow::ArrayVector> MakeForeignKeyColumn(::arrow::random::RandomArrayGenerator& g, uint32_t pk_batch_size,
                                                       int32_t num_batches, int32_t batch_size) {
  arrow::ArrayVector out;
  for (uint32_t i = 0; i < (uint32_t)num_batches; ++i) {
    auto md = arrow::key_value_metadata({{"min", std::to_string((uint64_t)i * pk_batch_size)},
                                         {"max", std::to_string((uint64_t)(i + 1) * pk_batch_size - 1)}});
    out.push_back(g.ArrayOf(*arrow::field("fk", arrow::uint32(), false, md), batch_size));
  }
  return out;
}

std::vector<arrow::compute::ExecBatch> ToExecBatches(const arrow::RecordBatchVector& batches) {
  std::vector<arrow::compute::ExecBatch> out;
  out.reserve(batches.size());
  for (const auto& b : batches) out.emplace_back(*b);
  return out;
}

arrow::AsyncGenerator<std::optional<arrow::compute::ExecBatch>> MakeAsyncGenerator(
    std::vector<arrow::compute::ExecBatch> batches, bool parallel) {
  std::vector<std::optional<arrow::compute::ExecBatch>> opt;
  opt.reserve(batches.size());
  for (auto& b : batches) opt.emplace_back(std::move(b));
  if (!parallel) return arrow::MakeVectorGenerator(std::move(opt));
  // batches "arrive" from a background thread and continue on the CPU pool, like a scan
  auto gen = arrow::MakeBackgroundGenerator(arrow::MakeVectorIterator(std::move(opt)),
                                            ::arrow::internal::GetCpuThreadPool())
                 .ValueOrDie();
  return arrow::MakeTransferredGenerator(std::move(gen), ::arrow::internal::GetCpuThreadPool());
}

std::shared_ptr<::arrow::RecordBatch> RecordBatchOf(std::vector<std::string> names,
                                                    std::vector<std::shared_ptr<arrow::Array>> data) {
  arrow::FieldVector fields;
  for (size_t i = 0; i < names.size(); ++i) fields.push_back(arrow::field(names[i], data[i]->type()));
  return arrow::RecordBatch::Make(arrow::schema(fields), data[0]->length(), data);
}

}  // namespace generator
}  // namespace upmemeval
