// generator.h -- synthetic record batches for tests and benchmarks; same entry points as
// the reference's host/generator/generator.h:16-46.  The reference draws from Arrow's
// testing RNG (arrow::random::RandomArrayGenerator, libarrow_testing -- not shipped with
// the Arrow build in this image); this build uses its own counter-based generator with the
// same distributions (uniform integers in [min, max] taken from the field metadata keys
// "min"/"max", as Arrow's generator does).
#pragma once

#include <arrow/api.h>
#include <arrow/compute/api.h>
#include <arrow/util/async_generator.h>

#include <optional>
#include <string>
#include <vector>

namespace arrow {
namespace random {

// Minimal stand-in for the one class of arrow/testing/random.h the harness names.
class RandomArrayGenerator {
 public:
  explicit RandomArrayGenerator(uint64_t seed) : seed_(seed) {}
  std::shared_ptr<Array> ArrayOf(const Field& field, int64_t length);
  std::shared_ptr<RecordBatch> BatchOf(const FieldVector& fields, int64_t length);

 private:
  uint64_t seed_;
  uint64_t stream_ = 0;  // every generated array uses a fresh stream
};

}  // namespace random
}  // namespace arrow

namespace upmemeval {
namespace generator {

arrow::RecordBatchVector MakeRandomRecordBatches(::arrow::random::RandomArrayGenerator& g,
                                                 const std::shared_ptr<arrow::Schema>& schema, int num_batches,
                                                 int batch_size);

arrow::Result<arrow::ArrayVector> MakeIndexColumn(int num_batches, int batch_size);

arrow::RecordBatchVector AddColumn(const std::string& name, const arrow::RecordBatchVector& batches,
                                   arrow::ArrayVector column);

arrow::Result<arrow::ArrayVector> MakeForeignKeyColumn(::arrow::random::RandomArrayGenerator& g, uint32_t pk_batch_size,
                                                       int32_t num_batches, int32_t batch_size);

std::vector<arrow::compute::ExecBatch> ToExecBatches(const arrow::RecordBatchVector& batches);

arrow::AsyncGenerator<std::optional<arrow::compute::ExecBatch>> MakeAsyncGenerator(
    std::vector<arrow::compute::ExecBatch> batches, bool parallel);

template <typename T>
std::shared_ptr<::arrow::Array> ArrayOf(std::vector<typename T::c_type> values) {
  ::arrow::NumericBuilder<T> builder;
  if (!builder.AppendValues(values).ok()) std::abort();
  return builder.Finish().ValueOrDie();
}

std::shared_ptr<::arrow::RecordBatch> RecordBatchOf(std::vector<std::string> names,
                                                    std::vector<std::shared_ptr<arrow::Array>> data);

}  // namespace generator
}  // namespace upmemeval
