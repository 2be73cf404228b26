// batches.h -- synthetic record batches for the tests and benchmarks.  Entry points named
// after the reference's generator (host/generator/generator.h:16-46) so its fixtures read
// the same; the values come from this project's counter-based generator (the reference
// draws from libarrow_testing's RandomArrayGenerator, which the Arrow build in this image
// does not ship) with the same distributions: uniform over the full type range, or over
// [min, max] when the field carries "min"/"max" metadata.
#pragma once

#include <arrow/api.h>
#include <arrow/compute/api.h>
#include <arrow/util/async_generator.h>

#include <optional>
#include <string>
#include <vector>

namespace arrow {
namespace random {

// The one class of arrow/testing/random.h the harness names.
class RandomArrayGenerator {
 public:
  explicit RandomArrayGenerator(uint64_t seed) : seed_(seed) {}
  std::shared_ptr<Array> ArrayOf(const Field& field, int64_t length);  // a fresh stream per array
  std::shared_ptr<RecordBatch> BatchOf(const FieldVector& fields, int64_t length);

 private:
  uint64_t seed_, next_stream_ = 0;
};

}  // namespace random
}  // namespace arrow

namespace upmemeval {
namespace generator {

using ExecBatchGenerator = arrow::AsyncGenerator<std::optional<arrow::compute::ExecBatch>>;

arrow::RecordBatchVector MakeRandomRecordBatches(arrow::random::RandomArrayGenerator& g,
                                                 const std::shared_ptr<arrow::Schema>& schema, int num_batches, int batch_size);
// 0, 1, 2, ... continuing across batches (the primary key of the join benchmark)
arrow::Result<arrow::ArrayVector> MakeIndexColumn(int num_batches, int batch_size);
// batch i uniform in [i * pk_batch_size, (i + 1) * pk_batch_size - 1]
arrow::Result<arrow::ArrayVector> MakeForeignKeyColumn(arrow::random::RandomArrayGenerator& g, uint32_t pk_batch_size,
                                                       int32_t num_batches, int32_t batch_size);
// prepends `column[i]` as column 0 named `name` to batch i
arrow::RecordBatchVector AddColumn(const std::string& name, const arrow::RecordBatchVector& batches, arrow::ArrayVector column);

std::vector<arrow::compute::ExecBatch> ToExecBatches(const arrow::RecordBatchVector& batches);
ExecBatchGenerator MakeAsyncGenerator(std::vector<arrow::compute::ExecBatch> batches, bool parallel);

template <typename T>
std::shared_ptr<arrow::Array> ArrayOf(std::vector<typename T::c_type> values) {
  arrow::NumericBuilder<T> builder;
  if (!builder.AppendValues(values).ok()) std::abort();
  return builder.Finish().ValueOrDie();
}

std::shared_ptr<arrow::RecordBatch> RecordBatchOf(std::vector<std::string> names, std::vector<std::shared_ptr<arrow::Array>> data);

}  // namespace generator
}  // namespace upmemeval
