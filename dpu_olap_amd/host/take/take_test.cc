// Take operator tests; cases of the reference's host/take/take_test.cc (SimpleTestNative
// :24-35, SimpleTestDpu :37-49, LargeTest :51-72).
#include <arrow/api.h>

#include "generator/generator.h"
#include "take/take_gpu.h"
#include "take/take_native.h"
#include "testing/minitest.h"

using namespace upmemeval::generator;
using namespace upmemeval::take;

namespace {
template <typename T>
arrow::Result<std::shared_ptr<arrow::Table>> PrepareAndRun(T&& t) {
  ARROW_RETURN_NOT_OK(t.Prepare());
  return t.Run();
}

void ExpectZeroTwoNine(const std::shared_ptr<arrow::Table>& result) {
  auto arr = std::dynamic_pointer_cast<arrow::UInt32Array>(result->column(0)->chunk(0));
  EXPECT_EQ(arr->length(), 3);
  EXPECT_EQ(arr->Value(0), 0u);
  EXPECT_EQ(arr->Value(1), 2u);
  EXPECT_EQ(arr->Value(2), 9u);
}
}  // namespace

TEST(TakeTest, SimpleTestNative) {
  auto rb = RecordBatchOf({"v"}, {ArrayOf<arrow::UInt32Type>({0, 2, 3, 8, 9})});
  auto idx = RecordBatchOf({"i"}, {ArrayOf<arrow::UInt32Type>({0, 1, 4})});
  ExpectZeroTwoNine(PrepareAndRun(TakeNative{rb->schema(), {rb}, {idx}}).ValueOrDie());
}

TEST(TakeTest, SimpleTestGpu) {
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  auto rb = RecordBatchOf({"v"}, {ArrayOf<arrow::UInt32Type>({0, 2, 3, 8, 9})});
  auto idx = RecordBatchOf({"i"}, {ArrayOf<arrow::UInt32Type>({0, 1, 4})});
  ExpectZeroTwoNine(PrepareAndRun(TakeGpu{*dev, {rb}, {idx}}).ValueOrDie());
}

TEST(TakeTest, LargeTest) {
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  const int num_batches = 128, batch_size = 64 << 10, indices_batch_size = 8 << 10;
  arrow::random::RandomArrayGenerator rng(42);
  auto schema = arrow::schema({arrow::field("v", arrow::uint32(), false)});
  auto batches = MakeRandomRecordBatches(rng, schema, num_batches, batch_size);
  auto md = arrow::key_value_metadata({{"min", "0"}, {"max", std::to_string(batch_size - 1)}});
  auto idx_schema = arrow::schema({arrow::field("i", arrow::uint32(), false, md)});
  auto idx_batches = MakeRandomRecordBatches(rng, idx_schema, num_batches, indices_batch_size);
  auto native = PrepareAndRun(TakeNative{schema, batches, idx_batches}).ValueOrDie();
  auto gpu = PrepareAndRun(TakeGpu{*dev, batches, idx_batches}).ValueOrDie();
  EXPECT_TRUE(native->Equals(*gpu));
}
