// Sum operator tests; cases of the reference's host/aggr/aggr_test.cc (:24-49).
#include <arrow/api.h>

#include "aggr/aggr_gpu.h"
#include "aggr/aggr_native.h"
#include "generator/generator.h"
#include "testing/minitest.h"

using namespace upmemeval::generator;
using namespace upmemeval::aggr;

namespace {
template <typename A>
arrow::Result<uint64_t> PrepareAndRun(A&& a) {
  ARROW_RETURN_NOT_OK(a.Prepare());
  return a.Run();
}
}  // namespace

TEST(SumTest, SimpleTestNative) {
  auto rb = RecordBatchOf({"v"}, {ArrayOf<arrow::UInt32Type>({0, 2, 3, 8, 9})});
  EXPECT_EQ(PrepareAndRun(AggrNative<arrow::UInt64Array>{rb->schema(), {rb}, "sum"}).ValueOrDie(), 22u);
}

TEST(SumTest, SimpleTestGpu) {
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  auto rb = RecordBatchOf({"v"}, {ArrayOf<arrow::UInt32Type>({0, 2, 3, 8, 9})});
  EXPECT_EQ(PrepareAndRun(SumGpu{*dev, {rb}}).ValueOrDie(), 22u);
}

TEST(SumTest, LargeTest) {
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  arrow::random::RandomArrayGenerator rng(42);
  auto schema = arrow::schema({arrow::field("v", arrow::uint32(), false)});
  auto batches = MakeRandomRecordBatches(rng, schema, 128, 64 << 10);
  auto native = PrepareAndRun(AggrNative<arrow::UInt64Array>{schema, batches, "sum"}).ValueOrDie();
  auto gpu = PrepareAndRun(SumGpu{*dev, batches}).ValueOrDie();
  EXPECT_EQ(native, gpu);
}

TEST(SumTest, Int64Column) {
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  arrow::random::RandomArrayGenerator rng(3);
  auto schema = arrow::schema({arrow::field("v", arrow::int64(), false)});
  auto batches = MakeRandomRecordBatches(rng, schema, 16, 100003);
  auto native = PrepareAndRun(AggrNative<arrow::Int64Array>{schema, batches, "sum"}).ValueOrDie();
  auto gpu = PrepareAndRun(SumGpu{*dev, batches}).ValueOrDie();
  EXPECT_EQ((uint64_t)native, gpu);  // both wrap mod 2^64
}
