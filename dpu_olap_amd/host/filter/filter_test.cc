// Gpu-vs-Native equality tests for the filter operator; the cases are those of the
// reference's host/filter/filter_test.cc (SimpleTest :24-31, ResultTest :33-61,
// LongerTest :63-78) with FilterGpu in the place of FilterDpu.
#include <arrow/api.h>

#include "filter/filter_gpu.h"
#include "filter/filter_native.h"
#include "generator/generator.h"
#include "testing/minitest.h"

using namespace upmemeval::generator;
using namespace upmemeval::filter;

namespace {
template <typename F>
arrow::Result<uint64_t> PrepareAndRun(F&& f) {
  ARROW_RETURN_NOT_OK(f.Prepare());
  return f.Run();
}
}  // namespace

TEST(FilterTest, SimpleTest) {
  auto rb = RecordBatchOf({"v"}, {ArrayOf<arrow::UInt32Type>({0, 2, 3, 8, 9})});
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  auto gpu = PrepareAndRun(FilterGpu{*dev, {rb}}).ValueOrDie();
  auto native = PrepareAndRun(FilterNative{rb->schema(), {rb}}).ValueOrDie();
  EXPECT_EQ(native, gpu);
  EXPECT_EQ(gpu, 5u);
}

TEST(FilterTest, ResultTest) {
  arrow::UInt32Builder b;
  for (uint32_t i = 0; i < 4096; ++i) {
    const bool small = i == 5 || i == 8 || i == 9 || i == 100 || i == 270;
    EXPECT_TRUE(b.Append(small ? i : i + (1u << 30)).ok());
  }
  auto rb = RecordBatchOf({"v"}, {b.Finish().ValueOrDie()});
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  FilterGpu gpu{*dev, {rb}};
  EXPECT_TRUE(gpu.Prepare().ok());
  auto gpu_result = gpu.GetResult().ValueOrDie();
  FilterNative native{rb->schema(), {rb}};
  EXPECT_TRUE(native.Prepare().ok());
  auto native_result = native.GetResult().ValueOrDie()->column(0);
  EXPECT_TRUE(gpu_result->Equals(native_result));
  EXPECT_EQ(gpu_result->length(), 5);
}

TEST(FilterTest, LongerTest) {
  auto schema = arrow::schema({arrow::field("v", arrow::uint32(), /*nullable=*/false)});
  arrow::random::RandomArrayGenerator rng(42);
  auto batches = MakeRandomRecordBatches(rng, schema, 1, 1 << 16);
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  FilterGpu gpu{*dev, batches};
  EXPECT_TRUE(gpu.Prepare().ok());
  auto gpu_result = gpu.GetResult().ValueOrDie();
  FilterNative native{schema, batches};
  EXPECT_TRUE(native.Prepare().ok());
  auto native_result = native.GetResult().ValueOrDie()->column(0);
  EXPECT_TRUE(gpu_result->Equals(native_result));
  auto timers = gpu.Timers();
  EXPECT_TRUE(timers != nullptr && timers->get().count("gpu-work") == 1);
}

TEST(FilterTest, ManyBatchesKeepBatchOrder) {
  auto schema = arrow::schema({arrow::field("v", arrow::uint32(), false)});
  arrow::random::RandomArrayGenerator rng(7);
  auto batches = MakeRandomRecordBatches(rng, schema, 64, 10007);
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  FilterGpu gpu{*dev, batches};
  EXPECT_TRUE(gpu.Prepare().ok());
  auto res = gpu.GetResult().ValueOrDie();
  EXPECT_EQ(res->num_chunks(), 64);
  for (int b = 0; b < 64; ++b) {  // per batch against Arrow, in batch order
    FilterNative native{schema, {batches[b]}};
    EXPECT_TRUE(native.Prepare().ok());
    auto expect = native.GetResult().ValueOrDie()->column(0);
    EXPECT_TRUE(arrow::ChunkedArray(res->chunk(b)).Equals(*expect));
  }
}
