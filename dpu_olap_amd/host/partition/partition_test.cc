// Partition operator tests; the cases the reference holds but skips
// (host/partition/partition_test.cc SimpleTest :21-58, LargeTest :60-92).
#include <arrow/api.h>
#include <arrow/compute/api.h>

#include <cmath>

#include "generator/generator.h"
#include "partition/partition_gpu.h"
#include "testing/minitest.h"

using namespace upmemeval::generator;
using namespace upmemeval::partition;

TEST(PartitionTest, SimpleTest) {
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  using U = arrow::UInt32Type;
  auto batch = RecordBatchOf({"pk", "x"}, {ArrayOf<U>({0, 2, 3, 8}), ArrayOf<U>({100, 101, 102, 103})});
  PartitionGpu part(*dev, batch->schema(), {batch}, 2, "pk");
  EXPECT_TRUE(part.Prepare().ok());
  auto parts = part.Run().ValueOrDie();
  EXPECT_EQ(parts.size(), 2u);
  EXPECT_EQ(parts[0]->num_rows(), 3);
  EXPECT_EQ(parts[1]->num_rows(), 1);
  auto table = arrow::Table::FromRecordBatches(parts).ValueOrDie();
  EXPECT_EQ(table->num_rows(), 4);
  EXPECT_TRUE(arrow::compute::Sum(table->GetColumnByName("pk")).ValueOrDie().scalar()->Equals(arrow::UInt64Scalar(13)));
  EXPECT_TRUE(arrow::compute::Sum(table->GetColumnByName("x")).ValueOrDie().scalar()->Equals(arrow::UInt64Scalar(406)));
}

TEST(PartitionTest, LargeTest) {
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  const int nr_partitions = 32, num_batches = 128, batch_size = 64 << 10;
  const double expect = 1.0 * num_batches * batch_size / nr_partitions;
  arrow::random::RandomArrayGenerator rng(42);
  auto schema = arrow::schema({arrow::field("pk", arrow::uint32(), false), arrow::field("x", arrow::uint32(), false),
                               arrow::field("y", arrow::uint32(), false)});
  auto batches = MakeRandomRecordBatches(rng, schema, num_batches, batch_size);
  PartitionGpu part(*dev, schema, batches, nr_partitions, "pk");
  EXPECT_TRUE(part.Prepare().ok());
  auto parts = part.Run().ValueOrDie();
  EXPECT_EQ(parts.size(), (size_t)nr_partitions);
  int64_t rows = 0;
  uint64_t sum_x = 0, sum_x_in = 0;
  for (auto& p : parts) {
    EXPECT_EQ(p->num_columns(), schema->num_fields());
    EXPECT_GT(p->num_rows(), 0);
    EXPECT_LE(std::abs(p->num_rows() - expect) / expect, 0.1);
    rows += p->num_rows();
    sum_x += std::static_pointer_cast<arrow::UInt64Scalar>(arrow::compute::Sum(p->column(1)).ValueOrDie().scalar())->value;
  }
  for (auto& b : batches)
    sum_x_in += std::static_pointer_cast<arrow::UInt64Scalar>(arrow::compute::Sum(b->column(1)).ValueOrDie().scalar())->value;
  EXPECT_EQ(rows, (int64_t)num_batches * batch_size);
  EXPECT_EQ(sum_x, sum_x_in);  // payload columns travelled with their keys
}
