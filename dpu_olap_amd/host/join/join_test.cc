// Join operator tests; cases of the reference's host/join/join_test.cc (SimpleTest :40-80,
// LargeTest :82-121): Gpu result == Native (Arrow hash join) result after sorting.
#include <arrow/api.h>
#include <arrow/compute/api.h>

#include "generator/generator.h"
#include "join/join_gpu.h"
#include "join/join_native.h"
#include "testing/minitest.h"

using namespace upmemeval::generator;
using namespace upmemeval::join;

namespace {
template <typename J>
arrow::Result<std::shared_ptr<arrow::Table>> PrepareAndRun(J&& j) {
  ARROW_RETURN_NOT_OK(j.Prepare());
  return j.Run();
}

// test-side helper: canonical row order (Arrow sort_indices + take), as join_test.cc:27-38
std::shared_ptr<arrow::Table> Sorted(const std::shared_ptr<arrow::Table>& t, std::vector<std::string> keys) {
  arrow::compute::SortOptions opts;
  for (auto& k : keys) opts.sort_keys.emplace_back(arrow::FieldRef(k));
  auto idx = arrow::compute::SortIndices(t, opts).ValueOrDie();
  return arrow::compute::Take(t, idx).ValueOrDie().table()->CombineChunks().ValueOrDie();
}
}  // namespace

TEST(JoinTest, SimpleTest) {
  using U = arrow::UInt32Type;
  arrow::RecordBatchVector left{
      RecordBatchOf({"fk", "v_l"}, {ArrayOf<U>({0, 2, 3, 8, 9}), ArrayOf<U>({100, 102, 103, 108, 109})}),
      RecordBatchOf({"fk", "v_l"}, {ArrayOf<U>({10, 12, 13, 18, 19}), ArrayOf<U>({110, 112, 113, 118, 119})})};
  arrow::RecordBatchVector right{
      RecordBatchOf({"pk", "v_r"}, {ArrayOf<U>({3, 8, 9, 0, 12}), ArrayOf<U>({53, 58, 59, 50, 62})}),
      RecordBatchOf({"pk", "v_r"}, {ArrayOf<U>({13, 18, 19, 10, 2}), ArrayOf<U>({63, 68, 69, 60, 52})})};
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  auto native = PrepareAndRun(JoinNative{left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie();
  auto gpu = PrepareAndRun(JoinGpu{*dev, left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie();
  EXPECT_EQ(gpu->num_rows(), 10);
  EXPECT_TRUE(gpu->schema()->Equals(*native->schema()));
  EXPECT_TRUE(Sorted(native, {"v_l", "fk"})->Equals(*Sorted(gpu, {"v_l", "fk"})));
}

TEST(JoinTest, LargeTest) {
  arrow::random::RandomArrayGenerator rng(42);
  const int num_batches = 128, batch_size = 64 << 10;
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  auto right_schema = arrow::schema({arrow::field("x", arrow::uint32(), false)});
  auto right = MakeRandomRecordBatches(rng, right_schema, num_batches, batch_size);
  right = AddColumn("pk", right, MakeIndexColumn(num_batches, batch_size).ValueOrDie());
  right_schema = right[0]->schema();
  auto left_schema = arrow::schema({arrow::field("y", arrow::uint32(), false)});
  auto left = MakeRandomRecordBatches(rng, left_schema, num_batches, batch_size);
  left = AddColumn("fk", left, MakeForeignKeyColumn(rng, batch_size, num_batches, batch_size).ValueOrDie());
  left_schema = left[0]->schema();
  auto native = Sorted(PrepareAndRun(JoinNative{left_schema, right_schema, left, right}).ValueOrDie(), {"fk", "y"});
  auto gpu = Sorted(PrepareAndRun(JoinGpu{*dev, left_schema, right_schema, left, right}).ValueOrDie(), {"fk", "y"});
  EXPECT_EQ(native->num_rows(), (int64_t)num_batches * batch_size);
  EXPECT_EQ(gpu->num_rows(), (int64_t)num_batches * batch_size);
  for (const auto& name : native->schema()->field_names())
    EXPECT_TRUE(native->GetColumnByName(name)->Equals(gpu->GetColumnByName(name)));
}

TEST(JoinTest, MissingForeignKeyIsAnError) {
  using U = arrow::UInt32Type;
  auto left = RecordBatchOf({"fk", "y"}, {ArrayOf<U>({1, 77}), ArrayOf<U>({5, 6})});
  auto right = RecordBatchOf({"pk", "x"}, {ArrayOf<U>({1, 2, 3}), ArrayOf<U>({7, 8, 9})});
  auto dev = olap::Device::Allocate(0).ValueOrDie();
  auto r = PrepareAndRun(JoinGpu{*dev, left->schema(), right->schema(), {left}, {right}});
  EXPECT_FALSE(r.ok());  // the reference asserts (dpu/shared/kernels/hash_probe.c:32-33)
}
