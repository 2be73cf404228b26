// operator_tests.cc -- Gpu-vs-Native equality tests of the five operators, on the gtest-style
// mini runner (testing/minitest.h).  The cases are the ones the reference's gtests hold
// (host/filter/filter_test.cc, host/take/take_test.cc, host/aggr/aggr_test.cc,
// host/join/join_test.cc and the skipped host/partition/partition_test.cc), with the *Gpu
// operators in the place of the *Dpu ones, plus a few of our own.
#include <arrow/api.h>
#include <arrow/compute/api.h>

#include <cmath>

#include "batches.h"
#include "operators_gpu.h"
#include "operators_native.h"
#include "testing/minitest.h"

using namespace upmemeval;
using namespace upmemeval::generator;
using U32 = arrow::UInt32Type;

namespace {

// Prepare() then Run(), whatever Run() returns.
template <typename Op>
auto Go(Op&& op) -> decltype(op.Run()) {
  ARROW_RETURN_NOT_OK(op.Prepare());
  return op.Run();
}

// NR_GPUS ranks (default 1); OLAP_GROUP_LOOPBACK=1 runs them as logical ranks of device 0, so the whole multi-device
// path -- batches dealt to the ranks, owner split, exchange, per-owner join -- is tested on a one-GPU box.
std::shared_ptr<olap::Device> Gpu() { return olap::Device::Allocate(variables::max_gpus()).ValueOrDie(); }

std::shared_ptr<arrow::Schema> OneColumn(const char* name, std::shared_ptr<arrow::DataType> type = arrow::uint32()) {
  return arrow::schema({arrow::field(name, std::move(type), /*nullable=*/false)});
}

// canonical row order for join results (Arrow sort_indices + take; test-side only)
std::shared_ptr<arrow::Table> SortedBy(const std::shared_ptr<arrow::Table>& t, std::vector<std::string> keys) {
  arrow::compute::SortOptions options;
  for (auto& k : keys) options.sort_keys.emplace_back(arrow::FieldRef(k));
  auto order = arrow::compute::SortIndices(t, options).ValueOrDie();
  return arrow::compute::Take(t, order).ValueOrDie().table()->CombineChunks().ValueOrDie();
}

uint64_t SumOf(const arrow::Datum& column) {
  return std::static_pointer_cast<arrow::UInt64Scalar>(arrow::compute::Sum(column).ValueOrDie().scalar())->value;
}

}  // namespace

// ------------------------------------------------------------------ filter
TEST(FilterTest, SimpleTest) {
  auto batch = RecordBatchOf({"v"}, {ArrayOf<U32>({0, 2, 3, 8, 9})});
  auto dev = Gpu();
  const uint64_t gpu = Go(filter::FilterGpu{*dev, {batch}}).ValueOrDie();
  const uint64_t native = Go(filter::FilterNative{batch->schema(), {batch}}).ValueOrDie();
  EXPECT_EQ(native, gpu);
  EXPECT_EQ(gpu, 5u);
}

TEST(FilterTest, ResultTest) {
  std::vector<uint32_t> values(4096);
  for (uint32_t i = 0; i < values.size(); ++i) values[i] = i + (1u << 30);
  for (uint32_t i : {5u, 8u, 9u, 100u, 270u}) values[i] = i;  // the only rows below 2^30
  auto batch = RecordBatchOf({"v"}, {ArrayOf<U32>(values)});
  auto dev = Gpu();
  filter::FilterGpu gpu{*dev, {batch}};
  EXPECT_TRUE(gpu.Prepare().ok());
  auto got = gpu.GetResult().ValueOrDie();
  filter::FilterNative native{batch->schema(), {batch}};
  EXPECT_TRUE(native.Prepare().ok());
  EXPECT_TRUE(got->Equals(native.GetResult().ValueOrDie()->column(0)));
  EXPECT_EQ(got->length(), 5);
}

TEST(FilterTest, LongerTest) {
  arrow::random::RandomArrayGenerator rng(42);
  auto schema = OneColumn("v");
  auto batches = MakeRandomRecordBatches(rng, schema, 1, 1 << 16);
  auto dev = Gpu();
  filter::FilterGpu gpu{*dev, batches};
  EXPECT_TRUE(gpu.Prepare().ok());
  auto got = gpu.GetResult().ValueOrDie();
  filter::FilterNative native{schema, batches};
  EXPECT_TRUE(native.Prepare().ok());
  EXPECT_TRUE(got->Equals(native.GetResult().ValueOrDie()->column(0)));
  EXPECT_TRUE(gpu.Timers() != nullptr && gpu.Timers()->get().count("gpu-work") == 1);
}

TEST(FilterTest, ManyBatchesKeepBatchOrder) {
  arrow::random::RandomArrayGenerator rng(7);
  auto schema = OneColumn("v");
  auto batches = MakeRandomRecordBatches(rng, schema, 64, 10007);
  auto dev = Gpu();
  filter::FilterGpu gpu{*dev, batches};
  EXPECT_TRUE(gpu.Prepare().ok());
  auto got = gpu.GetResult().ValueOrDie();
  EXPECT_EQ(got->num_chunks(), 64);
  for (int b = 0; b < 64; ++b) {
    filter::FilterNative native{schema, {batches[b]}};
    EXPECT_TRUE(native.Prepare().ok());
    EXPECT_TRUE(arrow::ChunkedArray(got->chunk(b)).Equals(*native.GetResult().ValueOrDie()->column(0)));
  }
}

// -------------------------------------------------------------------- take
static void ExpectZeroTwoNine(const std::shared_ptr<arrow::Table>& result) {
  auto column = std::dynamic_pointer_cast<arrow::UInt32Array>(result->column(0)->chunk(0));
  EXPECT_EQ(column->length(), 3);
  EXPECT_EQ(column->Value(0), 0u);
  EXPECT_EQ(column->Value(1), 2u);
  EXPECT_EQ(column->Value(2), 9u);
}

TEST(TakeTest, SimpleTestNative) {
  auto data = RecordBatchOf({"v"}, {ArrayOf<U32>({0, 2, 3, 8, 9})});
  auto idx = RecordBatchOf({"i"}, {ArrayOf<U32>({0, 1, 4})});
  ExpectZeroTwoNine(Go(take::TakeNative{data->schema(), {data}, {idx}}).ValueOrDie());
}

TEST(TakeTest, SimpleTestGpu) {
  auto dev = Gpu();
  auto data = RecordBatchOf({"v"}, {ArrayOf<U32>({0, 2, 3, 8, 9})});
  auto idx = RecordBatchOf({"i"}, {ArrayOf<U32>({0, 1, 4})});
  ExpectZeroTwoNine(Go(take::TakeGpu{*dev, {data}, {idx}}).ValueOrDie());
}

TEST(TakeTest, LargeTest) {
  auto dev = Gpu();
  arrow::random::RandomArrayGenerator rng(42);
  auto schema = OneColumn("v");
  auto batches = MakeRandomRecordBatches(rng, schema, 128, 64 << 10);
  auto bounds = arrow::key_value_metadata({"min", "max"}, {"0", std::to_string((64 << 10) - 1)});
  auto idx_schema = arrow::schema({arrow::field("i", arrow::uint32(), false, bounds)});
  auto idx = MakeRandomRecordBatches(rng, idx_schema, 128, 8 << 10);
  auto native = Go(take::TakeNative{schema, batches, idx}).ValueOrDie();
  auto gpu = Go(take::TakeGpu{*dev, batches, idx}).ValueOrDie();
  EXPECT_TRUE(native->Equals(*gpu));
}

// --------------------------------------------------------------------- sum
TEST(SumTest, SimpleTestNative) {
  auto batch = RecordBatchOf({"v"}, {ArrayOf<U32>({0, 2, 3, 8, 9})});
  EXPECT_EQ(Go(aggr::AggrNative<arrow::UInt64Array>{batch->schema(), {batch}, "sum"}).ValueOrDie(), 22u);
}

TEST(SumTest, SimpleTestGpu) {
  auto dev = Gpu();
  auto batch = RecordBatchOf({"v"}, {ArrayOf<U32>({0, 2, 3, 8, 9})});
  EXPECT_EQ(Go(aggr::SumGpu{*dev, {batch}}).ValueOrDie(), 22u);
}

TEST(SumTest, LargeTest) {
  auto dev = Gpu();
  arrow::random::RandomArrayGenerator rng(42);
  auto schema = OneColumn("v");
  auto batches = MakeRandomRecordBatches(rng, schema, 128, 64 << 10);
  EXPECT_EQ(Go(aggr::AggrNative<arrow::UInt64Array>{schema, batches, "sum"}).ValueOrDie(), Go(aggr::SumGpu{*dev, batches}).ValueOrDie());
}

TEST(SumTest, Int64Column) {
  auto dev = Gpu();
  arrow::random::RandomArrayGenerator rng(3);
  auto schema = OneColumn("v", arrow::int64());
  auto batches = MakeRandomRecordBatches(rng, schema, 16, 100003);
  const int64_t native = Go(aggr::AggrNative<arrow::Int64Array>{schema, batches, "sum"}).ValueOrDie();
  EXPECT_EQ((uint64_t)native, Go(aggr::SumGpu{*dev, batches}).ValueOrDie());  // both wrap mod 2^64
}

// -------------------------------------------------------------------- join
TEST(JoinTest, SimpleTest) {
  // keys shuffled across the two batches of each side
  arrow::RecordBatchVector left{RecordBatchOf({"fk", "v_l"}, {ArrayOf<U32>({0, 2, 3, 8, 9}), ArrayOf<U32>({100, 102, 103, 108, 109})}),
                                RecordBatchOf({"fk", "v_l"}, {ArrayOf<U32>({10, 12, 13, 18, 19}), ArrayOf<U32>({110, 112, 113, 118, 119})})};
  arrow::RecordBatchVector right{RecordBatchOf({"pk", "v_r"}, {ArrayOf<U32>({3, 8, 9, 0, 12}), ArrayOf<U32>({53, 58, 59, 50, 62})}),
                                 RecordBatchOf({"pk", "v_r"}, {ArrayOf<U32>({13, 18, 19, 10, 2}), ArrayOf<U32>({63, 68, 69, 60, 52})})};
  auto dev = Gpu();
  auto native = Go(join::JoinNative{left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie();
  auto gpu = Go(join::JoinGpu{*dev, left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie();
  EXPECT_EQ(gpu->num_rows(), 10);
  EXPECT_TRUE(gpu->schema()->Equals(*native->schema()));
  EXPECT_TRUE(SortedBy(native, {"v_l", "fk"})->Equals(*SortedBy(gpu, {"v_l", "fk"})));
}

TEST(JoinTest, LargeTest) {
  arrow::random::RandomArrayGenerator rng(42);
  const int num_batches = 128, batch_size = 64 << 10;
  auto dev = Gpu();
  auto right = AddColumn("pk", MakeRandomRecordBatches(rng, OneColumn("x"), num_batches, batch_size),
                         MakeIndexColumn(num_batches, batch_size).ValueOrDie());
  auto left = AddColumn("fk", MakeRandomRecordBatches(rng, OneColumn("y"), num_batches, batch_size),
                        MakeForeignKeyColumn(rng, batch_size, num_batches, batch_size).ValueOrDie());
  auto native = SortedBy(Go(join::JoinNative{left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie(), {"fk", "y"});
  auto gpu = SortedBy(Go(join::JoinGpu{*dev, left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie(), {"fk", "y"});
  EXPECT_EQ(native->num_rows(), (int64_t)num_batches * batch_size);
  EXPECT_EQ(gpu->num_rows(), (int64_t)num_batches * batch_size);
  for (const auto& name : native->schema()->field_names())
    EXPECT_TRUE(native->GetColumnByName(name)->Equals(gpu->GetColumnByName(name)));
}

TEST(JoinTest, WideTablesGoThroughSelectionVectors) {
  // several non-key columns per side, mixed widths: the join returns row ids
  // and every column is gathered through them (the reference's probe + take per column)
  arrow::random::RandomArrayGenerator rng(5);
  const int num_batches = 24, batch_size = 50021;
  auto dev = Gpu();
  auto rschema = arrow::schema({arrow::field("x1", arrow::uint64(), false), arrow::field("x2", arrow::uint32(), false)});
  auto lschema = arrow::schema({arrow::field("y1", arrow::uint32(), false), arrow::field("y2", arrow::int64(), false),
                                arrow::field("y3", arrow::int32(), false)});
  auto right = AddColumn("pk", MakeRandomRecordBatches(rng, rschema, num_batches, batch_size),
                         MakeIndexColumn(num_batches, batch_size).ValueOrDie());
  auto left = AddColumn("fk", MakeRandomRecordBatches(rng, lschema, num_batches, batch_size),
                        MakeForeignKeyColumn(rng, batch_size, num_batches, batch_size).ValueOrDie());
  auto native = SortedBy(Go(join::JoinNative{left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie(), {"fk", "y1", "y2"});
  auto got = Go(join::JoinGpu{*dev, left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie();
  // same column names and types in the same order (Acero marks its output fields nullable)
  EXPECT_TRUE(got->schema()->field_names() == native->schema()->field_names());
  for (int c = 0; c < native->num_columns(); ++c) EXPECT_TRUE(got->field(c)->type()->Equals(native->field(c)->type()));
  auto gpu = SortedBy(got, {"fk", "y1", "y2"});
  EXPECT_EQ(gpu->num_rows(), (int64_t)num_batches * batch_size);
  for (const auto& name : native->schema()->field_names())
    EXPECT_TRUE(native->GetColumnByName(name)->Equals(gpu->GetColumnByName(name)));
  // a single 64-bit payload per side takes the same route
  auto r64 = AddColumn("pk", MakeRandomRecordBatches(rng, OneColumn("x", arrow::int64()), 4, 1000), MakeIndexColumn(4, 1000).ValueOrDie());
  auto l64 = AddColumn("fk", MakeRandomRecordBatches(rng, OneColumn("y", arrow::uint64()), 4, 1000),
                       MakeForeignKeyColumn(rng, 1000, 4, 1000).ValueOrDie());
  auto n64 = SortedBy(Go(join::JoinNative{l64[0]->schema(), r64[0]->schema(), l64, r64}).ValueOrDie(), {"fk", "y"});
  auto g64 = SortedBy(Go(join::JoinGpu{*dev, l64[0]->schema(), r64[0]->schema(), l64, r64}).ValueOrDie(), {"fk", "y"});
  for (const auto& name : n64->schema()->field_names()) EXPECT_TRUE(n64->GetColumnByName(name)->Equals(g64->GetColumnByName(name)));
}

TEST(JoinTest, InnerJoinDropsUnmatched) {
  // half of the foreign keys have no primary key: JoinNative (Arrow inner hash join) drops those rows,
  // and so does JoinGpu when asked to
  arrow::random::RandomArrayGenerator rng(8);
  const int num_batches = 8, batch_size = 40000;
  auto dev = Gpu();
  auto right = AddColumn("pk", MakeRandomRecordBatches(rng, OneColumn("x"), num_batches, batch_size),
                         MakeIndexColumn(num_batches, batch_size).ValueOrDie());
  // foreign keys drawn from twice the key range of the matching batch pair
  auto left = AddColumn("fk", MakeRandomRecordBatches(rng, OneColumn("y"), num_batches, batch_size),
                        MakeForeignKeyColumn(rng, 2 * batch_size, num_batches, batch_size).ValueOrDie());
  auto native = SortedBy(Go(join::JoinNative{left[0]->schema(), right[0]->schema(), left, right}).ValueOrDie(), {"fk", "y"});
  join::JoinGpu strict{*dev, left[0]->schema(), right[0]->schema(), left, right};
  EXPECT_FALSE(Go(strict).ok());
  join::JoinGpu inner{*dev, left[0]->schema(), right[0]->schema(), left, right};
  inner.DropUnmatched(true);
  auto gpu = SortedBy(Go(inner).ValueOrDie(), {"fk", "y"});
  EXPECT_GT(native->num_rows(), 0);
  EXPECT_LT(native->num_rows(), (int64_t)num_batches * batch_size);
  EXPECT_EQ(gpu->num_rows(), native->num_rows());
  EXPECT_EQ((int64_t)inner.Stats().missing_keys, (int64_t)num_batches * batch_size - native->num_rows());
  for (const auto& name : native->schema()->field_names())
    EXPECT_TRUE(native->GetColumnByName(name)->Equals(gpu->GetColumnByName(name)));
}

TEST(JoinTest, SignedInt32Columns) {
  // BASELINE's "int32 PK-FK": keys and payloads of Arrow type int32, negative values included -- the join
  // works on the 32-bit patterns, the result columns keep the input types
  const int n = 50000;
  arrow::Int32Builder pkb, xb, fkb, yb;
  for (int i = 0; i < n; ++i) {
    EXPECT_TRUE(pkb.Append(i * 7 - 150000).ok());   // negative and positive keys
    EXPECT_TRUE(xb.Append(-i).ok());
  }
  for (int i = 0; i < 2 * n; ++i) {
    EXPECT_TRUE(fkb.Append(((i * 31) % n) * 7 - 150000).ok());
    EXPECT_TRUE(yb.Append(i - n).ok());
  }
  auto right = RecordBatchOf({"pk", "x"}, {pkb.Finish().ValueOrDie(), xb.Finish().ValueOrDie()});
  auto left = RecordBatchOf({"fk", "y"}, {fkb.Finish().ValueOrDie(), yb.Finish().ValueOrDie()});
  auto dev = Gpu();
  auto native = SortedBy(Go(join::JoinNative{left->schema(), right->schema(), {left}, {right}}).ValueOrDie(), {"fk", "y"});
  auto got = Go(join::JoinGpu{*dev, left->schema(), right->schema(), {left}, {right}}).ValueOrDie();
  EXPECT_TRUE(got->field(0)->type()->Equals(arrow::int32()));
  EXPECT_TRUE(got->field(2)->type()->Equals(arrow::int32()));
  auto gpu = SortedBy(got, {"fk", "y"});
  EXPECT_EQ(gpu->num_rows(), 2 * n);
  for (const auto& name : native->schema()->field_names()) EXPECT_TRUE(native->GetColumnByName(name)->Equals(gpu->GetColumnByName(name)));
}

TEST(JoinTest, Int64Keys) {
  // 64-bit join keys (the reference's HT_64BIT_KEYS table, hashtable.h:13-18): int64 keys spread over the whole
  // domain, negative ones included, 32-bit payloads; Gpu result equals Arrow's hash join, key column keeps its type
  const int n = 200000;
  arrow::Int64Builder pkb, fkb;
  arrow::UInt32Builder xb, yb;
  auto key = [](int64_t i) { return (i - 100000) * 0x100000001LL + (i % 7); };   // distinct, straddles zero, uses the high word
  for (int i = 0; i < n; ++i) {
    EXPECT_TRUE(pkb.Append(key(i)).ok());
    EXPECT_TRUE(xb.Append((uint32_t)(i * 2654435761u)).ok());
  }
  for (int i = 0; i < 3 * n; ++i) {
    EXPECT_TRUE(fkb.Append(key(((int64_t)i * 31) % n)).ok());
    EXPECT_TRUE(yb.Append((uint32_t)i).ok());
  }
  auto right = RecordBatchOf({"pk", "x"}, {pkb.Finish().ValueOrDie(), xb.Finish().ValueOrDie()});
  auto left = RecordBatchOf({"fk", "y"}, {fkb.Finish().ValueOrDie(), yb.Finish().ValueOrDie()});
  auto dev = Gpu();
  auto native = SortedBy(Go(join::JoinNative{left->schema(), right->schema(), {left}, {right}}).ValueOrDie(), {"fk", "y"});
  auto got = Go(join::JoinGpu{*dev, left->schema(), right->schema(), {left}, {right}}).ValueOrDie();
  EXPECT_TRUE(got->field(0)->type()->Equals(arrow::int64()));
  auto gpu = SortedBy(got, {"fk", "y"});
  EXPECT_EQ(gpu->num_rows(), 3 * n);
  for (const auto& name : native->schema()->field_names()) EXPECT_TRUE(native->GetColumnByName(name)->Equals(gpu->GetColumnByName(name)));
}

TEST(JoinTest, Int64KeysInt64Payloads) {
  // all-int64 tables ("join over Arrow int64 columns"): 64-bit keys with 64-bit payload columns and a second payload column on
  // the right -- the join returns selection vectors and the columns are gathered through them (join_dpu.cc:325-363)
  const int n = 50000;
  arrow::Int64Builder pkb, fkb, xb, yb;
  arrow::UInt32Builder zb;
  auto key = [](int64_t i) { return (i - 25000) * 0x100000001LL + (i % 5); };
  for (int i = 0; i < n; ++i) {
    EXPECT_TRUE(pkb.Append(key(i)).ok());
    EXPECT_TRUE(xb.Append((int64_t)i * 0x1234567890LL - 7).ok());
    EXPECT_TRUE(zb.Append((uint32_t)(i * 40503u)).ok());
  }
  for (int i = 0; i < 2 * n; ++i) {
    EXPECT_TRUE(fkb.Append(key(((int64_t)i * 17) % n)).ok());
    EXPECT_TRUE(yb.Append(-(int64_t)i * 0x100000000LL + i).ok());
  }
  auto right = RecordBatchOf({"pk", "x", "z"}, {pkb.Finish().ValueOrDie(), xb.Finish().ValueOrDie(), zb.Finish().ValueOrDie()});
  auto left = RecordBatchOf({"fk", "y"}, {fkb.Finish().ValueOrDie(), yb.Finish().ValueOrDie()});
  auto dev = Gpu();
  auto native = SortedBy(Go(join::JoinNative{left->schema(), right->schema(), {left}, {right}}).ValueOrDie(), {"fk", "y"});
  auto got = Go(join::JoinGpu{*dev, left->schema(), right->schema(), {left}, {right}}).ValueOrDie();
  EXPECT_EQ(got->num_columns(), 4);
  EXPECT_TRUE(got->GetColumnByName("x")->type()->Equals(arrow::int64()));
  auto gpu = SortedBy(got, {"fk", "y"});
  EXPECT_EQ(gpu->num_rows(), 2 * n);
  for (const auto& name : native->schema()->field_names()) EXPECT_TRUE(native->GetColumnByName(name)->Equals(gpu->GetColumnByName(name)));
}

TEST(JoinTest, MissingForeignKeyIsAnError) {
  auto left = RecordBatchOf({"fk", "y"}, {ArrayOf<U32>({1, 77}), ArrayOf<U32>({5, 6})});
  auto right = RecordBatchOf({"pk", "x"}, {ArrayOf<U32>({1, 2, 3}), ArrayOf<U32>({7, 8, 9})});
  auto dev = Gpu();
  EXPECT_FALSE(Go(join::JoinGpu{*dev, left->schema(), right->schema(), {left}, {right}}).ok());
}

// --------------------------------------------------------------- partition
TEST(PartitionTest, SimpleTest) {
  auto dev = Gpu();
  auto batch = RecordBatchOf({"pk", "x"}, {ArrayOf<U32>({0, 2, 3, 8}), ArrayOf<U32>({100, 101, 102, 103})});
  partition::PartitionGpu op(*dev, batch->schema(), {batch}, 2, "pk");
  EXPECT_TRUE(op.Prepare().ok());
  auto parts = op.Run().ValueOrDie();
  EXPECT_EQ(parts.size(), 2u);
  EXPECT_EQ(parts[0]->num_rows(), 3);
  EXPECT_EQ(parts[1]->num_rows(), 1);
  auto table = arrow::Table::FromRecordBatches(parts).ValueOrDie();
  EXPECT_EQ(SumOf(table->GetColumnByName("pk")), 13u);
  EXPECT_EQ(SumOf(table->GetColumnByName("x")), 406u);
}

TEST(PartitionTest, LargeTest) {
  auto dev = Gpu();
  const int nr_partitions = 32, num_batches = 128, batch_size = 64 << 10;
  const double expected = 1.0 * num_batches * batch_size / nr_partitions;
  arrow::random::RandomArrayGenerator rng(42);
  auto schema = arrow::schema({arrow::field("pk", arrow::uint32(), false), arrow::field("x", arrow::uint32(), false),
                               arrow::field("y", arrow::uint32(), false)});
  auto batches = MakeRandomRecordBatches(rng, schema, num_batches, batch_size);
  partition::PartitionGpu op(*dev, schema, batches, nr_partitions, "pk");
  EXPECT_TRUE(op.Prepare().ok());
  auto parts = op.Run().ValueOrDie();
  EXPECT_EQ(parts.size(), (size_t)nr_partitions);
  int64_t rows = 0;
  uint64_t x_out = 0, x_in = 0;
  for (const auto& p : parts) {
    EXPECT_EQ(p->num_columns(), schema->num_fields());
    EXPECT_GT(p->num_rows(), 0);
    EXPECT_LE(std::abs(p->num_rows() - expected) / expected, 0.1);  // balanced within 10 %
    rows += p->num_rows();
    x_out += SumOf(p->column(1));
  }
  for (const auto& b : batches) x_in += SumOf(b->column(1));
  EXPECT_EQ(rows, (int64_t)num_batches * batch_size);
  EXPECT_EQ(x_out, x_in);  // payload columns travelled with their keys
}

int main(int argc, char** argv) {
  if (!arrow::compute::Initialize().ok()) return 2;
  return minitest::run_all(argc, argv);
}
