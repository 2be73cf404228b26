// BM_JoinNative / BM_JoinGpu -- shapes, accounting and counters of the reference's
// host/join/join_benchmark.cc: BM_JoinNative {SF << 5, 64Ki, 64Ki, Partitioned 1|0} (:161-162),
// BM_JoinDpu -> BM_JoinGpu {SF, 2Mi, 2Mi, GPUs} (:170-171); items = sum of rows x columns of
// both tables, bytes = items x 4, not multiplied by iterations (:114-125,141-142,152-153).
#include <iostream>

#include "benchmark/benchmark.h"
#include "generator/generator.h"
#include "join/join_gpu.h"
#include "join/join_native.h"
#include "system/system.h"

namespace upmemeval {
namespace join {

template <class T>
void BM_Join(benchmark::State& state, T& op) {
  uint64_t total_rows = 0;
  for (auto _ : state) {
    auto prepared = op.Prepare();
    if (!prepared.ok()) {
      state.SkipWithError(("Prepare failed with error: " + prepared.ToString()).c_str());
      break;
    }
    auto result = op.Run();
    if (!result.ok()) {
      state.SkipWithError(("Run failed with error: " + result.status().ToString()).c_str());
      break;
    }
    total_rows += (uint64_t)result->get()->num_rows();
    if (auto timers = op.Timers()) {
      for (auto& kv : timers->get())
        state.counters[kv.first].value += std::chrono::duration<double, std::milli>(kv.second->Result()).count();
    }
  }
  std::cout << "Total Rows: " << total_rows << " Iterations: " << state.iterations() << std::endl;
}

class PartitionedBatchGeneratorFixture : public benchmark::Fixture {
 public:
  PartitionedBatchGeneratorFixture() : rng_(42) {}

  void SetUp(::benchmark::State& state) override {
    const int num_batches = (int)state.range(0), lsize = (int)state.range(1), rsize = (int)state.range(2);
    if (right_.size() != (size_t)num_batches || right_[0]->num_rows() != rsize) {
      auto schema = arrow::schema({arrow::field("x", arrow::uint32(), false)});
      auto batches = generator::MakeRandomRecordBatches(rng_, schema, num_batches, rsize);
      right_ = generator::AddColumn("pk", batches, generator::MakeIndexColumn(num_batches, rsize).ValueOrDie());
      right_schema_ = right_[0]->schema();
    }
    if (left_.size() != (size_t)num_batches || left_[0]->num_rows() != lsize) {
      auto schema = arrow::schema({arrow::field("y", arrow::uint32(), false)});
      auto batches = generator::MakeRandomRecordBatches(rng_, schema, num_batches, lsize);
      left_ = generator::AddColumn("fk", batches,
                                   generator::MakeForeignKeyColumn(rng_, rsize, num_batches, lsize).ValueOrDie());
      left_schema_ = left_[0]->schema();
    }
    std::cout << left_.size() << " left / " << right_.size() << " right batches generated" << std::endl;
  }

  void TearDown(::benchmark::State&) override {
    left_.clear();
    right_.clear();
  }

  int64_t total_items() const {
    return (int64_t)left_.size() * left_[0]->num_rows() * left_[0]->num_columns() +
           (int64_t)right_.size() * right_[0]->num_rows() * right_[0]->num_columns();
  }

 protected:
  arrow::random::RandomArrayGenerator rng_;
  std::shared_ptr<arrow::Schema> left_schema_, right_schema_;
  arrow::RecordBatchVector left_, right_;
};

BENCHMARK_DEFINE_F(PartitionedBatchGeneratorFixture, BM_JoinNative)(benchmark::State& state) {
  const int64_t items = total_items();
  JoinNative op{left_schema_, right_schema_, left_, right_, state.range(3) != 0};
  BM_Join<>(state, op);
  state.SetItemsProcessed(items);
  state.SetBytesProcessed(items * (int64_t)sizeof(uint32_t));
}

BENCHMARK_DEFINE_F(PartitionedBatchGeneratorFixture, BM_JoinGpu)(benchmark::State& state) {
  const int64_t items = total_items();
  auto dev = olap::Device::Allocate(0);
  if (!dev.ok()) {
    state.SkipWithError(dev.status().ToString().c_str());
    return;
  }
  JoinGpu op{**dev, left_schema_, right_schema_, left_, right_};
  BM_Join<>(state, op);
  state.SetItemsProcessed(items);
  state.SetBytesProcessed(items * (int64_t)sizeof(uint32_t));
}

BENCHMARK_REGISTER_F(PartitionedBatchGeneratorFixture, BM_JoinNative)
    ->ArgNames({"Batches", "L-Batch-Size", "R-Batch-Size", "Partitioned"})
    ->Args({variables::scale_factor() << 5, 64 << 10, 64 << 10, 1})
    ->Args({variables::scale_factor() << 5, 64 << 10, 64 << 10, 0})
    ->MeasureProcessCPUTime()
    ->UseRealTime()
    ->Unit(benchmark::kMillisecond);

BENCHMARK_REGISTER_F(PartitionedBatchGeneratorFixture, BM_JoinGpu)
    ->ArgNames({"Batches", "L-Batch-Size", "R-Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 2 << 20, 2 << 20, variables::max_gpus()})
    ->MeasureProcessCPUTime()
    ->UseRealTime()
    ->Unit(benchmark::kMillisecond);

}  // namespace join
}  // namespace upmemeval
