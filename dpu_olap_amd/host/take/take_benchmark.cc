// BM_TakeNative / BM_TakeGpu -- shapes of the reference's host/take/take_benchmark.cc
// (:139-159): {SF, 4Mi, 512Ki} and {SF << 6, 64Ki, 8Ki}; items = DATA rows (:54-56).
#include <iostream>

#include "benchmark/benchmark.h"
#include "generator/generator.h"
#include "system/system.h"
#include "take/take_gpu.h"
#include "take/take_native.h"

namespace upmemeval {
namespace take {

template <typename T>
void BM_Take(benchmark::State& state, T& op) {
  const int64_t num_batches = state.range(0), batch_size = state.range(1);
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
    if (auto timers = op.Timers()) {
      for (auto& kv : timers->get())
        state.counters[kv.first].value += std::chrono::duration<double, std::milli>(kv.second->Result()).count();
    }
  }
  const int64_t processed = num_batches * batch_size * state.iterations();
  state.SetBytesProcessed(processed * (int64_t)sizeof(uint32_t));
  state.SetItemsProcessed(processed);
}

class TakeBatchFixture : public benchmark::Fixture {
 public:
  TakeBatchFixture() : rng_(42) {}
  void SetUp(::benchmark::State& state) override {
    const int num_batches = (int)state.range(0);
    if (num_batches == 0) {
      schema_.reset();
      batches_.clear();
      indices_batches_.clear();
      (void)arrow::SetCpuThreadPoolCapacity(variables::max_threads());
      state.SkipWithError("not an error :)");
      return;
    }
    const int batch_size = (int)state.range(1), indices_batch_size = (int)state.range(2);
    if (!arrow::SetCpuThreadPoolCapacity((int)std::max<int64_t>(1, state.range(3))).ok()) state.SkipWithError("Set CPU threads failed");
    if (num_batches == (int)batches_.size() && batch_size == batches_[0]->num_rows()) return;
    schema_ = arrow::schema({arrow::field("v", arrow::uint32(), false)});
    batches_ = generator::MakeRandomRecordBatches(rng_, schema_, num_batches, batch_size);
    auto md = arrow::key_value_metadata({{"min", "0"}, {"max", std::to_string(batch_size - 1)}});
    auto idx_schema = arrow::schema({arrow::field("i", arrow::uint32(), false, md)});
    indices_batches_ = generator::MakeRandomRecordBatches(rng_, idx_schema, num_batches, indices_batch_size);
    std::cout << "generated " << batches_.size() << " batches of " << batch_size << " rows and " << indices_batch_size
              << " indices each" << std::endl;
  }

 protected:
  arrow::random::RandomArrayGenerator rng_;
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_, indices_batches_;
};

BENCHMARK_DEFINE_F(TakeBatchFixture, BM_TakeNative)(benchmark::State& state) {
  if (state.range(0) == 0) {
    state.SkipWithError("skip");
    return;
  }
  TakeNative op{schema_, batches_, indices_batches_};
  BM_Take(state, op);
}

BENCHMARK_DEFINE_F(TakeBatchFixture, BM_TakeGpu)(benchmark::State& state) {
  if (state.range(0) == 0) {
    state.SkipWithError("skip");
    return;
  }
  auto dev = olap::Device::Allocate(0);
  if (!dev.ok()) {
    state.SkipWithError(dev.status().ToString().c_str());
    return;
  }
  TakeGpu op{**dev, batches_, indices_batches_};
  BM_Take(state, op);
}

BENCHMARK_REGISTER_F(TakeBatchFixture, BM_TakeNative)
    ->ArgNames({"Batches", "Batch-Size", "Indices-Batch-Size", "Threads"})
    ->ArgsProduct({{variables::scale_factor()}, {4 << 20}, {512 << 10}, {variables::max_threads_no_ht(), variables::max_threads()}})
    ->Args({0, 0, 0, 0})
    ->ArgsProduct({{variables::scale_factor() << 6}, {64 << 10}, {8 << 10}, {variables::max_threads_no_ht(), variables::max_threads()}})
    ->Args({0, 0, 0, 0})
    ->MeasureProcessCPUTime()
    ->UseRealTime()
    ->Iterations(1)
    ->Unit(benchmark::kMillisecond);

BENCHMARK_REGISTER_F(TakeBatchFixture, BM_TakeGpu)
    ->ArgNames({"Batches", "Batch-Size", "Indices-Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 4 << 20, 512 << 10, variables::max_gpus()})
    ->Args({0, 0, 0, 0})
    ->Args({variables::scale_factor() << 6, 64 << 10, 8 << 10, variables::max_gpus()})
    ->Args({0, 0, 0, 0})
    ->MeasureProcessCPUTime()
    ->UseRealTime()
    ->Iterations(1)
    ->Unit(benchmark::kMillisecond);

}  // namespace take
}  // namespace upmemeval
