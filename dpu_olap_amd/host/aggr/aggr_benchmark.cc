// BM_SumNative / BM_SumGpu -- shapes and counters of the reference's
// host/aggr/aggr_benchmark.cc (:22-63, :132-150): {SF, 2Mi} and {SF << 5, 64Ki}.
#include <iostream>

#include "aggr/aggr_gpu.h"
#include "aggr/aggr_native.h"
#include "benchmark/benchmark.h"
#include "generator/generator.h"
#include "system/system.h"

namespace upmemeval {
namespace aggr {

template <class R, class T>
void BM_Aggregate(benchmark::State& state, T& op) {
  const int64_t num_batches = state.range(0), batch_size = state.range(1);
  double total_result = 0.0;
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
    total_result += (double)(R)*result;
    if (auto timers = op.Timers()) {
      for (auto& kv : timers->get())
        state.counters[kv.first].value += std::chrono::duration<double, std::milli>(kv.second->Result()).count();
    }
  }
  const int64_t processed = num_batches * batch_size * state.iterations();
  state.SetBytesProcessed(processed * (int64_t)sizeof(uint32_t));
  state.SetItemsProcessed(processed);
  state.counters["Result"] = benchmark::Counter(total_result, benchmark::Counter::kAvgIterations);
}

class SumBatchFixture : public benchmark::Fixture {
 public:
  SumBatchFixture() : rng_(42) {}
  void SetUp(::benchmark::State& state) override {
    const int num_batches = (int)state.range(0);
    if (num_batches == 0) {
      schema_.reset();
      batches_.clear();
      (void)arrow::SetCpuThreadPoolCapacity(variables::max_threads());
      state.SkipWithError("not an error :)");
      return;
    }
    const int batch_size = (int)state.range(1);
    if (!arrow::SetCpuThreadPoolCapacity((int)std::max<int64_t>(1, state.range(2))).ok()) state.SkipWithError("Set CPU threads failed");
    if (num_batches == (int)batches_.size() && batch_size == batches_[0]->num_rows()) return;
    schema_ = arrow::schema({arrow::field("v", arrow::uint32(), false)});
    batches_ = generator::MakeRandomRecordBatches(rng_, schema_, num_batches, batch_size);
    std::cout << "generated " << batches_.size() << " batches of " << batch_size << " rows" << std::endl;
  }

 protected:
  arrow::random::RandomArrayGenerator rng_;
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_;
};

BENCHMARK_DEFINE_F(SumBatchFixture, BM_SumNative)(benchmark::State& state) {
  if (state.range(0) == 0) {
    state.SkipWithError("skip");
    return;
  }
  AggrNative<arrow::UInt64Array> op{schema_, batches_, "sum"};
  BM_Aggregate<uint64_t>(state, op);
}

BENCHMARK_DEFINE_F(SumBatchFixture, BM_SumGpu)(benchmark::State& state) {
  if (state.range(0) == 0) {
    state.SkipWithError("skip");
    return;
  }
  auto dev = olap::Device::Allocate(0);
  if (!dev.ok()) {
    state.SkipWithError(dev.status().ToString().c_str());
    return;
  }
  SumGpu op{**dev, batches_};
  BM_Aggregate<uint64_t>(state, op);
}

BENCHMARK_REGISTER_F(SumBatchFixture, BM_SumNative)
    ->ArgNames({"Batches", "Batch-Size", "Threads"})
    ->ArgsProduct({{variables::scale_factor()}, {2 << 20}, {variables::max_threads_no_ht(), variables::max_threads()}})
    ->Args({0, 0, 0})
    ->ArgsProduct({{variables::scale_factor() << 5}, {64 << 10}, {variables::max_threads_no_ht(), variables::max_threads()}})
    ->Args({0, 0, 0})
    ->MeasureProcessCPUTime()
    ->UseRealTime()
    ->Iterations(1)
    ->Unit(benchmark::kMillisecond);

BENCHMARK_REGISTER_F(SumBatchFixture, BM_SumGpu)
    ->ArgNames({"Batches", "Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 2 << 20, variables::max_gpus()})
    ->Args({0, 0, 0})
    ->Args({variables::scale_factor() << 5, 64 << 10, variables::max_gpus()})
    ->Args({0, 0, 0})
    ->MeasureProcessCPUTime()
    ->UseRealTime()
    ->Iterations(1)
    ->Unit(benchmark::kMillisecond);

}  // namespace aggr
}  // namespace upmemeval
