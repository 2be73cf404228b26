// BM_FilterNative / BM_FilterGpu -- registrations, argument shapes and counters follow the
// reference's host/filter/filter_benchmark.cc (:22-72 loop and counters, :142,153 shapes):
// {Batches = SF << 7, Batch-Size = 64 Ki, Threads | GPUs}, Prepare() inside the timed
// region, Iterations(1), real time, items = rows, bytes = rows * 4, "Processed (AVG)" and
// "Results (AVG)" counters, timers in milliseconds.
#include <iostream>

#include "benchmark/benchmark.h"
#include "filter/filter_gpu.h"
#include "filter/filter_native.h"
#include "generator/generator.h"
#include "system/system.h"

namespace upmemeval {
namespace filter {

template <class T>
void BM_Filter(benchmark::State& state, T& op) {
  const int64_t num_batches = state.range(0), batch_size = state.range(1);
  double total_results = 0.0;
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
    total_results += *result;
    if (auto timers = op.Timers()) {
      for (auto& kv : timers->get())
        state.counters[kv.first].value += std::chrono::duration<double, std::milli>(kv.second->Result()).count();
    }
  }
  const int64_t processed = num_batches * batch_size * state.iterations();
  state.SetBytesProcessed(processed * (int64_t)sizeof(uint32_t));
  state.SetItemsProcessed(processed);
  state.counters["Processed (AVG)"] = benchmark::Counter((double)processed, benchmark::Counter::kAvgIterations);
  state.counters["Results (AVG)"] = benchmark::Counter(total_results, benchmark::Counter::kAvgIterations);
}

class BatchGeneratorFixture : public benchmark::Fixture {
 public:
  BatchGeneratorFixture() : rng_(42) {}
  void SetUp(::benchmark::State& state) override {
    const int num_batches = (int)state.range(0);
    if (num_batches == 0) {  // the {0,0,0} entries free the batches between shapes
      schema_.reset();
      batches_.clear();
      (void)arrow::SetCpuThreadPoolCapacity(variables::max_threads());
      state.SkipWithError("not an error :)");
      return;
    }
    const int batch_size = (int)state.range(1);
    if (!arrow::SetCpuThreadPoolCapacity((int)std::max<int64_t>(1, state.range(2))).ok()) state.SkipWithError("Set CPU threads failed");
    if (num_batches == (int)batches_.size() && batch_size == batches_[0]->num_rows()) return;
    schema_ = arrow::schema({arrow::field("v", arrow::uint32(), /*nullable=*/false)});
    batches_ = generator::MakeRandomRecordBatches(rng_, schema_, num_batches, batch_size);
    std::cout << "generated " << batches_.size() << " batches of " << batch_size << " rows" << std::endl;
  }

 protected:
  arrow::random::RandomArrayGenerator rng_;
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_;
};

BENCHMARK_DEFINE_F(BatchGeneratorFixture, BM_FilterNative)(benchmark::State& state) {
  if (state.range(0) == 0) {
    state.SkipWithError("skip");
    return;
  }
  FilterNative op{schema_, batches_};
  BM_Filter<>(state, op);
}

BENCHMARK_DEFINE_F(BatchGeneratorFixture, BM_FilterGpu)(benchmark::State& state) {
  if (state.range(0) == 0) {
    state.SkipWithError("skip");
    return;
  }
  auto dev = olap::Device::Allocate(0);
  if (!dev.ok()) {
    state.SkipWithError(dev.status().ToString().c_str());
    return;
  }
  FilterGpu op{**dev, batches_};
  BM_Filter<>(state, op);
}

BENCHMARK_REGISTER_F(BatchGeneratorFixture, BM_FilterNative)
    ->ArgNames({"Batches", "Batch-Size", "Threads"})
    ->Args({variables::scale_factor() << 7, 64 << 10, variables::max_threads_no_ht()})
    ->Args({0, 0, 0})
    ->MeasureProcessCPUTime()
    ->UseRealTime()
    ->Iterations(1)
    ->Unit(benchmark::kMillisecond);

BENCHMARK_REGISTER_F(BatchGeneratorFixture, BM_FilterGpu)
    ->ArgNames({"Batches", "Batch-Size", "GPUs"})
    ->Args({variables::scale_factor() << 7, 64 << 10, variables::max_gpus()})
    ->Args({0, 0, 0})
    ->MeasureProcessCPUTime()
    ->UseRealTime()
    ->Iterations(1)
    ->Unit(benchmark::kMillisecond);

}  // namespace filter
}  // namespace upmemeval
