// bench_harness.cc -- the BM_* benchmarks of the operator hot path on the Google-Benchmark-
// compatible runner (benchmark/benchmark.h).  Names, ArgNames, argument shapes, counters and
// the items/bytes accounting are the reference's (host/filter/filter_benchmark.cc,
// host/take/take_benchmark.cc, host/aggr/aggr_benchmark.cc, host/join/join_benchmark.cc,
// host/main_benchmark.cc), with *Gpu / "GPUs" in the place of *Dpu / "DPUs":
//   BM_Filter{Native,Gpu} {Batches = SF<<7, Batch-Size = 64Ki, Threads|GPUs}
//   BM_Take{Native,Gpu}   {SF, 4Mi, 512Ki, ..} and {SF<<6, 64Ki, 8Ki, ..}   (items = DATA rows)
//   BM_Sum{Native,Gpu}    {SF, 2Mi, ..} and {SF<<5, 64Ki, ..}
//   BM_JoinNative         {SF<<5, 64Ki, 64Ki, Partitioned 1|0};  BM_JoinGpu {SF, 2Mi, 2Mi, GPUs}
//   BM_PartitionGpu       {SF, 2Mi, GPUs, Partitions = max(SF, 2)}
//   BM_{Filter,Sum,Take,Join}GpuResident   the same shapes with the columns generated in HBM (device pointers through the C ABI)
// Prepare() runs inside the timed region like in the reference; filter/take/sum force one
// iteration; {0,...} argument rows free the generated batches between shapes.
#include <arrow/compute/api.h>

#include <iostream>

#include "batches.h"
#include "benchmark/benchmark.h"
#include "olap_host.h"
#include "operators_gpu.h"
#include "operators_native.h"

namespace upmemeval {
namespace {

// One Prepare() + Run() per benchmark iteration; errors end the run the GB way; the
// operator's timers are accumulated into counters (milliseconds).
template <class Op, class OnResult>
void TimedLoop(benchmark::State& state, Op& op, OnResult&& on_result) {
  for (auto _ : state) {
    auto prepared = op.Prepare();
    if (!prepared.ok()) {
      state.SkipWithError(("Prepare failed with error: " + prepared.ToString()).c_str());
      return;
    }
    auto result = op.Run();
    if (!result.ok()) {
      state.SkipWithError(("Run failed with error: " + result.status().ToString()).c_str());
      return;
    }
    on_result(*result);
    if (auto timers = op.Timers())
      for (const auto& [name, t] : timers->get())
        state.counters[name].value += std::chrono::duration<double, std::milli>(t->Result()).count();
  }
}

void CountRows(benchmark::State& state, int64_t rows_per_iteration) {
  const int64_t rows = rows_per_iteration * state.iterations();
  state.SetItemsProcessed(rows);
  state.SetBytesProcessed(rows * (int64_t)sizeof(uint32_t));
}

// The device set of a *Gpu benchmark: as many ranks as its "GPUs" argument says.  A host with fewer devices
// refuses (or runs the ranks on device 0 under OLAP_GROUP_LOOPBACK=1 and says so in the label): a line named
// GPUs:N always used N ranks.
std::shared_ptr<olap::Device> DeviceOrSkip(benchmark::State& state, int gpus = 1) {
  auto dev = olap::Device::Allocate(std::max(1, gpus));
  if (!dev.ok()) {
    state.SkipWithError(dev.status().ToString().c_str());
    return nullptr;
  }
  if ((*dev)->loopback()) state.SetLabel("loopback: " + std::to_string(gpus) + " logical ranks on device 0");
  return *dev;
}

// The device-resident variants drive ONE context through the C ABI: a GPUs argument above 1 is refused.
std::shared_ptr<olap::Device> OneDeviceOrSkip(benchmark::State& state, int64_t gpus) {
  if (gpus > 1) {
    state.SkipWithError("the device-resident variants run on one GPU: set NR_GPUS=1 (the multi-GPU join is BM_JoinGpu / bench.py)");
    return nullptr;
  }
  return DeviceOrSkip(state, 1);
}

// Fixture for the single-column benchmarks: range(0) batches of range(1) rows of "v";
// `threads_arg` is the index of the Threads/GPUs argument; an all-zero row drops the data.
class ColumnFixture : public benchmark::Fixture {
 public:
  explicit ColumnFixture(int threads_arg) : threads_arg_(threads_arg), rng_(42) {}

  void SetUp(benchmark::State& state) override {
    const int num_batches = (int)state.range(0), batch_size = (int)state.range(1);
    if (num_batches == 0) {
      Drop();
      (void)arrow::SetCpuThreadPoolCapacity(variables::max_threads());
      state.SkipWithError("not an error :)");
      return;
    }
    if (!arrow::SetCpuThreadPoolCapacity((int)std::max<int64_t>(1, state.range(threads_arg_))).ok())
      state.SkipWithError("Set CPU threads failed");
    if (num_batches == (int)batches_.size() && batch_size == batches_.front()->num_rows()) return;
    schema_ = arrow::schema({arrow::field("v", arrow::uint32(), /*nullable=*/false)});
    batches_ = generator::MakeRandomRecordBatches(rng_, schema_, num_batches, batch_size);
    Generated(state);
    std::cout << "generated " << batches_.size() << " batches of " << batch_size << " rows" << std::endl;
  }

 protected:
  virtual void Generated(benchmark::State&) {}
  virtual void Drop() {
    schema_.reset();
    batches_.clear();
  }
  bool Skipped(benchmark::State& state) {
    if (state.range(0) != 0) return false;
    state.SkipWithError("skip");
    return true;
  }
  int threads_arg_;
  arrow::random::RandomArrayGenerator rng_;
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_;
};

// ------------------------------------------------------------------ filter
class FilterFixture : public ColumnFixture {
 public:
  FilterFixture() : ColumnFixture(2) {}

 protected:
  template <class Op>
  void Measure(benchmark::State& state, Op& op) {
    double results = 0;
    TimedLoop(state, op, [&](uint64_t n) { results += (double)n; });
    const double processed = (double)state.range(0) * state.range(1) * state.iterations();
    CountRows(state, state.range(0) * state.range(1));
    state.counters["Processed (AVG)"] = benchmark::Counter(processed, benchmark::Counter::kAvgIterations);
    state.counters["Results (AVG)"] = benchmark::Counter(results, benchmark::Counter::kAvgIterations);
  }
};

BENCHMARK_DEFINE_F(FilterFixture, BM_FilterNative)(benchmark::State& state) {
  if (Skipped(state)) return;
  filter::FilterNative op{schema_, batches_};
  Measure(state, op);
}
BENCHMARK_DEFINE_F(FilterFixture, BM_FilterGpu)(benchmark::State& state) {
  if (Skipped(state)) return;
  auto dev = DeviceOrSkip(state, (int)state.range(2));
  if (!dev) return;
  filter::FilterGpu op{*dev, batches_};
  Measure(state, op);
}

#define OLAP_ONE_SHOT ->MeasureProcessCPUTime()->UseRealTime()->Iterations(1)->Unit(benchmark::kMillisecond)

BENCHMARK_REGISTER_F(FilterFixture, BM_FilterNative)
    ->ArgNames({"Batches", "Batch-Size", "Threads"})
    ->Args({variables::scale_factor() << 7, 64 << 10, variables::max_threads_no_ht()})
    ->Args({0, 0, 0}) OLAP_ONE_SHOT;
BENCHMARK_REGISTER_F(FilterFixture, BM_FilterGpu)
    ->ArgNames({"Batches", "Batch-Size", "GPUs"})
    ->Args({variables::scale_factor() << 7, 64 << 10, variables::max_gpus()})
    ->Args({0, 0, 0}) OLAP_ONE_SHOT;

// --------------------------------------------------------------------- sum
class SumFixture : public ColumnFixture {
 public:
  SumFixture() : ColumnFixture(2) {}

 protected:
  template <class Op>
  void Measure(benchmark::State& state, Op& op) {
    double total = 0;
    TimedLoop(state, op, [&](uint64_t s) { total += (double)s; });
    CountRows(state, state.range(0) * state.range(1));
    state.counters["Result"] = benchmark::Counter(total, benchmark::Counter::kAvgIterations);
  }
};

BENCHMARK_DEFINE_F(SumFixture, BM_SumNative)(benchmark::State& state) {
  if (Skipped(state)) return;
  aggr::AggrNative<arrow::UInt64Array> op{schema_, batches_, "sum"};
  Measure(state, op);
}
BENCHMARK_DEFINE_F(SumFixture, BM_SumGpu)(benchmark::State& state) {
  if (Skipped(state)) return;
  auto dev = DeviceOrSkip(state, (int)state.range(2));
  if (!dev) return;
  aggr::SumGpu op{*dev, batches_};
  Measure(state, op);
}

BENCHMARK_REGISTER_F(SumFixture, BM_SumNative)
    ->ArgNames({"Batches", "Batch-Size", "Threads"})
    ->ArgsProduct({{variables::scale_factor()}, {2 << 20}, {variables::max_threads_no_ht(), variables::max_threads()}})
    ->Args({0, 0, 0})
    ->ArgsProduct({{variables::scale_factor() << 5}, {64 << 10}, {variables::max_threads_no_ht(), variables::max_threads()}})
    ->Args({0, 0, 0}) OLAP_ONE_SHOT;
BENCHMARK_REGISTER_F(SumFixture, BM_SumGpu)
    ->ArgNames({"Batches", "Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 2 << 20, variables::max_gpus()})
    ->Args({0, 0, 0})
    ->Args({variables::scale_factor() << 5, 64 << 10, variables::max_gpus()})
    ->Args({0, 0, 0}) OLAP_ONE_SHOT;

// -------------------------------------------------------------------- take
class TakeFixture : public ColumnFixture {
 public:
  TakeFixture() : ColumnFixture(3) {}

 protected:
  void Generated(benchmark::State& state) override {  // range(2) indices per batch, uniform over the batch's rows
    auto bounds = arrow::key_value_metadata({"min", "max"}, {"0", std::to_string(state.range(1) - 1)});
    auto index_schema = arrow::schema({arrow::field("i", arrow::uint32(), false, bounds)});
    indices_ = generator::MakeRandomRecordBatches(rng_, index_schema, (int)state.range(0), (int)state.range(2));
  }
  void Drop() override {
    ColumnFixture::Drop();
    indices_.clear();
  }
  template <class Op>
  void Measure(benchmark::State& state, Op& op) {
    TimedLoop(state, op, [](const std::shared_ptr<arrow::Table>&) {});
    CountRows(state, state.range(0) * state.range(1));  // the reference counts the DATA rows
  }
  arrow::RecordBatchVector indices_;
};

BENCHMARK_DEFINE_F(TakeFixture, BM_TakeNative)(benchmark::State& state) {
  if (Skipped(state)) return;
  take::TakeNative op{schema_, batches_, indices_};
  Measure(state, op);
}
BENCHMARK_DEFINE_F(TakeFixture, BM_TakeGpu)(benchmark::State& state) {
  if (Skipped(state)) return;
  auto dev = DeviceOrSkip(state, (int)state.range(3));
  if (!dev) return;
  take::TakeGpu op{*dev, batches_, indices_};
  Measure(state, op);
}

BENCHMARK_REGISTER_F(TakeFixture, BM_TakeNative)
    ->ArgNames({"Batches", "Batch-Size", "Indices-Batch-Size", "Threads"})
    ->ArgsProduct({{variables::scale_factor()}, {4 << 20}, {512 << 10}, {variables::max_threads_no_ht(), variables::max_threads()}})
    ->Args({0, 0, 0, 0})
    ->ArgsProduct({{variables::scale_factor() << 6}, {64 << 10}, {8 << 10}, {variables::max_threads_no_ht(), variables::max_threads()}})
    ->Args({0, 0, 0, 0}) OLAP_ONE_SHOT;
BENCHMARK_REGISTER_F(TakeFixture, BM_TakeGpu)
    ->ArgNames({"Batches", "Batch-Size", "Indices-Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 4 << 20, 512 << 10, variables::max_gpus()})
    ->Args({0, 0, 0, 0})
    ->Args({variables::scale_factor() << 6, 64 << 10, 8 << 10, variables::max_gpus()})
    ->Args({0, 0, 0, 0}) OLAP_ONE_SHOT;

// -------------------------------------------------------------------- join
// right {pk = 0,1,2,.., x random}, left {fk in the matching batch's key range, y random}
class JoinFixture : public benchmark::Fixture {
 public:
  JoinFixture() : rng_(42) {}

  void SetUp(benchmark::State& state) override {
    const int num_batches = (int)state.range(0), left_rows = (int)state.range(1), right_rows = (int)state.range(2);
    if ((int)right_.size() != num_batches || right_.front()->num_rows() != right_rows) {
      auto payload = generator::MakeRandomRecordBatches(rng_, arrow::schema({arrow::field("x", arrow::uint32(), false)}), num_batches, right_rows);
      right_ = generator::AddColumn("pk", payload, generator::MakeIndexColumn(num_batches, right_rows).ValueOrDie());
    }
    if ((int)left_.size() != num_batches || left_.front()->num_rows() != left_rows) {
      auto payload = generator::MakeRandomRecordBatches(rng_, arrow::schema({arrow::field("y", arrow::uint32(), false)}), num_batches, left_rows);
      left_ = generator::AddColumn("fk", payload, generator::MakeForeignKeyColumn(rng_, right_rows, num_batches, left_rows).ValueOrDie());
    }
    std::cout << left_.size() << " left / " << right_.size() << " right batches generated" << std::endl;
  }
  void TearDown(benchmark::State&) override {
    left_.clear();
    right_.clear();
  }

 protected:
  // rows x columns over both tables, x 4 bytes; NOT multiplied by the iteration count
  // (the reference's accounting, join_benchmark.cc:114-125,141-142)
  template <class Op>
  void Measure(benchmark::State& state, Op& op) {
    const int64_t items = (int64_t)left_.size() * left_.front()->num_rows() * left_.front()->num_columns() +
                          (int64_t)right_.size() * right_.front()->num_rows() * right_.front()->num_columns();
    uint64_t rows = 0;
    TimedLoop(state, op, [&](const std::shared_ptr<arrow::Table>& t) { rows += (uint64_t)t->num_rows(); });
    std::cout << "Total Rows: " << rows << " Iterations: " << state.iterations() << std::endl;
    state.SetItemsProcessed(items);
    state.SetBytesProcessed(items * (int64_t)sizeof(uint32_t));
  }
  arrow::random::RandomArrayGenerator rng_;
  arrow::RecordBatchVector left_, right_;
};

BENCHMARK_DEFINE_F(JoinFixture, BM_JoinNative)(benchmark::State& state) {
  join::JoinNative op{left_.front()->schema(), right_.front()->schema(), left_, right_, state.range(3) != 0};
  Measure(state, op);
}
BENCHMARK_DEFINE_F(JoinFixture, BM_JoinGpu)(benchmark::State& state) {
  auto dev = DeviceOrSkip(state, (int)state.range(3));
  if (!dev) return;
  join::JoinGpu op{*dev, left_.front()->schema(), right_.front()->schema(), left_, right_};
  Measure(state, op);
}

BENCHMARK_REGISTER_F(JoinFixture, BM_JoinNative)
    ->ArgNames({"Batches", "L-Batch-Size", "R-Batch-Size", "Partitioned"})
    ->Args({variables::scale_factor() << 5, 64 << 10, 64 << 10, 1})
    ->Args({variables::scale_factor() << 5, 64 << 10, 64 << 10, 0})
    ->MeasureProcessCPUTime()->UseRealTime()->Unit(benchmark::kMillisecond);
BENCHMARK_REGISTER_F(JoinFixture, BM_JoinGpu)
    ->ArgNames({"Batches", "L-Batch-Size", "R-Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 2 << 20, 2 << 20, variables::max_gpus()})
    ->MeasureProcessCPUTime()->UseRealTime()->Unit(benchmark::kMillisecond);

// --------------------------------------------------------------- partition
// The standalone radix partition (the reference registers it DISABLED_, its operator is a
// stub: partition_benchmark.cc:71,117-127, partition_dpu.cc:50-135): batches {pk, v1} random,
// output = one record batch per partition.
class PartitionFixture : public benchmark::Fixture {
 public:
  PartitionFixture() : rng_(42) {}
  void SetUp(benchmark::State& state) override {
    const int num_batches = (int)state.range(0), batch_size = (int)state.range(1);
    if (num_batches == (int)batches_.size() && batch_size == batches_.front()->num_rows()) return;
    schema_ = arrow::schema({arrow::field("pk", arrow::uint32(), false), arrow::field("v1", arrow::uint32(), false)});
    batches_ = generator::MakeRandomRecordBatches(rng_, schema_, num_batches, batch_size);
  }
  void TearDown(benchmark::State&) override { batches_.clear(); }

 protected:
  arrow::random::RandomArrayGenerator rng_;
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_;
};

BENCHMARK_DEFINE_F(PartitionFixture, BM_PartitionGpu)(benchmark::State& state) {
  if (state.range(2) > 1) {  // the standalone partition runs on ONE device (the reference's PartitionDpu is a stub, partition_dpu.cc:50-135)
    state.SkipWithError("BM_PartitionGpu runs on one GPU: set NR_GPUS=1 for this benchmark");
    return;
  }
  auto dev = DeviceOrSkip(state);
  if (!dev) return;
  partition::PartitionGpu op{*dev, schema_, batches_, (uint32_t)state.range(3), "pk"};
  int64_t rows = 0;
  TimedLoop(state, op, [&](const arrow::RecordBatchVector& parts) {
    for (const auto& p : parts) rows += p->num_rows();
  });
  std::cout << "Total Rows: " << rows << " Iterations: " << state.iterations() << std::endl;
  CountRows(state, state.range(0) * state.range(1) * schema_->num_fields());
}
BENCHMARK_REGISTER_F(PartitionFixture, BM_PartitionGpu)
    ->ArgNames({"Batches", "Batch-Size", "GPUs", "Partitions"})
    ->Args({variables::scale_factor(), 2 << 20, variables::max_gpus(), std::max(2, variables::scale_factor())})
    ->MeasureProcessCPUTime()->UseRealTime()->Unit(benchmark::kMillisecond);

// ------------------------------------------------ device-resident variants
// The same operators with the columns already in HBM -- what a pipeline that keeps its
// intermediates on the GPU sees, and the configuration bench.py reports as the headline.  Data
// come from the library's counter-based generator (olap_gen_*, same distributions as the
// reference's generator: uniform values, pk = row index, fk inside the matching batch's key
// range); the operators are driven through the C ABI on device pointers.
class ResidentColumns {
 public:
  // The columns are carved out of ONE allocation with a skew of (2c + 1) MiB + c * 4352 B: equal-sized columns from
  // separate allocations lie an exact power of two apart and alias on the HBM channels when a kernel streams them in
  // lockstep (the join's probe: 20.7 vs 23.7 ms at SF=2048).
  ResidentColumns(olap::Device& dev, int64_t batches, int64_t batch_rows, int columns) : dev_(dev), rows_(batches * batch_rows) {
    const uint64_t bytes = ((uint64_t)std::max<int64_t>(rows_, 1) * 4 + 255) & ~255ull;
    std::vector<uint64_t> offset((size_t)columns);
    uint64_t total = 0;
    for (int c = 0; c < columns; ++c) {
      total += ((2ull * c + 1) << 20) + (uint64_t)c * 4352;
      total = (total + 255) & ~255ull;
      offset[(size_t)c] = total;
      total += bytes;
    }
    if (olap_malloc(dev_.ctx(), total, &base_) != OLAP_OK) {
      ok_ = false;
      base_ = nullptr;
    }
    for (int c = 0; c < columns && ok_; ++c) {
      char* p = static_cast<char*>(base_) + offset[(size_t)c];
      cols_.push_back(p);
      std::vector<const void*> ptrs;
      for (int64_t b = 0; b < batches; ++b) ptrs.push_back(p + (uint64_t)b * batch_rows * 4);
      tables_.push_back(std::move(ptrs));
    }
    lens_.assign((size_t)batches, (uint64_t)batch_rows);
  }
  ~ResidentColumns() {
    if (base_) (void)olap_free(dev_.ctx(), base_);
  }
  bool ok() const { return ok_; }
  void* column(int c) const { return cols_[c]; }
  const void* const* table(int c) const { return tables_[c].data(); }
  const uint32_t* const* table32(int c) const { return reinterpret_cast<const uint32_t* const*>(tables_[c].data()); }
  const uint64_t* lens() const { return lens_.data(); }
  uint64_t batches() const { return lens_.size(); }
  int64_t rows() const { return rows_; }

 private:
  olap::Device& dev_;
  int64_t rows_;
  bool ok_ = true;
  void* base_ = nullptr;
  std::vector<void*> cols_;
  std::vector<std::vector<const void*>> tables_;
  std::vector<uint64_t> lens_;
};

// Prepare()/Run()/Timers() over a callable, so TimedLoop drives it like any operator.
template <typename R>
class ResidentOp : public GpuOperator {
 public:
  ResidentOp(olap::Device& dev, std::function<arrow::Result<R>()> body) : GpuOperator(dev), body_(std::move(body)) {}
  arrow::Result<R> Run() {
    ARROW_RETURN_NOT_OK(Prepared());
    ARROW_ASSIGN_OR_RAISE(R r, body_());
    ARROW_RETURN_NOT_OK(Finish());
    return r;
  }

 private:
  std::function<arrow::Result<R>()> body_;
};

constexpr uint64_t kSeed = 42;

static void BM_FilterGpuResident(benchmark::State& state) {
  auto dev = OneDeviceOrSkip(state, state.range(2));
  if (!dev) return;
  const int64_t nb = state.range(0), rows = state.range(1);
  ResidentColumns in(*dev, nb, rows, 1), out(*dev, nb, rows, 1);
  if (!in.ok() || !out.ok() || olap_gen_uniform_u32(dev->ctx(), (uint32_t*)in.column(0), (uint64_t)in.rows(), kSeed, 1, 0) != OLAP_OK) {
    state.SkipWithError("device allocation / generation failed");
    return;
  }
  std::vector<void*> outs(nb);
  for (int64_t b = 0; b < nb; ++b) outs[b] = static_cast<char*>(out.column(0)) + (uint64_t)b * rows * 4;
  std::vector<uint64_t> kept(nb);
  const uint32_t threshold = 1u << 30;
  ResidentOp<uint64_t> op(*dev, [&]() -> arrow::Result<uint64_t> {
    uint64_t total = 0;
    ARROW_RETURN_NOT_OK(dev->Check(olap_filter_lt(dev->ctx(), OLAP_U32, in.table(0), in.lens(), in.batches(), OLAP_MEM_DEVICE, &threshold,
                                                  outs.data(), OLAP_MEM_DEVICE, kept.data(), &total)));
    return total;
  });
  double results = 0;
  TimedLoop(state, op, [&](uint64_t n) { results += (double)n; });
  CountRows(state, nb * rows);
  state.counters["Results (AVG)"] = benchmark::Counter(results, benchmark::Counter::kAvgIterations);
}
BENCHMARK(BM_FilterGpuResident)
    ->ArgNames({"Batches", "Batch-Size", "GPUs"})
    ->Args({variables::scale_factor() << 7, 64 << 10, variables::max_gpus()})
    ->MeasureProcessCPUTime()->UseRealTime()->Unit(benchmark::kMillisecond);

static void BM_SumGpuResident(benchmark::State& state) {
  auto dev = OneDeviceOrSkip(state, state.range(2));
  if (!dev) return;
  const int64_t nb = state.range(0), rows = state.range(1);
  ResidentColumns in(*dev, nb, rows, 1);
  if (!in.ok() || olap_gen_uniform_u32(dev->ctx(), (uint32_t*)in.column(0), (uint64_t)in.rows(), kSeed, 1, 0) != OLAP_OK) {
    state.SkipWithError("device allocation / generation failed");
    return;
  }
  ResidentOp<uint64_t> op(*dev, [&]() -> arrow::Result<uint64_t> {
    uint64_t sum = 0;
    ARROW_RETURN_NOT_OK(dev->Check(olap_sum(dev->ctx(), OLAP_U32, in.table(0), in.lens(), in.batches(), OLAP_MEM_DEVICE, &sum)));
    return sum;
  });
  TimedLoop(state, op, [&](uint64_t) {});
  CountRows(state, nb * rows);
}
BENCHMARK(BM_SumGpuResident)
    ->ArgNames({"Batches", "Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 2 << 20, variables::max_gpus()})
    ->Args({variables::scale_factor() << 5, 64 << 10, variables::max_gpus()})
    ->MeasureProcessCPUTime()->UseRealTime()->Unit(benchmark::kMillisecond);

static void BM_TakeGpuResident(benchmark::State& state) {
  auto dev = OneDeviceOrSkip(state, state.range(3));
  if (!dev) return;
  const int64_t nb = state.range(0), rows = state.range(1), picks = state.range(2);
  ResidentColumns data(*dev, nb, rows, 1), idx(*dev, nb, picks, 1), out(*dev, nb, picks, 1);
  olap_ctx* ctx = dev->ctx();
  // indices uniform in [0, rows): the foreign-key generator with one "left batch" spanning everything
  if (!data.ok() || !idx.ok() || !out.ok() ||
      olap_gen_uniform_u32(ctx, (uint32_t*)data.column(0), (uint64_t)data.rows(), kSeed, 1, 0) != OLAP_OK ||
      olap_gen_fk_u32(ctx, (uint32_t*)idx.column(0), (uint64_t)idx.rows(), kSeed, 2, 0, 1ull << 62, (uint64_t)rows) != OLAP_OK) {
    state.SkipWithError("device allocation / generation failed");
    return;
  }
  std::vector<void*> outs(nb);
  for (int64_t b = 0; b < nb; ++b) outs[b] = static_cast<char*>(out.column(0)) + (uint64_t)b * picks * 4;
  ResidentOp<uint64_t> op(*dev, [&]() -> arrow::Result<uint64_t> {
    ARROW_RETURN_NOT_OK(dev->Check(olap_take(ctx, OLAP_U32, data.table(0), data.lens(), idx.table32(0), idx.lens(), data.batches(),
                                             OLAP_MEM_DEVICE, outs.data(), OLAP_MEM_DEVICE)));
    return (uint64_t)idx.rows();
  });
  TimedLoop(state, op, [&](uint64_t) {});
  CountRows(state, nb * rows);  // items = DATA rows, as take_benchmark.cc:54-56 counts them
}
BENCHMARK(BM_TakeGpuResident)
    ->ArgNames({"Batches", "Batch-Size", "Indices-Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 4 << 20, 512 << 10, variables::max_gpus()})
    ->Args({variables::scale_factor() << 6, 64 << 10, 8 << 10, variables::max_gpus()})
    ->MeasureProcessCPUTime()->UseRealTime()->Unit(benchmark::kMillisecond);

static void BM_JoinGpuResident(benchmark::State& state) {
  auto dev = OneDeviceOrSkip(state, state.range(3));
  if (!dev) return;
  const int64_t nb = state.range(0), lrows = state.range(1), rrows = state.range(2);
  ResidentColumns left(*dev, nb, lrows, 2), right(*dev, nb, rrows, 2), out(*dev, nb, lrows, 3);
  olap_ctx* ctx = dev->ctx();
  if (!left.ok() || !right.ok() || !out.ok() ||
      olap_gen_fk_u32(ctx, (uint32_t*)left.column(0), (uint64_t)left.rows(), kSeed, 3, 0, (uint64_t)lrows, (uint64_t)rrows) != OLAP_OK ||
      olap_gen_uniform_u32(ctx, (uint32_t*)left.column(1), (uint64_t)left.rows(), kSeed, 4, 0) != OLAP_OK ||
      olap_gen_index_u32(ctx, (uint32_t*)right.column(0), (uint64_t)right.rows(), 0) != OLAP_OK ||
      olap_gen_uniform_u32(ctx, (uint32_t*)right.column(1), (uint64_t)right.rows(), kSeed, 5, 0) != OLAP_OK) {
    state.SkipWithError("device allocation / generation failed");
    return;
  }
  olap_join_stats stats{};
  ResidentOp<uint64_t> op(*dev, [&]() -> arrow::Result<uint64_t> {
    uint64_t n = 0;
    ARROW_RETURN_NOT_OK(dev->Check(olap_join_u32(ctx, left.table32(0), left.table32(1), left.lens(), left.batches(), right.table32(0),
                                                 right.table32(1), right.lens(), right.batches(), OLAP_MEM_DEVICE, (uint32_t*)out.column(0),
                                                 (uint32_t*)out.column(1), (uint32_t*)out.column(2), (uint64_t)left.rows(), OLAP_MEM_DEVICE,
                                                 &n, &stats)));
    return n;
  });
  // One untimed join first: it sizes the context's scratch arena (a ~120 GB hipMalloc at SF=2048, seconds of
  // page-table work) -- device allocation, which the reference also does outside its timed loop
  // (DpuSet::allocate before `for (auto _ : state)`, join_benchmark.cc:86-100).
  if (!op.Prepare().ok() || !op.Run().ok()) {
    state.SkipWithError(olap_last_error(ctx));
    return;
  }
  uint64_t rows = 0;
  TimedLoop(state, op, [&](uint64_t n) { rows += n; });
  std::cout << "Total Rows: " << rows << " Iterations: " << state.iterations() << std::endl;
  const int64_t items = (left.rows() + right.rows()) * 2;  // rows x columns over both tables (join_benchmark.cc:114-125)
  state.SetItemsProcessed(items);
  state.SetBytesProcessed(items * (int64_t)sizeof(uint32_t));
  state.counters["partitions"] = (double)stats.nr_partitions;
}
BENCHMARK(BM_JoinGpuResident)
    ->ArgNames({"Batches", "L-Batch-Size", "R-Batch-Size", "GPUs"})
    ->Args({variables::scale_factor(), 2 << 20, 2 << 20, variables::max_gpus()})
    ->MeasureProcessCPUTime()->UseRealTime()->Iterations(3)->Unit(benchmark::kMillisecond);

}  // namespace
}  // namespace upmemeval

int main(int argc, char** argv) {
  if (!arrow::compute::Initialize().ok()) return 2;
  benchmark::AddCustomContext("NR_GPUS", std::to_string(variables::max_gpus()));
  benchmark::AddCustomContext("SF", std::to_string(variables::scale_factor()));
  // what the *Native operators run on: Arrow's CPU pool (BM_JoinNative uses all of it; the single-column
  // baselines set their "Threads" argument), next to what the host has
  if (const char* cap = std::getenv("MAX_THREADS")) (void)arrow::SetCpuThreadPoolCapacity(std::max(1, std::atoi(cap)));
  benchmark::AddCustomContext("arrow_cpu_threads", std::to_string(arrow::GetCpuThreadPoolCapacity()));
  benchmark::AddCustomContext("hardware_concurrency", std::to_string(std::thread::hardware_concurrency()));
  benchmark::Initialize(&argc, argv);
  if (benchmark::ReportUnrecognizedArguments(argc, argv)) return 1;
  benchmark::RunSpecifiedBenchmarks();
  benchmark::Shutdown();
  return 0;
}
