// Benchmark entry point (reference: host/main_benchmark.cc:6-17): context keys NR_GPUS (in
// place of NR_DPUS) and SF, then the Google-Benchmark-style runner.
#include <arrow/compute/api.h>

#include <string>

#include "benchmark/benchmark.h"
#include "system/system.h"

int main(int argc, char** argv) {
  if (!arrow::compute::Initialize().ok()) return 2;
  benchmark::AddCustomContext("NR_GPUS", std::to_string(variables::max_gpus()));
  benchmark::AddCustomContext("SF", std::to_string(variables::scale_factor()));
  benchmark::Initialize(&argc, argv);
  if (benchmark::ReportUnrecognizedArguments(argc, argv)) return 1;
  benchmark::RunSpecifiedBenchmarks();
  benchmark::Shutdown();
  return 0;
}
