// PartitionGpu -- counterpart of the reference's PartitionDpu
// (host/partition/partition_dpu.{h,cc}; upstream marks it non-functional, README.md:114-118,
// and skips its tests): radix partition of the batches by the top bits of the Wang hash of
// one key column; Run() returns one record batch per partition ({key, payload}).
#pragma once

#include <arrow/api.h>

#include "gpuext/device.h"
#include "timer/timer.h"

namespace upmemeval {
namespace partition {

class PartitionGpu {
 public:
  PartitionGpu(olap::Device& system, std::shared_ptr<arrow::Schema> schema, arrow::RecordBatchVector batches,
               uint32_t nr_partitions, std::string key_column)
      : system_(system),
        schema_(std::move(schema)),
        batches_(std::move(batches)),
        nr_partitions_(nr_partitions),
        key_column_(std::move(key_column)) {}

  arrow::Status Prepare();
  arrow::Result<arrow::RecordBatchVector> Run();
  std::shared_ptr<timer::Timers> Timers() { return timers_; }

 private:
  olap::Device& system_;
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_;
  uint32_t nr_partitions_;
  std::string key_column_;
  std::shared_ptr<timer::Timers> timers_;
};

}  // namespace partition
}  // namespace upmemeval
