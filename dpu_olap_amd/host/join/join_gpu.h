// JoinGpu -- counterpart of the reference's JoinDpu (host/join/join_dpu.h:14-48): inner
// PK-FK join of left {fk, payload...} with right {pk, payload...}; output table = left
// columns followed by the right non-key columns (join_dpu.cc:314-363), rows partition-major.
#pragma once

#include <arrow/api.h>

#include "gpuext/device.h"
#include "timer/timer.h"

namespace upmemeval {
namespace join {

class JoinGpu {
 public:
  JoinGpu(olap::Device& system, std::shared_ptr<arrow::Schema> left_schema, std::shared_ptr<arrow::Schema> right_schema,
          arrow::RecordBatchVector left_batches, arrow::RecordBatchVector right_batches)
      : system_(system),
        left_schema_(std::move(left_schema)),
        right_schema_(std::move(right_schema)),
        left_batches_(std::move(left_batches)),
        right_batches_(std::move(right_batches)) {}

  arrow::Status Prepare();
  arrow::Result<std::shared_ptr<arrow::Table>> Run();
  std::shared_ptr<timer::Timers> Timers() { return timers_; }
  const olap_join_stats& Stats() const { return stats_; }

 private:
  olap::Device& system_;
  std::shared_ptr<arrow::Schema> left_schema_, right_schema_;
  arrow::RecordBatchVector left_batches_, right_batches_;
  std::shared_ptr<timer::Timers> timers_;
  olap_join_stats stats_{};
};

}  // namespace join
}  // namespace upmemeval
