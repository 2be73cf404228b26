// operators_gpu.h -- the MI355X operators, drop-in for the reference's *Dpu classes
// (FilterDpu host/filter/filter_dpu.h, TakeDpu host/take/take_dpu.h, SumDpu
// host/aggr/aggr_dpu.h, JoinDpu host/join/join_dpu.h, PartitionDpu
// host/partition/partition_dpu.h): same namespaces, same constructor argument order with
// olap::Device& in the place of dpu::DpuSet&, same Prepare()/Run()/GetResult()/Timers().
// All of them share one base: Prepare() arms the HIP-event timers (the reference loads the
// DPU program there, inside the timed region), Timers() hands out what Run() folded in.
#pragma once

#include <arrow/api.h>

#include "olap_host.h"

namespace upmemeval {

class GpuOperator {
 public:
  explicit GpuOperator(olap::Device& system) : system_(system) {}
  arrow::Status Prepare() {
    timers_ = std::make_shared<timer::Timers>(system_.nr_ranks());
    return system_.BeginTimers();
  }
  std::shared_ptr<timer::Timers> Timers() { return timers_; }

 protected:
  arrow::Status Prepared() const { return timers_ ? arrow::Status::OK() : arrow::Status::Invalid("Prepare() was not called"); }
  arrow::Status Finish() { return system_.FoldTimers(timers_.get()); }
  olap::Device& system_;
  std::shared_ptr<timer::Timers> timers_;
};

namespace filter {
// v < 2^30 on column 0 (dpu/shared/kernels/filter.c:25), input order kept.
class FilterGpu : public GpuOperator {
 public:
  FilterGpu(olap::Device& system, arrow::RecordBatchVector batches) : GpuOperator(system), batches_(std::move(batches)) {}
  arrow::Result<std::shared_ptr<arrow::ChunkedArray>> GetResult();  // one chunk per batch, batch order
  arrow::Result<uint64_t> Run();                                    // number of passing rows

 private:
  arrow::RecordBatchVector batches_;
};
}  // namespace filter

namespace take {
// out[b][i] = data[b][indices[b][i]], no bounds check (dpu/shared/kernels/take.c:33-39).
class TakeGpu : public GpuOperator {
 public:
  TakeGpu(olap::Device& system, arrow::RecordBatchVector batches, arrow::RecordBatchVector indices_batches)
      : GpuOperator(system), batches_(std::move(batches)), indices_batches_(std::move(indices_batches)) {}
  arrow::Result<std::shared_ptr<arrow::Table>> Run();  // one record batch per input batch

 private:
  arrow::RecordBatchVector batches_, indices_batches_;
};
}  // namespace take

namespace aggr {
// sum of column 0 over all batches (dpu/aggr/main.c:44-50).
class SumGpu : public GpuOperator {
 public:
  SumGpu(olap::Device& system, arrow::RecordBatchVector batches) : GpuOperator(system), batches_(std::move(batches)) {}
  arrow::Result<uint64_t> Run();

 private:
  arrow::RecordBatchVector batches_;
};
}  // namespace aggr

namespace partition {
// Radix partition by the top bits of the Wang hash of one key column; one record batch per
// partition (the reference's PartitionDpu is marked non-functional upstream, README.md:114-118).
class PartitionGpu : public GpuOperator {
 public:
  PartitionGpu(olap::Device& system, std::shared_ptr<arrow::Schema> schema, arrow::RecordBatchVector batches,
               uint32_t nr_partitions, std::string key_column)
      : GpuOperator(system), schema_(std::move(schema)), batches_(std::move(batches)), nr_partitions_(nr_partitions),
        key_column_(std::move(key_column)) {}
  arrow::Result<arrow::RecordBatchVector> Run();

 private:
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_;
  uint32_t nr_partitions_;
  std::string key_column_;
};
}  // namespace partition

namespace join {
// Inner PK-FK join of left {fk, y} with right {pk, x}; result = left columns, then the right
// non-key column (join_dpu.cc:314-363), partition-major rows.
class JoinGpu : public GpuOperator {
 public:
  JoinGpu(olap::Device& system, std::shared_ptr<arrow::Schema> left_schema, std::shared_ptr<arrow::Schema> right_schema,
          arrow::RecordBatchVector left_batches, arrow::RecordBatchVector right_batches)
      : GpuOperator(system), left_schema_(std::move(left_schema)), right_schema_(std::move(right_schema)),
        left_batches_(std::move(left_batches)), right_batches_(std::move(right_batches)) {}
  arrow::Result<std::shared_ptr<arrow::Table>> Run();
  const olap_join_stats& Stats() const { return stats_; }
  // true: inner-join semantics like JoinNative's Arrow hash join (left rows without a match are dropped);
  // false (default): they are an error, as in the reference's DPU path (hash_probe.c:33-34)
  JoinGpu& DropUnmatched(bool drop) {
    drop_unmatched_ = drop;
    return *this;
  }

 private:
  arrow::Result<std::shared_ptr<arrow::Table>> Run64(int fk, int pk);
  arrow::Result<std::shared_ptr<arrow::Array>> Gather(const arrow::RecordBatchVector& batches, int column,
                                                      const std::shared_ptr<arrow::Field>& field, const uint32_t* ids,
                                                      uint64_t n);

 private:
  std::shared_ptr<arrow::Schema> left_schema_, right_schema_;
  arrow::RecordBatchVector left_batches_, right_batches_;
  olap_join_stats stats_{};
  bool drop_unmatched_ = false;
};
}  // namespace join

}  // namespace upmemeval
