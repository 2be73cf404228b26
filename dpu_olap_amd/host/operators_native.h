// operators_native.h -- the Arrow CPU operators the reference compares its DPU path with
// ("Native": host/filter/filter_native.cc, host/take/take_native.cc, host/aggr/aggr_native.cc,
// host/join/join_native.cc), re-stated for the Arrow C++ 25 in this image (arrow::acero; the
// reference is written against Arrow 8.0's arrow::compute exec plans).  Same class names,
// namespaces and method surface, so the BM_* templates and tests take either kind.
//   filter : source -> filter(v < 2^30) -> sink
//   sum    : source -> scalar aggregate(fn on "v") -> sink
//   take   : compute::Take(NoBoundsCheck) per batch on the CPU pool
//   join   : source x2 -> hashjoin{INNER, fk = pk} -> sink, "pk" dropped
#pragma once

#include <arrow/acero/api.h>
#include <arrow/api.h>
#include <arrow/compute/api.h>

#include "batches.h"
#include "olap_host.h"

namespace upmemeval {

// An Acero plan "sources -> node -> sink" that can be run once into a Table.
class SinkPlan {
 public:
  arrow::Status Open();
  arrow::Result<arrow::acero::ExecNode*> AddSource(const std::shared_ptr<arrow::Schema>& schema,
                                                   std::vector<arrow::compute::ExecBatch> batches);
  arrow::Result<arrow::acero::ExecNode*> Add(const std::string& factory, std::vector<arrow::acero::ExecNode*> inputs,
                                             const arrow::acero::ExecNodeOptions& options);
  arrow::Status Close(arrow::acero::ExecNode* last);
  arrow::Result<std::shared_ptr<arrow::Table>> Collect();

 private:
  std::shared_ptr<arrow::acero::ExecPlan> plan_;
  std::shared_ptr<arrow::Schema> out_schema_;
  generator::ExecBatchGenerator sink_;
};

namespace filter {
class FilterNative {
 public:
  FilterNative(std::shared_ptr<arrow::Schema> schema, arrow::RecordBatchVector batches)
      : schema_(std::move(schema)), batches_(generator::ToExecBatches(batches)) {}
  arrow::Status Prepare();
  arrow::Result<std::shared_ptr<arrow::Table>> GetResult() { return plan_.Collect(); }
  arrow::Result<uint64_t> Run();
  std::shared_ptr<timer::Timers> Timers() { return nullptr; }

 private:
  std::shared_ptr<arrow::Schema> schema_;
  std::vector<arrow::compute::ExecBatch> batches_;
  SinkPlan plan_;
};
}  // namespace filter

namespace aggr {
template <class ArrayType>
class AggrNative {
 public:
  AggrNative(std::shared_ptr<arrow::Schema> schema, arrow::RecordBatchVector batches, std::string fn)
      : schema_(std::move(schema)), batches_(generator::ToExecBatches(batches)), fn_(std::move(fn)) {}
  arrow::Status Prepare();
  arrow::Result<typename ArrayType::value_type> Run();
  std::shared_ptr<timer::Timers> Timers() { return nullptr; }

 private:
  std::shared_ptr<arrow::Schema> schema_;
  std::vector<arrow::compute::ExecBatch> batches_;
  std::string fn_;
  SinkPlan plan_;
};
}  // namespace aggr

namespace take {
class TakeNative {
 public:
  TakeNative(std::shared_ptr<arrow::Schema> schema, arrow::RecordBatchVector batches, arrow::RecordBatchVector indices_batches)
      : schema_(std::move(schema)), batches_(std::move(batches)), indices_batches_(std::move(indices_batches)) {}
  arrow::Status Prepare() { return arrow::compute::Initialize(); }
  arrow::Result<std::shared_ptr<arrow::Table>> Run();
  std::shared_ptr<timer::Timers> Timers() { return nullptr; }

 private:
  std::shared_ptr<arrow::Schema> schema_;
  arrow::RecordBatchVector batches_, indices_batches_;
};
}  // namespace take

namespace join {
class JoinNative {
 public:
  // partitioned = true: one plan per aligned batch pair, results concatenated (valid for the
  // benchmark's data, whose foreign keys stay inside the matching batch's key range)
  JoinNative(std::shared_ptr<arrow::Schema> left_schema, std::shared_ptr<arrow::Schema> right_schema,
             arrow::RecordBatchVector left_batches, arrow::RecordBatchVector right_batches, bool partitioned = false)
      : left_schema_(std::move(left_schema)), right_schema_(std::move(right_schema)),
        left_batches_(generator::ToExecBatches(left_batches)), right_batches_(generator::ToExecBatches(right_batches)),
        partitioned_(partitioned) {}
  arrow::Status Prepare() { return arrow::compute::Initialize(); }
  arrow::Result<std::shared_ptr<arrow::Table>> Run();
  std::shared_ptr<timer::Timers> Timers() { return nullptr; }

 private:
  arrow::Result<std::shared_ptr<arrow::Table>> HashJoin(std::vector<arrow::compute::ExecBatch> left,
                                                        std::vector<arrow::compute::ExecBatch> right);
  std::shared_ptr<arrow::Schema> left_schema_, right_schema_;
  std::vector<arrow::compute::ExecBatch> left_batches_, right_batches_;
  bool partitioned_;
};
}  // namespace join

}  // namespace upmemeval
