// Host <-> HBM staging for OLAP_MEM_HOST operands: the role of the reference's
// arrow_copy_to_dpus / arrow_copy_from_dpus (host/dpuext/arrow_utils.cc:47-73,147-266), which
// hand pageable Arrow buffers to the device runtime.
//
// A pageable buffer cannot be DMA'd directly; hipMemcpyAsync on one bounces it through the
// runtime's own small pinned buffer on the calling thread (~10 GB/s measured).  Here the
// bounce is explicit and parallel: a ring of pinned slots, a pool of host threads that
// memcpy between the caller's buffers and a slot, and one large asynchronous DMA per slot
// on the context's stream -- the threads fill (or drain) slot i+1 while the DMA engine moves
// slot i, so throughput is min(parallel memcpy, PCIe).
//
// Copies are described as items {device address, host address, bytes}.  Consecutive items
// whose device addresses are contiguous up to the arena's 256-byte alignment gaps form a
// run; a run is cut into windows of at most one slot, and the slot mirrors the window's
// device layout (gaps included), so a window is ONE DMA however many batches it spans.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "common.h"

namespace olap {

namespace {

struct Piece {   // one memcpy between a slot and a caller buffer
  uint64_t slot_off;
  char* host;
  uint64_t bytes;
};

struct Window {  // one DMA
  char* dev;
  uint64_t bytes;
  std::vector<Piece> pieces;
};

constexpr uint64_t kGap = 256;        // ws_alloc alignment: the largest gap inside a run
constexpr uint64_t kMinSplit = 1 << 20;  // do not hand a thread less than this

}  // namespace

class Stager {
 public:
  Stager(int threads, uint64_t slot_bytes, int slots) : slot_bytes_(slot_bytes), n_slots_(slots) {
    if (threads < 1) threads = 1;
    for (int t = 1; t < threads; ++t) workers_.emplace_back([this] { worker(); });
    n_threads_ = threads;
  }

  ~Stager() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
    for (hipEvent_t e : events_) (void)hipEventDestroy(e);
    if (ring_) (void)hipHostFree(ring_);
  }

  int threads() const { return n_threads_; }

  int init(olap_ctx* ctx) {
    if (ring_) return OLAP_OK;
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, slot_bytes_ * n_slots_, hipHostMallocDefault);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return ctx->fail(OLAP_ERR_NOMEM, "pinned staging ring hipHostMalloc(%llu bytes) failed: %s",
                       (unsigned long long)(slot_bytes_ * n_slots_), hipGetErrorString(e));
    }
    ring_ = (char*)p;
    events_.resize(n_slots_);
    busy_.assign(n_slots_, false);
    for (auto& ev : events_) OLAP_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return OLAP_OK;
  }

  // host -> device.  Returns when every DMA has been ENQUEUED on ctx->stream (the data of the
  // last windows may still be in flight; later work on the stream is ordered after it).
  int h2d(olap_ctx* ctx, const std::vector<StageItem>& items) {
    OLAP_TRY(init(ctx));
    std::vector<Window> wins;
    make_windows(items, &wins);
    int slot = 0;
    for (const Window& w : wins) {
      OLAP_TRY(acquire(ctx, slot));
      char* base = ring_ + (uint64_t)slot * slot_bytes_;
      copy_pieces(w.pieces, base, /*to_slot=*/true);
      OLAP_HIP(ctx, hipMemcpyAsync(w.dev, base, w.bytes, hipMemcpyHostToDevice, ctx->stream));
      OLAP_HIP(ctx, hipEventRecord(events_[slot], ctx->stream));
      busy_[slot] = true;
      slot = (slot + 1) % n_slots_;
    }
    return OLAP_OK;
  }

  // device -> host.  Returns when the caller's buffers hold the data.
  int d2h(olap_ctx* ctx, const std::vector<StageItem>& items) {
    OLAP_TRY(init(ctx));
    std::vector<Window> wins;
    make_windows(items, &wins);
    const size_t n = wins.size();
    size_t issued = 0;
    auto issue = [&](size_t i) -> int {
      const int slot = (int)(i % n_slots_);
      OLAP_TRY(acquire(ctx, slot));  // drained already (windows are consumed in order)
      OLAP_HIP(ctx, hipMemcpyAsync(ring_ + (uint64_t)slot * slot_bytes_, wins[i].dev, wins[i].bytes, hipMemcpyDeviceToHost,
                                   ctx->stream));
      OLAP_HIP(ctx, hipEventRecord(events_[slot], ctx->stream));
      busy_[slot] = true;
      return OLAP_OK;
    };
    for (size_t i = 0; i < n; ++i) {
      // keep n_slots - 1 DMAs ahead of the drain
      while (issued < n && issued < i + (size_t)n_slots_ - (n_slots_ > 1 ? 1 : 0)) OLAP_TRY(issue(issued++));
      if (issued <= i) OLAP_TRY(issue(issued++));
      const int slot = (int)(i % n_slots_);
      OLAP_HIP(ctx, hipEventSynchronize(events_[slot]));
      busy_[slot] = false;
      copy_pieces(wins[i].pieces, ring_ + (uint64_t)slot * slot_bytes_, /*to_slot=*/false);
    }
    return OLAP_OK;
  }

 private:
  int acquire(olap_ctx* ctx, int slot) {
    if (busy_[slot]) {
      OLAP_HIP(ctx, hipEventSynchronize(events_[slot]));
      busy_[slot] = false;
    }
    return OLAP_OK;
  }

  void make_windows(const std::vector<StageItem>& items, std::vector<Window>* out) const {
    Window cur{nullptr, 0, {}};
    auto flush = [&] {
      if (cur.bytes) out->push_back(std::move(cur));
      cur = Window{nullptr, 0, {}};
    };
    for (const StageItem& it : items) {
      uint64_t done = 0;
      while (done < it.bytes) {
        char* d = (char*)it.dev + done;
        if (cur.bytes) {
          const char* end = cur.dev + cur.bytes;
          const bool adjacent = d >= end && (uint64_t)(d - end) < kGap && done == 0;
          if (!adjacent || (uint64_t)(d - cur.dev) >= slot_bytes_) flush();
        }
        if (!cur.bytes) cur.dev = d;
        const uint64_t off = (uint64_t)(d - cur.dev);
        const uint64_t take = std::min(it.bytes - done, slot_bytes_ - off);
        cur.pieces.push_back(Piece{off, (char*)it.host + done, take});
        cur.bytes = off + take;
        done += take;
        if (cur.bytes == slot_bytes_) flush();
      }
    }
    flush();
  }

  // Split the pieces of one window over the pool (large pieces are cut) and run them.
  void copy_pieces(const std::vector<Piece>& pieces, char* slot, bool to_slot) {
    uint64_t total = 0;
    for (const Piece& p : pieces) total += p.bytes;
    int parts = (int)std::min<uint64_t>((uint64_t)n_threads_, std::max<uint64_t>(1, total / kMinSplit));
    if (parts <= 1) {
      for (const Piece& p : pieces) move(slot + p.slot_off, p.host, p.bytes, to_slot);
      return;
    }
    const uint64_t share = (total + parts - 1) / parts;
    std::vector<std::vector<Piece>> tasks(parts);
    int k = 0;
    uint64_t acc = 0;
    for (const Piece& p : pieces) {
      uint64_t done = 0;
      while (done < p.bytes) {
        const uint64_t take = std::min(p.bytes - done, share - acc);
        tasks[k].push_back(Piece{p.slot_off + done, p.host + done, take});
        done += take;
        acc += take;
        if (acc == share && k + 1 < parts) {
          ++k;
          acc = 0;
        }
      }
    }
    {
      std::lock_guard<std::mutex> g(mu_);  // workers are idle here: the previous job was waited for
      tasks_.swap(tasks);
      job_slot_ = slot;
      job_to_slot_ = to_slot;
      next_task_ = 1;  // task 0 is the caller's
      pending_ = parts - 1;
      ++generation_;
    }
    cv_.notify_all();
    for (const Piece& p : tasks_[0]) move(slot + p.slot_off, p.host, p.bytes, to_slot);
    // help with whatever is left, then wait for the workers
    for (;;) {
      int t;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (next_task_ >= (int)tasks_.size()) break;
        t = next_task_++;
      }
      for (const Piece& p : tasks_[t]) move(slot + p.slot_off, p.host, p.bytes, to_slot);
      std::lock_guard<std::mutex> g(mu_);
      --pending_;
    }
    std::unique_lock<std::mutex> g(mu_);
    done_cv_.wait(g, [this] { return pending_ == 0; });
  }

  static void move(char* slot_addr, char* host, uint64_t bytes, bool to_slot) {
    if (to_slot) std::memcpy(slot_addr, host, bytes);
    else std::memcpy(host, slot_addr, bytes);
  }

  void worker() {
    uint64_t seen = 0;
    for (;;) {
      int t = -1;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return stop_ || (generation_ != seen && next_task_ < (int)tasks_.size()) ; });
        if (stop_) return;
        if (next_task_ < (int)tasks_.size()) t = next_task_++;
        if (next_task_ >= (int)tasks_.size()) seen = generation_;
      }
      if (t < 0) continue;
      for (const Piece& p : tasks_[t]) move(job_slot_ + p.slot_off, p.host, p.bytes, job_to_slot_);
      bool last;
      {
        std::lock_guard<std::mutex> g(mu_);
        last = --pending_ == 0;
      }
      if (last) done_cv_.notify_all();
    }
  }

  uint64_t slot_bytes_;
  int n_slots_;
  int n_threads_ = 1;
  char* ring_ = nullptr;
  std::vector<hipEvent_t> events_;
  std::vector<bool> busy_;

  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  bool stop_ = false;
  uint64_t generation_ = 0;
  std::vector<std::vector<Piece>> tasks_;
  int next_task_ = 0;
  int pending_ = 0;
  char* job_slot_ = nullptr;
  bool job_to_slot_ = true;
};

static Stager* stager_of(olap_ctx* ctx) {
  if (!ctx->stager) {
    int threads = ctx->host_copy_threads;
    if (threads <= 0) {
      const unsigned hw = std::thread::hardware_concurrency();
      threads = (int)std::min<unsigned>(hw ? hw : 1, 8);
    }
    ctx->stager = new Stager(threads, 32ull << 20, 4);
  }
  return ctx->stager;
}

void stager_destroy(olap_ctx* ctx) {
  delete ctx->stager;
  ctx->stager = nullptr;
}

int stage_h2d(olap_ctx* ctx, const std::vector<StageItem>& items) {
  if (items.empty()) return OLAP_OK;
  if (!ctx->host_stage) {  // A/B: let the runtime bounce each buffer itself
    for (const StageItem& it : items)
      if (it.bytes) OLAP_HIP(ctx, hipMemcpyAsync(it.dev, it.host, it.bytes, hipMemcpyHostToDevice, ctx->stream));
    return OLAP_OK;
  }
  return stager_of(ctx)->h2d(ctx, items);
}

int stage_d2h(olap_ctx* ctx, const std::vector<StageItem>& items) {
  if (items.empty()) return OLAP_OK;
  if (!ctx->host_stage) {
    for (const StageItem& it : items)
      if (it.bytes) OLAP_HIP(ctx, hipMemcpyAsync(it.host, it.dev, it.bytes, hipMemcpyDeviceToHost, ctx->stream));
    OLAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OLAP_OK;
  }
  return stager_of(ctx)->d2h(ctx, items);
}

}  // namespace olap
