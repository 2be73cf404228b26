// group.hip -- several devices behind ONE handle (olap_group): the join operator spans the device
// set the way JoinDpu spans the dpu::DpuSet (reference host/join/join_dpu.cc:82-142,168-400;
// the set itself: host/dpuext/dpuext.hpp:687-715).  Single process, one host thread per rank
// for the device-local steps, one exchange step in between:
//
//   1. batches are dealt to the ranks (batch b -> rank b mod G, like the reference deals batches
//      to DPUs round by round, host/join/join_dpu.cc:82-142);
//   2. every rank splits its share in ONE radix pass into G x S buckets, owner = top log2 G hash
//      bits, S = 512 / G low sub-buckets (olap_split_tuples_u32): what goes to an owner is one
//      contiguous run;
//   3. the runs travel -- this is the shuffle of host/partition/partitioner.cc:280-375, over xGMI
//      instead of host DRAM: RCCL ncclSend / ncclRecv inside one group call over a communicator
//      made by ncclCommInitAll (one device per rank), or plain peer copies (hipMemcpyPeerAsync;
//      same-device copies when several logical ranks share a device: the loopback mode that lets
//      a one-GPU box run the whole path);
//   4. every rank joins what it received, reading the received runs in place
//      (olap_join_presplit_u32 with the owner bits skipped); the result stays sharded by owner,
//      rank-major in the caller's output.
//
// The multi-PROCESS form of the same exchange (one process per GPU, torch.distributed over RCCL)
// is dpu_olap_amd/dist_join.py, which adds the sub-bucket pipeline; this file is what the C++
// operators and any other single-process host link against.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is opened on first use of the RCCL transport

#include <thread>

#include "common.h"

namespace {

constexpr int kSplitBits = 9;  // owner bits + low sub-bucket bits of the split: one 512-bucket pass

struct Buf {  // grow-only device buffer of one rank
  void* p = nullptr;
  uint64_t bytes = 0;
};

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::vector<ncclComm_t> comms;
};

}  // namespace

struct olap_group {
  std::vector<olap_ctx*> ctx;
  std::vector<int> device;
  int transport = OLAP_GROUP_TRANSPORT_AUTO;
  int used_transport = 0;
  bool distinct_devices = true;
  mutable std::string err;
  Rccl rccl;
  std::vector<Buf> tl, tr, rl, rr, out;  // per rank: split outputs, receive buffers, result columns
  double phase_ms[4] = {0, 0, 0, 0};     // last call: split, exchange, join, copy-out (wall)

  int fail(int code, const char* fmt, ...) const {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

namespace {

int grow(olap_group* g, int r, Buf* b, uint64_t bytes) {
  if (bytes <= b->bytes) return OLAP_OK;
  if (hipSetDevice(g->device[r]) != hipSuccess) return g->fail(OLAP_ERR_DEVICE, "hipSetDevice(%d) failed", g->device[r]);
  if (b->p) (void)hipFree(b->p);
  b->p = nullptr;
  b->bytes = 0;
  const uint64_t want = olap::align_up(bytes + bytes / 16 + 4096, 1 << 20);
  if (hipMalloc(&b->p, want) != hipSuccess) {
    (void)hipGetLastError();
    return g->fail(OLAP_ERR_NOMEM, "rank %d: hipMalloc(%llu bytes) failed", r, (unsigned long long)want);
  }
  b->bytes = want;
  return OLAP_OK;
}

int load_rccl(olap_group* g) {
  Rccl& R = g->rccl;
  if (!R.comms.empty()) return OLAP_OK;
  if (!g->distinct_devices)
    return g->fail(OLAP_ERR_INVALID, "the RCCL transport needs one device per rank (ranks share a device here: use the copy transport)");
  if (!R.lib) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (R.lib) break;
    }
    if (!R.lib) return g->fail(OLAP_ERR_DEVICE, "librccl.so not found: %s", dlerror());
    auto sym = [&](const char* n) { return dlsym(R.lib, n); };
    R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.Send = (decltype(R.Send))sym("ncclSend");
    R.Recv = (decltype(R.Recv))sym("ncclRecv");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    if (!R.CommInitAll || !R.CommDestroy || !R.GroupStart || !R.GroupEnd || !R.Send || !R.Recv || !R.GetErrorString)
      return g->fail(OLAP_ERR_DEVICE, "librccl.so lacks a required entry point");
  }
  R.comms.assign(g->ctx.size(), nullptr);
  const ncclResult_t rc = R.CommInitAll(R.comms.data(), (int)g->ctx.size(), g->device.data());
  if (rc != ncclSuccess) {
    R.comms.clear();
    return g->fail(OLAP_ERR_DEVICE, "ncclCommInitAll failed: %s", R.GetErrorString(rc));
  }
  return OLAP_OK;
}

struct Side {                              // one table, all ranks
  std::vector<std::vector<uint64_t>> off;  // [rank][G * S + 1] bucket offsets of the rank's split output
  std::vector<uint64_t> n_in;              // tuples the rank holds
  std::vector<uint64_t> n_out;             // tuples the rank receives
  std::vector<std::vector<uint64_t>> lo, hi;  // [dst][S * G] chunk tables (segment-major, one chunk per sender)
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

int olap_group_create(int n_ranks, const int* device_ids, olap_group** out_group) {
  if (!out_group) return OLAP_ERR_INVALID;
  *out_group = nullptr;
  if (n_ranks < 1 || n_ranks > (1 << (kSplitBits - 1)) || (n_ranks & (n_ranks - 1)) || !device_ids) return OLAP_ERR_INVALID;
  olap_group* g = new (std::nothrow) olap_group();
  if (!g) return OLAP_ERR_NOMEM;
  for (int r = 0; r < n_ranks; ++r) {
    olap_ctx* c = nullptr;
    const int rc = olap_create(device_ids[r], &c);
    if (rc != OLAP_OK) {
      for (olap_ctx* q : g->ctx) olap_destroy(q);
      delete g;
      return rc;
    }
    g->ctx.push_back(c);
    g->device.push_back(device_ids[r]);
    for (int q = 0; q < r; ++q)
      if (device_ids[q] == device_ids[r]) g->distinct_devices = false;
  }
  // peer access between the devices of the group (best effort: without it hipMemcpyPeerAsync stages through the host)
  for (int a = 0; a < n_ranks; ++a)
    for (int b = 0; b < n_ranks; ++b) {
      if (g->device[a] == g->device[b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, g->device[a], g->device[b]) == hipSuccess && can && hipSetDevice(g->device[a]) == hipSuccess)
        if (hipDeviceEnablePeerAccess(g->device[b], 0) != hipSuccess) (void)hipGetLastError();  // already enabled
    }
  g->tl.resize(n_ranks);
  g->tr.resize(n_ranks);
  g->rl.resize(n_ranks);
  g->rr.resize(n_ranks);
  g->out.resize(n_ranks);
  *out_group = g;
  return OLAP_OK;
}

void olap_group_destroy(olap_group* g) {
  if (!g) return;
  if (!g->rccl.comms.empty())
    for (ncclComm_t c : g->rccl.comms)
      if (c) (void)g->rccl.CommDestroy(c);
  for (size_t r = 0; r < g->ctx.size(); ++r) {
    (void)hipSetDevice(g->device[r]);
    for (Buf* b : {&g->tl[r], &g->tr[r], &g->rl[r], &g->rr[r], &g->out[r]})
      if (b->p) (void)hipFree(b->p);
    olap_destroy(g->ctx[r]);
  }
  if (g->rccl.lib) dlclose(g->rccl.lib);
  delete g;
}

int olap_group_size(const olap_group* g) { return g ? (int)g->ctx.size() : 0; }
olap_ctx* olap_group_ctx(olap_group* g, int rank) { return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[rank] : nullptr; }
int olap_group_device(const olap_group* g, int rank) { return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->device[rank] : -1; }
const char* olap_group_last_error(const olap_group* g) { return g ? g->err.c_str() : "null group"; }

int olap_group_set_transport(olap_group* g, int transport) {
  if (!g) return OLAP_ERR_INVALID;
  if (transport != OLAP_GROUP_TRANSPORT_AUTO && transport != OLAP_GROUP_TRANSPORT_COPY && transport != OLAP_GROUP_TRANSPORT_RCCL)
    return g->fail(OLAP_ERR_INVALID, "unknown transport %d", transport);
  g->transport = transport;
  return OLAP_OK;
}

int olap_group_last_transport(const olap_group* g) { return g ? g->used_transport : 0; }

int olap_group_last_phase_ms(const olap_group* g, double out_ms[4]) {
  if (!g || !out_ms) return OLAP_ERR_INVALID;
  for (int i = 0; i < 4; ++i) out_ms[i] = g->phase_ms[i];
  return OLAP_OK;
}

int olap_group_join_u32(olap_group* g, const uint32_t* const* left_fk_ptrs, const uint32_t* const* left_y_ptrs,
                        const uint64_t* left_rows, uint64_t n_left_batches, const uint32_t* const* right_pk_ptrs,
                        const uint32_t* const* right_x_ptrs, const uint64_t* right_rows, uint64_t n_right_batches, int in_mem,
                        uint32_t* out_fk, uint32_t* out_y, uint32_t* out_x, uint64_t out_capacity, uint64_t* out_rows,
                        uint64_t* out_rank_rows, olap_join_stats* stats) {
  if (!g) return OLAP_ERR_INVALID;
  if (!out_rows) return g->fail(OLAP_ERR_INVALID, "out_rows is null");
  *out_rows = 0;
  const int G = (int)g->ctx.size();
  if ((n_left_batches && (!left_fk_ptrs || !left_rows)) || (n_right_batches && (!right_pk_ptrs || !right_rows)))
    return g->fail(OLAP_ERR_INVALID, "null batch table");
  if (in_mem != OLAP_MEM_HOST && in_mem != OLAP_MEM_DEVICE) return g->fail(OLAP_ERR_INVALID, "bad memory space");
  for (int i = 0; i < 4; ++i) g->phase_ms[i] = 0;
  if (G == 1) {  // one rank: the plain join, results to the host
    if (out_rank_rows) out_rank_rows[0] = 0;
    const int rc = olap_join_u32(g->ctx[0], left_fk_ptrs, left_y_ptrs, left_rows, n_left_batches, right_pk_ptrs, right_x_ptrs, right_rows,
                                 n_right_batches, in_mem, out_fk, out_y, out_x, out_capacity, OLAP_MEM_HOST, out_rows, stats);
    if (rc != OLAP_OK) return g->fail(rc, "%s", olap_last_error(g->ctx[0]));
    if (out_rank_rows) out_rank_rows[0] = *out_rows;
    return OLAP_OK;
  }
  int gbits = 0;
  while ((1 << gbits) < G) ++gbits;
  const uint32_t lbits = kSplitBits - gbits, S = 1u << lbits;
  const bool l_pay = left_y_ptrs != nullptr, r_pay = right_x_ptrs != nullptr;

  // ---- 1. deal the batches: batch b -> rank b mod G
  struct Share {
    std::vector<const uint32_t*> k, p;
    std::vector<uint64_t> rows;
    uint64_t n = 0;
  };
  std::vector<Share> L(G), Rt(G);
  uint64_t NL = 0;
  for (uint64_t b = 0; b < n_left_batches; ++b) {
    Share& s = L[b % G];
    s.k.push_back(left_fk_ptrs[b]);
    if (l_pay) s.p.push_back(left_y_ptrs[b]);
    s.rows.push_back(left_rows[b]);
    s.n += left_rows[b];
    NL += left_rows[b];
  }
  for (uint64_t b = 0; b < n_right_batches; ++b) {
    Share& s = Rt[b % G];
    s.k.push_back(right_pk_ptrs[b]);
    if (r_pay) s.p.push_back(right_x_ptrs[b]);
    s.rows.push_back(right_rows[b]);
    s.n += right_rows[b];
  }
  if (NL > out_capacity) return g->fail(OLAP_ERR_CAPACITY, "output capacity %llu < %llu left rows", (unsigned long long)out_capacity, (unsigned long long)NL);
  if (NL && (!out_fk || !out_y || !out_x)) return g->fail(OLAP_ERR_INVALID, "null output column");
  if (!l_pay || !r_pay) return g->fail(OLAP_ERR_INVALID, "the group join carries one payload column per side (selection vectors: single-device join)");

  std::vector<int> rc(G, OLAP_OK);
  auto first_error = [&](const char* what) {
    for (int r = 0; r < G; ++r)
      if (rc[r] != OLAP_OK) return g->fail(rc[r], "rank %d (%s): %s", r, what, olap_last_error(g->ctx[r]));
    return (int)OLAP_OK;
  };
  auto on_ranks = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (int r = 0; r < G; ++r) th.emplace_back([&, r] { rc[r] = fn(r); });
    for (auto& t : th) t.join();
  };

  // ---- 2. split: one radix pass per rank and table into G x S buckets, owner-major
  double t0 = now_ms();
  Side SL, SR;
  for (Side* s : {&SL, &SR}) {
    s->off.assign(G, std::vector<uint64_t>((size_t)G * S + 1, 0));
    s->n_in.assign(G, 0);
    s->n_out.assign(G, 0);
    s->lo.assign(G, std::vector<uint64_t>((size_t)S * G, 0));
    s->hi.assign(G, std::vector<uint64_t>((size_t)S * G, 0));
  }
  for (int r = 0; r < G; ++r) {
    SL.n_in[r] = L[r].n;
    SR.n_in[r] = Rt[r].n;
    OLAP_TRY(grow(g, r, &g->tl[r], (L[r].n + 2) * 8));
    OLAP_TRY(grow(g, r, &g->tr[r], (Rt[r].n + 2) * 8));
  }
  on_ranks([&](int r) {
    int e = olap_split_tuples_u32(g->ctx[r], L[r].k.data(), L[r].p.data(), L[r].rows.data(), L[r].k.size(), in_mem, (uint32_t)gbits, 0, lbits,
                                  (uint64_t*)g->tl[r].p, L[r].n, SL.off[r].data());
    if (e != OLAP_OK) return e;
    return olap_split_tuples_u32(g->ctx[r], Rt[r].k.data(), Rt[r].p.data(), Rt[r].rows.data(), Rt[r].k.size(), in_mem, (uint32_t)gbits, 0, lbits,
                                 (uint64_t*)g->tr[r].p, Rt[r].n, SR.off[r].data());
  });
  OLAP_TRY(first_error("split"));
  g->phase_ms[0] = now_ms() - t0;

  // ---- 3. exchange: rank q's run for owner d lands sender-major in d's receive buffer
  t0 = now_ms();
  struct Move {
    int src, dst;
    uint64_t src_off, dst_off, n;  // tuples
  };
  std::vector<Move> moves_l, moves_r;
  for (auto [side, moves] : {std::pair<Side*, std::vector<Move>*>{&SL, &moves_l}, std::pair<Side*, std::vector<Move>*>{&SR, &moves_r}}) {
    for (int d = 0; d < G; ++d) {
      uint64_t at = 0;
      for (int q = 0; q < G; ++q) {
        const uint64_t b0 = (uint64_t)d * S, lo = side->off[q][b0], hi = side->off[q][b0 + S];
        if (hi > lo) moves->push_back(Move{q, d, lo, at, hi - lo});
        for (uint32_t l = 0; l < S; ++l) {  // chunk (segment l, sender q)
          side->lo[d][(size_t)l * G + q] = at + (side->off[q][b0 + l] - lo);
          side->hi[d][(size_t)l * G + q] = at + (side->off[q][b0 + l + 1] - lo);
        }
        at += hi - lo;
      }
      side->n_out[d] = at;
    }
  }
  for (int r = 0; r < G; ++r) {
    OLAP_TRY(grow(g, r, &g->rl[r], (SL.n_out[r] + 2) * 8));
    OLAP_TRY(grow(g, r, &g->rr[r], (SR.n_out[r] + 2) * 8));
    OLAP_TRY(grow(g, r, &g->out[r], 3 * olap::align_up(SL.n_out[r] * 4 + 16, 256)));
  }
  int transport = g->transport;
  if (transport == OLAP_GROUP_TRANSPORT_AUTO) transport = g->distinct_devices ? OLAP_GROUP_TRANSPORT_RCCL : OLAP_GROUP_TRANSPORT_COPY;
  if (transport == OLAP_GROUP_TRANSPORT_RCCL) {
    OLAP_TRY(load_rccl(g));
    Rccl& R = g->rccl;
    ncclResult_t e = R.GroupStart();
    for (int side = 0; side < 2 && e == ncclSuccess; ++side) {
      const std::vector<Move>& mv = side ? moves_r : moves_l;
      for (const Move& m : mv) {
        const char* sp = (const char*)(side ? g->tr[m.src].p : g->tl[m.src].p) + m.src_off * 8;
        char* dp = (char*)(side ? g->rr[m.dst].p : g->rl[m.dst].p) + m.dst_off * 8;
        // (pieces of at most 512 MiB: this RCCL delivers a message of more than 1 GiB only in part, silently -- scripts/dbg_a2a.py)
        constexpr uint64_t kPiece = 1ull << 26;
        for (uint64_t at = 0; at < m.n && e == ncclSuccess; at += kPiece) {
          const uint64_t cnt = std::min(kPiece, m.n - at);
          e = R.Send(sp + at * 8, cnt, ncclUint64, m.dst, R.comms[m.src], (hipStream_t)olap_get_stream(g->ctx[m.src]));
          if (e == ncclSuccess) e = R.Recv(dp + at * 8, cnt, ncclUint64, m.src, R.comms[m.dst], (hipStream_t)olap_get_stream(g->ctx[m.dst]));
        }
      }
    }
    const ncclResult_t e2 = R.GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess)
      return g->fail(OLAP_ERR_DEVICE, "RCCL exchange failed: %s", R.GetErrorString(e != ncclSuccess ? e : e2));
  } else {
    for (int side = 0; side < 2; ++side) {
      const std::vector<Move>& mv = side ? moves_r : moves_l;
      for (const Move& m : mv) {
        const char* sp = (const char*)(side ? g->tr[m.src].p : g->tl[m.src].p) + m.src_off * 8;
        char* dp = (char*)(side ? g->rr[m.dst].p : g->rl[m.dst].p) + m.dst_off * 8;
        hipStream_t st = (hipStream_t)olap_get_stream(g->ctx[m.src]);
        if (hipSetDevice(g->device[m.src]) != hipSuccess) return g->fail(OLAP_ERR_DEVICE, "hipSetDevice failed");
        const hipError_t he = g->device[m.src] == g->device[m.dst]
                                  ? hipMemcpyAsync(dp, sp, m.n * 8, hipMemcpyDeviceToDevice, st)
                                  : hipMemcpyPeerAsync(dp, g->device[m.dst], sp, g->device[m.src], m.n * 8, st);
        if (he != hipSuccess) return g->fail(OLAP_ERR_DEVICE, "exchange copy %d -> %d failed: %s", m.src, m.dst, hipGetErrorString(he));
      }
    }
  }
  for (int r = 0; r < G; ++r)
    if (olap_synchronize(g->ctx[r]) != OLAP_OK) return g->fail(OLAP_ERR_DEVICE, "rank %d: %s", r, olap_last_error(g->ctx[r]));
  g->used_transport = transport;
  g->phase_ms[1] = now_ms() - t0;

  // ---- 4. every rank joins what it owns
  t0 = now_ms();
  std::vector<uint64_t> rows(G, 0);
  std::vector<olap_join_stats> st(G);
  on_ranks([&](int r) {
    const uint64_t n = SL.n_out[r], colb = olap::align_up(n * 4 + 16, 256);
    char* o = (char*)g->out[r].p;
    int e = olap_join_set_hash_skip_bits(g->ctx[r], gbits);
    if (e != OLAP_OK) return e;
    e = olap_join_presplit_u32(g->ctx[r], (const uint64_t*)g->rl[r].p, n, SL.lo[r].data(), SL.hi[r].data(), (const uint64_t*)g->rr[r].p,
                               SR.n_out[r], SR.lo[r].data(), SR.hi[r].data(), S, (uint32_t)G, (uint32_t*)o, (uint32_t*)(o + colb),
                               (uint32_t*)(o + 2 * colb), n, &rows[r], &st[r]);
    (void)olap_join_set_hash_skip_bits(g->ctx[r], 0);
    return e;
  });
  OLAP_TRY(first_error("join"));
  g->phase_ms[2] = now_ms() - t0;

  // ---- 5. results to the caller, rank-major
  t0 = now_ms();
  std::vector<uint64_t> base(G + 1, 0);
  for (int r = 0; r < G; ++r) base[r + 1] = base[r] + rows[r];
  on_ranks([&](int r) {
    const uint64_t colb = olap::align_up(SL.n_out[r] * 4 + 16, 256);
    char* o = (char*)g->out[r].p;
    uint32_t* dst[3] = {out_fk, out_y, out_x};
    for (int c = 0; c < 3; ++c) {
      const int e = olap_memcpy_d2h(g->ctx[r], dst[c] + base[r], o + c * colb, rows[r] * 4);
      if (e != OLAP_OK) return e;
    }
    return (int)OLAP_OK;
  });
  OLAP_TRY(first_error("copy-out"));
  g->phase_ms[3] = now_ms() - t0;
  *out_rows = base[G];
  if (out_rank_rows)
    for (int r = 0; r < G; ++r) out_rank_rows[r] = rows[r];
  if (stats) {
    *stats = st[0];
    for (int r = 1; r < G; ++r) {
      stats->left_rows += st[r].left_rows;
      stats->right_rows += st[r].right_rows;
      stats->missing_keys += st[r].missing_keys;
      stats->duplicate_keys += st[r].duplicate_keys;
      stats->nr_partitions += st[r].nr_partitions;
      if (st[r].max_build_rows > stats->max_build_rows) stats->max_build_rows = st[r].max_build_rows;
    }
  }
  return OLAP_OK;
}

}  // extern "C"
