"""GPU parity tests of the chunk-by-chunk last pass of pre-split joins (olap_join_presplit_u32 with ONE radix pass left: what
a rank runs on the tuples it received from the all-to-all, host/partition/partitioner.cc:280-375 being the shuffle it
replaces): the pass runs one workgroup per (segment, sender) chunk into packed sub-partitions (csrc/pack6.h,
PassSrc::out_by_block) and k_join_probe_direct6_sub builds a partition's table from the sub-partitions of all of its
segment's chunks.  Rows against the Arrow golden join and the generator's digest; bit-exact.  Forced with presplit_sub = 2
at sizes the automatic choice leaves to the histogram pipeline."""
import numpy as np
import pytest

import oracle
from conftest import sort_rows

pytestmark = pytest.mark.gpu

SUB_KERNEL = "olap::k_join_probe_direct6_sub<512, 256>"


@pytest.fixture(scope="module")
def dev():
    from dpu_olap_amd import Device
    d = Device(0)
    yield d
    d.close()


def _forced(dev, total_bits, fn):
    dev.join_configure(total_bits, 0)
    dev.set_tuning("presplit_sub", 2)
    try:
        return fn()
    finally:
        dev.join_configure(-1, 0)
        dev.set_tuning("presplit_sub", 1)


def _side(ops, t, k, p, n_seg, chunks, empty_sender=True):
    """k, p split over `chunks` senders (odd boundaries, one sender empty), every sender's share grouped into n_seg segments."""
    import torch
    from dpu_olap_amd.dist_join import _chunk_tables
    cuts = np.linspace(0, k.size, chunks + 1).astype(np.int64)
    cuts = np.minimum(cuts | 1, k.size) if chunks > 1 else cuts
    cuts[0], cuts[-1] = 0, k.size
    if chunks > 2 and empty_sender:
        cuts[2] = cuts[1]
    parts, cnt = [], []
    for c in range(chunks):
        kk, pp = k[cuts[c]:cuts[c + 1]], p[cuts[c]:cuts[c + 1]]
        if kk.size:
            z, off = ops.partition_tuples([t(kk)], [t(pp)], n_seg)
        else:
            z, off = ops.empty_tuples(0), [0] * (n_seg + 1)
        parts.append(z)
        cnt.append([off[j + 1] - off[j] for j in range(n_seg)])
    buf = torch.cat(parts)
    pad = ops.empty_tuples(buf.numel() + 2)
    pad[:buf.numel()] = buf
    return (pad[:buf.numel()],) + _chunk_tables(torch.tensor(cnt, dtype=torch.int64))


@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("n_seg,chunks", [(512, 3), (1024, 1), (2048, 5)])
def test_presplit_sub_matches_arrow_golden(dev, golden, case, n_seg, chunks):
    """Ragged chunks, an empty sender, mostly empty sub-partitions (the golden tables are small): 18 radix bits in all, of which
    log2(n_seg) are fixed by the split -- one pass of 7..9 bits is left, the direct table takes the 14 bits below."""
    import torch
    from dpu_olap_amd.dist_join import GpuLocalOps
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    t = lambda a: torch.from_numpy(a.view(np.int32).copy()).to(device)
    fk, y, pk, x = (golden[f"join_{case}_{k}"] for k in ("fk", "y", "pk", "x"))
    L, l_lo, l_hi = _side(ops, t, fk, y, n_seg, chunks)
    R, r_lo, r_hi = _side(ops, t, pk, x, n_seg, chunks)
    ofk, oy, ox, stats = _forced(dev, 18, lambda: ops.join_presplit(L, l_lo, l_hi, R, r_lo, r_hi, n_seg, chunks, 0))
    assert dev.phase_kernel("join-probe") == SUB_KERNEL and dev.get_counter("join_attempts") == 1
    assert stats["passes"] == 1 and stats["nr_partitions"] == 1 << 18 and stats["missing_keys"] == 0 and stats["table_slots"] == 1 << 14
    u = lambda tt: tt.cpu().numpy().view(np.uint32)
    rfk, ry, rx = sort_rows(u(ofk), u(oy), u(ox))
    np.testing.assert_array_equal(rfk, golden[f"join_{case}_out_fk"])
    np.testing.assert_array_equal(ry, golden[f"join_{case}_out_y"])
    np.testing.assert_array_equal(rx, golden[f"join_{case}_out_x"])


@pytest.mark.parametrize("world", [2, 8])
def test_presplit_sub_fused_loopback_digest(dev, world):
    """The fused multi-GPU path on one GPU (as test_multirank_fused_split_loopback_on_one_gpu): every logical rank splits into
    G x S = 512 buckets, rank r joins the S runs it would receive from every sender -- here through the chunk-by-chunk pass.
    The union of the shards digests like the whole join; rows come segment-major; misses and duplicates are counted."""
    import torch
    from dpu_olap_amd import datagen as dg
    from dpu_olap_amd.dist_join import GpuLocalOps, _chunk_tables
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    rows, nb = 1 << 20, 16
    n = rows * nb
    col = lambda: torch.empty(n, dtype=torch.int32, device=device)
    fk, y, pk, x, exp = col(), col(), col(), col(), col()
    dev.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(pk.data_ptr(), n, 0)
    dev.gen_uniform_u32(x.data_ptr(), n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(exp.data_ptr(), fk.data_ptr(), n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(fk.data_ptr(), y.data_ptr(), exp.data_ptr(), n)
    per, S, bits = nb // world, 512 // world, world.bit_length() - 1
    sends = []
    for r in range(world):
        sl = lambda t: [t[(r * per + b) * rows:(r * per + b + 1) * rows] for b in range(per)]
        sends.append(ops.partition_tuples(sl(fk), sl(y), world * S) + ops.partition_tuples(sl(pk), sl(x), world * S))
    tsum, txor, trows = 0, 0, 0
    for r in range(world):
        def recv(ti, oi):
            buf = torch.cat([z[ti][z[oi][r * S]:z[oi][(r + 1) * S]] for z in sends])
            cnt = torch.tensor([[z[oi][r * S + j + 1] - z[oi][r * S + j] for j in range(S)] for z in sends], dtype=torch.int64)
            return (buf,) + _chunk_tables(cnt)
        RL, l_lo, l_hi = recv(0, 1)
        RR, r_lo, r_hi = recv(2, 3)
        ofk, oy, ox, stats = _forced(dev, 18 - bits, lambda: ops.join_presplit(RL, l_lo, l_hi, RR, r_lo, r_hi, S, world, bits))
        assert dev.phase_kernel("join-probe") == SUB_KERNEL and "k_part_scatter_ring6" in dev.phase_kernel("partition-scatter-capacity")
        assert dev.get_counter("join_attempts") == 1 and stats["missing_keys"] == 0 and stats["duplicate_keys"] == 0 and stats["passes"] == 1
        h = oracle.wang_hash_np(ofk.cpu().numpy().view(np.uint32))
        assert ((h >> np.uint32(32 - bits)) == r).all() if bits else True
        seg = (h << np.uint32(bits)) >> np.uint32(32 - (S.bit_length() - 1))   # the split's sub-bucket of every row: never decreases
        assert (np.diff(seg.astype(np.int64)) >= 0).all()
        s_, xo = dev.checksum_rows3_u32(ofk.data_ptr(), oy.data_ptr(), ox.data_ptr(), ofk.numel())
        tsum = (tsum + s_) & ((1 << 64) - 1)
        txor ^= xo
        trows += ofk.numel()
    assert trows == n and (tsum, txor) == expect


def test_presplit_sub_inner_join_duplicates_and_skew_fallback(dev):
    """Inner-join mode drops the foreign keys without a match (order inside the result aside), a duplicate build key overwrites
    and is counted; then a hot foreign key overruns its sub-partition's region: both sides are redone with the exact layout
    (join_attempts == 2) and the rows are the same."""
    import torch
    from dpu_olap_amd.dist_join import GpuLocalOps
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    t = lambda a: torch.from_numpy(a.view(np.int32).copy()).to(device)
    rng = np.random.default_rng(77)
    nr, nl = 1 << 19, 1 << 20
    pk = rng.permutation(1 << 24)[:nr].astype(np.uint32)
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = pk[rng.integers(0, nr, size=nl)]
    fk[rng.choice(nl, size=100, replace=False)] = np.uint32(0xFFFFFFF0)          # no such primary key (pk < 2^24)
    y = np.arange(nl, dtype=np.uint32)
    pk2, x2 = np.concatenate([pk, pk[9:10]]), np.concatenate([x, x[9:10]])
    keep = fk != np.uint32(0xFFFFFFF0)
    order = np.argsort(pk, kind="stable")
    lut = lambda keys: x[order][np.searchsorted(pk[order], keys)]

    def run(fkc):
        L, l_lo, l_hi = _side(ops, t, fkc, y, 64, 4, empty_sender=False)
        R, r_lo, r_hi = _side(ops, t, pk2, x2, 64, 4, empty_sender=False)
        dev.join_set_unmatched(True)
        try:
            return _forced(dev, 18, lambda: ops.join_presplit(L, l_lo, l_hi, R, r_lo, r_hi, 64, 4, 0))
        finally:
            dev.join_set_unmatched(False)
    ofk, oy, ox, stats = run(fk)
    # 18 bits, 6 fixed by the split: 12 are left -> two passes; the chunk-by-chunk form needs ONE pass: not taken here
    assert dev.phase_kernel("join-probe") != SUB_KERNEL
    u = lambda tt: tt.cpu().numpy().view(np.uint32)

    def run512(fkc):
        L, l_lo, l_hi = _side(ops, t, fkc, y, 512, 4, empty_sender=False)
        R, r_lo, r_hi = _side(ops, t, pk2, x2, 512, 4, empty_sender=False)
        dev.join_set_unmatched(True)
        try:
            return _forced(dev, 18, lambda: ops.join_presplit(L, l_lo, l_hi, R, r_lo, r_hi, 512, 4, 0))
        finally:
            dev.join_set_unmatched(False)
    ofk, oy, ox, stats = run512(fk)
    assert dev.phase_kernel("join-probe") == SUB_KERNEL and dev.get_counter("join_attempts") == 1
    assert stats["missing_keys"] == 100 and stats["duplicate_keys"] == 1 and ofk.numel() == int(keep.sum())
    g = sort_rows(u(oy), u(ofk), u(ox))
    np.testing.assert_array_equal(g[0], y[keep])
    np.testing.assert_array_equal(g[1], fk[keep])
    np.testing.assert_array_equal(g[2], lut(fk[keep]))
    hot = fk.copy()
    hot[rng.random(nl) < 0.2] = pk[5]
    keep = hot != np.uint32(0xFFFFFFF0)
    ofk, oy, ox, stats = run512(hot)
    assert dev.get_counter("join_attempts") == 2 and dev.get_counter("join_region_overflows") > 0 and dev.phase_kernel("join-probe") != SUB_KERNEL
    g = sort_rows(u(oy), u(ofk), u(ox))
    np.testing.assert_array_equal(g[0], y[keep])
    np.testing.assert_array_equal(g[1], hot[keep])
    np.testing.assert_array_equal(g[2], lut(hot[keep]))


def test_presplit_sub_groups_at_sf256_refines_the_chunks(dev):
    """The shape a rank of an 8-GPU job sees, at SF=256 (2^29 rows per side in all), default settings: the tables travel as K = 4
    sub-bucket groups, so a group join holds 16 segments x 8 senders = 128 chunks -- fewer than CUs.  The chunk places are cut
    into pieces (one per CU), the pass runs chunk by chunk and the sub-partition probe joins them; rank 0's four group joins
    must give exactly the rows whose key it owns, x = generator(fk)."""
    import torch
    from dpu_olap_amd import datagen as dg
    from dpu_olap_amd.dist_join import GpuLocalOps, _chunk_tables
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    rows, nb, G, K = 2 << 20, 256, 8, 4
    n = rows * nb
    col = lambda: torch.empty(n, dtype=torch.int32, device=device)
    fk, y, pk, x = col(), col(), col(), col()
    dev.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(pk.data_ptr(), n, 0)
    dev.gen_uniform_u32(x.data_ptr(), n, dg.SEED, dg.STREAM_X, 0)
    per, bits, gb = nb // G, 3, 2
    lb = (512 // G // K).bit_length() - 1
    Lw = 1 << lb
    recvL, recvR = [[] for _ in range(K)], [[] for _ in range(K)]
    cntL, cntR = [[] for _ in range(K)], [[] for _ in range(K)]
    for q in range(G):
        sl = lambda t: [t[(q * per + b) * rows:(q * per + b + 1) * rows] for b in range(per)]
        lt, loff = ops.split_tuples(sl(fk), sl(y), bits, gb, lb)
        rt, roff = ops.split_tuples(sl(pk), sl(x), bits, gb, lb)
        for g in range(K):                       # owner 0's runs of group g: buckets (g, 0, 0..Lw)
            b0 = (g * G + 0) * Lw
            recvL[g].append(lt[loff[b0]:loff[b0 + Lw]].clone())
            cntL[g].append([loff[b0 + l + 1] - loff[b0 + l] for l in range(Lw)])
            recvR[g].append(rt[roff[b0]:roff[b0 + Lw]].clone())
            cntR[g].append([roff[b0 + l + 1] - roff[b0 + l] for l in range(Lw)])
        del lt, rt
    del y, x
    total = 0
    for g in range(K):
        RLg, RRg = torch.cat(recvL[g]), torch.cat(recvR[g])
        l_lo, l_hi = _chunk_tables(torch.tensor(cntL[g], dtype=torch.int64))
        r_lo, r_hi = _chunk_tables(torch.tensor(cntR[g], dtype=torch.int64))
        ofk, oy, ox, st = ops.join_presplit(RLg, l_lo, l_hi, RRg, r_lo, r_hi, Lw, G, bits + gb)
        assert dev.phase_kernel("join-probe") == SUB_KERNEL and dev.get_counter("join_attempts") == 1, (dev.phase_kernel("join-probe"), st)
        assert st["missing_keys"] == 0 and st["duplicate_keys"] == 0 and st["passes"] == 1 and st["table_slots"] == 1 << 14
        exp = torch.empty_like(ox)
        dev.gen_uniform_u32_at(exp.data_ptr(), ofk.data_ptr(), ofk.numel(), dg.SEED, dg.STREAM_X)
        assert torch.equal(ox, exp)
        h = oracle.wang_hash_np(ofk[:1 << 16].cpu().numpy().view(np.uint32))
        assert ((h >> np.uint32(32 - bits)) == 0).all() and (((h >> np.uint32(32 - bits - gb)) & np.uint32(K - 1)) == g).all()
        total += ofk.numel()
        del RLg, RRg, ofk, oy, ox, exp
    assert abs(total / (n / G) - 1.0) < 0.01
