#!/usr/bin/env python3
"""Timeline model of the multi-GPU join step (dpu_olap_amd/dist_join.py, fused path) under the two issue orders of
its exchanges -- VERDICT r1 #6: "print the earliest time each group's join can start under each order for G=2/4/8
with a stated link rate".  No GPU needed: the per-rank DEVICE costs are the ones measured on one MI355X with the
real received sizes (scripts/loopback_fused_cost.py -> profiles/r1_loopback_fused_rank_cost.jsonl); the links are
a parameter.

Model (one rank; all ranks are symmetric on uniform data):
  * the device runs one thing at a time, in program order: split of left slice 0 .. H-1, split of the right table,
    then per sub-bucket group g = 0 .. K-1 the join of g = left pass -> right pass -> probe;
  * RCCL executes the rank's collectives in ISSUE order on its own stream, concurrently with the device work; a
    collective starts when the previous one has finished AND its send buffer exists (the split that produces it has
    ended);
  * an all-to-all of b bytes per rank moves b / G to each of the G - 1 peers, each over its own xGMI link, so it
    takes (b / G) / link_rate (+ a fixed latency per collective);
  * the left pass of join g needs ALL left slices of group g, its right pass (and probe) the right table's group g.
Orders:
  table : L slice 0 (g0..gK-1), L slice 1 (...), then R (g0..gK-1)     -- links start after 1/H of the left split
  group : everything is split first; then L(all slices) g0, R g0, L g1, R g1, ...

usage: exchange_order_model.py [--link-gbps 76.8] [--latency-us 30] [--costs profiles/r1_loopback_fused_rank_cost.jsonl]
                               [--rows 4294967296] [--groups 4] [--slices 2]"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def simulate(G, K, H, rows, split_ms, join_phases_ms, link_gbps, latency_us, order):
    """Returns (step_ms, [(group, join_start_ms, left_arrived_ms, right_arrived_ms)], exchange_busy_ms)."""
    per_rank = rows // G                                    # tuples per rank and table
    # device costs: the split of BOTH tables took split_ms; left and right are equal-sized
    s_left_slice, s_right = split_ms / 2 / H, split_ms / 2
    part = (join_phases_ms.get("partition-hist", 0) + join_phases_ms.get("partition-scan", 0) + join_phases_ms.get("partition-scatter", 0)) / 2 / K
    probe = join_phases_ms.get("join-probe", 0) / K
    msg = lambda tuples: tuples * 8 / G / (link_gbps * 1e6) + latency_us * 1e-3      # ms for one all-to-all of `tuples` tuples per rank
    t = 0.0
    ready = {}                                               # (table, slice, group) -> time its send buffer exists
    for h in range(H):
        t += s_left_slice
        for g in range(K):
            ready[("L", h, g)] = t
    t += s_right
    for g in range(K):
        ready[("R", 0, g)] = t
    splits_done = t
    if order == "table":
        issue = [("L", h, g) for h in range(H) for g in range(K)] + [("R", 0, g) for g in range(K)]
    else:
        issue = []
        for g in range(K):
            issue += [("L", h, g) for h in range(H)] + [("R", 0, g)]
        for k in ready:                                      # nothing is issued before the last split has ended
            ready[k] = splits_done
    arrive, link_t, busy = {}, 0.0, 0.0
    for key in issue:
        tuples = per_rank // K // (H if key[0] == "L" else 1)
        start = max(link_t, ready[key])
        link_t = start + msg(tuples)
        busy += msg(tuples)
        arrive[key] = link_t
    dev, rows_out = splits_done, []
    for g in range(K):
        l_arr = max(arrive[("L", h, g)] for h in range(H))
        r_arr = arrive[("R", 0, g)]
        start = max(dev, l_arr)
        end_left = start + part
        end = max(end_left, r_arr) + part + probe
        rows_out.append((g, start, l_arr, r_arr))
        dev = end
    return dev, rows_out, busy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--link-gbps", type=float, default=76.8, help="payload rate of ONE xGMI link in one direction (GB/s); 7 links x ~153 GB/s bidirectional per GPU")
    ap.add_argument("--latency-us", type=float, default=30.0, help="fixed cost per collective")
    ap.add_argument("--costs", default=os.path.join(ROOT, "profiles", "r1_loopback_fused_rank_cost.jsonl"))
    ap.add_argument("--rows", type=int, default=1 << 32)
    ap.add_argument("--groups", type=int, default=4)
    ap.add_argument("--slices", type=int, default=2)
    a = ap.parse_args()
    costs = {}
    for line in open(a.costs):
        d = json.loads(line)
        if d.get("groups") == a.groups:
            costs[d["G"]] = (d["split_ms_per_rank"], d["group_join_phases_total"])
    print(f"# link {a.link_gbps} GB/s per direction and peer, {a.latency_us} us per collective, K = {a.groups} groups, H = {a.slices} left slices, "
          f"{a.rows} rows per side; device costs per rank from {os.path.relpath(a.costs, ROOT)}")
    print("| G | order | join of group g can start at (ms): g0, g1, ... | all exchanges issued by | links busy | step (ms) | vs 1 GPU |")
    print("|---|---|---|---|---|---|---|")
    best = {}
    for G in sorted(costs):
        split_ms, phases = costs[G]
        for order in ("table", "group"):
            step, rows, busy = simulate(G, a.groups, a.slices, a.rows, split_ms, phases, a.link_gbps, a.latency_us, order)
            best.setdefault(G, []).append((step, order))
            starts = ", ".join(f"{r[1]:.1f}" for r in rows)
            print(f"| {G} | {order} | {starts} | {split_ms if order == 'group' else split_ms:.1f} | {busy:.1f} | {step:.1f} | |")
    print()
    for G, v in best.items():
        v.sort()
        print(f"G={G}: best order = {v[0][1]} ({v[0][0]:.1f} ms vs {v[1][0]:.1f} ms)")


if __name__ == "__main__":
    main()
