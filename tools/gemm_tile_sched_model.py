"""What would cheaper half tiles and a cost-aware tile assignment buy the persistent 256 x 256 GEMM?  A host-side model (no GPU).

The kernel (csrc/gemm256.hip) runs one workgroup per CU (256) and assigns output tiles statically: workgroup w takes tiles w, w + 256, ...
of the linear tile order.  Outputs whose width is not a multiple of 256 (1408 = 5.5 tiles, 4224 = 16.5) end in a half-empty column tile,
and M = 53376 = 208.5 row tiles ends in a half-empty row tile; today every tile costs the same (the padding multiplies zeros).
This script prices each GEMM of a 1B block (forward, dgrad; wgrad is grouped and K-long, see below) under four policies:
  static/full   today: static round-robin, padded tiles at full cost
  static/cheap  padded tiles cost rho of a full tile (the kernel change alone)
  greedy/cheap  cheap padded tiles + dynamic assignment, longest tile first (a work queue: the scheduler change on top)
  tail-split    cheap padded tiles ordered LAST + the full tiles that remain in the last round cut into S K-slices (the kernel's existing
                tail split, today only used when the last round is mostly empty): the last round then holds units of ~rho each
  ideal         total work / 256 (no quantisation at all)
in units of one full tile (K steps x per-step time + fixed per-tile cost folded in), and sums the step-level effect with the measured
launch times of profiles/r3_kernel_stats_b128_v3_final_sources.md.   python tools/gemm_tile_sched_model.py [--rho 0.55] [--batch 128]
"""
import argparse
import heapq
import json


def tiles(M, N, rho):
    """list of tile costs (full tile = 1): half-width / half-height tiles cost rho (both: rho * rho is optimistic -> max(rho * rho, 0.3))"""
    tm, tn = -(-M // 256), -(-N // 256)
    half_m = (M % 256) != 0 and (M % 256) <= 128
    half_n = (N % 256) != 0 and (N % 256) <= 128
    out = []
    for i in range(tm):
        for j in range(tn):
            hm = half_m and i == tm - 1
            hn = half_n and j == tn - 1
            out.append((1.0, (rho if hm else 1.0) * (rho if hn else 1.0)))
    return out          # (cost today, cost with cheap padded tiles)


def static_makespan(costs, wgs=256):
    load = [0.0] * wgs
    for i, c in enumerate(costs):
        load[i % wgs] += c
    return max(load)


def greedy_makespan(costs, wgs=256):
    h = [0.0] * wgs
    heapq.heapify(h)
    for c in sorted(costs, reverse=True):
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)


def today_makespan(n_tiles, wgs=256, xch=0.04):
    """static round-robin at full cost, with the kernel's existing tail split: when the tiles of the last round times S (<= 4) still fit the
    workgroups, they are cut into S K-slices (csrc/gemm256.hip g2_split_plan)"""
    rounds, left = divmod(n_tiles, wgs)
    if left == 0:
        return float(rounds)
    for S in (4, 3, 2):
        if rounds >= 1 and left * S <= wgs:
            return rounds + 1.0 / S + xch
    return rounds + 1.0


def tail_split_makespan(costs_cheap, wgs=256, xch=0.04):
    """full tiles first in whole rounds; the last round = leftover full tiles + all padded (cheap) tiles, every unit of it optionally cut
    into S = 1..4 K-slices when the slices fit the workgroups; the round lasts as long as its largest unit (+ the slice exchange)"""
    full = [c for c in costs_cheap if c >= 0.999]
    part = [c for c in costs_cheap if c < 0.999]
    rounds, left = divmod(len(full), wgs)
    last = [1.0] * left + part
    if not last:
        return float(rounds)
    best = None
    for S_full in (1, 2, 3, 4):
        for S_part in (1, 2):
            units = [1.0 / S_full + (xch if S_full > 1 else 0.0)] * (left * S_full) + [c / S_part + (xch if S_part > 1 else 0.0) for c in part for _ in range(S_part)]
            units.sort(reverse=True)
            t, i = 0.0, 0
            while i < len(units):          # units beyond one round spill into further rounds
                t += units[i]
                i += wgs
            best = t if best is None else min(best, t)
    return rounds + best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rho", type=float, default=0.55, help="cost of a half-width (or half-height) tile relative to a full one")
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    M = a.batch * 417
    D, Q, H = 1408, 4224, 6144
    # (name, N, K, launches per step of 40 blocks, measured us per launch at B = 128 or None)
    gemms = [("qkv fwd", Q, D, 40), ("proj fwd", D, D, 40), ("fc1 fwd (+GELU)", H, D, 40), ("fc2 fwd", D, H, 40),
             ("qkv dgrad", D, Q, 40), ("proj dgrad", D, D, 40), ("fc1 dgrad", D, H, 40), ("fc2 dgrad (x gelu')", H, D, 40)]
    rows, tot = [], dict(today=0.0, cheap=0.0, greedy=0.0, tail=0.0, ideal=0.0)
    for name, N, K, n in gemms:
        t = tiles(M, N, a.rho)
        today = today_makespan(len(t))
        cheap = min(today, static_makespan([c[1] for c in t]))          # (where the existing tail split already applies it stays)
        greedy = min(today, greedy_makespan([c[1] for c in t]))
        tail = tail_split_makespan([c[1] for c in t])
        useful = (M / 256.0) * (N / 256.0)                       # tiles' worth of real work
        ideal = useful / 256.0
        w = K / 1408.0 * n                                        # weight: K steps x launches (a tile's time is ~proportional to K)
        for k, v in (("today", today), ("cheap", cheap), ("greedy", greedy), ("tail", tail), ("ideal", ideal)):
            tot[k] += v * w
        rows.append(dict(gemm=name, N=N, K=K, tiles=len(t), today=round(today, 3), static_cheap=round(cheap, 3), greedy_cheap=round(greedy, 3),
                         tail_split=round(tail, 3), ideal=round(ideal, 3)))
    print(f"# persistent 256 x 256 GEMM, M = {M} (batch {a.batch}), 256 workgroups, rho = {a.rho}: makespan in full-tile units\n")
    print("| GEMM | N | K | tiles | static / full (today) | static / cheap | greedy / cheap | tail-split / cheap | ideal |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        print(f"| {r['gemm']} | {r['N']} | {r['K']} | {r['tiles']} | {r['today']} | {r['static_cheap']} | {r['greedy_cheap']} | {r['tail_split']} | {r['ideal']} |")
    print("\nK-weighted sum over a block's eight forward / dgrad GEMMs, relative to today:")
    for k in ("cheap", "greedy", "tail", "ideal"):
        print(f"  {k:7s} {tot[k] / tot['today']:.4f}")
    print(json.dumps(dict(batch=a.batch, rho=a.rho, relative={k: round(tot[k] / tot['today'], 4) for k in tot})))


if __name__ == "__main__":
    main()
