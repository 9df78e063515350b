#!/usr/bin/env python3
"""Device idle-time analysis of a rocprofv3 --kernel-trace CSV: union of kernel intervals, gaps between
them, and which kernels bracket the largest gaps.   python tools/trace_gaps.py <kernel_trace.csv> [t0_frac]"""
import csv
import sys
from collections import defaultdict


def short(name):
    for key in ("update_kernel<double, true, 0>", "update_kernel<double, true, 1>", "update_kernel<double, true, 2>",
                "trsm_kernel", "potrf_coop", "potrf_diag", "copyBuffer", "fillBuffer", "layout", "max_norm"):
        if key in name:
            return key
    return name[:40]


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # the last factorization: from the last-but-one... simply analyse the window of the last N kernels after
    # the final big fill/copy; the caller passes the window as fractions of the trace
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    w0, w1 = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
    rows = [r for r in rows if r[0] >= w0 and r[1] <= w1]
    busy_by = defaultdict(int)
    cur_s, cur_e, last = rows[0][0], rows[0][1], rows[0][2]
    gaps = []
    busy = 0
    for s, e, n in rows:
        busy_by[n] += e - s
    for s, e, n in rows[1:]:
        if s > cur_e:
            gaps.append((s - cur_e, last, n, cur_e))
            busy += cur_e - cur_s
            cur_s, cur_e, last = s, e, n
        elif e > cur_e:
            cur_e, last = e, n
    busy += cur_e - cur_s
    span = max(r[1] for r in rows) - rows[0][0]
    print(f"window span {span / 1e6:.3f} ms, device busy (union) {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
    for n, v in sorted(busy_by.items(), key=lambda kv: -kv[1]):
        print(f"   {n:34s} {v / 1e6:10.3f} ms")
    by_pair = defaultdict(lambda: [0, 0])
    for g, a, b, _ in gaps:
        by_pair[(a, b)][0] += g
        by_pair[(a, b)][1] += 1
    print("gaps by (kernel before -> kernel after):")
    for (a, b), (g, c) in sorted(by_pair.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"   {a:32s} -> {b:32s} {g / 1e6:8.3f} ms in {c} gaps")
    # exclusive time: intervals where ONLY one kind of kernel runs
    ev = []
    for s, e, n in rows:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort()
    active = defaultdict(int)
    excl = defaultdict(int)
    prev = ev[0][0]
    for t, d, n in ev:
        live = [k for k, v in active.items() if v > 0]
        if len(live) == 1:
            excl[live[0]] += t - prev
        active[n] += d
        prev = t
    print("time with exactly one kernel kind on the device:")
    for n, v in sorted(excl.items(), key=lambda kv: -kv[1]):
        print(f"   {n:34s} {v / 1e6:10.3f} ms")


if __name__ == "__main__":
    main()
