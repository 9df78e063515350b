#!/usr/bin/env python3
"""Where one factorization spends its time, from a rocprofv3 --kernel-trace CSV: time per kernel class, and the
gaps of the "full GPU" sequence (bulk launches + the role-1 launches that run at more than min_rate) -- the time in
which only panel kernels (POTRF / TRSM / U1 in the free slots) are running.
   python tools/trace_gaps2.py <kernel_trace.csv> <flops_per_workgroup_block e.g. 2*128*128*nb> [min_rate_TF]"""
import csv
import sys


def main():
    per_wg = float(eval(sys.argv[2]))
    min_rate = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        n = r["Kernel_Name"]
        tag = ("bulk" if "update_kernel" in n and ", 0>" in n else "role1" if "update_kernel" in n and ", 1>" in n else
               "trsm" if "trsm" in n else "potrf" if "potrf" in n else "other")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
    rows.sort()
    rows = rows[next(i for i, r in enumerate(rows) if r[2] == "potrf"):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cls = {}
    for s, e, t, g in rows:
        c = cls.setdefault(t, [0, 0.0])
        c[0] += 1
        c[1] += (e - s) / 1e6
    print("span %.2f ms; per class (launches, summed ms):" % ((t1 - t0) / 1e6), {k: (v[0], round(v[1], 1)) for k, v in cls.items()})
    main_seq = []
    for s, e, t, g in rows:
        if t == "bulk":
            main_seq.append((s, e, t))
        elif t == "role1" and g * per_wg / ((e - s) / 1e9) / 1e12 > min_rate:
            main_seq.append((s, e, "role1-fast"))
    main_seq.sort()
    busy = {"bulk": 0.0, "role1-fast": 0.0}
    for s, e, t in main_seq:
        busy[t] += (e - s) / 1e6
    gaps, cur = [], main_seq[0][1]
    for s, e, t in main_seq[1:]:
        if s > cur:
            gaps.append(((s - cur) / 1e6, (cur - t0) / 1e6))
        cur = max(cur, e)
    tot = sum(g[0] for g in gaps)
    head = (main_seq[0][0] - t0) / 1e6
    tail = (t1 - cur) / 1e6
    print("full-GPU sequence: bulk %.1f ms + fast role-1 %.1f ms; gaps %.1f ms in %d pieces (+ %.2f ms before the first, %.2f ms after the last)"
          % (busy["bulk"], busy["role1-fast"], tot, len(gaps), head, tail))
    q = (t1 - t0) / 1e6 / 4
    for i in range(4):
        print("  gaps in quarter %d of the run: %.1f ms" % (i + 1, sum(g[0] for g in gaps if i * q <= g[1] < (i + 1) * q)))


if __name__ == "__main__":
    main()
