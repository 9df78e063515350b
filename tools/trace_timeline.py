#!/usr/bin/env python3
"""Compact timeline of one factorization from a rocprofv3 --kernel-trace CSV: per dispatch start / duration (us),
kernel class, workgroups; for the role-1 update launches (lookahead / U1) an estimate of their rate from the grid
(one 128 x 128 block per workgroup).   python tools/trace_timeline.py <kernel_trace.csv> [K] [first] [count]"""
import csv
import sys


def short(name):
    for key, tag in (("update_kernel<double, true, 0>", "bulk"), ("update_kernel<double, true, 1>", "la/u1"),
                     ("update_kernel<double, true, 2>", "upd2"), ("trsm_rows", "trsm"), ("trsm_kernel", "trsm_s"),
                     ("potrf_coop", "potrf"), ("potrf_diag", "diag"), ("copy", "copy"), ("fill", "fill")):
        if key in name:
            return tag
    return name[:24]


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 256)) or 256)
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), grid // max(wg, 1)))
    rows.sort()
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
    # the factorization = from the first potrf to the last kernel
    i0 = next(i for i, r in enumerate(rows) if r[2] == "potrf")
    rows = rows[i0:]
    t0 = rows[0][0]
    tot = {}
    for s, e, n, g in rows:
        tot.setdefault(n, [0, 0.0])
        tot[n][0] += 1
        tot[n][1] += (e - s) / 1e3
    print("classes:", {k: (v[0], round(v[1] / 1e3, 2)) for k, v in tot.items()}, "ms; span", round((max(r[1] for r in rows) - t0) / 1e6, 2), "ms")
    # role-1 launches: split by K guess -- LA has K = 2 nb, U1 K = nb; both print with K = 2 nb rate and K = nb rate
    la = [(s, e, g) for s, e, n, g in rows if n == "la/u1"]
    print("role-1 launches:", len(la))
    for idx, (s, e, g) in enumerate(la[first:first + count]):
        us = (e - s) / 1e3
        fl1 = g * 2.0 * 128 * 128 * nb
        print(f"  #{first + idx:3d} t={((s - t0) / 1e6):9.3f} ms  {us:9.1f} us  wgs {g:6d}  rate if K=nb {fl1 / us / 1e6:6.1f}  if K=2nb {2 * fl1 / us / 1e6:6.1f} TF")
    if "--all" in sys.argv:
        for s, e, n, g in rows[first:first + count]:
            print(f"{(s - t0) / 1e6:10.3f} {(e - s) / 1e3:10.1f} us {n:8s} {g}")


if __name__ == "__main__":
    main()
