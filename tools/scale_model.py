#!/usr/bin/env python3
"""scale_model.py -- critical-path model of the tiled Cholesky on a Pr x Pc grid of MI355X, fed with the kernel
times MEASURED on one GPU (profiles/r02_*, profiles/r03_*, BENCH_r02.json) and the guide's xGMI figures.  No
multi-GPU run exists on this pool (one-GPU boxes), so this is what the first real scaling run is to be held against.

    python tools/scale_model.py            # prints the tables of DESIGN.md section 4

Per step k (nt = N / nb tile columns, r = nt - k - 1 trailing tile rows) a rank of the grid holds about r / Pr tile
rows and r / Pc tile columns of the trailing matrix.  Kernel times:

  bulk update   t_bulk(tiles)  = tiles * 2 nb^3 / R_bulk(tiles)        R_bulk saturates at the in-situ rate of the
                                                                       persistent launch; small launches pay a fill /
                                                                       drain time (measured: 70.7 one block per
                                                                       workgroup, 68.8 persistent, 62 on 256 slots)
  panel TRSM    t_trsm(rows)   = rows * nb^3 / R_trsm                  alone on the GPU (55 TFlop/s at nb = 1024)
  tile POTRF    t_potrf                                                0.92 ms alone / 3.15 ms beside the bulk (nb 1024)
  broadcasts    t_bc(bytes, members) = lat + bytes / bw                xGMI: one link per peer, root -> members - 1 peers
                                                                       concurrently (fully connected node)

Schedules (both take the update + solve flops of a step at R_eff, the whole-factorization in-situ rate of ONE GPU --
68.0 TFlop/s at C2 (BENCH_r02), 63.1 for z N=32768 nb=512 (profiles/r02_final_*) -- so that the 1 x 1 row reproduces the
measured run by construction; what the model adds is the dependency chain of a step at stand-alone kernel speeds):
  early  (what the grid executor issues today, runtime.cpp "early diagonal"): the panel TRSM runs on the main stream
         in front of the bulk at full width (serial with it), POTRF(k+1) beside the bulk on shared compute units:
             T_k = max( work_k / R_eff + t_trsm_alone ,  potrf_in_situ + bc(diag) + trsm(1) + bc(head) + upd(1) ,
                        bc(panel) + bc(panel^T) )
  pairs  (the one-process order extended to grids: panel chain on the side stream with an adaptive slot reservation,
         POTRF strips on compute units of their own -- VERDICT r02 items 3b, 4a, 4b):
             T_k = max( work_k / R_eff ,  potrf_alone + bc(diag) + t_trsm_alone + bc(panel) + bc(panel^T) + t_la_alone )
"""
import argparse

PEAK = 78.6e12  # fp64 MFMA peak per MI355X


class Rates:
    def __init__(self, nb, cx):
        self.nb, self.cx = nb, cx
        f = 4.0 if cx else 1.0
        self.flop_gemm_tile = f * 2.0 * nb ** 3
        self.flop_trsm_tile = f * 1.0 * nb ** 3
        # measured on one MI355X (profiles/r02_final_*, profiles/r03_update_kernel_wide4_lean_ab_timing.txt)
        if nb >= 1024:
            self.r_bulk = 70.7e12           # one block per workgroup, alone (profiles/r03_update_kernel_wide4_lean_ab_timing.txt)
            self.r_eff = 69.3e12            # whole factorization on one GPU (round 3, second half: 68.7-69.8)
            self.r_trsm = 55.0e12           # trsm_rows_kernel alone (BENCH_r02 trsm_panel.achieved_TFlops)
            # in situ: shared compute units with the neighbouring bulk workgroups sitting out 1.0 ms (2.15 ms with
            # raised wave priority alone; profiles/r03_potrf_yield_ab.txt); on exclusive compute units 0.87 ms
            # (profiles/r03_exclusive_cus_ab.txt) -- the grid order's reservation with a device-side transport
            # round 4 (profiles/r04_chain_bench.txt): grid order, strips registered for the yield (on grids again since
            # round 4): 0.62 ms in situ, 0.60 alone; a ONE-tile panel solve 0.34 ms (16 workgroups, MFMA-latency bound),
            # the update of the next column alone 0.16 ms
            self.potrf_alone, self.potrf_insitu, self.potrf_excl = 0.60e-3, 0.62e-3, 0.60e-3
            self.potrf_pairs = 0.97e-3      # in situ under the one-process (pairs) order
            self.trsm_one, self.upd_one = 0.34e-3, 0.16e-3
        else:
            self.r_bulk = (67.8e12 if cx else 62.0e12)   # z in situ 67.8 (4N^3/3 model); d nb=512 bulk 0.79 of peak
            self.r_eff = (65.1e12 if cx else 60.6e12)    # round 3: z N=32768 nb=512 65.1, C1 60.6
            self.r_trsm = (30.0e12 if cx else 45.0e12)
            # round 4 (profiles/r04_chain_bench.txt, z nb = 512): 0.525 ms in situ under the grid order, 0.54 alone,
            # one-tile solve 0.30 ms, next-column update 0.15-0.19 ms
            self.potrf_alone, self.potrf_insitu = ((0.54e-3, 0.525e-3) if cx else (0.375e-3, 0.37e-3))
            self.potrf_excl = 0.525e-3 if cx else 0.32e-3
            self.potrf_pairs = 0.945e-3 if cx else 0.37e-3
            self.trsm_one, self.upd_one = ((0.30e-3, 0.17e-3) if cx else (0.12e-3, 0.06e-3))
        self.fill = 60e-6                   # fill + drain of a bulk launch (70.7 vs 68.8 TFlop/s at 33 ms: ~0.9 ms / 15 waves)
        self.elem = 16 if cx else 8

    def t_bulk(self, tiles):
        if tiles <= 0:
            return 0.0
        return tiles * self.flop_gemm_tile / self.r_bulk + self.fill

    def t_trsm(self, rows):
        # a tile is 16 (nb = 1024) workgroups of the row-owner kernel: below a GPU-full of tiles the launch takes as long
        # as ONE wave needs for its 16 rows (measured: trsm_one), not flops / rate
        if rows <= 0:
            return 0.0
        return max(self.trsm_one, rows * self.flop_trsm_tile / self.r_trsm + 20e-6)


class Fabric:
    # MI355X_MICROARCH.md / task statement: 7 xGMI links x ~153 GB/s per GPU (bidirectional) -> ~64 GB/s one way per
    # peer after protocol overhead; a collective costs a launch + rendezvous
    def __init__(self, bw=64e9, lat=15e-6):
        self.bw, self.lat = bw, lat

    def bcast(self, nbytes, members):
        if members <= 1 or nbytes <= 0:
            return 0.0
        return self.lat + nbytes / self.bw   # direct peer copies / pipelined ring: one link time


def model(n, nb, pr, pc, cx, schedule, fab, reserve=32.0 / 512.0):
    rt = Rates(nb, cx)
    nt = n // nb
    tile_bytes = nb * nb * rt.elem
    total = 0.0
    chain_bound = 0
    for k in range(nt):
        r = nt - k - 1
        rows = -(-r // pr)                     # local tile rows of the panel (worst rank)
        cols = -(-r // pc)
        tiles = (rows * cols + 1) // 2 + (min(rows, cols) + 1) // 2   # local trailing tiles incl. diagonal ones
        t_bulk = rt.t_bulk(max(tiles - rows, 0))                         # minus the lookahead column
        t_la = rt.t_bulk(rows) if r > 0 else 0.0
        t_trsm = rt.t_trsm(rows)
        bc_diag = fab.bcast(tile_bytes, pr)
        bc_panel = fab.bcast(rows * tile_bytes, pc)
        bc_panel_t = fab.bcast(cols * tile_bytes, pr)
        work = (tiles * rt.flop_gemm_tile + rows * rt.flop_trsm_tile) / rt.r_eff   # (launch overheads are in R_eff)
        if schedule in ("early", "early-x"):
            chain = (rt.potrf_excl if schedule == "early-x" else rt.potrf_insitu) + bc_diag + rt.t_trsm(1) + fab.bcast(tile_bytes, pc) + rt.upd_one
            step = max(tiles * rt.flop_gemm_tile / rt.r_eff + t_trsm, chain, bc_panel + bc_panel_t)
        else:
            chain = rt.potrf_pairs + bc_diag + t_trsm + bc_panel + bc_panel_t + t_la
            step = max(work, chain)
        if step > work * 1.02 + 1e-9:
            chain_bound += 1
        total += step
    total += rt.potrf_alone                                  # the first diagonal tile is on nobody's shadow
    flops = (4.0 if cx else 1.0) * n ** 3 / 3.0
    tf = flops / total / 1e12
    return total, tf, tf * 1e12 / (pr * pc * PEAK), chain_bound, nt


def model_c5(pr, pc, fab):
    """BASELINE configs[4]: the Hermitian eigensolver at N = 20480, nb = 512 (band 128), fp64, on a pr x pc grid.  Stage
    times of one MI355X measured in round 4 (profiles/r04_eigensolver_N20480_nb512_bench.txt: 0.350 + 0.354 + 0.292 + 0.462 + 0.309 s), split into the part every
    rank repeats (replicated) and the part the grid shares; communication from the guide's xGMI figures.  What is
    replicated today: the panel factorization of reduction_to_band inside the owning process column (and the T / W / W2
    products every rank recomputes), band_to_tridiagonal (one persistent bulge-chasing launch per rank),
    the deflation / secular part of the divide & conquer solver; bt_band_to_tridiagonal is shared over process COLUMNS
    only (each rank applies it to all rows of its columns)."""
    n, b = 20480, 128
    P = pr * pc
    npanels = (n - b - 1 + b - 1) // b
    # stage 1: panel chain (blocked: 0.65 ms per panel) + T / W / W2 (0.10 ms) replicated; hemm + her2k shared
    t1_rep, t1_sh = npanels * 0.70e-3, 0.241
    # per panel: the factored panel along the process row, the all-reduce of X over the grid (m x b doubles each, m ~ n / 2)
    msg = (n / 2) * b * 8
    t1_comm = npanels * (fab.bcast(msg, pc) + (2 * fab.bcast(msg, P) if P > 1 else 0.0))
    t2 = 0.354 + (2 * fab.bcast(2 * b * n * 8, P) if P > 1 else 0.0)
    # stage 3: Q U of the large merges shared (0.17 s of GEMMs), the rest replicated; all-gather of n x n doubles at the top
    # level, half of that at each level below (seven links per GPU)
    t3 = 0.12 + 0.17 / P + ((2.0 * n * n * 8 * (P - 1) / P) / (7 * fab.bw) + 4 * fab.lat if P > 1 else 0.0)
    t4 = 0.462 / pc
    t5 = 0.309 / P + (npanels / 4) * fab.bcast((n / 2) * 512 * 8, pc) * (1 if P > 1 else 0)
    t1 = t1_rep + t1_sh / P + t1_comm
    return t1, t2, t3, t4, t5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bw", type=float, default=64e9)
    ap.add_argument("--lat", type=float, default=15e-6)
    a = ap.parse_args()
    fab = Fabric(a.bw, a.lat)
    for name, n, nb, cx in (("C2  d N=65536 nb=1024", 65536, 1024, False), ("C3  z N=32768 nb=512", 32768, 512, True)):
        print(f"## {name}   (xGMI {a.bw / 1e9:.0f} GB/s per peer, {a.lat * 1e6:.0f} us per collective)")
        print("| grid | schedule | time [ms] | TFlop/s | fraction of N x 78.6 | steps bound by the chain |")
        print("|---|---|---|---|---|---|")
        for pr, pc in ((1, 1), (1, 2), (2, 2), (2, 4)):
            for sched in ("early", "early-x", "pairs"):
                t, tf, frac, cb, nt = model(n, nb, pr, pc, cx, sched, fab)
                print(f"| {pr}x{pc} | {sched} | {t * 1e3:.0f} | {tf:.1f} | {frac:.3f} | {cb} / {nt} |")
        print()
    print(f"## C5  d hermitian_eigensolver N=20480 nb=512 band=128   (xGMI {a.bw / 1e9:.0f} GB/s per peer, {a.lat * 1e6:.0f} us per collective)")
    print("| grid | reduction_to_band | band_to_tridiagonal | tridiagonal_eigensolver | bt_band_to_tridiagonal | bt_reduction_to_band | total [s] | speed-up |")
    print("|---|---|---|---|---|---|---|---|")
    base = None
    for pr, pc in ((1, 1), (1, 2), (2, 2), (2, 4)):
        st = model_c5(pr, pc, fab)
        tot = sum(st)
        base = base or tot
        print(f"| {pr}x{pc} | " + " | ".join(f"{x:.3f}" for x in st) + f" | {tot:.2f} | {base / tot:.2f} |")
    print()


if __name__ == "__main__":
    main()
