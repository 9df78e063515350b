#!/usr/bin/env python3
"""bench.py -- fp64 tiled Cholesky TFlop/s on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is ONE factorization of the synthetic SPD matrix (set_random_hermitian_positive_definite,
include/dlaf/util_matrix.h:498-501) that is already resident in HBM in tile layout when the
timed region starts -- the window of the reference's miniapp (miniapp/miniapp_cholesky.cpp:137-155).
Default workload: the configuration the metric is quoted on, d N=65536 nb=1024 uplo=L, which fits one
GPU (32 GiB); grids 1x1, 1x2, 2x2, 2x4 (column-major rank order like the miniapp, :113).
Flop model: N^3/3 (miniapp_cholesky.cpp:157-162).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X fp64 matrix peak: 256 CU x 4 SIMD x (16x16x4 MFMA = 2048 flop / 64 cycles) x 2.4 GHz
# = 78.6 TFlop/s (vendor datasheet figure; issue rate confirmed by tools/mfma_f64_rate.hip)
PEAK_FP64_MFMA_TFLOPS = 78.6
GRIDS = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4), 3: (1, 3), 6: (2, 3)}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    # --matrix-size / --block-size as in the reference miniapp (miniapp_cholesky.cpp:223-224);
    # the short forms are for single-process use (torchrun's own parser trips over "--n")
    p.add_argument("--matrix-size", "--n", dest="n", type=int, default=65536)
    p.add_argument("--block-size", "--nb", dest="nb", type=int, default=1024)
    p.add_argument("--type", default="d", choices=["s", "d", "c", "z"])
    p.add_argument("--uplo", default="L", choices=["L", "U"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-n", type=int, default=16384)
    p.add_argument("--cpu-nb", type=int, default=512)
    p.add_argument("--check", action="store_true", help="check_cholesky of the miniapp on the device after the timed runs (any size, any grid)")
    p.add_argument("--no-check", action="store_true", help="skip the (untimed) device-side residual check at N = 1")
    p.add_argument("--no-trsm-profile", action="store_true", help="skip the stand-alone panel-TRSM timing (PMC passes: only the factorization's own launches are counted)")
    p.add_argument("--no-red2band", action="store_true",
                   help="skip the extra (untimed-for-the-metric) reduction_to_band line at N = 1")
    p.add_argument("--no-eigensolver", action="store_true",
                   help="skip the extra (untimed-for-the-metric) whole-eigensolver line at N = 1")
    p.add_argument("--r2b-n", type=int, default=20480)
    p.add_argument("--r2b-nb", type=int, default=512)
    p.add_argument("--transport", default="rccl", choices=["rccl", "host", "peer"],
                   help="host = gloo-staged broadcasts: lets several ranks rehearse the N > 1 path on ONE GPU; "
                        "peer = device-to-device copies out of hipIpc-mapped staging buffers, control messages over gloo")
    return p.parse_args()


def cpu_baseline(args):
    """The reference's CPU path restated (oracle/baseline_blas.py): the same right-looking tile DAG, one
    tile task per host thread, single-threaded vendor BLAS/LAPACK per tile -- timed on this box's host
    cores on a bounded sample of the same workload and the same generator.  Checker-side code, used here
    ONLY as the reported CPU baseline; the oracle's plain-C tile kernels are timed too for reference."""
    from oracle import baseline_blas, oracle
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))  # the GPU box's CPU share for one GPU
    n, nb = args.cpu_n, args.cpu_nb
    a = oracle.set_random_hpd(n, nb, np.float64)
    dt, tiles = baseline_blas.tiled_cholesky_blas(a, nb, threads)
    out = {"value": (n ** 3 / 3) / dt / 1e12, "unit": "TFlop/s", "cores": threads, "kind": "port",
           "sample": f"fp64 lower Cholesky N={n} nb={nb}, reference generator, right-looking tile DAG with one "
                     f"single-threaded MKL BLAS/LAPACK tile task per host thread, {dt:.2f} s wall"}
    try:
        nc, nbc = 4096, 256
        ac = oracle.set_random_hpd(nc, nbc, np.float64)
        t0 = time.perf_counter()
        assert oracle.baseline_cholesky_d(ac, nbc, threads) == 0
        out["plain_c_tile_kernels_TFlops"] = round((nc ** 3 / 3) / (time.perf_counter() - t0) / 1e12, 5)
    except Exception:
        pass
    return out


def red2band_line(dlaf, grid, n, nb, runs=2, mats=None):
    """SURVEY.md 8(f)4 / BASELINE configs[4], first stage on one GPU: reduction_to_band of a random Hermitian matrix
    (the reference's set_random_hermitian = the SPD generator without the 2 n on the diagonal) resident in HBM, band =
    get_band_size(nb) as the reference's eigensolver picks it; flop model of miniapp_reduction_to_band.cpp:163-168.
    An extra line next to the metric, not part of it."""
    band = dlaf.get_band_size(nb)
    a = np.zeros((n, n), dtype=np.float64, order="F")
    dlaf.set_random_hermitian_positive_definite(grid, a, n, nb)
    a[np.arange(n), np.arange(n)] -= 2.0 * n
    ref, work = mats if mats is not None else (dlaf.DeviceMatrix(grid, np.float64, "L", n, nb),
                                               dlaf.DeviceMatrix(grid, np.float64, "L", n, nb))
    ref.upload(a)
    del a
    best_ms = None
    for r in range(runs + 1):
        work.copy_from(ref)
        dlaf.reduction_to_band_device(work, band)
        ms, flops = dlaf.red2band_profile()
        if r > 0:
            best_ms = ms if best_ms is None else min(best_ms, ms)
    ref.close()
    work.close()
    tf = flops / best_ms / 1e9
    return {"workload": f"reduction_to_band_d N={n} nb={nb} band={band} (BASELINE configs[4], first stage, 1 GPU)",
            "value": round(tf, 3), "unit": "TFlop/s", "ms": round(best_ms, 2), "runs": runs,
            "flop_model": "2 (2/3 n^3 - n^2 nb)  (miniapp_reduction_to_band.cpp:163-168)",
            "fraction_of_fp64_mfma_peak": round(tf / PEAK_FP64_MFMA_TFLOPS, 4)}


def eigensolver_line(dlaf, grid, n, nb, runs=2):
    """BASELINE configs[4] on one GPU: the whole Hermitian eigensolver (reduction_to_band, band_to_tridiagonal,
    tridiagonal_eigensolver, bt_band_to_tridiagonal, bt_reduction_to_band; eigensolver/impl.h:38-55) through the
    reference's C entry dlaf_symmetric_eigensolver_d on a random symmetric matrix.  Reported: the sum of the per-stage
    device times (HIP events; operands resident, like the window of miniapp_eigensolver.cpp:150-165 whose mirrors are
    created outside the timer), the stages, the wall time of the host-array entry (PCIe staging included) and a
    sampled check of the result.  An extra line next to the metric, not part of it."""
    import time as _t
    a0 = np.zeros((n, n), dtype=np.float64, order="F")
    dlaf.set_random_hermitian_positive_definite(grid, a0, n, nb)
    a0[np.arange(n), np.arange(n)] -= 2.0 * n
    names = ["reduction_to_band", "band_to_tridiagonal", "tridiagonal_eigensolver", "bt_band_to_tridiagonal",
             "bt_reduction_to_band"]
    best = None
    for r in range(runs + 1):
        a = a0.copy(order="F")
        t0 = _t.time()
        w, z = dlaf.hermitian_eigensolver(grid, "L", a, nb)
        wall = _t.time() - t0
        ms = dlaf.eigensolver_profile()
        del a
        if r > 0 or runs == 0:
            if best is None or sum(ms) < sum(best[0]):
                best = (ms, wall)
    ms, wall = best
    cols = np.unique(np.concatenate([np.arange(0, n, max(1, n // 16)), [n - 1]]))
    zc = z[:, cols]
    az = np.tril(a0) @ zc + np.tril(a0, -1).T @ zc
    res = float(np.abs(az - zc * w[cols][None, :]).max())
    orth = float(np.abs(zc.T @ zc - np.eye(len(cols))).max())
    err = 2 * float(np.finfo(np.float64).eps)
    return {"workload": f"hermitian_eigensolver_d N={n} nb={nb} band={dlaf.get_band_size(nb)} (BASELINE configs[4], 1 GPU)",
            "value": round(sum(ms) / 1e3, 4), "unit": "s", "higher_is_better": False,
            "stages_ms": {k: round(v, 1) for k, v in zip(names, ms)},
            "wall_s_host_arrays_incl_pcie": round(wall, 3), "runs": runs,
            "sampled_check": {"columns": int(len(cols)), "max|A z - w z|": res, "bar_2_n_err_wmax": 2 * n * err * float(np.abs(w).max()),
                              "max|Z^T Z - I|": orth, "bar_10_n_err": 10 * n * err, "sorted": bool(np.all(np.diff(w) >= 0))}}


def host_grid(dlaf, dist, torch, nprow, npcol):
    """Column-major grid whose broadcasts go through gloo (rehearsal transport)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    myrow, mycol = rank % nprow, rank // nprow
    rank_of = lambda r, c: c * nprow + r  # noqa: E731
    rows = [dist.new_group([rank_of(r, c) for c in range(npcol)]) for r in range(nprow)]
    cols = [dist.new_group([rank_of(r, c) for r in range(nprow)]) for c in range(npcol)]

    def bcast(axis, root, buf):
        t = torch.frombuffer(buf, dtype=torch.uint8)
        if axis == 0:
            dist.broadcast(t, src=rank_of(myrow, root), group=rows[myrow])
        else:
            dist.broadcast(t, src=rank_of(root, mycol), group=cols[mycol])

    return dlaf.Grid.host(world, rank, nprow, npcol, "C", bcast, dist.barrier)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: dla_future_amd has no CPU path")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    if world > 1:
        if args.transport == "rccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group("gloo")

    if args.transport == "peer":
        os.environ["DLAF_MI355X_TRANSPORT"] = "peer"   # read when the grid's transport is built (csrc/host/transport_peer.cpp)
    import dla_future_amd as dlaf
    dlaf.initialize()
    nprow, npcol = GRIDS.get(world, (1, world))
    if world == 1:
        grid = dlaf.Grid.single()
    elif args.transport == "rccl":
        grid = dlaf.Grid.from_torch(nprow, npcol, "C")
    else:
        grid = host_grid(dlaf, dist, torch, nprow, npcol)

    n, nb = args.n, args.nb
    dt = {"s": np.float32, "d": np.float64, "c": np.complex64, "z": np.complex128}[args.type]
    lrows, lcols = grid.local_shape(n, nb)
    t_gen = time.perf_counter()
    a = np.zeros((max(1, lrows), max(1, lcols)), dtype=dt, order="F")[:lrows, :lcols]
    dlaf.set_random_hermitian_positive_definite(grid, a, n, nb)
    t_gen = time.perf_counter() - t_gen
    # The two matrices of the reduction_to_band line are allocated FIRST: device memory handed out after the release of
    # the factorization's 137 GB is slower on this stack -- the same stage 407 ms on matrices allocated behind that release,
    # 344 ms on matrices allocated before it or in a fresh process (tools/diag_r2b_after_alloc.py)
    r2b_mats = None
    if world == 1 and not args.no_red2band and args.type == "d":
        r2b_mats = (dlaf.DeviceMatrix(grid, np.float64, "L", args.r2b_n, args.r2b_nb),
                    dlaf.DeviceMatrix(grid, np.float64, "L", args.r2b_n, args.r2b_nb))
    ref = dlaf.DeviceMatrix(grid, dt, args.uplo, n, nb)
    t_up = time.perf_counter()
    ref.upload(a)
    t_up = time.perf_counter() - t_up
    del a

    # work matrices: as many as HBM holds (at most one per timed step); a reused one is restored from the
    # pristine copy outside the timer (the miniapp takes a fresh copy of the input outside its timer, :133-141)
    local_bytes = lrows * lcols * np.dtype(dt).itemsize
    free_b, _total = torch.cuda.mem_get_info()
    npool = max(1, min(args.steps, int((free_b * 0.8) // max(local_bytes * 1.05, 1))))
    pool = [dlaf.DeviceMatrix(grid, dt, args.uplo, n, nb) for _ in range(npool)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pool[0].copy_from(ref)
        info = pool[0].factorize()
        assert info == 0, info
    for w in pool:
        w.copy_from(ref)

    prof = {k: {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0} for k in dlaf.DeviceMatrix.PROFILE_KINDS}
    # K timed steps.  Each step's window is barrier + synchronize -> factorize -> barrier + synchronize; a
    # step that has to reuse a work matrix restores it from the pristine copy BEFORE its window opens
    # (barrier - copy - barrier - start timer, miniapp_cholesky.cpp:133-143), so the timed region holds
    # factorizations only; the restore time is reported next to it.
    elapsed = 0.0
    restore_s = 0.0
    for s in range(args.steps):
        w = pool[s % npool]
        if s >= npool:
            barrier()
            t_r = time.perf_counter()
            w.copy_from(ref)
            barrier()
            restore_s += time.perf_counter() - t_r
        barrier()
        t0 = time.perf_counter()
        info = w.factorize()
        barrier()
        elapsed += time.perf_counter() - t0
        assert info == 0, info
        for k in prof:
            p = w.profile(k)
            for f in prof[k]:
                prof[k][f] += p[f]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.transport == "rccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    flops = (4.0 if args.type in "cz" else 1.0) * n ** 3 / 3.0
    sec_per_step = elapsed / args.steps
    tflops = flops / sec_per_step / 1e12

    # --check: the miniapp's check_cholesky on the device (ref is overwritten with A - L L^H): works at
    # the full size and on every grid; OK bar of the miniapp = n * eps (miniapp_cholesky.cpp:432-442)
    check = None
    if args.check or (world == 1 and not args.no_check):
        diff, norm_a = ref.residual_against(pool[(args.steps - 1) % npool])
        check = diff / norm_a

    # the panel TRSM alone on the device (in the factorization it runs beside the bulk update on a few free
    # workgroup slots, so its in-situ duration does not describe the kernel)
    trsm_alone = None
    if world == 1 and nb * 2 <= n and not args.no_trsm_profile:
        trsm_alone = pool[(args.steps - 1) % npool].trsm_profile(5)

    if rank == 0:
        # HBM-side bytes of the dominant kernel come from PMC passes (FETCH_SIZE / WRITE_SIZE cannot be
        # read from inside the process): profiles/pmc_traffic.json holds the figure measured for one
        # workload, quoted here only when this run is that workload
        traffic = traffic_src = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pt = json.load(fh)
            # quoted only for the very workload AND launch structure it was measured on: a PMC pass of a build
            # that issues a different number of bulk launches per factorization is not this run's traffic
            per_fact = prof["update_bulk"]["launches"] // max(1, args.steps)
            if pt["workload"] == f"cholesky_{args.type} N={n} nb={nb} uplo={args.uplo}" and pt["grid"] == f"{nprow}x{npcol}" \
                    and int(pt["launches_per_factorization"]) == per_fact:
                traffic, traffic_src = pt["bytes_per_launch"], pt["source"]
            else:
                traffic_src = (f"not quoted: profiles/pmc_traffic.json was measured on {pt.get('workload')} with "
                               f"{pt.get('launches_per_factorization')} bulk launches per factorization, this run has {per_fact}")
        except (OSError, KeyError, ValueError):
            pass
        bulk = prof["update_bulk"]
        trsm = prof["trsm_panel"]
        ach = bulk["flops"] / (bulk["ms"] * 1e-3) / 1e12 if bulk["ms"] > 0 else 0.0
        line = {
            "metric": "fp64 Cholesky TFlop/s" if args.type == "d" else f"{args.type} Cholesky TFlop/s",
            "value": round(tflops, 4), "unit": "TFlop/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": {"s": "f32", "d": "f64", "c": "c64", "z": "c128"}[args.type],
            "data": "synthetic",
            "config": {"workload": f"cholesky_{args.type} N={n} nb={nb} uplo={args.uplo}", "grid": f"{nprow}x{npcol}",
                       "rank_order": "column-major", "transport": args.transport if world > 1 else "none",
                       "work_copies": npool, "restore_in_timed_region": False,
                       "restore_ms_outside_timer": round(restore_s * 1e3, 3)},
            "fraction_of_fp64_mfma_peak": round(tflops / (world * PEAK_FP64_MFMA_TFLOPS), 4),
            "roofline": {"kernel": "update_kernel<T,VEC,0> (grouped trailing herk+gemm)", "bound": "mfma",
                         "achieved": round(ach, 3), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFlop/s",
                         "frac": round(ach / PEAK_FP64_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, Infinity-Cache hits included)",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bulk["bytes"] / max(1, bulk["launches"]),
                         "launches": bulk["launches"], "avg_launch_ms": round(bulk["ms"] / max(1, bulk["launches"]), 4),
                         "algorithmic_flop_per_launch": bulk["flops"] / max(1, bulk["launches"])},
            # achieved_*: the kernel alone on the device (first panel of the factorization, HIP events around 5
            # launches); in_situ_*: summed over the launches of the timed factorizations, which under the default
            # issue order share the GPU with the bulk update and get only the workgroup slots it leaves free
            # (at nb = 1024 the panel solve sits at ~51 flop/byte, five times past the ridge: it is judged against the
            # MFMA peak; the GB/s the north_star asks for stand beside it)
            "trsm_panel": {"bound": "mfma",
                           "frac_of_fp64_mfma_peak": round(trsm_alone["flops"] / max(trsm_alone["ms"], 1e-9) / 1e9 / PEAK_FP64_MFMA_TFLOPS, 4) if trsm_alone else None,
                           "achieved_GBps": round(trsm_alone["bytes"] / max(trsm_alone["ms"], 1e-9) / 1e6, 1) if trsm_alone else None,
                           "achieved_TFlops": round(trsm_alone["flops"] / max(trsm_alone["ms"], 1e-9) / 1e9, 3) if trsm_alone else None,
                           "alone_launch_ms": round(trsm_alone["ms"], 4) if trsm_alone else None,
                           "in_situ_TFlops": round(trsm["flops"] / max(trsm["ms"], 1e-9) / 1e9, 3),
                           "launches": trsm["launches"], "in_situ_avg_launch_ms": round(trsm["ms"] / max(1, trsm["launches"]), 4)},
            "update_lookahead": {"launches": prof["update_lookahead"]["launches"],
                                 "total_ms_per_step": round(prof["update_lookahead"]["ms"] / max(1, args.steps), 3),
                                 "achieved_TFlops": round(prof["update_lookahead"]["flops"] / max(prof["update_lookahead"]["ms"], 1e-9) / 1e9, 3)},
            "potrf_tile": {"launches": prof["potrf_tile"]["launches"],
                           "avg_ms": round(prof["potrf_tile"]["ms"] / max(1, prof["potrf_tile"]["launches"]), 4)},
            "setup": {"generate_s": round(t_gen, 2), "upload_s": round(t_up, 2)},
        }
        if check is not None:
            eps = float(np.finfo(np.float32 if args.type in "sc" else np.float64).eps)
            line["residual"] = {"max|A-LL^H|/max|A|": check, "bar_n_eps": n * eps, "ok": bool(check <= n * eps)}
        if world == 1 and not args.no_red2band and args.type == "d":
            try:
                for w in pool:
                    w.close()
                ref.close()
            except Exception:
                pass
            if not args.no_eigensolver:
                try:
                    line["eigensolver"] = eigensolver_line(dlaf, grid, args.r2b_n, args.r2b_nb)
                except Exception as e:
                    line["eigensolver"] = {"value": None, "error": repr(e)}
            try:
                line["reduction_to_band"] = red2band_line(dlaf, grid, args.r2b_n, args.r2b_nb, runs=3, mats=r2b_mats)
            except Exception as e:  # an extra line: it must never take the metric down with it
                line["reduction_to_band"] = {"value": None, "error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
