import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dla_future_amd as dlaf
dlaf.initialize()
g = dlaf.Grid.single()
for n, nb in [(2048, 512), (4096, 1024), (1024, 256)]:
    a = np.zeros((n, n), order="F"); dlaf.set_random_hermitian_positive_definite(g, a, n, nb)
    for sched in ["classic", "sidecar", "pairs", "early"]:
        os.environ["DLAF_MI355X_SCHEDULE"] = sched
        m = dlaf.DeviceMatrix(g, np.float64, "L", n, nb); m.upload(a)
        nt = n // nb
        up0 = [np.triu(m.fetch_tile(k, k), 1) for k in range(nt)]
        ok_up = all(np.array_equal(up0[k], np.triu(a[k*nb:(k+1)*nb, k*nb:(k+1)*nb], 1)) for k in range(nt))
        assert m.factorize() == 0
        bad = []
        for k in range(nt):
            d = np.triu(m.fetch_tile(k, k), 1)
            if not np.array_equal(d, up0[k]):
                diff = np.argwhere(d != up0[k])
                bad.append((k, len(diff), diff.min(0).tolist(), diff.max(0).tolist(), float(np.abs(d[d != up0[k]]).max())))
        print(n, nb, sched, "upload_ok", ok_up, "changed:", bad, flush=True)
        m.close()
