"""CPU model of the wave reductions of csrc/device/lane_ops.hpp / kernels_tridiag.hip (band_to_tridiagonal's register kernel):
the lane maps of v_permlane32_swap, v_permlane16_swap and the DPP row controls as measured on gfx950
(profiles/r04_permlane_probe.txt, tools/permlane_probe.hip), the reductions restated on them in numpy.  Pins on the CPU what the
kernels rely on: which lane ends up with which total (wave_reduce8_index / wave_reduce16_index) and that wave_sum_fast leaves
the total in every lane."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LANES = np.arange(64)


def swap32(a, b):   # a <- [a.lo | b.lo], b <- [a.hi | b.hi]
    return np.concatenate([a[:32], b[:32]]), np.concatenate([a[32:], b[32:]])


def swap16(a, b):   # a <- rows [a0 b0 a2 b2], b <- rows [a1 b1 a3 b3]
    r = lambda x, i: x[16 * i:16 * i + 16]
    return np.concatenate([r(a, 0), r(b, 0), r(a, 2), r(b, 2)]), np.concatenate([r(a, 1), r(b, 1), r(a, 3), r(b, 3)])


def dpp(x, ctrl):
    row, l = LANES // 16 * 16, LANES % 16
    src = {"ror8": row + (l + 8) % 16, "half_mirror": row + (l // 8) * 8 + 7 - l % 8, "xor1": LANES ^ 1, "xor2": LANES ^ 2}[ctrl]
    return x[src]


def test_lane_maps_match_the_probe():
    """the model's maps against the lane maps the probe printed on the GPU"""
    rows = {}
    with open(os.path.join(ROOT, "profiles", "r04_permlane_probe.txt")) as fh:
        for ln in fh:
            if ":" in ln:
                name, vals = ln.rsplit(":", 1)
                rows[name.strip()] = np.array([int(v) for v in vals.split()])
    a, b = LANES.copy(), 100 + LANES
    x, y = swap32(a, b)
    assert (rows["permlane32_swap [0] (vdst)"] == x).all() and (rows["permlane32_swap [1] (src)"] == y).all()
    x, y = swap16(a, b)
    assert (rows["permlane16_swap [0]"] == x).all() and (rows["permlane16_swap [1]"] == y).all()
    assert (rows["row_half_mirror"] == dpp(a, "half_mirror")).all() and (rows["row_ror:8"] == dpp(a, "ror8")).all()
    assert (rows["quad_perm[1,0,3,2]"] == dpp(a, "xor1")).all() and (rows["quad_perm[2,3,0,1]"] == dpp(a, "xor2")).all()


def swap_add(a, b, rows):
    x, y = (swap16 if rows else swap32)(a, b)
    return x + y


def reduce4_tail(b4):
    h3, h2 = (LANES & 8) != 0, (LANES & 4) != 0
    c2 = []
    for i in range(2):
        mine = np.where(h3, b4[2 + i], b4[i])
        theirs = np.where(h3, b4[i], b4[2 + i])
        c2.append(mine + dpp(theirs, "ror8"))
    r = np.where(h2, c2[1], c2[0]) + dpp(np.where(h2, c2[0], c2[1]), "half_mirror")
    r = r + dpp(r, "xor2")
    return r + dpp(r, "xor1")


def wave_reduce16(v):
    a = [swap_add(v[i], v[8 + i], False) for i in range(8)]
    b4 = [swap_add(a[i], a[4 + i], True) for i in range(4)]
    return reduce4_tail(b4)


def wave_reduce8(v):
    h3 = (LANES & 8) != 0
    b4 = [swap_add(v[i], v[4 + i], False) for i in range(4)]
    c2 = [swap_add(b4[i], b4[2 + i], True) for i in range(2)]
    r = np.where(h3, c2[1], c2[0]) + dpp(np.where(h3, c2[0], c2[1]), "ror8")
    for c in ("half_mirror", "xor2", "xor1"):
        r = r + dpp(r, c)
    return r


def wave_sum_fast(v):
    for c in ("xor1", "xor2", "half_mirror", "ror8"):
        v = v + dpp(v, c)
    v = swap_add(v, v, True)
    return swap_add(v, v, False)


def test_transposing_reductions_leave_the_totals_where_the_kernel_reads_them():
    rng = np.random.default_rng(0)
    v = rng.integers(-1000, 1000, (16, 64)).astype(np.int64)     # (integers: exact sums)
    r = wave_reduce16(list(v))
    idx16 = ((LANES >> 5) & 1) * 8 + ((LANES >> 4) & 1) * 4 + ((LANES >> 3) & 1) * 2 + ((LANES >> 2) & 1)   # wave_reduce16_index
    assert (r == v.sum(axis=1)[idx16]).all()
    r = wave_reduce8(list(v[:8]))
    idx8 = ((LANES >> 5) & 1) * 4 + ((LANES >> 4) & 1) * 2 + ((LANES >> 3) & 1)                            # wave_reduce8_index
    assert (r == v[:8].sum(axis=1)[idx8]).all()
    # the kernel writes from the lanes with (l & 3) == 0 / (l & 7) == 0: every value index has such a lane
    assert set(idx16[(LANES & 3) == 0]) == set(range(16)) and set(idx8[(LANES & 7) == 0]) == set(range(8))
    assert (wave_sum_fast(v[0].copy()) == v[0].sum()).all()
