"""The hand-off protocol of band_to_tridiagonal's register kernel (csrc/device/kernels_tridiag.hip, b2t_reg_kernel) checked
on the CPU by enumeration.  Sweep s runs its steps in order; step t works on the block rows [j, j + 2b) x columns [j, j + b),
j = 1 + s + t b (lower part).  A step (1) loads all columns but the last once the predecessor sweep has FINISHED its step t,
(2) loads the last column once the predecessor has published the FIRST COLUMN of its step t + 1, (3) stores its own first
column and publishes it, (4) stores the rest and publishes the step.  The sweep's prologue reads and rewrites column s once
the predecessor has published the first column of its step 0.  (progress word: 2 per finished step + 1 for the first column.)
A sweep that has run out of steps answers every wait ("done") -- and publishes that only once its own predecessor is done:
without that last rule the model finds a pair, in the tail of the matrix, that nothing orders (sweep s - 1's last step
still storing its second column while sweep s, one step shorter, is done and lets sweep s + 1 read that column).

The model builds the happens-before graph these waits and the program order give and checks, for every element of the band,
that any two accesses of different sweeps with a write among them are ordered the way the sequential algorithm orders them
(earlier sweep first).  A wait that is too weak shows up as an unordered pair."""
import itertools

import pytest


def nsteps(n, b, s):
    return (n - s - 2 + b - 1) // b


def build(n, b, weaken=None):
    nsweeps = n - 2
    ev = {}           # name -> index
    prog = []         # edges (u, v): u happens before v
    acc = []          # (event, 'r' | 'w', set of (row, col))

    def E(name):
        if name not in ev:
            ev[name] = len(ev)
        return ev[name]

    def elems(rows, cols):
        return {(r, c) for c in cols for r in rows if r >= c and r < n and c < n}

    for s in range(nsweeps):
        last = E(("P", s))
        col = range(s, s + 1)
        acc.append((last, "r", elems(range(s + 1, s + 1 + b), col)))
        acc.append((last, "w", elems(range(s + 1, s + 1 + b), col)))
        if s > 0 and nsteps(n, b, s - 1) > 0 and weaken == "generic_kernel":
            prog.append((E(("ST", s - 1, 0)), last))   # (the predecessor has finished one step)
        elif s > 0 and nsteps(n, b, s - 1) > 0 and weaken != "prologue":
            prog.append((E(("FC", s - 1, 0)), last))
        for t in range(nsteps(n, b, s)):
            j = 1 + s + t * b
            rows = range(j, j + 2 * b)
            le, ll, fc, st = E(("LE", s, t)), E(("LL", s, t)), E(("FC", s, t)), E(("ST", s, t))
            for u, v in ((last, le), (le, ll), (ll, fc), (fc, st)):
                prog.append((u, v))
            last = st
            acc.append((le, "r", elems(rows, range(j, j + b - 1))))
            acc.append((ll, "r", elems(rows, range(j + b - 1, j + b))))
            acc.append((fc, "w", elems(rows, range(j, j + 1))))
            acc.append((st, "w", elems(rows, range(j + 1, j + b))))
            if s > 0 and weaken == "generic_kernel":
                # b2t_kernel (complex double, bands above 128) and the reference's semaphores (mc.h:683-709): a step runs once
                # the predecessor has finished t + 2 steps, or all it has; no early column, no chain of "done"
                ps = nsteps(n, b, s - 1)
                prog.append((E(("ST", s - 1, min(t + 1, ps - 1))), le))
            elif s > 0:
                ps = nsteps(n, b, s - 1)
                # (1) the predecessor has finished its step t (or is done)
                prog.append((E(("ST", s - 1, t)) if t < ps else E(("DONE", s - 1)), le))
                # (2) the predecessor has published the first column of its step t + 1 (or is done)
                if weaken != "last_column":
                    prog.append((E(("FC", s - 1, t + 1)) if t + 1 < ps else E(("DONE", s - 1)), ll))
        done = E(("DONE", s))
        prog.append((last, done))
        if s > 0 and weaken not in ("done_chain", "generic_kernel"):
            prog.append((E(("DONE", s - 1)), done))   # a sweep is done only once its predecessor is
    return ev, prog, acc


def unordered_conflicts(n, b, weaken=None):
    ev, prog, acc = build(n, b, weaken)
    m = len(ev)
    succ = [[] for _ in range(m)]
    for u, v in prog:
        succ[u].append(v)
    # reachability by bitsets, events in topological order (a sweep's events are numbered in program order and every
    # cross edge goes from sweep s - 1 to sweep s: index order is a topological order)
    reach = [0] * m
    for u in range(m - 1, -1, -1):
        r = 1 << u
        for v in succ[u]:
            assert v > u
            r |= reach[v]
        reach[u] = r
    sweep_of = {i: name[1] for name, i in ev.items()}
    per_elem = {}
    for e, kind, els in acc:
        for el in els:
            per_elem.setdefault(el, []).append((e, kind))
    bad = []
    for el, lst in per_elem.items():
        for (e1, k1), (e2, k2) in itertools.combinations(lst, 2):
            if k1 == "r" and k2 == "r":
                continue
            s1, s2 = sweep_of[e1], sweep_of[e2]
            if s1 == s2:
                continue
            first, second = (e1, e2) if s1 < s2 else (e2, e1)
            if not (reach[first] >> second) & 1:
                bad.append((el, first, second))
    return bad, {i: name for name, i in ev.items()}


@pytest.mark.parametrize("n,b", [(14, 2), (23, 3), (30, 4), (41, 5), (37, 8)])
def test_every_conflicting_pair_of_accesses_is_ordered(n, b):
    bad, names = unordered_conflicts(n, b)
    assert not bad, [(el, names[u], names[v]) for el, u, v in bad[:5]]


def test_the_model_notices_a_wait_that_is_too_weak():
    """without the wait for the predecessor's first column of the next step (or the one of the prologue) pairs are unordered"""
    assert unordered_conflicts(23, 3, weaken="last_column")[0]
    assert unordered_conflicts(23, 3, weaken="prologue")[0]
    assert unordered_conflicts(23, 3, weaken="done_chain")[0]   # (what the kernel did before the model was written)


@pytest.mark.parametrize("n,b", [(23, 3), (41, 5), (37, 8)])
def test_the_generic_kernels_protocol_orders_every_pair_too(n, b):
    """step t after the predecessor has finished t + 2 steps (or all of them): the protocol of b2t_kernel and of the reference"""
    bad, names = unordered_conflicts(n, b, weaken="generic_kernel")
    assert not bad, [(el, names[u], names[v]) for el, u, v in bad[:5]]
