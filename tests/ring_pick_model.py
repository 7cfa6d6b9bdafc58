"""Two models of PointProcessor's pick loops on ONE ring (PointProcessor.cc:685-732 + MaskPickedInRing :624-645), numpy only.

`greedy` is the reference's order of operations: per subregion the candidates are visited in curvature order (corners from the
largest, then flats from the smallest), a candidate is taken if it is unmasked, and taking it masks the up-to-nc neighbours on
either side that are connected to it by small gaps.

`rounds` is the form a GPU wants (k_ring_pick): inside one (subregion, phase) group the conflict relation "taking a
masks b" is symmetric (b = a + k is masked by a iff k <= nc and the k gaps between them are all small — the same gaps b's backward
walk checks), so the greedy result is the lexicographically-first maximal independent set of the conflict graph in priority order,
and that set can be grown in parallel rounds: every remaining candidate that out-ranks all remaining candidates it conflicts with
joins at once.  The per-group quota (20 corners, 4 flats) is a truncation of that set to its best-ranked members — a member's
membership depends only on better-ranked members — and only the kept members leave masks behind for the later groups.
tests/test_ring_pick_rounds.py checks that the two agree on random rings; the number of rounds is what a kernel would pay."""
import numpy as np


def subregions(n, nc=5, ns=6):
    out = []
    for j in range(ns):
        sp = (nc * (ns - j) + (n - nc) * j) // ns
        ep = (nc * (ns - 1 - j) + (n - nc) * (j + 1)) // ns - 1
        if ep > sp:
            out.append((sp, ep))
    return out


def reach(gap_ok, i, nc):
    """indices masked by taking i (besides i): forward / backward runs over small gaps; gap_ok[k] = gap between k and k + 1"""
    out = []
    for k in range(1, nc + 1):
        if not gap_ok[i + k - 1]:
            break
        out.append(i + k)
    for k in range(1, nc + 1):
        if not gap_ok[i - k]:
            break
        out.append(i - k)
    return out


def greedy(curv, gap_ok, mask0, th, nc=5, ns=6, q_corner=20, q_flat=4):
    n = len(curv)
    m = mask0.copy()
    picks = []
    for sp, ep in subregions(n, nc, ns):
        order = sorted(range(sp, ep + 1), key=lambda i: (curv[i], i))
        got = []
        for i in reversed(order):
            if len(got) >= q_corner:
                break
            if m[i] == 0 and curv[i] > th:
                got.append(i)
                m[i] = 1
                m[reach(gap_ok, i, nc)] = 1
        picks.append(("corner", sp, got))
        got = []
        for i in order:
            if len(got) >= q_flat:
                break
            if m[i] == 0 and curv[i] < th:
                got.append(i)
                m[i] = 1
                m[reach(gap_ok, i, nc)] = 1
        picks.append(("flat", sp, got))
    return picks, m


def rounds(curv, gap_ok, mask0, th, nc=5, ns=6, q_corner=20, q_flat=4):
    n = len(curv)
    m = mask0.copy()
    picks, n_rounds = [], []
    for sp, ep in subregions(n, nc, ns):
        order = sorted(range(sp, ep + 1), key=lambda i: (curv[i], i))
        for phase, quota in (("corner", q_corner), ("flat", q_flat)):
            seq = list(reversed(order)) if phase == "corner" else order
            rank = {i: r for r, i in enumerate(seq)}          # smaller = visited earlier
            alive = {i for i in seq if m[i] == 0 and (curv[i] > th if phase == "corner" else curv[i] < th)}
            members, nr = [], 0
            while alive:
                nr += 1
                # everything decided from the state at the start of the round (what the lanes of a wave would see)
                join = [i for i in alive if all(rank[i] < rank[j] for j in reach(gap_ok, i, nc) if j in alive)]
                assert join, "no candidate out-ranks its neighbourhood: the order is not total"
                members += join
                dead = set(join)
                for i in join:
                    dead.update(j for j in reach(gap_ok, i, nc) if j in alive)
                alive -= dead
            members.sort(key=lambda i: rank[i])
            kept = members[:quota]
            for i in kept:                                     # only the kept members leave masks behind
                m[i] = 1
                m[reach(gap_ok, i, nc)] = 1
            picks.append((phase, sp, kept))
            n_rounds.append(nr)
    return picks, m, n_rounds
