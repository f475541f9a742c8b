"""Executable model of k_match_resolve's speculative rounds (orb_match_kernels.h), checked against the plain in-order loop.

The reference resolves map points strictly in order (ORBmatcher.cc:75-135): a keypoint taken by an earlier map point with
observations is skipped by later ones.  The kernel instead lets the 64 queries of a chunk decide in parallel and iterates
to the unique fix-point of  D_i = f_i(D_0..D_i-1);  a query whose TOPK list is exhausted cuts the prefix: the prefix is committed and the
query gets a fresh list (the best REFRESH_K keypoints no committed claim holds) and decides again.  This file restates both in pure Python on abstract candidate lists so that the equivalence (including the
"withdraw unless committed" rule and the last-writer slot) is tested without a GPU.
"""
import numpy as np

FREE = 0xFFFFFFFF
TOPK = 8


def accept_rule(th, ratio, use_second, best, second):
    # best/second: (dist, level) or None  -- ORBmatcher.cc:124-130
    if best is None or best[0] > th:
        return False
    if use_second and second is not None and best[1] == second[1] and best[0] > ratio * second[0]:
        return False
    return True


def sequential(cands, obs, n, th, ratio, use_second, preclaimed):
    """cands[q] = list of (dist, level, kp) sorted by key; returns (match_of_query, slot, slot_obs)."""
    claimed = set(preclaimed)
    slot = [-1] * n
    sobs = [0] * n
    moq = []
    for q, lst in enumerate(cands):
        alive = [c for c in lst if c[2] not in claimed]
        best = alive[0] if alive else None
        second = alive[1] if len(alive) > 1 else None
        if accept_rule(th, ratio, use_second, best, second):
            k = best[2]
            moq.append(k)
            slot[k] = q
            sobs[k] = obs[q]
            if obs[q]:
                claimed.add(k)
        else:
            moq.append(-1)
    return moq, slot, sobs


REFRESH_K = 4


def decide(top, cap, owner_view, th, ratio, use_second):
    """Decision from a lane's current list `top` (<= cap entries, sorted); owner_view(k) -> True if claimed for this lane.
    Every candidate that is not in the list (and was unclaimed when the list was made) has key >= top[cap-1].
    Returns (accept, exhausted, kp)."""
    truncated = len(top) >= cap                # the kernel sees only "entry cap-1 exists"
    alive = [c for c in top if not owner_view(c[2])]
    found = len(alive)
    lb = top[cap - 1][0] if truncated else None
    bd = alive[0][0] if found else 256
    exhausted = False
    if truncated:
        if found == 0:
            exhausted = lb <= th
        elif found == 1 and use_second:
            exhausted = bd <= th and bd > ratio * lb
    acc = (not exhausted) and accept_rule(th, ratio, use_second, alive[0] if found else None, alive[1] if found > 1 else None)
    return acc, exhausted, (alive[0][2] if found else 0)


def speculative(cands, obs, n, th, ratio, use_second, preclaimed, stats=None):
    owner = [FREE] * (n + 1)
    for k in preclaimed:
        owner[k] = 0
    slotv = [-1] * n
    nq = len(cands)
    moq = [-1] * nq

    def bump(key, by=1):
        if stats is not None:
            stats[key] = stats.get(key, 0) + by

    for base in range(0, nq, 64):
        cnt = min(64, nq - base)
        lists = [cands[base + l][:TOPK] if l < cnt else [] for l in range(64)]   # what k_match_scan delivered
        caps = [TOPK] * 64
        D = [(False, False, 0)] * 64
        s = 0
        first_round = True

        def refresh(lanes):
            # new list = the REFRESH_K best candidates that no COMMITTED claim holds (posts are all withdrawn here)
            assert all(o in (0, FREE) for o in owner), "a post was left behind"
            for l in lanes:
                lists[l] = [c for c in cands[base + l] if owner[c[2]] != 0][:REFRESH_K]
                caps[l] = REFRESH_K
            bump("refresh_batches", (len(lanes) + 7) // 8)
            bump("refreshed", len(lanes))

        while s < cnt:
            while True:
                pend = [s <= l < cnt for l in range(64)]
                post = [pend[l] and D[l][0] and obs[base + l] for l in range(64)]
                for l in range(64):                       # ds_min
                    if post[l]:
                        owner[D[l][2]] = min(owner[D[l][2]], l + 1)
                snap = list(owner)                        # every lane reads before anyone withdraws
                for l in range(64):
                    if post[l] and snap[D[l][2]] != 0:    # a claim committed since my (stale) decision must survive
                        owner[D[l][2]] = FREE
                newD = list(D)
                for l in range(64):
                    if pend[l]:
                        newD[l] = decide(lists[l], caps[l], lambda k, l=l: snap[k] <= l, th, ratio, use_second)
                changed = [newD[l] != D[l] for l in range(64)]
                D = newD
                bump("rounds")
                flagged = [l for l in range(64) if pend[l] and D[l][1]]
                if first_round:
                    first_round = False
                    if flagged:                           # exhausted by the claims of earlier chunks: refresh at once
                        refresh(flagged)
                        continue
                r = flagged[0] if flagged else cnt
                if not any(changed[l] and l <= r for l in range(64)):
                    break
            for l in range(s, r):                         # commit the settled prefix
                if D[l][0]:
                    k = D[l][2]
                    moq[base + l] = k
                    if obs[base + l]:
                        owner[k] = 0
                    slotv[k] = max(slotv[k], ((base + l) << 1) | int(obs[base + l]))
            s = r
            if r < cnt:                                   # lane r (and every other exhausted lane) gets a fresh list
                refresh(flagged)
        assert all(o in (0, FREE) for o in owner), "a post was left behind"
    slot = [v >> 1 if v >= 0 else -1 for v in slotv]
    sobs = [v & 1 if v >= 0 else 0 for v in slotv]
    return moq, slot, sobs


def random_problem(rng, n, nq, density, p_obs, junk):
    """Lists with heavy overlap: every query prefers 'its' keypoint (if it has one) and a random set of others."""
    cands = []
    for q in range(nq):
        m = rng.integers(0, max(2, int(density * n)))
        ks = rng.choice(n, size=min(n, m), replace=False)
        lst = []
        for k in ks:
            lst.append((int(rng.integers(60, 140)), int(rng.integers(0, 3)), int(k)))
        if rng.random() > junk:
            k = int(rng.integers(0, n))
            lst = [c for c in lst if c[2] != k] + [(int(rng.integers(5, 60)), int(rng.integers(0, 3)), k)]
        lst.sort(key=lambda c: (c[0], c[2]))
        cands.append(lst)
    obs = (rng.random(nq) < p_obs).astype(int).tolist()
    pre = set(rng.choice(n, size=n // 20, replace=False).tolist())
    return cands, obs, pre
