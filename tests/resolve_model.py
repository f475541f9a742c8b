"""Executable model of k_match_resolve's speculative rounds (orb_match_kernels.h), checked against the plain in-order loop.

The reference resolves map points strictly in order (ORBmatcher.cc:75-135): a keypoint taken by an earlier map point with
observations is skipped by later ones.  The kernel instead lets the 64 queries of a chunk decide in parallel and iterates
to the unique fix-point of  D_i = f_i(D_0..D_i-1);  a query whose TOPK list is exhausted cuts the prefix and is rescanned
exactly.  This file restates both in pure Python on abstract candidate lists so that the equivalence (including the
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


def decide(lst_full, owner_view, th, ratio, use_second):
    """Decision from the TOPK prefix of a list; owner_view(k) -> True if claimed for this lane.
    Returns (accept, rescan, kp)."""
    top = lst_full[:TOPK]
    truncated = len(lst_full) >= TOPK          # the kernel sees only "entry TOPK-1 exists"
    alive = [c for c in top if not owner_view(c[2])]
    found = len(alive)
    lb = top[TOPK - 1][0] if truncated else None
    bd = alive[0][0] if found else 256
    rescan = False
    if truncated:
        if found == 0:
            rescan = lb <= th
        elif found == 1 and use_second:
            rescan = bd <= th and bd > ratio * lb
    acc = (not rescan) and accept_rule(th, ratio, use_second, alive[0] if found else None, alive[1] if found > 1 else None)
    return acc, rescan, (alive[0][2] if found else 0)


def speculative(cands, obs, n, th, ratio, use_second, preclaimed, stats=None):
    owner = [FREE] * (n + 1)
    for k in preclaimed:
        owner[k] = 0
    slotv = [-1] * n
    nq = len(cands)
    moq = [-1] * nq
    for base in range(0, nq, 64):
        cnt = min(64, nq - base)
        D = [(False, False, 0)] * 64
        s = 0
        while s < cnt:
            while True:
                pend = [s <= l < cnt for l in range(64)]
                post = [pend[l] and D[l][0] and obs[base + l] for l in range(64)]
                for l in range(64):                       # ds_min
                    if post[l]:
                        owner[D[l][2]] = min(owner[D[l][2]], l + 1)
                snap = list(owner)                        # every lane reads before anyone withdraws
                for l in range(64):
                    if post[l] and snap[D[l][2]] != 0:
                        owner[D[l][2]] = FREE
                newD = list(D)
                for l in range(64):
                    if pend[l]:
                        newD[l] = decide(cands[base + l], lambda k, l=l: snap[k] <= l, th, ratio, use_second)
                changed = [newD[l] != D[l] for l in range(64)]
                D = newD
                flagged = [l for l in range(64) if pend[l] and D[l][1]]
                r = flagged[0] if flagged else cnt
                if stats is not None:
                    stats["rounds"] = stats.get("rounds", 0) + 1
                if not any(changed[l] and l <= r for l in range(64)):
                    break
            for l in range(s, r):                         # commit the settled prefix
                if D[l][0]:
                    k = D[l][2]
                    moq[base + l] = k
                    if obs[base + l]:
                        owner[k] = 0
                    slotv[k] = max(slotv[k], ((base + l) << 1) | int(obs[base + l]))
            if r < cnt:                                   # exact rescan with the committed claims
                alive = [c for c in cands[base + r] if owner[c[2]] != 0]
                best = alive[0] if alive else None
                second = alive[1] if len(alive) > 1 else None
                if accept_rule(th, ratio, use_second, best, second):
                    k = best[2]
                    moq[base + r] = k
                    if obs[base + r]:
                        owner[k] = 0
                    slotv[k] = max(slotv[k], ((base + r) << 1) | int(obs[base + r]))
                D[r] = (False, False, 0)
                if stats is not None:
                    stats["rescans"] = stats.get("rescans", 0) + 1
            s = r + 1
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
