import importlib, sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np
pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
from oracle import oracle_py as oracle
import fuzz_parity as FP, test_gpu_mfma as T
seed, case = int(sys.argv[1]), int(sys.argv[2])
cache = {}
log = print
# monkeypatch run_batch to capture inputs
orig = T.run_batch
cap = {}
def rb(pkg_, m, cand, qry, bounds, nnratio, th, second):
    cap.update(cand=cand, qry=qry, bounds=bounds, nnratio=nnratio, th=th, second=second)
    return orig(pkg_, m, cand, qry, bounds, nnratio, th, second)
T.run_batch = rb
ok = FP.fuzz_batch_open(pkg, oracle, synth, np.random.default_rng([seed, case, 6]), log, cache)
print("ok", ok)
cand, qry, bounds = cap["cand"], cap["qry"], cap["bounds"]
sf = cache["sf"]
for eng in (2, 1, 0):
    m = pkg.ORBmatcher(cap["nnratio"], True); m.set_hamming_engine(eng)
    got = orig(pkg, m, cand, qry, bounds, cap["nnratio"], cap["th"], cap["second"]); m.close()
    for p, (c, q) in enumerate(zip(cand, qry)):
        r = T.oracle_pair(oracle, c, q, bounds, sf, cap["nnratio"], cap["th"], cap["second"])
        g = got[p]
        if not (g[0] == r[0] and np.array_equal(g[1], r[1])):
            diff = np.nonzero(g[1] != r[1])[0]
            print("engine", eng, "pair", p, "n", len(c["k"]), "nq", len(q["u"]), "nm", g[0], r[0], "first differing queries", diff[:10], "gpu", g[1][diff[:10]], "ref", r[1][diff[:10]])
            qi = diff[0]
            D = np.unpackbits(q["d"][qi][None, :] ^ c["d"], axis=1).sum(axis=1)
            print("   query", qi, "flags", q["flags"][qi], "r", q["r"][qi], "gpu pick", g[1][qi], "dist", D[g[1][qi]] if g[1][qi] >= 0 else None, "ref pick", r[1][qi], "dist", D[r[1][qi]] if r[1][qi] >= 0 else None, "bd gpu/ref", g[2][qi], r[2][qi])
            for which, pick in (("gpu", g[1][qi]), ("ref", r[1][qi])):
                if pick >= 0:
                    print("   ", which, "pick", pick, "pre-slot", c["slot"][pick], c["sobs"][pick], "x,y", c["k"]["x"][pick], c["k"]["y"][pick], "final slot gpu/ref", g[3][pick], r[3][pick])
