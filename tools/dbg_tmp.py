import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import oracle as O
from parsy_bench_amd import api, inspector as I, matrices as M
for name in ["tiny2d","small3d"]:
    A, perm = M.workload(name); sym = I.analyze(A, perm)
    plan = api.Plan(sym, 0)
    lv, sec = plan.factor(sym.A2x)
    ok, lo, _ = O.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    err = np.abs(lv-lo)
    print(name, "status", plan.status(), "maxerr", err.max(), "sec", sec, plan.info["chol_launches"])
    w = np.diff(sym.super); r = np.diff(sym.i_ptr[sym.super].astype(np.int64))
    bad = np.nonzero(err > 1e-9)[0]
    if len(bad):
        sn = np.searchsorted(sym.p[sym.super], bad, side='right')-1
        for s in np.unique(sn)[:6]:
            idx = bad[sn==s] - int(sym.p[sym.super[s]])
            print("  sn", s, "w", w[s], "r", r[s], "nbad", len(idx), "cols", np.unique(idx//r[s])[:10], "rows", np.unique(idx % r[s])[:10])
