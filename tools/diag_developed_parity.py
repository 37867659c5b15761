"""Distribution of the one-step HIP-vs-oracle differences from developed states (why tests/test_gpu_developed_state.py
allows two position outliers): python tools/diag_developed_parity.py [repetitions]"""
import sys, ctypes as C, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_developed_state import _develop
from oracle import Oracle
from wcsph_amd import _lib
for rep in range(int(sys.argv[1])):
    for mode in ("faithful", "clean"):
        n, steps = 48, 450
        dfsph, pd, reached = _develop(n, mode, steps)
        nl = pd.liquid_count
        ctx, lib = pd._need_ctx(), _lib.load()
        pos = pd.pos.to_numpy()
        state = {f: getattr(pd, f).to_numpy() for f in ("vel", "omega", "vel_guess")}
        state["kappa"], state["kappa_v"] = dfsph.kappa.to_numpy(), dfsph.kappa_v.to_numpy()
        dt0 = float(dfsph.deltaT[0]); err0 = C.c_float()
        _lib.check(ctx, lib.wcsph_get_scalar(ctx, _lib.SCALAR_ID["avg_density_err"], C.byref(err0)))
        o = Oracle(pos, nl, 0.025, mode=mode, bounds=(pd.maxboundarynp, pd.minboundarynp))
        o.reset_param()
        for f, a in state.items(): getattr(o, f)[...] = a
        o.deltaT = dt0; o.avg_density_err = err0.value
        o.vs_iter, o.dv_iter, o.pr_iter = dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter
        dt = dfsph.step(); o.dfsph_step()
        dp = np.abs(pd.pos.to_numpy()[:nl] - o.pos[:nl]).max(1)
        dv = np.abs(pd.vel.to_numpy() - o.vel).max(1)
        w = int(dp.argmax())
        drho = np.abs(pd.rho.to_numpy() - o.rho)
        dk = np.abs(dfsph.kappa.to_numpy() - o.kappa); dkv = np.abs(dfsph.kappa_v.to_numpy() - o.kappa_v)
        print(mode, "iters", (dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter), (o.vs_iter, o.dv_iter, o.pr_iter), "dt", dt,
              "pos max %.2e (particle %d, count>1e-6: %d)" % (dp.max(), w, (dp > 1e-6).sum()), "vel max %.2e vmax %.2f" % (dv.max(), np.abs(o.vel).max()),
              "worst: dv %.2e ncount %d rho %.2f drho %.2e dkappa %.2e (max %.2e of %.2e) dkv %.2e" % (dv[w], o.neighborCount[w], o.rho[w], drho[w], dk[w], dk.max(), np.abs(o.kappa).max(), dkv[w]), flush=True)
        pd.close(); o.close()
