import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, oracle
from mcmc_amd import abi, api
from rhelpers import glm_binomial, load_logit
api.init(0)
X, y = load_logit(); mle = glm_binomial(X, y)
rng = np.random.default_rng(0)
C, nbatch, blen, nspac = 4, 3, 2, 2
niter = nbatch*blen*nspac
init = mle + 0.05*rng.standard_normal((C,5))
z, u = rng.standard_normal((C,niter,5)), rng.random((C,niter))
tg = api.logistic_target(X,y)
print("logdens gpu", api.logdens(tg, init, path=abi.PATH_STREAM))
ot = oracle.Target(abi.T_LOGISTIC,5,x=X,y=y,prior_c=0.125)
print("logdens ora", oracle.logdens(ot, init))
g = api.metrop(tg, init, nbatch, blen, nspac, scale=0.4, z=z, u=u, debug=True)
r = oracle.metrop(ot, init, nbatch, blen, nspac, scale=0.4, z=z, u=u, debug=True)
for k,kr in [("current","current"),("proposal","proposal"),("z","z"),("log.green","log_green"),("u","u"),("debug.accept","debug_accept")]:
    a, b = np.asarray(g[k], dtype=float), np.asarray(r[kr], dtype=float)
    print(k, "maxdiff", np.nanmax(np.abs(a-b)), "nan-pattern-equal", np.array_equal(np.isnan(a), np.isnan(b)))
print(g["log.green"][0], r["log_green"][0])
print(g["final"][0], r["final"][0])
print(g["accept"], r["accept"])
