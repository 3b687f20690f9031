"""mcmchip_logdens_noalloc (include/mcmchip.h): the allocation-free variant of the MFMA log density, called through the
C-ABI with device pointers, against the existing entry point mcmchip_logdens(path = MFMA, device_ptrs = 1) -- same
launch plan, so the same bits -- and against the oracle (1e-12 relative).  Two calls in a row on the shared static
scratch array, the second with a smaller input; then a size beyond the array, which must be refused.
Collected last (file name) so that a failure here cannot hide the other GPU tests under `pytest -x`."""
import ctypes as C

import numpy as np
import pytest

from mcmc_amd import abi

pytestmark = pytest.mark.gpu


def _synthetic(n, d, seed):
    rng = np.random.default_rng(seed)
    X = np.empty((n, d))
    X[:, 0] = 1.0
    X[:, 1:] = rng.standard_normal((n, d - 1)) / np.sqrt(d)
    beta = 0.5 * rng.standard_normal(d)
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    return X, y, beta


def test_logdens_noalloc_matches_logdens(oracle_mod):
    import torch
    from mcmc_amd import api
    api.init(0)
    L = api.load_library()
    dev = torch.device("cuda", 0)

    def both(tg, states):
        st = torch.from_numpy(np.ascontiguousarray(states)).to(dev)
        want = torch.full((states.shape[0],), -1.0, dtype=torch.float64, device=dev)
        got = torch.full((states.shape[0],), -2.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        rc = L.mcmchip_logdens(tg.handle, st.data_ptr(), states.shape[0], abi.PATH_MFMA, 1, want.data_ptr())
        assert rc == 0, L.mcmchip_last_error().decode()
        rc = L.mcmchip_logdens_noalloc(tg.handle, st.data_ptr(), states.shape[0], got.data_ptr())
        assert rc == 0, L.mcmchip_last_error().decode()
        torch.cuda.synchronize()
        return got.cpu().numpy(), want.cpu().numpy()

    # first call: 8192 observations x 64 coefficients, 1000 chains (4 chain blocks, observation splits > 1)
    X, y, beta = _synthetic(8192, 64, 1)
    tg = api.logistic_target(X, y)
    states = beta + 0.3 * np.random.default_rng(2).standard_normal((1000, 64))
    got, want = both(tg, states)
    assert np.array_equal(got, want)
    ref = oracle_mod.logdens(oracle_mod.Target(abi.T_LOGISTIC, 64, x=X, y=y, prior_c=0.125), states[:64])
    assert np.max(np.abs(got[:64] - ref) / np.abs(ref)) < 1e-12

    # second call, smaller input (fewer chains, another descriptor with fewer, padded, columns): the scratch array
    # still holds the first call's contents
    X2, y2, beta2 = _synthetic(5003, 20, 3)
    tg2 = api.logistic_target(X2, y2)
    states2 = beta2 + 0.3 * np.random.default_rng(4).standard_normal((70, 20))
    got2, want2 = both(tg2, states2)
    assert np.array_equal(got2, want2)
    ref2 = oracle_mod.logdens(oracle_mod.Target(abi.T_LOGISTIC, 20, x=X2, y=y2, prior_c=0.125), states2)
    assert np.max(np.abs(got2 - ref2) / np.abs(ref2)) < 1e-12

    # and the first shape again after the smaller one
    got3, want3 = both(tg, states[:300])
    assert np.array_equal(got3, want3)
    assert np.max(np.abs(got3 - want[:300]) / np.abs(want[:300])) < 1e-12      # another plan (fewer chains): rounding only

    # beyond the fixed array: refused with a clear message, nothing launched
    big = torch.zeros((65536, 64), dtype=torch.float64, device=dev)
    out = torch.empty(65536, dtype=torch.float64, device=dev)
    rc = L.mcmchip_logdens_noalloc(tg.handle, big.data_ptr(), 65536, out.data_ptr())
    assert rc == abi.ECAP and b"scratch" in L.mcmchip_last_error()
    # a descriptor without MFMA tiles is refused as well
    tb = api.Target(abi.T_BALL, 5)
    rc = L.mcmchip_logdens_noalloc(tb.handle, big.data_ptr(), 16, out.data_ptr())
    assert rc == abi.EINVAL
