#!/usr/bin/env python3
"""sha256 over the sources of the headline kernel (logdens_mfma_kernel): a recorded PMC measurement of it
(profiles/r*_pmc_traffic.json) is quoted by bench.py only while this hash matches the tree.
usage: python tools/kernel_source_hash.py  -> the hash"""
import hashlib
import os

FILES = ("mcmc_amd/csrc/metrop_mfma.hip", "mcmc_amd/csrc/softplus.h", "mcmc_amd/csrc/philox.h")


def logdens_kernel_hash(root):
    h = hashlib.sha256()
    for f in FILES:
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(logdens_kernel_hash(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
