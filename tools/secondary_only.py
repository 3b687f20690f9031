"""blocks of bench_secondary.py alone; usage: python tools/secondary_only.py [c2|c4|c5|temper_models ...] (default: c5, initseq / olbm on the 204.8 GB
resident series)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mcmc_amd import api  # noqa: E402
import bench_secondary  # noqa: E402

api.init()
print(json.dumps(bench_secondary.run(api, torch, torch.device("cuda:0"), tuple(sys.argv[1:]) or ("c5",)), indent=1))
