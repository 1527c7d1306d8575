"""fused (P2P all-reduce + Adam in one kernel) vs unfused data-parallel step: where do the parameters differ?"""
import os, sys
import numpy as np, torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_dp as T
variant = sys.argv[1] if len(sys.argv) > 1 else "mmd"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if __name__ == "__main__":
    res = {}
    for fused in (True, False):
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(T._worker, args=(world, T._free_port(), fused, ret, variant, "fp32"), nprocs=world, join=True)
        assert "error" not in ret[0], ret[0].get("error")
        res[fused] = ret[0]["params"]
    from factorized_amd import engine, configs
    lay = engine.FlatLayout(engine.param_shapes(configs.canonical_configs(dropout=False), variant), variant)
    print("total", lay.total, "numel", lay.numel)
    for n in res[True]:
        d = np.abs(res[True][n] - res[False][n])
        if d.max() > 1e-7:
            idx = np.argwhere(d > 1e-7)
            print("%-36s offset %8d size %7d  max diff %.3e  n_diff %d first %s last %s" % (n, lay.offsets[n], d.size, d.max(), len(idx), idx[0], idx[-1]))
