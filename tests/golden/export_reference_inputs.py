"""Exports the inputs of the small committed fixtures as HDF5 files the REAL CMF.jl can read (tools/reference_fixtures.jl).

The reference is Julia and cannot run in the build image, so parity is pinned by the oracle only.  This
script and tools/reference_fixtures.jl make the missing pin a one-command job for anyone who has Julia: the .npz fixtures'
inputs (data, W0, H0, max_itr, the regularisers, the rule) go into tests/golden/ref_inputs/<name>.h5 in HDF5.jl's conventions
(cmf.jl_amd/_hdf5.py, the writer behind save_model: src/model.jl:149-163 reads the same layout), the Julia script runs the
reference's own fit_cnmf on them and writes tests/golden/ref_<name>.h5, and tests/test_reference_fixtures.py compares the
oracle and the HIP path with those files when they exist.

    python tests/golden/export_reference_inputs.py          # (re)writes tests/golden/ref_inputs/*.h5
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "ref_inputs")
# fixture -> the reference rule that produced its expected outputs in the oracle (make_golden.py)
FIXTURES = {"mu_small": "mult", "mu_small_reg": "mult", "mu_k5": "mult", "hals_small": "hals", "pgd_small": "pgd"}


def _hdf5():
    spec = importlib.util.spec_from_file_location("cmf_hdf5", os.path.join(ROOT, "cmf.jl_amd", "_hdf5.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def inputs_of(name):
    with np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False) as f:
        items = {k: np.array(f[k], dtype=np.float64) for k in ("data", "W0", "H0")}
        for k in ("max_itr", "l1W", "l2W", "l1H", "l2H"):
            items[k] = float(f[k])
    items["rule"] = FIXTURES[name]
    return items


def main():
    h5 = _hdf5()
    os.makedirs(OUT, exist_ok=True)
    for name in FIXTURES:
        path = os.path.join(OUT, name + ".h5")
        h5.write_file(path, inputs_of(name))
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    sys.exit(main())
