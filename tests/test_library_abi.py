"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/cmf_hip.h declares, and fails loudly (no fallback) without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cmf():
    import __graft_entry__

    __graft_entry__.build()
    import cmf_jl_amd as m

    return m


def test_header_symbols_all_exported(cmf):
    hdr = open(os.path.join(ROOT, "include", "cmf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(cmf_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(cmf.SYMBOLS), "binding table and header disagree"
    lib = ctypes.CDLL(cmf.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cmf_hip.h but not exported"


def test_version_and_host_only_entries(cmf):
    lib = cmf.load_library()
    assert lib.cmf_version().startswith(b"cmf_hip gfx950")
    # converged is pure host arithmetic (model.jl:91-107)
    assert not cmf.converged([1.0, 1.0, 1.0], 3, 1e-4)
    assert cmf.converged([1.0, 1.0, 1.0, 1.0], 3, 1e-4)
    assert not cmf.converged([2.0, 1.0, 1.0, 1.0], 3, 1e-4)
    assert cmf.converged([1.0, 1.00005], 1, 1e-4)


def test_no_cpu_fallback(cmf):
    """Without a HIP device every compute entry must raise, never silently compute on the CPU."""
    lib = cmf.load_library()
    if lib.cmf_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(cmf.CMFError) as ei:
        cmf.tensor_conv(np.ones((2, 3, 2)), np.ones((2, 5)))
    assert ei.value.code == 2
    with pytest.raises(cmf.CMFError):
        cmf.fit_cnmf(np.ones((4, 16)), L=2, K=2, max_itr=1)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under cmf.jl_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cmf.jl_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".jl")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"


def test_results_on_disk_and_convergence_helper(cmf, tmp_path):
    """save_model / load_model (model.jl:149-181 schema names) and evaluate_convergence (evaluate.jl:29-44)
    are host-only."""
    r = cmf.CNMF_results(np.ones((3, 5)), np.ones((2, 3, 4)), np.ones((2, 5)), np.arange(4.0), np.array([1.0, 0.5, 0.402, 0.4]))
    path = str(tmp_path / "m.npz")
    cmf.save_model(r, path, l1_H=0.1, alg="mult")
    r2, meta = cmf.load_model(path)
    for k in ("W", "H", "data", "loss_hist", "time_hist"):
        np.testing.assert_array_equal(getattr(r, k), getattr(r2, k))
    assert float(meta["l1_H"]) == 0.1 and str(meta["alg"]) == "mult"
    assert cmf.evaluate_convergence(r) == 2        # 0.402/0.4 < 1.01
    assert cmf.evaluate_convergence(r, thresh=0.3) == 1
    assert (r.num_lags(), r.num_units(), r.num_components(), r.num_iter()) == (4, 3, 2, 4)


def test_results_as_hdf5_in_the_reference_schema(cmf, tmp_path):
    """save_model / load_model with a .h5 path: a real HDF5 file in HDF5.jl's conventions (src/model.jl:149-181):
    Float64 datasets whose dimensions are the Julia dimensions reversed and whose bytes are Julia's column-major
    buffer; Float64 scalars; `alg` as a string."""
    from cmf_jl_amd import _hdf5  # noqa: the package shim

    if not _hdf5.available():
        pytest.skip("no libhdf5 in this environment")
    rng = np.random.default_rng(0)
    K, N, L, T = 2, 3, 4, 5
    r = cmf.CNMF_results(rng.random((N, T)), rng.random((K, N, L)), rng.random((K, T)), np.arange(4.0), np.array([1.0, 0.5, 0.402, 0.4]))
    path = str(tmp_path / "model.h5")
    cmf.save_model(r, path, l1_H=0.1, l2_H=0.2, l1_W=0.0, l2_W=0.5, alg=":mult")
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n"
    # the W tensor is stored contiguously in Julia order: W[k, n, l] with k fastest
    assert np.asfortranarray(r.W).tobytes(order="A") in raw or np.asfortranarray(r.W).T.tobytes() in raw
    r2, meta = cmf.load_model(path)
    for k in ("W", "H", "data", "loss_hist", "time_hist"):
        np.testing.assert_array_equal(getattr(r, k), getattr(r2, k))
    assert meta == {"l1_H": 0.1, "l2_H": 0.2, "l1_W": 0.0, "l2_W": 0.5, "alg": "mult"}
    # HDF5 sees the reversed dimensions (what h5dump / h5py would print for a file written by HDF5.jl)
    lib = _hdf5._load()
    f = lib.H5Fopen(path.encode(), 0, 0)
    d = lib.H5Dopen2(f, b"W", 0)
    s = lib.H5Dget_space(d)
    dims = (_hdf5.hsize_t * 3)()
    assert lib.H5Sget_simple_extent_ndims(s) == 3
    lib.H5Sget_simple_extent_dims(s, dims, None)
    assert tuple(dims) == (L, N, K)
    lib.H5Sclose(s); lib.H5Dclose(d); lib.H5Fclose(f)
    with pytest.raises(KeyError):
        _hdf5.write_file(str(tmp_path / "partial.h5"), {"W": r.W})
        cmf.load_model(str(tmp_path / "partial.h5"))


def option_names(lib):
    buf = ctypes.create_string_buffer(1024)
    assert lib.cmf_option_names(buf, 1024) == 0
    return buf.value.decode().split(",")


def test_every_code_path_choice_is_a_documented_option_and_the_environment_is_ten_variables(cmf):
    """Round 6: the measurement knobs read from the environment are gone.  Whatever selects a code path is a cmf_set_option name
    that the header documents (cmf_option_names is the table), and the sources read exactly the variables the header lists."""
    lib = cmf.load_library()
    names = option_names(lib)
    assert len(names) == len(set(names)) >= 15
    hdr = open(os.path.join(ROOT, "include", "cmf_hip.h")).read()
    for n in names:
        assert f'"{n}"' in hdr, f"option {n} is not documented in include/cmf_hip.h"
    assert lib.cmf_option_names(ctypes.create_string_buffer(8), 8) != 0  # truncation is an error, not a silent cut
    csrc = os.path.join(ROOT, "cmf.jl_amd", "csrc")
    text = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)))
    code = re.sub(r"//[^\n]*", "", text)
    assert len(re.findall(r"\bgetenv\s*\(", code)) <= 10
    read = set(re.findall(r'getenv\("(CMF_[A-Z_]+)"\)', code)) | set(re.findall(r'test_hook\("(CMF_[A-Z_]+)"', code))
    listed = set(re.findall(r"^ \*\s+(CMF_[A-Z_]+)\s", hdr[: hdr.index("#ifndef CMF_HIP_H")], flags=re.M))
    assert read == listed, (sorted(read - listed), sorted(listed - read))
    assert len(read) == 10
    # ... and nothing else in the repository still sets a variable the library no longer reads
    stale = re.compile(r"CMF_(HALS_[A-Z]+|CONV_[A-Z_]+|SK_[A-Z_0-9]+|GRAM_FW|PGD_TRANSPOSE|LOSS_POLL|SPECULATE_W|SMALL_K\b|HXT_EXACT|LOOPBACK_[A-Z_]+|EXP_CU_MASK)")
    for sub in ("tests", "tools", "cmf.jl_amd", "."):
        d = os.path.join(ROOT, sub)
        for f in sorted(os.listdir(d)):
            if f.endswith((".py", ".sh", ".jl", ".hip", ".h", ".c")) and f != os.path.basename(__file__):
                m = stale.search(open(os.path.join(d, f), errors="replace").read())
                assert not m, f"{sub}/{f} still mentions {m.group(0)}"


@pytest.mark.gpu
def test_every_option_name_is_accepted_and_unknown_names_are_refused(cmf, oracle):
    data, _, _ = oracle.c_gen_synthetic(N=24, T=200, K=3, L=6, seed=2)
    W0, H0 = oracle.c_init_rand(data, L=6, K=4, seed=1)
    lib = cmf.load_library()
    defaults = {"reuse_est": 1, "speculate": 1, "gram": 0, "conv_kernel": 0, "conv_split": 1, "small_k": 1, "small_k_fuse": 1, "hals_prepare": 1, "hals_gram": 2,
                "hals_persist": 1, "hals_general": 0, "hals_seg": 384, "hals_lag": 2, "hals_debug": 0, "hals_chase": -1, "profile": 0, "profile_mask": 0,
                "allreduce_overlap": 0, "enqueue_threads": 1, "halo_in_allreduce": 1}
    assert sorted(defaults) == sorted(option_names(lib))
    for devices in (None, [0, 0]):
        rule = cmf.MultUpdate(data, W0, H0, devices=devices)
        try:
            ref = rule.iterate(2)
            for name, value in defaults.items():
                if devices is not None and name.startswith("hals_"):
                    continue  # (the HALS rule does not shard: its options belong to single handles)
                rule.set_option(name, value)
            with pytest.raises(cmf.CMFError):
                rule.set_option("no_such_option", 1)
            rule.upload(W0, H0)
            assert np.array_equal(rule.iterate(2), ref)  # the defaults are the defaults
        finally:
            rule.close()
