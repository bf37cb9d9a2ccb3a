"""Minimal HDF5 reader/writer over the system libhdf5 (ctypes), for the reference's results schema.

The reference stores a fit with HDF5.jl (src/model.jl:149-181): datasets ``W, H, data, loss_hist, time_hist``
(Float64 arrays), ``l1_H, l2_H, l1_W, l2_W`` (Float64 scalars) and ``alg`` (a string).  HDF5.jl writes a Julia
array with its dimensions reversed (HDF5 is row-major, Julia column-major), so a K x N x L tensor is an HDF5
dataset of shape (L, N, K) holding the column-major bytes: exactly the buffer of a Fortran-ordered NumPy array.
This module writes and reads that convention, so files move between CMF.jl and this package unchanged.

There is no h5py in the image; libhdf5 itself ships with it (probed in this order: $CMF_LIBHDF5,
/opt/conda/lib, the loader path).  When it cannot be loaded, HDF5Unavailable is raised and callers fall back to
the .npz container.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import os

import numpy as np


class HDF5Unavailable(RuntimeError):
    pass


_lib = None
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get("CMF_LIBHDF5"), "/opt/conda/lib/libhdf5.so", ctypes.util.find_library("hdf5"),
             "libhdf5.so", "libhdf5_serial.so"]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            lib = ctypes.CDLL(c)
            break
        except OSError as e:  # keep looking
            err = e
    else:
        raise HDF5Unavailable(f"libhdf5 could not be loaded ({err}); use a .npz path instead")
    if lib.H5open() < 0:
        raise HDF5Unavailable("H5open failed")
    maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise HDF5Unavailable(f"libhdf5 {maj.value}.{mnr.value} has 32-bit ids; 1.10 or newer is needed")

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, list(args)

    cp, vp = ctypes.c_char_p, ctypes.c_void_p
    sig("H5Fcreate", hid_t, cp, ctypes.c_uint, hid_t, hid_t)
    sig("H5Fopen", hid_t, cp, ctypes.c_uint, hid_t)
    sig("H5Fclose", ctypes.c_int, hid_t)
    sig("H5Screate", hid_t, ctypes.c_int)
    sig("H5Screate_simple", hid_t, ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t))
    sig("H5Sclose", ctypes.c_int, hid_t)
    sig("H5Sget_simple_extent_ndims", ctypes.c_int, hid_t)
    sig("H5Sget_simple_extent_dims", ctypes.c_int, hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t))
    sig("H5Dcreate2", hid_t, hid_t, cp, hid_t, hid_t, hid_t, hid_t, hid_t)
    sig("H5Dopen2", hid_t, hid_t, cp, hid_t)
    sig("H5Dwrite", ctypes.c_int, hid_t, hid_t, hid_t, hid_t, hid_t, vp)
    sig("H5Dread", ctypes.c_int, hid_t, hid_t, hid_t, hid_t, hid_t, vp)
    sig("H5Dclose", ctypes.c_int, hid_t)
    sig("H5Dget_space", hid_t, hid_t)
    sig("H5Dget_type", hid_t, hid_t)
    sig("H5Tcopy", hid_t, hid_t)
    sig("H5Tset_size", ctypes.c_int, hid_t, ctypes.c_size_t)
    sig("H5Tget_size", ctypes.c_size_t, hid_t)
    sig("H5Tget_class", ctypes.c_int, hid_t)
    sig("H5Tis_variable_str", ctypes.c_int, hid_t)
    sig("H5Tclose", ctypes.c_int, hid_t)
    sig("H5Lexists", ctypes.c_int, hid_t, cp, hid_t)
    sig("H5Eset_auto2", ctypes.c_int, hid_t, vp, vp)
    sig("H5Dvlen_reclaim", ctypes.c_int, hid_t, hid_t, hid_t, vp)
    lib.H5Eset_auto2(0, None, None)  # errors come back as negative ids; no stderr stack dumps
    lib._f64 = ctypes.c_int64.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
    lib._i64 = ctypes.c_int64.in_dll(lib, "H5T_NATIVE_INT64_g").value
    lib._cs1 = ctypes.c_int64.in_dll(lib, "H5T_C_S1_g").value
    _lib = lib
    return lib


def available():
    try:
        _load()
        return True
    except HDF5Unavailable:
        return False


def _chk(v, what):
    if v < 0:
        raise OSError(f"HDF5: {what} failed")
    return v


def write_file(path, items):
    """items: name -> float array (any rank; stored with reversed dimensions like HDF5.jl does), float scalar or str."""
    lib = _load()
    f = _chk(lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"H5Fcreate({path})")
    try:
        for name, val in items.items():
            key = name.encode()
            if isinstance(val, str):
                raw = val.encode("utf-8")
                t = _chk(lib.H5Tcopy(lib._cs1), "H5Tcopy")
                lib.H5Tset_size(t, max(1, len(raw)))
                s = _chk(lib.H5Screate(H5S_SCALAR), "H5Screate")
                d = _chk(lib.H5Dcreate2(f, key, t, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({name})")
                buf = ctypes.create_string_buffer(raw, max(1, len(raw)))
                _chk(lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), f"H5Dwrite({name})")
                lib.H5Dclose(d); lib.H5Sclose(s); lib.H5Tclose(t)
                continue
            a = np.asarray(val, dtype=np.float64)
            if a.ndim == 0:
                s = _chk(lib.H5Screate(H5S_SCALAR), "H5Screate")
                buf = np.array([float(a)], dtype=np.float64)
            else:
                buf = np.asfortranarray(a)  # Julia's memory order
                dims = (hsize_t * a.ndim)(*reversed(a.shape))
                s = _chk(lib.H5Screate_simple(a.ndim, dims, None), "H5Screate_simple")
            d = _chk(lib.H5Dcreate2(f, key, lib._f64, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({name})")
            _chk(lib.H5Dwrite(d, lib._f64, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.ctypes.data_as(ctypes.c_void_p)), f"H5Dwrite({name})")
            lib.H5Dclose(d); lib.H5Sclose(s)
    finally:
        lib.H5Fclose(f)


def read_file(path, names):
    """Returns {name: value} for those of `names` present in the file (arrays come back in Julia's shape)."""
    lib = _load()
    f = _chk(lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT), f"H5Fopen({path})")
    out = {}
    try:
        for name in names:
            key = name.encode()
            if lib.H5Lexists(f, key, H5P_DEFAULT) <= 0:
                continue
            d = _chk(lib.H5Dopen2(f, key, H5P_DEFAULT), f"H5Dopen2({name})")
            t = lib.H5Dget_type(d)
            s = lib.H5Dget_space(d)
            try:
                cls = lib.H5Tget_class(t)
                nd = max(0, lib.H5Sget_simple_extent_ndims(s))
                dims = (hsize_t * max(nd, 1))()
                if nd:
                    lib.H5Sget_simple_extent_dims(s, dims, None)
                shape = tuple(int(x) for x in dims[:nd])
                if cls == H5T_STRING:
                    if lib.H5Tis_variable_str(t) > 0:
                        p = ctypes.c_char_p()
                        _chk(lib.H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, ctypes.byref(p)), f"H5Dread({name})")
                        out[name] = (p.value or b"").decode("utf-8")
                        lib.H5Dvlen_reclaim(t, s, H5P_DEFAULT, ctypes.byref(p))
                    else:
                        n = lib.H5Tget_size(t)
                        buf = ctypes.create_string_buffer(n + 1)
                        _chk(lib.H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), f"H5Dread({name})")
                        out[name] = buf.raw[:n].split(b"\0", 1)[0].decode("utf-8")
                elif cls in (H5T_FLOAT, H5T_INTEGER):
                    buf = np.empty(int(np.prod(shape)) if shape else 1, dtype=np.float64)
                    _chk(lib.H5Dread(d, lib._f64, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.ctypes.data_as(ctypes.c_void_p)), f"H5Dread({name})")
                    out[name] = buf.reshape(tuple(reversed(shape)), order="F") if shape else float(buf[0])
                else:
                    raise OSError(f"HDF5: dataset {name} has an unsupported type class {cls}")
            finally:
                lib.H5Sclose(s); lib.H5Tclose(t); lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)
    return out
