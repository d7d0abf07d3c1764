"""libpcad_host.so: native host-side helpers that are NOT on the GPU path (today: the XGBoost tree-ensemble evaluator behind
`xgb_predict.py`, host/xgb_eval.c).  Plain C + OpenMP, built in-tree with gcc by `__graft_entry__.build()` / on first use;
bound through ctypes.  Separate from libpcad.so on purpose: no HIP, loads on a machine without a GPU."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "host", "xgb_eval.c")
LIB_PATH = os.path.join(_HERE, "libpcad_host.so")
_lib = None


def build_library(force: bool = False) -> str:
    """gcc -O3 -fopenmp -shared, rebuilt when the source is newer than the library."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(SRC):
        tmp = LIB_PATH + ".%d.tmp" % os.getpid()
        r = subprocess.run(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-Wall", "-o", tmp, SRC, "-lm"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building libpcad_host.so failed:\n" + r.stdout + r.stderr)
        os.replace(tmp, LIB_PATH)          # atomic: several ranks may build at once
    return LIB_PATH


def load_library():
    global _lib
    if _lib is None:
        build_library()
        lib = C.CDLL(LIB_PATH)
        lib.pcad_host_version.restype = C.c_int
        lib.pcad_xgb_margin.restype = C.c_int
        lib.pcad_xgb_margin.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 6 + [C.c_double, C.c_void_p]
        _lib = lib
    return _lib


def xgb_margin(X: np.ndarray, n_features: int, tree_off: np.ndarray, left: np.ndarray, right: np.ndarray, feat: np.ndarray,
               cond: np.ndarray, dleft: np.ndarray, base_margin: float) -> np.ndarray:
    """Margins [rows] (float64) of a flattened gbtree ensemble (arrays as documented in host/xgb_eval.c) on X [rows, >= n_features] fp32."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    if X.ndim != 2 or X.shape[1] < n_features:
        raise ValueError(f"X must be [rows, >= {n_features}] (got {X.shape})")
    out = np.empty(X.shape[0], dtype=np.float64)
    arrs = [np.ascontiguousarray(tree_off, dtype=np.int64), np.ascontiguousarray(left, dtype=np.int32),
            np.ascontiguousarray(right, dtype=np.int32), np.ascontiguousarray(feat, dtype=np.int32),
            np.ascontiguousarray(cond, dtype=np.float32), np.ascontiguousarray(dleft, dtype=np.uint8)]
    rc = load_library().pcad_xgb_margin(X.ctypes.data, X.shape[0], X.shape[1], int(n_features), len(arrs[0]) - 1,
                                        *[a.ctypes.data for a in arrs], float(base_margin), out.ctypes.data)
    if rc != 0:
        raise RuntimeError("pcad_xgb_margin rejected its arguments")
    return out
