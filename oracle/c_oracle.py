"""TEST INFRASTRUCTURE ONLY: ctypes access to oracle/libpdp_oracle.so (C restatement of the reference's IRL inner loop)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpdp_oracle.so")
_lib = None


def build_native(out_dir=None):
    """Build the same sources with -O3 -march=native for the CPU this process runs on (bench.py's cpu_baseline leg: the portable
    libpdp_oracle.so is compiled in the build container with -O2 and no -march so that it loads on any host; the timed baseline
    should not be handicapped by that).  Returns the path, or None when no compiler is available."""
    import glob
    import tempfile
    out_dir = out_dir or tempfile.gettempdir()
    out = os.path.join(out_dir, "libpdp_oracle_native_%d.so" % os.getuid())
    src = [os.path.join(HERE, "pdp_oracle.c")] + sorted(glob.glob(os.path.join(HERE, "gen", "*_oc.c")))
    try:
        subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-ffp-contract=off", "-shared", "-o", out] + src + ["-lm"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    except Exception:
        return None
    return out


def load(build=True, path=None):
    global _lib
    if path is not None:                      # a specific build (build_native): not cached as the default library
        return _prepare(C.CDLL(path))
    if _lib is None:
        if not os.path.exists(LIB):
            if not build:
                raise RuntimeError("oracle/libpdp_oracle.so not built")
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _lib = _prepare(C.CDLL(LIB))
    return _lib


def _prepare(lib):
    lib.pdp_oracle_model_name.restype = C.c_char_p
    lib.pdp_oracle_oc_unit.restype = C.c_int
    lib.pdp_oracle_oc_unit.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 8 + [C.c_int]
    lib.models = {lib.pdp_oracle_model_name(i).decode(): i for i in range(lib.pdp_oracle_n_models())}
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def oc_unit(lib, system, u, theta, demo_x, demo_u, x0=None, x=None, lam=None, want_sens=False, threads=0, out=None):
    """out: dict of preallocated x, lam, loss, grad arrays (timed loops: no allocation / page faults inside the measurement)"""
    mid = lib.models[system]
    d = (C.c_int * 3)()
    lib.pdp_oracle_model_dims(mid, d)
    n, m, p = d[0], d[1], d[2]
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    u, demo_x, demo_u, theta = c(u), c(demo_x), c(demo_u), c(theta)
    B, T = u.shape[0], u.shape[1]
    tb = p if theta.ndim == 2 and theta.shape[0] == B and B > 1 else 0
    given = 0
    if x is not None:
        xs, ls, given = c(x).copy(), c(lam).copy(), 1
    else:
        xs, ls = (out["x"], out["lam"]) if out is not None else (np.zeros((B, T + 1, n)), np.zeros((B, T, n)))
        x0 = c(x0).reshape(B, n)
    loss, grad = (out["loss"], out["grad"]) if out is not None else (np.zeros(B), np.zeros((B, p)))
    X = np.zeros((B, T + 1, n, p)) if want_sens else None
    U = np.zeros((B, T, m, p)) if want_sens else None
    rc = lib.pdp_oracle_oc_unit(mid, B, T, given, _p(x0), _p(u), _p(theta), tb, _p(demo_x), _p(demo_u), _p(xs), _p(ls), _p(loss), _p(grad), _p(X), _p(U),
                                int(threads))
    assert rc == 0
    return dict(loss=loss, grad=grad, x=xs, lam=ls, dxdp=X, dudp=U)


def quadrotor_oc_unit(lib, x0, u, theta, demo_x, demo_u, threads=0, out=None):
    return oc_unit(lib, "quadrotor", u, theta, demo_x, demo_u, x0=x0, threads=threads, out=out)
