"""Build-container-only helper: a sympy-backed stand-in for the tiny part of the `casadi` API that the
reference's PDP/PDP.py and JinEnv/JinEnv.py touch, so that the reference can be IMPORTED UNMODIFIED
from /root/reference to produce golden outputs (tests/golden/make_ref_outputs.py).  CasADi itself is
not installable here.  Independent of the product's symbolic engine on purpose.
Never used by the product, never needed on the GPU box (only the resulting .npz files travel).
"""
import sys
import types
import numpy
import numpy as np
import sympy as sp

_counter = [0]


class SX:
    __array_ufunc__ = None          # make numpy defer to our reflected operators
    __array_priority__ = 1000

    def __init__(self, mat):
        self.m = sp.Matrix(mat) if not isinstance(mat, sp.MatrixBase) else mat

    # -- construction ------------------------------------------------------------------------
    @staticmethod
    def sym(name, r=1, c=1):
        _counter[0] += 1
        uid = _counter[0]
        if r == 1 and c == 1:
            return SX(sp.Matrix([[sp.Symbol("%s__%d" % (name, uid), real=True)]]))
        return SX(sp.Matrix(r, c, lambda i, j: sp.Symbol("%s__%d_%d_%d" % (name, uid, i, j), real=True)))

    @staticmethod
    def _lift(o):
        if isinstance(o, SX):
            return o
        if isinstance(o, (int, float, numpy.floating, numpy.integer)):
            return SX(sp.Matrix([[sp.Float(float(o)) if isinstance(o, (float, numpy.floating)) else sp.Integer(int(o))]]))
        a = numpy.asarray(o, dtype=float)
        if a.ndim == 0:
            return SX(sp.Matrix([[sp.Float(float(a))]]))
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        return SX(sp.Matrix(a.shape[0], a.shape[1], lambda i, j: sp.Float(a[i, j])))

    # -- shape -----------------------------------------------------------------------------
    @property
    def shape(self):
        return self.m.shape

    def numel(self):
        return self.m.shape[0] * self.m.shape[1]

    def size1(self):
        return self.m.shape[0]

    def size2(self):
        return self.m.shape[1]

    @property
    def T(self):
        return SX(self.m.T)

    def _colmajor(self):
        r, c = self.m.shape
        return [self.m[i, j] for j in range(c) for i in range(r)]

    def reshape(self, shp):
        flat = self._colmajor()
        r, c = shp
        n = len(flat)
        if r == -1:
            r = n // c
        if c == -1:
            c = n // r
        return SX(sp.Matrix(r, c, lambda i, j: flat[i + j * r]))

    def __getitem__(self, k):
        if isinstance(k, tuple):
            sub = self.m[k[0], k[1]]
            return SX(sub) if isinstance(sub, sp.MatrixBase) else SX(sp.Matrix([[sub]]))
        flat = self._colmajor()
        if isinstance(k, slice):
            return SX(sp.Matrix(flat[k]))
        return SX(sp.Matrix([[flat[k]]]))

    # -- arithmetic (CasADi semantics: * and / are element-wise with scalar broadcasting) ----------
    def _bin(self, o, f):
        o = SX._lift(o)
        a, b = self.m, o.m
        if a.shape == b.shape:
            return SX(sp.Matrix(a.shape[0], a.shape[1], lambda i, j: f(a[i, j], b[i, j])))
        if a.shape == (1, 1):
            return SX(sp.Matrix(b.shape[0], b.shape[1], lambda i, j: f(a[0, 0], b[i, j])))
        if b.shape == (1, 1):
            return SX(sp.Matrix(a.shape[0], a.shape[1], lambda i, j: f(a[i, j], b[0, 0])))
        raise ValueError("shape mismatch %s vs %s" % (a.shape, b.shape))

    def __add__(self, o): return self._bin(o, lambda x, y: x + y)
    def __radd__(self, o): return SX._lift(o)._bin(self, lambda x, y: x + y)
    def __sub__(self, o): return self._bin(o, lambda x, y: x - y)
    def __rsub__(self, o): return SX._lift(o)._bin(self, lambda x, y: x - y)
    def __mul__(self, o): return self._bin(o, lambda x, y: x * y)
    def __rmul__(self, o): return SX._lift(o)._bin(self, lambda x, y: x * y)
    def __truediv__(self, o): return self._bin(o, lambda x, y: x / y)
    def __rtruediv__(self, o): return SX._lift(o)._bin(self, lambda x, y: x / y)
    def __pow__(self, o): return self._bin(o, lambda x, y: x ** y)
    def __neg__(self): return SX(-self.m)

    def __repr__(self):
        return "SX(%s)" % (self.m,)


MX = SX


def _cat_v(items):
    mats = [SX._lift(i).m for i in items]
    mats = [m for m in mats if m.shape[0] * m.shape[1] > 0]
    if not mats:
        return SX(sp.zeros(0, 1))
    return SX(sp.Matrix.vstack(*mats))


def _cat_h(items):
    mats = [SX._lift(i).m for i in items]
    return SX(sp.Matrix.hstack(*mats))


def vertcat(*a): return _cat_v(a)
def horzcat(*a): return _cat_h(a)
def vcat(lst): return _cat_v(lst)
def hcat(lst): return _cat_h(lst)
def mtimes(a, b): return SX(SX._lift(a).m * SX._lift(b).m)
def transpose(a): return SX(SX._lift(a).m.T)
def trace(a): return SX(sp.Matrix([[SX._lift(a).m.trace()]]))
def diag(a):
    v = SX._lift(a)._colmajor()
    return SX(sp.diag(*v))
def inv(a): return SX(SX._lift(a).m.inv())
def dot(a, b):
    x, y = SX._lift(a)._colmajor(), SX._lift(b)._colmajor()
    return SX(sp.Matrix([[sum(p * q for p, q in zip(x, y))]]))
def _ew(f):
    def g(a):
        if isinstance(a, SX):
            return SX(a.m.applyfunc(f))
        return getattr(numpy, f.__name__)(a)
    return g
sin, cos, tanh, exp, sqrt = _ew(sp.sin), _ew(sp.cos), _ew(sp.tanh), _ew(sp.exp), _ew(sp.sqrt)


def jacobian(expr, var):
    e, v = SX._lift(expr), SX._lift(var)
    return SX(sp.Matrix(e._colmajor()).jacobian(sp.Matrix(v._colmajor())))


class DM:
    def __init__(self, a):
        self.a = numpy.atleast_2d(numpy.asarray(a, dtype=float))

    def full(self):
        return self.a

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __float__(self):
        return float(self.a.squeeze())


class Function:
    def __init__(self, name, ins, outs, *a, **k):
        self.name = name
        self.ins = [SX._lift(i) for i in ins]
        self.outs = [SX._lift(o) for o in outs]
        args = [i._colmajor() for i in self.ins]
        self._f = [sp.lambdify(args, o.m, modules="numpy", cse=True) for o in self.outs]

    def __call__(self, *vals):
        if any(isinstance(v, SX) for v in vals):                 # symbolic call: substitution (used by warp_dynCost / recmat)
            sub = {}
            for v, i in zip(vals, self.ins):
                v = SX._lift(v)
                for a, b in zip(i._colmajor(), v._colmajor()):
                    sub[a] = b
            res = [SX(o.m.xreplace(sub)) for o in self.outs]
            return res[0] if len(res) == 1 else tuple(res)
        flat = []
        for v, i in zip(vals, self.ins):
            if isinstance(v, DM):
                v = v.a
            a = numpy.asarray(v, dtype=float)
            a = a.reshape(-1, order="F") if a.ndim == 2 else a.reshape(-1)
            if a.size == 1 and i.numel() > 1:
                a = numpy.full(i.numel(), float(a[0]))
            assert a.size == i.numel(), "%s: arg size %d != %d" % (self.name, a.size, i.numel())
            flat.append(a)
        res = [DM(numpy.asarray(f(*flat), dtype=float).reshape(o.shape)) for f, o in zip(self._f, self.outs)]
        return res[0] if len(res) == 1 else tuple(res)


def nlpsol(*a, **k):
    raise RuntimeError("IPOPT is not available in the build container (casadi shim)")


def install():
    """Register this module as `casadi` so that `from casadi import *` inside the reference works."""
    mod = types.ModuleType("casadi")
    names = ["SX", "MX", "DM", "Function", "vertcat", "horzcat", "vcat", "hcat", "mtimes", "transpose", "trace", "diag", "inv", "dot",
             "sin", "cos", "tanh", "exp", "sqrt", "jacobian", "nlpsol", "np", "numpy"]
    g = globals()
    for n in names:
        setattr(mod, n, g[n])
    mod.casadi = mod
    mod.__all__ = names + ["casadi"]
    sys.modules["casadi"] = mod
    return mod
