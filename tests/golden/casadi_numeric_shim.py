"""Build-container-only stand-in for the `casadi` module, NUMERIC and forward-mode: expressions are lazy matrix-valued graphs, nothing is ever expanded
symbolically; `jacobian(expr, var)` is a node that - when a Function holding it is evaluated at numbers - runs the graph of `expr` on dual numbers (value [r, c] and
d value / d var [r, c, numel(var)] as numpy arrays, the var symbols seeded with the identity).  A Function called on symbolic arguments (the reference composes its
dynamics over grid cells that way, PDP.py:901-904, and builds the whole-horizon recovery matrix from Jacobian Functions called at symbolic points, PDP.py:1039-1079)
is a CALL node: its arguments are evaluated first, the callee's graph then runs on those numbers (and their derivatives: forward mode passes straight through).

Why it exists (round-3 verdict, item 3): the long recovery-matrix fixtures (tests/golden/ref_recmat_long_*.npz: the reference's own recmat_init_step(horizon, -1) at
T = 50 / 35 / 20) were generated with the PRODUCT's expression engine (pdp_amd/sx.py) standing in for CasADi, because the sympy stand-in cannot compose that
expression at length.  This module shares no code, no data structure and no differentiation method with sx.py (scalar hash-consed DAG, reverse mode, code emission):
matrix-valued nodes, forward-mode duals, numpy evaluation - an independent derivation of the same numbers.  make_recmat_long.py generates the fixtures with it and
cross-checks both other engines against it.

Only what PDP.ControlPlanning / SysID and JinEnv use of CasADi is implemented; second derivatives (a jacobian node evaluated under outer seeds) are refused -
OCSys.diffPMP needs them, the fixtures made here do not."""
import sys

import numpy as np

sys.setrecursionlimit(100000)


class SX:
    __array_ufunc__ = None              # numpy operands defer to the reflected operators below (np.identity(3) - SX ...)
    __array_priority__ = 1000

    def __init__(self, op, args=(), shape=(1, 1), data=None):
        self.op, self.args, self.shape, self.data = op, tuple(args), (int(shape[0]), int(shape[1])), data

    # ---- construction
    @staticmethod
    def sym(name, r=1, c=1):
        return SX("sym", (), (r, c), name)

    @staticmethod
    def lift(o):
        if isinstance(o, SX):
            return o
        if isinstance(o, DM):
            o = o.a
        a = np.array(o, dtype=float)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        return SX("const", (), a.shape, a)

    @staticmethod
    def zeros(r, c=1):
        return SX.lift(np.zeros((r, c)))

    def numel(self): return self.shape[0] * self.shape[1]
    def size1(self): return self.shape[0]
    def size2(self): return self.shape[1]
    def rows(self): return self.shape[0]
    def columns(self): return self.shape[1]

    @property
    def T(self):
        return SX("T", (self,), (self.shape[1], self.shape[0]))

    def reshape(self, shp, c=None):
        r, c = (shp if c is None else (shp, c))
        n = self.numel()
        r, c = (n // c if r == -1 else r), (n // r if c == -1 else c)
        assert r * c == n
        return SX("reshape", (self,), (r, c))

    def __getitem__(self, k):
        r, c = self.shape
        if isinstance(k, tuple):
            probe = np.zeros((r, c))[tuple(slice(i, i + 1) if isinstance(i, (int, np.integer)) else i for i in k)]
        else:                                   # one index: column-major linear indexing, the result is a column
            probe = np.zeros(r * c)[slice(k, k + 1) if isinstance(k, (int, np.integer)) else k].reshape(-1, 1)
        return SX("index", (self,), probe.shape, k)

    def _bin(self, o, op, swap=False):
        a, b = (SX.lift(o), self) if swap else (self, SX.lift(o))
        if a.shape != b.shape:
            assert a.numel() == 1 or b.numel() == 1, "shape mismatch %s vs %s" % (a.shape, b.shape)
        return SX(op, (a, b), a.shape if a.numel() >= b.numel() else b.shape)

    def __add__(self, o): return self._bin(o, "add")
    def __radd__(self, o): return self._bin(o, "add", True)
    def __sub__(self, o): return self._bin(o, "sub")
    def __rsub__(self, o): return self._bin(o, "sub", True)
    def __mul__(self, o): return self._bin(o, "mul")
    def __rmul__(self, o): return self._bin(o, "mul", True)
    def __truediv__(self, o): return self._bin(o, "div")
    def __rtruediv__(self, o): return self._bin(o, "div", True)
    def __pow__(self, o): return self._bin(o, "pow")
    def __rpow__(self, o): return self._bin(o, "pow", True)
    def __neg__(self): return SX("neg", (self,), self.shape)
    def __pos__(self): return self
    def __matmul__(self, o): return mtimes(self, o)


MX = SX


class DM:
    def __init__(self, a):
        self.a = np.array(a.a if isinstance(a, DM) else a, dtype=float)
        if self.a.ndim == 0:
            self.a = self.a.reshape(1, 1)
        elif self.a.ndim == 1:
            self.a = self.a.reshape(-1, 1)

    def full(self): return self.a
    def toarray(self): return self.a
    def __array__(self, dtype=None, copy=None): return self.a if dtype is None else self.a.astype(dtype)
    def __float__(self): return float(self.a.reshape(-1)[0])
    @property
    def shape(self): return self.a.shape


# ---------------------------------------------------------------------------------------------------------------------------------------
# evaluation on (value, derivative) pairs
# ---------------------------------------------------------------------------------------------------------------------------------------
_UN = {"sin": (np.sin, np.cos), "cos": (np.cos, lambda a: -np.sin(a)), "tan": (np.tan, lambda a: 1.0 / np.cos(a) ** 2), "tanh": (np.tanh, lambda a: 1.0 - np.tanh(a) ** 2),
       "exp": (np.exp, np.exp), "log": (np.log, lambda a: 1.0 / a), "sqrt": (np.sqrt, lambda a: 0.5 / np.sqrt(a))}


class _Ctx:
    def __init__(self, bind, nv):
        self.bind, self.nv, self.cache = bind, nv, {}


def _b(a, shape):                       # scalar broadcast of a value (r, c) or a derivative (r, c, nv)
    return np.broadcast_to(a, shape + a.shape[2:]) if a.shape[:2] != shape else a


def _ev(n, ctx):
    key = id(n)
    got = ctx.cache.get(key)
    if got is not None:
        return got
    nv, op = ctx.nv, n.op
    Z = lambda shape: np.zeros(shape + (nv,)) if nv else None
    if op == "sym":
        if key not in ctx.bind:
            raise KeyError("free symbol %s: not an input of the Function being evaluated" % n.data)
        res = ctx.bind[key]
    elif op == "const":
        res = (n.data, Z(n.shape))
    elif op in ("add", "sub", "mul", "div", "pow"):
        (a, da), (b, db) = _ev(n.args[0], ctx), _ev(n.args[1], ctx)
        a, b = _b(a, n.shape), _b(b, n.shape)
        if nv:
            da, db = _b(da, n.shape), _b(db, n.shape)
        if op == "add":
            res = (a + b, da + db if nv else None)
        elif op == "sub":
            res = (a - b, da - db if nv else None)
        elif op == "mul":
            res = (a * b, da * b[..., None] + a[..., None] * db if nv else None)
        elif op == "div":
            v = a / b
            res = (v, (da - v[..., None] * db) / b[..., None] if nv else None)
        else:
            v = a ** b
            d = None
            if nv:
                d = (b * a ** (b - 1))[..., None] * da
                if np.any(db != 0.0):
                    d = d + (np.log(a) * v)[..., None] * db
            res = (v, d)
    elif op == "neg":
        a, da = _ev(n.args[0], ctx)
        res = (-a, -da if nv else None)
    elif op == "matmul":
        (a, da), (b, db) = _ev(n.args[0], ctx), _ev(n.args[1], ctx)
        res = (a @ b, np.einsum("ikv,kj->ijv", da, b) + np.einsum("ik,kjv->ijv", a, db) if nv else None)
    elif op == "T":
        a, da = _ev(n.args[0], ctx)
        res = (a.T, da.transpose(1, 0, 2) if nv else None)
    elif op == "reshape":
        a, da = _ev(n.args[0], ctx)
        r, c = n.shape
        res = (a.reshape((r, c), order="F"), da.transpose(1, 0, 2).reshape(c, r, nv).transpose(1, 0, 2) if nv else None)
    elif op == "index":
        a, da = _ev(n.args[0], ctx)
        k = n.data
        if isinstance(k, tuple):
            kk = tuple(slice(i, i + 1) if isinstance(i, (int, np.integer)) else i for i in k)
            res = (a[kk], da[kk] if nv else None)
        else:
            kk = slice(k, k + 1) if isinstance(k, (int, np.integer)) else k
            res = (a.reshape(-1, order="F")[kk].reshape(-1, 1), da.transpose(1, 0, 2).reshape(-1, nv)[kk].reshape(-1, 1, nv) if nv else None)
    elif op in ("vcat", "hcat"):
        parts = [_ev(x, ctx) for x in n.args]
        ax = 0 if op == "vcat" else 1
        res = (np.concatenate([p[0] for p in parts], axis=ax), np.concatenate([p[1] for p in parts], axis=ax) if nv else None)
    elif op in _UN:
        a, da = _ev(n.args[0], ctx)
        f, df = _UN[op]
        res = (f(a), df(a)[..., None] * da if nv else None)
    elif op == "inv":
        a, da = _ev(n.args[0], ctx)
        ia = np.linalg.inv(a)
        res = (ia, -np.einsum("ik,klv,lj->ijv", ia, da, ia) if nv else None)
    elif op == "call":
        fn, k = n.data
        vals = [_ev(x, ctx) for x in n.args]
        res = fn._eval([v[0] for v in vals], [v[1] for v in vals], nv)[k]
    elif op == "jac":
        if nv:
            raise NotImplementedError("second derivatives (a jacobian inside a differentiated expression) are not part of the numeric stand-in")
        expr, leaves = n.args[0], n.data
        nvar = sum(s.numel() for s in leaves)
        bind = {k_: (v[0], np.zeros(v[0].shape + (nvar,))) for k_, v in ctx.bind.items()}
        off = 0
        for s in leaves:
            v = ctx.bind[id(s)][0]
            d = np.zeros(v.shape + (nvar,))
            for q in range(s.numel()):                      # column-major element order
                d[q % s.shape[0], q // s.shape[0], off + q] = 1.0
            bind[id(s)] = (v, d)
            off += s.numel()
        v, d = _ev(expr, _Ctx(bind, nvar))
        res = (d.transpose(1, 0, 2).reshape(v.size, nvar), None)
    else:
        raise NotImplementedError(op)
    ctx.cache[key] = res
    return res


def _leaves(var):
    """the symbols a pure symbolic matrix is made of, in column-major element order (the reference differentiates with respect to vertcat's of scalar symbols)"""
    var = SX.lift(var)
    if var.op == "sym":
        return [var]
    if var.op == "vcat" and var.shape[1] == 1 or var.op == "hcat" and var.shape[0] == 1:
        return [s for a in var.args for s in _leaves(a)]
    if var.op == "reshape" and var.args[0].shape[1] == 1 and var.shape[1] == 1:
        return _leaves(var.args[0])
    raise NotImplementedError("jacobian / Function input must be purely symbolic (symbols, vertcat of symbols): got '%s'" % var.op)


# ---------------------------------------------------------------------------------------------------------------------------------------
# the casadi functions the reference uses
# ---------------------------------------------------------------------------------------------------------------------------------------
def _cat(items, op):
    items = [SX.lift(i) for i in items if not (isinstance(i, (list, tuple)) and len(i) == 0)]
    items = [i for i in items if i.numel() > 0]
    if not items:
        return SX.zeros(0, 1)
    if op == "vcat":
        assert len(set(i.shape[1] for i in items)) == 1
        return SX(op, items, (sum(i.shape[0] for i in items), items[0].shape[1]))
    assert len(set(i.shape[0] for i in items)) == 1
    return SX(op, items, (items[0].shape[0], sum(i.shape[1] for i in items)))


def vertcat(*a): return _cat(a, "vcat")
def horzcat(*a): return _cat(a, "hcat")
def vcat(lst): return _cat(lst, "vcat")
def hcat(lst): return _cat(lst, "hcat")
def veccat(*a): return _cat([SX.lift(x).reshape((-1, 1)) for x in a], "vcat")


def mtimes(a, b, *more):
    a, b = SX.lift(a), SX.lift(b)
    if a.numel() == 1 or b.numel() == 1:
        out = a * b
    else:
        assert a.shape[1] == b.shape[0], "mtimes: %s x %s" % (a.shape, b.shape)
        out = SX("matmul", (a, b), (a.shape[0], b.shape[1]))
    return mtimes(out, *more) if more else out


def transpose(a): return SX.lift(a).T
def reshape(a, r, c=None): return SX.lift(a).reshape(r, c)
def inv(a): return SX("inv", (SX.lift(a),), SX.lift(a).shape)
def dot(a, b): return mtimes(SX.lift(a).reshape((1, -1)), SX.lift(b).reshape((-1, 1)))
def sumsqr(a): return dot(a, a)
def norm_2(a): return sqrt(sumsqr(a))
def power(a, b): return SX.lift(a) ** b


def trace(a):
    a = SX.lift(a)
    t = a[0, 0]
    for i in range(1, a.shape[0]):
        t = t + a[i, i]
    return t


def diag(a):
    a = SX.lift(a)
    if a.shape[1] == 1:
        n = a.shape[0]
        return vcat([hcat([a[i] if i == j else SX.zeros(1, 1) for j in range(n)]) for i in range(n)])
    return vcat([a[i, i] for i in range(a.shape[0])])


def _un(name):
    def f(a):
        if isinstance(a, (SX, DM)) or (isinstance(a, np.ndarray) and a.dtype == object):
            a = SX.lift(a)
            return SX(name, (a,), a.shape)
        return _UN[name][0](a)
    f.__name__ = name
    return f


sin, cos, tan, tanh, exp, log, sqrt = (_un(k) for k in ("sin", "cos", "tan", "tanh", "exp", "log", "sqrt"))


def jacobian(expr, var):
    expr, leaves = SX.lift(expr), _leaves(var)
    return SX("jac", (expr,), (expr.numel(), sum(s.numel() for s in leaves)), leaves)


class Function:
    def __init__(self, name, ins, outs, *a, **k):
        self.name = name
        self.ins = [_leaves(i) for i in ins]
        self.in_numel = [SX.lift(i).numel() for i in ins]
        self.outs = [SX.lift(o) for o in outs]

    def _eval(self, vals, dots, nv):
        bind = {}
        for leaves, total, v, d in zip(self.ins, self.in_numel, vals, dots):
            v = np.asarray(v, dtype=float)
            if v.size == 1 and total > 1:                         # CasADi broadcasts a scalar argument to the declared size
                v = np.full((total, 1), float(v.reshape(-1)[0]))
                d = np.broadcast_to(d, (total, 1, nv)) if nv else None
            assert v.size == total, "%s: an argument has %d elements, %d expected" % (self.name, v.size, total)
            flat = v.reshape(-1, order="F")
            dflat = d.transpose(1, 0, 2).reshape(total, nv) if nv else None
            off = 0
            for s in leaves:
                k = s.numel()
                bind[id(s)] = (flat[off:off + k].reshape(s.shape, order="F"), dflat[off:off + k].reshape(s.shape[1], s.shape[0], nv).transpose(1, 0, 2) if nv else None)
                off += k
        ctx = _Ctx(bind, nv)
        return [_ev(o, ctx) for o in self.outs]

    def __call__(self, *args):
        assert len(args) == len(self.ins), "%s: expected %d arguments" % (self.name, len(self.ins))
        if any(isinstance(a, SX) for a in args):
            args = [SX.lift(a) for a in args]
            res = [SX("call", args, o.shape, (self, k)) for k, o in enumerate(self.outs)]
        else:
            vals = []
            for a in args:
                a = np.array(a.a if isinstance(a, DM) else a, dtype=float)
                vals.append(a.reshape(-1, 1) if a.ndim <= 1 else a)
            res = [DM(v) for v, _ in self._eval(vals, [None] * len(vals), 0)]
        return res[0] if len(res) == 1 else tuple(res)


def nlpsol(*a, **k):
    raise NotImplementedError("IPOPT is not part of the numeric stand-in")


casadi = sys.modules[__name__]          # `from casadi import *` also brings the name `casadi` (the reference writes casadi.Function, casadi.jacobian)
numpy = np                              # ... and `numpy` (JinEnv.py uses it without importing it)


def install():
    """make `import casadi` / `from casadi import *` resolve to this module"""
    sys.modules["casadi"] = sys.modules[__name__]
    return sys.modules[__name__]


__all__ = [k for k in dir() if not k.startswith("_") and k not in ("sys",)]
