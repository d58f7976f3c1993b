"""One-time symbolic front-end of the MI355X PDP framework: a CasADi-`SX`-compatible scalar expression
DAG with hash-consing (= common-subexpression elimination by construction), reverse-mode automatic
differentiation, and straight-line code emission (HIP device code, C, or Python).

Role: the reference delegates `jacobian(...)` and Function evaluation to CasADi (PDP/PDP.py:235-270,
674-759, 1180-1188).  Here symbolic work happens ONCE per problem, at setup; its product is generated
HIP source (see codegen.py) that the batched kernels call.  CasADi is not a dependency.

API subset mirrored (what PDP.py / JinEnv.py / the example scripts use):
  SX.sym, MX.sym, vertcat, horzcat, vcat, hcat, jacobian, Function, mtimes, dot, inv, diag, transpose/.T,
  trace, reshape (column-major), numel, sin, cos, tan, tanh, exp, log, sqrt, power, DM(.full()).
Element order is column-major like CasADi (PDP.py:740 relies on it for the MLP weight layout).
"""
import math
import numbers
import struct
import zlib

import numpy as np

# ------------------------------------------------------------------------------------------------
# scalar DAG
# ------------------------------------------------------------------------------------------------
_UNARY = ("neg", "sin", "cos", "tan", "tanh", "exp", "log", "sqrt")
_BINARY = ("add", "sub", "mul", "div", "pow")


class Node:
    __slots__ = ("op", "a", "b", "val", "name", "id", "h")

    def __repr__(self):
        if self.op == "const":
            return repr(self.val)
        if self.op == "sym":
            return self.name
        if self.op in _UNARY:
            return "%s(%r)" % (self.op, self.a)
        return "%s(%r,%r)" % (self.op, self.a, self.b)


_table = {}
_next_id = [0]
_OPCODE = {k: i + 3 for i, k in enumerate(_UNARY + _BINARY)}


def _mk(op, a=None, b=None, val=None, name=None):
    if op == "const":
        key = ("const", val if val == val else "nan", math.copysign(1.0, val))
    elif op == "sym":
        key = None
    else:
        key = (op, a.id, b.id if b is not None else -1)
    if key is not None:
        n = _table.get(key)
        if n is not None:
            return n
    n = Node()
    n.op, n.a, n.b, n.val, n.name = op, a, b, val, name
    # structural hash (deterministic across processes and creation orders): canonical operand order for + and *
    if op == "const":
        n.h = zlib.crc32(struct.pack("<d", val)) | (1 << 40)
    elif op == "sym":
        n.h = zlib.crc32(name.encode()) | (2 << 40)
    else:
        n.h = ((a.h * 1000003) ^ ((b.h if b is not None else 7) * 998244353) ^ (_OPCODE[op] << 50)) & 0xFFFFFFFFFFFFFFF
    n.id = _next_id[0]
    _next_id[0] += 1
    if key is not None:
        _table[key] = n
    return n


def const(v):
    return _mk("const", val=float(v))


ZERO = const(0.0)
ONE = const(1.0)


def _is(n, v):
    return n.op == "const" and n.val == v


def add(a, b):
    if a.op == "const" and b.op == "const":
        return const(a.val + b.val)
    if _is(a, 0.0):
        return b
    if _is(b, 0.0):
        return a
    if b.op == "neg":
        return sub(a, b.a)
    if a.op == "neg":
        return sub(b, a.a)
    if (a.h, a.id) > (b.h, b.id):
        a, b = b, a
    return _mk("add", a, b)


def sub(a, b):
    if a.op == "const" and b.op == "const":
        return const(a.val - b.val)
    if _is(b, 0.0):
        return a
    if _is(a, 0.0):
        return neg(b)
    if a is b:
        return ZERO
    if b.op == "neg":
        return add(a, b.a)
    return _mk("sub", a, b)


def neg(a):
    if a.op == "const":
        return const(-a.val)
    if a.op == "neg":
        return a.a
    if a.op == "sub":
        return sub(a.b, a.a)
    return _mk("neg", a)


def mul(a, b):
    if a.op == "const" and b.op == "const":
        return const(a.val * b.val)
    if _is(a, 0.0) or _is(b, 0.0):
        return ZERO
    if _is(a, 1.0):
        return b
    if _is(b, 1.0):
        return a
    if _is(a, -1.0):
        return neg(b)
    if _is(b, -1.0):
        return neg(a)
    if a.op == "neg" and b.op == "neg":
        return mul(a.a, b.a)
    if a.op == "neg":
        return neg(mul(a.a, b))
    if b.op == "neg":
        return neg(mul(a, b.a))
    if (a.h, a.id) > (b.h, b.id):
        a, b = b, a
    return _mk("mul", a, b)


def div(a, b):
    if a.op == "const" and b.op == "const":
        return const(a.val / b.val)
    if _is(a, 0.0):
        return ZERO
    if _is(b, 1.0):
        return a
    if _is(b, -1.0):
        return neg(a)
    if a.op == "neg":
        return neg(div(a.a, b))
    if b.op == "neg":
        return neg(div(a, b.a))
    return _mk("div", a, b)


def powr(a, b):
    if a.op == "const" and b.op == "const":
        return const(a.val ** b.val)
    if b.op == "const":
        e = b.val
        if e == 0.0:
            return ONE
        if e == 1.0:
            return a
        if e == 2.0:
            return mul(a, a)
        if e == 3.0:
            return mul(mul(a, a), a)
        if e == 4.0:
            s = mul(a, a)
            return mul(s, s)
        if e == -1.0:
            return div(ONE, a)
        if e == -2.0:
            return div(ONE, mul(a, a))
        if e == 0.5:
            return unary("sqrt", a)
    return _mk("pow", a, b)


_PYF = {"sin": math.sin, "cos": math.cos, "tan": math.tan, "tanh": math.tanh, "exp": math.exp, "log": math.log, "sqrt": math.sqrt}


def unary(op, a):
    if op == "neg":
        return neg(a)
    if a.op == "const":
        return const(_PYF[op](a.val))
    if op in ("sin", "tan", "tanh") and a.op == "neg":      # odd functions: pull the sign out (better CSE)
        return neg(unary(op, a.a))
    if op == "cos" and a.op == "neg":
        return unary(op, a.a)
    return _mk(op, a)


def sym(name):
    return _mk("sym", name=name)


def as_node(v):
    if isinstance(v, Node):
        return v
    if isinstance(v, SX):
        assert v.numel() == 1, "expected a scalar expression"
        return v.data[0]
    return const(float(v))


# ------------------------------------------------------------------------------------------------
# graph utilities
# ------------------------------------------------------------------------------------------------
def topo_order(outputs, stop=None):
    """Nodes reachable from `outputs` in dependency order (iterative DFS); nodes whose id is in `stop` are listed
    but not descended into."""
    seen, order = set(), []
    stop = stop or ()
    for root in outputs:
        if root.id in seen:
            continue
        stack = [(root, 0)]
        while stack:
            n, st = stack.pop()
            if st == 0:
                if n.id in seen:
                    continue
                seen.add(n.id)
                stack.append((n, 1))
                if n.id in stop:
                    continue
                if n.b is not None and n.b.id not in seen:
                    stack.append((n.b, 0))
                if n.a is not None and n.a.id not in seen:
                    stack.append((n.a, 0))
            else:
                order.append(n)
    return order


def gradient(out, wrt):
    """Reverse-mode AD of scalar node `out` w.r.t. the list of sym nodes `wrt`; returns list of nodes."""
    order = topo_order([out])
    adj = {out.id: ONE}
    for n in reversed(order):
        g = adj.get(n.id)
        if g is None or n.op in ("const", "sym"):
            continue
        a, b, op = n.a, n.b, n.op

        def acc(t, v):
            if v is ZERO:
                return
            cur = adj.get(t.id)
            adj[t.id] = v if cur is None else add(cur, v)
        if op == "add":
            acc(a, g); acc(b, g)
        elif op == "sub":
            acc(a, g); acc(b, neg(g))
        elif op == "mul":
            acc(a, mul(g, b)); acc(b, mul(g, a))
        elif op == "div":
            acc(a, div(g, b)); acc(b, neg(div(mul(g, n), b)))       # d(a/b)/db = -(a/b)/b
        elif op == "neg":
            acc(a, neg(g))
        elif op == "sin":
            acc(a, mul(g, unary("cos", a)))
        elif op == "cos":
            acc(a, neg(mul(g, unary("sin", a))))
        elif op == "tan":
            acc(a, mul(g, add(ONE, mul(n, n))))
        elif op == "tanh":
            acc(a, mul(g, sub(ONE, mul(n, n))))
        elif op == "exp":
            acc(a, mul(g, n))
        elif op == "log":
            acc(a, div(g, a))
        elif op == "sqrt":
            acc(a, div(g, mul(const(2.0), n)))
        elif op == "pow":
            if b.op == "const":
                acc(a, mul(g, mul(b, powr(a, const(b.val - 1.0)))))
            else:
                acc(a, mul(g, mul(b, powr(a, sub(b, ONE)))))
                acc(b, mul(g, mul(n, unary("log", a))))
        else:
            raise NotImplementedError(op)
    return [adj.get(w.id, ZERO) for w in wrt]


def count_ops(outputs):
    return sum(1 for n in topo_order(outputs) if n.op not in ("const", "sym"))


# ------------------------------------------------------------------------------------------------
# matrix container with CasADi semantics
# ------------------------------------------------------------------------------------------------
_sym_counter = [0]


class SX:
    __array_ufunc__ = None
    __array_priority__ = 1000.0

    def __init__(self, data=None, shape=None):
        if data is None:
            self.data, self.shp = [], (0, 1)
            return
        if isinstance(data, SX):
            self.data, self.shp = list(data.data), data.shp
            return
        if isinstance(data, Node):
            self.data, self.shp = [data], (1, 1)
            return
        if isinstance(data, numbers.Real):
            self.data, self.shp = [const(data)], (1, 1)
            return
        if shape is not None:
            self.data, self.shp = list(data), tuple(shape)
            assert len(self.data) == shape[0] * shape[1]
            return
        a = np.asarray(data, dtype=object)
        if a.ndim == 0:
            self.data, self.shp = [as_node(a.item())], (1, 1)
        elif a.ndim == 1:
            self.data, self.shp = [as_node(v) for v in a], (a.shape[0], 1)
        else:
            self.data = [as_node(a[i, j]) for j in range(a.shape[1]) for i in range(a.shape[0])]
            self.shp = (a.shape[0], a.shape[1])

    # ---- construction
    @staticmethod
    def sym(name, r=1, c=1):
        _sym_counter[0] += 1
        if r == 1 and c == 1:
            return SX([sym(name)], (1, 1))
        return SX([sym("%s_%d" % (name, k)) for k in range(r * c)], (r, c))

    @staticmethod
    def zeros(r, c=1):
        return SX([ZERO] * (r * c), (r, c))

    @staticmethod
    def eye(n):
        return SX([ONE if i == j else ZERO for j in range(n) for i in range(n)], (n, n))

    # ---- shape
    @property
    def shape(self):
        return self.shp

    def numel(self):
        return self.shp[0] * self.shp[1]

    def size1(self):
        return self.shp[0]

    def size2(self):
        return self.shp[1]

    def rows(self):
        return self.shp[0]

    def columns(self):
        return self.shp[1]

    def is_scalar(self):
        return self.numel() == 1

    def at(self, i, j):
        return self.data[i + j * self.shp[0]]

    @property
    def T(self):
        r, c = self.shp
        return SX([self.data[i + j * r] for i in range(r) for j in range(c)], (c, r))

    def reshape(self, shp, c=None):
        if c is not None:
            shp = (shp, c)
        r, c = shp
        n = self.numel()
        if r == -1:
            r = n // c
        if c == -1:
            c = n // r
        assert r * c == n
        return SX(self.data, (r, c))

    def nz(self):
        return list(self.data)

    def __len__(self):
        return self.numel()

    def __iter__(self):
        for d in self.data:
            yield SX(d)

    def __getitem__(self, k):
        r, c = self.shp
        if isinstance(k, tuple):
            ri = range(r)[k[0]] if isinstance(k[0], slice) else [range(r)[k[0]]]
            ci = range(c)[k[1]] if isinstance(k[1], slice) else [range(c)[k[1]]]
            return SX([self.data[i + j * r] for j in ci for i in ri], (len(ri), len(ci)))
        if isinstance(k, slice):
            d = self.data[k]
            return SX(d, (len(d), 1))
        return SX([self.data[range(len(self.data))[k]]], (1, 1))

    # ---- arithmetic: element-wise with scalar broadcasting (CasADi `*` is element-wise)
    def _bin(self, o, f, swap=False):
        o = o if isinstance(o, SX) else SX(o)
        a, b = (o, self) if swap else (self, o)
        if a.shp == b.shp:
            return SX([f(x, y) for x, y in zip(a.data, b.data)], a.shp)
        if a.numel() == 1:
            return SX([f(a.data[0], y) for y in b.data], b.shp)
        if b.numel() == 1:
            return SX([f(x, b.data[0]) for x in a.data], a.shp)
        if a.numel() == b.numel() and 1 in a.shp and 1 in b.shp:          # row vs column vector of equal length
            return SX([f(x, y) for x, y in zip(a.data, b.data)], a.shp)
        raise ValueError("dimension mismatch %s vs %s" % (a.shp, b.shp))

    def __add__(self, o): return self._bin(o, add)
    def __radd__(self, o): return self._bin(o, add, True)
    def __sub__(self, o): return self._bin(o, sub)
    def __rsub__(self, o): return self._bin(o, sub, True)
    def __mul__(self, o): return self._bin(o, mul)
    def __rmul__(self, o): return self._bin(o, mul, True)
    def __truediv__(self, o): return self._bin(o, div)
    def __rtruediv__(self, o): return self._bin(o, div, True)
    def __pow__(self, o): return self._bin(o, powr)
    def __rpow__(self, o): return self._bin(o, powr, True)
    def __neg__(self): return SX([neg(x) for x in self.data], self.shp)
    def __pos__(self): return self
    def __matmul__(self, o): return mtimes(self, o)

    def __float__(self):
        assert self.numel() == 1 and self.data[0].op == "const"
        return self.data[0].val

    def __repr__(self):
        if self.numel() == 1:
            return "SX(%r)" % (self.data[0],)
        return "SX(%dx%d)" % self.shp


MX = SX   # the reference uses MX only to build the IPOPT NLP (PDP.py:138-166); same expression type here


def _lift(v):
    return v if isinstance(v, SX) else SX(v)


def vertcat(*args):
    items = [_lift(a) for a in args]
    items = [a for a in items if a.numel() > 0]
    if not items:
        return SX()
    c = items[0].shp[1]
    assert all(a.shp[1] == c for a in items), "vertcat: column mismatch"
    rows = sum(a.shp[0] for a in items)
    data = []
    for j in range(c):
        for a in items:
            data.extend(a.data[j * a.shp[0]:(j + 1) * a.shp[0]])
    return SX(data, (rows, c))


def horzcat(*args):
    items = [_lift(a) for a in args]
    items = [a for a in items if a.numel() > 0]
    if not items:
        return SX()
    r = items[0].shp[0]
    assert all(a.shp[0] == r for a in items), "horzcat: row mismatch"
    data = []
    for a in items:
        data.extend(a.data)
    return SX(data, (r, sum(a.shp[1] for a in items)))


def vcat(lst): return vertcat(*lst)
def hcat(lst): return horzcat(*lst)
def veccat(*args): return vertcat(*[_lift(a).reshape((-1, 1)) for a in args])


def mtimes(a, b, *more):
    a, b = _lift(a), _lift(b)
    if a.numel() == 1 or b.numel() == 1:
        out = a * b
    else:
        assert a.shp[1] == b.shp[0], "mtimes: inner dimension mismatch %s %s" % (a.shp, b.shp)
        r, k, c = a.shp[0], a.shp[1], b.shp[1]
        data = []
        for j in range(c):
            for i in range(r):
                s = ZERO
                for l in range(k):
                    s = add(s, mul(a.data[i + l * r], b.data[l + j * k]))
                data.append(s)
        out = SX(data, (r, c))
    for m in more:
        out = mtimes(out, m)
    return out


def transpose(a): return _lift(a).T


def dot(a, b):
    a, b = _lift(a), _lift(b)
    assert a.numel() == b.numel()
    s = ZERO
    for x, y in zip(a.data, b.data):
        s = add(s, mul(x, y))
    return SX(s)


def sumsqr(a):
    return dot(a, a)


def trace(a):
    a = _lift(a)
    s = ZERO
    for i in range(a.shp[0]):
        s = add(s, a.at(i, i))
    return SX(s)


def diag(a):
    a = _lift(a)
    if 1 in a.shp:
        n = a.numel()
        return SX([a.data[i] if i == j else ZERO for j in range(n) for i in range(n)], (n, n))
    return SX([a.at(i, i) for i in range(a.shp[0])], (a.shp[0], 1))


def inv(a):
    """Symbolic inverse by Gauss-Jordan without pivoting on structurally non-zero pivots (small matrices:
    JinEnv uses it for a diagonal 3x3 inertia and the 2x2 robot-arm mass matrix)."""
    a = _lift(a)
    n = a.shp[0]
    assert a.shp == (n, n)
    A = [[a.at(i, j) for j in range(n)] for i in range(n)]
    B = [[ONE if i == j else ZERO for j in range(n)] for i in range(n)]
    for k in range(n):
        piv = k
        while piv < n and A[piv][k] is ZERO:
            piv += 1
        assert piv < n, "inv: structurally singular"
        A[k], A[piv] = A[piv], A[k]
        B[k], B[piv] = B[piv], B[k]
        d = A[k][k]
        A[k] = [div(v, d) for v in A[k]]
        B[k] = [div(v, d) for v in B[k]]
        for i in range(n):
            if i != k and A[i][k] is not ZERO:
                f = A[i][k]
                A[i] = [sub(x, mul(f, y)) for x, y in zip(A[i], A[k])]
                B[i] = [sub(x, mul(f, y)) for x, y in zip(B[i], B[k])]
    return SX([B[i][j] for j in range(n) for i in range(n)], (n, n))


def _ew(op, npf):
    def f(a):
        if isinstance(a, SX):
            return SX([unary(op, x) for x in a.data], a.shp)
        if isinstance(a, Node):
            return unary(op, a)
        return npf(a)
    f.__name__ = op
    return f


sin, cos, tan, tanh = _ew("sin", np.sin), _ew("cos", np.cos), _ew("tan", np.tan), _ew("tanh", np.tanh)
exp, log, sqrt = _ew("exp", np.exp), _ew("log", np.log), _ew("sqrt", np.sqrt)


def power(a, b): return _lift(a) ** b
def norm_2(a): return sqrt(sumsqr(a))


def jacobian(expr, var):
    """d vec(expr) / d vec(var) as an (numel(expr) x numel(var)) SX (casadi.jacobian)."""
    e, v = _lift(expr), _lift(var)
    for w in v.data:
        assert w.op == "sym", "jacobian: second argument must be purely symbolic"
    rows = [gradient(o, v.data) for o in e.data]
    ne, nv = e.numel(), v.numel()
    return SX([rows[i][j] for j in range(nv) for i in range(ne)], (ne, nv))


def gradient_sx(expr, var):
    return jacobian(expr, var).T


def hessian(expr, var):
    g = jacobian(expr, var).T
    return jacobian(g, var), g


def symvar(expr):
    e = _lift(expr)
    return [n for n in topo_order(e.data) if n.op == "sym"]


def substitute(expr, old, new):
    """Replace sym nodes `old` by expressions `new` (both SX of equal numel)."""
    e, old, new = _lift(expr), _lift(old), _lift(new)
    m = {o.id: n for o, n in zip(old.data, new.data)}
    memo = {}
    for n in topo_order(e.data):
        if n.op == "sym":
            memo[n.id] = m.get(n.id, n)
        elif n.op == "const":
            memo[n.id] = n
        elif n.op in _UNARY:
            memo[n.id] = unary(n.op, memo[n.a.id])
        else:
            memo[n.id] = {"add": add, "sub": sub, "mul": mul, "div": div, "pow": powr}[n.op](memo[n.a.id], memo[n.b.id])
    return SX([memo[d.id] for d in e.data], e.shp)


# ------------------------------------------------------------------------------------------------
# code emission
# ------------------------------------------------------------------------------------------------
_CFUN = {"sin": "sin", "cos": "cos", "tan": "tan", "tanh": "tanh", "exp": "exp", "log": "log", "sqrt": "sqrt"}


def _cnum(v):
    if v != v or v in (float("inf"), float("-inf")):
        raise ValueError("non-finite constant in expression")
    r = repr(float(v))
    if "e" not in r and "." not in r:
        r += ".0"
    return r


def emit(outputs, inputs, lang="c", tmp="w", result=lambda k, e: "out[%d] = %s;" % (k, e), skip_zero=False, indent="    ", replace=None):
    """Straight-line code computing `outputs` (list of Nodes) from `inputs` ({sym node id: source string}).
    lang: 'c' (C / HIP device code, doubles) or 'py' (Python; functions prefixed by `_m.`).
    replace: {node id: source string} - sub-expressions available precomputed (not re-emitted, not descended into).
    Returns list of source lines."""
    lines, name = [], {}
    use = {}
    replace = replace or {}
    order = topo_order(outputs, stop=replace)
    for n in order:
        for ch in (n.a, n.b):
            if ch is not None:
                use[ch.id] = use.get(ch.id, 0) + 1
    k = 0
    for n in order:
        if n.id in replace:
            name[n.id] = replace[n.id]
            continue
        if n.op == "const":
            name[n.id] = _cnum(n.val) if n.val >= 0 else "(%s)" % _cnum(n.val)
            continue
        if n.op == "sym":
            if n.id not in inputs:
                raise KeyError("free symbol %s is not an input of the generated function" % n.name)
            name[n.id] = inputs[n.id]
            continue
        a = name[n.a.id]
        b = name[n.b.id] if n.b is not None else None
        if n.op == "add":
            e = "%s + %s" % (a, b)
        elif n.op == "sub":
            e = "%s - %s" % (a, b)
        elif n.op == "mul":
            e = "%s * %s" % (a, b)
        elif n.op == "div":
            e = "%s / %s" % (a, b)
        elif n.op == "neg":
            e = "-%s" % a
        elif n.op == "pow":
            e = ("pow(%s, %s)" if lang == "c" else "_m.pow(%s, %s)") % (a, b)
        else:
            e = ("%s(%s)" if lang == "c" else "_m.%s(%%s)" % n.op) % ((_CFUN[n.op], a) if lang == "c" else (a,))
        v = "%s%d" % (tmp, k)
        k += 1
        lines.append("%s%s%s = %s;" % (indent, "const double " if lang == "c" else "", v, e) if lang == "c" else "%s%s = %s" % (indent, v, e))
        name[n.id] = v
    for i, o in enumerate(outputs):
        if skip_zero and o is ZERO:
            continue
        r = result(i, name[o.id])
        if r:
            lines.append(indent + r)
    return lines


# ------------------------------------------------------------------------------------------------
# DM / Function (host-side numeric evaluation, casadi.Function call semantics)
# ------------------------------------------------------------------------------------------------
class DM:
    """Dense numeric matrix returned by Function calls (`.full()` -> 2-D ndarray)."""
    __array_priority__ = 2000.0

    def __init__(self, a):
        a = np.asarray(a, dtype=float)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        self.a = a

    def full(self): return self.a
    def toarray(self): return self.a
    def __array__(self, dtype=None, copy=None): return self.a if dtype is None else self.a.astype(dtype)
    def __float__(self): return float(self.a.reshape(-1)[0])
    @property
    def shape(self): return self.a.shape
    def __getitem__(self, k): return DM(self.a[k])
    def __add__(self, o): return DM(self.a + np.asarray(o, float))
    __radd__ = __add__
    def __sub__(self, o): return DM(self.a - np.asarray(o, float))
    def __rsub__(self, o): return DM(np.asarray(o, float) - self.a)
    def __mul__(self, o): return DM(self.a * np.asarray(o, float))
    __rmul__ = __mul__
    def __neg__(self): return DM(-self.a)
    def __repr__(self): return "DM(%r)" % (self.a,)


class _PyMath:
    sin, cos, tan, tanh, exp, log, sqrt, pow = math.sin, math.cos, math.tan, math.tanh, math.exp, math.log, math.sqrt, math.pow


class _NpMath:
    sin, cos, tan, tanh, exp, log, sqrt, pow = np.sin, np.cos, np.tan, np.tanh, np.exp, np.log, np.sqrt, np.power


class Function:
    """casadi.Function(name, [inputs], [outputs]) for purely symbolic inputs; numeric call returns DM
    (or a tuple of DM).  Scalars are broadcast to the declared input size like CasADi does."""

    def __init__(self, name, ins, outs, *unused, **unused_kw):
        self.name = name
        self.ins = [_lift(i) for i in ins]
        self.outs = [_lift(o) for o in outs]
        for i in self.ins:
            for n in i.data:
                assert n.op == "sym", "Function inputs must be symbolic"
        self._fn = {}

    def _compile(self, vec):
        inputs = {}
        for ai, i in enumerate(self.ins):
            for k, n in enumerate(i.data):
                inputs[n.id] = "a%d[%d]" % (ai, k)
        flat = [n for o in self.outs for n in o.data]
        body = emit(flat, inputs, lang="py", result=lambda k, e: "out[%d] = %s" % (k, e))
        src = "def _f(%s, out, _m):\n" % ", ".join("a%d" % k for k in range(len(self.ins))) + ("\n".join(body) if body else "    pass") + "\n"
        env = {}
        exec(compile(src, "<sx.Function %s>" % self.name, "exec"), env)
        return env["_f"]

    def __call__(self, *vals):
        assert len(vals) == len(self.ins), "%s: expected %d arguments" % (self.name, len(self.ins))
        if any(isinstance(v, SX) for v in vals):                      # symbolic call = substitution
            outs = self.outs
            for i, v in zip(self.ins, vals):
                v = _lift(v)
                if v.numel() == 1 and i.numel() > 1:
                    v = SX([v.data[0]] * i.numel(), i.shp)
                outs = [substitute(o, i, v) for o in outs]
            return outs[0] if len(outs) == 1 else tuple(outs)
        args = []
        for v, i in zip(vals, self.ins):
            if isinstance(v, DM):
                v = v.a
            a = np.asarray(v, dtype=float)
            a = a.reshape(-1, order="F") if a.ndim == 2 else a.reshape(-1)
            if a.size == 1 and i.numel() != 1:
                a = np.full(i.numel(), a[0])
            assert a.size == i.numel(), "%s: argument has %d elements, expected %d" % (self.name, a.size, i.numel())
            args.append(a.tolist())
        if "s" not in self._fn:
            self._fn["s"] = self._compile(False)
        n_out = sum(o.numel() for o in self.outs)
        out = [0.0] * n_out
        self._fn["s"](*args, out, _PyMath)
        res, k = [], 0
        for o in self.outs:
            res.append(DM(np.array(out[k:k + o.numel()], dtype=float).reshape(o.shp, order="F")))
            k += o.numel()
        return res[0] if len(res) == 1 else tuple(res)

    def map_batch(self, *arrays):
        """Vectorised evaluation: every argument is an array [B, numel]; returns list of [B, rows, cols]."""
        if "v" not in self._fn:
            self._fn["v"] = self._compile(True)
        B = max(np.asarray(a).shape[0] for a in arrays)
        args = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, float), (B, i.numel())).T) for a, i in zip(arrays, self.ins)]
        n_out = sum(o.numel() for o in self.outs)
        out = [None] * n_out
        self._fn["v"](*args, out, _NpMath)
        res, k = [], 0
        for o in self.outs:
            cols = [np.broadcast_to(np.asarray(out[k + q], float), (B,)) for q in range(o.numel())]
            res.append(np.stack(cols, axis=1).reshape(B, o.shp[1], o.shp[0]).transpose(0, 2, 1) if cols else np.zeros((B,) + o.shp))
            k += o.numel()
        return res


def nlpsol(*a, **k):
    raise NotImplementedError("IPOPT is not part of this framework; OCSys.ocSolver uses the batched Newton-KKT solver instead")
