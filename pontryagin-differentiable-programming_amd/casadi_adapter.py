"""CasADi front-end adapter: a `casadi.Function` built from `SX` expressions -> this package's expression DAG (sx.py) -> code generation.

Every script of the reference states its model with real `casadi.SX` (PDP/PDP.py:23; JinEnv/JinEnv.py:19) and hands the expressions to
`OCSys.setDyn / setPathCost / setFinalCost` (PDP.py:96-119), where they are wrapped into `casadi.Function`s and differentiated (PDP.py:235-270).
A user who keeps such a model can hand it to `pdp_amd.PDP` unchanged: the adapter walks the Function's SX instruction tape - the documented
introspection interface `n_instructions / instruction_id / instruction_input / instruction_output / instruction_constant` (CasADi's example
"accessing_sx_algorithm": every instruction is `work[o] = op(work[i0], work[i1])`, an input `work[o] = arg[i0][i1]`, an output `res[o0][o1] = work[i0]`,
or a constant) - and replays it on sx.py nodes.  From there on CasADi is no longer needed: differentiation, code generation and compilation are this
package's own (BASELINE.json: "CasADi retained only for one-time symbolic Jacobian codegen" - here not even for that).

CasADi is absent from the build image, so the walk is written against the INTERFACE, not the module: anything that offers the methods below can be
converted - a live `casadi.Function`, or a `RecordedTape` (the same calls answered from a JSON dump: tests/golden/casadi_tape_*.json).  `record(fn)` writes
such a dump from a live Function; `tape_of(sx_function)` writes one from this package's own Function in the same format (how the fixture was produced,
and a way to hand a model back to CasADi tooling).  Operation codes are matched by NAME (`casadi.OP_ADD` ...; a dump carries its own code -> name table),
never by number, so a renumbering between CasADi versions cannot silently change the meaning of a tape."""
import json

from . import sx

# operation names of casadi/core/calculus.hpp (enum Operation) that have a counterpart on sx.py nodes; n = number of work-vector operands
_ONE, _TWO = sx.const(1.0), sx.const(2.0)
_E = lambda f: (lambda a: sx.unary(f, a))
OPS = {
    "OP_ASSIGN": (1, lambda a: a),
    "OP_ADD": (2, sx.add), "OP_SUB": (2, sx.sub), "OP_MUL": (2, sx.mul), "OP_DIV": (2, sx.div),
    "OP_NEG": (1, sx.neg), "OP_EXP": (1, _E("exp")), "OP_LOG": (1, _E("log")), "OP_POW": (2, sx.powr), "OP_CONSTPOW": (2, sx.powr),
    "OP_SQRT": (1, _E("sqrt")), "OP_SQ": (1, lambda a: sx.mul(a, a)), "OP_TWICE": (1, lambda a: sx.mul(_TWO, a)),
    "OP_SIN": (1, _E("sin")), "OP_COS": (1, _E("cos")), "OP_TAN": (1, _E("tan")), "OP_TANH": (1, _E("tanh")),
    "OP_INV": (1, lambda a: sx.div(_ONE, a)),
    "OP_SINH": (1, lambda a: sx.mul(sx.const(0.5), sx.sub(sx.unary("exp", a), sx.unary("exp", sx.neg(a))))),
    "OP_COSH": (1, lambda a: sx.mul(sx.const(0.5), sx.add(sx.unary("exp", a), sx.unary("exp", sx.neg(a))))),
}
SPECIAL = ("OP_CONST", "OP_INPUT", "OP_OUTPUT")
# how sx.py's node kinds are written on a tape (tape_of)
_TO_CASADI = {"add": "OP_ADD", "sub": "OP_SUB", "mul": "OP_MUL", "div": "OP_DIV", "pow": "OP_POW", "neg": "OP_NEG", "sin": "OP_SIN", "cos": "OP_COS",
              "tan": "OP_TAN", "tanh": "OP_TANH", "exp": "OP_EXP", "log": "OP_LOG", "sqrt": "OP_SQRT"}


def op_names(module=None):
    """{operation code: name} of a CasADi module (its OP_* integer constants); the live `casadi` when none is given"""
    if module is None:
        import casadi as module
    return {int(getattr(module, n)): n for n in dir(module) if n.startswith("OP_") and isinstance(getattr(module, n), int)}


class RecordedTape:
    """A casadi.Function's introspection interface answered from a dict / JSON dump (see record, tape_of)."""

    def __init__(self, d):
        self.d = json.load(open(d)) if isinstance(d, str) else d
        self.op_names = {int(k): v for k, v in self.d["op_names"].items()}

    def name(self): return self.d["name"]
    def n_in(self): return len(self.d["in"])
    def n_out(self): return len(self.d["out"])
    def name_in(self, i): return self.d["in"][i]["name"]
    def name_out(self, i): return self.d["out"][i]["name"]
    def size_in(self, i): return tuple(self.d["in"][i]["size"])
    def size_out(self, i): return tuple(self.d["out"][i]["size"])
    def nnz_in(self, i): return self.d["in"][i]["size"][0] * self.d["in"][i]["size"][1]
    def nnz_out(self, i): return len(self.d["out"][i]["row"])
    def sz_w(self): return self.d["sz_w"]
    def n_instructions(self): return len(self.d["instructions"])
    def instruction_id(self, k): return self.d["instructions"][k][0]
    def instruction_input(self, k): return self.d["instructions"][k][1]
    def instruction_output(self, k): return self.d["instructions"][k][2]
    def instruction_constant(self, k): return self.d["instructions"][k][3]
    def out_pattern(self, i): return self.d["out"][i]["row"], self.d["out"][i]["colind"]


def _out_pattern(fn, i):
    """(row of every structural non-zero, column pointers) of output i - CasADi's compressed-column Sparsity"""
    if hasattr(fn, "out_pattern"):
        return fn.out_pattern(i)
    sp = fn.sparsity_out(i)
    return [int(r) for r in sp.row()], [int(c) for c in sp.colind()]


def record(fn, names=None):
    """A live casadi.Function (SX) as a JSON-able dict: everything from_casadi asks of it."""
    names = op_names() if names is None else names
    ins = []
    for i in range(fn.n_in()):
        assert fn.nnz_in(i) == fn.size1_in(i) * fn.size2_in(i), "input %d is sparse: state / control / parameter vectors are dense" % i
        ins.append({"name": fn.name_in(i), "size": [int(fn.size1_in(i)), int(fn.size2_in(i))]})
    outs = []
    for i in range(fn.n_out()):
        row, colind = _out_pattern(fn, i)
        outs.append({"name": fn.name_out(i), "size": [int(fn.size1_out(i)), int(fn.size2_out(i))], "row": row, "colind": colind})
    instr = []
    for k in range(fn.n_instructions()):
        op = int(fn.instruction_id(k))
        instr.append([op, [int(v) for v in fn.instruction_input(k)], [int(v) for v in fn.instruction_output(k)],
                      float(fn.instruction_constant(k)) if names.get(op) == "OP_CONST" else 0.0])
    used = sorted(set(e[0] for e in instr))
    return {"name": fn.name(), "in": ins, "out": outs, "sz_w": int(fn.sz_w()), "instructions": instr, "op_names": {str(c): names[c] for c in used}}


def tape_of(function, codes=None):
    """This package's sx.Function as an instruction tape in CasADi's format (dense outputs, column-major non-zeros, one work slot per node - no slot re-use;
    CasADi's own tapes re-use slots, which from_casadi handles: the work vector is replayed as mutable state).  codes: {name: code}; default: the numbering of
    CasADi 3.5 / 3.6 for the operations used here."""
    codes = codes or {"OP_ASSIGN": 0, "OP_ADD": 1, "OP_SUB": 2, "OP_MUL": 3, "OP_DIV": 4, "OP_NEG": 5, "OP_EXP": 6, "OP_LOG": 7, "OP_POW": 8, "OP_CONSTPOW": 9,
                      "OP_SQRT": 10, "OP_SQ": 11, "OP_TWICE": 12, "OP_SIN": 13, "OP_COS": 14, "OP_TAN": 15, "OP_TANH": 39, "OP_CONST": 44, "OP_INPUT": 45, "OP_OUTPUT": 46}
    slot, instr = {}, []
    for ai, a in enumerate(function.ins):
        for k, n in enumerate(a.data):
            slot[n.id] = len(slot)
            instr.append([codes["OP_INPUT"], [ai, k], [slot[n.id]], 0.0])
    flat = [n for o in function.outs for n in o.data]
    for n in sx.topo_order(flat):
        if n.id in slot:
            continue
        assert n.op != "sym", "free symbol %s: every symbol must be a Function input" % n.name
        slot[n.id] = len(slot)
        if n.op == "const":
            instr.append([codes["OP_CONST"], [], [slot[n.id]], float(n.val)])
        else:
            instr.append([codes[_TO_CASADI[n.op]], [slot[n.a.id]] + ([slot[n.b.id]] if n.b is not None else []), [slot[n.id]], 0.0])
    outs = []
    for oi, o in enumerate(function.outs):
        r, c = o.shp
        for k, n in enumerate(o.data):
            instr.append([codes["OP_OUTPUT"], [slot[n.id]], [oi, k], 0.0])
        outs.append({"name": "o%d" % oi, "size": [r, c], "row": [k % r for k in range(r * c)], "colind": [j * r for j in range(c + 1)]})
    used = sorted(set(e[0] for e in instr))
    inv = {v: k for k, v in codes.items()}
    return {"name": function.name, "in": [{"name": "i%d" % ai, "size": list(a.shp)} for ai, a in enumerate(function.ins)], "out": outs, "sz_w": len(slot),
            "instructions": instr, "op_names": {str(c): inv[c] for c in used}}


def from_casadi(fn, inputs=None, names=None):
    """casadi.Function (or RecordedTape) -> sx.Function over this package's own symbols.
    inputs: optional list of sx.SX (dense, sizes of the Function's inputs) to express the outputs in - e.g. the state / control / auxvar symbols an OCSys already
    holds; default: fresh symbols named after the Function's inputs.  names: {operation code: name}; default: the tape's own table, else the live casadi module's."""
    if names is None:
        names = getattr(fn, "op_names", None)
        names = op_names() if names is None else names
    nin = fn.n_in()
    if inputs is None:
        inputs = [sx.SX.sym(fn.name_in(i), *[int(v) for v in fn.size_in(i)]) for i in range(nin)]
    assert len(inputs) == nin, "%s has %d inputs" % (fn.name(), nin)
    ins = [sx._lift(a) for a in inputs]
    for i, a in enumerate(ins):
        assert a.numel() == fn.nnz_in(i), "input %d of %s: %d elements given, %d expected (inputs must be dense)" % (i, fn.name(), a.numel(), fn.nnz_in(i))
    work = [None] * int(fn.sz_w())
    res = [dict() for _ in range(fn.n_out())]
    for k in range(fn.n_instructions()):
        name = names.get(int(fn.instruction_id(k)))
        i, o = list(fn.instruction_input(k)), list(fn.instruction_output(k))
        if name == "OP_CONST":
            work[o[0]] = sx.const(float(fn.instruction_constant(k)))
        elif name == "OP_INPUT":
            work[o[0]] = ins[i[0]].data[i[1]]
        elif name == "OP_OUTPUT":
            res[o[0]][o[1]] = work[i[0]]
        elif name in OPS:
            n, f = OPS[name]
            assert len(i) >= n and all(work[j] is not None for j in i[:n]), "instruction %d (%s) reads an unset work slot" % (k, name)
            work[o[0]] = f(*[work[j] for j in i[:n]])
        else:
            raise NotImplementedError("CasADi operation %s (code %d, instruction %d of %s) has no counterpart in pdp_amd.sx (supported: %s)"
                                      % (name, int(fn.instruction_id(k)), k, fn.name(), ", ".join(sorted(OPS))))
    outs = []
    for oi in range(fn.n_out()):
        r, c = (int(v) for v in fn.size_out(oi))
        row, colind = _out_pattern(fn, oi)
        data = [sx.ZERO] * (r * c)                              # structural zeros of a sparse output stay exact zeros (code generation drops them)
        for j in range(c):
            for q in range(colind[j], colind[j + 1]):
                data[j * r + row[q]] = res[oi][q]
        outs.append(sx.SX(data, (r, c)))
    return sx.Function(fn.name(), ins, outs)


def is_casadi(expr):
    """a CasADi symbolic object (SX / MX / DM), recognised without importing casadi"""
    return type(expr).__module__.split(".")[0] == "casadi"


def convert_expression(expr, casadi_vars, own_vars, name="expr"):
    """A casadi.SX expression in the casadi symbols `casadi_vars` (list of SX vectors) -> the same expression on `own_vars` (list of sx.SX of the same sizes):
    wraps it into a casadi.Function - as OCSys.setDyn does in the reference (PDP.py:101) - and converts that.  Needs the live casadi module."""
    import casadi
    fn = casadi.Function(name, list(casadi_vars), [expr])
    return from_casadi(fn, inputs=own_vars).outs[0]
