"""One-time code generation: symbolic problem -> HIP device code (a `PdpModel` struct) -> libpdp_model_<name>.so.

This replaces the reference's setup-time CasADi work: `OCSys.diffPMP` (PDP/PDP.py:222-270) builds 14
casadi.Function objects that `getAuxSys` (272-314) then calls 9T+2 times per trajectory through SWIG; here the
same derivatives are differentiated once (sx.py), emitted as straight-line device code with shared
sub-expressions, and compiled by hipcc for gfx950 into the batched kernels of csrc/pdp_model_kernels.h.

Generated interface (consumed by csrc/pdp_model_kernels.h):
  struct PdpModel { KIND, NX, NU, NP, NAME, CHUNK;
      dyn(x,u,th,out)  path_cost(x,u,th)  final_cost(x,th)  costate_step(x,u,lam,th,out)  dhx(x,th,out)
      group <g> in {path, fwd, fin}:  <G>_NVAR, <G>_NCONST, <g>_const(c), eval_<g>(x,u,lam,th,sink),
                                      <g>_code(mat, i), <G>_MAT[k], <G>_OFF[k], <G>_ROWS[mat], <G>_COLS[mat] }
Matrix entries are classified at generation time: structural zero (code -1), constant (code -2-c, value in
the constant pool) or variable (code k >= 0, written by eval_<g> through sink.put<k>(v)).  Only variable
entries cost arithmetic and LDS space; the quadrotor's eight per-step matrices have ~200 of 728 entries non-zero.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

from . import sx

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
GEN_DIR = os.path.join(CSRC, "generated")
LIB_DIR = os.path.join(HERE, "lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: let the fp64 MFMA accumulators live in VGPRs (the kernels sit at the 256-VGPR ceiling; with AGPR accumulators every
# VALU/LDS use of a product costs v_accvgpr moves - measured +4% on the headline kernel)
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form"] + os.environ.get("PDP_HIP_EXTRA_FLAGS", "").split()
# The core library (16 instantiations of lqr_solve_kernel at 256 VGPRs + up to 180 AGPRs) is built WITHOUT -amdgpu-mfma-vgpr-form: with that
# (hidden, experimental) LLVM option a variant of lqr_solve_kernel<2,3> that streams one more operand is miscompiled at -O3 - deterministic
# out-of-bounds global reads on guard-banded operands, gone at -O1 and gone without the option (probes/lqr_oob_probe.py, DESIGN.md section 8;
# profiles/r02_lqr_oob_root_cause.txt).  The kernel is HBM-bound: the option bought nothing there.
CORE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + os.environ.get("PDP_HIP_EXTRA_FLAGS", "").split()
# Exact generated names (label + kind + content hash of the generated code) of the models that may be built with HIP_FLAGS: zoo.py registers
# the reference's benchmark systems here.  Keyed on the HASH, not on the label: a user's own "quadrotor" with other dynamics, sizes or
# horizon has another hash and is built with plain -O3 (round-2 advisor finding).  PDP_MFMA_VGPR_FORM=0 builds everything without the option,
# =1 everything with it (tests/test_gpu_flag_fence.py compares the two builds of the headline models bit for bit).
# The set is read at import from csrc/generated/tuned_names.txt (tracked; written by zoo.register_tuned, held current by tests/test_abi_and_host.py), so that
# tuned(name) answers the same whatever ran before in the process - a set filled lazily made the flags of a zoo-identical user model depend on call order.
TUNED_FILE = os.path.join(GEN_DIR, "tuned_names.txt")


def _read_tuned():
    try:
        with open(TUNED_FILE) as f:
            return set(ln.strip() for ln in f if ln.strip() and not ln.startswith("#"))
    except OSError:
        return set()


TUNED_NAMES = _read_tuned()
# OC models (the fused kernel): loop strength reduction rewrites the running LDS addresses of the step loops as (induction variable + 0)
# and leaves a `v_add_u32 v, 0, v` in front of every ds_read (39 VALU instructions per time step); the loops already carry their own
# running addresses.  Measured: fused kernel +5 % without LSR; the SysID / ControlPlanning kernels lose up to 12 % -> OC models only.
OC_EXTRA_FLAGS = ["-mllvm", "-disable-lsr"]

KIND_OC, KIND_CP, KIND_SYSID = 0, 1, 2
KIND_NAME = {0: "oc", 1: "cp", 2: "sysid"}

# matrix order inside each group (indices used by the kernels)
OC_PATH = ["F", "G", "E", "Hxx", "Hxu", "Hxe", "Huu", "Hue"]
OC_FWD = ["F", "G", "E"]
OC_FIN = ["hxx", "hxe"]
CP_PATH = ["F", "G", "cx", "cu"]
ID_PATH = ["F", "E"]


def _col(v):
    v = sx._lift(v)
    return v.reshape((-1, 1))


class Problem:
    """Symbolic statement of one PDP problem (what the reference passes to setDyn/setPathCost/setFinalCost)."""

    def __init__(self, kind, state, control, dyn, auxvar=None, path_cost=None, final_cost=None, label="model"):
        self.kind = kind
        self.x, self.u = _col(state), _col(control)
        self.th = _col(auxvar) if auxvar is not None else sx.SX()
        self.dyn = _col(dyn)
        self.path_cost = sx._lift(path_cost) if path_cost is not None else None
        self.final_cost = sx._lift(final_cost) if final_cost is not None else None
        self.label = label
        self.n, self.m, self.p = self.x.numel(), self.u.numel(), self.th.numel()
        assert self.dyn.numel() == self.n, "dynamics must have as many rows as the state"
        for s in list(self.x.data) + list(self.u.data) + list(self.th.data):
            assert s.op == "sym", "state / control / auxvar must be purely symbolic"


class _Group:
    def __init__(self, gname, mats, dedup=False):
        """mats: list of (name, rows, cols, SX).  Builds codes / pools.
        dedup: entries that are the SAME expression (the two halves of a symmetric Hessian, repeated Jacobian entries) share one
        variable slot - less arithmetic-free LDS traffic and a shorter pool row (quadrotor: 193 -> 120 entries per step).  Only for
        groups consumed through their codes (the OC kernels); the dense scatter of the CP / SysID kernels maps slot -> one element."""
        self.gname = gname
        self.names = [m[0] for m in mats]
        self.rows = [m[1] for m in mats]
        self.cols = [m[2] for m in mats]
        self.codes = []          # per matrix: list (row-major) of codes
        self.var_nodes = []      # k -> node
        self.var_mat, self.var_off = [], []
        self.consts = []         # c -> value
        cidx = {}
        slot = {}
        for mi, (name, r, c, M) in enumerate(mats):
            M = sx._lift(M)
            assert M.shape == (r, c), "%s: shape %s, expected %s" % (name, M.shape, (r, c))
            codes = []
            for i in range(r):
                for j in range(c):
                    nd = M.at(i, j)
                    if nd.op == "const":
                        if nd.val == 0.0:
                            codes.append(-1)
                        else:
                            if nd.val not in cidx:
                                cidx[nd.val] = len(self.consts)
                                self.consts.append(nd.val)
                            codes.append(-2 - cidx[nd.val])
                    elif dedup and nd.id in slot:
                        codes.append(slot[nd.id])
                    else:
                        slot[nd.id] = len(self.var_nodes)
                        codes.append(len(self.var_nodes))
                        self.var_nodes.append(nd)
                        self.var_mat.append(mi)
                        self.var_off.append(i * c + j)
            self.codes.append(codes)

    @property
    def nvar(self):
        return len(self.var_nodes)


class _DeviceOpt:
    """Device-oriented rewrite of the expression DAG before emission:
      * a / b with b depending only on the parameters theta (or constant) becomes a * (1/b): fp64 division costs ~74
        cycles on gfx950 against ~5 for a multiply, and the models divide by masses / inertias at every time step;
      * maximal sub-expressions that depend only on theta are hoisted into `precompute(th, pc)`, evaluated once per
        trajectory; the per-step functions read them from pc[].
    Results differ from the un-rewritten expressions by at most an ulp per rewritten division."""

    def __init__(self, theta_nodes):
        self.theta = set(n.id for n in theta_nodes)
        self.memo = {}           # old id -> new node
        self.inv = {}            # new id -> depends only on theta/consts
        self.pc_index = {}       # new id -> slot in pc[]
        self.pc_nodes = []

    def _isinv(self, n):
        return self.inv[n.id]

    def rewrite(self, outputs):
        B = {"add": sx.add, "sub": sx.sub, "mul": sx.mul, "pow": sx.powr}
        for n in sx.topo_order(outputs):
            if n.id in self.memo:
                continue
            if n.op == "const":
                m, iv = n, True
            elif n.op == "sym":
                m, iv = n, n.id in self.theta
            elif n.b is None:
                a = self.memo[n.a.id]
                m, iv = sx.unary(n.op, a), self.inv[a.id]
            else:
                a, b = self.memo[n.a.id], self.memo[n.b.id]
                iv = self.inv[a.id] and self.inv[b.id]
                if n.op == "div":
                    if self.inv[b.id] and not self.inv[a.id]:
                        r = sx.div(sx.ONE, b)
                        self.inv.setdefault(r.id, True)
                        m = sx.mul(a, r)
                    else:
                        m = sx.div(a, b)
                else:
                    m = B[n.op](a, b)
            self.memo[n.id] = m
            self.inv.setdefault(m.id, iv)
            # simplification may return an existing sub-node: make sure its children are classified too
            if m.id not in self.inv:
                self.inv[m.id] = iv
        return [self.memo[o.id] for o in outputs]

    def classify(self, outputs):
        """(re)compute invariance on the rewritten DAG"""
        for n in sx.topo_order(outputs):
            if n.op == "const":
                self.inv[n.id] = True
            elif n.op == "sym":
                self.inv[n.id] = n.id in self.theta
            else:
                self.inv[n.id] = self.inv[n.a.id] and (n.b is None or self.inv[n.b.id])

    def mark_hoisted(self, outputs):
        """Register the theta-only operator nodes that x/u/lambda-dependent code (or an output) uses directly."""
        self.classify(outputs)

        def want(n):
            return n.op not in ("const", "sym") and self.inv.get(n.id, False)
        for n in sx.topo_order(outputs, stop=self.pc_index):
            if self.inv.get(n.id, False):
                continue
            for ch in (n.a, n.b):
                if ch is not None and want(ch) and ch.id not in self.pc_index:
                    self.pc_index[ch.id] = len(self.pc_nodes)
                    self.pc_nodes.append(ch)
        for o in outputs:
            if want(o) and o.id not in self.pc_index:
                self.pc_index[o.id] = len(self.pc_nodes)
                self.pc_nodes.append(o)

    def replace_map(self):
        return {nid: "pc[%d]" % k for nid, k in self.pc_index.items()}


def _arr(vals, per_line=24):
    vals = list(vals)
    if not vals:
        return "0"
    out = []
    for i in range(0, len(vals), per_line):
        out.append(", ".join(str(v) for v in vals[i:i + per_line]))
    return ",\n            ".join(out)


def _emit_group(g, inputs, L, replace=None):
    G = g.gname.upper()
    L.append("    // ---- group '%s': %s" % (g.gname, ", ".join("%s[%dx%d]" % (n, r, c) for n, r, c in zip(g.names, g.rows, g.cols))))
    L.append("    static constexpr int %s_NMAT = %d, %s_NVAR = %d, %s_NCONST = %d;" % (G, len(g.names), G, g.nvar, G, len(g.consts)))
    L.append("    static constexpr short %s_ROWS[%d] = {%s};" % (G, len(g.names), _arr(g.rows)))
    L.append("    static constexpr short %s_COLS[%d] = {%s};" % (G, len(g.names), _arr(g.cols)))
    L.append("    static constexpr short %s_MAT[%d] = {%s};" % (G, max(1, g.nvar), _arr(g.var_mat)))
    L.append("    static constexpr short %s_OFF[%d] = {%s};" % (G, max(1, g.nvar), _arr(g.var_off)))
    flat, base = [], []
    for codes in g.codes:
        base.append(len(flat))
        flat.extend(codes)
    L.append("    PDP_HD static int %s_code(int mat, int i) {" % g.gname)
    L.append("        constexpr short tbl[] = {%s};" % _arr(flat))
    L.append("        constexpr short base[] = {%s};" % _arr(base))
    L.append("        return tbl[base[mat] + i];")
    L.append("    }")
    L.append("    PDP_HD static double %s_const(int c) {" % g.gname)
    L.append("        constexpr double tbl[] = {%s};" % (_arr([sx._cnum(v) for v in g.consts], 8) if g.consts else "0.0"))
    L.append("        return tbl[c];")
    L.append("    }")
    L.append("    template <class Sink>")
    L.append("    PDP_HD static void eval_%s(const double* x, const double* u, const double* lam, const double* th, const double* pc, Sink& s) {" % g.gname)
    L.append("        (void)x; (void)u; (void)lam; (void)th; (void)pc;")
    L.extend(sx.emit(g.var_nodes, inputs, lang="c", result=lambda k, e: "s.template put<%d>(%s);" % (k, e), indent="        ", replace=replace))
    L.append("    }")


def _emit_vecfn(name, args, outputs, inputs, L, scalar=False, replace=None):
    if scalar:
        L.append("    PDP_HD static double %s(%s) {" % (name, ", ".join("const double* %s" % a for a in args)))
        L.append("        " + " ".join("(void)%s;" % a for a in args))
        body = sx.emit(outputs, inputs, lang="c", result=lambda k, e: "return %s;" % e, indent="        ", replace=replace)
        L.extend(body)
        L.append("    }")
    else:
        L.append("    PDP_HD static void %s(%s, double* out) {" % (name, ", ".join("const double* %s" % a for a in args)))
        L.append("        " + " ".join("(void)%s;" % a for a in args))
        L.extend(sx.emit(outputs, inputs, lang="c", result=lambda k, e: "out[%d] = %s;" % (k, e), indent="        ", replace=replace))
        L.append("    }")


def _pick_chunk(nvar, extra, other_doubles=0):
    """Time steps per lane-parallel aux pass (lane = step, so at most 64); the LDS pool is chunk * (nvar+extra, odd) doubles.
    Fused OC kernel (`other_doubles` = its Riccati scratch, constants and parameters): the most steps that still let 4 workgroups
    (one wavefront per SIMD) share a CU's 160 KB, i.e. 40 KB per workgroup - the kernel splits a horizon into ceil(T / CHUNK)
    chunks of equal length, so any value helps, not just powers of two (quadrotor: 21 -> T = 50 runs as 3 chunks of 17).
    Other kernels also stage the trajectory in LDS and want several workgroups per SIMD: powers of two within 34 KB."""
    stride = (nvar + extra) | 1
    if other_doubles:
        return max(2, min(64, (40 * 1024 - 8 * other_doubles) // (stride * 8)))
    for c in (32, 16, 8, 4):
        if c * stride * 8 <= 34 * 1024:
            return c
    return 2


def generate(problem):
    """Returns (header_source, info dict).  The model name embeds a hash of the generated code."""
    pb = problem
    n, m, p = pb.n, pb.m, pb.p
    x, u, th = pb.x, pb.u, pb.th
    lam = sx.SX.sym("lam", n)
    inputs = {}
    for nm, v in (("x", x), ("u", u), ("th", th), ("lam", lam)):
        for k, nd in enumerate(v.data):
            inputs[nd.id] = "%s[%d]" % (nm, k)
    opt = _DeviceOpt(th.data)
    R = lambda M: sx.SX(opt.rewrite(sx._lift(M).data), sx._lift(M).shape)      # device-oriented rewrite of a matrix
    funcs = []        # (name, args, output nodes, scalar?)
    groups = {}
    dyn = pb.dyn
    fx, fu = sx.jacobian(dyn, x), sx.jacobian(dyn, u)
    funcs.append(("dyn", ["x", "u", "th"], R(dyn).data, False))
    if pb.kind == KIND_OC:
        c, h = pb.path_cost, pb.final_cost
        assert c is not None and h is not None and c.numel() == 1 and h.numel() == 1
        fe = sx.jacobian(dyn, th)
        H = c + sx.dot(dyn, lam)                                   # Hamiltonian, PDP.py:231
        dHx, dHu = sx.jacobian(H, x).T, sx.jacobian(H, u).T        # PDP.py:243-246
        dhx = sx.jacobian(h, x).T                                  # PDP.py:263
        funcs.append(("path_cost", ["x", "u", "th"], R(c).data, True))
        funcs.append(("final_cost", ["x", "th"], R(h).data, True))
        funcs.append(("costate_step", ["x", "u", "lam", "th"], R(dHx).data, False))     # c_x + f_x' lam  (PDP.py:205-209)
        funcs.append(("dhx", ["x", "th"], R(dhx).data, False))
        funcs.append(("dHu", ["x", "u", "lam", "th"], R(dHu).data, False))               # c_u + f_u' lam  (PMP stationarity residual)
        mats = {"F": fx, "G": fu, "E": fe, "Hxx": sx.jacobian(dHx, x), "Hxu": sx.jacobian(dHx, u), "Hxe": sx.jacobian(dHx, th),
                "Huu": sx.jacobian(dHu, u), "Hue": sx.jacobian(dHu, th), "hxx": sx.jacobian(dhx, x), "hxe": sx.jacobian(dhx, th)}
        mats = {k: R(v) for k, v in mats.items()}
        dims = {"F": (n, n), "G": (n, m), "E": (n, p), "Hxx": (n, n), "Hxu": (n, m), "Hxe": (n, p), "Huu": (m, m), "Hue": (m, p),
                "hxx": (n, n), "hxe": (n, p)}
        groups["path"] = _Group("path", [(k,) + dims[k] + (mats[k],) for k in OC_PATH], dedup=True)
        # the fused kernel evaluates the path matrices in two stages per chunk: lambda-independent (patha: F, G, E, c_x) first,
        # then - once the costates of the chunk have been propagated with F and c_x - the lambda-weighted Hessians (pathb)
        mats["cx"] = R(sx.jacobian(c, x).T)
        dims["cx"] = (n, 1)
        groups["patha"] = _Group("patha", [(k,) + dims[k] + (mats[k],) for k in ("F", "G", "E", "cx")], dedup=True)
        groups["pathb"] = _Group("pathb", [(k,) + dims[k] + (mats[k],) for k in ("Hxx", "Hxu", "Hxe", "Huu", "Hue")], dedup=True)
        groups["fwd"] = _Group("fwd", [(k,) + dims[k] + (mats[k],) for k in OC_FWD], dedup=True)
        groups["fin"] = _Group("fin", [(k,) + dims[k] + (mats[k],) for k in OC_FIN], dedup=True)
        # the multiple-shooting OC solver (oc_solve_ms_kernel) knows (x, u, lambda) of every stage before its sweep: one group with
        # the matrices of the KKT system for the backward pass, F and G alone for the forward pass
        groups["sol"] = _Group("sol", [(k,) + dims[k] + (mats[k],) for k in ("F", "G", "Hxx", "Hxu", "Huu")], dedup=True)
        groups["solf"] = _Group("solf", [(k,) + dims[k] + (mats[k],) for k in ("F", "G")], dedup=True)
        chunk = None                                          # needs the number of hoisted values: decided below
    elif pb.kind == KIND_CP:
        c, h = pb.path_cost, pb.final_cost
        assert p == 0, "ControlPlanning dynamics / costs carry no auxvar (PDP.py:672-697)"
        funcs.append(("path_cost", ["x", "u", "th"], R(c).data, True))
        funcs.append(("final_cost", ["x", "th"], R(h).data, True))
        funcs.append(("dhx", ["x", "th"], R(sx.jacobian(h, x).T).data, False))
        groups["path"] = _Group("path", [("F", n, n, R(fx)), ("G", n, m, R(fu)), ("cx", 1, n, R(sx.jacobian(c, x))), ("cu", 1, m, R(sx.jacobian(c, u)))])
        chunk = _pick_chunk(groups["path"].nvar, 0)
    elif pb.kind == KIND_SYSID:
        fe = sx.jacobian(dyn, th)
        groups["path"] = _Group("path", [("F", n, n, R(fx)), ("E", n, p, R(fe))])
        chunk = _pick_chunk(groups["path"].nvar, n)
    else:
        raise ValueError("unknown problem kind")
    for _, _, outs, _ in funcs:
        opt.mark_hoisted(outs)
    for g in groups.values():
        opt.mark_hoisted(g.var_nodes)
    repl = opt.replace_map()
    npc = len(opt.pc_nodes)
    if chunk is None:
        # fused kernel (csrc/pdp_model_kernels.h fused_lds_bytes): Riccati scratch 608 + constants + dl_T + theta + pc + pad
        nconst = 1 + max(len(groups["patha"].consts) + len(groups["pathb"].consts), len(groups["fwd"].consts), len(groups["fin"].consts))
        chunk = _pick_chunk(groups["patha"].nvar + groups["pathb"].nvar, n, other_doubles=608 + nconst + n + p + max(1, npc) + 8)
        # solver kernel: pool row = sol entries + defect (n) + Lagrangian gradients (n + m); its own constants, theta, pc
        nconst_s = 1 + max(len(groups["sol"].consts), len(groups["solf"].consts), len(groups["fin"].consts))
        ms_chunk = _pick_chunk(groups["sol"].nvar, 2 * n + m, other_doubles=608 + nconst_s + n + p + max(1, npc) + 16)
    else:
        ms_chunk = 0
    L = []
    L.append("    // ---- theta-only sub-expressions, evaluated once per trajectory (pc[NPC])")
    L.append("    static constexpr int NPC = %d;" % max(1, npc))
    L.append("    PDP_HD static void precompute(const double* th, double* pc) {")
    L.append("        (void)th; (void)pc;")
    L.extend(sx.emit(opt.pc_nodes, inputs, lang="c", result=lambda k, e: "pc[%d] = %s;" % (k, e), indent="        "))
    L.append("    }")
    for name, args, outs, scalar in funcs:
        _emit_vecfn(name, args + ["pc"], outs, inputs, L, scalar=scalar, replace=repl)
    for g in groups.values():
        _emit_group(g, inputs, L, replace=repl)
    body = "\n".join(L)
    digest = hashlib.sha1(("%d|%d|%d|%d|" % (pb.kind, n, m, p) + body).encode()).hexdigest()[:10]
    name = "%s_%s_%s" % (pb.label, KIND_NAME[pb.kind], digest)
    nops = sx.count_ops(groups["path"].var_nodes)
    head = [
        "// AUTO-GENERATED by pdp_amd.codegen (symbolic problem -> HIP device code).  Do not edit.",
        "// model %s : kind=%s n=%d m=%d p=%d ; theta-only precomputed values: %d" % (name, KIND_NAME[pb.kind], n, m, p, npc),
        "#pragma once",
        "#ifndef PDP_HD",
        "#define PDP_HD __host__ __device__ inline",
        "#endif",
        "struct PdpModel {",
        "    static constexpr int KIND = %d, NX = %d, NU = %d, NP = %d, CHUNK = %d, MS_CHUNK = %d;" % (pb.kind, n, m, p, chunk, ms_chunk),
        "    static constexpr const char* NAME = \"%s\";" % name,
    ]
    src = "\n".join(head) + "\n" + body + "\n};\n"
    info = dict(name=name, kind=pb.kind, n=n, m=m, p=p, chunk=chunk, npc=npc, nvar={k: g.nvar for k, g in groups.items()},
                nconst={k: len(g.consts) for k, g in groups.items()}, ops_path=nops)
    return src, info


# ------------------------------------------------------------------------------------------------------
# build
# ------------------------------------------------------------------------------------------------------
def header_path(name):
    return os.path.join(GEN_DIR, name + ".h")


def lib_path(name):
    return os.path.join(LIB_DIR, "libpdp_model_%s.so" % name)


def write_header(problem):
    src, info = generate(problem)
    os.makedirs(GEN_DIR, exist_ok=True)
    path = header_path(info["name"])
    if not os.path.exists(path) or open(path).read() != src:
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "w") as f:
            f.write(src)
        os.replace(tmp, path)
    return path, info


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout[-4000:]))
    return r.stdout


def _stamp_of(deps, flags):
    """content hash of everything a library is built from (sources, headers, flags): decides rebuilds, not file mtimes - the
    snapshot copied to a GPU box carries arbitrary mtimes, and 8 ranks must not all decide to rebuild because of them"""
    h = hashlib.sha1(" ".join(flags).encode())
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _build(out, deps, cmd_tail, force, flags=None):
    flags = HIP_FLAGS if flags is None else flags
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp_path = out + ".stamp"
    stamp = _stamp_of(deps, flags + cmd_tail[:-1])
    if not force and os.path.exists(out) and os.path.exists(stamp_path) and open(stamp_path).read().strip() == stamp:
        return out
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s: cannot build %s" % (HIPCC, out))
    tmp = "%s.%d.tmp" % (out, os.getpid())           # several ranks may build the same library at once: write aside, rename atomically
    _run([HIPCC] + flags + cmd_tail + ["-o", tmp])
    os.replace(tmp, out)
    with open(stamp_path + ".%d.tmp" % os.getpid(), "w") as f:
        f.write(stamp)
    os.replace(stamp_path + ".%d.tmp" % os.getpid(), stamp_path)
    return out


def source_closure(src):
    """`src` and every file of this repository it includes, transitively (`#include "..."` resolved against the including file's directory, csrc/ and include/;
    the generated model header comes in through a macro and is listed by the caller).  The content-hash stamp of a library is taken over exactly this list, so a
    header added to a translation unit can never be missing from it (round-5 advice: pdp_cp_generic_kernels.h was, and edits to it left stale model libraries)."""
    import re
    roots = [CSRC, os.path.join(os.path.dirname(HERE), "include")]
    seen, todo = [], [os.path.abspath(src)]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.append(f)
        with open(f) as fh:
            for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), flags=re.M):
                for d in [os.path.dirname(f)] + roots:
                    c = os.path.abspath(os.path.join(d, inc))
                    if os.path.exists(c):
                        todo.append(c)
                        break
    return [seen[0]] + sorted(seen[1:])


def kernel_sources_digest():
    """sha1 over every hand-written source the shipped kernels are built from (the include closures of both translation units; generated model headers excluded):
    recorded beside counter-derived figures (profiles/traffic.json) so that bench.py can tell whether they were collected on the kernels it is running."""
    files = sorted(set(source_closure(os.path.join(CSRC, "pdp_model.hip")) + source_closure(os.path.join(CSRC, "pdp_lqr.hip"))))
    h = hashlib.sha1()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def tuned(name):
    env = os.environ.get("PDP_MFMA_VGPR_FORM")
    if env is not None:
        return env not in ("0", "")
    return name in TUNED_NAMES


def compile_model(name, force=False, plain_twin=False):
    """hipcc csrc/pdp_model.hip with the generated header -> lib/libpdp_model_<name>.so (in-tree).  plain_twin: the same model built with
    CORE_FLAGS whatever tuned() says, as lib/libpdp_model_<name>__plain.so - the reference build tests/test_gpu_flag_fence.py compares with."""
    deps = [header_path(name)] + source_closure(os.path.join(CSRC, "pdp_model.hip"))
    extra = OC_EXTRA_FLAGS if ("_%s_" % KIND_NAME[KIND_OC]) in name else []
    # -amdgpu-mfma-vgpr-form (hidden LLVM option, +4 % on the headline kernel, but see CORE_FLAGS above) only for the exact benchmark models
    # (TUNED_NAMES), whose kernels are parity-tested one by one on NaN-dirtied memory and compared with their plain -O3 twins; a user's model
    # is built with plain -O3
    flags = HIP_FLAGS if (tuned(name) and not plain_twin) else CORE_FLAGS
    out = lib_path(name + "__plain") if plain_twin else lib_path(name)
    return _build(out, deps, extra + ["-DPDP_MODEL_HEADER=\"generated/%s.h\"" % name, "-I", CSRC, os.path.join(CSRC, "pdp_model.hip")], force, flags=flags)


CORE_LIB_PATH = os.path.join(LIB_DIR, "libpdp_hip.so")


def compile_core(force=False):
    deps = source_closure(os.path.join(CSRC, "pdp_lqr.hip"))
    return _build(os.path.join(LIB_DIR, "libpdp_hip.so"), deps, ["-I", CSRC, os.path.join(CSRC, "pdp_lqr.hip")], force, flags=CORE_FLAGS)


def kernel_resources(lib, count_scratch_instructions=False):
    """Register / scratch usage of every kernel in a built library, read from the gfx950 code object's metadata notes (llvm-objcopy ->
    clang-offload-bundler -> llvm-readelf --notes): {demangled-ish kernel name: dict(vgpr, agpr, sgpr, spill, scratch, lds)}.  Used by the no-spill
    test (tests/test_abi_and_host.py) and by bench.py, which prints the dominant kernel's figures into its roofline entry."""
    import re
    import tempfile
    llvm = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        _run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
        _run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = _run([os.path.join(llvm, "llvm-readelf"), "--notes", co])
        dis = _run([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn", co]) if count_scratch_instructions else ""
    nscr = {}
    if dis:                                                   # scratch_load / scratch_store instructions per kernel symbol
        cur = None
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                cur = m.group(1)
                nscr[cur] = 0
            elif cur is not None and "scratch_" in ln:
                nscr[cur] += 1
    out = {}
    for ent in notes.split("- .agpr_count:")[1:]:
        get = lambda k: int(re.search(r"\.%s:\s*(\d+)" % k, ent).group(1))
        sym = re.search(r"\.name:\s*(\S+)", ent).group(1)
        m = re.match(r"_ZN3pdp(\d+)", sym)                   # pdp::<name><template args...>
        name = sym[m.end():m.end() + int(m.group(1))] if m else sym
        targs = re.findall(r"L[ib](\d+)E", sym[m.end() + int(m.group(1)):].split("EE")[0] + "E") if m and "I" in sym[m.end() + int(m.group(1)):][:1] else []
        key = name + ("<%s>" % ",".join(targs) if targs else "")
        out[key] = dict(vgpr=get("vgpr_count"), agpr=int(ent.split()[0]), sgpr=get("sgpr_count"), spill=get("vgpr_spill_count"),
                        scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"), symbol=sym)
        if dis:
            out[key]["scratch_instructions"] = nscr.get(sym, 0)
    return out


def build_problem(problem, force=False):
    """generate + compile (cached by content hash).  Returns (lib path, info)."""
    _, info = write_header(problem)
    return compile_model(info["name"], force=force), info


def build_many(problems, force=False, workers=None):
    infos = [write_header(pb)[1] for pb in problems]
    with ThreadPoolExecutor(max_workers=workers or min(8, os.cpu_count() or 1)) as ex:
        libs = list(ex.map(lambda i: compile_model(i["name"], force=force), infos))
    return list(zip(libs, infos))
