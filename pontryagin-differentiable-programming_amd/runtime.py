"""ctypes binding of the C-ABI in include/pdp_hip.h + torch tensor plumbing (device memory, streams).

PyTorch is used for HBM allocation / stream handles only; every arithmetic result of the hot path comes
from the HIP kernels in libpdp_hip.so / libpdp_model_*.so.  There is NO CPU fallback: if the shared
library is missing or the GPU is absent, calls raise (RuntimeError) instead of silently computing elsewhere.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
CORE_LIB = os.path.join(LIB_DIR, "libpdp_hip.so")

PDP_E = {-1: "PDP_E_ARG (null pointer / bad size)", -2: "PDP_E_SIZE (dimension outside kernel limits)",
         -3: "PDP_E_LAUNCH (HIP launch failed)", -4: "PDP_E_MODE (entry point not provided by this model kind)"}


class PdpMat(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bstride", C.c_int64), ("tstride", C.c_int64)]


class PdpLqrProblem(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("n", C.c_int), ("m", C.c_int), ("p", C.c_int)] + \
               [(k, PdpMat) for k in ("F", "G", "E", "Hxx", "Hxu", "Hxe", "Huu", "Hue", "hxx", "hxe", "X0")]


class PdpModelInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("n", C.c_int), ("m", C.c_int), ("p", C.c_int), ("nnz_path", C.c_int), ("chunk", C.c_int),
                ("name", C.c_char_p)]


class PdpOcAuxsys(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue", "hxx", "hxe", "dHu", "Huu_damp")]


class PdpOcSolveOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("newton_switch", C.c_double), ("max_iter", C.c_int), ("check_every", C.c_int), ("ls_trials", C.c_int),
                ("straggler_patience", C.c_int), ("print_level", C.c_int)]


class PdpOcMsOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int), ("flags", C.c_int), ("log_rows", C.c_int), ("dtheta_bstride", C.c_int),
                ("dtheta", C.c_void_p), ("dxdp", C.c_void_p), ("dudp", C.c_void_p), ("riccati", C.c_void_p), ("predict_record", C.c_void_p)]


class PdpOcSensOut(C.Structure):
    _fields_ = [("dxdp", C.c_void_p), ("dudp", C.c_void_p), ("riccati", C.c_void_p), ("predict_record", C.c_void_p)]


class PdpPolicy(C.Structure):
    _fields_ = [("kind", C.c_int), ("n_pivots", C.c_int), ("pivots", C.c_double * 16), ("n_layers", C.c_int), ("sizes", C.c_int * 16), ("n_basis", C.c_int),
                ("table", C.c_void_p)]


CORE_SYMBOLS = ["pdp_hip_version", "pdp_lqr_workspace_bytes", "pdp_lqr_solve_batched", "pdp_cp_aux_integrate_batched",
                "pdp_sysid_aux_integrate_batched", "pdp_gd_update_batched"]
MODEL_SYMBOLS = ["pdp_model_get_info", "pdp_oc_rollout_batched", "pdp_oc_rollout_feedback_batched", "pdp_oc_costate_batched", "pdp_oc_ms_residuals_batched",
                 "pdp_oc_auxsys_batched",
                 "pdp_oc_solve_workspace_bytes", "pdp_oc_solve_batched", "pdp_oc_solve_ms_workspace_bytes", "pdp_oc_solve_ms_batched",
                 "pdp_oc_pdp_workspace_bytes", "pdp_oc_pdp_grad_batched", "pdp_oc_riccati_doubles", "pdp_oc_predict_record_floats", "pdp_oc_pdp_grad_sens_batched",
                 "pdp_oc_predict_batched", "pdp_oc_predict_record_batched",
                 "pdp_cp_integrate_batched", "pdp_cp_auxsys_batched",
                 "pdp_cp_step_workspace_bytes", "pdp_cp_step_batched", "pdp_sysid_integrate_batched", "pdp_sysid_auxsys_batched", "pdp_sysid_step_batched",
                 "pdp_sysid_step_workspace_bytes", "pdp_sysid_step_ws_batched"]

_core = None


def _dlopen(path):
    """dlopen one of our libraries AFTER torch, so that its libamdhip64.so.7 dependency resolves to the HIP runtime
    torch already loaded (one runtime per process; loading /opt/rocm's copy first leaves torch without a device)."""
    import torch  # noqa: F401
    return C.CDLL(path)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, PDP_E.get(rc, rc)))


def load_core():
    """dlopen libpdp_hip.so (built by __graft_entry__.build()); raises if it is not there."""
    global _core
    if _core is None:
        if not os.path.exists(CORE_LIB):
            raise RuntimeError("libpdp_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback" % CORE_LIB)
        lib = _dlopen(CORE_LIB)
        lib.pdp_hip_version.restype = C.c_char_p
        lib.pdp_lqr_workspace_bytes.restype = C.c_int64
        lib.pdp_lqr_workspace_bytes.argtypes = [C.c_int] * 6
        lib.pdp_lqr_solve_batched.restype = C.c_int
        lib.pdp_lqr_solve_batched.argtypes = [C.POINTER(PdpLqrProblem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int64, C.c_void_p]
        lib.pdp_cp_aux_integrate_batched.restype = C.c_int
        lib.pdp_cp_aux_integrate_batched.argtypes = [C.c_int] * 5 + [C.c_void_p] * 7 + [C.c_void_p]
        lib.pdp_sysid_aux_integrate_batched.restype = C.c_int
        lib.pdp_sysid_aux_integrate_batched.argtypes = [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_void_p]
        lib.pdp_gd_update_batched.restype = C.c_int
        lib.pdp_gd_update_batched.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double] + [C.c_void_p] * 4 + \
                                             [C.c_int64, C.c_void_p, C.c_void_p]
        _core = lib
    return _core


def gd_update(loss, grad, lr, theta, dtheta, counters, status=None, converged=None, iterations=None, loss_trace=None, parameter_trace=None):
    """theta <- theta - lr * mean gradient with traces and health counters, one launch (pdp_gd_update_batched).  loss [B], grad [B, p] (rows may be strided: a view of
    the packed [B, p + 1] output), theta / dtheta [p] fp64, counters int64 [4] = (iterations done, unconverged solves, numerical trouble, Newton iterations);
    status / converged / iterations: int32 [B] or None."""
    torch = torch_cuda()
    B, p = grad.shape
    assert loss.dtype == torch.float64 and grad.dtype == torch.float64 and grad.stride(1) == 1 and loss.is_contiguous() and counters.dtype == torch.int64 and counters.numel() >= 4
    for t in (status, converged, iterations):
        assert t is None or (t.dtype == torch.int32 and t.is_contiguous() and t.numel() == B)
    assert theta.dtype == torch.float64 and theta.numel() == p and dtheta.numel() == p and theta.is_contiguous() and dtheta.is_contiguous()
    tl = 0
    if loss_trace is not None:
        tl = loss_trace.shape[0]
        assert loss_trace.is_contiguous() and (parameter_trace is None or (parameter_trace.is_contiguous() and tuple(parameter_trace.shape) == (tl, p)))
    elif parameter_trace is not None:
        tl = parameter_trace.shape[0]
        assert parameter_trace.is_contiguous() and parameter_trace.shape[1] == p
    check(load_core().pdp_gd_update_batched(B, p, ptr(loss), ptr(grad), int(grad.stride(0)), ptr(status), ptr(converged), ptr(iterations), float(lr), ptr(theta), ptr(dtheta),
                                            ptr(loss_trace), ptr(parameter_trace), int(tl), ptr(counters), current_stream_ptr()), "pdp_gd_update_batched")


def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible (torch.cuda.is_available() is False): the PDP kernels need a GPU, there is no CPU fallback")
    return torch


def current_stream_ptr():
    torch = torch_cuda()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a, device=None):
    """numpy / list / tensor -> contiguous fp64 CUDA tensor."""
    torch = torch_cuda()
    if isinstance(a, torch.Tensor):
        return a.to(device=device or "cuda", dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=device or "cuda")


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _mat(t, B, time_varying):
    """tensor [B?,T?,r,c] -> PdpMat with zero strides on broadcast axes."""
    if t is None:
        return PdpMat(None, 0, 0), None
    t = dev(t)
    r, c = t.shape[-2], t.shape[-1]
    if time_varying:
        if t.dim() == 4:
            bs, ts = (t.shape[1] * r * c if t.shape[0] > 1 else 0), (r * c if t.shape[1] > 1 else 0)
        elif t.dim() == 3:
            bs, ts = 0, (r * c if t.shape[0] > 1 else 0)
        else:
            bs, ts = 0, 0
    else:
        bs, ts = (r * c if (t.dim() == 3 and t.shape[0] > 1) else 0), 0
    return PdpMat(t.data_ptr(), bs, ts), t


def lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=None, Hxu=None, Hxe=None, Hue=None, X0=None, T=None, want_costate=True):
    """Batched LQR.lqrSolver.  Time-varying families are [B,T,r,c] (or [T,r,c] shared over the batch, or [r,c]
    time-invariant); terminal / initial ones [B,r,c] or [r,c].  Returns (X [B,T+1,n,p], U [B,T,m,p], Lam or None, status [B])."""
    torch = torch_cuda()
    lib = load_core()
    keep = []
    F_t = dev(F)
    n = F_t.shape[-1]
    G_t = dev(G)
    m = G_t.shape[-1]
    hxe_t = dev(hxe)
    p = hxe_t.shape[-1]
    B = 1
    for a in (F_t, G_t):
        if a.dim() == 4:
            B = max(B, a.shape[0])
    for a in (hxe_t, dev(hxx)) + ((dev(X0),) if X0 is not None else ()):
        if a.dim() == 3:
            B = max(B, a.shape[0])
    if T is None:
        T = F_t.shape[-3] if F_t.dim() >= 3 else None
    assert T is not None, "horizon T is required for time-invariant problems"
    pr = PdpLqrProblem()
    pr.B, pr.T, pr.n, pr.m, pr.p = B, int(T), n, m, p
    for name, val, tv in (("F", F_t, True), ("G", G_t, True), ("E", E, True), ("Hxx", Hxx, True), ("Hxu", Hxu, True), ("Hxe", Hxe, True),
                          ("Huu", Huu, True), ("Hue", Hue, True), ("hxx", hxx, False), ("hxe", hxe_t, False), ("X0", X0, False)):
        mat, t = _mat(val, B, tv)
        setattr(pr, name, mat)
        keep.append(t)
    # parameter columns one launch carries: 4 tiles of 16 (the first shared with the controls) in the tile kernels, 32 in the generic one
    pmax = 64 - m if (n <= 16 and m <= 4) else 32
    if p > pmax:
        # more columns than the kernel's tiles hold (a neural policy's auxvar, a wide X0): the columns of (E, Hxe, Hue, hxe, X0) -> (X, U, Lam)
        # are independent given the gains, so the problem is solved in column blocks (each launch repeats the n x n backward recursion)
        cut = lambda a, c0, c1: None if a is None else dev(a)[..., c0:c1].contiguous()
        parts = [lqr_solve(F_t, G_t, Hxx, Huu, hxx, cut(hxe_t, c0, min(p, c0 + pmax)), E=cut(E, c0, min(p, c0 + pmax)), Hxu=Hxu, Hxe=cut(Hxe, c0, min(p, c0 + pmax)),
                           Hue=cut(Hue, c0, min(p, c0 + pmax)), X0=cut(X0, c0, min(p, c0 + pmax)), T=T, want_costate=want_costate) for c0 in range(0, p, pmax)]
        status = parts[0][3]
        for q in parts[1:]:
            status = status | q[3]
        return (torch.cat([q[0] for q in parts], dim=-1), torch.cat([q[1] for q in parts], dim=-1),
                torch.cat([q[2] for q in parts], dim=-1) if want_costate else None, status)
    X = torch.empty((B, T + 1, n, p), dtype=torch.float64, device="cuda")
    U = torch.empty((B, T, m, p), dtype=torch.float64, device="cuda")
    Lam = torch.empty((B, T, n, p), dtype=torch.float64, device="cuda") if want_costate else None
    status = torch.zeros((B,), dtype=torch.int32, device="cuda")
    nbytes = lib.pdp_lqr_workspace_bytes(B, T, n, m, p, 1 if want_costate else 0)
    ws = torch.empty((max(nbytes, 8) // 8,), dtype=torch.float64, device="cuda")
    rc = lib.pdp_lqr_solve_batched(C.byref(pr), ptr(X), ptr(U), ptr(Lam), ptr(status), ptr(ws), nbytes, current_stream_ptr())
    check(rc, "pdp_lqr_solve_batched")
    return X, U, Lam, status


def cp_aux_integrate(F, G, Ux, Ue, X0=None):
    torch = torch_cuda()
    lib = load_core()
    F, G, Ux, Ue = dev(F), dev(G), dev(Ux), dev(Ue)
    B, T, n, _ = F.shape
    m, p = Ue.shape[-2], Ue.shape[-1]
    X0 = dev(X0) if X0 is not None else None
    X = torch.empty((B, T + 1, n, p), dtype=torch.float64, device="cuda")
    U = torch.empty((B, T, m, p), dtype=torch.float64, device="cuda")
    check(lib.pdp_cp_aux_integrate_batched(B, T, n, m, p, ptr(F), ptr(G), ptr(Ux), ptr(Ue), ptr(X0), ptr(X), ptr(U), current_stream_ptr()),
          "pdp_cp_aux_integrate_batched")
    return X, U


def sysid_aux_integrate(F, E, X0=None):
    torch = torch_cuda()
    lib = load_core()
    F, E = dev(F), dev(E)
    B, T, n, _ = F.shape
    p = E.shape[-1]
    X0 = dev(X0) if X0 is not None else None
    X = torch.empty((B, T + 1, n, p), dtype=torch.float64, device="cuda")
    check(lib.pdp_sysid_aux_integrate_batched(B, T, n, p, ptr(F), ptr(E), ptr(X0), ptr(X), current_stream_ptr()),
          "pdp_sysid_aux_integrate_batched")
    return X


# ------------------------------------------------------------------------------------------------------
# per-model libraries (section B of include/pdp_hip.h)
# ------------------------------------------------------------------------------------------------------
_VP, _I, _I64 = C.c_void_p, C.c_int, C.c_int64
_MODEL_SIGS = {
    "pdp_model_get_info": (None, [C.POINTER(PdpModelInfo)]),
    "pdp_oc_rollout_batched": (_I, [_I, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "pdp_oc_costate_batched": (_I, [_I, _I, _VP, _VP, _VP, _I, _VP, _VP]),
    "pdp_oc_ms_residuals_batched": (_I, [_I, _I, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "pdp_oc_rollout_feedback_batched": (_I, [_I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "pdp_oc_auxsys_batched": (_I, [_I, _I, _VP, _VP, _VP, _VP, _I, C.POINTER(PdpOcAuxsys), _VP]),
    "pdp_oc_solve_workspace_bytes": (_I64, [_I, _I, _I]),
    "pdp_oc_solve_batched": (_I, [_I, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(PdpOcSolveOpts), C.POINTER(C.c_int), _VP, _I64, _VP]),
    "pdp_oc_solve_ms_workspace_bytes": (_I64, [_I, _I, _I]),
    "pdp_oc_solve_ms_batched": (_I, [_I, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(PdpOcMsOpts), _VP, _I64, _VP]),
    "pdp_oc_pdp_workspace_bytes": (_I64, [_I, _I]),
    "pdp_oc_pdp_grad_batched": (_I, [_I, _I, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _VP]),
    "pdp_oc_riccati_doubles": (_I64, []),
    "pdp_oc_predict_record_floats": (_I64, []),
    "pdp_oc_pdp_grad_sens_batched": (_I, [_I, _I, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(PdpOcSensOut), _VP, _VP, _I64, _VP]),
    "pdp_oc_predict_batched": (_I, [_I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pdp_oc_predict_record_batched": (_I, [_I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "pdp_cp_integrate_batched": (_I, [_I, _I, C.POINTER(PdpPolicy), _I, _VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "pdp_cp_auxsys_batched": (_I, [_I, _I, C.POINTER(PdpPolicy), _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pdp_cp_step_workspace_bytes": (_I64, [_I, _I, C.POINTER(PdpPolicy), _I]),
    "pdp_cp_step_batched": (_I, [_I, _I, C.POINTER(PdpPolicy), _I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _I64, _VP]),
    "pdp_sysid_integrate_batched": (_I, [_I, _I, _VP, _VP, _VP, _I, _VP, _VP]),
    "pdp_sysid_auxsys_batched": (_I, [_I, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "pdp_sysid_step_batched": (_I, [_I, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "pdp_sysid_step_workspace_bytes": (_I64, [_I, _I]),
    "pdp_sysid_step_ws_batched": (_I, [_I, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _I64, _VP]),
}
_models = {}


def make_policy(kind, pivots=None, layers=None, table=None):
    """pdp_policy (include/pdp_hip.h): "poly" (Lagrange pivots, PDP.py:699-725), "mlp" (layers = hidden + [m], PDP.py:727-759) or "table" (table [T, n_basis]: the basis
    values of an open-loop policy at every step, u_t = sum_i table[t, i] theta_i - the warped / recovery-matrix variants, PDP.py:882-1141).  A table policy keeps its
    device tensor alive through the returned struct (pol._table)."""
    pol = PdpPolicy()
    if kind == "poly":
        pol.kind = 0
        pol.n_pivots = len(pivots)
        assert pol.n_pivots <= 16, "at most 16 Lagrange pivots"
        for i, v in enumerate(pivots):
            pol.pivots[i] = float(v)
    elif kind == "table":
        pol.kind = 2
        pol._table = dev(np.ascontiguousarray(np.asarray(table, dtype=float)))
        assert pol._table.dim() == 2
        pol.n_basis = int(pol._table.shape[1])
        pol.table = pol._table.data_ptr()
    else:
        pol.kind = 1
        pol.n_layers = len(layers)
        assert pol.n_layers <= 16, "at most 16 weight layers"
        for i, v in enumerate(layers):
            pol.sizes[i] = int(v)
    return pol


class ModelLib:
    """One generated model library; methods mirror the C entry points on torch tensors."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError("model library %s not built; there is no CPU fallback" % path)
        self.path = path
        self.lib = _dlopen(path)
        for name, (res, args) in _MODEL_SIGS.items():
            fn = getattr(self.lib, name)
            fn.restype, fn.argtypes = res, args
        info = PdpModelInfo()
        self.lib.pdp_model_get_info(C.byref(info))
        self.kind, self.n, self.m, self.p = info.kind, info.n, info.m, info.p
        self.nnz_path, self.chunk, self.name = info.nnz_path, info.chunk, info.name.decode()

    # -- helpers
    def _theta(self, theta, B, p=None):
        th = dev(theta)
        p = self.p if p is None else p
        th = th.reshape(-1, p)
        assert th.shape[0] in (1, B), "theta must be [p] (shared) or [B,p]"
        return th, (p if th.shape[0] == B and B > 1 else 0)

    # -- OC
    def oc_rollout(self, x0, u, theta, want_cost=True):
        torch = torch_cuda()
        x0, u = dev(x0).reshape(-1, self.n), dev(u)
        B, T = x0.shape[0], u.shape[1]
        th, tb = self._theta(theta, B)
        x = torch.empty((B, T + 1, self.n), dtype=torch.float64, device="cuda")
        cost = torch.empty((B,), dtype=torch.float64, device="cuda") if want_cost else None
        check(self.lib.pdp_oc_rollout_batched(B, T, ptr(x0), ptr(u), ptr(th), tb, ptr(x), ptr(cost), current_stream_ptr()), "pdp_oc_rollout_batched")
        return x, cost

    def oc_rollout_feedback(self, x0, ubar, xbar, gains, alpha, theta):
        torch = torch_cuda()
        x0, ubar, xbar, gains, alpha = dev(x0).reshape(-1, self.n), dev(ubar), dev(xbar), dev(gains), dev(alpha)
        B, T = ubar.shape[0], ubar.shape[1]
        th, tb = self._theta(theta, B)
        x = torch.empty((B, T + 1, self.n), dtype=torch.float64, device="cuda")
        u = torch.empty((B, T, self.m), dtype=torch.float64, device="cuda")
        cost = torch.empty((B,), dtype=torch.float64, device="cuda")
        check(self.lib.pdp_oc_rollout_feedback_batched(B, T, ptr(x0), ptr(ubar), ptr(xbar), ptr(gains), ptr(alpha), ptr(th), tb, ptr(x), ptr(u), ptr(cost),
                                                       current_stream_ptr()), "pdp_oc_rollout_feedback_batched")
        return x, u, cost

    def oc_solve(self, x0, u_init, theta, tol=1e-9, newton_switch=1e-2, max_iter=300, check_every=2, ls_trials=10, straggler_patience=0,
                 print_level=0, want_gains=False):
        """Batched Newton solve of the OC problem (pdp_oc_solve_batched; stands where OCSys.ocSolver calls IPOPT).  u_init [B,T,m] is
        not modified.  Returns dict(state, control, costate, cost, grad_norm, converged (bool), iterations[, gains])."""
        torch = torch_cuda()
        x0, u = dev(x0).reshape(-1, self.n), dev(u_init).clone()
        B, T = u.shape[0], u.shape[1]
        th, tb = self._theta(theta, B)
        f64 = dict(dtype=torch.float64, device="cuda")
        x, lam = torch.empty((B, T + 1, self.n), **f64), torch.empty((B, T, self.n), **f64)
        cost, gnorm = torch.empty((B,), **f64), torch.empty((B,), **f64)
        conv = torch.zeros((B,), dtype=torch.int32, device="cuda")
        gains = torch.empty((B, T, self.n * self.m + self.m), **f64) if want_gains else None
        nbytes = self.lib.pdp_oc_solve_workspace_bytes(B, T, ls_trials)
        ws = torch.empty((max(nbytes, 8) // 8,), **f64)
        opts = PdpOcSolveOpts(tol, newton_switch, int(max_iter), int(check_every), int(ls_trials), int(straggler_patience), int(print_level))
        it = C.c_int(0)
        check(self.lib.pdp_oc_solve_batched(B, T, ptr(x0), ptr(th), tb, ptr(u), ptr(x), ptr(lam), ptr(cost), ptr(gnorm), ptr(conv), ptr(gains),
                                            C.byref(opts), C.byref(it), ptr(ws), nbytes, current_stream_ptr()), "pdp_oc_solve_batched")
        out = {"state": x, "control": u, "costate": lam, "cost": cost, "grad_norm": gnorm, "converged": conv != 0, "iterations": it.value}
        if want_gains:
            out["gains"] = gains
        return out

    def oc_solve_ms(self, x0, theta, T, tol=1e-10, max_iter=300, warm=None, want_gains=False, log_rows=0, restoration=True, u_init=None, consume_warm=False, predict=None,
                    soc=False, watchdog=False):
        """watchdog=True: IPOPT's watchdog in the line search (PDP_MS_WITH_WATCHDOG, include/pdp_hip.h: opt-in, not together with soc) - for cold solves that crawl.
        The reference's multiple-shooting NLP (PDP.py:131-182) solved by IPOPT's algorithm from its all-zero initial guess
        (pdp_oc_solve_ms_batched: a persistent pair of wavefronts per trajectory, all iterations in one launch).  warm = (x, u, lam) starts
        from a given point instead (x[:, 0] is replaced by x0).  restoration=False: a line search that falls below alpha_min ends the trajectory with
        PDP_MS_RESTORATION instead of entering the feasibility restoration (include/pdp_hip.h).  u_init [B, T, m] (instead of warm): start from these
        controls, their rollout and the least-squares multiplier estimate (PDP_MS_FROM_CONTROLS).  consume_warm: the warm tensors themselves become the outputs
        (no copies: for callers that built the starting point for this call, e.g. oc_predict).  predict = dict(dtheta [p] or [B, p], dxdp, dudp[, riccati]) with
        warm = the solution at the previous parameter: the kernel starts from its first-order prediction for the step dtheta (PDP_MS_PREDICT: what oc_predict computes,
        applied while the point is loaded - no extra launch; with consume_warm the previous solution's tensors are overwritten by the new one, as an IRL loop wants it).
        predict["guard"] (default True, PDP_MS_PREDICT_GUARD): the previous solution is evaluated beside its prediction and the solve starts from whichever has the smaller
        scaled KKT error (a first-order prediction across a large parameter step can be worse than no prediction: status & 512 marks the trajectories where it was dropped).
        soc=True (PDP_MS_WITH_SOC): IPOPT's second-order correction in the line search (status & 1024: a corrected step was taken; include/pdp_hip.h says why it is off by default).
        Returns dict(state, control, costate, cost,
        resid [B,2], converged (bool), iterations [B], status [B][, gains])."""
        torch = torch_cuda()
        x0 = dev(x0).reshape(-1, self.n)
        B, T = x0.shape[0], int(T)
        th, tb = self._theta(theta, B)
        f64 = dict(dtype=torch.float64, device="cuda")
        if warm is not None:
            assert u_init is None, "warm and u_init exclude each other"
            if consume_warm:        # in place: the caller's tensors ARE the outputs - a silent copy would leave an IRL loop re-solving from a stale point
                for a in warm:          # (an exception, not an assert: under `python -O` the silent copy would come back)
                    if not (torch.is_tensor(a) and a.is_cuda and a.dtype == torch.float64 and a.is_contiguous()):
                        raise TypeError("oc_solve_ms(consume_warm=True) works IN PLACE on the warm start: state, control and costate must be contiguous fp64 CUDA tensors "
                                        "(got %s); pass consume_warm=False to start from a copy of anything dev() converts" % (type(a).__name__ if not torch.is_tensor(a)
                                        else "%s %s tensor, contiguous=%s" % (a.device.type, a.dtype, a.is_contiguous())))
                x, u, lam = warm
            else:
                x, u, lam = (dev(a).clone().contiguous() for a in warm)
            assert x.shape == (B, T + 1, self.n) and u.shape == (B, T, self.m) and lam.shape == (B, T, self.n)
        elif u_init is not None:
            u = dev(u_init).clone().contiguous()
            assert u.shape == (B, T, self.m)
            x, lam = torch.zeros((B, T + 1, self.n), **f64), torch.zeros((B, T, self.n), **f64)
        else:
            x, u, lam = torch.empty((B, T + 1, self.n), **f64), torch.empty((B, T, self.m), **f64), torch.empty((B, T, self.n), **f64)
        cost, resid = torch.empty((B,), **f64), torch.empty((B, 2), **f64)
        conv, iters, status = (torch.zeros((B,), dtype=torch.int32, device="cuda") for _ in range(3))
        gains = torch.empty((B, T, self.n * self.m + self.m), **f64) if want_gains else None
        nbytes = self.lib.pdp_oc_solve_ms_workspace_bytes(B, T, int(max_iter))
        ws = torch.empty((max(nbytes, 8) // 8,), **f64)
        log = torch.zeros((B, int(log_rows), 8), **f64) if log_rows > 0 else None
        opts = PdpOcMsOpts(float(tol), int(max_iter), (1 if warm is not None else 0) | (0 if restoration else 2) | (9 if u_init is not None else 0) | (128 if soc else 0) | (256 if watchdog else 0), int(log_rows))
        keep = None
        if predict is not None:
            assert warm is not None, "predict needs the previous solution as the warm point"
            dth, dtb = self._theta(predict["dtheta"], B)
            opts.flags |= 16
            opts.dtheta_bstride, opts.dtheta = dtb, dth.data_ptr()
            if predict.get("record") is not None:           # the packed fp32 record (oc_pdp_grad(want_predict_record=True))
                keep = (dth, predict["record"])
                assert keep[1].dtype == torch.float32 and tuple(keep[1].shape) == (B, T, int(self.lib.pdp_oc_predict_record_floats())) and keep[1].is_contiguous()
                opts.predict_record = keep[1].data_ptr()
            else:
                keep = (dth, dev(predict["dxdp"]), dev(predict["dudp"]), dev(predict["riccati"]) if predict.get("riccati") is not None else None)
                assert keep[1].shape == (B, T + 1, self.n, self.p) and keep[2].shape == (B, T, self.m, self.p)
                opts.dxdp, opts.dudp = keep[1].data_ptr(), keep[2].data_ptr()
                opts.riccati = keep[3].data_ptr() if keep[3] is not None else None
            if predict.get("primal"):                       # PDP_MS_PREDICT_PRIMAL: states and controls only (a record written with want_predict_record="primal" holds nothing else)
                opts.flags |= 32
            if predict.get("guard", True):                  # PDP_MS_PREDICT_GUARD (default): the prediction is kept only if its KKT error does not exceed the previous solution's;
                opts.flags |= 64                            # status bit 512 (PDP_MS_PREDICT_REJECTED) says where it was not
        check(self.lib.pdp_oc_solve_ms_batched(B, T, ptr(x0), ptr(th), tb, ptr(x), ptr(u), ptr(lam), ptr(cost), ptr(resid), ptr(conv), ptr(iters),
                                               ptr(status), ptr(gains), ptr(log), C.byref(opts), ptr(ws), nbytes, current_stream_ptr()), "pdp_oc_solve_ms_batched")
        out = {"state": x, "control": u, "costate": lam, "cost": cost, "resid": resid, "converged": conv != 0, "converged_flags": conv, "iterations": iters, "status": status}
        if want_gains:
            out["gains"] = gains
        if log is not None:
            out["log"] = log            # [B, log_rows, 8]: iteration, objective, inf_pr, inf_du, dw, alpha (minus alpha: a second-order-corrected step), grad(phi)'d, theta
        return out

    def oc_costate(self, x, u, theta):
        torch = torch_cuda()
        x, u = dev(x), dev(u)
        B, T = u.shape[0], u.shape[1]
        th, tb = self._theta(theta, B)
        lam = torch.empty((B, T, self.n), dtype=torch.float64, device="cuda")
        check(self.lib.pdp_oc_costate_batched(B, T, ptr(x), ptr(u), ptr(th), tb, ptr(lam), current_stream_ptr()), "pdp_oc_costate_batched")
        return lam

    def oc_ms_residuals(self, x, u, lam, theta):
        """Residuals of ocSolver's multiple-shooting NLP at (x, u, lam) (pdp_oc_ms_residuals_batched): dict(c [B,T,n] defects, rx [B,T+1,n], ru [B,T,m] Lagrangian
        gradients, cost [B,T+1] stage costs with the final cost last)."""
        torch = torch_cuda()
        x, u, lam = dev(x), dev(u), dev(lam)
        B, T = u.shape[0], u.shape[1]
        th, tb = self._theta(theta, B)
        f64 = dict(dtype=torch.float64, device="cuda")
        out = {"c": torch.empty((B, T, self.n), **f64), "rx": torch.empty((B, T + 1, self.n), **f64), "ru": torch.empty((B, T, self.m), **f64),
               "cost": torch.empty((B, T + 1), **f64)}
        check(self.lib.pdp_oc_ms_residuals_batched(B, T, ptr(x), ptr(u), ptr(lam), ptr(th), tb, ptr(out["c"]), ptr(out["rx"]), ptr(out["ru"]), ptr(out["cost"]),
                                                   current_stream_ptr()), "pdp_oc_ms_residuals_batched")
        return out

    def oc_auxsys(self, x, u, lam, theta, only=None):
        """materialised OCSys.getAuxSys; `only` = iterable of keys to produce (others are skipped), may include 'dHu'"""
        torch = torch_cuda()
        x, u, lam = dev(x), dev(u), dev(lam)
        B, T = u.shape[0], u.shape[1]
        n, m, p = self.n, self.m, self.p
        th, tb = self._theta(theta, B)
        shp = dict(dynF=(B, T, n, n), dynG=(B, T, n, m), dynE=(B, T, n, p), Hxx=(B, T, n, n), Hxu=(B, T, n, m), Hxe=(B, T, n, p),
                   Hux=(B, T, m, n), Huu=(B, T, m, m), Hue=(B, T, m, p), hxx=(B, n, n), hxe=(B, n, p))
        if only is not None:
            shp["dHu"] = (B, T, m)
            shp = {k: v for k, v in shp.items() if k in only}
        out = {k: torch.empty(s, dtype=torch.float64, device="cuda") for k, s in shp.items()}
        o = PdpOcAuxsys(**{k: v.data_ptr() for k, v in out.items()})
        check(self.lib.pdp_oc_auxsys_batched(B, T, ptr(x), ptr(u), ptr(lam), ptr(th), tb, C.byref(o), current_stream_ptr()), "pdp_oc_auxsys_batched")
        return out

    def oc_pdp_grad(self, u, theta, demo_x, demo_u, x0=None, x=None, lam=None, want_sens=False, buffers=None, packed=False, want_riccati=False,
                    want_predict_record=False):
        """Fused forward + Riccati + PDP gradient.  Give (x, lam) to use an optimal trajectory (PDP_OC_GIVEN_TRAJ),
        else x0 and the kernel integrates u and the costates itself.  Returns dict(loss, grad, x, lam, status[, dxdp, dudp][, riccati]).
        packed: the kernel writes gradient and loss as one [B, p+1] tensor (PDP_OC_PACKED; out["packed"], out["grad"] is a view of it).
        want_riccati (with want_sens: everything oc_predict needs): also the Riccati matrices of the auxiliary control system,
        out["riccati"] [B, T, n n + n p + 1] = P_{t+1} | W_{t+1} | one scratch word per stage (pdp_oc_pdp_grad_sens_batched).
        want_predict_record: out["predict_record"], float32 [B, T, 2 n p + m p + n (n + 1) / 2] - the same information packed in single precision, what an IRL loop
        hands to the next oc_solve_ms(predict=dict(dtheta=..., record=...)) (2.4 times less memory traffic than the fp64 outputs).  want_predict_record="primal"
        (PDP_OC_RECORD_PRIMAL): only the X | U part of the record is written - for oc_solve_ms(predict=dict(..., primal=True)), the prediction of states and controls."""
        torch = torch_cuda()
        u, demo_x, demo_u = dev(u), dev(demo_x), dev(demo_u)
        B, T = u.shape[0], u.shape[1]
        n, m, p = self.n, self.m, self.p
        th, tb = self._theta(theta, B)
        flags = 0
        if x is not None:
            assert lam is not None
            x, lam, flags = dev(x), dev(lam), 1
            x0 = None
        else:
            x0 = dev(x0).reshape(B, n)
        bufs = buffers if buffers is not None else {}

        def buf(key, shape, dtype=torch.float64):
            t = bufs.get(key)
            if t is None or tuple(t.shape) != tuple(shape):
                t = torch.empty(shape, dtype=dtype, device="cuda")
                bufs[key] = t
            return t
        if x is None:
            x, lam = buf("x", (B, T + 1, n)), buf("lam", (B, T, n))
        loss = buf("loss", (B,))
        if packed:
            pk = buf("packed", (B, p + 1))
            grad, flags = pk[:, :p], flags | 2
        else:
            pk = grad = buf("grad", (B, p))
        status = buf("status", (B,), torch.int32)
        dxdp = buf("dxdp", (B, T + 1, n, p)) if want_sens else None
        dudp = buf("dudp", (B, T, m, p)) if want_sens else None
        ric = buf("riccati", (B, T, int(self.lib.pdp_oc_riccati_doubles()))) if want_riccati else None
        prec = buf("predict_record", (B, T, int(self.lib.pdp_oc_predict_record_floats())), torch.float32) if want_predict_record else None
        if want_predict_record == "primal":
            flags |= 4
        nbytes = self.lib.pdp_oc_pdp_workspace_bytes(B, T)
        ws = buf("ws", (max(nbytes, 8) // 8,))
        if want_riccati or want_predict_record:
            so = PdpOcSensOut(dxdp.data_ptr() if dxdp is not None else None, dudp.data_ptr() if dudp is not None else None,
                              ric.data_ptr() if ric is not None else None, prec.data_ptr() if prec is not None else None)
            rc = self.lib.pdp_oc_pdp_grad_sens_batched(B, T, flags, ptr(x0), ptr(u), ptr(th), tb, ptr(demo_x), ptr(demo_u), ptr(x), ptr(lam), ptr(loss),
                                                       ptr(pk), C.byref(so), ptr(status), ptr(ws), nbytes, current_stream_ptr())
            if rc == -2:
                raise RuntimeError("pdp_oc_pdp_grad_sens_batched: the Riccati / prediction records are outputs of the fused kernels (n <= 16, m <= 4, m + p <= 16)")
        else:
            rc = self.lib.pdp_oc_pdp_grad_batched(B, T, flags, ptr(x0), ptr(u), ptr(th), tb, ptr(demo_x), ptr(demo_u), ptr(x), ptr(lam), ptr(loss),
                                                  ptr(pk), ptr(dxdp), ptr(dudp), ptr(status), ptr(ws), nbytes, current_stream_ptr())
        if rc == -2:
            # m + p > 16 (beyond the fused kernel's single parameter tile), n > 16 / m > 4 (beyond one tile per matrix: the size-generic LQR
            # kernel takes over - any n, m), or a horizon whose staging exceeds the LDS: the reference's own route, kernel by kernel
            if not getattr(self, "_warned_materialised", False):
                import warnings
                warnings.warn("pdp_oc_pdp_grad_batched: problem outside the fused kernel's limits (n = %d, m = %d, p = %d, T = %d): taking the "
                              "kernel-by-kernel route through HBM (several launches, roughly ten times slower)" % (self.n, self.m, p, T), RuntimeWarning)
                self._warned_materialised = True
            self._oc_pdp_grad_materialised(u, theta, demo_x, demo_u, x0, x, lam, flags, loss, grad, status, dxdp, dudp)
            if packed:
                pk[:, p].copy_(loss)
            rc = 0
        check(rc, "pdp_oc_pdp_grad_batched")
        out = dict(loss=loss, grad=grad, x=x, lam=lam, status=status)
        if packed:
            out["packed"] = pk
        if want_sens:
            out.update(dxdp=dxdp, dudp=dudp)
        if want_riccati:
            out["riccati"] = ric
        if want_predict_record:
            out["predict_record"] = prec
        return out

    def oc_pdp_grad_prepared(self, u, theta, demo_x, demo_u, x0, packed_out=None):
        """The fused OC unit as a PREPARED call: every argument of pdp_oc_pdp_grad_batched marshalled once; the returned step() is one foreign call on the stream that is
        current when it runs (no tensor conversions, no dictionary, no allocation per step) - what a driver that calls the unit in a loop on fixed buffers should use, and what
        bench.py times: oc_pdp_grad's Python wrapper costs 25 - 60 us per call, which a busy host cannot always hide under a 97 us kernel (the driver's 0.104 - 0.109 ms per step
        against 0.097 - 0.099 ms of kernel time).  Returns (step, out) with out = dict(loss, grad, x, lam, status, packed): the tensors every step() overwrites.
        packed_out: a [B, p + 1] tensor to receive gradient | loss (PDP_OC_PACKED), e.g. the buffer a collective sends; allocated when None."""
        torch = torch_cuda()
        u, demo_x, demo_u = dev(u), dev(demo_x), dev(demo_u)
        B, T = u.shape[0], u.shape[1]
        n, p = self.n, self.p
        th, tb = self._theta(theta, B)
        x0 = dev(x0).reshape(B, n)
        f64 = dict(dtype=torch.float64, device="cuda")
        x, lam, loss = torch.empty((B, T + 1, n), **f64), torch.empty((B, T, n), **f64), torch.empty((B,), **f64)
        pk = torch.empty((B, p + 1), **f64) if packed_out is None else packed_out
        assert pk.is_cuda and pk.dtype == torch.float64 and pk.is_contiguous() and tuple(pk.shape) == (B, p + 1)
        status = torch.empty((B,), dtype=torch.int32, device="cuda")
        nbytes = self.lib.pdp_oc_pdp_workspace_bytes(B, T)
        ws = torch.empty((max(nbytes, 8) // 8,), **f64)
        keep = (u, th, demo_x, demo_u, x0, x, lam, loss, pk, status, ws)                 # (the closure owns every buffer the kernel touches)
        fn = self.lib.pdp_oc_pdp_grad_batched
        args = (B, T, 2, ptr(x0), ptr(u), ptr(th), tb, ptr(demo_x), ptr(demo_u), ptr(x), ptr(lam), ptr(loss), ptr(pk), ptr(None), ptr(None), ptr(status), ptr(ws), nbytes)
        rc0 = fn(*args, current_stream_ptr())
        if rc0 != 0:
            raise RuntimeError("pdp_oc_pdp_grad_batched (prepared call): %s - problems outside the fused kernel's limits take oc_pdp_grad" % PDP_E.get(rc0, rc0))
        cur = torch.cuda.current_stream

        def step(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args, C.c_void_p(cur().cuda_stream))
            if rc != 0:
                check(rc, "pdp_oc_pdp_grad_batched")
        return step, dict(loss=loss, grad=pk[:, :p], x=x, lam=lam, status=status, packed=pk)

    def oc_predict_from_record(self, x, u, lam, dtheta, record, primal=False):
        """oc_predict from the packed fp32 record (pdp_oc_predict_record_batched): new tensors (x, u, lam) + first-order change for the step dtheta
        (primal: states and controls only, lam is returned as it came)"""
        x, u, lam = dev(x).clone(), dev(u).clone(), dev(lam).clone()
        B, T = u.shape[0], u.shape[1]
        dth, dtb = self._theta(dtheta, B)
        assert record.dtype == torch_cuda().float32 and tuple(record.shape) == (B, T, int(self.lib.pdp_oc_predict_record_floats()))
        check(self.lib.pdp_oc_predict_record_batched(B, T, ptr(dth), dtb, ptr(record), ptr(x), ptr(u), None if primal else ptr(lam), current_stream_ptr()),
              "pdp_oc_predict_record_batched")
        return x, u, lam

    def oc_predict(self, x, u, lam, dtheta, dxdp, dudp, riccati=None):
        """First-order prediction of the optimal (x, u, lam) at theta + dtheta from the gradient unit's sensitivity outputs (pdp_oc_predict_batched):
        returns NEW tensors x + X dtheta, u + U dtheta and - with `riccati` - lam_t + P_{t+1} X_{t+1} dtheta + W_{t+1} dtheta (else lam unchanged).
        dtheta [p] (shared) or [B, p].  What an IRL loop hands to oc_solve_ms(warm=...) at the next parameter."""
        torch = torch_cuda()
        x, u, lam = dev(x).clone(), dev(u).clone(), dev(lam).clone()
        B, T = u.shape[0], u.shape[1]
        dth, dtb = self._theta(dtheta, B)
        dxdp, dudp = dev(dxdp), dev(dudp)
        assert dxdp.shape == (B, T + 1, self.n, self.p) and dudp.shape == (B, T, self.m, self.p)
        ric = dev(riccati) if riccati is not None else None
        check(self.lib.pdp_oc_predict_batched(B, T, ptr(dth), dtb, ptr(dxdp), ptr(dudp), ptr(ric), ptr(x), ptr(u), ptr(lam) if ric is not None else None,
                                              current_stream_ptr()), "pdp_oc_predict_batched")
        return x, u, lam

    def _oc_pdp_grad_materialised(self, u, theta, demo_x, demo_u, x0, x, lam, flags, loss, grad, status, dxdp=None, dudp=None):
        """The PDP gradient unit by the reference's own route (PDP.py:272-314, 557-608 and the chain rule of cartpole_PDP.py:63-74), one
        kernel per stage: trajectory and costates (unless given), aux matrices to HBM, lqrSolver (column blocks for any p), contraction
        with (x - x_demo, u - u_demo).  Fills the given output tensors."""
        torch = torch_cuda()
        B, T = u.shape[0], u.shape[1]
        n, m, p = self.n, self.m, self.p
        if not (flags & 1):
            x.copy_(self.oc_rollout(x0, u, theta, want_cost=False)[0])
            lam.copy_(self.oc_costate(x, u, theta))
        aux = self.oc_auxsys(x, u, lam, theta)
        X, U, _, st = lqr_solve(aux["dynF"], aux["dynG"], aux["Hxx"], aux["Huu"], aux["hxx"], aux["hxe"], E=aux["dynE"], Hxu=aux["Hxu"], Hxe=aux["Hxe"],
                                Hue=aux["Hue"], want_costate=False)
        ex, eu = x - demo_x, u - demo_u
        loss.copy_((ex ** 2).sum(dim=(1, 2)) + (eu ** 2).sum(dim=(1, 2)))
        g = torch.empty((B, p), dtype=torch.float64, device="cuda")
        core = load_core()
        core.pdp_cp_grad_contract_batched.restype = C.c_int
        core.pdp_cp_grad_contract_batched.argtypes = [C.c_int] * 5 + [C.c_void_p] * 7
        ex_path, ex_fin, eu_c = ex[:, :T].contiguous(), ex[:, T].contiguous(), eu.contiguous()      # (named: they must outlive the launch)
        check(core.pdp_cp_grad_contract_batched(B, T, n, m, p, ptr(ex_path), ptr(eu_c), ptr(ex_fin), ptr(X), ptr(U), ptr(g), current_stream_ptr()),
              "pdp_cp_grad_contract_batched")
        grad.copy_(g)
        status.copy_(st)
        if dxdp is not None:
            dxdp.copy_(X)
        if dudp is not None:
            dudp.copy_(U)

    def oc_pdp_grad_materialised(self, u, theta, demo_x, demo_u, x0=None, x=None, lam=None, want_sens=False):
        """same results as oc_pdp_grad through the materialised kernels (getAuxSys -> lqrSolver -> chain rule): the reference's data flow"""
        torch = torch_cuda()
        u, demo_x, demo_u = dev(u), dev(demo_x), dev(demo_u)
        B, T = u.shape[0], u.shape[1]
        f64 = dict(dtype=torch.float64, device="cuda")
        flags = 0
        if x is not None:
            x, lam, flags = dev(x).clone(), dev(lam).clone(), 1
        else:
            x0 = dev(x0).reshape(B, self.n)
            x, lam = torch.empty((B, T + 1, self.n), **f64), torch.empty((B, T, self.n), **f64)
        loss, grad = torch.empty((B,), **f64), torch.empty((B, self.p), **f64)
        status = torch.zeros((B,), dtype=torch.int32, device="cuda")
        dxdp = torch.empty((B, T + 1, self.n, self.p), **f64) if want_sens else None
        dudp = torch.empty((B, T, self.m, self.p), **f64) if want_sens else None
        self._oc_pdp_grad_materialised(u, theta, demo_x, demo_u, x0, x, lam, flags, loss, grad, status, dxdp, dudp)
        out = dict(loss=loss, grad=grad, x=x, lam=lam, status=status)
        if want_sens:
            out.update(dxdp=dxdp, dudp=dudp)
        return out

    # -- ControlPlanning
    def cp_integrate(self, pol, p, x0, theta):
        torch = torch_cuda()
        x0 = dev(x0).reshape(-1, self.n)
        B = x0.shape[0]
        th, tb = self._theta(theta, B, p)
        T = self._T
        x = torch.empty((B, T + 1, self.n), dtype=torch.float64, device="cuda")
        u = torch.empty((B, T, self.m), dtype=torch.float64, device="cuda")
        cost = torch.empty((B,), dtype=torch.float64, device="cuda")
        rc = self.lib.pdp_cp_integrate_batched(B, T, C.byref(pol), p, ptr(x0), ptr(th), tb, ptr(x), ptr(u), ptr(cost), current_stream_ptr())
        if rc == -2:        # a network wider than the lane-per-trajectory integrator's local arrays: the size-generic step kernel rolls out as well (its gradient is dropped)
            loss, _, x, u = self.cp_step(pol, p, x0, th, T, want_traj=True)
            return x, u, loss
        check(rc, "pdp_cp_integrate_batched")
        return x, u, cost

    def cp_integrate_T(self, pol, p, x0, theta, T):
        self._T = int(T)
        return self.cp_integrate(pol, p, x0, theta)

    def cp_auxsys(self, pol, p, x, u, theta, want_cost_grads=True):
        torch = torch_cuda()
        x, u = dev(x), dev(u)
        B, T = u.shape[0], u.shape[1]
        n, m = self.n, self.m
        th, tb = self._theta(theta, B, p)
        e = lambda *s: torch.empty(s, dtype=torch.float64, device="cuda")
        out = dict(dynF=e(B, T, n, n), dynG=e(B, T, n, m), dUx=e(B, T, m, n), dUe=e(B, T, m, p))
        if want_cost_grads:
            out.update(dcx=e(B, T, n), dcu=e(B, T, m), dhx=e(B, n))
        check(self.lib.pdp_cp_auxsys_batched(B, T, C.byref(pol), p, ptr(x), ptr(u), ptr(th), tb, ptr(out["dynF"]), ptr(out["dynG"]), ptr(out["dUx"]),
                                             ptr(out["dUe"]), ptr(out.get("dcx")), ptr(out.get("dcu")), ptr(out.get("dhx")), current_stream_ptr()),
              "pdp_cp_auxsys_batched")
        return out

    def cp_step(self, pol, p, x0, theta, T, want_traj=False):
        """ControlPlanning.step (pdp_cp_step_batched): forward-sensitivity kernels for the Lagrange policy, adjoint kernels for the MLP
        policy, the size-generic adjoint kernel for everything beyond their limits and for table policies (the workspace any of them asks for is allocated here)."""
        torch = torch_cuda()
        x0 = dev(x0).reshape(-1, self.n)
        B = x0.shape[0]
        th, tb = self._theta(theta, B, p)
        loss = torch.empty((B,), dtype=torch.float64, device="cuda")
        grad = torch.empty((B, p), dtype=torch.float64, device="cuda")
        x = torch.empty((B, T + 1, self.n), dtype=torch.float64, device="cuda") if want_traj else None
        u = torch.empty((B, T, self.m), dtype=torch.float64, device="cuda") if want_traj else None
        nbytes = self.lib.pdp_cp_step_workspace_bytes(B, int(T), C.byref(pol), p)
        ws = torch.empty((nbytes // 8,), dtype=torch.float64, device="cuda") if nbytes > 0 else None
        rc = self.lib.pdp_cp_step_batched(B, int(T), C.byref(pol), p, ptr(x0), ptr(th), tb, ptr(loss), ptr(grad), ptr(x), ptr(u), ptr(ws), nbytes,
                                          current_stream_ptr())
        check(rc, "pdp_cp_step_batched")        # (no size is refused since round 5: what the tuned kernels do not take runs on the size-generic adjoint kernel)
        return (loss, grad, x, u) if want_traj else (loss, grad)

    def cp_step_materialised(self, pol, p, x0, theta, T, want_traj=False):
        """ControlPlanning.step exactly as the reference composes it (PDP.py:850-878): integrateSys -> getAuxSys ->
        integrateAuxSys (forward sensitivities, MFMA tiles over the parameter dimension) -> chain rule; every stage materialised in HBM."""
        torch = torch_cuda()
        x0 = dev(x0).reshape(-1, self.n)
        B = x0.shape[0]
        th, tb = self._theta(theta, B, p)
        x, u, loss = self.cp_integrate_T(pol, p, x0, th, T)
        aux = self.cp_auxsys(pol, p, x, u, th)
        X, U = cp_aux_integrate(aux["dynF"], aux["dynG"], aux["dUx"], aux["dUe"])
        grad = torch.empty((B, p), dtype=torch.float64, device="cuda")
        lib = load_core()
        lib.pdp_cp_grad_contract_batched.restype = C.c_int
        lib.pdp_cp_grad_contract_batched.argtypes = [C.c_int] * 5 + [C.c_void_p] * 7
        check(lib.pdp_cp_grad_contract_batched(B, int(T), self.n, self.m, p, ptr(aux["dcx"]), ptr(aux["dcu"]), ptr(aux["dhx"]), ptr(X), ptr(U),
                                               ptr(grad), current_stream_ptr()), "pdp_cp_grad_contract_batched")
        return (loss, grad, x, u) if want_traj else (loss, grad)

    # -- SysID
    def sysid_integrate(self, x0, u, theta):
        torch = torch_cuda()
        x0, u = dev(x0).reshape(-1, self.n), dev(u)
        B, T = u.shape[0], u.shape[1]
        th, tb = self._theta(theta, B)
        x = torch.empty((B, T + 1, self.n), dtype=torch.float64, device="cuda")
        check(self.lib.pdp_sysid_integrate_batched(B, T, ptr(x0), ptr(u), ptr(th), tb, ptr(x), current_stream_ptr()), "pdp_sysid_integrate_batched")
        return x

    def sysid_auxsys(self, x, u, theta):
        torch = torch_cuda()
        x, u = dev(x), dev(u)
        B, T = u.shape[0], u.shape[1]
        th, tb = self._theta(theta, B)
        F = torch.empty((B, T, self.n, self.n), dtype=torch.float64, device="cuda")
        E = torch.empty((B, T, self.n, self.p), dtype=torch.float64, device="cuda")
        check(self.lib.pdp_sysid_auxsys_batched(B, T, ptr(x), ptr(u), ptr(th), tb, ptr(F), ptr(E), current_stream_ptr()), "pdp_sysid_auxsys_batched")
        return F, E

    def sysid_step(self, u, xobs, theta):
        """SysID.step per trajectory (pdp_sysid_step_ws_batched).  Models beyond the fused kernels' tiles (n > 16 or p > 64) take the reference's own route kernel by kernel -
        integrateDyn -> getAuxSys -> integrateAuxSys (size-generic kernels) -> the chain rule of PDP.py:1285-1291 as two tensor contractions: no size is refused."""
        torch = torch_cuda()
        u, xobs = dev(u), dev(xobs)
        B, T = u.shape[0], u.shape[1]
        th, tb = self._theta(theta, B)
        loss = torch.empty((B,), dtype=torch.float64, device="cuda")
        grad = torch.empty((B, self.p), dtype=torch.float64, device="cuda")
        nbytes = int(self.lib.pdp_sysid_step_workspace_bytes(B, T))              # > 0: large batch, the trajectories are rolled out beforehand, one lane each
        ws = None
        if nbytes > 0:
            # kept per (B, T): an SGD loop calls this every step, and a captured hipGraph (irl.GDLoop) holds the raw pointer - a buffer that a later, larger call freed
            # would leave the graph writing through a dangling pointer (round-4 advice).  Bounded (round-5 advice: a caller with ragged batches or horizon sweeps grew
            # one buffer per shape for the lifetime of the library): the eight most recently used shapes stay, and every shape that was used DURING a graph capture
            # stays for good (its pointer is baked into the graph).
            from collections import OrderedDict
            cache = self.__dict__.setdefault("_sysid_ws", OrderedDict())
            pinned = self.__dict__.setdefault("_sysid_ws_pinned", set())
            ws = cache.get((B, T))
            if ws is None:
                ws = cache[(B, T)] = torch.empty((nbytes // 8,), dtype=torch.float64, device="cuda")
            cache.move_to_end((B, T))
            if torch.cuda.is_current_stream_capturing():
                pinned.add((B, T))
            for key in [k for k in cache if k not in pinned][:max(0, len(cache) - len(pinned) - 8)]:
                del cache[key]
        rc = self.lib.pdp_sysid_step_ws_batched(B, T, ptr(u), ptr(xobs), ptr(th), tb, ptr(loss), ptr(grad), ptr(ws), nbytes, current_stream_ptr())
        if rc == -2:
            x = self.sysid_integrate(xobs[:, 0].contiguous(), u, th)
            F, E = self.sysid_auxsys(x, u, th)
            X = sysid_aux_integrate(F, E)                                           # [B, T+1, n, p]
            d = x - xobs
            return (d * d).sum(dim=(1, 2)), torch.einsum("bti,btip->bp", d, X)
        check(rc, "pdp_sysid_step_ws_batched")
        return loss, grad


def load_model(path):
    if path not in _models:
        _models[path] = ModelLib(path)
    return _models[path]
