"""Host-side mirror of the reference's solver interface
    [x,fval,exitflag,iter,lambda,auxOutput] = qpOASES(H,g,A,lb,ub,lbA,ubA{,options})
(optimizers/matlab/qpOASES/qpOASES.m:22-23, bounds-only form :34-35, multi-column form :65-67),
running on the MI355X through libfsaempc.so.  Same argument meaning, same exit flags; the build's
additive extension is a leading batch dimension (3-D H / A)."""
import ctypes as C

import numpy as np

from ._lib import QpAux, QpDesc, check, default_opts, lib


def _col(a, n, name):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    if a.shape[0] != n:
        raise ValueError("ERROR (qpOASES): Input dimension mismatch for argument %s" % name)
    return a


def qpOASES(H, g, *args, options=None):
    """qpOASES(H,g,A,lb,ub,lbA,ubA) or qpOASES(H,g,lb,ub).  g/lb/ub/lbA/ubA may have k columns => k QPs
    sharing H and A (qpOASES.m:65-67).  H (nV,nV[,B]) / A (nC,nV[,B]) with a trailing batch axis => B
    independent QPs.  Returns x, fval, exitflag, iter, lambda, auxOutput (arrays gain a trailing batch axis
    when more than one QP is solved)."""
    if len(args) == 5:
        A, lb, ub, lbA, ubA = args
    elif len(args) == 2:
        lb, ub = args
        A, lbA, ubA = None, None, None
    else:
        raise TypeError("qpOASES(H,g,A,lb,ub,lbA,ubA) or qpOASES(H,g,lb,ub)")
    H = np.asarray(H, dtype=np.float64)
    nV = H.shape[0]
    if H.shape[1] != nV:
        raise ValueError("ERROR (qpOASES): Input dimension mismatch for argument 1")
    batched_HA = H.ndim == 3
    g = _col(g, nV, "2")
    B = H.shape[2] if batched_HA else g.shape[1]
    lb, ub = _col(lb, nV, "lb"), _col(ub, nV, "ub")
    if A is None or np.size(A) == 0:
        nC = 0
        A3 = np.zeros((0, nV, 1))
        lbA = ubA = np.zeros((0, B))
    else:
        A = np.asarray(A, dtype=np.float64)
        nC = A.shape[0]
        if A.shape[1] != nV:
            raise ValueError("ERROR (qpOASES): Input dimension mismatch for argument 3")
        A3 = A if A.ndim == 3 else A[:, :, None]
        lbA, ubA = _col(lbA, nC, "lbA"), _col(ubA, nC, "ubA")

    def bcast(a):
        return np.broadcast_to(a, (a.shape[0], B)) if a.shape[1] != B else a

    g, lb, ub, lbA, ubA = (bcast(a) for a in (g, lb, ub, lbA, ubA))
    # instance-major stacking, each instance column-major
    Hs = np.ascontiguousarray(np.transpose(H if batched_HA else H[:, :, None], (2, 1, 0)))
    As = np.ascontiguousarray(np.transpose(A3, (2, 1, 0)))
    vec = lambda a: np.ascontiguousarray(a.T)
    gs, lbs, ubs, lbAs, ubAs = vec(g), vec(lb), vec(ub), vec(lbA), vec(ubA)
    x = np.zeros((B, nV)); fval = np.zeros(B); flag = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
    lam = np.zeros((B, nV + nC))
    desc = QpDesc(nV, nC, B, 0 if batched_HA else 1)
    opts = options if options is not None else default_opts()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().fsaempc_qp_solve_batch(C.byref(desc), p(Hs), p(gs), p(As), p(lbs), p(ubs), p(lbAs), p(ubAs), C.byref(opts),
                                      p(x), p(fval), p(flag), p(it), p(lam))
    check(rc, "qpOASES")
    # working sets in the reference's encoding (qpOASES.m:58-61), derived from the multipliers
    ws = np.sign(lam) * -1.0
    ws[np.abs(lam) <= 1e-9 * np.maximum(1.0, np.abs(lam).max(axis=1, keepdims=True))] = 0.0
    aux = {"workingSetB": ws[:, :nV].T.squeeze(), "workingSetC": ws[:, nV:].T.squeeze(), "cpuTime": None}
    if B == 1:
        return x[0], float(fval[0]), int(flag[0]), int(it[0]), lam[0], aux
    return x.T, fval, flag, it, lam.T, aux


def qp_solve_batch_device(H, g, A, lb, ub, lbA, ubA, options=None, want_lambda=False, workspace=None, stream=None,
                          shared_HA=False, want_aux=False, x_init=None, difficulty=None):
    """Device-resident batched solve on torch CUDA(HIP) tensors (instance-major, each instance column-major):
    H (B,nV,nV), g (B,nV), A (B,nV,nC) [memory of a column-major nC x nV matrix], lb/ub (B,nV), lbA/ubA (B,nC).
    Asynchronous on `stream` (default: torch's current stream).  Returns dict of device tensors; want_aux adds `kkt`
    (relative KKT residual of the returned point) and `polished` (> 0: the active-set refinement reached the vertex); x_init (B,nV):
    optional starting point of the interior-point iteration (clamped to the bounds)."""
    import torch
    B, nV = g.shape
    nC = lbA.shape[1] if lbA is not None else 0
    dev = g.device
    for t in (H, g, A, lb, ub, lbA, ubA):
        if t is not None and (t.dtype != torch.float64 or not t.is_contiguous() or not t.is_cuda):
            raise ValueError("device tensors must be contiguous float64 on the GPU")
    desc = QpDesc(nV, nC, B, 1 if shared_HA else 0)
    need = lib().fsaempc_qp_workspace_bytes(C.byref(desc))
    if need < 0:
        check(int(need), "fsaempc_qp_workspace_bytes")
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need + 7) // 8, dtype=torch.float64, device=dev)
    x = torch.empty((B, nV), dtype=torch.float64, device=dev)
    fval = torch.empty(B, dtype=torch.float64, device=dev)
    flag = torch.empty(B, dtype=torch.int32, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    lam = torch.empty((B, nV + nC), dtype=torch.float64, device=dev) if want_lambda else None
    opts = options if options is not None else default_opts()
    st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    kkt = torch.empty(B, dtype=torch.float64, device=dev) if want_aux else None
    pol = torch.empty(B, dtype=torch.int32, device=dev) if want_aux else None
    if x_init is not None and (x_init.dtype != torch.float64 or not x_init.is_contiguous() or not x_init.is_cuda or tuple(x_init.shape) != (B, nV)):
        raise ValueError("x_init must be a contiguous float64 (B, nV) tensor on the GPU")
    if difficulty is not None and (difficulty.dtype != torch.int32 or not difficulty.is_contiguous() or not difficulty.is_cuda or tuple(difficulty.shape) != (B,)):
        raise ValueError("difficulty must be a contiguous int32 (B,) tensor on the GPU")
    aux = QpAux(P(kkt), P(pol), P(x_init), P(difficulty))   # per-instance diagnostics (the analogue of qpOASES' auxOutput) + optional starting point
    rc = lib().fsaempc_qp_solve_batch_device_aux(C.byref(desc), P(H), P(g), P(A), P(lb), P(ub), P(lbA), P(ubA), C.byref(opts),
                                                 P(x), P(fval), P(flag), P(it), P(lam), C.byref(aux), P(workspace),
                                                 C.c_longlong(workspace.numel() * 8), C.c_void_p(st))
    check(rc, "fsaempc_qp_solve_batch_device_aux")
    return dict(x=x, fval=fval, exitflag=flag, iter=it, lam=lam, workspace=workspace, kkt=kkt, polished=pol)


def qpOASES_sequence(cmd, *args, options=None, aux=False):
    """Mirror of qpOASES_sequence (optimizers/matlab/qpOASES/qpOASES_sequence.m):
      [QP,x,fval,exitflag,iter,lambda,auxOutput] = qpOASES_sequence('i', H,g,A,lb,ub,lbA,ubA)     (:23)
      [QP,x,fval,exitflag,iter,lambda,auxOutput] = qpOASES_sequence('i', H,g,lb,ub)               (:25, bounds-only)
      [x,fval,exitflag,iter,lambda,auxOutput]    = qpOASES_sequence('h', QP, g,lb,ub,lbA,ubA)     (:39)
      [x,fval,exitflag,iter,lambda,auxOutput]    = qpOASES_sequence('h', QP, g,lb,ub)             (:41, bounds-only)
      [x,fval,exitflag,iter,lambda,auxOutput]    = qpOASES_sequence('m', QP, H,g,A,lb,ub,lbA,ubA) (:51)
      [x,lambda,workingSetB,workingSetC]         = qpOASES_sequence('e', QP, g,lb,ub{,lbA,ubA})   (:64)
                                                   qpOASES_sequence('c', QP)                      (:76)
    g, lb, ub, lbA, ubA may carry k columns ((nV, k) / (nC, k) arrays) in 'h' and 'e' -- k QPs sharing the handle's H and A; x
    and lambda then come back with k columns, fval / exitflag / iter as length-k arrays.  auxOutput (the optional last output of
    the reference) is appended when aux=True: dict(workingSetB, workingSetC) from the sign of the multipliers (qpOASES.m:58-61).  The handle owns device copies of H and A, the workspace
    and the per-call vectors; every call is a cold solve."""
    L = lib()
    opts = options if options is not None else default_opts()
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    colmajor = lambda M: np.ascontiguousarray(np.asarray(M, dtype=np.float64).T)

    def vec(a, rows, k):   # (rows,) or (rows, k) -> k stacked columns (one column is broadcast)
        if a is None or rows == 0:
            return None
        a = np.asarray(a, dtype=np.float64)
        a = a.reshape(rows, -1) if a.ndim > 1 or a.size != rows else a.reshape(rows, 1)
        if a.shape[1] not in (1, k):
            raise ValueError("ERROR (qpOASES): Input dimension mismatch")
        return np.ascontiguousarray(np.broadcast_to(a, (rows, k)).T)

    def ncols(g, nV):
        g = np.asarray(g, dtype=np.float64)
        return g.shape[1] if g.ndim == 2 and g.shape[0] == nV else 1

    def outs(nV, nC, k):
        return np.zeros((k, nV)), np.zeros(k), np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32), np.zeros((k, nV + nC))

    def pack(nV, nC, k, x, fv, fl, it, lam):
        aux = dict(workingSetB=np.where(lam[:, :nV] > 0, -1, np.where(lam[:, :nV] < 0, 1, 0)).T,
                   workingSetC=np.where(lam[:, nV:] > 0, -1, np.where(lam[:, nV:] < 0, 1, 0)).T)
        if k == 1:
            aux = {k_: v[:, 0] for k_, v in aux.items()}
            return x[0], float(fv[0]), int(fl[0]), int(it[0]), lam[0], aux
        return x.T, fv, fl, it, lam.T, aux

    if cmd == "i":
        if len(args) == 7:
            H, g, A, lb, ub, lbA, ubA = args
        elif len(args) == 4:
            (H, g, lb, ub), A, lbA, ubA = args, None, None, None
        else:
            raise ValueError("ERROR (qpOASES): Invalid number of input arguments!")
        H = np.asarray(H, dtype=np.float64)
        nV = H.shape[0]
        A = np.asarray(A, dtype=np.float64).reshape(-1, nV) if A is not None else np.zeros((0, nV))
        nC = A.shape[0]
        k = ncols(g, nV)
        x, fv, fl, it, lam = outs(nV, nC, k)
        h = C.c_int(0)
        check(L.fsaempc_seq_init(nV, nC, p(colmajor(H)), p(vec(g, nV, k)), p(colmajor(A)) if nC else None, p(vec(lb, nV, k)), p(vec(ub, nV, k)),
                                 p(vec(lbA, nC, k)), p(vec(ubA, nC, k)), k, C.byref(opts), C.byref(h), p(x), p(fv), p(fl), p(it), p(lam)), "qpOASES_sequence('i')")
        _SEQ_DIMS[h.value] = (nV, nC)
        return (h.value,) + pack(nV, nC, k, x, fv, fl, it, lam)[:6 if aux else 5]
    if cmd in ("h", "m"):
        QP = int(args[0])
        nV, nC = _SEQ_DIMS.get(QP, (1, 0))
        if cmd == "h":
            if len(args) == 6:
                g, lb, ub, lbA, ubA = args[1:]
            elif len(args) == 4:
                (g, lb, ub), lbA, ubA = args[1:], None, None
            else:
                raise ValueError("ERROR (qpOASES): Invalid number of input arguments!")
            k = ncols(g, nV)
            x, fv, fl, it, lam = outs(nV, nC, k)
            rc = L.fsaempc_seq_hotstart(QP, nV, nC, p(vec(g, nV, k)), p(vec(lb, nV, k)), p(vec(ub, nV, k)), p(vec(lbA, nC, k)), p(vec(ubA, nC, k)), k,
                                        C.byref(opts), p(x), p(fv), p(fl), p(it), p(lam))
        else:
            H, g, A, lb, ub, lbA, ubA = args[1:]
            H = np.asarray(H, dtype=np.float64); A = np.asarray(A, dtype=np.float64).reshape(-1, H.shape[0])
            nVm, nCm = H.shape[0], A.shape[0]
            k = ncols(g, nVm)
            x, fv, fl, it, lam = outs(nVm, nCm, k)
            rc = L.fsaempc_seq_hotstart_matrices(QP, nVm, nCm, p(colmajor(H)), p(vec(g, nVm, k)), p(colmajor(A)), p(vec(lb, nVm, k)), p(vec(ub, nVm, k)),
                                                 p(vec(lbA, nCm, k)), p(vec(ubA, nCm, k)), k, C.byref(opts), p(x), p(fv), p(fl), p(it), p(lam))
        check(rc, "qpOASES_sequence('%s')" % cmd)
        return pack(nV, nC, k, x, fv, fl, it, lam)[:6 if aux else 5]
    if cmd == "e":
        QP = int(args[0])
        nV, nC = _SEQ_DIMS.get(QP, (1, 0))
        if len(args) == 6:
            g, lb, ub, lbA, ubA = args[1:]
        elif len(args) == 4:
            (g, lb, ub), lbA, ubA = args[1:], None, None
        else:
            raise ValueError("ERROR (qpOASES): Invalid number of input arguments!")
        k = ncols(g, nV)
        x, lam = np.zeros((k, nV)), np.zeros((k, nV + nC))
        wb, wc = np.zeros(nV, dtype=np.int32), np.zeros(max(nC, 1), dtype=np.int32)
        check(L.fsaempc_seq_equality(QP, nV, nC, p(vec(g, nV, k)), p(vec(lb, nV, k)), p(vec(ub, nV, k)), p(vec(lbA, nC, k)), p(vec(ubA, nC, k)), k,
                                     C.byref(opts), p(x), p(lam), p(wb), p(wc)), "qpOASES_sequence('e')")
        return (x[0], lam[0], wb, wc[:nC]) if k == 1 else (x.T, lam.T, wb, wc[:nC])
    if cmd == "c":
        check(L.fsaempc_seq_cleanup(int(args[0])), "qpOASES_sequence('c')")
        _SEQ_DIMS.pop(int(args[0]), None)
        return None
    raise ValueError("ERROR (qpOASES): unknown command '%s'" % cmd)


_SEQ_DIMS = {}
