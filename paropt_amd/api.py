"""
Python host harness over the C ABI (paropt_amd/lib.py).  Class and method names follow the
reference's Python layer (paropt/ParOpt.pyx:761-1521: PVec, LBFGS, LSR1, Problem,
InteriorPoint) so tests read like the reference's own; all arithmetic happens in
libparopt_amd.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import lib as L
from .lib import check, lib


def live_objects():
    """(device vectors alive in this process, HBM bytes behind them): the leak check of the C ABI."""
    a, b = C.c_int64(), C.c_int64()
    check(lib.po_live_objects(C.byref(a), C.byref(b)))
    return a.value, b.value


def live_host_mirrors():
    """Pinned host mirrors (getArray) alive in this process."""
    a = C.c_int64()
    check(lib.po_live_host_mirrors(C.byref(a)))
    return a.value


class Context:
    """One per process / GPU: HIP stream + communicator (replaces the MPI communicator)."""

    def __init__(self, device=0):
        self._h = L.po_ctx()
        check(lib.po_ctx_create(int(device), C.byref(self._h)))
        self._keep = []

    @property
    def handle(self):
        return self._h

    def rank_size(self):
        r, s = C.c_int(), C.c_int()
        check(lib.po_ctx_rank(self._h, C.byref(r), C.byref(s)))
        return r.value, s.value

    def synchronize(self):
        check(lib.po_ctx_synchronize(self._h))

    def time_mdot(self, nvecs):
        """Bracket every mdot launch of exactly `nvecs` vectors with HIP events (0: off); resets the totals."""
        check(lib.po_ctx_time_mdot(self._h, int(nvecs)))

    def time_mdot_result(self):
        """(accumulated kernel milliseconds, launches) since time_mdot()."""
        ms, cnt = C.c_double(), C.c_int64()
        check(lib.po_ctx_time_mdot_result(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def time_wgram(self, on):
        """Bracket every weighted-Gram launch with HIP events (resets the totals)."""
        check(lib.po_ctx_time_wgram(self._h, int(bool(on))))

    def time_wgram_result(self, which):
        """(kernel ms, launches, panel width, algorithmic bytes) of the timed weighted-Gram launches;
        which = 0 plain launches, 1 launches that also form the L-SR1 columns."""
        ms, cnt, cols, byt = C.c_double(), C.c_int64(), C.c_int(), C.c_double()
        check(lib.po_ctx_time_wgram_result(self._h, int(which), C.byref(ms), C.byref(cnt), C.byref(cols),
                                           C.byref(byt)))
        return ms.value, cnt.value, cols.value, byt.value

    def comm_info(self):
        """(communicator kind: 0 self / 1 RCCL / 2 host callback, ncclAllReduce calls, ncclAllGather calls)."""
        k, a, b = C.c_int(), C.c_int64(), C.c_int64()
        check(lib.po_ctx_comm_info(self._h, C.byref(k), C.byref(a), C.byref(b)))
        return k.value, a.value, b.value

    def set_reduction_batching(self, on):
        """Let independent reductions share one collective + host sync (default on)."""
        check(lib.po_ctx_set_reduction_batching(self._h, int(bool(on))))
        return self

    def batched_reductions(self):
        """Reductions that shared another reduction's collective + host sync so far."""
        a = C.c_int64()
        check(lib.po_ctx_batched_reductions(self._h, C.byref(a)))
        return a.value

    def counters(self):
        """(host-synchronising reductions, kernel launches) issued on this context so far."""
        a, b = C.c_int64(), C.c_int64()
        check(lib.po_ctx_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sync_counters(self):
        """{flag_waits, flag_timeouts, allreduces, allgathers}: the polled completions of reductions, those that fell back
        to the stream synchronisation after the bounded spin, and the RCCL collectives issued (po_ctx_sync_counters)."""
        v = [C.c_int64() for _ in range(4)]
        check(lib.po_ctx_sync_counters(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("flag_waits", "flag_timeouts", "allreduces", "allgathers"), (x.value for x in v)))

    def bench_collective(self, count, pure_sum=True, reps=50):
        """{median, min, max} host microseconds of one reduction exchange of `count` doubles (collective)."""
        out = (C.c_double * 3)()
        check(lib.po_ctx_bench_collective(self._h, int(count), 1 if pure_sum else 0, int(reps), out))
        return {"median_us": out[0], "min_us": out[1], "max_us": out[2], "count": int(count),
                "form": "allreduce" if pure_sum else "allgather+host combine"}

    def allreduce(self, values, op="sum"):
        """MPI_Allreduce counterpart on host values (numpy float64 array, in place)."""
        import numpy as np

        a = np.ascontiguousarray(values, dtype=np.float64)
        check(lib.po_ctx_allreduce(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size,
                                   {"sum": 0, "min": 1, "max": 2}[op]))
        return a

    def algorithmic_bytes(self):
        """(total, issued from inside problem callbacks): algorithmic HBM bytes of the n-sized launches so far."""
        a, b = C.c_double(), C.c_double()
        check(lib.po_ctx_algorithmic_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def init_rccl_from_torch(self):
        """One process per GPU: ship the RCCL unique id over torch.distributed, then init."""
        import torch.distributed as dist

        rank, size = dist.get_rank(), dist.get_world_size()
        if size == 1:
            return

        def _all_ok(flag, what):
            # every rank takes the same decision, so a failure on one rank cannot leave the others
            # waiting inside a collective
            flags = [None] * size
            dist.all_gather_object(flags, bool(flag))
            if not all(flags):
                raise RuntimeError("%s failed on rank(s) %s" % (what, [r for r, f in enumerate(flags) if not f]))

        # 1. librccl loads and hands out an id on EVERY rank (dlopen + symbols), before any collective
        buf = (C.c_char * 128)()
        _all_ok(lib.po_rccl_unique_id(buf) == 0, "loading RCCL")
        # 2. rank 0's id to everyone, collective init, and agreement that it worked
        obj = [bytes(buf) if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0)
        idbuf = (C.c_char * 128).from_buffer_copy(obj[0])
        _all_ok(lib.po_ctx_comm_init_rccl(self._h, rank, size, idbuf) == 0, "ncclCommInitRank")

    def init_callback_from_torch(self, device=None):
        """Host-side allgather through torch.distributed (gloo, or nccl with `device`): the
        maintainer's-MPI hook."""
        import torch
        import torch.distributed as dist

        rank, size = dist.get_rank(), dist.get_world_size()

        def _gather(inp, out, count, user):
            try:
                loc = torch.from_numpy(np.ctypeslib.as_array(inp, shape=(count,)).copy())
                if device is not None:
                    loc = loc.to(device)
                parts = [torch.empty_like(loc) for _ in range(size)]
                dist.all_gather(parts, loc)
                dst = np.ctypeslib.as_array(out, shape=(count * size,))
                for r in range(size):
                    dst[r * count : (r + 1) * count] = parts[r].cpu().numpy()
                return 0
            except Exception:  # pragma: no cover
                return 1

        cb = L.ALLGATHER_FN(_gather)
        self._keep.append(cb)
        check(lib.po_ctx_comm_init_callback(self._h, rank, size, cb, None))

    def close(self):
        """Destroy the context.  Objects created from it must not be used afterwards (their
        destructors become no-ops: the device memory went with the context's process)."""
        if self._h:
            lib.po_ctx_destroy(self._h)
            self._h = None


class PVec:
    """ParOptVec in HBM (reference: paropt/ParOpt.pyx PVec, src/ParOptVec.h:53-70)."""

    def __init__(self, ctx, n=None, handle=None, owned=True):
        self.ctx = ctx
        if handle is None:
            h = L.po_vec()
            check(lib.po_vec_create(ctx.handle, int(n), C.byref(h)))
            self._h = h
            self._owned = True
        else:
            self._h = handle if isinstance(handle, C.c_void_p) else L.po_vec(handle)
            self._owned = owned

    @property
    def handle(self):
        return self._h

    def __len__(self):
        n = C.c_int64()
        check(lib.po_vec_size(self._h, C.byref(n)))
        return n.value

    def __del__(self):
        try:
            if self._owned and self._h and self.ctx._h:
                lib.po_vec_decref(self._h)
        except Exception:
            pass

    # -- the 11 virtuals of ParOptVec --------------------------------------------------------
    def set(self, alpha):
        check(lib.po_vec_set(self._h, float(alpha)))

    def zeroEntries(self):
        check(lib.po_vec_zero(self._h))

    def copyValues(self, other):
        check(lib.po_vec_copy(self._h, other.handle))

    def norm(self):
        out = C.c_double()
        check(lib.po_vec_norm(self._h, C.byref(out)))
        return out.value

    def maxabs(self):
        out = C.c_double()
        check(lib.po_vec_maxabs(self._h, C.byref(out)))
        return out.value

    def l1norm(self):
        out = C.c_double()
        check(lib.po_vec_l1norm(self._h, C.byref(out)))
        return out.value

    def dot(self, other):
        out = C.c_double()
        check(lib.po_vec_dot(self._h, other.handle, C.byref(out)))
        return out.value

    def mdot(self, vecs):
        nv = len(vecs)
        out = np.zeros(max(nv, 1))
        arr = (L.po_vec * max(nv, 1))(*[v.handle for v in vecs])
        check(lib.po_vec_mdot(self._h, arr, nv, out.ctypes.data_as(L.c_double_p)))
        return out[:nv]

    def scale(self, alpha):
        check(lib.po_vec_scale(self._h, float(alpha)))

    def axpy(self, alpha, other):
        check(lib.po_vec_axpy(self._h, float(alpha), other.handle))

    def maxpy(self, beta, alphas, vecs):
        nv = len(vecs)
        a = np.ascontiguousarray(alphas, dtype=np.float64)
        arr = (L.po_vec * max(nv, 1))(*[v.handle for v in vecs])
        check(lib.po_vec_maxpy(self._h, float(beta), a.ctypes.data_as(L.c_double_p), arr, nv))

    def getArray(self):
        """The vector's data as a numpy view of its pinned host mirror, LIVE as in the reference (ParOptVec::getArray):
        writes through the view are seen by every later operation on the vector and results of operations show
        up in the view (po_vec_get_array); releaseArray() ends that state."""
        p = L.c_double_p()
        check(lib.po_vec_get_array(self._h, C.byref(p)))
        n = len(self)
        if n == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(p, shape=(n,))

    def _peek(self):
        p = L.c_double_p()
        check(lib.po_vec_peek_array(self._h, C.byref(p)))
        n = len(self)
        return np.ctypeslib.as_array(p, shape=(n,)) if n else np.zeros(0)

    def releaseArray(self, upload=True):
        """End the live state of the host mirror getArray() handed out (final upload unless upload=False).
        Required for vectors the solver owns before the solver runs on (po_vec_release_array)."""
        check(lib.po_vec_release_array(self._h, int(bool(upload))))

    def syncToDevice(self):
        check(lib.po_vec_sync_to_device(self._h))

    def syncToHost(self):
        check(lib.po_vec_sync_to_host(self._h))

    # -- conveniences --------------------------------------------------------------------------
    def to_numpy(self):
        return np.array(self._peek(), copy=True)

    def from_numpy(self, arr):
        a = self._peek()
        a[:] = arr
        self.syncToDevice()
        return self

    def fill_hash(self, seed, aid, offset=0, scale=1.0, shift=0.0):
        check(lib.po_vec_fill_hash(self._h, int(seed), int(aid), int(offset), float(scale), float(shift)))
        return self

    def device_ptr(self):
        p = L.c_double_p()
        check(lib.po_vec_get_device_array(self._h, C.byref(p)))
        return C.cast(p, C.c_void_p).value


class _QuasiNewton:
    def __init__(self, ctx, kind, n, subspace, handle=None):
        self.ctx = ctx
        if handle is None:
            self._h = L.po_qn()
            check(lib.po_qn_create(ctx.handle, kind, int(n), int(subspace), C.byref(self._h)))
            self._owned = True
        else:
            self._h = handle
            self._owned = False

    def __del__(self):
        try:
            if self._owned and self._h and self.ctx._h:
                lib.po_qn_destroy(self._h)
        except Exception:
            pass

    def reset(self):
        check(lib.po_qn_reset(self._h))

    def update(self, s, y):
        rc = C.c_int()
        check(lib.po_qn_update(self._h, s.handle, y.handle, C.byref(rc)))
        return rc.value

    def mult(self, x, y):
        check(lib.po_qn_mult(self._h, x.handle, y.handle))

    def multAdd(self, alpha, x, y):
        check(lib.po_qn_mult_add(self._h, float(alpha), x.handle, y.handle))

    def setInitDiagonalType(self, t):
        check(lib.po_qn_set_diag_type(self._h, 1 if t in (1, "yts_over_sts") else 0))

    def debugLoad(self, b0, B, Lm, D, S, Y):
        """po_qn_debug_load: take over a complete limited-memory state (test hook).  B, Lm are (ld x ld) arrays in the
        reference's column-major storage (flat), D has ld entries, S / Y are lists of numpy vectors of the local size."""
        msub = len(S)
        ld = len(D)
        Bf = np.ascontiguousarray(B, dtype=np.float64).ravel()
        Lf = np.ascontiguousarray(Lm, dtype=np.float64).ravel()
        Df = np.ascontiguousarray(D, dtype=np.float64).ravel()
        sv = [PVec(self.ctx, len(v)).from_numpy(v) for v in S]
        yv = [PVec(self.ctx, len(v)).from_numpy(v) for v in Y]
        sh = (L.po_vec * max(msub, 1))(*[v.handle.value for v in sv])
        yh = (L.po_vec * max(msub, 1))(*[v.handle.value for v in yv])
        check(lib.po_qn_debug_load(self._h, msub, float(b0), Bf.ctypes.data_as(L.c_double_p),
                                   Lf.ctypes.data_as(L.c_double_p), Df.ctypes.data_as(L.c_double_p), ld, sh, yh))

    def getCompactMat(self):
        k, b0 = C.c_int(), C.c_double()
        d0, M, Z = L.c_double_p(), L.c_double_p(), L.vec_p()
        check(lib.po_qn_get_compact(self._h, C.byref(k), C.byref(b0), C.byref(d0), C.byref(M), C.byref(Z)))
        n = k.value
        d = np.array([d0[i] for i in range(n)])
        Mm = np.array([M[i] for i in range(n * n)]).reshape(n, n).T if n else np.zeros((0, 0))
        Zs = [PVec(self.ctx, handle=L.po_vec(Z[i]), owned=False) for i in range(n)]
        return b0.value, d, Mm, Zs


class LBFGS(_QuasiNewton):
    def __init__(self, ctx, n, subspace=10, update_type="skip_negative_curvature"):
        super().__init__(ctx, 0, n, subspace)
        check(lib.po_qn_set_update_type(self._h, 1 if update_type in (1, "damped_update", "damped") else 0))


class LSR1(_QuasiNewton):
    def __init__(self, ctx, n, subspace=10):
        super().__init__(ctx, 1, n, subspace)


class Problem:
    """Base class for user problems implemented in Python (host arrays through getArray).

    Mirrors paropt.ParOpt.Problem: override getVarsAndBounds(x, lb, ub), evalObjCon(x) ->
    (fail, fobj, con) and evalObjConGradient(x, g, A) -> fail, all on numpy views.  With
    nwcon > 0 (sparse constraints, nwblock = 1) also override evalSparseCon(x, out),
    addSparseJacobian(alpha, x, px, out), addSparseJacobianTranspose(alpha, x, pzw, out) and
    addSparseInnerProduct(alpha, x, cvec, A) (A is the w-sized diagonal; with nwblock > 1 the packed upper
    triangles of the nwblock x nwblock blocks, nwcon (nwblock+1)/2 entries), all in place on numpy views.

    With rowp / cols (the reference's CSR form, paropt/ParOpt.pyx:849-881) override instead
    evalSparseObjCon(x, sparse_cons) -> (fail, fobj, con) and evalSparseObjConGradient(x, g, A, data) -> fail;
    sparse_cons (nwcon) and data (nnz, in the order of cols) are filled in place.
    """

    def __init__(self, ctx, nvars, ncon, ninequality=-1, nwcon=0, nwinequality=0, use_lower=True,
                 use_upper=True, rowp=None, cols=None, nwblock=1):
        self.ctx = ctx
        self.nvars, self.ncon = int(nvars), int(ncon)
        self.nwcon = int(nwcon)
        self._csr = rowp is not None and cols is not None
        if self._csr and len(rowp) != self.nwcon + 1:
            raise ValueError("rowp is incorrect length")  # paropt/ParOpt.pyx:861-862
        cb = L.ProblemCallbacks()
        self._pending_exc = None

        def _guard(fn):
            # A Python exception must not look like success to the solver: it is kept (the first one), the
            # callback reports failure through its non-zero return code, and optimize() re-raises it.
            def _g(*args):
                if self._pending_exc is not None:
                    return 1  # fail fast until optimize() has re-raised the first exception
                try:
                    return fn(*args)
                except BaseException as e:  # noqa: BLE001 - re-raised by _raise_pending()
                    if self._pending_exc is None:
                        self._pending_exc = e
                    return 1
            return _g

        @_guard
        def _gvb(user, x, lb, ub):
            vx, vl, vu = (PVec(ctx, handle=L.po_vec(h), owned=False) for h in (x, lb, ub))
            ax, al, au = vx.getArray(), vl.getArray(), vu.getArray()
            fail = self.getVarsAndBounds(ax, al, au)
            vx.releaseArray(True), vl.releaseArray(True), vu.releaseArray(True)
            return int(fail or 0)

        @_guard
        def _eval(user, x, fobj, cons):
            vx = PVec(ctx, handle=L.po_vec(x), owned=False)
            fail, f, con = self.evalObjCon(vx.to_numpy())
            fobj[0] = float(f)
            for j in range(self.ncon):
                cons[j] = float(con[j])
            return int(fail)

        @_guard
        def _grad(user, x, g, Ac):
            vx = PVec(ctx, handle=L.po_vec(x), owned=False)
            vg = PVec(ctx, handle=L.po_vec(g), owned=False)
            # Ac is NULL when the problem declared linear constraints and only the objective gradient is wanted
            va = [PVec(ctx, handle=L.po_vec(Ac[j]), owned=False) for j in range(self.ncon)] if Ac else []
            ag = vg.getArray()
            aa = [v.getArray() for v in va] if Ac else None
            fail = self.evalObjConGradient(vx.to_numpy(), ag, aa)
            vg.releaseArray(True)
            for v in va:
                v.releaseArray(True)
            return int(fail or 0)

        self._cbs = (L.GET_VARS_FN(_gvb), L.EVAL_FN(_eval), L.GRAD_FN(_grad))
        cb.user = None
        cb.get_vars_and_bounds, cb.eval_obj_con, cb.eval_obj_con_gradient = self._cbs
        cb.qn_update_correction = L.QNCORR_FN()
        cb.write_output = L.WRITE_FN()
        self._cb_struct = cb
        self._h = L.po_problem()
        check(lib.po_problem_create_callbacks(ctx.handle, self.nvars, self.ncon, int(ninequality),
                                              C.byref(cb), C.byref(self._h)))
        if not (use_lower and use_upper):
            check(lib.po_problem_set_var_bound_options(self._h, int(bool(use_lower)), int(bool(use_upper))))
        # optional second-order callbacks (use_hvec_product / use_diag_hessian): evalHvecProduct(x, z, zw,
        # px, hvec) and evalHessianDiag(x, z, zw, hdiag) on numpy views, as in paropt.ParOpt
        has_hvec, has_hdiag = hasattr(self, "evalHvecProduct"), hasattr(self, "evalHessianDiag")
        if has_hvec or has_hdiag:
            def _views(x, z, zw):
                vx = PVec(ctx, handle=L.po_vec(x), owned=False)
                za = np.array([z[j] for j in range(self.ncon)])
                zwa = PVec(ctx, handle=L.po_vec(zw), owned=False).to_numpy() if zw else None
                return vx.to_numpy(), za, zwa

            @_guard
            def _hvec(user, x, z, zw, px, hvec):
                xa, za, zwa = _views(x, z, zw)
                vp = PVec(ctx, handle=L.po_vec(px), owned=False)
                vh = PVec(ctx, handle=L.po_vec(hvec), owned=False)
                ah = vh.getArray()
                fail = self.evalHvecProduct(xa, za, zwa, vp.to_numpy(), ah)
                vh.releaseArray(True)
                return int(fail or 0)

            @_guard
            def _hdiag(user, x, z, zw, hdiag):
                xa, za, zwa = _views(x, z, zw)
                vh = PVec(ctx, handle=L.po_vec(hdiag), owned=False)
                ah = vh.getArray()
                fail = self.evalHessianDiag(xa, za, zwa, ah)
                vh.releaseArray(True)
                return int(fail or 0)

            self._hcbs = (L.HVEC_FN(_hvec) if has_hvec else L.HVEC_FN(), L.HDIAG_FN(_hdiag) if has_hdiag else L.HDIAG_FN())
            check(lib.po_problem_set_hessian_callbacks(self._h, self._hcbs[0], self._hcbs[1]))
        if self._csr:
            self._rowp = np.ascontiguousarray(rowp, dtype=np.intc)
            self._cols = np.ascontiguousarray(cols, dtype=np.intc)
            nnz = int(self._rowp[-1])
            if len(self._cols) != nnz:
                raise ValueError("cols is incorrect length")
            self._data_host = np.zeros(max(nnz, 1))

            @_guard
            def _sobjcon(user, x, fobj, cons, sparse):
                vx = PVec(ctx, handle=L.po_vec(x), owned=False)
                vs = PVec(ctx, handle=L.po_vec(sparse), owned=False)
                asp = vs.getArray()
                fail, f, con = self.evalSparseObjCon(vx.to_numpy(), asp)
                vs.releaseArray(True)
                fobj[0] = float(f)
                for j in range(self.ncon):
                    cons[j] = float(con[j])
                return int(fail or 0)

            @_guard
            def _sgrad(user, x, g, Ac, data, nnz_):
                vx = PVec(ctx, handle=L.po_vec(x), owned=False)
                vg = PVec(ctx, handle=L.po_vec(g), owned=False)
                va = [PVec(ctx, handle=L.po_vec(Ac[j]), owned=False) for j in range(self.ncon)]
                ag = vg.getArray()
                aa = [v.getArray() for v in va]
                fail = self.evalSparseObjConGradient(vx.to_numpy(), ag, aa, self._data_host[:nnz])
                vg.releaseArray(True)
                for v in va:
                    v.releaseArray(True)
                check(lib.po_ctx_memcpy(ctx.handle, data, self._data_host.ctypes.data, 8 * nnz, 1))
                return int(fail or 0)

            self._csr_cbs = (L.SPARSE_OBJCON_FN(_sobjcon), L.SPARSE_GRAD_FN(_sgrad))
            check(lib.po_problem_set_sparse_jacobian_data(
                self._h, self.nwcon, int(nwinequality), self._rowp.ctypes.data_as(L.c_int_p),
                self._cols.ctypes.data_as(L.c_int_p), self._csr_cbs[0], self._csr_cbs[1]))
        elif self.nwcon > 0:
            def _wrap(method):
                @_guard
                def _f(user, alpha, x, v, out):
                    vx = PVec(ctx, handle=L.po_vec(x), owned=False)
                    vv = PVec(ctx, handle=L.po_vec(v), owned=False)
                    vo = PVec(ctx, handle=L.po_vec(out), owned=False)
                    ao = vo.getArray()
                    fail = method(float(alpha), vx.to_numpy(), vv.to_numpy(), ao)
                    vo.releaseArray(True)
                    return int(fail or 0)
                return _f

            @_guard
            def _wcon(user, x, out):
                vx = PVec(ctx, handle=L.po_vec(x), owned=False)
                vo = PVec(ctx, handle=L.po_vec(out), owned=False)
                ao = vo.getArray()
                fail = self.evalSparseCon(vx.to_numpy(), ao)
                vo.releaseArray(True)
                return int(fail or 0)

            scb = L.ProblemSparseCallbacks()
            self._scbs = (L.SPARSE_CON_FN(_wcon), L.SPARSE_JAC_FN(_wrap(self.addSparseJacobian)),
                          L.SPARSE_JAC_FN(_wrap(self.addSparseJacobianTranspose)),
                          L.SPARSE_JAC_FN(_wrap(self.addSparseInnerProduct)))
            (scb.eval_sparse_con, scb.add_sparse_jacobian, scb.add_sparse_jacobian_transpose,
             scb.add_sparse_inner_product) = self._scbs
            self._scb_struct = scb
            check(lib.po_problem_set_sparse_callbacks(self._h, self.nwcon, int(nwinequality), C.byref(scb)))
            if int(nwblock) > 1:  # addSparseInnerProduct then fills packed upper nwblock x nwblock blocks
                check(lib.po_problem_set_sparse_block_size(self._h, int(nwblock)))

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib.po_problem_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _raise_pending(self):
        """Re-raise the first exception a callback threw during the last solver call."""
        e, self._pending_exc = self._pending_exc, None
        if e is not None:
            raise e

    def setLinearConstraints(self, flag=True):
        """Declare the dense constraints linear: after the first gradient evaluation of an optimize() call
        evalObjConGradient is called with A = None (objective gradient only)."""
        check(lib.po_problem_set_linear_constraints(self._h, int(bool(flag))))
        return self

    @property
    def handle(self):
        return self._h


class SeparableProblem:
    """Built-in device-resident workloads: 'quadratic', 'convex', 'rosenbrock'."""

    KINDS = {"quadratic": 0, "convex": 1, "rosenbrock": 2}

    def __init__(self, ctx, kind, n, c=2, seed=0, eig_min=1.0, eig_max=100.0):
        self.ctx = ctx
        self._h = L.po_problem()
        check(lib.po_problem_create_separable(ctx.handle, self.KINDS[kind], int(n), int(c), int(seed),
                                              float(eig_min), float(eig_max), C.byref(self._h)))
        nl, off, nc = C.c_int64(), C.c_int64(), C.c_int()
        check(lib.po_problem_sizes(self._h, C.byref(nl), C.byref(off), C.byref(nc)))
        self.nvars, self.offset, self.ncon = nl.value, off.value, nc.value
        self.nwcon = 0

    def setWeighting(self, nwcon, nw, nwstart=0, nwskip=0, nwinequality=None):
        """cw_i = 1 - sum_{k<nw} x[nwstart + i (nw + nwskip) + k], i < nwcon (global indices);
        the first nwinequality of them are inequalities (all by default)."""
        if nwinequality is None:
            nwinequality = nwcon
        check(lib.po_problem_set_weighting(self._h, int(nwcon), int(nw), int(nwstart), int(nwskip),
                                           int(nwinequality)))
        a, b = C.c_int64(), C.c_int64()
        check(lib.po_problem_sparse_sizes(self._h, C.byref(a), C.byref(b)))
        self.nwcon = a.value
        return self

    def setChain(self, span=2, stride=1, reverse_cols=False):
        """Rank-local overlapping sparse constraints in CSR form: cw_i = 1 - sum_{k<span} x[i*stride+k]^2 >= 0
        (examples/rosenbrock/sparse_rosenbrock.cpp is span 2, stride 1)."""
        check(lib.po_problem_set_chain(self._h, int(span), int(stride), int(bool(reverse_cols))))
        a, b = C.c_int64(), C.c_int64()
        check(lib.po_problem_sparse_sizes(self._h, C.byref(a), C.byref(b)))
        self.nwcon = a.value
        return self

    def setVarBoundOptions(self, use_lower=True, use_upper=True):
        check(lib.po_problem_set_var_bound_options(self._h, int(bool(use_lower)), int(bool(use_upper))))
        return self

    def setBoundsMode(self, mode):
        """Deliberately broken bounds for the bound-repair tests (po_problem_set_bounds_mode)."""
        check(lib.po_problem_set_bounds_mode(self._h, int(mode)))
        return self

    def setLinearConstraints(self, flag=True):
        """The dense constraints are linear: the solver keeps the Jacobian of the first gradient evaluation
        of each optimize() and asks for the objective gradient only afterwards (po_problem_set_linear_constraints)."""
        check(lib.po_problem_set_linear_constraints(self._h, int(bool(flag))))
        return self

    @property
    def handle(self):
        return self._h

    def evalObjCon(self, x):
        f = C.c_double()
        con = np.zeros(max(self.ncon, 1))
        check(lib.po_problem_eval_obj_con(self._h, x.handle, C.byref(f), con.ctypes.data_as(L.c_double_p)))
        return 0, f.value, con[: self.ncon]

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib.po_problem_destroy(self._h)
        except Exception:
            pass


class UserLibraryProblem:
    """A ParOptProblem subclass that lives in a USER'S shared library built on include/ParOptAMD.hpp (e.g.
    examples/librandom_convex_user.so): the library hands over the `po_problem` of its facade object, the solver
    classes of this module drive it like any other problem.  Entry points expected from the library (extern "C"):
    <prefix>_problem_create(ctx, nglobal, ncon, seed) -> void*, <prefix>_problem_handle(void*) -> po_problem,
    <prefix>_problem_sizes(void*, int64*, int64*), <prefix>_problem_destroy(void*), and optionally
    <prefix>_problem_set_linear_constraints / _set_deferred_reductions(void*, int), _own_kernel_bytes(void*, int)."""

    def __init__(self, ctx, path, nglobal, ncon, seed=0, prefix="rc", nwcon=0, nw=0):
        """nwcon > 0: <prefix>_problem_create_weighting(ctx, nglobal, ncon, seed, nwcon, nw) instead (a problem with one
        sparse constraint per group of nw consecutive variables, examples/weighting_amd.cpp)."""
        self.ctx = ctx
        self._lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        f = lambda name: getattr(self._lib, "%s_problem_%s" % (prefix, name))
        f("create").restype = C.c_void_p
        f("create").argtypes = [L.po_ctx, C.c_int64, C.c_int, C.c_uint64]
        f("handle").restype = L.po_problem
        f("handle").argtypes = [C.c_void_p]
        f("sizes").argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        f("destroy").argtypes = [C.c_void_p]
        self._f = f
        if nwcon > 0:
            f("create_weighting").restype = C.c_void_p
            f("create_weighting").argtypes = [L.po_ctx, C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_int]
            self._obj = C.c_void_p(f("create_weighting")(ctx.handle, int(nglobal), int(ncon), int(seed), int(nwcon),
                                                         int(nw)))
        else:
            self._obj = C.c_void_p(f("create")(ctx.handle, int(nglobal), int(ncon), int(seed)))
        if not self._obj:
            raise L.ParOptAMDError(-1, "user library failed to create its problem")
        self._h = L.po_problem(f("handle")(self._obj))
        nl, off = C.c_int64(), C.c_int64()
        f("sizes")(self._obj, C.byref(nl), C.byref(off))
        self.nvars, self.offset, self.ncon, self.nwcon = nl.value, off.value, int(ncon), int(nwcon)

    @property
    def handle(self):
        return self._h

    def setLinearConstraints(self, flag=True):
        fn = self._f("set_linear_constraints")
        fn.argtypes = [C.c_void_p, C.c_int]
        fn(self._obj, int(bool(flag)))
        return self

    def setDeferredReductions(self, flag=True):
        fn = self._f("set_deferred_reductions")
        fn.argtypes = [C.c_void_p, C.c_int]
        fn(self._obj, int(bool(flag)))
        return self

    def ownKernelBytes(self, jacobian_rewritten=True):
        """Algorithmic HBM bytes of the user's own kernels so far (the library cannot count them)."""
        fn = self._f("own_kernel_bytes")
        fn.restype = C.c_double
        fn.argtypes = [C.c_void_p, C.c_int]
        return fn(self._obj, int(bool(jacobian_rewritten)))

    def close(self):
        if self._obj:
            self._f("destroy")(self._obj)
            self._obj = None
            self._h = None

    def __del__(self):
        try:
            if self.ctx._h:
                self.close()
        except Exception:
            pass


class CsrSymbolic:
    """The one-time host analysis of a CSR sparse Jacobian pattern (no device needed): ordering,
    elimination tree, pattern of the Cholesky factor of S = C + Aw D^-1 Aw^T, dependency level sets."""

    def __init__(self, nvars, rowp, cols):
        rowp = np.ascontiguousarray(rowp, dtype=np.intc)
        cols = np.ascontiguousarray(cols, dtype=np.intc)
        w = len(rowp) - 1
        h = L.po_csr_symbolic()
        check(lib.po_csr_symbolic_create(int(nvars), w, rowp.ctypes.data_as(L.c_int_p),
                                         cols.ctypes.data_as(L.c_int_p), C.byref(h)))
        try:
            info = (C.c_int64 * 7)()
            check(lib.po_csr_symbolic_info(h, info))
            self.nnz, self.nnzS, self.nnzL, self.nlevels = (int(v) for v in info[:4])
            self.sorted_input = bool(info[4])
            self.nfronts, self.max_front = int(info[5]), int(info[6])
            ptrs = [L.c_int_p() for _ in range(6)]
            check(lib.po_csr_symbolic_arrays(h, *[C.byref(p) for p in ptrs]))

            def arr(p, n):
                return np.ctypeslib.as_array(p, shape=(n,)).copy() if n > 0 else np.zeros(0, dtype=np.intc)

            self.perm, self.parent = arr(ptrs[0], w), arr(ptrs[1], w)
            self.Lrowp, self.Lcols = arr(ptrs[2], w + 1), arr(ptrs[3], self.nnzL)
            self.level_ptr = arr(ptrs[4], self.nlevels + 1)
            self.front_of = arr(ptrs[5], w)
        finally:
            lib.po_csr_symbolic_destroy(h)


def quasidef_factor(problem, x, dinv, c):
    """ParOptQuasiDefMat::factor (src/ParOptSparseMat.h:25) of the problem's sparse constraints."""
    check(lib.po_quasidef_factor(problem.handle, x.handle, dinv.handle, c.handle))


def quasidef_apply(problem, x, dinv, c, bx, bw, yx, yw):
    """ParOptQuasiDefMat::apply (:39-56); bw may be None."""
    check(lib.po_quasidef_apply(problem.handle, x.handle, dinv.handle, c.handle, bx.handle,
                                bw.handle if bw is not None else None, yx.handle, yw.handle))


def quasidef_factor_info(problem):
    s = lib.po_quasidef_factor_info(problem.handle)
    return s.decode() if s else None


class InteriorPoint:
    """ParOptInteriorPoint (reference src/ParOptInteriorPoint.h:128-217)."""

    def __init__(self, problem, options=None):
        self.problem = problem
        self.ctx = problem.ctx
        self._h = L.po_ip()
        check(lib.po_ip_create(problem.handle, C.byref(self._h)))
        self._iter_cb = None
        opts = dict(options or {})
        # the C ABI keeps the reference's default ("paropt.out" in the working directory); the
        # Python harness only writes the iteration table when asked to
        opts.setdefault("output_file", "")
        for k, v in opts.items():
            self.setOption(k, v)

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib.po_ip_destroy(self._h)
        except Exception:
            pass

    def setOption(self, name, value):
        nm = name.encode()
        if isinstance(value, bool):
            check(lib.po_ip_set_option_int(self._h, nm, int(value)))
        elif isinstance(value, int):
            # int given for a float option is a type error in the reference too; be lenient only
            # in the obvious direction (python ints for float-valued settings)
            rc = lib.po_ip_set_option_int(self._h, nm, value)
            if rc != 0:
                check(lib.po_ip_set_option_float(self._h, nm, float(value)))
        elif isinstance(value, float):
            check(lib.po_ip_set_option_float(self._h, nm, value))
        else:
            check(lib.po_ip_set_option_str(self._h, nm, str(value).encode()))

    def setIterationCallback(self, fn):
        def _cb(user, k):
            fn(k)
            return 0

        self._iter_cb = L.ITER_FN(_cb)
        check(lib.po_ip_set_iteration_callback(self._h, self._iter_cb, None))

    def optimize(self, checkpoint=None):
        rc = lib.po_ip_optimize(self._h, checkpoint.encode() if checkpoint else None)
        if hasattr(self.problem, "_raise_pending"):
            self.problem._raise_pending()  # an exception thrown inside a problem callback
        if rc not in (0,):
            raise L.ParOptAMDError(rc, lib.po_last_error().decode(errors="replace"))
        return rc

    def writeSolutionFile(self, filename):
        check(lib.po_ip_write_solution_file(self._h, filename.encode()))

    def readSolutionFile(self, filename):
        check(lib.po_ip_read_solution_file(self._h, filename.encode()))

    def getOptimizedPoint(self):
        x, zl, zu = L.po_vec(), L.po_vec(), L.po_vec()
        z = L.c_double_p()
        check(lib.po_ip_get_optimized_point(self._h, C.byref(x), C.byref(z), C.byref(zl), C.byref(zu)))
        c = self.problem.ncon
        # zl / zu are None for a side the problem declares unused (as the reference returns NULL)
        return (PVec(self.ctx, handle=x, owned=False), np.array([z[i] for i in range(c)]),
                PVec(self.ctx, handle=zl, owned=False) if zl else None,
                PVec(self.ctx, handle=zu, owned=False) if zu else None)

    def getOptimizedSlacks(self):
        ptrs = [L.c_double_p() for _ in range(4)]
        check(lib.po_ip_get_optimized_slacks(self._h, *[C.byref(p) for p in ptrs]))
        c = self.problem.ncon
        return tuple(np.array([p[i] for i in range(c)]) for p in ptrs)

    def getOptimizedSparse(self):
        """(zw, sw, tw, zsw, ztw) as borrowed PVec handles, or None when nwcon = 0."""
        hs = [L.po_vec() for _ in range(5)]
        check(lib.po_ip_get_optimized_sparse(self._h, *[C.byref(h) for h in hs]))
        if not hs[0]:
            return None
        return tuple(PVec(self.ctx, handle=h, owned=False) for h in hs)

    def getIterationCounters(self):
        a, b, d = C.c_int(), C.c_int(), C.c_int()
        check(lib.po_ip_get_counters(self._h, C.byref(a), C.byref(b), C.byref(d)))
        return a.value, b.value, d.value

    def setPenaltyGamma(self, gamma):
        check(lib.po_ip_set_penalty_gamma(self._h, float(gamma)))

    def setMultiplePenaltyGamma(self, gamma):
        arr = np.ascontiguousarray(gamma, dtype=np.float64)
        assert len(arr) == self.problem.ncon
        check(lib.po_ip_set_penalty_gamma_array(self._h, arr.ctypes.data_as(L.c_double_p)))

    def setQuasiNewton(self, qn):
        """Use a caller-owned LBFGS / LSR1 (kept alive by this object); None detaches it."""
        self._qn_ref = qn
        check(lib.po_ip_set_quasi_newton(self._h, qn._h if qn is not None else None))

    def resetProblemInstance(self, problem):
        self._prob_ref = problem
        check(lib.po_ip_reset_problem_instance(self._h, problem.handle))

    def resetQuasiNewtonHessian(self):
        check(lib.po_ip_reset_quasi_newton(self._h))

    def resetDesignAndBounds(self):
        check(lib.po_ip_reset_design_and_bounds(self._h))

    def checkGradients(self, dh=1e-6):
        """ParOptInteriorPoint::checkGradients(dh) (reference .cpp:6196-6199): the problem's finite-difference check
        at the solver's current point; returns (and prints) the report."""
        t = C.c_char_p()
        check(lib.po_ip_check_gradients(self._h, float(dh), C.byref(t)))
        text = (t.value or b"").decode()
        print(text, end="")
        return text

    def checkMeritFuncGradient(self, xpt=None, dh=1e-6):
        """ParOptInteriorPoint::checkMeritFuncGradient(xpt, dh) (.cpp:3280-3432): returns (finite difference, actual)."""
        fd, act = C.c_double(), C.c_double()
        check(lib.po_ip_check_merit_func_gradient(self._h, xpt.handle if xpt is not None else None, float(dh),
                                                  C.byref(fd), C.byref(act)))
        return fd.value, act.value

    def setBFGSUpdateType(self, update_type):
        """setBFGSUpdateType (.cpp:1179-1186): applies to the solver's own L-BFGS object."""
        h = L.po_qn()
        check(lib.po_ip_get_quasi_newton(self._h, C.byref(h)))
        if h:
            check(lib.po_qn_set_update_type(h, 1 if update_type in (1, "damped_update", "damped") else 0))

    def setUseDiagHessian(self, truth):
        self.setOption("use_diag_hessian", bool(truth))

    def getHvecCount(self):
        v = C.c_int()
        check(lib.po_ip_get_hvec_count(self._h, C.byref(v)))
        return v.value

    def getBarrierParameter(self):
        v = C.c_double()
        check(lib.po_ip_get_barrier_parameter(self._h, C.byref(v)))
        return v.value

    def getComplementarity(self):
        v = C.c_double()
        check(lib.po_ip_get_complementarity(self._h, C.byref(v)))
        return v.value

    def getObjective(self):
        f, rho = C.c_double(), C.c_double()
        check(lib.po_ip_get_objective(self._h, C.byref(f), C.byref(rho)))
        return f.value, rho.value

    def getQuasiNewton(self):
        h = L.po_qn()
        check(lib.po_ip_get_quasi_newton(self._h, C.byref(h)))
        return _QuasiNewton(self.ctx, 0, 0, 0, handle=h) if h else None

    def getHistory(self):
        t = C.c_char_p()
        check(lib.po_ip_get_history(self._h, C.byref(t)))
        return t.value.decode()

    def setCallbackTiming(self, on=True):
        """Event timing of the problem's callbacks (the "user_eval" entry of getPhaseTimes): off by default, every
        event record costs the stream some dispatch latency (po_ip_set_callback_timing)."""
        check(lib.po_ip_set_callback_timing(self._h, 1 if on else 0))
        return self

    def getPhaseTimes(self):
        names, secs, cnt = C.c_char_p(), L.c_double_p(), C.c_int()
        check(lib.po_ip_get_phase_times(self._h, C.byref(names), C.byref(secs), C.byref(cnt)))
        nm = names.value.decode().split(";") if cnt.value else []
        return {nm[i]: secs[i] for i in range(cnt.value)}

    def debugKKTStep(self, mu):
        px, pzl, pzu = L.po_vec(), L.po_vec(), L.po_vec()
        d = [L.c_double_p() for _ in range(5)]
        check(lib.po_ip_debug_kkt_step(self._h, float(mu), C.byref(px), C.byref(pzl), C.byref(pzu),
                                       *[C.byref(p) for p in d]))
        c = self.problem.ncon
        out = dict(x=PVec(self.ctx, handle=px, owned=False).to_numpy(),
                   zl=PVec(self.ctx, handle=pzl, owned=False).to_numpy(),
                   zu=PVec(self.ctx, handle=pzu, owned=False).to_numpy())
        for name, p in zip(("z", "s", "t", "zs", "zt"), d):
            out[name] = np.array([p[i] for i in range(c)])
        wh = [L.po_vec() for _ in range(5)]
        check(lib.po_ip_debug_kkt_step_sparse(self._h, *[C.byref(h) for h in wh]))
        if wh[0]:
            for name, h in zip(("zw", "sw", "tw", "zsw", "ztw"), wh):
                out[name] = PVec(self.ctx, handle=h, owned=False).to_numpy()
        return out

    def debugSetState(self, z, s, t, zs, zt, mu):
        """po_ip_debug_set_state: dense blocks + barrier parameter of an injected state (x, zl, zu and the sparse
        blocks are written through getOptimizedPoint() / getOptimizedSparse() handles beforehand)."""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (z, s, t, zs, zt)]
        check(lib.po_ip_debug_set_state(self._h, *[a.ctypes.data_as(L.c_double_p) for a in arrs], float(mu)))

    def debugKKT(self, mu, mode, tau=0.95):
        """po_ip_debug_kkt: the pieces of one KKT step at the current state as numpy copies (mode 0: one bordered
        solve; mode 1: the fused kernel sequence of a plain iteration = the step after one refinement)."""
        d = L.KKTDump()
        check(lib.po_ip_debug_kkt(self._h, float(mu), int(mode), float(tau), C.byref(d)))
        c, k = d.c, d.k
        m = c + k

        def arr(p, cnt):
            return np.array([p[i] for i in range(cnt)], dtype=np.float64)

        def vec(h):
            return PVec(self.ctx, handle=L.po_vec(h), owned=False).to_numpy()

        out = dict(c=c, k=k, Dinv=vec(d.Dinv), res_x=vec(d.res_x), res_z=arr(d.res_z, c), res_s=arr(d.res_s, c),
                   res_t=arr(d.res_t, c), res_zs=arr(d.res_zs, c), res_zt=arr(d.res_zt, c),
                   res_norms=np.array(list(d.res_norms)),
                   W=arr(d.W, m * m).reshape(m, m).T, G=arr(d.G, c * c).reshape(c, c).T,
                   Ce=arr(d.Ce, k * k).reshape(k, k).T if k > 0 else np.zeros((0, 0)),
                   gpiv=np.array([d.gpiv[i] for i in range(c)], dtype=np.int64) + 1,
                   cpiv=np.array([d.cpiv[i] for i in range(k)], dtype=np.int64) + 1,
                   step_x=vec(d.px), step_zl=vec(d.pzl), step_zu=vec(d.pzu), step_z=arr(d.pz, c),
                   step_s=arr(d.ps, c), step_t=arr(d.pt, c), step_zs=arr(d.pzs, c), step_zt=arr(d.pzt, c),
                   step_mins=np.array(list(d.step_mins)))
        wh = [L.po_vec() for _ in range(5)]
        check(lib.po_ip_debug_kkt_step_sparse(self._h, *[C.byref(h) for h in wh]))
        if wh[0]:
            for name, h in zip(("zw", "sw", "tw", "zsw", "ztw"), wh):
                out["step_" + name] = PVec(self.ctx, handle=h, owned=False).to_numpy()
        return out

    def snapshot(self):
        """State in the layout of oracle snapshots / golden 'itNNN/' records."""
        x, z, zl, zu = self.getOptimizedPoint()
        s, t, zs, zt = self.getOptimizedSlacks()
        niter, neval, ngeval = self.getIterationCounters()
        f, rho = self.getObjective()
        d = dict(mu=self.getBarrierParameter(), rho=rho, fobj=f, z=z, s=s, t=t, zs=zs, zt=zt,
                 counters=np.array([niter, neval, ngeval]),
                 norms=np.array([x.norm(), zl.norm() if zl is not None else np.nan,
                                 zu.norm() if zu is not None else np.nan]))
        wv = self.getOptimizedSparse()
        if wv is not None:
            d["wnorms"] = np.array([v.norm() for v in wv])
        qn = self.getQuasiNewton()
        if qn is not None:
            k, b0 = C.c_int(), C.c_double()
            check(lib.po_qn_get_compact(qn._h, C.byref(k), C.byref(b0), None, None, None))
            d["qn_size"] = k.value
            d["qn_b0"] = b0.value
            pp, pn = L.c_int_p(), C.c_int()
            check(lib.po_qn_get_pivots(qn._h, C.byref(pp), C.byref(pn)))
            d["mfpiv"] = np.array([pp[i] for i in range(pn.value)], dtype=np.int64) + 1  # LAPACK numbering
        d.update(self.getDebugInts())
        return d

    def getDebugInts(self):
        """SURVEY 8a' integers in the reference's numbering: gpiv (1-based LAPACK rows), check_flag, clamped[8]."""
        pp, pn, fl = L.c_int_p(), C.c_int(), C.c_int()
        cl = (C.c_int64 * 8)()
        check(lib.po_ip_get_debug_ints(self._h, C.byref(pp), C.byref(pn), C.byref(fl), cl))
        return dict(gpiv=np.array([pp[i] for i in range(pn.value)], dtype=np.int64) + 1, check_flag=fl.value,
                    clamped=np.array(list(cl), dtype=np.int64))

    def getBounds(self):
        """(lb, ub) as repaired by initAndCheckDesignAndBounds (borrowed)."""
        a, b = L.po_vec(), L.po_vec()
        check(lib.po_ip_get_bounds(self._h, C.byref(a), C.byref(b)))
        return PVec(self.ctx, handle=a, owned=False), PVec(self.ctx, handle=b, owned=False)


class EigenApprox:
    """ParOptCompactEigenApprox as seen by the model-update callback: c0 (settable), g0 (PVec), N,
    M and Minv (row-major N x N numpy views, writable), hvecs (list of PVec)."""

    def __init__(self, ctx, handle):
        c0, g0, N = L.c_double_p(), L.po_vec(), C.c_int()
        M, Minv, hv = L.c_double_p(), L.c_double_p(), L.vec_p()
        check(lib.po_eig_get_approximation(handle, C.byref(c0), C.byref(g0), C.byref(N), C.byref(M),
                                           C.byref(Minv), C.byref(hv)))
        self.N = N.value
        self._c0 = c0
        self.g0 = PVec(ctx, handle=g0, owned=False)
        self.M = np.ctypeslib.as_array(M, shape=(self.N, self.N))
        self.Minv = np.ctypeslib.as_array(Minv, shape=(self.N, self.N))
        self.hvecs = [PVec(ctx, handle=L.po_vec(hv[i]), owned=False) for i in range(self.N)]

    @property
    def c0(self):
        return self._c0[0]

    @c0.setter
    def c0(self, v):
        self._c0[0] = float(v)


class CompactEigenApprox(EigenApprox):
    """ParOptCompactEigenApprox(problem, N) (reference src/ParOptCompactEigenvalueApprox.h:7-32): the object the user's
    code creates and hands to EigenQuasiNewton; c0 / g0 / M / Minv / hvecs as in EigenApprox."""

    def __init__(self, problem, N):
        self.ctx = problem.ctx
        self.problem = problem
        self._eh = L.po_eig()
        check(lib.po_eig_create(problem.handle, int(N), C.byref(self._eh)))
        super().__init__(problem.ctx, self._eh)

    def __del__(self):
        try:
            if self._eh and self.ctx._h:
                lib.po_eig_destroy(self._eh)
        except Exception:
            pass

    @property
    def handle(self):
        return self._eh

    def multAdd(self, alpha, x, y):
        check(lib.po_eig_mult_add(self._eh, float(alpha), x.handle, y.handle))

    def evalApproximation(self, s=None, t=None):
        v = C.c_double()
        check(lib.po_eig_eval_approximation(self._eh, s.handle if s is not None else None,
                                            t.handle if t is not None else None, C.byref(v)))
        return v.value

    def evalApproximationGradient(self, s, grad):
        check(lib.po_eig_eval_approximation_gradient(self._eh, s.handle, grad.handle))


class EigenQuasiNewton(_QuasiNewton):
    """ParOptEigenQuasiNewton(qn, eigh, index) (.h:34-84): B = B_qn - z0 H M H^T as one compact matrix; qn may be
    None.  A quasi-Newton object like any other (mult / multAdd / getCompactMat)."""

    def __init__(self, qn, eigh, index=0):
        self.qn, self.eigh = qn, eigh  # kept alive: the library object borrows both
        h = L.po_qn()
        check(lib.po_eigqn_create(qn._h if qn is not None else None, eigh.handle, int(index), C.byref(h)))
        super().__init__(eigh.ctx, 0, 0, 0, handle=h)
        self._owned = True
        self.index = int(index)

    def setUseQuasiNewtonObjective(self, truth):
        check(lib.po_eigqn_set_use_quasi_newton_objective(self._h, int(bool(truth))))

    def updateMultipliers(self, z):
        za = (C.c_double * len(z))(*[float(v) for v in z])
        check(lib.po_eigqn_update_multipliers(self._h, za))


class TrustRegionSubproblem:
    """ParOptTrustRegionSubproblem (reference src/ParOptTrustRegion.h:15-151) in its library forms.  Also the
    ParOptProblem the interior-point solver is built on: ``InteriorPoint(subproblem, options)``."""

    def __init__(self, problem):
        self.problem = problem
        self.ctx = problem.ctx
        self.nvars, self.ncon = problem.nvars, problem.ncon
        self.nwcon = getattr(problem, "nwcon", 0)
        self._h = L.po_trsub()
        self._keep = []

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib.po_trsub_destroy(self._h)
        except Exception:
            pass

    @property
    def handle(self):
        """the subproblem as a po_problem (borrowed)"""
        p = L.po_problem()
        check(lib.po_trsub_problem(self._h, C.byref(p)))
        return p

    def _raise_pending(self):
        if hasattr(self.problem, "_raise_pending"):
            self.problem._raise_pending()

    def initModelAndBounds(self, tr_size):
        check(lib.po_trsub_init_model_and_bounds(self._h, float(tr_size)))

    def setTrustRegionBounds(self, tr_size):
        check(lib.po_trsub_set_trust_region_bounds(self._h, float(tr_size)))

    def evalTrialStepAndUpdate(self, update_flag, step, z, zw=None):
        za = (C.c_double * max(1, self.ncon))(*[float(v) for v in z])
        f, cons = C.c_double(), (C.c_double * max(1, self.ncon))()
        check(lib.po_trsub_eval_trial_step_and_update(self._h, int(update_flag), step.handle, za,
                                                      zw.handle if zw is not None else None, C.byref(f), cons))
        return f.value, np.array(cons[:self.ncon])

    def acceptTrialStep(self, step, z, zw=None):
        za = (C.c_double * max(1, self.ncon))(*[float(v) for v in z]) if z is not None else None
        check(lib.po_trsub_accept_trial_step(self._h, step.handle, za, zw.handle if zw is not None else None))

    def rejectTrialStep(self):
        check(lib.po_trsub_reject_trial_step(self._h))

    def getQuasiNewtonUpdateType(self):
        t = C.c_int()
        check(lib.po_trsub_get_quasi_newton_update_type(self._h, C.byref(t)))
        return t.value

    # the ParOptProblem side (the model in the step), as the interior point sees it
    def getVarsAndBounds(self, step, lower, upper):
        check(lib.po_problem_get_vars_and_bounds(self.handle, step.handle, lower.handle, upper.handle))

    def evalObjCon(self, step):
        f = C.c_double()
        con = np.zeros(max(self.ncon, 1))
        rc = lib.po_problem_eval_obj_con(self.handle, step.handle, C.byref(f), con.ctypes.data_as(L.c_double_p))
        return rc, f.value, con[: self.ncon]

    def evalObjConGradient(self, step, g, A):
        hs = (L.po_vec * max(1, self.ncon))(*[a.handle for a in A])
        return lib.po_problem_eval_obj_con_gradient(self.handle, step.handle, g.handle, hs)

    def getLinearModel(self):
        """(xk, fk, gk, ck, Ak, lb, ub) of the current model (borrowed vectors)."""
        xk, gk, lb, ub, Ak = L.po_vec(), L.po_vec(), L.po_vec(), L.po_vec(), L.vec_p()
        fk, ck, m = C.c_double(), L.c_double_p(), C.c_int()
        check(lib.po_trsub_get_linear_model(self._h, C.byref(xk), C.byref(fk), C.byref(gk), C.byref(ck), C.byref(Ak),
                                            C.byref(lb), C.byref(ub), C.byref(m)))
        wrap = lambda h: PVec(self.ctx, handle=h, owned=False)  # noqa: E731
        return (wrap(xk), fk.value, wrap(gk), np.array([ck[i] for i in range(m.value)]),
                [wrap(L.po_vec(Ak[i])) for i in range(m.value)], wrap(lb), wrap(ub))


class UserTrustRegionSubproblem(TrustRegionSubproblem):
    """A trust-region subproblem written by the USER: subclass and override the virtuals of
    ParOptTrustRegionSubproblem (reference src/ParOptTrustRegion.h:15-151) -- all arguments are device vectors (PVec),
    multipliers numpy arrays:

        getQuasiNewton() -> LBFGS / LSR1 / EigenQuasiNewton or None
        initModelAndBounds(tr_size); setTrustRegionBounds(tr_size)
        evalTrialStepAndUpdate(update_flag, step, z, zw) -> (fail, fobj, cons)
        acceptTrialStep(step, z, zw) -> fail;  rejectTrialStep();  getQuasiNewtonUpdateType() -> int
        getLinearModel() -> (xk, fk, gk, ck, Ak, lb, ub)            (the user's own vectors; borrowed by the driver)

    and the ParOptProblem side the interior point solves (the model in the step):

        getVarsAndBounds(step, lower, upper)                         (filled in place)
        evalObjCon(step or None) -> (fail, fobj, cons)               (None: the values at a zero step)
        evalObjConGradient(step, g, A) -> fail                       (A is None when only g is wanted)

    ``TrustRegion(sub, options).optimize(InteriorPoint(sub, options))`` then drives it exactly like the library's own
    subproblems (po_trsub_create_callbacks)."""

    def __init__(self, problem):
        super().__init__(problem)
        ctx, m = self.ctx, self.ncon
        self._pending_exc = None
        wrap = lambda h: PVec(ctx, handle=L.po_vec(h), owned=False) if h else None  # noqa: E731

        def guard(fn):
            def g(*args):
                if self._pending_exc is not None:
                    return 1
                try:
                    return int(fn(*args) or 0)
                except BaseException as e:  # noqa: BLE001 - re-raised by _raise_pending()
                    self._pending_exc = e
                    return 1
            return g

        @guard
        def _getqn(user, out):
            q = self.getQuasiNewton()
            out[0] = q._h if q is not None else None
            return 0

        @guard
        def _init(user, tr):
            return self.initModelAndBounds(tr)

        @guard
        def _setb(user, tr):
            return self.setTrustRegionBounds(tr)

        @guard
        def _trial(user, flag, step, z, zw, fobj, cons):
            fail, f, c = self.evalTrialStepAndUpdate(flag, wrap(step), np.array([z[i] for i in range(m)]), wrap(zw))
            fobj[0] = float(f)
            for i in range(m):
                cons[i] = float(c[i])
            return fail

        @guard
        def _accept(user, step, z, zw):
            za = np.array([z[i] for i in range(m)]) if z else None
            return self.acceptTrialStep(wrap(step), za, wrap(zw))

        @guard
        def _reject(user):
            return self.rejectTrialStep()

        def _utype(user):
            try:
                return int(self.getQuasiNewtonUpdateType())
            except BaseException as e:  # noqa: BLE001
                self._pending_exc = self._pending_exc or e
                return 0

        @guard
        def _model(user, xk, fk, gk, ck, Ak, lb, ub):
            vx, f, vg, c, A, vl, vu = self.getLinearModel()
            # the arrays handed out must outlive the call: kept on the object until the next call
            self._ck_arr = (C.c_double * max(1, m))(*[float(v) for v in c])
            self._ak_arr = (L.po_vec * max(1, m))(*[a.handle.value for a in A])
            xk[0], gk[0], lb[0], ub[0] = vx.handle.value, vg.handle.value, vl.handle.value, vu.handle.value
            fk[0] = float(f)
            ck[0] = C.cast(self._ck_arr, L.c_double_p)
            Ak[0] = C.cast(self._ak_arr, L.vec_p)
            return 0

        @guard
        def _bounds(user, step, lo, up):
            return self.getVarsAndBounds(wrap(step), wrap(lo), wrap(up))

        @guard
        def _eval(user, step, fobj, cons):
            fail, f, c = self.evalObjCon(wrap(step))
            fobj[0] = float(f)
            for i in range(m):
                cons[i] = float(c[i])
            return fail

        @guard
        def _grad(user, step, g, Ac):
            A = [wrap(Ac[i]) for i in range(m)] if Ac else None
            return self.evalObjConGradient(wrap(step), wrap(g), A)

        cb = L.TrSubCallbacks()
        self._fns = (L.TRSUB_GETQN_FN(_getqn), L.TRSUB_SIZE_FN(_init), L.TRSUB_SIZE_FN(_setb), L.TRSUB_TRIAL_FN(_trial),
                     L.TRSUB_ACCEPT_FN(_accept), L.TRSUB_VOID_FN(_reject), L.TRSUB_VOID_FN(_utype),
                     L.TRSUB_MODEL_FN(_model), L.TRSUB_BOUNDS_FN(_bounds), L.TRSUB_EVAL_FN(_eval), L.TRSUB_GRAD_FN(_grad))
        cb.user = None
        (cb.get_quasi_newton, cb.init_model_and_bounds, cb.set_trust_region_bounds, cb.eval_trial_step_and_update,
         cb.accept_trial_step, cb.reject_trial_step, cb.get_quasi_newton_update_type, cb.get_linear_model,
         cb.get_vars_and_bounds, cb.eval_obj_con, cb.eval_obj_con_gradient) = self._fns
        self._cb_struct = cb
        check(lib.po_trsub_create_callbacks(problem.handle, C.byref(cb), C.byref(self._h)))

    def _raise_pending(self):
        e, self._pending_exc = self._pending_exc, None
        if e is not None:
            raise e
        super()._raise_pending()

    # the virtuals (the base class's versions of these names call INTO the library: a user subproblem defines them)
    def getQuasiNewton(self):
        return None

    def getQuasiNewtonUpdateType(self):
        return 0

    def rejectTrialStep(self):
        return 0


class QuadraticSubproblem(TrustRegionSubproblem):
    """ParOptQuadraticSubproblem(problem, qn) (.h:153-300); qn may be None."""

    def __init__(self, problem, qn=None):
        super().__init__(problem)
        self.qn = qn
        check(lib.po_trsub_create_quadratic(problem.handle, qn._h if qn is not None else None, C.byref(self._h)))

    def getQuasiNewton(self):
        return self.qn


class EigenSubproblem(TrustRegionSubproblem):
    """ParOptEigenSubproblem(problem, eig_qn) (src/ParOptCompactEigenvalueApprox.h:86-206)."""

    def __init__(self, problem, eig_qn):
        super().__init__(problem)
        self.eig_qn = eig_qn
        check(lib.po_trsub_create_eigen(problem.handle, eig_qn._h, C.byref(self._h)))

    def getQuasiNewton(self):
        return self.eig_qn

    def setEigenModelUpdate(self, update):
        """update(x: PVec, approx: CompactEigenApprox) is called at the starting point and at every accepted point with
        c0 / g0 preset; it fills hvecs, M and Minv (setEigenModelUpdate, .h:166-170)."""
        approx = self.eig_qn.eigh

        def _cb(user, x, eig):
            update(PVec(self.ctx, handle=L.po_vec(x), owned=False), approx)
            return 0

        fn = L.EIG_UPDATE_FN(_cb)
        self._keep.append(fn)
        check(lib.po_trsub_set_eigen_model_update(self._h, fn, None))


class InfeasSubproblem:
    """ParOptInfeasSubproblem(subproblem, subproblem_objective, subproblem_constraint) (reference
    src/ParOptTrustRegion.h:293-374, .cpp:468-650): the problem of the trust-region driver's steering step as a problem
    of its own -- ``InteriorPoint(InfeasSubproblem(sub, LINEAR_OBJECTIVE, LINEAR_CONSTRAINT), options)``.  The selector
    constants are the reference's."""

    SUBPROBLEM_OBJECTIVE, LINEAR_OBJECTIVE, CONSTANT_OBJECTIVE = 1, 2, 3
    SUBPROBLEM_CONSTRAINT, LINEAR_CONSTRAINT = 1, 2

    def __init__(self, subproblem, subproblem_objective, subproblem_constraint):
        self.subproblem = subproblem  # borrowed by the library object: kept alive here
        self.ctx = subproblem.ctx
        self.nvars, self.ncon, self.nwcon = subproblem.nvars, subproblem.ncon, subproblem.nwcon
        self._h = L.po_problem()
        check(lib.po_infeas_create(subproblem._h, int(subproblem_objective), int(subproblem_constraint),
                                   C.byref(self._h)))

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib.po_problem_destroy(self._h)
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _raise_pending(self):
        self.subproblem._raise_pending()

    def setObjectiveScaling(self, scale):
        check(lib.po_infeas_set_objective_scaling(self._h, float(scale)))

    def getVarsAndBounds(self, x, lb, ub):
        check(lib.po_problem_get_vars_and_bounds(self._h, x.handle, lb.handle, ub.handle))

    def evalObjCon(self, x):
        f = C.c_double()
        con = np.zeros(max(self.ncon, 1))
        rc = lib.po_problem_eval_obj_con(self._h, x.handle, C.byref(f), con.ctypes.data_as(L.c_double_p))
        return rc, f.value, con[: self.ncon]

    def evalObjConGradient(self, x, g, A):
        hs = (L.po_vec * max(1, self.ncon))(*[a.handle for a in A])
        return lib.po_problem_eval_obj_con_gradient(self._h, x.handle, g.handle, hs)


class TrustRegion:
    """ParOptTrustRegion.  ``TrustRegion(problem, options)`` assembles quasi-Newton object, quadratic (or eigenvalue)
    subproblem and interior-point solver the way ParOptOptimizer does for algorithm='tr' (reference
    src/ParOptOptimizer.cpp:108-183); ``TrustRegion(subproblem, options)`` with a TrustRegionSubproblem is the
    reference's own constructor (src/ParOptTrustRegion.cpp:660-718) and takes the solver at ``optimize(ip)``.
    `options` may mix interior-point and trust-region option names (one shared registry)."""

    COLS = ("fobj", "infeas", "l1", "linfty", "smax", "tr", "rho", "model_reduc", "zav", "zmax", "gav", "gmax")

    def __init__(self, problem, options=None):
        self.problem = problem
        self.ctx = problem.ctx
        self._h = L.po_tr()
        self.subproblem = problem if isinstance(problem, TrustRegionSubproblem) else None
        if self.subproblem is not None:
            check(lib.po_tr_create_subproblem(problem._h, C.byref(self._h)))
        else:
            check(lib.po_tr_create(problem.handle, C.byref(self._h)))
        self._cbs = []
        opts = dict(options or {})
        opts.setdefault("tr_output_file", "")
        for k, v in opts.items():
            self.setOption(k, v)

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib.po_tr_destroy(self._h)
        except Exception:
            pass

    def setOption(self, name, value):
        nm = name.encode()
        if isinstance(value, bool):
            check(lib.po_tr_set_option_int(self._h, nm, int(value)))
        elif isinstance(value, int):
            rc = lib.po_tr_set_option_int(self._h, nm, value)
            if rc != 0:
                check(lib.po_tr_set_option_float(self._h, nm, float(value)))
        elif isinstance(value, float):
            check(lib.po_tr_set_option_float(self._h, nm, value))
        else:
            check(lib.po_tr_set_option_str(self._h, nm, str(value).encode()))

    def setEigenModel(self, N, index, update):
        """update(x: PVec, approx: EigenApprox) fills approx.hvecs / M / Minv (c0, g0 are preset)."""
        def _cb(user, x, approx):
            update(PVec(self.ctx, handle=L.po_vec(x), owned=False), EigenApprox(self.ctx, L.po_eig(approx)))
            return 0

        fn = L.EIG_UPDATE_FN(_cb)
        self._cbs.append(fn)
        check(lib.po_tr_set_eigen_model(self._h, int(N), int(index), fn, None))

    def setEigenModelSynthetic(self, N, index, seed=0, curv=1.0):
        check(lib.po_tr_set_eigen_model_synthetic(self._h, int(N), int(index), int(seed), float(curv)))

    def setIterationCallback(self, fn):
        def _cb(user, k):
            fn(k)
            return 0

        cb = L.TR_ITER_FN(_cb)
        self._cbs.append(cb)
        check(lib.po_tr_set_iteration_callback(self._h, cb, None))

    def optimize(self, ip=None):
        """optimize() for the self-assembled form; optimize(ip) with an InteriorPoint built on the subproblem for the
        reference's form (ParOptTrustRegion::optimize(ParOptInteriorPoint*), .cpp:2365-2384)."""
        if ip is not None:
            self._ip = ip
            rc = lib.po_tr_optimize_with(self._h, ip._h)
        else:
            rc = lib.po_tr_optimize(self._h)
        if hasattr(self.problem, "_raise_pending"):
            self.problem._raise_pending()
        if rc != 0:
            raise L.ParOptAMDError(rc, lib.po_last_error().decode(errors="replace"))
        return rc

    def initialize(self):
        check(lib.po_tr_initialize(self._h))

    def setPenaltyGamma(self, gamma):
        if np.isscalar(gamma):
            check(lib.po_tr_set_penalty_gamma(self._h, float(gamma)))
        else:
            ga = (C.c_double * len(gamma))(*[float(v) for v in gamma])
            check(lib.po_tr_set_penalty_gamma_array(self._h, ga))

    def getOptimizedPoint(self):
        x, z, zw = L.po_vec(), L.c_double_p(), L.po_vec()
        check(lib.po_tr_get_optimized_point(self._h, C.byref(x), C.byref(z), C.byref(zw)))
        c = self.problem.ncon
        return (PVec(self.ctx, handle=x, owned=False), np.array([z[i] for i in range(c)]),
                PVec(self.ctx, handle=zw, owned=False) if zw else None)

    def getState(self):
        tr, it, si, ai, fk = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
        pg, ck = L.c_double_p(), L.c_double_p()
        check(lib.po_tr_get_state(self._h, C.byref(tr), C.byref(it), C.byref(si), C.byref(ai), C.byref(pg),
                                  C.byref(fk), C.byref(ck)))
        c = self.problem.ncon
        return dict(tr_size=tr.value, iter_count=it.value, subproblem_iters=si.value,
                    adaptive_subproblem_iters=ai.value, penalty_gamma=np.array([pg[i] for i in range(c)]),
                    fk=fk.value, ck=np.array([ck[i] for i in range(c)]))

    def getLastRow(self):
        row, info = L.c_double_p(), C.c_char_p()
        check(lib.po_tr_get_last_row(self._h, C.byref(row), C.byref(info)))
        return [row[i] for i in range(12)], info.value.decode().split()

    def getLastSolveLines(self):
        """Last iteration-table line of the steering (or restoration) solve and of the QP solve of the latest
        trust-region iteration: how each interior-point solve ended."""
        a, b = C.c_char_p(), C.c_char_p()
        check(lib.po_tr_get_last_solve_lines(self._h, C.byref(a), C.byref(b)))
        return (a.value or b"").decode(), (b.value or b"").decode()

    def getHistory(self):
        t = C.c_char_p()
        check(lib.po_tr_get_history(self._h, C.byref(t)))
        return t.value.decode()

    def getQuasiNewton(self):
        h = L.po_qn()
        check(lib.po_tr_get_quasi_newton(self._h, C.byref(h)))
        return _QuasiNewton(self.ctx, 0, 0, 0, handle=h) if h else None

    def getModelVectors(self):
        xk, gk = L.po_vec(), L.po_vec()
        check(lib.po_tr_get_model_vectors(self._h, C.byref(xk), C.byref(gk)))
        return PVec(self.ctx, handle=xk, owned=False), PVec(self.ctx, handle=gk, owned=False)

    def snapshot(self):
        """State in the layout of the golden 'trNNN/' records (oracle/ref_driver.cpp TrHook)."""
        s = self.getState()
        xk, gk = self.getModelVectors()
        d = dict(tr_size=s["tr_size"], penalty_gamma=s["penalty_gamma"], fk=s["fk"], ck=s["ck"],
                 iters=np.array([s["iter_count"], s["subproblem_iters"], s["adaptive_subproblem_iters"]]),
                 norms=np.array([xk.norm(), gk.norm()]), qn_size=0, qn_b0=0.0)
        qn = self.getQuasiNewton()
        if qn is not None:
            k, b0 = C.c_int(), C.c_double()
            check(lib.po_qn_get_compact(qn._h, C.byref(k), C.byref(b0), None, None, None))
            d["qn_size"], d["qn_b0"] = k.value, b0.value
        return d


class MMA:
    """ParOptMMA (reference src/ParOptMMA.h:22-192) with its interior-point sub-solver, assembled the
    way ParOptOptimizer does for algorithm='mma'.  `options` may mix interior-point and mma_* names."""

    def __init__(self, problem, options=None):
        self.problem = problem
        self.ctx = problem.ctx
        self._h = L.po_mma()
        check(lib.po_mma_create(problem.handle, C.byref(self._h)))
        self._cbs = []
        opts = dict(options or {})
        opts.setdefault("mma_output_file", "")
        for k, v in opts.items():
            self.setOption(k, v)

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib.po_mma_destroy(self._h)
        except Exception:
            pass

    def setOption(self, name, value):
        nm = name.encode()
        if isinstance(value, bool):
            check(lib.po_mma_set_option_int(self._h, nm, int(value)))
        elif isinstance(value, int):
            rc = lib.po_mma_set_option_int(self._h, nm, value)
            if rc != 0:
                check(lib.po_mma_set_option_float(self._h, nm, float(value)))
        elif isinstance(value, float):
            check(lib.po_mma_set_option_float(self._h, nm, value))
        else:
            check(lib.po_mma_set_option_str(self._h, nm, str(value).encode()))

    def setIterationCallback(self, fn):
        def _cb(user, k):
            fn(k)
            return 0

        cb = L.TR_ITER_FN(_cb)
        self._cbs.append(cb)
        check(lib.po_mma_set_iteration_callback(self._h, cb, None))

    def optimize(self):
        rc = lib.po_mma_optimize(self._h)
        if hasattr(self.problem, "_raise_pending"):
            self.problem._raise_pending()
        if rc != 0:
            raise L.ParOptAMDError(rc, lib.po_last_error().decode(errors="replace"))
        return rc

    def getOptimizedPoint(self):
        x, z, zw, zl, zu = L.po_vec(), L.c_double_p(), L.po_vec(), L.po_vec(), L.po_vec()
        check(lib.po_mma_get_optimized_point(self._h, C.byref(x), C.byref(z), C.byref(zw), C.byref(zl), C.byref(zu)))
        c = self.problem.ncon
        wrap = lambda h: PVec(self.ctx, handle=h, owned=False) if h else None  # noqa: E731
        return wrap(x), np.array([z[i] for i in range(c)]), wrap(zw), wrap(zl), wrap(zu)

    def getAsymptotes(self):
        lo, up = L.po_vec(), L.po_vec()
        check(lib.po_mma_get_asymptotes(self._h, C.byref(lo), C.byref(up)))
        return PVec(self.ctx, handle=lo, owned=False), PVec(self.ctx, handle=up, owned=False)

    def getState(self):
        a, b, f, cons = C.c_int(), C.c_int(), C.c_double(), L.c_double_p()
        check(lib.po_mma_get_state(self._h, C.byref(a), C.byref(b), C.byref(f), C.byref(cons)))
        c = self.problem.ncon
        return dict(mma_iter=a.value, subproblem_iter=b.value, fobj=f.value, cons=np.array([cons[i] for i in range(c)]))

    def getLastRow(self):
        row = L.c_double_p()
        check(lib.po_mma_get_last_row(self._h, C.byref(row)))
        return [row[i] for i in range(5)]

    def getHistory(self):
        t = C.c_char_p()
        check(lib.po_mma_get_history(self._h, C.byref(t)))
        return t.value.decode()


def wgram(d, vecs, rhs_last=False):
    """W = P^T diag(d) P; rhs_last: the last vector is pre-weighted (its row / column are plain dots)."""
    nv = len(vecs)
    W = np.zeros((nv, nv))
    arr = (L.po_vec * max(nv, 1))(*[v.handle for v in vecs])
    fn = lib.po_wgram_with_rhs if rhs_last else lib.po_wgram
    check(fn(d.handle, arr, nv, W.ctypes.data_as(L.c_double_p)))
    return W.T  # column-major symmetric


def group_panel(d, vecs, nwcon, nw, skip, alpha, U):
    """U_j = alpha * (sums of d o vecs_j over groups of nw consecutive variables, period nw + skip): the structured
    sparse-Jacobian panel image in a pass of its own."""
    nv = len(vecs)
    arr = (L.po_vec * max(nv, 1))(*[v.handle for v in vecs])
    uarr = (L.po_vec * max(nv, 1))(*[u.handle for u in U])
    check(lib.po_group_panel(d.handle, arr, nv, nwcon, nw, skip, alpha, uarr))


def wgram_with_groups(d, vecs, nwcon, nw, skip, alpha, U, rhs_last=False):
    """wgram() with the panel image of the first len(U) columns riding in the same pass; returns (W, fused)."""
    nv = len(vecs)
    W = np.zeros((nv, nv))
    arr = (L.po_vec * max(nv, 1))(*[v.handle for v in vecs])
    uarr = (L.po_vec * max(len(U), 1))(*[u.handle for u in U])
    fused = C.c_int(0)
    check(lib.po_wgram_with_groups(d.handle, arr, nv, 1 if rhs_last else 0, nwcon, nw, skip, alpha, uarr, len(U),
                                   W.ctypes.data_as(L.c_double_p), C.byref(fused)))
    return W.T, bool(fused.value)


def bench_mdot(x, vecs, reps=10):
    nv = len(vecs)
    arr = (L.po_vec * max(nv, 1))(*[v.handle for v in vecs])
    ms = C.c_double()
    out = np.zeros(max(nv, 1))
    check(lib.po_bench_mdot(x.handle, arr, nv, int(reps), C.byref(ms), out.ctypes.data_as(L.c_double_p)))
    return ms.value, out[:nv]


def bench_kernels(ctx, n, c, k, reps=5):
    """Every hot kernel of an interior-point iteration timed in isolation (po_bench_kernels): list of dicts."""
    import json

    buf = C.create_string_buffer(16384)
    check(lib.po_bench_kernels(ctx.handle, int(n), int(c), int(k), int(reps), buf, len(buf)))
    return json.loads(buf.value.decode())


def bench_vec_api(ctx, n, reps=10):
    """Roofline rows of the vector API and the quasi-Newton products at size n (po_bench_vec_api): list of dicts."""
    import json

    buf = C.create_string_buffer(32768)
    check(lib.po_bench_vec_api(ctx.handle, int(n), int(reps), buf, len(buf)))
    return json.loads(buf.value.decode())


def bench_stream(x, y, kind, reps=10):
    """Average kernel ms of a read-only stream (kind 0: x.y) or a copy (kind 1: y <- x), HIP events."""
    ms = C.c_double()
    check(lib.po_bench_stream(x.handle, y.handle, int(kind), int(reps), C.byref(ms)))
    return ms.value


def bench_wgram(d, vecs, reps=10):
    nv = len(vecs)
    arr = (L.po_vec * max(nv, 1))(*[v.handle for v in vecs])
    ms = C.c_double()
    check(lib.po_bench_wgram(d.handle, arr, nv, int(reps), C.byref(ms)))
    return ms.value
