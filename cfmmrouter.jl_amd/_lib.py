"""ctypes binding of libcfmm_amd.so (include/cfmm_amd.h).

This is the only door from the Python host mirror into the device code.  There is NO fallback:
if the shared library is missing or no gfx950 device is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CFMM_AMD_LIB") or os.path.join(_HERE, "libcfmm_amd.so")   # override: A/B builds in scripts/
CSRC = os.path.join(_HERE, "csrc")

OK = 0
ERR_INVALID_ARG = -1
ERR_HIP = -2
ERR_STATE = -3
ERR_UNSUPPORTED = -4

KIND_PRODUCT, KIND_GEOMEAN, KIND_UNIV3 = 0, 1, 2

_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_ctx = C.c_void_p

OBJ_LINEAR_NONNEGATIVE, OBJ_BASKET_LIQUIDATION = 0, 1


class RouteInfo(C.Structure):
    _fields_ = [("f", C.c_double), ("proj_grad", C.c_double), ("iterations", C.c_int32),
                ("evaluations", C.c_int32), ("sweeps", C.c_int32), ("status", C.c_int32),
                ("sweep_seconds", C.c_double), ("total_seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PolishInfo(C.Structure):
    _fields_ = [("residual0", C.c_double), ("residual", C.c_double), ("iterations", C.c_int32), ("sweeps", C.c_int32),
                ("total_seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


FG_CALLBACK = C.CFUNCTYPE(C.c_double, C.c_void_p, _f64p, _f64p)


class ArgumentError(ValueError):
    """The reference's ArgumentError (src/cfmms.jl:77-78, src/objectives.jl:54,97)."""


class CFMMDeviceError(RuntimeError):
    """A HIP runtime failure or a missing device/extension."""


def build(force: bool = False) -> str:
    """Compile libcfmm_amd.so for gfx950 with hipcc (csrc/Makefile)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "cfmm_amd.h"))
    stale = not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), check=True)
    return LIB_PATH


_lib = None


def lib():
    """Load the extension (once).  Raises CFMMDeviceError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CFMMDeviceError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cfmmrouter.jl_amd/csrc`).  There is no CPU fallback.")
    # One HIP runtime per process: if torch is going to be used (streams, torch.distributed), its
    # bundled libamdhip64 must be the one that is loaded, so import it before our library.
    if "torch" not in sys.modules and os.environ.get("CFMM_AMD_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    L.cfmm_ctx_create.argtypes = [C.c_int, C.c_int32, C.POINTER(_ctx)]
    L.cfmm_ctx_create_multi.argtypes = [C.c_int32, _i32p, C.c_int32, C.POINTER(_ctx)]
    L.cfmm_device_count.argtypes = [_ctx]
    L.cfmm_device_count.restype = C.c_int32
    L.cfmm_ctx_destroy.argtypes = [_ctx]
    L.cfmm_ctx_destroy.restype = None
    L.cfmm_last_error.argtypes = [_ctx]
    L.cfmm_last_error.restype = C.c_char_p
    L.cfmm_version.restype = C.c_char_p
    L.cfmm_set_stream.argtypes = [_ctx, C.c_void_p]
    L.cfmm_reset_stream.argtypes = [_ctx]
    L.cfmm_set_option.argtypes = [_ctx, C.c_char_p, C.c_int64]
    L.cfmm_get_option.argtypes = [_ctx, C.c_char_p, _i64p]
    L.cfmm_pools_add_product.argtypes = [_ctx, C.c_int64, _f64p, _f64p, _i32p]
    L.cfmm_pools_add_geomean.argtypes = [_ctx, C.c_int64, _f64p, _f64p, _f64p, _i32p]
    L.cfmm_pools_add_univ3.argtypes = [_ctx, C.c_int64, _f64p, _f64p, _i32p, _i64p, _f64p, _f64p]
    L.cfmm_pools_clear.argtypes = [_ctx]
    L.cfmm_pools_count.argtypes = [_ctx]
    L.cfmm_pools_count.restype = C.c_int64
    L.cfmm_n_tokens.argtypes = [_ctx]
    L.cfmm_n_tokens.restype = C.c_int32
    L.cfmm_find_arb.argtypes = [_ctx, _f64p]
    L.cfmm_eval.argtypes = [_ctx, _f64p, _f64p, _f64p]
    L.cfmm_get_trades.argtypes = [_ctx, _f64p, _f64p]
    L.cfmm_get_trades_range.argtypes = [_ctx, C.c_int32, C.c_int64, C.c_int64, _f64p, _f64p]
    L.cfmm_netflows.argtypes = [_ctx, _f64p]
    L.cfmm_dual_value.argtypes = [_ctx, _f64p]
    L.cfmm_update_reserves.argtypes = [_ctx]
    L.cfmm_get_reserves.argtypes = [_ctx, C.c_int32, _f64p]
    L.cfmm_get_prices.argtypes = [_ctx, C.c_int32, _f64p]
    L.cfmm_sweep_dev.argtypes = [_ctx, C.c_void_p, C.c_void_p, C.c_int]
    L.cfmm_trades_dev.argtypes = [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.cfmm_kernel_times.argtypes = [_ctx, _i64p, _f64p, _i64p, _f64p]
    L.cfmm_route.argtypes = [_ctx, C.c_int32, _f64p, C.c_int32, _f64p, C.c_int32, C.c_double, C.c_double,
                             C.c_int32, C.c_int32, _f64p, _f64p, C.POINTER(RouteInfo)]
    L.cfmm_polish.argtypes = [_ctx, C.c_int32, _f64p, C.c_int32, _f64p, C.c_int32, C.c_double, _f64p, C.POINTER(PolishInfo)]
    L.cfmm_lbfgsb_minimize.argtypes = [C.c_int32, _f64p, _f64p, _f64p, _i32p, FG_CALLBACK, C.c_void_p, C.c_int32,
                                       C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.POINTER(RouteInfo)]
    L.cfmm_set_peers.argtypes = [_ctx, C.POINTER(C.c_uint64), C.c_int32, C.c_int32, C.c_uint64]
    L.cfmm_peer_buffer_bytes.argtypes = [C.c_int32]
    L.cfmm_peer_buffer_bytes.restype = C.c_int64
    L.cfmm_peer_buffer_alloc.argtypes = [_ctx, C.POINTER(C.c_uint64), C.c_char_p]
    L.cfmm_peer_buffer_open.argtypes = [_ctx, C.c_char_p, C.POINTER(C.c_uint64)]
    L.cfmm_peer_buffer_close.argtypes = [_ctx, C.c_uint64]
    L.cfmm_peer_buffer_free.argtypes = [_ctx, C.c_uint64]
    L.cfmm_rccl_unique_id.argtypes = [C.c_char_p]
    L.cfmm_rccl_init_rank.argtypes = [_ctx, C.c_char_p, C.c_int32, C.c_int32]
    L.cfmm_set_rccl_comm.argtypes = [_ctx, C.c_void_p]
    L.cfmm_segment_count.argtypes = [_ctx]
    L.cfmm_segment_count.restype = C.c_int32
    L.cfmm_segment_info.argtypes = [_ctx, C.c_int32, _i32p, _i64p, _i32p, _i32p]
    _lib = L
    return L


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def ptr(a):
    if a is None:
        return None
    if a.dtype == np.float64:
        return a.ctypes.data_as(_f64p)
    if a.dtype == np.int32:
        return a.ctypes.data_as(_i32p)
    if a.dtype == np.int64:
        return a.ctypes.data_as(_i64p)
    raise TypeError(a.dtype)


class Context:
    """RAII wrapper of a cfmm_ctx: one device and its pool store, or -- `device` a list of HIP
    ordinals -- the single-process multi-device context of cfmm_ctx_create_multi (pools split in
    contiguous blocks over the devices, {Ψ, acc} summed on the host)."""

    def __init__(self, n_tokens: int, device=0):
        self._L = lib()
        h = _ctx()
        if isinstance(device, (list, tuple, np.ndarray)):
            ids = np.ascontiguousarray(device, dtype=np.int32)
            rc = self._L.cfmm_ctx_create_multi(int(ids.size), ptr(ids), int(n_tokens), C.byref(h))
            self.device = [int(d) for d in ids]
        else:
            rc = self._L.cfmm_ctx_create(int(device), int(n_tokens), C.byref(h))
            self.device = int(device)
        if rc != OK:
            self._h = None
            self._raise(rc, None)
        self._h = h
        self.n_tokens = int(n_tokens)

    @property
    def device_count(self) -> int:
        return int(self._L.cfmm_device_count(self._h))

    # -- errors --------------------------------------------------------------------------------
    def _raise(self, rc, h):
        msg = self._L.cfmm_last_error(h).decode()
        if rc == ERR_INVALID_ARG:
            raise ArgumentError(msg)
        if rc == ERR_STATE:
            raise RuntimeError(msg)
        if rc == ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise CFMMDeviceError(msg)

    def _check(self, rc):
        if rc != OK:
            self._raise(rc, self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.cfmm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        """Launch on exactly this hipStream_t handle (0 / None = HIP's default stream)."""
        self._check(self._L.cfmm_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def reset_stream(self):
        self._check(self._L.cfmm_reset_stream(self._h))

    def set_option(self, key: str, value: int):
        self._check(self._L.cfmm_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        self._check(self._L.cfmm_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    # -- pools ---------------------------------------------------------------------------------
    def add_product(self, R, gamma, Ai0):
        R, gamma = f64(R), f64(gamma)
        Ai0 = np.ascontiguousarray(Ai0, dtype=np.int32)
        m = gamma.size
        if R.size != 2 * m or Ai0.size != 2 * m:
            raise ArgumentError("R and Ai must have shape [m, 2]")
        self._check(self._L.cfmm_pools_add_product(self._h, m, ptr(R), ptr(gamma), ptr(Ai0)))

    def add_geomean(self, R, w, gamma, Ai0):
        R, w, gamma = f64(R), f64(w), f64(gamma)
        Ai0 = np.ascontiguousarray(Ai0, dtype=np.int32)
        m = gamma.size
        if R.size != 2 * m or w.size != 2 * m or Ai0.size != 2 * m:
            raise ArgumentError("R, w and Ai must have shape [m, 2]")
        self._check(self._L.cfmm_pools_add_geomean(self._h, m, ptr(R), ptr(w), ptr(gamma), ptr(Ai0)))

    def add_univ3(self, current_price, gamma, Ai0, tick_off, lower_ticks, liquidity):
        current_price, gamma = f64(current_price), f64(gamma)
        lower_ticks, liquidity = f64(lower_ticks), f64(liquidity)
        Ai0 = np.ascontiguousarray(Ai0, dtype=np.int32)
        tick_off = np.ascontiguousarray(tick_off, dtype=np.int64)
        m = gamma.size
        if current_price.size != m or Ai0.size != 2 * m or tick_off.size != m + 1:
            raise ArgumentError("inconsistent UniV3 array sizes")
        if m and (lower_ticks.size != tick_off[-1] or liquidity.size != tick_off[-1]):
            raise ArgumentError("tick arrays must have tick_off[-1] entries")
        self._check(self._L.cfmm_pools_add_univ3(self._h, m, ptr(current_price), ptr(gamma), ptr(Ai0),
                                                 ptr(tick_off), ptr(lower_ticks), ptr(liquidity)))

    def clear(self):
        self._check(self._L.cfmm_pools_clear(self._h))

    @property
    def pool_count(self) -> int:
        return int(self._L.cfmm_pools_count(self._h))

    # -- hot path ------------------------------------------------------------------------------
    def find_arb(self, v):
        v = f64(v)
        if v.size != self.n_tokens:
            raise ArgumentError("v must have n_tokens entries")
        self._check(self._L.cfmm_find_arb(self._h, ptr(v)))

    def eval(self, v):
        v = f64(v)
        if v.size != self.n_tokens:
            raise ArgumentError("v must have n_tokens entries")
        psi = np.empty(self.n_tokens)
        acc = C.c_double()
        self._check(self._L.cfmm_eval(self._h, ptr(v), ptr(psi), C.byref(acc)))
        return psi, acc.value

    def trades(self, out=None):
        """r.Δs / r.Λs of the latest materialising sweep as [m, 2] arrays (cfmm_get_trades).  `out` = (Δ, Λ): fill
        caller-owned C-contiguous float64 arrays instead of allocating (what a binding that owns r.Δs / r.Λs does)."""
        m = self.pool_count
        if out is None:
            D, Lm = np.empty((m, 2)), np.empty((m, 2))
        else:
            D, Lm = out
            for a in (D, Lm):
                if a.dtype != np.float64 or a.shape != (m, 2) or not a.flags.c_contiguous:
                    raise ArgumentError("out arrays must be C-contiguous float64 of shape [m, 2]")
        self._check(self._L.cfmm_get_trades(self._h, ptr(D), ptr(Lm)))
        return D, Lm

    def trades_range(self, seg, first, count):
        D, Lm = np.empty((count, 2)), np.empty((count, 2))
        self._check(self._L.cfmm_get_trades_range(self._h, int(seg), int(first), int(count), ptr(D), ptr(Lm)))
        return D, Lm

    def netflows(self):
        psi = np.empty(self.n_tokens)
        self._check(self._L.cfmm_netflows(self._h, ptr(psi)))
        return psi

    def dual_value(self) -> float:
        acc = C.c_double()
        self._check(self._L.cfmm_dual_value(self._h, C.byref(acc)))
        return acc.value

    def route(self, objective_kind, objective_vec, objective_index=0, v0=None, m=5, factr=1e1, pgtol=1e-5,
              maxfun=15_000, maxiter=15_000):
        """cfmm_route: route! entirely inside the library (own L-BFGS-B).  -> (v, psi, info dict)."""
        ov = f64(objective_vec)
        if ov.size != self.n_tokens:
            raise ArgumentError("objective vector must have n_tokens entries")
        v0a = None if v0 is None else f64(v0)
        v, psi, info = np.empty(self.n_tokens), np.empty(self.n_tokens), RouteInfo()
        self._check(self._L.cfmm_route(self._h, int(objective_kind), ptr(ov), int(objective_index), ptr(v0a),
                                       int(m), float(factr), float(pgtol), int(maxfun), int(maxiter), ptr(v),
                                       ptr(psi), C.byref(info)))
        return v, psi, info.as_dict()

    def polish(self, objective_kind, objective_vec, objective_index, v, max_iters=8, rel_step=1e-7):
        """cfmm_polish: the gradient-only projected chord-Newton polish inside the library.  -> (v, psi, info dict)."""
        ov = f64(objective_vec)
        if ov.size != self.n_tokens:
            raise ArgumentError("objective vector must have n_tokens entries")
        vv, psi, info = f64(np.array(v, dtype=np.float64).copy()), np.empty(self.n_tokens), PolishInfo()
        self._check(self._L.cfmm_polish(self._h, int(objective_kind), ptr(ov), int(objective_index), ptr(vv), int(max_iters),
                                        float(rel_step), ptr(psi), C.byref(info)))
        return vv, psi, info.as_dict()

    def update_reserves(self):
        """update_reserves!(r) on the device (cfmm_update_reserves): consumes the latest materialised trades."""
        self._check(self._L.cfmm_update_reserves(self._h))

    def reserves(self, seg: int, m: int):
        R = np.empty((int(m), 2))
        self._check(self._L.cfmm_get_reserves(self._h, int(seg), ptr(R)))
        return R

    def prices(self, seg: int, m: int):
        p = np.empty(int(m))
        self._check(self._L.cfmm_get_prices(self._h, int(seg), ptr(p)))
        return p

    def set_peers(self, peer_ptrs, world: int, rank: int, seq: int):
        """Sharded operation: every host-pointer sweep of this context ends with the one-shot peer
        all-reduce over the given symmetric buffers (cfmm_set_peers).  world=0 switches it off."""
        arr = (C.c_uint64 * max(world, 1))(*[int(p) for p in peer_ptrs][:max(world, 1)]) if world else None
        self._check(self._L.cfmm_set_peers(self._h, arr, int(world), int(rank), C.c_uint64(int(seq))))

    def peer_buffer_alloc(self):
        """This rank's symmetric buffer for cfmm_set_peers -> (device pointer, 64-byte IPC handle)."""
        p = C.c_uint64()
        h = C.create_string_buffer(64)
        self._check(self._L.cfmm_peer_buffer_alloc(self._h, C.byref(p), h))
        return p.value, h.raw

    def peer_buffer_open(self, handle: bytes) -> int:
        p = C.c_uint64()
        self._check(self._L.cfmm_peer_buffer_open(self._h, C.create_string_buffer(bytes(handle), 64), C.byref(p)))
        return p.value

    def peer_buffer_close(self, ptr_: int):
        self._check(self._L.cfmm_peer_buffer_close(self._h, C.c_uint64(int(ptr_))))

    def peer_buffer_free(self, ptr_: int):
        self._check(self._L.cfmm_peer_buffer_free(self._h, C.c_uint64(int(ptr_))))

    # sharded operation through RCCL (include/cfmm_amd.h): rank 0 makes the id, every rank joins with it
    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = lib().cfmm_rccl_unique_id(buf)
        if rc != OK:
            raise CFMMDeviceError(lib().cfmm_last_error(None).decode())
        return buf.raw

    def rccl_init_rank(self, uid: bytes, world: int, rank: int):
        self._check(lib().cfmm_rccl_init_rank(self._h, uid, int(world), int(rank)))

    def set_rccl_comm(self, comm_ptr):
        self._check(lib().cfmm_set_rccl_comm(self._h, comm_ptr))

    def sweep_dev(self, d_v_ptr: int, d_out_ptr: int, materialize: bool):
        self._check(self._L.cfmm_sweep_dev(self._h, C.c_void_p(d_v_ptr), C.c_void_p(d_out_ptr),
                                           1 if materialize else 0))

    def trades_dev(self):
        a, b = C.c_void_p(), C.c_void_p()
        self._check(self._L.cfmm_trades_dev(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_times(self):
        sn, rn = C.c_int64(), C.c_int64()
        sm, rm = C.c_double(), C.c_double()
        self._check(self._L.cfmm_kernel_times(self._h, C.byref(sn), C.byref(sm), C.byref(rn), C.byref(rm)))
        return {"sweep_launches": sn.value, "sweep_ms": sm.value, "reduce_launches": rn.value,
                "reduce_ms": rm.value}

    def segments(self):
        out = []
        for s in range(self._L.cfmm_segment_count(self._h)):
            k, b, g = C.c_int32(), C.c_int32(), C.c_int32()
            m = C.c_int64()
            self._check(self._L.cfmm_segment_info(self._h, s, C.byref(k), C.byref(m), C.byref(b), C.byref(g)))
            out.append({"kind": k.value, "m": m.value, "block": b.value, "grid": g.value})
        return out


def lbfgsb_minimize(fun, x0, bounds, m=5, factr=1e1, pgtol=1e-5, maxfun=15_000, maxiter=15_000,
                    reference_boxed=False):
    """The library's own L-BFGS-B on a Python objective `fun(x) -> (f, g)` (host only; used by the
    CPU tests to compare the solver with SciPy's).  bounds: list of (lo, hi) with None = unbounded.
    reference_boxed=True passes nbd = 2 for every variable (infinite bounds included) and lets the
    solver take the Fortran code's "boxed" first step, as the reference's call does."""
    n = len(x0)
    x = f64(np.array(x0, dtype=np.float64).copy())
    lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=np.float64)
    hi = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=np.float64)
    nbd = np.array([(1 if np.isfinite(l) else 0) + (2 if np.isfinite(h) else 0) for l, h in zip(lo, hi)])
    nbd = np.array([{0: 0, 1: 1, 3: 2, 2: 3}[int(k)] for k in nbd], dtype=np.int32)
    if reference_boxed:
        nbd[:] = 2

    def cb(_user, xp, gp):
        xx = np.ctypeslib.as_array(xp, shape=(n,))
        fval, g = fun(xx.copy())
        np.ctypeslib.as_array(gp, shape=(n,))[:] = g
        return float(fval)

    info = RouteInfo()
    rc = lib().cfmm_lbfgsb_minimize(n, ptr(x), ptr(lo), ptr(hi), ptr(nbd), FG_CALLBACK(cb), None, int(m),
                                    float(factr), float(pgtol), int(maxfun), int(maxiter),
                                    1 if reference_boxed else 0, C.byref(info))
    if rc != OK:
        raise ArgumentError(lib().cfmm_last_error(None).decode())
    return x, info.as_dict()
