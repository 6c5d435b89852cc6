"""ctypes binding of libgpet_hip.so (include/gpet_hip.h).

The HIP library is the product's only compute path: if it is missing or fails to load this
module raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPET_LIB_PATH") or os.path.join(_HERE, "libgpet_hip.so")  # (override: instrumented builds)

# status codes (gpet_status)
OK, ERR_BAD_ARG, ERR_HIP, ERR_NOT_PD, ERR_ITER_CAP, ERR_RANK_CAP, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_STATE = range(9)

# gpet_buf
(BUF_X_TRAIN, BUF_Y_TRAIN, BUF_CHOL, BUF_ALPHA, BUF_MEAN, BUF_STD, BUF_COV, BUF_FACTOR, BUF_EIGVALS, BUF_NORMALS,
 BUF_SAMPLES, BUF_COSTS, BUF_BEST_IDX, BUF_BEST_COSTS, BUF_SCALARS, BUF_OBS, BUF_KDE, BUF_GRAD_KDE, BUF_GRAD,
 BUF_NOISE_W, BUF_FIN_TRAIN, BUF_FIN_PAR, BUF_FIN_STARTS) = range(23)

KERNEL_RBF, KERNEL_MATERN = 0, 1
GRAD_ON_DEVICE = 1  # gpet_batch_create2 / gpet_batch_set_images flag: the gradient image pointers are device pointers
IMAGES_NEXT_FRAME = 2  # gpet_batch_set_images: the images continue the sequences just traced


class GpetParams(C.Structure):
    _fields_ = [("kernel_type", C.c_int32), ("nu", C.c_double), ("sigma_f", C.c_double),
                ("length_scale", C.c_double), ("noise_y", C.c_double), ("n_samples", C.c_int32),
                ("n_keep", C.c_int32), ("delta_x", C.c_int32), ("pixel_thresh", C.c_int32),
                ("score_thresh", C.c_double), ("fix_endpoints", C.c_int32), ("x_st", C.c_int32),
                ("x_en", C.c_int32), ("n_init", C.c_int32), ("obs_cap", C.c_int32), ("factor_cap", C.c_int32),
                ("z_cols", C.c_int32), ("jitter", C.c_double)]


class GpetScalars(C.Structure):
    _fields_ = [("y_s", C.c_double), ("amp", C.c_double), ("y_mean", C.c_double), ("y_std", C.c_double),
                ("score_thresh", C.c_double), ("lml", C.c_double), ("n", C.c_int32), ("n_obs", C.c_int32),
                ("rank", C.c_int32), ("status", C.c_int32), ("iter", C.c_int32), ("done", C.c_int32),
                ("n_removed", C.c_int32), ("force", C.c_int32)]


class GpetError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgpet_hip status {code}: {msg}")
        self.code = code


# every symbol include/gpet_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "gpet_abi_version": (C.c_int, []),
    "gpet_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "gpet_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "gpet_option_count": (C.c_int, []),
    "gpet_option_info": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "gpet_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "gpet_ctx_destroy": (None, [_P]),
    "gpet_last_error": (C.c_char_p, [_P]),
    "gpet_sync": (C.c_int, [_P]),
    "gpet_ctx_stream": (_P, [_P]),
    "gpet_timer_start": (C.c_int, [_P]),
    "gpet_timer_stop_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "gpet_grad_image": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "gpet_normalise_f32": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "gpet_batch_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P), C.c_int,
                                    C.POINTER(GpetParams), C.POINTER(_P), C.POINTER(_P)]),
    "gpet_batch_create2": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P), C.c_int,
                                     C.POINTER(GpetParams), C.POINTER(_P), C.c_uint, C.POINTER(_P)]),
    "gpet_batch_set_images": (C.c_int, [_P, C.POINTER(_P), C.c_uint]),
    "gpet_batch_destroy": (None, [_P]),
    "gpet_batch_size": (C.c_int, [_P]),
    "gpet_batch_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "gpet_batch_reset": (C.c_int, [_P]),
    "gpet_profile_stage": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "gpet_batch_set_obs": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "gpet_batch_read": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t]),
    "gpet_batch_write": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t, C.c_int]),
    "gpet_batch_clear_injected_factor": (C.c_int, [_P, C.c_int]),
    "gpet_gp_fit_predict": (C.c_int, [_P, C.c_int]),
    "gpet_gp_factor": (C.c_int, [_P]),
    "gpet_gp_normals": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "gpet_gp_sample": (C.c_int, [_P]),
    "gpet_score_curves": (C.c_int, [_P]),
    "gpet_select_pixels": (C.c_int, [_P]),
    "gpet_curve_kde": (C.c_int, [_P]),
    "gpet_final_cov": (C.c_int, [_P]),
    "gpet_select_pixels_only": (C.c_int, [_P]),
    "gpet_final_set_training": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int]),
    "gpet_lml_batch": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "gpet_lml_stats": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "gpet_final_set_training_all": (C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    "gpet_batch_read_obs_all": (C.c_int, [_P, _P, _P, C.c_int]),
    "gpet_batch_read_scalars_all": (C.c_int, [_P, _P]),
    "gpet_final_predict_all": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "gpet_final_fit_all": (C.c_int, [_P, C.POINTER(C.c_uint32), _P, _P, _P, C.c_int, C.POINTER(C.c_int32)]),
    "gpet_final_optimize": (C.c_int, [_P, C.c_int, _P, _P, _P, C.POINTER(C.c_int32)]),
    "gpet_batch_set_sample_dtype": (C.c_int, [_P, C.c_int]),
    "gpet_batch_set_rng": (C.c_int, [_P, C.c_int]),
    "gpet_batch_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "gpet_batch_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int)]),
    "gpet_trace_iterate": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int)]),
    "gpet_comm_unique_id": (C.c_int, [_P]),
    "gpet_comm_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "gpet_comm_destroy": (None, [_P]),
    "gpet_comm_rank": (C.c_int, [_P]),
    "gpet_comm_world": (C.c_int, [_P]),
    "gpet_comm_block": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gpet_bcast_grad": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "gpet_dev_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "gpet_dev_free": (C.c_int, [_P, _P]),
    "gpet_dev_copy": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int]),
    "gpet_allgather_i64": (C.c_int, [_P, _P, _P, _P]),
    "gpet_gather_traces": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
}
COMM_ID_BYTES = 128

_lib = None


def get_option(name):
    """Current value of a process-wide tuning switch (gpet_get_option; -1 = chosen automatically)."""
    v = C.c_int()
    if load().gpet_get_option(name.encode(), C.byref(v)) != 0:
        raise ValueError("unknown option %r" % (name,))
    return v.value


_live_batches = []  # weak references to the Batch objects of this process (set_option(..., live_batches=True))


def set_option(name, value, live_batches=True):
    """Process-wide tuning switch of the library (gpet_set_option); returns the previous value (-1 = automatic).
    The library gives every batch its own copy of the table when the batch is created (gpet_batch_set_option changes that
    copy), so a switch set here reaches only batches created afterwards -- unless ``live_batches`` (the default, what tools
    and tests that flip a switch on an existing object mean): then it is also written into every live Batch of this process."""
    old = get_option(name)
    if load().gpet_set_option(name.encode(), int(value)) < 0:
        raise ValueError("unknown option %r" % (name,))
    if live_batches:
        for ref in list(_live_batches):
            b = ref()
            if b is None or not getattr(b, "h", None):
                _live_batches.remove(ref)
            else:
                b.set_option(name, value)
    return old


def options():
    """The table of tuning switches: {name: dict(value, default, lo, hi, doc)} (gpet_option_info)."""
    lib = load()
    out = {}
    for i in range(lib.gpet_option_count()):
        name, doc = C.c_char_p(), C.c_char_p()
        v, d, lo, hi = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib.gpet_option_info(i, C.byref(name), C.byref(v), C.byref(d), C.byref(lo), C.byref(hi), C.byref(doc))
        out[name.value.decode()] = dict(value=v.value, default=d.value, lo=lo.value, hi=hi.value, doc=doc.value.decode())
    return out


def load():
    """Load libgpet_hip.so (once).  Raises if it is absent: the product has no other path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  gaussian_process_edge_trace_amd has no CPU fallback.")
    # Deployment setting of the package: eight HIP hardware queues per process instead of the runtime's four.  A tracer drives
    # three streams per batch object (loop, RNG look-ahead, converged fits) and a server keeps several objects in flight; on four
    # queues they alias (+2-3 % throughput at 1 024 edges, +9 % at 256 with eight: profiles/r05_hw_queues.txt; 12 and more
    # hurt the latency of small batches).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the
    # process: a value already in the environment wins, and a process that touched the GPU before importing this keeps its own.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.gpet_abi_version() != 1:
        raise ImportError("libgpet_hip.so ABI version mismatch")
    _lib = lib
    return lib


class Context:
    """gpet_ctx: one device + one HIP stream."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = _P()
        rc = self.lib.gpet_ctx_create(int(device), _P(stream) if stream else None, C.byref(h))
        if rc == ERR_NO_DEVICE:
            raise GpetError(rc, "no HIP device visible (this package needs an MI355X; there is no CPU fallback)")
        if rc != OK:
            raise GpetError(rc, "gpet_ctx_create failed")
        self.h = h
        self.device = device
        self._comms = []  # (weak references to the communicators built on this context: close() closes them first)

    def check(self, rc):
        if rc != OK:
            raise GpetError(rc, (self.lib.gpet_last_error(self.h) or b"").decode())

    def sync(self):
        self.check(self.lib.gpet_sync(self.h))

    def timer_start(self):
        self.check(self.lib.gpet_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self.check(self.lib.gpet_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def grad_image(self, img, kernel):
        img = np.ascontiguousarray(img, dtype=np.float64)
        kernel = np.ascontiguousarray(kernel, dtype=np.float64)
        out = np.empty(img.shape, dtype=np.float32)
        self.check(self.lib.gpet_grad_image(self.h, img.ctypes.data, img.shape[0], img.shape[1], kernel.ctypes.data,
                                            kernel.shape[0], kernel.shape[1], out.ctypes.data))
        return out

    def normalise_f32(self, img):
        a = np.ascontiguousarray(img, dtype=np.float32)
        out = np.empty_like(a)
        self.check(self.lib.gpet_normalise_f32(self.h, a.ctypes.data, a.size, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "h", None):
            for ref in list(getattr(self, "_comms", [])):  # (their buffers are freed through this context's handle)
                cm = ref()
                if cm is not None:
                    cm.close()
            self._comms = []
            self.lib.gpet_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    """gpet_comm_unique_id: the 128 bytes rank 0 ships to every rank (any transport) before Comm(ctx, id, world, rank)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().gpet_comm_unique_id(buf)
    if rc:
        raise GpetError(rc, "gpet_comm_unique_id failed (is RCCL installed?)")
    return buf.raw


class Comm:
    """gpet_comm: the C ABI's RCCL communicator of one rank (one process per GPU) and its two collectives --
    the broadcast of the shared gradient image(s) into device memory and the gather of the finished traces."""

    def __init__(self, ctx: Context, unique_id, world, rank):
        self.ctx, self.lib = ctx, ctx.lib
        h = _P()
        idbuf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES) if unique_id is not None else None
        ctx.check(self.lib.gpet_comm_create(ctx.h, idbuf, int(world), int(rank), C.byref(h)))
        self.h = h
        self.world, self.rank = int(world), int(rank)
        self._bufs = []
        import weakref
        ctx._comms.append(weakref.ref(self))

    def block(self, n_units):
        lo, hi = C.c_int64(), C.c_int64()
        self.ctx.check(self.lib.gpet_comm_block(self.h, int(n_units), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def bcast_grad(self, grad, shape, root=0):
        """Broadcast float32 image(s) of ``shape`` from ``root`` (``grad`` is read there only); returns the DEVICE pointer the
        collective filled, for GP_Edge_Tracing_Batch(..., grad_device_ptrs=[ptr], grad_shape=...) -- no trip through host memory
        on the receiving ranks.  The buffer lives until close()."""
        count = int(np.prod(shape))
        d = _P()
        self.ctx.check(self.lib.gpet_dev_alloc(self.ctx.h, count * 4, C.byref(d)))
        self._bufs.append(d)
        if self.rank == root:
            a = np.ascontiguousarray(grad, dtype=np.float32).reshape(-1)
            assert a.size == count
            self.ctx.check(self.lib.gpet_dev_copy(self.ctx.h, d, a.ctypes.data, count * 4, 0))
        self.ctx.check(self.lib.gpet_bcast_grad(self.h, d, count, int(root)))
        return d.value

    def download(self, dev_ptr, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        self.ctx.check(self.lib.gpet_dev_copy(self.ctx.h, out.ctypes.data, _P(dev_ptr), out.nbytes, 1))
        return out

    def gather_traces(self, local, n_edges, edge_len):
        """(n_local, edge_len, 2) int64 traces of this rank's block -> (n_edges, edge_len, 2) in global order on every rank."""
        loc = np.ascontiguousarray(np.asarray(local, dtype=np.int64).reshape(-1, int(edge_len), 2))
        lo, hi = self.block(n_edges)
        assert loc.shape[0] == hi - lo, (loc.shape, lo, hi)
        out = np.empty((int(n_edges), int(edge_len), 2), dtype=np.int64)
        self.ctx.check(self.lib.gpet_gather_traces(self.h, loc.ctypes.data if loc.size else None, int(n_edges), int(edge_len),
                                                   out.ctypes.data))
        return out

    def allgather_i64(self, local, counts):
        loc = np.ascontiguousarray(np.asarray(local, dtype=np.int64).reshape(-1))
        cn = np.ascontiguousarray(np.asarray(counts, dtype=np.int64))
        assert cn.size == self.world and loc.size == cn[self.rank]
        out = np.empty(int(cn.sum()), dtype=np.int64)
        self.ctx.check(self.lib.gpet_allgather_i64(self.h, loc.ctypes.data if loc.size else None, cn.ctypes.data, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):  # (a context closed first has already closed this communicator: see Context.close)
                for d in self._bufs:
                    self.lib.gpet_dev_free(self.ctx.h, d)
            self._bufs = []
            self.lib.gpet_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DT = {BUF_X_TRAIN: np.float64, BUF_Y_TRAIN: np.float64, BUF_CHOL: np.float64, BUF_ALPHA: np.float64,
       BUF_MEAN: np.float64, BUF_STD: np.float64, BUF_COV: np.float64, BUF_FACTOR: np.float64,
       BUF_EIGVALS: np.float64, BUF_NORMALS: np.float64, BUF_SAMPLES: np.float64, BUF_COSTS: np.float64,
       BUF_BEST_IDX: np.int32, BUF_BEST_COSTS: np.float64, BUF_OBS: np.int64, BUF_KDE: np.float32,
       BUF_GRAD_KDE: np.float32, BUF_GRAD: np.float32, BUF_NOISE_W: np.float64, BUF_FIN_TRAIN: np.float64,
       BUF_FIN_PAR: np.float64, BUF_FIN_STARTS: np.float64}


class Batch:
    """gpet_batch: B independent edges processed together."""

    def __init__(self, ctx: Context, grads, params, inits, share_image=False, device_ptrs=None, shape=None):
        """``grads``: float32 (M, N) arrays on the host -- or, with ``device_ptrs`` (a list of integer device
        addresses of f32 [M*N] images on the context's device, e.g. ``tensor.data_ptr()`` after an RCCL broadcast) and
        ``shape`` = (M, N), nothing on the host at all: the library consumes the device images in place."""
        self.ctx = ctx
        self.lib = ctx.lib
        B = len(params)
        inits = [np.ascontiguousarray(i, dtype=np.int64) for i in inits]
        if device_ptrs is not None:
            self.M, self.N = int(shape[0]), int(shape[1])
            gp = (_P * len(device_ptrs))(*[int(p) for p in device_ptrs])
            flags = GRAD_ON_DEVICE
            grads = None
        else:
            grads = [np.ascontiguousarray(g, dtype=np.float32) for g in grads]
            self.M, self.N = grads[0].shape
            gp = (_P * len(grads))(*[g.ctypes.data for g in grads])
            flags = 0
        ip = (_P * B)(*[i.ctypes.data for i in inits])
        pa = (GpetParams * B)(*params)
        h = _P()
        ctx.check(self.lib.gpet_batch_create2(ctx.h, B, self.M, self.N, gp, 1 if share_image else 0, pa, ip, flags,
                                              C.byref(h)))
        self.h = h
        self.B = B
        import weakref
        _live_batches.append(weakref.ref(self))
        if len(_live_batches) > 4096:  # (drop the references of batches that are gone)
            _live_batches[:] = [r for r in _live_batches if r() is not None and getattr(r(), "h", None)]
        self._scored = False
        self.share_image = bool(share_image)
        self._keep = (grads, inits)

    def set_images(self, grads=None, device_ptrs=None, next_frame=False):
        """New gradient image(s) for the same edges, gradient KDE recomputed, loop state reset (gpet_batch_set_images).
        ``next_frame``: the images are the next frames of the sequences just traced, so an any-rank factor may start from
        the last trace's rows (GPET_IMAGES_NEXT_FRAME); otherwise nothing of an earlier trace is used."""
        n_img = 1 if self.share_image else self.B
        nf = IMAGES_NEXT_FRAME if next_frame else 0
        if device_ptrs is not None:
            assert len(device_ptrs) == n_img
            gp = (_P * n_img)(*[int(p) for p in device_ptrs])
            self.ctx.check(self.lib.gpet_batch_set_images(self.h, gp, GRAD_ON_DEVICE | nf))
            return
        grads = [np.ascontiguousarray(g, dtype=np.float32) for g in grads]
        assert len(grads) == n_img and all(g.shape == (self.M, self.N) for g in grads)
        gp = (_P * n_img)(*[g.ctypes.data for g in grads])
        self.ctx.check(self.lib.gpet_batch_set_images(self.h, gp, nf))

    def _max_info(self, key):
        """max over the edges of a creation-time constant of gpet_batch_info (cached)."""
        cache = self.__dict__.setdefault("_info_max", {})
        if key not in cache:
            cache[key] = max(self.info(e)[key] for e in range(self.B))
        return cache[key]

    def info(self, e=0):
        v = (C.c_int32 * 14)()
        self.ctx.check(self.lib.gpet_batch_info(self.h, e, v, 14))
        keys = ["Lg", "S", "n_keep", "n_cap", "factor_cap", "z_cols", "factor_rows_cap", "n_bins", "obs_cap",
                "algo_thresh", "structured", "r0", "z_ring", "arena_mib"]
        return dict(zip(keys, list(v)))

    def scalars(self, e=0) -> GpetScalars:
        s = GpetScalars()
        self.ctx.check(self.lib.gpet_batch_read(self.h, e, BUF_SCALARS, C.byref(s), C.sizeof(s)))
        return s

    def write_scalars(self, s, e=0):
        self.ctx.check(self.lib.gpet_batch_write(self.h, e, BUF_SCALARS, C.byref(s), C.sizeof(s), 0))

    @property
    def have_scores(self):
        """True once a scoring pass has left costs / best indices on the device (gpet_score_curves or the loop)."""
        return bool(self._scored)

    def all_scalars(self):
        arr = (GpetScalars * self.B)()
        self.ctx.check(self.lib.gpet_batch_read_scalars_all(self.h, arr))
        return list(arr)

    def set_obs(self, e, obs_xy):
        o = np.ascontiguousarray(np.asarray(obs_xy).reshape(-1, 2), dtype=np.int64)
        self.ctx.check(self.lib.gpet_batch_set_obs(self.h, e, o.ctypes.data if o.size else None, o.shape[0]))

    def read(self, which, e=0):
        inf = self.info(e)
        s = self.scalars(e)
        Lg, S = inf["Lg"], inf["S"]
        shape = {BUF_X_TRAIN: (s.n,), BUF_Y_TRAIN: (s.n,), BUF_NOISE_W: (s.n,), BUF_ALPHA: (s.n,),
                 BUF_CHOL: (s.n, s.n), BUF_MEAN: (Lg,), BUF_STD: (Lg,), BUF_COV: (Lg, Lg),
                 BUF_FACTOR: (s.rank, Lg), BUF_EIGVALS: (s.rank,), BUF_NORMALS: (S, inf["z_cols"]),
                 BUF_SAMPLES: (S, Lg), BUF_COSTS: (S,), BUF_BEST_IDX: (inf["n_keep"],),
                 BUF_BEST_COSTS: (inf["n_keep"],), BUF_OBS: (s.n_obs, 2), BUF_KDE: (self.M, self.N),
                 BUF_GRAD_KDE: (self.M, self.N), BUF_GRAD: (self.M, self.N), BUF_FIN_TRAIN: (3, inf["n_cap"]),
                 BUF_FIN_PAR: (12,), BUF_FIN_STARTS: (13, 3)}[which]
        out = np.zeros(shape, dtype=_DT[which])
        if out.size:
            self.ctx.check(self.lib.gpet_batch_read(self.h, e, which, out.ctypes.data, out.nbytes))
        return out

    def write(self, which, arr, e=0, rows=0):
        a = np.ascontiguousarray(arr, dtype=_DT[which])
        self.ctx.check(self.lib.gpet_batch_write(self.h, e, which, a.ctypes.data, a.nbytes, int(rows)))

    def clear_injected_factor(self, e=0):
        self.ctx.check(self.lib.gpet_batch_clear_injected_factor(self.h, e))

    def fit_predict(self, want_cov=True):
        self.ctx.check(self.lib.gpet_gp_fit_predict(self.h, 1 if want_cov else 0))

    def factor(self):
        self.ctx.check(self.lib.gpet_gp_factor(self.h))

    def normals(self, seeds):
        s = (C.c_uint32 * self.B)(*[int(v) & 0xFFFFFFFF for v in seeds])
        self.ctx.check(self.lib.gpet_gp_normals(self.h, s))

    def sample(self):
        self.ctx.check(self.lib.gpet_gp_sample(self.h))

    def score(self):
        self.ctx.check(self.lib.gpet_score_curves(self.h))
        self._scored = True

    def reset(self):
        self.ctx.check(self.lib.gpet_batch_reset(self.h))

    def profile_stage(self, stage, reps=10):
        ms = C.c_float()
        self.ctx.check(self.lib.gpet_profile_stage(self.h, int(stage), int(reps), C.byref(ms)))
        return ms.value

    def final_set_training(self, e, xs, ys, w):
        xs, ys, w = (np.ascontiguousarray(a, dtype=np.float64) for a in (xs, ys, w))
        self.ctx.check(self.lib.gpet_final_set_training(self.h, e, xs.ctypes.data, ys.ctypes.data, w.ctypes.data,
                                                        xs.shape[0]))

    def read_obs_all(self):
        cap = self._max_info("obs_cap")
        dst = np.zeros((self.B, cap, 2), dtype=np.int64)
        cnt = np.zeros(self.B, dtype=np.int32)
        self.ctx.check(self.lib.gpet_batch_read_obs_all(self.h, dst.ctypes.data, cnt.ctypes.data, cap))
        return [dst[e, :cnt[e]].copy() for e in range(self.B)]

    def final_set_training_all(self, xs_list, ys_list, w_list):
        n = np.asarray([len(x) for x in xs_list], dtype=np.int32)
        stride = int(n.max())

        def pack(lst):
            out = np.zeros((len(lst), stride))
            for i, a in enumerate(lst):
                out[i, :n[i]] = a
            return out
        xs, ys, w = pack(xs_list), pack(ys_list), pack(w_list)
        self.ctx.check(self.lib.gpet_final_set_training_all(self.h, xs.ctypes.data, ys.ctypes.data, w.ctypes.data,
                                                            n.ctypes.data, stride))

    def final_predict_all(self, par):
        par = np.ascontiguousarray(par, dtype=np.float64).reshape(self.B, 12)
        Lg = self._max_info("Lg")
        mean = np.zeros((self.B, Lg))
        std = np.zeros((self.B, Lg))
        self.ctx.check(self.lib.gpet_final_predict_all(self.h, par.ctypes.data, mean.ctypes.data, std.ctypes.data, Lg))
        return mean, std

    def final_fit_all(self, seeds):
        """Converged fits of every edge on the device (gpet_final_fit_all).  Returns (mean [B, Lg_max] in pixels,
        std [B, Lg_max] in standardised units, theta [B, 3], minimum of -LML [B], objective launches)."""
        s = (C.c_uint32 * self.B)(*[int(v) & 0xFFFFFFFF for v in seeds])
        Lg = self._max_info("Lg")
        mean = np.zeros((self.B, Lg))
        std = np.zeros((self.B, Lg))
        th = np.zeros((self.B, 4))
        rounds = C.c_int32()
        self.ctx.check(self.lib.gpet_final_fit_all(self.h, s, mean.ctypes.data, std.ctypes.data, th.ctypes.data, Lg,
                                                   C.byref(rounds)))
        return mean, std, th[:, :3].copy(), th[:, 3].copy(), rounds.value

    def set_option(self, name, value):
        """This batch's own copy of a tuning switch (gpet_batch_set_option); returns the previous value (-1 = automatic)."""
        old = self.get_option(name)
        if self.lib.gpet_batch_set_option(self.h, name.encode(), int(value)) < 0:
            raise ValueError("unknown option %r" % (name,))
        return old

    def get_option(self, name):
        v = C.c_int()
        if self.lib.gpet_batch_get_option(self.h, name.encode(), C.byref(v)) != 0:
            raise ValueError("unknown option %r" % (name,))
        return v.value

    def set_rng(self, rng):
        """Random numbers of the batch: "mt19937" (default: numpy's RandomState stream, the reference's) or "philox"
        (gpet_batch_set_rng: counter-based Philox4x32-10 + Box-Muller, opt-in, not the reference's numbers)."""
        if rng not in (None, "mt19937", "philox"):
            raise ValueError("rng must be 'mt19937' or 'philox'")
        self.ctx.check(self.lib.gpet_batch_set_rng(self.h, 1 if rng == "philox" else 0))

    def set_sample_dtype(self, dtype):
        """Storage type of the posterior samples: "f64" (default, the reference's) or "f32" (gpet_batch_set_sample_dtype:
        the GEMM rounds on store, consumers widen; opt-in, BASELINE config 2's "fp32 posterior samples")."""
        if dtype not in (None, "f64", "f32", "float64", "float32"):
            raise ValueError("sample_dtype must be 'f64' or 'f32'")
        self.ctx.check(self.lib.gpet_batch_set_sample_dtype(self.h, 1 if dtype in ("f32", "float32") else 0))

    def final_optimize(self, starts, bounds):
        """Device L-BFGS-B on the training sets of final_set_training(_all) (gpet_final_optimize): starts
        (B, n_starts, 3) = log(constant, length_scale, noise_level), bounds (3, 2) in the same space.  Returns
        (theta [B, 3] of the best start per edge, its -LML [B], objective rounds)."""
        starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(self.B, -1, 3)
        bounds = np.ascontiguousarray(bounds, dtype=np.float64).reshape(3, 2)
        th = np.zeros((self.B, 4))
        rounds = C.c_int32()
        self.ctx.check(self.lib.gpet_final_optimize(self.h, starts.shape[1], starts.ctypes.data, bounds.ctypes.data,
                                                    th.ctypes.data, C.byref(rounds)))
        return th[:, :3].copy(), th[:, 3].copy(), rounds.value

    def lml_batch(self, edge_of, theta):
        edge_of = np.ascontiguousarray(edge_of, dtype=np.int32)
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(-1, 3)
        P = theta.shape[0]
        f = np.empty(P)
        g = np.empty((P, 3))
        self.ctx.check(self.lib.gpet_lml_batch(self.h, P, edge_of.ctypes.data, theta.ctypes.data, f.ctypes.data,
                                               g.ctypes.data))
        return f, g

    def lml_stats(self, reset=False):
        ms, ev, la = C.c_double(), C.c_int64(), C.c_int32()
        self.ctx.check(self.lib.gpet_lml_stats(self.h, int(bool(reset)), C.byref(ms), C.byref(ev), C.byref(la)))
        return dict(kernel_ms=ms.value, evaluations=ev.value, launches=la.value)

    def select_pixels(self):
        self.ctx.check(self.lib.gpet_select_pixels(self.h))

    def select_pixels_only(self):
        self.ctx.check(self.lib.gpet_select_pixels_only(self.h))

    def curve_kde(self):
        self.ctx.check(self.lib.gpet_curve_kde(self.h))

    def final_cov(self):
        self.ctx.check(self.lib.gpet_final_cov(self.h))

    def iterate(self, base_seeds, max_iters):
        s = (C.c_uint32 * self.B)(*[int(v) & 0xFFFFFFFF for v in base_seeds])
        n = C.c_int()
        self.ctx.check(self.lib.gpet_trace_iterate(self.h, s, int(max_iters), C.byref(n)))
        self._scored = self._scored or int(max_iters) > 0
        return n.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.gpet_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
