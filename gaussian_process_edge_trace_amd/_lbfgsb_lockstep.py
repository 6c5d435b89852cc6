"""Many independent L-BFGS-B minimisations advanced in lock step.

``scipy.optimize.minimize(method="L-BFGS-B")`` (what the reference's GaussianProcessRegressor
uses, sklearn_gpr.py:587-595) is a reverse-communication loop around ``_lbfgsb.setulb``
(scipy/optimize/_lbfgsb_py.py).  Driving that same routine for all (edge, restart) problems at
once lets every round of objective evaluations run as ONE batched GPU launch, while each
problem follows exactly the iterates scipy's own driver would produce for the same f/g values.
Falls back to plain ``scipy.optimize.minimize`` per problem if the private routine is missing.
"""
from __future__ import annotations

import os

import numpy as np

try:  # private but stable across the scipy versions this image ships
    from scipy.optimize import _lbfgsb
    _HAVE_SETULB = hasattr(_lbfgsb, "setulb")
except Exception:  # pragma: no cover
    _lbfgsb = None
    _HAVE_SETULB = False


def _bound_arrays(bounds):
    lo, hi = bounds[:, 0], bounds[:, 1]
    low = np.where(np.isinf(lo), 0.0, lo).astype(np.float64)
    up = np.where(np.isinf(hi), 0.0, hi).astype(np.float64)
    code = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}
    nbd = np.array([code[(bool(np.isfinite(a)), bool(np.isfinite(b)))] for a, b in zip(lo, hi)], dtype=np.int32)
    return low, up, nbd


def minimize_many(eval_batch, x0s, bounds, m=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxiter=15000,
                  maxfun=15000, maxls=20):
    """Minimise len(x0s) problems that share ``bounds`` (n, 2).

    ``eval_batch(idx, X)``: idx = list of problem indices, X = (len(idx), n) points;
    returns (f (len(idx),), g (len(idx), n)).  Returns (xs (P, n), funs (P,), n_rounds).
    Every problem sees exactly the call sequence ``scipy.optimize.minimize(method='L-BFGS-B',
    jac=True, bounds=...)`` would issue (same workspace sizes, factr, pgtol, maxls, stopping
    rules); state lives in row views of a few 2-D arrays so that a step costs one ``setulb`` call."""
    bounds = np.asarray(bounds, dtype=np.float64)
    P = len(x0s)
    n = bounds.shape[0]
    if not _HAVE_SETULB:  # pragma: no cover - same results, one problem at a time
        import scipy.optimize
        xs, fs = [], []
        for i, x0 in enumerate(x0s):
            def fun(x, i=i):
                f, g = eval_batch([i], x[None, :])
                return float(f[0]), g[0]
            r = scipy.optimize.minimize(fun, x0, method="L-BFGS-B", jac=True, bounds=bounds)
            xs.append(r.x)
            fs.append(r.fun)
        return np.array(xs), np.array(fs), -1
    factr = ftol / np.finfo(float).eps
    low, up, nbd = _bound_arrays(bounds)
    X = np.clip(np.asarray(x0s, dtype=np.float64).reshape(P, n), bounds[:, 0], bounds[:, 1])
    G = np.zeros((P, n))
    WA = np.zeros((P, 2 * m * n + 5 * n + 11 * m * m + 8 * m))
    IWA = np.zeros((P, 3 * n), dtype=np.int32)
    TASK = np.zeros((P, 2), dtype=np.int32)
    LNT = np.zeros((P, 2), dtype=np.int32)
    LSAVE = np.zeros((P, 4), dtype=np.int32)
    ISAVE = np.zeros((P, 44), dtype=np.int32)
    DSAVE = np.zeros((P, 29))
    rows = [(X[i], G[i], WA[i], IWA[i], TASK[i], LSAVE[i], ISAVE[i], DSAVE[i], LNT[i]) for i in range(P)]
    F = [0.0] * P
    n_iter = [0] * P
    nfev = [0] * P
    setulb = _lbfgsb.setulb
    active = list(range(P))
    rounds = 0
    while active:
        pending = []
        for i in active:
            x, g, wa, iwa, task, lsave, isave, dsave, lnt = rows[i]
            f = F[i]
            while True:
                setulb(m, x, low, up, nbd, f, g, factr, gtol, wa, iwa, task, lsave, isave, dsave, maxls, lnt)
                t0 = task[0]
                if t0 == 3:
                    pending.append(i)
                    break
                if t0 == 1:
                    n_iter[i] += 1
                    if n_iter[i] >= maxiter:
                        task[0], task[1] = 5, 504
                    elif nfev[i] > maxfun:
                        task[0], task[1] = 5, 502
                    continue
                break
        if not pending:
            break
        idx = np.asarray(pending, dtype=np.intp)
        f, g = eval_batch(pending, X[idx])
        G[idx] = g
        fl = np.asarray(f, dtype=np.float64).tolist()
        for k, i in enumerate(pending):
            F[i] = fl[k]
            nfev[i] += 1
        active = pending
        rounds += 1
    return X, np.asarray(F), rounds


# ---------------------------------------------------------------------------------------------
# The same lock-step scheme spread over worker processes.  A setulb step costs ~5-7 us of real
# L-BFGS-B arithmetic and holds the GIL, so thousands of problems are advanced by W processes
# that share X / F / G / state arrays with the parent; the parent only runs the batched GPU
# objective between two shared-memory hand-offs per round (counters polled, no semaphores).
# ---------------------------------------------------------------------------------------------
def _attach(name, shape, dtype):
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(name=name)
    return shm, np.ndarray(shape, dtype=dtype, buffer=shm.buf)


def _spin_until(arr, i, value):
    """Wait until arr[i] == value: the rounds are ~0.1-1 ms apart, far below what a semaphore hand-off
    costs, so poll shared memory; yield the core after a short burst (workers may outnumber cores)."""
    k = 0
    while arr[i] != value:
        k += 1
        if k > 200:
            os.sched_yield()


def _wait_until_all(arr, n, value):
    """Parent side: same polling, but sleeping between looks -- a pure-Python spin would hold the GIL and
    starve the thread that is driving the device loop of the next batch."""
    import time
    while not (arr[:n] == value).all():
        time.sleep(2e-5)


def _advance(job, X, Fs, G, S, wid):
    """One lock-step round of a job's problems owned by this worker: run setulb on every active
    problem until it asks for f/g (S=1) or stops (S=2)."""
    setulb = _lbfgsb.setulb
    m, low, up, nbd, factr, gtol, maxiter, maxfun, maxls = job["consts"]
    rows, n_iter, nfev, first = job["rows"], job["n_iter"], job["nfev"], job["first"]
    pending = []
    for i in job["active"]:
        x, g, wa, iwa, task, lsave, isave, dsave, lnt = rows[i]
        f = 0.0 if first else float(Fs[i])
        while True:
            setulb(m, x, low, up, nbd, f, g, factr, gtol, wa, iwa, task, lsave, isave, dsave, maxls, lnt)
            t0 = task[0]
            if t0 == 3:
                pending.append(i)
                break
            if t0 == 1:
                n_iter[i] += 1
                if n_iter[i] >= maxiter:
                    task[0], task[1] = 5, 504
                elif nfev[i] > maxfun:
                    task[0], task[1] = 5, 502
                continue
            S[i] = 2
            break
    for i in pending:
        S[i] = 1
        nfev[i] += 1
    job["first"] = False
    job["active"] = pending


def _farm_worker(wid, W, slots, pmax, n, m, names, conn):
    """Worker of a LockstepFarm: serves up to `slots` concurrent jobs (one per parent thread), taking the
    problems i = wid (mod W) of each; a job alone in the farm therefore gets all W workers."""
    try:
        from threadpoolctl import threadpool_limits
        _lim = threadpool_limits(limits=1)  # noqa: F841  (kept alive)
    except Exception:
        pass
    shms = []
    arrs = {}
    for key, (shape, dt) in dict(X=((slots, pmax, n), np.float64), F=((slots, pmax), np.float64),
                                 G=((slots, pmax, n), np.float64), S=((slots, pmax), np.int8),
                                 C=((slots, 4), np.int64), D=((slots, 256), np.int64)).items():
        shm, a = _attach(names[key], shape, dt)
        shms.append(shm)
        arrs[key] = a
    X, Fs, G, S, C, D = arrs["X"], arrs["F"], arrs["G"], arrs["S"], arrs["C"], arrs["D"]
    jobs = {}
    quit_ = False
    while not quit_:
        # new jobs: block when idle, poll when busy
        while (not jobs) or conn.poll(0):
            msg = conn.recv()
            if msg[0] == "quit":
                quit_ = True
                break
            _, slot, P, bounds, ftol, gtol, maxiter, maxfun, maxls = msg
            factr = ftol / np.finfo(float).eps
            low, up, nbd = _bound_arrays(bounds)
            mine = list(range(wid, P, W))  # interleaved: the 13 restarts of one edge spread over the workers
            cnt = len(mine)
            WA = np.zeros((cnt, 2 * m * n + 5 * n + 11 * m * m + 8 * m))
            IWA = np.zeros((cnt, 3 * n), dtype=np.int32)
            TASK = np.zeros((cnt, 2), dtype=np.int32)
            LNT = np.zeros((cnt, 2), dtype=np.int32)
            LSAVE = np.zeros((cnt, 4), dtype=np.int32)
            ISAVE = np.zeros((cnt, 44), dtype=np.int32)
            DSAVE = np.zeros((cnt, 29))
            Xs, Gs = X[slot], G[slot]
            job = dict(consts=(m, low, up, nbd, factr, gtol, maxiter, maxfun, maxls),
                       rows={i: (Xs[i], Gs[i], WA[k], IWA[k], TASK[k], LSAVE[k], ISAVE[k], DSAVE[k], LNT[k])
                             for k, i in enumerate(mine)},
                       n_iter=dict.fromkeys(mine, 0), nfev=dict.fromkeys(mine, 0), first=True, active=list(mine),
                       rnd=int(C[slot, 1]))  # the parent's release counter at the start of this job
            _advance(job, Xs, Fs[slot], Gs, S[slot], wid)
            job["rnd"] += 1
            D[slot, wid] = job["rnd"]  # parent may now collect the pending points
            jobs[slot] = job
        if quit_:
            break
        progressed = False
        for slot in list(jobs):
            job = jobs[slot]
            if C[slot, 1] != job["rnd"]:
                continue  # parent has not yet written F / G for this round
            progressed = True
            if C[slot, 0] == 0:
                del jobs[slot]
                continue
            _advance(job, X[slot], Fs[slot], G[slot], S[slot], wid)
            job["rnd"] += 1
            D[slot, wid] = job["rnd"]
        if not progressed:
            os.sched_yield()
    for shm in shms:
        shm.close()


class _FarmSlot:
    """One job lane of a LockstepFarm (use one per concurrently running parent thread)."""

    def __init__(self, farm, slot):
        self.farm, self.slot = farm, slot
        self.stats = {}

    def minimize(self, eval_batch, x0s, bounds, **kw):
        out = self.farm.minimize(eval_batch, x0s, bounds, slot=self.slot, **kw)
        self.stats = self.farm.stats_of[self.slot]
        return out


class LockstepFarm:
    """W worker processes advancing L-BFGS-B problems in lock step (see minimize_many), shared by up to
    `slots` concurrent jobs.  Create it BEFORE the process initialises HIP: the fork server is exec'ed here."""

    def __init__(self, workers, pmax=16384, n=3, m=10, slots=1):
        import multiprocessing as mp
        import threading
        from multiprocessing import shared_memory
        self.W, self.pmax, self.n, self.m, self.slots = int(workers), int(pmax), n, m, int(slots)
        if self.W > 256:
            raise ValueError("at most 256 workers")
        self._shms = {}
        self._arr = {}
        self._send_lock = threading.Lock()
        self.stats_of = [dict() for _ in range(self.slots)]
        self.stats = self.stats_of[0]
        for key, (shape, dt) in dict(X=((self.slots, pmax, n), np.float64), F=((self.slots, pmax), np.float64),
                                     G=((self.slots, pmax, n), np.float64), S=((self.slots, pmax), np.int8),
                                     C=((self.slots, 4), np.int64), D=((self.slots, 256), np.int64)).items():
            shm = shared_memory.SharedMemory(create=True, size=int(np.prod(shape)) * np.dtype(dt).itemsize)
            self._shms[key] = shm
            self._arr[key] = np.ndarray(shape, dtype=dt, buffer=shm.buf)
            self._arr[key][...] = 0
        saved = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}
        for k in saved:
            os.environ[k] = "1"
        # the workers are CPU-only (numpy + scipy's L-BFGS-B routine): start the fork server without a tool library a
        # profiler may have preloaded into this process (rocprofv3's installs signal handlers the fork server cannot
        # hand on to its children)
        tool_env = [k for k in os.environ if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES") or
                    k.startswith("ROCPROF")]
        for k in tool_env:
            saved[k] = os.environ.pop(k)
        try:
            ctx = mp.get_context("forkserver")
            ctx.set_forkserver_preload(["numpy", "scipy.optimize"])
            names = {k: s.name for k, s in self._shms.items()}
            self._conns, self._procs = [], []
            for w in range(self.W):
                a, b = ctx.Pipe()
                p = ctx.Process(target=_farm_worker, args=(w, self.W, self.slots, self.pmax, n, m, names, b),
                                daemon=True)
                p.start()
                self._conns.append(a)
                self._procs.append(p)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def slot(self, i):
        if not 0 <= i < self.slots:
            raise ValueError("no such slot")
        return _FarmSlot(self, i)

    def minimize(self, eval_batch, x0s, bounds, ftol=2.2204460492503131e-09, gtol=1e-5, maxiter=15000,
                 maxfun=15000, maxls=20, slot=0):
        import time
        bounds = np.asarray(bounds, dtype=np.float64)
        P = len(x0s)
        if P > self.pmax:
            raise ValueError("too many problems for this farm")
        X, F, G, S, C, D = (self._arr[k][slot] for k in "XFGSCD")
        X[:P] = np.clip(np.asarray(x0s, dtype=np.float64).reshape(P, self.n), bounds[:, 0], bounds[:, 1])
        S[:P] = 0
        C[0] = 1
        rnd = int(C[1])
        D[:self.W] = rnd
        with self._send_lock:
            for c in self._conns:
                c.send(("go", slot, P, bounds, ftol, gtol, maxiter, maxfun, maxls))
        rounds = 0
        st = self.stats_of[slot] = dict(wait_workers=0.0, eval=0.0, wait_release=0.0, host=0.0)
        if slot == 0:
            self.stats = st
        while True:
            t0 = time.perf_counter()
            rnd += 1
            _wait_until_all(D, self.W, rnd)
            t1 = time.perf_counter()
            idx = np.nonzero(S[:P] == 1)[0]
            if idx.size == 0:
                C[0] = 0
                C[1] = rnd
                break
            Xi = X[idx]
            t2 = time.perf_counter()
            f, g = eval_batch(idx, Xi)
            t3 = time.perf_counter()
            F[idx] = f
            G[idx] = g
            S[idx] = 0
            t4 = time.perf_counter()
            C[1] = rnd
            t5 = time.perf_counter()
            st["wait_workers"] += t1 - t0
            st["host"] += (t2 - t1) + (t4 - t3)
            st["eval"] += t3 - t2
            st["wait_release"] += t5 - t4
            rounds += 1
        return X[:P].copy(), F[:P].copy(), rounds

    def close(self):
        with self._send_lock:
            for c in self._conns:
                try:
                    c.send(("quit",))
                except Exception:
                    pass
        for p in self._procs:
            p.join(timeout=2)
            if p.is_alive():
                p.terminate()
        for s in self._shms.values():
            try:
                s.close()
                s.unlink()
            except Exception:
                pass
        self._procs = []

    def __del__(self):
        try:
            if self._procs:
                self.close()
        except Exception:
            pass
