"""Whole-job throughput when the device loops of SEVERAL batch objects run concurrently (one host thread per object in
flight, each doing reset -> loop -> converged fits), against the bench's pipeline (one loop at a time, the fits of the
previous steps beside it).  python tools/time_concurrent_loops.py [edges] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import gaussian_process_edge_trace_amd as amd
import bench
from bench import synth_image, README_KW, timed_steps
L = amd._lib
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx0 = L.Context(0)
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx0)
seeds = list(range(1, E + 1))
NOBJ = int(sys.argv[3]) if len(sys.argv) > 3 else 4
objs = [amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=(ctx0 if k == 0 else L.Context(0))) for k in range(NOBJ)]
print("%d batch objects of %d edges, arena %d MiB each" % (NOBJ, E, objs[0]._batch.info()["arena_mib"]), flush=True)

def sync_all():
    for o in objs:
        o._ctx.sync()

def whole(o):
    o.reset()
    it = o.run_loop()
    return o.finish(it)

# bench pipeline: 3 objects, depth 2
ex = ThreadPoolExecutor(max_workers=2)
timed_steps(objs[:3], 3, 2, ex, [])
sync_all()
for rep in range(2):
    t0 = time.time()
    timed_steps(objs[:3], STEPS, 2, ex, [])
    sync_all()
    dt = time.time() - t0
    print("bench pipeline (one loop at a time, 2 steps' fits beside it): %.0f traces/s (%.1f ms per step)" % (STEPS * E / dt, 1e3 * dt / STEPS), flush=True)
for workers in [w for w in (2, 3, 4, 6, 8) if w <= NOBJ]:
    exw = ThreadPoolExecutor(max_workers=workers)
    list(exw.map(whole, objs[:workers]))
    sync_all()
    for rep in range(2):
        t0 = time.time()
        futs = []
        for k in range(STEPS):
            futs.append(exw.submit(whole, objs[k % workers]) if False else None)
        # (an object must not be used by two tasks at once: each worker walks its own object)
        def worker(o, n):
            for _ in range(n):
                whole(o)
        per = max(1, STEPS // workers)
        list(exw.map(lambda o: worker(o, per), objs[:workers]))
        sync_all()
        dt = time.time() - t0
        print("%d whole traces in flight (loop + fits per host thread): %.0f traces/s (%.1f ms per step)"
              % (workers, per * workers * E / dt, 1e3 * dt / (per * workers)), flush=True)
    exw.shutdown()
