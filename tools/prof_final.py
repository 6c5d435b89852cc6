"""Ad-hoc timing of the final-fit path on the GPU box (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    from gaussian_process_edge_trace_amd._lbfgsb_lockstep import LockstepFarm
    farm = LockstepFarm(W) if W > 1 else None
    import gaussian_process_edge_trace_amd as pkg
    from gaussian_process_edge_trace_amd import gpet as G
    from bench import README_KW, synth_image
    ctx = pkg._lib.Context(0)
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    tr = pkg.GP_Edge_Tracing_Batch([init] * E, grad, list(range(1, E + 1)), **README_KW, _ctx=ctx, fit_farm=farm)
    iters = tr.run_loop()
    b = tr._batch
    orig = b.lml_batch
    dev = [0.0, 0, 0]
    def timed(edge_of, theta):
        s = time.time(); r = orig(edge_of, theta); dev[0] += time.time() - s; dev[1] += len(edge_of); dev[2] += 1
        return r
    b.lml_batch = timed
    for rep in range(3):
        dev[:] = [0.0, 0, 0]
        t0 = time.time(); obs = b.read_obs_all(); t1 = time.time()
        fits, rounds = G.device_final_fits(b, tr._ps, obs, iters, farm)
        t2 = time.time()
        print("W=%d E=%d read_obs %.1f ms | final fits %.1f ms (rounds %d; device objective %.1f ms, %d evals in %d launches)"
              % (W, E, 1e3 * (t1 - t0), 1e3 * (t2 - t1), rounds, 1e3 * dev[0], dev[1], dev[2]), flush=True)
        if farm:
            print("   farm ms:", {k: round(1e3 * v, 1) for k, v in farm.stats.items()}, flush=True)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    G.device_final_fits(b, tr._ps, obs, iters, farm)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    if farm:
        farm.close()


if __name__ == "__main__":
    main()
