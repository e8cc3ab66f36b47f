"""Generates tests/golden/ref_window_noise.npz: every window of tests/ref_window_cases.py run by the REFERENCE's own pipeline
(oracle/_ref) with its own random seed (s0 = the run of tests/golden/ref_window.npz) and with N_SALTS other seeds of its random
streams (ref_set_rand_salt(1..N_SALTS): curand_init(seed ^ salt...) for the depth samples and the P3P draws alike; RAND_SEED 233 of
utils.h:18 is as good as any other).  The distances {re-seeded run vs s0} are, per window, a sample of the estimator's own
run-to-run spread -- two independent draws of the same estimator on the same flows.  tests/test_gpu_vs_ref_window.py ranks the
distance {fast HIP run vs s0} of the same window inside that sample (one-sided rank-sum test over all windows) instead of comparing
it with a hand-set tolerance.  Why seeds and not 1-ulp jitter here (the yardstick of the cfg2 / cfg3 ensembles): the reference draws
its hypotheses by INDEX into the compacted list of valid pixels, so two runs either share the valid set exactly -- identical draws,
near-identical output: what 1-ulp jitter gives on these small, prior-constrained windows (90th-percentile depth difference 0) -- or
differ in one pixel of it and re-draw all 8192 tuples.  Any implementation that is not bit-identical falls into the second case, and
the second case is what another seed produces.  Stored per run: registered count, poses, covariances, depth and confidence (every
2nd pixel per axis, every 4th of the larger windows).

Build container only: `python tests/golden/gen_golden_window_noise.py [workers]`; one process per run, a few seconds each."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)

import ref_window_cases as cases  # noqa: E402

N_SALTS = 8


def run_one(job):
    name, salt = job
    os.environ["OMP_NUM_THREADS"] = "1"
    from gen_golden_window import run_reference
    from oracle import orc
    c = dict(cases.window_cases())[name]
    ref = orc.ref()
    ref.ref_set_rand_salt(salt)
    r = run_reference(c)
    ref.ref_set_rand_salt(0)
    sub = 2 if c["exact"] else 4  # enough pixels for percentiles, a quarter / sixteenth of the bytes
    return job, {"n_registered": np.int32(r["n_registered"]), "poses": r["poses"], "poses_covar": r["poses_covar"],
                 "depth": r["depth"][::sub, ::sub].copy(), "depth_conf": r["depth_conf"][::sub, ::sub].copy()}


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, (os.cpu_count() or 2) - 1)
    names = [n for n, _ in cases.window_cases()]
    jobs = [(n, s) for n in names for s in range(N_SALTS + 1)]
    out = {}
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers, maxtasksperchild=1) as pool:
        for (name, salt), r in pool.imap_unordered(run_one, jobs):
            for k, v in r.items():
                out[f"{name}/s{salt}/{k}"] = v
            print(f"[{time.time() - t0:5.0f}s] {name:20s} salt {salt} n_registered {int(r['n_registered'])}", flush=True)
    # the glibc run must be the run of tests/golden/ref_window.npz
    g = np.load(os.path.join(HERE, "ref_window.npz"))
    for n in names:
        assert np.array_equal(out[f"{n}/s0/poses"], g[f"{n}/poses"]), n
    path = os.path.join(HERE, "ref_window_noise.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
