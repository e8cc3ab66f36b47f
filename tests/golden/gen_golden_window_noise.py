"""Generates tests/golden/ref_window_noise.npz: every window of tests/ref_window_cases.py run by the REFERENCE's own pipeline
(oracle/_ref) under glibc and under N_SALTS independent 1-ulp jitter patterns of expf / powf / logf (ref_set_math_mode(2),
ref_set_jitter_salt(1..N_SALTS); oracle/ref_stubs/emul/cuda_emul.h).  The distances {jitter run vs glibc run} are, per window,
a sample of the estimator's reproducibility under a last-bit change of its transcendentals; tests/test_gpu_vs_ref_window.py
ranks the distance {fast HIP run vs glibc run} of the same window inside that sample (rank test over all windows) instead of
comparing it with a hand-set tolerance.  Stored per run: registered count, poses, covariances, depth and confidence (every 2nd
pixel of the larger windows).

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
    ref.ref_set_math_mode(2 if salt else 0); ref.ref_set_jitter_salt(salt)
    r = run_reference(c)
    ref.ref_set_math_mode(0); ref.ref_set_jitter_salt(0)
    sub = 1 if c["exact"] else 2
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
