"""Generates tests/golden/ref_ensemble.npz: the REFERENCE's own pipeline (oracle/_ref: voldor/*.cpp + gpu-kernels/*.cu executed on
the CPU) over an ensemble of independent windows, each run under three libms that differ only in the last bit of expf/powf/logf:

  g   glibc, as the reference calls it                         (ref_set_math_mode(0))
  jA  glibc with every result moved by -1/0/+1 ulp, salt 1      (ref_set_math_mode(2), ref_set_jitter_salt(1))
  jB  the same with an independent pattern, salt 2

  * BASELINE cfg2 (640x480, N=5, monocular, 8 iterations): seeds CFG2_SEEDS (144 windows)
  * BASELINE cfg3 (1241x376, N=8, stereo prior):            seeds CFG3_SEEDS (48 windows)

What the distances between those runs are is the estimator's reproducibility under a 1-ulp change of its transcendentals -- the
yardstick the fast HIP path (hardware v_exp / v_log) is held to in tests/test_gpu_ensemble.py: a two-sample Kolmogorov-Smirnov
test between {fast HIP vs g} and {jA vs g, jB vs g}, and between the errors against analytic ground truth of the two estimators.
Per run: registered count, poses, covariances, the depth / confidence maps subsampled by 8 (enough for medians over ~4800 / 7300
pixels), and the sha256 of the full maps.  Also: the reference's STRICT-math cfg2 window of seed 233 (poses, covariances, sha256
of the maps) -- the window tests/test_gpu_vs_ref_window.py::test_strict_cfg2_window_equals_the_reference holds the HIP path to.

Build container only (needs /root/reference through oracle/_ref): `python tests/golden/gen_golden_ensemble.py [workers]`.
One process per run (the reference keeps file-static device buffers); ~20 s (cfg2) / ~60 s (cfg3) per run on one core."""
import hashlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)

import ensemble_cases as ens  # noqa: E402

MODES = {"g": (0, 0), "jA": (2, 1), "jB": (2, 2), "strict": (1, 0)}
SUB = 8


FIVE_POINT = "--five-point" in sys.argv  # cfg2 only, into ref_ensemble5.npz: the reference pipeline started from the product's five-point two-view pose (round 6)


def run_one(job):
    kind, seed, mode = job[:3]
    five = len(job) > 3 and job[3]
    os.environ["OMP_NUM_THREADS"] = "1"
    from gen_golden_window import run_reference
    from oracle import orc
    c = ens.make(kind, seed)
    ref = orc.ref()
    m, salt = MODES[mode]
    ref.ref_set_math_mode(m); ref.ref_set_jitter_salt(salt)
    if m == 1:
        orc.lib().orc_set_strict_math(1)  # the injected two-view pose: one mode throughout
    t0 = time.time()
    r = run_reference(c, five_point=five)
    dt = time.time() - t0
    ref.ref_set_math_mode(0); ref.ref_set_jitter_salt(0)
    out = {"n_registered": np.int32(r["n_registered"]), "poses": r["poses"], "poses_covar": r["poses_covar"],
           "depth_sub": r["depth"][::SUB, ::SUB].copy(), "conf_sub": r["depth_conf"][::SUB, ::SUB].copy(),
           "depth_sha256": np.frombuffer(hashlib.sha256(r["depth"].tobytes()).digest(), np.uint8),
           "conf_sha256": np.frombuffer(hashlib.sha256(r["depth_conf"].tobytes()).digest(), np.uint8)}
    return job[:3], out, dt


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else max(1, (os.cpu_count() or 2) - 1)
    jobs = [("cfg2", 233, "strict")]
    jobs += [("cfg3", s, m) for s in ens.CFG3_SEEDS for m in ("g", "jA", "jB")]  # the long ones first
    jobs += [("cfg2", s, m) for s in ens.CFG2_SEEDS for m in ("g", "jA", "jB")]
    if FIVE_POINT:
        jobs = [("cfg2", s, m, True) for s in ens.CFG2_SEEDS for m in ("g", "jA", "jB")]
    out = {}
    path = os.path.join(HERE, "ref_ensemble5.npz" if FIVE_POINT else "ref_ensemble.npz")
    if os.path.exists(path) and "--fresh" not in sys.argv:  # incremental: runs already in the file are kept (each is a pure function of (kind, seed, mode))
        with np.load(path) as old:
            out = {k: old[k] for k in old.files}
        jobs = [j for j in jobs if f"{j[0]}/s{j[1]}/{j[2]}/n_registered" not in out]
        print(f"{len(out)} arrays kept, {len(jobs)} runs to do", flush=True)
    t0 = time.time(); done = 0
    with mp.get_context("spawn").Pool(workers, maxtasksperchild=1) as pool:
        for (kind, seed, mode), r, dt in pool.imap_unordered(run_one, jobs):
            for k, v in r.items():
                out[f"{kind}/s{seed}/{mode}/{k}"] = v
            print(f"[{time.time() - t0:6.0f}s] {kind} seed {seed} {mode:6s} n_registered {int(r['n_registered'])}  ({dt:.0f} s)", flush=True)
            done += 1
            if done % 30 == 0:  # (a run is ~20-60 s of one core: keep what is there if the generator is interrupted)
                np.savez_compressed(path, **out)
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
