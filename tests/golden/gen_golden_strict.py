"""Generates tests/golden/ref_window_strict.npz and tests/golden/ref_selfnoise.npz with the REFERENCE's own pipeline
(oracle/_ref: voldor/*.cpp + gpu-kernels/*.cu executed on the CPU).  Build container only:
`python tests/golden/gen_golden_strict.py` after `make -C oracle ref`.

ref_window_strict.npz -- the reference pipeline with its libm calls served by voldor_amd/csrc/vk_strict_math.h
  (ref_set_math_mode(1), oracle/ref_stubs/emul/cuda_emul.h): what "strict math" means on the reference's own code.
  tests/test_oracle_vs_ref_window.py holds the oracle in strict mode to it bit for bit; tests/test_gpu_strict.py holds the HIP
  path to the oracle bit for bit.
ref_selfnoise.npz -- the same window run by the reference pipeline three times: glibc, strict math, and glibc with the last
  bit of every expf/powf/logf result jittered (mode 2).  The three runs differ ONLY in the rounding of the transcendentals;
  the distance between them is the self-noise of the estimator under 1-ulp perturbations, the yardstick for
  "fast HIP path vs strict HIP path" (tests/test_gpu_strict.py::test_fast_vs_strict_*)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)

import ref_window_cases as cases  # noqa: E402
from gen_golden_window import run_reference  # noqa: E402
from oracle import orc  # noqa: E402

STRICT_CASES = ("mono_nonexclusive", "stereo_default", "stereo_ap3p", "depth_priors")


def main():
    ref = orc.ref()
    allc = dict(cases.window_cases())
    out = {}
    if "--default-only" in sys.argv:
        default_mode_windows(ref, allc)
        return
    if "--cuda-only" not in sys.argv:
        strict_windows(ref, allc)
        default_mode_windows(ref, allc)
    cuda_windows_and_noise(ref, allc)


DEFAULT_MODE_CASES = ("mono_default_b1", "truncated_b1")


def default_mode_windows(ref, allc):
    """ref_window_default.npz -- round 4: monocular windows under the reference's DEFAULT `exclusive_gpu_context 1`, in strict math: the runs
    that show SURVEY Appendix B-1 (optimize_depth.cu searches from a device copy of the depth map that never saw normalize_world_scale()).
    The product reproduces them with `--reference_stale_depth 1` on top of the reference mode, the oracle with ORC_EMULATE_B1=1.  Two small
    windows (one of them truncated) and BASELINE cfg2 at full size (sha256 + every 8th pixel)."""
    import hashlib
    out = {}
    ref.ref_set_math_mode(1)
    orc.lib().orc_set_strict_math(1)
    try:
        for name in DEFAULT_MODE_CASES:
            r = run_reference(allc[name])
            out[f"{name}/n_registered"] = np.int32(r["n_registered"])
            for k in ("poses", "poses_covar", "depth", "depth_conf"):
                out[f"{name}/{k}"] = r[k]
            print(f"strict, default exclusive mode {name:20s} n_registered {r['n_registered']}")
        name, c = cases.cfg2_case()
        c = dict(c, ref_config=c["config"])  # the reference's default: exclusive_gpu_context 1
        r = run_reference(c)
        name = name + "_default"
        out[f"{name}/n_registered"], out[f"{name}/poses"], out[f"{name}/poses_covar"] = np.int32(r["n_registered"]), r["poses"], r["poses_covar"]
        for k in ("depth", "depth_conf"):
            out[f"{name}/{k}_sub8"] = r[k][::8, ::8].copy()
            out[f"{name}/{k}_sha256"] = np.frombuffer(hashlib.sha256(r[k].tobytes()).digest(), np.uint8)
        print(f"strict, default exclusive mode {name:20s} n_registered {r['n_registered']}")
    finally:
        ref.ref_set_math_mode(0); orc.lib().orc_set_strict_math(0)
    path = os.path.join(HERE, "ref_window_default.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def strict_windows(ref, allc):
    out = {}
    ref.ref_set_math_mode(1)
    orc.lib().orc_set_strict_math(1)  # the injected two-view pose comes from the oracle's bootstrap: no transcendental in it, but keep one mode
    try:
        for name in STRICT_CASES:
            r = run_reference(allc[name])
            out[f"{name}/n_registered"] = np.int32(r["n_registered"])
            for k in ("poses", "poses_covar", "depth", "depth_conf"):
                out[f"{name}/{k}"] = r[k]
            print(f"strict {name:20s} n_registered {r['n_registered']}")
    finally:
        ref.ref_set_math_mode(0); orc.lib().orc_set_strict_math(0)
    path = os.path.join(HERE, "ref_window_strict.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def cuda_windows_and_noise(ref, allc):
    # ---- round 4: the same windows with the reference's LAST two stand-ins switched off as well -- cuRAND XORWOW streams (D1) and CUDA's
    # linear texture filter over the stacked layers (D2), both restated in voldor_amd/csrc/vk_ref_cuda.h (ref_set_reference_rng / _tex)
    out = {}
    ref.ref_set_math_mode(1); ref.ref_set_reference_rng(1); ref.ref_set_reference_tex(1)
    orc.lib().orc_set_strict_math(1)
    try:
        for name in STRICT_CASES:
            r = run_reference(allc[name])
            out[f"{name}/n_registered"] = np.int32(r["n_registered"])
            for k in ("poses", "poses_covar", "depth", "depth_conf"):
                out[f"{name}/{k}"] = r[k]
            print(f"strict + xorwow + tex {name:20s} n_registered {r['n_registered']}")
        import hashlib
        name, c = cases.cfg2_case()
        r = run_reference(c)
        out[f"{name}/n_registered"], out[f"{name}/poses"], out[f"{name}/poses_covar"] = np.int32(r["n_registered"]), r["poses"], r["poses_covar"]
        for k in ("depth", "depth_conf"):
            out[f"{name}/{k}_sub8"] = r[k][::8, ::8].copy()
            out[f"{name}/{k}_sha256"] = np.frombuffer(hashlib.sha256(r[k].tobytes()).digest(), np.uint8)
        print(f"strict + xorwow + tex {name:20s} n_registered {r['n_registered']}")
    finally:
        ref.ref_set_math_mode(0); ref.ref_set_reference_rng(0); ref.ref_set_reference_tex(0); orc.lib().orc_set_strict_math(0)
    path = os.path.join(HERE, "ref_window_cuda.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
    if "--cuda-only" in sys.argv:
        return

    noise = {}
    big = dict(cases.window_cases())["mono_320x240"]
    for name, c in (("mono_320x240", big), cases.cfg2_case()):
        runs = {}
        for mode in (0, 1, 2):
            ref.ref_set_math_mode(mode)
            try:
                runs[mode] = run_reference(c)
            finally:
                ref.ref_set_math_mode(0)
            r = runs[mode]
            noise[f"{name}/m{mode}/n_registered"] = np.int32(r["n_registered"])
            noise[f"{name}/m{mode}/poses"], noise[f"{name}/m{mode}/poses_covar"] = r["poses"], r["poses_covar"]
            noise[f"{name}/m{mode}/depth_sub4"] = r["depth"][::4, ::4].copy()
            noise[f"{name}/m{mode}/depth_conf_sub4"] = r["depth_conf"][::4, ::4].copy()
        for a, b in ((0, 1), (0, 2), (1, 2)):
            ra, rb = runs[a], runs[b]
            m = (ra["depth_conf"] > 0.5) & (rb["depth_conf"] > 0.5)
            rel = np.abs(ra["depth"][m] - rb["depth"][m]) / ra["depth"][m]
            stats = np.array([np.mean(ra["depth"] == rb["depth"]), np.median(rel), np.percentile(rel, 90), np.mean(rel < 1e-3)], np.float64)
            noise[f"{name}/depth_stats_m{a}_m{b}"] = stats  # identical fraction, median / p90 relative difference, fraction within 1e-3
            print(f"{name} modes {a}/{b}: depth identical {stats[0]:.4f} median rel {stats[1]:.2e} p90 {stats[2]:.2e} within 1e-3 {stats[3]:.4f}; "
                  f"pose diff rot/trans {np.abs(ra['poses'][:, :3] - rb['poses'][:, :3]).max():.2e} {np.abs(ra['poses'][:, 3:] - rb['poses'][:, 3:]).max():.2e}")
    path = os.path.join(HERE, "ref_selfnoise.npz")
    np.savez_compressed(path, **noise)
    print(f"wrote {path}: {len(noise)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
