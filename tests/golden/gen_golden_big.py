"""Generates tests/golden/ref_big.npz: BASELINE cfg3 (1241x376, N=8, stereo) and cfg5 (1920x1080, N=10, 12 iterations,
RGB-D style disparity prior) windows -- the sizes bench.py --workload cfg3/cfg5 measures.  Build container only; takes
~20-30 minutes (the reference pipeline runs on one core):  python tests/golden/gen_golden_big.py [cfg3] [cfg5]

Two records per window:
  ref/...     the REFERENCE's own py_voldor_wrapper executed on the CPU (oracle/_ref, as tests/golden/gen_golden_window.py): poses,
              covariances, depth / confidence sub-sampled 4x4.  The HIP path is compared with it statistically (the reference's
              approximate-SVD rodrigues and its draw differ from the product's).
  strict/...  the oracle in strict-math mode (orc_set_strict_math(1), rejection draw D3b = ORC_REFERENCE_DRAW=0): sha256 of the full depth and confidence maps,
              poses, covariances.  The HIP path in --strict_math 1 must reproduce these BIT FOR BIT at full size
              (tests/test_gpu_configs.py).

python tests/golden/gen_golden_big.py --noise cfg3   writes tests/golden/ref_big_noise_cfg3.npz: the reference pipeline run twice more on the
same window, with strict-math transcendentals (ref_set_math_mode(1)) and with the last bit of every expf/powf/logf result jittered
(mode 2), poses and covariances only: the reference's own sensitivity to the last bit of its libm at this size, which is what the
fast-mode comparison is held to (as tests/golden/gen_golden_strict.py does for cfg2)."""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import orc  # noqa: E402
import big_window_cases as big  # noqa: E402


def noise(which):
    ref = orc.ref()
    for name in which:
        c = big.make(name)
        fx, fy, cx, cy = c["K"]
        out = {}
        for mode in (1, 2):
            t0 = time.time()
            ref.ref_set_math_mode(mode)
            try:
                r = orc.ref_voldor(c["flows"], fx, fy, cx, cy, config=c["config"], basefocal=c["basefocal"], disparity=c["disparity"])
            finally:
                ref.ref_set_math_mode(0)
            print(f"{name}: reference pipeline, math mode {mode}: {time.time() - t0:.0f} s, n_registered {r['n_registered']}", flush=True)
            out[f"{name}/m{mode}/n_registered"] = np.int32(r["n_registered"])
            out[f"{name}/m{mode}/poses"], out[f"{name}/m{mode}/poses_covar"] = r["poses"], r["poses_covar"]
        path = os.path.join(HERE, f"ref_big_noise_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}", flush=True)


def strict_ref(which):
    """python tests/golden/gen_golden_big.py --strict-ref cfg3 -> tests/golden/ref_big_strict_cfg3.npz: the REFERENCE pipeline itself in
    strict math (ref_set_math_mode(1)) at full size -- what the HIP window in `--strict_math 1 --reference_draw 1 --reference_svd 1`
    must equal bit for bit (tests/test_gpu_configs.py): registered count, poses, covariances, sha256 of the depth / confidence maps
    and every 8th pixel of them.  cfg3 ~4 min, cfg5 ~40 min on one core."""
    ref = orc.ref()
    cuda = "--cuda" in sys.argv  # with cuRAND's XORWOW streams and CUDA's linear texture filter as well (vk_ref_cuda.h; product: --reference_rng 1 --reference_tex 1) -> ref_big_cuda_<name>.npz
    for name in which:
        c = big.make(name)
        fx, fy, cx, cy = c["K"]
        t0 = time.time()
        ref.ref_set_math_mode(1)
        if cuda: ref.ref_set_reference_rng(1); ref.ref_set_reference_tex(1)
        try:
            r = orc.ref_voldor(c["flows"], fx, fy, cx, cy, config=c["config"], basefocal=c["basefocal"], disparity=c["disparity"])
        finally:
            ref.ref_set_math_mode(0)
            if cuda: ref.ref_set_reference_rng(0); ref.ref_set_reference_tex(0)
        print(f"{name}: reference pipeline in strict math{' + xorwow + texture filter' if cuda else ''}: {time.time() - t0:.0f} s, n_registered {r['n_registered']}", flush=True)
        out = {f"{name}/ref_strict/n_registered": np.int32(r["n_registered"]), f"{name}/ref_strict/poses": r["poses"],
               f"{name}/ref_strict/poses_covar": r["poses_covar"]}
        for k in ("depth", "depth_conf"):
            out[f"{name}/ref_strict/{k}_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(r[k]).tobytes()).digest(), np.uint8)
            out[f"{name}/ref_strict/{k}_sub8"] = r[k][::8, ::8].copy()
        path = os.path.join(HERE, f"ref_big_{'cuda' if cuda else 'strict'}_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}", flush=True)


def main():
    which = [a for a in sys.argv[1:] if a in big.CASES] or list(big.CASES)
    if "--noise" in sys.argv:
        return noise(which)
    if "--strict-ref" in sys.argv:
        return strict_ref(which)
    path = os.path.join(HERE, "ref_big.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in which:
        c = big.make(name)
        fx, fy, cx, cy = c["K"]
        kw = dict(basefocal=c["basefocal"], disparity=c["disparity"])
        t0 = time.time()
        orc.lib().orc_set_strict_math(1)
        os.environ["ORC_REFERENCE_DRAW"] = "0"  # these records keep the rejection draw D3b covered at full size (product: --reference_draw 0); the reference's draw is held by ref_big_strict_*.npz
        try:
            s = orc.voldor(c["flows"], fx, fy, cx, cy, config=c["config"], **kw)
        finally:
            orc.lib().orc_set_strict_math(0)
            os.environ.pop("ORC_REFERENCE_DRAW", None)
        print(f"{name}: strict oracle {time.time() - t0:.0f} s, n_registered {s['n_registered']}", flush=True)
        out[f"{name}/strict/n_registered"] = np.int32(s["n_registered"])
        out[f"{name}/strict/poses"], out[f"{name}/strict/poses_covar"] = s["poses"], s["poses_covar"]
        for k in ("depth", "depth_conf"):
            out[f"{name}/strict/{k}_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(s[k]).tobytes()).digest(), np.uint8)
            out[f"{name}/strict/{k}_sub8"] = s[k][::8, ::8].copy()
        if "--oracle-only" in sys.argv:  # only the strict/... records (e.g. after a change of the oracle that leaves the reference's run as it is)
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {len(out)} arrays (strict oracle records of {name} renewed)", flush=True)
            continue
        t0 = time.time()
        r = orc.ref_voldor(c["flows"], fx, fy, cx, cy, config=c["config"], **kw)
        print(f"{name}: reference pipeline {time.time() - t0:.0f} s, n_registered {r['n_registered']}", flush=True)
        out[f"{name}/ref/n_registered"] = np.int32(r["n_registered"])
        out[f"{name}/ref/poses"], out[f"{name}/ref/poses_covar"] = r["poses"], r["poses_covar"]
        for k in ("depth", "depth_conf"):
            out[f"{name}/ref/{k}_sub4"] = r[k][::4, ::4].copy()
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
