"""GPU diagnostic: does gfx950 give the host's bits for the arithmetic strict mode rests on, and where does the device build of
the LambdaTwist solver start to differ from its host build?  (tests/cxx/vk_testhooks.hip)   python scripts/diag_device_math.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hooks  # noqa: E402
from oracle import orc  # noqa: E402
from voldor_amd import kernels, synth  # noqa: E402

out = {}
rng = np.random.default_rng(1)
n = 1 << 18
mag = np.exp(rng.uniform(-12, 12, n))
a = (rng.choice([-1, 1], n) * mag).astype(np.float32)
b = (rng.choice([-1, 1], n) * np.exp(rng.uniform(-6, 6, n))).astype(np.float32)
a2 = rng.normal(0, 3, n).astype(np.float32); b2 = rng.normal(0, 3, n).astype(np.float32)
names = ["f32 div", "f32 sqrt", "f64 div", "f64 sqrt", "1/sqrt f64->f32", "a*b+a nofma", "vsm_expf", "vsm_logf", "vsm_powf", "vsm_atan2f",
         "vsm_sinf", "vsm_cosf", "vsm_cbrtf", "floorf", "f32->int->f32", "1/a f32", "vsm_exp f64", "vsm_log f64", "f64 mul->f32", "f32*0.5(double)",
         "strict rigidness", "strict depth rigidness", "strict cost term", "f32 div expr", "l2norm"]
for op in range(hooks.lib().vkt_probe_ops_count()):
    for tag, (x, y) in (("wide", (a, b)), ("unit", (a2, b2))):
        if op == 14 and tag == "wide":
            x = np.clip(x, -1e9, 1e9)
        h = hooks.probe(op, x, y, False); d = hooks.probe(op, x, y, True)
        neq = (h.view(np.uint64) != d.view(np.uint64)) & ~(np.isnan(h) & np.isnan(d))
        out[f"probe/{names[op]}/{tag}"] = int(neq.sum())
        if neq.any():
            i = int(np.flatnonzero(neq)[0])
            print(f"MISMATCH {names[op]} [{tag}]: {int(neq.sum())}/{n}  e.g. a={x[i]!r} b={y[i]!r} host={h[i]!r} dev={d[i]!r}")
print("probes:", json.dumps({k: v for k, v in out.items() if v}, indent=0) or "all identical")

# ---- LambdaTwist on realistic correspondences
sc = synth.make_scene(w=320, h=240, n_flows=4, fx=160, fy=160, cx=160, cy=120, seed=7)
fx, fy, cx, cy = sc["K"]
K = np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], np.float32)
N, h_, w_ = sc["flows"].shape[:3]
gt_R = np.stack([synth.rodrigues(p[:3]) for p in sc["poses_gt"]]).astype(np.float32)
gt_t = np.stack([p[3:] for p in sc["poses_gt"]]).astype(np.float32)
depth = (sc["depth_gt"] * (1 + rng.normal(0, 0.02, sc["depth_gt"].shape))).astype(np.float32)
rig = np.ones((N, h_, w_), np.float32)
p2, p3 = orc.collect_p3p(sc["flows"], rig, depth, K, gt_R, gt_t, 1)
pts2, pts3 = orc.compact_p3p(p2, p3)
npose = 8192
idx = np.array([orc.pose_sample_indices(i, pts2.shape[0]) for i in range(npose)])
y8 = pts2[idx].reshape(npose, 8); x12 = pts3[idx].reshape(npose, 12)
for dbl in (0, 1):
    ok_h, R_h, t_h, dbg_h = hooks.p4p(y8, x12, (fx, fy, cx, cy), dbl, False)
    ok_d, R_d, t_d, dbg_d = hooks.p4p(y8, x12, (fx, fy, cx, cy), dbl, True)
    both = (ok_h == 1) & (ok_d == 1)
    same_t = (t_h.view(np.uint32) == t_d.view(np.uint32)).all(1)
    first = np.full(npose, -1)
    dd = (dbg_h.view(np.uint64) != dbg_d.view(np.uint64)) & ~(np.isnan(dbg_h) & np.isnan(dbg_d))
    for i in np.flatnonzero(dd.any(1)):
        first[i] = int(np.flatnonzero(dd[i])[0])
    hist = {int(k): int(v) for k, v in zip(*np.unique(first[first >= 0], return_counts=True))}
    out[f"p4p/{'f64' if dbl else 'f32'}"] = dict(ok_mismatch=int((ok_h != ok_d).sum()), t_identical=float(same_t[both].mean()), n_both=int(both.sum()),
                                                  first_diff_hist=hist)
    print(f"p4p {'f64' if dbl else 'f32'} sequential path, device vs host: ok mismatch {(ok_h != ok_d).sum()}, t bit-identical {same_t[both].mean():.4f} of {both.sum()}; "
          f"first differing tap: {hist}")
    if dd.any():
        i = int(np.flatnonzero(dd.any(1))[0]); k = first[i]
        print("   example", i, "tap", k, "host", dbg_h[i, max(0, k - 2):k + 3], "dev", dbg_d[i, max(0, k - 2):k + 3])
    if not dbl:
        # the product kernel (4 lanes per hypothesis + fold) against the host build
        grv, gtv = kernels.solve_batch_p3p_lambdatwist_gpu(pts3, pts2, K, npose)
        fin_g = np.isfinite(gtv.sum(1)); fin_h = ok_h == 1
        same = (gtv.view(np.uint32) == t_h.view(np.uint32)).all(1)
        out["k_solve/f32"] = dict(finite_mismatch=int((fin_g != fin_h).sum()), t_identical=float(same[fin_g & fin_h].mean()))
        print(f"k_solve<lambdatwist f32> vs host build: finite mismatch {(fin_g != fin_h).sum()}, t bit-identical {same[fin_g & fin_h].mean():.4f}")
        orv, otv = orc.solve_batch_p3p(pts3, pts2, K, npose)
        same_o = (otv.view(np.uint32) == t_h.view(np.uint32)).all(1)
        print(f"   oracle vs host build: t bit-identical {same_o[fin_h & np.isfinite(otv.sum(1))].mean():.4f}")
        rv_h = hooks.rodrigues(R_h[fin_h], True, False); rv_d = hooks.rodrigues(R_h[fin_h], True, True)
        out["rodrigues_strict"] = int((rv_h.view(np.uint32) != rv_d.view(np.uint32)).any(1).sum())
        print("strict rodrigues device vs host mismatches:", out["rodrigues_strict"], "of", int(fin_h.sum()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag_device_math.json"), "w"), indent=1)
