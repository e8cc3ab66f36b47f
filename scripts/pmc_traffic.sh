#!/bin/bash
# HBM traffic per kernel launch from the PMC counters (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only), per-launch averages, gfx950 correction
# (FETCH_SIZE reports half of the bytes of coalesced reads).  Run on the GPU box:  WL=cfg2 bash scripts/pmc_traffic.sh <tag>
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=${1:-run}; WL=${WL:-cfg2}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${c}_$tag -- python bench.py --workload $WL --steps 2 --warmup 1 --no-extras > gpurun_out/pmc_${c}_$tag.log 2>&1
done
python - "$tag" "$WL" <<'PY'
import csv, glob, json, collections, os, sys
tag, wl = sys.argv[1], sys.argv[2]
def load(c):
    f = glob.glob(f"gpurun_out/pmc_{c}_{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(float); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if r["Counter_Name"] != c: continue
        acc[k] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); n[k] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}
F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --workload WL --steps 2 --warmup 1 "
                "--no-cpu-baseline --in-flight 0; per-launch averages in KiB as reported.  gfx950: FETCH_SIZE reports 1/2 of the bytes of coalesced "
                "reads (MI355X_MICROARCH.md), WRITE_SIZE is exact.  hbm_bytes = (fetch_factor*FETCH_SIZE + WRITE_SIZE)*1024; fetch_factor 2 for kernels of 8/16-byte "
                "loads, 1 for kernels of 4-byte loads (calibrated on kernels of known byte count, see scripts/pmc_traffic.sh).",
       "workload": wl, "commit": os.environ.get("COMMIT"), "kernels": {}}
import hashlib
h = hashlib.sha256()
for f in ("vk_depth.hip", "vk_depth_impl.hpp", "vk_fb.hpp", "vk_cum_poses.hpp", "vk_pose.hip", "vk_device.hpp", "vk_p3p.hpp", "vk_common.hpp"):  # = bench.KERNEL_SOURCES: bench.py withholds a pass taken on other sources
    h.update(open(os.path.join("voldor_amd", "csrc", f), "rb").read())
out["kernel_source_sha256"] = h.hexdigest()[:16]
# Calibration of the x2 on kernels whose byte count is known (profiles/r02h_pmc_traffic_cfg2.json, r02j_pmc_traffic_cfg5.json): FETCH_SIZE reports
# HALF the bytes of 8- and 16-byte loads (k_fb_rows: 20 B/px read as float4 -> 10.5 reported; k_depth_conf 44 -> 22; k_disp_to_depth 4 -> 2;
# k_update_rigidness_lean, 16-byte texel-pair gathers: ~44 -> 19.9) and ALL the bytes of 4-byte loads (k_fb_cols: 20 -> 20.5).  The flow
# gathers of the depth kernels are 16-byte texel pairs, so the x2 holds for them; kernels made of dword loads get factor 1.
FETCH_FACTOR = {"vk::k_fb_cols": 1.0}
for k in sorted(F, key=lambda k: -F[k][0] * F[k][1]):
    w = W.get(k, (0.0, 0))[0]
    fac = next((v for n, v in FETCH_FACTOR.items() if k.startswith(n)), 2.0)
    out["kernels"][k] = {"launches": F[k][1], "FETCH_SIZE_KiB": round(F[k][0], 1), "WRITE_SIZE_KiB": round(w, 1), "fetch_factor": fac,
                         "hbm_bytes_per_launch": int((fac * F[k][0] + w) * 1024)}
json.dump(out, open(f"gpurun_out/pmc_traffic_{tag}.json", "w"), indent=1)
for k, v in list(out["kernels"].items())[:14]: print(k[:44].ljust(44), v)
PY
