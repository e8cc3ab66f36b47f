"""Kernel-level picture of 4 windows in flight (vk_voldor_device_batch): run under
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/inflight -- python scripts/inflight_profile.py
and summarise with scripts/inflight_profile.py --summarise <kernel_trace.csv>."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def summarise(path):
    import csv
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(path))]
    rows.sort()
    # steady state: the last 60 % of the trace
    t_lo = rows[0][0] + 0.4 * (rows[-1][1] - rows[0][0])
    rows = [r for r in rows if r[0] >= t_lo]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    ev = sorted([(s, 1) for s, e, _ in rows] + [(e, -1) for s, e, _ in rows])
    cur, last, hist = 0, ev[0][0], {}
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
    print(f"steady-state span {span/1e6:.2f} ms, summed kernel time {busy/1e6:.2f} ms -> average {busy/span:.2f} kernels running")
    tot = sum(hist.values())
    print("time share by number of kernels running at once:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
    nwin = sum(1 for r in rows if r[2] == "vk::k_extract_corr")
    print(f"{nwin} windows started in the span -> {span/1e6/max(nwin,1):.2f} ms per window")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2]); sys.exit(0)
    import torch
    from voldor_amd import pyvoldor, synth
    W, H, N, B = 640, 480, 5, 4
    CONFIG = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 8"
    fls = [torch.from_numpy(synth.make_scene(w=W, h=H, n_flows=N, fx=320, fy=320, cx=320, cy=240, seed=1000 + b)["flows"]).cuda() for b in range(B)]
    for _ in range(12): pyvoldor.voldor_device_batch(fls, 320, 320, 320, 240, config=CONFIG)
    torch.cuda.synchronize()
