"""Per-call durations of one kernel family inside a window, from a rocprofv3 --kernel-trace csv: the calls of a window are numbered
in launch order and averaged over the windows of the trace.  usage: ktrace_seq.py <kernel_trace.csv> <name substring> <calls per window>"""
import csv
import sys

import numpy as np

path, key, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if key in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows], float) / 1e3
n = len(d) // per * per
d = d[len(d) - n:].reshape(-1, per)  # drop incomplete leading windows
m = d.mean(axis=0)
print(f"{key}: {d.shape[0]} windows x {per} calls, mean {m.mean():.2f} us")
for i in range(0, per, 4):
    print("  calls %2d..%2d: " % (i, i + 3) + " ".join(f"{v:7.2f}" for v in m[i:i + 4]))
