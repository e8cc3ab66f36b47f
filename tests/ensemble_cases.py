"""The window ensembles of tests/golden/gen_golden_ensemble.py (the reference pipeline, three libms per window) and
tests/test_gpu_ensemble.py (the fast HIP path on the same windows): BASELINE cfg2 and cfg3 scenes over
independent seeds -- scene geometry, camera motion, flow noise, outliers and disparity noise all re-drawn per seed."""
import numpy as np

import big_window_cases as big
import ref_window_cases as small
from voldor_amd import synth

CFG2_SEEDS = (233,) + tuple(range(400, 543))  # 144 windows (24 in round 3, 72 in round 4); 233 = the window bench.py times
CFG3_SEEDS = (233,) + tuple(range(500, 547))  # 48 windows (8 in round 3)


def make(kind, seed):
    if kind == "cfg2":
        sc = synth.make_scene(w=640, h=480, n_flows=5, fx=320.0, fy=320.0, cx=320.0, cy=240.0, seed=seed)
        c = small._case(sc, small.MONO, exact=False, ref_config=small.MONO + " --exclusive_gpu_context 0")
        c["depth_gt"] = sc["depth_gt"]
        return c
    b = big.CASES[kind]
    sc = synth.make_scene(w=b["w"], h=b["h"], n_flows=b["n"], fx=b["fx"], fy=b["fx"], cx=b["cx"], cy=b["cy"], seed=seed, basefocal=b["basefocal"])
    c = small._case(sc, b["config"], basefocal=b["basefocal"], disparity=True, exact=False)
    c["depth_gt"] = sc["depth_gt"]
    return c
