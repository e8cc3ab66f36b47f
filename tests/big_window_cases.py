"""BASELINE cfg3 / cfg5 windows (the scenes bench.py --workload cfg3/cfg5 times on rank 0), shared by
tests/golden/gen_golden_big.py and tests/test_gpu_configs.py."""
import numpy as np

from voldor_amd import synth

CASES = {
    "cfg3": dict(w=1241, h=376, n=8, fx=718.856, cx=607.19, cy=185.22, basefocal=386.1,
                 config="--silent --meanshift_kernel_var 0.1 --disp_delta 1 --delta 0.2 --max_iters 8"),
    "cfg5": dict(w=1920, h=1080, n=10, fx=960.0, cx=960.0, cy=540.0, basefocal=480.0,
                 config="--silent --max_iters 12 --fb_smooth 1 --disp_delta 1 --delta 0.2"),
}


def make(name):
    c = CASES[name]
    sc = synth.make_scene(w=c["w"], h=c["h"], n_flows=c["n"], fx=c["fx"], fy=c["fx"], cx=c["cx"], cy=c["cy"], seed=233, basefocal=c["basefocal"])
    return dict(flows=np.ascontiguousarray(sc["flows"], np.float32), K=tuple(float(v) for v in sc["K"]), basefocal=float(c["basefocal"]),
                disparity=np.ascontiguousarray(sc["disparity"], np.float32), config=c["config"], poses_gt=sc["poses_gt"], depth_gt=sc["depth_gt"])
