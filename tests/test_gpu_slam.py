"""SURVEY.md 8(f)-4 on the GPU: eval_covisibility through the C-ABI against the reference's own scores
(tests/golden/ref_covis.npz) and the oracle's counts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    z = np.load(os.path.join(GOLD, "ref_covis.npz"))
    for i in range(int(z["n"])):
        mask = z[f"mask_{i}"]
        yield dict(stride=int(z[f"stride_{i}"]), depth=z[f"depth_{i}"], T=z[f"T_{i}"], K=z[f"K_{i}"],
                   mask=None if mask.size == 0 else mask.astype(bool), score=float(z[f"score_{i}"]))


def test_covisibility_matches_reference_scores():
    from oracle import orc_slam
    from voldor_amd import slam_utils
    worst = 0.0
    for c in _cases():
        s, nv, nc = slam_utils.eval_covisibility(c["depth"], c["T"], c["K"], c["mask"], c["stride"], return_counts=True)
        so, nvo, nco = orc_slam.eval_covisibility(c["depth"], c["T"], c["K"], c["mask"], c["stride"], return_counts=True)
        # integer counts; a projected point within float rounding of an image / cell border may fall on the other side
        # (numpy runs the three matmuls through BLAS, the kernel in a fixed order)
        assert abs(nv - nvo) <= 2 and abs(nc - nco) <= 2, (nv, nvo, nc, nco)
        assert abs(s - c["score"]) < 2e-3
        worst = max(worst, abs(s - c["score"]))
    assert worst < 2e-3


def test_covisibility_on_device_resident_maps():
    import torch
    from voldor_amd import slam_utils
    c = next(iter(_cases()))
    conf = np.random.default_rng(0).uniform(0, 1, c["depth"].shape).astype(np.float32)
    host = slam_utils.eval_covisibility(c["depth"], c["T"], c["K"], conf > 0.4, c["stride"], return_counts=True)
    d = torch.from_numpy(c["depth"]).cuda()
    m = torch.from_numpy(conf).cuda() > 0.4
    dev = slam_utils.eval_covisibility(d, c["T"], c["K"], m, c["stride"], return_counts=True)
    assert host == dev
    # identity motion, no mask: everything but the border samples stays visible
    s = slam_utils.eval_covisibility(d, np.eye(4, dtype=np.float32), c["K"], None, c["stride"])
    assert s > 0.97
