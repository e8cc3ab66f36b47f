"""The dealing of a riding fb_smooth (voldor_amd/csrc/vk_voldor.hip fb_ride_plan / fb_ride_of_camera, through vk_debug_fb_ride_plan): host arithmetic that
decides which 256-thread block of which pass runs in which camera's mode-kernel launch.  Held to its invariants over a sweep of window geometries, no
device involved: every row block in exactly one launch, every column block in exactly one LATER launch, at most 480 blocks (240 riding workgroups: one
per otherwise idle compute unit) per launch, and no riding where it cannot work (a single camera, more blocks than the launches hold); since round 6 every stack in the
segments its own launches use (12 | 20 | 40 steps per lane, rows and columns apart), the rigidness maps alone where the prior confidences do not fit.
The arithmetic of the blocks themselves is held on the GPU (tests/test_gpu_riders.py: no output bit of a window changes)."""
import ctypes as C

import numpy as np
import pytest


def _plan(w, h, n_flows, n_dp):
    from voldor_amd import capi
    out = (C.c_int * (5 + 3 * 16))()
    n = capi.lib().vk_debug_fb_ride_plan(w, h, n_flows, n_dp, out, len(out))
    assert n == 5 + 3 * n_flows
    a = np.array(out[:n])
    return dict(on=int(a[0]), seg_rows=int(a[1]) // 100, seg_cols=int(a[1]) % 100, R=int(a[2]), C=int(a[3]), k_rows=int(a[4]), cams=a[5:].reshape(n_flows, 3))


def _segs(w, h, n_maps):
    """steps per lane of the row / column pass over a stack of n_maps maps: the rule of fb_smooth_plan (vk_depth.hip), restated"""
    want = 40 if w * h * n_maps >= (8 << 20) else 12
    pick = lambda ln, mx: 12 if (want == 12 and ln <= 12 * mx) else 20 if (want <= 20 and ln <= 20 * mx) else 40
    return pick(w, 256), pick(h, 64)


def _expected_blocks(w, h, n_maps):
    rs, cs = _segs(w, h, n_maps)
    sr, sc = -(-w // rs), -(-h // cs)
    if sr > 256 or sc > 256:
        return None
    lpb, cw = 256 // sr, min(16, 256 // sc)
    return -(-h // lpb) * n_maps, -(-w // cw) * n_maps


SIZES = [(640, 480), (1241, 376), (320, 240), (333, 171), (64, 48), (1280, 720), (2048, 64), (17, 900), (1920, 1080), (3100, 200), (200, 3100)]


@pytest.mark.parametrize("w,h", SIZES)
def test_every_block_rides_exactly_once_and_rows_come_first(w, h):
    seen_on = False
    for n_flows in range(1, 17):
        for n_dp in (0, 1, 3):
            p = _plan(w, h, n_flows, n_dp)
            if not p["on"]:
                assert (p["cams"] == 0).all()
                continue
            seen_on = True
            assert n_flows >= 2 and p["on"] in (1, 2) and (p["seg_rows"], p["seg_cols"]) == _segs(w, h, n_flows)  # (the segments the pass's own launches use)
            assert 40 not in (p["seg_rows"], p["seg_cols"])
            r0, c0 = _expected_blocks(w, h, n_flows)
            pad = lambda n: (n + 1) & ~1
            if p["on"] == 2:  # the prior confidences ride too, in THEIR segments, behind the rigidness maps' slots padded to an even number
                assert n_dp > 0
                r1, c1 = _expected_blocks(w, h, n_dp)
            else:
                r1, c1 = 0, 0
            assert p["R"] == pad(r0) + r1 and p["C"] == pad(c0) + c1
            assert (p["cams"][:, 1] % 2 == 0).all()  # a riding workgroup takes two consecutive slots from an even one: both of ONE stack
            assert 1 <= p["k_rows"] <= n_flows - 1
            kinds, first, count = p["cams"][:, 0], p["cams"][:, 1], p["cams"][:, 2]
            assert (count <= 480).all() and (count >= 0).all()
            assert (kinds[:p["k_rows"]] != 2).all() and (kinds[p["k_rows"]:] != 1).all()  # rows only in the first k_rows launches, columns only after
            assert ((kinds == 0) == (count == 0)).all()
            for kind, total in ((1, p["R"]), (2, p["C"])):
                cover = np.zeros(total, int)
                for k, f, c in p["cams"]:
                    if k == kind:
                        assert 0 <= f and f + c <= total
                        cover[f:f + c] += 1
                assert (cover == 1).all(), (w, h, n_flows, n_dp, kind)
    if (w, h) in ((640, 480), (1241, 376), (320, 240), (333, 171)):
        assert seen_on


def test_where_nothing_rides():
    assert _plan(640, 480, 1, 0)["on"] == 0          # one camera: no launch to put the column blocks behind the row blocks
    assert _plan(1920, 1080, 10, 1)["on"] == 0       # 8 M map pixels and more: 40-step segments -- measured as riders in round 6 (mode kernels 12.7 -> 25.6 us, the 1080p window 3 % slower): they stay in the depth half
    assert _plan(400, 800, 3, 0)["on"] == 1          # round 6: rows in 12-step, columns in 20-step segments (a tall image) ride, each pass in its own segments
    assert _plan(1280, 720, 2, 0)["on"] == 0         # 720 row blocks for the one launch that may carry rows: more than its 480
    assert _plan(1280, 720, 5, 0)["on"] == 0         # (blocks and launches both grow with the frame count: 360 + 320 blocks per map never fit 480 per launch)
    assert _plan(1241, 376, 2, 0)["on"] == 1
