"""Work that rides in another kernel's launch (round 5): the 256-thread blocks of fb_smooth in the mode kernels of the pose half (FbRide,
voldor_amd/csrc/vk_common.hpp; vk_debug_switch "fb_ride") and the density reduction of an E-step in the next correspondence trace
(OdParams::defer_reduce; "defer_reduce").  Both move work between launches, not arithmetic: every output bit of a window must be that of the plain
launch chain -- windows with 1, 2, 5 and 8 cameras (how the row and column blocks fall onto the mode kernels), with depth priors (a second stack of
maps), ragged sizes, a refit in every iteration (nothing rides in a refit kernel), a window that truncates half way, device-resident inputs, and two
windows in a row on one context (the maps change names after every pose half)."""
import numpy as np
import pytest

import hooks

pytestmark = pytest.mark.gpu

MONO = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 4"
STEREO = "--silent --meanshift_kernel_var 0.1 --disp_delta 1 --delta 0.2 --max_iters 4"
CASES = {
    "five_cameras": dict(w=320, h=240, n=5, cfg=MONO),
    "two_cameras": dict(w=320, h=240, n=2, cfg=MONO),
    "one_camera": dict(w=200, h=160, n=1, cfg=MONO),
    "eight_cameras_with_prior": dict(w=411, h=203, n=8, cfg=STEREO, basefocal=40.0),
    "ragged_size": dict(w=333, h=171, n=4, cfg=MONO),
    "refit_in_every_iteration": dict(w=320, h=240, n=4, cfg=MONO + " --rg_refine_last_only 0"),
    "no_fb_smooth": dict(w=320, h=240, n=4, cfg=MONO + " --fb_smooth 0"),
    "truncates": dict(w=160, h=120, n=5, cfg=MONO + " --max_iters 6", noise_from=3),
    # round 6: every stack rides in the segments its own launches use -- a tall window: rows in 12-step, columns in 20-step segments (rounds 1-5: no riding there)
    "tall_rows_12_columns_20": dict(w=400, h=800, n=3, cfg=MONO.replace("--max_iters 4", "--max_iters 3")),
    "tall_with_prior": dict(w=300, h=900, n=4, cfg=STEREO.replace("--max_iters 4", "--max_iters 3"), basefocal=40.0),
    # 40-step segments (what 1080p windows get; forced here on a small one) do not ride -- measured in round 6, vk_voldor.hip fb_ride_plan
    "forty_step_segments": dict(w=320, h=240, n=5, cfg=MONO, switches={"fb_segment": 40}),
}


def _run(case, switches, device):
    import torch
    from voldor_amd import kernels, pyvoldor, synth
    c = CASES[case]
    bf = c.get("basefocal", 0.0)
    sc = synth.make_scene(w=c["w"], h=c["h"], n_flows=c["n"], fx=c["w"] / 2, fy=c["w"] / 2, cx=c["w"] / 2, cy=c["h"] / 2, seed=241, basefocal=bf)
    flows = sc["flows"].copy()
    if "noise_from" in c:
        flows[c["noise_from"]:] = np.random.default_rng(0).uniform(-40, 40, flows[c["noise_from"]:].shape).astype(np.float32)
    fx, fy, cx, cy = sc["K"]
    switches = dict(c.get("switches", {}), **switches)
    prev = {k: hooks.debug_switch(k, v) for k, v in switches.items()}
    plan = _plan_rides(c)  # (under the case's own switches: a forced segment length changes the dealing)
    hooks.debug_counter("fb_blocks_rode"); hooks.debug_counter("reduces_rode"); hooks.debug_counter("fb_side_passes")  # (read and clear)
    try:
        outs = []
        for _ in range(2):  # two windows in a row on one context
            kernels.set_rand_epoch(0)
            if device:
                depth = torch.empty(c["h"], c["w"], device="cuda"); conf = torch.empty_like(depth)
                kw = dict(basefocal=bf, disparity=torch.from_numpy(sc["disparity"]).cuda()) if bf else {}
                o = pyvoldor.voldor_device(torch.from_numpy(flows).cuda(), fx, fy, cx, cy, config=c["cfg"], depth_out=depth, depth_conf_out=conf, **kw)
                o = dict(o, depth=depth.cpu().numpy(), depth_conf=conf.cpu().numpy())
            else:
                kw = dict(basefocal=bf, disparity=sc["disparity"]) if bf else {}
                o = pyvoldor.voldor(flows, fx, fy, cx, cy, config=c["cfg"], **kw)
            outs.append(o)
    finally:
        for k, v in prev.items():
            hooks.debug_switch(k, v)
    return outs, dict(fb_blocks=hooks.debug_counter("fb_blocks_rode"), reduces=hooks.debug_counter("reduces_rode"), side=hooks.debug_counter("fb_side_passes"), plan=plan)


def _plan_rides(c):
    """what the host-side dealing (vk_debug_fb_ride_plan, held to its invariants in tests/test_fb_ride_plan.py) says for this geometry"""
    import ctypes as C
    from voldor_amd import capi
    out = (C.c_int * (5 + 3 * 16))()
    n_dp = 1 if c.get("basefocal") else 0
    assert capi.lib().vk_debug_fb_ride_plan(c["w"], c["h"], c["n"], n_dp, out, len(out)) == 5 + 3 * c["n"]
    return int(out[0]), int(out[2]) + int(out[3])  # (stacks that ride: 0 | 1 | 2, slots per EM iteration)


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("device", [False, True])
def test_riding_work_changes_no_bit_of_a_window(case, device):
    plain, rode0 = _run(case, {"fb_ride": 0, "defer_reduce": 0}, device)
    riding, rode1 = _run(case, {"fb_ride": 1, "defer_reduce": 1}, device)
    # the comparison below is only worth something if the second run DID move work into other launches: counted where the
    # launches are built.  Nothing rides with the switches off; with them on every EM iteration but the last leaves its density reduction to the
    # next trace, and fb_smooth rides in every iteration without the refit where the dealing says the geometry allows it -- whole passes at a time
    assert rode0["fb_blocks"] == 0 and rode0["reduces"] == 0, rode0
    assert rode1["reduces"] >= 2 * 2, rode1  # two windows, at least three EM iterations each
    on, blocks_per_iter = rode1["plan"]
    if case.startswith("tall"):
        assert on >= 1
    if case == "forty_step_segments":
        assert on == 0
    expect_fb = on and "--fb_smooth 0" not in CASES[case]["cfg"] and "--rg_refine_last_only 0" not in CASES[case]["cfg"]
    if case == "truncates":
        assert rode1["fb_blocks"] > 0, rode1  # (the frame count changes on the way: the per-iteration block count with it)
    elif expect_fb:
        assert rode1["fb_blocks"] > 0 and rode1["fb_blocks"] % blocks_per_iter == 0 and rode1["fb_blocks"] >= 2 * 2 * blocks_per_iter, (rode1, blocks_per_iter)
    else:
        assert rode1["fb_blocks"] == 0, rode1
    if case == "truncates":
        assert plain[0]["n_registered"] < CASES[case]["n"]
    for a, b in zip(plain, riding):
        assert a["n_registered"] == b["n_registered"]
        for k in ("depth", "depth_conf", "poses", "poses_covar"):
            x, y = np.ascontiguousarray(a[k], np.float32), np.ascontiguousarray(b[k], np.float32)
            np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32), err_msg=f"{case}: {k}")
    # and the two windows of a run are one window twice
    for k in ("depth", "poses"):
        np.testing.assert_array_equal(np.asarray(riding[0][k], np.float32).view(np.uint32), np.asarray(riding[1][k], np.float32).view(np.uint32))


SIDE_CASES = {
    # strict mode: fb_smooth (the reference's recurrence, step by step) runs on a second stream next to the pose half (round 6)
    "strict": dict(case="five_cameras", suffix=" --strict_math 1 --reference_draw 1 --reference_svd 1", switches={}),
    "strict_with_prior": dict(case="eight_cameras_with_prior", suffix=" --strict_math 1", switches={}),
    "strict_refit_everywhere": dict(case="refit_in_every_iteration", suffix=" --strict_math 1", switches={}),
    "strict_truncates": dict(case="truncates", suffix=" --strict_math 1", switches={}),
    "strict_cuda_mode": dict(case="five_cameras", suffix=" --strict_math 1 --reference_draw 1 --reference_svd 1 --reference_rng 1 --reference_tex 1", switches={}),
}


@pytest.mark.parametrize("name", sorted(SIDE_CASES))
def test_fb_smooth_next_to_the_pose_half_changes_no_bit_of_a_window(name):
    sc_ = SIDE_CASES[name]
    old = CASES[sc_["case"]]
    CASES["_side"] = dict(old, cfg=old["cfg"] + sc_["suffix"])
    try:
        inside, c0 = _run("_side", dict(sc_["switches"], fb_side=0), device=False)
        beside, c1 = _run("_side", dict(sc_["switches"], fb_side=1), device=False)
    finally:
        CASES.pop("_side")
    assert c0["side"] == 0 and c0["fb_blocks"] == 0, (c0["side"], c0["fb_blocks"])
    assert c1["side"] >= 2 * 3 and c1["fb_blocks"] == 0, (c1["side"], c1["fb_blocks"])  # two windows, every EM iteration (the refit iterations too: nothing rides in a kernel there, the stream does not care)
    for a, b in zip(inside, beside):
        assert a["n_registered"] == b["n_registered"]
        for k in ("depth", "depth_conf", "poses", "poses_covar"):
            np.testing.assert_array_equal(np.ascontiguousarray(a[k], np.float32).view(np.uint32), np.ascontiguousarray(b[k], np.float32).view(np.uint32), err_msg=f"{name}: {k}")
