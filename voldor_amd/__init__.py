"""voldor_amd -- MI355X-native (gfx950, hand-written HIP) implementation of VOLDOR's per-frame
dense visual-odometry inner loop, behind the reference's own pyvoldor / gpu_kernels boundary.

    from voldor_amd import pyvoldor
    out = pyvoldor.voldor(flows, fx, fy, cx, cy, basefocal, disparity=..., config="--silent ...")

See DESIGN.md for the hot path, INTEGRATION.md for the reference-side bindings.
"""
__all__ = ["pyvoldor", "capi", "kernels", "synth", "dist", "build"]
