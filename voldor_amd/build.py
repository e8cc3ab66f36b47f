"""Build libvoldor_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m voldor_amd.build [--force]

The shared library lands in voldor_amd/lib/ (git-ignored, but shipped to the GPU box with the
repo snapshot).  hipcc cross-compiles gfx950 code objects without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvoldor_hip.so")
SOURCES = ["vk_depth_i6.hip", "vk_depth_i12.hip", "vk_depth_i16.hip", "vk_depth_s8.hip", "vk_depth_s16.hip", "vk_depth_i8.hip", "vk_depth_i4.hip", "vk_abi.hip", "vk_depth.hip", "vk_pose.hip", "vk_strict.hip", "vk_bootstrap.hip", "vk_voldor.hip", "vk_slam.hip", "vk_align.hip", "vk_dist.hip"]
# test-only library (host builds of the per-lane math + device-vs-host probes): tests/cxx/vk_testhooks.hip, built by build_test_lib()
# with the product flags; nothing in the product loads it and the product build does not depend on it.
TEST_LIB = os.path.join(LIBDIR, "libvoldor_hip_test.so")
TEST_SRC = os.path.join(os.path.dirname(HERE), "tests", "cxx", "vk_testhooks.hip")
FAKE_RCCL_LIB = os.path.join(LIBDIR, "libfake_rccl_test.so")
# The pose half (one hypothesis per lane) must reproduce the reference's fp32/fp64 rounding
# sequence to stay inside the pose tolerance (vk_p3p.hpp NUMERICS NOTE): no fma contraction there.
# It is a few hundred microseconds of work per window, so this costs nothing measurable; the
# per-pixel kernels of vk_depth.hip keep contraction (geometry there opts out via pragmas).
PER_FILE_FLAGS = {"vk_pose.hip": ["-ffp-contract=off"], "vk_bootstrap.hip": ["-ffp-contract=off"],
                  "vk_strict.hip": ["-ffp-contract=off"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result",
         # the SLP vectorizer packs scalar fp32 math into v_pk_* pairs and pays for it with ~30 % extra
         # v_mov traffic and 2x the registers in the latency-bound per-pixel kernels (k_cost_rand: 3335 ->
         # 2541 VALU instructions, 151 -> fewer VGPRs without it); results are equal up to fma-contraction choices.
         "-fno-slp-vectorize",
         # the device code of the per-frame-bound instantiations is most of the library (12 MB of 12.3): stored compressed (zstd, unpacked by
         # the HIP runtime when the module loads) the library is ~2 MB
         "--offload-compress"]


LINK_LIBS = ["-ldl"]  # vk_dist.hip binds RCCL with dlopen at the first vk_dist_* call


def _hipcc() -> str:
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.sep not in p or os.path.exists(p)):
            return p
    return "hipcc"


def _deps():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    inc = os.path.join(os.path.dirname(HERE), "include")
    out += [os.path.join(inc, f) for f in os.listdir(inc)]
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """The product library.  Depends on nothing under tests/ (ADVICE r2): the test-hook library is build_test_lib()."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    newest = max(os.path.getmtime(p) for p in _deps())
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
            return obj
        cmd = [hipcc] + FLAGS + PER_FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # -z now: every HIP entry point is bound when the library loads.  With lazy binding a call made for the first time AFTER another HIP runtime entered the
    # process (torch ships its own libamdhip64 and loads it globally) resolved to THAT runtime, while the streams and events the library already held came
    # from the first one: hipErrorUnknown on the first hipStreamWaitEvent (round 6: __graft_entry__.build() loads the library, smoke() then imports torch)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,now", "-o", LIB] + objs + LINK_LIBS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_test_lib(force: bool = False, verbose: bool = False):
    """tests/cxx/vk_testhooks.hip -> libvoldor_hip_test.so (tests/hooks.py, __graft_entry__.build()).  Returns None where tests/ is
    not shipped."""
    if not os.path.exists(TEST_SRC):
        return None
    os.makedirs(LIBDIR, exist_ok=True)
    newest = max(os.path.getmtime(p) for p in _deps() + [TEST_SRC] + [os.path.join(os.path.dirname(TEST_SRC), f) for f in os.listdir(os.path.dirname(TEST_SRC))])
    if not force and os.path.exists(TEST_LIB) and os.path.getmtime(TEST_LIB) >= newest and os.path.exists(FAKE_RCCL_LIB) and os.path.getmtime(FAKE_RCCL_LIB) >= newest:
        return TEST_LIB
    cmd = [_hipcc()] + FLAGS + ["-ffp-contract=off", "-shared", "-o", TEST_LIB, TEST_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    fake = os.path.join(os.path.dirname(TEST_SRC), "fake_rccl.cpp")  # file-backed stand-in for librccl (two ranks on one GPU, tests only)
    if os.path.exists(fake):
        cmd = [_hipcc(), "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", FAKE_RCCL_LIB, fake]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return TEST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--no-tests" not in sys.argv:
        print(build_test_lib(force="--force" in sys.argv, verbose=True))
