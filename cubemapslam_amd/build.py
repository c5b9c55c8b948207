"""Build libcubemapslam_hip.so (hand-written HIP kernels + C-ABI) for gfx950, in-tree.

    python -m cubemapslam_amd.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
-ffp-contract=off is part of the numerical contract: key-point angles and tap positions must match the CPU
reference bit for bit (see csrc/cms_detmath.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libcubemapslam_hip.so")
HOST_LIB = os.path.join(LIB_DIR, "libcubemapslam_host.so")
DRIVER = os.path.join(LIB_DIR, "cubemap_closed_loop")


def _newer(srcs, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _sources(d):
    out = []
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith((".hip", ".h", ".inc", ".cpp", ".hpp")):
                out.append(os.path.join(root, f))
    return out


def build(force=False, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = _sources(CSRC) + [os.path.join(os.path.dirname(HERE), "include", "cubemapslam_hip.h")]
    if force or _newer(srcs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result",
               os.path.join(CSRC, "cms_lib.hip"), "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    host_dir = os.path.join(HERE, "host")
    hsrcs = _sources(host_dir)
    cpps = [s for s in hsrcs if s.endswith(".cpp") and not s.endswith("closed_loop_driver.cpp")]
    if cpps and (force or _newer(hsrcs + [LIB], HOST_LIB)):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
               "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", host_dir] + cpps + \
              ["-L", LIB_DIR, "-lcubemapslam_hip", "-Wl,-rpath,$ORIGIN", "-o", HOST_LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # the Python-free closed-loop driver (Examples/cubemap_lafida.cpp's role): a plain C++ program over the C-ABI and io_formats
    drv_src = os.path.join(host_dir, "closed_loop_driver.cpp")
    if os.path.exists(drv_src) and (force or _newer(hsrcs + [LIB], DRIVER)):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", host_dir,
               drv_src, os.path.join(host_dir, "io_formats.cpp"), "-L", LIB_DIR, "-lcubemapslam_hip", "-Wl,-rpath,$ORIGIN", "-o", DRIVER]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
