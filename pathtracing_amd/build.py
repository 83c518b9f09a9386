"""Builds the native libraries in-tree.

    libptrace.so    pathtracing_amd/csrc   HIP kernels + C ABI (hipcc, gfx950)
    libptmaster.so  pathtracing_amd/host   C++ host mirror + synthetic scenes (g++)
    libptoracle.so  oracle/_build          CPU oracle, test infrastructure (gcc)

Run as `python -m pathtracing_amd.build` or through __graft_entry__.build().
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pathtracing_amd", "csrc")
HOST = os.path.join(ROOT, "pathtracing_amd", "host")
ORACLE = os.path.join(ROOT, "oracle")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# -ffp-contract=off: the only fused multiply-adds are the explicit ones of
# include/ptrace_fmath.h, on the device and on the host side alike.
HIP_FLAGS = [
    "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=cwd, check=True)


def build_ptrace(force=False):
    out = os.path.join(CSRC, "libptrace.so")
    srcs = [os.path.join(CSRC, f) for f in ("ptrace_api.hip", "bvh.cpp", "bvh.hpp", "pt_device.hpp", "pt_wavefront.hpp", "pt_pool.hpp", "pt_screen.hpp", "pt_refit.hpp", "pt_build.hpp", "pt_build_sah.hpp")]
    srcs += [os.path.join(ROOT, "include", f) for f in ("ptrace.h", "ptrace_fmath.h")]
    if force or _newer(out, srcs):
        _run([HIPCC] + HIP_FLAGS + ["-o", out, "ptrace_api.hip", "bvh.cpp", "-lpthread"], CSRC)
    return out


def build_ptmaster(force=False):
    out = os.path.join(HOST, "libptmaster.so")
    srcs = [os.path.join(HOST, f) for f in ("scene_gen.cpp", "ray_tracing_master.cpp", "scene.hpp",
                                            "ray_tracing_master.hpp", "ptmaster.h")]
    srcs.append(os.path.join(CSRC, "libptrace.so"))
    if force or _newer(out, srcs):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
              "-o", out, "scene_gen.cpp", "ray_tracing_master.cpp",
              "-L" + CSRC, "-lptrace", "-Wl,-rpath,$ORIGIN/../csrc"], HOST)
    return out


def build_demo(force=False):
    """The compiled host (no Python): host/ptrace_demo."""
    out = os.path.join(HOST, "ptrace_demo")
    srcs = [os.path.join(HOST, f) for f in ("ptrace_demo.cpp", "ray_tracing_master.hpp", "scene.hpp", "libptmaster.so")]
    if force or _newer(out, srcs):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-o", out, "ptrace_demo.cpp", "-L" + HOST, "-lptmaster", "-L" + CSRC, "-lptrace",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../csrc"], HOST)
    return out


def build_oracle(force=False):
    out = os.path.join(ORACLE, "_build", "libptoracle.so")
    if force:
        _run(["make", "-C", ORACLE, "clean"], ROOT)
    _run(["make", "-C", ORACLE], ROOT)
    return out


def build_all(force=False):
    return [build_ptrace(force), build_ptmaster(force), build_demo(force), build_oracle(force)]


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
