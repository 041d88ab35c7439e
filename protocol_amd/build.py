"""Build libpm_engine.so (HIP kernels + engine + host helpers) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(HERE, "libpm_engine.so")
SOURCES = ["pm_kernels.hip", "pm_engine.cpp", "pm_host.cpp"]
HEADERS = ["pm_device.h", "pm_internal.h", "pm_members.h", "pm_validate.inc", "pm_propose.inc", "pm_prep.inc",
           "pm_carve_kernel.inc", "pm_stream.inc", "pm_launch.inc",
           "pm_engine_types.inc", "pm_engine_state.inc", "pm_engine_groups.inc", "pm_engine_carve.inc", "pm_engine_match.inc",
           "pm_engine_merge.inc", "pm_engine_workers.inc", "pm_engine_tasks.inc", "pm_engine_api.inc", "pm_engine_tick.inc",
           "pm_engine_dist.inc", "pm_engine_debug.inc", "pm_unicode_lower.inc"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(INCLUDE, h) for h in ("pm_engine.h", "pm_engine_debug.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defines=(), out: str | None = None) -> str:
    out = out or LIB_PATH
    if not force and out == LIB_PATH and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-I", INCLUDE, "-I", CSRC, "-Wall", "-Wno-unused-result", *[f"-D{d}" for d in defines],
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


# ---- the host side above the C ABI, compiled: protocol_amd/plugin (GpuMatchPlugin, Scheduler — the C++ twin of
# rust/gpu_match_plugin.rs).  Plain C++17, no HIP: g++, linked against libpm_engine.so next to it.
PLUGIN_DIR = os.path.join(HERE, "plugin")
PLUGIN_LIB = os.path.join(HERE, "libpm_plugin.so")
PLUGIN_SOURCES = ["gpu_match_plugin.cpp", "pm_plugin_c.cpp"]
PLUGIN_HEADERS = ["gpu_match_plugin.hpp", "pm_plugin_c.h", "pm_plugin_c_internal.hpp"]
# the communicators of GpuMatchPlugin::tick_dist (RCCL over xGMI; in-process ranks) + their C face: links librccl, libamdhip64
PLUGIN_DIST_LIB = os.path.join(HERE, "libpm_plugin_dist.so")
PLUGIN_DIST_SOURCES = ["rccl_all_gather.cpp", "pm_plugin_dist_c.cpp"]
PLUGIN_DIST_HEADERS = ["rccl_all_gather.hpp", "pm_plugin_dist_c.h"]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _cxx() -> str:
    for cand in (shutil.which("g++"), shutil.which("c++"), _hipcc()):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("no C++ compiler found")


def plugin_needs_build() -> bool:
    if not os.path.exists(PLUGIN_LIB):
        return True
    t = os.path.getmtime(PLUGIN_LIB)
    deps = [os.path.join(PLUGIN_DIR, f) for f in PLUGIN_SOURCES + PLUGIN_HEADERS] + [os.path.join(INCLUDE, "pm_engine.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_plugin(force: bool = False, verbose: bool = False) -> str:
    """libpm_plugin.so: its pm_* symbols are libpm_engine.so's (found beside it through $ORIGIN)."""
    build()  # (the engine library it links against)
    if not force and not plugin_needs_build():
        return PLUGIN_LIB
    cmd = [_cxx(), "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", INCLUDE, "-I", PLUGIN_DIR,
           *[os.path.join(PLUGIN_DIR, s) for s in PLUGIN_SOURCES], "-L", HERE, "-lpm_engine", "-Wl,-rpath,$ORIGIN",
           "-lpthread", "-o", PLUGIN_LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(PLUGIN_LIB + ".tmp", PLUGIN_LIB)
    return PLUGIN_LIB


def plugin_dist_needs_build() -> bool:
    if not os.path.exists(PLUGIN_DIST_LIB):
        return True
    t = os.path.getmtime(PLUGIN_DIST_LIB)
    deps = [os.path.join(PLUGIN_DIST_DIR, f) for f in PLUGIN_DIST_SOURCES + PLUGIN_DIST_HEADERS + PLUGIN_HEADERS] + [PLUGIN_LIB]
    return any(os.path.getmtime(d) > t for d in deps)


PLUGIN_DIST_DIR = PLUGIN_DIR


def build_plugin_dist(force: bool = False, verbose: bool = False) -> str:
    """libpm_plugin_dist.so: RcclAllGather / LocalAllGather + pmx_tick_dist & co.; plain host C++ (g++) against the HIP
    runtime's and RCCL's C APIs."""
    build_plugin()
    if not force and not plugin_dist_needs_build():
        return PLUGIN_DIST_LIB
    cmd = [_cxx(), "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I", INCLUDE, "-I", PLUGIN_DIR,
           "-isystem", os.path.join(ROCM, "include"), *[os.path.join(PLUGIN_DIR, s) for s in PLUGIN_DIST_SOURCES],
           "-L", HERE, "-lpm_plugin", "-lpm_engine", "-L", os.path.join(ROCM, "lib"), "-lrccl", "-lamdhip64",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread", "-o", PLUGIN_DIST_LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(PLUGIN_DIST_LIB + ".tmp", PLUGIN_DIST_LIB)
    return PLUGIN_DIST_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_plugin(force=True, verbose=True))
    print(build_plugin_dist(force=True, verbose=True))
