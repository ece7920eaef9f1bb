"""Builds libphylocsf_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m phylocsfpp_amd._build          # or: from phylocsfpp_amd._build import build; build()

hipcc cross-compiles without a GPU.  The library lands next to this file so that it travels
with the repository snapshot to the GPU box and is visible as an in-tree native extension.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libphylocsf_amd.so")
SOURCES = ["kernels.hip", "expm.hip", "eigen.hip", "frontend.hip", "factored.hip", "tiles.hip", "chain.hip", "wig.hip", "hmm.hip", "api.cpp", "blocks.cpp", "model_abi.cpp", "hmm_abi.cpp", "comm.cpp"]
HEADERS = ["kernels.h", "hostmath.hpp", "schedule.hpp", "ctx.hpp", "optimizer.hpp", "machine.hpp", "frames.hpp", os.path.join("..", "..", "include", "phylocsf_amd.h"),
           os.path.join("..", "host", "model_io.hpp")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm is required to build phylocsfpp_amd)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    build_dir = os.path.join(HERE, "_build")
    os.makedirs(build_dir, exist_ok=True)
    common = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result",
              "-ffp-contract=off"]
    # The op loop of prune_fixed is a switch on a wave-uniform op code.  By default the AMDGPU back end
    # structurises even uniform regions, which turns the switch into a chain of "Flow" blocks whose phi copies
    # cost ~80 v_mov_b64 per op (and 26 spilled VGPRs).  Leaving uniform regions as plain scalar branches
    # removes both (no spills).
    per_source = {"kernels.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"],
                  "factored.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"],
                  "tiles.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"]}
    header_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    header_time = max(header_time, os.path.getmtime(os.path.abspath(__file__)))
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(build_dir, src + ".o")
        objs.append(obj)
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(header_time, os.path.getmtime(path)):
            continue  # object is newer than its source and every header
        cmd = [hipcc()] + common + per_source.get(src, []) + ["-x", "hip", "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((src, subprocess.Popen(cmd)))  # the translation units compile side by side
    failed = [src for src, pr in jobs if pr.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
