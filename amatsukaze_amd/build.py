"""Builds the in-tree HIP library (amatsukaze_amd/libamt_gpu.so) for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libamt_gpu.so")

SOURCES = [
    "amt_gpu.hip",
    "amt_gpu_erase_scan.hip",
    "amt_gpu_stats.hip",
    "amt_gpu_ingest.hip",
    "amt_gpu_upload.hip",
    "eval_engine.hip",
    "eval_fused_kernels.hip",
    "eval_linear_kernels.hip",
    "eval_pair_kernels.hip",
    "erase_scan_kernels.hip",
    "stats_kernels.hip",
    "ingest_kernels.hip",
    "logo_model.cpp",
    "logo_fit.cpp",
    "decisions.cpp",
    "stats_decisions.cpp",
    "amts_file.cpp",
]

# -ffp-contract=off: the reference is built without FMA contraction (MSVC /fp:precise) and its scores
# feed discontinuous decisions -- the kernels must round exactly where it rounds.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


# per-file extras.  eval_fused_kernels.hip: the fade loop is ~200 straight-line VALU instructions per iteration with
# 8-cycle dependent-issue latency (tools/ubench/valu_rate.hip); the max-ILP scheduler spaces dependent packed ops
# further apart than the default occupancy-driven one (measured: 2.58 -> 2.40 ms per 2048-frame analysis).
# eval_linear_kernels.hip: no SLP vectorisation -- a packed fp32 op issues in 4 cycles against 2 for a plain one (tools/ubench/valu_rate.hip), so
# pairing the per-fade scalar code gains nothing and costs the v_mov shuffles that assemble the pairs, and a DPP operand cannot fold into
# a packed add (measured: 3.21 -> 3.12 ms per 10 000 frames, outputs identical; profiles/r05_notes.md).
EXTRA_FLAGS = {"eval_fused_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
               "eval_linear_kernels.hip": ["-fno-slp-vectorize"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build(srcs) -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [CSRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "amt_gpu.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, defines, extra_flags=(), file_flags=None) -> str:
    """An instrumented copy of the library (tools/phase_timing.py): amatsukaze_amd/libamt_gpu_<name>.so
    file_flags: {source file: [flags]} on top of EXTRA_FLAGS, for experiments with one translation unit's compiler options"""
    file_flags = file_flags or {}
    out = os.path.join(HERE, f"libamt_gpu_{name}.so")
    bdir = os.path.join(HERE, "build", name)
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], []
    for f in SOURCES:
        o = os.path.join(bdir, f + ".o")
        objs.append(o)
        # (-DAMT_INSTRUMENTED_BUILD: csrc/build_knobs.h refuses every knob without it -- the release build() below never sets it)
        cmd = [hipcc(), *FLAGS, *EXTRA_FLAGS.get(f, []), *file_flags.get(f, []), *extra_flags, "-DAMT_INSTRUMENTED_BUILD", *[f"-D{d}" for d in defines], "-x", "hip", "-c",
               os.path.join(CSRC, f), "-o", o]
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for f, p in procs:
        o_, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"variant build failed on {f}:\n" + o_.decode(errors="replace"))
    r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("variant link failed:\n" + r.stdout.decode(errors="replace"))
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not needs_build(srcs):
        return OUT
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        newest_dep = max([os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
                         + [os.path.getmtime(s), os.path.getmtime(os.path.join(HERE, "..", "include", "amt_gpu.h"))])
        if not force and os.path.exists(o) and os.path.getmtime(o) > newest_dep:
            continue
        cmd = [hipcc(), *FLAGS, *EXTRA_FLAGS.get(os.path.basename(s), []), "-x", "hip", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
