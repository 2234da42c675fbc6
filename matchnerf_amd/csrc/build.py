"""Build libmnerf_hip.so (gfx950) in-tree:  python -m matchnerf_amd.csrc.build

hipcc cross-compiles without a GPU.  The .so lands next to the package
(matchnerf_amd/libmnerf_hip.so): git-ignored, but it travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libmnerf_hip.so")
SOURCES = ["api.cpp", "backward.hip", "composite.hip", "conv.hip", "cost_volume.hip", "decoder.hip", "encoder_block.hip", "geometry.hip",
           "instance_norm.hip", "qkv.hip", "render_chunk.hip", "window_attention.hip"]
DECODER_SOURCES = ["decoder.hip", "split_f16.hpp", "cv_walk.hpp", "common.hpp"]  # what decoder_kernel is compiled from
# -amdgpu-use-amdgpu-trackers: the AMDGPU register-pressure trackers in the scheduler; the fused decoder spills 57 instead
# of 108 vector registers with them (decoder 20.8 -> 20.15 ms per frame on MI355X), everything else is unchanged
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm",
         "-amdgpu-use-amdgpu-trackers=1"]


def source_hash():
    """sha256 over everything the decoder kernel is compiled from (its sources, the argument structs it takes from the
    ABI header, the compiler flags): identifies the build that profile-derived numbers (profiles/decoder_counters.json)
    belong to, so that bench.py can tell when they have gone stale.  (Other kernels' sources and the rest of the header
    are left out: a change in the encoder does not change the decoder's counters.)"""
    import hashlib
    import re
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for name in sorted(DECODER_SOURCES):
        with open(os.path.join(HERE, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(PKG, "..", "include", "mnerf.h")) as f:
        header = f.read()
    for st in ("mnerf_view", "mnerf_rays", "mnerf_scene", "mnerf_decoder"):
        m = re.search(r"typedef struct %s \{.*?\} %s;" % (st, st), header, re.S)
        h.update(m.group(0).encode())
    return h.hexdigest()[:16]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(HERE, h) for h in ("common.hpp", "cv_walk.hpp", "split_f16.hpp")] + [os.path.join(PKG, "..", "include", "mnerf.h")]
    objs = []
    for src in SOURCES:
        sp = os.path.join(HERE, src)
        if not os.path.exists(sp):
            raise FileNotFoundError(sp)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if force or _newer(sp, obj) or any(_newer(d, obj) for d in deps):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or any(_newer(o, OUT) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
