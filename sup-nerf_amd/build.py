"""Build recipe for libsupnerf_hip.so (gfx950 only).  In-tree output so the .so travels with the repo
snapshot to the GPU box.  Used by __graft_entry__.build() and runnable by hand:

    python sup-nerf_amd/build.py [--force] [--verbose]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsupnerf_hip.so")
STAMP = os.path.join(HERE, ".libsupnerf_hip.stamp")
SOURCES = ["snr_aux.hip", "snr_loss.hip", "snr_loop.hip", "snr_wgrad.hip", "snr_mlp.hip", "snr_mlp16.hip", "snr_mlp16_bwd.hip", "snr_mlp_bwd.hip", "snr_bf16.hip"]
HEADERS = ["snr_layout.h", "snr_device.hpp", "snr_host.hpp", "snr_mlp_core.hpp", "snr_mlp16_core.hpp", os.path.join("..", "..", "include", "supnerf_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            h.update(f.encode()); h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_library(force=False, verbose=False):
    """Compile every .hip source for gfx950 and link the shared library.  Returns its path."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if verbose or p.returncode != 0:
            print(out)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
