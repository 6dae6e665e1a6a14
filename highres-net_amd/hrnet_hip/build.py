"""Build libhrnet_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python highres-net_amd/hrnet_hip/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libhrnet_hip.so")
SOURCES = ["api.hip", "prof.hip", "conv3x3.hip", "conv3x3_r64.hip", "conv3x3_v6.hip", "conv3x3_v6x3.hip", "backward.hip", "wgrad_x3.hip", "decoder_bwd.hip", "train.hip", "stem.hip", "decoder.hip", "lanczos.hip", "lanczos_bwd.hip", "shiftnet.hip", "shiftnet_bwd.hip", "adam.hip", "losses.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdr.append(os.path.join(HERE, "..", "..", "include", "hrnet_hip.h"))
    return hdr


def build_library(force=False, verbose=True):
    """Compile every .hip source and link the shared library; returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr_time = _newest(_deps())
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append((s, o))

    def run(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    failed = False
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for cmd, r in ex.map(run, jobs):
            if verbose:
                print(" ".join(cmd))
            if r.returncode != 0:
                failed = True
            if r.stdout or r.stderr:
                sys.stderr.write(r.stdout + r.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


IO_SRC = os.path.join(HERE, "csrc_io", "hrnet_io.cpp")
IO_LIB = os.path.join(HERE, "libhrnet_io.so")


def build_io_library(force=False, verbose=True):
    """Compile the host-side input-pipeline library (g++, zlib, pthreads): libhrnet_io.so."""
    hdr = os.path.join(HERE, "..", "..", "include", "hrnet_io.h")
    if force or not os.path.exists(IO_LIB) or os.path.getmtime(IO_LIB) < max(os.path.getmtime(IO_SRC), os.path.getmtime(hdr)):
        cmd = [os.environ.get("CXX", "g++"), "-O3", "-fPIC", "-shared", "-std=c++17", "-Wall", "-o", IO_LIB, IO_SRC, "-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return IO_LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
    print(build_io_library(force="--force" in sys.argv))
