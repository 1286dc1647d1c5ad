"""Build libohw.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m openhush_amd.build [--force]

Objects are cached under openhush_amd/csrc/_build/ keyed by a hash of the source, the headers and
the flags; the shared library is written to openhush_amd/libohw.so so it travels with the tree.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# OHW_BUILD_VARIANT=trace builds libohw_trace.so with -DOHW_TRACE (in-kernel timestamps for tools/dec_trace.py);
# the product library never carries that code.  OHW_BUILD_VARIANT=asan builds libohw_asan.so with AddressSanitizer and
# UBSan on the HOST code only (GPU sanitizers are not available on the pool): tools/asan_host.sh runs the CPU suite on it.
VARIANT = os.environ.get("OHW_BUILD_VARIANT", "")
OUT = os.path.join(HERE, f"libohw_{VARIANT}.so" if VARIANT else "libohw.so")
BUILD = os.path.join(CSRC, f"_build_{VARIANT}" if VARIANT else "_build")
SOURCES = ["gemm.hip", "gemm256.hip", "attention.hip", "weights.hip", "mel.hip", "misc.hip", "decode.hip", "decode_persist.hip", "model.hip", "engine.hip", "host_engine.cpp", "pool.cpp", "dsp.cpp", "vad.cpp", "resample.hip", "tracker.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-ffp-contract=fast-honor-pragmas",
         "-fno-gpu-rdc"] + (["-DOHW_TRACE"] + os.environ.get("OHW_EXP_FLAGS", "").split() if VARIANT == "trace" else []) + (
             ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-g"]
             if VARIANT == "asan" else []) + ["-x", "hip"]
LINK_EXTRA = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined"] if VARIANT == "asan" else []


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _headers_digest() -> str:
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for name in sorted(os.listdir(root)):
            if name.endswith((".hpp", ".h")):
                with open(os.path.join(root, name), "rb") as f:
                    h.update(name.encode()); h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str, hdr: str, force: bool) -> str:
    path = os.path.join(CSRC, src)
    with open(path, "rb") as f:
        key = hashlib.sha256(f.read() + hdr.encode()).hexdigest()[:16]
    obj = os.path.join(BUILD, f"{os.path.splitext(src)[0]}.{key}.o")
    if force or not os.path.exists(obj):
        for old in os.listdir(BUILD):
            if old.startswith(os.path.splitext(src)[0] + ".") and old.endswith(".o"):
                os.remove(os.path.join(BUILD, old))
        cmd = [_hipcc()] + FLAGS + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    hdr = _headers_digest()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr, force), SOURCES))
    stamp = os.path.join(BUILD, "link.stamp")
    want = hashlib.sha256("\n".join(objs).encode()).hexdigest()
    have = open(stamp).read() if os.path.exists(stamp) else ""
    if force or have != want or not os.path.exists(OUT):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + LINK_EXTRA + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(want)
    if verbose:
        print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
