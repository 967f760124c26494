"""Builds libmpn_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m multiposenet_amd.build [--force]

Every `csrc/*.hip` is compiled to an object (in parallel, only when out of date) and
linked into `multiposenet_amd/libmpn_hip.so`. The .so is git-ignored but travels to the
GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmpn_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-munsafe-fp-atomics"] + os.environ.get("MPN_EXTRA_FLAGS", "").split()   # diagnostic builds


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, force, hdr_m):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src[:-4] + ".o")
    if (not force and os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(s)
            and os.path.getmtime(o) >= hdr_m):
        return o, False
    cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return o, True


def build_all(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _headers_mtime()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr_m), srcs))
    objs = [o for o, _ in res]
    rebuilt = any(c for _, c in res)
    if rebuilt or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[mpn.build] linked {LIB} from {len(objs)} objects")
    elif verbose:
        print(f"[mpn.build] {LIB} up to date")
    return LIB


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
