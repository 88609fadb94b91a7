"""Build librag_mi355x.so (HIP, gfx950) in-tree.

`python -m rag_system_amd.build` cross-compiles every csrc/*.hip with hipcc for gfx950 (no
GPU needed) and links them into rag_system_amd/librag_mi355x.so, which travels to the GPU
box with the source tree.  Nothing is built at import time.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
LIB_PATH = os.path.join(HERE, "librag_mi355x.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off for sparse.hip/cand.hip: float64 BM25 / TF-IDF scores must be the same
# IEEE operation sequence numpy performs (no fused multiply-add).
COMMON = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function"]
PER_FILE = {
    "sparse.hip": ["-ffp-contract=off"],
    "cand.hip": ["-ffp-contract=off"],
}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "rag_mi355x.h"))
    return max(os.path.getmtime(p) for p in hdrs)


def _compile(src, force):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(sp)
            and os.path.getmtime(obj) >= _deps_mtime()):
        return obj
    cmd = [HIPCC] + COMMON + PER_FILE.get(src, []) + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_lib(force=False, jobs=8):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs)):
        return LIB_PATH
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
