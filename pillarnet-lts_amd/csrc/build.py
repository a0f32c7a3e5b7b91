"""Builds libpillarnet_hip.so (gfx950) in-tree with hipcc.

One object per .hip file (compiled in parallel, rebuilt only when the source or
a header is newer), then one shared link.  No torch headers: the library is a
plain C-ABI .so (include/pillarnet_hip.h) that Python reaches through ctypes.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
INCLUDE = os.path.join(ROOT, "include")
OUT = os.path.join(PKG, "libpillarnet_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = [
    "-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}",
    "-ffp-contract=off",  # torch runs sub / mul / add as separately rounded kernels
    # no packed-fp32 VALU forms (v_pk_{add,mul,fma}_f32): hipcc -O3 would otherwise emit them from
    # ordinary float code
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{HERE}",
]


# tuning builds: extra hipcc flags (e.g. -DPN_DECODE_STAMPS) from the environment
FLAGS += os.environ.get("PN_EXTRA_HIPCC_FLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, hdr_mtime, verbose):
    obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
    spath = os.path.join(HERE, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), hdr_mtime):
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", spath, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj, True


def build(verbose=True, force=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_mtime = float("inf") if force else _headers_mtime()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr_mtime, verbose), srcs))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(OUT):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
