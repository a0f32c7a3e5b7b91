"""Compiles the reference's own CPU rotated-IoU source, where it lies under /root/reference, into
oracle/_ref/libref_iou3d_cpu.so (git-ignored; travels to the GPU box with the snapshot).

Recipe: g++ on det3d/ops/iou3d_nms/src/iou3d_cpu.cpp + oracle/ref_driver.cpp, against libtorch
(the file includes <torch/extension.h>) and the CUDA runtime headers the image already ships inside
the triton wheel (the file includes <cuda.h> / <cuda_runtime_api.h> but uses nothing from them
beyond the `__device__` qualifier).  No reference build system is run and no stand-in header is
written.  Nothing else of the reference compiles here: its other sources are CUDA (`<<<>>>`).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/det3d/ops/iou3d_nms/src/iou3d_cpu.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libref_iou3d_cpu.so")


def cuda_header_dir():
    import triton
    d = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
    return d if os.path.exists(os.path.join(d, "cuda_runtime_api.h")) else None


def build(verbose=True):
    """Returns the .so path, or None when the reference (or a header it needs) is not present."""
    if os.path.exists(OUT) and (not os.path.exists(REF_SRC)
                                or os.path.getmtime(OUT) > os.path.getmtime(REF_SRC)):
        return OUT
    if not os.path.exists(REF_SRC):
        return None
    cuda_inc = cuda_header_dir()
    if cuda_inc is None:
        return None
    import torch
    tdir = os.path.dirname(torch.__file__)
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
           f"-I{cuda_inc}", f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include",
           f"-I{sysconfig.get_paths()['include']}",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           REF_SRC, os.path.join(HERE, "ref_driver.cpp"), "-o", OUT,
           f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
           f"-Wl,-rpath,{tdir}/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build())
