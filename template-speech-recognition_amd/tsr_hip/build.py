"""Build libtsr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).
Sources are compiled to objects one by one (only the stale ones) and linked."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "lib", "obj")
LIB = os.path.join(PKG, "lib", "libtsr_hip.so")
SOURCES = ["tsr_api.hip", "detect_exact.hip", "detect_exact_mfma.hip", "detect_mfma.hip", "detect_mfma_i0.hip", "detect_mfma_i1.hip", "detect_mfma_i2.hip", "detect_mfma_i3.hip", "detect_mfma_i4.hip", "detect_mfma_i5.hip", "parts.hip", "parts_mfma.hip", "classify.hip", "edges.hip", "edges2.hip", "detect_mfma_bf16.hip", "detect_mfma_tap2_i0.hip", "detect_mfma_tap2_i1.hip", "detect_mfma_tap2_i2.hip", "detect_mx.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]
# per-source extras: matrix instructions with VGPR accumulators (with AGPR accumulators the register allocator
# rotates the three accumulator tiles of detect_exact_mfma.hip through sixteen copies per loop iteration)
EXTRA = {"detect_exact_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
HEADERS = [os.path.join(CSRC, "tsr_internal.h"), os.path.join(CSRC, "mx_plan.h"), os.path.join(CSRC, "detect_mfma_kernel.inc"), os.path.join(CSRC, "detect_mfma_tap2.inc"),
           os.path.join(os.path.dirname(PKG), "include", "tsr.h")]


def _newer(path, deps):
    return not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for src in SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(OBJ, src + ".o")
        objs.append(o)
        if force or _newer(o, [s] + HEADERS):
            cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if force or procs or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
