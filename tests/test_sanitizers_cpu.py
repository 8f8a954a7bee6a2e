"""Sanitizer builds of the CPU-side code ("race detection / sanitizers"; the reference compiles
its Cython with boundscheck=False, so out-of-range is undefined behaviour there).

* oracle/tsr_oracle.c under AddressSanitizer + UndefinedBehaviorSanitizer: the golden-vector and random tests of
  tests/test_oracle.py run against that build in a child process (libasan preloaded into the interpreter);
* the host (non-kernel) code of libtsr_hip.so under the same sanitizers (hipcc -fsanitize=address,undefined
  -fno-gpu-sanitize): the argument validation, error paths and the no-device behaviour that
  tests/test_abi_cpu.py exercises.  GPU AddressSanitizer is not available on the GPU pool; the kernels' own
  indexing is covered by the shape checks in front of every launch and by the parity tests."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer")


def _gcc_runtime(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_c_under_asan_ubsan():
    asan = _gcc_runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    env = dict(os.environ, TSR_ORACLE_SANITIZE="1", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    text = r.stdout + r.stderr
    assert not any(b in text for b in BAD), text[-4000:]
    assert r.returncode == 0, text[-4000:]
    assert os.path.exists(os.path.join(ROOT, "oracle", "liboracle_san.so"))


def test_host_code_of_the_library_under_asan_ubsan(tmp_path):
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    hipcc = "/opt/rocm/bin/hipcc"
    if not rt or not os.path.exists(hipcc):
        pytest.skip("no clang AddressSanitizer runtime / hipcc here")
    csrc = os.path.join(ROOT, "template-speech-recognition_amd", "csrc")
    lib = str(tmp_path / "libtsr_hip_san.so")
    # the host entry points live in tsr_api.hip; the kernel units are compiled without instrumentation of
    # device code (-fno-gpu-sanitize) and only at -O1: this build is never timed
    srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
    cmd = [hipcc, "-O1", "-g", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan", "-Wno-unused-parameter",
           "-DTSR_SAN_BUILD", "-o", lib] + srcs + ["-ldl"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, TSR_HIP_LIB=lib, LD_PRELOAD=rt[0],
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi_cpu.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    text = r.stdout + r.stderr
    assert not any(b in text for b in BAD), text[-4000:]
    assert r.returncode == 0, text[-4000:]
