#!/usr/bin/env python3
"""Build the reference's own Cython kernels into oracle/_ref/ (TEST INFRASTRUCTURE).

Compiles, unmodified and from where they lie under /root/reference, the three
.pyx files that ARE the hot path:

    template_speech_rec/compute_likelihood_linear_filter.pyx
    template_speech_rec/code_parts.pyx
    template_speech_rec/spread_waliji_patches.pyx

with this image's Cython + gcc (-O2, no -ffast-math: the reference's
setup.py:35-41 passes no extra flags either).  This is our own recipe, not the
reference's setup.py.  Generated C and object files go to a throw-away temp
directory; ONLY the resulting extension modules (.so) are placed in
oracle/_ref/, which is git-ignored (never committed) but travels to the GPU box
with the snapshot like our own built libraries.

Used for: (i) pinning oracle/tsr_oracle.c bit-for-bit, (ii) generating
tests/golden/*.npz (tests/golden/make_golden.py), (iii) optionally the
"reference" CPU baseline in bench.py.  Never imported by the product path.
"""
import os
import shutil
import sys
import tempfile

REF_PKG = "/root/reference/template_speech_rec"
MODULES = ["compute_likelihood_linear_filter", "code_parts", "spread_waliji_patches"]
# compiled the same way, for the widenings of SURVEY 8(f): cluster_times.pyx (detection clustering, N2)
EXTRA_MODULES = ["cluster_times"]
# code_parts.pyx once more with language_level=2 into _ref/py2/: code_parts_mag_aux (:138-242) divides
# unsigned ints with `/`, which is C integer division only under Python-2 semantics (N4)
PY2_MODULES = ["code_parts"]
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def built(mods=None, sub=""):
    mods = MODULES + EXTRA_MODULES if mods is None else mods
    d = os.path.join(OUT, sub) if sub else OUT
    ok = all(any(f.startswith(m + ".") and f.endswith(".so") for f in os.listdir(d))
             for m in mods) if os.path.isdir(d) else False
    if ok and mods is not None and not sub:
        return ok
    return ok and (sub != "" or built(PY2_MODULES, "py2"))


def build(force=False):
    if not os.path.isdir(REF_PKG):
        return False  # GPU box: use whatever was prebuilt
    if built() and not force:
        return True
    import numpy
    from setuptools import Extension
    from setuptools.dist import Distribution
    from Cython.Build import cythonize

    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="tsr_ref_build_")
    try:
        exts = [Extension(m, [os.path.join(REF_PKG, m + ".pyx")],
                          include_dirs=[numpy.get_include()],
                          extra_compile_args=["-O2"],
                          define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")])
                for m in MODULES + EXTRA_MODULES]
        # generated .c goes under tmp (the reference tree is read-only and no
        # reference text may land in this repository)
        exts = cythonize(exts, language_level="3str", force=True, quiet=True,
                         build_dir=os.path.join(tmp, "cy"))
        dist = Distribution({"name": "tsr_ref", "ext_modules": exts})
        cmd = dist.get_command_obj("build_ext")
        cmd.build_lib = os.path.join(tmp, "lib")
        cmd.build_temp = os.path.join(tmp, "obj")
        cmd.ensure_finalized()
        cmd.run()
        for f in os.listdir(cmd.build_lib):
            if f.endswith(".so"):
                shutil.copy2(os.path.join(cmd.build_lib, f), os.path.join(OUT, f))
        # the Python-2-semantics build
        os.makedirs(os.path.join(OUT, "py2"), exist_ok=True)
        exts2 = [Extension(m, [os.path.join(REF_PKG, m + ".pyx")], include_dirs=[numpy.get_include()],
                           extra_compile_args=["-O2"],
                           define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")]) for m in PY2_MODULES]
        exts2 = cythonize(exts2, language_level=2, force=True, quiet=True, build_dir=os.path.join(tmp, "cy2"))
        dist2 = Distribution({"name": "tsr_ref_py2", "ext_modules": exts2})
        cmd2 = dist2.get_command_obj("build_ext")
        cmd2.build_lib = os.path.join(tmp, "lib2")
        cmd2.build_temp = os.path.join(tmp, "obj2")
        cmd2.ensure_finalized()
        cmd2.run()
        for f in os.listdir(cmd2.build_lib):
            if f.endswith(".so"):
                shutil.copy2(os.path.join(cmd2.build_lib, f), os.path.join(OUT, "py2", f))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return built()


def load():
    """Import the three compiled reference modules (or return None if absent)."""
    if not built(MODULES):
        return None
    import importlib
    sys.path.insert(0, OUT)
    try:
        return tuple(importlib.import_module(m) for m in MODULES)
    finally:
        sys.path.remove(OUT)


def load_extra(name, py2=False):
    """Import one of EXTRA_MODULES, or (py2=True) the Python-2-semantics build of a module (or None)."""
    sub = "py2" if py2 else ""
    if not built([name], sub):
        return None
    import importlib
    import importlib.util
    d = os.path.join(OUT, sub) if sub else OUT
    if py2:  # same module name as the regular build: load by path under an alias
        f = [x for x in os.listdir(d) if x.startswith(name + ".") and x.endswith(".so")][0]
        spec = importlib.util.spec_from_file_location(name, os.path.join(d, f))
        saved = sys.modules.pop(name, None)
        try:
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        finally:
            sys.modules.pop(name, None)
            if saved is not None:
                sys.modules[name] = saved
        return mod
    sys.path.insert(0, d)
    try:
        return importlib.import_module(name)
    finally:
        sys.path.remove(d)


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref:", "built" if ok else "NOT built", sorted(os.listdir(OUT)) if os.path.isdir(OUT) else [])
    sys.exit(0 if ok else 1)
