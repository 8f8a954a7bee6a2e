"""Host logic of TSR_FAST | TSR_MX without a device: the plan of pair records, tiles and workgroups
(template-speech-recognition_amd/csrc/mx_plan.h, the code csrc/detect_mx.hip calls) compiled with g++ under
AddressSanitizer + UBSan and held to its invariants on random corpora by tests/cpp/mx_plan_check.cpp: every window start
of the shortest filter in exactly one pair record, pairs on even virtual rows inside their utterance, every tile and
every workgroup inside one LDS image, workgroups partitioning the tiles in order within the slots of the rounds the plan
settled on -- and the two shapes the design quotes: config 2 = 253 workgroups of 21 tiles on 256 CUs, the same corpus with
Lmin = 28 = two rounds of 11."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_mx_plan_invariants(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++ here")
    exe = str(tmp_path / "mx_plan_check")
    b = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-I", os.path.join(ROOT, "template-speech-recognition_amd", "csrc"),
                        os.path.join(ROOT, "tests", "cpp", "mx_plan_check.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    for seed in (1, 2, 3):
        r = subprocess.run([exe, "1500", str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip() == "ok 1500 cases", (r.stdout + r.stderr)[-3000:]
