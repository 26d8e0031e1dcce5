"""The host coders that read untrusted bytes (csrc/octree_host.cpp: geometry blob version 1; csrc/rans_host.cpp: a
piece of a y string from a seek point) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU: damaged and cut
blobs, garbage states and positions — error codes or decoded data, never a report.  (Sanitizers run on the CPU build
only: the GPU boxes do not offer them.)"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "demo-learned-point-cloud-compression_amd", "csrc")


@pytest.mark.parametrize("name,needle", [("fuzz_octree_host", "fuzz:"), ("fuzz_rans_seek", "pieces equal serial: 1")])
def test_host_coders_under_sanitizers(tmp_path, name, needle):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / name)
    src = os.path.join(ROOT, "tests", "fuzz", name + ".cpp")
    build = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-pthread",
                            "-w", "-I", CSRC, "-I", os.path.join(ROOT, "include"), src, "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this toolchain has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert needle in run.stdout and "ERROR" not in run.stderr and "runtime error" not in run.stderr
