"""The C ABI used from plain C++ (no Python / torch in the process): builds examples/capi_demo.cpp
against libdesco_hip.so and runs it."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_demo_builds_and_runs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "capi_demo")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "capi_demo.cpp"),
                           "-L", os.path.join(ROOT, "desco_amd"), "-ldesco_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "desco_amd"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("ok")
