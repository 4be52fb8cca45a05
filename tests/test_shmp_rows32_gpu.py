"""The 32-row form of the bf16x6 layer kernel (DESCO_SHMP_ROWS=32, kept for A/B runs against the 16-row
kernel of the product path) still passes the fused-layer and fused-pooling parity tests.  The tile form is
read once per process, so the tests run in a child process with the variable set."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_32_row_layer_kernel_parity_in_child_process():
    env = dict(os.environ, DESCO_SHMP_ROWS="32")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_kernels_gpu.py"), "-k",
                        "fused_shmp_layer or fused_pooling"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
