"""The opt-in 32-row-slab transposed K = 256 GEMM (csrc/experimental/gemm_t32.hip, COATI_T32=1): the operator tests whose shapes it takes
(24 577 .. 65 536 rows, K = 256: plain bf16, QKV + RoPE, NewGELU + derivative codes, FC2 input gradient, lm_head partial
cross-entropy and its gradient) and the packed-against-padded engine steps at the wide sizes, re-run in a process that has
the switch set (the library reads it once)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _needs_experimental_build():
    """the kernel under test lives in csrc/experimental/ and is only compiled under COATI_AMD_EXPERIMENTAL=1 (build.py)"""
    from coati_amd import _lib
    if not _lib.has_experimental():
        pytest.skip("csrc/experimental/ is not in this library: build and run with COATI_AMD_EXPERIMENTAL=1")


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("what", [
    ["tests/test_gpu_ops.py", "-k", "gemm_epilogues or lmhead_ce or gemm_qkv_rope"],
    ["tests/test_gpu_packed.py", "-k", "wide"],
])
def test_operator_and_packed_tests_with_the_t32_kernel(what):
    env = dict(os.environ, COATI_T32="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + what, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
