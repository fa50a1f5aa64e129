"""Compile-time resource check of the hot kernels (no GPU needed: hipcc cross-compiles gfx950 and reports registers,
spills and scratch per kernel).  Guards against the class of regression found in round 2: one store into a by-value
kernel-parameter struct made the compiler keep a private copy of the whole struct in scratch memory (gram_kernel_v3:
896 bytes per lane, 4.5 x slower) without a single line of the kernel's arithmetic changing."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpmp_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


_CACHE = {}


def _resources(src):
    if src in _CACHE:                      # one compile per source and test session (gemm_f64.hip: 24 template instances of the GEMM)
        return _CACHE[src]
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.devnull]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            res[name][m.group(1).strip()] = int(m.group(2))
    _CACHE[src] = res
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src, must_match", [
    ("gram.hip", ("gram_kernel_v3", "grad_trace_kernel", "pairwise_kernel", "gram_deriv_kernel")),
    ("gemm_f64.hip", ("gemm_f64_kernel", "gemm_nt_lean_kernel", "gemm_nt_small_kernel", "trsm_leaf_kernel")),
    ("trsv.hip", ("trsv_persist_kernel",)),
])
def test_hot_kernels_use_no_scratch_memory(src, must_match):
    res = _resources(src)
    assert res, "no resource remarks parsed"
    for frag in must_match:
        hit = {k: v for k, v in res.items() if frag in k}
        assert hit, (frag, sorted(res))
        for k, v in hit.items():
            assert v.get("ScratchSize", 0) == 0, (k, v)
            assert v.get("VGPRs Spill", 0) == 0, (k, v)       # (scalar registers spilled into vector lanes cost no memory access)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_register_budgets_the_design_depends_on():
    """The LDS-direct GEMM must stay at two workgroups per CU (<= 256 registers), the small-footprint kernel must fit beside them
    (<= 48), and the diagonal-block kernel must fit beside ONE GEMM wave per SIMD (<= 128)."""
    g = _resources("gemm_f64.hip")
    for k, v in g.items():
        if "gemm_f64_kernel_v2" in k:
            assert v["VGPRs"] <= 256, (k, v)
        if "gemm_nt_lean_kernel" in k:
            assert v["VGPRs"] <= 48, (k, v)
    p = _resources("potf2.hip")
    k = [x for x in p if "potf2_inv_kernel" in x]
    assert k and p[k[0]]["VGPRs"] <= 128, p
