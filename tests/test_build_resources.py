"""The BUILT library's own metadata (llvm-readelf --notes on its gfx950 code objects, tools/kernel_resources.py): no kernel may spill vector
registers or use scratch memory.  A spill reload is a `scratch_load` followed by `s_waitcnt vmcnt(0)` - it waits for every global load the
kernel has in flight, i.e. for exactly the prefetches the persistent kernels are built around - so "no scratch" is a property the schedule
relies on, not a nicety.  (SGPR spills go to lanes of a VGPR - v_writelane / v_readlane, no memory - and are reported, not refused.)"""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

LLVM = os.environ.get("TVC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
pytestmark = pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, "llvm-readelf")) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))),
                                reason="needs ROCm's llvm-readelf / llvm-objdump")


@pytest.fixture(scope="module")
def kernels():
    from tinyvc_amd import build
    import kernel_resources
    return kernel_resources.kernels(build.build(verbose=False))


def test_the_library_holds_its_kernels(kernels):
    names = [k["demangled"] for k in kernels]
    assert len(kernels) >= 150
    # the kernels every timeline shows (bench step, B = 1, a streaming block): each family must be in the binary that is checked
    for family in ("conv3s_kernel", "conv_s2_kernel", "film_s2_kernel", "gemm_s2_kernel", "cnx1_kernel", "cnx2_kernel", "conv48s_kernel", "conv48p_kernel",
                   "up24s_kernel", "down24f_kernel", "down0s_kernel", "knn_coarse_kernel", "knn_rescore_kernel", "knn_topk_split_kernel",
                   "stft_fft_kernel", "noise_ifft_kernel", "harm_synth_kernel", "sola_corr_kernel", "sola_kernel"):
        assert any(family in n for n in names), family


def test_no_kernel_spills_vector_registers_or_uses_scratch(kernels):
    bad = [(k["demangled"], k.get("vgpr_spill", 0), k.get("scratch", 0), k.get("dyn_stack"))
           for k in kernels if k.get("vgpr_spill", 0) or k.get("scratch", 0) or k.get("dyn_stack") == "true"]
    assert not bad, "kernels with VGPR spills / scratch:\n" + "\n".join(f"  {n}: vgpr_spill {v}, scratch {s} B, dynamic stack {d}" for n, v, s, d in bad)


def test_register_budgets_match_the_launch_bounds(kernels):
    """A kernel's unified VGPR + AGPR count must fit the waves its workgroup size puts on a SIMD (512 registers per SIMD lane, granule 8)."""
    for k in kernels:
        waves_per_simd = -(-(k["wg"] // 64) // 4)
        budget = min(512 // max(waves_per_simd, 1) // 8 * 8, 512)
        assert k.get("vgpr", 0) + k.get("agpr", 0) <= budget, (k["demangled"], k.get("vgpr"), k.get("agpr"), k["wg"])
