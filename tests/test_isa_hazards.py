"""CPU: static check of the compiled GEMM kernels for reads of registers whose inline-asm loads are still in flight.

csrc/gemm_x6.hip issues its operand loads from inline asm and counts `s_waitcnt vmcnt(N)` by hand (the compiler's own
bookkeeping turns prefetch distance 2 into 1).  hipcc does not know those registers are not ready: when it decides to keep a
loaded value elsewhere it may place the copy in front of the wait, and the kernel then multiplies garbage -- silently, and only
in the template instances where register allocation happens to do so (round 5 met it twice while adding the fp16 form's
range check).  scripts/isa_hazards.py walks every kernel's instruction stream over all branch edges with the queue of in-flight
asm loads as its state; this test cross-compiles the file exactly as the Makefile does and requires zero findings."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "pcrcg_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["gemm_x6.hip"])
def test_no_use_of_registers_with_loads_in_flight(tmp_path, src):
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + CSRC,
           "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    import isa_hazards
    kernels = isa_hazards.parse_kernels(open(out).read().splitlines())
    findings, checked = [], 0
    for name, ins in kernels.items():
        if any(a and t and t.startswith("global_load") for _, t, a in ins):
            checked += 1
            findings += isa_hazards.check_kernel(name, ins)
    assert checked >= 10, "the scan found no kernels with asm loads: has the file's structure changed?"
    assert not findings, "\n".join("%s #%d %s <- %s" % (n[:90], i, t, w) for n, i, t, w in findings[:10])


def test_the_scanner_sees_a_planted_hazard():
    """The checker on a hand-made stream: a copy of a load's destination in front of the wait is reported, the same copy behind
    the wait is not, and a loop's back edge carries the in-flight state."""
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    import isa_hazards
    bad = """_Zbad:
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[0:1], off
	;;#ASMEND
.LBB0_1:
	v_mov_b32_e32 v9, v5
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_f32_e32 v10, v4, v9
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[0:1], off
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	s_endpgm
"""
    good = bad.replace("\tv_mov_b32_e32 v9, v5\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n",
                       "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v9, v5\n")
    kb = isa_hazards.parse_kernels(bad.splitlines())
    kg = isa_hazards.parse_kernels(good.splitlines())
    assert len(isa_hazards.check_kernel("_Zbad", kb["_Zbad"])) >= 1
    assert isa_hazards.check_kernel("_Zbad", kg["_Zbad"]) == []
