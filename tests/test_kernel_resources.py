"""Design budgets of the compiled kernels, read from the code objects embedded in the in-tree library (no GPU needed).  The occupancy of the
fused control-step kernel is LDS-bound: one wavefront per SIMD up to four environments per CU, so every configuration is sized to a whole
number of environments per 160 KB of LDS.  A change that grows the per-env LDS object past its budget silently costs
25-33 % of the throughput; this test makes it loud."""
import os

import pytest

from tools.kernel_resources import LLVM, kernels

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosuite_amd", "librsim_hip.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))), reason="needs the built library and llvm-readelf")

LDS_PER_CU = 160 * 1024
# k_step<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR>: environments per CU the configuration is designed for
STEP_BUDGET = {"ILi32ELi16ELi16ELi24ELi16ELi16ELi64ELi192E": 4, "ILi32ELi16ELi32ELi24ELi16ELi32ELi64ELi192E": 3,
               "ILi64ELi16ELi16ELi32ELi32ELi32ELi64ELi320E": 3, "ILi64ELi32ELi64ELi64ELi32ELi32ELi128ELi640E": 1}


def test_fused_kernel_configurations_keep_their_lds_and_register_budgets():
    ks = kernels(LIB)
    steps = {n: r for n, r in ks.items() if n.startswith("_Z6k_step")}
    assert len(steps) == 4
    for tag, envs in STEP_BUDGET.items():
        (name,) = [n for n in steps if tag in n]
        r = steps[name]
        # LDS is allocated in granules (512 B assumed): the budget is checked on the rounded size
        assert envs * (-(-r["lds"] // 512) * 512) <= LDS_PER_CU, (name, r)
        assert (envs + 1) * r["lds"] > LDS_PER_CU or envs == 4          # the table above states the real occupancy, not a lower bound
        assert r["vgpr"] <= 512                                        # VGPR + AGPR of one wavefront per SIMD
        # no private segment: besides spills, a run-time index into the by-value DModel kernel argument makes the compiler keep a 1.9 KB
        # copy of it there and read every model scalar from that copy (the eight-tree configuration did, through m.dynroot[r])
        assert r["scratch"] == 0, (name, r)


def test_auxiliary_kernels_use_no_scratch():
    for name, r in kernels(LIB).items():
        if not name.startswith("_Z6k_step"):
            assert r["scratch"] == 0, (name, r)
