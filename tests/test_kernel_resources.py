"""Design budgets of the compiled kernels, read from the code objects embedded in the in-tree library (no GPU needed).  The occupancy of the
fused control-step kernel is bound twice: by LDS (a whole number of environments per 160 KB) and by registers (one wavefront per SIMD above
256 VGPR + AGPR, two up to 256).  The Lift configuration is sized to EIGHT environments per CU: 20 KB of LDS and 256 registers, i.e. two
wavefronts per SIMD; growing either past its budget silently halves the occupancy -- this test makes it loud."""
import os

import pytest

from tools.kernel_resources import LLVM, kernels

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosuite_amd", "librsim_hip.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))), reason="needs the built library and llvm-readelf")

LDS_PER_CU = 160 * 1024
# k_step<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR>: (environments per CU by LDS, register budget VGPR + AGPR, private-segment bytes tolerated)
STEP_BUDGET = {"ILi32ELi16ELi16ELi24ELi16ELi16ELi64ELi192E": (8, 256, 32),    # round 6: 0 B at 234 registers with BOTH bodies (native + fused wide tier, 20 408 B of LDS) reading their arguments from the kernarg segment; 52 B in round 5 (round 3: 156, round 4 before the MachineLICM flags: 124): profiles/r04_y_ab_spills.txt
                "ILi32ELi16ELi32ELi24ELi16ELi32ELi64ELi192E": (8, 256, 32),   # Stack: J, M and the contact block in global memory, no LDS hull pool, two wavefronts per SIMD (round 5, sessions 12 / 13: 36.1 -> 19.8 KB, four -> eight envs per CU); round 6: native + fused wide body in 20 328 B, 249 registers, no scratch
               "ILi64ELi16ELi16ELi32ELi32ELi32ELi64ELi320E": (5, 256, 32),   # round 6: 31 892 B = 25 granules, 247 registers: five envs per CU (before: 32 084 B = 26 granules = four, whatever the registers)
                "ILi64ELi32ELi48ELi64ELi32ELi32ELi128ELi640E": (4, 512, 0),   # J and M in global memory (RSIM_JGLOBAL round 4: 74.8 -> 49.7 KB = 3; RSIM_MGLOBAL round 5: 40.3 KB = 4, one wavefront per SIMD)
               "ILi64ELi32ELi64ELi64ELi32ELi32ELi128ELi640E": (1, 512, 0),
               # the capacity tiers (round 4): above 64 x 48, above the Lift configuration, above the Stack configuration.  The 256-row tier (four rows per lane, J in
               # global memory since round 5: two per CU) spills in the polish's fp64 line search (424 B); it steps the few envs beyond 128 rows
               # round 6: the row staging holds one slot (64 rows) at a time -- the 256-row tier 65 972 -> 54 516 B (still two per CU, see below), the Stack-class tier's own unit 28 276 -> 23 124 B = six (granule)
               "ILi64ELi32ELi48ELi64ELi32ELi64ELi256ELi640E": (2, 512, 512),   # (54 516 B: 244 B over three per CU at the 512-B allocation granule)
                "ILi32ELi16ELi16ELi24ELi16ELi32ELi128ELi192E": (5, 512, 0), "ILi32ELi16ELi32ELi24ELi16ELi32ELi128ELi192E": (6, 512, 64)}   # Stack tier: J and M in global memory (50.1 -> 28.3 KB), native-style entry held to 256 registers (60 B)


def test_fused_kernel_configurations_keep_their_lds_and_register_budgets():
    ks = kernels(LIB)
    steps = {n: r for n, r in ks.items() if n.startswith("_Z6k_stepI")}
    assert len(steps) == 8
    for tag, (envs, regs, scratch) in STEP_BUDGET.items():
        (name,) = [n for n in steps if tag in n]
        r = steps[name]
        # LDS is allocated in granules of 1280 B (measured in round 6: tools/kernel_resources.py LDS_GRANULE): the budget is checked on the rounded size
        assert envs * (-(-r["lds"] // 1280) * 1280) <= LDS_PER_CU, (name, r)
        assert (envs + 1) * (-(-r["lds"] // 1280) * 1280) > LDS_PER_CU   # the table above states the real LDS occupancy (at the allocation granule), not a lower bound
        assert r["vgpr"] <= regs, (name, r)                            # vgpr_count of the code object = VGPR + AGPR; <= 256: two wavefronts per SIMD
        # (almost) no private segment: besides spills, a run-time index into the by-value DModel kernel argument -- or a select between two
        # pointers derived from it -- makes the compiler keep a 1.9 KB copy of DModel there and read every model scalar from that copy
        assert r["scratch"] <= scratch, (name, r)


def test_auxiliary_kernels_use_no_scratch():
    for name, r in kernels(LIB).items():
        if name.startswith("_Z11k_reset_obs") or name.startswith("_Z10k_step_dbg"):
            # (the Lift-class tier's own translation unit, 32 x 16 with 128 rows, is not on the Lift path any more -- round 6: that tier is a body of the native
            # kernel -- and its reset / debug kernels are never launched: the reset observation of every env comes from the native build)
            assert r["scratch"] <= (64 if "ELi256E" not in name else 1024) or "ILi32ELi16ELi16ELi24ELi16ELi32ELi128ELi192E" in name, (name, r)   # the reset-observation pass (a few envs per control step) and the B = 1 debug entries share the step body (256-row tier: see STEP_BUDGET)
        elif not (name.startswith("_Z6k_step") or name.startswith("_Z11k_step_list")):
            assert r["scratch"] == 0, (name, r)


def test_pmc_evidence_is_keyed_to_the_machine_code_of_its_configuration(tmp_path, monkeypatch):
    """profiles/valu_count*.json / hbm_traffic*.json carry the sha of the library they were measured on AND of the machine code (.text + kernel descriptors) of
    their configuration's code object.  bench.py accepts a figure when either matches the library it runs -- a build that changed another configuration runs the
    bit-identical kernel for this one -- and says "stale:<sha>" otherwise (never a silent null).  The committed evidence must be valid for the in-tree build."""
    import hashlib
    import json
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from tools.kernel_resources import CONFIG_TAG, config_code_sha16

    lib_sha = hashlib.sha256(open(LIB, "rb").read()).hexdigest()[:16]
    shas = {c: config_code_sha16(LIB, c) for c in CONFIG_TAG}
    assert all(shas.values()) and len(set(shas.values())) == len(shas)          # one code object per configuration, all different
    for cfg, sfx in (("lift", ""), ("stack", "_stack"), ("peg", "_peg"), ("pickplace", "_pickplace")):
        for stem, key in (("valu_count", "valu_per_env_substep"), ("hbm_traffic", "bytes_per_launch")):
            d = json.load(open(os.path.join(root, "profiles", f"{stem}{sfx}.json")))
            assert d["code_sha16"] == shas[cfg], (stem, cfg, "evidence of another kernel: re-run tools/gpu_session.sh <tag> pmc:" + cfg)
            v = bench.pmc_evidence(f"{stem}{sfx}.json", key, lib_sha, cfg)
            assert isinstance(v, float) and v > 0, (stem, cfg, v)
    # evidence of another kernel is said out loud
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    json.dump({"valu_per_env_substep": 1.0, "lib_sha16": "0123456789abcdef", "code_sha16": "fedcba9876543210"}, open(tmp_path / "profiles" / "valu_count.json", "w"))
    assert bench.pmc_evidence("valu_count.json", "valu_per_env_substep", lib_sha, "lift") == "stale:0123456789abcdef"
    assert bench.pmc_evidence("hbm_traffic.json", "bytes_per_launch", lib_sha, "lift") == "absent"
    # another library build: the figure stands only with the configuration's code object, its capacity tier's and the host-side dispatch / solver settings all equal
    # (round-5 advisor finding: a host-only change of polish passes or tier thresholds left code_sha16 equal)
    from robosuite_amd import backend
    from tools.kernel_resources import wide_code_sha16
    rec = {"valu_per_env_substep": 2.0, "lib_sha16": "0123456789abcdef", "code_sha16": shas["lift"], "wide_code_sha16": wide_code_sha16(LIB, "lift"), "tuning_sha16": backend.tuning_sha16()}
    json.dump(rec, open(tmp_path / "profiles" / "valu_count.json", "w"))
    assert bench.pmc_evidence("valu_count.json", "valu_per_env_substep", lib_sha, "lift") == 2.0
    json.dump(dict(rec, tuning_sha16="0000000000000000"), open(tmp_path / "profiles" / "valu_count.json", "w"))
    assert bench.pmc_evidence("valu_count.json", "valu_per_env_substep", lib_sha, "lift") == "stale:0123456789abcdef"
    monkeypatch.setenv("RSIM_NEWTON_REFINE", "3")            # an override in force is another setting than the one the evidence was taken under
    json.dump(rec, open(tmp_path / "profiles" / "valu_count.json", "w"))
    assert bench.pmc_evidence("valu_count.json", "valu_per_env_substep", lib_sha, "lift") == "stale:0123456789abcdef"
    monkeypatch.delenv("RSIM_NEWTON_REFINE")
    assert wide_code_sha16(LIB, "stack") not in (None, "same-object", shas["stack"]) and wide_code_sha16(LIB, "pickplace") not in (None, shas["pickplace"])
