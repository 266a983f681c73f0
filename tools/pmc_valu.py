"""VALU wave-instructions per env-substep of the fused kernel from a rocprofv3 PMC pass (SQ_INSTS_VALU), on the bench workload.
Writes profiles/valu_count.json, keyed by the sha of the library build it was measured on; bench.py reports roofline.issue from it
only when that sha matches the library it runs.  Usage (on the GPU box): python tools/pmc_valu.py gpurun_out/<tag>.sq1 [last_n]"""
import csv, glob, hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import backend  # noqa: E402
from tools.kernel_resources import config_code_sha16, wide_code_sha16  # noqa: E402


def per_dispatch(d, counter, pat="k_step<"):   # the control-step kernel, not k_step_dbg
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals[int(r["Dispatch_Id"])] = vals.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return np.array([vals[k] for k in sorted(vals)])


if __name__ == "__main__":
    d = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    CONFIG = os.environ.get("RSIM_CONFIG", "lift")       # bench.py --config: the file becomes profiles/valu_count_<config>.json
    SFX = "" if CONFIG == "lift" else "_" + CONFIG
    B, n_sub = int(os.environ.get("RSIM_B", {"lift": 4096, "stack": 4096, "peg": 2048, "pickplace": 8192}[CONFIG])), 25
    v = per_dispatch(d, "SQ_INSTS_VALU")[-last:]        # the timed control steps are the last dispatches of the run
    out = {"valu_per_env_substep": float(np.mean(v) / (B * n_sub)), "dispatches": int(len(v)), "envs": B,
           "lib_sha16": hashlib.sha256(open(backend.LIB_PATH, "rb").read()).hexdigest()[:16], "code_sha16": config_code_sha16(backend.LIB_PATH, CONFIG), "wide_code_sha16": wide_code_sha16(backend.LIB_PATH, CONFIG), "tuning_sha16": backend.tuning_sha16(),
           "config": CONFIG,
           "note": "rocprofv3 --pmc SQ_INSTS_VALU, mean over the last control-step dispatches of `bench.py --steps 4 --warmup 1` (steady-state episode phase)"}
    for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"):
        x = per_dispatch(d, c)[-last:]
        if len(x):
            out[c.lower() + "_per_env_substep"] = float(np.mean(x) / (B * n_sub))
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_count" + SFX + ".json"), "w"), indent=1)
    print(out)
