"""HBM traffic per launch of the fused kernel from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes as the
MI355X guide prescribes).  Writes profiles/hbm_traffic.json, which bench.py reports as roofline.traffic.
Usage (on the GPU box): python tools/pmc_traffic.py gpurun_out/<tag>.hbm1 gpurun_out/<tag>.hbm2"""
import csv, glob, hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import backend  # noqa: E402
from tools.kernel_resources import config_code_sha16, wide_code_sha16  # noqa: E402
def per_dispatch(d, counter):
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_step<" in r["Kernel_Name"] and r["Counter_Name"] == counter:   # not k_step_dbg (forward / shim entries)
                vals[int(r["Dispatch_Id"])] = vals.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return np.array([vals[k] for k in sorted(vals)])
LAST = int(sys.argv[3]) if len(sys.argv) > 3 else 4
CONFIG = os.environ.get("RSIM_CONFIG", "lift")
SFX = "" if CONFIG == "lift" else "_" + CONFIG
rd, wr = per_dispatch(sys.argv[1], "FETCH_SIZE")[-LAST:], per_dispatch(sys.argv[2], "WRITE_SIZE")[-LAST:]
# the timed control steps are the last dispatches of the run (before them: forward(), controller reset, the untimed pre-roll)
out = {"bytes_per_launch": float((np.median(rd) + np.median(wr)) * 1024.0), "fetch_kb_median": float(np.median(rd)), "write_kb_median": float(np.median(wr)),
       "dispatches": int(len(rd)), "config": CONFIG, "lib_sha16": hashlib.sha256(open(backend.LIB_PATH, "rb").read()).hexdigest()[:16], "code_sha16": config_code_sha16(backend.LIB_PATH, CONFIG), "wide_code_sha16": wide_code_sha16(backend.LIB_PATH, CONFIG), "tuning_sha16": backend.tuning_sha16(), "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, KiB per dispatch; gfx950 FETCH_SIZE may under-report wide "
       "coalesced reads by 2x (MI355X guide) - this kernel issues dword loads; reads include the whole float-table FIELDS that carry per-env values "
       "(cube size / mass / inertia / inverse weights: 273 floats per env, of which 17 differ), the shared tables stay L2-resident; writes: the private segment (52 B per lane x 64 lanes x one wavefront per env in the Lift configuration since round 4; 124 B before), the per-step state, the narrow-phase warm-start records and the broadphase pair list"}
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic" + SFX + ".json"), "w"), indent=1)
print(out)
