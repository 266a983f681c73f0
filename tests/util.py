"""Shared helpers for the test-suite: golden fixtures, oracle and HIP object construction."""
import json
import os

import numpy as np

from robosuite_amd import mjcf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = ("seed0_gentle", "seed1_full")


def load_golden(tag):
    g = np.load(os.path.join(GOLD, f"lift_panda_{tag}.npz"))
    cfg = json.load(open(os.path.join(GOLD, f"lift_panda_{tag}.cfg.json")))
    flat = mjcf.load_model(os.path.join(GOLD, f"lift_panda_{tag}.rsim"))
    return g, cfg, flat


def make_oracle(flat, cfg=None):
    from oracle.oracle import OracleController, OracleData, OracleModel

    om = OracleModel(mjcf.to_blob(flat))
    d = OracleData(om)
    c = OracleController(cfg) if cfg is not None else None
    return om, d, c


def make_hip(flat, cfg=None, B=1, per_env=False):
    from robosuite_amd.backend import HipBatch, HipModel

    hm = HipModel(flat)
    if cfg is not None:
        hm.set_controller(cfg)
    return hm, HipBatch(hm, B, 0, per_env)
