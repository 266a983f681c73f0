"""B = 1 compatibility backend for robosuite_amd.shim on the HIP C-ABI: robosuite's own Python (MjSim, Controller.update, env.step) runs
unchanged, one environment, with float64 numpy mirrors of the device state (SURVEY.md section 8b "B=1 compat: host mirrors").

Every forward/step call uploads the arrays robosuite may have written through its views (qpos, qvel, ctrl, qacc_warmstart, and model
edits via sync_model), launches the same fused kernel the batched path uses (phase flags select step1 / step2 / forward), and refreshes
the mirrors IN PLACE, because robosuite holds on to the numpy views (`sim.data.ctrl[idx] = ...`, fixed_base_robot.py:153).
This is the plumbing path (BASELINE config 1), not the fast path; there is still no CPU fallback for the arithmetic.
"""
from __future__ import annotations

import numpy as np

from . import mjcf
from .backend import HipBatch, HipModel

_PARAM_FIELDS = ("body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "body_invweight0", "body_subtreemass",
                 "jnt_pos", "jnt_axis", "jnt_range", "jnt_margin", "jnt_solref", "jnt_solimp", "qpos0", "dof_armature", "dof_damping",
                 "dof_frictionloss", "dof_solref", "dof_solimp", "dof_invweight0", "geom_size", "geom_pos", "geom_quat", "geom_friction",
                 "geom_solref", "geom_solimp", "geom_solmix", "geom_margin", "geom_gap", "geom_rbound", "site_pos", "site_quat",
                 "actuator_gear", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange", "actuator_forcerange")


class HipShimBackend:
    def __init__(self, flat: mjcf.FlatModel, device: int = 0):
        self.flat = flat
        self.model = HipModel(flat)
        self.batch = HipBatch(self.model, 1, device, per_env_params=False)
        m = flat
        f64 = lambda n: np.zeros(n, dtype=np.float64)
        self.d = {
            "qpos": np.array(m.qpos0, dtype=np.float64).ravel().copy(), "qvel": f64(m.nv), "qacc": f64(m.nv), "qacc_warmstart": f64(m.nv),
            "ctrl": f64(m.nu), "qfrc_applied": f64(m.nv), "mocap_pos": f64(3 * int(m.arrays["nmocap"][0]) if "nmocap" in m.arrays else 0),
            "mocap_quat": f64(4 * int(m.arrays["nmocap"][0]) if "nmocap" in m.arrays else 0), "xpos": f64(3 * m.nbody), "xquat": f64(4 * m.nbody),
            "xmat": f64(9 * m.nbody), "xipos": f64(3 * m.nbody), "ximat": f64(9 * m.nbody), "geom_xpos": f64(3 * m.ngeom), "geom_xmat": f64(9 * m.ngeom),
            "site_xpos": f64(3 * m.nsite), "site_xmat": f64(9 * m.nsite), "subtree_com": f64(3 * m.nbody), "qM": f64(m.nv * m.nv), "qfrc_bias": f64(m.nv),
            "qfrc_passive": f64(m.nv), "qfrc_actuator": f64(m.nv), "qfrc_constraint": f64(m.nv), "actuator_force": f64(m.nu),
            "sensordata": f64(int(m.arrays["sensor_dim"].sum()) if m.nsensor else 0), "time": f64(1),
        }
        self._ncon = 0
        self._contacts = []

    # ---- shim protocol -------------------------------------------------------------------------
    def model_array(self, name):
        return None  # the FlatModel's own arrays are the host copy; sync_model() pushes edits

    def sync_model(self):
        for f in _PARAM_FIELDS:
            if f in self.flat.arrays:
                self.batch.param_set(f, np.asarray(self.flat.arrays[f], dtype=np.float64).reshape(1, -1))
        opt = np.concatenate([[self.flat.timestep], np.asarray(self.flat.arrays["gravity"]).ravel(), [self.flat.density, self.flat.viscosity, self.flat.impratio],
                              np.asarray(self.flat.arrays["wind"]).ravel()])
        self.batch.param_set("opt", opt.reshape(1, -1))

    def data_array(self, name):
        return self.d[name]

    def _push(self):
        b = self.batch
        for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "qfrc_applied"):
            b.set(k, self.d[k][None])
        b.set("time", self.d["time"])

    def _pull(self, stepped, acc=True):
        b, d, m = self.batch, self.d, self.flat
        if acc and len(d["sensordata"]):      # acceleration-stage sensors (mj_sensorAcc): force / torque at the gripper's ft_frame site
            d["sensordata"][:] = b.get("sensordata")[0]
        if stepped:
            for k in ("qpos", "qvel", "qacc_warmstart"):
                d[k][:] = b.get(k)[0]
            d["time"][:] = b.get("time")
        for k in ("xpos", "xquat", "qM", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qfrc_constraint", "qacc"):
            d[k][:] = b.get(k)[0].ravel()
        # frames the kernel keeps in LDS only: rebuild on the host from body frames (same formulas as mjcf.kinematics_np)
        xq = d["xquat"].reshape(-1, 4)
        xp = d["xpos"].reshape(-1, 3)
        R = np.stack([mjcf.quat2mat(q) for q in xq])
        d["xmat"][:] = R.reshape(-1)
        d["xipos"][:] = (xp + np.einsum("bij,bj->bi", R, m.body_ipos)).ravel()
        d["ximat"][:] = np.stack([mjcf.quat2mat(mjcf.quat_mul(xq[i], m.body_iquat[i])) for i in range(m.nbody)]).reshape(-1)
        gb, sb = m.geom_bodyid, m.site_bodyid
        d["geom_xpos"][:] = (xp[gb] + np.einsum("gij,gj->gi", R[gb], m.geom_pos)).ravel()
        d["geom_xmat"][:] = np.stack([mjcf.quat2mat(mjcf.quat_mul(xq[gb[i]], m.geom_quat[i])) for i in range(m.ngeom)]).reshape(-1)
        if m.nsite:
            d["site_xpos"][:] = (xp[sb] + np.einsum("sij,sj->si", R[sb], m.site_pos)).ravel()
            d["site_xmat"][:] = np.stack([mjcf.quat2mat(mjcf.quat_mul(xq[sb[i]], m.site_quat[i])) for i in range(m.nsite)]).reshape(-1)
        d["subtree_com"][:] = b.get("rootcom")[0].ravel()
        self._ncon = int(b.get("ncon")[0])
        self._contacts = b.contacts(0)

    def forward(self):
        self._push(); self.batch.forward(); self._pull(False)

    def step1(self):
        self._push(); self.batch.step1(); self._pull(False, acc=False)

    def step2(self):
        self._push(); self.batch.step2(); self._pull(True)

    def step(self):
        self._push(); self.batch.step(); self._pull(True)

    def reset(self):
        self.batch.reset()
        d = self.d
        d["qpos"][:] = np.asarray(self.flat.qpos0).ravel()
        for k in ("qvel", "qacc", "qacc_warmstart", "ctrl", "qfrc_applied", "time"):
            d[k][:] = 0

    def jac(self, kind, idx):
        if kind == "site":
            return self.batch.jac_site(0, idx)
        if kind == "body":
            return self.batch.jac_body(0, idx)
        raise NotImplementedError("geom Jacobians are not on robosuite's hot path (binding_utils.py:759-773)")

    def full_M(self):
        return self.d["qM"].reshape(self.flat.nv, self.flat.nv).copy()

    @property
    def ncon(self):
        return self._ncon

    def contacts(self):
        return self._contacts

    @property
    def nefc(self):
        return int(self.batch.get("nefc")[0])

    def efc_array(self, name):
        """Constraint rows of the last evaluation (shim.MjData.efc_*).  The kernel exports the forces; aref / R / type live in LDS only."""
        if name != "efc_force":
            raise KeyError(f"the HIP backend does not export {name} (constraint forces only: RSIM_EFC_FORCE)")
        return self.batch.get("efc_force")[0][: self.nefc].astype(np.float64)
