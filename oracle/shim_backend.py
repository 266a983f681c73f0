"""Oracle-backed backend for robosuite_amd.shim (TEST INFRASTRUCTURE ONLY; see oracle/oracle.py)."""
import numpy as np

from robosuite_amd import mjcf

from .oracle import OracleData, OracleModel


class OracleBackend:
    def __init__(self, flat):
        self.flat = flat
        self.om = OracleModel(mjcf.to_blob(flat))
        self.d = OracleData(self.om)

    def model_array(self, name):
        try:
            return self.om.field(name)  # zero-copy view into the C model: edits take effect immediately
        except KeyError:
            return None

    def sync_model(self):
        pass

    def data_array(self, name):
        return self.d.field(name)

    def forward(self):
        self.d.forward()

    def step(self):
        self.d.step()

    def step1(self):
        self.d.step1()

    def step2(self):
        self.d.step2()

    def reset(self):
        self.d.reset()

    def jac(self, kind, idx):
        return self.d.jac(kind, idx)

    def full_M(self):
        return self.d.full_M()

    @property
    def ncon(self):
        return self.d.ncon

    def contacts(self):
        return self.d.contacts()

    @property
    def nefc(self):
        return self.d.nefc

    def efc_array(self, name):
        if name == "efc_type":
            return np.array(self.d.efc_types(), dtype=np.int32)
        return np.array(self.d.field(name)[: self.d.nefc])
