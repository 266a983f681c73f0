"""Batched Controller plugin protocol: user-defined part controllers written against [B, ...] device tensors.

The reference's plugin boundary is `Controller` (controllers/parts/controller.py:35-44, 140-147): a part controller is constructed with the sim, its
joint index tables and its actuator range, `set_goal(action)` is called once per control step, `run_controller()` once per substep BETWEEN
`sim.step1()` and `sim.step2()` (environments/base.py:494-504), and what it returns is clipped into `sim.data.ctrl` (robots/fixed_base_robot.py:143-153).
The built-in types (OSC_POSE / OSC_POSITION / JOINT_POSITION / JOINT_TORQUE / JOINT_VELOCITY / GRIP) run inside the fused kernel; a controller the
kernel does not know could so far only run through the B = 1 shim.  This module is the batched form of the same contract:

    state = BatchState(batch)                                    # device tensors of all envs, valid between step1 and step2
    class MyController(BatchedController):
        def set_goal(self, action): ...                          # action: [B, control_dim] in [-1, 1]
        def run_controller(self): return torques                 # [B, n_joints], un-clipped (the caller clips, as the reference's robot does)
    env = HostControlledEnv(task_batch, [Part(MyController(state, ...), action_slice, actuator_ids)])
    env.step(actions)                                            # 25 x { rsim_step1; controllers on [B, ...] tensors; rsim_step2 }, obs / reward / horizon as rsim_control_step

Nothing here leaves the device: the controllers are torch code on tensors that alias the backend's buffers, one evaluation per substep for the
whole batch (50 kernel launches + the controllers' torch ops per control step, against ONE launch for the built-in types: this is the general
path, not the fast one).  `TorchJointTorqueController` and `TorchGripController` re-express the reference's JointTorqueController
(controllers/parts/generic/joint_tor.py:111-167) and SimpleGripController (controllers/parts/gripper/simple_grip.py:110-186) in this protocol;
tests/test_controllers_plugin.py checks them against the in-kernel versions step for step.
"""
from __future__ import annotations

import numpy as np


class BatchState:
    """Device views of a HipBatch, as robosuite's `sim.data` / `sim.model` would show them for one env (utils/binding_utils.py MjData accessors)."""

    def __init__(self, batch):
        import torch

        self.batch, self.flat = batch, batch.model.flat
        self.B, self.device = batch.B, torch.device("cuda", batch.device)
        for k in ("qpos", "qvel", "ctrl", "qM", "qfrc_bias", "qacc", "xpos", "xquat", "cdof", "rootcom", "time"):
            setattr(self, k, batch.tensor(k))
        m = self.flat
        self._site_body = torch.as_tensor(np.asarray(m.site_bodyid), device=self.device, dtype=torch.long)
        self._site_pos = torch.as_tensor(np.asarray(m.site_pos, dtype=np.float32), device=self.device)
        self._site_quat = torch.as_tensor(np.asarray(m.site_quat, dtype=np.float32), device=self.device)
        self._dofmask = {}

    @property
    def actuator_ctrlrange(self):
        return np.asarray(self.flat.actuator_ctrlrange, dtype=np.float32)

    @staticmethod
    def quat2mat(q):
        """wxyz quaternions [..., 4] -> rotation matrices [..., 3, 3]."""
        import torch

        w, x, y, z = q.unbind(-1)
        return torch.stack([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), w * w - x * x + y * y - z * z,
                            2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z], dim=-1).reshape(q.shape[:-1] + (3, 3))

    def site_pose(self, site: int):
        """(sim.data.site_xpos[site], site_xmat[site]) for every env: [B, 3], [B, 3, 3]."""
        import torch

        b = int(self._site_body[site])
        R = self.quat2mat(self.xquat[:, b])
        pos = self.xpos[:, b] + torch.einsum("bij,j->bi", R, self._site_pos[site])
        return pos, torch.einsum("bij,jk->bik", R, self.quat2mat(self._site_quat[site]))

    def body_dofs(self, body: int):
        """bool [nv]: the dofs that move `body` (its own joints and those of its ancestors)."""
        import torch

        if body not in self._dofmask:
            m = self.flat
            mask = np.zeros(m.nv, dtype=bool)
            b = body
            while b > 0:
                a, n = int(m.body_dofadr[b]), int(m.body_dofnum[b])
                if n > 0:
                    mask[a:a + n] = True
                b = int(m.body_parentid[b])
            self._dofmask[body] = torch.as_tensor(mask, device=self.device)
        return self._dofmask[body]

    def site_jacobian(self, site: int):
        """mj_jacSite for every env (binding_utils.py:826-851): jacp, jacr [B, 3, nv], from the motion axes about the tree's centre of mass that the
        position stage leaves in RSIM_CDOF / RSIM_ROOTCOM (the same formula as rsim_jac_site)."""
        import torch

        body = int(self._site_body[site])
        pos, _ = self.site_pose(site)
        off = pos - self.rootcom[:, body]                                  # [B, 3]
        ang, lin = self.cdof[:, :, 0:3], self.cdof[:, :, 3:6]              # [B, nv, 3]
        mask = self.body_dofs(body).to(self.cdof.dtype)[None, :, None]
        jacr = (ang * mask).transpose(1, 2)
        jacp = ((lin + torch.cross(ang, off[:, None, :].expand_as(ang), dim=-1)) * mask).transpose(1, 2)
        return jacp, jacr


class BatchedController:
    """Plugin base class: the reference's `Controller` contract (controllers/parts/controller.py) on [B, ...] tensors.

    joint_indexes: dict(joints=[...], qpos=[...], qvel=[...]) as the reference passes it; actuator_range: (low [n], high [n])."""

    name = "BATCHED"

    def __init__(self, state: BatchState, joint_indexes: dict, actuator_range, part_name="right", naming_prefix="robot0_"):
        import torch

        self.state, self.part_name, self.naming_prefix = state, part_name, naming_prefix
        self.qpos_index = torch.as_tensor(np.asarray(joint_indexes["qpos"]), device=state.device, dtype=torch.long)
        self.qvel_index = torch.as_tensor(np.asarray(joint_indexes["qvel"]), device=state.device, dtype=torch.long)
        self.joint_dim = len(joint_indexes["qvel"])
        self.actuator_min = torch.as_tensor(np.asarray(actuator_range[0], dtype=np.float32), device=state.device)
        self.actuator_max = torch.as_tensor(np.asarray(actuator_range[1], dtype=np.float32), device=state.device)
        self.control_dim = self.joint_dim
        self.input_min = self.input_max = self.output_min = self.output_max = None

    # controller.py:199-232 (the batched state is always current between step1 and step2: nothing to cache)
    @property
    def joint_pos(self):
        return self.state.qpos[:, self.qpos_index]

    @property
    def joint_vel(self):
        return self.state.qvel[:, self.qvel_index]

    @property
    def torque_compensation(self):
        """controller.py:303-311: gravity + Coriolis compensation = qfrc_bias of the part's dofs."""
        return self.state.qfrc_bias[:, self.qvel_index]

    @property
    def mass_matrix(self):
        """controller.py:226-232: the part's block of the dense mass matrix."""
        return self.state.qM[:, self.qvel_index][:, :, self.qvel_index]

    def scale_action(self, action):
        """controller.py:149-168: clip to [input_min, input_max], affine map onto [output_min, output_max]."""
        import torch

        t = lambda v: torch.as_tensor(np.broadcast_to(np.asarray(v, dtype=np.float32), (self.control_dim,)).copy(), device=self.state.device)   # noqa: E731
        imin, imax, omin, omax = t(self.input_min), t(self.input_max), t(self.output_min), t(self.output_max)
        scale = (omax - omin).abs() / (imax - imin).abs()
        a = torch.minimum(torch.maximum(action, imin), imax)
        return (a - 0.5 * (imax + imin)) * scale + 0.5 * (omax + omin)

    def clip_torques(self, torques):
        """controller.py:264-274."""
        import torch

        return torch.minimum(torch.maximum(torques, self.actuator_min), self.actuator_max)

    def reset_goal(self, mask=None):
        """Fresh controller state for the envs in `mask` (bool [B]; None = all): the reference constructs new controller objects on every reset
        (robots/robot.py:271)."""

    def set_goal(self, action):
        raise NotImplementedError

    def run_controller(self):
        raise NotImplementedError


class TorchJointTorqueController(BatchedController):
    """JointTorqueController (controllers/parts/generic/joint_tor.py:111-167, no interpolator): goal_torque = clip(scale_action(action), torque_limits),
    torques = goal_torque + torque_compensation."""

    name = "JOINT_TORQUE"

    def __init__(self, state, joint_indexes, actuator_range, input_max=1, input_min=-1, output_max=0.05, output_min=-0.05, torque_limits=None, **kw):
        import torch

        super().__init__(state, joint_indexes, actuator_range, **kw)
        self.input_max, self.input_min, self.output_max, self.output_min = input_max, input_min, output_max, output_min
        lim = torque_limits if torque_limits is not None else actuator_range
        self.torque_lo = torch.as_tensor(np.asarray(lim[0], dtype=np.float32), device=state.device)
        self.torque_hi = torch.as_tensor(np.asarray(lim[1], dtype=np.float32), device=state.device)
        self.goal_torque = torch.zeros(state.B, self.joint_dim, device=state.device)

    def reset_goal(self, mask=None):
        import torch

        self.goal_torque = torch.zeros_like(self.goal_torque) if mask is None else torch.where(mask[:, None], torch.zeros_like(self.goal_torque), self.goal_torque)   # no host sync

    def set_goal(self, action):
        import torch

        self.goal_torque = torch.minimum(torch.maximum(self.scale_action(action), self.torque_lo), self.torque_hi)

    def run_controller(self):
        return self.goal_torque + self.torque_compensation


class TorchJointPositionController(BatchedController):
    """JointPositionController (controllers/parts/generic/joint_pos.py:184-266, fixed impedance, delta inputs, no interpolator): goal_qpos = joint_pos +
    scale_action(action), clipped to `qpos_limits` if given; torques = M_part (kp (goal - q) - kd qd) + torque_compensation."""

    name = "JOINT_POSITION"

    def __init__(self, state, joint_indexes, actuator_range, input_max=1, input_min=-1, output_max=0.05, output_min=-0.05, kp=50.0, damping_ratio=1.0,
                 qpos_limits=None, use_torque_compensation=True, **kw):
        import torch

        super().__init__(state, joint_indexes, actuator_range, **kw)
        self.input_max, self.input_min, self.output_max, self.output_min = input_max, input_min, output_max, output_min
        t = lambda v: torch.as_tensor(np.broadcast_to(np.asarray(v, dtype=np.float32), (self.joint_dim,)).copy(), device=state.device)   # noqa: E731
        self.kp = t(kp)
        self.kd = 2.0 * torch.sqrt(self.kp) * t(damping_ratio)
        self.limits = None if qpos_limits is None else (t(qpos_limits[0]), t(qpos_limits[1]))
        self.use_torque_compensation = bool(use_torque_compensation)
        self.goal_qpos = torch.zeros(state.B, self.joint_dim, device=state.device)

    def reset_goal(self, mask=None):
        import torch

        self.goal_qpos = self.joint_pos.clone() if mask is None else torch.where(mask[:, None], self.joint_pos, self.goal_qpos)

    def set_goal(self, action):
        import torch

        g = self.joint_pos + self.scale_action(action)
        self.goal_qpos = g if self.limits is None else torch.minimum(torch.maximum(g, self.limits[0]), self.limits[1])

    def run_controller(self):
        import torch

        want = (self.goal_qpos - self.joint_pos) * self.kp - self.joint_vel * self.kd
        if not self.use_torque_compensation:
            return want
        return torch.einsum("bij,bj->bi", self.mass_matrix, want) + self.torque_compensation


class TorchOSCController(BatchedController):
    """OperationalSpaceController (controllers/parts/arm/osc.py) as a batched plugin: OSC_POSE / OSC_POSITION with fixed impedance, delta inputs in the
    robot-base frame, goals updated from the achieved pose -- the configuration of the reference's default robot configs
    (controllers/config/default/parts/osc_pose.json) and of the in-kernel controller (rsim_step.hip ctrl_run_osc).

      set_goal        osc.py:225-288, 306-401: scaled delta -> goal position / orientation in the base frame
      run_controller  osc.py:403-495 with utils/control_utils.py:43-140: errors -> desired force / torque -> Lambda (pseudo-inverses of
                      J M^-1 J^T and of its position / orientation blocks) -> J^T wrench + torque compensation + nullspace posture torques
    eef_site / base_site: model site ids of `{gripper}grip_site` and `{robot}{part}_center` (controller.py:88-100)."""

    name = "OSC_POSE"

    def __init__(self, state, joint_indexes, actuator_range, eef_site, base_site, kp=150.0, damping_ratio=1.0, input_max=1, input_min=-1,
                 output_max=(0.05, 0.05, 0.05, 0.5, 0.5, 0.5), output_min=(-0.05, -0.05, -0.05, -0.5, -0.5, -0.5), uncouple_pos_ori=True,
                 control_ori=True, nullspace_kp=10.0, **kw):
        import torch

        super().__init__(state, joint_indexes, actuator_range, **kw)
        self.eef_site, self.base_site, self.use_ori, self.uncoupling = int(eef_site), int(base_site), bool(control_ori), bool(uncouple_pos_ori)
        self.control_dim = 6 if self.use_ori else 3
        t = lambda v, n: torch.as_tensor(np.broadcast_to(np.asarray(v, dtype=np.float32), (n,)).copy(), device=state.device)   # noqa: E731
        self.kp = t(kp, 6)
        self.kd = 2.0 * torch.sqrt(self.kp) * t(damping_ratio, 6)
        self.input_max, self.input_min = np.broadcast_to(np.asarray(input_max, dtype=np.float32), (self.control_dim,)), np.broadcast_to(np.asarray(input_min, dtype=np.float32), (self.control_dim,))
        self.output_max, self.output_min = np.asarray(output_max, dtype=np.float32)[:self.control_dim], np.asarray(output_min, dtype=np.float32)[:self.control_dim]
        self.nullspace_kp = float(nullspace_kp)
        self.goal_pos = torch.zeros(state.B, 3, device=state.device)
        self.goal_ori = torch.eye(3, device=state.device).repeat(state.B, 1, 1)
        self.initial_joint = torch.zeros(state.B, self.joint_dim, device=state.device)

    # ---- what Controller.update() reads (controller.py:170-232), for every env -------------------
    def frames(self):
        """(eef position, eef rotation, base position, base rotation)"""
        ep, eR = self.state.site_pose(self.eef_site)
        op, oR = self.state.site_pose(self.base_site)
        return ep, eR, op, oR

    def jacobians(self):
        """(J_full [B, 6, n] over the part's dofs, eef velocity [B, 6], base velocity [B, 6])"""
        import torch

        jp, jr = self.state.site_jacobian(self.eef_site)
        bp, br = self.state.site_jacobian(self.base_site)
        qv = self.state.qvel
        ev = torch.cat([torch.einsum("bij,bj->bi", jp, qv), torch.einsum("bij,bj->bi", jr, qv)], dim=1)
        bv = torch.cat([torch.einsum("bij,bj->bi", bp, qv), torch.einsum("bij,bj->bi", br, qv)], dim=1)
        return torch.cat([jp[:, :, self.qvel_index], jr[:, :, self.qvel_index]], dim=1), ev, bv

    def reset_goal(self, mask=None):
        """Controller construction at (re)set (robots/robot.py:271, controller.py:128-130, osc.py:520-532): initial_joint = joint positions, goal = current pose."""
        import torch

        ep, eR, op, oR = self.frames()
        m = torch.ones(self.state.B, dtype=torch.bool, device=self.state.device) if mask is None else mask
        self.initial_joint = torch.where(m[:, None], self.joint_pos, self.initial_joint)
        self.goal_pos = torch.where(m[:, None], ep, self.goal_pos)
        self.goal_ori = torch.where(m[:, None, None], eR, self.goal_ori)

    def set_goal(self, action):
        import torch

        d = self.scale_action(action)
        ep, eR, op, oR = self.frames()
        self.goal_pos = torch.einsum("bji,bj->bi", oR, ep - op) + d[:, :3]                      # world_to_origin_frame(ref_pos) + delta (osc.py:306-345)
        cur = torch.einsum("bji,bjk->bik", oR, eR)                                              # eef orientation in the base frame
        if self.use_ori:
            ang = torch.linalg.norm(d[:, 3:6], dim=1, keepdim=True)                             # axis-angle delta -> rotation (osc.py:347-401)
            ax = d[:, 3:6] / torch.where(ang > 0, ang, torch.ones_like(ang))
            w, v = torch.cos(0.5 * ang), ax * torch.sin(0.5 * ang)
            w = torch.where(ang > 0, w, torch.ones_like(w))
            Rerr = BatchState.quat2mat(torch.cat([w, v], dim=1))
            self.goal_ori = torch.einsum("bij,bjk->bik", Rerr, cur)
        else:
            self.goal_ori = cur

    def torques_from(self, ep, eR, ev, op, oR, bv, J, M, bias, q, qd):
        """The torque law on explicit inputs ([B, ...] tensors): run_controller() gathers them from the batch state; tests feed recorded ones."""
        import torch

        # float64 inside: J M^-1 J^T of a 7-dof arm reaches condition numbers of 1e5-1e6 (tests/golden/lift_panda_singular), which a float32 inverse + pseudo-
        # inverse turns into per-cent errors of the nullspace term; the fused kernel avoids that with Cholesky solves, a plugin can simply afford doubles
        out_dtype = J.dtype
        ep, eR, ev, op, oR, bv, J, M, bias, q, qd = (x.double() for x in (ep, eR, ev, op, oR, bv, J, M, bias, q, qd))
        goal_pos, goal_ori, q0, kp, kd = self.goal_pos.double(), self.goal_ori.double(), self.initial_joint.double(), self.kp.double(), self.kd.double()
        dpos = op + torch.einsum("bij,bj->bi", oR, goal_pos)
        dori = torch.einsum("bij,bjk->bik", oR, goal_ori)
        # orientation_error (control_utils.py:85-112): half the sum of the cross products of corresponding axes
        oerr = 0.5 * sum(torch.cross(eR[:, :, c], dori[:, :, c], dim=1) for c in range(3))
        F = (dpos - ep) * kp[:3] - (ev[:, :3] - bv[:, :3]) * kd[:3]
        T = oerr * kp[3:] - (ev[:, 3:] - bv[:, 3:]) * kd[3:]
        Minv = torch.linalg.inv(M)
        MiJT = torch.einsum("bij,bkj->bik", Minv, J)                                            # M^-1 J^T  [B, n, 6]
        lfi = torch.einsum("bij,bjk->bik", J, MiJT)                                             # J M^-1 J^T
        lf = torch.linalg.pinv(lfi)
        if self.uncoupling:
            wrench = torch.cat([torch.einsum("bij,bj->bi", torch.linalg.pinv(lfi[:, :3, :3]), F), torch.einsum("bij,bj->bi", torch.linalg.pinv(lfi[:, 3:, 3:]), T)], dim=1)
        else:
            wrench = torch.einsum("bij,bj->bi", lf, torch.cat([F, T], dim=1))
        tau = torch.einsum("bji,bj->bi", J, wrench) + bias
        # nullspace_torques (control_utils.py:115-140): N^T M (kp (q0 - q) - 2 sqrt(kp) qd), N = I - Jbar J, Jbar = M^-1 J^T Lambda
        N = torch.eye(J.shape[2], device=J.device, dtype=J.dtype)[None] - torch.einsum("bij,bjk->bik", torch.einsum("bij,bjk->bik", MiJT, lf), J)
        pose = torch.einsum("bij,bj->bi", M, self.nullspace_kp * (q0 - q) - 2.0 * float(np.sqrt(self.nullspace_kp)) * qd)
        return (tau + torch.einsum("bji,bj->bi", N, pose)).to(out_dtype)

    def run_controller(self):
        ep, eR, op, oR = self.frames()
        J, ev, bv = self.jacobians()
        return self.torques_from(ep, eR, ev, op, oR, bv, J, self.mass_matrix, self.torque_compensation, self.joint_pos, self.joint_vel)


class TorchGripController(BatchedController):
    """PandaGripper.format_action + SimpleGripController (models/grippers/panda_gripper.py:43-58, controllers/parts/gripper/simple_grip.py:110-186): the
    one-dimensional action moves an internal state in [-1, 1] by `speed * sign(action)` per control step; the actuators' position targets are
    bias + weight * (sign_i * state) over their ctrl ranges."""

    name = "GRIP"

    def __init__(self, state, joint_indexes, actuator_range, signs=(-1.0, 1.0), speed=0.2, **kw):
        import torch

        super().__init__(state, joint_indexes, actuator_range, **kw)
        self.control_dim = 1
        self.signs = torch.as_tensor(np.asarray(signs, dtype=np.float32), device=state.device)
        self.speed = float(speed)
        self.current_action = torch.zeros(state.B, len(signs), device=state.device)

    def reset_goal(self, mask=None):
        import torch

        self.current_action = torch.zeros_like(self.current_action) if mask is None else torch.where(mask[:, None], torch.zeros_like(self.current_action), self.current_action)

    def set_goal(self, action):
        import torch

        self.current_action = torch.clamp(self.current_action + self.signs[None] * self.speed * torch.sign(action[:, :1]), -1.0, 1.0)

    def run_controller(self):
        bias, weight = 0.5 * (self.actuator_max + self.actuator_min), 0.5 * (self.actuator_max - self.actuator_min)
        return bias + weight * self.current_action


class Part:
    """One part of a composite controller (composite_controller.py:70-116): the controller, its slice of the flat action, its actuators."""

    def __init__(self, controller: BatchedController, action_slice: slice, actuator_ids):
        self.controller, self.action_slice, self.actuator_ids = controller, action_slice, list(actuator_ids)


class HostControlledEnv:
    """MujocoEnv.step's substep loop (environments/base.py:494-504) for a batch whose controllers are BatchedController plugins: per substep one
    `rsim_step1` of all envs, the parts' `run_controller()` on device tensors, the clip into `ctrl` (fixed_base_robot.py:143-153), one `rsim_step2`;
    the last substep's step2 also produces the observation record / reward / success / horizon bookkeeping exactly as rsim_control_step does
    (rsim_step2_last).  `task` is one of the task batches (lift.LiftBatch, ...); its in-kernel controller configuration is not used."""

    def __init__(self, task, parts, n_sub: int = 25):
        import torch

        self.task, self.batch, self.parts, self.n_sub = task, task.batch, list(parts), n_sub
        self.state = parts[0].controller.state
        cr = self.state.actuator_ctrlrange
        self._ids = [torch.as_tensor(p.actuator_ids, device=self.state.device, dtype=torch.long) for p in self.parts]
        self._lo = [torch.as_tensor(cr[p.actuator_ids, 0], device=self.state.device) for p in self.parts]
        self._hi = [torch.as_tensor(cr[p.actuator_ids, 1], device=self.state.device) for p in self.parts]
        self.action_dim = max(p.action_slice.stop for p in self.parts)
        self._restarted = None

    def reset(self, block: int = 0):
        self.task.reset(block)
        self._restarted = None
        for p in self.parts:
            p.controller.reset_goal()

    def step(self, actions):
        import torch

        b, ctrl = self.batch, self.state.ctrl
        for i in range(self.n_sub):
            b.step1()
            for p, ids, lo, hi in zip(self.parts, self._ids, self._lo, self._hi):
                if i == 0:
                    if self._restarted is not None:
                        # envs restarted on the device by the previous step get fresh controllers (robots/robot.py:271) -- now, after the first step1
                        # of their new episode: only then do the frames a task-space controller resets its goal from belong to the new state
                        p.controller.reset_goal(self._restarted)
                    p.controller.set_goal(actions[:, p.action_slice])          # policy step: composite_controller.set_goal (composite_controller.py:97-103)
                ctrl[:, ids] = torch.minimum(torch.maximum(p.controller.run_controller(), lo), hi)
            if i < self.n_sub - 1:
                b.step2()
            else:
                b.step2_last()
        self._restarted = b.tensor("done").to(torch.bool).clone()
        self.task._bank_tick()
        return self.task.obs(), self.task.reward(), b.tensor("done"), {"success": self.task.success()}
