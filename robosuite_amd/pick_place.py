"""PickPlace / IIWA + Robotiq140 (BASELINE configs[4]) host side for the fused kernel's 64 x 64 configuration: the task program.

  robot keys   robots/robot.py:334-484 (one arm; the Robotiq gripper has six finger joints)
  object keys  pick_place.py:585-668, per object: {obj}_to_robot0_eef_pos, {obj}_to_robot0_eef_quat, {obj}_pos, {obj}_quat.  The relative
               sensors precede the object's own pos / quat sensors in the Observable order and read them from the observation cache, so they
               see the object pose of the PREVIOUS control step against the current gripper pose (and zeros in the record reset() returns);
               the kernel reproduces that (include/rsim.h RSIM_OBS_REL_POS).
  reward       pick_place.py:274-429 (all-objects mode), success :737-762
The bin placement samplers (pick_place.py:431-513) and the visual-object bodies are not restated yet: batches start from given states.
"""
from __future__ import annotations


def pick_place_task(flat, cfg, reward_scale: float = 1.0, reward_shaping: bool = True):
    names, t = flat.names, cfg["task"]
    body, geom, site = names["body"], names["geom"], names["site"]
    gsite = site.index(t["grip_site"])
    eef_body = body.index(t["eef_body"])
    qi, di = cfg["qpos_idx"], cfg["dof_idx"]
    gq, gd = cfg["grip_qpos_idx"], cfg["grip_dof_idx"]
    obs = []
    obs += [("qpos", q, 0) for q in qi] + [("cos", q, 0) for q in qi] + [("sin", q, 0) for q in qi]
    obs += [("qvel", d, 0) for d in di] + [("qacc", d, 0) for d in di]
    obs += [("site_pos", gsite, k) for k in range(3)] + [("body_quat", eef_body, k) for k in range(4)] + [("site_quat", gsite, k) for k in range(4)]
    obs += [("qpos", q, 0) for q in gq] + [("qvel", d, 0) for d in gd]
    obj_body = [body.index(b) for b in t["object_bodies"]]
    pos_slot = []
    for i, ob in enumerate(obj_body):
        obs += [("rel_pos", i, k) for k in range(3)] + [("rel_quat", i, k) for k in range(4)]
        pos_slot.append(len(obs))
        obs += [("body_pos", ob, k) for k in range(3)] + [("body_quat", ob, k) for k in range(4)]
    return dict(obs=obs, task="pick_place", grip_site=gsite, eef_body=eef_body, obj_body=obj_body, pos_slot=pos_slot,
                obj_geoms=[[geom.index(g) for g in gs] for gs in t["object_geoms"]], bin2_pos=t["bin2_pos"], bin_size=t["bin_size"][:2],
                bin_target=[r[:2] for r in t["target_bin_placements"]], left_pad_geoms=[geom.index(g) for g in t["left_pad"]],
                right_pad_geoms=[geom.index(g) for g in t["right_pad"]], reward_scale=reward_scale, reward_shaping=reward_shaping)
