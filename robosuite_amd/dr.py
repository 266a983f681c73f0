"""Host mirror of the device's dynamics randomisation (rsim_step.hip k_randomize; include/rsim.h rsim_randomize_dynamics): the same counter-based draws, item
numbering, ratio / size rules and clips, in numpy -- so that (a) a GPU test can hold the kernel to it element by element and (b) a CPU test can hold IT to
the reference's DynamicsModder.randomize (utils/mjmod.py:1705-1729) run over the shim: same means, ranges and clip asymmetries per parameter.  The reference
draws from the unseeded global numpy generator, so only distributions can be compared, never a stream."""
from __future__ import annotations

import numpy as np

M64 = (1 << 64) - 1


def dr_uniform(seed: int, step: int, env: int, item) -> np.ndarray:
    """U(-1, 1) with 24 bits, keyed by (seed, step, env, item): rsim_step.hip dr_uniform (two rounds of the splitmix64 finaliser)."""
    item = np.asarray(item, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64((int(seed) + 0x9E3779B97F4A7C15 * (int(step) + 1)) & M64) + ((np.uint64(int(env)) << np.uint64(32)) | item)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9); z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB); z = z ^ (z >> np.uint64(31))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9); z = z ^ (z >> np.uint64(29))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 / 16777216.0) - np.float32(1.0)).astype(np.float32)


def randomize_host(flat, cgeoms, args: dict, seed: int, step: int, env: int, base=None) -> dict:
    """The float model arrays env `env` holds after rsim_randomize_dynamics(args, seed, step), drawn around `base` (default: the compiled model).
    cgeoms: model geom id of every colliding geom, in colliding-geom order (rsim_model_cgeom inverted).  float32 arithmetic as on the device."""
    f32 = np.float32
    a = {k: f32(v) for k, v in args.items() if not k.endswith("_mask")}
    bm, gm, jm = (int(args.get(k, 0)) or M64 for k in ("body_mask", "geom_mask", "joint_mask"))
    src = {k: np.asarray(v, dtype=np.float64).copy() for k, v in (base or flat.arrays).items() if k in
           ("density", "viscosity", "body_pos", "body_quat", "body_inertia", "body_mass", "geom_friction", "geom_solref", "geom_solimp", "dof_frictionloss", "dof_damping", "dof_armature")}
    out = {k: v.astype(np.float32) for k, v in src.items()}
    u = lambda ids: dr_uniform(seed, step, env, ids)     # noqa: E731
    INF = f32(3.0e38)

    def ratio(x, mag, ids, lo=f32(0), hi=INF):
        return np.minimum(hi, np.maximum(lo, x.astype(np.float32) * (f32(1) + mag * u(ids)))) if mag > 0 else x.astype(np.float32)

    def size(x, mag, ids, lo):
        return np.maximum(lo, x.astype(np.float32) + mag * u(ids)) if mag > 0 else x.astype(np.float32)

    out["density"] = ratio(src["density"].ravel(), a["density_ratio"], np.array([1])); out["viscosity"] = ratio(src["viscosity"].ravel(), a["viscosity_ratio"], np.array([2]))
    for bd in range(1, flat.nbody):
        if not (bm >> bd) & 1:
            continue
        i0 = 16 + 16 * bd
        out["body_pos"][bd] = size(src["body_pos"][bd], a["position_size"], i0 + np.arange(3), -INF)
        if a["quaternion_size"] > 0:
            q = src["body_quat"][bd].astype(np.float32) + a["quaternion_size"] * u(i0 + 3 + np.arange(4))
            out["body_quat"][bd] = q * (f32(1) / np.sqrt(np.maximum(np.sum(q * q, dtype=np.float32), f32(1e-20))))
        out["body_inertia"][bd] = ratio(src["body_inertia"][bd], a["inertia_ratio"], i0 + 7 + np.arange(3))
        out["body_mass"][bd] = ratio(src["body_mass"][bd:bd + 1], a["mass_ratio"], np.array([i0 + 10]))[0]
    for c, g in enumerate(cgeoms):
        if not (gm >> c) & 1:
            continue
        i0 = 16 + 16 * 64 + 16 * c
        out["geom_friction"][g] = ratio(src["geom_friction"][g], a["friction_ratio"], i0 + np.arange(3))
        out["geom_solref"][g] = ratio(src["geom_solref"][g], a["solref_ratio"], i0 + 3 + np.arange(2), hi=f32(1))
        out["geom_solimp"][g] = ratio(src["geom_solimp"][g], a["solimp_ratio"], i0 + 5 + np.arange(5))
    for i in range(flat.nv):
        j = int(flat.dof_jntid[i])
        if int(flat.jnt_type[j]) == 0 or not (jm >> j) & 1:
            continue
        i0 = 16 + 32 * 64 + 4 * i
        out["dof_frictionloss"][i] = size(src["dof_frictionloss"][i:i + 1], a["frictionloss_size"], np.array([i0]), f32(0))[0]
        out["dof_damping"][i] = size(src["dof_damping"][i:i + 1], a["damping_size"], np.array([i0 + 1]), f32(0))[0]
        out["dof_armature"][i] = size(src["dof_armature"][i:i + 1], a["armature_size"], np.array([i0 + 2]), f32(0))[0]
    return out
