"""Inputs for the batched keyframe refinement tests (BASELINE.json configs[4]): local maps and down-sampled keyframe
stacks captured from a scan-to-map run of the ORACLE, plus perturbed initial poses.  Both libraries get the same bytes."""
import numpy as np

from lio_amd import capi, synth
from mapping_util import drifting_inputs


def perturbed(T, rng, dpos, drot):
    q, p = T
    R = synth.rot_from_quat(np.asarray(q, np.float64)) @ synth.small_rot(rng.uniform(-drot, drot, 3))
    return synth.quat_from_rot(R).astype(np.float32), (np.asarray(p, np.float64) + rng.uniform(-dpos, dpos, 3)).astype(np.float32)


def keyframe_inputs(oracle, kind, n_frames, n_perturb, seed=7, dpos=0.08, drot=0.01, map_builder=0, **dataset_kw):
    """-> maps [(corner_from_map, surf_from_map)], keyframes [(map_index, corner_stack, surf_stack, T_init, T_ref)]
    T_ref = the pose the sequential scan-to-map run settled on for that frame (the perturbations should come back near it)."""
    rng = np.random.default_rng(seed)
    frames = drifting_inputs(oracle, kind, n_frames, **dataset_kw)
    mp = capi.PointMapping(oracle, map_builder=map_builder)
    maps, kfs = [], []
    for k, (corner, surf, T_sum, _) in enumerate(frames):
        r = mp.process(corner, surf, T_sum)
        if k == 0:
            continue
        maps.append((mp.cloud(capi.PointMapping.CORNER_FROM_MAP), mp.cloud(capi.PointMapping.SURF_FROM_MAP)))
        cs, ss = mp.cloud(capi.PointMapping.CORNER_STACK_DS), mp.cloud(capi.PointMapping.SURF_STACK_DS)
        for _ in range(n_perturb):
            kfs.append((len(maps) - 1, cs, ss, perturbed(r["T_aft"], rng, dpos, drot), r["T_aft"]))
    return maps, kfs


def load(batch, maps, kfs):
    for cm, sm in maps:
        batch.add_map(cm, sm)
    for mi, cs, ss, T0, _ in kfs:
        batch.add_keyframe(mi, cs, ss, T0)
    return batch
