"""Keyframe sets on which the reference's own Gauss-Newton loops (PointMapping::OptimizeTransformTobeMapped, MapBuilder::OptimizeMap) were
run one keyframe at a time for tests/golden/ref_kf_refine.json — shared by the generator (tests/golden/make_ref_kf_refine.py, build
container only) and tests/test_ref_kf_refine.py.  name -> (kind, frames, perturbations per frame, four_dof)"""
from kf_util import keyframe_inputs

CASES = {
    "indoor_6dof": ("indoor", 3, 3, 0),
    "outdoor_6dof": ("outdoor", 2, 2, 0),
    "indoor_4dof": ("indoor", 3, 3, 1),
}


def inputs(oracle, name):
    kind, n_frames, n_perturb, four_dof = CASES[name]
    return keyframe_inputs(oracle, kind, n_frames, n_perturb, map_builder=four_dof)
