"""The metamorphic tests must FAIL on a misread registration — otherwise they pin nothing.  oracle/orc_odo.c carries 26
deliberate misreadings behind `#if ORC_MUTANT == k` (compiled only into oracle/_build/liboracle_mutant_<k>.so by `make mutants`);
tools/mutation_report.py runs the whole metamorphic module against each (profiles/r05_metamorphic_mutation_report.txt).  Here, in
the suite, one quick case per kind of misreading: a Jacobian sign, a frame, a weight, a composition."""
import os
import subprocess

import pytest

import test_registration_metamorphic as tm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mutants_built(oracle_lib_built):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "mutants"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return oracle_lib_built


CASES = [
    (3, "the photometric row's rotational columns with the opposite sign", lambda o: tm.test_photometric_term_meets_the_half_pixel_bound_of_every_level(o, "room", "5px")),
    (1, "the ICP normal left in the world frame", lambda o: tm.test_icp_rows_live_in_the_previous_cameras_frame(o)),
    (6, "A_rgb + w A_icp", lambda o: tm.test_joint_system_moves_with_the_depth_unit_as_the_algebra_predicts(o)),
    (7, "b_rgb + w^2 b_icp (the consistent weighting the reference does not use)", lambda o: tm.test_joint_system_moves_with_the_depth_unit_as_the_algebra_predicts(o)),
    (9, "T_prev * dT instead of T_prev * dT^-1", lambda o: tm.test_icp_term_alone_recovers_the_motion_in_the_corner(o)),
    (11, "SO3 residual with the opposite sign", lambda o: tm.test_so3_prealignment_recovers_a_pure_rotation_to_half_a_level2_pixel(o, tm.VGA, (0.0, 0.03, 0.0))),
    (16, "the depth gate on the pixel's own depth", lambda o: tm.test_the_photometric_depth_gate_is_on_the_depth_in_the_model_camera(o)),
    (20, "the depth pyramid by plain subsampling", lambda o: tm.test_the_depth_pyramid_averages_over_the_valid_taps_only(o)),
    (22, "the increment composed on the right", lambda o: tm.test_the_loop_runs_4_5_10_iterations_and_every_increment_acts_on_the_left(o)),
    (23, "the gradient threshold unsquared", lambda o: tm.test_a_texture_below_the_gradient_threshold_contributes_nothing_on_level_0(o)),
    (19, "the photometric weight 1 / sigma", lambda o: tm.test_the_photometric_weight_depends_on_sigma_plus_the_residual_only(o)),
]


@pytest.mark.parametrize("mutant, what, check", CASES, ids=["m%d" % c[0] for c in CASES])
def test_a_misread_registration_fails_its_metamorphic_test(mutants_built, monkeypatch, mutant, what, check):
    check(mutants_built)                                   # the oracle as it is passes
    monkeypatch.setenv("HRBF_ORACLE_MUTANT", str(mutant))
    with pytest.raises(AssertionError):
        check(mutants_built)                               # ... and the misreading is caught
