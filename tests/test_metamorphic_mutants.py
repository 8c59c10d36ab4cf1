"""The metamorphic tests must FAIL on a misread registration — otherwise they pin nothing.  oracle/orc_odo.c and orc_ctx.c carry 58
deliberate misreadings behind `#if ORC_MUTANT == k` (compiled only into oracle/_build/liboracle_mutant_<k>.so by `make mutants`);
tools/mutation_report.py runs both metamorphic modules against each (profiles/r06_mutation_report.txt).  Here, in the suite, one
quick case per kind of misreading: a Jacobian sign, a frame, a weight, a composition — and, since round 6, a map transform, a resize
rule, a validity rule, a rejection threshold, the search's tie order, the multiplier update, the 0.3 m guard, the weighting clamp."""
import os
import subprocess

import pytest

import test_registration_metamorphic as tm
import test_registration_metamorphic2 as t2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    # round 6 (tests/test_registration_metamorphic2.py)
    (28, "model vertices rotated without the translation", lambda o: t2.test_the_model_maps_live_in_the_trackers_world_frame(o)),
    (30, "2 x 2 resize as a NaN-aware mean", lambda o: t2.test_resize_is_nan_if_any_tap_is_nan_and_renormalises_normals(o)),
    (33, "curvature validity without the lower bound", lambda o: t2.test_copy_validity_rules_cost_exactly_the_planted_pixels(o, ("CURV1", 3, -301.0, 100))),
    (35, "the distance threshold met by the squared distance", lambda o: t2.test_distance_rejection_is_euclidean_at_ten_centimetres(o)),
    (36, "the angle threshold met by 1 - cosine", lambda o: t2.test_angle_rejection_is_the_sine_of_twenty_degrees(o, 21.0)),
    (38, "search ties to the last candidate", lambda o: t2.test_search_ties_go_to_the_first_candidate_in_raster_order(t2._Window(o.load()))),
    (41, "the sparse-ICP target moved by z + lambda / mu", lambda o: t2.test_sparse_icp_multiplier_update_doubles_a_standing_offset(o)),
    (43, "no 0.3 m guard", lambda o: t2.test_an_estimate_beyond_thirty_centimetres_is_thrown_away(o)),
    (45, "velocity weighting without its floor", lambda o: t2.test_velocity_weighting_follows_the_stated_clamp(o, "2cm")),
    (48, "the intensity pyramid counting black pixels", lambda o: t2.test_the_intensity_pyramid_skips_black_pixels(o)),
    (49, "no far cut-off for the photometric depth", lambda o: t2.test_the_photometric_term_sees_nothing_beyond_six_metres(o, 6.2, False)),
]


@pytest.fixture(scope="module")
def mutants_built(oracle_lib_built):
    # only the misread builds the cases below load (22 of 58), in parallel: a fresh tree compiles them in ~15 s
    targets = ["_build/liboracle_mutant_%d.so" % k for k in sorted(set(c[0] for c in CASES))]
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle")] + targets, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return oracle_lib_built


@pytest.mark.parametrize("mutant, what, check", CASES, ids=["m%d" % c[0] for c in CASES])
def test_a_misread_registration_fails_its_metamorphic_test(mutants_built, monkeypatch, mutant, what, check):
    check(mutants_built)                                   # the oracle as it is passes
    monkeypatch.setenv("HRBF_ORACLE_MUTANT", str(mutant))
    with pytest.raises(AssertionError):
        check(mutants_built)                               # ... and the misreading is caught
