"""samplenerfro_amd.utils.default_flags and the workloads of bench.py / the tests against the reference's own flag table and shipped configs,
extracted as data from rnerf/utils.py:define_flags and configs/*.yaml (tests/golden/make_reference_flags.py -> tests/golden/reference_flags.json)."""
import importlib.util
import json
import os

import pytest

from samplenerfro_amd import synthetic as syn, utils

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "reference_flags.json")))
OURS_ONLY = {"backward_precision", "range_retry"}          # arithmetic / recovery switches of this implementation, not reference flags


def test_the_committed_table_is_what_the_reference_defines():
    spec = importlib.util.spec_from_file_location("make_reference_flags", os.path.join(HERE, "golden", "make_reference_flags.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    now = gen.extract()
    if now is None:
        pytest.skip("the reference is not on this machine")
    assert json.loads(json.dumps(now, sort_keys=True)) == REF


def test_default_flags_are_the_references_defaults():
    """Every hot-path flag default_flags carries has the reference's default (rnerf/utils.py:87-245) — use_online_sparsity = True and
    noise_std = None among them —, and everything it adds is listed."""
    mine = vars(utils.default_flags())
    assert set(mine) - set(REF["defaults"]) == OURS_ONLY
    for k, v in mine.items():
        if k in OURS_ONLY:
            continue
        assert REF["defaults"][k]["default"] == v, (k, v, REF["defaults"][k])
    assert REF["defaults"]["use_online_sparsity"]["default"] is True and REF["defaults"]["noise_std"]["default"] is None


def test_the_bench_workloads_carry_the_shipped_configs_numbers():
    """bench.py's train step (Stepper) uses the loss terms every shipped scene config but ball / glass / pen sets (bg_weight 0.025,
    bg_smooth_weight 1.0 on a 128 x 128 env-map patch, use_online_sparsity off, randomized on, no white background); its workloads take their
    sample counts and depth ranges from the configs BASELINE.json names."""
    cfgs = REF["configs"]
    ship = cfgs["ship_skydome-bkgd_no-partial-reflect_cycles"]
    for name in ("example", "dolphin", "ship_skydome-bkgd_no-partial-reflect_cycles"):
        c = cfgs[name]
        assert (c["bg_weight"], c["bg_smooth_weight"], c["bg_patch_size"], c["use_online_sparsity"], c["randomized"], c["white_bkgd"]) == (0.025, 1.0, 128, False, True, False)
    for c in cfgs.values():        # what EVERY shipped config agrees on: the branches this implementation's product path is built and measured for
        assert c["use_online_sparsity"] is False and c["sh_deg"] == -1 and c["sh_direnc_deg"] == -1 and c["use_viewdirs"] is True and "noise_std" not in c
    d = utils.default_flags()
    assert (syn.CONFIGS["example"]["S"], syn.CONFIGS["example"]["F"], syn.CONFIGS["example"]["P"]) == (cfgs["example"]["num_coarse_samples"], cfgs["example"]["num_fine_samples"], cfgs["example"]["num_path_samples"])
    assert (syn.CONFIGS["example"]["near"], syn.CONFIGS["example"]["far"]) == (d.near, d.far)                  # example.yaml keeps the defaults
    assert syn.CONFIGS["ship_straight"]["P"] == syn.CONFIGS["ship_refractive"]["P"] == ship["num_path_samples"]
    dol = cfgs["dolphin"]
    assert (syn.CONFIGS["dolphin_train"]["S"], syn.CONFIGS["dolphin_train"]["F"], syn.CONFIGS["dolphin_train"]["P"], syn.CONFIGS["dolphin_train"]["near"],
            syn.CONFIGS["dolphin_train"]["far"]) == (dol["num_coarse_samples"], dol["num_fine_samples"], dol["num_path_samples"], dol["near"], dol["far"])
    gl = cfgs["glass"]
    assert (syn.CONFIGS["glass_frame"]["P"], syn.CONFIGS["glass_frame"]["near"], syn.CONFIGS["glass_frame"]["far"]) == (gl["num_path_samples"], gl["near"], gl["far"])


def test_the_workloads_grids_and_model_switches_are_the_shipped_gin_bindings():
    """configs/*.gin: prefilter kernel (Config.kernel_size / kernel_sigma, G1), grid resolution (the number in Config.voxel_grid), and the two
    NerfModel switches a shipped file sets — bd_cut_dist = 6.0 in ball / glass / pen (row M4) and use_mask_bbox = False everywhere."""
    gin = REF["gin"]
    assert all(g["NerfModel.use_mask_bbox"] is False and g["VoxMLP.interp_method"] == "linear3" and g["VoxMLP.annealed"] is True for g in gin.values())
    cut = sorted(n for n, g in gin.items() if "NerfModel.bd_cut_dist" in g)
    assert cut == ["ball", "glass", "pen"] and all(gin[n]["NerfModel.bd_cut_dist"] == 6.0 for n in cut)      # the configs NerfModel._bd_cut_bbox knows
    w = syn.CONFIGS
    assert (w["example"]["ksize"], w["example"]["ksigma"]) == (gin["example"]["Config.kernel_size"], gin["example"]["Config.kernel_sigma"])
    assert (w["dolphin_train"]["ksize"], w["dolphin_train"]["ksigma"], w["dolphin_train"]["G"]) == (gin["dolphin"]["Config.kernel_size"], gin["dolphin"]["Config.kernel_sigma"], 256)
    assert "uni256" in gin["dolphin"]["Config.voxel_grid"]
    assert (w["glass_frame"]["ksize"], w["glass_frame"]["ksigma"], w["glass_frame"]["G"]) == (gin["glass"]["Config.kernel_size"], gin["glass"]["Config.kernel_sigma"], 384)
    assert "uni384" in gin["glass"]["Config.voxel_grid"]
    ship = gin["ship_skydome-bkgd_no-partial-reflect_cycles"]
    assert (w["ship_refractive"]["ksize"], w["ship_refractive"]["ksigma"]) == (ship["Config.kernel_size"], ship["Config.kernel_sigma"])
