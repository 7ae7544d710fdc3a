"""Pin the CPU oracle (oracle/jat_oracle.py) against fixtures produced by the reference's own classes
(oracle/gen_golden.py).  The reference runs fp32; its own fp32-vs-fp64 noise floor is ~6e-7 rel-L2
(stored in each fixture), so the restatement must agree to <= 1e-5 (SURVEY.md §8c gate)."""
import numpy as np
import pytest

import jatsr_amd.recipe as recipe
from helpers import fwd_inputs, load_golden, rel_l2, sampler_inputs, sub
from oracle import jat_oracle as O

TOL = 1e-5

FWD_FAST = ["fwd_micro_T24", "fwd_micro_T22_pad", "fwd_micro_ln_T24", "fwd_tiny_T128", "fwd_tiny_T516_pad"]
FWD_BIG = ["fwd_v3mod2_T512", "fwd_v3mod2_T1378"]


def _check_forward(name, dtype=np.float32):
    z, meta = load_golden(name)
    cfg, x_t, t, x_c = fwd_inputs(meta)
    m = O.OracleModel(cfg, recipe.make_state_dict(cfg, meta["norm"], meta["salt"]), meta["norm"], dtype)
    out = m.forward(x_t, t, x_c, record=True)
    assert out.shape == x_t.shape                       # reference test_model(): jat_audiosr_v3.py:509
    o = out if meta["full"] else sub(out, *meta["s_out"])
    assert rel_l2(o, z["out"]) < TOL
    assert rel_l2(o, z["out64"]) < TOL
    assert abs(np.linalg.norm(out.astype(np.float64)) / float(z["out_l2"]) - 1) < 1e-5
    for k in z.files:
        if k.startswith("st_"):
            s = m.stages[k[3:]]
            s = s if (meta["full"] or s.ndim == 2) else sub(s, *meta["s_st"])
            assert rel_l2(s, z[k]) < TOL, k
            assert abs(np.linalg.norm(m.stages[k[3:]].astype(np.float64)) / float(z["l2_" + k[3:]]) - 1) < 1e-5


@pytest.mark.parametrize("name", FWD_FAST)
def test_forward_matches_reference(name):
    _check_forward(name)


@pytest.mark.parametrize("name", FWD_BIG)
def test_forward_matches_reference_v3mod2(name):
    _check_forward(name)


def test_forward_fp64_oracle_is_ground_truth():
    z, meta = load_golden("fwd_micro_T24")
    cfg, x_t, t, x_c = fwd_inputs(meta)
    m = O.OracleModel(cfg, recipe.make_state_dict(cfg, "rms", meta["salt"]), "rms", np.float64)
    assert rel_l2(m.forward(x_t, t, x_c), z["out64"]) < 1e-7


@pytest.mark.parametrize("name", ["sampler_micro_cfg3", "sampler_micro_nocfg", "sampler_tiny_cfg3"])
def test_sampler_matches_reference(name):
    z, meta = load_golden(name)
    cfg, lr, z0 = sampler_inputs(meta)
    m = O.OracleModel(cfg, recipe.make_state_dict(cfg, "rms", meta["salt"]), "rms", np.float32)
    out = O.flow_matching_sample(m, lr, z0, meta["steps"], meta["cfg_scale"])
    o = out if meta["full"] else sub(out, *meta["s_out"])
    assert rel_l2(o, z["z"]) < 2e-5   # 50 chained fp32 forwards
    assert abs(np.linalg.norm(out.astype(np.float64)) / float(z["z_l2"]) - 1) < 1e-5


def test_misc_goldens():
    z, _ = load_golden("misc")
    for n in (51, 11, 8):
        assert np.array_equal(O.linspace_f32(0.0, 1.0, n), z[f"linspace{n}"])
    chunks = [recipe.gaussian("chunk", (1, 6, n), i) for i, n in enumerate((40, 40, 23))]
    assert np.allclose(O.crossfade_chunks(chunks, 8), z["xfade_ov8"], atol=1e-6)
    assert np.array_equal(O.crossfade_chunks(chunks, 0), z["xfade_ov0"])
    assert np.array_equal(O.crossfade_chunks(chunks[:1], 8), z["xfade_single"])
    assert O.crossfade_chunks([], 8) is None
    assert float(z["zero_init_absmax"]) == 0.0
    assert int(z["flops_tiny_B2_T128"]) == O.forward_flops(recipe.CONFIGS["tiny"], 2, 128) == 5395972096
    assert O.forward_flops(recipe.CONFIGS["v3mod2"], 1, 512) == 127627689984


def test_zero_init_gives_zero_output():
    """A freshly initialised model (adaLN + final linear zero, jat_audiosr_v3.py:395-404) outputs exact 0."""
    cfg = recipe.CONFIGS["micro"]
    sd = recipe.make_state_dict(cfg)
    for k in sd:
        if "adaLN_modulation" in k or k.startswith("final_layer.1"):
            sd[k] = np.zeros_like(sd[k])
    x_t, x_c = recipe.make_latents(1, 32, 16, salt=7)
    out = O.OracleModel(cfg, sd).forward(x_t, np.array([0.3], np.float32), x_c)
    assert np.abs(out).max() == 0.0


def test_sequence_too_long_raises():
    cfg = recipe.CONFIGS["micro"]
    m = O.OracleModel(cfg, recipe.make_state_dict(cfg))
    x = np.zeros((1, 32, 4 * 2049), np.float32)
    with pytest.raises(ValueError):
        m.forward(x, np.array([0.5], np.float32), x)


def test_bad_head_config_asserts():
    cfg = dict(recipe.CONFIGS["micro"], num_q_heads=3)
    with pytest.raises(AssertionError):
        O.OracleModel(cfg, {})


def test_chunk_plan_matches_reference_arithmetic():
    # infer_test_v3m2.py:340-361: T=4096 -> 4 chunks [0:1378],[1206:2584],[2412:3790],[3618:4096]
    assert O.chunk_plan(4096) == [(0, 1378), (1206, 2584), (2412, 3790), (3618, 4096)]
    assert O.chunk_plan(1378) == [(0, 1378)]
