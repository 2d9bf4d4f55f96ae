"""GPU: the attack's parity tests once more with the encoder's products as six bf16 piece products (GEOADV_ENC_ARITH_BF16X3).

tests/test_gpu_attack.py runs under the library default (f16x2: three fp16 piece products per fp32 product); the bf16x3 form
of the encoder stays a supported selection (and the fallback of models outside f16x2's range), so its forward, its masked /
recomputing / Jacobian backward and its tied-pool dense path are held to the same model at the same tolerances here -- the same test functions, collected a second time."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _bf16x3_encoder_default():
    from geometric_adv_amd import _lib
    lib = _lib.lib()
    assert lib.geoadv_set_default_encoder_arith(1) == 0            # GEOADV_ENC_ARITH_BF16X3: models created from here on
    yield
    assert lib.geoadv_set_default_encoder_arith(-1) == 0           # GEOADV_ENC_ARITH_AUTO


from test_gpu_attack import (setup,                                                      # noqa: E402,F401  (module-scoped: re-created here, after the switch)
                             test_ae_forward_matches_model, test_ae_forward_ragged_point_count,       # noqa: F401
                             test_single_iterations_match_model, test_maxpool_exact_ties_split_gradient,   # noqa: F401
                             test_masked_backward_equals_recomputing_backward,                        # noqa: F401
                             test_jacobian_backward_equals_masked_backward,                           # noqa: F401
                             test_forward_and_gradient_match_torch_golden)                            # noqa: F401


def test_the_models_of_this_module_run_the_bf16x3_encoder(setup):
    _, ae, _ = setup
    assert ae.encoder_arith == "bf16x3"
