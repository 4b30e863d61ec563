"""Pin the CPU oracle against the golden vectors produced by the live reference module
(tests/golden/make_golden.py; reference = cc_attention/functions.py:27-49 + autograd)."""
import math

import pytest
import torch

from conftest import SMALL_CASES, load_golden, regenerate_module_inputs
from oracle import cca_oracle as O

TOL = 2e-5   # fp32 reference vs fp32 einsum restatement: summation-order noise only


def params_of(g):
    return {k[len("param."):]: v for k, v in g.items() if k.startswith("param.")}


@pytest.mark.parametrize("case", SMALL_CASES)
def test_forward_intermediates_match_reference(case):
    g = load_golden(case)
    H = g["x"].shape[2]
    A = O.ca_softmax(O.ca_forward(g["q"], g["k"]))
    assert torch.allclose(A, g["A"], atol=TOL)
    # structural zero of the column self-slot (functions.py:11-12,38): exactly 0, not merely small
    idx = torch.arange(H)
    assert torch.all(A[:, idx, :, idx] == 0)
    assert torch.all(g["A"][:, idx, :, idx] == 0)
    o = O.ca_map_forward(g["A"], g["v"])
    y = g["param.gamma"] * o + g["x"]
    assert torch.allclose(y, g["y"], atol=TOL)


@pytest.mark.parametrize("case", SMALL_CASES)
def test_backward_pieces_match_reference_autograd(case):
    g = load_golden(case)
    gamma = g["param.gamma"]
    # dA captured on the reference's softmax output == gamma * t
    t, dv_unscaled = O.ca_map_backward(g["dy"], g["A"], g["v"])
    assert torch.allclose(gamma * t, g["dA"], atol=TOL)
    assert torch.allclose(gamma * dv_unscaled, g["dv"], atol=TOL)
    dE = O.ca_softmax_backward(g["A"], g["dA"])
    dq, dk = O.ca_backward(dE, g["q"], g["k"])
    assert torch.allclose(dq, g["dq"], atol=TOL)
    assert torch.allclose(dk, g["dk"], atol=TOL)
    core = O.cca_core_backward(g["dy"], g["q"], g["k"], g["v"], g["A"], gamma)
    assert torch.allclose(core["dgamma"], g["grad.gamma"], rtol=1e-4, atol=1e-4)
    for n in ("dq", "dk", "dv"):
        assert torch.allclose(core[n], g[n], atol=TOL)


@pytest.mark.parametrize("case", SMALL_CASES)
def test_module_level_matches_reference(case):
    g = load_golden(case)
    y, dx, grads = O.cca_module_forward_backward(g["x"], params_of(g), g["dy"])
    assert torch.allclose(y, g["y"], atol=TOL)
    assert torch.allclose(dx, g["dx"], atol=1e-4)
    for n, gr in grads.items():
        assert torch.allclose(gr, g["grad." + n], rtol=1e-4, atol=2e-4), n


@pytest.mark.parametrize("case", ["cfg1_2x64x32x32", "fast_1x64x97x97"])
def test_config1_matches_reference(case):
    """BASELINE.json configs[0]: (2,64,32,32) fp32, and the headline geometry 97x97 at C = 64; inputs regenerated from
    the seed."""
    g = load_golden(case)
    B, C, H, W = [int(v) for v in g["shape"]]
    x, dy, params = regenerate_module_inputs(B, C, H, W)
    if abs(float(x.double().sum()) - float(g["fingerprint.x"][0])) > 1e-6:
        pytest.skip("torch RNG stream differs from the build container's; fixture inputs not reproducible")
    assert float(x.flatten()[12345]) == pytest.approx(float(g["fingerprint.x"][1]), abs=0)
    y, dx, grads = O.cca_module_forward_backward(x, params, dy)
    assert torch.allclose(y, g["y"], atol=TOL)
    assert torch.allclose(dx, g["dx"], atol=1e-4)
    for n, gr in grads.items():
        assert torch.allclose(gr, g["grad." + n], rtol=1e-4, atol=1e-3), n


def test_loop_restatement_pins_index_map():
    torch.manual_seed(3)
    q, k = torch.randn(1, 3, 4, 5, dtype=torch.float64), torch.randn(1, 3, 4, 5, dtype=torch.float64)
    v = torch.randn(1, 6, 4, 5, dtype=torch.float64)
    e = O.ca_forward(q, k)
    el = O.ca_forward_loops(q, k)
    assert torch.equal(torch.isinf(e), torch.isinf(el))
    fin = ~torch.isinf(e)
    assert torch.allclose(e[fin], el[fin], atol=1e-12)
    A = O.ca_softmax(e)
    assert torch.allclose(O.ca_map_forward(A, v), O.ca_map_forward_loops(A, v), atol=1e-12)


def test_closed_form_backward_equals_autograd_fp64():
    """The closed form against autograd of the einsum forward, H != W."""
    torch.manual_seed(5)
    B, C, H, W = 2, 16, 5, 7
    q = torch.randn(B, 2, H, W, dtype=torch.float64, requires_grad=True)
    k = torch.randn(B, 2, H, W, dtype=torch.float64, requires_grad=True)
    v = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
    x = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
    gamma = torch.tensor([0.7], dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, C, H, W, dtype=torch.float64)
    y, A = O.cca_core_forward(q, k, v, x, gamma)
    y.backward(dy)
    g = O.cca_core_backward(dy, q.detach(), k.detach(), v.detach(), A.detach(), gamma.detach())
    for n, t in (("dq", q), ("dk", k), ("dv", v), ("dx", x), ("dgamma", gamma)):
        assert torch.allclose(g[n], t.grad, atol=1e-12), n


def test_gamma_zero_is_identity_and_kills_qkv_grads():
    torch.manual_seed(1)
    q, k = torch.randn(1, 2, 4, 4), torch.randn(1, 2, 4, 4)
    v, x, dy = torch.randn(1, 16, 4, 4), torch.randn(1, 16, 4, 4), torch.randn(1, 16, 4, 4)
    gamma = torch.zeros(1)
    y, A = O.cca_core_forward(q, k, v, x, gamma)
    assert torch.equal(y, x)
    g = O.cca_core_backward(dy, q, k, v, A, gamma)
    assert torch.all(g["dq"] == 0) and torch.all(g["dk"] == 0) and torch.all(g["dv"] == 0)
    assert float(g["dgamma"].abs()) > 0


def test_accounting_matches_survey():
    assert O.algorithmic_bytes(8, 512, 97, 97) == 1_040_560_128
    assert math.isclose(O.algorithmic_flops(8, 512, 97, 97) / 1e9, 50.47, rel_tol=1e-3)
