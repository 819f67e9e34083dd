"""CPU: loading reference-style checkpoints into the tcnn duck types (quadraturefields_amd/tinycudann.py).

The flat ``params`` order ``[network | grid]`` is restated from memory (tcnn's source is not in the
container): parity unpinned.  What CAN be done offline is to make the two failure modes loud -- a wrong size names the
expected split, and a right-sized vector whose dense MLP weights sit at the END is warned about instead of rendering
garbage silently (reference call sites: examples/radiance_fields/ngp.py:709-727,
examples/train_finetune.py:407-409).
"""
import warnings

import pytest
import torch

from quadraturefields_amd import tinycudann as tcnn

ENC = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 14, "base_resolution": 16,
       "per_level_scale": 1.4472692012786865}
NET = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}


def _module(log2_t=14):
    return tcnn.NetworkWithInputEncoding(3, 16, dict(ENC, log2_hashmap_size=log2_t), NET)


def _trained_like(m, seed=0):
    """A flat vector in the ASSUMED order with the statistics of a trained model: Xavier-scale dense weights, coarse
    levels grown to ~1e-2, the hashed fine levels mostly still near tcnn's U(-1e-4, 1e-4) initialisation."""
    g = torch.Generator().manual_seed(seed)
    n = m.n_network_params
    net = (torch.rand(n, generator=g) * 2 - 1) * 0.25
    rows = m.grid.n_rows
    grid = (torch.rand(rows * 2, generator=g) * 2 - 1) * 1e-4
    coarse = int(m.grid.desc.offset[4]) * 2
    grid[:coarse] = torch.randn(coarse, generator=g) * 2e-2
    touched = torch.rand(rows * 2, generator=g) < 0.3
    grid[touched] += torch.randn(int(touched.sum()), generator=g) * 3e-3
    return torch.cat([net, grid])


def test_right_order_loads_silently_and_transposed_order_warns():
    m = _module()
    flat = _trained_like(m)
    n = m.n_network_params
    with warnings.catch_warnings():
        warnings.simplefilter("error")                   # any warning would fail the test
        m.load_state_dict({"params": flat})
    assert torch.equal(m.params.detach(), flat)
    assert m.layout_check(flat)[0] == m.LAYOUT_OK
    # a freshly initialised module (tcnn's init statistics) is fine too
    assert m.layout_check(_module().params)[0] == m.LAYOUT_OK
    transposed = torch.cat([flat[n:], flat[:n]])         # [grid | network]: same size, total garbage if loaded as is
    verdict, assumed, swapped = m.layout_check(transposed)
    assert verdict == m.LAYOUT_SUSPECT and swapped > 10 * assumed
    with pytest.warns(UserWarning, match=r"looks like \[grid \| network\]") as rec:
        m.load_state_dict({"params": transposed})
    text = str(rec[0].message)
    assert f"network ({n})" in text and "SURVEY.md A.2" in text and f"p[-{n}:]" in text
    assert torch.equal(m.params.detach(), transposed)    # warned, not refused: the heuristic cannot prove it
    # nested, as the reference's fields hold it (NGPRadianceField.mlp_base): the prefix is named
    holder = torch.nn.Module()
    holder.mlp_base = _module()
    with pytest.warns(UserWarning, match=r"mlp_base\.params"):
        holder.load_state_dict({"mlp_base.params": transposed})


def test_wrong_size_names_the_expected_split_and_the_size_that_would_fit():
    m = _module(14)
    other = _module(15)
    with pytest.raises(RuntimeError) as e:
        m.load_state_dict({"params": other.params.detach().clone()})
    text = str(e.value)
    assert f"expects {m.params.numel()}" in text and f"[{m.n_network_params} network weights | 2 x {m.grid.n_rows} grid rows]" in text
    assert "log2_hashmap_size=14" in text and "fits log2_hashmap_size=15" in text
    # the plain Network and the grid-only Encoding explain themselves as well
    net = tcnn.Network(16, 3, NET)
    with pytest.raises(RuntimeError, match=r"expects 2048 = 2048 network weights"):
        net.load_state_dict({"params": torch.zeros(100)})
    enc = tcnn.Encoding(3, ENC)
    with pytest.raises(RuntimeError, match=r"grid rows"):
        enc.load_state_dict({"params": torch.zeros(100)})
    # the right sizes still load, strictly
    net.load_state_dict({"params": torch.zeros(net.params.numel())})
    enc.load_state_dict({"params": torch.zeros(enc.params.numel())})
