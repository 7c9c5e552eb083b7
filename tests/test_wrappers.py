"""The Python operator wrappers against tests/golden/wrappers.npz — the reference's own gridencoder/grid.py and ffmlp/ffmlp.py driven on
the CPU with oracle-backed stub backends (tests/golden/make_golden.py). CPU part: constructor arithmetic (level offsets, scale, weight
blob, seed-42 init). GPU part: the same calls through focnerf_amd on the device reproduce the wrapper-level results."""
import os

import numpy as np
import pytest
import torch

from util import to_np, assert_half_close

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "focnerf_amd", "libfocnerf_hip.so")
needs_lib = pytest.mark.skipif(not os.path.exists(LIB), reason="libfocnerf_hip.so not built")
SMALL = dict(input_dim=3, num_levels=8, level_dim=2, base_resolution=4, log2_hashmap_size=12, desired_resolution=96)


def _fx():
    return np.load(os.path.join(HERE, "golden", "wrappers.npz"))


@needs_lib
def test_constructors_match_reference_wrappers():
    from focnerf_amd.gridencoder import GridEncoder
    from focnerf_amd.ffmlp import FFMLP
    g = _fx()
    big = GridEncoder(desired_resolution=2048)
    assert np.array_equal(big.offsets.numpy(), g["big_offsets"]) and big.per_level_scale == float(g["big_per_level_scale"])
    assert big.embeddings.shape == (int(g["big_offsets"][-1]), 2) and big.output_dim == 32
    small = GridEncoder(**SMALL)
    assert np.array_equal(small.offsets.numpy(), g["ge_offsets"]) and small.per_level_scale == float(g["ge_per_level_scale"])
    assert small.output_dim == int(g["ge_output_dim"])
    mlp = FFMLP(input_dim=32, output_dim=3, hidden_dim=64, num_layers=2)
    assert np.array_equal(mlp.weights.detach().numpy(), g["ff_weights"]), "weight blob size / seed-42 initialisation (ffmlp.py:120-144)"
    assert list(g["ff_allocate_splitk"]) == [mlp.num_layers + 1]
    # the reference pads every batch to the next multiple of 128 — a full extra block when it is already aligned (ffmlp.py:157-159)
    assert list(g["ff_inference_B"]) == [256] and list(g["ff_forward_B"]) == [256] and int(g["ff_inference_B_aligned"]) == 256


@pytest.mark.gpu
def test_grid_encoder_module_matches_reference_wrapper():
    from focnerf_amd.gridencoder import GridEncoder
    g = _fx()
    enc = GridEncoder(**SMALL).cuda()
    enc.embeddings.data.copy_(torch.from_numpy(g["ge_embeddings"]))
    x = torch.from_numpy(g["ge_x"]).cuda().requires_grad_(True)
    y = enc(x, bound=2)                                             # fp32, no autocast: what the CPU run of the reference computed
    assert y.shape == g["ge_y"].shape and y.dtype == torch.float32
    assert np.array_equal(to_np(y), g["ge_y"]), "forward through normalisation, kernel and layout"
    y.backward(torch.from_numpy(g["ge_gy"]).cuda())
    np.testing.assert_allclose(to_np(enc.embeddings.grad), g["ge_grad_embeddings"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(to_np(x.grad), g["ge_grad_x"], atol=1e-3, rtol=1e-3)


@pytest.mark.gpu
def test_ffmlp_module_matches_reference_wrapper():
    from focnerf_amd.ffmlp import FFMLP
    g = _fx()
    mlp = FFMLP(input_dim=32, output_dim=3, hidden_dim=64, num_layers=2).cuda()
    assert np.array_equal(to_np(mlp.weights), g["ff_weights"])
    x = torch.from_numpy(g["ff_x"]).cuda()
    mlp.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y = mlp(x)
    assert y.shape == (200, 3) and y.dtype == torch.float16
    # the oracle behind the fixture accumulates in fp16 like the reference's WMMA fragments, the MFMA kernels in fp32: a few half-ulps
    assert_half_close(to_np(y), g["ff_y_eval"], ulps=8, atol=4e-3, what="inference")
    mlp.train()
    xt = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        yt = mlp(xt)
    assert_half_close(to_np(yt), g["ff_y_train"], ulps=8, atol=4e-3, what="training forward")
    yt.backward(torch.from_numpy(g["ff_gy"]).cuda())
    assert_half_close(to_np(xt.grad), g["ff_grad_x"], ulps=8, atol=2e-3, what="grad_inputs")
    gw, gw_ref = to_np(mlp.weights.grad).astype(np.float32), g["ff_grad_w"].astype(np.float32)
    assert gw.shape == gw_ref.shape
    assert np.abs(gw - gw_ref).max() <= 2e-2 * np.abs(gw_ref).max() + 1e-3
