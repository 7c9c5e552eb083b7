"""Reference `.pth` layout round trip (nerf/utils.py:1431-1530) on the CPU: keys, safe loading, best-checkpoint form."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _have_lib():
    return os.path.exists(os.path.join(os.path.dirname(HERE), "focnerf_amd", "libfocnerf_hip.so"))


@pytest.mark.skipif(not _have_lib(), reason="libfocnerf_hip.so not built")
def test_reference_layout_round_trip(tmp_path):
    from focnerf_amd.network import NeRFNetwork
    from focnerf_amd.checkpoint import save_checkpoint, load_checkpoint
    torch.manual_seed(0)
    a = NeRFNetwork(bound=2, cuda_ray=True)
    a.encoder.embeddings.data.uniform_(-1, 1)
    a.density_grid.uniform_(0, 5)
    a.density_bitfield.random_(0, 255)
    a.mean_count, a.mean_density = 77, 1.25
    keys = set(a.state_dict().keys())
    # the module paths the reference's network_ff / renderer register (network_ff.py:29-49, renderer.py:83-98, grid.py:131-137)
    assert {"encoder.embeddings", "encoder.offsets", "sigma_net.weights", "color_net.weights", "aabb_train", "aabb_infer",
            "density_grid", "density_bitfield", "step_counter"} <= keys
    p = tmp_path / "ngp_ep0001.pth"
    save_checkpoint(a, str(p), epoch=1, global_step=100, stats={"loss": [0.5], "checkpoints": [], "best_result": None})
    raw = torch.load(str(p), weights_only=True)
    assert set(raw.keys()) >= {"epoch", "global_step", "stats", "model", "mean_count", "mean_density"}
    b = NeRFNetwork(bound=2, cuda_ray=True)
    missing, unexpected = load_checkpoint(b, str(p))
    assert missing == [] and unexpected == []
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]), k
    assert b.mean_count == 77 and b.mean_density == 1.25
    # "best" checkpoints drop density_grid (utils.py:1484-1485): loads with that one key missing
    pb = tmp_path / "ngp.pth"
    save_checkpoint(a, str(pb), best=True)
    c = NeRFNetwork(bound=2, cuda_ray=True)
    missing, unexpected = load_checkpoint(c, str(pb))
    assert missing == ["density_grid"] and unexpected == []
    assert torch.equal(c.density_bitfield, a.density_bitfield)
    # a bare state_dict file (utils.py:1509-1512)
    ps = tmp_path / "bare.pth"
    torch.save(a.state_dict(), str(ps))
    d = NeRFNetwork(bound=2, cuda_ray=True)
    assert load_checkpoint(d, str(ps)) == ([], [])
