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


def _reference_shaped_state(model_sd, best):
    """What nerf/utils.py:1431-1470 writes: numpy float64 in stats (PSNRMeter.measure, :563-575), and for full checkpoints the
    optimizer / lr_scheduler / scaler / ema dictionaries; best checkpoints carry the model without density_grid."""
    import numpy as np
    state = {"epoch": 12, "global_step": 4800,
             "stats": {"loss": [0.01, 0.008], "valid_loss": [0.009], "results": [np.float64(30.25), np.float64(31.5)], "checkpoints": ["ngp_ep0011.pth"],
                       "best_result": np.float64(31.5)}}
    sd = {k: v.clone() for k, v in model_sd.items()}
    if best:
        sd.pop("density_grid", None)
    else:
        state["optimizer"] = {"state": {0: {"step": torch.tensor(4800.0), "exp_avg": torch.zeros(4), "exp_avg_sq": torch.zeros(4)}},
                              "param_groups": [{"lr": 0.01, "betas": (0.9, 0.99), "eps": 1e-15, "params": [0]}]}
        state["lr_scheduler"] = {"last_epoch": 4800, "_step_count": 4801, "base_lrs": [0.01]}
        state["scaler"] = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 17}
        state["ema"] = {"decay": 0.95, "num_updates": 12, "shadow_params": [torch.zeros(4)], "collected_params": None}
    state["model"] = sd
    return state


@pytest.mark.skipif(not _have_lib(), reason="libfocnerf_hip.so not built")
def test_reference_shaped_checkpoints_load_through_load_objects(tmp_path):
    """Best checkpoints always carry numpy scalars (they are written only once `stats['results']` is non-empty, utils.py:1472): the
    weights-only loader has to take them, for the K-object loader COMBINED's flow needs, without ever unpickling code."""
    import numpy as np
    from focnerf_amd.network import NeRFNetwork
    from focnerf_amd.checkpoint import load_objects, safe_load
    paths, nets = [], []
    for k in range(3):
        torch.manual_seed(k)
        n = NeRFNetwork(bound=1)
        n.encoder.embeddings.data.uniform_(-1, 1)
        nets.append(n)
        p = tmp_path / f"obj{k}.pth"
        torch.save(_reference_shaped_state(n.state_dict(), best=(k != 1)), str(p))
        paths.append(str(p))
    with pytest.raises(Exception):
        torch.load(paths[0], weights_only=True)               # the premise: torch's default allow-list refuses the file
    raw = safe_load(paths[0])
    assert isinstance(raw["stats"]["best_result"], np.float64) and float(raw["stats"]["best_result"]) == 31.5
    assert "optimizer" in safe_load(paths[1]) and "ema" in safe_load(paths[1])
    models = load_objects(paths, lambda: NeRFNetwork(bound=1), torch.device("cpu"))
    assert len(models) == 3 and all(not m.training for m in models)
    for n, m in zip(nets, models):
        for k, v in n.state_dict().items():
            assert torch.equal(v, m.state_dict()[k]), k


def test_files_that_need_code_are_refused(tmp_path):
    from focnerf_amd.checkpoint import safe_load

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    p = tmp_path / "evil.pth"
    torch.save({"model": {}, "stats": Evil()}, str(p))
    with pytest.raises(RuntimeError, match="refusing to unpickle"):
        safe_load(str(p))
