"""Checkpoint I/O in the reference trainer's `.pth` layout (nerf/utils.py:1431-1470 save, :1494-1530 load), so per-object
checkpoints written by FOC's trainer drop into these networks and the combiner.

    {'epoch', 'global_step', 'stats', ['mean_count', 'mean_density'], 'model': state_dict, [...optimizer state...]}

`model` keys are the module paths both code bases share: `encoder.embeddings`, `encoder.offsets`, `sigma_net.weights`,
`color_net.weights`, `aabb_train`, `aabb_infer`, and for occupancy-grid models `density_grid` (absent from "best" checkpoints,
utils.py:1484-1485), `density_bitfield`, `step_counter`.

Loading uses `torch.load(weights_only=True)`: nothing in the file is executed. The reference saves `stats` with plain Python
containers, which that loader accepts; a file it refuses is reported, not unpickled.
"""
import torch


def save_checkpoint(model, path, epoch=0, global_step=0, stats=None, best=False):
    state = {'epoch': epoch, 'global_step': global_step, 'stats': stats if stats is not None else {}}
    if getattr(model, 'cuda_ray', False):
        state['mean_count'] = model.mean_count
        state['mean_density'] = model.mean_density
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    if best and 'density_grid' in sd:
        del sd['density_grid']
    state['model'] = sd
    torch.save(state, path)


def load_checkpoint(model, path, map_location=None):
    """Returns (missing_keys, unexpected_keys) like the reference logs them; restores mean_count / mean_density when present."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    sd = ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if getattr(model, 'cuda_ray', False) and isinstance(ckpt, dict):
        if 'mean_count' in ckpt:
            model.mean_count = ckpt['mean_count']
        if 'mean_density' in ckpt:
            model.mean_density = ckpt['mean_density']
    return list(missing), list(unexpected)


def load_objects(paths, build_model, device):
    """COMBINED.py:592-618 re-reads one checkpoint per object per view; here every object's network is loaded once and stays
    resident (K x ~100 MB of parameters). `build_model()` -> a fresh network; returns the list of eval-mode models."""
    models = []
    for p in paths:
        m = build_model().to(device)
        load_checkpoint(m, p, map_location=device)
        models.append(m.eval())
    return models
