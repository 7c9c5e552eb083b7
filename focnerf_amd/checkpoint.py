"""Checkpoint I/O in the reference trainer's `.pth` layout (nerf/utils.py:1431-1470 save, :1494-1530 load), so per-object
checkpoints written by FOC's trainer drop into these networks and the combiner.

    {'epoch', 'global_step', 'stats', ['mean_count', 'mean_density'], ['optimizer', 'lr_scheduler', 'scaler', 'ema'], 'model': state_dict}

`model` keys are the module paths both code bases share: `encoder.embeddings`, `encoder.offsets`, `sigma_net.weights`,
`color_net.weights`, `aabb_train`, `aabb_infer`, and for occupancy-grid models `density_grid` (absent from "best" checkpoints,
utils.py:1484-1485), `density_bitfield`, `step_counter`.

Loading uses `torch.load(weights_only=True)`: nothing in the file is executed. The reference's `stats['results']` and
`stats['best_result']` hold numpy float64 scalars (PSNRMeter.measure returns `V / N` built with np.log10, utils.py:563-575, appended at
:1413; best checkpoints — the ones COMBINED.py consumes — are only written when `results` is non-empty, :1472), which the weights-only
unpickler refuses by default: the numpy *scalar reconstructors* (data only: `numpy.core.multiarray.scalar` under either of numpy's module
names, `numpy.dtype` and the dtype classes) are allow-listed for the duration of the load. A file that still does not load is reported,
never unpickled.
"""
import torch


def _numpy_scalar_globals():
    import numpy
    try:
        from numpy._core.multiarray import scalar            # numpy >= 2
    except ImportError:                                       # pragma: no cover
        from numpy.core.multiarray import scalar
    dtypes = [type(numpy.dtype(t)) for t in ("float64", "float32", "float16", "int64", "int32", "int16", "int8", "uint8", "bool")]
    # a checkpoint written under numpy 1.x names the reconstructor numpy.core.multiarray.scalar, one written under 2.x numpy._core...
    return [scalar, (scalar, "numpy.core.multiarray.scalar"), (scalar, "numpy._core.multiarray.scalar"), numpy.dtype] + dtypes


def safe_load(path, map_location=None):
    """torch.load(weights_only=True) that also accepts numpy scalars (see the module docstring). Raises RuntimeError naming the
    globals the file would need when it holds anything else."""
    try:
        with torch.serialization.safe_globals(_numpy_scalar_globals()):
            return torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:
        try:
            need = torch.serialization.get_unsafe_globals_in_checkpoint(path)
        except Exception:
            need = "unknown"
        raise RuntimeError(f"focnerf_amd.checkpoint: {path} is not loadable without executing code from the file (globals it asks for: {need}); "
                           f"refusing to unpickle it") from e


def save_checkpoint(model, path, epoch=0, global_step=0, stats=None, best=False, extra=None):
    """`extra`: further top-level entries (the reference trainer adds 'optimizer', 'lr_scheduler', 'scaler', 'ema' to full checkpoints)."""
    state = {'epoch': epoch, 'global_step': global_step, 'stats': stats if stats is not None else {}}
    if getattr(model, 'cuda_ray', False):
        state['mean_count'] = model.mean_count
        state['mean_density'] = model.mean_density
    if extra:
        state.update(extra)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    if best and 'density_grid' in sd:
        del sd['density_grid']
    state['model'] = sd
    torch.save(state, path)


def load_checkpoint(model, path, map_location=None):
    """Returns (missing_keys, unexpected_keys) like the reference logs them; restores mean_count / mean_density when present."""
    ckpt = safe_load(path, map_location=map_location)
    sd = ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if getattr(model, 'cuda_ray', False) and isinstance(ckpt, dict):
        if 'mean_count' in ckpt:
            model.mean_count = ckpt['mean_count']
        if 'mean_density' in ckpt:
            model.mean_density = ckpt['mean_density']
    return list(missing), list(unexpected)


def load_objects(paths, build_model, device):
    """COMBINED.py:592-618 re-reads one checkpoint per object per view (`self.load_checkpoint(self.ckpt)` inside the view loop); here
    every object's network is loaded once and stays resident (K x ~100 MB of parameters). `build_model()` -> a fresh network; returns
    the list of eval-mode models in checkpoint order (the order decides ties in the per-sample select).

    `device` must be the process's current GPU: one process drives one GPU (bench.py, the RCCL combine). Objects on OTHER GPUs of the
    same process are served by the ops (they run with their tensors' device current, `_lib._on_tensor_device`, the extension shims'
    FocStream) but that path has not run on a multi-GPU box yet (tests/test_gpu_edge_cases.py::test_ops_follow_their_tensors_device is
    skipped on one GPU), so it has to be asked for: FOC_ALLOW_CROSS_DEVICE=1."""
    import os
    dev = torch.device(device)
    if dev.type == "cuda" and torch.cuda.is_available():
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if idx != torch.cuda.current_device() and os.environ.get("FOC_ALLOW_CROSS_DEVICE") != "1":
            raise RuntimeError(f"load_objects: device cuda:{idx} is not the current device (cuda:{torch.cuda.current_device()}); one process drives one "
                               "GPU. Make it current (torch.cuda.set_device) or set FOC_ALLOW_CROSS_DEVICE=1 (unverified on multi-GPU hardware)")
    models = []
    for p in paths:
        m = build_model().to(device)
        load_checkpoint(m, p, map_location=device)
        models.append(m.eval())
    return models
