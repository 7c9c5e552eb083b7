#!/usr/bin/env python
"""bench.py — headline benchmark of the FOCNeRF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU, one OBJECT per rank (weak scaling). Either launched by torch.distributed.run (RANK / WORLD_SIZE in the
environment), or — `python bench.py --gpus N` on its own — this process starts the N ranks itself (torch.distributed.run as a CHILD
process, before anything here touches the GPU) and exits with their status.

Workload (BASELINE.json configs[1]): single-object hash-grid + fused-MLP NeRF, fp16 autocast, rays
drawn from synthetic 800x800 views, FOC's default fixed-step renderer (num_steps=512,
upsample_steps=0, nerf/renderer.py:126-238): one STEP = 4096 rays x 512 samples = 2 097 152 samples
through R1 -> G1 -> M1(sigma) -> weights -> M1(colour, masked) -> composite -> loss -> backward
(M2, G2) -> Adam. `value` = samples completed per second, whole job (sum over ranks), inputs resident
in HBM. Extra keys: `combined_render` (EVERY N: K = N objects, one per rank, full 800x800 view — per-rank field evaluation + exchange by
ray + select/composite + gather, COMBINED.py:592-618 — rays/s, bytes on the wire, share of xGMI and of the field evaluation),
`render` (full 800x800 view, rays/s), `occupancy_path` (config[2]: march + composite kernels), `roofline` (dominant kernel, timed with
events on the launch stream), `cpu_baseline` (rank 0 / N=1 only: the CPU oracle port on a bounded sample, and configs[0] — the
reference's pure-PyTorch network through the fixed-step renderer on the host cores).

`--dry-run-cpu`: no GPU — gloo ranks exercising the launch and the combined-render leg's host/collective logic with CPU ops injected
from the tests; the training metric is null there (the product has no CPU path).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

NUM_RAYS = 4096          # main_nerf.py:28
NUM_STEPS = 512          # main_nerf.py:31
VIEW = 800
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0

# algorithmic bytes per sample (fp16 table/activations, fp32 coords)
ALGO_BYTES = {
    "grid_encode_forward": 588.0,
    "grid_encode_backward": 588.0,
    "ffmlp_forward": None,    # per network, filled below from its shape
    "ffmlp_backward": None,
}


# C-ABI op -> the HIP kernels it launches (names as rocprofv3 prints them)
OP_KERNELS = {
    "grid_encode_backward": ["k_gbin_count", "k_gbin_scans", "k_gbin_scatter", "k_gbin_reduce"],
    "grid_encode_forward": ["k_grid_fwd_bl", "k_grid_fwd_lbc", "k_grid_fwd_counted"],
    "grid_encode_forward_counted": ["k_grid_fwd_counted", "k_gbin_scans"],
    "ffmlp_forward": ["k_mlp_fwd"],
    "ffmlp_inference": ["k_mlp_fwd"],
    "ffmlp_backward": ["k_mlp_bwd", "k_mlp_dw", "k_mlp_dw_finalize"],
}


def pmc_traffic_bytes(op, which="bench", kernels=None):
    """HBM bytes per launch of `op` from the newest committed PMC summary of the `which` run (profiles/rNN_<which>_pmc_hbm.csv, collected
    in separate rocprofv3 --pmc passes), or None."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"*_{which}_pmc_hbm.csv")))
    if not files:
        return None
    total, found = 0.0, False
    for row in csv.DictReader(open(files[-1])):
        if any(k in row["kernel"] for k in (kernels or OP_KERNELS.get(op, []))):
            total += float(row["hbm_bytes_per_launch"])
            found = True
    return total if found else None


def profile_occupancy_kernel_us_per_step():
    """Library / torch kernel time per configs[2] training step from the newest committed profiles/*_occupancy_train_kernel_stats.csv
    (rocprofv3 --kernel-trace --stats of tools/prof_occupancy.py: 37 steps, one k_march_count_wave launch each), or None."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_occupancy_train_kernel_stats.csv")))
    if not files:
        return None
    try:
        rows = list(csv.DictReader(open(files[-1])))
        steps = max(int(r["Calls"]) for r in rows if "k_march_count" in r["Name"])
        is_lib = lambda n: re.search(r"\bk_[a-z]|_Z\d+k_", n) is not None
        lib = sum(float(r["TotalDurationNs"]) for r in rows if is_lib(r["Name"]))
        launches = sum(int(r["Calls"]) for r in rows if is_lib(r["Name"]))
        rest = sum(float(r["TotalDurationNs"]) for r in rows) - lib
        return {"profile_library_kernels_us_per_step": lib / steps / 1e3, "profile_library_launches_per_step": launches / steps,
                "profile_torch_native_us_per_step": rest / steps / 1e3, "profile_source": os.path.basename(files[-1])}
    except Exception:
        return None


def profile_kernel_ms_per_step():
    """{library kernels ms/step, torch-native kernels ms/step, source file} from the newest committed profiles/*_bench_kernel_stats.csv
    (rocprofv3 --kernel-trace --stats of `bench.py --no-extras`), or None. A step launches k_gbin_reduce exactly once: its call count is
    the number of steps in the profiled run (initialisation and warm-up steps included — they are the same step)."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_bench_kernel_stats.csv")))
    if not files:
        return None
    try:
        rows = list(csv.DictReader(open(files[-1])))
        steps = max(int(r["Calls"]) for r in rows if "k_gbin_reduce" in r["Name"])
        lib = sum(float(r["TotalDurationNs"]) for r in rows if re.search(r"\bk_[a-z]|_Z\d+k_", r["Name"]))
        rest = sum(float(r["TotalDurationNs"]) for r in rows) - lib
        return {"profile_library_kernels_ms_per_step": lib / steps / 1e6, "profile_torch_native_ms_per_step": rest / steps / 1e6,
                "profile_source": os.path.basename(files[-1]), "profile_steps": steps}
    except Exception:
        return None


def mlp_bytes_per_row(input_dim, hidden, num_layers, train, backward=False):
    # forward (train): read input, write every hidden activation, write 16 outputs   (fp16)
    fwd = 2 * input_dim + (2 * hidden * num_layers if train else 0) + 32
    if not backward:
        return float(fwd)
    # backward: read grad 32, forward activations (mask + dW) 2x, write + re-read backward buffers, inputs, grad_inputs
    return float(32 + 2 * (2 * hidden * num_layers) + 2 * (2 * hidden * num_layers) + 2 * input_dim + 2 * input_dim)


def mlp_flops_per_row(input_dim, hidden, num_layers):
    return 2.0 * (input_dim * hidden + (num_layers - 1) * hidden * hidden + hidden * 16)


def step_op_table():
    """C-ABI entry points of the headline step -> (HIP kernels as rocprofv3 prints them, what bounds them, ALGORITHMIC bytes per sample
    (SURVEY.md §8d / DESIGN.md §4), USEFUL flops per sample: the forward products once, the backward's dX and dW products — the re-evaluated
    forward inside the fused backward is not counted)."""
    fs, fc = mlp_flops_per_row(32, 64, 2), mlp_flops_per_row(32, 64, 3)
    bwd_kernels = ["k_mlp_bwd_fused", "k_mlp_dw_reduce"]
    return {
        "foc_fixed_sample": (["k_fs_sample"], "hbm", 12.0, None),
        "foc_near_far_from_aabb": (["k_near_far_from_aabb"], "hbm", 32.0 / NUM_STEPS, None),
        "foc_grid_encode_forward_counted": (["k_grid_fwd_counted", "k_gbin_scans"], "hbm", 588.0, None),
        "foc_grid_encode_forward": (["k_grid_fwd_lbc"], "hbm", 588.0, None),
        "foc_ffmlp_forward_planar": (["k_mlp_fwd"], "mfma", 64.0 + 32.0, fs),
        "foc_color_head_forward": (["k_mlp_fwd"], "mfma", 32.0 + 8.0, fc),
        "foc_field_forward_train": (["k_field_fwd_train"], "mfma", 64.0 + 32.0 + 8.0, fs + fc),      # both networks' training forward in one kernel
        "foc_fixed_tail_forward": (["k_fs_tail_fwd"], "hbm", 40.0 + 12.0, None),
        "foc_fixed_tail_backward": (["k_fs_tail_bwd"], "hbm", 28.0 + 34.0, None),
        "foc_color_head_backward": (bwd_kernels, "mfma", 8.0 + 32.0 + 32.0, 2.0 * fc),
        "foc_ffmlp_backward_planar": (bwd_kernels, "mfma", 32.0 + 64.0 + 64.0, 2.0 * fs),
        "foc_grid_encode_backward_binned_counted": (["k_gbin_scatter", "k_gbin_reduce"], "hbm", 588.0, None),
        "foc_grid_encode_backward_binned": (["k_gbin_count", "k_gbin_scans", "k_gbin_scatter", "k_gbin_reduce"], "hbm", 588.0, None),
    }


def pmc_mfma_utilisation(patterns):
    """{kernel: matrix-pipe utilisation} of the kernels whose (mangled) name contains one of `patterns`, from the newest committed
    profiles/*_bench_pmc_mfma.csv (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... in its own pass), or None. A figure from a committed file, not from
    this run: the file name travels with it."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_bench_pmc_mfma.csv")))
    if not files:
        return None
    out = {}
    for row in csv.DictReader(open(files[-1])):
        if any(pat in row["kernel"] for pat in patterns):
            out[row["kernel"][:64]] = float(row["mfma_pipe_utilisation"])
    return {"source": os.path.basename(files[-1]), "by_kernel": out} if out else None


class KernelTimer:
    """Event pairs around the C-ABI calls of focnerf_amd.backend, on torch's current stream (the stream the kernels launch on)."""

    def __init__(self):
        self.records = {}
        self.enabled = False
        self._orig = {}

    def install(self):
        from focnerf_amd import backend
        targets = [(backend._gridencoder, "grid_encode_forward"), (backend._gridencoder, "grid_encode_forward_counted"),
                   (backend._gridencoder, "grid_encode_backward"),
                   (backend._ffmlp, "ffmlp_forward"), (backend._ffmlp, "ffmlp_inference"), (backend._ffmlp, "ffmlp_backward"),
                   (backend._raymarching, "march_rays_train"), (backend._raymarching, "composite_rays_train_forward"),
                   (backend._raymarching, "composite_rays_train_backward"), (backend._raymarching, "near_far_from_aabb")]
        for cls, name in targets:
            orig = getattr(cls, name)
            self._orig[(cls, name)] = orig

            def make(orig, name):
                def wrapped(*a, **k):
                    if not self.enabled:
                        return orig(*a, **k)
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    r = orig(*a, **k)
                    e.record()
                    units = a[4] if name.startswith("grid_encode_forward") else None
                    if name == "grid_encode_backward":
                        units = a[5]
                    elif name.startswith("ffmlp"):
                        units = a[2] if name != "ffmlp_backward" else a[4]
                    self.records.setdefault(name, []).append((s, e, units))      # no tensor references: they would pin every step's buffers
                    if name == "grid_encode_forward_counted":
                        # shapes and scalars only (count_share_ms re-creates the tensors afterwards): holding the step's encoder planes and
                        # fp16 table here kept 90 MB out of the caching allocator's reach and cost one device allocation in the timed region
                        self.last_counted_args = (tuple(a[0].shape), a[1].shape, a[1].dtype, a[2]) + tuple(a[4:13])
                    return r
                return staticmethod(wrapped)
            setattr(cls, name, make(orig, name))

    def count_share_ms(self, reps=5):
        """The backward's count pass rides in the training forward launch (k_grid_fwd_counted): its cost there = that launch minus the
        plain forward on the same inputs, both timed here, outside the timed region."""
        a = getattr(self, "last_counted_args", None)
        if a is None:
            return None
        from focnerf_amd import backend
        plain = self._orig[(backend._gridencoder, "grid_encode_forward")]
        counted = self._orig[(backend._gridencoder, "grid_encode_forward_counted")]
        in_shape, emb_shape, emb_dtype, offsets, B, D, C, L, S, H, gridtype, ac, interp = a
        dev = offsets.device
        # ray-ordered positions like the step's own (4096 rays x 512 samples through the unit cube), a table of the same shape
        t = torch.linspace(0.02, 0.98, NUM_STEPS, device=dev)
        p0, p1 = torch.rand(B // NUM_STEPS + 1, 1, 3, device=dev), torch.rand(B // NUM_STEPS + 1, 1, 3, device=dev)
        inputs = (p0 + (p1 - p0) * t[None, :, None]).reshape(-1, 3)[:B].contiguous()
        emb = (torch.rand(emb_shape, device=dev) - 0.5).to(emb_dtype)
        outputs = torch.empty(L, B, C, device=dev, dtype=emb_dtype)

        def timed(fn):
            fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / reps
        t_plain = timed(lambda: plain(inputs, emb, offsets, outputs, B, D, C, L, S, H, None, gridtype, ac, interp))
        t_counted = timed(lambda: counted(inputs, emb, offsets, outputs, B, D, C, L, S, H, gridtype, ac, interp))
        backend._gridencoder._invalidate_precount(inputs.device)
        self.last_counted_args = None
        return {"plain_forward_ms": t_plain, "counted_forward_ms": t_counted, "count_share_ms": max(0.0, t_counted - t_plain)}

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _ in recs]
            units = [u for _, _, u in recs if u is not None]
            out[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                         "avg_units": (sum(units) / len(units)) if units else None}
        return out


class LibTimer:
    """Event pairs around chosen C-ABI entry points (attributes of focnerf_amd._lib.lib), recorded on torch's current stream — the stream
    the entry point launches on. `with LibTimer([...]) as t: ...; t.summary()` -> {name: (launches, avg_ms)}."""

    def __init__(self, names):
        self.names, self.records, self._orig = names, {}, {}

    def __enter__(self):
        from focnerf_amd._lib import lib
        for name in self.names:
            orig = getattr(lib, name)
            self._orig[name] = orig

            def wrapped(*a, _orig=orig, _name=name):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = _orig(*a)
                e.record()
                self.records.setdefault(_name, []).append((s, e))
                return r
            setattr(lib, name, wrapped)
        return self

    def __exit__(self, *exc):
        from focnerf_amd._lib import lib
        for name, orig in self._orig.items():
            setattr(lib, name, orig)
        return False

    def summary(self):
        torch.cuda.synchronize()
        return {n: (len(r), sum(s.elapsed_time(e) for s, e in r) / len(r)) for n, r in self.records.items()}


# C-ABI entry point -> (kernel name as rocprofv3 prints it, algorithmic bytes per sample: SURVEY.md §8d / DESIGN.md §4)
RENDER_OPS = {
    "foc_grid_encode_forward": ("k_grid_fwd_lbc", 588.0),            # 12 r + 512 gathered + 64 w
    "foc_nerf_field_inference": ("k_nerf_infer", 64.0 + 16.0),       # planes 64 r; sigma 4 + rgb 12 w (directions: per ray)
    "foc_fixed_render_inference": ("k_fs_render_infer_blk", 16.0),   # sigma + rgb r (image / depth: per ray)
    "foc_fixed_sample": ("k_fs_sample", 12.0),                       # positions w
}


def render_roofline(model, view_rays, rkw, n_views=2):
    """Per-kernel launch durations of the fixed-step render with every chunk on ONE stream (FOC_RENDER_STREAMS=1: with the default two
    streams the kernels of neighbouring chunks overlap and an event pair sees both), and the roofline entry of the dominant one."""
    old = os.environ.get("FOC_RENDER_STREAMS")
    os.environ["FOC_RENDER_STREAMS"] = "1"
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            model.render(*view_rays[0], return_fields=False, **rkw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with LibTimer(list(RENDER_OPS)) as lt:
                for i in range(n_views):
                    model.render(*view_rays[i % len(view_rays)], return_fields=False, **rkw)
                ks = lt.summary()
            one_stream_s = (time.perf_counter() - t0) / n_views
    finally:
        if old is None:
            os.environ.pop("FOC_RENDER_STREAMS", None)
        else:
            os.environ["FOC_RENDER_STREAMS"] = old
    if not ks:
        return None
    chunks = max(v[0] for v in ks.values()) / n_views
    units = VIEW * VIEW * NUM_STEPS / chunks                        # samples per launch (the last chunk of a view is shorter)
    dom = max(ks, key=lambda k: ks[k][0] * ks[k][1])
    kern, bpu = RENDER_OPS[dom]
    achieved = bpu * units / (ks[dom][1] * 1e-3) / 1e9
    total = sum(v[0] * v[1] for v in ks.values())
    traffic = pmc_traffic_bytes(dom, which="render_fixed", kernels=[kern])
    hbm_real = (traffic / (ks[dom][1] * 1e-3) / 1e9) if traffic else None
    return {"kernel": f"{dom} = {kern}", "bound": "cache-request (L2 / Infinity-Cache gathers; HBM carries `hbm_real_frac`)", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "hbm_real_GBps": hbm_real, "hbm_real_frac": (hbm_real / HBM_PEAK_GBS) if hbm_real else None,
            "traffic": traffic, "avg_launch_ms": ks[dom][1], "units_per_launch": units,
            "algorithmic_bytes_per_unit": bpu, "share_of_render_kernel_time": ks[dom][0] * ks[dom][1] / total,
            "kernels_ms_per_chunk": {RENDER_OPS[k][0]: round(v[1], 4) for k, v in ks.items()}, "one_stream_s_per_view": one_stream_s,
            "note": "events on the launch stream around the C-ABI call, every chunk on one stream (FOC_RENDER_STREAMS=1); the rays/s figure is the "
                    "default two-stream render. `frac` prices the ALGORITHMIC 588 B per sample against the HBM peak as the contract asks, but the hash "
                    "tables (24 MB fp16) are cache resident: the 512 gathered bytes per sample are L2 / Infinity-Cache requests, not HBM traffic. What "
                    "reaches HBM is `traffic` (profiles/*_render_fixed_pmc_hbm.csv; the PMC run's piece size may differ from this run's: the real "
                    "fraction is quoted per launch of that file's size when they agree) = `hbm_real_frac` of the peak"}


def build_model(bound, device, cuda_ray=False, seed=0):
    from focnerf_amd.network import NeRFNetwork
    from focnerf_amd import synthetic
    torch.manual_seed(seed)
    model = NeRFNetwork(bound=bound, cuda_ray=cuda_ray).to(device)
    if cuda_ray:
        model.set_density_grid(synthetic.analytic_density_grid(bound, device=device))
    return model


def build_foc_model(bound, device, seed=0):
    """FOC's object-conditioned network (nerf/network_tcnn.py:451-681 topology: 48-wide colour input [SH16 | geo15 | object feature 16 | 0]) —
    the network main_nerf.py:108 and COMBINED.py:84 actually construct."""
    from focnerf_amd.network_foc import NeRFNetwork
    torch.manual_seed(seed)
    return NeRFNetwork(bound=bound).to(device)


def foc_yolo_details(device, n_rays, seed):
    """Synthetic `yolo_details` = (object mask of the batch's rays [1,N] bool, bbox (unused on the path), raw object feature [144] fp32) —
    what nerf/provider.py hands the trainer per image (utils.py:818-823)."""
    g = torch.Generator().manual_seed(seed)
    mask = (torch.rand(1, n_rays, generator=g) < 0.7).to(device)
    feat = torch.randn(144, generator=g).to(device)          # resident like the rays (the reference's --preload): no H2D copy inside a step
    return (mask, None, feat)


def foc_train_step(model, opt, scaler, rays_o, rays_d, target, yolo, sched=None):
    """nerf/utils.py:818-902 (train_step of the FOC trainer): render with yolo_details, MSE + 1e-8 * outside-mask density criterion."""
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        out = model.render(rays_o, rays_d, yolo, staged=False, num_steps=NUM_STEPS, upsample_steps=0, perturb=True, bg_color=None, fused=True)
        loss = torch.nn.functional.mse_loss(out["image"], target)
        if out.get("criterion_outside_mask") is not None:
            loss = loss + 1e-8 * out["criterion_outside_mask"]
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    if sched is not None:
        sched.step()
    return loss


def make_training_rays(device, bound, n_views, seed):
    from focnerf_amd import synthetic
    g = torch.Generator().manual_seed(seed)
    poses = synthetic.rand_poses(n_views, device, radius=2.0, generator=g)
    intr = synthetic.intrinsics(VIEW, VIEW)
    return poses, intr


def sample_batch(poses, intr, device, gen):
    from focnerf_amd import synthetic
    v = int(torch.randint(0, poses.shape[0], (1,), generator=gen).item())
    inds = torch.randint(0, VIEW * VIEW, (1, NUM_RAYS), generator=gen).to(device)
    if os.environ.get("FOC_BENCH_SORT_RAYS") == "1":        # experiment only: the same kind of batch with its rays in pixel order
        inds = inds.sort(dim=1).values
    rays_o, rays_d = synthetic.get_rays(poses[v:v + 1], intr, VIEW, VIEW, inds)
    # synthetic target: colour of the analytic scene does not matter for throughput; use a smooth function of the ray
    target = 0.5 + 0.5 * torch.sin(3.0 * rays_d)
    return rays_o, rays_d, target


def train_step(model, opt, scaler, rays_o, rays_d, target, fused=True, sched=None):
    """One iteration of the reference trainer's loop (nerf/utils.py:1108-1119): zero_grad, autocast forward + loss, scaled backward,
    GradScaler step + update, per-step LR schedule (its EMA update runs once per epoch, :1124, not per step)."""
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        out = model.render(rays_o, rays_d, staged=False, num_steps=NUM_STEPS, upsample_steps=0, perturb=True, bg_color=None, fused=fused)
        loss = torch.nn.functional.mse_loss(out["image"], target)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    if sched is not None:
        sched.step()
    return loss


def cuda_ray_train_step(model, opt, scaler, rays_o, rays_d, target):
    with torch.autocast("cuda", dtype=torch.float16):
        out = model.render(rays_o, rays_d, staged=False, perturb=True, force_all_rays=False, dt_gamma=1 / 128, max_steps=1024, bg_color=None)
        loss = torch.nn.functional.mse_loss(out["image"], target)
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    return loss


def cpu_baseline(budget_s=10.0, one_core_s=4.0):
    """The CPU oracle port of the same step's kernels (G1 -> M1 sigma -> M1 colour -> fixed-step composite, forward and backward)
    on a bounded sample: as many 512-sample rays as the host cores available to this process get through in ~budget_s seconds
    (one worker thread per core, up to 16; the C calls release the GIL), after a short single-core run for reference."""
    import numpy as np
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    from focnerf_amd.gridencoder import level_offsets
    pls = np.exp2(np.log2(2048 / 16) / 15)
    S = float(np.log2(pls))
    off = level_offsets(3, 16, pls, 16, 19)
    rng0 = np.random.default_rng(0)
    table = rng0.uniform(-1, 1, (int(off[-1]), 2)).astype(np.float16)
    Ws = (rng0.uniform(-1, 1, 64 * (32 + 64 + 16)) * 0.2).astype(np.float16)
    Wc = (rng0.uniform(-1, 1, 64 * (32 + 128 + 16)) * 0.2).astype(np.float16)
    rays_per_chunk = 8
    B = rays_per_chunk * NUM_STEPS            # 4096 samples, a multiple of 128
    oracle.lib()                              # build / load once, before the threads start

    def worker(seed, seconds):
        rng = np.random.default_rng(seed)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            x = rng.random((B, 3)).astype(np.float32)
            enc = oracle.grid_encode_forward(x, table, off, 3, 2, 16, S, 16)
            enc_bl = np.ascontiguousarray(np.transpose(enc, (1, 0, 2)).reshape(B, 32))
            h, fb_s = oracle.ffmlp_forward(enc_bl, Ws, 32, 64, 2, 0)
            sigma = np.exp(h[:, 0].astype(np.float32))
            cin = np.concatenate([h, h], 1).astype(np.float16)
            c, fb_c = oracle.ffmlp_forward(cin, Wc, 32, 64, 3, 0)
            rgb = 1 / (1 + np.exp(-c[:, :3].astype(np.float32)))
            nears = np.full(rays_per_chunk, 0.5, np.float32); fars = np.full(rays_per_chunk, 2.5, np.float32)
            oracle.composite_fixed_steps(sigma.reshape(rays_per_chunk, NUM_STEPS), rgb.reshape(rays_per_chunk, NUM_STEPS, 3), nears, fars, 1.0)
            g = (rng.standard_normal((B, 16)) * 0.01).astype(np.float16)
            gw_c, gi_c, _ = oracle.ffmlp_backward(g, cin, Wc, fb_c, 32, 64, 3, 0, True)
            gw_s, gi_s, _ = oracle.ffmlp_backward(gi_c[:, :16].copy(), enc_bl, Ws, fb_s, 32, 64, 2, 0, True)
            genc = np.ascontiguousarray(np.transpose(gi_s.reshape(B, 16, 2), (1, 0, 2)))
            oracle.grid_encode_backward(genc, x, off, int(off[-1]), 3, 2, 16, S, 16)
            done += B
        return done, time.perf_counter() - t0

    d1, e1 = worker(1, one_core_s)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        res = list(pool.map(lambda sd: worker(sd, budget_s), range(100, 100 + cores)))
    el = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    return {"value": done / el, "unit": "samples/s", "cores": cores, "kind": "port", "one_core_value": d1 / e1,
            "sample": f"{done} samples ({done // NUM_STEPS} rays x {NUM_STEPS}) of the same step's kernels "
                      f"(hash-grid fwd/bwd, both fused MLPs fwd/bwd, fixed-step composite), scalar C oracle on {cores} threads, {el:.1f} s "
                      f"(+ {e1:.1f} s single-core run)"}


def progress(msg):
    """Progress marker on stderr (rank 0): the one JSON line comes last, and a long silent run looks hung to whoever is watching.
    Also where Python's cycle collector runs: it is switched off for the measurements (main(), timeit's convention) and called here, between
    the legs — a generation-2 pass over this process's heap inside a timed loop is a 50-100 ms pause of the ENQUEUEING thread (measured: the
    eight timed views of the render leg 42 ms each with the collector off or after three warm-up views, 51-62 with it on and one)."""
    import gc
    gc.collect()
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child `python -m torch.distributed.run` (one rank per GPU over
    RCCL) and return its exit status. Nothing in THIS process has touched the GPU (no HIP call, no torch.cuda query) — the ranks are
    fresh processes, never an exec of a process that initialised a device."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def resident_object_fields(device, bound, vo, vd, K, first=None, rank=0):
    """K objects RESIDENT on one device, as the field functions `ObjectCombiner.render_view` takes (`fn(lo, hi, out)` -> packed field4 of the
    object on rays lo:hi of the view vo / vd): FOC object-conditioned networks with their own object features, seeded — the same objects
    whenever this is called with the same arguments (bench leg `resident_4_objects_one_gpu`; tests/test_gpu_combine.py walks the same view)."""
    from focnerf_amd.fixedstep import render_field4
    if first is None:
        first = (build_foc_model(bound, device, seed=rank).eval(), foc_yolo_details(device, 1, 50 + rank))
    objs = [first[0]] + [build_foc_model(bound, device, seed=100 + k).eval() for k in range(K - 1)]
    yolos = [first[1]] + [foc_yolo_details(device, 1, 60 + k) for k in range(K - 1)]
    return [(lambda lo, hi, out, m=m, y=y: render_field4(m, vo[lo:hi], vd[lo:hi], num_steps=NUM_STEPS, yolo_details=y, out=out)) for m, y in zip(objs, yolos)]


def combined_render_leg(rank, world, device, model, views, fused_fn, barrier, max_over_ranks, chunk=16384, ops=None, n_side=VIEW, T=NUM_STEPS, overlap=True,
                        ray_order=None, collectives_at_world_1=False):
    """COMBINED.py:592-618 on one object per rank: every rank evaluates ITS object on all rays of an n_side^2 view (16384-ray pieces, packed
    per-sample fields), the chunks are exchanged by ray (all-to-all over xGMI), every rank selects + composites its ray slices for both
    backgrounds, one all-gather per view assembles the images. Timed against the same view with the field evaluation alone.
    `fused_fn(lo, hi, out)` = this rank's object; `ops` = None (HIP) or injected CPU ops (dry run)."""
    import torch.distributed as dist
    from focnerf_amd.combine import ObjectCombiner, HipCombineOps
    comb = ObjectCombiner(rank=rank, world_size=world, ops=ops or HipCombineOps, collectives_at_world_1=collectives_at_world_1)
    n_rays = n_side * n_side
    nears, fars = model["nears"], model["fars"]

    def one_view(ov=overlap):
        img, dep = comb.render_view([fused_fn], n_rays, nears, fars, T, bgs=(1.0, 0.0), max_ray_batch=chunk, overlap=ov)
        if ray_order is not None:          # the view was walked in pixel tiles (focnerf_amd/rayorder.py): rows back to the caller's ray order, inside the timed region
            img = torch.empty_like(img).index_copy_(1, ray_order, img)
            dep = torch.empty_like(dep).index_copy_(0, ray_order, dep)
        return img, dep

    def eval_only():
        buf = torch.empty(chunk, T, 4, dtype=torch.float32, device=device)
        for lo in range(0, n_rays, chunk):
            hi = min(lo + chunk, n_rays)
            fused_fn(lo, hi, buf[: hi - lo])

    img, dep = one_view()                                  # untimed: allocator block sizes, RCCL channel set-up
    barrier()
    t0 = time.perf_counter()
    for _ in range(views):
        img, dep = one_view()
    barrier()
    el = max_over_ranks(time.perf_counter() - t0) / views
    sent = comb.bytes_sent
    eval_only()
    barrier()
    t0 = time.perf_counter()
    for _ in range(views):
        eval_only()
    barrier()
    el_eval = max_over_ranks(time.perf_counter() - t0) / views
    links = max(world - 1, 1)
    el_other = None
    if comb.xch:                                           # the other exchange mode on the same box, for an A/B on real links
        one_view(not overlap)
        barrier()
        t0 = time.perf_counter()
        for _ in range(views):
            one_view(not overlap)
        barrier()
        el_other = max_over_ranks(time.perf_counter() - t0) / views
    out = {"metric": "combined_render_rays_per_sec", "value": n_rays / el, "unit": "rays/s", "objects": world, "object_rays_per_sec": world * n_rays / el,
           "world_size": dist.get_world_size() if dist.is_initialized() else 1, "backend": dist.get_backend() if dist.is_initialized() else None,
           "overlap": bool(overlap), "s_per_view_other_exchange_mode": el_other,
           "s_per_view": el, "views": views, "view": f"{n_side}x{n_side}", "samples_per_ray": T, "backgrounds": 2,
           "field_eval_only_s_per_view": el_eval, "exchange_and_composite_share_of_field_eval": (el - el_eval) / el_eval if el_eval > 0 else None,
           "bytes_sent_per_view_per_gpu": int(sent),
           "xgmi": {"achieved_GBps_per_gpu": sent / el / 1e9, "peak_GBps_per_gpu": links * 153.0 if world > 1 else None,
                    "frac_of_peak": (sent / el / 1e9) / (links * 153.0) if world > 1 else None,
                    "note": "bytes this rank put on the wire per view / wall time per view; peak = (N-1) point-to-point links x 153 GB/s"},
           "image_checksum": float(img.double().sum().item()),
           "path": "per rank: near/far -> sample -> hash-grid -> whole-field kernel -> own weights + mask + pack (16 B/sample); all-to-all by ray per "
                   "16384-ray piece (134 MB per object; 4096-ray pieces are 10 % slower: 44.4 against 40.4 ms per view at N = 1), overlapped with the next piece's evaluation; fused select + composite of the rank's ray slices, both backgrounds; "
                   "one all-gather per view (COMBINED.py:592-618, 141-200, 247-251)"}
    return out


class ExchangeGuard:
    """The N-rank exchange is the one part of this file in which a rank can wait for another one. If it makes no progress (a rank that
    failed and left the others inside a collective), every rank leaves on its own timer — before the process group's timeout would abort the
    job. Rank 0 still prints ONE line: the headline and what had been measured when the leg started, with the leg's `error` — from a
    snapshot serialised BEFORE the leg (the timer thread never walks a dictionary the main thread may be changing) — and every rank then
    exits NON-ZERO: a process that has touched the GPU and cannot finish its job must not report success to the launcher (it does not
    restart or re-exec anything; it exits)."""
    EXIT_CODE = 3

    def __init__(self, rank, world, timeout, result):
        self.rank, self.world, self.timeout = rank, world, timeout
        self.base_line = json.dumps(result)
        self.timer = None

    def start(self):
        if self.world > 1:
            import threading
            self.timer = threading.Timer(self.timeout, self.abandon, args=(f"no progress for {self.timeout} s in the {self.world}-rank exchange: leg abandoned",))
            self.timer.daemon = True
            self.timer.start()

    def cancel(self):
        if self.timer is not None:
            self.timer.cancel()

    def abandon(self, reason, trace=None):
        try:
            if self.rank == 0:
                try:
                    snap = json.loads(self.base_line)
                    snap["combined_render"] = {"error": reason, "world_size": self.world}
                    if trace:
                        snap["combined_render"]["trace"] = trace
                    line = json.dumps(snap)
                except Exception:
                    line = self.base_line
                print(line, flush=True)
            else:
                time.sleep(5.0)          # rank 0's line first: the launcher tears every rank down as soon as one of them has exited
        finally:
            os._exit(self.EXIT_CODE)


def dry_run_cpu(args):
    """CPU / gloo rehearsal of the N-rank launch and of the combined-render leg's host and collective logic (no GPU, no HIP kernel: the
    per-sample fields are synthetic arrays and the select / composite are the CPU ops the tests inject)."""
    import numpy as np
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("FOC_BENCH_FAIL_RANK") == str(rank):      # test hook: a rank that dies must fail the whole launch
        raise SystemExit(3)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_combine_gloo import CpuOps
    device = torch.device("cpu")
    n_side, T, chunk = 24, 32, 96
    rng = np.random.default_rng(7 + rank)
    n_rays = n_side * n_side
    f4 = torch.from_numpy(rng.random((n_rays, T, 4)).astype(np.float32))
    f4[..., 0] *= 20.0

    def fn(lo, hi, out):
        if os.environ.get("FOC_BENCH_STALL_RANK") == str(rank):      # test hook: a rank that never reaches the exchange (the others wait inside it)
            time.sleep(600)
        if out is not None:
            out.copy_(f4[lo:hi])
            return out
        return f4[lo:hi].clone()

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x
    geom = {"nears": torch.full((n_rays,), 0.5), "fars": torch.full((n_rays,), 2.5)}
    result = {"metric": "train_samples_per_sec", "value": None, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
              "config": {"workload": "DRY RUN on CPU/gloo: launch + combined-render collective logic only; the training step has no CPU path"}}
    guard = ExchangeGuard(rank, world, args.exchange_timeout, result)
    guard.start()
    try:
        result["combined_render"] = combined_render_leg(rank, world, device, geom, 1, fn, barrier, max_over_ranks, chunk=chunk, ops=CpuOps, n_side=n_side, T=T)
    except Exception as e:
        import traceback
        if world > 1:                                       # the other ranks may be inside a collective: no orderly shutdown of the group
            guard.abandon(repr(e), traceback.format_exc()[-800:])
        raise
    guard.cancel()
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--render-views", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the render / occupancy-path extras")
    ap.add_argument("--no-fused", action="store_true", help="headline step through the torch glue of NeRFRenderer.run instead of csrc/fixedstep.hip")
    ap.add_argument("--dry-run-cpu", action="store_true", help="no GPU: gloo ranks, combined-render leg on injected CPU ops (launch / collective logic only)")
    ap.add_argument("--combined-views", type=int, default=2)
    ap.add_argument("--exchange-timeout", type=float, default=150.0, help="N > 1: seconds the combined-render leg may take before every rank abandons it")
    ap.add_argument("--combined-no-overlap", action="store_true", help="combined-render leg: wait for each chunk's all-to-all before evaluating the next chunk")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.dry_run_cpu:
        return dry_run_cpu(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FOC_BENCH_SHARE_GPU") == "1":      # rehearsal of the N-rank path on a one-GPU box (every rank on device 0); never set by the driver
        local_rank = 0
    # FOC_BENCH_REHEARSE_RCCL=1 with ONE rank (never set by the driver): this file's N-rank branch — nccl process group, barrier, the MAX over
    # ranks, the combined leg with the combiner issuing its collectives, the teardown — runs through RCCL on a one-GPU box. Labelled in the line.
    rehearse = world == 1 and os.environ.get("FOC_BENCH_REHEARSE_RCCL") == "1"
    dist_on = world > 1 or rehearse
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            os.environ.setdefault("MASTER_PORT", "29751"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        import datetime
        # a rank that fails inside a collective leg must not leave the others waiting for the driver's kill: collectives time out
        if os.environ.get("FOC_BENCH_SHARE_GPU") == "1":  # RCCL refuses two ranks on one device: the rehearsal moves the same collectives over gloo
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=240))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=240))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist_on:
            import torch.distributed as dist
            t = torch.tensor([x], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    bound = 1
    model = build_model(bound, device, cuda_ray=False, seed=rank).train()
    opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
    import warnings
    warnings.filterwarnings("ignore", message="Detected call of `lr_scheduler.step\\(\\)` before")     # a GradScaler-skipped first step
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / 30000, 1))              # main_nerf.py:477
    scaler = torch.amp.GradScaler("cuda")
    poses, intr = make_training_rays(device, bound, 8, seed=rank)
    gen = torch.Generator().manual_seed(1000 + rank)
    # one distinct ray batch per timed step (and per warm-up step), drawn before the timed region: data loading is not part of the path
    batches = [sample_batch(poses, intr, device, gen) for _ in range(max(4, min(args.steps + args.warmup, 256)))]

    timer = KernelTimer()
    timer.install()
    import gc
    gc.collect()
    gc.disable()                      # see progress(): collections happen between the legs, not inside a timed loop

    fused = not args.no_fused
    # Initialisation, before the W warm-up steps: the first steps create the persistent scratch buffers (2 GB of records), size the
    # caching allocator's blocks and bring the device out of its idle power state; on a freshly started process the first few hundred
    # milliseconds of work have been seen to run 3-10x slower than steady state. Bounded by wall time, not part of W or K.
    t_init = time.perf_counter()
    i = 0
    while i < 8 or (time.perf_counter() - t_init < 0.75 and i < 400):
        train_step(model, opt, scaler, *batches[i % len(batches)], fused=fused, sched=sched)
        i += 1
        if i % 8 == 0:
            torch.cuda.synchronize()
    init_steps = i
    for i in range(args.warmup):
        train_step(model, opt, scaler, *batches[i % len(batches)], fused=fused, sched=sched)
    barrier()
    timer.enabled = True
    step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    alloc0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    host_marks = [0.0] * (args.steps + 1)                  # when the host had enqueued step i (a spike in a step's GPU time with a host pause at the same
    t0 = time.perf_counter()                                 # step is the host's: the region starts from an empty queue, the host's lead grows by ~0.6 ms per step)
    host_marks[0] = t0
    step_marks[0].record()
    for i in range(args.steps):
        train_step(model, opt, scaler, *batches[(args.warmup + i) % len(batches)], fused=fused, sched=sched)
        step_marks[i + 1].record()
        host_marks[i + 1] = time.perf_counter()
    barrier()
    el = max_over_ranks(time.perf_counter() - t0)
    timer.enabled = False
    ksum = timer.summary()
    count_share = timer.count_share_ms()
    raw_steps = [step_marks[i].elapsed_time(step_marks[i + 1]) for i in range(args.steps)]
    host_steps = [1e3 * (host_marks[i + 1] - host_marks[i]) for i in range(args.steps)]
    per_step = sorted(raw_steps)
    step_stats = {"min": per_step[0], "median": per_step[len(per_step) // 2], "max": per_step[-1],
                  "slowest_step": max(range(args.steps), key=lambda i: raw_steps[i]) if args.steps else None,
                  "host_enqueue_ms": {"median": sorted(host_steps)[len(host_steps) // 2], "max": max(host_steps),
                                      "max_at_step": max(range(args.steps), key=lambda i: host_steps[i])} if args.steps else None,
                  "device_allocs_in_timed_region": torch.cuda.memory_stats(device).get("num_device_alloc", 0) - alloc0,
                  "init_steps_before_warmup": init_steps}

    samples_per_step = NUM_RAYS * NUM_STEPS
    value = world * samples_per_step * args.steps / el
    progress(f"headline: {1000.0 * el / args.steps:.3f} ms/step")
    # kernel-only step time (the optimizer is excluded or reported separately): a few more steps with an event pair around
    # EVERY C-ABI call of the library — what the step costs in this library's kernels, the rest being torch's own (fused Adam, GradScaler
    # unscale / inf check, fp16 <-> fp32 casts of parameters and gradients, fills, the loss)
    ks_all, n_k = {}, 5
    try:
        from focnerf_amd import _lib
        names = [n for n, (_, a) in _lib.SIGNATURES.items() if a and a[-1] is _lib.c_vp and not n.endswith("_bytes") and not n.endswith("_option")]
        with LibTimer(names) as lt:
            m0 = torch.cuda.Event(enable_timing=True)
            m1 = torch.cuda.Event(enable_timing=True)
            m0.record()
            for i in range(n_k):
                train_step(model, opt, scaler, *batches[i % len(batches)], fused=fused, sched=sched)
            m1.record()
            ks_all = lt.summary()
        lib_ms = sum(n * avg for n, avg in ks_all.values()) / n_k
        step_stats["library_kernels_ms_per_step"] = lib_ms
        step_stats["torch_native_ms_per_step"] = max(0.0, m0.elapsed_time(m1) / n_k - lib_ms)
        step_stats["kernel_only_samples_per_sec"] = samples_per_step / (lib_ms * 1e-3)
        step_stats["kernel_only_note"] = ("library_kernels = sum of event-timed C-ABI calls per step (every kernel of this library: sample, encoder fwd + "
                                          "count, both MLPs fwd/bwd, tail fwd/bwd, grid backward); torch_native = the step's remaining GPU time "
                                          "(Adam, GradScaler, casts, fills, loss), measured over 5 extra steps after the timed region")
    except Exception as e:
        step_stats["kernel_only_error"] = repr(e)
    # the same figure from the committed rocprofv3 kernel trace of this command (profiles/rNN_bench_kernel_stats.csv): sum over this library's
    # kernels of total duration / steps — no event pairs, no gaps between launches (the event-timed sum above moves by 10 % from run to run)
    prof = profile_kernel_ms_per_step()
    if prof is not None:
        step_stats["from_committed_profile"] = prof       # numbers parsed from a file of an EARLIER run of this command, not measured by this run
    result = {
        "metric": "train_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic", "step_ms": step_stats,
        "numerics": "fp16 storage, fp32 MFMA accumulation (the reference accumulates in fp16): sample indices / positions bit-exact, hash-grid forward "
                    "bit-exact vs the oracle, composites 1e-4. END TO END against the oracle chain in the reference-literal half accumulation (measured on "
                    "MI355X, pinned by tests/test_gpu_network.py::test_end_to_end_distance_to_reference_literal_numerics; BASELINE field, seed-1 U(-1,1) table): "
                    "|dRGB| max / p99.9 = 4.88e-4 / 4.88e-4 (ONE fp16 ulp of a colour in [0.5,1); 89.8 % of colours identical), |dsigma|/sigma max / p99.9 = "
                    "4.89e-4 / 3.66e-4 at the initialisation's weight scale (131 072 samples); with the weights x 16 (a trained field's logit range, 32 768 "
                    "samples) 5.37e-3 / 3.42e-3 and 7.84e-3 / 5.84e-3 — north_star's 1e-4 is below the resolution of the reference's own fp16 outputs and is "
                    "met only by the fp32 stages. Hash-grid gradients: fixed-point (2^-24) sums per 32768-record chunk, the unmerged levels' addends "
                    "rebuilt from factored records (<= ~1 half-ulp per addend; the reference adds with rounding half2 atomics) — tolerances stated per "
                    "test in tests/",
        "config": {"workload": "configs[1]: single-object hash-grid(L16,C2,2^19)+ffmlp fp16 NeRF, rays from synthetic 800x800 views, "
                               "fixed-step renderer num_steps=512", "rays_per_step": NUM_RAYS, "samples_per_step": samples_per_step,
                   "objects": world, "parallelism": f"one object per GPU x{world}", "optimizer": "Adam(fused) inside the timed step",
                   "render_glue": "csrc/fixedstep.hip (fused density head + composite)" if fused else "torch ops of NeRFRenderer.run"},
    }

    # ---- roofline of the dominant op of the step. EVERY C-ABI entry point of the step is event-timed (ks_all: LibTimer over the five extra steps
    # above, events on the launch stream around each call); the dominant one is chosen among all of them. For the ops the KernelTimer also timed
    # INSIDE the timed region (the public grid / MLP ops) the timed region's own average is used.
    if ks_all:
        ops = step_op_table()
        step_ms_now = 1000.0 * el / args.steps
        per_step = {k: n * avg / n_k for k, (n, avg) in ks_all.items()}
        share_ms = count_share["count_share_ms"] if count_share is not None else 0.0
        ranked = dict(per_step)
        if "foc_grid_encode_backward_binned_counted" in ranked and "foc_grid_encode_forward_counted" in ranked:
            # the backward's count pass + scans ride in the training forward launch (k_grid_fwd_counted): they are the backward's
            ranked["foc_grid_encode_backward_binned_counted"] += share_ms
            ranked["foc_grid_encode_forward_counted"] = max(0.0, ranked["foc_grid_encode_forward_counted"] - share_ms)
        timed_region = {"foc_grid_encode_backward_binned_counted": "grid_encode_backward", "foc_grid_encode_backward_binned": "grid_encode_backward",
                        "foc_grid_encode_forward_counted": "grid_encode_forward_counted", "foc_grid_encode_forward": "grid_encode_forward"}

        def entry(name):
            kernels, op_bound, bpu, fpu = ops.get(name, ([name], "hbm", None, None))
            launches, avg = ks_all[name]
            src = "LibTimer, 5 steps after the timed region"
            kt = ksum.get(timed_region.get(name, ""))
            if kt is not None:
                avg, src = kt["avg_ms"], "KernelTimer, inside the timed region"
            e = {"kernels": kernels, "launches_per_step": launches / n_k, "avg_ms": round(avg, 4), "share_of_step": round(ranked.get(name, per_step[name]) / step_ms_now, 4),
                 "bound": op_bound, "timed": src}
            units = samples_per_step
            if bpu is not None:
                e["algorithmic_bytes_per_unit"] = bpu
                e["achieved_GBps"] = round(bpu * units / (avg * 1e-3) / 1e9, 1)
                e["hbm_frac"] = round(e["achieved_GBps"] / HBM_PEAK_GBS, 4)
            if fpu is not None:
                e["useful_flops_per_unit"] = fpu
                e["achieved_TFLOPs"] = round(fpu * units / (avg * 1e-3) / 1e12, 1)
                e["mfma_frac"] = round(e["achieved_TFLOPs"] / MFMA_F16_PEAK_TFLOPS, 4)
            return e
        result["kernels"] = {k: entry(k) for k in sorted(ks_all, key=lambda k: -ranked[k])}
        dom = max(ranked, key=lambda k: ranked[k])
        kernels, op_bound, bpu, fpu = ops.get(dom, ([dom], "hbm", None, None))     # (not `bound`: that is the scene's box in this function)
        d = result["kernels"][dom]
        op_ms = d["avg_ms"]
        note = ("chosen among ALL C-ABI entry points of the step (`kernels`); timed with events on the launch stream around the C-ABI call (all kernels of "
                "the op); traffic = HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE), profiles/*_pmc_hbm.csv")
        if dom == "foc_grid_encode_backward_binned_counted" and count_share is not None:
            op_ms += share_ms
            note += (f"; avg_launch_ms = scatter + reduce ({d['avg_ms']:.4f} ms) + the op's count pass and scans, which ride in the forward launch "
                     f"(counted forward {count_share['counted_forward_ms']:.4f} ms - plain forward {count_share['plain_forward_ms']:.4f} ms, timed after the run)")
        if op_bound == "mfma" and fpu is not None:
            achieved = fpu * samples_per_step / (op_ms * 1e-3) / 1e12
            result["roofline"] = {"kernel": f"{dom} = {'+'.join(kernels)}", "bound": "mfma", "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": achieved / MFMA_F16_PEAK_TFLOPS, "traffic": pmc_traffic_bytes(dom, kernels=kernels), "avg_launch_ms": op_ms,
                                  "units_per_launch": samples_per_step, "useful_flops_per_unit": fpu, "note": note}
        else:
            bpu = bpu if bpu is not None else 588.0
            achieved = bpu * samples_per_step / (op_ms * 1e-3) / 1e9
            result["roofline"] = {"kernel": f"{dom} = {'+'.join(kernels)}", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(dom, kernels=kernels), "avg_launch_ms": op_ms,
                                  "units_per_launch": samples_per_step, "algorithmic_bytes_per_unit": bpu, "note": note}
        # the MFMA side of the step: both networks' backward and forward against the dense fp16 matrix peak, on USEFUL flops
        mf = {}
        for label, names in (("mlp_backward", ("foc_ffmlp_backward_planar", "foc_color_head_backward")), ("mlp_forward", ("foc_field_forward_train", "foc_ffmlp_forward_planar", "foc_color_head_forward"))):
            have = [n for n in names if n in ks_all]
            if not have:
                continue
            ms = sum(ks_all[n][1] * ks_all[n][0] / n_k for n in have)
            fl = sum(ops[n][3] for n in have) * samples_per_step
            mf[label] = {"entry_points": {n: result["kernels"][n]["avg_ms"] for n in have}, "ms_per_step": round(ms, 4), "useful_flops_per_step": fl,
                         "achieved_TFLOPs": round(fl / (ms * 1e-3) / 1e12, 1), "peak_TFLOPs": MFMA_F16_PEAK_TFLOPS,
                         "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)}
        util = pmc_mfma_utilisation(["k_mlp_bwd", "k_mlp_fwd", "k_field_fwd"])
        if util is not None:
            mf["mfma_pipe_utilisation_from_committed_profile"] = util
        mf["note"] = ("useful flops = 2 x 36 864 per sample backward (dX and dW products; the forward the fused backward re-evaluates is NOT counted, so the "
                      "matrix pipe's own busy share — the PMC figure — is about 1.5 x the useful fraction), 36 864 forward; sigma 32-64-64-16, colour 32-64-64-64-16")
        result["roofline"]["mfma"] = mf
    elif ksum:
        # the all-entry-point timing failed (step_ms.kernel_only_error): fall back to the ops the KernelTimer saw inside the timed region
        dom = max(ksum, key=lambda k: ksum[k]["total_ms"])
        op_ms = ksum[dom]["avg_ms"] + (count_share["count_share_ms"] if (dom == "grid_encode_backward" and count_share is not None) else 0.0)
        achieved = 588.0 * samples_per_step / (op_ms * 1e-3) / 1e9
        result["roofline"] = {"kernel": f"{dom} = {'+'.join(OP_KERNELS.get(dom, [dom]))}", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(dom), "avg_launch_ms": op_ms, "units_per_launch": samples_per_step,
                              "algorithmic_bytes_per_unit": 588.0, "note": "fallback: chosen among the public grid / MLP ops only (KernelTimer, timed region)"}
        result["kernels"] = {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4)} for k, v in sorted(ksum.items(), key=lambda kv: -kv[1]["total_ms"])}

    from focnerf_amd import synthetic
    rays_o, rays_d = synthetic.get_rays(poses[:1], intr, VIEW, VIEW)
    # single-GPU properties (other paths of the same object) are reported at N = 1 only; the multi-rank runs keep to the headline
    # step plus the one exchange step the path has (combine), so that no rank-local failure can leave the others in a collective
    if not args.no_extras and world == 1:
        progress("single-GPU extras (torch-glue step, renders, occupancy path)")
        try:
            # ---- the same step through the reference caller's torch glue (NeRFRenderer.run) instead of the fused kernels
            if fused:
                for i in range(3):
                    train_step(model, opt, scaler, *batches[i % len(batches)], fused=False)
                barrier()
                t0 = time.perf_counter()
                nu = max(5, args.steps // 2)
                for i in range(nu):
                    train_step(model, opt, scaler, *batches[i % len(batches)], fused=False)
                barrier()
                elu = max_over_ranks(time.perf_counter() - t0)
                result["torch_glue_path"] = {"metric": "train_samples_per_sec", "value": world * samples_per_step * nu / elu, "unit": "samples/s",
                                             "ms_per_step": 1000.0 * elu / nu, "path": "same step, NeRFRenderer.run torch glue around the same kernels"}
            # ---- what an UNCHANGED FOCNeRF checkout gets: the same step through the public ops only, in the reference's call sequence
            # (nerf/network_ff.py:51-134 inside nerf/renderer.py:145-221): grid_encode -> FFMLP -> trunc_exp -> SH -> cat / pad -> FFMLP (on
            # the samples whose weight passes the threshold) -> sigmoid -> torch cumprod composite, every fusion of this library's own
            # callers switched off (FOC_FUSED_FIELD / HEAD / TAIL / INFER = 0). The four ops are the drop-in boundary; everything between
            # them is the reference's torch code.
            if fused:
                switches = ("FOC_FUSED_FIELD", "FOC_FUSED_HEAD", "FOC_FUSED_TAIL", "FOC_FUSED_INFER", "FOC_FUSED_OCC")
                saved = {k: os.environ.get(k) for k in switches}
                try:
                    for k in switches:
                        os.environ[k] = "0"
                    for i in range(3):
                        train_step(model, opt, scaler, *batches[i % len(batches)], fused=False)
                    barrier()
                    timer.records.clear()
                    timer.enabled = True
                    nd = max(5, args.steps // 2)
                    t0 = time.perf_counter()
                    for i in range(nd):
                        train_step(model, opt, scaler, *batches[i % len(batches)], fused=False)
                    barrier()
                    eld = max_over_ranks(time.perf_counter() - t0)
                    timer.enabled = False
                    kd = timer.summary()
                    result["dropin_ops_path"] = {
                        "metric": "train_samples_per_sec", "value": world * samples_per_step * nd / eld, "unit": "samples/s", "ms_per_step": 1000.0 * eld / nd,
                        "vs_headline": (world * samples_per_step * nd / eld) / value,
                        "path": "configs[1] step through the public ops only (grid_encode, ffmlp_forward x2, trunc_exp, SH encoder) in the reference's "
                                "sequence, NeRFRenderer.run torch glue, all fusions off: what nerf/renderer.py + nerf/network_ff.py get on these ops",
                        "op_share_of_step": {k: {"launches_per_step": v["launches"] / nd, "avg_ms": round(v["avg_ms"], 4), "share": round(v["total_ms"] / (1000.0 * eld), 4)}
                                             for k, v in sorted(kd.items(), key=lambda kv: -kv[1]["total_ms"])},
                        "torch_glue_share": round(max(0.0, 1.0 - sum(v["total_ms"] for v in kd.values()) / (1000.0 * eld)), 4)}
                finally:
                    timer.enabled = False
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
            # ---- the headline step replayed as ONE HIP graph (focnerf_amd.graph.GraphedStep; shapes are static on the fixed-step path): the same
            # kernels with no host in the loop. `value` above stays the eager step (the host's lead over the GPU grows by ~0.5-0.9 ms per step from
            # the empty queue the timed region starts with, so one pause of the host inside the first steps shows up in its mean; a replayed graph
            # is enqueued in microseconds). No LR scheduler inside the graph (a host-side scalar); everything else is train_step().
            try:
                from focnerf_amd.graph import GraphedStep
                model_g = build_model(bound, device, cuda_ray=False, seed=rank).train()      # a copy of the trained model: the render legs below (and their
                model_g.load_state_dict(model.state_dict())                                    # checksums) see `model` as the timed steps left it
                opt_g = torch.optim.Adam(model_g.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
                sc_g = torch.amp.GradScaler("cuda")
                g_head = GraphedStep(lambda o, d, t: train_step(model_g, opt_g, sc_g, o, d, t, fused=fused), batches[0])
                for i in range(4):
                    g_head(*batches[i % len(batches)])
                barrier()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    g_head(*batches[i % len(batches)])
                barrier()
                el_g = max_over_ranks(time.perf_counter() - t0)
                result["graph_replay"] = {"metric": "train_samples_per_sec", "value": world * samples_per_step * args.steps / el_g, "unit": "samples/s",
                                          "ms_per_step": 1000.0 * el_g / args.steps, "steps": args.steps,
                                          "note": "the headline step (forward, backward, GradScaler, fused Adam; a fresh batch copied in per step) captured once and "
                                                  "replayed as one HIP graph: the host-independent figure beside the eager `value`"}
                del g_head, opt_g, sc_g, model_g
            except Exception as e:
                result["graph_replay"] = {"error": repr(e)}
            # ---- render: full 800x800 views through the same fixed-step path, staged in 4096-ray chunks (max_ray_batch, flags default)
            progress("render leg (fixed-step, 800x800 views)")
            model.eval()
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                # one untimed full view first: the caching allocator sizes its blocks for the view's chunk shapes
                rkw = dict(staged=True, max_ray_batch=4096, num_steps=NUM_STEPS, upsample_steps=0, perturb=False, fused=fused)
                # the views cycle through the eight camera poses: the encoder's cost depends on how a view's rows lie to the x axis
                # (x-neighbours share cache lines of the hash tables), 0.17 - 0.24 ms per chunk over these poses
                view_rays = [synthetic.get_rays(poses[v:v + 1], intr, VIEW, VIEW) for v in range(poses.shape[0])]
                model.render(rays_o, rays_d, return_fields=False, **rkw)
                barrier()
                t0 = time.perf_counter()
                for i in range(args.render_views):
                    model.render(*view_rays[i % len(view_rays)], return_fields=False, **rkw)
                barrier()
            rel = max_over_ranks(time.perf_counter() - t0)
            result["render"] = {"metric": "render_rays_per_sec", "value": world * VIEW * VIEW * args.render_views / rel, "unit": "rays/s",
                                "samples_per_sec": world * VIEW * VIEW * NUM_STEPS * args.render_views / rel, "views": args.render_views,
                                "s_per_view": rel / args.render_views, "path": "fixed-step run(), 512 samples/ray, 4096-ray chunks, image + depth; views cycle through 8 camera poses"}
            # the render half of BASELINE's metric inside `roofline` (the driver's record keeps `roofline` and `config` whole)
            try:
                rr = render_roofline(model, view_rays, rkw)
                if rr is not None and "roofline" in result:
                    rr.update({"metric": "render_rays_per_sec", "rays_per_sec": result["render"]["value"], "s_per_view": result["render"]["s_per_view"]})
                    result["roofline"]["render"] = rr
            except Exception as e:
                result.setdefault("roofline", {})["render"] = {"error": repr(e)}
            # the same views on an OPAQUE field (density_scale 1e4: alpha ~ 1 at a ray's first sample, as behind a trained object's surface): the
            # reference evaluates the colour network only where weights > 1e-10 (nerf/renderer.py:185-187), this path evaluates it densely and
            # masks in the composite — so the time of a view must not depend on the field, and `colour_mask_fraction` says how sparse the
            # reference's mask would be here (4096 rays of view 0, weights formed in torch from the returned densities)
            saved_scale = model.density_scale
            try:
                from focnerf_amd import raymarching
                model.density_scale = 1.0e4
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                    model.render(rays_o, rays_d, return_fields=False, **rkw)
                    barrier()
                    t0 = time.perf_counter()
                    for i in range(args.render_views):
                        model.render(*view_rays[i % len(view_rays)], return_fields=False, **rkw)
                    barrier()
                    relq = max_over_ranks(time.perf_counter() - t0) / max(1, args.render_views)
                    so, sd = view_rays[0][0][:, :4096].contiguous(), view_rays[0][1][:, :4096].contiguous()
                    part = model.render(so, sd, return_fields=True, **rkw)
                    sig = part["densities"].float().view(-1, NUM_STEPS)
                    nf_near, nf_far = raymarching.near_far_from_aabb(so.view(-1, 3), sd.view(-1, 3), model.aabb_infer, model.min_near)
                    step = ((nf_far - nf_near) / NUM_STEPS).view(-1, 1)
                    alpha = 1 - torch.exp(-step * model.density_scale * sig)
                    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-15], dim=-1), dim=-1)[:, :-1]
                    frac = float(((alpha * trans) > 1e-10).float().mean())
                result["render"]["opaque_field"] = {"s_per_view": relq, "vs_render": relq / result["render"]["s_per_view"], "density_scale": 1.0e4,
                                                    "colour_mask_fraction": frac,
                                                    "note": "same 8 poses; the colour network runs densely, the weights > 1e-10 mask is applied in the composite"}
            except Exception as e:
                result["render"]["opaque_field"] = {"error": repr(e)}
            finally:
                model.density_scale = saved_scale
            # the reference's eval render also assembles the per-sample fields of the whole view for the combiner (renderer.py:524-547)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                # the SAME views as the `render` leg, in the same order: a view costs 40 - 45 ms depending on its pose, and with one fixed view here
                # the two legs were not comparable (round 3: 41.5 ms for pose 0 against 42.4 ms for the mean over eight poses)
                model.render(rays_o, rays_d, return_fields=True, **rkw)
                barrier()
                t0 = time.perf_counter()
                for i in range(args.render_views):
                    model.render(*view_rays[i % len(view_rays)], return_fields=True, **rkw)
                barrier()
            relf = max_over_ranks(time.perf_counter() - t0) / max(1, args.render_views)
            result["render_with_fields"] = {"metric": "render_rays_per_sec", "value": world * VIEW * VIEW / relf, "unit": "rays/s", "s_per_view": relf,
                                            "path": "same, plus densities [1,N,512] and rgbs [1,N,512,3] of the whole view (5.2 GB) as the reference's render() returns them"}

            # ---- what an UNCHANGED FOCNeRF checkout gets for a RENDER: one 800 x 800 view through the public ops only, in the sequence of the
            # reference's run() (nerf/renderer.py:145-221 with nerf/network_ff.py:51-134: near_far_from_aabb -> torch linspace / clamp -> grid_encode ->
            # FFMLP -> trunc_exp -> cumprod weights -> SH + cat / pad + FFMLP on the samples whose weight passes 1e-10 -> sigmoid -> torch composite),
            # every fusion of this library's own callers off, as `dropin_ops_path` does for training
            if fused and "dropin_ops_path" in result:
                switches = ("FOC_FUSED_FIELD", "FOC_FUSED_HEAD", "FOC_FUSED_TAIL", "FOC_FUSED_INFER", "FOC_FUSED_OCC")
                saved = {k: os.environ.get(k) for k in switches}
                try:
                    for k in switches:
                        os.environ[k] = "0"
                    rkd = dict(staged=True, max_ray_batch=4096, num_steps=NUM_STEPS, upsample_steps=0, perturb=False, fused=False)
                    nv = 2
                    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                        model.render(*view_rays[0], **rkd)
                        barrier()
                        timer.records.clear()
                        timer.enabled = True
                        t0 = time.perf_counter()
                        for i in range(nv):
                            model.render(*view_rays[i % len(view_rays)], **rkd)
                        barrier()
                        eldr = max_over_ranks(time.perf_counter() - t0)
                        timer.enabled = False
                    kdr = timer.summary()
                    result["dropin_ops_path"]["render"] = {
                        "metric": "render_rays_per_sec", "value": world * VIEW * VIEW * nv / eldr, "unit": "rays/s", "s_per_view": eldr / nv, "views": nv,
                        "vs_render": (world * VIEW * VIEW * nv / eldr) / result["render"]["value"],
                        "path": "800 x 800 view through the public ops only (near_far_from_aabb, grid_encode, ffmlp x2, trunc_exp, SH encoder) in the sequence of "
                                "the reference's run(), 4096-ray chunks, torch glue between the ops, all fusions off",
                        "op_share_of_view": {k: {"launches_per_view": v["launches"] / nv, "avg_ms": round(v["avg_ms"], 4), "share": round(v["total_ms"] / (1000.0 * eldr), 4)}
                                             for k, v in sorted(kdr.items(), key=lambda kv: -kv[1]["total_ms"])},
                        "torch_glue_share": round(max(0.0, 1.0 - sum(v["total_ms"] for v in kdr.values()) / (1000.0 * eldr)), 4)}
                except Exception as e:
                    result["dropin_ops_path"]["render"] = {"error": repr(e)}
                finally:
                    timer.enabled = False
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v

            # ---- the same two measurements on FOC's object-conditioned network (network_tcnn.py topology, 48-wide colour input)
            progress("FOC object-conditioned network legs")
            try:
                mf = build_foc_model(bound, device, seed=rank).train()
                optf = torch.optim.Adam(mf.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
                scf = torch.amp.GradScaler("cuda")
                yolo = foc_yolo_details(device, NUM_RAYS, 7 + rank)
                for i in range(max(args.warmup, 10)):
                    foc_train_step(mf, optf, scf, *batches[i % len(batches)], yolo)
                barrier()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    foc_train_step(mf, optf, scf, *batches[(args.warmup + i) % len(batches)], yolo)
                barrier()
                elf = max_over_ranks(time.perf_counter() - t0)
                foc = {"network": "focnerf_amd.network_foc.NeRFNetwork: hash grid 32 -> 64 -> 64 -> 16; colour [SH16 | geo15 | encoded object feature 16 | 0] = 48 -> 64 -> 64 -> 3; "
                                  "object-feature encoder 144 -> 16 -> 16 (nerf/network_tcnn.py:451-681)",
                       "train": {"metric": "train_samples_per_sec", "value": world * samples_per_step * args.steps / elf, "unit": "samples/s",
                                 "ms_per_step": 1000.0 * elf / args.steps, "vs_plain_topology": (world * samples_per_step * args.steps / elf) / value,
                                 "path": "same step as the headline with yolo_details: fused tail with the object feature, outside-mask criterion, Adam over 5 groups"}}
                mf.eval()
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                    mf.render(rays_o, rays_d, yolo, return_fields=False, **rkw)
                    barrier()
                    t0 = time.perf_counter()
                    for i in range(args.render_views):
                        mf.render(*view_rays[i % len(view_rays)], yolo, return_fields=False, **rkw)
                    barrier()
                relo = max_over_ranks(time.perf_counter() - t0)
                foc["render"] = {"metric": "render_rays_per_sec", "value": world * VIEW * VIEW * args.render_views / relo, "unit": "rays/s",
                                 "s_per_view": relo / args.render_views, "vs_plain_topology": (world * VIEW * VIEW * args.render_views / relo) / result["render"]["value"],
                                 "path": "fixed-step run() with yolo_details, whole-field kernel with the object feature"}
                result["foc_network"] = foc
                del mf, optf
            except Exception as e:
                import traceback
                result["foc_network"] = {"error": repr(e), "trace": traceback.format_exc()[-800:]}

            # ---- configs[2]: occupancy-grid path (march_rays_train -> encode -> MLPs -> composite_rays_train -> backward -> Adam)
            progress("occupancy-grid path (configs[2]): training step, render loop")
            m2 = build_model(2, device, cuda_ray=True, seed=rank).train()
            opt2 = torch.optim.Adam(m2.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
            sc2 = torch.amp.GradScaler("cuda")
            poses2, _ = make_training_rays(device, 2, 8, seed=rank)
            b2 = [sample_batch(poses2, intr, device, gen) for _ in range(4)]
            for i in range(17):
                cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
                if i == 15:
                    m2.mean_count = int(m2.step_counter[:16, 0].sum().item() / 16)    # what update_extra_state would set (renderer.py:533)
            barrier()
            n2 = max(3 * args.steps, 30)                  # the eager step is host-bound: more steps, less jitter in the figure
            c0 = 0
            t0 = time.perf_counter()
            for i in range(n2):
                cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
            barrier()
            el2 = max_over_ranks(time.perf_counter() - t0)
            per_step = float(m2.step_counter[:, 0].float().mean().item())
            result["occupancy_path"] = {"metric": "train_samples_per_sec", "value": world * per_step * n2 / el2, "unit": "samples/s",
                                        "ms_per_step": 1000 * el2 / n2, "rays_per_step": NUM_RAYS, "samples_per_step": per_step,
                                        "path": "configs[2]: occupancy-grid training step, bound 2, eager: NeRFRenderer.run_cuda -> focnerf_amd/occtrain.py (march in the "
                                                "field's layout, encoder + count, both MLPs, ragged tail: ONE autograd node, 16 library launches) + MSE + GradScaler + fused "
                                                "Adam; FOC_FUSED_OCC=0 is the chain of separate ops (march_rays_train, grid_encode, FFMLP x2, composite_rays_train)"}
            po = profile_occupancy_kernel_us_per_step()
            if po is not None:
                result["occupancy_path"]["from_committed_profile"] = po   # parsed from a file of an earlier run, not measured by this run
            # the same step as the chain of separate public ops (what round 3 measured as `occupancy_path`)
            saved_occ = os.environ.get("FOC_FUSED_OCC")
            try:
                os.environ["FOC_FUSED_OCC"] = "0"
                for i in range(4):
                    cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
                barrier()
                t0 = time.perf_counter()
                for i in range(n2):
                    cuda_ray_train_step(m2, opt2, sc2, *b2[i % 4])
                barrier()
                elc = max_over_ranks(time.perf_counter() - t0)
                result["occupancy_path"]["op_chain"] = {"ms_per_step": 1000 * elc / n2, "value": world * per_step * n2 / elc,
                                                        "note": "FOC_FUSED_OCC=0: one autograd node per op, the caller-side torch glue between them"}
            except Exception as e:
                result["occupancy_path"]["op_chain"] = {"error": repr(e)}
            finally:
                if saved_occ is None:
                    os.environ.pop("FOC_FUSED_OCC", None)
                else:
                    os.environ["FOC_FUSED_OCC"] = saved_occ

            # replayed as one HIP graph (focnerf_amd.graph.GraphedStep, static shapes thanks to the sample budget): what the step costs with no
            # host in the loop (round 3's chain of ops needed 1.2 ms of Python per step for 1.05 ms of GPU work; the fused node 0.75 for 0.8)
            try:
                from focnerf_amd.graph import GraphedStep
                optg = torch.optim.Adam(m2.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
                gstep = GraphedStep(lambda o, d, t: cuda_ray_train_step(m2, optg, sc2, o, d, t), b2[0])
                for i in range(4):
                    gstep(*b2[i % 4])
                barrier()
                t0 = time.perf_counter()
                for i in range(n2):
                    gstep(*b2[i % 4])
                barrier()
                elg = max_over_ranks(time.perf_counter() - t0)
                result["occupancy_path"]["graph_replay"] = {"ms_per_step": 1000 * elg / n2, "value": world * per_step * n2 / elg,
                                                            "note": "same step captured once and replayed as one HIP graph"}
            except Exception as e:
                result["occupancy_path"]["graph_replay"] = {"error": repr(e)}

            # ---- configs[2] render half: full 800x800 view through the incremental march_rays / composite_rays loop
            m2.eval()
            ro2, rd2 = synthetic.get_rays(poses2[:1], intr, VIEW, VIEW)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                m2.render(ro2, rd2, staged=False, perturb=False, dt_gamma=1 / 128, max_steps=1024, T_thresh=1e-4, device_compaction=True)
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.render_views):
                    m2.render(ro2, rd2, staged=False, perturb=False, dt_gamma=1 / 128, max_steps=1024, T_thresh=1e-4, device_compaction=True)
                barrier()
            rel2 = max_over_ranks(time.perf_counter() - t0)
            result["render_occupancy"] = {"metric": "render_rays_per_sec", "value": world * VIEW * VIEW * args.render_views / rel2, "unit": "rays/s",
                                          "s_per_view": rel2 / args.render_views, "path": "configs[2]: march_rays + composite_rays loop (occupancy grid), bound 2; one native call per iteration, "
                                                  + os.environ.get("FOC_RENDER_BURST", "8") + " samples per ray and iteration (image identical to the reference's schedule, FOC_RENDER_BURST=1)"}
            # ---- density-grid maintenance (update_extra_state, every 16 steps in the reference trainer, utils.py:850-852), timed on its own:
            # running it inside the loop above would replace the analytic occupancy grid by the untrained network's density
            m2.train()
            with torch.autocast("cuda", dtype=torch.float16):
                for _ in range(17):                      # the first 16 calls sweep the full grid (renderer.py:461-476)
                    m2.update_extra_state()
                barrier()
                t0 = time.perf_counter()
                for _ in range(5):
                    m2.update_extra_state()
                barrier()
            result["occupancy_path"]["grid_update_ms"] = 1000 * (time.perf_counter() - t0) / 5
            result["occupancy_path"]["grid_update_note"] = "update_extra_state, steady-state branch, once per 16 training steps; not inside ms_per_step"
        except Exception as e:   # an extra must never take the headline number down with it
            result["extras_error"] = repr(e)

    if not args.no_extras:
        # ---- configs[3]/[4]: K = N objects, one per rank, a full 800x800 view end to end (field evaluation + exchange + composite + gather)
        progress("combined-render leg")
        model.eval()
        # The N-rank exchange is the one part of this file in which a rank can wait for another one. If it makes no progress (a rank that
        # failed and left the others inside a collective), every rank leaves on its own timer — before the process group's 240 s timeout
        # would abort the job — and rank 0 still prints the line with the headline and what has been measured so far.
        guard = ExchangeGuard(rank, world, args.exchange_timeout, result)
        guard.start()
        try:
            from focnerf_amd import raymarching as rm
            from focnerf_amd.field import half_cache_scope
            from focnerf_amd.fixedstep import render_field4
            vo, vd = rays_o[0].contiguous(), rays_d[0].contiguous()
            # the view's rays in 8 x 8 pixel tiles, as the staged render walks them (rayorder.py): every rank permutes the same way, the exchange and
            # the composite are per ray, the image rows go back to the caller's order at the end of each view
            from focnerf_amd.rayorder import view_tiling
            tile_order = view_tiling(vd)
            if tile_order is not None:
                vo, vd = vo.index_select(0, tile_order), vd.index_select(0, tile_order)
            vn, vf = rm.near_far_from_aabb(vo, vd, model.aabb_infer, model.min_near)
            # the objects COMBINED.py loads are network_tcnn networks (COMBINED.py:84): one FOC object-conditioned network per rank, each
            # with its own object feature (gather_obj_feats, utils.py:177-187)
            obj_model = build_foc_model(bound, device, seed=rank).eval()
            obj_yolo = foc_yolo_details(device, 1, 50 + rank)

            def my_object(lo, hi, out):
                return render_field4(obj_model, vo[lo:hi], vd[lo:hi], num_steps=NUM_STEPS, yolo_details=obj_yolo, out=out)
            with torch.no_grad(), half_cache_scope():
                result["combined_render"] = combined_render_leg(rank, world, device, {"nears": vn, "fars": vf}, args.combined_views, my_object, barrier,
                                                                max_over_ranks, overlap=not args.combined_no_overlap, ray_order=tile_order,
                                                                collectives_at_world_1=rehearse)
            if rehearse:
                result["combined_render"]["rehearsal"] = ("ONE rank with the nccl process group, every collective issued (RCCL copies on the device): NOT an N > 1 "
                                                          "measurement — no byte leaves the GPU")
            result["combined_render"]["ray_order"] = "8x8 pixel tiles" if tile_order is not None else "as given"
            cr = result["combined_render"]
            # the N-rank half of BASELINE's metric inside `roofline` (the driver's record keeps `roofline` and `config` whole)
            result.setdefault("roofline", {})["combined"] = {
                "metric": "combined_render_rays_per_sec", "rays_per_sec": cr["value"], "object_rays_per_sec": cr["object_rays_per_sec"], "n_ranks": cr["world_size"],
                "backend": cr["backend"], "objects": cr["objects"], "s_per_view": cr["s_per_view"], "field_eval_only_s_per_view": cr["field_eval_only_s_per_view"],
                "bytes_sent_per_view_per_gpu": cr["bytes_sent_per_view_per_gpu"], "bound": "xgmi", "achieved": cr["xgmi"]["achieved_GBps_per_gpu"],
                "peak": cr["xgmi"]["peak_GBps_per_gpu"], "unit": "GB/s", "frac": cr["xgmi"]["frac_of_peak"], "overlap": cr["overlap"],
                "s_per_view_other_exchange_mode": cr["s_per_view_other_exchange_mode"]}
            result["combined_render"]["objects_network"] = "focnerf_amd.network_foc.NeRFNetwork (object-conditioned, 48-wide colour input), one per rank"
            if world == 1:
                # the single-GPU form of the same job: K = 4 objects RESIDENT on one device (COMBINED.py reloads a checkpoint per object per view)
                from focnerf_amd.combine import ObjectCombiner
                comb1 = ObjectCombiner(rank=0, world_size=1)
                fns = resident_object_fields(device, bound, vo, vd, 4, first=(obj_model, obj_yolo))
                with torch.no_grad(), half_cache_scope():
                    comb1.render_view(fns, VIEW * VIEW, vn, vf, NUM_STEPS, max_ray_batch=16384)
                    barrier()
                    t0 = time.perf_counter()
                    img4_4, _ = comb1.render_view(fns, VIEW * VIEW, vn, vf, NUM_STEPS, max_ray_batch=16384)
                    barrier()
                el4 = time.perf_counter() - t0
                result["combined_render"]["resident_4_objects_one_gpu"] = {"s_per_view": el4, "rays_per_sec": VIEW * VIEW / el4,
                                                                           "object_rays_per_sec": 4 * VIEW * VIEW / el4,
                                                                           "image_checksum": float(img4_4.double().sum().item()),
                                                                           "note": "configs[3] on ONE GPU: 4 resident objects, per chunk 4 field evaluations + "
                                                                                   "one select/composite kernel; the N-GPU job's single-device baseline"}
        except Exception as e:   # at N = 1 an extra must never take the headline number down with it
            import traceback
            if world > 1:                                    # the other ranks may be inside a collective: no orderly shutdown of the group, and
                guard.abandon(repr(e), traceback.format_exc()[-800:])      # the launcher must see the failure — one line from rank 0, exit code 3
            result["combined_render"] = {"error": repr(e), "trace": traceback.format_exc()[-800:]}
        guard.cancel()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The baseline north_star names: the reference's pure-PyTorch network (--ff off, --tcnn off: nerf/network.py) through the fixed-step
        # renderer on the node's host cores — configs[0] by the protocol of BASELINE.md section 2, IN FULL (one whole 400 x 400 view, median of
        # five full 4096-ray steps: ~2 min on 16 cores). oracle/torch_cpu_nerf.py is the torch-CPU restatement of that configuration, pinned to
        # the reference's own network class + run() by tests/golden/cpu_network.npz. Nested beside it: the scalar C oracle port of this
        # step's own kernels on a bounded sample.
        progress("cpu baseline: configs[0] torch CPU (BASELINE.md section 2 protocol, ~2 min)")
        try:
            from oracle import torch_cpu_nerf
            c0 = torch_cpu_nerf.time_baseline()
            result["cpu_baseline"] = {"value": c0["train"]["samples_per_sec"], "unit": "samples/s", "cores": c0["cores"], "kind": "port",
                                      "cpu_model": c0["cpu_model"], "extrapolated": c0["extrapolated"],
                                      "sample": "configs[0], BASELINE.md section 2: " + c0["train"]["sample"] + "; render: " + c0["render"]["sample"],
                                      "render_rays_per_sec": c0["render"]["rays_per_sec"], "render_s_per_view": c0["render"]["s_per_view"],
                                      "train_s_per_step": c0["train"]["s_per_step"], "config": c0["config"], "protocol": c0["protocol"]}
        except Exception as e:
            result["cpu_baseline"] = {"error": repr(e)}
        progress("cpu baseline: C oracle port of this step's kernels")
        try:
            result["cpu_baseline"]["c_oracle_port"] = cpu_baseline()
        except Exception as e:
            result["cpu_baseline"]["c_oracle_port"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(result))
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
