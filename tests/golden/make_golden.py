"""Generates tests/golden/*.npz from the parts of the reference that run on CPU.

Run ONLY in the build container (needs /root/reference); the GPU box never runs this.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from the reference (nothing is copied into this repo; the fixtures hold data only):
  * activation.trunc_exp                      -> trunc_exp.npz   (forward + backward values)
  * nerf.renderer.NeRFRenderer.run            -> run_foc.npz     (FOC fixed-step compositing, mask w > 1e-10)
  * nerf.renderer.NeRFRenderer.mark_untrained_grid / update_extra_state -> grid_maintenance.npz (grid size 32)
  * COMBINED.py best_densities_and_colors_v3 / image_depth_generation (compiled from the file with ast; not importable) -> combined.npz
  * nerf.network.NeRFNetwork (the pure-PyTorch network class) through nerf.renderer.NeRFRenderer.run, on the encoders of
    oracle/torch_cpu_nerf.py -> cpu_network.npz (pins the CPU baseline of configs[0])
  * nerf.renderer.sample_pdf and the legacy renderer's run() with upsample_steps > 0 (compiled from the file with ast) -> upsample.npz
  * editable.py modify_rays_for_object / get_object_type_from_ckpt / batch_run / run / select / composite (same way), 8 objects, 2 views -> editable.npz
  * gridencoder.GridEncoder / grid_encode and ffmlp.FFMLP / ffmlp_forward (the Python wrappers, on oracle-backed stub backends) -> wrappers.npz
  * raymarching/raymarching.py wrappers (same arrangement; Tensor.cuda patched to the identity for the run) -> raymarching_wrappers.npz
  (legacy/nerf/renderer.py is not importable here: its `from .utils import custom_meshgrid` pulls in
   imageio, cv2, tensorboardX, mcubes, lpips, torchmetrics, torch_ema ... none of which are installed.)
`raymarching` (a CUDA extension that would JIT-build on import) and `trimesh`
(absent) are stubbed in sys.modules BEFORE the import; the stub's near_far_from_aabb is this
repo's CPU oracle, and its outputs are stored in the fixture as inputs of the composite.
The density / colour fields are analytic functions defined here.
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
import oracle  # noqa: E402

# ---- stubs, installed before any reference import
rm = types.ModuleType("raymarching")


def _near_far(rays_o, rays_d, aabb, min_near=0.2):
    n, f = oracle.near_far_from_aabb(rays_o.detach().numpy(), rays_d.detach().numpy(), aabb.detach().numpy(), min_near)
    return torch.from_numpy(n), torch.from_numpy(f)


rm.near_far_from_aabb = _near_far
sys.modules["raymarching"] = rm
sys.modules["trimesh"] = types.ModuleType("trimesh")
sys.path.insert(0, REF)

from activation import trunc_exp  # noqa: E402
import nerf.renderer as foc_renderer  # noqa: E402


def sigma_field(x):
    c = torch.tensor([0.1, -0.05, 0.2])
    r2 = ((x - c) ** 2).sum(-1)
    return 40.0 * torch.exp(-r2 / 0.12) + 3.0 * torch.exp(-((x + 0.4) ** 2).sum(-1) / 0.02)


def color_field(x, d):
    return 0.5 + 0.5 * torch.sin(3.0 * x + 0.7 * d)


def make_rays(N, seed, bound):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(N, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * (1.6 * bound)
    tgt = (torch.rand(N, 3, generator=g) - 0.5) * bound
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    # a few rays whose line misses the box entirely: tangent direction at distance 2*bound > sqrt(3)*bound
    o[::17] = o[::17] * 1.25
    t = torch.cross(o[::17], torch.tensor([[0.3, -0.5, 0.8]]).expand_as(o[::17]), dim=-1)
    d[::17] = t / t.norm(dim=-1, keepdim=True)
    return o.float(), d.float()


def run_reference(mod, foc, bound, N, T, seed):
    class Toy(mod.NeRFRenderer):
        def density(self, x):
            return {'sigma': sigma_field(x), 'geo_feat': x[..., :2] * 0.0}

        if foc:
            def color(self, x, d, yolo_details, mask=None, geo_feat=None, **kw):
                rgbs = torch.zeros(mask.shape[0], 3)
                rgbs[mask] = color_field(x[mask], d[mask])
                return rgbs
        else:
            def color(self, x, d, mask=None, geo_feat=None, **kw):
                rgbs = torch.zeros(mask.shape[0], 3)
                rgbs[mask] = color_field(x[mask], d[mask])
                return rgbs

    m = Toy(bound=bound, min_near=0.2)
    m.eval()
    o, d = make_rays(N, seed, bound)
    with torch.no_grad():
        if foc:
            res = m.run(o[None], d[None], None, num_steps=T, upsample_steps=0, bg_color=None, perturb=False)
        else:
            res = m.run(o[None], d[None], num_steps=T, upsample_steps=0, bg_color=None, perturb=False)
    nears, fars = _near_far(o, d, m.aabb_infer, m.min_near)
    # fields at the sample positions the renderer used (recomputed exactly as run() does)
    z = torch.linspace(0.0, 1.0, T)[None].expand(N, T)
    z = nears[:, None] + (fars - nears)[:, None] * z
    xyz = o[:, None, :] + d[:, None, :] * z[..., None]
    xyz = torch.min(torch.max(xyz, m.aabb_infer[:3]), m.aabb_infer[3:])
    sig = sigma_field(xyz.reshape(-1, 3)).view(N, T)
    out = dict(rays_o=o.numpy(), rays_d=d.numpy(), aabb=m.aabb_infer.numpy(), min_near=np.float32(m.min_near), T=np.int32(T),
               nears=nears.numpy(), fars=fars.numpy(), sigmas=sig.numpy(),
               image=res['image'][0].numpy(), depth=res['depth'][0].numpy(), weights_sum=res['weights_sum'].numpy())
    if foc:
        out['rgbs'] = res['rgbs'].numpy()
        assert np.array_equal(res['densities'].squeeze(-1).numpy(), sig.numpy())
    else:
        # legacy run() does not return the fields: recompute colour with its w > 1e-4 mask
        dl = z[:, 1:] - z[:, :-1]
        dl = torch.cat([dl, ((fars - nears) / T)[:, None]], -1)
        al = 1 - torch.exp(-dl * sig)
        w = al * torch.cumprod(torch.cat([torch.ones_like(al[:, :1]), 1 - al + 1e-15], -1), -1)[:, :-1]
        msk = w > 1e-4
        rgbs = torch.zeros(N, T, 3)
        dd = d[:, None, :].expand(N, T, 3)
        rgbs[msk] = color_field(xyz[msk], dd[msk])
        out['rgbs'] = rgbs.numpy()
    return out


def poly_sigma(x):
    """Analytic density of the grid-maintenance fixture: add / multiply / max only, so that every IEEE-754 machine gets the same bits
    (torch's CPU exp differs in the last place between vector ISAs)."""
    c = torch.tensor([0.1, -0.05, 0.2])
    r2 = ((x - c) * (x - c)).sum(-1)
    a = torch.clamp(1.0 - r2 * 2.5, min=0.0)
    q = ((x + 0.4) * (x + 0.4)).sum(-1)
    b = torch.clamp(1.0 - q * 16.0, min=0.0)
    return a * a * 40.0 + b * 3.0


def grid_maintenance(H=32, bound=2):
    """NeRFRenderer.mark_untrained_grid and update_extra_state (nerf/renderer.py:356-508) run on the CPU at grid size H (the reference
    hard-codes 128; the attribute and the two buffers are replaced after construction, every use goes through self.grid_size).
    morton3D / morton3D_invert / packbits are this repo's oracle (the reference's are CUDA); torch.rand_like is pinned to 0.5 (cell
    centres) and the torch.randint draws of the steady-state branch are recorded, so the call can be replayed exactly."""
    # nerf/renderer.py calls custom_meshgrid without importing it (the reference's FOC renderer cannot run its own occupancy path,
    # SURVEY.md H4); the helper is torch.meshgrid(..., indexing='ij') in the reference's utils.py
    foc_renderer.custom_meshgrid = lambda *a: torch.meshgrid(*a, indexing='ij')
    rm.morton3D = lambda c: torch.from_numpy(oracle.morton3D(c.numpy()))
    rm.morton3D_invert = lambda i: torch.from_numpy(oracle.morton3D_invert(i.numpy().astype(np.int32)))
    rm.packbits = lambda grid, thresh, bitfield=None: torch.from_numpy(oracle.packbits(grid.contiguous().numpy(), thresh))

    class Toy(foc_renderer.NeRFRenderer):
        def density(self, x):
            return {'sigma': poly_sigma(x)}

    m = Toy(bound=bound, cuda_ray=True, density_thresh=0.01)
    C = m.cascade
    m.grid_size = H
    m.density_grid = torch.zeros(C, H ** 3)
    m.density_bitfield = torch.zeros(C * H ** 3 // 8, dtype=torch.uint8)
    out = dict(H=np.int32(H), bound=np.float32(bound), cascade=np.int32(C), decay=np.float32(0.95), density_thresh=np.float32(m.density_thresh),
               density_scale=np.float32(m.density_scale))
    # ---- mark_untrained_grid
    g = torch.Generator().manual_seed(7)
    th = torch.rand(5, generator=g) * (np.pi / 3) + np.pi / 3
    ph = torch.rand(5, generator=g) * 2 * np.pi
    centers = torch.stack([torch.sin(th) * torch.sin(ph), torch.cos(th), torch.sin(th) * torch.cos(ph)], -1) * 2.2
    fw = -centers / centers.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, -1.0, 0.0]).expand(5, 3)
    rt = torch.cross(fw, up, dim=-1); rt = rt / rt.norm(dim=-1, keepdim=True)
    up2 = torch.cross(rt, fw, dim=-1); up2 = up2 / up2.norm(dim=-1, keepdim=True)
    poses = torch.eye(4).repeat(5, 1, 1)
    poses[:, :3, 0], poses[:, :3, 1], poses[:, :3, 2], poses[:, :3, 3] = rt, up2, fw, centers
    intr = (700.0, 650.0, 200.0, 180.0)                     # a narrow frustum, so that some cells are seen by nobody
    m.density_grid.fill_(1.0)
    out.update(mark_poses=poses.numpy(), mark_intrinsics=np.array(intr, np.float32), mark_grid_before=m.density_grid.numpy().copy())
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        m.mark_untrained_grid(poses, intr)
    out['mark_grid_after'] = m.density_grid.numpy().copy()
    # ---- two full sweeps (iter_density < 16), jitter pinned to the cell centre
    m.density_grid.zero_()
    m.density_grid[0, ::9] = -1.0                           # some untrained cells, as mark_untrained_grid leaves them
    out['sweep_grid_before'] = m.density_grid.numpy().copy()
    orig_rand_like, orig_randint = torch.rand_like, torch.randint
    torch.rand_like = lambda t, **k: torch.full_like(t, 0.5)
    try:
        for it in range(2):
            m.update_extra_state()
            out[f'sweep{it}_grid'] = m.density_grid.numpy().copy()
            out[f'sweep{it}_mean'] = np.float32(m.mean_density)
            out[f'sweep{it}_bits'] = m.density_bitfield.numpy().copy()
        # ---- one steady-state update (iter_density >= 16), its torch.randint draws recorded
        m.iter_density = 16
        drawn = []

        def recording_randint(*a, **k):
            t = orig_randint(*a, **k)
            drawn.append(t.clone())
            return t
        torch.randint = recording_randint
        torch.manual_seed(11)
        n_occ = [int((m.density_grid[c] > 0).sum()) for c in range(C)]
        m.update_extra_state()
    finally:
        torch.rand_like, torch.randint = orig_rand_like, orig_randint
    assert len(drawn) == 2 * C
    out['steady_n_occ'] = np.array(n_occ, np.int32)
    out['steady_coords'] = np.stack([drawn[2 * c].numpy() for c in range(C)]).astype(np.uint8)          # [C, N, 3]
    out['steady_pick'] = np.stack([drawn[2 * c + 1].numpy() for c in range(C)]).astype(np.int32)      # [C, N] index into the occupied list
    out['steady_grid'] = m.density_grid.numpy().copy()
    out['steady_mean'] = np.float32(m.mean_density)
    out['steady_bits'] = m.density_bitfield.numpy().copy()
    return out


def combined_fixture(K=4, N=40, T=64):
    """COMBINED.py keeps its logic in methods of a class defined inside its `__main__` block and imports ultralytics, lpips, cv2, ...
    at the top, so the file cannot be imported. The two methods on the path — best_densities_and_colors_v3 (:247-251) and
    image_depth_generation (:141-200) — are located with `ast`, compiled from the reference file where it lies and run here with the
    objects they reach for: `raymarching` (near_far_from_aabb = this repo's oracle), `opt.num_steps`, `self.model`. The object loop is
    the one of :592-618 (first object initialises, later ones go through the select)."""
    import ast
    import textwrap
    path = os.path.join(REF, "COMBINED.py")
    src = open(path).read()
    tree = ast.parse(src)
    wanted = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in ("best_densities_and_colors_v3", "image_depth_generation"):
            wanted[node.name] = textwrap.dedent(ast.get_source_segment(src, node))
    assert len(wanted) == 2
    ns = {"torch": torch, "raymarching": rm, "opt": types.SimpleNamespace(num_steps=T)}
    for code in wanted.values():
        exec(compile(code, path, "exec"), ns)
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    me = types.SimpleNamespace(model=types.SimpleNamespace(aabb_train=aabb, aabb_infer=aabb, training=False, min_near=0.2))
    g = torch.Generator().manual_seed(5)
    o, d = make_rays(N, 9, 1)
    dens = torch.rand(K, 1, N, T, generator=g) * 3
    dens[dens < 0.6] = 0.0                                     # empty space, and therefore ties at zero between objects
    dens[2, 0, :, ::5] = dens[0, 0, :, ::5]                    # exact ties between objects 0 and 2: the first one must win
    rgbs = torch.rand(K, 1, N, T, 3, generator=g)
    max_d, max_rgb = None, None
    for k in range(K):                                         # COMBINED.py:592-618
        if max_d is None:
            max_d, max_rgb = dens[k], rgbs[k]
        else:
            max_d, max_rgb = ns["best_densities_and_colors_v3"](me, dens[k], max_d, rgbs[k], max_rgb)
    data = {"rays_o": o, "rays_d": d}
    out = dict(rays_o=o.numpy(), rays_d=d.numpy(), aabb=aabb.numpy(), min_near=np.float32(0.2), densities=dens.numpy(), rgbs=rgbs.numpy(),
               max_densities=max_d.numpy(), max_rgbs=max_rgb.numpy())
    for bg in ("white", "black"):
        img, dep = ns["image_depth_generation"](me, data, max_d.clone(), max_rgb.clone(), bg)
        out[f"image_{bg}"] = img.numpy()
        out[f"depth_{bg}"] = dep.numpy()
    nears, fars = _near_far(o, d, aabb, 0.2)
    out["nears"], out["fars"] = nears.numpy(), fars.numpy()
    return out


def editable_fixture(N_hw=(5, 8), T=32, chunk=16):
    """configs[4]: editable.py's per-view object loop (:640-700) on EIGHT objects, two views, with the edited object's ray offset.
    editable.py cannot be imported (moviepy, ultralytics, lpips, cv2 ... at the top; its logic sits in methods of a class defined inside
    the `__main__` block), so the methods on the path — modify_rays_for_object (:443-471), get_object_type_from_ckpt (:500-508),
    batch_run (:215-252), run (:511-…), best_densities_and_colors_v3 (:259-263), image_depth_generation — are located with `ast`,
    compiled from the reference file where it lies and bound to a stand-in object carrying what they reach for (`self.model`, `self.B/N/H/W`;
    globals `opt`, `model`, `device`, `raymarching` = this repo's oracle near/far, `F`). The objects are analytic fields; `model` (the
    script's one network whose weights it reloads per object) is switched to object k where the script calls load_checkpoint.
    Stored per view: the data rays, every object's modified rays, its raw per-sample field and what `run` returned (densities, rgbs
    masked by its own weights > 1e-10, the weights themselves), the merged field and both composites."""
    import ast
    import textwrap
    path = os.path.join(REF, "editable.py")
    src = open(path).read()
    tree = ast.parse(src)
    names = ("modify_rays_for_object", "get_object_type_from_ckpt", "batch_run", "run", "best_densities_and_colors_v3", "image_depth_generation")
    wanted = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in wanted:
            wanted[node.name] = textwrap.dedent(ast.get_source_segment(src, node))
    assert set(wanted) == set(names), sorted(wanted)
    H, W = N_hw
    N = H * W
    dev = torch.device("cpu")
    opt = types.SimpleNamespace(num_steps=T, upsample_steps=0, max_ray_batch=chunk, edit_object="bottle", offset_x=0.01, offset_y=0.01, offset_z=0.60)
    ns = {"torch": torch, "raymarching": rm, "opt": opt, "F": torch.nn.functional, "device": dev, "model": None}
    for code in wanted.values():
        exec(compile(code, path, "exec"), ns)
    Ed = type("Ed", (), {n: ns[n] for n in names})
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    me = Ed()
    me.model = types.SimpleNamespace(aabb_train=aabb, aabb_infer=aabb, training=False, min_near=0.2)
    # checkpoint names: the type is the first of ['book','chair','bottle','cup'] that is a SUBSTRING of the path (:500-508)
    ckpts = ["ws/book_a/ngp.pth", "ws/cup_a/ngp.pth", "ws/bottle_a/ngp.pth", "ws/box_a/ngp.pth", "ws/chair/ngp.pth", "ws/cupboard/ngp.pth",
             "ws/bottle_b/ngp.pth", "ws/notebook/ngp.pth"]
    K = len(ckpts)
    raw_log = {}

    class Toy:
        def __init__(self, k):
            g = torch.Generator().manual_seed(100 + k)
            self.k = k
            self.c = (torch.rand(3, generator=g) - 0.5) * 0.9
            self.amp = 20.0 + 10.0 * k
            self.width = 0.05 + 0.02 * k

        def density(self, x):
            r2 = ((x - self.c) ** 2).sum(-1)
            sig = self.amp * torch.exp(-r2 / self.width)
            sig = torch.where(sig < 0.05, torch.zeros_like(sig), sig)        # empty space: exact zeros, hence ties between objects
            return {'sigma': sig, 'geo_feat': x[..., :2] * 0.0}

        def color(self, x, d, yolo_details, mask=None, geo_feat=None, **kw):
            raw = 0.5 + 0.5 * torch.sin(3.0 * x + 0.7 * d + float(self.k))
            raw_log.setdefault(self.k, []).append(raw.clone())
            rgbs = torch.zeros(mask.shape[0], 3)
            rgbs[mask] = raw[mask]
            return rgbs

    toys = [Toy(k) for k in range(K)]
    toys[5].c = toys[1].c.clone(); toys[5].amp = toys[1].amp; toys[5].width = toys[1].width   # objects 1 and 5 have the SAME density: exact ties, 1 wins
    out = dict(K=np.int32(K), T=np.int32(T), chunk=np.int32(chunk), HW=np.array([H, W], np.int32), aabb=aabb.numpy(), min_near=np.float32(0.2),
               ckpts=np.array(ckpts), edit_object=np.array(opt.edit_object), offset=np.array([opt.offset_x, opt.offset_y, opt.offset_z], np.float64),
               object_types=np.array([str(me.get_object_type_from_ckpt(c)) for c in ckpts]))
    for v in range(2):
        o, d = make_rays(N, 40 + v, 1)
        data = {"rays_o": o, "rays_d": d}
        me.B, me.H, me.W, me.N = 1, H, W, N
        max_d = None
        mod_o, mod_d, dens_all, rgbs_all, raw_all = [], [], [], [], []
        for k, ck in enumerate(ckpts):                                        # editable.py:660-700
            ns["model"] = toys[k]                                             # self.load_checkpoint(self.ckpt)
            raw_log.clear()
            mo, md = me.modify_rays_for_object(o, d, None, ck)
            res = me.batch_run(mo, md, None, **vars(opt))
            dens, rgbs = res['densities'], res['rgbs']
            mod_o.append(mo.clone().numpy()); mod_d.append(md.clone().numpy())
            dens_all.append(dens.numpy().copy()); rgbs_all.append(rgbs.numpy().copy())
            raw_all.append(torch.cat(raw_log[k], 0).view(N, T, 3).numpy())
            if max_d is None:
                max_d, max_rgb = dens, rgbs
            else:
                max_d, max_rgb = me.best_densities_and_colors_v3(dens, max_d, rgbs, max_rgb)
        out.update({f"v{v}_rays_o": o.numpy(), f"v{v}_rays_d": d.numpy(), f"v{v}_mod_o": np.stack(mod_o), f"v{v}_mod_d": np.stack(mod_d),
                    f"v{v}_densities": np.stack(dens_all)[:, 0], f"v{v}_rgbs": np.stack(rgbs_all)[:, 0], f"v{v}_raw_rgbs": np.stack(raw_all),
                    f"v{v}_max_densities": max_d.numpy()[0], f"v{v}_max_rgbs": max_rgb.numpy()[0]})
        for bg in ("white", "black"):
            img, dep = me.image_depth_generation(data, max_d.clone(), max_rgb.clone(), bg)
            out[f"v{v}_image_{bg}"] = img.numpy()
            out[f"v{v}_depth_{bg}"] = dep.numpy()
        nears, fars = _near_far(o, d, aabb, 0.2)
        out[f"v{v}_nears"], out[f"v{v}_fars"] = nears.numpy(), fars.numpy()
        # every object's own near / far (of its modified rays) and compositing weights, to tell borderline mask decisions apart
        own = []
        for k in range(K):
            n_k, f_k = _near_far(torch.from_numpy(mod_o[k]), torch.from_numpy(mod_d[k]), aabb, 0.2)
            own.append(np.stack([n_k.numpy(), f_k.numpy()]))
        out[f"v{v}_own_near_far"] = np.stack(own)
    return out


def cpu_network_fixture(N=48, T=64, bound=2):
    """configs[0]: the reference's pure-PyTorch network CLASS (nerf/network.py:10-210: nn.Linear sigma / colour nets, trunc_exp, sigmoid,
    masked colour query) driven by the reference's own NeRFRenderer.run (nerf/renderer.py:126-238), on the CPU. What the reference lacks
    to do that is supplied from oracle/torch_cpu_nerf.py — the thing this fixture pins: `encoding.get_encoder` (the module nerf/network.py:5
    imports and the tree does not contain; a SMALL hash grid here — 8 levels, 2^12 rows — so that the parameters fit in the fixture) and
    near_far_from_aabb (CUDA-only in the reference). One adapter: FOC's run() calls `self.color(x, d, yolo_details, mask=...)`
    (nerf/renderer.py:187) but nerf/network.py's color() has no yolo_details parameter (only network_tcnn.py's has) — the subclass below
    drops that positional argument and calls the reference method.
    Stored: parameters, rays, eval-mode image / depth / weights_sum, and a training-mode loss with its parameter gradients."""
    from oracle import torch_cpu_nerf as tcn
    small = dict(num_levels=8, base_resolution=8, log2_hashmap_size=12, desired_resolution=128)
    enc_stub = types.ModuleType("encoding")
    enc_stub.get_encoder = lambda encoding, **kw: tcn.get_encoder(encoding, **{**kw, **(small if encoding == "hashgrid" else {})})
    sys.modules["encoding"] = enc_stub
    import nerf.network as ref_net
    assert ref_net.__file__.startswith(REF)
    saved_nf = rm.near_far_from_aabb
    rm.near_far_from_aabb = lambda o, d, aabb, min_near=0.2: tcn.near_far(o, d, aabb, min_near)
    try:
        class Net(ref_net.NeRFNetwork):
            def color(self, x, d, yolo_details=None, mask=None, geo_feat=None, **kw):
                return ref_net.NeRFNetwork.color(self, x, d, mask=mask, geo_feat=geo_feat)

        torch.manual_seed(123)
        m = Net(bound=bound)
        m.encoder.embeddings.data.uniform_(-0.6, 0.6)
        o, d = make_rays(N, 77, bound)
        out = dict(bound=np.float32(bound), T=np.int32(T), rays_o=o.numpy(), rays_d=d.numpy(), encoder_cfg=np.array([small[k] for k in
                   ("num_levels", "base_resolution", "log2_hashmap_size", "desired_resolution")], np.int32))
        for k, v in m.state_dict().items():
            out["param/" + k] = v.detach().numpy().copy()
        m.eval()
        with torch.no_grad():
            ev = m.run(o[None], d[None], None, num_steps=T, upsample_steps=0, bg_color=None, perturb=False)
        out.update(eval_image=ev["image"][0].numpy(), eval_depth=ev["depth"][0].numpy(), eval_weights_sum=ev["weights_sum"].numpy(),
                   eval_densities=ev["densities"].squeeze(-1).numpy(), eval_rgbs=ev["rgbs"].numpy())
        m.train()
        target = 0.5 + 0.5 * torch.sin(3.0 * d)
        tr = m.run(o[None], d[None], (torch.ones(1, N, T, dtype=torch.bool), None, None), num_steps=T, upsample_steps=0, bg_color=None, perturb=False)
        hit = tr["weights_sum"] > -1
        loss = torch.nn.functional.mse_loss(tr["image"][0][torch.isfinite(tr["depth"][0])], target[torch.isfinite(tr["depth"][0])])
        loss.backward()
        out.update(train_target=target.numpy(), train_loss=np.float32(loss.item()), train_outside=np.float32(tr["criterion_outside_mask"].item()),
                   grad_embeddings=m.encoder.embeddings.grad.numpy().copy())
        for i, lin in enumerate(m.sigma_net):
            out[f"grad_sigma_net_{i}"] = lin.weight.grad.numpy().copy()
        for i, lin in enumerate(m.color_net):
            out[f"grad_color_net_{i}"] = lin.weight.grad.numpy().copy()
        nears, fars = tcn.near_far(o, d, m.aabb_infer, m.min_near)
        n2, f2 = _near_far(o, d, m.aabb_infer, m.min_near)
        assert torch.equal(nears, n2) and torch.equal(fars, f2)            # torch restatement == C oracle on these rays
        out["nears"], out["fars"] = nears.numpy(), fars.numpy()
    finally:
        rm.near_far_from_aabb = saved_nf
    return out


def upsample_fixture(N=40, T=48, t=32, bound=1):
    """Hierarchical resampling (upsample_steps > 0): `sample_pdf` as the FOC renderer module defines it (nerf/renderer.py:13-46, imported)
    and the legacy renderer's `run` (legacy/nerf/renderer.py:125-254), whose module cannot be imported (its `from .utils import ...`
    pulls in imageio, cv2, mcubes, lpips, ...): the method is located with `ast`, compiled from the reference file where it lies and run
    with the names it reaches for — `raymarching` (near/far = this repo's oracle), `sample_pdf` (the imported one), `torch` — on an
    analytic field, in eval mode (deterministic strata)."""
    import ast
    import textwrap
    out = {}
    g = torch.Generator().manual_seed(31)
    bins = torch.sort(torch.rand(25, 20, generator=g) * 3, dim=-1).values
    w = torch.rand(25, 19, generator=g) ** 3
    w[3] = 0.0                                                      # an empty ray: uniform after the +1e-5
    w[4, :10] = 0.0
    out.update(pdf_bins=bins.numpy(), pdf_weights=w.numpy(), pdf_samples_det=foc_renderer.sample_pdf(bins, w, 16, det=True).numpy())
    path = os.path.join(REF, "legacy", "nerf", "renderer.py")
    src = open(path).read()
    run_src = None
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == "run":
            run_src = textwrap.dedent(ast.get_source_segment(src, node))
            break
    assert run_src is not None and "sample_pdf" in run_src
    ns = {"torch": torch, "raymarching": rm, "sample_pdf": foc_renderer.sample_pdf}
    exec(compile(run_src, path, "exec"), ns)
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32)

    class Toy:
        aabb_train = aabb_infer = aabb
        training, min_near, density_scale, bg_radius = False, 0.2, 1, -1

        def density(self, x):
            return {'sigma': sigma_field(x), 'geo_feat': x[..., :2] * 0.5}

        def color(self, x, d, mask=None, geo_feat=None, **kw):
            rgbs = torch.zeros(mask.shape[0], 3)
            rgbs[mask] = color_field(x[mask], d[mask]) * (0.5 + geo_feat[mask][..., :1].abs().clamp(max=0.5))
            return rgbs
    o, d = make_rays(N, 55, bound)
    with torch.no_grad():
        res = ns["run"](Toy(), o[None], d[None], num_steps=T, upsample_steps=t, bg_color=None, perturb=False)
        res0 = ns["run"](Toy(), o[None], d[None], num_steps=T, upsample_steps=0, bg_color=None, perturb=False)
    nears, fars = _near_far(o, d, aabb, 0.2)
    out.update(rays_o=o.numpy(), rays_d=d.numpy(), aabb=aabb.numpy(), T=np.int32(T), t=np.int32(t), nears=nears.numpy(), fars=fars.numpy(),
               image=res["image"][0].numpy(), depth=res["depth"][0].numpy(), weights_sum=res["weights_sum"].numpy(),
               image_coarse=res0["image"][0].numpy(), depth_coarse=res0["depth"][0].numpy())
    return out


def wrapper_fixture():
    """The reference's Python operator wrappers (gridencoder/grid.py GridEncoder + grid_encode, ffmlp/ffmlp.py FFMLP + ffmlp_forward)
    driven on the CPU with their pybind11 backends replaced by stubs of the same names that call this repo's oracle. What the fixture
    pins is the WRAPPER layer a drop-in has to reproduce: the level-offset table and per_level_scale of GridEncoder.__init__, the
    (x + bound) / (2 bound) normalisation, the [L,B,C] <-> [B,L*C] permutes, FFMLP's weight-blob size and seed-42 initialisation, its
    padding of the batch to a multiple of 128 and the slicing of the padded output. `_gridencoder` / `_ffmlp` are put into sys.modules
    BEFORE the packages are imported, so their JIT-building backend.py is never reached; `turtle` (ffmlp.py:2, a stray
    import that needs tkinter) is stubbed too."""
    ge = types.ModuleType("_gridencoder")

    def ge_fwd(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
        res = oracle.grid_encode_forward(inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(), D, C, L, float(S), H,
                                         dy_dx is not None, gridtype, align_corners, interp)
        if dy_dx is not None:
            outputs.copy_(torch.from_numpy(res[0])); dy_dx.copy_(torch.from_numpy(res[1]).reshape(dy_dx.shape))
        else:
            outputs.copy_(torch.from_numpy(res))

    def ge_bwd(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):
        res = oracle.grid_encode_backward(grad.numpy(), inputs.detach().numpy(), offsets.numpy(), embeddings.shape[0], D, C, L, float(S), H,
                                          dy_dx.numpy().reshape(B, L, D, C) if dy_dx is not None else None, gridtype, align_corners, interp)
        if dy_dx is not None:
            grad_embeddings.copy_(torch.from_numpy(res[0])); grad_inputs.copy_(torch.from_numpy(res[1]))
        else:
            grad_embeddings.copy_(torch.from_numpy(res))
    ge.grid_encode_forward, ge.grid_encode_backward = ge_fwd, ge_bwd
    sys.modules["_gridencoder"] = ge

    ff = types.ModuleType("_ffmlp")
    calls = {"allocate_splitk": [], "forward_B": [], "inference_B": []}

    def ff_fwd(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer, outputs):
        calls["forward_B"].append(int(B))
        out, fb = oracle.ffmlp_forward(inputs.detach().numpy(), weights.detach().half().numpy(), input_dim, hidden_dim, num_layers, activation)
        outputs.copy_(torch.from_numpy(out)); forward_buffer.copy_(torch.from_numpy(fb))

    def ff_inf(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference_buffer, outputs):
        calls["inference_B"].append(int(B))
        outputs.copy_(torch.from_numpy(oracle.ffmlp_forward(inputs.detach().numpy(), weights.detach().half().numpy(), input_dim, hidden_dim, num_layers,
                                                             activation, training=False)))

    def ff_bwd(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs,
               backward_buffer, grad_inputs, grad_weights):
        gw, gi, bb = oracle.ffmlp_backward(grad.numpy(), inputs.detach().numpy(), weights.detach().half().numpy(), forward_buffer.numpy(), input_dim, hidden_dim,
                                           num_layers, activation, bool(calc_grad_inputs))
        grad_weights.copy_(torch.from_numpy(gw).to(grad_weights.dtype)); backward_buffer.copy_(torch.from_numpy(bb))
        if calc_grad_inputs:
            grad_inputs.copy_(torch.from_numpy(gi))
    ff.ffmlp_forward, ff.ffmlp_inference, ff.ffmlp_backward = ff_fwd, ff_inf, ff_bwd
    ff.allocate_splitk = lambda n: calls["allocate_splitk"].append(int(n))
    ff.free_splitk = lambda: None
    sys.modules["_ffmlp"] = ff
    sys.modules.setdefault("turtle", types.SimpleNamespace(backward=None, forward=None))

    import gridencoder as ref_ge            # /root/reference/gridencoder (sys.path), backend = the stub above
    import ffmlp as ref_ff
    assert ref_ge.__file__.startswith(REF) and ref_ff.__file__.startswith(REF)
    out = {}
    # ---- GridEncoder: default FOC table (offsets only: 6.1 M rows are not stored) and a small one driven end to end
    big = ref_ge.GridEncoder(desired_resolution=2048)
    out["big_offsets"] = big.offsets.numpy()
    out["big_per_level_scale"] = np.float64(big.per_level_scale)
    torch.manual_seed(3)
    cfg = dict(input_dim=3, num_levels=8, level_dim=2, base_resolution=4, log2_hashmap_size=12, desired_resolution=96)
    enc = ref_ge.GridEncoder(**cfg)
    enc.embeddings.data.uniform_(-1, 1)
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(77, 3, generator=g) * 2 - 1) * 2.0            # bound 2
    x[5] = 2.0; x[6] = -2.0                                        # the faces of the box
    xg = x.clone().requires_grad_(True)
    y = enc(xg, bound=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update(ge_offsets=enc.offsets.numpy(), ge_per_level_scale=np.float64(enc.per_level_scale), ge_embeddings=enc.embeddings.detach().numpy(),
               ge_x=x.numpy(), ge_y=y.detach().numpy(), ge_gy=gy.numpy(), ge_grad_embeddings=enc.embeddings.grad.numpy(), ge_grad_x=xg.grad.numpy(),
               ge_output_dim=np.int32(enc.output_dim))
    # ---- FFMLP
    mlp = ref_ff.FFMLP(input_dim=32, output_dim=3, hidden_dim=64, num_layers=2)
    out["ff_weights"] = mlp.weights.detach().numpy()
    out["ff_allocate_splitk"] = np.array(calls["allocate_splitk"], np.int32)
    xin = (torch.randn(200, 32, generator=g) * 0.5).half()
    mlp.eval()
    with torch.no_grad():
        out["ff_y_eval"] = mlp(xin).numpy()
    mlp.train()
    xt = xin.clone().requires_grad_(True)
    yt = mlp(xt)
    gyt = (torch.randn(200, 3, generator=g) * 0.1).half()
    yt.backward(gyt)
    out.update(ff_x=xin.numpy(), ff_y_train=yt.detach().numpy(), ff_gy=gyt.numpy(), ff_grad_x=xt.grad.numpy(), ff_grad_w=mlp.weights.grad.numpy(),
               ff_forward_B=np.array(calls["forward_B"], np.int32), ff_inference_B=np.array(calls["inference_B"], np.int32))
    # a batch that is already a multiple of 128 still gets one more block of padding (ffmlp.py:157-159)
    mlp.eval()
    with torch.no_grad():
        mlp(xin[:128])
    out["ff_inference_B_aligned"] = np.int32(calls["inference_B"][-1])
    return out


def raymarching_wrapper_fixture(H=32, bound=2.0):
    """The reference's raymarching/raymarching.py wrappers (output allocation, the mean_count / align / force_all_rays sizing and
    slicing of march_rays_train, the padding of march_rays, the autograd plumbing of composite_rays_train) on an oracle-backed
    `_raymarching` stub. The wrappers move their inputs with `.cuda()`; for this CPU run Tensor.cuda is the identity and
    torch.cuda.empty_cache a no-op. perturb=False throughout (the noise comes from torch's generator). The package contains a
    prebuilt _raymarching*.so; it is never loaded: the name is already in sys.modules when the package is imported."""
    C = 2
    st = types.ModuleType("_raymarching")

    def t2n(t):
        return t.detach().numpy()

    def s_near_far(rays_o, rays_d, aabb, N, min_near, nears, fars):
        n, f = oracle.near_far_from_aabb(t2n(rays_o), t2n(rays_d), t2n(aabb), min_near)
        nears.copy_(torch.from_numpy(n)); fars.copy_(torch.from_numpy(f))

    def s_morton(coords, N, indices):
        indices.copy_(torch.from_numpy(oracle.morton3D(t2n(coords))))

    def s_morton_inv(indices, N, coords):
        coords.copy_(torch.from_numpy(oracle.morton3D_invert(t2n(indices))))

    def s_packbits(grid, N, thresh, bitfield):
        bitfield.copy_(torch.from_numpy(oracle.packbits(t2n(grid.contiguous()), thresh)))

    def s_march_train(rays_o, rays_d, grid, bnd, dt_gamma, max_steps, N, Cc, Hh, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
        x, d, dl, r, c = oracle.march_rays_train(t2n(rays_o), t2n(rays_d), t2n(grid), bnd, dt_gamma, max_steps, Cc, Hh, M, t2n(nears), t2n(fars), t2n(noises), t2n(counter))
        xyzs.copy_(torch.from_numpy(x)); dirs.copy_(torch.from_numpy(d)); deltas.copy_(torch.from_numpy(dl)); rays.copy_(torch.from_numpy(r))
        counter.copy_(torch.from_numpy(c))

    def s_comp_fwd(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        w, dp, im = oracle.composite_rays_train_forward(t2n(sigmas), t2n(rgbs), t2n(deltas), t2n(rays), N, T_thresh)
        weights_sum.copy_(torch.from_numpy(w)); depth.copy_(torch.from_numpy(dp)); image.copy_(torch.from_numpy(im))

    def s_comp_bwd(gws, gim, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs):
        gs, gc = oracle.composite_rays_train_backward(t2n(gws), t2n(gim), t2n(sigmas), t2n(rgbs), t2n(deltas), t2n(rays), t2n(weights_sum), t2n(image), T_thresh)
        grad_sigmas.copy_(torch.from_numpy(gs)); grad_rgbs.copy_(torch.from_numpy(gc))

    def s_march(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bnd, dt_gamma, max_steps, Cc, Hh, grid, nears, fars, xyzs, dirs, deltas, noises):
        x, d, dl = oracle.march_rays(n_alive, n_step, t2n(rays_alive), t2n(rays_t), t2n(rays_o), t2n(rays_d), bnd, dt_gamma, max_steps, Cc, Hh, t2n(grid),
                                     t2n(nears), t2n(fars), t2n(noises), M=xyzs.shape[0])
        xyzs.copy_(torch.from_numpy(x)); dirs.copy_(torch.from_numpy(d)); deltas.copy_(torch.from_numpy(dl))

    def s_comp(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        ra, rt, w, dp, im = oracle.composite_rays(n_alive, n_step, T_thresh, t2n(rays_alive), t2n(rays_t), t2n(sigmas), t2n(rgbs), t2n(deltas), t2n(weights_sum),
                                                  t2n(depth), t2n(image))
        rays_alive.copy_(torch.from_numpy(ra)); rays_t.copy_(torch.from_numpy(rt)); weights_sum.copy_(torch.from_numpy(w))
        depth.copy_(torch.from_numpy(dp)); image.copy_(torch.from_numpy(im))

    st.near_far_from_aabb, st.morton3D, st.morton3D_invert, st.packbits = s_near_far, s_morton, s_morton_inv, s_packbits
    st.march_rays_train, st.composite_rays_train_forward, st.composite_rays_train_backward = s_march_train, s_comp_fwd, s_comp_bwd
    st.march_rays, st.composite_rays = s_march, s_comp
    sys.modules["_raymarching"] = st
    saved = sys.modules.pop("raymarching")
    orig_cuda, orig_empty = torch.Tensor.cuda, torch.cuda.empty_cache
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    try:
        import raymarching as ref_rm
        assert ref_rm.__file__.startswith(REF)
        out = dict(H=np.int32(H), C=np.int32(C), bound=np.float32(bound))
        # occupancy: a ball of radius 0.7 in both cascades
        idx = torch.arange(H ** 3, dtype=torch.int32)
        coords = ref_rm.morton3D_invert(idx)
        out["morton_roundtrip_ok"] = np.bool_(torch.equal(ref_rm.morton3D(coords), idx))
        grid = torch.zeros(C, H ** 3)
        for cas in range(C):
            b = min(2 ** cas, bound)
            xyz = (2 * coords.float() / (H - 1) - 1) * (b - b / H)
            grid[cas] = (xyz.norm(dim=-1) < 0.7).float() * 5.0
        bitfield = ref_rm.packbits(grid, 0.5)
        out["grid"], out["bitfield"] = grid.numpy(), bitfield.numpy()
        o, d = make_rays(96, 21, bound * 0.9)
        aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound])
        nears, fars = ref_rm.near_far_from_aabb(o, d, aabb, 0.2)
        out.update(rays_o=o.numpy(), rays_d=d.numpy(), aabb=aabb.numpy(), nears=nears.numpy(), fars=fars.numpy())
        # march_rays_train: first epochs (mean_count <= 0: slice to the used count rounded up to 128), steady state (M = mean_count rounded
        # up, here too small so that rays are dropped), force_all_rays
        for name, kw in (("first", dict(mean_count=-1, align=128)), ("steady", dict(mean_count=700, align=128)), ("force", dict(mean_count=700, align=128, force_all_rays=True)),
                         ("noalign", dict(mean_count=-1, align=-1))):
            counter = torch.zeros(2, dtype=torch.int32)
            xyzs, dirs, deltas, rays = ref_rm.march_rays_train(o, d, bound, bitfield, C, H, nears, fars, counter, kw.get("mean_count", -1), False, kw.get("align", -1),
                                                               kw.get("force_all_rays", False), 1 / 128, 256)
            out.update({f"mt_{name}_xyzs": xyzs.numpy(), f"mt_{name}_dirs": dirs.numpy(), f"mt_{name}_deltas": deltas.numpy(), f"mt_{name}_rays": rays.numpy(),
                        f"mt_{name}_counter": counter.numpy()})
        # composite_rays_train with autograd through the wrapper
        xyzs, deltas, rays = torch.from_numpy(out["mt_first_xyzs"]), torch.from_numpy(out["mt_first_deltas"]), torch.from_numpy(out["mt_first_rays"])
        g = torch.Generator().manual_seed(8)
        sig = (torch.rand(xyzs.shape[0], generator=g) * 8).requires_grad_(True)
        rgb = torch.rand(xyzs.shape[0], 3, generator=g).requires_grad_(True)
        ws, dep, img = ref_rm.composite_rays_train(sig, rgb, deltas, rays, 1e-4)
        gws, gimg = torch.rand(ws.shape, generator=g), torch.rand(img.shape, generator=g)
        (ws * gws).sum().add((img * gimg).sum()).backward()
        out.update(ct_sigmas=sig.detach().numpy(), ct_rgbs=rgb.detach().numpy(), ct_ws=ws.detach().numpy(), ct_depth=dep.detach().numpy(), ct_image=img.detach().numpy(),
                   ct_gws=gws.numpy(), ct_gimg=gimg.numpy(), ct_grad_sigmas=sig.grad.numpy(), ct_grad_rgbs=rgb.grad.numpy())
        # one inference iteration: march_rays (padded to 128) + composite_rays (in place)
        N = o.shape[0]
        n_alive, n_step = N, 3
        rays_alive = torch.arange(N, dtype=torch.int32)
        rays_t = nears.clone()
        x2, d2, dl2 = ref_rm.march_rays(n_alive, n_step, rays_alive, rays_t, o, d, bound, bitfield, C, H, nears, fars, 128, False, 1 / 128, 256)
        sig2 = torch.rand(x2.shape[0], generator=g) * 6
        rgb2 = torch.rand(x2.shape[0], 3, generator=g)
        wsum, dpt, im = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
        ref_rm.composite_rays(n_alive, n_step, rays_alive, rays_t, sig2, rgb2, dl2, wsum, dpt, im, 1e-2)
        out.update(mi_xyzs=x2.numpy(), mi_dirs=d2.numpy(), mi_deltas=dl2.numpy(), mi_sigmas=sig2.numpy(), mi_rgbs=rgb2.numpy(), mi_rays_alive=rays_alive.numpy(),
                   mi_rays_t=rays_t.numpy(), mi_ws=wsum.numpy(), mi_depth=dpt.numpy(), mi_image=im.numpy())
        # ---- NeRFRenderer.run_cuda (nerf/renderer.py:243-352) on top of these wrappers: the training branch (march_rays_train ->
        # network -> composite_rays_train) and the inference loop (n_step schedule, alive-ray filtering), analytic network, perturb off
        class ToyField(foc_renderer.NeRFRenderer):
            def forward(self, x, d):
                return poly_sigma(x), 0.5 + 0.25 * (x[..., :3] * 0.5 + d)

        old_rm = foc_renderer.raymarching
        foc_renderer.raymarching = ref_rm
        try:
            m = ToyField(bound=bound, cuda_ray=True)
            m.grid_size = H
            m.density_grid = grid.clone()
            m.density_bitfield = bitfield.clone()
            o2, d2r = make_rays(80, 33, bound * 0.9)
            m.train()
            tr0 = m.run_cuda(o2[None], d2r[None], dt_gamma=1 / 128, bg_color=None, perturb=False, force_all_rays=False, max_steps=256, T_thresh=1e-4)
            m.mean_count = 640                                       # what update_extra_state would set; rays beyond M are dropped
            tr1 = m.run_cuda(o2[None], d2r[None], dt_gamma=1 / 128, bg_color=0.25, perturb=False, force_all_rays=False, max_steps=256, T_thresh=1e-4)
            m.eval()
            with torch.no_grad():
                ev = m.run_cuda(o2[None], d2r[None], dt_gamma=1 / 128, bg_color=None, perturb=False, max_steps=256, T_thresh=1e-4)
            out.update(rc_rays_o=o2.numpy(), rc_rays_d=d2r.numpy(), rc_step_counter=m.step_counter.numpy().copy(),
                       rc_train0_image=tr0["image"][0].detach().numpy(), rc_train0_depth=tr0["depth"][0].detach().numpy(), rc_train0_ws=tr0["weights_sum"].detach().numpy(),
                       rc_train1_image=tr1["image"][0].detach().numpy(), rc_train1_depth=tr1["depth"][0].detach().numpy(),
                       rc_eval_image=ev["image"][0].numpy(), rc_eval_depth=ev["depth"][0].numpy())
        finally:
            foc_renderer.raymarching = old_rm
    finally:
        torch.Tensor.cuda, torch.cuda.empty_cache = orig_cuda, orig_empty
        sys.modules["raymarching"] = saved
    return out


def main():
    # trunc_exp
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(256, generator=g) * 8).requires_grad_(True)
    y = trunc_exp(x)
    gy = torch.randn(256, generator=g)
    y.backward(gy)
    np.savez_compressed(os.path.join(HERE, "trunc_exp.npz"), x=x.detach().numpy(), y=y.detach().numpy(), gy=gy.numpy(), gx=x.grad.numpy())

    np.savez_compressed(os.path.join(HERE, "run_foc.npz"), **run_reference(foc_renderer, True, bound=1, N=96, T=128, seed=1))
    np.savez_compressed(os.path.join(HERE, "run_foc_b2.npz"), **run_reference(foc_renderer, True, bound=2, N=64, T=512, seed=2))
    np.savez_compressed(os.path.join(HERE, "grid_maintenance.npz"), **grid_maintenance())
    np.savez_compressed(os.path.join(HERE, "combined.npz"), **combined_fixture())
    np.savez_compressed(os.path.join(HERE, "editable.npz"), **editable_fixture())
    np.savez_compressed(os.path.join(HERE, "cpu_network.npz"), **cpu_network_fixture())
    np.savez_compressed(os.path.join(HERE, "upsample.npz"), **upsample_fixture())
    np.savez_compressed(os.path.join(HERE, "wrappers.npz"), **wrapper_fixture())
    np.savez_compressed(os.path.join(HERE, "raymarching_wrappers.npz"), **raymarching_wrapper_fixture())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
