"""Order of the rays of a view inside the library's 64-ray blocks.

A caller hands a view's rays over row by row (nerf/provider.py: `get_rays` flattens the H x W pixel grid), so 64 consecutive rays are a
64 x 1 strip of pixels. What a ray receives does not depend on its neighbours in the list — but how fast the hash-grid encoder runs
does: the 64 rays of a block are the 64 lanes of a wave at one depth, and the cache lines their corner rows share are what the gather
saves. An 8 x 8 pixel tile is a more compact set of points than a 64 x 1 strip at every level (measured on the 800 x 800 fixed-step render,
tools/time_render_fixed.py TILE=8x8: 44.8 -> 41.5 ms per view; 16x4 41.8, 4x16 42.7, 2x32 43.9).

`view_tiling(rays_d)` recognises a row-major pixel grid from the directions alone (inside a row consecutive steps point the same way,
at a row's end the step jumps back across the image) and returns the permutation that lists the rays tile by tile (the identity for a
ray set that is no such grid), computed on the device without a host round trip; `detect_image_width` / `tile_permutation` are the
same recognition and order as torch expressions (host-side, used by the tests and the bench's combined-render leg). The
staged fixed-step render walks the view in that order and puts the results back where the caller's rays were; nothing else changes (the
occupancy-grid loop gains nothing from it — 14.8 against 14.6 ms per view: its march and composite kernels read the rays' state through the
list — and keeps the caller's order). A wrong guess could
only cost speed, never a result — and a ray set that is not such a grid (a training batch, a hand-made list) is left as it is.
"""
import os

import torch

_PERMS = {}


def detect_image_width(rays_d):
    """rays_d [N,3] (one view) -> W if the rays are the rows of an H x W pixel grid (W >= 16, H >= 8), else None. One small device -> host copy."""
    n = rays_d.shape[0]
    if n < 4096 or rays_d.dim() != 2:
        return None
    step = (rays_d[1:] - rays_d[:-1]).float()
    # inside a row consecutive steps point the same way; the jump from a row's last pixel to the next row's first points back across the
    # image: it is anti-parallel to the step before it AND to the step after it
    turns = torch.nonzero((step[1:] * step[:-1]).sum(-1) < 0).view(-1)          # the only synchronisation: where the direction turns
    if turns.numel() == 0 or turns.numel() % 2:
        return None
    w = int(turns[0]) + 2                                                       # turn i = between steps i and i + 1; the jump is step w - 1
    h = n // w
    if w < 16 or n % w != 0 or h < 8 or turns.numel() != 2 * (h - 1):
        return None
    ends = torch.arange(1, h, device=rays_d.device) * w
    expect = torch.stack([ends - 2, ends - 1], 1).view(-1)
    return w if bool((turns == expect).all()) else None


def tile_permutation(n, w, device, th=8, tw=8):
    """Indices of the n = H * w rays of a row-major view listed tile by tile (th x tw pixels, row-major inside a tile and over the tiles)."""
    key = (n, w, th, tw, str(device))
    if key not in _PERMS:
        h = n // w
        yy, xx = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
        tiles_per_row = (w + tw - 1) // tw
        order = ((yy // th) * tiles_per_row + (xx // tw)) * (th * tw) + (yy % th) * tw + (xx % tw)
        _PERMS[key] = torch.argsort(order.reshape(-1), stable=True)
        if len(_PERMS) > 16:
            _PERMS.pop(next(iter(_PERMS)))
    return _PERMS[key]


def view_tiling(rays_d):
    """int64 [N] order in which the staged render walks the view's rays: 8 x 8 pixel tiles if `rays_d` is a row-major pixel grid, else the
    identity — found and built ON THE DEVICE (csrc/fixedstep.hip, foc_view_tile_order): the caller's thread never waits for the GPU, so
    the chunks of the next view are enqueued while the previous view is still being rendered. None: not applicable (FOC_RAY_TILES=0, fewer
    than 4096 rays, not on the GPU). A ray set of >= 4096 rays that is NO pixel grid gets the identity as a tensor (the decision lives on the
    device): its caller gathers the rays and scatters the results through it — 2 x 24 B per ray of extra traffic for random-ray evaluation,
    the price of not waiting for the GPU once per view; FOC_RAY_TILES=0 skips both."""
    shape = os.environ.get("FOC_RAY_TILES", "8x8")
    if shape in ("0", "", "off") or not rays_d.is_cuda or rays_d.dim() != 2 or rays_d.shape[0] < 4096:
        return None
    if rays_d.shape[1] != 3:
        raise ValueError(f"view_tiling: rays_d must be [N, 3] (got {tuple(rays_d.shape)})")
    try:
        th, tw = (int(v) for v in shape.lower().split("x"))
    except ValueError:
        raise ValueError(f"FOC_RAY_TILES must be 0 or <rows>x<columns> (got {shape!r})") from None
    if th < 1 or tw < 1:
        return None
    from ._lib import lib, ptr, stream_of, check
    d = rays_d.contiguous().float()
    n = d.shape[0]
    perm = torch.empty(n, dtype=torch.int64, device=d.device)
    state = torch.empty(4, dtype=torch.int32, device=d.device)
    check(lib.foc_view_tile_order(ptr(d), n, th, tw, ptr(perm), ptr(state), stream_of(d)), "view_tile_order")
    return perm
