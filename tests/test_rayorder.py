"""Ray order inside the library's 64-ray blocks (focnerf_amd/rayorder.py): recognition of a row-major pixel grid from the ray directions,
the tile permutation, and — on the GPU — that the renderers give the caller the same image and depth with and without it."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focnerf_amd import rayorder, synthetic


def _view(h, w, seed=0):
    gen = torch.Generator().manual_seed(seed)
    poses = synthetic.rand_poses(1, "cpu", radius=2.5, generator=gen)
    o, d = synthetic.get_rays(poses, synthetic.intrinsics(h, w), h, w)
    return o[0], d[0]


def test_detects_the_width_of_a_row_major_view_and_nothing_else():
    for h, w in ((64, 64), (48, 96), (100, 60), (8, 600)):
        _, d = _view(h, w)
        assert rayorder.detect_image_width(d) == w, (h, w)
    _, d = _view(80, 80)
    g = torch.Generator().manual_seed(1)
    assert rayorder.detect_image_width(d[torch.randperm(6400, generator=g)]) is None            # a training batch: random pixels
    assert rayorder.detect_image_width(d[:4000]) is None                                         # too few rays
    assert rayorder.detect_image_width(d[: 80 * 60 + 17]) is None                                # a view cut inside a row
    assert rayorder.detect_image_width(d.flip(0)[:: 1].contiguous()) == 80                       # rows right to left are a grid as well
    two = torch.cat([_view(64, 72, 1)[1], _view(64, 72, 2)[1]])                                  # two views behind each other: a taller grid
    assert rayorder.detect_image_width(two) == 72


def test_tile_permutation_lists_every_ray_once_tile_by_tile():
    for h, w, th, tw in ((16, 24, 8, 8), (20, 30, 8, 8), (32, 32, 4, 16)):
        p = rayorder.tile_permutation(h * w, w, "cpu", th, tw)
        assert sorted(p.tolist()) == list(range(h * w))
        y, x = p // w, p % w
        if h % th == 0 and w % tw == 0:
            blocks = torch.stack([y // th, x // tw], 1).view(-1, th * tw, 2)
            assert bool((blocks == blocks[:, :1]).all()), "every run of th*tw rays lies in one tile"
        first = torch.stack([y[: th * tw] if h >= th else y, x[: th * tw]], 1)
        assert int(first[:, 0].max()) < th and int(first[:, 1].max()) < tw


@pytest.mark.gpu
def test_renders_are_the_same_with_and_without_the_tile_order(monkeypatch):
    import bench
    dev = torch.device("cuda", 0)
    poses, intr = bench.make_training_rays(dev, 1, 2, seed=3)
    o, d = synthetic.get_rays(poses[:1], intr, 96, 120)
    assert rayorder.view_tiling(d[0]) is not None
    m = bench.build_model(1, dev, seed=2).eval()
    kw = dict(staged=True, max_ray_batch=4096, num_steps=64, upsample_steps=0, perturb=False, fused=True, return_fields=False)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = m.render(o, d, **kw)
        monkeypatch.setenv("FOC_RAY_TILES", "0")
        b = m.render(o, d, **kw)
        monkeypatch.delenv("FOC_RAY_TILES")
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"].nan_to_num(), b["depth"].nan_to_num())
    assert (a["image"] < 0.99).any()


@pytest.mark.gpu
@pytest.mark.parametrize("fields", [False, True])
def test_staged_render_in_larger_pieces_is_the_callers_chunking_bit_for_bit(fields, monkeypatch):
    """The fused inference path walks a view in pieces of at least 16384 rays whatever `max_ray_batch` says (focnerf_amd/renderer.py):
    same image, depth and per-sample fields as the caller's own chunks (FOC_RENDER_MIN_CHUNK=0), ragged last piece included."""
    import bench
    dev = torch.device("cuda", 0)
    poses, intr = bench.make_training_rays(dev, 1, 2, seed=5)
    o, d = synthetic.get_rays(poses[:1], intr, 150, 141)                  # 21150 rays: one piece of 16384 + a ragged one; six chunks of 4096
    m = bench.build_model(1, dev, seed=3).eval()
    kw = dict(staged=True, max_ray_batch=4096, num_steps=32, upsample_steps=0, perturb=False, fused=True, return_fields=fields)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = m.render(o, d, **kw)
        monkeypatch.setenv("FOC_RENDER_MIN_CHUNK", "0")
        b = m.render(o, d, **kw)
    for k in ("image", "depth") + (("densities", "rgbs") if fields else ()):
        assert torch.equal(a[k].nan_to_num(), b[k].nan_to_num()), k
    assert (a["image"] < 0.99).any()


@pytest.mark.gpu
def test_device_side_tile_order_equals_the_host_recognition_and_permutation(monkeypatch):
    """foc_view_tile_order (no host round trip) against detect_image_width + tile_permutation: the same order for pixel grids (ragged last
    tiles, non-square tiles, two views behind each other), the identity for everything the host form rejects."""
    g = torch.Generator().manual_seed(1)
    cases = [_view(64, 64)[1], _view(70, 90)[1], _view(100, 61)[1], _view(8, 600)[1], _view(80, 80)[1].flip(0).contiguous(),
             torch.cat([_view(64, 72, 1)[1], _view(64, 72, 2)[1]]),
             _view(80, 80)[1][torch.randperm(6400, generator=g)], _view(80, 80)[1][: 80 * 60 + 17].contiguous(), _view(7, 700)[1]]
    for shape in ("8x8", "4x16", "3x5"):
        monkeypatch.setenv("FOC_RAY_TILES", shape)
        th, tw = (int(v) for v in shape.split("x"))
        for d in cases:
            w = rayorder.detect_image_width(d)
            want = rayorder.tile_permutation(d.shape[0], w, "cpu", th, tw) if w is not None else torch.arange(d.shape[0])
            got = rayorder.view_tiling(d.cuda())
            assert got is not None and torch.equal(got.cpu(), want), (shape, tuple(d.shape), w)
    assert rayorder.view_tiling(_view(60, 60)[1].cuda()) is None                                  # fewer than 4096 rays: left alone
