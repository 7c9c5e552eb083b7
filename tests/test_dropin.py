"""CPU: the reference's own caller modules import and construct on top of this repo's ops when
focnerf_amd/dropin is put on sys.path. Needs the reference tree, so it only runs in
the build container; nothing is copied, and no bytecode is written into the reference."""
import os
import sys
import types

import pytest

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_network_ff_constructs_on_dropin_ops():
    import torch
    sys.dont_write_bytecode = True
    dropin = os.path.join(REPO, "focnerf_amd", "dropin")
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    try:
        for k in [k for k in sys.modules if k.split(".")[0] in ("nerf", "raymarching", "gridencoder", "freqencoder", "ffmlp", "encoding", "activation")]:
            del sys.modules[k]
        sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))       # absent third-party viewer lib
        sys.path.insert(0, REF)
        sys.path.insert(0, dropin)                                            # ahead of the reference's CUDA packages
        import raymarching
        assert raymarching.march_rays_train.__module__ == "focnerf_amd.raymarching"
        from nerf.network_ff import NeRFNetwork                               # the reference file, unmodified
        net = NeRFNetwork(bound=2, cuda_ray=True)
        assert type(net.encoder).__module__ == "focnerf_amd.gridencoder"
        assert type(net.sigma_net).__module__ == "focnerf_amd.ffmlp"
        assert net.in_dim == 32 and net.in_dim_color == 32
        assert net.sigma_net.weights.numel() == 64 * (32 + 64 + 16)
        assert net.color_net.weights.numel() == 64 * (32 + 64 * 2 + 16)
        assert net.encoder.embeddings.shape == (6328848, 2)                   # bound 2 table
        assert net.density_bitfield.numel() == 2 * 128 ** 3 // 8
        assert len(net.get_params(1e-2)) == 4
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved_mods:
                del sys.modules[k]
