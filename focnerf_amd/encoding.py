"""encoding.get_encoder — the module the reference's networks import (nerf/network_ff.py:5,
nerf/network.py:5) but the reference tree does not contain. Signature follows
torch-ngp's: returns (encoder module, output_dim)."""

from .gridencoder import GridEncoder
from .freqencoder import FreqEncoder
from .shencoder import SHEncoder


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, **kwargs):
    if encoding == 'None':
        return (lambda x, **kw: x), input_dim
    if encoding == 'frequency':
        encoder = FreqEncoder(input_dim=input_dim, degree=multires)
    elif encoding == 'sphere_harmonics':
        encoder = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding == 'hashgrid':
        encoder = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                              log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution, gridtype='hash',
                              align_corners=align_corners)
    elif encoding == 'tiledgrid':
        encoder = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                              log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution, gridtype='tiled',
                              align_corners=align_corners)
    else:
        raise NotImplementedError('Unknown encoding mode, choose from [None, frequency, sphere_harmonics, hashgrid, tiledgrid]')
    return encoder, encoder.output_dim
