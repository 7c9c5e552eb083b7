"""Drop-in for the reference's `raymarching` package: `import raymarching` resolves here when
focnerf_amd/dropin is on PYTHONPATH ahead of the reference's own package directory."""
from focnerf_amd.raymarching import *  # noqa: F401,F403
from focnerf_amd.raymarching import (near_far_from_aabb, sph_from_ray, morton3D, morton3D_invert, packbits,  # noqa: F401
                                     march_rays_train, composite_rays_train, march_rays, composite_rays, compact_alive)
