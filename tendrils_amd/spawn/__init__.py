"""Host mirror of the reference's respawn helpers (src/spawn/): `init.spawner`, `ball.spawnBall`,
`pixels.PixelSpawner`.  Each wraps an opaque program (kernel family) + uniforms and calls
`tendrils.spawnShader(...)`, exactly like the reference objects wrap a gl-shader."""
from . import ball, geometry, init, pixels
from .geometry import GeometrySpawner, bright_sample_frag
from .ball import spawnBall
from .init import spawner
from .pixels import ImageBuffer, PixelSpawner, best_sample_frag, data_sample_frag, flow_sample_frag, pixels_frag

__all__ = ["init", "ball", "pixels", "spawner", "spawnBall", "PixelSpawner", "flow_sample_frag", "data_sample_frag",
           "best_sample_frag", "pixels_frag", "ImageBuffer", "geometry", "GeometrySpawner", "bright_sample_frag"]
