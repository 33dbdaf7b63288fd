"""ray-tracing_amd — MI355X (gfx950) trace loop behind the cft_ray_tracer scene/camera API.

The directory name carries a hyphen (it is fixed by the build contract), so import it with
    rt = importlib.import_module("ray-tracing_amd")
"""
from .api import (EPS, Camera, Cube, Device, Dielectric, LambertianModel, LightShape, Metal, Object,  # noqa: F401
                  ParallelLight, PhongModel, PointLight, SkyLight, Specular, Sphere, Square, Transparent,
                  Triangle, World, default_device, vec3_to_rgb)
from . import image_io, multi, scenes  # noqa: F401
from ._lib import CftrtError, LIB_PATH, SYMBOLS, load  # noqa: F401
