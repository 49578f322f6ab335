"""MI355X-native differentiable heightfield intersector: host-side mirror of the
reference's Shape plugin interface over the C ABI of libhf.so (include/hf.h)."""
from . import build, workload  # noqa: F401
from ._capi import HfError  # noqa: F401
from .shape import (Adam, Frame3f, Heightfield, ParamFlags, PreliminaryIntersection3f, Ray3f, RayFlags,  # noqa: F401
                    SurfaceInteraction3f, allreduce_gradient, direct_lighting, film_gaussian, point_lighting, reparameterize_ray)
