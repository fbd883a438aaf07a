"""Drop-in for the reference's `Sim3DR` package (Sim3DR/__init__.py:3-4): `get_normal`, `rasterize`,
`RenderPipeline`, running on the MI355X through libdad3d_hip.so, plus the batched device-resident
`Mesh` the reference has no counterpart for."""
from .Sim3DR import get_normal, rasterize, rasterize_triangles  # noqa: F401
from .lighting import RenderPipeline  # noqa: F401
from .mesh import Mesh  # noqa: F401
