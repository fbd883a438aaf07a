"""`RenderPipeline` of the reference (Sim3DR/lighting.py:23-71) on the GPU: normals -> per-vertex Phong ->
z-buffer raster in three launches, vertices uploaded once, only the finished image comes back."""
from __future__ import annotations

import numpy as np
import torch

from .mesh import Mesh


def norm_vertices(vertices):  # lighting.py:9-14 (host helper kept for API parity)
    vertices -= vertices.min(0)[None, :]
    vertices /= vertices.max()
    vertices *= 2
    vertices -= vertices.max(0)[None, :] / 2
    return vertices


def convert_type(obj):  # lighting.py:17-20
    if isinstance(obj, (tuple, list)):
        return np.array(obj, dtype=np.float32)[None, :]
    return obj


class RenderPipeline(object):
    def __init__(self, **kwargs):
        self.intensity_ambient = convert_type(kwargs.get("intensity_ambient", 0.3))
        self.intensity_directional = convert_type(kwargs.get("intensity_directional", 0.6))
        self.intensity_specular = convert_type(kwargs.get("intensity_specular", 0.1))
        self.specular_exp = kwargs.get("specular_exp", 5)
        self.color_ambient = convert_type(kwargs.get("color_ambient", (1, 1, 1)))
        self.color_directional = convert_type(kwargs.get("color_directional", (1, 1, 1)))
        self.light_pos = convert_type(kwargs.get("light_pos", (0, 0, 5)))
        self.view_pos = convert_type(kwargs.get("view_pos", (0, 0, 5)))
        self._mesh = None
        self._mesh_key = None

    def update_light_pos(self, light_pos):
        self.light_pos = convert_type(light_pos)

    def _mesh_for(self, triangles: np.ndarray, nver: int) -> Mesh:
        key = (triangles.shape, nver, hash(triangles.tobytes()))
        if self._mesh is None or key != self._mesh_key:
            self._mesh, self._mesh_key = Mesh(triangles, nver), key
        return self._mesh

    def _light_kwargs(self):
        vec = lambda a: [float(x) for x in np.asarray(a, dtype=np.float32).reshape(-1)]  # noqa: E731
        return dict(ambient=float(self.intensity_ambient), directional=float(self.intensity_directional),
                    specular=float(self.intensity_specular), specular_exp=float(self.specular_exp),
                    color_ambient=vec(self.color_ambient), color_directional=vec(self.color_directional),
                    light_pos=vec(self.light_pos), view_pos=vec(self.view_pos))

    def __call__(self, vertices, triangles, bg, texture=None):
        mesh = self._mesh_for(triangles, vertices.shape[0])
        dev = mesh.torch_device
        v = torch.from_numpy(np.ascontiguousarray(vertices, dtype=np.float32)).to(dev)[None]
        img = torch.from_numpy(bg).to(dev)[None].contiguous()
        if texture is None and img.shape[-1] == 3:
            mesh.render(v, img, **self._light_kwargs())  # normals + Phong inside the raster's geometry kernel: two launches
        else:
            light = mesh.phong_light(v, None, **self._light_kwargs())  # vertex normals + Phong terms in one launch
            colors = light
            if texture is not None:
                tex = torch.from_numpy(np.ascontiguousarray(texture, dtype=np.float32)).to(dev)[None] * light
                texture[...] = tex[0].cpu().numpy()  # `texture *= light` is in place in the reference (lighting.py:69)
                colors = tex.contiguous()
            mesh.rasterize(v, colors, img)
        bg[...] = img[0].cpu().numpy()  # the reference renders into `bg` and returns it
        return bg
