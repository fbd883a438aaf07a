// C++-linkage doubles of the reference's Sim3DR entry points, prototype for prototype
// (Sim3DR/lib/rasterize.h:88-100), forwarding to the C ABI. With these exported, the reference's Cython
// binding (Sim3DR/lib/rasterize.pyx, `cdef extern from "rasterize.h"`) links against libdad3d_hip.so
// instead of rasterize_kernel.cpp without a source change -- see INTEGRATION.md.
#include "../../include/dad3d.h"

DAD3D_EXPORT void _get_tri_normal(float* tri_normal, float* vertices, int* triangles, int ntri, bool norm_flg) {
    dad3d_sim3dr_get_tri_normal(tri_normal, vertices, triangles, ntri, norm_flg ? 1 : 0);
}

DAD3D_EXPORT void _get_ver_normal(float* ver_normal, float* tri_normal, int* triangles, int nver, int ntri) {
    dad3d_sim3dr_get_ver_normal(ver_normal, tri_normal, triangles, nver, ntri);
}

DAD3D_EXPORT void _get_normal(float* ver_normal, float* vertices, int* triangles, int nver, int ntri) {
    dad3d_sim3dr_get_normal(ver_normal, vertices, triangles, nver, ntri);
}

DAD3D_EXPORT void _rasterize_triangles(float* vertices, int* triangles, float* depth_buffer, int* triangle_buffer,
                                       float* barycentric_weight, int ntri, int h, int w) {
    dad3d_sim3dr_rasterize_triangles(vertices, triangles, depth_buffer, triangle_buffer, barycentric_weight, ntri, h, w);
}

DAD3D_EXPORT void _rasterize(unsigned char* image, float* vertices, int* triangles, float* colors, float* depth_buffer,
                             int ntri, int h, int w, int c, float alpha, bool reverse) {
    dad3d_sim3dr_rasterize(image, vertices, triangles, colors, depth_buffer, ntri, h, w, c, alpha, reverse ? 1 : 0);
}
