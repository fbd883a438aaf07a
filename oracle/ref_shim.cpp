// extern "C" doorway onto the reference's Sim3DR C++ (TEST INFRASTRUCTURE, not product code).
//
// Built by oracle/Makefile together with /root/reference/Sim3DR/lib/rasterize_kernel.cpp, compiled
// from where it lies (nothing is copied), into oracle/_ref/libsim3dr_ref.so. The reference exposes
// these functions with C++ linkage only (Sim3DR/lib/rasterize.h:84-100) and reaches them from Python
// through Cython (Sim3DR/lib/rasterize.pyx:44-102); ctypes needs unmangled names, hence this shim.
#include "rasterize.h"

extern "C" {

int ref_is_point_in_tri(float px, float py, float x0, float y0, float x1, float y1, float x2, float y2) {
    return is_point_in_tri(Point(px, py), Point(x0, y0), Point(x1, y1), Point(x2, y2)) ? 1 : 0;
}

void ref_get_point_weight(float *weight, float px, float py, float x0, float y0, float x1, float y1, float x2,
                          float y2) {
    get_point_weight(weight, Point(px, py), Point(x0, y0), Point(x1, y1), Point(x2, y2));
}

void ref_get_tri_normal(float *tri_normal, float *vertices, int *triangles, int ntri, int norm_flg) {
    _get_tri_normal(tri_normal, vertices, triangles, ntri, norm_flg != 0);
}

void ref_get_ver_normal(float *ver_normal, float *tri_normal, int *triangles, int nver, int ntri) {
    _get_ver_normal(ver_normal, tri_normal, triangles, nver, ntri);
}

void ref_get_normal(float *ver_normal, float *vertices, int *triangles, int nver, int ntri) {
    _get_normal(ver_normal, vertices, triangles, nver, ntri);
}

void ref_rasterize_triangles(float *vertices, int *triangles, float *depth_buffer, int *triangle_buffer,
                             float *barycentric_weight, int ntri, int h, int w) {
    _rasterize_triangles(vertices, triangles, depth_buffer, triangle_buffer, barycentric_weight, ntri, h, w);
}

void ref_rasterize(unsigned char *image, float *vertices, int *triangles, float *colors, float *depth_buffer,
                   int ntri, int h, int w, int c, float alpha, int reverse) {
    _rasterize(image, vertices, triangles, colors, depth_buffer, ntri, h, w, c, alpha, reverse != 0);
}

}  // extern "C"
