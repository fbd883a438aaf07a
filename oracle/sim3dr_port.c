/*
 * CPU oracle for the Sim3DR path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the algorithms in the reference's only native file,
 * Sim3DR/lib/rasterize_kernel.cpp. It exists so that the oracle is available on machines without
 * /root/reference (the GPU box) and so that every arithmetic step the HIP kernels must reproduce is
 * written down once, explicitly, in evaluation order. It is validated bit-for-bit against the
 * reference's own C++ (oracle/_ref/libsim3dr_ref.so, built by oracle/Makefile) in
 * tests/test_sim3dr_oracle.py, including the known-answer inputs of Sim3DR/tests/test.cpp:10-48.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off (no -march, no -ffast-math): every float operation is an
 * individually rounded IEEE binary32 op, as in the reference's SSE2 build (Sim3DR/setup.py:12-18).
 *
 * Function <-> reference map
 *   port_point_weight        get_point_weight      rasterize_kernel.cpp:54-82
 *   port_point_in_tri        is_point_in_tri       rasterize_kernel.cpp:26-52
 *   port_get_tri_normal      _get_tri_normal       rasterize_kernel.cpp:87-120
 *   port_get_ver_normal      _get_ver_normal       rasterize_kernel.cpp:125-153
 *   port_get_normal          _get_normal           rasterize_kernel.cpp:158-215
 *   port_rasterize           _rasterize            rasterize_kernel.cpp:219-292
 *   port_rasterize_triangles _rasterize_triangles  rasterize_kernel.cpp:295-353
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>

/* (int) of a float as x86-64 `cvttss2si` does it: out-of-range and NaN give INT_MIN. The reference's
 * `(int) ceil(...)` / `(int) floor(...)` (rasterize_kernel.cpp:246-250) inherit exactly that. */
static int f2i_x86(float f) {
    if (f >= -2147483648.0f && f < 2147483648.0f) return (int)f;
    return INT_MIN;
}

/* std::min / std::max on floats: `(b < a) ? b : a` and `(a < b) ? b : a`. */
static float minf2(float a, float b) { return (b < a) ? b : a; }
static float maxf2(float a, float b) { return (a < b) ? b : a; }
static int mini2(int a, int b) { return (b < a) ? b : a; }
static int maxi2(int a, int b) { return (a < b) ? b : a; }

typedef struct {
    float u, v, inv;
} uv_t;

/* Shared core of is_point_in_tri / get_point_weight: edge vectors from p0, five dot products,
 * Cramer inverse (0 when the determinant is exactly 0). */
static uv_t bary_uv(float px, float py, float x0, float y0, float x1, float y1, float x2, float y2) {
    float ax = x2 - x0, ay = y2 - y0; /* v0 = p2 - p0 */
    float bx = x1 - x0, by = y1 - y0; /* v1 = p1 - p0 */
    float cx = px - x0, cy = py - y0; /* v2 = p  - p0 */
    float d00 = ax * ax + ay * ay;
    float d01 = ax * bx + ay * by;
    float d02 = ax * cx + ay * cy;
    float d11 = bx * bx + by * by;
    float d12 = bx * cx + by * cy;
    float den = d00 * d11 - d01 * d01;
    uv_t r;
    r.inv = (den == 0) ? 0.0f : 1 / den;
    r.u = (d11 * d02 - d01 * d12) * r.inv;
    r.v = (d00 * d12 - d01 * d02) * r.inv;
    return r;
}

void port_point_weight(float *w, float px, float py, float x0, float y0, float x1, float y1, float x2, float y2) {
    uv_t r = bary_uv(px, py, x0, y0, x1, y1, x2, y2);
    w[0] = 1 - r.u - r.v;
    w[1] = r.v;
    w[2] = r.u;
}

int port_point_in_tri(float px, float py, float x0, float y0, float x1, float y1, float x2, float y2) {
    uv_t r = bary_uv(px, py, x0, y0, x1, y1, x2, y2);
    return (r.u >= 0) && (r.v >= 0) && (r.u + r.v < 1);
}

static void face_cross(const float *vert, const int *tri, float *n) {
    const float *a = vert + 3 * tri[0], *b = vert + 3 * tri[1], *c = vert + 3 * tri[2];
    float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    n[0] = e1y * e2z - e1z * e2y;
    n[1] = e1z * e2x - e1x * e2z;
    n[2] = e1x * e2y - e1y * e2x;
}

static void unit3(float *n) {
    float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (len <= 0) len = (float)1e-6;
    n[0] = n[0] / len;
    n[1] = n[1] / len;
    n[2] = n[2] / len;
}

void port_get_tri_normal(float *tri_normal, const float *vertices, const int *triangles, int ntri, int norm_flg) {
    for (int t = 0; t < ntri; ++t) {
        face_cross(vertices, triangles + 3 * t, tri_normal + 3 * t);
        if (norm_flg) unit3(tri_normal + 3 * t);
    }
}

/* ver_normal is ACCUMULATED INTO (the python wrapper pre-zeroes it, Sim3DR/Sim3DR.py:9). */
void port_get_ver_normal(float *ver_normal, const float *tri_normal, const int *triangles, int nver, int ntri) {
    for (int t = 0; t < ntri; ++t)
        for (int corner = 0; corner < 3; ++corner) {
            float *dst = ver_normal + 3 * triangles[3 * t + corner];
            dst[0] += tri_normal[3 * t + 0];
            dst[1] += tri_normal[3 * t + 1];
            dst[2] += tri_normal[3 * t + 2];
        }
    for (int v = 0; v < nver; ++v) unit3(ver_normal + 3 * v);
}

void port_get_normal(float *ver_normal, const float *vertices, const int *triangles, int nver, int ntri) {
    float *tn = (float *)malloc(sizeof(float) * 3 * (size_t)(ntri > 0 ? ntri : 1));
    port_get_tri_normal(tn, vertices, triangles, ntri, 0);
    port_get_ver_normal(ver_normal, tn, triangles, nver, ntri);
    free(tn);
}

typedef struct {
    int x0, x1, y0, y1;
} box_t;

static int tri_box(const float *p0, const float *p1, const float *p2, int h, int w, box_t *b) {
    b->x0 = maxi2(f2i_x86(ceilf(minf2(p0[0], minf2(p1[0], p2[0])))), 0);
    b->x1 = mini2(f2i_x86(floorf(maxf2(p0[0], maxf2(p1[0], p2[0])))), w - 1);
    b->y0 = maxi2(f2i_x86(ceilf(minf2(p0[1], minf2(p1[1], p2[1])))), 0);
    b->y1 = mini2(f2i_x86(floorf(maxf2(p0[1], maxf2(p1[1], p2[1])))), h - 1);
    return !(b->x1 < b->x0 || b->y1 < b->y0);
}

/* float -> unsigned char the way gcc/x86-64 does `(unsigned char) f`: cvttss2si then keep 8 bits. */
static unsigned char f2u8_x86(float f) { return (unsigned char)(f2i_x86(f) & 0xff); }

void port_rasterize(unsigned char *image, const float *vertices, const int *triangles, const float *colors,
                    float *depth_buffer, int ntri, int h, int w, int c, float alpha, int reverse) {
    for (int t = 0; t < ntri; ++t) {
        const int i0 = triangles[3 * t], i1 = triangles[3 * t + 1], i2 = triangles[3 * t + 2];
        const float *p0 = vertices + 3 * i0, *p1 = vertices + 3 * i1, *p2 = vertices + 3 * i2;
        box_t b;
        if (!tri_box(p0, p1, p2, h, w, &b)) continue;
        for (int y = b.y0; y <= b.y1; ++y)
            for (int x = b.x0; x <= b.x1; ++x) {
                float wgt[3];
                port_point_weight(wgt, (float)x, (float)y, p0[0], p0[1], p1[0], p1[1], p2[0], p2[1]);
                if (!(wgt[2] > 0 && wgt[1] > 0 && wgt[0] > 0)) continue; /* strictly inside */
                float z = wgt[0] * p0[2] + wgt[1] * p1[2] + wgt[2] * p2[2];
                if (!(z > depth_buffer[y * w + x])) continue;
                int row = reverse ? (h - 1 - y) : y;
                unsigned char *px = image + ((size_t)row * w + x) * c;
                for (int k = 0; k < c; ++k) {
                    float col = wgt[0] * colors[c * i0 + k] + wgt[1] * colors[c * i1 + k] + wgt[2] * colors[c * i2 + k];
                    px[k] = f2u8_x86((1 - alpha) * px[k] + alpha * 255 * col);
                }
                depth_buffer[y * w + x] = z;
            }
    }
}

/* TEST DIAGNOSTIC (no counterpart in the reference): port_rasterize that ALSO records, per written byte, the float it
 * was cast from -- `(1 - alpha) * image + alpha * 255 * p_color` of rasterize_kernel.cpp:276-281 -- into
 * `precast` [h][w][c] (untouched where nothing is drawn). tests/render_checks.py uses it to show that a byte which
 * differs from the reference's sits on a truncation boundary. Same loop, same expressions as port_rasterize; the test
 * suite holds its byte output to port_rasterize's. */
void port_rasterize_precast(unsigned char *image, float *precast, const float *vertices, const int *triangles,
                            const float *colors, float *depth_buffer, int ntri, int h, int w, int c, float alpha, int reverse) {
    for (int t = 0; t < ntri; ++t) {
        const int i0 = triangles[3 * t], i1 = triangles[3 * t + 1], i2 = triangles[3 * t + 2];
        const float *p0 = vertices + 3 * i0, *p1 = vertices + 3 * i1, *p2 = vertices + 3 * i2;
        box_t b;
        if (!tri_box(p0, p1, p2, h, w, &b)) continue;
        for (int y = b.y0; y <= b.y1; ++y)
            for (int x = b.x0; x <= b.x1; ++x) {
                float wgt[3];
                port_point_weight(wgt, (float)x, (float)y, p0[0], p0[1], p1[0], p1[1], p2[0], p2[1]);
                if (!(wgt[2] > 0 && wgt[1] > 0 && wgt[0] > 0)) continue;
                float z = wgt[0] * p0[2] + wgt[1] * p1[2] + wgt[2] * p2[2];
                if (!(z > depth_buffer[y * w + x])) continue;
                int row = reverse ? (h - 1 - y) : y;
                unsigned char *px = image + ((size_t)row * w + x) * c;
                float *pf = precast + ((size_t)row * w + x) * c;
                for (int k = 0; k < c; ++k) {
                    float col = wgt[0] * colors[c * i0 + k] + wgt[1] * colors[c * i1 + k] + wgt[2] * colors[c * i2 + k];
                    pf[k] = (1 - alpha) * px[k] + alpha * 255 * col;
                    px[k] = f2u8_x86(pf[k]);
                }
                depth_buffer[y * w + x] = z;
            }
    }
}

void port_rasterize_triangles(const float *vertices, const int *triangles, float *depth_buffer, int *triangle_buffer,
                              float *barycentric_weight, int ntri, int h, int w) {
    for (int t = 0; t < ntri; ++t) {
        const float *p0 = vertices + 3 * triangles[3 * t];
        const float *p1 = vertices + 3 * triangles[3 * t + 1];
        const float *p2 = vertices + 3 * triangles[3 * t + 2];
        box_t b;
        if (!tri_box(p0, p1, p2, h, w, &b)) continue;
        for (int y = b.y0; y <= b.y1; ++y)
            for (int x = b.x0; x <= b.x1; ++x) {
                if (!port_point_in_tri((float)x, (float)y, p0[0], p0[1], p1[0], p1[1], p2[0], p2[1])) continue;
                float wgt[3];
                port_point_weight(wgt, (float)x, (float)y, p0[0], p0[1], p1[0], p1[1], p2[0], p2[1]);
                float z = wgt[0] * p0[2] + wgt[1] * p1[2] + wgt[2] * p2[2];
                if (!(z > depth_buffer[y * w + x])) continue;
                depth_buffer[y * w + x] = z;
                triangle_buffer[y * w + x] = t;
                barycentric_weight[(y * w + x) * 3 + 0] = wgt[0];
                barycentric_weight[(y * w + x) * 3 + 1] = wgt[1];
                barycentric_weight[(y * w + x) * 3 + 2] = wgt[2];
            }
    }
}
