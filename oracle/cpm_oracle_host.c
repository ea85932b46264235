/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT (see cpm_oracle.h).
 *
 * cpm_oracle_host.c -- restatement of the two pieces of HOST arithmetic on the path that decide device inputs:
 *
 *   E2  the light's rectangle: 8 proxy-mesh vertices -> plane coordinates -> hull -> minimum-area rectangle
 *       (ref lightcl/pointplaneprojection.cpp:39-54, lightcl/convexhull2d.cpp:38-130,
 *        lightcl/orientedboundingbox2d.cpp:40-100).  It defines the emitted photon set.
 *   C2  the break points of |TF_new - TF_old| (ref importancesamplingcl/processors/
 *       minmaxuniformgrid3dimportanceclprocessor.cpp:364-524).  It defines the importance grid of a TF edit.
 *
 * This file FOLLOWS THE REFERENCE'S CONTROL FLOW statement by statement, on purpose: it is what the product's own,
 * differently built host code (host/cpm_hostmath.cpp) is checked against (tests/test_host_logic.py).
 *
 * PARITY UNPINNED: the reference has no tests for either function, and both lean on glm, which is not in the reference
 * tree.  glm's semantics as restated here, flagged [GLM]:
 *   normalize(v)            = v * (1 / sqrt(dot(v, v)))            (a zero vector gives NaNs)
 *   dot(a, b)               = a.x*b.x + a.y*b.y (+ a.z*b.z), left to right
 *   mix(x, y, a), a double  = (T)((double)x * (1 - a) + (double)y * a)   (GLSL's definition, evaluated in a's type)
 *   epsilonNotEqual(x,y,e)  = |x - y| >= e
 * and Inviwo's TFPrimitive ordering = by position [INVIWO].
 */
#include "cpm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;

static float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 sub3(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static v3 add3(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static v3 mul3(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static v3 norm3(v3 a) { return mul3(a, 1.0f / sqrtf(dot3(a, a))); }  /* [GLM] */
static v3 cross3(v3 a, v3 b) { v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; return r; }

static int cmp_xy(const void* pa, const void* pb) {
    const v2 *a = (const v2*)pa, *b = (const v2*)pb;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    return 0;
}

/* ref convexhull2d.cpp:56-58 */
static float left_of(v2 p0, v2 p1, v2 p) { return (p1.x - p0.x) * (p.y - p0.y) - (p.x - p0.x) * (p1.y - p0.y); }

/* ref convexhull2d.cpp:38-130.  hull must have room for n + 2 points; returns the number written.  The reference's
 * std::sort is not stable, but elements that compare equal are identical points, so the order is the same. */
int cpmo_convex_hull_2d(const float* xy, int n, float* hull_xy) {
    v2* pts = (v2*)malloc(sizeof(v2) * (size_t)(n > 0 ? n : 1));
    v2* hull = (v2*)hull_xy;
    int count = 0, i;
    memcpy(pts, xy, sizeof(v2) * (size_t)n);
    qsort(pts, (size_t)n, sizeof(v2), cmp_xy);
    if (n < 4) { /* :44-46 */
        memcpy(hull, pts, sizeof(v2) * (size_t)n);
        free(pts);
        return n;
    }
    int minmin = 0, minmax = 1; /* :62-69 */
    for (; minmax < n; ++minmax)
        if (pts[0].x != pts[minmax].x) break;
    --minmax;
    if (minmax == n - 1) { /* :71-79: all x equal */
        hull[count++] = pts[minmin];
        if (pts[minmax].y != pts[minmin].y) hull[count++] = pts[minmax];
        hull[count++] = pts[minmin];
        free(pts);
        return count;
    }
    int maxmin = n - 1, maxmax = n - 2; /* :81-88 (the reference's names: maxXMinYId, maxXMaxYId) */
    for (; maxmax >= 0; --maxmax)
        if (pts[n - 1].x > pts[maxmax].x) break;
    ++maxmax;
    hull[count++] = pts[minmin]; /* :91-105 lower hull */
    for (i = minmax + 1; i <= maxmin; ++i) {
        if (left_of(pts[minmin], pts[maxmin], pts[i]) >= 0 && i < maxmin) continue;
        while (count >= 2) {
            if (left_of(hull[count - 2], hull[count - 1], pts[i]) > 0) break;
            --count;
        }
        hull[count++] = pts[i];
    }
    if (maxmax != maxmin) hull[count++] = pts[maxmax]; /* :107-109 */
    int bottom = count - 1;                            /* :110-125 upper hull */
    for (i = maxmax; i > minmax; --i) {
        if (left_of(pts[maxmax], pts[minmax], pts[i]) >= 0 && i > minmax) continue;
        while (count - bottom >= 2) {
            if (left_of(hull[count - 2], hull[count - 1], pts[i]) > 0) break;
            --count;
        }
        hull[count++] = pts[i];
    }
    if (minmax != minmin) hull[count++] = pts[maxmin]; /* :126-127 */
    free(pts);
    return count;
}

/* ref orientedboundingbox2d.cpp:40-78: out = origin.xy, u.xy, v.xy */
void cpmo_minimum_bounding_rectangle(const float* hull_xy, int n, float out[6]) {
    const v2* h = (const v2*)hull_xy;
    float min_area = FLT_MAX;
    int i, j, k;
    memset(out, 0, 6 * sizeof(float));
    if (n <= 0) return;
    for (i = 0, j = n - 1; i < n; j = i, ++i) {
        v2 e = { h[i].x - h[j].x, h[i].y - h[j].y };
        float inv = 1.0f / sqrtf(e.x * e.x + e.y * e.y); /* [GLM] normalize */
        v2 e0 = { e.x * inv, e.y * inv };
        if (isnan(e0.x) || isnan(e0.y)) continue;
        v2 e1 = { -e0.y, e0.x };
        float min0 = 0.f, min1 = 0.f, max0 = 0.f, max1 = 0.f;
        for (k = 0; k < n; ++k) {
            v2 d = { h[k].x - h[j].x, h[k].y - h[j].y };
            float t = d.x * e0.x + d.y * e0.y;
            min0 = fminf(min0, t);
            max0 = fmaxf(max0, t);
            t = d.x * e1.x + d.y * e1.y;
            min1 = fminf(min1, t);
            max1 = fmaxf(max1, t);
        }
        float area = (max0 - min0) * (max1 - min1);
        if (area < min_area) {
            min_area = area;
            float a = fminf(min0, 0.f), b = fminf(min1, 0.f);
            /* vec2 + float*vec2 + float*vec2, left to right */
            out[0] = (h[j].x + a * e0.x) + b * e1.x;
            out[1] = (h[j].y + a * e0.y) + b * e1.y;
            out[2] = e0.x * (max0 - min0);
            out[3] = e0.y * (max0 - min0);
            out[4] = e1.x * (max1 - min1);
            out[5] = e1.y * (max1 - min1);
        }
    }
}

/* ref orientedboundingbox2d.cpp:80-100 with pointplaneprojection.cpp:39-54.  plane_normal must be normalised (the caller
 * normalises the light direction, directionallightsamplercl.cpp:60-63).  Plane::projectPoint is Inviwo's [INVIWO]: restated
 * with the arithmetic of projectPointsOnPlane, which IS in the tree.  out = origin.xyz, u.xyz, v.xyz */
void cpmo_fit_obb(const float* points_xyz, int n, const float plane_point[3], const float plane_normal[3], float out[9]) {
    const v3 pp = { plane_point[0], plane_point[1], plane_point[2] }, nn = { plane_normal[0], plane_normal[1], plane_normal[2] };
    const float d = dot3(nn, pp);
    v3 axis = { 0.f, 0.f, 0.f }, u, v;
    int i;
    if (fabsf(nn.x) > fabsf(nn.y)) axis.x = 1.f; else axis.y = 1.f;
    u = norm3(sub3(sub3(axis, mul3(nn, dot3(nn, axis) - d)), pp));
    v = norm3(cross3(nn, u));
    float* proj = (float*)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    float* hull = (float*)malloc(sizeof(float) * 2 * (size_t)(n + 2));
    for (i = 0; i < n; ++i) {
        const v3 p = { points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2] };
        const float dist = dot3(nn, p) - d;
        const v3 o2p = sub3(sub3(p, mul3(nn, dist)), pp);
        proj[2 * i] = dot3(u, o2p);
        proj[2 * i + 1] = dot3(v, o2p);
    }
    float r[6];
    const int nh = cpmo_convex_hull_2d(proj, n, hull);
    cpmo_minimum_bounding_rectangle(hull, nh, r);
    const v3 origin = add3(add3(pp, mul3(u, r[0])), mul3(v, r[1]));
    const v3 bu = add3(mul3(u, r[2]), mul3(v, r[3])), bv = add3(mul3(u, r[4]), mul3(v, r[5]));
    out[0] = origin.x; out[1] = origin.y; out[2] = origin.z;
    out[3] = bu.x; out[4] = bu.y; out[5] = bu.z;
    out[6] = bv.x; out[7] = bv.y; out[8] = bv.z;
    free(proj);
    free(hull);
}

/* ---- C2 ------------------------------------------------------------------------------------------------------------ */

typedef struct { double pos; float c[4]; } prim;

static prim prim_at(const double* pos, const float* rgba, int i) {
    prim p;
    p.pos = pos[i];
    memcpy(p.c, rgba + 4 * i, sizeof p.c);
    return p;
}
/* ref :503-507 */
static void color_diff(const float p1[4], const float p2[4], int associated, float out[4]) {
    const float s1 = associated ? p1[3] : 1.f, s2 = associated ? p2[3] : 1.f;
    int k;
    for (k = 0; k < 4; ++k) out[k] = fabsf(p2[k] * s2 - p1[k] * s1);
}
/* ref :509-524 [GLM] mix with a double parameter */
static prim mix_at(prim a, prim b, prim at) {
    const double t = (at.pos - a.pos) / (b.pos - a.pos);
    prim r;
    int k;
    r.pos = a.pos * (1.0 - t) + b.pos * t;
    for (k = 0; k < 4; ++k) r.c[k] = (float)((double)a.c[k] * (1.0 - t) + (double)b.c[k] * t);
    return r;
}
static int ne0(const float c[4], float eps) { /* glm::any(glm::epsilonNotEqual(c, vec4(0), eps)) [GLM] */
    return fabsf(c[0]) >= eps || fabsf(c[1]) >= eps || fabsf(c[2]) >= eps || fabsf(c[3]) >= eps;
}

/* ref :364-501.  tf / prev: sorted points (position, rgba).  Room for n_tf + n_prev + 2 points in the outputs; returns the
 * number of points, -1 when exactly one of the two functions is empty (the reference reads point 0 of an empty function there). */
int cpmo_tf_difference_points(const double* tf_pos, const float* tf_rgba, int n_tf, const double* prev_pos, const float* prev_rgba, int n_prev,
                              float eps, int associated, float* out_pos, float* out_rgba) {
    int out = 0, id = 0, prev_id = 0;
    const float zero[4] = { 0.f, 0.f, 0.f, 0.f };
#define PUT(P, C) do { out_pos[out] = (float)(P); memcpy(out_rgba + 4 * out, (C), 4 * sizeof(float)); ++out; } while (0)
    if (n_tf == 0 && n_prev == 0) { /* :365-376 (both positions 0, as written there) */
        PUT(0.0, zero);
        PUT(0.0, zero);
        return out;
    }
    if (n_tf == 0 || n_prev == 0) return -1;
    const prim first = prim_at(tf_pos, tf_rgba, 0), pfirst = prim_at(prev_pos, prev_rgba, 0);
    prim p1, p2;
    p1.pos = first.pos < pfirst.pos ? first.pos : pfirst.pos; /* :397-399 */
    color_diff(first.c, pfirst.c, associated, p1.c);
    p2 = p1;
    if (first.pos != pfirst.pos && first.c[3] == 0.f && pfirst.c[3] == 0.f) { /* :400-416 */
        if (first.pos < pfirst.pos) {
            const prim a2 = prim_at(tf_pos, tf_rgba, 1 < n_tf - 1 ? 1 : n_tf - 1);
            const prim p = mix_at(first, a2, pfirst);
            p2.pos = pfirst.pos;
            color_diff(pfirst.c, p.c, associated, p2.c);
        } else {
            const prim a2 = prim_at(prev_pos, prev_rgba, 1 < n_prev - 1 ? 1 : n_prev - 1);
            const prim p = mix_at(pfirst, a2, first);
            p2.pos = first.pos;
            color_diff(first.c, p.c, associated, p2.c);
        }
    }
    if (p1.pos > 0. && (first.c[3] > 0.f || pfirst.c[3] > 0.f) && ne0(p1.c, eps)) PUT(0.0, p1.c); /* :417-433 */
    else PUT(0.0, zero);
    while (id < n_tf || prev_id < n_prev) { /* :435-486 */
        if ((ne0(p1.c, eps) || ne0(p2.c, eps)) && (p1.c[3] > 0.f || p2.c[3] > 0.f)) {
            if (out == 1) PUT(p1.pos, p1.c);
            PUT(p2.pos, p2.c);
        }
        const prim a1 = prim_at(tf_pos, tf_rgba, id < n_tf - 1 ? id : n_tf - 1);
        prim a2, b2;
        if (id + 1 < n_tf - 1) a2 = prim_at(tf_pos, tf_rgba, id + 1);
        else { a2 = prim_at(tf_pos, tf_rgba, n_tf - 1); a2.pos = 1.; }
        const prim b1 = prim_at(prev_pos, prev_rgba, prev_id < n_prev - 1 ? prev_id : n_prev - 1);
        if (prev_id + 1 < n_prev - 1) b2 = prim_at(prev_pos, prev_rgba, prev_id + 1);
        else { b2 = prim_at(prev_pos, prev_rgba, n_prev - 1); b2.pos = 1.; }
        p1 = p2;
        if (a2.pos < b2.pos) {
            const prim p = mix_at(b1, b2, a2);
            p2.pos = a2.pos;
            color_diff(a2.c, p.c, associated, p2.c);
            ++id;
        } else if (b2.pos < a2.pos) {
            const prim p = mix_at(a1, a2, b2);
            p2.pos = b2.pos;
            color_diff(b2.c, p.c, associated, p2.c);
            ++prev_id;
        } else {
            p2.pos = a2.c[3] < b2.c[3] ? b2.pos : a2.pos;
            color_diff(a2.c, b2.c, associated, p2.c);
            ++id;
            ++prev_id;
        }
    }
    if (p2.pos < 1. && p2.c[3] > 0.f) PUT(p2.pos, p2.c); /* :487-491 */
    if (out_pos[out - 1] < 1.f) PUT(1.0, zero);           /* :492-497 */
#undef PUT
    return out;
}
