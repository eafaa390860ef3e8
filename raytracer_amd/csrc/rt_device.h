// Device-side building blocks of the renderer (gfx950): vector arithmetic in the reference's evaluation order, the keyed
// counter RNG, portable sinf/atan2f, the primitive tests, the stackless two-level BVH traversal, hit finalisation and
// shading.  Compiled with -ffp-contract=off: the reference's x86-64 build rounds every product and every sum separately,
// and primary-hit ids and radiance are compared bit for bit.  File:line citations are relative to the reference's
// Raytracer/src.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "rt_flat.h"

namespace rtd {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 divs(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }   // glm vec3 / scalar: true division
// glm/detail/func_geometric.inl:48-55,68-79,82-90: dot = (x*x' + y*y') + z*z'; normalize = v * (1 / sqrt(dot))
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 x, V3 y) { return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
__device__ __forceinline__ V3 normalize(V3 v) { float s = 1.0f / sqrtf(dot(v, v)); return v * s; }
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }   // glm/detail/func_common.inl:20
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }   // :29
__device__ __forceinline__ float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
// selects, not indexed stores: a run-time index into a register-resident vector would push it out to scratch memory
__device__ __forceinline__ void setc(V3& v, int i, float f) { v.x = i == 0 ? f : v.x; v.y = i == 1 ? f : v.y; v.z = i >= 2 ? f : v.z; }
__device__ __forceinline__ V3 xyz(float4 f) { return mk(f.x, f.y, f.z); }

// ---------------------------------------------------------------------------------------------------- RNG
// Stand-in for std::rand (DataUtility.h:11-12): value i of stream `key` is fmix32(key ^ i*0x9E3779B9) >> 1 (31 bits, glibc's
// RAND_MAX).  One stream per (frame seed, pixel, sample): the draws of a path never depend on another path.
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rand31(uint32_t key, uint32_t ctr) { return fmix32(key ^ (ctr * 0x9E3779B9u)) >> 1; }
__device__ __forceinline__ uint32_t sample_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = fmix32(seed ^ 0xA511E9B3u);
    k = fmix32(k + pixel * 0x9E3779B1u + 0x68E31DA4u);
    k = fmix32(k ^ (sample * 0x85EBCA77u + 0xC2B2AE3Du));
    return k;
}
struct Rng { uint32_t key, ctr; };
// RandomUnitDistance (DataUtility.h:12): (float)(rand() - RAND_MAX/2) / (RAND_MAX/2); the divisor converts to 2^30, so the
// division is an exact scaling
__device__ __forceinline__ float unit_distance(Rng& g) {
    int r = (int)rand31(g.key, g.ctr++);
    return (float)(r - 1073741823) * 9.31322574615478515625e-10f;
}
__device__ __forceinline__ float random01(Rng& g) { return (float)(int)rand31(g.key, g.ctr++) / 2147483648.0f; }   // DataUtility.h:11
__device__ __forceinline__ V3 unit_sphere_point(Rng& g) {   // DataUtility.h:28-34
    for (;;) {
        V3 s;
        s.x = unit_distance(g); s.y = unit_distance(g); s.z = unit_distance(g);
        if (s.x * s.x + s.y * s.y + s.z * s.z < 1.0f) return s;
    }
}
__device__ __forceinline__ void unit_circle_point(Rng& g, float& x, float& y) {   // DataUtility.h:35-41
    for (;;) {
        x = unit_distance(g); y = unit_distance(g);
        if (x * x + y * y < 1.0f) return;
    }
}

// ---------------------------------------------------------------------------------------------------- portable sinf / atan2f
// Evaluated in IEEE double with a fixed operation order (fdlibm's kernel polynomials) and rounded to float once: within
// half an ulp of the true value almost everywhere, and the same bits on any IEEE machine — the CPU checker evaluates the
// same formulas.  The reference calls libm's sinf/atan2f (DataUtility.cpp:91, World.cpp:174, Object.cpp:25).
__device__ inline float pm_sinf(float x) {
    if (!(fabsf(x) <= FLT_MAX)) return x - x;
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079631090164184570e+00, P2 = 1.58932547122958567343e-08, P3 = 6.12323399573676603587e-17;
    const double xd = (double)x;
    const double k = rint(xd * TWO_OVER_PI);
    const double r = ((xd - k * P1) - k * P2) - k * P3;
    const int q = (int)(k - 4.0 * floor(k * 0.25));
    const double z = r * r;
    double res;
    if (q & 1) {
        const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        const double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        res = 1.0 - (0.5 * z - z * p);
    } else {
        const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
        const double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
        res = r + r * (z * p);
    }
    if (q & 2) res = -res;
    return (float)res;
}
__device__ inline double pm_atan_pos(double x) {
    // the four reductions of fdlibm's s_atan.c; numerator and denominator are picked by selects and divided once: the lanes
    // of a wave fall into all of the ranges, and a division per range would be executed four times over
    int id = -1;
    double num = x, den = 1.0, hi = 0.0, lo = 0.0;
    if (!(x < 0.4375)) {
        const bool r0 = x < 0.6875, r1 = x < 1.1875, r2 = x < 2.4375;
        id = r0 ? 0 : (r1 ? 1 : (r2 ? 2 : 3));
        num = r0 ? 2.0 * x - 1.0 : (r1 ? x - 1.0 : (r2 ? x - 1.5 : -1.0));
        den = r0 ? 2.0 + x : (r1 ? x + 1.0 : (r2 ? 1.0 + 1.5 * x : x));
        hi = r0 ? 4.63647609000806093515e-01 : (r1 ? 7.85398163397448278999e-01 : (r2 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00));
        lo = r0 ? 2.26987774529616870924e-17 : (r1 ? 3.06161699786838301793e-17 : (r2 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17));
        x = num / den;
    }
    const double z = x * x, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                      w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                      w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    if (id < 0) return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}
__device__ inline float pm_atan2f(float y, float x) {
    if (!(fabsf(x) <= FLT_MAX) || !(fabsf(y) <= FLT_MAX)) return (x - x) + (y - y);
    const double PI = 3.14159265358979311600e+00;
    const double ax = fabs((double)x), ay = fabs((double)y);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else a = pm_atan_pos(ay / ax);
    if (__float_as_uint(x) >> 31) a = PI - a;
    if (__float_as_uint(y) >> 31) a = -a;
    return (float)a;
}

// logf for the fog's free path (Object.cpp:579): fdlibm's e_log.c reduction and polynomial in IEEE double, rounded to float
// once; x >= 0 (it is a Random01 draw).  The CPU checker evaluates the same formulas.
__device__ inline float pm_logf(float x) {
    if (x != x) return x;
    if (x <= 0.0f) return x == 0.0f ? -INFINITY : (x - x) / 0.0f;
    if (x > FLT_MAX) return x;
    unsigned long long u = (unsigned long long)__double_as_longlong((double)x);
    int k = (int)((u >> 52) & 0x7FF) - 1023;
    u = (u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = __longlong_as_double((long long)u);
    if (m > 1.41421356237309514547) { m *= 0.5; k += 1; }
    const double f = m - 1.0, s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    return (float)(dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f));
}

// ---------------------------------------------------------------------------------------------------- scene view
struct Scene {
    const float4* ent;   // 2 x float4 per entry: lo, hi
    const float4* tris; const float4* inst; const float4* objects; const float4* planes;
    const rt::ImageRec* images; const uint8_t* texels;
    const uint32_t* alpha_bits;   // one bit per texel of every image: alpha >= 1
    uint32_t sky_off, sky_w, sky_h;
    int top_end, stale_entry, n_planes;
    int n_entries, n_tris, n_inst, fast_div_ok, idle_entry;
    int has_fog;   // some entry is a fog volume: traversal consumes random numbers (Object.cpp:579)
    int ranked;    // leaf entries carry their rank in the reference's visiting order (every scene without fog)
    float entry_guard;   // rays that start further out (some |origin coordinate| above this) take every mesh instance's entry
                         // box for hit: the boxes' margin is proven up to there (rt_scene.cpp, "entry boxes"); FLT_MAX without such boxes
    int has_ref;   // the entries are a regrouped layout: rays that are not ray_monotone are not walked here but handed
                   // over to the walk through the reference's layout (rt_engine.hip, exceptions)
    // distance culling (see "distance culling" below): scene-wide constants of the bound; cull_on = 0 switches it off
    float cull_delta, cull_cabs, cull_2rmax;
    int cull_on;
};
// What the ordered traversal (see "ordered traversal" below) needs beside the scene; ent_ord == nullptr: the scene has none
struct OrdScene {
    const float4* ent_ord;        // the entries with second-child links in place of skip links
    const uint32_t* top_tab;      // records of the tree's upper levels, copied into LDS by every block
    const int32_t* inst_end;      // per instance: the END entry that closes its range
    int n_top_tab, stack_slots;   // records in top_tab; stack slots a ray can need
    uint32_t root_link;           // where a ray starts: entry 0, or its record in the table
    // pair steps (see "pair steps"): the records by entry index, where a ray that enters an instance starts, and whether
    // the scene is walked that way (then ent_ord and top_tab are the entries and the table that go with the records)
    const float4* pair;
    const uint32_t* inst_start;
    int pairs;
};
struct Ray { V3 o, d; };

// BoundingBox::DoesRayHit (DataUtility.cpp:35-82): true division per axis, branch on dir >= 0 (so -0.0 takes the >= branch),
// NaN comparisons fall through as "not rejected".
__device__ __forceinline__ bool box_hit(V3 mn, V3 mx, const Ray& r) {
    const bool px = r.d.x >= 0, py = r.d.y >= 0, pz = r.d.z >= 0;
    float tmin = ((px ? mn.x : mx.x) - r.o.x) / r.d.x;
    float tmax = ((px ? mx.x : mn.x) - r.o.x) / r.d.x;
    const float tymin = ((py ? mn.y : mx.y) - r.o.y) / r.d.y;
    const float tymax = ((py ? mx.y : mn.y) - r.o.y) / r.d.y;
    if ((tmin > tymax) || (tymin > tmax)) return false;
    tmin = gmax(tmin, tymin);
    tmax = gmin(tmax, tymax);
    const float tzmin = ((pz ? mn.z : mx.z) - r.o.z) / r.d.z;
    const float tzmax = ((pz ? mx.z : mn.z) - r.o.z) / r.d.z;
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    tmin = gmax(tmin, tzmin);
    tmax = gmin(tmax, tzmax);
    if (tmin < 0.0f && tmax < 0.0f) return false;
    return true;
}

// Sphere::Intersect up to the accept/reject decision (Object.cpp:10-19).  The reference's sqrt() call binds to the double
// overload, so distance = (float)((double)p1 - sqrt((double)p2sqr)).  `wrote` tells that HitInfo.distance was assigned
// even though the test failed (near root behind the origin).
__device__ __forceinline__ bool sphere_test(V3 c, float radius, const Ray& r, float& t, bool& wrote) {
    const V3 L = r.o - c;
    const float p1 = -dot(r.d, L);
    const float p2sqr = p1 * p1 - dot(L, L) + radius * radius;
    wrote = false;
    if (p2sqr < 0) return false;
    t = (float)((double)p1 - sqrt((double)p2sqr));
    wrote = true;
    return !(t < 0);
}
// AxisAlignedCube::Intersect (Object.cpp:36-119): entry distance and face normal (exit face when the origin is inside)
__device__ __forceinline__ bool cube_test(V3 mn, V3 mx, const Ray& r, float& t, V3& normal) {
    const bool px = r.d.x >= 0, py = r.d.y >= 0, pz = r.d.z >= 0;
    float tmin = ((px ? mn.x : mx.x) - r.o.x) / r.d.x;
    float tmax = ((px ? mx.x : mn.x) - r.o.x) / r.d.x;
    V3 nmin = mk(px ? -1.0f : 1.0f, 0, 0), nmax = nmin;
    const float tymin = ((py ? mn.y : mx.y) - r.o.y) / r.d.y;
    const float tymax = ((py ? mx.y : mn.y) - r.o.y) / r.d.y;
    V3 cand = mk(0, py ? -1.0f : 1.0f, 0);
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) { tmin = tymin; nmin = cand; }
    if (tymax < tmax) { tmax = tymax; nmax = cand; }
    const float tzmin = ((pz ? mn.z : mx.z) - r.o.z) / r.d.z;
    const float tzmax = ((pz ? mx.z : mn.z) - r.o.z) / r.d.z;
    cand = mk(0, 0, pz ? -1.0f : 1.0f);
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    if (tzmin > tmin) { tmin = tzmin; nmin = cand; }
    if (tzmax < tmax) { tmax = tzmax; nmax = cand; }
    if (tmin < 0.0f && tmax < 0.0f) return false;
    if (tmin < 0) { nmin = nmax; tmin = tmax; }
    t = tmin;
    normal = nmin;
    return true;
}
// Fog::Intersect (Object.cpp:558-593): where the ray would scatter in an unbounded medium of this density, and whether
// that lies inside the boundary cube.  Draws one Random01 when the ray meets the cube.
__device__ __forceinline__ bool fog_test(V3 mn, V3 mx, float neg_inv_density, const Ray& r, Rng& rng, float& t) {
    float t1, t2;
    V3 n1, n2;
    if (!cube_test(mn, mx, r, t1, n1)) return false;
    Ray past;
    past.o = (r.o + r.d * t1) - n1 * 0.001f;
    past.d = r.d;
    const bool inside = !cube_test(mn, mx, past, t2, n2);
    const V3 enter = inside ? r.o : past.o;
    const V3 exit_ = inside ? past.o : past.o + past.d * t2;
    const float supposed = neg_inv_density * pm_logf(random01(rng));
    const V3 dd = exit_ - enter;
    const float travel = sqrtf(dot(dd, dd));          // glm::distance
    if (supposed > travel) return false;
    t = inside ? supposed : supposed + t1;
    return true;
}
// float -> unsigned as the reference's x86-64 build converts it (glm::uvec2 cast, Object.cpp:545): truncate to 64 bits
// signed, keep the low 32
__device__ __forceinline__ uint32_t f2u_x86(float f) {
    if (f >= 0.0f && f < 2147483648.0f) return (uint32_t)(int)f;
    if (!(f > -9.2233720368547758e18f && f < 9.2233720368547758e18f)) return 0u;
    return (uint32_t)(unsigned long long)(long long)f;
}
// Texture coordinates the reference can produce lie in [0, 1] and index at most one row past the image (the tables are
// padded by that row); the index is held there so that coordinates no scene produces (a caller's ray with components
// that overflow, say) read the padding instead of memory outside the table.  Nothing changes for indices in range.
__device__ __forceinline__ uint32_t texel_in_table(uint32_t texel, uint32_t w, uint32_t h) { return min(texel, w * h + w); }
__device__ __forceinline__ int image_index(const rt::ImageRec im, float u, float v) {   // DataUtility.h:82 (uvec2 size: unsigned arithmetic)
    const uint32_t row = (uint32_t)(int)(v * (float)im.h), col = (uint32_t)(int)(u * (float)im.w);
    return 4 * (int)texel_in_table(row * im.w + col, im.w, im.h);
}
// Triangle::Intersect (Object.cpp:498-550).  alpha_off >= 0: the material's image texture decides solidity
// (DataUtility.cpp:123-127: alpha byte > 0.01f, i.e. alpha >= 1) — looked up in a one-bit-per-texel copy of the alpha
// channel at word alpha_off of Scene::alpha_bits (alpha_wh = height << 16 | width), 32 times smaller than the texels and
// therefore resident in the vector cache.  `wrote`: HitInfo.distance was assigned although the triangle was rejected
// by the cut-out.
//
// The test is cut in two.  tri_reach is the geometric part (Object.cpp:498-531): does the ray meet the triangle's plane
// inside the triangle, and at which distance.  tri_surface is everything after it (Object.cpp:532-550): texture
// coordinates, the cut-out, the assignment of HitInfo.distance.  Most rays that reach a triangle do so behind a hit they
// already have, where the second part cannot change anything; the traversal runs it only where it can, and for many
// lanes at once.
__device__ __forceinline__ bool tri_reach(const float4* tris, int ti, const Ray& r, float& tt) {
    const float EPSILON = 0.0000001f;
    const float4* T = tris + (size_t)ti * rt::RT_TRI_F4;
    const V3 v0 = xyz(T[0]), edge1 = xyz(T[1]), edge2 = xyz(T[2]);
    const V3 h = cross(r.d, edge2);
    const float a = dot(edge1, h);
    if (a > -EPSILON && a < EPSILON) return false;
    const float f = 1.0f / a;
    const V3 s = r.o - v0;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const V3 q = cross(s, edge1);
    const float v = f * dot(r.d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    tt = f * dot(edge2, q);
    return tt > EPSILON;
}
__device__ __forceinline__ bool tri_surface(const Scene& sc, const float4* tris, int ti, int alpha_off, uint32_t alpha_wh, const Ray& r, float tt, float& t, float& uo, float& vo, bool& wrote) {
    const float4* T = tris + (size_t)ti * rt::RT_TRI_F4;
    const float4 ta = T[0], t5 = T[5];
    const V3 p1 = xyz(ta), p2 = mk(ta.w, T[1].w, T[2].w), p3 = mk(t5.y, t5.z, t5.w);
    wrote = false;
    const V3 hp = r.o + r.d * tt;
    const float denominator = p1.y * (p2.z - p3.z) - p2.y * (p1.z - p3.z) + p3.y * (p1.z - p2.z);
    if (denominator == 0) return false;
    t = tt;
    wrote = true;
    const float4 t3 = T[3], t4 = T[4];
    const float percent0 = (hp.y * (p2.z - p3.z) - hp.z * (p2.y - p3.y) + p2.y * p3.z - p3.y * p2.z) / denominator;
    const float percent1 = -(hp.y * (p1.z - p3.z) - hp.z * (p1.y - p3.y) + p1.y * p3.z - p3.y * p1.z) / denominator;
    const float percent2 = (hp.y * (p1.z - p2.z) - hp.z * (p1.y - p2.y) + p1.y * p2.z - p2.y * p1.z) / denominator;
    float uu = t3.x * percent0 + t3.z * percent1 + t4.x * percent2;
    float vv = t3.y * percent0 + t3.w * percent1 + t4.y * percent2;
    uu -= (float)f2u_x86(uu);
    vv -= (float)f2u_x86(vv);
    if (uu < 0) uu += 1;
    if (vv < 0) vv += 1;
    uo = uu; vo = vv;
    if (alpha_off >= 0) {
        const uint32_t w = alpha_wh & 0xFFFFu, h = alpha_wh >> 16;
        const uint32_t texel = texel_in_table((uint32_t)(int)(vv * (float)h) * w + (uint32_t)(int)(uu * (float)w), w, h);   // DataUtility.h:82
        if (!((sc.alpha_bits[(uint32_t)alpha_off + (texel >> 5)] >> (texel & 31u)) & 1u)) return false;
    }
    return true;
}

// The rotation of RotationIntersect (Object.cpp:306-316 in, :322-333 out) about one axis: components (a1, a2) = (1,2) for
// X, (0,2) for Y, (0,1) for Z are mixed, the third is kept.  Written per axis with plain components: a run-time index
// into a register-resident vector would push it out to scratch memory.
__device__ __forceinline__ V3 rotate_in(int axis, float sinT, float cosT, V3 v) {
    if (axis == 0) return mk(v.x, cosT * v.y - sinT * v.z, sinT * v.y + cosT * v.z);
    if (axis == 1) return mk(cosT * v.x - sinT * v.z, v.y, sinT * v.x + cosT * v.z);
    return mk(cosT * v.x - sinT * v.y, sinT * v.x + cosT * v.y, v.z);
}
__device__ __forceinline__ V3 rotate_out(int axis, float sinT, float cosT, V3 v) {
    if (axis == 0) return mk(v.x, cosT * v.y + sinT * v.z, -sinT * v.y + cosT * v.z);
    if (axis == 1) return mk(cosT * v.x + sinT * v.z, v.y, -sinT * v.x + cosT * v.z);
    return mk(cosT * v.x + sinT * v.y, -sinT * v.x + cosT * v.y, v.z);
}
// ApplyMovement::Intersect + RotationIntersect on the way in (Object.cpp:601-603, 306-316)
__device__ __forceinline__ Ray to_instance(const float4 i0, const float4 i1, const float4 i2, const Ray& r) {
    Ray q = r;
    if (i0.w != 0.0f) q.o = r.o - xyz(i0);
    const int axis = __float_as_int(i1.w);
    if (axis >= 0) {
        const V3 pivot = xyz(i1);
        q.o = rotate_in(axis, i2.x, i2.y, q.o - pivot) + pivot;
        q.d = rotate_in(axis, i2.x, i2.y, q.d);
    }
    return q;
}
// and on the way out (Object.cpp:322-333, 606)
__device__ __forceinline__ void from_instance(const float4 i0, const float4 i1, const float4 i2, V3& point, V3& normal) {
    const int axis = __float_as_int(i1.w);
    if (axis >= 0) {
        const V3 pivot = xyz(i1);
        point = rotate_out(axis, i2.x, i2.y, point - pivot) + pivot;
        normal = rotate_out(axis, i2.x, i2.y, normal);
    }
    if (i0.w != 0.0f) point = point + xyz(i0);
}

// ---------------------------------------------------------------------------------------------------- exact division, cheaply
// A slab test divides six differences by the three direction components of the SAME ray.  With y = RN(1/b) computed once
// per ray by a true division, RN(a/b) is recovered from y by two residual corrections (Markstein's scheme: q ~ a*y,
// r = a - b*q exactly by FMA, q += r*y), five ordinary instructions instead of the ~11 of the hardware's IEEE sequence.
// The scheme is exact (a zero quotient possibly with the other sign, which no comparison can tell) while nothing
// over/underflows; `usable` below admits a ray only if that holds for every box of the scene: |dir| components in
// [2^-20, 2^20], origin components 0 or in [2^-50, 2^40] (dividends down to 2^-80 are still exact) — the host checks the same of all box coordinates
// (Scene::fast_div_ok).  Other rays take true division.
__device__ __forceinline__ float div_by(float a, float b, float y) {
    float q = a * y;
    float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}
// RN(1/d) for |d| in [2^-20, 2^20]: the hardware reciprocal refined by the FMA steps of the IEEE division sequence, without
// that sequence's operand scaling and special-case fix-up, which do nothing in this range (seven instructions, not eleven)
__device__ __forceinline__ float recip_in_range(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    const float q = __builtin_fmaf(__builtin_fmaf(-d, r1, 1.0f), r1, r1);
    return __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), r1, q);
}
__device__ __forceinline__ bool mag_in(float v, float lo, float hi) { const float a = fabsf(v); return a >= lo && a <= hi; }
__device__ __forceinline__ bool coord_ok(float v) { return v == 0.0f || mag_in(v, 8.8817841970012523e-16f, 1.099511627776e12f); }
// The slab test as the reference writes it (true divisions) is monotone in the box — a box that contains another is hit
// whenever the other is — for every ray whose quotients are numbers: a direction without a zero component and nothing
// that is not finite.  (With d = -0 the reference's test takes the entry side for the exit side; with 0 / 0 or inf - inf
// its comparisons see NaNs.)  A regrouped layout walks only such rays; see walker_special, WALK_HANDED_OVER.
__device__ __forceinline__ bool ray_monotone(const Ray& r) {
    const bool finite = fabsf(r.d.x) <= FLT_MAX && fabsf(r.d.y) <= FLT_MAX && fabsf(r.d.z) <= FLT_MAX &&
                        fabsf(r.o.x) <= FLT_MAX && fabsf(r.o.y) <= FLT_MAX && fabsf(r.o.z) <= FLT_MAX;
    return finite && r.d.x != 0.0f && r.d.y != 0.0f && r.d.z != 0.0f;
}
__device__ __forceinline__ bool ray_usable(const Ray& r) {
    return mag_in(r.d.x, 9.5367431640625e-07f, 1048576.0f) && mag_in(r.d.y, 9.5367431640625e-07f, 1048576.0f) &&
           mag_in(r.d.z, 9.5367431640625e-07f, 1048576.0f) && coord_ok(r.o.x) && coord_ok(r.o.y) && coord_ok(r.o.z);
}

// ---------------------------------------------------------------------------------------------------- closest hit
// What World::GetRayColor learns from rootNode->Intersect plus the plane list (World.cpp:105-119).
//   best >= 0: entry index of the winning leaf; best <= -2: plane -(best+2); best == -1: nothing hit.
//   dist: HitInfo.distance as the reference leaves it — FLT_MAX, the winner's distance, or the stale value the leftmost
//   leaf assigned before rejecting (Object.cpp:17-19, 534-550), which the plane is then compared against.
struct Closest { float dist; int best; float u, v; bool hit; };

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sk_kind(uint32_t sk) { return (int)(sk >> rt::SK_SHIFT); }
__device__ __forceinline__ int sk_payload(uint32_t sk) { return (int)(sk & rt::SK_MASK); }

// Per-lane traversal state.  The lane holds the entry it stands on so that the wave can decide which phase to run: lanes
// on inner entries step together (box tests), lanes on leaves test primitives together.  A lane without a ray stands on
// the scene's IDLE entry.
struct Walker {
    // the ray in the current frame (the scene's, or a mesh instance's), kept as the operand pairs of the box step:
    // (o.x, o.y) (o.z, 1/d.x) (1/d.y, 1/d.z) (d.x, d.y) (d.z, -)
    f2 p0, p1, p2, p3, p4;
    bool usable;             // the reciprocal path reproduces true division for this ray
    // while inside an instance: the same for the ray as given, to come back to
    f2 s0, s1, s2, s3, s4;
    bool s_usable;
    int i, resume, inst, stale, alpha_off;
    uint32_t alpha_wh;
    f2 e01, e23, e45;        // dwords 0..5 of the entry
    uint32_t sk;             // dword 6: kind << 28 | payload; for an inner entry (kind 0) simply the skip index
    float rho;               // dword 7 of an inner entry: its culling parameter (+inf: never culled)
    // distance culling: per-ray constants (cull_w = +inf: this ray never culls) and the current threshold
    // base + k * rho(entry); base = +inf until the ray has a hit
    float cull_w, cull_kray, cull_c, cull_base, cull_k;
    uint32_t f0, f1, f2, cnt;   // leaves visited but not yet tested, oldest first, and their number (0..3)
    float reach;                // > 0: the ray reaches the triangle f0 at this distance and tri_surface has yet to run
    Rng rng;                    // the path's random stream: fog volumes draw from it during traversal
    Closest c;
    // ordered traversal only: LDS byte address of the lane's next free stack slot; first-child link of the table record
    // the lane stands on
    uint32_t sp, eleft;
    // pair steps only: while the lane stands on a pair record, e01 / e23 / e45 hold the first child's box and these the
    // second child's, the two links and the two culling parameters; while it stands on an entry, g01 holds dwords 6, 7
    ::rtd::f2 g01, g23, g45, lk, rh;
};
// entries (and, when staged, triangles) either in LDS or in global memory
struct Store {
    const float4* ent; const float4* tris; const float4* inst; const float4* planes;   // 2 x float4 per entry
    // ordered traversal only: the block's dynamic LDS as dwords and its LDS byte address; byte addresses of the table of
    // upper-level records and of slot 0 of thread 0's stack
    uint32_t* lds0; uint32_t lds0_addr, top_addr, stack_addr;
    uint32_t root_link; const int32_t* inst_end;   // where a ray starts; per instance the END entry that closes its range
    const float4* pair; const uint32_t* inst_start;   // pair steps: the records by entry index; per instance where a ray starts in it
};

__device__ __forceinline__ Ray walker_ray(const Walker& w) {
    Ray r;
    r.o = mk(w.p0.x, w.p0.y, w.p1.x); r.d = mk(w.p3.x, w.p3.y, w.p4.x);
    return r;
}
// usable_known: 0 or 1 when the caller has kept the answer of ray_usable for this very ray, -1 to work it out
__device__ __forceinline__ void walker_set_ray(Walker& w, const Ray r, bool scene_ok, int usable_known = -1) {
    w.usable = usable_known >= 0 ? usable_known != 0 : scene_ok && ray_usable(r);
    // the reciprocals are only used when the ray is admitted to the reciprocal path, i.e. |d| within [2^-20, 2^20]
    w.p0 = f2{r.o.x, r.o.y}; w.p1 = f2{r.o.z, recip_in_range(r.d.x)}; w.p2 = f2{recip_in_range(r.d.y), recip_in_range(r.d.z)};
    w.p3 = f2{r.d.x, r.d.y}; w.p4 = f2{r.d.z, 0.0f};
}
__device__ __forceinline__ void walker_load(const Store& st, Walker& w) {
    const f2* e = reinterpret_cast<const f2*>(st.ent) + 4 * w.i;
    w.e01 = e[0]; w.e23 = e[1]; w.e45 = e[2];
    w.sk = reinterpret_cast<const uint32_t*>(e)[6];
    w.rho = reinterpret_cast<const float*>(e)[7];
}
__device__ __forceinline__ void walker_idle(const Scene& sc, Walker& w) {
    w.i = sc.idle_entry; w.sk = (uint32_t)rt::ENT_IDLE << rt::SK_SHIFT;
    w.e01 = f2{0, 0}; w.e23 = f2{0, 0}; w.e45 = f2{0, 0}; w.rho = INFINITY;
    w.cull_w = INFINITY; w.cull_kray = INFINITY; w.cull_c = INFINITY; w.cull_base = INFINITY; w.cull_k = 0.0f;
    w.p0 = f2{0, 0}; w.p1 = f2{0, 1}; w.p2 = f2{1, 1}; w.p3 = f2{1, 1}; w.p4 = f2{1, 0}; w.usable = true;
    w.s0 = w.p0; w.s1 = w.p1; w.s2 = w.p2; w.s3 = w.p3; w.s4 = w.p4; w.s_usable = true;
    w.resume = 0; w.inst = -1; w.stale = -1; w.alpha_off = -1; w.alpha_wh = 0;
    w.f0 = 0; w.f1 = 0; w.f2 = 0; w.cnt = 0; w.reach = 0.0f;
    w.g01 = f2{__uint_as_float(w.sk), 0.0f}; w.g23 = f2{0, 0}; w.g45 = f2{0, 0}; w.lk = f2{0, 0}; w.rh = f2{0, 0};   // (pair steps: the idle entry as its loop sees it)
    w.sp = 0; w.eleft = 0;
    w.rng.key = 0; w.rng.ctr = 0;
    w.c.dist = FLT_MAX; w.c.best = -1; w.c.u = 0; w.c.v = 0; w.c.hit = false;
}
// ---------------------------------------------------------------------------------------------------- distance culling
// BVH_Node::Intersect (Object.cpp:241-252) descends into every box the ray hits, also when the box lies wholly behind the
// closest hit found so far.  Nothing in such a box can win: a leaf replaces the closest hit only with a strictly smaller
// distance (Object.cpp:248).  The traversal of large scenes (entries in memory) therefore treats a box as missed when
// its entry distance is PROVABLY beyond the current closest distance D — provably in the reference's own arithmetic,
// because what must not be lost is a leaf whose COMPUTED distance is below D, whatever rounding did to it.
//
// Which boxes: inner entries of the top level whose subtree holds only spheres with a positive radius and cubes with
// min <= max (the host marks every other inner entry with rho = +inf: below it may be a mesh instance — a triangle
// nearly edge-on to the ray can get any distance at all —, a fog volume, which draws random numbers, or an inverted
// box).  Which rays: those admitted to the reciprocal path whose direction has |d.d - 1| <= 2^-10.  A ray without a
// hit culls nothing, so the leaf whose rejected HitInfo.distance survives (the first one tested) is never skipped.
//
// The bound.  Every inner box is the fmin/fmax union of its children's boxes (Object.cpp:156-166) and a leaf's box is
// computed from the leaf in float, so with round-to-nearest being monotone, the entry distance the reference's slab
// test computes for a box is <= the one it computes for any box below it, down to the float box B of a leaf.
//   * Cube leaf: its Intersect IS that slab computation on B, so its distance is >= the entry of every box above.
//   * Sphere leaf (c, r), accepted with distance t_c < D: with P, L as the reference computes them (Object.cpp:11-16)
//     but exact, and E the rounding error of its discriminant, the point o + tau d at tau = P - sqrt(disc_c) lies
//     exactly on the sphere of radius sqrt(r^2 + E + eta tau^2) about c, where |d|^2 = 1 + eta: for a far, small sphere
//     the unit-length assumption of the formula is worth more than all rounding.  That point is inside B widened per
//     axis by  g = (|E| + |eta| tau^2) / (2 r) + rounding of B and of L,  so  tau >= entry(B) - g / min|d_i|,  and
//     t_c >= tau - 2^-21 (D + 2r).  With |E| <= 11 * 2^-24 (D + 2r)^2:
//         t_c >= entry(box above) (1 - 2^-21) - W [delta + 2^-23 (D + 2 r_max) + rho eta' (D + 2 r_max)^2] - 2^-21 (D + 2 r_max)
//     W = max |1/d_i| (1 + 2^-20), eta' = |fl(d.d) - 1| + 2^-20, rho = max 1 / (2r) below the box,
//     delta = 2^-23 max(|c_i| + r).
// The one-multiplication entry distance the box step decides on is within 1.5 * 2^-23 of the reference's, hence a box
// is skipped when that distance exceeds
//         D (1 + 2^-16 + 2^-23 W) + 2^-16 r_max + W (delta + 2^-22 r_max)  +  W eta' (D + 2 r_max)^2 * rho
//         \_______________________________ cull_base _________________/     \________ cull_k ________/
// (margins 2^-16 where 2^-19 would do).  cull_base and cull_k change only when the closest hit does.
// The planes (tested after the tree, World.cpp:110-118) give no further bound worth having: a box entered beyond a
// plane lies on its far side, where scenes keep nothing, and a skipped hit there would let the plane be compared with
// a stale HitInfo.distance where the reference compares it with the hit.
template <bool CULL>
__device__ __forceinline__ void cull_start(const Scene& sc, Walker& w, const Ray& ray) {
    w.cull_base = INFINITY; w.cull_k = 0.0f;
    w.cull_w = INFINITY; w.cull_kray = INFINITY; w.cull_c = INFINITY;
    if (!CULL) return;
    const float eta = fabsf(dot(ray.d, ray.d) - 1.0f) + 9.5367431640625e-07f;
    if (sc.cull_on && w.usable && eta <= 9.765625e-04f) {
        const float W = fmaxf(fabsf(w.p1.y), fmaxf(fabsf(w.p2.x), fabsf(w.p2.y))) * 1.000001f;
        w.cull_w = W;
        w.cull_kray = W * eta * 1.000001f;
        w.cull_c = sc.cull_cabs + W * sc.cull_delta * 1.000001f;
    }
}
template <bool CULL>
__device__ __forceinline__ void cull_update(const Scene& sc, Walker& w, float D) {
    if (!CULL) return;
    w.cull_base = D * (1.0000153f + w.cull_w * 1.1920928955078125e-07f) + w.cull_c;
    const float dd = D + sc.cull_2rmax;
    w.cull_k = w.cull_kray * dd * dd;
}

template <bool CULL = false>
__device__ __forceinline__ void walker_start(const Scene& sc, const Store& st, Walker& w, const Ray ray, const Rng rng) {
    w.rng = rng;
    walker_set_ray(w, ray, sc.fast_div_ok != 0);
    w.usable = w.usable && fmaxf(fabsf(ray.o.x), fmaxf(fabsf(ray.o.y), fabsf(ray.o.z))) <= sc.entry_guard;   // (entry boxes: see Scene)
    cull_start<CULL>(sc, w, ray);
    w.i = 0; w.resume = 0; w.inst = -1; w.stale = sc.stale_entry; w.alpha_off = -1; w.alpha_wh = 0;
    w.f0 = 0; w.f1 = 0; w.f2 = 0; w.cnt = 0; w.reach = 0.0f;
    w.c.dist = FLT_MAX; w.c.best = -1; w.c.u = 0; w.c.v = 0; w.c.hit = false;
    walker_load(st, w);
}

// The traversal loop.  BVH_Node::Intersect (Object.cpp:241-252) visits both children of every node whose box is hit, no
// matter what the primitives below return, so the sequence of entries a ray visits does not depend on any primitive
// test.  A lane therefore does not stop at a leaf: it notes the leaf in a short per-lane FIFO (f0, f1, f2; oldest
// first) and walks on; the primitive tests are run later, in visiting order, with many lanes at once (prim_step).
//
// One iteration, for the lanes of the wave that can move:
//   * lanes on an inner entry whose ray is admitted to the reciprocal path: BoundingBox::DoesRayHit
//     (DataUtility.cpp:35-82), then  i = hit ? i + 1 : skip;
//   * lanes on a sphere / cube / triangle / fog entry (kinds 1..4) with room in their FIFO: note i, then i = i + 1;
//   * both: fetch entry i.
// It repeats while at least `quorum` lanes stand on inner entries (and runs once more for the stragglers).  Lanes on
// anything else (END, INSTANCE, boxes or rays that need the general test, a full FIFO) wait for the caller.
//
// This is the instruction stream the renderer spends most of its time in, so it is written out by hand: 25 vector
// instructions per box step where the compiler's schedule of the same arithmetic needs 50 to 70 (it keeps the entry in
// four-register tuples, rebuilds operand pairs and copies loop-carried values around the loop).
//   * the entry holds the bounds of an axis side by side (min.x, max.x) (min.y, max.y) (min.z, max.z): one packed
//     subtract, multiply and four packed FMAs per axis give both quotients (div_by above, two lanes wide); per-ray
//     constants are broadcast out of their pairs with op_sel, so they need no registers of their own;
//   * the three axes are interleaved: a packed result is never consumed by the next instruction (which would need a
//     wait state on gfx950);
//   * only boxes with min <= max and rays admitted to the reciprocal path come here (finite quotients only).  The
//     reference rejects when an axis's entry distance exceeds ANOTHER axis's exit distance (DataUtility.cpp:56,71); an
//     axis's own entry never exceeds its own exit for such a box, so "not rejected" is max3(entries) <= min3(exits), and
//     the last test, both distances negative (:78), is then "exit < 0".  Entry/exit per axis are picked by the sign of
//     the direction (px, py, pz: lane masks of d >= 0), as the reference does;
//   * the decision is first made on one-multiplication quotients (c - o) * (1/d), each within 1.5 * 2^-23 of the true
//     quotient's rounding; when the largest entry and the smallest exit distance are further apart than 2^-20 of their
//     magnitude the exactly rounded quotients order the same way (and a quotient's sign is that of its product), so the
//     decision stands.  Only if some lane is closer than that does the wave compute the exact quotients (the four packed
//     FMAs per axis) and decide again — a few box tests in a million.  "Closer than that" is tested on the bit patterns:
//     two floats within 2^-20 of their magnitude have the same sign and lie at most 16 representable values apart, so
//     the difference of their patterns is within [-32, 31] (three integer instructions; distances of either sign that
//     are both zero need no second look: a zero quotient is exactly zero);
//   * the loads land in the registers that already hold the entry: nothing is copied (from LDS as tied operands; from
//     memory the entry is bound to v22..v29, because a 16-byte load needs a register quadruple and the operands are
//     pairs);
//   * the lanes that act are enabled through exec inside the block (restored before every test and at the end), so the
//     surrounding control flow is wave-uniform.
// Scratch registers are used by name because the halves of a register pair cannot be named through operands: v2..v15
// with the scene in memory, v8..v15 with the scene in LDS (there products and selects are formed in place, and the exact
// quotients are computed one axis after the other, so that the kernel variants that run six waves per SIMD stay within
// their 80 registers).  The exact quotients' residuals go into the registers of the entry itself, which the box step has
// consumed by then and the fetch at the end of the iteration overwrites for every lane that stepped.
// Every lane of the wave must be enabled at the call.
// RT_PROFILE builds (make PROF=1) time the phases of every round with the shader clock; see intersect_kernel
struct Prof { unsigned long long box = 0, prim = 0, special = 0, refill = 0; unsigned rounds = 0, prims = 0, prim_lanes = 0, specials = 0, box_iters = 0, box_lanes = 0, finishes = 0, finish_lanes = 0;
              unsigned sp_exec[4] = {0, 0, 0, 0}, sp_lanes[4] = {0, 0, 0, 0}; };   // state-changing entries by kind: end of walk, end of instance, instance, box tested as written
#ifdef RT_PROFILE
#define RT_PROF(x) x
#else
#define RT_PROF(x)
#endif
#ifdef RT_PROFILE   // iterations of the loop and lanes that moved, summed per wave (the counters ride in two SGPRs)
#define RT_BOX_COUNT "s_add_u32 %[iters], %[iters], 1\n\ts_bcnt1_i32_b64 %[cntS], %[mall]\n\ts_add_u32 %[lanes], %[lanes], %[cntS]\n\t"
#define RT_BOX_COUNT_OPERANDS , [iters] "+s"(box_iters), [lanes] "+s"(box_lanes)
#else
#define RT_BOX_COUNT
#define RT_BOX_COUNT_OPERANDS
#endif
template <bool LDS>
__device__ __forceinline__ void box_steps(const Store& st, Walker& w, int quorum, Prof& prof, uint32_t& n_box) {
    const unsigned long long px = __ballot(w.p3.x >= 0), py = __ballot(w.p3.y >= 0), pz = __ballot(w.p4.x >= 0);
    const unsigned long long usable = __ballot(w.usable);
    unsigned long long m, m2, m3, m4, mall, saved;
    int n_inner;
    uint32_t nb = 0;   // slab tests executed by this call (lanes x iterations), summed in a scalar register
    RT_PROF(int box_iters = 0; int box_lanes = 0;)
    if (LDS) {
        const uint32_t base = (uint32_t)(uintptr_t)st.ent;   // LDS addresses are 32-bit offsets
        asm volatile(
                 "s_mov_b64 %[saved], exec\n\t"
                 "1:\n\t"
                 "v_cmp_gt_u32_e32 vcc, 0x10000000, %[sk]\n\t"
                 "s_and_b64 %[m], vcc, %[usable]\n\t"
                 "v_add_u32_e32 v9, 0xf0000000, %[sk]\n\t"
                 "v_cmp_gt_u32_e64 %[m2], 3, %[cnt]\n\t"
                 "v_cmp_gt_u32_e32 vcc, 0x40000000, v9\n\t"
                 "s_and_b64 %[m2], %[m2], vcc\n\t"
                 "s_or_b64 %[mall], %[m], %[m2]\n\t"
                 "s_cmp_eq_u64 %[mall], 0\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_cmp_eq_u64 %[m], 0\n\t"
                 "s_cbranch_scc1 3f\n\t"
                 "s_and_b64 exec, %[saved], %[m]\n\t"
                 "v_pk_add_f32 v[10:11], %[e01], %[p0] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 v[12:13], %[e23], %[p0] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 v[14:15], %[e45], %[p1] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_mul_f32 v[10:11], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                 "v_pk_mul_f32 v[12:13], v[12:13], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
                 "v_pk_mul_f32 v[14:15], v[14:15], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                 "v_cndmask_b32_e64 v8, v11, v10, %[px]\n\t"
                 "v_cndmask_b32_e64 v11, v10, v11, %[px]\n\t"
                 "v_cndmask_b32_e64 v10, v13, v12, %[py]\n\t"
                 "v_cndmask_b32_e64 v13, v12, v13, %[py]\n\t"
                 "v_cndmask_b32_e64 v12, v15, v14, %[pz]\n\t"
                 "v_cndmask_b32_e64 v15, v14, v15, %[pz]\n\t"
                 "v_max3_f32 v8, v8, v10, v12\n\t"
                 "v_min3_f32 v11, v11, v13, v15\n\t"
                 "v_cmp_le_f32_e32 vcc, v8, v11\n\t"
                 "v_cmp_nlt_f32_e64 %[m3], v11, 0\n\t"
                 "v_sub_u32_e32 v9, v11, v8\n\t"
                 "s_and_b64 vcc, vcc, %[m3]\n\t"
                 "v_add_u32_e32 v9, 32, v9\n\t"
                 "v_add_u32_e32 v10, 1, %[i]\n\t"
                 "v_cmp_gt_u32_e64 %[m3], 64, v9\n\t"
                 "s_cmp_eq_u64 %[m3], 0\n\t"
                 "s_cbranch_scc1 5f\n\t"
                 "v_pk_add_f32 v[10:11], %[e01], %[p0] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_mul_f32 v[12:13], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                 "v_pk_fma_f32 %[e01], %[p3], v[12:13], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[12:13], %[e01], %[p1], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %[e01], %[p3], v[12:13], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[14:15], %[e01], %[p1], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_add_f32 v[10:11], %[e23], %[p0] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_mul_f32 v[12:13], v[10:11], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
                 "v_pk_fma_f32 %[e23], %[p3], v[12:13], v[10:11] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[12:13], %[e23], %[p2], v[12:13] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 %[e23], %[p3], v[12:13], v[10:11] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[10:11], %[e23], %[p2], v[12:13] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_add_f32 %[e01], %[e45], %[p1] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_mul_f32 v[12:13], %[e01], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                 "v_pk_fma_f32 %[e45], %[p4], v[12:13], %[e01] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[12:13], %[e45], %[p2], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %[e45], %[p4], v[12:13], %[e01] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[12:13], %[e45], %[p2], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_cndmask_b32_e64 v8, v15, v14, %[px]\n\t"
                 "v_cndmask_b32_e64 v15, v14, v15, %[px]\n\t"
                 "v_cndmask_b32_e64 v14, v11, v10, %[py]\n\t"
                 "v_cndmask_b32_e64 v11, v10, v11, %[py]\n\t"
                 "v_cndmask_b32_e64 v10, v13, v12, %[pz]\n\t"
                 "v_cndmask_b32_e64 v13, v12, v13, %[pz]\n\t"
                 "v_max3_f32 v8, v8, v14, v10\n\t"
                 "v_min3_f32 v11, v15, v11, v13\n\t"
                 "v_cmp_le_f32_e32 vcc, v8, v11\n\t"
                 "v_cmp_nlt_f32_e64 %[m3], v11, 0\n\t"
                 "v_add_u32_e32 v10, 1, %[i]\n\t"
                 "s_and_b64 vcc, vcc, %[m3]\n\t"
                 "5:\n\t"
                 "s_nop 0\n\t"
                 "v_cndmask_b32_e32 %[i], %[sk], v10, vcc\n\t"
                 "3:\n\t"
                 "s_cmp_eq_u64 %[m2], 0\n\t"
                 "s_cbranch_scc1 4f\n\t"
                 "s_and_b64 exec, %[saved], %[m2]\n\t"
                 "v_cmp_eq_u32_e64 %[m3], 0, %[cnt]\n\t"
                 "v_cmp_eq_u32_e32 vcc, 1, %[cnt]\n\t"
                 "v_mov_b32_e32 %[f2], %[i]\n\t"
                 "v_add_u32_e32 %[cnt], 1, %[cnt]\n\t"
                 "v_cndmask_b32_e64 %[f0], %[f0], %[i], %[m3]\n\t"
                 "v_cndmask_b32_e32 %[f1], %[f1], %[i], vcc\n\t"
                 "v_add_u32_e32 %[i], 1, %[i]\n\t"
                 "4:\n\t"
                 "s_and_b64 exec, %[saved], %[mall]\n\t"
                 RT_BOX_COUNT
                 "v_lshl_add_u32 v8, %[i], 5, %[base]\n\t"
                 "ds_read_b64 %[e01], v8\n\t"
                 "ds_read_b64 %[e23], v8 offset:8\n\t"
                 "ds_read_b64 %[e45], v8 offset:16\n\t"
                 "ds_read_b32 %[sk], v8 offset:24\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b64 exec, %[saved]\n\t"
                 "s_bcnt1_i32_b64 %[cntS], %[m]\n\t"
                 "s_add_u32 %[nbox], %[nbox], %[cntS]\n\t"
                 "s_cmp_ge_u32 %[cntS], %[quorum]\n\t"
                 "s_cbranch_scc1 1b\n\t"
                 "2:\n\t"
                 : [e01] "+v"(w.e01), [e23] "+v"(w.e23), [e45] "+v"(w.e45), [sk] "+v"(w.sk), [i] "+v"(w.i), [f0] "+v"(w.f0),
                   [f1] "+v"(w.f1), [f2] "+v"(w.f2), [cnt] "+v"(w.cnt), [m] "=&s"(m), [m2] "=&s"(m2), [m3] "=&s"(m3),
                   [mall] "=&s"(mall), [saved] "=&s"(saved), [cntS] "=&s"(n_inner), [nbox] "+s"(nb) RT_BOX_COUNT_OPERANDS
                 : [p0] "v"(w.p0), [p1] "v"(w.p1), [p2] "v"(w.p2), [p3] "v"(w.p3), [p4] "v"(w.p4), [px] "s"(px), [py] "s"(py),
                   [pz] "s"(pz), [base] "s"(base), [usable] "s"(usable), [quorum] "s"(quorum)
                 : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "vcc", "scc", "memory");
    } else {
        const float4* base = st.ent;
        asm volatile(
                 "s_mov_b64 %[saved], exec\n\t"
                 "1:\n\t"
                 "v_cmp_gt_u32_e32 vcc, 0x10000000, %[sk]\n\t"
                 "s_and_b64 %[m], vcc, %[usable]\n\t"
                 "v_add_u32_e32 v9, 0xf0000000, %[sk]\n\t"
                 "v_cmp_gt_u32_e64 %[m2], 3, %[cnt]\n\t"
                 "v_cmp_gt_u32_e32 vcc, 0x40000000, v9\n\t"
                 "s_and_b64 %[m2], %[m2], vcc\n\t"
                 "s_or_b64 %[mall], %[m], %[m2]\n\t"
                 "s_cmp_eq_u64 %[mall], 0\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_cmp_eq_u64 %[m], 0\n\t"
                 "s_cbranch_scc1 3f\n\t"
                 "s_and_b64 exec, %[saved], %[m]\n\t"
                 "v_pk_add_f32 v[10:11], %[e01], %[p0] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 v[12:13], %[e23], %[p0] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 v[14:15], %[e45], %[p1] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_mul_f32 v[2:3], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                 "v_pk_mul_f32 v[4:5], v[12:13], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
                 "v_pk_mul_f32 v[6:7], v[14:15], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                 "v_cndmask_b32_e64 v8, v3, v2, %[px]\n\t"
                 "v_cndmask_b32_e64 v3, v2, v3, %[px]\n\t"
                 "v_cndmask_b32_e64 v2, v5, v4, %[py]\n\t"
                 "v_cndmask_b32_e64 v5, v4, v5, %[py]\n\t"
                 "v_cndmask_b32_e64 v4, v7, v6, %[pz]\n\t"
                 "v_cndmask_b32_e64 v7, v6, v7, %[pz]\n\t"
                 "v_max3_f32 v8, v8, v2, v4\n\t"
                 "v_min3_f32 v3, v3, v5, v7\n\t"
                 "v_fma_f32 v5, %[ck], %[rho], %[cbase]\n\t"
                 "v_cmp_le_f32_e32 vcc, v8, v3\n\t"
                 "v_cmp_nlt_f32_e64 %[m3], v3, 0\n\t"
                 "v_cmp_ngt_f32_e64 %[m4], v8, v5\n\t"
                 "v_sub_u32_e32 v9, v3, v8\n\t"
                 "s_and_b64 vcc, vcc, %[m3]\n\t"
                 "v_add_u32_e32 v9, 32, v9\n\t"
                 "v_add_u32_e32 v4, 1, %[i]\n\t"
                 "v_cmp_gt_u32_e64 %[m3], 64, v9\n\t"
                 "s_cmp_eq_u64 %[m3], 0\n\t"
                 "s_cbranch_scc1 5f\n\t"
                 "v_pk_mul_f32 v[2:3], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                 "v_pk_mul_f32 v[4:5], v[12:13], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
                 "v_pk_mul_f32 v[6:7], v[14:15], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                 "v_pk_fma_f32 %[e01], %[p3], v[2:3], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 %[e23], %[p3], v[4:5], v[12:13] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 %[e45], %[p4], v[6:7], v[14:15] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[2:3], %[e01], %[p1], v[2:3] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[4:5], %[e23], %[p2], v[4:5] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[6:7], %[e45], %[p2], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %[e01], %[p3], v[2:3], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 %[e23], %[p3], v[4:5], v[12:13] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 %[e45], %[p4], v[6:7], v[14:15] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[2:3], %[e01], %[p1], v[2:3] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[4:5], %[e23], %[p2], v[4:5] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[6:7], %[e45], %[p2], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_cndmask_b32_e64 v8, v3, v2, %[px]\n\t"
                 "v_cndmask_b32_e64 v3, v2, v3, %[px]\n\t"
                 "v_cndmask_b32_e64 v2, v5, v4, %[py]\n\t"
                 "v_cndmask_b32_e64 v5, v4, v5, %[py]\n\t"
                 "v_cndmask_b32_e64 v4, v7, v6, %[pz]\n\t"
                 "v_cndmask_b32_e64 v7, v6, v7, %[pz]\n\t"
                 "v_max3_f32 v8, v8, v2, v4\n\t"
                 "v_min3_f32 v3, v3, v5, v7\n\t"
                 "v_cmp_le_f32_e32 vcc, v8, v3\n\t"
                 "v_cmp_nlt_f32_e64 %[m3], v3, 0\n\t"
                 "v_add_u32_e32 v4, 1, %[i]\n\t"
                 "s_and_b64 vcc, vcc, %[m3]\n\t"
                 "5:\n\t"
                 "s_and_b64 vcc, vcc, %[m4]\n\t"
                 "s_nop 0\n\t"
                 "v_cndmask_b32_e32 %[i], %[sk], v4, vcc\n\t"
                 "3:\n\t"
                 "s_cmp_eq_u64 %[m2], 0\n\t"
                 "s_cbranch_scc1 4f\n\t"
                 "s_and_b64 exec, %[saved], %[m2]\n\t"
                 "v_cmp_eq_u32_e64 %[m3], 0, %[cnt]\n\t"
                 "v_cmp_eq_u32_e32 vcc, 1, %[cnt]\n\t"
                 "v_mov_b32_e32 %[f2], %[i]\n\t"
                 "v_add_u32_e32 %[cnt], 1, %[cnt]\n\t"
                 "v_cndmask_b32_e64 %[f0], %[f0], %[i], %[m3]\n\t"
                 "v_cndmask_b32_e32 %[f1], %[f1], %[i], vcc\n\t"
                 "v_add_u32_e32 %[i], 1, %[i]\n\t"
                 "4:\n\t"
                 "s_and_b64 exec, %[saved], %[mall]\n\t"
                 RT_BOX_COUNT
                 "v_lshlrev_b32_e32 v2, 5, %[i]\n\t"
                 "global_load_dwordx4 v[22:25], v2, %[base]\n\t"
                 "global_load_dwordx4 v[26:29], v2, %[base] offset:16\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %[saved]\n\t"
                 "s_bcnt1_i32_b64 %[cntS], %[m]\n\t"
                 "s_add_u32 %[nbox], %[nbox], %[cntS]\n\t"
                 "s_cmp_ge_u32 %[cntS], %[quorum]\n\t"
                 "s_cbranch_scc1 1b\n\t"
                 "2:\n\t"
                 : [e01] "+{v[22:23]}"(w.e01), [e23] "+{v[24:25]}"(w.e23), [e45] "+{v[26:27]}"(w.e45), [sk] "+{v28}"(w.sk), [i] "+v"(w.i), [f0] "+v"(w.f0),
                   [f1] "+v"(w.f1), [f2] "+v"(w.f2), [cnt] "+v"(w.cnt), [m] "=&s"(m), [m2] "=&s"(m2), [m3] "=&s"(m3),
                   [mall] "=&s"(mall), [saved] "=&s"(saved), [cntS] "=&s"(n_inner), [m4] "=&s"(m4), [rho] "+{v29}"(w.rho), [nbox] "+s"(nb) RT_BOX_COUNT_OPERANDS
                 : [p0] "v"(w.p0), [p1] "v"(w.p1), [p2] "v"(w.p2), [p3] "v"(w.p3), [p4] "v"(w.p4), [px] "s"(px), [py] "s"(py),
                   [pz] "s"(pz), [base] "s"(base), [usable] "s"(usable), [quorum] "s"(quorum), [ck] "v"(w.cull_k), [cbase] "v"(w.cull_base)
                 : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "vcc", "scc", "memory");
    }
    n_box += nb;
    RT_PROF(prof.box_iters += box_iters; prof.box_lanes += box_lanes;)
}

// ---------------------------------------------------------------------------------------------------- ordered traversal
// For scenes that are read from memory (no fog).  The reference takes a node's children in the order it stored them
// and, having no limit on the distance, descends into both.  Distance culling (above) prunes the boxes that lie behind
// the closest hit — the more the sooner a close hit is found.  This traversal therefore visits the child on the ray's
// near side first: the children of a node were ordered by their boxes' minimum along the node's axis, so the first
// child is the near one when the ray's direction component along that axis is not negative.  The other child waits on
// a per-lane stack in LDS (slot s of thread t at dword s * BLOCK + t: conflict-free), and a lane that misses a box or
// notes a leaf continues with what it pops.  The boxes a ray tests are still a subset of the reference's and every one
// of them is decided as the reference decides it; what changes is the order in which leaves are met, which the
// reference's result depends on in exactly two places, both restated in closest_update:
//   * equal distances: BVH_Node::Intersect keeps the hit it met first (Object.cpp:248, strict <).  Entries are numbered
//     in the reference's visiting order, so of two leaves at the same distance the one with the smaller rank wins;
//   * the first leaf's rejected HitInfo.distance survives only if nothing is hit at all: it is recorded only while the
//     ray has no hit, and any hit overwrites it.
// A culled box holds no leaf that could tie (the bound is strict), so culling and ordering do not interact.
// The records of the tree's upper levels (most of the 160 KB that the stacks leave) are copied into LDS by every block:
// steps on them cost no memory access.  A link with LINK_LDS set names such a record; its first child is named in the
// record (eleft), an entry's first child is the next entry.  Slot 0 of every stack holds the END entry of the top-level
// list, so the pop after the last subtree ends the walk; entering a mesh instance pushes the END entry of its range.
__device__ __forceinline__ void ord_push(const Store& st, Walker& w, uint32_t link, uint32_t stride) {
    st.lds0[(w.sp - st.lds0_addr) >> 2] = link;
    w.sp += stride;
}
__device__ __forceinline__ uint32_t ord_pop(const Store& st, Walker& w, uint32_t stride) {
    w.sp -= stride;
    return st.lds0[(w.sp - st.lds0_addr) >> 2];
}
__device__ __forceinline__ void walker_load_ord(const Store& st, Walker& w) {
    const uint32_t i = (uint32_t)w.i;
    if (i & rt::LINK_LDS) {
        const uint32_t* e = st.lds0 + ((st.top_addr - st.lds0_addr) >> 2) + rt::TOP_REC_DWORDS * (i & rt::LINK_INDEX);
        w.e01 = f2{__uint_as_float(e[0]), __uint_as_float(e[1])}; w.e23 = f2{__uint_as_float(e[2]), __uint_as_float(e[3])};
        w.e45 = f2{__uint_as_float(e[4]), __uint_as_float(e[5])};
        w.sk = e[6]; w.rho = __uint_as_float(e[7]); w.eleft = e[8];
    } else walker_load(st, w);
}
template <bool CULL, int BLOCK>
__device__ __forceinline__ void walker_start_ord(const Scene& sc, const Store& st, Walker& w, const Ray ray, const Rng rng) {
    w.rng = rng;
    walker_set_ray(w, ray, sc.fast_div_ok != 0);
    w.usable = w.usable && fmaxf(fabsf(ray.o.x), fmaxf(fabsf(ray.o.y), fabsf(ray.o.z))) <= sc.entry_guard;   // (entry boxes: see Scene)
    cull_start<CULL>(sc, w, ray);
    w.i = (int)st.root_link; w.resume = 0; w.inst = -1; w.stale = sc.stale_entry; w.alpha_off = -1; w.alpha_wh = 0;
    w.f0 = 0; w.f1 = 0; w.f2 = 0; w.cnt = 0; w.reach = 0.0f;
    w.c.dist = FLT_MAX; w.c.best = -1; w.c.u = 0; w.c.v = 0; w.c.hit = false;
    w.sp = st.stack_addr + 4u * ((uint32_t)BLOCK + threadIdx.x);     // slot 1: slot 0 holds the end of the top-level list
    walker_load_ord(st, w);
}
// Does leaf entry a come before leaf entry b in the reference's visiting order?  Every leaf carries its rank in that
// order in a dword its kind leaves free (a triangle in dword 0, a sphere or a cube in dword 7).  (In the reference's layout
// kept beside a regrouped one that dword names the leaf's entry in the regrouped layout instead.)
__device__ inline uint32_t leaf_extra(const Scene& sc, int e) {
    const uint32_t* E = reinterpret_cast<const uint32_t*>(sc.ent) + 8 * (size_t)e;
    return sk_kind(E[6]) == rt::ENT_TRI ? E[0] : E[7];
}
__device__ inline bool rank_before(const Scene& sc, int a, int b) { return leaf_extra(sc, a) < leaf_extra(sc, b); }
// The traversal loop of the ordered traversal: box_steps (above) with the stack in place of the skip links, and with
// the entry fetched from the LDS table or from memory, whichever the link says.  Per iteration, for the lanes that act:
//   * lanes on an inner entry: the slab test and the culling bound as in box_steps; on a hit the far child is pushed
//     (if there are two) and the lane goes to the near one, otherwise it pops;
//   * lanes on a leaf with room in their FIFO: note it, pop;
//   * both: fetch the entry they stand on now.
// `signs`: bit k of a lane is set when component k of its ray's direction is negative.
template <int BLOCK>
__device__ __forceinline__ void box_steps_ord(const Store& st, Walker& w, int quorum, uint32_t& n_box) {
    const unsigned long long px = __ballot(w.p3.x >= 0), py = __ballot(w.p3.y >= 0), pz = __ballot(w.p4.x >= 0);
    const unsigned long long usable = __ballot(w.usable);
    const uint32_t signs = (w.p3.x >= 0 ? 0u : 1u) | (w.p3.y >= 0 ? 0u : 2u) | (w.p4.x >= 0 ? 0u : 4u);
    unsigned long long m, m2, m3, m4, mall, saved, mlds, mhit, mpop;
    int n_inner;
    uint32_t nb = 0;
    const float4* base = st.ent;
    const uint32_t ldsbase = st.top_addr;
    asm volatile(
             "s_mov_b64 %[saved], exec\n\t"
             "1:\n\t"
             "v_cmp_le_u32_e32 vcc, 0x4000000, %[i]\n\t"
             "s_mov_b64 %[mlds], vcc\n\t"
             "v_cmp_gt_u32_e32 vcc, 0x10000000, %[sk]\n\t"
             "s_and_b64 %[m], vcc, %[usable]\n\t"
             "v_add_u32_e32 v9, 0xf0000000, %[sk]\n\t"
             "v_cmp_gt_u32_e64 %[m2], 3, %[cnt]\n\t"
             "v_cmp_gt_u32_e32 vcc, 0x40000000, v9\n\t"
             "s_and_b64 %[m2], %[m2], vcc\n\t"
             "s_or_b64 %[mall], %[m], %[m2]\n\t"
             "s_cmp_eq_u64 %[mall], 0\n\t"
             "s_cbranch_scc1 2f\n\t"
             "s_mov_b64 %[mpop], 0\n\t"
             "s_cmp_eq_u64 %[m], 0\n\t"
             "s_cbranch_scc1 3f\n\t"
             "s_and_b64 exec, %[saved], %[m]\n\t"
             "v_pk_add_f32 v[10:11], %[e01], %[p0] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_add_f32 v[12:13], %[e23], %[p0] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_add_f32 v[14:15], %[e45], %[p1] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_mul_f32 v[2:3], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_mul_f32 v[4:5], v[12:13], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
             "v_pk_mul_f32 v[6:7], v[14:15], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_cndmask_b32_e64 v8, v3, v2, %[px]\n\t"
             "v_cndmask_b32_e64 v3, v2, v3, %[px]\n\t"
             "v_cndmask_b32_e64 v2, v5, v4, %[py]\n\t"
             "v_cndmask_b32_e64 v5, v4, v5, %[py]\n\t"
             "v_cndmask_b32_e64 v4, v7, v6, %[pz]\n\t"
             "v_cndmask_b32_e64 v7, v6, v7, %[pz]\n\t"
             "v_max3_f32 v8, v8, v2, v4\n\t"
             "v_min3_f32 v3, v3, v5, v7\n\t"
             "v_fma_f32 v5, %[ck], %[rho], %[cbase]\n\t"
             "v_cmp_le_f32_e32 vcc, v8, v3\n\t"
             "v_cmp_nlt_f32_e64 %[m3], v3, 0\n\t"
             "v_cmp_ngt_f32_e64 %[m4], v8, v5\n\t"
             "v_sub_u32_e32 v9, v3, v8\n\t"
             "s_and_b64 vcc, vcc, %[m3]\n\t"
             "v_add_u32_e32 v9, 32, v9\n\t"
             "v_add_u32_e32 v4, 1, %[i]\n\t"
             "v_cmp_gt_u32_e64 %[m3], 64, v9\n\t"
             "s_cmp_eq_u64 %[m3], 0\n\t"
             "s_cbranch_scc1 5f\n\t"
             "v_pk_mul_f32 v[2:3], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_mul_f32 v[4:5], v[12:13], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
             "v_pk_mul_f32 v[6:7], v[14:15], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_fma_f32 %[e01], %[p3], v[2:3], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 %[e23], %[p3], v[4:5], v[12:13] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 %[e45], %[p4], v[6:7], v[14:15] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[2:3], %[e01], %[p1], v[2:3] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_fma_f32 v[4:5], %[e23], %[p2], v[4:5] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
             "v_pk_fma_f32 v[6:7], %[e45], %[p2], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_fma_f32 %[e01], %[p3], v[2:3], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 %[e23], %[p3], v[4:5], v[12:13] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 %[e45], %[p4], v[6:7], v[14:15] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[2:3], %[e01], %[p1], v[2:3] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_fma_f32 v[4:5], %[e23], %[p2], v[4:5] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
             "v_pk_fma_f32 v[6:7], %[e45], %[p2], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_cndmask_b32_e64 v8, v3, v2, %[px]\n\t"
             "v_cndmask_b32_e64 v3, v2, v3, %[px]\n\t"
             "v_cndmask_b32_e64 v2, v5, v4, %[py]\n\t"
             "v_cndmask_b32_e64 v5, v4, v5, %[py]\n\t"
             "v_cndmask_b32_e64 v4, v7, v6, %[pz]\n\t"
             "v_cndmask_b32_e64 v7, v6, v7, %[pz]\n\t"
             "v_max3_f32 v8, v8, v2, v4\n\t"
             "v_min3_f32 v3, v3, v5, v7\n\t"
             "v_cmp_le_f32_e32 vcc, v8, v3\n\t"
             "v_cmp_nlt_f32_e64 %[m3], v3, 0\n\t"
             "v_add_u32_e32 v4, 1, %[i]\n\t"
             "s_and_b64 vcc, vcc, %[m3]\n\t"
             "5:\n\t"
             "s_and_b64 %[mhit], vcc, %[m4]\n\t"                    // box hit and not culled (zero outside the acting lanes)
             "v_and_b32_e32 v2, 3, %[rho]\n\t"                      // axis; 3 = a single child
             "v_lshrrev_b32_e32 v3, v2, %[signs]\n\t"
             "v_and_b32_e32 v3, 1, v3\n\t"                          // 1: the direction is negative along the axis
             "v_cndmask_b32_e64 v5, v4, %[eleft], %[mlds]\n\t"      // first child: the next entry, or what the table record says
             "v_cmp_eq_u32_e32 vcc, 1, v3\n\t"
             "v_cndmask_b32_e32 v6, v5, %[sk], vcc\n\t"             // near child (an inner entry's dword 6 IS its second child's link)
             "v_cndmask_b32_e32 v7, %[sk], v5, vcc\n\t"             // far child
             "v_cmp_ne_u32_e32 vcc, 3, v2\n\t"
             "s_and_b64 vcc, vcc, %[mhit]\n\t"
             "s_mov_b64 exec, vcc\n\t"                              // hit, two children: the far one waits on the stack
             "ds_write_b32 %[sp], v7\n\t"
             "v_add_u32_e32 %[sp], %[stride], %[sp]\n\t"
             "s_mov_b64 exec, %[mhit]\n\t"
             "v_mov_b32_e32 %[i], v6\n\t"
             "s_andn2_b64 %[mpop], %[m], %[mhit]\n\t"               // missed or culled: pop
             "3:\n\t"
             "s_cmp_eq_u64 %[m2], 0\n\t"
             "s_cbranch_scc1 4f\n\t"
             "s_and_b64 exec, %[saved], %[m2]\n\t"
             "v_cmp_eq_u32_e64 %[m3], 0, %[cnt]\n\t"
             "v_cmp_eq_u32_e32 vcc, 1, %[cnt]\n\t"
             "v_mov_b32_e32 %[f2], %[i]\n\t"
             "v_add_u32_e32 %[cnt], 1, %[cnt]\n\t"
             "v_cndmask_b32_e64 %[f0], %[f0], %[i], %[m3]\n\t"
             "v_cndmask_b32_e32 %[f1], %[f1], %[i], vcc\n\t"
             "s_or_b64 %[mpop], %[mpop], %[m2]\n\t"
             "4:\n\t"
             "s_cmp_eq_u64 %[mpop], 0\n\t"
             "s_cbranch_scc1 6f\n\t"
             "s_mov_b64 exec, %[mpop]\n\t"
             "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n\t"
             "ds_read_b32 %[i], %[sp]\n\t"
             "s_waitcnt lgkmcnt(0)\n\t"
             "6:\n\t"
             "s_and_b64 exec, %[saved], %[mall]\n\t"
             "v_cmp_le_u32_e32 vcc, 0x4000000, %[i]\n\t"            // which of the new entries are table records
             "s_andn2_b64 exec, exec, vcc\n\t"
             "v_lshlrev_b32_e32 v2, 5, %[i]\n\t"
             "global_load_dwordx4 v[22:25], v2, %[base]\n\t"
             "global_load_dwordx4 v[26:29], v2, %[base] offset:16\n\t"
             "s_cmp_eq_u64 vcc, 0\n\t"
             "s_cbranch_scc1 7f\n\t"
             "s_mov_b64 exec, vcc\n\t"
             "v_and_b32_e32 v2, 0x3ffffff, %[i]\n\t"
             "v_mad_u32_u24 v2, v2, 40, %[ldsbase]\n\t"
             "ds_read_b64 v[22:23], v2\n\t"
             "ds_read_b64 v[24:25], v2 offset:8\n\t"
             "ds_read_b64 v[26:27], v2 offset:16\n\t"
             "ds_read_b64 v[28:29], v2 offset:24\n\t"
             "ds_read_b32 %[eleft], v2 offset:32\n\t"
             "7:\n\t"
             "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
             "s_mov_b64 exec, %[saved]\n\t"
             "s_bcnt1_i32_b64 %[cntS], %[m]\n\t"
             "s_add_u32 %[nbox], %[nbox], %[cntS]\n\t"
             "s_cmp_ge_u32 %[cntS], %[quorum]\n\t"
             "s_cbranch_scc1 1b\n\t"
             "2:\n\t"
             : [e01] "+{v[22:23]}"(w.e01), [e23] "+{v[24:25]}"(w.e23), [e45] "+{v[26:27]}"(w.e45), [sk] "+{v28}"(w.sk), [i] "+v"(w.i), [f0] "+v"(w.f0),
               [f1] "+v"(w.f1), [f2] "+v"(w.f2), [cnt] "+v"(w.cnt), [m] "=&s"(m), [m2] "=&s"(m2), [m3] "=&s"(m3),
               [mall] "=&s"(mall), [saved] "=&s"(saved), [cntS] "=&s"(n_inner), [m4] "=&s"(m4), [rho] "+{v29}"(w.rho), [nbox] "+s"(nb),
               [mlds] "=&s"(mlds), [mhit] "=&s"(mhit), [mpop] "=&s"(mpop), [sp] "+v"(w.sp), [eleft] "+v"(w.eleft)
             : [p0] "v"(w.p0), [p1] "v"(w.p1), [p2] "v"(w.p2), [p3] "v"(w.p3), [p4] "v"(w.p4), [px] "s"(px), [py] "s"(py),
               [pz] "s"(pz), [base] "s"(base), [usable] "s"(usable), [quorum] "s"(quorum), [ck] "v"(w.cull_k), [cbase] "v"(w.cull_base),
               [signs] "v"(signs), [ldsbase] "s"(ldsbase), [stride] "n"(BLOCK * 4)
             : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "vcc", "scc", "memory");
    n_box += nb;
}

// ---------------------------------------------------------------------------------------------------- pair steps
// The ordered traversal above tests one box per step: a lane loads a node, tests its box, and pops when it misses —
// half of all steps on large scenes are such misses, each a round trip to memory.  Pair steps test a node's TWO children
// in one step, from one record that holds both boxes (rt_flat.h, "Pair records"): a child that is missed or culled costs
// its share of the arithmetic and nothing else; of two children that are hit the one the ray enters FIRST (by the entry
// distances just computed, not by a guess from the split axis) is visited first and the other waits on the stack.
// Nothing is tested that the reference does not test, and a node's own box need not be tested on arrival: it was, as
// its parent's child — and for a root (the scene's, a mesh's) not at all, because a box that is missed contains no box
// that is hit (rt_scene.cpp, "regrouping": the slab test is monotone in the box), so the first pair step finds nothing.
// Per iteration, for the lanes that act:
//   * lanes on a pair record: both slab tests and culling bounds as in box_steps_ord; then go to the nearer child that
//     is hit, push the other if it is hit too, pop if none is;
//   * lanes on a leaf entry with room in their FIFO: note it; go on to the next entry when the leaf says that its sibling
//     follows (LEAF_MORE), else pop;
//   * both: fetch the record or entry they stand on now (records of the tree's upper levels from the table in LDS).
// Every other entry (end of a list, instance, a box with min > max, an instance's entry box, an inner entry reached as
// an entry) and every ray outside the reciprocal path waits for walker_special.
// Registers: the record occupies v16..v31 (four 16-byte loads); an entry its first half, dword 6 in v22, dword 7 in v23.
template <int BLOCK>
__device__ __forceinline__ void box_steps_pairs(const Store& st, Walker& w, int quorum, Prof& prof, uint32_t& n_box) {
    const unsigned long long px = __ballot(w.p3.x >= 0), py = __ballot(w.p3.y >= 0), pz = __ballot(w.p4.x >= 0);
    const unsigned long long usable = __ballot(w.usable);
    unsigned long long m, m2, m3, m4, mall, saved, mhit, mpop, ma, mb;
    int n_pair;
    uint32_t nb = 0;
    RT_PROF(uint32_t box_iters = 0; uint32_t box_lanes = 0;)
    const float4* base = st.ent;
    const float4* pbase = st.pair;
    const uint32_t ldsbase = st.top_addr;
    asm volatile(
             "s_mov_b64 %[saved], exec\n\t"
             "1:\n\t"
             "v_cmp_le_u32_e32 vcc, 0x40000000, %[i]\n\t"           // lanes on a pair record
             "s_and_b64 %[m], vcc, %[usable]\n\t"
             "v_add_u32_e32 v9, 0xf0000000, v22\n\t"                // an entry's dword 6
             "v_cmp_gt_u32_e64 %[m2], 3, %[cnt]\n\t"
             "s_andn2_b64 %[m2], %[m2], vcc\n\t"
             "v_cmp_gt_u32_e32 vcc, 0x40000000, v9\n\t"             // kinds 1..4: a leaf
             "s_and_b64 %[m2], %[m2], vcc\n\t"
             "s_or_b64 %[mall], %[m], %[m2]\n\t"
             "s_cmp_eq_u64 %[mall], 0\n\t"
             "s_cbranch_scc1 2f\n\t"
             RT_BOX_COUNT
             "s_mov_b64 %[mpop], 0\n\t"
             "s_cmp_eq_u64 %[m], 0\n\t"
             "s_cbranch_scc1 3f\n\t"
             "s_and_b64 exec, %[saved], %[m]\n\t"
             // both children on one-multiplication quotients: A into v8 (entry), v3 (exit); B into v4, v5
             "v_pk_add_f32 v[10:11], v[16:17], %[p0] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_add_f32 v[12:13], v[18:19], %[p0] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_add_f32 v[14:15], v[20:21], %[p1] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_mul_f32 v[10:11], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_mul_f32 v[12:13], v[12:13], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
             "v_pk_mul_f32 v[14:15], v[14:15], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_add_f32 v[4:5], v[22:23], %[p0] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_add_f32 v[6:7], v[24:25], %[p0] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_cndmask_b32_e64 v8, v11, v10, %[px]\n\t"
             "v_cndmask_b32_e64 v3, v10, v11, %[px]\n\t"
             "v_cndmask_b32_e64 v2, v13, v12, %[py]\n\t"
             "v_cndmask_b32_e64 v9, v12, v13, %[py]\n\t"
             "v_cndmask_b32_e64 v10, v15, v14, %[pz]\n\t"
             "v_cndmask_b32_e64 v11, v14, v15, %[pz]\n\t"
             "v_pk_add_f32 v[12:13], v[26:27], %[p1] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_max3_f32 v8, v8, v2, v10\n\t"
             "v_min3_f32 v3, v3, v9, v11\n\t"
             "v_pk_mul_f32 v[4:5], v[4:5], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_mul_f32 v[6:7], v[6:7], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
             "v_pk_mul_f32 v[12:13], v[12:13], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_sub_u32_e32 v14, v3, v8\n\t"                       // A: entry and exit within 2^-20 of each other?
             "v_add_u32_e32 v14, 32, v14\n\t"
             "v_cndmask_b32_e64 v2, v5, v4, %[px]\n\t"
             "v_cndmask_b32_e64 v5, v4, v5, %[px]\n\t"
             "v_cndmask_b32_e64 v9, v7, v6, %[py]\n\t"
             "v_cndmask_b32_e64 v7, v6, v7, %[py]\n\t"
             "v_cndmask_b32_e64 v10, v13, v12, %[pz]\n\t"
             "v_cndmask_b32_e64 v13, v12, v13, %[pz]\n\t"
             "v_max3_f32 v4, v2, v9, v10\n\t"
             "v_min3_f32 v5, v5, v7, v13\n\t"
             "v_cmp_gt_u32_e64 %[m3], 64, v14\n\t"
             "v_sub_u32_e32 v15, v5, v4\n\t"                       // B likewise
             "v_add_u32_e32 v15, 32, v15\n\t"
             "v_cmp_gt_u32_e64 %[m4], 64, v15\n\t"
             "s_or_b64 %[m3], %[m3], %[m4]\n\t"
             "s_cmp_eq_u64 %[m3], 0\n\t"
             "s_cbranch_scc1 5f\n\t"
             // some lane is that close: the exactly rounded quotients for both children (div_by, two lanes wide)
             "v_pk_add_f32 v[10:11], v[16:17], %[p0] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_mul_f32 v[2:3], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p3], v[2:3], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[2:3], v[14:15], %[p1], v[2:3] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p3], v[2:3], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[2:3], v[14:15], %[p1], v[2:3] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_add_f32 v[10:11], v[18:19], %[p0] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_mul_f32 v[6:7], v[10:11], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
             "v_pk_fma_f32 v[14:15], %[p3], v[6:7], v[10:11] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[6:7], v[14:15], %[p2], v[6:7] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p3], v[6:7], v[10:11] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[6:7], v[14:15], %[p2], v[6:7] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
             "v_pk_add_f32 v[10:11], v[20:21], %[p1] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_mul_f32 v[12:13], v[10:11], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p4], v[12:13], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[12:13], v[14:15], %[p2], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p4], v[12:13], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[12:13], v[14:15], %[p2], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_cndmask_b32_e64 v8, v3, v2, %[px]\n\t"
             "v_cndmask_b32_e64 v3, v2, v3, %[px]\n\t"
             "v_cndmask_b32_e64 v9, v7, v6, %[py]\n\t"
             "v_cndmask_b32_e64 v7, v6, v7, %[py]\n\t"
             "v_cndmask_b32_e64 v10, v13, v12, %[pz]\n\t"
             "v_cndmask_b32_e64 v13, v12, v13, %[pz]\n\t"
             "v_max3_f32 v8, v8, v9, v10\n\t"
             "v_min3_f32 v3, v3, v7, v13\n\t"
             "v_pk_add_f32 v[10:11], v[22:23], %[p0] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_mul_f32 v[4:5], v[10:11], %[p1] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p3], v[4:5], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[4:5], v[14:15], %[p1], v[4:5] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p3], v[4:5], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[4:5], v[14:15], %[p1], v[4:5] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_add_f32 v[10:11], v[24:25], %[p0] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_mul_f32 v[6:7], v[10:11], %[p2] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
             "v_pk_fma_f32 v[14:15], %[p3], v[6:7], v[10:11] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[6:7], v[14:15], %[p2], v[6:7] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p3], v[6:7], v[10:11] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[6:7], v[14:15], %[p2], v[6:7] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
             "v_pk_add_f32 v[10:11], v[26:27], %[p1] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
             "v_pk_mul_f32 v[12:13], v[10:11], %[p2] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p4], v[12:13], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[12:13], v[14:15], %[p2], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_pk_fma_f32 v[14:15], %[p4], v[12:13], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
             "v_pk_fma_f32 v[12:13], v[14:15], %[p2], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
             "v_cndmask_b32_e64 v2, v5, v4, %[px]\n\t"
             "v_cndmask_b32_e64 v5, v4, v5, %[px]\n\t"
             "v_cndmask_b32_e64 v9, v7, v6, %[py]\n\t"
             "v_cndmask_b32_e64 v7, v6, v7, %[py]\n\t"
             "v_cndmask_b32_e64 v10, v13, v12, %[pz]\n\t"
             "v_cndmask_b32_e64 v13, v12, v13, %[pz]\n\t"
             "v_max3_f32 v4, v2, v9, v10\n\t"
             "v_min3_f32 v5, v5, v7, v13\n\t"
             "5:\n\t"
             // hit: entry <= exit, exit not negative, entry not beyond the culling bound of the child
             "v_fma_f32 v6, %[ck], v30, %[cbase]\n\t"
             "v_fma_f32 v7, %[ck], v31, %[cbase]\n\t"
             "v_cmp_le_f32_e32 vcc, v8, v3\n\t"
             "v_cmp_nlt_f32_e64 %[m3], v3, 0\n\t"
             "v_cmp_ngt_f32_e64 %[m4], v8, v6\n\t"
             "s_and_b64 %[ma], vcc, %[m3]\n\t"
             "s_and_b64 %[ma], %[ma], %[m4]\n\t"
             "v_cmp_le_f32_e32 vcc, v4, v5\n\t"
             "v_cmp_nlt_f32_e64 %[m3], v5, 0\n\t"
             "v_cmp_ngt_f32_e64 %[m4], v4, v7\n\t"
             "s_and_b64 %[mb], vcc, %[m3]\n\t"
             "s_and_b64 %[mb], %[mb], %[m4]\n\t"
             "s_and_b64 %[mhit], %[ma], %[mb]\n\t"                  // both: the one entered first goes first
             "v_cmp_lt_f32_e32 vcc, v4, v8\n\t"
             "s_and_b64 vcc, vcc, %[mhit]\n\t"
             "s_andn2_b64 %[m3], %[ma], vcc\n\t"                    // lanes that go to A
             "s_or_b64 %[m4], %[ma], %[mb]\n\t"
             "s_andn2_b64 %[mpop], %[m], %[m4]\n\t"                 // neither: pop
             "v_cndmask_b32_e64 v6, v29, v28, %[m3]\n\t"
             "v_cndmask_b32_e64 v7, v28, v29, %[m3]\n\t"
             "s_mov_b64 exec, %[mhit]\n\t"
             "ds_write_b32 %[sp], v7\n\t"
             "v_add_u32_e32 %[sp], %[stride], %[sp]\n\t"
             "s_mov_b64 exec, %[m4]\n\t"
             "v_mov_b32_e32 %[i], v6\n\t"
             "3:\n\t"
             "s_cmp_eq_u64 %[m2], 0\n\t"
             "s_cbranch_scc1 4f\n\t"
             "s_and_b64 exec, %[saved], %[m2]\n\t"
             "v_cmp_eq_u32_e64 %[m3], 0, %[cnt]\n\t"
             "v_cmp_eq_u32_e32 vcc, 1, %[cnt]\n\t"
             "v_mov_b32_e32 %[f2], %[i]\n\t"
             "v_add_u32_e32 %[cnt], 1, %[cnt]\n\t"
             "v_cndmask_b32_e64 %[f0], %[f0], %[i], %[m3]\n\t"
             "v_cndmask_b32_e32 %[f1], %[f1], %[i], vcc\n\t"
             "v_cmp_gt_i32_e32 vcc, 0, v23\n\t"                     // LEAF_MORE: the sibling leaf is the next entry
             "v_add_u32_e32 v2, 1, %[i]\n\t"
             "s_andn2_b64 %[m3], exec, vcc\n\t"
             "s_or_b64 %[mpop], %[mpop], %[m3]\n\t"
             "v_cndmask_b32_e32 %[i], %[i], v2, vcc\n\t"
             "4:\n\t"
             "s_cmp_eq_u64 %[mpop], 0\n\t"
             "s_cbranch_scc1 6f\n\t"
             "s_mov_b64 exec, %[mpop]\n\t"
             "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n\t"
             "ds_read_b32 %[i], %[sp]\n\t"
             "s_waitcnt lgkmcnt(0)\n\t"
             "6:\n\t"
             // fetch: an entry (two loads), a record from memory (four), or a record from the table in LDS
             "s_and_b64 exec, %[saved], %[mall]\n\t"
             "v_cmp_le_u32_e32 vcc, 0x40000000, %[i]\n\t"
             "s_and_b64 %[m3], vcc, exec\n\t"                       // records
             "s_andn2_b64 exec, exec, vcc\n\t"                      // entries
             "s_cbranch_execz 7f\n\t"
             "v_lshlrev_b32_e32 v2, 5, %[i]\n\t"
             "global_load_dwordx4 v[16:19], v2, %[base]\n\t"
             "global_load_dwordx4 v[20:23], v2, %[base] offset:16\n\t"
             "7:\n\t"
             "s_mov_b64 exec, %[m3]\n\t"
             "s_cbranch_execz 9f\n\t"
             "v_and_b32_e32 v3, 0x4000000, %[i]\n\t"
             "v_and_b32_e32 v2, 0x3ffffff, %[i]\n\t"
             "v_cmp_ne_u32_e32 vcc, 0, v3\n\t"
             "s_and_b64 %[m4], vcc, exec\n\t"                       // records in LDS
             "s_andn2_b64 exec, exec, vcc\n\t"                      // records in memory
             "s_cbranch_execz 8f\n\t"
             "v_lshlrev_b32_e32 v3, 6, v2\n\t"
             "global_load_dwordx4 v[16:19], v3, %[pbase]\n\t"
             "global_load_dwordx4 v[20:23], v3, %[pbase] offset:16\n\t"
             "global_load_dwordx4 v[24:27], v3, %[pbase] offset:32\n\t"
             "global_load_dwordx4 v[28:31], v3, %[pbase] offset:48\n\t"
             "8:\n\t"
             "s_mov_b64 exec, %[m4]\n\t"
             "s_cbranch_execz 9f\n\t"
             "v_lshl_add_u32 v2, v2, 6, %[ldsbase]\n\t"
             "ds_read_b128 v[16:19], v2\n\t"
             "ds_read_b128 v[20:23], v2 offset:16\n\t"
             "ds_read_b128 v[24:27], v2 offset:32\n\t"
             "ds_read_b128 v[28:31], v2 offset:48\n\t"
             "9:\n\t"
             "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
             "s_mov_b64 exec, %[saved]\n\t"
             "s_bcnt1_i32_b64 %[cntS], %[m]\n\t"
             "s_add_u32 %[nbox], %[nbox], %[cntS]\n\t"
             "s_add_u32 %[nbox], %[nbox], %[cntS]\n\t"
             "s_cmp_ge_u32 %[cntS], %[quorum]\n\t"
             "s_cbranch_scc1 1b\n\t"
             "2:\n\t"
             : [e01] "+{v[16:17]}"(w.e01), [e23] "+{v[18:19]}"(w.e23), [e45] "+{v[20:21]}"(w.e45), [g01] "+{v[22:23]}"(w.g01), [g23] "+{v[24:25]}"(w.g23),
               [g45] "+{v[26:27]}"(w.g45), [lk] "+{v[28:29]}"(w.lk), [rh] "+{v[30:31]}"(w.rh), [i] "+v"(w.i), [f0] "+v"(w.f0),
               [f1] "+v"(w.f1), [f2] "+v"(w.f2), [cnt] "+v"(w.cnt), [m] "=&s"(m), [m2] "=&s"(m2), [m3] "=&s"(m3),
               [mall] "=&s"(mall), [saved] "=&s"(saved), [cntS] "=&s"(n_pair), [m4] "=&s"(m4), [nbox] "+s"(nb),
               [mhit] "=&s"(mhit), [mpop] "=&s"(mpop), [ma] "=&s"(ma), [mb] "=&s"(mb), [sp] "+v"(w.sp) RT_BOX_COUNT_OPERANDS
             : [p0] "v"(w.p0), [p1] "v"(w.p1), [p2] "v"(w.p2), [p3] "v"(w.p3), [p4] "v"(w.p4), [px] "s"(px), [py] "s"(py),
               [pz] "s"(pz), [base] "s"(base), [pbase] "s"(pbase), [usable] "s"(usable), [quorum] "s"(quorum), [ck] "v"(w.cull_k), [cbase] "v"(w.cull_base),
               [ldsbase] "s"(ldsbase), [stride] "n"(BLOCK * 4)
             : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "vcc", "scc", "memory");
    n_box += nb;
    RT_PROF(prof.box_iters += box_iters; prof.box_lanes += box_lanes;)
}
// What the lane stands on, for the code around the loop: a pair record reads as an inner entry (kind 0), an entry as itself
__device__ __forceinline__ void pairs_sync_kind(Walker& w) {
    const bool on_record = ((uint32_t)w.i & rt::LINK_PAIR) != 0u;
    w.sk = on_record ? 0u : __float_as_uint(w.g01.x);
    w.rho = on_record ? 0.0f : w.g01.y;
}
__device__ __forceinline__ void walker_load_pairs(const Store& st, Walker& w) {
    const uint32_t i = (uint32_t)w.i;
    if (i & rt::LINK_PAIR) {
        const uint32_t* r = (i & rt::LINK_LDS) ? st.lds0 + ((st.top_addr - st.lds0_addr) >> 2) + rt::PAIR_REC_DWORDS * (i & rt::LINK_INDEX)
                                               : reinterpret_cast<const uint32_t*>(st.pair) + rt::PAIR_REC_DWORDS * (size_t)(i & rt::LINK_INDEX);
        auto pr = [&](int k) { return f2{__uint_as_float(r[k]), __uint_as_float(r[k + 1])}; };
        w.e01 = pr(0); w.e23 = pr(2); w.e45 = pr(4); w.g01 = pr(6); w.g23 = pr(8); w.g45 = pr(10); w.lk = pr(12); w.rh = pr(14);
    } else {
        // (every field is assigned on both sides: a field assigned on one side only makes the compiler keep it in memory)
        const f2* e = reinterpret_cast<const f2*>(st.ent) + 4 * (size_t)i;
        w.e01 = e[0]; w.e23 = e[1]; w.e45 = e[2]; w.g01 = e[3]; w.g23 = f2{0, 0}; w.g45 = f2{0, 0}; w.lk = f2{0, 0}; w.rh = f2{0, 0};
    }
    pairs_sync_kind(w);
}
template <bool CULL, int BLOCK>
__device__ __forceinline__ void walker_start_pairs(const Scene& sc, const Store& st, Walker& w, const Ray ray, const Rng rng) {
    w.rng = rng;
    walker_set_ray(w, ray, sc.fast_div_ok != 0);
    w.usable = w.usable && fmaxf(fabsf(ray.o.x), fmaxf(fabsf(ray.o.y), fabsf(ray.o.z))) <= sc.entry_guard;   // (entry boxes: see Scene)
    cull_start<CULL>(sc, w, ray);
    w.i = (int)st.root_link; w.resume = 0; w.inst = -1; w.stale = sc.stale_entry; w.alpha_off = -1; w.alpha_wh = 0;
    w.f0 = 0; w.f1 = 0; w.f2 = 0; w.cnt = 0; w.reach = 0.0f;
    w.c.dist = FLT_MAX; w.c.best = -1; w.c.u = 0; w.c.v = 0; w.c.hit = false;
    w.sp = st.stack_addr + 4u * ((uint32_t)BLOCK + threadIdx.x);     // slot 1: slot 0 holds the end of the top-level list
    walker_load_pairs(st, w);
}

// A lane's queue position as the kernels that read the ray again keep it: the top bit carries Walker::usable of the ray
// as given (queues hold fewer than 2^31 positions)
constexpr uint32_t IDX_MASK = 0x7FFFFFFFu, IDX_USABLE = 0x80000000u;
// The lane's number within its wave, computed where it is asked for (two instructions) instead of being held in a
// register through the kernel; volatile so that the compiler does not compute it once and keep it.
__device__ __forceinline__ uint32_t lane_now() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// The oldest noted leaf of the lane: Sphere / AxisAlignedCube / Triangle ::Intersect and the closest-hit bookkeeping of
// BVH_Node::Intersect (Object.cpp:245-251) and World::GetRayColor (World.cpp:110-118).  A triangle the ray reaches stays
// at the head of the FIFO with its distance in `reach` when the rest of its test can matter — the distance would win, or
// the leaf is the one whose assignment of HitInfo.distance survives a rejection — and is finished by tri_finish_step.
// uv_out: where a winning triangle's barycentrics go instead of the lane's registers (the second half of the ray's hit
// record; the last triangle to win is the last to write, and no other kind of hit reads them).
// Leaves need not be met in the reference's order (the ordered traversal, a regrouped layout): of two leaves at the same
// distance the one the reference meets first wins (Object.cpp:248 keeps the earlier hit; sc.ranked: the leaves carry
// their ranks), and a rejected first leaf's distance is recorded only while the ray has no hit — met first, as in the
// reference, that is always so; met later, a hit stands.  RANKED = false: kernels for scenes with fog, whose leaves are
// always met in the reference's order (and whose fog entries have no room for a rank).
template <bool CULL, bool RANKED = true>
__device__ __forceinline__ void closest_update(const Scene& sc, Walker& w, int e, bool ok, bool wrote, float t, float u, float v, float2* uv_out = nullptr) {
    // selects rather than branches around the updates: the record stays in registers
    bool better_ = ok && (!w.c.hit || t < w.c.dist);
    if constexpr (RANKED) { if (ok && w.c.hit && t == w.c.dist && sc.ranked) better_ = rank_before(sc, e, w.c.best); }   // a rare tie: settled with two loads
    const bool better = better_;
    if (CULL && better) cull_update<CULL>(sc, w, t);
    const bool stale = !ok && wrote && e == w.stale && !w.c.hit;
    w.c.dist = (better || stale) ? t : w.c.dist;
    w.c.best = better ? e : w.c.best;
    if (uv_out) { if (better) *uv_out = make_float2(u, v); }
    else { w.c.u = better ? u : w.c.u; w.c.v = better ? v : w.c.v; }
    w.c.hit = w.c.hit || better;
}
template <bool FOG, bool CULL, bool ORD = false>
__device__ __forceinline__ void prim_step(const Scene& sc, const Store& st, Walker& w) {
    const int e = (int)w.f0;
    const f2* E = reinterpret_cast<const f2*>(st.ent) + 4 * (size_t)e;
    const uint32_t sk = reinterpret_cast<const uint32_t*>(E)[6];
    const int kind = sk_kind(sk);
    const Ray cur = walker_ray(w);
    float t = 0;
    bool ok, wrote = false, wait = false;
    if (kind == rt::ENT_TRI) {
        wait = tri_reach(st.tris, sk_payload(sk), cur, t) && (!w.c.hit || (FOG ? t < w.c.dist : t <= w.c.dist) || e == w.stale);   // (an equal distance may win by rank)
        ok = false;
    } else {
        const f2 a = E[0], b = E[1], c = E[2];
        if (kind == rt::ENT_SPHERE) ok = sphere_test(mk(a.x, a.y, b.x), b.y, cur, t, wrote);
        else if (FOG && kind == rt::ENT_FOG) ok = fog_test(mk(a.x, a.y, b.x), mk(b.y, c.x, c.y), sc.objects[(size_t)sk_payload(sk) * rt::RT_OBJ_F4].w, cur, w.rng, t);
        else { V3 n; ok = cube_test(mk(a.x, a.y, b.x), mk(b.y, c.x, c.y), cur, t, n); }   // kernels without fog support only see scenes without fog
    }
    w.reach = wait ? t : 0.0f;
    w.f0 = wait ? w.f0 : w.f1; w.f1 = wait ? w.f1 : w.f2; w.cnt -= wait ? 0u : 1u;
    closest_update<CULL>(sc, w, e, ok, wrote, t, 0.0f, 0.0f);
}
// The second part of Triangle::Intersect for a lane whose head-of-FIFO triangle is waiting for it.
// LEAN: the alpha mask's place and size are read from the instance record instead of being carried by every lane, and
// the barycentrics of a winning triangle go straight to the ray's hit record, hit_rec[idx].
template <bool CULL, bool LEAN = false, bool ORD = false>
__device__ __forceinline__ void tri_finish_step(const Scene& sc, const Store& st, Walker& w, float4* hit_rec = nullptr, const uint32_t* idx_slot = nullptr) {
    const int e = (int)w.f0;
    w.f0 = w.f1; w.f1 = w.f2; w.cnt--;
    const uint32_t sk = reinterpret_cast<const uint32_t*>(st.ent)[8 * (size_t)e + 6];
    float t = 0, u = 0, v = 0;
    bool wrote = false;
    int alpha_off = w.alpha_off;
    uint32_t alpha_wh = w.alpha_wh;
    if (LEAN) {
        const float4 i3 = st.inst[(size_t)w.inst * rt::RT_INST_F4 + 3];
        alpha_off = __float_as_int(i3.z); alpha_wh = __float_as_uint(i3.w);
    }
    const bool ok = tri_surface(sc, st.tris, sk_payload(sk), alpha_off, alpha_wh, walker_ray(w), w.reach, t, u, v, wrote);
    w.reach = 0.0f;
    closest_update<CULL>(sc, w, e, ok, wrote, t, u, v, LEAN ? reinterpret_cast<float2*>(hit_rec + (*idx_slot & IDX_MASK)) + 1 : nullptr);
}
// Entries that change the lane's state beyond its position; they wait until every noted leaf has been tested.  Returns
// true when the ray has finished (closest hit known, plane list merged); the lane is idle from then on.
// RELOAD: the ray as given is not kept in registers while the lane is inside an instance (ten registers: what decides
// whether a fifth wave fits a SIMD); on the way out it is read again from the queue record it came from (outer_a,
// outer_b: the two halves of record `outer_idx`) and its reciprocals are computed again, bit for bit as at the start.
// LEAN: nor are the entry to go on at after the instance and the alpha mask of its texture (three more registers): they
// are read from the instance record where they are needed.
// SAVE: the ray as given has no queue record (it came from the camera): on the way in it is written to the 32 bytes at
// outer_a + 2 * outer_idx (one memory sector; outer_b is not used), and read from there on the way out.
// With RELOAD the top bit of outer_tagged says whether the ray as given is admitted to the reciprocal path (IDX_USABLE):
// that was worked out when the ray was taken up and is not worked out again on the way out (0.25 ms of a 1080p, 64 spp
// frame; installing the new ray once for the lanes going in and the lanes going out changed nothing).
// ORD > 0: the ordered traversal with stacks of ORD-thread blocks: where the other variants follow a skip link or go
// on behind an instance's entry, this one pops; what it leaves for later, it pushes.
// Returns WALK_ON, WALK_DONE (the ray has finished) or WALK_HANDED_OVER: with a regrouped layout (sc.has_ref) a ray that
// enters a mesh instance and has, in the instance's frame, a zero direction component (ray_monotone) cannot be walked
// here; the lane goes idle and the caller hands the ray over to the walk through the reference's layout.
enum : int { WALK_ON = 0, WALK_DONE = 1, WALK_HANDED_OVER = 2 };
// PAIRS (with ORD): pair steps; a lane on a pair record comes here when its ray is outside the reciprocal path, and an
// inner entry reached as an entry (a root that has no record, a box with min > max, an instance's entry box) always.
template <bool RELOAD = false, bool LEAN = false, bool SAVE = false, int ORD = 0, bool PAIRS = false>
__device__ __forceinline__ int walker_special(const Scene& sc, const Store& st, Walker& w, float4* outer_a = nullptr, float4* outer_b = nullptr, uint32_t outer_tagged = 0) {
    const uint32_t outer_idx = outer_tagged & IDX_MASK;
    int done = WALK_ON;
    const int kind = sk_kind(w.sk);
    if constexpr (PAIRS) {
        if ((uint32_t)w.i & rt::LINK_PAIR) {
            // both children as the reference tests them; any order of the two gives the reference's result
            const Ray r = walker_ray(w);
            const bool ha = box_hit(mk(w.e01.x, w.e23.x, w.e45.x), mk(w.e01.y, w.e23.y, w.e45.y), r);
            const bool hb = box_hit(mk(w.g01.x, w.g23.x, w.g45.x), mk(w.g01.y, w.g23.y, w.g45.y), r);
            const uint32_t la = __float_as_uint(w.lk.x), lb = __float_as_uint(w.lk.y);
            if (ha && hb) ord_push(st, w, lb, ORD * 4);
            w.i = ha ? (int)la : (hb ? (int)lb : (int)ord_pop(st, w, ORD * 4));
            walker_load_pairs(st, w);
            return WALK_ON;
        }
    }
    if (kind == rt::ENT_END) {
        if (w.inst >= 0) {   // PolygonMesh and its wrappers return to the top level
            if constexpr (ORD != 0) w.i = (int)ord_pop(st, w, ORD * 4);
            else w.i = LEAN ? __float_as_int(st.inst[(size_t)w.inst * rt::RT_INST_F4 + 2].w) : w.resume;
            w.inst = -1; w.stale = sc.stale_entry;
            if (RELOAD) {
                const float4 a = SAVE ? outer_a[2 * (size_t)outer_idx] : outer_a[outer_idx], b = SAVE ? outer_a[2 * (size_t)outer_idx + 1] : outer_b[outer_idx];
                Ray r;
                r.o = mk(a.x, a.y, a.z); r.d = mk(a.w, b.x, b.y);
                walker_set_ray(w, r, sc.fast_div_ok != 0, (int)(outer_tagged >> 31));
            } else { w.p0 = w.s0; w.p1 = w.s1; w.p2 = w.s2; w.p3 = w.s3; w.p4 = w.s4; w.usable = w.s_usable; }
        } else {
            const Ray top = walker_ray(w);
            for (int p = 0; p < sc.n_planes; p++) {   // YPlane::Intersect (Object.cpp:139-149), merged as World.cpp:110-118
                const float t = -(top.o.y - st.planes[p].x) / top.d.y;
                if (t < 0.001f) continue;             // "t < 1e-3" against a double: the same floats pass
                if (t < w.c.dist) { w.c.hit = true; w.c.dist = t; w.c.best = -(p + 2); }
            }
            w.i = sc.idle_entry;
            done = WALK_DONE;
        }
    } else if (kind == rt::ENT_INSTANCE) {
        const int ii = sk_payload(w.sk);
        const float4* I = st.inst + (size_t)ii * rt::RT_INST_F4;
        const float4 i0 = I[0], i1 = I[1], i2 = I[2], i3 = I[3];
        if (!RELOAD) { w.s0 = w.p0; w.s1 = w.p1; w.s2 = w.p2; w.s3 = w.p3; w.s4 = w.p4; w.s_usable = w.usable; }
        if (RELOAD && SAVE) {
            const Ray r = walker_ray(w);
            outer_a[2 * (size_t)outer_idx] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
            *reinterpret_cast<float2*>(outer_a + 2 * (size_t)outer_idx + 1) = make_float2(r.d.y, r.d.z);
        }
        walker_set_ray(w, to_instance(i0, i1, i2, walker_ray(w)), sc.fast_div_ok != 0);
        if constexpr (ORD != 0) ord_push(st, w, (uint32_t)st.inst_end[ii], ORD * 4);   // the way back out: the END entry of the instance's range
        else if (!LEAN) w.resume = w.i + 1;
        w.stale = (w.i == sc.stale_entry) ? __float_as_int(i3.y) : -1;
        w.inst = ii;
        if (!LEAN) { w.alpha_off = __float_as_int(i3.z); w.alpha_wh = __float_as_uint(i3.w); }   // alpha cut-out of the mesh's texture, or -1
        w.i = __float_as_int(i2.z);
        if constexpr (PAIRS) w.i = (int)st.inst_start[ii];   // (the mesh's root: its pair record if it has one)
        if (sc.has_ref && !w.usable && !ray_monotone(walker_ray(w))) {   // not a ray for this layout: hand it over, as it was given
            w.i = sc.idle_entry; w.inst = -1; w.cnt = 0; w.reach = 0.0f;
            done = WALK_HANDED_OVER;
        }
    } else {
        // an inner entry tested as the reference writes it, with true divisions: a box with min > max on some axis, or
        // a ray that is not admitted to the reciprocal path
        // (a mesh instance's entry box is proven for rays of the reciprocal path only: here it counts as hit)
        // (dword 7 of the entry: the ordered traversal holds it — its entry may be a record of the table in LDS —, the others
        // read it here, the traversal loop of a scene in LDS does not load it)
        uint32_t dword7;
        if constexpr (ORD != 0) dword7 = __float_as_uint(w.rho);
        else dword7 = kind == rt::ENT_INNER ? reinterpret_cast<const uint32_t*>(st.ent)[8 * (size_t)w.i + 7] : 0u;
        const bool entry_box = kind == rt::ENT_INNER && (dword7 & ~3u) == rt::ENTRY_BOX_MARK;
        if constexpr (ORD != 0) {
            // (no preference between the children here: any order gives the reference's result)
            const bool hit = entry_box || box_hit(mk(w.e01.x, w.e23.x, w.e45.x), mk(w.e01.y, w.e23.y, w.e45.y), walker_ray(w));
            const uint32_t second = kind == rt::ENT_INNER ? w.sk : __float_as_uint(w.rho);
            const uint32_t first = ((uint32_t)w.i & rt::LINK_LDS) ? w.eleft : (uint32_t)w.i + 1u;
            if (PAIRS && kind == rt::ENT_INNER && (second & rt::ENTRY_HAS_PAIR))   // its record takes over
                w.i = hit ? (int)(rt::LINK_PAIR | (second & (rt::LINK_LDS | rt::LINK_INDEX))) : (int)ord_pop(st, w, ORD * 4);
            else {
                if (hit && second != rt::LINK_NONE) ord_push(st, w, second, ORD * 4);
                w.i = hit ? (int)first : (int)ord_pop(st, w, ORD * 4);
            }
        } else {
            const int skip = kind == rt::ENT_INNER ? (int)w.sk : __float_as_int(reinterpret_cast<const float*>(st.ent)[8 * (size_t)w.i + 7]);
            w.i = (entry_box || box_hit(mk(w.e01.x, w.e23.x, w.e45.x), mk(w.e01.y, w.e23.y, w.e45.y), walker_ray(w))) ? w.i + 1 : skip;
        }
    }
    if constexpr (PAIRS) walker_load_pairs(st, w); else if constexpr (ORD != 0) walker_load_ord(st, w); else walker_load(st, w);
    return done;
}
// One round for a wave: the traversal loop; then primitive tests, repeated while at least `prim_quorum` lanes have a
// noted leaf; then the state-changing entries of lanes that have none left.  Returns, per lane, what walker_special
// returned: WALK_DONE in lanes whose ray finished, WALK_HANDED_OVER in lanes whose ray has to be walked elsewhere.
template <bool LDS, bool FOG, bool RELOAD = false, bool SAVE = false, int ORD = 0, bool PAIRS = false>
__device__ __forceinline__ int walk_round(const Scene& sc, const Store& st, Walker& w, int quorum, int prim_quorum, int finish_quorum, Prof& prof, uint32_t& n_box,
                                           float4* outer_a = nullptr, float4* outer_b = nullptr, uint32_t outer_idx = 0, float4* hit_rec = nullptr, const uint32_t* idx_slots = nullptr) {
    RT_PROF(unsigned long long t0 = __builtin_readcyclecounter(); prof.rounds++;)
    // with few rays in the wave (the last rays of a pass) a fixed quorum is never met and every round would advance each
    // ray by a single step: ask for half of the lanes that hold a ray instead
    const int busy = __popcll(__ballot(sk_kind(w.sk) != rt::ENT_IDLE));
    quorum = max(1, min(quorum, (busy + 1) >> 1));
    prim_quorum = max(1, min(prim_quorum, (busy + 1) >> 1));
    finish_quorum = max(1, min(finish_quorum, (busy + 1) >> 1));
    constexpr bool CULL = !LDS && !FOG;   // see "distance culling": large scenes, and never when traversal draws random numbers
    if constexpr (PAIRS) { box_steps_pairs<ORD>(st, w, quorum, prof, n_box); pairs_sync_kind(w); }
    else if constexpr (ORD != 0) box_steps_ord<ORD>(st, w, quorum, n_box); else box_steps<LDS>(st, w, quorum, prof, n_box);
    RT_PROF(unsigned long long t1 = __builtin_readcyclecounter(); prof.box += t1 - t0;)
    for (;;) {
        const bool has = w.cnt != 0 && w.reach == 0.0f;
        const int n = __popcll(__ballot(has));
        if (n == 0) break;
        RT_PROF(prof.prims++; prof.prim_lanes += n;)
        if (has) prim_step<FOG, CULL, ORD != 0>(sc, st, w);
        if (n < prim_quorum) break;
    }
    // triangles waiting for the second part of their test: run it once enough lanes have one (a waiting lane goes on
    // walking and noting leaves meanwhile, but tests no further primitive, so the order of the updates is the reference's)
    {
        const bool waiting = w.reach != 0.0f;
        const int n = __popcll(__ballot(waiting));
        if (n >= finish_quorum) {
            RT_PROF(prof.finishes++; prof.finish_lanes += n;)
            if (waiting) tri_finish_step<CULL, RELOAD && LDS, ORD != 0>(sc, st, w, hit_rec, idx_slots ? idx_slots + lane_now() : nullptr);
        }
    }
    RT_PROF(unsigned long long t2 = __builtin_readcyclecounter(); prof.prim += t2 - t1;)
    const int kind = sk_kind(w.sk);
    const bool as_entry = PAIRS && !((uint32_t)w.i & rt::LINK_PAIR);   // pair steps: the loop does not test an inner entry reached as an entry
    const bool special = w.cnt == 0 && (kind == rt::ENT_END || kind == rt::ENT_INSTANCE || kind == rt::ENT_INNER_INV ||
                                        (kind == rt::ENT_INNER && (!w.usable || as_entry)));
    int done = WALK_ON;
    RT_PROF(if (__any(special)) prof.specials++;)
#ifdef RT_PROFILE
    {
        const bool k0 = special && kind == rt::ENT_END && w.inst < 0;
        const bool k1 = special && kind == rt::ENT_END && w.inst >= 0;
        const bool k2 = special && kind == rt::ENT_INSTANCE;
        const bool k3 = special && !k0 && !k1 && !k2;
        const int n0 = __popcll(__ballot(k0));
        const int n1 = __popcll(__ballot(k1));
        const int n2 = __popcll(__ballot(k2));
        const int n3 = __popcll(__ballot(k3));
        prof.sp_exec[0] += n0 != 0; prof.sp_exec[1] += n1 != 0; prof.sp_exec[2] += n2 != 0; prof.sp_exec[3] += n3 != 0;
        prof.sp_lanes[0] += n0; prof.sp_lanes[1] += n1; prof.sp_lanes[2] += n2; prof.sp_lanes[3] += n3;
    }
#endif
    n_box += (uint32_t)__popcll(__ballot(special && (kind == rt::ENT_INNER_INV || kind == rt::ENT_INNER)));   // boxes tested as written
    if (PAIRS) n_box += (uint32_t)__popcll(__ballot(special && kind == rt::ENT_INNER && !as_entry));   // (a record: two boxes)
    if (special) done = walker_special<RELOAD, RELOAD && LDS, SAVE, ORD, PAIRS>(sc, st, w, outer_a, outer_b, idx_slots ? idx_slots[lane_now()] : outer_idx);
    RT_PROF(prof.special += __builtin_readcyclecounter() - t2;)
    return done;
}
// Run the lanes of a wave until all of them have finished.
// Returns true in lanes whose ray was handed over (see walker_special) instead of finished.
template <bool FOG>
__device__ __forceinline__ bool walk_wave_to_end(const Scene& sc, const Store& st, Walker& w) {
    Prof prof;
    uint32_t n_box = 0;
    bool handed_over = false;
    while (__any(sk_kind(w.sk) != rt::ENT_IDLE)) handed_over = (walk_round<false, FOG>(sc, st, w, 24, 32, 16, prof, n_box) == WALK_HANDED_OVER) || handed_over;
    return handed_over;
}
__device__ __forceinline__ bool walker_is_idle(const Walker& w) { return sk_kind(w.sk) == rt::ENT_IDLE; }

struct Hit { V3 point, normal; float u, v; int object; bool sphere; };   // sphere: u, v are computed when a texture asks (sphere_uv)

// The remaining HitInfo fields of the winning primitive (Object.cpp:21-25, 119-130, 145-147, 534-547 and the wrappers' way out)
__device__ inline void finalize_hit(const Scene& sc, const Ray& ray, const Closest& c, Hit& h) {
    h.u = 0; h.v = 0; h.sphere = false;
    if (c.best <= -2) {
        const float4 pl = sc.planes[-(c.best + 2)];
        h.point = ray.o + ray.d * c.dist;
        h.normal = mk(0.0f, 1.0f, 0.0f);
        h.object = __float_as_int(pl.y);
        return;
    }
    const float4 ea = sc.ent[2 * c.best], eb = sc.ent[2 * c.best + 1];
    const uint32_t sk = __float_as_uint(eb.z);
    const int kind = sk_kind(sk);
    if (kind == rt::ENT_SPHERE) {
        h.point = ray.o + ray.d * c.dist;
        h.normal = normalize(h.point - xyz(ea));
        h.object = sk_payload(sk);
        h.sphere = true;
    } else if (kind == rt::ENT_CUBE) {
        float t; V3 n;
        const V3 mn = xyz(ea), mx = mk(ea.w, eb.x, eb.y);
        cube_test(mn, mx, ray, t, n);
        h.normal = n;
        h.object = sk_payload(sk);
        h.point = ray.o + ray.d * t;
        h.point = h.point + n * 0.02f;
        const V3 m01 = mk((h.point.x - mn.x) / (mx.x - mn.x), (h.point.y - mn.y) / (mx.y - mn.y), (h.point.z - mn.z) / (mx.z - mn.z));
        if (n.x == 1 || n.x == -1) { h.u = m01.z; h.v = m01.y; }
        else if (n.y == 1 || n.y == -1) { h.u = m01.x; h.v = m01.z; }
        else { h.u = m01.x; h.v = m01.y; }
    } else if (kind == rt::ENT_FOG) {   // Object.cpp:585-591, in the frame the volume was tested in
        const int ii = __float_as_int(eb.w);
        Ray q = ray;
        float4 i0, i1, i2;
        if (ii >= 0) {
            const float4* I = sc.inst + (size_t)ii * rt::RT_INST_F4;
            i0 = I[0]; i1 = I[1]; i2 = I[2];
            q = to_instance(i0, i1, i2, ray);
        }
        h.normal = mk(0.0f, 1.0f, 0.0f);
        h.object = sk_payload(sk);
        h.point = q.o + (q.d * c.dist) * 1.001f;
        if (ii >= 0) from_instance(i0, i1, i2, h.point, h.normal);
    } else {   // triangle inside a mesh instance
        const float4* I = sc.inst + (size_t)__float_as_int(eb.w) * rt::RT_INST_F4;
        const float4 i0 = I[0], i1 = I[1], i2 = I[2], i3 = I[3];
        const Ray q = to_instance(i0, i1, i2, ray);
        const float4* T = sc.tris + (size_t)sk_payload(sk) * rt::RT_TRI_F4;
        const V3 n = mk(T[4].z, T[4].w, T[5].x);
        const V3 hp = q.o + q.d * c.dist;
        h.point = hp + n * 0.01f;
        h.normal = n;
        from_instance(i0, i1, i2, h.point, h.normal);
        h.object = __float_as_int(i3.x);   // PolygonMesh::Intersect reports the mesh, not the triangle (Object.cpp:490-495)
        h.u = c.u; h.v = c.v;
    }
}

// ---------------------------------------------------------------------------------------------------- shading
// byte / 255.0f (World.cpp:192, DataUtility.cpp:120): exactly the division, by two residual corrections of byte * RN(1/255)
__device__ __forceinline__ V3 texel_color(const uint8_t* p) {
    const float y = 1.0f / 255.0f;
    return mk(div_by((float)p[0], 255.0f, y), div_by((float)p[1], 255.0f, y), div_by((float)p[2], 255.0f, y));
}
// World::GetSkyboxPixel (World.cpp:173-193): equirectangular nearest lookup without clamping
__device__ inline V3 skybox_pixel(const Scene& sc, V3 d) {
    const float u = pm_atan2f(d.x, d.z) / (2 * 3.14159265358979323846264338327950288f) + 0.5f;
    const float v = d.y * 0.5f + 0.5f;
    const int ix = (int)(u * (float)sc.sky_w), iy = (int)(v * (float)sc.sky_h);
    const int index = 3 * (int)texel_in_table((uint32_t)iy * sc.sky_w + (uint32_t)ix, sc.sky_w, sc.sky_h);
    return texel_color(sc.texels + sc.sky_off + (uint32_t)index);
}
// The sign of pm_sinf(x) without its polynomial: 1 negative, 0 positive, -1 when this shortcut cannot tell (the value is
// or may round to zero, or x is not finite).  With r the reduced argument, |r| <= pi/4 plus rounding, and q the quadrant,
// pm_sinf returns +-(r + r*(z*p)) with |z*p| < 0.2 for even q — the sign of r — and +-(1 - (z/2 - z*p)) >= 0.6 for odd q.
__device__ __forceinline__ int pm_sinf_sign(float x) {
    if (!(fabsf(x) <= FLT_MAX)) return -1;
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079631090164184570e+00, P2 = 1.58932547122958567343e-08, P3 = 6.12323399573676603587e-17;
    const double xd = (double)x;
    const double k = rint(xd * TWO_OVER_PI);
    const double r = ((xd - k * P1) - k * P2) - k * P3;
    const int q = (int)(k - 4.0 * floor(k * 0.25));
    if (!(q & 1) && !(fabs(r) > 1.0e-12)) return -1;    // sin(r) of a tiny r: its float may be 0 or underflow the product
    const int neg = (q & 1) ? 0 : (r < 0.0 ? 1 : 0);
    return neg ^ ((q >> 1) & 1);
}
// Sphere::Intersect's texture coordinates (Object.cpp:24-25), for the textures that read them
__device__ __forceinline__ void sphere_uv(Hit& h) {
    h.u = pm_atan2f(h.normal.x, h.normal.z) / (2 * 3.14159265358979323846264338327950288f) + 0.5f;
    h.v = h.normal.y * 0.5f + 0.5f;
}
// Texture::GetColorValue of the three texture classes (DataUtility.h:91, DataUtility.cpp:84-96, 113-121) — in two steps.
// The recursion of World::GetRayColor multiplies a bounce's texture colour in on the way BACK (World.cpp:138,151,166), so
// the colour is needed when the path ends, not when the bounce happens.  A bounce therefore only records WHERE its colour
// is — four bytes — and the colour is looked up when the path unwinds:
//   bits 31-30: 0 = the first colour of object `payload` (a plain colour, or the checker's colour for a negative product
//               of sines), 1 = the object's second colour, 2 = the texel at byte 4 * payload of the texel pool
//   bit 29:     the bounce halves what comes back (lightDarknerFactor of a diffuse surface, World.cpp:129)
//   bits 28-0:  payload (object ids stay below 2^28, the texel pool below 2^31 bytes: rt_engine_set_scene checks)
constexpr uint32_t REF_SECOND = 1u << 30, REF_TEXEL = 2u << 30, REF_HALF = 1u << 29, REF_PAYLOAD = (1u << 29) - 1u;
__device__ inline uint32_t texture_ref(const Scene& sc, const float4* ob, int object, Hit& h) {
    const int kind = __float_as_int(ob[0].y);
    if (kind == 0) return (uint32_t)object;
    if (kind == 1) {
        // the checker only asks whether the product of three sines is negative: with every factor well away from zero
        // that is the parity of their signs; otherwise the product itself is formed
        const float scale = 5.0f;
        const float ax = scale * h.point.x, ay = scale * h.point.y, az = scale * h.point.z;
        const int sx = pm_sinf_sign(ax), sy = pm_sinf_sign(ay), sz = pm_sinf_sign(az);
        bool negative;
        if ((sx | sy | sz) >= 0) negative = ((sx ^ sy ^ sz) & 1) != 0;
        else negative = pm_sinf(ax) * pm_sinf(ay) * pm_sinf(az) < 0.0f;
        return negative ? (uint32_t)object : (REF_SECOND | (uint32_t)object);
    }
    if (h.sphere) sphere_uv(h);
    const rt::ImageRec im = sc.images[__float_as_int(ob[0].z)];
    return REF_TEXEL | ((im.offset + (uint32_t)image_index(im, h.u, h.v)) >> 2);   // (image offsets are multiples of 16)
}
__device__ inline V3 ref_color(const Scene& sc, uint32_t ref) {
    if (ref & REF_TEXEL) return texel_color(sc.texels + 4u * (size_t)(ref & REF_PAYLOAD));
    const float4* ob = sc.objects + (size_t)(ref & REF_PAYLOAD) * rt::RT_OBJ_F4;
    return xyz((ref & REF_SECOND) ? ob[2] : ob[1]);
}

}  // namespace rtd
