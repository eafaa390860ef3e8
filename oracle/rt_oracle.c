/* TEST INFRASTRUCTURE — see rt_oracle.h.  Plain-C restatement of the reference's per-pixel path.
 * Every function cites the reference lines it follows (paths relative to /root/reference/Raytracer/src).
 * Compile with -ffp-contract=off: every a*b+c below is two roundings, as in the reference's x86-64 build. */
#include "rt_oracle.h"
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { v3 o, d; } Ray;
typedef struct { v3 mn, mx; } Box;
/* HitInfo (DataUtility.h:52-58); object = index into scene->objs, -1 = nullptr */
typedef struct { v3 point, normal; int object; float distance; v2 uv; } Hit;

enum { K_SPHERE, K_CUBE, K_PLANE, K_TRI, K_NODE, K_MESH, K_ROT, K_MOVE, K_FOG };
enum { C_RAYS, C_AABB, C_SPH_TEST, C_SPH_HIT, C_CUBE_TEST, C_CUBE_HIT, C_PLANE_TEST, C_PLANE_HIT, C_TRI_TEST, C_TRI_UV,
       C_INST_ENTER, C_INST_HIT, C_SAMPLES, C_SHADE, C_RNG, C_FOG_TEST, C_COUNT };

typedef struct {
    int kind, mat, id;
    v3 c; float r;              /* sphere */
    float neg_inv_density;      /* fog (its boundary cube is `box`) */
    Box box;                    /* cube extents | node/tri/rotation box */
    float y;                    /* plane */
    v3 p[3]; v2 tc[3]; v3 n[3]; /* triangle vertices */
    int left, right;            /* node */
    int root;                   /* mesh: its triangle BVH */
    int target, axis; float sin_t, cos_t; v3 pivot, offset;   /* rotation / movement wrappers */
    /* deferred mesh creation */
    float* verts; int n_tri; float size; int rot_axis; float degrees; int has_move;
} Obj;

typedef struct { int kind; v3 c1, c2; uint8_t* img; unsigned w, h; } Tex;
typedef struct { int type, tex; v3 emit; } Mat;

struct orc_scene {
    Obj* objs; int n_objs, cap_objs;
    Tex* texs; int n_texs;
    Mat* mats; int n_mats;
    int* top; int n_top;        /* creation-order list of BVH'd objects */
    int* planes; int n_planes;  /* noBoundingBoxObjects */
    int root;
    uint8_t* sky; unsigned sky_w, sky_h;
    int math_mode;
    int next_id;
    uint64_t counters[C_COUNT];
    uint64_t repeats[C_COUNT];   /* the part of `counters` spent on the second visit of a leaf listed as both children */
    pthread_mutex_t lock;
};

typedef struct { const orc_scene* s; uint32_t key, ctr; uint64_t cnt[C_COUNT]; float sky_brightness; uint64_t rep[C_COUNT]; int in_repeat; } Ctx;

/* ------------------------------------------------------------------------------------------------------------
 * keyed counter RNG standing in for std::rand (DataUtility.h:11-12), RAND_MAX = 2^31-1 as on glibc
 * ---------------------------------------------------------------------------------------------------------- */
static uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
uint32_t orc_rand31(uint32_t key, uint32_t ctr) { return fmix32(key ^ (ctr * 0x9E3779B9u)) >> 1; }
uint32_t orc_sample_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = fmix32(seed ^ 0xA511E9B3u);
    k = fmix32(k + pixel * 0x9E3779B1u + 0x68E31DA4u);
    k = fmix32(k ^ (sample * 0x85EBCA77u + 0xC2B2AE3Du));
    return k;
}
static int ctx_rand(Ctx* c) { c->cnt[C_RNG]++; return (int)orc_rand31(c->key, c->ctr++); }
/* DataUtility.h:11 — (float)rand() / RAND_MAX, RAND_MAX converts to 2^31 */
static float random01(Ctx* c) { return (float)ctx_rand(c) / 2147483648.0f; }
/* DataUtility.h:12 — int subtraction, int->float, divide by (float)(RAND_MAX/2) = 2^30 */
static float random_unit_distance(Ctx* c) { return (float)(ctx_rand(c) - 1073741823) / 1073741824.0f; }
/* DataUtility.h:28-34 */
static v3 unit_sphere_point(Ctx* c) {
    for (;;) {
        v3 s; s.x = random_unit_distance(c); s.y = random_unit_distance(c); s.z = random_unit_distance(c);
        if (s.x * s.x + s.y * s.y + s.z * s.z < 1.0f) return s;
    }
}
/* DataUtility.h:35-41 */
static v2 unit_circle_point(Ctx* c) {
    for (;;) {
        v2 s; s.x = random_unit_distance(c); s.y = random_unit_distance(c);
        if (s.x * s.x + s.y * s.y < 1.0f) return s;
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * portable sinf / atan2f: evaluated in IEEE double with a fixed operation order so that a GPU implementation
 * of the same formulas gives the same bits.  Polynomial coefficients are fdlibm's (k_sin.c, k_cos.c, s_atan.c).
 * ---------------------------------------------------------------------------------------------------------- */
float orc_pm_sinf(float x) {
    if (!(fabsf(x) <= FLT_MAX)) return x - x;   /* inf/NaN -> NaN */
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    /* pi/2 = P1 + P2 + P3: 25 + 25 + 53 significant bits, so k*P1 and k*P2 are exact for |k| < 2^28 */
    const double P1 = 1.57079631090164184570e+00;   /* 0x3FF921FB50000000 */
    const double P2 = 1.58932547122958567343e-08;   /* 0x3E5110B460000000 */
    const double P3 = 6.12323399573676603587e-17;   /* 0x3C91A62633145C07 */
    double xd = (double)x;
    double k = rint(xd * TWO_OVER_PI);
    double r = ((xd - k * P1) - k * P2) - k * P3;
    int q = (int)(k - 4.0 * floor(k * 0.25));
    double z = r * r;
    double res;
    if (q & 1) {
        const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        res = 1.0 - (0.5 * z - z * p);
    } else {
        const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
        double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
        res = r + r * (z * p);
    }
    if (q & 2) res = -res;
    return (float)res;
}
static double pm_atan_pos(double x) {   /* atan for x >= 0 (inf allowed), fdlibm s_atan.c reduction */
    static const double hi[4] = { 4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00 };
    static const double lo[4] = { 2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17 };
    static const double aT[11] = { 3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
        9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
        4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02 };
    int id;
    if (x < 0.4375) id = -1;
    else if (x < 0.6875) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
    else if (x < 1.1875) { id = 1; x = (x - 1.0) / (x + 1.0); }
    else if (x < 2.4375) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
    else { id = 3; x = -1.0 / x; }
    double z = x * x, w = z * z;
    double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    return hi[id] - ((x * (s1 + s2) - lo[id]) - x);
}
float orc_pm_atan2f(float y, float x) {
    if (!(fabsf(x) <= FLT_MAX) || !(fabsf(y) <= FLT_MAX)) return (x - x) + (y - y);   /* inf/NaN -> NaN */
    const double PI = 3.14159265358979311600e+00;
    double ax = fabs((double)x), ay = fabs((double)y);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else a = pm_atan_pos(ay / ax);
    if (signbit(x)) a = PI - a;
    if (signbit(y)) a = -a;
    return (float)a;
}
static float m_sinf(const orc_scene* s, float x) { return s->math_mode == ORC_MATH_PORTABLE ? orc_pm_sinf(x) : sinf(x); }
/* logf for the fog's free path (Object.cpp:579), evaluated in IEEE double with fdlibm's e_log.c reduction and polynomial
 * and rounded to float once; the GPU kernels evaluate the same formulas.  x >= 0 (it is a Random01 draw). */
float orc_pm_logf(float x) {
    if (x != x) return x;
    if (x <= 0.0f) return x == 0.0f ? -INFINITY : (x - x) / 0.0f;
    if (x > FLT_MAX) return x;
    union { double d; uint64_t u; } v;
    v.d = (double)x;                                   /* exact; never subnormal as a double */
    int k = (int)((v.u >> 52) & 0x7FF) - 1023;
    v.u = (v.u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;   /* m in [1, 2) */
    double m = v.d;
    if (m > 1.41421356237309514547) { m *= 0.5; k += 1; }         /* m in (sqrt(2)/2, sqrt(2)] */
    const double f = m - 1.0, s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    return (float)(dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f));
}
static float m_logf(const orc_scene* s, float x) { return s->math_mode == ORC_MATH_PORTABLE ? orc_pm_logf(x) : logf(x); }
static float m_atan2f(const orc_scene* s, float y, float x) { return s->math_mode == ORC_MATH_PORTABLE ? orc_pm_atan2f(y, x) : atan2f(y, x); }

/* ------------------------------------------------------------------------------------------------------------
 * GLM 0.9.9.7 vector arithmetic in the order its templates evaluate it
 * ---------------------------------------------------------------------------------------------------------- */
static v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
static v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static v3 divs3(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }               /* glm/detail/type_vec3.inl:715-721 */
static float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            /* glm/detail/func_geometric.inl:48-55 */
static v3 cross3(v3 x, v3 y) { return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }  /* :68-79 */
static v3 normalize3(v3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return scale3(v, s); }  /* :82-90, func_exponential.inl:134-139 */
static float gmin(float x, float y) { return (y < x) ? y : x; }   /* glm/detail/func_common.inl:20 */
static float gmax(float x, float y) { return (x < y) ? y : x; }   /* :29 */
static float get3(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
static void set3(v3* v, int i, float f) { if (i == 0) v->x = f; else if (i == 1) v->y = f; else v->z = f; }

/* ------------------------------------------------------------------------------------------------------------
 * BoundingBox::DoesRayHit (DataUtility.cpp:35-82)
 * ---------------------------------------------------------------------------------------------------------- */
static int box_hit(const Box* b, const Ray* r) {
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    if (r->d.x >= 0) { tmin = (b->mn.x - r->o.x) / r->d.x; tmax = (b->mx.x - r->o.x) / r->d.x; }
    else { tmin = (b->mx.x - r->o.x) / r->d.x; tmax = (b->mn.x - r->o.x) / r->d.x; }
    if (r->d.y >= 0) { tymin = (b->mn.y - r->o.y) / r->d.y; tymax = (b->mx.y - r->o.y) / r->d.y; }
    else { tymin = (b->mx.y - r->o.y) / r->d.y; tymax = (b->mn.y - r->o.y) / r->d.y; }
    if ((tmin > tymax) || (tymin > tmax)) return 0;
    tmin = gmax(tmin, tymin);
    tmax = gmin(tmax, tymax);
    if (r->d.z >= 0) { tzmin = (b->mn.z - r->o.z) / r->d.z; tzmax = (b->mx.z - r->o.z) / r->d.z; }
    else { tzmin = (b->mx.z - r->o.z) / r->d.z; tzmax = (b->mn.z - r->o.z) / r->d.z; }
    if ((tmin > tzmax) || (tzmin > tmax)) return 0;
    tmin = gmax(tmin, tzmin);
    tmax = gmin(tmax, tzmax);
    if (tmin < 0.0f && tmax < 0.0f) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------------------------
 * textures (DataUtility.h:75-107, DataUtility.cpp:84-127)
 * ---------------------------------------------------------------------------------------------------------- */
static int image_index(const Tex* t, v2 uv) {   /* DataUtility.h:82, size is uvec2 -> unsigned arithmetic */
    unsigned row = (unsigned)(int)(uv.y * (float)t->h);
    unsigned col = (unsigned)(int)(uv.x * (float)t->w);
    return 4 * (int)(row * t->w + col);
}
static v3 tex_color(const orc_scene* s, const Tex* t, v2 uv, v3 p) {
    if (t->kind == ORC_TEX_COLOR) return t->c1;
    if (t->kind == ORC_TEX_CHECKER) {           /* DataUtility.cpp:84-96 */
        const float scale = 5.0f;
        float sines = m_sinf(s, scale * p.x) * m_sinf(s, scale * p.y) * m_sinf(s, scale * p.z);
        return (sines < 0.0f) ? t->c1 : t->c2;
    }
    int idx = image_index(t, uv);               /* DataUtility.cpp:113-121 */
    return divs3(V3((float)t->img[idx], (float)t->img[idx + 1], (float)t->img[idx + 2]), 255.0f);
}
static int tex_solid(const Tex* t, v2 uv) {     /* DataUtility.cpp:123-127; base class: always solid (DataUtility.h:67) */
    if (t->kind != ORC_TEX_IMAGE) return 1;
    uint8_t alpha = t->img[image_index(t, uv) + 3];
    return (float)alpha > 0.01f;
}

/* ------------------------------------------------------------------------------------------------------------
 * primitives
 * ---------------------------------------------------------------------------------------------------------- */
/* Sphere::Intersect (Object.cpp:10-28).  sqrt(p2sqr) resolves to ::sqrt(double): the subtraction is done in double
 * and rounded to float on assignment (cvtss2sd/sqrtsd/subsd/cvtsd2ss in the compiled reference). */
static int sphere_hit(Ctx* c, const Obj* o, int self, const Ray* ray, Hit* hi) {
    c->cnt[C_SPH_TEST]++;
    v3 L = sub3(ray->o, o->c);
    float p1 = -dot3(ray->d, L);
    float p2sqr = p1 * p1 - dot3(L, L) + o->r * o->r;
    if (p2sqr < 0) return 0;
    hi->distance = (float)((double)p1 - sqrt((double)p2sqr));
    if (hi->distance < 0) return 0;
    c->cnt[C_SPH_HIT]++;
    hi->point = add3(ray->o, scale3(ray->d, hi->distance));
    hi->normal = normalize3(sub3(hi->point, o->c));
    hi->object = self;
    hi->uv.x = m_atan2f(c->s, hi->normal.x, hi->normal.z) / (2 * 3.14159265358979323846264338327950288f) + 0.5f;
    hi->uv.y = hi->normal.y * 0.5f + 0.5f;
    return 1;
}
/* AxisAlignedCube::Intersect (Object.cpp:36-133) */
static int cube_hit(Ctx* c, const Obj* o, int self, const Ray* ray, Hit* hi) {
    c->cnt[C_CUBE_TEST]++;
    const v3 mn = o->box.mn, mx = o->box.mx;
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    v3 cand, tmaxN;
    if (ray->d.x >= 0) { tmin = (mn.x - ray->o.x) / ray->d.x; tmax = (mx.x - ray->o.x) / ray->d.x; hi->normal = V3(-1, 0, 0); tmaxN = hi->normal; }
    else { tmin = (mx.x - ray->o.x) / ray->d.x; tmax = (mn.x - ray->o.x) / ray->d.x; hi->normal = V3(1, 0, 0); tmaxN = hi->normal; }
    if (ray->d.y >= 0) { tymin = (mn.y - ray->o.y) / ray->d.y; tymax = (mx.y - ray->o.y) / ray->d.y; cand = V3(0, -1, 0); }
    else { tymin = (mx.y - ray->o.y) / ray->d.y; tymax = (mn.y - ray->o.y) / ray->d.y; cand = V3(0, 1, 0); }
    if ((tmin > tymax) || (tymin > tmax)) return 0;
    if (tymin > tmin) { tmin = tymin; hi->normal = cand; }
    if (tymax < tmax) { tmax = tymax; tmaxN = cand; }
    if (ray->d.z >= 0) { tzmin = (mn.z - ray->o.z) / ray->d.z; tzmax = (mx.z - ray->o.z) / ray->d.z; cand = V3(0, 0, -1); }
    else { tzmin = (mx.z - ray->o.z) / ray->d.z; tzmax = (mn.z - ray->o.z) / ray->d.z; cand = V3(0, 0, 1); }
    if ((tmin > tzmax) || (tzmin > tmax)) return 0;
    if (tzmin > tmin) { tmin = tzmin; hi->normal = cand; }
    if (tzmax < tmax) { tmax = tzmax; tmaxN = cand; }
    if (tmin < 0.0f && tmax < 0.0f) return 0;
    if (tmin < 0) { hi->normal = tmaxN; tmin = tmax; }
    c->cnt[C_CUBE_HIT]++;
    hi->distance = tmin;
    hi->object = self;
    hi->point = add3(ray->o, scale3(ray->d, tmin));
    hi->point = add3(hi->point, scale3(hi->normal, 0.02f));
    v3 m01 = V3((hi->point.x - mn.x) / (mx.x - mn.x), (hi->point.y - mn.y) / (mx.y - mn.y), (hi->point.z - mn.z) / (mx.z - mn.z));
    if (hi->normal.x == 1 || hi->normal.x == -1) { hi->uv.x = m01.z; hi->uv.y = m01.y; }
    else if (hi->normal.y == 1 || hi->normal.y == -1) { hi->uv.x = m01.x; hi->uv.y = m01.z; }
    else { hi->uv.x = m01.x; hi->uv.y = m01.y; }
    return 1;
}
/* YPlane::Intersect (Object.cpp:139-149); uv is left as it was */
static int plane_hit(Ctx* c, const Obj* o, int self, const Ray* ray, Hit* hi) {
    c->cnt[C_PLANE_TEST]++;
    float t = -(ray->o.y - o->y) / ray->d.y;
    if ((double)t < 1e-3) return 0;
    c->cnt[C_PLANE_HIT]++;
    hi->distance = t;
    hi->point = add3(ray->o, scale3(ray->d, t));
    hi->normal = V3(0.0f, 1.0f, 0.0f);
    hi->object = self;
    return 1;
}
/* float -> glm::uvec2 component as x86-64 g++ performs it: cvttss2si to 64 bits, keep the low 32 */
static uint32_t f2u_x86(float f) {
    if (!(f > -9.2233720368547758e18f && f < 9.2233720368547758e18f)) return 0u;   /* indefinite 0x8000000000000000 */
    return (uint32_t)(uint64_t)(int64_t)f;
}
/* Triangle::Intersect (Object.cpp:498-556) */
static int tri_hit(Ctx* c, const Obj* o, int self, const Ray* ray, Hit* hi) {
    c->cnt[C_TRI_TEST]++;
    const float EPSILON = 0.0000001f;
    const v3 vertex0 = o->p[0], vertex1 = o->p[1], vertex2 = o->p[2];
    v3 edge1 = sub3(vertex1, vertex0);
    v3 edge2 = sub3(vertex2, vertex0);
    v3 h = cross3(ray->d, edge2);
    float a = dot3(edge1, h);
    if (a > -EPSILON && a < EPSILON) return 0;
    float f = 1.0f / a;
    v3 s = sub3(ray->o, vertex0);
    float u = f * dot3(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = cross3(s, edge1);
    float v = f * dot3(ray->d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = f * dot3(edge2, q);
    if (!(t > EPSILON)) return 0;
    c->cnt[C_TRI_UV]++;
    v3 hp = add3(ray->o, scale3(ray->d, t));
    const v3 p1 = vertex0, p2 = vertex1, p3 = vertex2;
    float denominator = p1.y * (p2.z - p3.z) - p2.y * (p1.z - p3.z) + p3.y * (p1.z - p2.z);
    if (denominator == 0) return 0;
    hi->distance = t;
    hi->normal = o->n[0];
    hi->object = self;
    hi->point = add3(hp, scale3(hi->normal, 0.01f));
    float percent0 = (hp.y * (p2.z - p3.z) - hp.z * (p2.y - p3.y) + p2.y * p3.z - p3.y * p2.z) / denominator;
    float percent1 = -(hp.y * (p1.z - p3.z) - hp.z * (p1.y - p3.y) + p1.y * p3.z - p3.y * p1.z) / denominator;
    float percent2 = (hp.y * (p1.z - p2.z) - hp.z * (p1.y - p2.y) + p1.y * p2.z - p2.y * p1.z) / denominator;
    hi->uv.x = o->tc[0].x * percent0 + o->tc[1].x * percent1 + o->tc[2].x * percent2;
    hi->uv.y = o->tc[0].y * percent0 + o->tc[1].y * percent1 + o->tc[2].y * percent2;
    hi->uv.x -= (float)f2u_x86(hi->uv.x);
    hi->uv.y -= (float)f2u_x86(hi->uv.y);
    if (hi->uv.x < 0) hi->uv.x += 1;
    if (hi->uv.y < 0) hi->uv.y += 1;
    const Mat* m = &c->s->mats[o->mat];
    if (m->tex >= 0 && !tex_solid(&c->s->texs[m->tex], hi->uv)) return 0;
    return 1;
}

/* Fog::Intersect (Object.cpp:558-593): a constant-density volume inside an axis-aligned cube.  Draws one Random01 per test
 * that reaches the free-path computation — inside the traversal, so the order of tests matters. */
static int fog_hit(Ctx* c, const Obj* o, int self, const Ray* ray, Hit* hi) {
    c->cnt[C_FOG_TEST]++;
    Hit info1, info2;
    memset(&info1, 0, sizeof info1); info1.object = -1; info1.distance = FLT_MAX;
    info2 = info1;
    if (!cube_hit(c, o, self, ray, &info1)) return 0;
    Ray past;
    past.o = sub3(add3(ray->o, scale3(ray->d, info1.distance)), scale3(info1.normal, 0.001f));
    past.d = ray->d;
    const int inside = !cube_hit(c, o, self, &past, &info2);
    v3 enter, exit_;
    if (inside) { enter = ray->o; exit_ = past.o; }
    else { enter = past.o; exit_ = add3(past.o, scale3(past.d, info2.distance)); }
    const float supposed = o->neg_inv_density * m_logf(c->s, random01(c));
    const v3 dd = sub3(exit_, enter);                       /* glm::distance(a, b) = length(b - a) */
    const float travel = sqrtf(dot3(dd, dd));
    if (supposed > travel) return 0;
    hi->normal = V3(0, 1, 0);
    hi->uv.x = 0; hi->uv.y = 0;
    hi->object = self;
    hi->distance = supposed;
    if (!inside) hi->distance += info1.distance;
    hi->point = add3(ray->o, scale3(scale3(ray->d, hi->distance), 1.001f));
    return 1;
}
static int obj_hit(Ctx* c, int idx, const Ray* ray, Hit* hi);

/* BVH_Node::Intersect (Object.cpp:241-252): the caller's record goes down the left child, a fresh one down the right */
static int node_hit(Ctx* c, const Obj* o, const Ray* ray, Hit* hi) {
    c->cnt[C_AABB]++;
    if (!box_hit(&o->box, ray)) return 0;
    int hit_left = obj_hit(c, o->left, ray, hi);
    Hit info2;
    memset(&info2, 0, sizeof info2);
    info2.object = -1; info2.distance = FLT_MAX;
    int hit_right;
    if (o->right == o->left && !c->in_repeat) {
        /* a node over one object has it as both children (Object.cpp:198-200): the reference does everything below it a
         * second time.  The events of that second visit are also booked apart, so that a traversal which lists such a
         * leaf once can be priced by the work it executes. */
        uint64_t before[C_COUNT];
        memcpy(before, c->cnt, sizeof before);
        c->in_repeat = 1;
        hit_right = obj_hit(c, o->right, ray, &info2);
        c->in_repeat = 0;
        for (int i = 0; i < C_COUNT; i++) c->rep[i] += c->cnt[i] - before[i];
    } else hit_right = obj_hit(c, o->right, ray, &info2);
    if (hit_right && (!hit_left || (hit_left && info2.distance < hi->distance))) *hi = info2;
    return hit_left || hit_right;
}
/* RotationIntersect (Object.cpp:303-336); axis pairs from :338-346 */
static int rot_hit(Ctx* c, const Obj* o, const Ray* ray, Hit* hi) {
    c->cnt[C_INST_ENTER]++;
    const int a1 = o->axis == 0 ? 1 : 0, a2 = o->axis == 2 ? 1 : 2;
    const float sinT = o->sin_t, cosT = o->cos_t;
    v3 oo = sub3(ray->o, o->pivot);
    v3 om = oo;
    v3 dir = ray->d;
    set3(&om, a1, cosT * get3(oo, a1) - sinT * get3(oo, a2));
    set3(&om, a2, sinT * get3(oo, a1) + cosT * get3(oo, a2));
    set3(&dir, a1, cosT * get3(ray->d, a1) - sinT * get3(ray->d, a2));
    set3(&dir, a2, sinT * get3(ray->d, a1) + cosT * get3(ray->d, a2));
    Ray rotated; rotated.o = add3(om, o->pivot); rotated.d = dir;
    if (!obj_hit(c, o->target, &rotated, hi)) return 0;
    c->cnt[C_INST_HIT]++;
    oo = sub3(hi->point, o->pivot);
    om = oo;
    v3 n = hi->normal;
    set3(&om, a1, cosT * get3(oo, a1) + sinT * get3(oo, a2));
    set3(&om, a2, -sinT * get3(oo, a1) + cosT * get3(oo, a2));
    set3(&n, a1, cosT * get3(hi->normal, a1) + sinT * get3(hi->normal, a2));
    set3(&n, a2, -sinT * get3(hi->normal, a1) + cosT * get3(hi->normal, a2));
    hi->point = add3(om, o->pivot);
    hi->normal = n;
    return 1;
}
static int obj_hit(Ctx* c, int idx, const Ray* ray, Hit* hi) {
    const Obj* o = &c->s->objs[idx];
    switch (o->kind) {
    case K_SPHERE: return sphere_hit(c, o, idx, ray, hi);
    case K_CUBE: return cube_hit(c, o, idx, ray, hi);
    case K_PLANE: return plane_hit(c, o, idx, ray, hi);
    case K_TRI: return tri_hit(c, o, idx, ray, hi);
    case K_FOG: return fog_hit(c, o, idx, ray, hi);
    case K_NODE: return node_hit(c, o, ray, hi);
    case K_MESH:   /* PolygonMesh::Intersect (Object.cpp:490-495) */
        if (obj_hit(c, o->root, ray, hi)) { hi->object = idx; return 1; }
        return 0;
    case K_ROT: return rot_hit(c, o, ray, hi);
    case K_MOVE: { /* ApplyMovement::Intersect (Object.cpp:601-608) */
        Ray tr; tr.o = sub3(ray->o, o->offset); tr.d = ray->d;
        if (!obj_hit(c, o->target, &tr, hi)) return 0;
        hi->point = add3(hi->point, o->offset);
        return 1;
    }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * bounding boxes and BVH construction
 * ---------------------------------------------------------------------------------------------------------- */
static Box obj_box(const orc_scene* s, int idx) {
    const Obj* o = &s->objs[idx];
    Box b;
    switch (o->kind) {
    case K_SPHERE:  /* Object.cpp:29-35 (a negative radius yields an inverted box) */
        b.mn = sub3(o->c, V3(o->r, o->r, o->r)); b.mx = add3(o->c, V3(o->r, o->r, o->r)); return b;
    case K_MESH: return obj_box(s, o->root);          /* Object.h:97 */
    case K_MOVE:    /* Object.cpp:610-615 */
        b = obj_box(s, o->target); b.mn = add3(b.mn, o->offset); b.mx = add3(b.mx, o->offset); return b;
    default: return o->box;                           /* cube, triangle, node, rotation */
    }
}
static int new_obj(orc_scene* s, int kind) {
    if (s->n_objs == s->cap_objs) {
        s->cap_objs = s->cap_objs ? s->cap_objs * 2 : 1024;
        s->objs = (Obj*)realloc(s->objs, sizeof(Obj) * (size_t)s->cap_objs);
    }
    Obj* o = &s->objs[s->n_objs];
    memset(o, 0, sizeof *o);
    o->kind = kind; o->mat = -1; o->id = -1; o->left = o->right = o->root = o->target = -1;
    return s->n_objs++;
}

/* std::sort of libstdc++ 11 (bits/stl_algo.h: __sort, __introsort_loop, __unguarded_partition_pivot,
 * __move_median_to_first, __unguarded_partition, __final_insertion_sort, __heap_select/__sort_heap fallback),
 * which Object.cpp:212 calls with a strict "<" on one box coordinate.  Elements are ints; cmp(a,b) = key[a] < key[b]. */
typedef struct { const float* key; } SortCtx;
#define LESS(a, b) (sc->key[(a)] < sc->key[(b)])
static void iswap(int* a, int* b) { int t = *a; *a = *b; *b = t; }
static void unguarded_linear_insert(int* last, const SortCtx* sc) {
    int val = *last;
    int* next = last - 1;
    while (LESS(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void insertion_sort(int* first, int* last, const SortCtx* sc) {
    if (first == last) return;
    for (int* i = first + 1; i != last; ++i) {
        if (LESS(*i, *first)) {
            int val = *i;
            memmove(first + 1, first, sizeof(int) * (size_t)(i - first));
            *first = val;
        } else unguarded_linear_insert(i, sc);
    }
}
static void push_heap_(int* first, long hole, long top, int value, const SortCtx* sc) {
    long parent = (hole - 1) / 2;
    while (hole > top && LESS(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
static void adjust_heap(int* first, long hole, long len, int value, const SortCtx* sc) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (LESS(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value, sc);
}
static void heap_sort_(int* first, int* last, const SortCtx* sc) {   /* __partial_sort(first,last,last) */
    long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) { int v = first[parent]; adjust_heap(first, parent, len, v, sc); if (parent == 0) break; parent--; }
    }
    while (last - first > 1) {
        --last;
        int v = *last; *last = *first;
        adjust_heap(first, 0, last - first, v, sc);
    }
}
static void move_median_to_first(int* result, int* a, int* b, int* c, const SortCtx* sc) {
    if (LESS(*a, *b)) {
        if (LESS(*b, *c)) iswap(result, b);
        else if (LESS(*a, *c)) iswap(result, c);
        else iswap(result, a);
    } else if (LESS(*a, *c)) iswap(result, a);
    else if (LESS(*b, *c)) iswap(result, c);
    else iswap(result, b);
}
static int* unguarded_partition(int* first, int* last, int* pivot, const SortCtx* sc) {
    for (;;) {
        while (LESS(*first, *pivot)) ++first;
        --last;
        while (LESS(*pivot, *last)) --last;
        if (!(first < last)) return first;
        iswap(first, last);
        ++first;
    }
}
static void introsort_loop(int* first, int* last, long depth_limit, const SortCtx* sc) {
    while (last - first > 16) {
        if (depth_limit == 0) { heap_sort_(first, last, sc); return; }
        --depth_limit;
        int* mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1, sc);
        int* cut = unguarded_partition(first + 1, last, first, sc);
        introsort_loop(cut, last, depth_limit, sc);
        last = cut;
    }
}
static void std_sort(int* first, int* last, const SortCtx* sc) {
    if (first == last) return;
    long n = last - first, lg = 0;
    while ((n >> (lg + 1)) > 0) lg++;          /* std::__lg */
    introsort_loop(first, last, lg * 2, sc);
    if (last - first > 16) {
        insertion_sort(first, first + 16, sc);
        for (int* i = first + 16; i != last; ++i) unguarded_linear_insert(i, sc);
    } else insertion_sort(first, last, sc);
}
void orc_std_sort(int n, const float* keys, int* order) {
    SortCtx sc; sc.key = keys;
    std_sort(order, order + n, &sc);
}

/* BVH_Node::BVH_Node (Object.cpp:181-225).  `objs` is the node's working copy of the object list; children sort
 * disjoint sub-ranges, so one shared array gives the same result as the reference's per-node copies. */
static int build_node(orc_scene* s, Ctx* rng, int* objs, int start, int end, float* keybuf) {
    int axis = (int)(random01(rng) * 3);
    int span = end - start;
    int left, right;
    if (span == 1) left = right = objs[start];
    else if (span == 2) {
        Box a = obj_box(s, objs[start]), b = obj_box(s, objs[start + 1]);
        int ax = axis == 0 ? 0 : (axis == 1 ? 1 : 2);
        if (get3(a.mn, ax) < get3(b.mn, ax)) { left = objs[start]; right = objs[start + 1]; }
        else { left = objs[start + 1]; right = objs[start]; }
    } else {
        int ax = axis == 0 ? 0 : (axis == 1 ? 1 : 2);
        for (int i = start; i < end; i++) keybuf[objs[i]] = get3(obj_box(s, objs[i]).mn, ax);
        SortCtx sc; sc.key = keybuf;
        std_sort(objs + start, objs + end, &sc);
        int mid = start + span / 2;
        left = build_node(s, rng, objs, start, mid, keybuf);
        right = build_node(s, rng, objs, mid, end, keybuf);
    }
    Box bl = obj_box(s, left), br = obj_box(s, right);
    int me = new_obj(s, K_NODE);
    Obj* o = &s->objs[me];
    o->left = left; o->right = right;
    /* GetSurroundingBox (Object.cpp:156-166) */
    o->box.mn = V3(fminf(bl.mn.x, br.mn.x), fminf(bl.mn.y, br.mn.y), fminf(bl.mn.z, br.mn.z));
    o->box.mx = V3(fmaxf(bl.mx.x, br.mx.x), fmaxf(bl.mx.y, br.mx.y), fmaxf(bl.mx.z, br.mx.z));
    return me;
}
static int build_bvh(orc_scene* s, Ctx* rng, const int* list, int n) {
    int* objs = (int*)malloc(sizeof(int) * (size_t)n);
    memcpy(objs, list, sizeof(int) * (size_t)n);
    /* keys are indexed by object index; the pool grows while building, so size for the final count */
    float* keybuf = (float*)malloc(sizeof(float) * (size_t)(s->n_objs + 2 * n + 16));
    int root = build_node(s, rng, objs, 0, n, keybuf);
    free(objs); free(keybuf);
    return root;
}

/* PolygonMesh::PolygonMesh (Object.cpp:451-489) from already-parsed vertices, then the wrappers'
 * constructors: InitializeRotationTransform (Object.cpp:254-301) and ApplyMovement (Object.h:148). */
static void finish_mesh(orc_scene* s, Ctx* rng, int mesh_idx) {
    Obj m = s->objs[mesh_idx];
    const int nv = m.n_tri * 3;
    float* vv = m.verts;
    v3 pmin = V3(vv[0], vv[1], vv[2]), pmax = pmin;       /* GetExtentsFromPositionArray (Object.cpp:437-449) */
    for (int i = 0; i < nv; i++) {
        v3 pt = V3(vv[8 * i], vv[8 * i + 1], vv[8 * i + 2]);
        pmin = V3(gmin(pmin.x, pt.x), gmin(pmin.y, pt.y), gmin(pmin.z, pt.z));
        pmax = V3(gmax(pmax.x, pt.x), gmax(pmax.y, pt.y), gmax(pmax.z, pt.z));
    }
    const v3 center = add3(scale3(pmin, 0.5f), scale3(pmax, 0.5f));
    v3 ext = sub3(pmax, pmin);
    const float factor = m.size / gmax(gmax(ext.x, ext.y), ext.z);   /* glm::compMax */
    int* tris = (int*)malloc(sizeof(int) * (size_t)m.n_tri);
    for (int t = 0; t < m.n_tri; t++) {
        int ti = new_obj(s, K_TRI);
        Obj* o = &s->objs[ti];
        o->mat = m.mat; o->id = m.id;
        for (int k = 0; k < 3; k++) {
            const float* r = vv + 8 * (3 * t + k);
            o->p[k] = scale3(sub3(V3(r[0], r[1], r[2]), center), factor);
            o->tc[k].x = r[3]; o->tc[k].y = r[4];
            o->n[k] = V3(r[5], r[6], r[7]);
        }
        /* Triangle box (Object.h:84): glm::min(v1, glm::min(v2, v3)) */
        o->box.mn = V3(gmin(o->p[0].x, gmin(o->p[1].x, o->p[2].x)), gmin(o->p[0].y, gmin(o->p[1].y, o->p[2].y)), gmin(o->p[0].z, gmin(o->p[1].z, o->p[2].z)));
        o->box.mx = V3(gmax(o->p[0].x, gmax(o->p[1].x, o->p[2].x)), gmax(o->p[0].y, gmax(o->p[1].y, o->p[2].y)), gmax(o->p[0].z, gmax(o->p[1].z, o->p[2].z)));
        tris[t] = ti;
    }
    int root = build_bvh(s, rng, tris, m.n_tri);
    free(tris);
    s->objs[mesh_idx].root = root;
}
static void finish_rotation(orc_scene* s, int rot_idx) {
    Obj* o = &s->objs[rot_idx];
    const float angle = o->degrees * 0.01745329251994329576923690768489f;   /* glm::radians */
    o->sin_t = sinf(angle);
    o->cos_t = cosf(angle);
    Box box = obj_box(s, o->target);
    o->pivot = add3(scale3(box.mn, 1.0f - 0.5f), scale3(box.mx, 0.5f));   /* glm::lerp -> mix: x*(1-a) + y*a */
    box.mn = sub3(box.mn, o->pivot);
    box.mx = sub3(box.mx, o->pivot);
    const int a1 = o->axis == 0 ? 1 : 0, a2 = o->axis == 2 ? 1 : 2, ra = o->axis;
    v3 mn = V3(FLT_MAX, FLT_MAX, FLT_MAX), mx = V3(FLT_MIN, FLT_MIN, FLT_MIN);   /* numeric_limits<float>::min() */
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
        float axis1 = i * get3(box.mx, a1) + (1 - i) * get3(box.mn, a1);
        float axis2 = j * get3(box.mx, a2) + (1 - j) * get3(box.mn, a2);
        float leftover = k * get3(box.mx, ra) + (1 - k) * get3(box.mn, ra);
        float new1 = o->cos_t * axis1 + o->sin_t * axis2;
        float new2 = -o->sin_t * axis1 + o->cos_t * axis2;
        v3 tester = V3(0, 0, 0);
        set3(&tester, a1, new1); set3(&tester, a2, new2); set3(&tester, ra, leftover);
        mn = V3(fminf(mn.x, tester.x), fminf(mn.y, tester.y), fminf(mn.z, tester.z));
        mx = V3(fmaxf(mx.x, tester.x), fmaxf(mx.y, tester.y), fmaxf(mx.z, tester.z));
    }
    o->box.mn = add3(mn, o->pivot);
    o->box.mx = add3(mx, o->pivot);
}

/* ------------------------------------------------------------------------------------------------------------
 * scene API
 * ---------------------------------------------------------------------------------------------------------- */
orc_scene* orc_scene_new(void) {
    orc_scene* s = (orc_scene*)calloc(1, sizeof *s);
    s->root = -1;
    pthread_mutex_init(&s->lock, NULL);
    return s;
}
void orc_scene_free(orc_scene* s) {
    if (!s) return;
    for (int i = 0; i < s->n_objs; i++) free(s->objs[i].verts);
    for (int i = 0; i < s->n_texs; i++) free(s->texs[i].img);
    free(s->objs); free(s->texs); free(s->mats); free(s->top); free(s->planes); free(s->sky);
    pthread_mutex_destroy(&s->lock);
    free(s);
}
void orc_set_math(orc_scene* s, int mode) { s->math_mode = mode; }
int orc_add_texture(orc_scene* s, int kind, const float* c1, const float* c2, const uint8_t* rgba, int w, int h) {
    s->texs = (Tex*)realloc(s->texs, sizeof(Tex) * (size_t)(s->n_texs + 1));
    Tex* t = &s->texs[s->n_texs];
    memset(t, 0, sizeof *t);
    t->kind = kind;
    if (c1) t->c1 = V3(c1[0], c1[1], c1[2]);
    if (c2) t->c2 = V3(c2[0], c2[1], c2[2]);
    if (kind == ORC_TEX_IMAGE) {
        /* one extra row: uv.y can wrap to exactly 1.0 (Object.cpp:546-547), which indexes row h */
        size_t n = (size_t)w * (size_t)h * 4;
        t->img = (uint8_t*)calloc(n + (size_t)w * 4 + 16, 1);
        memcpy(t->img, rgba, n);
        t->w = (unsigned)w; t->h = (unsigned)h;
    }
    return s->n_texs++;
}
int orc_add_material(orc_scene* s, int type, int tex, const float* emit) {
    s->mats = (Mat*)realloc(s->mats, sizeof(Mat) * (size_t)(s->n_mats + 1));
    Mat* m = &s->mats[s->n_mats];
    m->type = type; m->tex = tex;
    m->emit = emit ? V3(emit[0], emit[1], emit[2]) : V3(0, 0, 0);
    return s->n_mats++;
}
static int push_top(orc_scene* s, int idx) {
    s->top = (int*)realloc(s->top, sizeof(int) * (size_t)(s->n_top + 1));
    s->top[s->n_top++] = idx;
    return s->objs[idx].id;
}
int orc_add_sphere(orc_scene* s, const float* c, float r, int mat) {
    int i = new_obj(s, K_SPHERE);
    Obj* o = &s->objs[i];
    o->c = V3(c[0], c[1], c[2]); o->r = r; o->mat = mat; o->id = s->next_id++;
    return push_top(s, i);
}
int orc_add_cube_minmax(orc_scene* s, const float* mn, const float* mx, int mat) {   /* Object.h:30 */
    int i = new_obj(s, K_CUBE);
    Obj* o = &s->objs[i];
    o->box.mn = V3(mn[0], mn[1], mn[2]); o->box.mx = V3(mx[0], mx[1], mx[2]); o->mat = mat; o->id = s->next_id++;
    return push_top(s, i);
}
int orc_add_cube_center(orc_scene* s, const float* c, float half, int mat) {          /* Object.h:29 */
    float mn[3] = { c[0] - half, c[1] - half, c[2] - half }, mx[3] = { c[0] + half, c[1] + half, c[2] + half };
    return orc_add_cube_minmax(s, mn, mx, mat);
}
int orc_add_mesh(orc_scene* s, const float* verts8, int n_tri, float size, int mat, int rot_axis, float degrees,
                 int has_move, const float* offset) {
    int id = s->next_id++;
    int mi = new_obj(s, K_MESH);
    Obj* m = &s->objs[mi];
    m->mat = mat; m->id = id; m->n_tri = n_tri; m->size = size;
    m->verts = (float*)malloc(sizeof(float) * 24 * (size_t)n_tri);
    memcpy(m->verts, verts8, sizeof(float) * 24 * (size_t)n_tri);
    int cur = mi;
    if (rot_axis >= 0) {
        int ri = new_obj(s, K_ROT);
        Obj* r = &s->objs[ri];
        r->target = cur; r->axis = rot_axis; r->degrees = degrees; r->id = id;
        cur = ri;
    }
    if (has_move) {
        int vi = new_obj(s, K_MOVE);
        Obj* v = &s->objs[vi];
        v->target = cur; v->offset = V3(offset[0], offset[1], offset[2]); v->id = id;
        cur = vi;
    }
    return push_top(s, cur);
}
/* Fog (Object.h:48-57) inside optional ApplyX/Y/ZRotation and ApplyMovement wrappers (World.cpp:40).  `mat` must be an
 * Isotropic material: the reference's Fog builds its own from the texture it is given. */
int orc_add_fog(orc_scene* s, const float* center, float radius, float density, int mat, int rot_axis, float degrees,
                int has_move, const float* offset) {
    int id = s->next_id++;
    int fi = new_obj(s, K_FOG);
    Obj* f = &s->objs[fi];
    f->mat = mat; f->id = id;
    const v3 c = V3(center[0], center[1], center[2]);
    f->box.mn = sub3(c, V3(radius, radius, radius));        /* AxisAlignedCube(pos, radius): Object.h:29 */
    f->box.mx = add3(c, V3(radius, radius, radius));
    f->neg_inv_density = -1 / density;
    int cur = fi;
    if (rot_axis >= 0) {
        int ri = new_obj(s, K_ROT);
        Obj* r = &s->objs[ri];
        r->target = cur; r->axis = rot_axis; r->degrees = degrees; r->id = id;
        finish_rotation(s, ri);
        cur = ri;
    }
    if (has_move) {
        int vi = new_obj(s, K_MOVE);
        Obj* v = &s->objs[vi];
        v->target = cur; v->offset = V3(offset[0], offset[1], offset[2]); v->id = id;
        cur = vi;
    }
    return push_top(s, cur);
}
int orc_add_plane(orc_scene* s, float y, int mat) {
    int i = new_obj(s, K_PLANE);
    Obj* o = &s->objs[i];
    o->y = y; if (y == 0) o->y = 0.001f;   /* Object.h:40 */
    o->mat = mat; o->id = s->next_id++;
    s->planes = (int*)realloc(s->planes, sizeof(int) * (size_t)(s->n_planes + 1));
    s->planes[s->n_planes++] = i;
    return o->id;
}
void orc_set_skybox(orc_scene* s, const uint8_t* rgb, int w, int h) {
    free(s->sky);
    /* one extra row + pixel: direction.y == 1 or atan2 == pi index one past the image (World.cpp:174-183) */
    size_t n = (size_t)w * (size_t)h * 3;
    s->sky = (uint8_t*)calloc(n + (size_t)w * 3 + 16, 1);
    memcpy(s->sky, rgb, n);
    s->sky_w = (unsigned)w; s->sky_h = (unsigned)h;
}
void orc_build(orc_scene* s, uint32_t build_key) {
    Ctx rng; memset(&rng, 0, sizeof rng);
    rng.s = s; rng.key = build_key; rng.ctr = 0;
    /* constructors run in creation order: each mesh builds its triangle BVH (drawing from the RNG), then its
     * wrappers; the plane list and the top-level BVH follow (World.cpp:47-52 member order: planes draw nothing) */
    const int n_before = s->n_objs;
    for (int i = 0; i < n_before; i++) {
        if (s->objs[i].kind == K_MESH && s->objs[i].root < 0) finish_mesh(s, &rng, i);
        else if (s->objs[i].kind == K_ROT) finish_rotation(s, i);
    }
    s->root = build_bvh(s, &rng, s->top, s->n_top);
}
/* World.cpp:15-32 (the "#if 1" scene) */
void orc_default_scene(orc_scene* s, const float* tree_verts8, int n_tri, const uint8_t* tree_rgba, int tw, int th,
                       const uint8_t* sky_rgb, int sw, int sh) {
    orc_set_skybox(s, sky_rgb, sw, sh);
    float grey[3] = { 0.4f, 0.4f, (float)0.4 };
    float c6[3] = { 6, 2, -9 };
    orc_add_cube_center(s, c6, 2, orc_add_material(s, ORC_MAT_DIFFUSE, orc_add_texture(s, ORC_TEX_COLOR, grey, NULL, NULL, 0, 0), NULL));
    float brown[3] = { 0.6f, 0.3f, 0.2f }, white[3] = { 1.0f, 1.0f, 1.0f };
    float c8[3] = { 8, 2, -4 };
    orc_add_sphere(s, c8, 2, orc_add_material(s, ORC_MAT_DIFFUSE, orc_add_texture(s, ORC_TEX_CHECKER, brown, white, NULL, 0, 0), NULL));
    float off[3] = { 0, 5, 0 };
    orc_add_mesh(s, tree_verts8, n_tri, 10, orc_add_material(s, ORC_MAT_DIFFUSE, orc_add_texture(s, ORC_TEX_IMAGE, NULL, NULL, tree_rgba, tw, th), NULL),
                 0, 90, 1, off);
    for (int i = 0; i < 15; i++) {
        float c[3] = { i * 2.5f + 10, sinf(i * 78.0f) * 1.9f + 1.0f, 15.0f * sinf(i * 34.4f) };
        float r = sinf(i * 78.0f) * 1.9f + 1.0f;
        /* glm::sin(int) is std::sin(int): evaluated in double, narrowed by the vec3 constructor */
        float col[3] = { 0.5f + (float)sin((double)i) * 0.5f, 0.5f + sinf(i * 1.7f) * 0.5f, 0.5f + (float)sin((double)i) * 0.5f };
        orc_add_sphere(s, c, r, orc_add_material(s, ORC_MAT_DIFFUSE, orc_add_texture(s, ORC_TEX_COLOR, col, NULL, NULL, 0, 0), NULL));
    }
    float green[3] = { 0.2f, 0.6f, 0.3f };
    orc_add_plane(s, 0, orc_add_material(s, ORC_MAT_METAL, orc_add_texture(s, ORC_TEX_CHECKER, white, green, NULL, 0, 0), NULL));
}

/* Camera::SetForwardVector (DataUtility.cpp:10-15) */
static void camera_basis(v3 look, v3* fwd, v3* u, v3* v) {
    *fwd = normalize3(look);
    *u = cross3(V3(0, 1, 0), *fwd);
    *v = cross3(*fwd, *u);
}
void orc_camera(const float* pos, const float* look, float* out12) {
    v3 f, u, v;
    camera_basis(V3(look[0], look[1], look[2]), &f, &u, &v);
    out12[0] = pos[0]; out12[1] = pos[1]; out12[2] = pos[2];
    out12[3] = f.x; out12[4] = f.y; out12[5] = f.z;
    out12[6] = u.x; out12[7] = u.y; out12[8] = u.z;
    out12[9] = v.x; out12[10] = v.y; out12[11] = v.z;
}
void orc_camera_input(float* cam, float xrel, float yrel, int key_w, int key_a, int key_s, int key_d) {
    v3 pos = V3(cam[0], cam[1], cam[2]), f = V3(cam[3], cam[4], cam[5]), u = V3(cam[6], cam[7], cam[8]), v = V3(cam[9], cam[10], cam[11]);
    if (xrel != 0.0f || yrel != 0.0f) {   /* App.cpp:98 */
        v3 look = sub3(f, scale3(add3(scale3(u, xrel), scale3(v, yrel)), 0.001f));
        /* MOUSE_SENSITIVITY * (vec): scalar * vec multiplies componentwise — same products */
        camera_basis(look, &f, &u, &v);
    }
    const float speed = 0.25f;            /* App.cpp:114-137 */
    if (key_d) { v3 d = u; d.y = 0; pos = sub3(pos, scale3(d, speed)); }
    if (key_a) { v3 d = u; d.y = 0; pos = add3(pos, scale3(d, speed)); }
    if (key_w) { v3 d = f; d.y = 0; pos = add3(pos, scale3(d, speed)); }
    if (key_s) { v3 d = f; d.y = 0; pos = sub3(pos, scale3(d, speed)); }
    cam[0] = pos.x; cam[1] = pos.y; cam[2] = pos.z; cam[3] = f.x; cam[4] = f.y; cam[5] = f.z;
    cam[6] = u.x; cam[7] = u.y; cam[8] = u.z; cam[9] = v.x; cam[10] = v.y; cam[11] = v.z;
}

/* ------------------------------------------------------------------------------------------------------------
 * World::GetRayColor and helpers (World.cpp:104-193)
 * ---------------------------------------------------------------------------------------------------------- */
static void fresh_hit(Hit* h) { memset(h, 0, sizeof *h); h->object = -1; h->distance = FLT_MAX; }
/* World.cpp:105-119 */
static int closest_hit(Ctx* c, const Ray* ray, Hit* hi) {
    const orc_scene* s = c->s;
    c->cnt[C_RAYS]++;
    int hit = obj_hit(c, s->root, ray, hi);
    Hit th; fresh_hit(&th);
    for (int i = 0; i < s->n_planes; i++) {
        if (!obj_hit(c, s->planes[i], ray, &th)) continue;
        if (th.distance < hi->distance) { hit = 1; *hi = th; }
    }
    return hit;
}
/* World.cpp:173-193 */
static v3 skybox_pixel(const orc_scene* s, v3 d) {
    float u = m_atan2f(s, d.x, d.z) / (2 * 3.14159265358979323846264338327950288f) + 0.5f;
    float v = d.y * 0.5f + 0.5f;
    float px = u * (float)s->sky_w, py = v * (float)s->sky_h;
    int ix = (int)px, iy = (int)py;
    int index = 3 * (int)((unsigned)iy * s->sky_w + (unsigned)ix);
    return divs3(V3((float)s->sky[index], (float)s->sky[index + 1], (float)s->sky[index + 2]), 255.0f);
}
static v3 ray_color(Ctx* c, const Ray* ray, int bounce) {
    const orc_scene* s = c->s;
    Hit hi; fresh_hit(&hi);
    if (!closest_hit(c, ray, &hi)) return scale3(skybox_pixel(s, ray->d), c->sky_brightness);
    c->cnt[C_SHADE]++;
    const Obj* o = &s->objs[hi.object];
    const Mat* m = &s->mats[o->mat];
    switch (m->type) {
    case ORC_MAT_DIFFUSE: {
        if (bounce == 0) return V3(0, 0, 0);
        Ray r;
        r.d = normalize3(add3(hi.normal, unit_sphere_point(c)));
        r.o = hi.point;
        v3 child = ray_color(c, &r, bounce - 1);
        return mul3(scale3(child, 0.5f), tex_color(s, &s->texs[m->tex], hi.uv, hi.point));
    }
    case ORC_MAT_METAL: {
        if (bounce == 0) return V3(0, 0, 0);
        Ray r;
        r.o = hi.point;
        r.d = normalize3(sub3(ray->d, scale3(scale3(hi.normal, 2.0f), dot3(ray->d, hi.normal))));
        v3 child = ray_color(c, &r, bounce - 1);
        return mul3(tex_color(s, &s->texs[m->tex], hi.uv, hi.point), child);
    }
    case ORC_MAT_LIGHT: return m->emit;
    case ORC_MAT_ISOTROPIC: {
        if (bounce == 0) return V3(0, 0, 0);
        Ray r;
        r.o = hi.point;
        r.d = unit_sphere_point(c);
        v3 child = ray_color(c, &r, bounce - 1);
        return mul3(tex_color(s, &s->texs[m->tex], hi.uv, hi.point), child);
    }
    default: return V3(0, 0, 0);
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * World::CalculateColorForScreenPosition (World.cpp:68-102) and the frame-buffer store (FrameManager.cpp:89-94)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct { int W, H, S, B; float fov, dof, focus, sky; v3 pos, fwd, u, v; } Cfg;
static Ray camera_ray(Ctx* c, const Cfg* g, int px, int py, int sx, int sy) {
    float one_x = -(1.0f / g->W) * ((float)g->W / g->H) * g->fov, one_y = -(1.0f / g->H) * g->fov;
    float off_x = -((float)px / g->W - 0.5f) * ((float)g->W / g->H) * g->fov, off_y = -((float)py / g->H - 0.5f) * g->fov;
    Ray ray;
    ray.d = g->fwd;
    ray.d = add3(ray.d, add3(scale3(g->u, off_x + one_x * ((float)sx / g->S)), scale3(g->v, off_y + one_y * ((float)sy / g->S))));
    ray.d = normalize3(ray.d);
    v2 lens = unit_circle_point(c);
    lens.x = lens.x * g->dof; lens.y = lens.y * g->dof;
    v3 wo = add3(scale3(g->u, lens.x), scale3(g->v, lens.y));
    ray.o = g->pos;
    ray.o = add3(ray.o, wo);
    ray.d = sub3(ray.d, divs3(wo, g->focus));
    ray.d = normalize3(ray.d);
    return ray;
}
static uint8_t to_u8(float f) {   /* (char)(f * 255): cvttss2si then the low byte; NaN/out of range -> 0x80000000 -> 0 */
    float v = f * 255;
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return 0;
    return (uint8_t)(int)v;
}
static void resolve_pixel(v3 color, int S, uint8_t* out4) {
    color = divs3(color, (float)(S * S));
    const float steepness = 5.0f;
    color = V3(1.0f - 1.0f / (1.0f + steepness * color.x), 1.0f - 1.0f / (1.0f + steepness * color.y), 1.0f - 1.0f / (1.0f + steepness * color.z));
    out4[0] = 255; out4[1] = to_u8(color.z); out4[2] = to_u8(color.y); out4[3] = to_u8(color.x);
}
static void cfg_init(Cfg* g, const int* ci, const float* cf, const float* cam) {
    g->W = ci[0]; g->H = ci[1]; g->S = ci[2]; g->B = ci[3];
    g->fov = cf[0]; g->dof = cf[1]; g->focus = cf[2]; g->sky = cf[3];
    g->pos = V3(cam[0], cam[1], cam[2]); g->fwd = V3(cam[3], cam[4], cam[5]); g->u = V3(cam[6], cam[7], cam[8]); g->v = V3(cam[9], cam[10], cam[11]);
}
static void merge_counters(orc_scene* s, const Ctx* c) {
    pthread_mutex_lock(&s->lock);
    for (int i = 0; i < C_COUNT; i++) { s->counters[i] += c->cnt[i]; s->repeats[i] += c->rep[i]; }
    pthread_mutex_unlock(&s->lock);
}

typedef struct {
    orc_scene* s; Cfg g; uint32_t seed; int x0, y0, x1, y1; long p_begin, p_end;
    float* radiance; int32_t* primary_id; float* primary_t; float* ray6; uint32_t* rng_draws; uint8_t* rgba;
} Job;
static void* render_job(void* arg) {
    Job* j = (Job*)arg;
    const Cfg* g = &j->g;
    Ctx c; memset(&c, 0, sizeof c);
    c.s = j->s; c.sky_brightness = g->sky;
    const int rw = j->x1 - j->x0, spp = g->S * g->S;
    for (long lp = j->p_begin; lp < j->p_end; lp++) {
        int x = j->x0 + (int)(lp % rw), y = j->y0 + (int)(lp / rw);
        v3 color = V3(0, 0, 0);
        for (int sx = 0; sx < g->S; sx++) for (int sy = 0; sy < g->S; sy++) {
            int sidx = sx * g->S + sy;
            long si = lp * spp + sidx;
            c.key = orc_sample_key(j->seed, (uint32_t)(y * g->W + x), (uint32_t)sidx); c.ctr = 0;
            c.cnt[C_SAMPLES]++;
            Ray ray = camera_ray(&c, g, x, y, sx, sy);
            if (j->ray6) { float* r = j->ray6 + 6 * si; r[0] = ray.o.x; r[1] = ray.o.y; r[2] = ray.o.z; r[3] = ray.d.x; r[4] = ray.d.y; r[5] = ray.d.z; }
            if (j->primary_id || j->primary_t) {
                Ctx probe = c;   /* does not disturb the RNG stream or the counters */
                Hit hi; fresh_hit(&hi);
                int ok = closest_hit(&probe, &ray, &hi);
                if (j->primary_id) j->primary_id[si] = ok ? j->s->objs[hi.object].id : -1;
                if (j->primary_t) j->primary_t[si] = hi.distance;
            }
            if (!j->radiance && !j->rng_draws && !j->rgba) continue;
            v3 col = ray_color(&c, &ray, g->B);
            if (j->radiance) { j->radiance[3 * si] = col.x; j->radiance[3 * si + 1] = col.y; j->radiance[3 * si + 2] = col.z; }
            if (j->rng_draws) j->rng_draws[si] = c.ctr;
            color = add3(color, col);
        }
        if (j->rgba) resolve_pixel(color, g->S, j->rgba + 4 * lp);
    }
    merge_counters(j->s, &c);
    return NULL;
}
void orc_render(orc_scene* s, const int* cfg_i, const float* cfg_f, const float* cam12, uint32_t seed,
                int x0, int y0, int x1, int y1, int n_threads,
                float* radiance, int32_t* primary_id, float* primary_t, float* ray6, uint32_t* rng_draws, uint8_t* rgba) {
    if (n_threads < 1) n_threads = 1;
    Job* jobs = (Job*)calloc((size_t)n_threads, sizeof(Job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    const long n_pix = (long)(x1 - x0) * (y1 - y0);
    for (int t = 0; t < n_threads; t++) {
        Job* j = &jobs[t];
        j->s = s; cfg_init(&j->g, cfg_i, cfg_f, cam12); j->seed = seed; j->x0 = x0; j->y0 = y0; j->x1 = x1; j->y1 = y1;
        /* contiguous pixel ranges per worker (FrameManager.cpp:80-82) */
        j->p_begin = (long)t * n_pix / n_threads; j->p_end = (long)(t + 1) * n_pix / n_threads;
        j->radiance = radiance; j->primary_id = primary_id; j->primary_t = primary_t; j->ray6 = ray6; j->rng_draws = rng_draws; j->rgba = rgba;
        if (n_threads == 1) render_job(j); else pthread_create(&th[t], NULL, render_job, j);
    }
    if (n_threads > 1) for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(jobs); free(th);
}
void orc_render_pixel_keyed(orc_scene* s, const int* cfg_i, const float* cfg_f, const float* cam12, uint32_t seed,
                            int x0, int y0, int x1, int y1, uint8_t* rgba) {
    Cfg g; cfg_init(&g, cfg_i, cfg_f, cam12);
    Ctx c; memset(&c, 0, sizeof c);
    c.s = s; c.sky_brightness = g.sky;
    long lp = 0;
    for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++, lp++) {
        c.key = orc_sample_key(seed, (uint32_t)(y * g.W + x), 0xFFFFFFFFu); c.ctr = 0;
        v3 color = V3(0, 0, 0);
        for (int sx = 0; sx < g.S; sx++) for (int sy = 0; sy < g.S; sy++) {
            c.cnt[C_SAMPLES]++;
            Ray ray = camera_ray(&c, &g, x, y, sx, sy);
            color = add3(color, ray_color(&c, &ray, g.B));
        }
        resolve_pixel(color, g.S, rgba + 4 * lp);
    }
    merge_counters(s, &c);
}
/* One frame as the reference's own scheduler produces it when rand() is the keyed generator with thread-local
 * state: every worker starts at key 0 / counter 0 and the counter runs on through the pixels in the worker's
 * visiting order (FrameManager.cpp:78-108).  rgba is the full W*H*4 buffer, zeroed first (FrameManager.cpp:13). */
int orc_thread_pixel_order(int W, int H, int n_threads, int spread, int thread, int max_out, uint32_t* pixel_index);
void orc_render_framemanager(orc_scene* s, const int* cfg_i, const float* cfg_f, const float* cam12, int n_threads, int spread, uint8_t* rgba) {
    Cfg g; cfg_init(&g, cfg_i, cfg_f, cam12);
    memset(rgba, 0, (size_t)g.W * (size_t)g.H * 4);
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)g.W * (size_t)g.H);
    for (int t = 0; t < n_threads; t++) {
        int n = orc_thread_pixel_order(g.W, g.H, n_threads, spread, t, g.W * g.H, order);
        Ctx c; memset(&c, 0, sizeof c);
        c.s = s; c.sky_brightness = g.sky; c.key = 0; c.ctr = 0;
        for (int i = 0; i < n; i++) {
            int x = (int)(order[i] % (uint32_t)g.W), y = (int)(order[i] / (uint32_t)g.W);
            v3 color = V3(0, 0, 0);
            for (int sx = 0; sx < g.S; sx++) for (int sy = 0; sy < g.S; sy++) {
                Ray ray = camera_ray(&c, &g, x, y, sx, sy);
                color = add3(color, ray_color(&c, &ray, g.B));
            }
            resolve_pixel(color, g.S, rgba + 4 * (size_t)order[i]);
        }
    }
    free(order);
}
/* FrameManager::ThreadWork's visiting order (FrameManager.cpp:80-107): pixel indices in the order worker `thread`
 * computes them */
int orc_thread_pixel_order(int W, int H, int n_threads, int spread, int thread, int max_out, uint32_t* pixel_index) {
    const uint32_t pixelAmount = (uint32_t)W * (uint32_t)H;
    const uint32_t startIndex = ((uint32_t)thread * pixelAmount) / (uint32_t)n_threads * 4;
    const uint32_t endIndex = (((uint32_t)thread + 1) * pixelAmount) / (uint32_t)n_threads * 4;
    uint32_t cur = startIndex, next = 1;
    int n = 0;
    for (;;) {
        if (n < max_out) pixel_index[n] = cur / 4;
        n++;
        cur += 4 * (uint32_t)spread;
        if (cur >= endIndex) { cur = startIndex + next * 4; next++; }
        if (next > (uint32_t)spread) break;
    }
    return n;
}

void orc_trace(orc_scene* s, int n, const float* o3, const float* d3, int32_t* hit, float* rec9) {
    Ctx c; memset(&c, 0, sizeof c); c.s = s;
    for (int i = 0; i < n; i++) {
        Ray r; r.o = V3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]); r.d = V3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        Hit hi; fresh_hit(&hi);
        c.key = (uint32_t)i; c.ctr = 0;                 /* fog tests draw random numbers: one stream per ray */
        int ok = closest_hit(&c, &r, &hi);
        hit[i] = ok ? s->objs[hi.object].id : -1;
        float* q = rec9 + 9 * i;
        q[0] = hi.distance; q[1] = hi.point.x; q[2] = hi.point.y; q[3] = hi.point.z;
        q[4] = hi.normal.x; q[5] = hi.normal.y; q[6] = hi.normal.z; q[7] = hi.uv.x; q[8] = hi.uv.y;
    }
    merge_counters(s, &c);
}
void orc_ray_color(orc_scene* s, int n, const float* o3, const float* d3, int bounces, const uint32_t* keys, float sky_brightness, float* rgb) {
    Ctx c; memset(&c, 0, sizeof c); c.s = s; c.sky_brightness = sky_brightness;
    for (int i = 0; i < n; i++) {
        Ray r; r.o = V3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]); r.d = V3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        c.key = keys[i]; c.ctr = 0;
        v3 col = ray_color(&c, &r, bounces);
        rgb[3 * i] = col.x; rgb[3 * i + 1] = col.y; rgb[3 * i + 2] = col.z;
    }
    merge_counters(s, &c);
}
void orc_skybox(orc_scene* s, int n, const float* d3, float* rgb) {
    for (int i = 0; i < n; i++) {
        v3 c = skybox_pixel(s, V3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]));
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}
void orc_counters(orc_scene* s, uint64_t* out15, int reset) {
    pthread_mutex_lock(&s->lock);
    for (int i = 0; i < C_COUNT; i++) { out15[i] = s->counters[i]; if (reset) { s->counters[i] = 0; s->repeats[i] = 0; } }
    pthread_mutex_unlock(&s->lock);
}
void orc_repeat_counters(orc_scene* s, uint64_t* out16) {
    pthread_mutex_lock(&s->lock);
    for (int i = 0; i < C_COUNT; i++) out16[i] = s->repeats[i];
    pthread_mutex_unlock(&s->lock);
}

/* ------------------------------------------------------------------------------------------------------------
 * primitive-level entry points and dumps
 * ---------------------------------------------------------------------------------------------------------- */
void orc_aabb_hit(const float* b6, int n, const float* o3, const float* d3, uint8_t* out) {
    Box b; b.mn = V3(b6[0], b6[1], b6[2]); b.mx = V3(b6[3], b6[4], b6[5]);
    for (int i = 0; i < n; i++) {
        Ray r; r.o = V3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]); r.d = V3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        out[i] = (uint8_t)box_hit(&b, &r);
    }
}
void orc_normalize(int n, const float* v, float* out) {
    for (int i = 0; i < n; i++) {
        v3 r = normalize3(V3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
static void write_rec(int ok, const Hit* hi, int32_t* hit, float* q) {
    *hit = ok ? 0 : -1;
    q[0] = hi->distance; q[1] = hi->point.x; q[2] = hi->point.y; q[3] = hi->point.z;
    q[4] = hi->normal.x; q[5] = hi->normal.y; q[6] = hi->normal.z; q[7] = hi->uv.x; q[8] = hi->uv.y;
}
static void fill_hit(Hit* h, float fill) { fresh_hit(h); h->point = V3(fill, fill, fill); h->normal = h->point; h->uv.x = h->uv.y = fill; }
void orc_intersect_prim(int kind, const float* p, int n, const float* o3, const float* d3, float fill, int32_t* hit, float* rec9) {
    orc_scene* s = orc_scene_new();
    Ctx c; memset(&c, 0, sizeof c); c.s = s;
    Obj o; memset(&o, 0, sizeof o);
    if (kind == 0) { o.kind = K_SPHERE; o.c = V3(p[0], p[1], p[2]); o.r = p[3]; }
    else if (kind == 1) { o.kind = K_CUBE; o.box.mn = V3(p[0], p[1], p[2]); o.box.mx = V3(p[3], p[4], p[5]); }
    else { o.kind = K_PLANE; o.y = p[0]; if (p[0] == 0) o.y = 0.001f; }
    for (int i = 0; i < n; i++) {
        Ray r; r.o = V3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]); r.d = V3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        Hit hi; fill_hit(&hi, fill);
        int ok = kind == 0 ? sphere_hit(&c, &o, 0, &r, &hi) : (kind == 1 ? cube_hit(&c, &o, 0, &r, &hi) : plane_hit(&c, &o, 0, &r, &hi));
        write_rec(ok, &hi, hit + i, rec9 + 9 * i);
    }
    orc_scene_free(s);
}
void orc_intersect_tri(const float* v24, const uint8_t* rgba, int w, int h, int n, const float* o3, const float* d3,
                       float fill, int32_t* hit, float* rec9) {
    orc_scene* s = orc_scene_new();
    float white[3] = { 1, 1, 1 };
    int tex = rgba ? orc_add_texture(s, ORC_TEX_IMAGE, NULL, NULL, rgba, w, h) : orc_add_texture(s, ORC_TEX_COLOR, white, NULL, NULL, 0, 0);
    int mat = orc_add_material(s, ORC_MAT_DIFFUSE, tex, NULL);
    Ctx c; memset(&c, 0, sizeof c); c.s = s;
    Obj o; memset(&o, 0, sizeof o);
    o.kind = K_TRI; o.mat = mat;
    for (int k = 0; k < 3; k++) {
        const float* r = v24 + 8 * k;
        o.p[k] = V3(r[0], r[1], r[2]); o.tc[k].x = r[3]; o.tc[k].y = r[4]; o.n[k] = V3(r[5], r[6], r[7]);
    }
    for (int i = 0; i < n; i++) {
        Ray r; r.o = V3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]); r.d = V3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        Hit hi; fill_hit(&hi, fill);
        int ok = tri_hit(&c, &o, 0, &r, &hi);
        write_rec(ok, &hi, hit + i, rec9 + 9 * i);
    }
    orc_scene_free(s);
}

typedef struct { float* boxes; int* left; int* right; int n, max; int* leaves; int n_leaves, max_leaves; int* slot_of; } Dump;
static int dump_rec(const orc_scene* s, int idx, Dump* d) {
    const Obj* o = &s->objs[idx];
    if (o->kind == K_NODE) {
        int me = d->n++;
        if (me < d->max) {
            float* b = d->boxes + 6 * me;
            b[0] = o->box.mn.x; b[1] = o->box.mn.y; b[2] = o->box.mn.z; b[3] = o->box.mx.x; b[4] = o->box.mx.y; b[5] = o->box.mx.z;
        }
        int l = dump_rec(s, o->left, d);
        int r = dump_rec(s, o->right, d);
        if (me < d->max) { d->left[me] = l; d->right[me] = r; }
        return me;
    }
    if (d->slot_of[idx] >= 0) return ~d->slot_of[idx];
    int slot = d->n_leaves++;
    d->slot_of[idx] = slot;
    if (slot < d->max_leaves) d->leaves[slot] = idx;
    return ~slot;
}
int orc_dump_top(orc_scene* s, int max_nodes, float* boxes6, int* left, int* right, int* leaf_ids, int* n_leaves) {
    Dump d; memset(&d, 0, sizeof d);
    d.boxes = boxes6; d.left = left; d.right = right; d.max = max_nodes;
    d.leaves = (int*)malloc(sizeof(int) * (size_t)(max_nodes + 1)); d.max_leaves = max_nodes + 1;
    d.slot_of = (int*)malloc(sizeof(int) * (size_t)s->n_objs);
    for (int i = 0; i < s->n_objs; i++) d.slot_of[i] = -1;
    dump_rec(s, s->root, &d);
    free(d.slot_of);
    for (int i = 0; i < d.n_leaves && i < d.max_leaves; i++) leaf_ids[i] = s->objs[d.leaves[i]].id;
    *n_leaves = d.n_leaves;
    free(d.leaves);
    return d.n <= max_nodes ? d.n : -d.n;
}
static int find_top(const orc_scene* s, int object_id) {
    for (int i = 0; i < s->n_top; i++) if (s->objs[s->top[i]].id == object_id) return s->top[i];
    return -1;
}
static int find_mesh(const orc_scene* s, int idx) {
    while (idx >= 0 && s->objs[idx].kind != K_MESH) {
        if (s->objs[idx].kind == K_ROT || s->objs[idx].kind == K_MOVE) idx = s->objs[idx].target; else return -1;
    }
    return idx;
}
int orc_dump_mesh(orc_scene* s, int object_id, int max_nodes, float* boxes6, int* left, int* right, int max_tris,
                  float* tris24, int* n_tris) {
    int mi = find_mesh(s, find_top(s, object_id));
    if (mi < 0) return 0;
    Dump d; memset(&d, 0, sizeof d);
    d.boxes = boxes6; d.left = left; d.right = right; d.max = max_nodes;
    d.leaves = (int*)malloc(sizeof(int) * (size_t)max_tris); d.max_leaves = max_tris;
    d.slot_of = (int*)malloc(sizeof(int) * (size_t)s->n_objs);
    for (int i = 0; i < s->n_objs; i++) d.slot_of[i] = -1;
    dump_rec(s, s->objs[mi].root, &d);
    free(d.slot_of);
    for (int i = 0; i < d.n_leaves && i < max_tris; i++) {
        const Obj* t = &s->objs[d.leaves[i]];
        for (int k = 0; k < 3; k++) {
            float* r = tris24 + 24 * i + 8 * k;
            r[0] = t->p[k].x; r[1] = t->p[k].y; r[2] = t->p[k].z; r[3] = t->tc[k].x; r[4] = t->tc[k].y;
            r[5] = t->n[k].x; r[6] = t->n[k].y; r[7] = t->n[k].z;
        }
    }
    *n_tris = d.n_leaves;
    free(d.leaves);
    return d.n <= max_nodes ? d.n : -d.n;
}
int orc_dump_objects(orc_scene* s, int max_obj, float* out) {
    int count = 0;
    for (int pass = 0; pass < 2; pass++) {
        int n = pass == 0 ? s->n_top : s->n_planes;
        for (int i = 0; i < n; i++) {
            int idx = pass == 0 ? s->top[i] : s->planes[i];
            const Obj* o = &s->objs[idx];
            if (o->id < 0 || o->id >= max_obj) return -1;
            float* r = out + 20 * o->id;
            for (int k = 0; k < 20; k++) r[k] = 0;
            int mat = o->mat;
            if (o->kind == K_SPHERE) { r[0] = 0; r[1] = o->c.x; r[2] = o->c.y; r[3] = o->c.z; r[4] = o->r; }
            else if (o->kind == K_CUBE) { r[0] = 1; r[1] = o->box.mn.x; r[2] = o->box.mn.y; r[3] = o->box.mn.z; r[4] = o->box.mx.x; r[5] = o->box.mx.y; r[6] = o->box.mx.z; }
            else if (o->kind == K_PLANE) { r[0] = 4; r[1] = o->y; }
            else {
                r[0] = 3;
                Box b = obj_box(s, idx);
                r[1] = b.mn.x; r[2] = b.mn.y; r[3] = b.mn.z; r[4] = b.mx.x; r[5] = b.mx.y; r[6] = b.mx.z;
                mat = s->objs[find_mesh(s, idx)].mat;
            }
            const Mat* m = &s->mats[mat];
            r[9] = (float)m->type;
            if (m->tex < 0) r[10] = -1;
            else {
                const Tex* t = &s->texs[m->tex];
                r[10] = (float)t->kind;
                if (t->kind == ORC_TEX_COLOR || t->kind == ORC_TEX_CHECKER) { r[11] = t->c1.x; r[12] = t->c1.y; r[13] = t->c1.z; }
                if (t->kind == ORC_TEX_CHECKER) { r[14] = t->c2.x; r[15] = t->c2.y; r[16] = t->c2.z; }
            }
            r[17] = m->emit.x; r[18] = m->emit.y; r[19] = m->emit.z;
            count++;
        }
    }
    return count;
}
int orc_dump_instance(orc_scene* s, int object_id, float* out) {
    int idx = find_top(s, object_id);
    if (idx < 0) return 0;
    for (int i = 0; i < 17; i++) out[i] = 0;
    out[8] = -1;
    const Obj* o = &s->objs[idx];
    if (o->kind == K_MOVE) { out[0] = o->offset.x; out[1] = o->offset.y; out[2] = o->offset.z; out[15] = 1; o = &s->objs[o->target]; }
    if (o->kind == K_ROT) {
        out[3] = o->pivot.x; out[4] = o->pivot.y; out[5] = o->pivot.z; out[6] = o->sin_t; out[7] = o->cos_t; out[8] = (float)o->axis;
        out[9] = o->box.mn.x; out[10] = o->box.mn.y; out[11] = o->box.mn.z; out[12] = o->box.mx.x; out[13] = o->box.mx.y; out[14] = o->box.mx.z;
        out[16] = 1;
    }
    return 1;
}
