/* TEST INFRASTRUCTURE — CPU restatement (plain C) of the RaskiTech/Raytracer per-pixel path.
 *
 * This is the parity oracle for the HIP renderer: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  It is not shipped, and the renderer never calls it.
 *
 * Pinned: checked bit-for-bit (primary-hit ids, hit distances, per-sample radiance, 8-bit frames, BVH
 * topology and boxes) against the reference's own C++ compiled from /root/reference (oracle/_ref, see
 * oracle/Makefile) and against the golden fixtures that build wrote to tests/golden/.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* texture kinds / material types (MaterialType enum order of DataUtility.h:109-111) */
enum { ORC_TEX_COLOR = 0, ORC_TEX_CHECKER = 1, ORC_TEX_IMAGE = 2 };
enum { ORC_MAT_NONE = 0, ORC_MAT_DIFFUSE = 1, ORC_MAT_METAL = 2, ORC_MAT_ISOTROPIC = 3, ORC_MAT_LIGHT = 4 };
/* math modes: libm = host sinf/atan2f (what the reference links); portable = the fp64-evaluated sinf/atan2f
 * the HIP kernels implement, so GPU results can be compared bit-for-bit */
enum { ORC_MATH_LIBM = 0, ORC_MATH_PORTABLE = 1 };

orc_scene* orc_scene_new(void);
void orc_scene_free(orc_scene*);
void orc_set_math(orc_scene*, int mode);

int orc_add_texture(orc_scene*, int kind, const float* c1, const float* c2, const uint8_t* rgba, int w, int h);
int orc_add_material(orc_scene*, int type, int tex, const float* emit);
int orc_add_sphere(orc_scene*, const float* c, float r, int mat);
int orc_add_cube_minmax(orc_scene*, const float* mn, const float* mx, int mat);
int orc_add_cube_center(orc_scene*, const float* c, float half, int mat);
/* verts: what the reference's OBJ parser yields: n_tri*3 vertices x (pos3, uv2, normal3).  rot_axis -1 = none. */
int orc_add_mesh(orc_scene*, const float* verts8, int n_tri, float size, int mat, int rot_axis, float degrees,
                 int has_move, const float* offset);
/* Fog (Object.h:48-57): cube boundary centre +- radius, density; `mat` an Isotropic material; optional rotation /
 * movement wrappers as for meshes (World.cpp:40) */
int orc_add_fog(orc_scene*, const float* center, float radius, float density, int mat, int rot_axis, float degrees,
                int has_move, const float* offset);
int orc_add_plane(orc_scene*, float y, int mat);
void orc_set_skybox(orc_scene*, const uint8_t* rgb, int w, int h);
/* the reference's default scene (World.cpp:15-32); objects get ids 0..17, the plane 18 */
void orc_default_scene(orc_scene*, const float* tree_verts8, int n_tri, const uint8_t* tree_rgba, int tw, int th,
                       const uint8_t* sky_rgb, int sw, int sh);
/* BVH construction with the keyed RNG (Object.cpp:181-225); call once after all objects are added */
void orc_build(orc_scene*, uint32_t build_key);

/* camera (DataUtility.cpp:10-15): out12 = pos, fwd, u, v */
void orc_camera(const float* pos, const float* look, float* out12);
/* front-end input mapping (App.cpp:98,114-137) applied to cam12 in place */
void orc_camera_input(float* cam12, float xrel, float yrel, int key_w, int key_a, int key_s, int key_d);

/* dumps for topology parity: preorder inner nodes; child >=0 inner index, <0 ~leaf slot (first-visit order) */
int orc_dump_top(orc_scene*, int max_nodes, float* boxes6, int* left, int* right, int* leaf_ids, int* n_leaves);
int orc_dump_mesh(orc_scene*, int object_id, int max_nodes, float* boxes6, int* left, int* right, int max_tris,
                  float* tris24, int* n_tris);
int orc_dump_objects(orc_scene*, int max_obj, float* out20);
int orc_dump_instance(orc_scene*, int object_id, float* out17);

/* primitive level */
void orc_aabb_hit(const float* box6, int n, const float* o3, const float* d3, uint8_t* out);
void orc_normalize(int n, const float* v3, float* out3);
void orc_intersect_prim(int kind, const float* p, int n, const float* o3, const float* d3, float fill, int32_t* hit, float* rec9);
void orc_intersect_tri(const float* verts24, const uint8_t* rgba, int w, int h, int n, const float* o3, const float* d3,
                       float fill, int32_t* hit, float* rec9);

/* scene level */
void orc_trace(orc_scene*, int n, const float* o3, const float* d3, int32_t* hit, float* rec9);
void orc_ray_color(orc_scene*, int n, const float* o3, const float* d3, int bounces, const uint32_t* keys, float sky_brightness, float* rgb);
void orc_skybox(orc_scene*, int n, const float* d3, float* rgb);

/* cfg_i = {W,H,S,B}; cfg_f = {fov, dof, focus, sky_brightness}.  Outputs as oracle/ref_harness.cpp ref_render. */
void orc_render(orc_scene*, const int* cfg_i, const float* cfg_f, const float* cam12, uint32_t seed,
                int x0, int y0, int x1, int y1, int n_threads,
                float* radiance, int32_t* primary_id, float* primary_t, float* ray6, uint32_t* rng_draws, uint8_t* rgba);
/* one key per PIXEL, counter running through its samples — matches the reference's own
 * CalculateColorForScreenPosition under ref_check_pixel_loop's keying */
void orc_render_pixel_keyed(orc_scene*, const int* cfg_i, const float* cfg_f, const float* cam12, uint32_t seed,
                            int x0, int y0, int x1, int y1, uint8_t* rgba);
/* the reference scheduler's pixel visiting order for one worker (FrameManager.cpp:80-107) */
int orc_thread_pixel_order(int W, int H, int n_threads, int spread, int thread, int max_out, uint32_t* pixel_index);

/* a whole frame exactly as the reference's FrameManager renders it under a thread-local keyed rand() (all workers
 * key 0, counters running through each worker's pixel order); rgba = W*H*4 */
void orc_render_framemanager(orc_scene*, const int* cfg_i, const float* cfg_f, const float* cam12, int n_threads, int spread, uint8_t* rgba);

/* work counters accumulated by orc_render/orc_trace/orc_ray_color since the last reset (15 values):
 * rays, aabb, sph_test, sph_hit, cube_test, cube_hit, plane_test, plane_hit, tri_test, tri_uv, inst_enter, inst_hit,
 * samples, shade, rng */
void orc_counters(orc_scene*, uint64_t* out16, int reset);
/* the part of those counters that was spent on the second visit of a leaf the reference lists as both children of a node
 * (Object.cpp:198-200, 241-252): work a traversal that lists such a leaf once does not repeat.  Reset with orc_counters. */
void orc_repeat_counters(orc_scene*, uint64_t* out16);
float orc_pm_logf(float x);

/* RNG and portable math, exposed for known-answer tests */
uint32_t orc_rand31(uint32_t key, uint32_t ctr);
uint32_t orc_sample_key(uint32_t seed, uint32_t pixel, uint32_t sample);
float orc_pm_sinf(float x);
float orc_pm_atan2f(float y, float x);
/* libstdc++-11 std::sort restated (introsort + final insertion sort) on (key,index) pairs, comparing keys only */
void orc_std_sort(int n, const float* keys, int* order_inout);

#ifdef __cplusplus
}
#endif
#endif
