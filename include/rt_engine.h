/* rt_engine.h — C ABI of the MI355X renderer for the RaskiTech/Raytracer per-pixel path.
 *
 * The reference has no plugin or FFI layer: its seam is the C++ class FrameManager
 * (Raytracer/src/FrameManager.h:12-37) with World (World.h:12-32) and Camera (DataUtility.h:136-154)
 * behind it.  The functions below are what a binding for that seam calls; the C++ adapter in
 * raytracer_amd/csrc/compat/ (FrameManager / World / Camera with the reference's signatures) and the Python
 * binding raytracer_amd/engine.py are both written on top of them.  All citations are relative to
 * /root/reference/Raytracer/src.
 *
 * Conventions: every function returns RT_OK (0) or a negative RT_E_* code; rt_last_error() gives the message
 * of the last failure on the calling thread.  No exceptions cross this boundary.  An engine or a scene is
 * owned by one thread at a time.  Pointers are host pointers unless the name starts with d_.
 */
#ifndef RT_ENGINE_H
#define RT_ENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_INVALID (-1)   /* bad argument / object in the wrong state */
#define RT_E_DEVICE (-2)    /* HIP runtime error */
#define RT_E_IO (-3)        /* asset could not be read or decoded */
#define RT_E_NOMEM (-4)

/* MaterialType (DataUtility.h:109-111) and the Texture subclasses (DataUtility.h:75-107) */
enum { RT_MAT_NONE = 0, RT_MAT_DIFFUSE = 1, RT_MAT_METAL = 2, RT_MAT_ISOTROPIC = 3, RT_MAT_LIGHT = 4 };
enum { RT_TEX_COLOR = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2 };

typedef struct rt_scene rt_scene;     /* host-side scene under construction / flattened */
typedef struct rt_engine rt_engine;   /* one GPU: device copies of a scene, ray queues, frame buffer */

const char* rt_last_error(void);
/* A digest of the kernel sources this library was compiled from (first 16 hex digits of the sha256 of rt_engine.hip,
 * rt_device.h and rt_flat.h, in that order; "unknown" for builds that do not pass it): profiles record it, so that a
 * measurement can be tied to the code that ran. */
const char* rt_source_hash(void);

/* ---------------------------------------------------------------------------------------------------------
 * Quality settings: the QUALITY and PROGRAM blocks of Constants.h:4-17, made run-time values.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct rt_config {
    int32_t width, height;           /* WINDOW_WIDTH, WINDOW_HEIGHT            (Constants.h:4-5)   */
    int32_t samples_per_axis;        /* SAMPLES_PER_PIXEL_AXIS                  (Constants.h:17)    */
    int32_t bounces;                 /* LIGHT_BOUNCE_AMOUNT                     (Constants.h:16)    */
    float fov;                       /* FIELD_OF_VIEW                           (Constants.h:12)    */
    float skybox_brightness;         /* SKYBOX_BRIGHTNESS                       (Constants.h:13)    */
    float dof_intensity;             /* DEPTH_OF_FIELD_INTENSITY                (Constants.h:14)    */
    float focus_distance;            /* FOCUS_DISTANCE                          (Constants.h:15)    */
    int32_t device;                  /* HIP device ordinal                                           */
    int32_t max_batch_paths;         /* ray-queue capacity in paths (pixel samples) per batch; 0 = default */
    int32_t tile_w, tile_h;          /* image-space tile used to split a frame across GPUs; 0 = 64      */
    int32_t flags;                   /* RT_FLAG_* bits, 0 by default */
} rt_config;
/* BVH_Node::Intersect (Object.cpp:241-252) descends into every box the ray hits.  For scenes too large for on-chip
 * memory the engine skips boxes that provably lie behind the closest hit found so far (same results, fewer box tests);
 * this flag makes it visit them all, as the reference does — for comparisons. */
#define RT_FLAG_NO_CULL 1
/* A scene that fits is copied into on-chip memory by every block of the traversal kernel; this flag makes the kernel read
 * it from device memory as it reads large scenes — the same results through the other variant of the kernel, for
 * comparisons and tests. */
#define RT_FLAG_NO_LDS 2
/* A scene that is read from device memory is walked near side first, the children a ray has yet to visit on a stack in
 * on-chip memory; this flag makes the kernel take the children in the reference's order, without a stack — the same
 * results through the other variant of the kernel, for comparisons and tests. */
#define RT_FLAG_NO_ORDER 4
/* Which leaves BVH_Node::Intersect tests is decided by the boxes of the trees' lowest nodes alone (every box above is a
 * union of boxes below it and the slab test is monotone in the box), so the engine walks a hierarchy of its own above
 * those nodes — fewer box tests, the same leaves, the same results.  This flag makes it walk the reference's hierarchy
 * (Object.cpp:181-225) throughout — for comparisons and tests. */
#define RT_FLAG_NO_REGROUP 8
/* A scene that is read from device memory is walked two boxes at a time: a node's two children are tested in one step
 * from one record and the nearer one that is hit is visited first.  This flag makes the kernel test one box per step, the
 * child to visit first chosen by the split axis — the same results through the other variant of the kernel, for
 * comparisons and tests. */
#define RT_FLAG_NO_PAIRS 16
/* In a scene held in on-chip memory a node of the engine's own hierarchy may list more than two children: a box that
 * nearly fills its parent's is left out and its children are tested in its place (fewer box tests, the same leaves).
 * This flag keeps every node at two children — for comparisons and tests. */
#define RT_FLAG_NO_WIDE 32
void rt_config_defaults(rt_config* cfg);   /* the values of Constants.h:4-17 */

/* ---------------------------------------------------------------------------------------------------------
 * Scene construction — replaces the object graph built in World.cpp:15-45 and the constructors of Object.h.
 * Objects get stable ids in creation order (BVH'd objects first as added, planes continue the numbering).
 * ------------------------------------------------------------------------------------------------------- */
int rt_scene_create(rt_scene** out);
void rt_scene_destroy(rt_scene* s);
/* Texture::CreateColored / CreateCheckered / CreateFromImage (DataUtility.h:70-72).  Image data: RGBA8, rows as
 * stb_image returns them after stbi_set_flip_vertically_on_load(true) (DataUtility.cpp:99-102). Return texture id. */
int rt_scene_add_texture_color(rt_scene* s, const float rgb[3]);
int rt_scene_add_texture_checker(rt_scene* s, const float rgb1[3], const float rgb2[3]);
int rt_scene_add_texture_image(rt_scene* s, const uint8_t* rgba, int w, int h);
int rt_scene_add_texture_file(rt_scene* s, const char* png_path);     /* decodes a PNG itself, flipped, forced RGBA */
/* Material::Create* (DataUtility.h:114-117). Returns material id. */
int rt_scene_add_material(rt_scene* s, int type, int texture, const float emit[3]);
/* Sphere (Object.h:19), AxisAlignedCube (Object.h:29-30), YPlane (Object.h:40). Return object id. */
int rt_scene_add_sphere(rt_scene* s, const float center[3], float radius, int material);
int rt_scene_add_cube_center(rt_scene* s, const float center[3], float half, int material);
int rt_scene_add_cube_minmax(rt_scene* s, const float mn[3], const float mx[3], int material);
int rt_scene_add_plane(rt_scene* s, float y, int material);
/* ApplyMovement(offset, ApplyX/Y/ZRotation(degrees, PolygonMesh(obj, size, material))) (World.cpp:26, Object.h:93-156).
 * verts8: n_tri*3 vertices of (position3, texcoord2, normal3) as LoadOBJ yields them (Object.cpp:349-435).
 * rot_axis: 0/1/2 = X/Y/Z, -1 = no rotation wrapper.  has_move 0 = no movement wrapper. */
int rt_scene_add_mesh(rt_scene* s, const float* verts8, int n_tri, float size, int material,
                      int rot_axis, float degrees, int has_move, const float offset[3]);
int rt_scene_add_mesh_file(rt_scene* s, const char* obj_path, float size, int material,
                           int rot_axis, float degrees, int has_move, const float offset[3]);
/* Fog (Object.h:48-57, Object.cpp:558-599): a constant-density volume in the cube centre +- radius; `material` must be
 * Isotropic (the reference's Fog makes its own from a texture).  Optional rotation / movement wrappers as for meshes
 * (World.cpp:40: ApplyXRotation(20, Fog(...))). */
int rt_scene_add_fog(rt_scene* s, const float center[3], float radius, float density, int material,
                     int rot_axis, float degrees, int has_move, const float offset[3]);
/* Skybox (DataUtility.cpp:17-28): RGB8, flipped rows. */
int rt_scene_set_skybox(rt_scene* s, const uint8_t* rgb, int w, int h);
int rt_scene_set_skybox_file(rt_scene* s, const char* png_path);
/* The scene of World.cpp:15-32 with assets from asset_dir ("<asset_dir>/src/stb_image/tree.obj" ...). */
int rt_scene_add_default_objects(rt_scene* s, const char* asset_dir, float time);
/* The scene of World.cpp:34-40 (the reference's "#else" branch): four slabs, a light that moves with `time`, a fog
 * volume rotated about X, the plane of World.cpp:17. */
int rt_scene_add_alt_objects(rt_scene* s, const char* asset_dir, float time);
/* BVH_Node::BVH_Node (Object.cpp:181-225) for every mesh and for the object list, in the reference's order, with the
 * keyed counter RNG standing in for std::rand; then flattening into the device layout.  Call once. */
int rt_scene_build(rt_scene* s, uint32_t build_key);

/* Introspection used by the parity tests (same shapes as the oracle's dumps): preorder inner nodes, child >= 0 =
 * inner index, child < 0 = ~leaf slot in first-visit order. */
int rt_scene_dump_top(const rt_scene* s, int max_nodes, float* boxes6, int32_t* left, int32_t* right,
                      int32_t* leaf_ids, int32_t* n_leaves);
int rt_scene_dump_mesh(const rt_scene* s, int object_id, int max_nodes, float* boxes6, int32_t* left, int32_t* right,
                       int max_tris, float* tris24, int32_t* n_tris);
int rt_scene_dump_objects(const rt_scene* s, int max_obj, float* out20);
int rt_scene_dump_instance(const rt_scene* s, int object_id, float* out17);
/* sizes of the flattened scene: entries, triangles, instances, objects, planes, texel bytes, total device bytes */
int rt_scene_flat_info(const rt_scene* s, int64_t out7[7]);
/* the regrouped layout (RT_FLAG_NO_REGROUP): out8 = { has one, trees regrouped, lowest nodes in them, expected slab tests of
 * a random line through the top-level tree and through the meshes' trees in the reference's layout, the same two in the
 * regrouped layout, stack slots the ordered traversal needs there }; no counterpart in the reference */
int rt_scene_regroup_info(const rt_scene* s, double out8[8]);
/* the regrouped layout with wide nodes (RT_FLAG_NO_WIDE): out4 = { has one, entries, expected slab tests of a random line
 * through the meshes' trees, most children under one node }; no counterpart in the reference */
int rt_scene_wide_info(const rt_scene* s, double out4[4]);
/* the flattened entries of the reference's layout (regrouped = 0), the regrouped one (1) or the one with wide nodes (2): 8 floats per entry (box minimum or
 * leaf data, link; box maximum, kind — integers as bit patterns), the same leaf's entry in the other layout (-1: none) and
 * a leaf's rank in the reference's visiting order; returns the number of entries (all arrays null: only that) */
int rt_scene_dump_entries(const rt_scene* s, int regrouped, int max_entries, float* lo_hi8, int32_t* peer, uint32_t* rank);

/* Host asset decoders on their own (Object.cpp:349-435 LoadOBJ; DataUtility.cpp:17-28,98-109 image loads). */
int rt_load_obj(const char* path, int max_verts, float* verts8);              /* returns vertex count or <0 */
int rt_load_png(const char* path, int channels, int flip, int64_t max_bytes, uint8_t* out, int32_t* w, int32_t* h,
                int32_t* file_channels);                                      /* returns byte count or <0 */

/* ---------------------------------------------------------------------------------------------------------
 * Camera (DataUtility.h:136-154, DataUtility.cpp:10-15) and the front end's input mapping (App.cpp:98,114-137).
 * cam12 = pos, forward, u, v.
 * ------------------------------------------------------------------------------------------------------- */
void rt_camera_look(const float pos[3], const float look[3], float cam12[12]);
void rt_camera_input(float cam12[12], float xrel, float yrel, int key_w, int key_a, int key_s, int key_d);

/* ---------------------------------------------------------------------------------------------------------
 * Engine — replaces FrameManager::StartNewFrame/ThreadWork (FrameManager.cpp:7-30,78-122) and everything
 * World::CalculateColorForScreenPosition calls (World.cpp:68-193).
 * ------------------------------------------------------------------------------------------------------- */
int rt_engine_create(const rt_config* cfg, rt_engine** out);
void rt_engine_destroy(rt_engine* e);
int rt_engine_set_scene(rt_engine* e, const rt_scene* s);              /* uploads the flattened scene */
int rt_engine_set_camera(rt_engine* e, const float cam12[12]);
int rt_engine_get_config(const rt_engine* e, rt_config* out);

/* Render this rank's share of a frame, asynchronously on `stream` (a hipStream_t, NULL = the engine's own stream).
 * frame_seed keys the per-(pixel, sample) RNG streams.  The image is cut into tile_w x tile_h tiles numbered row-major;
 * rank r of n renders tiles r, r+n, r+2n, ...
 *   n_ranks == 1: d_rgba (device pointer, NULL = the engine's own frame buffer) receives the whole frame:
 *                 width*height*4 bytes, [A=255,B,G,R] per pixel, row-major, top row first (FrameManager.cpp:89-94).
 *   n_ranks  > 1: d_rgba (required: the engine's own frame buffer may be smaller than a rank's padded tiles) receives
 *                 rt_engine_tile_bytes() bytes: this rank's tiles packed one after another, each tile_h x tile_w x 4;
 *                 slots beyond the rank's tile count stay zero.
 * Rendering allocates nothing: queues and counters are sized when the engine is created. */
int rt_engine_render(rt_engine* e, uint32_t frame_seed, int rank, int n_ranks, uint8_t* d_rgba, void* stream);
int64_t rt_engine_tile_bytes(const rt_engine* e, int n_ranks);         /* per-rank packed size, equal on all ranks */
/* Restrict following renders to the pixel rectangle [x0,x1) x [y0,y1) of the frame (tiles are then cut from the window;
 * pixels outside keep their previous bytes).  Pixels keep their frame coordinates, so rays and RNG keys do not change —
 * the counterpart of calling CalculateColorForScreenPosition (World.cpp:68) for a sub-range of pixels. */
int rt_engine_set_window(rt_engine* e, int x0, int y0, int x1, int y1);
/* Scatter gathered tiles (n_ranks packed buffers back to back, as an all-gather / gather leaves them) into a frame. */
int rt_engine_assemble(rt_engine* e, const uint8_t* d_gathered, int n_ranks, uint8_t* d_rgba, void* stream);
/* Waits for everything the engine has queued: its own stream and, when the latest rt_engine_render / rt_engine_assemble
 * was given a stream of the caller's, that stream as well.  rt_engine_poll, rt_engine_read_frame, rt_engine_get_stats
 * and rt_engine_set_scene follow the same rule.  A caller that spreads work over several streams of its own orders
 * them itself.  The engine keeps the handle of a caller's stream from the render / assemble / clear_frame that named it
 * until the first of these waits has returned, and not beyond: the stream must stay valid until then (or until
 * rt_engine_destroy, which waits too), and may be destroyed afterwards. */
int rt_engine_sync(rt_engine* e);
/* 1 when everything queued on the engine's stream has finished, 0 while work is still running — what the front end's
 * FrameManager::NeedUpdatingTexture (FrameManager.cpp:40-48) learns from its per-thread flags.  Never blocks. */
int rt_engine_poll(rt_engine* e);
/* FrameManager::GetTexturePixelsToPresent (FrameManager.cpp:60-71): waits, then copies the engine's own frame buffer. */
int rt_engine_read_frame(rt_engine* e, uint8_t* rgba_out);
/* The pixels [x0,x1) x [y0,y1) of the engine's own frame buffer, rows of (x1 - x0) * 4 bytes one after another: what
 * World::CalculateColorForScreenPosition (World.cpp:68-102, FrameManager.cpp:89-94) yields for those pixels.  Waits like
 * rt_engine_read_frame. */
int rt_engine_read_rect(rt_engine* e, int x0, int y0, int x1, int y1, uint8_t* rgba_out);
/* The same frame without a second copy: a pointer to the engine's own page-locked host buffer (width*height*4 bytes, same
 * layout), valid until the next rt_engine_map_frame or rt_engine_destroy.  wait != 0: the complete frame, after everything
 * queued has finished; the device-to-host copy runs on the render's stream right behind it.  wait == 0: the frame buffer
 * as it stands while the render goes on (FrameManager.cpp:60-71 presents a frame in progress the same way): finished
 * batches show their pixels, the rest what the buffer held before. */
int rt_engine_map_frame(rt_engine* e, int wait, const uint8_t** rgba);
/* memset(texturePixels, 0) of FrameManager::StartNewFrame (FrameManager.cpp:13) on the engine's own frame buffer, in
 * stream order (NULL = the engine's stream). */
int rt_engine_clear_frame(rt_engine* e, void* stream);
/* FrameManager::TerminateAllThreads (FrameManager.cpp:31-38, state = Quit) without the wait: of the render in flight (and
 * only that one) the launch that is running finishes, every later launch leaves at once, and batches that have not been
 * resolved keep what the frame buffer held.  Returns immediately; rt_engine_sync then returns after what is left of one
 * launch (milliseconds; tens of milliseconds for a 2^27-path batch of a 100 000-primitive scene).  A render started
 * afterwards is not affected. */
int rt_engine_cancel(rt_engine* e);
/* Per-sample outputs (parity/debug).  rt_engine_keep_samples(e, 1) makes following renders record them; they can be read
 * after an rt_engine_render with n_ranks == 1 on a frame that fits one batch:
 * radiance[3] per sample, primary-hit object id (-1 = sky), primary HitInfo.distance, RNG draws.  Any pointer may be NULL.
 * Sample order: pixel-major (row-major pixels), sample index sx*S+sy within the pixel. */
int rt_engine_keep_samples(rt_engine* e, int enable);
int rt_engine_read_samples(rt_engine* e, float* radiance3, int32_t* primary_id, float* primary_t, uint32_t* rng_draws);

/* Closest-hit queries for arbitrary rays — World.cpp:105-119 (BVH + plane list merge).  Host arrays of n rays.
 * hit_id: object id or -1.  rec9: distance, point[3], normal[3], uv[2]; for a miss distance holds what the reference's
 * HitInfo.distance holds (FLT_MAX or a stale value, Object.cpp:17-19,534-550). */
int rt_engine_trace(rt_engine* e, int n, const float* origins3, const float* dirs3, int32_t* hit_id, float* rec9);
/* rt_engine_trace for callers that cannot allocate device memory (code that runs under HIP graph capture elsewhere in the
 * process, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and
 * results as rt_engine_trace, at most RT_TRACE_STATIC_MAX_RAYS rays per call (more: RT_E_INVALID).  The rays and results
 * are staged in a static 2 MiB device array that every GPU the library runs on holds until the process ends; calls on
 * the same GPU share it, so calls that overlap on one GPU (from several engines or threads) race and must be serialised
 * by the caller.  Like rt_engine_trace it first waits for what the engine has queued.  Calls no hipMalloc / hipFree and no
 * allocating library routine: the caller's arrays are copied through a page-locked mirror of the staging array that the
 * engine allocated when it was created, so every transfer is between page-locked and device memory. */
#define RT_TRACE_STATIC_MAX_RAYS 32768
int rt_engine_trace_static(rt_engine* e, int n, const float* origins3, const float* dirs3, int32_t* hit_id, float* rec9);
/* World::GetRayColor (World.cpp:104-171) for arbitrary rays, one RNG key per ray. */
int rt_engine_ray_color(rt_engine* e, int n, const float* origins3, const float* dirs3, int bounces, const uint32_t* keys, float* rgb3);

/* Counters of the last render (always kept: they are the queue counters of the passes): paths, rays (closest-hit queries), rays per bounce level [0..32). */
/* box_tests: slab tests (BoundingBox::DoesRayHit, DataUtility.cpp:35-82) the engine executed — with distance culling fewer
 * than the reference algorithm performs for the same rays. */
typedef struct rt_stats { uint64_t paths, rays; uint64_t rays_at_depth[33]; uint64_t box_tests; } rt_stats;
int rt_engine_get_stats(rt_engine* e, rt_stats* out);
/* Time of the dominant kernel over the last render, from HIP events recorded on the render stream around every
 * launch of it: out[0] = total ms, out[1] = launches.  Only when enabled with rt_engine_time_kernels(e, 1). */
int rt_engine_time_kernels(rt_engine* e, int enable);
int rt_engine_kernel_times(rt_engine* e, double out2[2]);

/* Device self-checks used by the tests: the portable sinf / atan2f and the RNG evaluated on the GPU. */
int rt_engine_eval_math(rt_engine* e, int n, const float* x, const float* y, float* sin_x, float* atan2_yx);
int rt_engine_eval_rng(rt_engine* e, int n, const uint32_t* key, const uint32_t* ctr, uint32_t* rand31, float* unit_distance);
/* The slab test divides through a per-ray reciprocal with two FMA corrections instead of the hardware's IEEE sequence.
 * This compares the two on the GPU: the n given pairs (a / b) plus threads x rounds x 4 generated pairs whose quotients sit
 * next to rounding boundaries.  *mismatches = pairs whose bit patterns differ; first_bad4 = (a, b, got, IEEE). */
int rt_engine_check_division(rt_engine* e, int n, const float* a, const float* b, uint32_t seed, int threads, int rounds,
                             uint64_t* mismatches, float* first_bad4);

/* ---------------------------------------------------------------------------------------------------------
 * Several GPUs behind one handle, in one process — for a front end that owns one FrameManager (App.h:27) and should use
 * a whole node: FrameManager.cpp:17-21 splits the frame over THREAD_COUNT threads, this splits it over n_dev devices.
 * The scene is replicated; device k renders tiles k, k+n, ...; the first device of the list gathers the others' tiles
 * with peer-to-peer copies and holds the frame.  All calls queue work and return, except sync and map_frame(wait);
 * frames may be queued one behind the other without waiting in between (a device's next render waits on the GPU until
 * the display device has taken its tiles of the frame before).
 * A device may be listed more than once.  rt_multi_engine(m, k) gives engine k (statistics, queries); engine 0 is the
 * display engine, whose own frame buffer receives the frame.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_config* cfg, const int32_t* device_ids, int n_dev, rt_multi** out);   /* cfg->device is ignored */
void rt_multi_destroy(rt_multi* m);
int rt_multi_count(const rt_multi* m);
rt_engine* rt_multi_engine(rt_multi* m, int k);
int rt_multi_set_scene(rt_multi* m, const rt_scene* s);
int rt_multi_set_camera(rt_multi* m, const float cam12[12]);
int rt_multi_render(rt_multi* m, uint32_t frame_seed);               /* FrameManager::StartNewFrame, FrameManager.cpp:7-30 */
int rt_multi_poll(rt_multi* m);                                      /* FrameManager::NeedUpdatingTexture, :40-48 */
int rt_multi_sync(rt_multi* m);
int rt_multi_cancel(rt_multi* m);                                    /* FrameManager::TerminateAllThreads, :31-38 */
int rt_multi_clear_frame(rt_multi* m);                               /* FrameManager.cpp:13 */
int rt_multi_map_frame(rt_multi* m, int wait, const uint8_t** rgba); /* FrameManager::GetTexturePixelsToPresent, :60-71 */

#ifdef __cplusplus
}
#endif
#endif
