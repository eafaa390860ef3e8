// Host-side types shared by the scene builder, the asset decoders and the engine.
#pragma once
#include <cmath>
#include <cstdint>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_flat.h"

namespace rt {

struct Vec2 { float x, y; };
struct Vec3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
struct Box3 { Vec3 mn, mx; };

// GLM 0.9.9.7 evaluation order (glm/detail/func_geometric.inl:48-90, func_common.inl:20,29): the scene constants
// computed here must carry the same bits as the reference's.
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 x, Vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline Vec3 normalize(Vec3 v) { float s = 1.0f / std::sqrt(dot(v, v)); return v * s; }
inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }

// keyed counter generator that stands in for std::rand (DataUtility.h:11-12) on the host (BVH axis choice) and on
// the device (lens and bounce sampling): value i of stream `key` is fmix32(key ^ i*0x9E3779B9) >> 1, 31 bits
// like glibc's RAND_MAX.
inline uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
inline uint32_t rand31(uint32_t key, uint32_t ctr) { return fmix32(key ^ (ctr * 0x9E3779B9u)) >> 1; }

void set_error(const std::string& msg);

// No exception may cross the C boundary: entry points whose work allocates host memory run it through guard(), which
// turns std::bad_alloc into RT_E_NOMEM (-4) and anything else into RT_E_INVALID (-1) with the message in rt_last_error().
template <typename F>
int guard(F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { set_error("out of host memory"); return -4; }
    catch (const std::exception& ex) { set_error(std::string("unexpected error: ") + ex.what()); return -1; }
    catch (...) { set_error("unexpected error"); return -1; }
}

bool load_obj(const std::string& path, std::vector<float>& verts8, std::string& err);
bool load_png(const std::string& path, int req_channels, bool flip, std::vector<uint8_t>& out, int& w, int& h,
              int& file_channels, std::string& err);

}  // namespace rt

// One arrangement of a scene's entries (layout: rt_flat.h).  A flattened scene has the reference's arrangement — every
// tree in the order BVH_Node::Intersect (Object.cpp:241-252) walks it — and may have a second one, `fast`, in which the
// levels of a tree ABOVE its lowest nodes are rebuilt (rt_scene.cpp, "regrouping"); leaves, lowest nodes, instances and
// triangles are the same objects in both.
struct rt_layout {
    std::vector<rt::F4> ent_lo, ent_hi;      // threaded preorder entries
    std::vector<rt::F4> inst;                // RT_INST_F4 float4 per mesh instance (entry ranges and stale entries are the layout's own)
    int32_t top_end = 0, stale_entry = -1, idle_entry = 0;
    // Distance culling (rt_device.h, "distance culling"): per entry, for inner entries of the top level, the largest
    // 1 / (2 radius) over the spheres below it (0 when only cubes are below), or +inf when something below may not be
    // culled (a mesh instance, a fog volume, a sphere with a negative radius, a cube with min > max).
    std::vector<float> ent_rho;
    // Ordered traversal: per inner entry the index of its second child's entry (the first child is the next entry; -1
    // when the node lists a single child) and the axis along which the first child's box starts no later than the
    // second's (3 for a single child); and the stack slots a ray can need at most.
    std::vector<int32_t> ent_right;
    std::vector<uint8_t> ent_axis;
    int32_t stack_need = 0;
    // Leaf entries (sphere, cube, fog, triangle): the entry of the same leaf in the scene's other layout (-1 when there
    // is none), and the leaf's place in the order in which the reference meets the scene's leaves (ties between equal
    // distances go to the leaf met first, Object.cpp:248).  0 / -1 for other kinds of entries.
    std::vector<int32_t> ent_peer;
    std::vector<uint32_t> ent_rank;
    std::vector<uint8_t> ent_entry_box;      // 1: the inner entry is a mesh instance's entry box (rt_scene.cpp, "entry boxes")
};

// The flattened scene as the engine uploads it: the reference's layout (the base), what both layouts share, and the
// regrouped layout when the scene has one.
struct rt_scene_flat : rt_layout {
    std::vector<rt::F4> tris;                // RT_TRI_F4 float4 per triangle
    std::vector<rt::F4> objects;             // RT_OBJ_F4 float4 per object id
    std::vector<rt::F4> planes;              // (y, object id bits, 0, 0)
    std::vector<rt::ImageRec> images;
    std::vector<uint8_t> texels;             // all images then the skybox, each padded by one row
    rt::ImageRec sky{0, 0, 0, 0};
    bool fast_div_ok = false;                // every box coordinate is 0 or has magnitude in [2^-50, 2^40] (see rt_device.h, div_by)
    float cull_delta = 0, cull_cabs = 0, cull_2rmax = 0;   // scene-wide constants of the culling bound
    float entry_guard = 3.4028234663852886e38f;   // entry boxes are proven for rays that start within this distance of the origin, per coordinate
    bool has_fast = false;                   // `fast` is filled: some tree of the scene was regrouped
    rt_layout fast;
    int32_t regrouped_trees = 0, regrouped_groups = 0;
    // The regrouped layout once more, for the kernels that hold the scene on chip and walk it by skip links alone: an
    // inner node of a rebuilt hierarchy whose box is more than half of its parent's (by surface) is left out and its
    // children are listed under the parent (rt_scene.cpp, "wide nodes").  ent_peer_wide: by entry of the reference's
    // layout, the same leaf's entry here.
    bool has_wide = false;
    rt_layout wide;
    std::vector<int32_t> ent_peer_wide;
};

const rt_scene_flat* rt_scene_get_flat(const struct rt_scene* s);
