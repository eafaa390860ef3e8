// TEST INFRASTRUCTURE — not part of the shipped renderer.
//
// Harness that is compiled TOGETHER WITH the unmodified reference sources where they lie
// (/root/reference/Raytracer/src/{World,Object,DataUtility,FrameManager}.cpp and
// stb_image/stb_image.cpp; see oracle/Makefile) into oracle/_ref/libref.so.  It exposes the
// reference's own classes through a plain C interface so that Python (oracle/gen_golden.py,
// tests/) can drive them: build the reference World, walk/dump its BVH, call the reference's
// Intersect functions and World::GetRayColor, and run the reference's FrameManager headless.
//
// Nothing of the reference is copied here.  The only restated piece is the 25-line sample loop of
// World::CalculateColorForScreenPosition (World.cpp:68-102), needed because the reference keeps
// W/H/S/B as compile-time macros (Constants.h:4-17) and because a per-sample RNG key has to be
// installed between samples; ref_check_pixel_loop() below validates that restatement bit-for-bit
// against the reference's own function at the reference's own constants.
//
// Determinism: the reference draws from std::rand() (DataUtility.h:11-12).  This file defines
// rand() itself (the library is linked -Bsymbolic so the reference objects bind to it) and backs
// it with a keyed counter generator, thread_local, with glibc's RAND_MAX (2^31-1).
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <typeinfo>
#include <vector>
#include <glm.hpp>
#include <gtx/compatibility.hpp>

#define private public
#include "World.h"
#include "FrameManager.h"
#undef private
#include "stb_image/stb_image.h"

// ---------------------------------------------------------------------------------------------
// keyed counter RNG behind rand()
// ---------------------------------------------------------------------------------------------
static inline uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
static inline uint32_t rand31(uint32_t key, uint32_t ctr) { return fmix32(key ^ (ctr * 0x9E3779B9u)) >> 1; }
static inline uint32_t sample_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = fmix32(seed ^ 0xA511E9B3u);
    k = fmix32(k + pixel * 0x9E3779B1u + 0x68E31DA4u);
    k = fmix32(k ^ (sample * 0x85EBCA77u + 0xC2B2AE3Du));
    return k;
}

static thread_local uint32_t g_key = 0, g_ctr = 0;
static std::atomic<uint64_t> g_calls{0};
static thread_local uint64_t t_calls = 0;

extern "C" int rand(void) noexcept {
    ++t_calls;
    return (int)rand31(g_key, g_ctr++);
}

// glm::vec3 World::GetSkyboxPixel etc. are private; "#define private public" above opens them.

namespace {

struct RefWorld {
    World* world = nullptr;
    bool custom = false;
    std::map<const Object*, int> ids;   // custom scenes: creation order
    std::vector<std::shared_ptr<Texture>> keepTex;
};

// Stable object IDs of the reference's default scene (World.cpp:24-32 creation order):
// 0 cube, 1 checker sphere, 2 tree mesh, 3+i sphere i, 18 plane, -1 none.
int default_id(const Object* o) {
    if (!o) return -1;
    if (dynamic_cast<const AxisAlignedCube*>(o)) return 0;
    if (dynamic_cast<const PolygonMesh*>(o) || dynamic_cast<const ApplyMovement*>(o) ||
        dynamic_cast<const ApplyXRotation*>(o) || dynamic_cast<const Triangle*>(o)) return 2;
    if (auto* s = dynamic_cast<const Sphere*>(o)) {
        if (s->pos.x == 8.0f) return 1;
        return 3 + (int)std::lround((s->pos.x - 10.0f) / 2.5f);
    }
    if (dynamic_cast<const YPlane*>(o)) return 18;
    return -2;
}

// unwrap transform wrappers down to the mesh (custom scenes register the wrapper and the mesh)
int object_id(const RefWorld* rw, const Object* o) {
    if (!o) return -1;
    if (!rw->custom) return default_id(o);
    auto it = rw->ids.find(o);
    return it == rw->ids.end() ? -2 : it->second;
}

struct Cfg { int W, H, S, B; float fov, dof, focus; };

// The sample loop of World::CalculateColorForScreenPosition (World.cpp:68-92) with run-time
// W/H/S/B/FOV/DoF/focus, one RNG key per sample.  Everything below GetRayColor is the reference.
inline Ray camera_ray(World* w, const Cfg& c, int px, int py, int sx, int sy) {
    Camera& camera = w->camera;
    glm::vec2 onePixelOffset = { -(1.0f / c.W) * ((float)c.W / c.H) * c.fov, -(1.0f / c.H) * c.fov };
    glm::vec2 offset = { -((float)px / c.W - 0.5f) * ((float)c.W / c.H) * c.fov, -((float)py / c.H - 0.5f) * c.fov };
    Ray ray;
    ray.direction = camera.GetForwardVector();
    ray.direction += camera.GetUDirection() * (offset.x + onePixelOffset.x * ((float)sx / c.S))
                   + camera.GetVDirection() * (offset.y + onePixelOffset.y * ((float)sy / c.S));
    ray.direction = glm::normalize(ray.direction);
    glm::vec2 lens = GetRandomUnitCirclePoint() * c.dof;
    glm::vec3 worldOffset = camera.GetUDirection() * lens.x + camera.GetVDirection() * lens.y;
    ray.pos = camera.pos;
    ray.pos += worldOffset;
    ray.direction -= worldOffset / c.focus;
    ray.direction = glm::normalize(ray.direction);
    return ray;
}

inline void resolve_pixel(glm::vec3 color, int S, uint8_t* out4) {
    color /= S * S;
    const float steepness = 5.0f;
    color = 1.0f - 1.0f / (1.0f + steepness * color);
    glm::u8vec3 c8 = { (char)(color.r * 255), (char)(color.g * 255), (char)(color.b * 255) };
    glm::vec3 col = c8;                       // FrameManager.cpp:89
    out4[0] = 255; out4[1] = (uint8_t)col.b; out4[2] = (uint8_t)col.g; out4[3] = (uint8_t)col.r;  // :91-94
}

// closest hit exactly as World::GetRayColor finds it (World.cpp:105-119), calling the reference's
// Intersect functions.  Returns hit flag; hitInfo as the reference leaves it.
inline bool closest_hit(World* w, const Ray& ray, HitInfo& hitInfo) {
    bool hit = w->rootNode->Intersect(ray, hitInfo);
    HitInfo thisHitInfo;
    for (size_t i = 0; i < w->noBoundingBoxObjects.size(); i++) {
        if (!w->noBoundingBoxObjects[i]->Intersect(ray, thisHitInfo)) continue;
        if (thisHitInfo.distance < hitInfo.distance) { hit = true; hitInfo = thisHitInfo; }
    }
    return hit;
}

struct NodeDump {
    std::vector<float> boxes;      // 6 per inner node, preorder
    std::vector<int> left, right;  // >=0 inner preorder index, <0: ~leaf slot
    std::vector<const Object*> leaves;  // leaf slots in first-visit order
    std::map<const Object*, int> slot;
};
int dump_rec(const Object* o, NodeDump& d) {
    if (auto* n = dynamic_cast<const BVH_Node*>(o)) {
        int me = (int)d.left.size();
        d.left.push_back(0); d.right.push_back(0);
        for (int c = 0; c < 3; c++) d.boxes.push_back(n->boundingBox.minCoord[c]);
        for (int c = 0; c < 3; c++) d.boxes.push_back(n->boundingBox.maxCoord[c]);
        int l = dump_rec(n->left, d);
        int r = dump_rec(n->right, d);
        d.left[me] = l; d.right[me] = r;
        return me;
    }
    auto it = d.slot.find(o);
    int s;
    if (it == d.slot.end()) { s = (int)d.leaves.size(); d.slot[o] = s; d.leaves.push_back(o); }
    else s = it->second;
    return ~s;
}

const PolygonMesh* find_mesh(const Object* o) {
    if (auto* m = dynamic_cast<const PolygonMesh*>(o)) return m;
    if (auto* a = dynamic_cast<const ApplyMovement*>(o)) return find_mesh(a->target);
    if (auto* a = dynamic_cast<const ApplyXRotation*>(o)) return find_mesh(a->target);
    if (auto* a = dynamic_cast<const ApplyYRotation*>(o)) return find_mesh(a->target);
    if (auto* a = dynamic_cast<const ApplyZRotation*>(o)) return find_mesh(a->target);
    return nullptr;
}

}  // namespace

std::vector<Vertex> LoadOBJ(std::istream& in);   // Object.cpp:349 (external linkage)

extern "C" {

void ref_rng_set(uint32_t key, uint32_t ctr) { g_key = key; g_ctr = ctr; }
uint32_t ref_rng_ctr(void) { return g_ctr; }
uint64_t ref_rng_calls(void) { return t_calls; }
uint32_t ref_rand31(uint32_t key, uint32_t ctr) { return rand31(key, ctr); }
uint32_t ref_sample_key(uint32_t seed, uint32_t pixel, uint32_t sample) { return sample_key(seed, pixel, sample); }
float ref_random01(void) { return Random01(); }                      // DataUtility.h:11
float ref_random_unit_distance(void) { return RandomUnitDistance(); }  // DataUtility.h:12
void ref_unit_sphere_point(float* out3) { glm::vec3 p = GetRandomUnitSpherePoint(); out3[0] = p.x; out3[1] = p.y; out3[2] = p.z; }
void ref_unit_circle_point(float* out2) { glm::vec2 p = GetRandomUnitCirclePoint(); out2[0] = p.x; out2[1] = p.y; }

// ---- world life cycle -------------------------------------------------------------------------
// cwd must be the reference's Raytracer/ directory (asset paths are relative, World.cpp:26,49).
void* ref_world_new(uint32_t build_key) {
    ref_rng_set(build_key, 0);
    auto* rw = new RefWorld();
    rw->world = new World(0.0f);
    return rw;
}
void ref_world_free(void* h) {
    auto* rw = (RefWorld*)h;
    delete rw->world;
    delete rw;
}
void ref_world_set_camera(void* h, const float* pos, const float* look) {
    Camera& cam = ((RefWorld*)h)->world->camera;
    cam.pos = { pos[0], pos[1], pos[2] };
    cam.SetForwardVector({ look[0], look[1], look[2] });
}
void ref_world_get_camera(void* h, float* out12) {
    Camera& cam = ((RefWorld*)h)->world->camera;
    glm::vec3 v[4] = { cam.pos, cam.GetForwardVector(), cam.GetUDirection(), cam.GetVDirection() };
    for (int i = 0; i < 4; i++) for (int c = 0; c < 3; c++) out12[i * 3 + c] = v[i][c];
}
// Mouse / key handling of the front end (App.cpp:98,114-137) applied to the reference's Camera.
void ref_world_camera_input(void* h, float xrel, float yrel, int key_w, int key_a, int key_s, int key_d) {
    Camera& camera = ((RefWorld*)h)->world->camera;
    if (xrel != 0.0f || yrel != 0.0f)
        camera.SetForwardVector(camera.GetForwardVector() - MOUSE_SENSITIVITY * (xrel * camera.GetUDirection() + yrel * camera.GetVDirection()));
    if (key_d) { glm::vec3 dir = camera.GetUDirection(); dir.y = 0; camera.pos -= dir * (float)MOVEMENT_SPEED; }
    if (key_a) { glm::vec3 dir = camera.GetUDirection(); dir.y = 0; camera.pos += dir * (float)MOVEMENT_SPEED; }
    if (key_w) { glm::vec3 dir = camera.GetForwardVector(); dir.y = 0; camera.pos += dir * (float)MOVEMENT_SPEED; }
    if (key_s) { glm::vec3 dir = camera.GetForwardVector(); dir.y = 0; camera.pos -= dir * (float)MOVEMENT_SPEED; }
}

// Replace the world's scene by a custom one built from the reference's own object types.
// tex[i]  : kind (0 colour, 1 checker, 2 image file "src/stb_image/tree_texture.png"), c1[3], c2[3]   (7 floats, kind as float)
// mat[i]  : type (MaterialType enum value), tex index, emit[3]                                         (5 floats)
// obj[i]  : kind (0 sphere c,r | 1 cube min,max | 2 cube centre,half | 3 tree mesh: size, rot_axis(-1,0,1,2), degrees, offset[3], has_move),
//           8 floats of params, then material index                                                   (10 floats)
// plane[i]: y, material index                                                                          (2 floats)
void ref_world_set_custom(void* h, uint32_t build_key, int n_tex, const float* tex, int n_mat, const float* mat,
                          int n_obj, const float* obj, int n_plane, const float* plane) {
    auto* rw = (RefWorld*)h;
    World* w = rw->world;
    for (Object* o : w->noBoundingBoxObjects) delete o;
    w->noBoundingBoxObjects.clear();
    delete w->rootNode;
    rw->custom = true;
    rw->ids.clear();
    ref_rng_set(build_key, 0);

    std::vector<std::shared_ptr<Texture>> texs;
    for (int i = 0; i < n_tex; i++) {
        const float* t = tex + 7 * i;
        glm::vec3 c1{ t[1], t[2], t[3] }, c2{ t[4], t[5], t[6] };
        int kind = (int)t[0];
        if (kind == 0) texs.push_back(Texture::CreateColored(c1));
        else if (kind == 1) texs.push_back(Texture::CreateCheckered(c1, c2));
        else texs.push_back(Texture::CreateFromImage("src/stb_image/tree_texture.png"));
    }
    std::vector<Material> mats;
    for (int i = 0; i < n_mat; i++) {
        const float* m = mat + 5 * i;
        Material mm;
        mm.materialType = (MaterialType)(int)m[0];
        if ((int)m[1] >= 0) mm.texture = texs[(int)m[1]];
        mm.emittingColor = { m[2], m[3], m[4] };
        mats.push_back(mm);
    }
    std::vector<Object*> objects;
    for (int i = 0; i < n_obj; i++) {
        const float* o = obj + 10 * i;
        int kind = (int)o[0];
        const Material& m = mats[(int)o[9]];
        Object* made = nullptr;
        if (kind == 0) made = new Sphere{ glm::vec3{ o[1], o[2], o[3] }, o[4], m };
        else if (kind == 1) made = new AxisAlignedCube{ glm::vec3{ o[1], o[2], o[3] }, glm::vec3{ o[4], o[5], o[6] }, m };
        else if (kind == 2) made = new AxisAlignedCube{ glm::vec3{ o[1], o[2], o[3] }, o[4], m };
        else if (kind == 4) {   // Fog(pos, radius, density, texture of material), rotation axis o[6] by o[7] degrees, moved by o[8] in x
            Object* fog = new Fog(glm::vec3{ o[1], o[2], o[3] }, o[4], o[5], m.texture);
            rw->ids[fog] = i;
            Object* cur = fog;
            int axis = (int)o[6];
            if (axis == 0) cur = new ApplyXRotation(o[7], cur);
            else if (axis == 1) cur = new ApplyYRotation(o[7], cur);
            else if (axis == 2) cur = new ApplyZRotation(o[7], cur);
            rw->ids[cur] = i;
            if (o[8] != 0.0f) cur = new ApplyMovement({ o[8], 0.0f, 0.0f }, cur);
            made = cur;
        }
        else {
            Object* mesh = new PolygonMesh("src/stb_image/tree.obj", o[1], m);
            rw->ids[mesh] = i;
            Object* cur = mesh;
            int axis = (int)o[2];
            if (axis == 0) cur = new ApplyXRotation(o[3], cur);
            else if (axis == 1) cur = new ApplyYRotation(o[3], cur);
            else if (axis == 2) cur = new ApplyZRotation(o[3], cur);
            rw->ids[cur] = i;
            if (o[7] != 0.0f) cur = new ApplyMovement({ o[4], o[5], o[6] }, cur);
            made = cur;
        }
        rw->ids[made] = i;
        objects.push_back(made);
    }
    w->rootNode = new BVH_Node(objects);
    for (int i = 0; i < n_plane; i++) {
        Object* p = new YPlane(plane[2 * i], mats[(int)plane[2 * i + 1]]);
        rw->ids[p] = n_obj + i;
        w->noBoundingBoxObjects.push_back(p);
    }
}

// ---- dumps ------------------------------------------------------------------------------------
// Top-level BVH in preorder.  Returns node count; leaf_ids[slot] = stable object id of leaf slot.
int ref_dump_top(void* h, int max_nodes, float* boxes6, int* left, int* right, int* leaf_ids, int* n_leaves) {
    auto* rw = (RefWorld*)h;
    NodeDump d;
    dump_rec(rw->world->rootNode, d);
    int n = (int)d.left.size();
    if (n > max_nodes) return -n;
    memcpy(boxes6, d.boxes.data(), sizeof(float) * 6 * n);
    memcpy(left, d.left.data(), sizeof(int) * n);
    memcpy(right, d.right.data(), sizeof(int) * n);
    for (size_t i = 0; i < d.leaves.size(); i++) leaf_ids[i] = object_id(rw, d.leaves[i]);
    *n_leaves = (int)d.leaves.size();
    return n;
}
// Object parameters by stable id: kind (0 sphere, 1 cube, 3 mesh instance, 4 plane), 8 floats, material type,
// texture colours (kind, c1, c2).  Returns number of objects written (BVH leaves + planes).
int ref_dump_objects(void* h, int max_obj, float* out /*20 per object*/) {
    auto* rw = (RefWorld*)h;
    NodeDump d;
    dump_rec(rw->world->rootNode, d);
    std::vector<const Object*> all(d.leaves.begin(), d.leaves.end());
    for (Object* o : rw->world->noBoundingBoxObjects) all.push_back(o);
    int count = 0;
    for (const Object* o : all) {
        int id = object_id(rw, o);
        if (id < 0 || id >= max_obj) return -1;
        float* r = out + 20 * id;
        for (int i = 0; i < 20; i++) r[i] = 0;
        const Material* m = &o->material;
        if (auto* s = dynamic_cast<const Sphere*>(o)) { r[0] = 0; r[1] = s->pos.x; r[2] = s->pos.y; r[3] = s->pos.z; r[4] = s->radius; }
        else if (auto* c = dynamic_cast<const AxisAlignedCube*>(o)) { r[0] = 1; for (int k = 0; k < 3; k++) { r[1 + k] = c->minCoord[k]; r[4 + k] = c->maxCoord[k]; } }
        else if (auto* p = dynamic_cast<const YPlane*>(o)) { r[0] = 4; r[1] = p->yPos; }
        else {
            r[0] = 3;
            BoundingBox b; o->GetBoundingBox(b);
            for (int k = 0; k < 3; k++) { r[1 + k] = b.minCoord[k]; r[4 + k] = b.maxCoord[k]; }
            m = &find_mesh(o)->material;
        }
        r[9] = (float)(int)m->materialType;
        if (auto* ct = dynamic_cast<const ColorTexture*>(m->texture.get())) { r[10] = 0; for (int k = 0; k < 3; k++) r[11 + k] = ct->color[k]; }
        else if (auto* ch = dynamic_cast<const CheckeredTexture*>(m->texture.get())) { r[10] = 1; for (int k = 0; k < 3; k++) { r[11 + k] = ch->color1[k]; r[14 + k] = ch->color2[k]; } }
        else if (dynamic_cast<const ImageTexture*>(m->texture.get())) r[10] = 2;
        else r[10] = -1;
        for (int k = 0; k < 3; k++) r[17 + k] = m->emittingColor[k];
        count++;
    }
    return count;
}
// Mesh instance `id`: transform parameters.  out: offset[3], pivot[3], sin, cos, axis, rotated box[6], has_move, has_rot
int ref_dump_instance(void* h, int id, float* out17) {
    auto* rw = (RefWorld*)h;
    NodeDump d;
    dump_rec(rw->world->rootNode, d);
    for (const Object* o : d.leaves) {
        if (object_id(rw, o) != id) continue;
        for (int i = 0; i < 17; i++) out17[i] = 0;
        out17[8] = -1;
        const Object* cur = o;
        if (auto* mv = dynamic_cast<const ApplyMovement*>(cur)) { for (int k = 0; k < 3; k++) out17[k] = mv->offset[k]; out17[15] = 1; cur = mv->target; }
        auto rot = [&](const glm::vec3& pivot, float s, float c, const BoundingBox& box, int axis) {
            for (int k = 0; k < 3; k++) { out17[3 + k] = pivot[k]; out17[9 + k] = box.minCoord[k]; out17[12 + k] = box.maxCoord[k]; }
            out17[6] = s; out17[7] = c; out17[8] = (float)axis; out17[16] = 1;
        };
        if (auto* r = dynamic_cast<const ApplyXRotation*>(cur)) rot(r->pivot, r->sinTheta, r->cosTheta, r->box, 0);
        else if (auto* r = dynamic_cast<const ApplyYRotation*>(cur)) rot(r->pivot, r->sinTheta, r->cosTheta, r->box, 1);
        else if (auto* r = dynamic_cast<const ApplyZRotation*>(cur)) rot(r->pivot, r->sinTheta, r->cosTheta, r->box, 2);
        return 1;
    }
    return 0;
}
// Triangle BVH of mesh instance `id` in preorder; tris: 24 floats per leaf slot (3 x pos3,uv2,normal3).
int ref_dump_mesh(void* h, int id, int max_nodes, float* boxes6, int* left, int* right, int max_tris, float* tris24, int* n_tris) {
    auto* rw = (RefWorld*)h;
    NodeDump top;
    dump_rec(rw->world->rootNode, top);
    for (const Object* o : top.leaves) {
        if (object_id(rw, o) != id) continue;
        const PolygonMesh* mesh = find_mesh(o);
        if (!mesh) return 0;
        NodeDump d;
        dump_rec(mesh->triangles, d);
        int n = (int)d.left.size();
        if (n > max_nodes || (int)d.leaves.size() > max_tris) return -n;
        memcpy(boxes6, d.boxes.data(), sizeof(float) * 6 * n);
        memcpy(left, d.left.data(), sizeof(int) * n);
        memcpy(right, d.right.data(), sizeof(int) * n);
        for (size_t i = 0; i < d.leaves.size(); i++) {
            auto* t = dynamic_cast<const Triangle*>(d.leaves[i]);
            for (int v = 0; v < 3; v++) {
                float* r = tris24 + 24 * i + 8 * v;
                for (int k = 0; k < 3; k++) r[k] = t->vertices[v].position[k];
                r[3] = t->vertices[v].texcoord.x; r[4] = t->vertices[v].texcoord.y;
                for (int k = 0; k < 3; k++) r[5 + k] = t->vertices[v].normal[k];
            }
        }
        *n_tris = (int)d.leaves.size();
        return n;
    }
    return 0;
}
// The reference's OBJ parser (Object.cpp:349-435) on a file: 8 floats per vertex (pos3, uv2, normal3).
int ref_load_obj(const char* path, int max_verts, float* out8) {
    std::ifstream f(path);
    if (!f) return -1;
    std::vector<Vertex> v = LoadOBJ(f);
    if ((int)v.size() > max_verts) return -(int)v.size();
    for (size_t i = 0; i < v.size(); i++) {
        float* r = out8 + 8 * i;
        for (int k = 0; k < 3; k++) r[k] = v[i].position[k];
        r[3] = v[i].texcoord.x; r[4] = v[i].texcoord.y;
        for (int k = 0; k < 3; k++) r[5 + k] = v[i].normal[k];
    }
    return (int)v.size();
}
// stb_image decode exactly as the reference asks for it (DataUtility.cpp:18-21,99-102): flipped, forced channels.
int ref_load_image(const char* path, int req_channels, int max_bytes, uint8_t* out, int* w, int* hgt, int* file_channels) {
    stbi_set_flip_vertically_on_load(true);
    unsigned char* d = stbi_load(path, w, hgt, file_channels, req_channels);
    if (!d) return -1;
    int n = (*w) * (*hgt) * req_channels;
    if (n <= max_bytes) memcpy(out, d, n);
    stbi_image_free(d);
    return n;
}
void ref_skybox_info(void* h, int* wh2) {
    Skybox& s = ((RefWorld*)h)->world->skybox;
    wh2[0] = (int)s.size.x; wh2[1] = (int)s.size.y;
}

// ---- primitive-level calls ----------------------------------------------------------------------
// BoundingBox::DoesRayHit (DataUtility.cpp:35-82); directions used as given.
void ref_aabb_hit(const float* box6, int n, const float* o3, const float* d3, uint8_t* out) {
    BoundingBox b({ box6[0], box6[1], box6[2] }, { box6[3], box6[4], box6[5] });
    for (int i = 0; i < n; i++) {
        Ray r({ o3[3 * i], o3[3 * i + 1], o3[3 * i + 2] }, { d3[3 * i], d3[3 * i + 1], d3[3 * i + 2] });
        out[i] = b.DoesRayHit(r) ? 1 : 0;
    }
}
// the Main.cpp:19 helper normalises before the call
void ref_normalize(int n, const float* v3, float* out3) {
    for (int i = 0; i < n; i++) {
        glm::vec3 v = glm::normalize(glm::vec3{ v3[3 * i], v3[3 * i + 1], v3[3 * i + 2] });
        out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
    }
}
static void write_hit(bool hit, const HitInfo& hi, int id, int32_t* hit_id, float* rec) {
    *hit_id = hit ? id : -1;
    rec[0] = hi.distance;
    for (int k = 0; k < 3; k++) { rec[1 + k] = hi.point[k]; rec[4 + k] = hi.normal[k]; }
    rec[7] = hi.uv.x; rec[8] = hi.uv.y;
}
// kind 0 sphere(c,r) | 1 cube(min,max) | 4 plane(y).  rec: 9 floats (t, point, normal, uv) — fields the
// primitive did not write keep the fill pattern given in `fill`.
void ref_intersect_prim(int kind, const float* p, int n, const float* o3, const float* d3, float fill, int32_t* hit, float* rec9) {
    Material m;
    std::unique_ptr<Object> obj;
    if (kind == 0) obj.reset(new Sphere({ p[0], p[1], p[2] }, p[3], m));
    else if (kind == 1) obj.reset(new AxisAlignedCube(glm::vec3{ p[0], p[1], p[2] }, glm::vec3{ p[3], p[4], p[5] }, m));
    else obj.reset(new YPlane(p[0], m));
    for (int i = 0; i < n; i++) {
        Ray r({ o3[3 * i], o3[3 * i + 1], o3[3 * i + 2] }, { d3[3 * i], d3[3 * i + 1], d3[3 * i + 2] });
        HitInfo hi;
        hi.point = glm::vec3(fill); hi.normal = glm::vec3(fill); hi.uv = glm::vec2(fill);
        bool h = obj->Intersect(r, hi);
        write_hit(h, hi, 0, hit + i, rec9 + 9 * i);
    }
}
// Triangle::Intersect (Object.cpp:498-556) with the tree texture as alpha source (use_alpha) or no cut-out.
void ref_intersect_tri(const float* verts24, int use_alpha, int n, const float* o3, const float* d3, float fill, int32_t* hit, float* rec9) {
    Material m = use_alpha ? Material::CreateDiffuse(Texture::CreateFromImage("src/stb_image/tree_texture.png"))
                           : Material::CreateDiffuse(Texture::CreateColored({ 1, 1, 1 }));
    Vertex v[3];
    for (int k = 0; k < 3; k++) {
        const float* r = verts24 + 8 * k;
        v[k].position = { r[0], r[1], r[2] }; v[k].texcoord = { r[3], r[4] }; v[k].normal = { r[5], r[6], r[7] };
    }
    Triangle t(v[0], v[1], v[2], m);
    for (int i = 0; i < n; i++) {
        Ray r({ o3[3 * i], o3[3 * i + 1], o3[3 * i + 2] }, { d3[3 * i], d3[3 * i + 1], d3[3 * i + 2] });
        HitInfo hi;
        hi.point = glm::vec3(fill); hi.normal = glm::vec3(fill); hi.uv = glm::vec2(fill);
        bool h = t.Intersect(r, hi);
        write_hit(h, hi, 0, hit + i, rec9 + 9 * i);
    }
}

// ---- scene-level calls --------------------------------------------------------------------------
// Closest hit for arbitrary rays: BVH + plane list merged as World::GetRayColor does (World.cpp:105-119).
// rec9 holds what the reference's HitInfo holds afterwards (stale values included when hit==-1).
void ref_trace(void* h, int n, const float* o3, const float* d3, int32_t* hit, float* rec9) {
    auto* rw = (RefWorld*)h;
    for (int i = 0; i < n; i++) {
        Ray r({ o3[3 * i], o3[3 * i + 1], o3[3 * i + 2] }, { d3[3 * i], d3[3 * i + 1], d3[3 * i + 2] });
        HitInfo hi;
        hi.point = glm::vec3(0); hi.normal = glm::vec3(0); hi.uv = glm::vec2(0);
        ref_rng_set((uint32_t)i, 0);   // Fog tests draw random numbers: one stream per ray
        bool ok = closest_hit(rw->world, r, hi);
        write_hit(ok, hi, object_id(rw, hi.object), hit + i, rec9 + 9 * i);
    }
}
// World::GetRayColor for arbitrary rays, RNG key per ray.
void ref_ray_color(void* h, int n, const float* o3, const float* d3, int bounces, const uint32_t* keys, float* rgb) {
    World* w = ((RefWorld*)h)->world;
    for (int i = 0; i < n; i++) {
        Ray r({ o3[3 * i], o3[3 * i + 1], o3[3 * i + 2] }, { d3[3 * i], d3[3 * i + 1], d3[3 * i + 2] });
        ref_rng_set(keys[i], 0);
        glm::vec3 c = w->GetRayColor(r, bounces);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}
// World::GetSkyboxPixel(direction) (World.cpp:173-181)
void ref_skybox(void* h, int n, const float* d3, float* rgb) {
    World* w = ((RefWorld*)h)->world;
    for (int i = 0; i < n; i++) {
        glm::vec3 c = w->GetSkyboxPixel(glm::vec3{ d3[3 * i], d3[3 * i + 1], d3[3 * i + 2] });
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}

// Render the pixel rectangle [x0,x1) x [y0,y1) of a W x H frame with per-sample keys.
// cfg_i = {W,H,S,B}; cfg_f = {fov, dof, focus}.  SKYBOX_BRIGHTNESS is whatever the reference compiled in.
// Any output pointer may be null.  Per sample s = sx*S+sy of pixel p = y*W+x (rect-local order: row-major pixels):
//   radiance[3], primary_id, primary_t (HitInfo.distance after the closest-hit search), ray6 (camera ray), rng draws.
// rgba: 4 bytes per pixel of the rectangle.
void ref_render(void* h, const int* cfg_i, const float* cfg_f, uint32_t seed, int x0, int y0, int x1, int y1, int n_threads,
                float* radiance, int32_t* primary_id, float* primary_t, float* ray6, uint32_t* rng_draws, uint8_t* rgba) {
    auto* rw = (RefWorld*)h;
    World* w = rw->world;
    Cfg c{ cfg_i[0], cfg_i[1], cfg_i[2], cfg_i[3], cfg_f[0], cfg_f[1], cfg_f[2] };
    const int rw_ = x1 - x0, rh_ = y1 - y0, spp = c.S * c.S;
    const long n_pix = (long)rw_ * rh_;
    const bool need_color = radiance || rng_draws || rgba;
    auto work = [&](long p_begin, long p_end) {
        for (long lp = p_begin; lp < p_end; lp++) {
            int x = x0 + (int)(lp % rw_), y = y0 + (int)(lp / rw_);
            glm::vec3 color{ 0 };
            for (int sx = 0; sx < c.S; sx++) for (int sy = 0; sy < c.S; sy++) {
                int s = sx * c.S + sy;
                long si = lp * spp + s;
                ref_rng_set(sample_key(seed, (uint32_t)(y * c.W + x), (uint32_t)s), 0);
                Ray ray = camera_ray(w, c, x, y, sx, sy);
                if (ray6) { for (int k = 0; k < 3; k++) { ray6[6 * si + k] = ray.pos[k]; ray6[6 * si + 3 + k] = ray.direction[k]; } }
                if (primary_id || primary_t) {
                    HitInfo hi;
                    const uint32_t key_now = g_key, ctr_now = g_ctr;   // a Fog test draws random numbers: probe, then rewind
                    bool ok = closest_hit(w, ray, hi);
                    ref_rng_set(key_now, ctr_now);
                    if (primary_id) primary_id[si] = ok ? object_id(rw, hi.object) : -1;
                    if (primary_t) primary_t[si] = hi.distance;
                }
                if (!need_color) continue;
                glm::vec3 col = w->GetRayColor(ray, c.B);
                if (radiance) { radiance[3 * si] = col.x; radiance[3 * si + 1] = col.y; radiance[3 * si + 2] = col.z; }
                if (rng_draws) rng_draws[si] = g_ctr;
                color += col;
            }
            if (rgba) resolve_pixel(color, c.S, rgba + 4 * lp);
        }
    };
    if (n_threads <= 1) { work(0, n_pix); return; }
    // contiguous pixel ranges per thread, as FrameManager.cpp:80-82 splits them
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work, (long)t * n_pix / n_threads, (long)(t + 1) * n_pix / n_threads);
    for (auto& t : th) t.join();
}

// Validate the restated loop: with ONE key per pixel (counter running through the pixel's samples) compare
// World::CalculateColorForScreenPosition (the reference's own, at its compiled constants) with camera_ray +
// GetRayColor + resolve_pixel at the same constants.  Returns the number of mismatching pixels in the rectangle
// and writes the reference's own bytes to rgba (4 per pixel) when non-null.
long ref_check_pixel_loop(void* h, uint32_t seed, int x0, int y0, int x1, int y1, uint8_t* rgba) {
    World* w = ((RefWorld*)h)->world;
    Cfg c{ WINDOW_WIDTH, WINDOW_HEIGHT, SAMPLES_PER_PIXEL_AXIS, LIGHT_BOUNCE_AMOUNT, FIELD_OF_VIEW, DEPTH_OF_FIELD_INTENSITY, FOCUS_DISTANCE };
    long bad = 0, lp = 0;
    for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++, lp++) {
        uint32_t key = sample_key(seed, (uint32_t)(y * c.W + x), 0xFFFFFFFFu);
        ref_rng_set(key, 0);
        glm::u8vec3 a = w->CalculateColorForScreenPosition(x, y);
        uint8_t ra[4];
        { glm::vec3 col = a; ra[0] = 255; ra[1] = (uint8_t)col.b; ra[2] = (uint8_t)col.g; ra[3] = (uint8_t)col.r; }
        ref_rng_set(key, 0);
        glm::vec3 color{ 0 };
        for (int sx = 0; sx < c.S; sx++) for (int sy = 0; sy < c.S; sy++) {
            Ray ray = camera_ray(w, c, x, y, sx, sy);
            color += w->GetRayColor(ray, c.B);
        }
        uint8_t rb[4];
        resolve_pixel(color, c.S, rb);
        if (memcmp(ra, rb, 4) != 0) bad++;
        if (rgba) memcpy(rgba + 4 * lp, ra, 4);
    }
    return bad;
}
void ref_constants(int* i4, float* f4) {
    i4[0] = WINDOW_WIDTH; i4[1] = WINDOW_HEIGHT; i4[2] = SAMPLES_PER_PIXEL_AXIS; i4[3] = LIGHT_BOUNCE_AMOUNT;
    f4[0] = FIELD_OF_VIEW; f4[1] = DEPTH_OF_FIELD_INTENSITY; f4[2] = FOCUS_DISTANCE; f4[3] = SKYBOX_BRIGHTNESS;
}

// One frame through the reference's own scheduler (FrameManager.cpp), headless.  Returns seconds between
// StartNewFrame returning and the last worker exiting (what LOG_BENCHMARK prints); copies the W*H*4 bytes.
double ref_framemanager_frame(uint32_t build_key, uint8_t* rgba_out) {
    ref_rng_set(build_key, 0);
    auto* fm = new FrameManager();
    fm->time = 0.0f;
    ref_rng_set(build_key, 0);
    std::streambuf* old = std::cout.rdbuf(nullptr);   // silence the benchmark print
    fm->StartNewFrame();
    auto t0 = std::chrono::steady_clock::now();
    while (fm->runningThreadAmount != 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
    auto t1 = std::chrono::steady_clock::now();
    while (fm->NeedUpdatingTexture()) {}
    std::cout.rdbuf(old);
    if (rgba_out) memcpy(rgba_out, fm->GetTexturePixelsToPresent(), (size_t)WINDOW_WIDTH * WINDOW_HEIGHT * 4);
    fm->TerminateAllThreads();
    delete fm;
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
