// Host scene builder: the object graph of the reference (Object.h) as a flat pool, the reference's BVH construction
// (BVH_Node::BVH_Node, Object.cpp:181-225) driven by the keyed counter RNG, and the flattening into the stackless
// device layout described in rt_flat.h.  Also the host side of Camera (DataUtility.cpp:10-15) and the default scene
// (World.cpp:15-32).
#include "../../include/rt_engine.h"
#include "rt_host.h"

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <thread>

using namespace rt;

namespace {

thread_local std::string g_error;

enum Kind { SPHERE, CUBE, PLANE, TRI, NODE, MESH, ROT, MOVE, FOG };

struct Obj {
    Kind kind;
    int material = -1, id = -1;
    Vec3 c{0, 0, 0};
    float r = 0;                      // sphere
    float neg_inv_density = 0;        // fog (its boundary cube is `box`)
    Box3 box{{0, 0, 0}, {0, 0, 0}};   // cube extents; node / triangle / rotation box
    float y = 0;                      // plane
    Vec3 p[3]; Vec2 tc[3]; Vec3 n[3]; // triangle
    int left = -1, right = -1;        // node
    int root = -1;                    // mesh: its triangle BVH
    int target = -1, axis = -1;       // wrappers
    float sin_t = 0, cos_t = 0, degrees = 0;
    Vec3 pivot{0, 0, 0}, offset{0, 0, 0};
    std::vector<float> verts;         // mesh: parsed vertices until build
    float size = 0;
};
struct Tex { int kind; Vec3 c1, c2; int image; };
struct Mat { int type, tex; Vec3 emit; };
struct Image { std::vector<uint8_t> px; int w, h, channels; };

}  // namespace

struct rt_scene {
    std::vector<Obj> objs;
    std::vector<Tex> texs;
    std::vector<Mat> mats;
    std::vector<Image> images;
    Image sky{{}, 0, 0, 3};
    std::vector<int> top, planes;
    int root = -1, next_id = 0;
    bool built = false;
    uint32_t rng_key = 0, rng_ctr = 0;
    rt_scene_flat flat;

    float random01() { return (float)(int)rand31(rng_key, rng_ctr++) / 2147483648.0f; }   // DataUtility.h:11
};

namespace rt { void set_error(const std::string& m) { g_error = m; } }
extern "C" const char* rt_last_error(void) { return g_error.c_str(); }
const rt_scene_flat* rt_scene_get_flat(const rt_scene* s) { return s && s->built ? &s->flat : nullptr; }

namespace {

int fail(int code, const std::string& msg) { set_error(msg); return code; }

Box3 box_of(const rt_scene& s, int idx) {
    const Obj& o = s.objs[idx];
    switch (o.kind) {
    case SPHERE: {   // Object.cpp:29-35; a negative radius gives an inverted box and is kept that way
        Vec3 rr{o.r, o.r, o.r};
        return {o.c - rr, o.c + rr};
    }
    case MESH: return box_of(s, o.root);   // Object.h:97
    case MOVE: {                           // Object.cpp:610-615
        Box3 b = box_of(s, o.target);
        return {b.mn + o.offset, b.mx + o.offset};
    }
    default: return o.box;
    }
}

// BVH_Node::BVH_Node (Object.cpp:181-225).  The reference copies the object vector in every node and sorts a
// sub-range of the copy; children only touch disjoint sub-ranges, so sorting one shared vector in place is the same.
//
// The constructor draws one random number per node, parent before children, left subtree before right
// (Object.cpp:191,215-216), so the counter a node draws with is its parent's plus one plus the nodes of everything built
// before it — a function of the spans alone.  A span of 1 or 2 is one node; a span n >= 3 is one node plus the nodes of
// its halves n/2 and n - n/2.  With the counters known up front, the two halves of a large span are built by different
// threads: each draws the numbers the serial order would have drawn, sorts the same sub-range with the same comparator,
// and writes its nodes into the slots the serial order (children before parent) would have used.
struct NodeCount {
    std::map<int, long long> memo;
    long long operator()(int span) {
        if (span <= 2) return 1;
        auto it = memo.find(span);
        if (it != memo.end()) return it->second;
        const long long n = 1 + (*this)(span / 2) + (*this)(span - span / 2);
        memo[span] = n;
        return n;
    }
};
struct TreeBuild {
    rt_scene& s;
    std::vector<int>& list;
    const std::vector<Vec3>& mins;   // box minimum of every object index that occurs in the list (the sort key, Object.cpp:168-179)
    NodeCount& count;                // filled for every span that can occur before the threads start: read-only afterwards
    int build(int start, int end, uint32_t ctr, int slot) {
        const float draw = (float)(int)rand31(s.rng_key, ctr) / 2147483648.0f;   // Random01, DataUtility.h:11
        const int axis_draw = (int)(draw * 3);
        const int axis = axis_draw == 0 ? 0 : (axis_draw == 1 ? 1 : 2);
        const std::vector<Vec3>& m = mins;
        auto less = [&m, axis](int a, int b) { return m[a][axis] < m[b][axis]; };
        const int span = end - start;
        int left, right;
        int me;
        if (span == 1) { left = right = list[start]; me = slot; }
        else if (span == 2) {
            if (less(list[start], list[start + 1])) { left = list[start]; right = list[start + 1]; }
            else { left = list[start + 1]; right = list[start]; }
            me = slot;
        } else {
            std::sort(list.begin() + start, list.begin() + end, less);
            const int mid = start + span / 2;
            const long long n_left = count.memo.count(mid - start) ? count.memo.at(mid - start) : 1, n_right = count.memo.count(end - mid) ? count.memo.at(end - mid) : 1;
            if (span >= 8192) {   // large halves: the left one on another thread
                int l = -1;
                std::thread other([&] { l = build(start, mid, ctr + 1, slot); });
                right = build(mid, end, ctr + 1 + (uint32_t)n_left, slot + (int)n_left);
                other.join();
                left = l;
            } else {
                left = build(start, mid, ctr + 1, slot);
                right = build(mid, end, ctr + 1 + (uint32_t)n_left, slot + (int)n_left);
            }
            me = slot + (int)(n_left + n_right);
        }
        const Box3 bl = box_of(s, left), br = box_of(s, right);
        Obj& node = s.objs[me];
        node.kind = NODE;
        node.left = left;
        node.right = right;
        node.axis = axis;      // the children were ordered by their boxes' minimum along this axis: `left` has the smaller one
        // GetSurroundingBox (Object.cpp:156-166): fmin / fmax per component
        node.box.mn = {std::fmin(bl.mn.x, br.mn.x), std::fmin(bl.mn.y, br.mn.y), std::fmin(bl.mn.z, br.mn.z)};
        node.box.mx = {std::fmax(bl.mx.x, br.mx.x), std::fmax(bl.mx.y, br.mx.y), std::fmax(bl.mx.z, br.mx.z)};
        return me;
    }
};
int build_node(rt_scene& s, std::vector<int>& list, int start, int end) {
    NodeCount count;
    const long long n_nodes = count(end - start);
    std::vector<Vec3> mins(s.objs.size(), Vec3{0, 0, 0});
    for (int k = start; k < end; k++) mins[list[k]] = box_of(s, list[k]).mn;
    const int slot = (int)s.objs.size();
    s.objs.resize(s.objs.size() + (size_t)n_nodes);   // every node has its slot before any thread runs: the vector does not move
    TreeBuild tb{s, list, mins, count};
    const int root = tb.build(start, end, s.rng_ctr, slot);
    s.rng_ctr += (uint32_t)n_nodes;
    return root;
}

// PolygonMesh::PolygonMesh after parsing (Object.cpp:462-488): recentre, scale to `size`, one Triangle per 3 vertices
// (box as Object.h:84), triangle BVH.
void finish_mesh(rt_scene& s, int mesh_idx) {
    const std::vector<float> vv = s.objs[mesh_idx].verts;
    const int n_tri = (int)vv.size() / 24, nv = n_tri * 3;
    const float size = s.objs[mesh_idx].size;
    const int material = s.objs[mesh_idx].material, id = s.objs[mesh_idx].id;
    Vec3 pmin{vv[0], vv[1], vv[2]}, pmax = pmin;
    for (int i = 0; i < nv; i++) {
        const Vec3 pt{vv[8 * i], vv[8 * i + 1], vv[8 * i + 2]};
        pmin = {gmin(pmin.x, pt.x), gmin(pmin.y, pt.y), gmin(pmin.z, pt.z)};
        pmax = {gmax(pmax.x, pt.x), gmax(pmax.y, pt.y), gmax(pmax.z, pt.z)};
    }
    const Vec3 center = pmin * 0.5f + pmax * 0.5f;
    const Vec3 ext = pmax - pmin;
    const float factor = size / gmax(gmax(ext.x, ext.y), ext.z);
    std::vector<int> tris(n_tri);
    for (int t = 0; t < n_tri; t++) {
        Obj tri;
        tri.kind = TRI;
        tri.material = material;
        tri.id = id;
        for (int k = 0; k < 3; k++) {
            const float* r = &vv[8 * (3 * t + k)];
            tri.p[k] = (Vec3{r[0], r[1], r[2]} - center) * factor;
            tri.tc[k] = {r[3], r[4]};
            tri.n[k] = {r[5], r[6], r[7]};
        }
        for (int c = 0; c < 3; c++) {
            tri.box.mn.at(c) = gmin(tri.p[0][c], gmin(tri.p[1][c], tri.p[2][c]));
            tri.box.mx.at(c) = gmax(tri.p[0][c], gmax(tri.p[1][c], tri.p[2][c]));
        }
        s.objs.push_back(tri);
        tris[t] = (int)s.objs.size() - 1;
    }
    const int root = build_node(s, tris, 0, n_tri);
    s.objs[mesh_idx].root = root;
}

// InitializeRotationTransform (Object.cpp:254-292) with the axis pairs of Object.cpp:293-301
void finish_rotation(rt_scene& s, int idx) {
    Obj& o = s.objs[idx];
    const float angle = o.degrees * 0.01745329251994329576923690768489f;   // glm::radians
    o.sin_t = std::sin(angle);
    o.cos_t = std::cos(angle);
    Box3 box = box_of(s, o.target);
    o.pivot = box.mn * (1.0f - 0.5f) + box.mx * 0.5f;   // glm::lerp(min, max, 0.5f) = mix
    box.mn = box.mn - o.pivot;
    box.mx = box.mx - o.pivot;
    const int a1 = o.axis == 0 ? 1 : 0, a2 = o.axis == 2 ? 1 : 2, ra = o.axis;
    Vec3 mn{FLT_MAX, FLT_MAX, FLT_MAX}, mx{FLT_MIN, FLT_MIN, FLT_MIN};   // numeric_limits<float>::min(), Object.cpp:265
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                const float axis1 = i * box.mx[a1] + (1 - i) * box.mn[a1];
                const float axis2 = j * box.mx[a2] + (1 - j) * box.mn[a2];
                const float rest = k * box.mx[ra] + (1 - k) * box.mn[ra];
                Vec3 t{0, 0, 0};
                t.at(a1) = o.cos_t * axis1 + o.sin_t * axis2;
                t.at(a2) = -o.sin_t * axis1 + o.cos_t * axis2;
                t.at(ra) = rest;
                for (int c = 0; c < 3; c++) {
                    mn.at(c) = std::fmin(mn[c], t[c]);
                    mx.at(c) = std::fmax(mx[c], t[c]);
                }
            }
    o.box = {mn + o.pivot, mx + o.pivot};
}

// ------------------------------------------------------------------------------------------------- regrouping
// What the levels of a tree above its lowest nodes are for.  BVH_Node::Intersect (Object.cpp:241-252) tests a leaf when
// the boxes of ALL nodes above it are hit.  Every node's box is the fmin / fmax union of its children's boxes
// (Object.cpp:156-166), hence contains, coordinate by coordinate, the box of every node below it, and
// BoundingBox::DoesRayHit (DataUtility.cpp:35-82) is monotone in the box: a subtraction and a division by the same
// direction component round monotonically, so a larger box gets entry distances that are no larger and exit distances
// that are no smaller, and none of the test's comparisons can reject the larger box when it accepts the smaller one.
// (This needs boxes with min <= max and a ray whose quotients are numbers: no zero direction component, nothing that is
// not finite.  Trees with other boxes are left alone, and other rays are walked through the reference's own arrangement:
// rt_device.h ray_monotone, rt_engine.hip exceptions_kernel.)
// So a leaf is tested exactly when the box of its LOWEST node — the node whose children are leaves — is hit, and the
// levels above decide nothing: they only save work.  The reference builds them with a random axis and a median cut
// (Object.cpp:191-216); a regrouped layout keeps every lowest node with its own box and its leaves, and rebuilds the
// levels above with boxes that are unions again (so the same monotonicity makes the new hierarchy visit every lowest
// node whose box is hit): small trees bottom-up, always joining the two clusters whose union has the smallest surface;
// large ones cut top-down by the surface-area heuristic into parts that are then joined bottom-up.  What depended on
// the order of the visits — which of two leaves at the same distance wins, and the rejected first leaf whose
// HitInfo.distance survives — is settled by each leaf's rank in the reference's order (rt_device.h, closest_update).
struct Regroup {
    const rt_scene& s;
    struct FNode { Box3 box; int a, b; };   // children: >= 0 another FNode, < 0 the group ~child
    std::vector<int> groups;                // the tree's lowest nodes (object indices), in the reference's order
    std::vector<FNode> nodes;
    int root = -1;
    int JOIN_MAX = 256;                     // clusters of at most this many groups are joined bottom-up (RT_JOIN_MAX: experiments)

    static bool ordered(const Box3& b) {
        for (int c = 0; c < 3; c++)
            if (!(b.mn[c] <= b.mx[c]) || !(std::fabs(b.mn[c]) <= FLT_MAX) || !(std::fabs(b.mx[c]) <= FLT_MAX)) return false;
        return true;
    }
    // the lowest nodes of the tree under idx; false when the tree cannot be regrouped (a box with min > max, or a
    // node with a leaf beside a subtree)
    bool collect(int idx) {
        const Obj& o = s.objs[idx];
        if (o.kind != NODE || !ordered(o.box)) return false;
        const bool ln = s.objs[o.left].kind == NODE, rn = s.objs[o.right].kind == NODE;
        if (!ln && !rn) { groups.push_back(idx); return true; }
        if (ln != rn) return false;
        return collect(o.left) && collect(o.right);
    }
    const Box3& box_of_child(int child) const { return child < 0 ? s.objs[groups[(size_t)~child]].box : nodes[(size_t)child].box; }
    static Box3 unite(const Box3& a, const Box3& b) {
        return {{std::fmin(a.mn.x, b.mn.x), std::fmin(a.mn.y, b.mn.y), std::fmin(a.mn.z, b.mn.z)},
                {std::fmax(a.mx.x, b.mx.x), std::fmax(a.mx.y, b.mx.y), std::fmax(a.mx.z, b.mx.z)}};
    }
    static double area(const Box3& b) {
        const double x = (double)b.mx.x - b.mn.x, y = (double)b.mx.y - b.mn.y, z = (double)b.mx.z - b.mn.z;
        return x * y + y * z + x * z;
    }
    int join(int a, int b) {
        nodes.push_back({unite(box_of_child(a), box_of_child(b)), a, b});
        return (int)nodes.size() - 1;
    }
    // bottom-up: join the two clusters whose union has the smallest surface until one is left
    int agglomerate(std::vector<int> code) {
        const int n = (int)code.size();
        if (n == 1) return code[0];
        std::vector<Box3> box((size_t)n);
        std::vector<char> active((size_t)n, 1);
        std::vector<int> best((size_t)n, -1);
        std::vector<double> cost((size_t)n, 0.0);
        for (int i = 0; i < n; i++) box[i] = box_of_child(code[i]);
        auto find_best = [&](int i) {
            best[i] = -1;
            cost[i] = std::numeric_limits<double>::infinity();
            for (int j = 0; j < n; j++) {
                if (j == i || !active[j]) continue;
                const double c = area(unite(box[i], box[j]));
                if (c < cost[i]) { cost[i] = c; best[i] = j; }
            }
        };
        for (int i = 0; i < n; i++) find_best(i);
        for (int left = n; left > 1; left--) {
            int i = -1;
            for (int k = 0; k < n; k++)
                if (active[k] && (i < 0 || cost[k] < cost[i])) i = k;
            const int j = best[i];
            code[i] = join(code[i], code[j]);
            box[i] = nodes[(size_t)code[i]].box;
            active[j] = 0;
            if (left == 2) return code[i];
            find_best(i);
            for (int k = 0; k < n; k++) {
                if (!active[k] || k == i) continue;
                if (best[k] == i || best[k] == j) find_best(k);
                else {
                    const double c = area(unite(box[k], box[i]));
                    if (c < cost[k]) { cost[k] = c; best[k] = i; }
                }
            }
        }
        return code[0];
    }
    // top-down: cut ids[begin, end) where the surface-area heuristic (16 bins along the axis the centres spread most on)
    // is smallest, until the parts are small enough to be joined bottom-up
    int split(std::vector<int>& ids, int begin, int end) {
        const int n = end - begin;
        if (n <= JOIN_MAX) {
            std::vector<int> code((size_t)n);
            for (int k = 0; k < n; k++) code[k] = ~ids[(size_t)begin + k];
            return agglomerate(std::move(code));
        }
        auto centre = [&](int g, int c) { const Box3& b = s.objs[groups[(size_t)g]].box; return (double)b.mn[c] + (double)b.mx[c]; };
        double lo[3], hi[3];
        for (int c = 0; c < 3; c++) { lo[c] = hi[c] = centre(ids[begin], c); }
        for (int k = begin; k < end; k++)
            for (int c = 0; c < 3; c++) { const double v = centre(ids[k], c); lo[c] = std::min(lo[c], v); hi[c] = std::max(hi[c], v); }
        int axis = 0;
        for (int c = 1; c < 3; c++) if (hi[c] - lo[c] > hi[axis] - lo[axis]) axis = c;
        int mid = begin + n / 2;
        const double extent = hi[axis] - lo[axis];
        bool cut = false;
        if (extent > 0) {
            constexpr int BINS = 16;
            Box3 bb[BINS];
            int cnt[BINS] = {0};
            auto bin_of = [&](int g) { const int b = (int)((centre(g, axis) - lo[axis]) / extent * BINS); return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b); };
            for (int k = begin; k < end; k++) {
                const int b = bin_of(ids[k]);
                const Box3& gb = s.objs[groups[(size_t)ids[k]]].box;
                bb[b] = cnt[b] ? unite(bb[b], gb) : gb;
                cnt[b]++;
            }
            double right_area[BINS];
            int right_cnt[BINS];
            Box3 acc{};
            int c_acc = 0;
            for (int b = BINS - 1; b > 0; b--) {
                if (cnt[b]) { acc = c_acc ? unite(acc, bb[b]) : bb[b]; c_acc += cnt[b]; }
                right_area[b] = c_acc ? area(acc) : 0.0;
                right_cnt[b] = c_acc;
            }
            double best_cost = std::numeric_limits<double>::infinity();
            int best_b = -1;
            c_acc = 0;
            for (int b = 1; b < BINS; b++) {          // cut between bin b - 1 and bin b
                if (cnt[b - 1]) { acc = c_acc ? unite(acc, bb[b - 1]) : bb[b - 1]; c_acc += cnt[b - 1]; }
                if (c_acc == 0 || right_cnt[b] == 0) continue;
                const double c = area(acc) * c_acc + right_area[b] * right_cnt[b];
                if (c < best_cost) { best_cost = c; best_b = b; }
            }
            if (best_b > 0) {
                auto it = std::stable_partition(ids.begin() + begin, ids.begin() + end, [&](int g) { return bin_of(g) < best_b; });
                mid = (int)(it - ids.begin());
                cut = mid > begin && mid < end;
            }
        }
        if (!cut) {   // all centres in one place: halves by their place in the reference's order
            mid = begin + n / 2;
        }
        const int a = split(ids, begin, mid), b = split(ids, mid, end);
        return join(a, b);
    }
    bool build(int tree_root) {
        groups.clear(); nodes.clear();
        if (const char* v = getenv("RT_JOIN_MAX")) JOIN_MAX = std::max(1, atoi(v));
        if (!collect(tree_root) || groups.size() < 4) return false;
        std::vector<int> ids(groups.size());
        for (size_t k = 0; k < ids.size(); k++) ids[k] = (int)k;
        nodes.reserve(groups.size());
        root = split(ids, 0, (int)ids.size());
        return true;
    }
};

// ------------------------------------------------------------------------------------------------- entry boxes
// ApplyMovement / RotationIntersect (Object.cpp:601-608, 303-336) take every ray that reaches them into the mesh's frame
// before the mesh's root box is tested there — and most rays then miss that box (in the default scene every ray enters
// the tree's frame, one in six meets its box).  An entry box is a box in the SCENE's frame, tested like any inner node
// just before the instance's entry, that contains everything a ray can hit in there: when the ray misses it, the
// instance is not entered.  It must never be missed by a ray for which the reference's test of the root box, on the
// transformed ray, says hit.  It is the bounding box of the root box's eight corners taken back into the scene's frame
// (in double, by the inverse of the very matrix of float sine and cosine the transformation uses), widened by a margin
// that covers what rounding can move: with every coordinate of the scene and of the ray's origin at most G in
// magnitude, the transformed origin (two subtractions of up to 3 G, two products and a sum, the pivot added back) is off
// by at most 24 * 2^-24 G = 1.4e-6 G, the transformed direction by 3 * 2^-24 of its length (8e-7 G of position over the
// 4 G a ray can travel between origin and box), and the slab test, which compares quotients rounded to 2^-24 of their
// size, calls a box hit that the ray passes within 1e-6 G: 3.2e-6 G together; the margin is 4e-5 G.  Rays
// that start further out than G take the path of rays outside the reciprocal path, where an entry box counts as hit
// (rt_device.h: walker_start, walker_special).
float down(double v) { float f = (float)v; return (double)f > v ? std::nextafter(f, -INFINITY) : f; }
float up(double v) { float f = (float)v; return (double)f < v ? std::nextafter(f, INFINITY) : f; }
// the chain MOVE? ROT? MESH at idx: the box of the mesh's root box in the scene's frame (no margin); false for other chains
bool entry_box(const rt_scene& s, int idx, double out[6]) {
    double offset[3] = {0, 0, 0}, pivot[3] = {0, 0, 0}, sn = 0, cs = 1;
    int axis = -1;
    if (s.objs[idx].kind == MOVE) { const Vec3 o = s.objs[idx].offset; offset[0] = o.x; offset[1] = o.y; offset[2] = o.z; idx = s.objs[idx].target; }
    if (s.objs[idx].kind == ROT) {
        const Obj& r = s.objs[idx];
        pivot[0] = r.pivot.x; pivot[1] = r.pivot.y; pivot[2] = r.pivot.z; axis = r.axis; sn = r.sin_t; cs = r.cos_t;
        idx = r.target;
    }
    if (s.objs[idx].kind != MESH || s.objs[idx].root < 0) return false;
    const Box3 b = s.objs[s.objs[idx].root].box;
    const double det = cs * cs + sn * sn;
    if (!(det > 0.5 && det < 2.0)) return false;
    for (int k = 0; k < 3; k++) { out[k] = INFINITY; out[3 + k] = -INFINITY; }
    for (int corner = 0; corner < 8; corner++) {
        double p[3];
        for (int c = 0; c < 3; c++) p[c] = ((corner >> c) & 1) ? (double)b.mx[c] : (double)b.mn[c];
        if (axis >= 0) {
            // forward (rt_device.h rotate_in): (a1', a2') = (cos a1 - sin a2, sin a1 + cos a2) about the pivot
            const int a1 = axis == 0 ? 1 : 0, a2 = axis == 2 ? 1 : 2;
            const double x = p[a1] - pivot[a1], y = p[a2] - pivot[a2];
            p[a1] = (cs * x + sn * y) / det + pivot[a1];
            p[a2] = (-sn * x + cs * y) / det + pivot[a2];
        }
        for (int c = 0; c < 3; c++) {
            const double v = p[c] + offset[c];
            if (!(std::fabs(v) <= 1e30)) return false;
            out[c] = std::min(out[c], v);
            out[3 + c] = std::max(out[3 + c], v);
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------- flattening
struct Flattener {
    const rt_scene& s;
    rt_layout& f;
    std::vector<F4>& tris;          // the scene's triangle records, shared by its layouts
    std::vector<int>& tri_of;       // by object index: the triangle's record, -1 until it has one
    std::vector<int>& ent_obj;      // by entry of this layout: the object of a leaf or instance entry, -1 for other entries
    bool regroup = false;           // rebuild the upper levels of the trees that allow it
    double collapse = 0;            // > 0: wide nodes (see emit_regrouped)
    bool behind_entry_box = false;  // the tree is a mesh instance's and the instance has an entry box (see emit_tree)
    double entry_margin = -1;       // >= 0: mesh instances get an entry box (see entry_box) widened by this much
    int regrouped_trees = 0, regrouped_groups = 0;
    int first_leaf = -1;            // first leaf entry of the list being emitted

    void push(const F4& lo, const F4& hi, int obj) {
        f.ent_lo.push_back(lo);
        f.ent_hi.push_back(hi);
        ent_obj.push_back(obj);
    }
    int emit_leaf(int idx, bool top_level) {
        const Obj& o = s.objs[idx];
        const int at = (int)f.ent_lo.size();
        if (first_leaf < 0) first_leaf = at;
        switch (o.kind) {
        case SPHERE:
            push({o.c.x, o.c.y, o.c.z, o.r}, {bits_f(o.id), 0, 0, bits_f(ENT_SPHERE)}, idx);
            break;
        case CUBE:
            push({o.box.mn.x, o.box.mn.y, o.box.mn.z, bits_f(o.id)}, {o.box.mx.x, o.box.mx.y, o.box.mx.z, bits_f(ENT_CUBE)}, idx);
            break;
        case FOG:    // the boundary cube; the density sits in the object table
            push({o.box.mn.x, o.box.mn.y, o.box.mn.z, bits_f(o.id)}, {o.box.mx.x, o.box.mx.y, o.box.mx.z, bits_f(ENT_FOG)}, idx);
            break;
        case TRI: {
            if (tri_of[(size_t)idx] < 0) {
                tri_of[(size_t)idx] = (int)tris.size() / RT_TRI_F4;
                tris.push_back({o.p[0].x, o.p[0].y, o.p[0].z, 0});
                tris.push_back({o.p[1].x, o.p[1].y, o.p[1].z, 0});
                tris.push_back({o.p[2].x, o.p[2].y, o.p[2].z, 0});
                tris.push_back({o.tc[0].x, o.tc[0].y, o.tc[1].x, o.tc[1].y});
                tris.push_back({o.tc[2].x, o.tc[2].y, o.n[0].x, o.n[0].y});
                tris.push_back({o.n[0].z, 0, 0, 0});
            }
            push({bits_f(tri_of[(size_t)idx]), 0, 0, 0}, {0, 0, 0, bits_f(ENT_TRI)}, idx);
            break;
        }
        default: {   // a MOVE / ROT chain over a mesh or a fog volume = one instance
            (void)top_level;
            double wb[6];
            if (entry_margin >= 0 && entry_box(s, idx, wb)) {
                // the instance's entry box: an inner entry with the instance as its only child (missed: go on behind it)
                const float lo[3] = {down(wb[0] - entry_margin), down(wb[1] - entry_margin), down(wb[2] - entry_margin)};
                const float hi[3] = {up(wb[3] + entry_margin), up(wb[4] + entry_margin), up(wb[5] + entry_margin)};
                const int box_at = (int)f.ent_lo.size();
                push({lo[0], lo[1], lo[2], bits_f(box_at + 2)}, {hi[0], hi[1], hi[2], bits_f(ENT_INNER)}, -1);
                f.ent_right.resize(f.ent_lo.size(), -1);
                f.ent_axis.resize(f.ent_lo.size(), 3);
                f.ent_entry_box.resize(f.ent_lo.size(), 0);
                f.ent_entry_box[(size_t)box_at] = 1;
                if (first_leaf == box_at) first_leaf = box_at + 1;   // (the instance's own entry is the leaf)
            }
            const int inst_at = (int)f.ent_lo.size();
            push({bits_f((int)pending.size()), 0, 0, 0}, {0, 0, 0, bits_f(ENT_INSTANCE)}, idx);   // (the chain's first object stands for the instance)
            pending.push_back(idx);
            return inst_at;
        }
        }
        return at;
    }
    void note_children(int at, int right_at, int axis) {
        f.ent_lo[(size_t)at].w = bits_f((int)f.ent_lo.size());
        f.ent_right.resize(f.ent_lo.size(), -1);
        f.ent_axis.resize(f.ent_lo.size(), 3);
        f.ent_right[(size_t)at] = right_at;
        f.ent_axis[(size_t)at] = (uint8_t)(right_at >= 0 ? (axis >= 0 && axis <= 2 ? axis : 0) : 3);
    }
    void emit(int idx, bool top_level) {
        const Obj& o = s.objs[idx];
        if (o.kind != NODE) { emit_leaf(idx, top_level); return; }
        const int at = (int)f.ent_lo.size();
        push({o.box.mn.x, o.box.mn.y, o.box.mn.z, 0}, {o.box.mx.x, o.box.mx.y, o.box.mx.z, bits_f(ENT_INNER)}, -1);
        emit(o.left, top_level);
        // a node over a single object has it as both children (Object.cpp:198-200) and tests it twice; for a fog volume
        // the second test draws a new random number and can hit where the first did not, so it is listed twice
        int right_at = -1;
        if (o.right != o.left || draws_random(o.left)) { right_at = (int)f.ent_lo.size(); emit(o.right, top_level); }
        note_children(at, right_at, o.axis);
    }
    // a node of a regrouped hierarchy: its box, then the child whose box begins first along the axis on which the two
    // children's centres lie furthest apart (what the ordered traversal takes for the near child), then the other
    void emit_regrouped(const Regroup& rg, int child, bool top_level) {
        if (child < 0) { emit(rg.groups[(size_t)~child], top_level); return; }
        const Regroup::FNode& n = rg.nodes[(size_t)child];
        const Box3 &ba = rg.box_of_child(n.a), &bb = rg.box_of_child(n.b);
        int axis = 0;
        double far = -1;
        for (int c = 0; c < 3; c++) {
            const double d = std::fabs(((double)ba.mn[c] + ba.mx[c]) - ((double)bb.mn[c] + bb.mx[c]));
            if (d > far) { far = d; axis = c; }
        }
        const bool a_first = ba.mn[axis] <= bb.mn[axis];
        const int at = (int)f.ent_lo.size();
        push({n.box.mn.x, n.box.mn.y, n.box.mn.z, 0}, {n.box.mx.x, n.box.mx.y, n.box.mx.z, bits_f(ENT_INNER)}, -1);
        if (collapse > 0) {
            int right_at = -1, listed = 0;
            emit_children(rg, child, n.box, top_level, right_at, listed);
            note_children(at, right_at, axis);
            return;
        }
        emit_regrouped(rg, a_first ? n.a : n.b, top_level);
        const int right_at = (int)f.ent_lo.size();
        emit_regrouped(rg, a_first ? n.b : n.a, top_level);
        note_children(at, right_at, axis);
    }
    // Wide nodes.  A walk that follows skip links tests the box of every child of a node whose box is hit.  An inner child
    // is worth its own test only if the test often fails: with p the chance that a ray through the parent's box meets the
    // child's (the ratio of their surfaces), listing the child costs 1 + 2p tests, listing its two children in its place
    // costs 2 — less when p > 1/2.  So the children of `child` are listed under the box `parent`, and an inner child
    // with more than `collapse` of the parent's surface is replaced by its own children, measured against the same
    // parent.  Every box that remains is a union of lowest nodes' boxes inside its parent's, which is all the walk's
    // exactness rests on ("regrouping").  (Not for the ordered traversal, whose entries name exactly two children.)
    static double area_of(const Box3& b) {
        const double x = (double)b.mx.x - b.mn.x, y = (double)b.mx.y - b.mn.y, z = (double)b.mx.z - b.mn.z;
        return x * y + y * z + x * z;
    }
    void emit_children(const Regroup& rg, int child, const Box3& parent, bool top_level, int& right_at, int& listed) {
        const Regroup::FNode& n = rg.nodes[(size_t)child];
        for (int c : {n.a, n.b}) {
            if (c >= 0 && area_of(rg.nodes[(size_t)c].box) > collapse * area_of(parent)) { emit_children(rg, c, parent, top_level, right_at, listed); continue; }
            if (listed == 1) right_at = (int)f.ent_lo.size();
            listed++;
            emit_regrouped(rg, c, top_level);
        }
    }
    // One tree of the scene: the top-level tree or a mesh's.  Only meshes are regrouped.  The top-level tree of a large
    // scene is walked near side first with distance culling, and there the reference's own upper levels do better than
    // rebuilt ones: its lowest nodes are the cells of its own partition of space, which no box of its upper levels cuts
    // through, while any other partition cuts through some of them and gets overlapping boxes for it (measured on the
    // 100 000-primitive scene: 79 slab tests per ray in the reference's hierarchy, 92 to 100 in rebuilt ones).  A mesh is
    // never culled (no distance bound exists for a triangle) and its tree of random axes (Object.cpp:191) over thin,
    // slanted triangles leaves much to gain.
    void emit_tree(int root, bool top_level) {
        if (regroup && (!top_level || getenv("RT_REGROUP_TOP"))) {
            Regroup rg{s};
            if (rg.build(root)) {
                regrouped_trees++;
                regrouped_groups += (int)rg.groups.size();
                // With wide nodes the root of a mesh behind an entry box is the first node to be left out: a ray comes
                // here through the entry box, which is the root's box taken to the scene's frame and widened, so the
                // root's own test hardly ever fails; its children start the instance's list.
                if (collapse > 0 && behind_entry_box && !top_level && rg.root >= 0) {
                    int right_at = -1, listed = 0;
                    emit_children(rg, rg.root, rg.nodes[(size_t)rg.root].box, top_level, right_at, listed);
                } else emit_regrouped(rg, rg.root, top_level);
                return;
            }
        }
        emit(root, top_level);
    }
    bool draws_random(int idx) const {
        while (s.objs[idx].kind == MOVE || s.objs[idx].kind == ROT) idx = s.objs[idx].target;
        return s.objs[idx].kind == FOG;
    }
    std::vector<int> pending;   // instance leaves of the top level, in instance-index order
};

// One layout of the scene: entries, instance records, what the ordered traversal and distance culling need per entry.
// inst_obj: by instance of this layout, the object (wrapper chain) it stands for.
void flatten_layout(const rt_scene& s, rt_layout& f, std::vector<F4>& tris, std::vector<int>& tri_of, std::vector<int>& ent_obj,
                    std::vector<int>& inst_obj, bool regroup, double entry_margin, int* regrouped_trees, int* regrouped_groups, double collapse = 0) {
    f = rt_layout();
    ent_obj.clear();
    Flattener fl{s, f, tris, tri_of, ent_obj};
    fl.regroup = regroup;
    fl.collapse = collapse;
    fl.entry_margin = entry_margin;
    fl.emit_tree(s.root, true);
    f.top_end = (int)f.ent_lo.size();
    fl.push({0, 0, 0, 0}, {0, 0, 0, bits_f(ENT_END)}, -1);
    f.stale_entry = fl.first_leaf;
    f.inst.resize(fl.pending.size() * RT_INST_F4);
    inst_obj = fl.pending;
    int trees = fl.regrouped_trees, groups = fl.regrouped_groups;
    for (size_t k = 0; k < fl.pending.size(); k++) {
        int idx = fl.pending[k];
        Vec3 offset{0, 0, 0}, pivot{0, 0, 0};
        float has_move = 0, sin_t = 0, cos_t = 1;
        int axis = -1;
        const int id = s.objs[idx].id;
        if (s.objs[idx].kind == MOVE) { offset = s.objs[idx].offset; has_move = 1; idx = s.objs[idx].target; }
        if (s.objs[idx].kind == ROT) {
            pivot = s.objs[idx].pivot; axis = s.objs[idx].axis; sin_t = s.objs[idx].sin_t; cos_t = s.objs[idx].cos_t;
            idx = s.objs[idx].target;
        }
        const int begin = (int)f.ent_lo.size();
        Flattener sub{s, f, tris, tri_of, ent_obj};
        sub.regroup = regroup;
        sub.collapse = collapse;
        { double wb[6]; sub.behind_entry_box = entry_margin >= 0 && entry_box(s, fl.pending[k], wb); }
        if (s.objs[idx].kind == MESH) sub.emit_tree(s.objs[idx].root, false);
        else sub.emit_leaf(idx, false);   // a wrapped fog volume: a range of one entry
        trees += sub.regrouped_trees; groups += sub.regrouped_groups;
        const int end = (int)f.ent_lo.size();
        sub.push({0, 0, 0, 0}, {0, 0, 0, bits_f(ENT_END)}, -1);
        f.inst[k * RT_INST_F4 + 0] = {offset.x, offset.y, offset.z, has_move};
        f.inst[k * RT_INST_F4 + 1] = {pivot.x, pivot.y, pivot.z, bits_f(axis)};
        f.inst[k * RT_INST_F4 + 2] = {sin_t, cos_t, bits_f(begin), bits_f(end)};
        f.inst[k * RT_INST_F4 + 3] = {bits_f(id), bits_f(sub.first_leaf), 0, 0};
    }
    if (regrouped_trees) *regrouped_trees = trees;
    if (regrouped_groups) *regrouped_groups = groups;
    f.idle_entry = (int)f.ent_lo.size();
    fl.push({0, 0, 0, 0}, {0, 0, 0, bits_f(ENT_IDLE)}, -1);
    f.ent_right.resize(f.ent_lo.size(), -1);
    f.ent_axis.resize(f.ent_lo.size(), 3);
    f.ent_peer.assign(f.ent_lo.size(), -1);
    f.ent_rank.assign(f.ent_lo.size(), 0u);
    f.ent_entry_box.resize(f.ent_lo.size(), 0);
    // Ordered traversal (rt_device.h, "ordered traversal") keeps the children it has yet to visit on a per-ray stack: an
    // inner entry with two children adds one while its first child's subtree is walked, a mesh instance one (the way
    // back out) while its own tree is walked; one more slot holds the end of the top-level list.  Children follow their
    // parents in the entry array and the instances' ranges follow the top-level list, so one pass downwards suffices.
    {
        std::vector<int> need(f.ent_lo.size(), 0);
        for (int i = (int)f.ent_lo.size() - 1; i >= 0; i--) {
            const int kind = bits_i(f.ent_hi[i].w);
            if (kind == ENT_INNER) {
                const int l = need[(size_t)i + 1], r = f.ent_right[i] >= 0 ? need[f.ent_right[i]] : -1;
                need[i] = r < 0 ? l : 1 + std::max(l, r);
            } else if (kind == ENT_INSTANCE) {
                const size_t k = (size_t)bits_i(f.ent_lo[i].x) * RT_INST_F4;
                if (k + RT_INST_F4 <= f.inst.size()) need[i] = 1 + need[bits_i(f.inst[k + 2].z)];
            }
        }
        f.stack_need = 1 + need[0];
    }
    // Distance culling data (see rt_device.h).  Entries are in preorder and an inner entry's skip index is the end of its
    // subtree, so walking the indices downwards meets the children of an entry before the entry itself.
    {
        const float INF = std::numeric_limits<float>::infinity();
        const size_t n = f.ent_lo.size();
        f.ent_rho.assign(n, INF);
        for (int i = f.top_end - 1; i >= 0; i--) {
            const int kind = bits_i(f.ent_hi[i].w);
            if (kind == ENT_SPHERE) {
                const float r = f.ent_lo[i].w;
                const float cm = std::fmax(std::fabs(f.ent_lo[i].x), std::fmax(std::fabs(f.ent_lo[i].y), std::fabs(f.ent_lo[i].z)));
                if (r > 0 && r <= FLT_MAX && cm <= FLT_MAX) f.ent_rho[i] = (0.5f / r) * 1.000001f;
            } else if (kind == ENT_CUBE) {
                const bool ordered = f.ent_lo[i].x <= f.ent_hi[i].x && f.ent_lo[i].y <= f.ent_hi[i].y && f.ent_lo[i].z <= f.ent_hi[i].z;
                if (ordered) f.ent_rho[i] = 0.0f;
            } else if (kind == ENT_INNER) {
                const int skip = bits_i(f.ent_lo[i].w);
                float rho = 0.0f;
                for (int c = i + 1; c < skip; c = bits_i(f.ent_hi[c].w) == ENT_INNER ? bits_i(f.ent_lo[c].w) : c + 1) rho = std::fmax(rho, f.ent_rho[c]);
                const bool ordered = f.ent_lo[i].x <= f.ent_hi[i].x && f.ent_lo[i].y <= f.ent_hi[i].y && f.ent_lo[i].z <= f.ent_hi[i].z;
                f.ent_rho[i] = ordered ? rho : INF;
            }
        }
    }
}

void flatten(rt_scene& s) {
    rt_scene_flat& f = s.flat;
    f = rt_scene_flat();
    std::vector<int> tri_of(s.objs.size(), -1), ent_obj, inst_obj;
    // entry boxes of mesh instances (see "entry boxes"): G = 16 x the largest coordinate of the scene, margin 4e-5 G
    double entry_margin = -1;
    f.entry_guard = FLT_MAX;
    if (!getenv("RT_NO_ENTRY_BOX")) {
        double big = 0;
        bool any = false, fine = true;
        auto see = [&](double v) { if (!(std::fabs(v) <= 1e30)) fine = false; else big = std::max(big, std::fabs(v)); };
        for (size_t i = 0; i < s.objs.size(); i++) {
            const Obj& o = s.objs[i];
            if (o.kind == SPHERE) { for (int c = 0; c < 3; c++) { see((double)o.c[c] + std::fabs(o.r)); see((double)o.c[c] - std::fabs(o.r)); } }
            else if (o.kind == CUBE || o.kind == NODE || o.kind == TRI || o.kind == ROT || o.kind == FOG) { for (int c = 0; c < 3; c++) { see(o.box.mn[c]); see(o.box.mx[c]); } }
            if (o.kind == ROT) for (int c = 0; c < 3; c++) see(o.pivot[c]);
            if (o.kind == MOVE) for (int c = 0; c < 3; c++) see(o.offset[c]);
        }
        for (int idx : s.top) {
            double wb[6];
            if (entry_box(s, idx, wb)) { any = true; for (double v : wb) see(v); }
        }
        if (any && fine && big > 0) {
            const double G = 16.0 * big;
            entry_margin = 4e-5 * G;
            f.entry_guard = (float)G;
        }
    }
    flatten_layout(s, f, f.tris, tri_of, ent_obj, inst_obj, false, entry_margin, nullptr, nullptr);
    // every leaf's place in the order in which the reference meets the leaves: the top-level list from its start, a
    // mesh's triangles where its instance's entry stands
    {
        uint32_t rank = 0;
        auto is_leaf = [&](int i) { const int k = bits_i(f.ent_hi[i].w); return k >= ENT_SPHERE && k <= ENT_FOG; };
        for (int i = 0; i < f.top_end; i++) {
            if (is_leaf(i)) f.ent_rank[i] = rank++;
            else if (bits_i(f.ent_hi[i].w) == ENT_INSTANCE) {
                const size_t k = (size_t)bits_i(f.ent_lo[i].x) * RT_INST_F4;
                for (int j = bits_i(f.inst[k + 2].z); j < bits_i(f.inst[k + 2].w); j++)
                    if (is_leaf(j)) f.ent_rank[j] = rank++;
            }
        }
    }
    f.fast_div_ok = true;
    bool has_fog = false;
    float r_max = 0, c_max = 0;
    for (size_t i = 0; i < f.ent_lo.size(); i++) {
        const int kind = bits_i(f.ent_hi[i].w);
        if (kind == ENT_FOG) has_fog = true;
        if (kind == ENT_SPHERE && i < (size_t)f.top_end) {
            const float r = f.ent_lo[i].w;
            const float cm = std::fmax(std::fabs(f.ent_lo[i].x), std::fmax(std::fabs(f.ent_lo[i].y), std::fabs(f.ent_lo[i].z)));
            if (r > 0 && r <= FLT_MAX && cm <= FLT_MAX) { r_max = std::fmax(r_max, r); c_max = std::fmax(c_max, cm + r); }
        }
        if (kind != ENT_INNER) continue;
        const float c[6] = {f.ent_lo[i].x, f.ent_lo[i].y, f.ent_lo[i].z, f.ent_hi[i].x, f.ent_hi[i].y, f.ent_hi[i].z};
        for (float v : c) {
            const float a = std::fabs(v);
            if (!(v == 0.0f || (a >= 8.8817841970012523e-16f && a <= 1.099511627776e12f))) f.fast_div_ok = false;
        }
    }
    f.cull_2rmax = 2.0f * r_max;
    f.cull_cabs = r_max * 1.52587890625e-05f;                                    // 2^-16 r_max
    f.cull_delta = (c_max + 2.0f * r_max) * 1.1920928955078125e-07f;             // 2^-23 (max |c| + r) + 2^-22 r_max
    // The regrouped layout (see "regrouping"); never with fog, whose random draws follow the reference's visiting order
    if (f.fast_div_ok && !has_fog && !getenv("RT_NO_REGROUP")) {
        // leaves of a layout of the engine's own and of the reference's name each other; ranks and the leaves whose
        // rejected HitInfo.distance survives (the reference's first leaves, wherever they stand now) go with them
        auto tie = [&](rt_layout& own, const std::vector<int>& ent_obj_own, const std::vector<int>& inst_obj_own, std::vector<int32_t>& peer_of_ref) {
            std::vector<int> at_own(s.objs.size(), -1);
            for (size_t e = 0; e < ent_obj_own.size(); e++) if (ent_obj_own[e] >= 0) at_own[(size_t)ent_obj_own[e]] = (int)e;
            peer_of_ref.assign(f.ent_lo.size(), -1);
            for (size_t e = 0; e < ent_obj.size(); e++) {
                if (ent_obj[e] < 0) continue;
                const int fe = at_own[(size_t)ent_obj[e]];
                peer_of_ref[e] = fe;
                own.ent_peer[(size_t)fe] = (int)e;
                own.ent_rank[(size_t)fe] = f.ent_rank[e];
            }
            own.stale_entry = f.stale_entry >= 0 ? peer_of_ref[(size_t)f.stale_entry] : -1;
            for (size_t kf = 0; kf < inst_obj_own.size(); kf++)
                for (size_t kr = 0; kr < inst_obj.size(); kr++)
                    if (inst_obj[kr] == inst_obj_own[kf]) {
                        const int first = bits_i(f.inst[kr * RT_INST_F4 + 3].y);
                        own.inst[kf * RT_INST_F4 + 3].y = bits_f(first >= 0 ? peer_of_ref[(size_t)first] : -1);
                    }
        };
        std::vector<int> ent_obj_fast, inst_obj_fast;
        flatten_layout(s, f.fast, f.tris, tri_of, ent_obj_fast, inst_obj_fast, true, entry_margin, &f.regrouped_trees, &f.regrouped_groups);
        if (f.regrouped_trees > 0) {
            f.has_fast = true;
            tie(f.fast, ent_obj_fast, inst_obj_fast, f.ent_peer);
            if (!getenv("RT_NO_WIDE")) {
                std::vector<int> ent_obj_wide, inst_obj_wide;
                flatten_layout(s, f.wide, f.tris, tri_of, ent_obj_wide, inst_obj_wide, true, entry_margin, nullptr, nullptr, 0.5);
                f.has_wide = true;
                tie(f.wide, ent_obj_wide, inst_obj_wide, f.ent_peer_wide);
            }
        } else f.fast = rt_layout();
    }
    // texel pool: images, then the skybox; each padded by one zero row because uv == 1.0 indexes row h
    // (Object.cpp:546-547 wrap; World.cpp:174-183 has no clamp)
    for (const Image& im : s.images) {
        ImageRec r{(uint32_t)f.texels.size(), (uint32_t)im.w, (uint32_t)im.h, 4};
        f.texels.insert(f.texels.end(), im.px.begin(), im.px.end());
        f.texels.resize(f.texels.size() + (size_t)im.w * 4 + 16, 0);
        f.texels.resize((f.texels.size() + 15) & ~(size_t)15, 0);
        f.images.push_back(r);
    }
    f.sky = {(uint32_t)f.texels.size(), (uint32_t)s.sky.w, (uint32_t)s.sky.h, 3};
    f.texels.insert(f.texels.end(), s.sky.px.begin(), s.sky.px.end());
    f.texels.resize(f.texels.size() + (size_t)s.sky.w * 3 + 16, 0);
    f.texels.resize((f.texels.size() + 15) & ~(size_t)15, 0);
    // objects by id
    f.objects.assign((size_t)s.next_id * RT_OBJ_F4, F4{0, 0, 0, 0});
    auto put_object = [&](int id, int material, float neg_inv_density = 0.0f) {
        const Mat& m = s.mats[material];
        int tk = -1, img = -1;
        Vec3 c1{0, 0, 0}, c2{0, 0, 0};
        if (m.tex >= 0) { const Tex& t = s.texs[m.tex]; tk = t.kind; img = t.image; c1 = t.c1; c2 = t.c2; }
        f.objects[(size_t)id * RT_OBJ_F4 + 0] = {bits_f(m.type), bits_f(tk), bits_f(img), neg_inv_density};
        f.objects[(size_t)id * RT_OBJ_F4 + 1] = {c1.x, c1.y, c1.z, 0};
        f.objects[(size_t)id * RT_OBJ_F4 + 2] = {c2.x, c2.y, c2.z, 0};
        f.objects[(size_t)id * RT_OBJ_F4 + 3] = {m.emit.x, m.emit.y, m.emit.z, 0};
    };
    auto mesh_of = [&](int idx) {
        while (s.objs[idx].kind == MOVE || s.objs[idx].kind == ROT) idx = s.objs[idx].target;
        return idx;
    };
    for (int idx : s.top) put_object(s.objs[idx].id, s.objs[mesh_of(idx)].material, s.objs[mesh_of(idx)].neg_inv_density);
    for (int idx : s.planes) {
        put_object(s.objs[idx].id, s.objs[idx].material);
        f.planes.push_back({s.objs[idx].y, bits_f(s.objs[idx].id), 0, 0});
    }
}

int check_material(rt_scene* s, int material) {
    if (!s) return fail(RT_E_INVALID, "null scene");
    if (s->built) return fail(RT_E_INVALID, "scene is already built");
    if (material < 0 || material >= (int)s->mats.size()) return fail(RT_E_INVALID, "material id out of range");
    return RT_OK;
}
int add_top(rt_scene* s, Obj&& o) {
    o.id = s->next_id++;
    s->objs.push_back(std::move(o));
    s->top.push_back((int)s->objs.size() - 1);
    return s->objs.back().id;
}

// first-visit leaf numbering dumps, shared by the two dump entry points
struct Dump {
    const rt_scene& s;
    std::vector<float> boxes;
    std::vector<int> left, right, leaves, slot_of;
    explicit Dump(const rt_scene& sc) : s(sc), slot_of(sc.objs.size(), -1) {}
    int rec(int idx) {
        const Obj& o = s.objs[idx];
        if (o.kind == NODE) {
            const int me = (int)left.size();
            left.push_back(0);
            right.push_back(0);
            const float b[6] = {o.box.mn.x, o.box.mn.y, o.box.mn.z, o.box.mx.x, o.box.mx.y, o.box.mx.z};
            boxes.insert(boxes.end(), b, b + 6);
            const int l = rec(o.left), r = rec(o.right);
            left[me] = l;
            right[me] = r;
            return me;
        }
        if (slot_of[idx] < 0) { slot_of[idx] = (int)leaves.size(); leaves.push_back(idx); }
        return ~slot_of[idx];
    }
};

}  // namespace

// =================================================================================================== C ABI: scene
extern "C" {

void rt_config_defaults(rt_config* c) {
    memset(c, 0, sizeof *c);
    c->width = 1280; c->height = 720; c->samples_per_axis = 3; c->bounces = 3;
    c->fov = 1.5f; c->skybox_brightness = 0.5f; c->dof_intensity = 0.05f; c->focus_distance = 10.0f;
    c->device = 0; c->max_batch_paths = 0; c->tile_w = 64; c->tile_h = 64; c->flags = 0;
}

int rt_scene_create(rt_scene** out) {
    if (!out) return fail(RT_E_INVALID, "null output pointer");
    *out = new rt_scene();
    return RT_OK;
}
void rt_scene_destroy(rt_scene* s) { delete s; }

int rt_scene_add_texture_color(rt_scene* s, const float rgb[3]) {
    if (!s || !rgb) return fail(RT_E_INVALID, "null argument");
    s->texs.push_back({RT_TEX_COLOR, {rgb[0], rgb[1], rgb[2]}, {0, 0, 0}, -1});
    return (int)s->texs.size() - 1;
}
int rt_scene_add_texture_checker(rt_scene* s, const float a[3], const float b[3]) {
    if (!s || !a || !b) return fail(RT_E_INVALID, "null argument");
    s->texs.push_back({RT_TEX_CHECKER, {a[0], a[1], a[2]}, {b[0], b[1], b[2]}, -1});
    return (int)s->texs.size() - 1;
}
static int rt_scene_add_texture_image_impl(rt_scene* s, const uint8_t* rgba, int w, int h) {
    if (!s || !rgba || w <= 0 || h <= 0) return fail(RT_E_INVALID, "bad image texture arguments");
    Image im{std::vector<uint8_t>(rgba, rgba + (size_t)w * h * 4), w, h, 4};
    s->images.push_back(std::move(im));
    s->texs.push_back({RT_TEX_IMAGE, {0, 0, 0}, {0, 0, 0}, (int)s->images.size() - 1});
    return (int)s->texs.size() - 1;
}
static int rt_scene_add_texture_file_impl(rt_scene* s, const char* path) {
    if (!s || !path) return fail(RT_E_INVALID, "null argument");
    std::vector<uint8_t> px;
    int w, h, fc;
    std::string err;
    if (!load_png(path, 4, true, px, w, h, fc, err)) return fail(RT_E_IO, err);
    // DataUtility.cpp:105-108 accepts 3- and 4-channel files only
    if (fc != 3 && fc != 4) return fail(RT_E_IO, std::string("An image had ") + std::to_string(fc) + " channels: " + path);
    return rt_scene_add_texture_image(s, px.data(), w, h);
}
int rt_scene_add_material(rt_scene* s, int type, int texture, const float emit[3]) {
    if (!s) return fail(RT_E_INVALID, "null scene");
    if (type < RT_MAT_NONE || type > RT_MAT_LIGHT) return fail(RT_E_INVALID, "unknown material type");
    if (texture >= (int)s->texs.size()) return fail(RT_E_INVALID, "texture id out of range");
    if ((type == RT_MAT_DIFFUSE || type == RT_MAT_METAL || type == RT_MAT_ISOTROPIC) && texture < 0)
        return fail(RT_E_INVALID, "this material type needs a texture");
    Vec3 e{0, 0, 0};
    if (emit) e = {emit[0], emit[1], emit[2]};
    s->mats.push_back({type, texture < 0 ? -1 : texture, e});
    return (int)s->mats.size() - 1;
}
int rt_scene_add_sphere(rt_scene* s, const float c[3], float radius, int material) {
    if (int rc = check_material(s, material)) return rc;
    Obj o;
    o.kind = SPHERE; o.c = {c[0], c[1], c[2]}; o.r = radius; o.material = material;
    return add_top(s, std::move(o));
}
int rt_scene_add_cube_minmax(rt_scene* s, const float mn[3], const float mx[3], int material) {
    if (int rc = check_material(s, material)) return rc;
    Obj o;
    o.kind = CUBE; o.box = {{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}}; o.material = material;
    return add_top(s, std::move(o));
}
int rt_scene_add_cube_center(rt_scene* s, const float c[3], float half, int material) {   // Object.h:29
    const float mn[3] = {c[0] - half, c[1] - half, c[2] - half}, mx[3] = {c[0] + half, c[1] + half, c[2] + half};
    return rt_scene_add_cube_minmax(s, mn, mx, material);
}
int rt_scene_add_plane(rt_scene* s, float y, int material) {
    if (int rc = check_material(s, material)) return rc;
    Obj o;
    o.kind = PLANE; o.y = y; o.material = material;
    if (y == 0) o.y = 0.001f;   // Object.h:40
    o.id = s->next_id++;
    s->objs.push_back(o);
    s->planes.push_back((int)s->objs.size() - 1);
    return o.id;
}
static int rt_scene_add_mesh_impl(rt_scene* s, const float* verts8, int n_tri, float size, int material, int rot_axis, float degrees,
                      int has_move, const float offset[3]) {
    if (int rc = check_material(s, material)) return rc;
    if (!verts8 || n_tri <= 0) return fail(RT_E_INVALID, "mesh has no triangles");
    if (rot_axis < -1 || rot_axis > 2) return fail(RT_E_INVALID, "rotation axis must be -1, 0, 1 or 2");
    const int id = s->next_id++;
    Obj m;
    m.kind = MESH; m.material = material; m.id = id; m.size = size;
    m.verts.assign(verts8, verts8 + (size_t)n_tri * 24);
    s->objs.push_back(std::move(m));
    int cur = (int)s->objs.size() - 1;
    if (rot_axis >= 0) {
        Obj r;
        r.kind = ROT; r.target = cur; r.axis = rot_axis; r.degrees = degrees; r.id = id;
        s->objs.push_back(r);
        cur = (int)s->objs.size() - 1;
    }
    if (has_move) {
        Obj v;
        v.kind = MOVE; v.target = cur; v.id = id;
        if (offset) v.offset = {offset[0], offset[1], offset[2]};
        s->objs.push_back(v);
        cur = (int)s->objs.size() - 1;
    }
    s->top.push_back(cur);
    return id;
}
// Fog (Object.h:48-57): boundary AxisAlignedCube(pos, radius) (Object.h:29), negInverseDensity = -1 / density, an
// Isotropic material of its own built from the texture it is given — here: `material` must be Isotropic.
int rt_scene_add_fog(rt_scene* s, const float center[3], float radius, float density, int material, int rot_axis, float degrees,
                     int has_move, const float offset[3]) {
    if (int rc = check_material(s, material)) return rc;
    if (!center) return fail(RT_E_INVALID, "null argument");
    if (s->mats[material].type != RT_MAT_ISOTROPIC) return fail(RT_E_INVALID, "a fog volume needs an isotropic material");
    if (rot_axis < -1 || rot_axis > 2) return fail(RT_E_INVALID, "rotation axis must be -1, 0, 1 or 2");
    const int id = s->next_id++;
    Obj g;
    g.kind = FOG; g.material = material; g.id = id;
    const Vec3 c{center[0], center[1], center[2]}, rr{radius, radius, radius};
    g.box = {c - rr, c + rr};
    g.neg_inv_density = -1 / density;
    s->objs.push_back(g);
    int cur = (int)s->objs.size() - 1;
    if (rot_axis >= 0) {
        Obj r;
        r.kind = ROT; r.target = cur; r.axis = rot_axis; r.degrees = degrees; r.id = id;
        s->objs.push_back(r);
        cur = (int)s->objs.size() - 1;
    }
    if (has_move) {
        Obj v;
        v.kind = MOVE; v.target = cur; v.id = id;
        if (offset) v.offset = {offset[0], offset[1], offset[2]};
        s->objs.push_back(v);
        cur = (int)s->objs.size() - 1;
    }
    s->top.push_back(cur);
    return id;
}
static int rt_scene_add_mesh_file_impl(rt_scene* s, const char* path, float size, int material, int rot_axis, float degrees, int has_move,
                           const float offset[3]) {
    if (!path) return fail(RT_E_INVALID, "null path");
    std::vector<float> v;
    std::string err;
    if (!load_obj(path, v, err)) return fail(RT_E_IO, err);
    if (v.size() % 24 != 0 || v.empty()) return fail(RT_E_IO, std::string("Vertex count was not a multiple of 3: ") + path);
    return rt_scene_add_mesh(s, v.data(), (int)v.size() / 24, size, material, rot_axis, degrees, has_move, offset);
}
static int rt_scene_set_skybox_impl(rt_scene* s, const uint8_t* rgb, int w, int h) {
    if (!s || !rgb || w <= 0 || h <= 0) return fail(RT_E_INVALID, "bad skybox arguments");
    if (s->built) return fail(RT_E_INVALID, "scene is already built");
    s->sky = Image{std::vector<uint8_t>(rgb, rgb + (size_t)w * h * 3), w, h, 3};
    return RT_OK;
}
static int rt_scene_set_skybox_file_impl(rt_scene* s, const char* path) {
    if (!s || !path) return fail(RT_E_INVALID, "null argument");
    std::vector<uint8_t> px;
    int w, h, fc;
    std::string err;
    if (!load_png(path, 3, true, px, w, h, fc, err)) return fail(RT_E_IO, err);
    if (fc != 3) return fail(RT_E_IO, "The skybox had " + std::to_string(fc) + " channels instead of 3.");   // DataUtility.cpp:24-27
    return rt_scene_set_skybox(s, px.data(), w, h);
}

// The alternative scene the reference keeps behind "#else" (World.cpp:34-40): a room of four slabs, an emissive sphere that
// moves with `time`, a fog volume rotated about X, and the checkered metal plane (World.cpp:17).  The light's x is
// "3 * sin(time * 0.1f)": the unqualified sin takes the double overload, and the product narrows into the vec3.
static int rt_scene_add_alt_objects_impl(rt_scene* s, const char* asset_dir, float time) {
    if (!s || !asset_dir) return fail(RT_E_INVALID, "null argument");
    const std::string dir = std::string(asset_dir) + "/src/stb_image/";
    if (int rc = rt_scene_set_skybox_file(s, (dir + "skybox12.png").c_str())) return rc;   // World.cpp:49
    auto colored = [&](int type, float r, float g, float b) {
        const float c[3] = {r, g, b};
        return rt_scene_add_material(s, type, rt_scene_add_texture_color(s, c), nullptr);
    };
    const float slabs[4][6] = {{-6, 0, -6, -5, 6, 6}, {5, 0, -6, 6, 6, 6}, {-6, 0, 5, 6, 6, 6}, {-6, 5, -6, 6, 6, 6}};
    const float cols[4][3] = {{0.8f, 0.2f, 0.3f}, {0.2f, 0.8f, 0.3f}, {0.4f, 0.4f, 0.4f}, {0.4f, 0.4f, 0.4f}};
    for (int i = 0; i < 4; i++)
        rt_scene_add_cube_minmax(s, slabs[i], slabs[i] + 3, colored(RT_MAT_DIFFUSE, cols[i][0], cols[i][1], cols[i][2]));
    const float light_pos[3] = {(float)(3 * std::sin((double)(time * 0.1f))), 4.5f, 3.5f}, emit[3] = {5.0f, 2.0f, 2.0f};
    rt_scene_add_sphere(s, light_pos, 2.5f, rt_scene_add_material(s, RT_MAT_LIGHT, -1, emit));
    const float fog_pos[3] = {-3, 1, 0};
    const int fog = rt_scene_add_fog(s, fog_pos, 3.0f, 0.5f, colored(RT_MAT_ISOTROPIC, 1.0f, 1.0f, 0.0f), 0, 20.0f, 0, nullptr);
    if (fog < 0) return fog;
    const float white[3] = {1.0f, 1.0f, 1.0f}, green[3] = {0.2f, 0.6f, 0.3f};   // World.cpp:17
    rt_scene_add_plane(s, 0, rt_scene_add_material(s, RT_MAT_METAL, rt_scene_add_texture_checker(s, white, green), nullptr));
    return RT_OK;
}

// World.cpp:15-32.  `time` only matters to the alternative scene of World.cpp:34-40, which the reference compiles out.
static int rt_scene_add_default_objects_impl(rt_scene* s, const char* asset_dir, float time) {
    (void)time;
    if (!s || !asset_dir) return fail(RT_E_INVALID, "null argument");
    const std::string dir = std::string(asset_dir) + "/src/stb_image/";
    if (int rc = rt_scene_set_skybox_file(s, (dir + "skybox12.png").c_str())) return rc;   // World.cpp:49
    auto diffuse = [&](int tex) { return rt_scene_add_material(s, RT_MAT_DIFFUSE, tex, nullptr); };
    const float grey[3] = {0.4f, 0.4f, (float)0.4};
    const float c6[3] = {6, 2, -9};
    rt_scene_add_cube_center(s, c6, 2, diffuse(rt_scene_add_texture_color(s, grey)));
    const float brown[3] = {0.6f, 0.3f, 0.2f}, white[3] = {1.0f, 1.0f, 1.0f};
    const float c8[3] = {8, 2, -4};
    rt_scene_add_sphere(s, c8, 2, diffuse(rt_scene_add_texture_checker(s, brown, white)));
    const int tree_tex = rt_scene_add_texture_file(s, (dir + "tree_texture.png").c_str());
    if (tree_tex < 0) return tree_tex;
    const float up[3] = {0, 5, 0};
    const int tree = rt_scene_add_mesh_file(s, (dir + "tree.obj").c_str(), 10, diffuse(tree_tex), 0, 90, 1, up);
    if (tree < 0) return tree;
    for (int i = 0; i < 15; i++) {
        const float radius = std::sin(i * 78.0f) * 1.9f + 1.0f;
        const float c[3] = {i * 2.5f + 10, radius, 15.0f * std::sin(i * 34.4f)};
        // glm::sin(i) with an int argument is std::sin(int): evaluated in double, narrowed by the vec3 constructor
        const float si = (float)std::sin((double)i);
        const float col[3] = {0.5f + si * 0.5f, 0.5f + std::sin(i * 1.7f) * 0.5f, 0.5f + si * 0.5f};
        rt_scene_add_sphere(s, c, radius, diffuse(rt_scene_add_texture_color(s, col)));
    }
    const float green[3] = {0.2f, 0.6f, 0.3f};   // World.cpp:17
    rt_scene_add_plane(s, 0, rt_scene_add_material(s, RT_MAT_METAL, rt_scene_add_texture_checker(s, white, green), nullptr));
    return RT_OK;
}

static int rt_scene_build_impl(rt_scene* s, uint32_t build_key) {
    if (!s) return fail(RT_E_INVALID, "null scene");
    if (s->built) return fail(RT_E_INVALID, "scene is already built");
    if (s->top.empty()) return fail(RT_E_INVALID, "scene has no bounded objects");
    if (s->sky.w <= 0) return fail(RT_E_INVALID, "scene has no skybox");
    s->rng_key = build_key;
    s->rng_ctr = 0;
    // constructors in creation order: each mesh builds its triangle BVH when it is created (World.cpp:26), wrappers after it
    const size_t n0 = s->objs.size();
    for (size_t i = 0; i < n0; i++) {
        if (s->objs[i].kind == MESH) finish_mesh(*s, (int)i);
        else if (s->objs[i].kind == ROT) finish_rotation(*s, (int)i);
    }
    std::vector<int> list = s->top;
    s->root = build_node(*s, list, 0, (int)list.size());   // World.cpp:44
    flatten(*s);
    s->built = true;
    return RT_OK;
}

int rt_scene_dump_top(const rt_scene* s, int max_nodes, float* boxes6, int32_t* left, int32_t* right, int32_t* leaf_ids, int32_t* n_leaves) {
    if (!s || !s->built) return fail(RT_E_INVALID, "scene is not built");
    Dump d(*s);
    d.rec(s->root);
    const int n = (int)d.left.size();
    if (n > max_nodes) return fail(RT_E_INVALID, "dump buffer too small");
    memcpy(boxes6, d.boxes.data(), sizeof(float) * 6 * n);
    memcpy(left, d.left.data(), sizeof(int) * n);
    memcpy(right, d.right.data(), sizeof(int) * n);
    for (size_t i = 0; i < d.leaves.size(); i++) leaf_ids[i] = s->objs[d.leaves[i]].id;
    *n_leaves = (int)d.leaves.size();
    return n;
}
static int find_top(const rt_scene* s, int object_id) {
    for (int idx : s->top) if (s->objs[idx].id == object_id) return idx;
    return -1;
}
int rt_scene_dump_mesh(const rt_scene* s, int object_id, int max_nodes, float* boxes6, int32_t* left, int32_t* right, int max_tris,
                       float* tris24, int32_t* n_tris) {
    if (!s || !s->built) return fail(RT_E_INVALID, "scene is not built");
    int idx = find_top(s, object_id);
    while (idx >= 0 && (s->objs[idx].kind == MOVE || s->objs[idx].kind == ROT)) idx = s->objs[idx].target;
    if (idx < 0 || s->objs[idx].kind != MESH) return fail(RT_E_INVALID, "object is not a mesh");
    Dump d(*s);
    d.rec(s->objs[idx].root);
    const int n = (int)d.left.size();
    if (n > max_nodes || (int)d.leaves.size() > max_tris) return fail(RT_E_INVALID, "dump buffer too small");
    memcpy(boxes6, d.boxes.data(), sizeof(float) * 6 * n);
    memcpy(left, d.left.data(), sizeof(int) * n);
    memcpy(right, d.right.data(), sizeof(int) * n);
    for (size_t i = 0; i < d.leaves.size(); i++) {
        const Obj& t = s->objs[d.leaves[i]];
        for (int k = 0; k < 3; k++) {
            float* r = tris24 + 24 * i + 8 * k;
            r[0] = t.p[k].x; r[1] = t.p[k].y; r[2] = t.p[k].z; r[3] = t.tc[k].x; r[4] = t.tc[k].y;
            r[5] = t.n[k].x; r[6] = t.n[k].y; r[7] = t.n[k].z;
        }
    }
    *n_tris = (int)d.leaves.size();
    return n;
}
int rt_scene_dump_objects(const rt_scene* s, int max_obj, float* out) {
    if (!s || !s->built) return fail(RT_E_INVALID, "scene is not built");
    int count = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int idx : (pass == 0 ? s->top : s->planes)) {
            const Obj& o = s->objs[idx];
            if (o.id >= max_obj) return fail(RT_E_INVALID, "dump buffer too small");
            float* r = out + 20 * o.id;
            std::fill(r, r + 20, 0.0f);
            int material = o.material;
            if (o.kind == SPHERE) { r[0] = 0; r[1] = o.c.x; r[2] = o.c.y; r[3] = o.c.z; r[4] = o.r; }
            else if (o.kind == CUBE) { r[0] = 1; r[1] = o.box.mn.x; r[2] = o.box.mn.y; r[3] = o.box.mn.z; r[4] = o.box.mx.x; r[5] = o.box.mx.y; r[6] = o.box.mx.z; }
            else if (o.kind == PLANE) { r[0] = 4; r[1] = o.y; }
            else {
                r[0] = 3;
                const Box3 b = box_of(*s, idx);
                r[1] = b.mn.x; r[2] = b.mn.y; r[3] = b.mn.z; r[4] = b.mx.x; r[5] = b.mx.y; r[6] = b.mx.z;
                int m = idx;
                while (s->objs[m].kind == MOVE || s->objs[m].kind == ROT) m = s->objs[m].target;
                material = s->objs[m].material;
            }
            const Mat& m = s->mats[material];
            r[9] = (float)m.type;
            if (m.tex < 0) r[10] = -1;
            else {
                const Tex& t = s->texs[m.tex];
                r[10] = (float)t.kind;
                if (t.kind != RT_TEX_IMAGE) { r[11] = t.c1.x; r[12] = t.c1.y; r[13] = t.c1.z; }
                if (t.kind == RT_TEX_CHECKER) { r[14] = t.c2.x; r[15] = t.c2.y; r[16] = t.c2.z; }
            }
            r[17] = m.emit.x; r[18] = m.emit.y; r[19] = m.emit.z;
            count++;
        }
    return count;
}
int rt_scene_dump_instance(const rt_scene* s, int object_id, float* out) {
    if (!s || !s->built) return fail(RT_E_INVALID, "scene is not built");
    int idx = find_top(s, object_id);
    if (idx < 0) return fail(RT_E_INVALID, "no such object");
    std::fill(out, out + 17, 0.0f);
    out[8] = -1;
    const Obj* o = &s->objs[idx];
    if (o->kind == MOVE) { out[0] = o->offset.x; out[1] = o->offset.y; out[2] = o->offset.z; out[15] = 1; o = &s->objs[o->target]; }
    if (o->kind == ROT) {
        out[3] = o->pivot.x; out[4] = o->pivot.y; out[5] = o->pivot.z; out[6] = o->sin_t; out[7] = o->cos_t; out[8] = (float)o->axis;
        out[9] = o->box.mn.x; out[10] = o->box.mn.y; out[11] = o->box.mn.z; out[12] = o->box.mx.x; out[13] = o->box.mx.y; out[14] = o->box.mx.z;
        out[16] = 1;
    }
    return 1;
}
int rt_scene_flat_info(const rt_scene* s, int64_t out[7]) {
    if (!s || !s->built) return fail(RT_E_INVALID, "scene is not built");
    const rt_scene_flat& f = s->flat;
    out[0] = (int64_t)f.ent_lo.size();
    out[1] = (int64_t)f.tris.size() / RT_TRI_F4;
    out[2] = (int64_t)f.inst.size() / RT_INST_F4;
    out[3] = (int64_t)f.objects.size() / RT_OBJ_F4;
    out[4] = (int64_t)f.planes.size();
    out[5] = (int64_t)f.texels.size();
    out[6] = (int64_t)(16 * (f.ent_lo.size() * 2 + f.tris.size() + f.inst.size() + f.objects.size() + f.planes.size()) + f.texels.size());
    return RT_OK;
}

// Expected number of slab tests for a random line through the root box of the tree in entries [begin, end) of a layout
// (surface-area measure: a node's box is tested when its parent's box is hit).
static double expected_box_tests(const rt_layout& L, int begin, int end) {
    auto area = [&](int i) {
        const double x = (double)L.ent_hi[i].x - L.ent_lo[i].x, y = (double)L.ent_hi[i].y - L.ent_lo[i].y, z = (double)L.ent_hi[i].z - L.ent_lo[i].z;
        return x * y + y * z + x * z;
    };
    if (begin >= end || bits_i(L.ent_hi[begin].w) != ENT_INNER) return 0.0;
    // (the range is a list of siblings — one root, or a left-out root's children: the line goes through the box around them)
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, sum = 0.0;
    for (int c = begin; c < end; c = bits_i(L.ent_hi[c].w) == ENT_INNER ? bits_i(L.ent_lo[c].w) : c + 1) {
        if (bits_i(L.ent_hi[c].w) != ENT_INNER) continue;
        sum += 1.0;
        const float mn[3] = {L.ent_lo[c].x, L.ent_lo[c].y, L.ent_lo[c].z}, mx[3] = {L.ent_hi[c].x, L.ent_hi[c].y, L.ent_hi[c].z};
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)mn[k]); hi[k] = std::max(hi[k], (double)mx[k]); }
    }
    const double root = (hi[0] - lo[0]) * (hi[1] - lo[1]) + (hi[1] - lo[1]) * (hi[2] - lo[2]) + (hi[0] - lo[0]) * (hi[2] - lo[2]);
    if (!(root > 0)) return 0.0;
    for (int i = begin; i < end; i++) {
        if (bits_i(L.ent_hi[i].w) != ENT_INNER) continue;
        const int skip = bits_i(L.ent_lo[i].w);
        int inner_children = 0;
        for (int c = i + 1; c < skip; c = bits_i(L.ent_hi[c].w) == ENT_INNER ? bits_i(L.ent_lo[c].w) : c + 1)
            if (bits_i(L.ent_hi[c].w) == ENT_INNER) inner_children++;
        sum += inner_children * area(i) / root;
    }
    return sum;
}
// The entries of a layout as the host holds them (rt_flat.h): per entry lo.xyzw, hi.xyzw, then the peer entry in the other
// layout and the rank of a leaf.  Returns the number of entries (the arrays may be null to ask for it), or an error.
int rt_scene_dump_entries(const rt_scene* s, int regrouped, int max_entries, float* lo_hi8, int32_t* peer, uint32_t* rank) {
    if (!s || !s->built) return fail(RT_E_INVALID, "scene is not built");
    const rt_scene_flat& f = s->flat;
    if (regrouped && !f.has_fast) return fail(RT_E_INVALID, "the scene has no regrouped layout");
    if (regrouped == 2 && !f.has_wide) return fail(RT_E_INVALID, "the scene has no layout with wide nodes");
    const rt_layout& L = regrouped == 2 ? f.wide : regrouped ? f.fast : static_cast<const rt_layout&>(f);
    const int n = (int)L.ent_lo.size();
    if (!lo_hi8 && !peer && !rank) return n;
    if (n > max_entries) return fail(RT_E_INVALID, "entry buffer too small");
    for (int i = 0; i < n; i++) {
        if (lo_hi8) { memcpy(lo_hi8 + 8 * (size_t)i, &L.ent_lo[i], 16); memcpy(lo_hi8 + 8 * (size_t)i + 4, &L.ent_hi[i], 16); }
        if (peer) peer[i] = L.ent_peer[i];
        if (rank) rank[i] = L.ent_rank[i];
    }
    return n;
}
int rt_scene_regroup_info(const rt_scene* s, double out[8]) {
    if (!s || !s->built || !out) return fail(RT_E_INVALID, "scene is not built");
    const rt_scene_flat& f = s->flat;
    for (int k = 0; k < 8; k++) out[k] = 0;
    out[0] = f.has_fast ? 1 : 0;
    out[1] = f.regrouped_trees;
    out[2] = f.regrouped_groups;
    out[3] = expected_box_tests(f, 0, f.top_end);
    for (size_t k = 0; k + RT_INST_F4 <= f.inst.size(); k += RT_INST_F4) out[4] += expected_box_tests(f, bits_i(f.inst[k + 2].z), bits_i(f.inst[k + 2].w));
    if (f.has_fast) {
        out[5] = expected_box_tests(f.fast, 0, f.fast.top_end);
        for (size_t k = 0; k + RT_INST_F4 <= f.fast.inst.size(); k += RT_INST_F4) out[6] += expected_box_tests(f.fast, bits_i(f.fast.inst[k + 2].z), bits_i(f.fast.inst[k + 2].w));
        out[7] = f.fast.stack_need;
    }
    return RT_OK;
}

int rt_scene_wide_info(const rt_scene* s, double out[4]) {
    if (!s || !s->built || !out) return fail(RT_E_INVALID, "scene is not built");
    const rt_scene_flat& f = s->flat;
    for (int k = 0; k < 4; k++) out[k] = 0;
    if (!f.has_wide) return RT_OK;
    out[0] = 1;
    out[1] = (double)f.wide.ent_lo.size();
    for (size_t k = 0; k + RT_INST_F4 <= f.wide.inst.size(); k += RT_INST_F4) out[2] += expected_box_tests(f.wide, bits_i(f.wide.inst[k + 2].z), bits_i(f.wide.inst[k + 2].w));
    for (size_t i = 0; i < f.wide.ent_lo.size(); i++) {
        if (bits_i(f.wide.ent_hi[i].w) != ENT_INNER) continue;
        int kids = 0;
        for (int c = (int)i + 1; c < bits_i(f.wide.ent_lo[i].w); c = bits_i(f.wide.ent_hi[c].w) == ENT_INNER ? bits_i(f.wide.ent_lo[c].w) : c + 1) kids++;
        out[3] = std::max(out[3], (double)kids);
    }
    return RT_OK;
}

static int rt_load_obj_impl(const char* path, int max_verts, float* verts8) {
    if (!path) return fail(RT_E_INVALID, "null path");
    std::vector<float> v;
    std::string err;
    if (!load_obj(path, v, err)) return fail(RT_E_IO, err);
    const int n = (int)v.size() / 8;
    if (n > max_verts) return fail(RT_E_INVALID, "vertex buffer too small");
    if (verts8) memcpy(verts8, v.data(), v.size() * sizeof(float));
    return n;
}
static int rt_load_png_impl(const char* path, int channels, int flip, int64_t max_bytes, uint8_t* out, int32_t* w, int32_t* h, int32_t* file_channels) {
    if (!path || !w || !h || !file_channels) return fail(RT_E_INVALID, "null argument");
    std::vector<uint8_t> px;
    std::string err;
    int ww, hh, fc;
    if (!load_png(path, channels, flip != 0, px, ww, hh, fc, err)) return fail(RT_E_IO, err);
    *w = ww; *h = hh; *file_channels = fc;
    if ((int64_t)px.size() > max_bytes) return fail(RT_E_INVALID, "pixel buffer too small");
    if (out) memcpy(out, px.data(), px.size());
    return (int)px.size();
}

// Camera::SetForwardVector (DataUtility.cpp:10-15): u and v are not normalised
static void camera_basis(Vec3 look, float* cam12) {
    const Vec3 f = normalize(look), u = cross(Vec3{0, 1, 0}, f), v = cross(f, u);
    cam12[3] = f.x; cam12[4] = f.y; cam12[5] = f.z; cam12[6] = u.x; cam12[7] = u.y; cam12[8] = u.z; cam12[9] = v.x; cam12[10] = v.y; cam12[11] = v.z;
}
void rt_camera_look(const float pos[3], const float look[3], float cam12[12]) {
    cam12[0] = pos[0]; cam12[1] = pos[1]; cam12[2] = pos[2];
    camera_basis({look[0], look[1], look[2]}, cam12);
}
void rt_camera_input(float cam[12], float xrel, float yrel, int key_w, int key_a, int key_s, int key_d) {
    Vec3 pos{cam[0], cam[1], cam[2]};
    if (xrel != 0.0f || yrel != 0.0f) {   // App.cpp:98, MOUSE_SENSITIVITY = 0.001 (Constants.h:22)
        const Vec3 f{cam[3], cam[4], cam[5]}, u{cam[6], cam[7], cam[8]}, v{cam[9], cam[10], cam[11]};
        camera_basis(f - (u * xrel + v * yrel) * 0.001f, cam);
    }
    const float speed = 0.25f;            // MOVEMENT_SPEED (Constants.h:21), App.cpp:114-137
    Vec3 du{cam[6], 0, cam[8]}, df{cam[3], 0, cam[5]};
    if (key_d) pos = pos - du * speed;
    if (key_a) pos = pos + du * speed;
    if (key_w) pos = pos + df * speed;
    if (key_s) pos = pos - df * speed;
    cam[0] = pos.x; cam[1] = pos.y; cam[2] = pos.z;
}

// The entry points above that allocate host memory, behind rt::guard: no exception crosses the C boundary.
int rt_scene_add_texture_image(rt_scene* s, const uint8_t* rgba, int w, int h) { return guard([&] { return rt_scene_add_texture_image_impl(s, rgba, w, h); }); }
int rt_scene_add_texture_file(rt_scene* s, const char* path) { return guard([&] { return rt_scene_add_texture_file_impl(s, path); }); }
int rt_scene_add_mesh(rt_scene* s, const float* verts8, int n_tri, float size, int material, int rot_axis, float degrees,
                      int has_move, const float offset[3]) { return guard([&] { return rt_scene_add_mesh_impl(s, verts8, n_tri, size, material, rot_axis, degrees, has_move, offset); }); }
int rt_scene_add_mesh_file(rt_scene* s, const char* path, float size, int material, int rot_axis, float degrees, int has_move,
                           const float offset[3]) { return guard([&] { return rt_scene_add_mesh_file_impl(s, path, size, material, rot_axis, degrees, has_move, offset); }); }
int rt_scene_set_skybox(rt_scene* s, const uint8_t* rgb, int w, int h) { return guard([&] { return rt_scene_set_skybox_impl(s, rgb, w, h); }); }
int rt_scene_set_skybox_file(rt_scene* s, const char* path) { return guard([&] { return rt_scene_set_skybox_file_impl(s, path); }); }
int rt_scene_add_alt_objects(rt_scene* s, const char* asset_dir, float time) { return guard([&] { return rt_scene_add_alt_objects_impl(s, asset_dir, time); }); }
int rt_scene_add_default_objects(rt_scene* s, const char* asset_dir, float time) { return guard([&] { return rt_scene_add_default_objects_impl(s, asset_dir, time); }); }
int rt_scene_build(rt_scene* s, uint32_t build_key) { return guard([&] { return rt_scene_build_impl(s, build_key); }); }
int rt_load_obj(const char* path, int max_verts, float* verts8) { return guard([&] { return rt_load_obj_impl(path, max_verts, verts8); }); }
int rt_load_png(const char* path, int channels, int flip, int64_t max_bytes, uint8_t* out, int32_t* w, int32_t* h, int32_t* file_channels) { return guard([&] { return rt_load_png_impl(path, channels, flip, max_bytes, out, w, h, file_channels); }); }

}  // extern "C"
