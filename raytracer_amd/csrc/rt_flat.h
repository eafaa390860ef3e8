// Device layout of a flattened scene.  Everything is arrays of 16-byte records so that each lane fetches a record with
// one or two dwordx4 loads; integer fields are stored as the bit pattern of a float lane.
//
// Entries — the two BVH levels of the reference (BVH_Node over objects, Object.h:59-71; BVH_Node over a mesh's triangles,
// Object.cpp:488) flattened in depth-first preorder, left child first, exactly the order BVH_Node::Intersect
// (Object.cpp:241-252) visits them.  An inner node carries the index to continue at when its box is missed ("skip"), so
// traversal needs no stack:  i = box_hit ? i + 1 : skip.   A node whose two children are the same object (span 1,
// Object.cpp:198-200) lists that object once: testing it twice cannot change the result.
// Host arrays (rt_scene_flat::ent_lo / ent_hi), one pair of records per entry:
//   kind INNER    lo = (min.x, min.y, min.z, skip)        hi = (max.x, max.y, max.z, kind)
//   kind SPHERE   lo = (c.x, c.y, c.z, radius)            hi = (object id, 0, 0, kind)
//   kind CUBE     lo = (min.x, min.y, min.z, object id)   hi = (max.x, max.y, max.z, kind)
//   kind FOG      like CUBE: the volume's boundary; -1 / density is in the object table
//   kind TRI      lo = (triangle index, 0, 0, 0)          hi = (0, 0, 0, kind)
//   kind INSTANCE lo = (instance index, 0, 0, 0)          hi = (0, 0, 0, kind)
//   kind END      closes the top-level list and every instance range; kind IDLE is one extra entry that lanes
//                 without a ray stand on.  A lane therefore always has an entry to load and its kind says what to do.
// The top-level list occupies [0, top_end], entry top_end being its END; each mesh instance owns a range after it.
//
// Device entries (what the engine uploads, 8 dwords): dword 6 = kind << 28 | payload, so that one compare tells inner
// entries (kind 0) from everything else and an inner entry's dword 6 IS its skip index.
//   INNER      min.x max.x | min.y max.y | min.z max.z | skip | rho      the bounds of an axis side by side: the operand
//                                                                        pairs of packed FP32 arithmetic; rho: see rt_device.h,
//                                                                        "distance culling" (ENTRY_BOX_MARK for an entry box)
//   INNER_INV  same six bounds | kind | skip      a box with min > max on some axis (unions over negative-radius spheres,
//                                                 Object.cpp:29-35,156-166): tested by the general slab test
//   SPHERE     c.x c.y c.z radius 0 0 | kind, object id | 0
//   CUBE       min.xyz max.xyz | kind, object id | 0
//   FOG        min.xyz max.xyz | kind, object id | instance the volume belongs to, or -1
//   TRI        0 0 0 0 0 0 | kind, triangle index | instance the triangle belongs to
//   INSTANCE   0 0 0 0 0 0 | kind, instance index | 0
//   END, IDLE  0 ... | kind | 0
//
// Ordered entries — a second copy of the device entries for scenes that are read from memory (rt_device.h, "ordered
// traversal"): the same indices and the same leaf entries; an inner entry names its SECOND child instead of the entry
// to skip to (its first child is still the next entry), because the traversal keeps the child it visits later on a
// per-ray stack and therefore can take the two in either order:
//   INNER      six bounds | link of the second child (LINK_NONE when the node lists one child) | rho, its low two
//              mantissa bits replaced by the axis along which the first child's box begins no later than the second's
//              (3 for a single child; rho is rounded up first, +inf becomes a NaN: both keep "never cull less")
//   INNER_INV  six bounds | kind | link of the second child
// A link is an entry index, or LINK_LDS | k for the k-th record of the table of the tree's upper levels that every
// block copies into LDS: ten dwords each — six bounds | link of the second child | rho and axis as above | link of the
// first child | 0.
//
// Triangle (RT_TRI_F4 = 6 records):  (v0.xyz, -) (v1.xyz, -) (v2.xyz, -) (uv0.xy, uv1.xy) (uv2.xy, n.x, n.y) (n.z, -, -, -)
//   on the device:                    (v0.xyz, v1.x) (v1 - v0, v1.y) (v2 - v0, v1.z) (uv0.xy, uv1.xy) (uv2.xy, n.x, n.y) (n.z, v2.xyz)
// Instance (RT_INST_F4 = 4):         (offset.xyz, has_move) (pivot.xyz, rot axis or -1) (sin, cos, begin, end)
//                                    (object id, stale entry, word offset of the texture's alpha bit mask or -1, height << 16 | width)   [the last two: device only]
//   on the device `end` is replaced by the index of the entry that follows the instance's own entry in the top-level list
// Object   (RT_OBJ_F4 = 4):          (material type, texture kind, image index, -1 / density of a fog volume) (colour1.rgb, 0) (colour2.rgb, 0) (emit.rgb, 0)
// Plane    (1):                      (y, object id, 0, 0)
#pragma once
#include <cstdint>

namespace rt {

struct F4 { float x, y, z, w; };
struct ImageRec { uint32_t offset, w, h, channels; };   // byte offset into the texel pool

// kinds 1..4 are the leaves a lane notes and walks past (the traversal loop tests for that range)
enum : int { ENT_INNER = 0, ENT_SPHERE = 1, ENT_CUBE = 2, ENT_TRI = 3, ENT_FOG = 4, ENT_END = 5, ENT_IDLE = 6, ENT_INNER_INV = 7, ENT_INSTANCE = 8 };
constexpr int RT_TRI_F4 = 6, RT_INST_F4 = 4, RT_OBJ_F4 = 4;
constexpr int SK_SHIFT = 28;
constexpr uint32_t SK_MASK = (1u << SK_SHIFT) - 1u;

// dword 7 of an inner entry that is a mesh instance's entry box (rt_scene.cpp, "entry boxes"): a NaN where other inner
// entries have their culling parameter ("never culled", like the +inf of every box over an instance), told apart by its
// payload; the ordered entries keep the axis in the two low bits
constexpr uint32_t ENTRY_BOX_MARK = 0x7FC00010u;

constexpr uint32_t LINK_LDS = 1u << 26, LINK_INDEX = LINK_LDS - 1u, LINK_NONE = LINK_INDEX;
constexpr int TOP_REC_DWORDS = 10;
// Pair records (rt_device.h, "pair steps"): an inner node both of whose children have boxes, as the two children's boxes
// side by side, 16 dwords:  A.min.x A.max.x A.min.y A.max.y | A.min.z A.max.z B.min.x B.max.x | B.min.y B.max.y B.min.z
// B.max.z | link of A, link of B, rho of A, rho of B.   A link with LINK_PAIR set names a pair record (LINK_LDS: its place
// in the table in LDS; otherwise the node's entry index, which also indexes the array of records); without it, an entry.
// In the entries that go with pair records a leaf whose sibling leaf follows at the next index has LEAF_MORE set in
// dword 7, and an inner entry that has a pair record names it in dword 6 (where the others name their second child):
// ENTRY_HAS_PAIR | the record's place (LINK_LDS | k, or the entry's own index) — inside the 28-bit payload, so that the
// entry's kind still reads as INNER (LINK_PAIR itself would read as a leaf's kind there).
constexpr uint32_t LINK_PAIR = 1u << 30, LEAF_MORE = 1u << 31, ENTRY_HAS_PAIR = 1u << 27;
constexpr int PAIR_REC_DWORDS = 16;

inline float bits_f(int32_t i) { union { int32_t i; float f; } u; u.i = i; return u.f; }
inline int32_t bits_i(float f) { union { int32_t i; float f; } u; u.f = f; return u.i; }

}  // namespace rt
