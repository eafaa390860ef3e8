// MI355X engine: wavefront-batched path tracing of the reference's per-pixel path.
//
// One batch of paths (a path = one pixel sample) is advanced bounce by bounce; every pass is two kernels over a ray queue:
//
//   intersect : persistent waves pull rays from the queue (pass 0: generate them from the camera, World.cpp:68-88) and
//               walk the stackless two-level BVH with the scene staged in LDS.  Lanes do not stop at leaves: they note
//               them and walk on; primitive tests run in batches; finished lanes are refilled with new rays, so a
//               long traversal does not idle the other 63 lanes (rt_device.h: box_steps, prim_step, walk_round).
//               Writes one 16-byte hit record per ray.
//   shade     : one lane per ray: finish the hit (point, normal, uv), material switch, texture lookup, bounce sampling
//               (World.cpp:121-170); survivors are compacted into the other queue with a wave64 ballot + prefix count;
//               finished paths unwind their colour stack and store the sample's radiance.
//   resolve   : per pixel, sum the samples in the reference's order, tone-map, pack [A,B,G,R] (World.cpp:94-100).
//
// A path owns a slot in the batch, slot = pixel * spp + sample: the 64 lanes of a wave start on the samples of one pixel,
// as coherent as primary rays get (enumerating the paths sample-major instead costs the first intersect pass 20 %).
// Its RNG stream is keyed by (frame seed, pixel, sample), so the result of a slot does not depend on scheduling.  The
// recursion of World::GetRayColor multiplies colours on the way back (innermost first); to reproduce that rounding order
// every bounce records where its texture colour is (four bytes: with the ray for the latest bounce, in a per-slot row for
// the ones before) and the terminating pass looks the colours up and multiplies them in, innermost first.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_engine.h"
#include "rt_device.h"
#include "rt_host.h"

using namespace rtd;

// Staging memory of rt_engine_trace_static, the closest-hit query that allocates nothing: the rays' origins and
// directions on the way in, hit ids and hit records on the way out, 64 bytes per ray.  One array per GPU the library
// runs on, held until the process ends.  Host code never uses this name as an address: every engine asks the runtime
// for its device's copy once (hipGetSymbolAddress in init_engine) and hands that pointer to copies and kernels.
constexpr size_t TRACE_STATIC_BYTES_PER_RAY = 12 + 12 + 4 + 36;
static __device__ __align__(256) unsigned char scratch[(size_t)RT_TRACE_STATIC_MAX_RAYS * TRACE_STATIC_BYTES_PER_RAY];

namespace {

constexpr int IBLOCK = 512;         // intersect: 8 waves per block; two blocks per CU, each with its own copy of the scene in LDS
constexpr long long WIDE_MIN_PATHS = 1ll << 24;   // batches of at least this many paths run their large passes with 768-thread blocks
constexpr int OBLOCK = 1024;        // intersect, ordered traversal: one block per CU, its stacks and the tree's upper levels share the CU's LDS
constexpr int MAX_ORD_BYTES = 160 * 1024 - 256;
constexpr int MAX_STAGED_BYTES = 80 * 1024;   // largest scene copy a block keeps in LDS (two blocks share a CU's 160 KiB)
// Reservation counters of the intersect passes: the queue of a pass is cut into NCUR ranges with a counter each (a single
// counter would take every reservation of the grid, and additions to one address are served one after another); the
// counters of a pass lie CUR_STRIDE words apart so that they fall into different memory channels.
constexpr int NCUR = 32, CUR_STRIDE = 1024;
constexpr int STRIPE = 512;          // positions per stripe; range r of a queue is every NCUR-th stripe, starting with stripe r
constexpr int COUNT_WORDS = CUR_STRIDE * (1 + NCUR);   // per batch: ray counts per pass, then NCUR rows of counters
constexpr int BOX_COUNT_AT = 512;   // words BOX_COUNT_AT + 2k, + 2k + 1 of a batch's first row: box tests executed by pass k (64 bits)
constexpr int EXC_COUNT_AT = 256;   // word EXC_COUNT_AT + k of a batch's first row: rays of pass k handed over to exceptions_kernel
constexpr long long MAX_BATCHES = 4096;   // batches per frame an engine reserves pass counters for (0.5 GiB of counters at most)
constexpr int STILE = 1024;         // shade: queue positions a block handles per round (survivors staged in 32 KiB of LDS)
// intersect scheduling knobs (defaults in rt_engine_create; RT_REFILL_AT / RT_QUORUM / RT_CHUNK override them for tuning)
struct Tune {
    int refill_at;   // a wave fetches new rays once this many lanes are idle (pass 0: generating a camera ray is expensive)
    int refill_at_bounce;   // the same for the bounce passes, where fetching a ray is two loads
    int quorum;      // lanes on inner entries needed to repeat the box-test phase
    int chunk;       // queue positions a wave reserves per atomic (shrunk when a pass holds few rays)
    int prim_quorum; // lanes with an untested leaf needed to repeat the primitive-test phase
    int min_chunk;      // smallest reservation (64 to 128 is best: thinner spreading leaves lanes idle, thicker starves waves)
    int guide;          // a reservation is 1/guide of an even share of the rays still in the queue
    int finish_quorum;  // lanes whose triangle waits for the second part of its test needed to run that part (1 measured best:
                        // a waiting lane tests no further leaf, and that costs more than running the part for few lanes)
};

// Division of a 32-bit unsigned number by a divisor fixed for the whole launch: n / d = (((n - q) >> 1) + q) >> shift with
// q = mulhi(n, magic) — four instructions instead of the ~40 of a hardware-less integer division.  make_fastdiv runs on
// the host; d = 1 is flagged (the scheme needs d >= 2).
struct FastDiv { uint32_t magic, shift, d; };
__host__ inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{0, 0, d};
    if (d < 2) return f;
    uint32_t L = 31;
    while (!((d >> L) & 1u)) L--;                       // floor(log2 d)
    if ((d & (d - 1)) == 0) { f.magic = 0; f.shift = L - 1; return f; }
    const unsigned long long two = 1ull << (32 + L);
    unsigned long long m = two / d;
    const unsigned long long rem = two % d;
    m = 2 * m + ((2 * rem >= d) ? 1 : 0);
    f.magic = (uint32_t)(m + 1);
    f.shift = L;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    const uint32_t q = __umulhi(n, f.magic);
    const uint32_t t = (((n - q) >> 1) + q) >> f.shift;
    return f.d == 1 ? n : t;
}

struct Cam { V3 pos, fwd, u, v; };
struct Cfg {
    int W, H, S, B;
    float fov, sky, dof, focus;
    FastDiv div_S, div_spp;
    // frame constants of camera-ray generation, computed once on the host with the reference's expressions (World.cpp:70)
    float one_x, one_y, aspect;
    float rW, rH, rS, rfocus;   // RN(1/W), RN(1/H), RN(1/S), RN(1/focus): reciprocals for div_by
    int fast_div;               // the divisions by W, H, S and focus may go through div_by (exact for these operand ranges)
};
// which pixels a batch covers: pixels are enumerated tile by tile over the tiles this rank owns
struct Batch {
    int first_pixel, n_pixels;       // in the rank's pixel enumeration
    int rank, n_ranks, tiles_x, n_tiles, tile_w, tile_h;
    int wx0, wy0, ww, wh;            // render window inside the frame (the whole frame unless a test narrows it)
    uint32_t seed;
    FastDiv div_per_tile, div_tile_w, div_tiles_x;
    // cancellation: the launches of one rt_engine_render carry its serial number; rt_engine_cancel stores the serial of
    // the render in flight in *cancel, and every block that finds its own serial there when it starts leaves at once: a
    // cancelled frame costs what is left of the launch that is running (reading the word with every reservation of
    // rays as well cost 1 % of a frame and bought nothing a viewer would notice)
    const uint32_t* cancel;
    uint32_t serial;
};
__device__ __forceinline__ bool cancelled(const Batch& bt) {
    return __hip_atomic_load(bt.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == bt.serial;
}
struct Queues {
    float4* a[2]; float4* b[2];      // ray records: a = (o.xyz, d.x), b = (d.y, d.z, slot, ctr | depth << 24); b[k] = a[k] + capacity
    float4* hit;                     // hit records by queue position: (distance, best entry, u, v)
    uint32_t* hit_ctr;               // scenes with fog: the path's random-number counter after the traversal
    uint32_t* exc;                   // queue positions of the rays of the running pass that were handed over (regrouped layouts)
    uint32_t* counts;                // counts[k] = rays entering pass k
    uint32_t* cursors;               // cursors[r * CUR_STRIDE + k] = positions of range r of pass k handed out to intersect waves so far
    uint32_t* c[2];                  // by queue position, beside the ray records: where the colour of the path's latest bounce is (texture_ref)
    uint32_t* stack;                 // [slot][depth]: the same for the bounces before the latest (paths of two bounces and more)
    int stack_depth;                 // depths per slot (the engine's bounce count)
    float4* radiance;                // [slot] = sample radiance
    int capacity;                    // slots
};
struct Debug { int32_t* primary_id; float* primary_t; uint32_t* rng_draws; const uint32_t* keys; unsigned long long* prof; };

__device__ __forceinline__ bool pixel_of(const Batch& bt, int lp, int& px, int& py) {
    const int per_tile = bt.tile_w * bt.tile_h;
    const int t = (int)fdiv((uint32_t)lp, bt.div_per_tile), in = lp - t * per_tile;
    const int tile = bt.rank + t * bt.n_ranks;
    if (tile >= bt.n_tiles) return false;
    const int ty = (int)fdiv((uint32_t)tile, bt.div_tiles_x), tx = tile - ty * bt.tiles_x;
    const int row = (int)fdiv((uint32_t)in, bt.div_tile_w);
    py = ty * bt.tile_h + row;
    px = tx * bt.tile_w + (in - row * bt.tile_w);
    if (px >= bt.ww || py >= bt.wh) return false;
    px += bt.wx0; py += bt.wy0;
    return true;
}

// World::CalculateColorForScreenPosition, one sample (World.cpp:70-88)
__device__ __forceinline__ Ray camera_ray(const Cam& cam, const Cfg& g, int px, int py, int sx, int sy, Rng& rng) {
    // quotients by frame constants: the ray's numerators are small integers or lens offsets (0 or at least 2^-36 in
    // magnitude), inside the range in which div_by reproduces the division exactly
    const bool fast = g.fast_div != 0;
    const float fW = (float)g.W, fH = (float)g.H, fS = (float)g.S;
    const float qx = fast ? div_by((float)px, fW, g.rW) : (float)px / fW, qy = fast ? div_by((float)py, fH, g.rH) : (float)py / fH;
    const float fx = fast ? div_by((float)sx, fS, g.rS) : (float)sx / fS, fy = fast ? div_by((float)sy, fS, g.rS) : (float)sy / fS;
    const float off_x = -(qx - 0.5f) * g.aspect * g.fov, off_y = -(qy - 0.5f) * g.fov;
    Ray r;
    r.d = cam.fwd + (cam.u * (off_x + g.one_x * fx) + cam.v * (off_y + g.one_y * fy));
    r.d = normalize(r.d);
    float lx, ly;
    unit_circle_point(rng, lx, ly);
    lx = lx * g.dof; ly = ly * g.dof;
    const V3 wo = cam.u * lx + cam.v * ly;
    r.o = cam.pos + wo;
    r.d = r.d - (fast ? mk(div_by(wo.x, g.focus, g.rfocus), div_by(wo.y, g.focus, g.rfocus), div_by(wo.z, g.focus, g.rfocus)) : divs(wo, g.focus));
    r.d = normalize(r.d);
    return r;
}

// The random stream of the path in `slot`: keyed by (frame seed, pixel, sample) (or given by the caller of a ray query)
__device__ __forceinline__ uint32_t slot_key(const Cfg& g, const Batch& bt, const Debug& dbg, uint32_t slot) {
    if (dbg.keys) return dbg.keys[slot];
    const int spp = g.S * g.S;
    const int lp = (int)fdiv(slot, g.div_spp), s = (int)(slot - (uint32_t)lp * (uint32_t)spp);
    int px = 0, py = 0;
    pixel_of(bt, bt.first_pixel + lp, px, py);
    return sample_key(bt.seed, (uint32_t)(py * g.W + px), (uint32_t)s);
}

// The ray at queue position `idx` of a pass.  FIRST: generated from the camera (slot = idx); otherwise read from queue
// `qin`.  Returns false for a position that holds no ray (a pixel outside the frame in a partial tile).
template <bool FIRST>
__device__ __forceinline__ bool fetch_ray(const Cam& cam, const Cfg& g, const Batch& bt, const Queues& q, int qin, const Debug& dbg,
                                          uint32_t idx, Ray& ray, Rng& rng, uint32_t& slot, uint32_t& depth, bool need_rng = true) {
    const int spp = g.S * g.S;
    if (FIRST) {
        slot = idx; depth = 0;
        const int lp = (int)fdiv(idx, g.div_spp), s = (int)(idx - (uint32_t)lp * (uint32_t)spp);
        int px, py;
        if (!pixel_of(bt, bt.first_pixel + lp, px, py)) return false;
        rng.key = sample_key(bt.seed, (uint32_t)(py * g.W + px), (uint32_t)s);
        rng.ctr = 0;
        const int sx = (int)fdiv((uint32_t)s, g.div_S);
        ray = camera_ray(cam, g, px, py, sx, s - sx * g.S, rng);
        return true;
    }
    const float4 a = q.a[qin][idx], b = q.b[qin][idx];
    ray.o = mk(a.x, a.y, a.z); ray.d = mk(a.w, b.x, b.y);
    slot = __float_as_uint(b.z);
    const uint32_t cd = __float_as_uint(b.w);
    rng.ctr = cd & 0xFFFFFFu; depth = cd >> 24;
    // the closest-hit search of a scene without fog draws nothing, and many rays end without a draw: those callers ask
    // for the key (slot_key) only when a draw comes up
    rng.key = need_rng ? slot_key(g, bt, dbg, slot) : 0u;
    return true;
}

// Where the traversal reads the scene from: the block's own copy in LDS (entries, triangles, instance and plane records,
// copied once; blocks are persistent), or global memory when the scene is too large for that.
// ORD_BLOCK > 0 (the ordered traversal, blocks of ORD_BLOCK threads): the block's LDS holds the threads' stacks (slot s
// of thread t at dword s * ORD_BLOCK + t) and behind them the table of the tree's upper levels, copied here once.
template <bool STAGED, int ORD_BLOCK = 0, bool PAIRS = false>
__device__ __forceinline__ Store stage_scene(const Scene& sc, float4* smem, const OrdScene* os = nullptr) {
    Store st;
    st.lds0 = nullptr; st.lds0_addr = 0; st.top_addr = 0; st.stack_addr = 0; st.root_link = 0; st.inst_end = nullptr;
    st.pair = nullptr; st.inst_start = nullptr;
    if (ORD_BLOCK) {
        uint32_t* l = reinterpret_cast<uint32_t*>(smem);
        const int stack_dwords = os->stack_slots * ORD_BLOCK, tab_dwords = os->n_top_tab * (PAIRS ? rt::PAIR_REC_DWORDS : rt::TOP_REC_DWORDS);
        for (int k = threadIdx.x; k < tab_dwords; k += ORD_BLOCK) l[stack_dwords + k] = os->top_tab[k];
        l[threadIdx.x] = (uint32_t)sc.top_end;   // slot 0: where the walk ends
        __syncthreads();
        st.ent = os->ent_ord; st.tris = sc.tris; st.inst = sc.inst; st.planes = sc.planes;
        st.root_link = os->root_link; st.inst_end = os->inst_end;
        st.pair = os->pair; st.inst_start = os->inst_start;
        st.lds0 = l; st.lds0_addr = (uint32_t)(uintptr_t)l;     // LDS addresses are 32-bit offsets
        st.stack_addr = st.lds0_addr; st.top_addr = st.lds0_addr + 4u * (uint32_t)stack_dwords;
        return st;
    }
    if (STAGED) {
        const int n_ent4 = 2 * sc.n_entries, n_tri4 = rt::RT_TRI_F4 * sc.n_tris, n_inst4 = rt::RT_INST_F4 * sc.n_inst;
        for (int k = threadIdx.x; k < n_ent4; k += blockDim.x) smem[k] = sc.ent[k];
        for (int k = threadIdx.x; k < n_tri4; k += blockDim.x) smem[n_ent4 + k] = sc.tris[k];
        for (int k = threadIdx.x; k < n_inst4; k += blockDim.x) smem[n_ent4 + n_tri4 + k] = sc.inst[k];
        for (int k = threadIdx.x; k < sc.n_planes; k += blockDim.x) smem[n_ent4 + n_tri4 + n_inst4 + k] = sc.planes[k];
        __syncthreads();
        st.ent = smem; st.tris = smem + n_ent4; st.inst = smem + n_ent4 + n_tri4; st.planes = smem + n_ent4 + n_tri4 + n_inst4;
    } else { st.ent = sc.ent; st.tris = sc.tris; st.inst = sc.inst; st.planes = sc.planes; }
    return st;
}

// offsets range r spans: STRIPE for each of its stripes (the queue's last stripe may hold fewer positions: those past the
// end are skipped when they come up)
__device__ __forceinline__ uint32_t range_size(uint32_t n_stripes, uint32_t r) { return r < n_stripes ? ((n_stripes - r + NCUR - 1) / NCUR) * STRIPE : 0u; }

// A by-value kernel argument read again from the kernarg segment (scalar loads, cached).  The persistent kernels below
// use some arguments only in a part that runs now and then (the camera, frame and batch records: 62 dwords that only the
// generation of camera rays reads); held in scalar registers for the whole kernel they push other values out into lanes
// of a vector register, and every use of those costs a vector instruction.  The pointer is passed through an empty asm
// statement so that the loads stay where they are written instead of being hoisted out of the loop.
struct IntersectArgs { Scene sc; Cam cam; Cfg g; Batch bt; Queues q; int pass; Debug dbg; Tune tune; OrdScene os; };   // the kernel's parameter list
template <typename T>
__device__ __forceinline__ T kernarg_read(size_t offset) {
    typedef const __attribute__((address_space(4))) uint32_t* KP;
    KP p = (KP)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offset);
    asm volatile("" : "+s"(p));
    T r;
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; k++) d[k] = p[k];
    return r;
}

// One dword per thread of the block in LDS, for the kernel variant that is short of registers
template <int N>
__device__ __forceinline__ uint32_t* lane_slots() { __shared__ uint32_t slots[N]; return slots; }

// A ray that the regrouped layout is not proven for (rt_scene.cpp, "regrouping"): its queue position goes on the list that
// exceptions_kernel works off after the pass.  One atomic per ray: rendered frames have no such rays.
__device__ __forceinline__ void hand_over(const Queues& q, int pass, uint32_t idx) {
    q.exc[atomicAdd(q.counts + EXC_COUNT_AT + pass, 1u)] = idx;
}

// ---- intersect (World.cpp:105-119) ----------------------------------------------------------------------------------
// (the parameter list is mirrored by IntersectArgs, through which kernarg_read finds cam, g and bt: change both together)
template <bool FIRST, bool STAGED, bool FOG, int BLOCK = IBLOCK, int MIN_WAVES = 1, bool ORD = false, bool PAIRS = false>
__global__ __launch_bounds__(BLOCK, MIN_WAVES) void intersect_kernel(Scene sc, Cam cam, Cfg g, Batch bt, Queues q, int pass, Debug dbg, Tune tune, OrdScene os) {
    static_assert(!ORD || (!STAGED && !FOG), "the ordered traversal is for scenes read from memory, without fog");
    static_assert(!PAIRS || ORD, "pair steps are a form of the ordered traversal");
    extern __shared__ float4 smem[];
    const uint32_t n_in = FIRST ? (uint32_t)bt.n_pixels * (uint32_t)(g.S * g.S) : q.counts[pass];
    // a pass with few rays: blocks beyond one smallest reservation per wave leave before they copy the scene
    const uint32_t blocks_used = min((uint32_t)gridDim.x, n_in / ((uint32_t)tune.min_chunk * (BLOCK / 64)) + 1u);
    if (n_in == 0u || blockIdx.x >= blocks_used || cancelled(bt)) return;
    const Store st = stage_scene<STAGED, ORD ? BLOCK : 0, PAIRS>(sc, smem, &os);
    const int qin = (pass + 1) & 1;
    // Range r of the queue is every NCUR-th stripe of STRIPE positions, starting with stripe r: all ranges move through
    // the queue side by side, so the grid still works on one neighbourhood of the image at a time.  A wave starts on the
    // range of its block and moves on when a range is used up.  Offset o of range r is queue position
    // ((o / STRIPE) * NCUR + r) * STRIPE + o % STRIPE.
    const uint32_t n_stripes = (n_in + STRIPE - 1) / STRIPE;
    uint32_t range = blockIdx.x % NCUR;
    // RELOAD, LEAN: see walker_special.  The lean variant also keeps the queue position of the lane's ray in LDS.
    constexpr bool RELOAD = BLOCK != IBLOCK || (!FIRST && !STAGED), LEAN = RELOAD && STAGED;
    uint32_t* my_slot = nullptr;   // wave-uniform: slot [lane] is this lane's
    if constexpr (LEAN) my_slot = lane_slots<BLOCK>() + __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
    Walker w;
    walker_idle(sc, w);
    uint32_t my_idx = 0;
    // the wave owns queue positions [mine, mine_end); it reserves CHUNK more with one atomic when they run out
    uint32_t mine = 0, mine_end = 0, mine_range = 0;   // as offsets of range mine_range
    // few rays in the pass: smaller reservations, so that every wave of the grid gets some
    const uint32_t waves = blocks_used * (BLOCK / 64);
    uint32_t chunk = (n_in / (waves * (uint32_t)tune.guide)) & ~7u;
    chunk = chunk < (uint32_t)tune.min_chunk ? (uint32_t)tune.min_chunk : (chunk > (uint32_t)tune.chunk ? (uint32_t)tune.chunk : chunk);
    bool drained = false;   // wave-uniform: every queue position has been handed out
    uint32_t n_box = 0;     // wave-uniform: slab tests this wave executed (lanes x iterations)
    constexpr bool CULL = !STAGED && !FOG;
    Prof prof;
    RT_PROF(const unsigned long long t_start = __builtin_readcyclecounter(); unsigned n_refills = 0;)
    for (;;) {
        RT_PROF(const unsigned long long t_r0 = __builtin_readcyclecounter();)
        const unsigned long long idle = __ballot(walker_is_idle(w));
        if (!drained && __popcll(idle) >= (FIRST ? tune.refill_at : tune.refill_at_bounce)) {
            if (mine == mine_end) {
                for (;;) {
                    const uint32_t size = range_size(n_stripes, range);
                    uint32_t off = 0xFFFFFFFFu;
                    if (size) {
                        if (lane_now() == 0) off = atomicAdd(q.cursors + range * CUR_STRIDE + pass, chunk);
                        off = __builtin_amdgcn_readfirstlane(off);   // (every lane is enabled here) a scalar: what follows from it stays out of the vector registers
                    }
                    if (off < size) {
                        mine = off;
                        mine_end = off + chunk < size ? off + chunk : size;
                        mine_range = range;
                        // guided: the next reservation shrinks with what is left, so that the waves run out of work together
                        const uint32_t left = (size - mine_end) * NCUR;
                        chunk = (left / (waves * (uint32_t)tune.guide)) & ~7u;
                        chunk = chunk < (uint32_t)tune.min_chunk ? (uint32_t)tune.min_chunk : (chunk > (uint32_t)tune.chunk ? (uint32_t)tune.chunk : chunk);
                        break;
                    }
                    // this range is used up: lane r looks at range r's counter, the wave moves to the range with the most left
                    uint32_t left = 0;
                    const uint32_t lane = lane_now();
                    if (lane < NCUR) {
                        const uint32_t sz = range_size(n_stripes, lane);
                        const uint32_t seen = __hip_atomic_load(q.cursors + lane * CUR_STRIDE + pass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        left = seen < sz ? sz - seen : 0u;
                    }
                    // the range with the most left, found with scalar instructions (a butterfly of lane exchanges would
                    // keep six lane numbers in vector registers through the whole kernel for this rare step)
                    uint32_t best = 0;                                    // left < 2^26: a queue holds fewer than 2^31 positions, a range 1/32 of them
#pragma unroll
                    for (int r = 0; r < NCUR; r++) {
                        const uint32_t c = __builtin_amdgcn_readlane(left, r) << 6 | (uint32_t)r;
                        best = c > best ? c : best;
                    }
                    if ((best >> 6) == 0u) break;                         // nothing left anywhere
                    range = best & 63u;
                    chunk = (uint32_t)tune.min_chunk;
                }
            }
            if (mine == mine_end) drained = true;
            else {
                const uint32_t o = mine + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));   // idle lanes below this one
                const uint32_t idx = ((o / STRIPE) * NCUR + mine_range) * STRIPE + o % STRIPE;
                if (FIRST) {
                    // camera, frame and batch records: read here, not kept in registers between refills
                    const Cam cam_now = kernarg_read<Cam>(offsetof(IntersectArgs, cam));
                    const Cfg g_now = kernarg_read<Cfg>(offsetof(IntersectArgs, g));
                    const Batch bt_now = kernarg_read<Batch>(offsetof(IntersectArgs, bt));
                    if (walker_is_idle(w) && o < mine_end && idx < n_in) {
                        Ray ray; Rng rng; uint32_t slot, depth;
                        if (fetch_ray<true>(cam_now, g_now, bt_now, q, qin, dbg, idx, ray, rng, slot, depth, FOG)) {
                            if constexpr (PAIRS) walker_start_pairs<CULL, BLOCK>(sc, st, w, ray, rng); else if constexpr (ORD) walker_start_ord<CULL, BLOCK>(sc, st, w, ray, rng); else walker_start<CULL>(sc, st, w, ray, rng);
                            const uint32_t tagged = RELOAD && w.usable ? idx | IDX_USABLE : idx;
                            if constexpr (LEAN) my_slot[lane_now()] = tagged; else my_idx = tagged;
                            if (!FOG && sc.has_ref && !w.usable && !ray_monotone(ray)) { hand_over(q, pass, idx); walker_idle(sc, w); }
                        }
                    }
                } else if (walker_is_idle(w) && o < mine_end && idx < n_in) {
                    Ray ray; Rng rng; uint32_t slot, depth;
                    if (fetch_ray<false>(cam, g, bt, q, qin, dbg, idx, ray, rng, slot, depth, FOG)) {
                        if constexpr (PAIRS) walker_start_pairs<CULL, BLOCK>(sc, st, w, ray, rng); else if constexpr (ORD) walker_start_ord<CULL, BLOCK>(sc, st, w, ray, rng); else walker_start<CULL>(sc, st, w, ray, rng);
                        const uint32_t tagged = RELOAD && w.usable ? idx | IDX_USABLE : idx;
                        if constexpr (LEAN) my_slot[lane_now()] = tagged; else my_idx = tagged;
                        if (!FOG && sc.has_ref && !w.usable && !ray_monotone(ray)) { hand_over(q, pass, idx); walker_idle(sc, w); }
                    }
                }
                // the lean variant reads the entry every lane stands on again (it is what the lane holds already: every
                // change of position ends in a load), so the entry's seven registers are free while rays are fetched
                if constexpr (LEAN) walker_load(st, w);
                const uint32_t took = (uint32_t)__popcll(idle);
                mine = mine + took < mine_end ? mine + took : mine_end;
                RT_PROF(n_refills++;)
            }
        }
        RT_PROF(prof.refill += __builtin_readcyclecounter() - t_r0;)
        if (!__any(!walker_is_idle(w))) {
            if (drained) break;
            continue;
        }
        // RELOAD — bounce passes of scenes read from memory: a lane inside an instance does not keep the ray as given, it
        // reads it again from the queue on the way out (ten registers).  With the scene in LDS only the 768-thread
        // variants do so (launch_intersect): they need the registers for their sixth wave per SIMD, while the 512-thread
        // blocks are bounded by LDS and reading the ray again cost them 0.3 ms of a 1080p, 64 spp frame.  (A fifth wave
        // per SIMD was measured on both kinds of scene and gained nothing: 256-thread blocks on the 100 000-primitive
        // scene 92.9 against 92.3 ms; 640-thread blocks with the scene in LDS lose a fifth, their ten waves do not spread
        // evenly over four SIMDs.)
        const int ended = walk_round<STAGED, FOG, RELOAD, FIRST, ORD ? BLOCK : 0, PAIRS>(sc, st, w, tune.quorum, tune.prim_quorum, tune.finish_quorum, prof, n_box, q.a[qin], q.b[qin], my_idx, q.hit, my_slot);
        if (ended == WALK_DONE) {
            // (the lean walker has written the barycentrics of its winning triangle already)
            if constexpr (LEAN) *reinterpret_cast<float2*>(q.hit + (my_slot[lane_now()] & IDX_MASK)) = make_float2(w.c.dist, __int_as_float(w.c.best));
            else q.hit[my_idx & IDX_MASK] = make_float4(w.c.dist, __int_as_float(w.c.best), w.c.u, w.c.v);
            if (FOG) q.hit_ctr[my_idx & IDX_MASK] = w.rng.ctr;
        } else if (!FOG && ended == WALK_HANDED_OVER) {
            if constexpr (LEAN) hand_over(q, pass, my_slot[lane_now()] & IDX_MASK); else hand_over(q, pass, my_idx & IDX_MASK);
        }
    }
    if (lane_now() == 0 && n_box) atomicAdd(reinterpret_cast<unsigned long long*>(q.counts + BOX_COUNT_AT) + pass, (unsigned long long)n_box);
#ifdef RT_PROFILE
    if (dbg.prof && lane_now() == 0 && prof.rounds) {   // per pass: 12 counters summed over the waves that had work
        unsigned long long* p = dbg.prof + 32 * (pass < 15 ? pass : 15);
        atomicAdd(p + 0, 1ull); atomicAdd(p + 1, __builtin_readcyclecounter() - t_start); atomicAdd(p + 2, (unsigned long long)prof.rounds);
        atomicAdd(p + 3, prof.box); atomicAdd(p + 4, prof.prim); atomicAdd(p + 5, prof.special); atomicAdd(p + 6, prof.refill);
        atomicAdd(p + 7, (unsigned long long)prof.prims); atomicAdd(p + 8, (unsigned long long)prof.prim_lanes);
        atomicAdd(p + 9, (unsigned long long)prof.specials); atomicAdd(p + 10, (unsigned long long)n_refills);
        atomicAdd(p + 11, (unsigned long long)prof.box_iters); atomicAdd(p + 12, (unsigned long long)prof.box_lanes);
        atomicAdd(p + 13, (unsigned long long)prof.finishes); atomicAdd(p + 14, (unsigned long long)prof.finish_lanes);
        for (int k = 0; k < 4; k++) { atomicAdd(p + 16 + k, (unsigned long long)prof.sp_exec[k]); atomicAdd(p + 20 + k, (unsigned long long)prof.sp_lanes[k]); }
    }
#endif
}

// ---- exceptions: the rays of a pass that intersect_kernel handed over, walked through the reference's own layout -------
// A regrouped layout (rt_scene.cpp, "regrouping") tests the leaves the reference tests for every ray for which the slab
// test is monotone in the box (rt_device.h, ray_monotone).  The others — a direction component that is exactly zero, or
// something that is not finite, in the scene's frame or in a mesh instance's: none among the rays of a rendered frame
// (the lens offsets see to that), any number among a caller's own rays — are walked here, from the start, through the
// reference's layout with the reference's slab test as written, by the same walker (scene read from memory, one ray per
// lane until all are done).  A launch that finds no such ray leaves at once.
// sc_ref: the reference's layout; its leaves name their entries in the regrouped layout, which is what a hit record holds.
template <bool FIRST>
__global__ __launch_bounds__(256) void exceptions_kernel(Scene sc_ref, Cam cam, Cfg g, Batch bt, Queues q, int pass, Debug dbg) {
    const uint32_t n = q.counts[EXC_COUNT_AT + pass];
    if (n == 0u || cancelled(bt)) return;
    Store st;
    st.ent = sc_ref.ent; st.tris = sc_ref.tris; st.inst = sc_ref.inst; st.planes = sc_ref.planes;
    st.lds0 = nullptr; st.lds0_addr = 0; st.top_addr = 0; st.stack_addr = 0; st.root_link = 0; st.inst_end = nullptr;
    st.pair = nullptr; st.inst_start = nullptr;
    const int qin = (pass + 1) & 1;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {   // (the same for a whole block)
        const uint32_t k = base + threadIdx.x;
        Walker w;
        walker_idle(sc_ref, w);
        uint32_t idx = 0;
        bool have = false;
        if (k < n) {
            idx = q.exc[k];
            Ray ray; Rng rng; uint32_t slot, depth;
            have = fetch_ray<FIRST>(cam, g, bt, q, qin, dbg, idx, ray, rng, slot, depth, false);
            if (have) walker_start<true>(sc_ref, st, w, ray, rng);
        }
        walk_wave_to_end<false>(sc_ref, st, w);
        if (have) {
            const int best = w.c.best >= 0 ? (int)leaf_extra(sc_ref, w.c.best) : w.c.best;
            q.hit[idx] = make_float4(w.c.dist, __int_as_float(best), w.c.u, w.c.v);
        }
    }
}

// ---- one bounce of World::GetRayColor (World.cpp:121-170) for a ray whose closest hit is known.  Returns true when the
// path goes on: `next` is the bounce ray, the (factor, texture colour) of this bounce is on the path's colour stack.
// Otherwise the path ends here: the stack is unwound and the sample's radiance stored.
// The colours the recursion multiplies in on the way back are looked up when the path ends (rt_device.h, texture_ref): a
// bounce records four bytes.  The latest bounce's travel with the ray (`carried`, beside its queue record: read and
// written in queue order); when the path bounces again they move to the path's own row stack[slot][depth - 1].  Four
// paths in five end with one bounce behind them and never touch that row; the others read it once, one 32-byte sector
// for up to eight bounces.  (Records of (colour, factor) by [depth][slot], as before, were 16 bytes written at every
// bounce and one scattered 16-byte read per depth at the end: a third of the shading kernels' memory traffic.)
template <bool FIRST>
__device__ __forceinline__ bool shade_ray(const Scene& sc, const Cfg& g, const Batch& bt, const Queues& q, const Debug& dbg, const Ray& ray, Rng& rng,
                                          uint32_t slot, uint32_t depth, const Closest& c, uint32_t carried, Ray& next, uint32_t& next_ref) {
    bool alive = false;
    Hit h;
    if (c.hit) finalize_hit(sc, ray, c, h);
    if (FIRST) {
        if (dbg.primary_t) dbg.primary_t[slot] = c.dist;
        if (dbg.primary_id) dbg.primary_id[slot] = c.hit ? h.object : -1;
    }
    V3 L = mk(0, 0, 0);
    if (!c.hit) L = skybox_pixel(sc, ray.d) * g.sky;
    else {
        const float4* ob = sc.objects + (size_t)h.object * rt::RT_OBJ_F4;
        const int mtype = __float_as_int(ob[0].x);
        const int remaining = g.B - (int)depth;       // the recursion's bounceAmount
        if (mtype == RT_MAT_LIGHT) L = xyz(ob[3]);
        else if (mtype != RT_MAT_NONE && remaining != 0) {
            next.o = h.point;
            if (!FIRST && mtype != RT_MAT_METAL) rng.key = slot_key(g, bt, dbg, slot);   // the bounce draws: now the stream is needed
            uint32_t half = 0u;
            if (mtype == RT_MAT_DIFFUSE) {
                next.d = normalize(h.normal + unit_sphere_point(rng));
                half = REF_HALF;                      // lightDarknerFactor (World.cpp:129)
            } else if (mtype == RT_MAT_METAL) {
                next.d = normalize(ray.d - (h.normal * 2.0f) * dot(ray.d, h.normal));
            } else next.d = unit_sphere_point(rng);    // Isotropic (World.cpp:163)
            next_ref = texture_ref(sc, ob, h.object, h) | half;
            if (!FIRST && depth != 0u) q.stack[(size_t)slot * q.stack_depth + (depth - 1u)] = carried;
            alive = true;
        }
    }
    if (!alive) {
        // unwind the recursion: colour = (factor * child) * texture, innermost first (World.cpp:138,151,166)
        if (!FIRST && depth != 0u) {
            L = (L * ((carried & REF_HALF) ? 0.5f : 1.0f)) * ref_color(sc, carried);
            const uint32_t* row = q.stack + (size_t)slot * q.stack_depth;
            for (int k = (int)depth - 2; k >= 0; k--) {
                const uint32_t r = row[k];
                L = (L * ((r & REF_HALF) ? 0.5f : 1.0f)) * ref_color(sc, r);
            }
        }
        q.radiance[slot] = make_float4(L.x, L.y, L.z, 0.0f);
        if (dbg.rng_draws) dbg.rng_draws[slot] = rng.ctr;
    }
    return alive;
}

// ---- shade + compact (World.cpp:121-170) ------------------------------------------------------------------------------
// (the parameter list begins like IntersectArgs, through which kernarg_read finds cam, g and bt: change both together)
template <bool FIRST>
__global__ __launch_bounds__(256) void shade_kernel(Scene sc, Cam cam, Cfg g, Batch bt, Queues q, int pass, Debug dbg) {
    // A block shades STILE consecutive queue positions per round and collects the surviving rays in LDS; one global
    // atomic per round then reserves their places in the outgoing queue (a per-wave atomic on the one counter would
    // serialise millions of requests on a single L2 line), and the copy-out is fully coalesced.
    __shared__ float4 s_a[STILE], s_b[STILE];
    __shared__ uint32_t s_c[STILE];
    __shared__ uint32_t s_count, s_base;
    const int qin = (pass + 1) & 1, qout = pass & 1;
    const uint32_t n_in = FIRST ? (uint32_t)bt.n_pixels * (uint32_t)(g.S * g.S) : q.counts[pass];
    const uint32_t n_tiles = (n_in + STILE - 1) / STILE;
    // once per block, not per tile: the read goes to the L2 and waiting for it before every tile cost 1.4 % of a frame
    if (cancelled(bt)) return;        // one address for the whole block: every thread takes the same way
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      if (threadIdx.x == 0) s_count = 0;
      __syncthreads();
      for (uint32_t k = 0; k < STILE / 256; k++) {
        const uint32_t i = tile * STILE + k * 256 + threadIdx.x;
        Ray ray;
        Rng rng;
        uint32_t slot = 0, depth = 0;
        ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1); rng.key = 0; rng.ctr = 0;
        // the frame and batch records are read from the kernarg segment where they are used (see kernarg_read): the
        // parameter list begins like the intersect kernel's
        const Cfg g_now = kernarg_read<Cfg>(offsetof(IntersectArgs, g));
        const Batch bt_now = kernarg_read<Batch>(offsetof(IntersectArgs, bt));
        bool live = i < n_in;
        if (FIRST) {
            const Cam cam_now = kernarg_read<Cam>(offsetof(IntersectArgs, cam));
            live = live && fetch_ray<true>(cam_now, g_now, bt_now, q, qin, dbg, i, ray, rng, slot, depth, true);
        } else live = live && fetch_ray<false>(cam, g_now, bt_now, q, qin, dbg, i, ray, rng, slot, depth, false);
        bool alive = false;        // continues into the next pass
        Ray next = ray;
        uint32_t next_ref = 0u;
        if (live) {
            const float4 hr = q.hit[i];
            if (sc.has_fog) rng.ctr = q.hit_ctr[i];      // fog volumes drew random numbers during the traversal
            const uint32_t carried = (!FIRST && depth != 0u) ? q.c[qin][i] : 0u;
            Closest c;
            c.dist = hr.x; c.best = __float_as_int(hr.y); c.u = hr.z; c.v = hr.w; c.hit = c.best != -1;
            alive = shade_ray<FIRST>(sc, g_now, bt_now, q, dbg, ray, rng, slot, depth, c, carried, next, next_ref);
        }
        // compact: survivors of this wave (ballot + prefix count) take consecutive places in the block's staging tile
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            const int lane = (int)(threadIdx.x & 63u);
            const int leader = __ffsll((long long)mask) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&s_count, (uint32_t)__popcll(mask));
            base = (uint32_t)__shfl((int)base, leader);
            if (alive) {
                const uint32_t at = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                s_a[at] = make_float4(next.o.x, next.o.y, next.o.z, next.d.x);
                s_b[at] = make_float4(next.d.y, next.d.z, __uint_as_float(slot), __uint_as_float(rng.ctr | ((depth + 1u) << 24)));
                s_c[at] = next_ref;
            }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) s_base = s_count ? atomicAdd(&q.counts[pass + 1], s_count) : 0u;
      __syncthreads();
      const uint32_t n_out = s_count, out0 = s_base;
      for (uint32_t j = threadIdx.x; j < n_out; j += 256) {
          q.a[qout][out0 + j] = s_a[j];
          q.b[qout][out0 + j] = s_b[j];
          q.c[qout][out0 + j] = s_c[j];
      }
      __syncthreads();
    }
}

// ---- resolve: World.cpp:94-100 + FrameManager.cpp:89-94.  (char)(c * 255): truncation, then the low byte ---------------
// The samples of a pixel are summed one after another in sample order, as the reference does (float addition does not
// associate), so a pixel is one thread's work; but a thread reading its pixel's spp consecutive records would make every
// wave access 64 distant addresses per load.  A block therefore copies the records of RPB pixels (RCH samples at a time)
// into LDS with consecutive lanes on consecutive records, and the pixel threads add them up from there.
constexpr int RTILE = 2048;   // float4 records staged per block and step (32 KiB)
__device__ __forceinline__ uint32_t to_u8(float c) { return (uint32_t)(int)(c * 255) & 0xFFu; }
__global__ __launch_bounds__(256) void resolve_kernel(Cfg g, Batch bt, const float4* radiance, uint8_t* out, int packed, int rpb, int rch, FastDiv div_rch) {
    __shared__ float4 tile[RTILE + 256];
    if (cancelled(bt)) return;        // the pixels of a cancelled batch keep what the frame buffer held
    const int spp = g.S * g.S;
    const int p0 = blockIdx.x * rpb;                                   // first pixel of the block
    const int n_p = min(rpb, bt.n_pixels - p0);
    V3 color = mk(0, 0, 0);
    for (int s0 = 0; s0 < spp; s0 += rch) {
        const int n_s = min(rch, spp - s0);
        for (int k = threadIdx.x; k < n_p * rch; k += 256) {
            const int p = (int)fdiv((uint32_t)k, div_rch), j = k - p * rch;
            if (j < n_s) tile[p * (rch + 1) + j] = radiance[(size_t)(p0 + p) * spp + s0 + j];
        }
        __syncthreads();
        if ((int)threadIdx.x < n_p)
            for (int j = 0; j < n_s; j++) { const float4 v = tile[threadIdx.x * (rch + 1) + j]; color = color + mk(v.x, v.y, v.z); }
        __syncthreads();
    }
    if ((int)threadIdx.x >= n_p) return;
    const int lp = p0 + threadIdx.x;
    int px, py;
    const int glp = bt.first_pixel + lp;
    if (!pixel_of(bt, glp, px, py)) return;
    color = divs(color, (float)spp);
    const float steepness = 5.0f;
    color = mk(1.0f - 1.0f / (1.0f + steepness * color.x), 1.0f - 1.0f / (1.0f + steepness * color.y), 1.0f - 1.0f / (1.0f + steepness * color.z));
    const uint32_t word = 255u | (to_u8(color.z) << 8) | (to_u8(color.y) << 16) | (to_u8(color.x) << 24);
    // packed: tile slot-major layout for the gather; otherwise the final frame
    const size_t at = packed ? (size_t)glp : (size_t)py * g.W + px;
    reinterpret_cast<uint32_t*>(out)[at] = word;
}
// scatter gathered per-rank tile buffers into the frame
__global__ __launch_bounds__(256) void assemble_kernel(Cfg g, Batch bt, const uint32_t* gathered, long long per_rank_pixels, uint32_t* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_rank_pixels * bt.n_ranks) return;
    Batch b = bt;
    b.rank = (int)(i / per_rank_pixels);
    int px, py;
    if (!pixel_of(b, (int)(i - (long long)b.rank * per_rank_pixels), px, py)) return;
    out[(size_t)py * g.W + px] = gathered[i];
}

// closest-hit queries for arbitrary rays (same walker, scene read from global memory; like the render kernels, compiled
// with and without fog support, and culling by distance without it)
template <bool FOG>
__global__ __launch_bounds__(256) void trace_kernel(Scene sc, Scene sc_ref, int n, const float* o3, const float* d3, int32_t* hit_id, float* rec9) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Store st;
    st.ent = sc.ent; st.tris = sc.tris; st.inst = sc.inst; st.planes = sc.planes;
    st.lds0 = nullptr; st.lds0_addr = 0; st.top_addr = 0; st.stack_addr = 0; st.root_link = 0; st.inst_end = nullptr;
    st.pair = nullptr; st.inst_start = nullptr;
    Walker w;
    walker_idle(sc, w);
    Ray r;
    r.o = mk(0, 0, 0); r.d = mk(0, 0, 1);
    Rng rng; rng.key = (uint32_t)i; rng.ctr = 0;   // fog volumes draw random numbers: one stream per ray
    bool elsewhere = false;   // a regrouped layout walks rays of the reciprocal path only (see exceptions_kernel)
    if (i < n) {
        r.o = mk(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]);
        r.d = mk(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        walker_start<!FOG>(sc, st, w, r, rng);
        if (!FOG && sc.has_ref && !w.usable && !ray_monotone(r)) { elsewhere = true; walker_idle(sc, w); }
    }
    elsewhere = walk_wave_to_end<FOG>(sc, st, w) || elsewhere;
    Closest c = w.c;
    if (!FOG && sc.has_ref && __any(elsewhere)) {   // those rays again, through the reference's layout
        Store sr = st;
        sr.ent = sc_ref.ent; sr.inst = sc_ref.inst;
        Walker w2;
        walker_idle(sc_ref, w2);
        if (elsewhere) walker_start<true>(sc_ref, sr, w2, r, rng);
        walk_wave_to_end<false>(sc_ref, sr, w2);
        if (elsewhere) { c = w2.c; if (c.best >= 0) c.best = (int)leaf_extra(sc_ref, c.best); }
    }
    if (i >= n) return;
    float* q = rec9 + 9 * (size_t)i;
    q[0] = c.dist;
    if (!c.hit) { hit_id[i] = -1; for (int k = 1; k < 9; k++) q[k] = 0; return; }
    Hit h;
    finalize_hit(sc, r, c, h);
    if (h.sphere) sphere_uv(h);   // the query reports HitInfo.uv whether or not a texture would read it
    hit_id[i] = h.object;
    q[1] = h.point.x; q[2] = h.point.y; q[3] = h.point.z; q[4] = h.normal.x; q[5] = h.normal.y; q[6] = h.normal.z; q[7] = h.u; q[8] = h.v;
}
__global__ void load_rays_kernel(int n, const float* o3, const float* d3, float4* a, float4* b, uint32_t* count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *count = (uint32_t)n;
    if (i >= n) return;
    a[i] = make_float4(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2], d3[3 * i]);
    b[i] = make_float4(d3[3 * i + 1], d3[3 * i + 2], __uint_as_float((uint32_t)i), __uint_as_float(0u));
}
__global__ void math_kernel(int n, const float* x, const float* y, float* s, float* a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = pm_sinf(x[i]);
    a[i] = pm_atan2f(y[i], x[i]);
}
__global__ void rng_kernel(int n, const uint32_t* key, const uint32_t* ctr, uint32_t* r31, float* ud) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    r31[i] = rand31(key[i], ctr[i]);
    Rng g; g.key = key[i]; g.ctr = ctr[i];
    ud[i] = unit_distance(g);
}
// the reciprocal-based division against the hardware's IEEE division: explicit pairs, then `rounds` pseudo-random pairs
// per thread biased towards quotients next to rounding boundaries; counts disagreements (bit patterns, NaN == NaN)
__global__ void div_check_kernel(int n, const float* a, const float* b, uint32_t seed, int rounds, unsigned long long* mismatches, float* first_bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    auto check = [&](float x, float y) {
        if (!mag_in(y, 9.5367431640625e-07f, 1048576.0f)) return;
        if (!(x == 0.0f || mag_in(x, 1.0e-24f, 2.0e12f))) return;
        const float k = 1.0f / y;
        if (recip_in_range(y) != k) {   // the short reciprocal against the full division
            if (bad == 0 && atomicAdd(mismatches + 1, 1ull) == 0) { first_bad[0] = 1.0f; first_bad[1] = y; first_bad[2] = recip_in_range(y); first_bad[3] = k; }
            bad++;
        }
        const float q1 = div_by(x, y, k), q2 = x / y;
        // a zero quotient may come out with the other sign (a = -0); no comparison of the slab test can tell
        if (!(q1 == q2) && !(q1 != q1 && q2 != q2)) {
            if (bad == 0 && atomicAdd(mismatches + 1, 1ull) == 0) { first_bad[0] = x; first_bad[1] = y; first_bad[2] = q1; first_bad[3] = q2; }
            bad++;
        }
    };
    if (i < n) check(a[i], b[i]);
    uint32_t key = fmix32(seed ^ (uint32_t)i * 0x9E3779B9u);
    for (int r = 0; r < rounds; r++) {
        const uint32_t r1 = fmix32(key + 3u * (uint32_t)r), r2 = fmix32(key ^ (0x85EBCA6Bu * (uint32_t)(r + 1)));
        // y: random mantissa, exponent in [-19, 1]; q: random mantissa; x = RN(y * q) nudged by -1..1 ulp: quotients that
        // sit on or next to representable values and midpoints
        const float y = __uint_as_float(((108u + (r1 >> 27) % 21u) << 23) | (r1 & 0x7FFFFFu)) * ((r2 & 1u) ? -1.0f : 1.0f);
        const float qq = __uint_as_float(((100u + (r2 >> 26) % 56u) << 23) | ((r2 >> 1) & 0x7FFFFFu));
        const float x0 = y * qq;
        check(x0, y);
        check(__uint_as_float(__float_as_uint(x0) + 1u), y);
        check(__uint_as_float(__float_as_uint(x0) - 1u), y);
        check(qq, y);
    }
    if (bad) atomicAdd(mismatches, bad);
}

int fail(int code, const std::string& msg) { rt::set_error(msg); return code; }
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(RT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// A device allocation that goes back when the scope ends, whichever way it ends.
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
        return RT_OK;
    }
    int from_host(const void* src, size_t bytes) {
        if (int rc = alloc(bytes)) return rc;
        if (bytes) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return RT_OK;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

template <typename T>
int upload(T** dst, const void* src, size_t bytes) {
    *dst = nullptr;
    HIP_TRY(hipMalloc((void**)dst, bytes ? bytes : 16));
    if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

}  // namespace

struct rt_engine {
    rt_config cfg;
    hipStream_t stream = nullptr;
    hipStream_t last_stream = nullptr;  // the stream of the latest render / assemble when the caller brought its own
    unsigned char* d_scratch = nullptr; // this device's copy of the static staging array (rt_engine_trace_static)
    unsigned char* h_stage = nullptr;   // page-locked host mirror of it, allocated with the engine: the copies of rt_engine_trace_static
                                        // then never go through a staging buffer the runtime would have to find at call time
    // cancellation and presentation
    hipStream_t side_stream = nullptr;  // carries the cancel word and frame snapshots past a render that is still running
    uint32_t* d_cancel = nullptr;       // serial of the cancelled render (kernels compare it with their own)
    uint32_t* h_serial = nullptr;       // pinned: source of the copy into d_cancel
    uint32_t serial = 0;                // serial of the latest render / ray query
    uint8_t* h_frame = nullptr;         // pinned host copy of the frame (rt_engine_map_frame), allocated on first use
    bool frame_dirty = true;            // the frame buffer differs from h_frame (true from the start: h_frame is allocated uninitialised,
                                        // and the first complete map must copy the zeroed frame buffer, FrameManager.cpp:13)
    Scene sc{};
    Scene sc_ref{};                     // the reference's layout of the same scene when `sc` is the regrouped one (sc.has_ref)
    OrdScene os{};
    bool have_scene = false, have_camera = false;
    Cam cam{};
    std::vector<void*> scene_allocs;
    size_t staged_bytes = 0;            // LDS the intersect kernel needs to hold the scene; 0 = read it from global memory
    size_t ord_bytes = 0;               // LDS of the ordered traversal (stacks + table of the upper levels); 0 = the scene is not walked that way
    Queues q{};
    int n_counts = 0;
    uint8_t* d_frame = nullptr;
    Debug dbg{nullptr, nullptr, nullptr, nullptr, nullptr};
    bool debug_valid = false;
    int n_cus = 0;
    Tune tune{32, 16, 8, 256, 16, 64, 16, 1};
    int quorum_mem = 48;                // Tune::quorum for scenes read from memory (RT_QUORUM sets both): in effect half of the lanes that hold a
                                        // ray (walk_round); the ordered traversal: 24 -> 55.7 ms, 32 -> 55.3, 48 -> 54.7, 56 -> 54.8 on the 960x540 cut
    int wide_blocks = -1;               // 768-thread bounce passes: -1 by batch size, 0 never, 1 always (RT_WIDE_BLOCKS)
    int refill_at_wide = 48;            // Tune::refill_at of the 768-thread camera-ray pass: with six waves per SIMD a wave can wait for
                                        // more idle lanes before it generates rays (22.59 -> 22.44 ms on the 1080p, 64 spp frame; RT_REFILL_AT sets both)
    int wide_first = 1;                 // ... and the camera rays' pass too (RT_WIDE_FIRST)
    int wide_max_pass = 3;              // ... up to this bounce (later passes hold a few per cent of the paths: 17.45 -> 17.35 ms)
    int win[4] = {0, 0, 0, 0};          // x0, y0, width, height of the render window
    // statistics of the last render
    std::vector<uint32_t> h_counts;     // per batch: counts (B + 2), cursors (B + 2)
    uint32_t* d_counts_all = nullptr;
    size_t counts_capacity = 0;         // words
    int last_batches = 0;
    uint64_t last_paths = 0;
    bool time_kernels = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
};

namespace {

Cfg make_cfg(const rt_config& c) {
    Cfg g{};
    g.W = c.width; g.H = c.height; g.S = c.samples_per_axis; g.B = c.bounces;
    g.fov = c.fov; g.sky = c.skybox_brightness; g.dof = c.dof_intensity; g.focus = c.focus_distance;
    g.div_S = make_fastdiv((uint32_t)c.samples_per_axis);
    g.div_spp = make_fastdiv((uint32_t)(c.samples_per_axis * c.samples_per_axis));
    // World.cpp:70: onePixelOffset = (-(1.0f / W) * ((float)W / H) * FOV, -(1.0f / H) * FOV); this file is compiled without
    // FMA contraction, so the host evaluates them as the reference's build does
    const float fW = (float)g.W, fH = (float)g.H;
    g.aspect = fW / fH;
    g.one_x = -(1.0f / fW) * g.aspect * g.fov;
    g.one_y = -(1.0f / fH) * g.fov;
    g.rW = 1.0f / fW; g.rH = 1.0f / fH; g.rS = 1.0f / (float)g.S; g.rfocus = 1.0f / g.focus;
    const float af = std::fabs(g.focus), ad = std::fabs(g.dof);
    // div_by is exact for divisors in [2^-20, 2^20] and dividends that are 0 or in [2^-80, 2^40]: pixel and sample
    // indices are; a lens offset is a sum of two products of dof * (a multiple of 2^-30 below 1) with camera axes
    g.fast_div = (g.W <= (1 << 20) && g.H <= (1 << 20) && g.S <= (1 << 20) && af >= 9.5367431640625e-07f && af <= 1048576.0f &&
                  (ad == 0.0f || (ad >= 1.0e-5f && ad <= 1.0e6f))) ? 1 : 0;
    return g;
}

void free_scene(rt_engine* e) {
    for (void* p : e->scene_allocs) (void)hipFree(p);
    e->scene_allocs.clear();
    e->have_scene = false;
}

int tiles_x(const rt_engine* e) { return (e->win[2] + e->cfg.tile_w - 1) / e->cfg.tile_w; }
int tiles_y(const rt_engine* e) { return (e->win[3] + e->cfg.tile_h - 1) / e->cfg.tile_h; }
int tiles_per_rank(const rt_engine* e, int n_ranks) { return (tiles_x(e) * tiles_y(e) + n_ranks - 1) / n_ranks; }

Batch make_batch(const rt_engine* e, int rank, int n_ranks, uint32_t seed) {
    const rt_config& c = e->cfg;
    Batch b;
    b.first_pixel = 0; b.n_pixels = 0; b.rank = rank; b.n_ranks = n_ranks;
    b.tiles_x = tiles_x(e); b.n_tiles = tiles_x(e) * tiles_y(e); b.tile_w = c.tile_w; b.tile_h = c.tile_h; b.seed = seed;
    b.wx0 = e->win[0]; b.wy0 = e->win[1]; b.ww = e->win[2]; b.wh = e->win[3];
    b.div_per_tile = make_fastdiv((uint32_t)(c.tile_w * c.tile_h)); b.div_tile_w = make_fastdiv((uint32_t)c.tile_w);
    b.div_tiles_x = make_fastdiv((uint32_t)b.tiles_x);
    b.cancel = e->d_cancel; b.serial = e->serial;
    return b;
}

hipEvent_t next_event(rt_engine* e) {
    if (e->ev_used == e->ev.size()) {
        hipEvent_t ev;
        (void)hipEventCreate(&ev);
        e->ev.push_back(ev);
    }
    return e->ev[e->ev_used++];
}

template <bool FIRST>
void launch_intersect(rt_engine* e, hipStream_t st, const Cfg& g, const Batch& bt, const Queues& q, int pass) {
    const int blocks = e->n_cus * 2;
    // the kernel is compiled with and without fog support: a scene without fog pays nothing for it
    const bool fog = e->sc.has_fog != 0;
    // a scene read from memory: every iteration of the traversal loop ends in a round trip to the cache, so the loop
    // is left to the primitive tests only when fewer lanes can go on (measured on the 100 000-primitive scene: 8 -> 24
    // lanes, 3.5 % of the frame; 24 -> 48 with the ordered traversal, 2 % more)
    Tune tune = e->tune;
    if (!e->staged_bytes) tune.quorum = e->quorum_mem;
#define RT_LAUNCH_INTERSECT(STAGED, FOG, LDS_BYTES) \
    hipLaunchKernelGGL((intersect_kernel<FIRST, STAGED, FOG>), dim3(blocks), dim3(IBLOCK), LDS_BYTES, st, e->sc, e->cam, g, bt, q, pass, e->dbg, tune, e->os)
    // Large passes over a scene in LDS: 768-thread blocks, twelve waves that spread evenly over the four SIMDs, two blocks
    // per CU as before: six waves per SIMD instead of four, for which the kernel stays within 80 registers (RELOAD, LEAN,
    // SAVE in rt_device.h; the variant keeps one dword per thread in LDS beside the scene).  Measured on the 1920 x 1080
    // x 64 spp frame: 18.0 -> 16.0 ms in the intersect kernels; on a half, a quarter and an eighth of it 14.4 -> 13.2,
    // 7.96 -> 7.84 and 4.89 -> 4.84 ms per frame; on 1280 x 720 x 9 spp 2.37 -> 2.45 ms: small passes lose with larger
    // blocks (more waves wait at the end), so the shape follows the size of the batch and the depth of the pass.
    const bool wide = e->wide_blocks < 0 ? (long long)bt.n_pixels * (g.S * g.S) >= WIDE_MIN_PATHS : e->wide_blocks != 0;
    if (e->staged_bytes && e->staged_bytes + 768 * sizeof(uint32_t) <= (size_t)MAX_STAGED_BYTES && !fog && wide && pass <= e->wide_max_pass && (!FIRST || e->wide_first)) {
        if (FIRST) tune.refill_at = e->refill_at_wide;
        hipLaunchKernelGGL((intersect_kernel<FIRST, true, false, 768, 6>), dim3(blocks), dim3(768), e->staged_bytes, st, e->sc, e->cam, g, bt, q, pass, e->dbg, tune, e->os);
        return;
    }
    if (e->ord_bytes && !e->staged_bytes && !fog) {
        // a scene read from memory: near child first, the far one on a stack in LDS, the tree's upper levels in LDS too
        // (rt_device.h, "ordered traversal"); one 1024-thread block per CU owns the CU's LDS
        if (e->os.pairs) hipLaunchKernelGGL((intersect_kernel<FIRST, false, false, OBLOCK, 1, true, true>), dim3(e->n_cus), dim3(OBLOCK), e->ord_bytes, st, e->sc, e->cam, g, bt, q, pass, e->dbg, tune, e->os);
        else hipLaunchKernelGGL((intersect_kernel<FIRST, false, false, OBLOCK, 1, true>), dim3(e->n_cus), dim3(OBLOCK), e->ord_bytes, st, e->sc, e->cam, g, bt, q, pass, e->dbg, tune, e->os);
        return;
    }
    if (e->staged_bytes) { if (fog) RT_LAUNCH_INTERSECT(true, true, e->staged_bytes); else RT_LAUNCH_INTERSECT(true, false, e->staged_bytes); }
    else { if (fog) RT_LAUNCH_INTERSECT(false, true, 0); else RT_LAUNCH_INTERSECT(false, false, 0); }
#undef RT_LAUNCH_INTERSECT
}

// the passes of one batch: pass 0 generates its rays, passes 1..B read the queue the previous pass filled.  counts points
// at COUNT_WORDS zeroed words: ray counts, then the intersect passes' reservation counters.
int run_passes(rt_engine* e, const Batch& bt, hipStream_t st, uint32_t* counts, bool first_from_queue, int bounces) {
    Queues q = e->q;
    q.counts = counts;
    q.cursors = counts + CUR_STRIDE;
    Cfg g = make_cfg(e->cfg);
    g.B = bounces;
    const int shade_blocks = e->n_cus * 8;
    for (int pass = 0; pass <= bounces; pass++) {
        const bool first = pass == 0 && !first_from_queue;
        if (e->time_kernels) (void)hipEventRecord(next_event(e), st);
        if (first) launch_intersect<true>(e, st, g, bt, q, pass);
        else launch_intersect<false>(e, st, g, bt, q, pass);
        if (e->time_kernels) (void)hipEventRecord(next_event(e), st);
        if (e->sc.has_ref) {   // the rays the pass handed over (none in a rendered frame: the launch leaves at once)
            if (first) hipLaunchKernelGGL(exceptions_kernel<true>, dim3(64), dim3(256), 0, st, e->sc_ref, e->cam, g, bt, q, pass, e->dbg);
            else hipLaunchKernelGGL(exceptions_kernel<false>, dim3(64), dim3(256), 0, st, e->sc_ref, e->cam, g, bt, q, pass, e->dbg);
        }
        if (first) {
            const uint32_t n = (uint32_t)bt.n_pixels * (uint32_t)(g.S * g.S);
            const int blocks = (int)std::min<uint32_t>((n + 255u) / 256u, (uint32_t)shade_blocks);
            hipLaunchKernelGGL(shade_kernel<true>, dim3(blocks), dim3(256), 0, st, e->sc, e->cam, g, bt, q, pass, e->dbg);
        } else {
            hipLaunchKernelGGL(shade_kernel<false>, dim3(shade_blocks), dim3(256), 0, st, e->sc, e->cam, g, bt, q, pass, e->dbg);
        }
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int init_engine(rt_engine* e, const rt_config* cfg);

// Wait for everything the engine has queued: its own stream and, when the latest render or assemble ran on a stream the
// caller brought, that stream.
int wait_all(rt_engine* e) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->last_stream) {
        // the caller's handle is used this once and then forgotten: everything queued on it for this engine has finished,
        // and the caller may destroy the stream from here on
        hipStream_t st = e->last_stream;
        e->last_stream = nullptr;
        HIP_TRY(hipStreamSynchronize(st));
    }
    return RT_OK;
}

}  // namespace

extern "C" {

#ifndef RT_SOURCE_HASH
#define RT_SOURCE_HASH "unknown"
#endif
const char* rt_source_hash(void) { return RT_SOURCE_HASH; }

int rt_engine_create(const rt_config* cfg, rt_engine** out) {
    if (!cfg || !out) return fail(RT_E_INVALID, "null argument");
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->samples_per_axis <= 0 || cfg->bounces < 0 || cfg->bounces > 32)
        return fail(RT_E_INVALID, "width, height and samples_per_axis must be positive, bounces within 0..32");
    if (!(cfg->focus_distance != 0.0f)) return fail(RT_E_INVALID, "focus_distance must not be zero");
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (cfg->device < 0 || cfg->device >= n_dev) return fail(RT_E_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(cfg->device));
    rt_engine* e = new (std::nothrow) rt_engine();
    if (!e) return fail(RT_E_NOMEM, "out of host memory");
    const int rc = init_engine(e, cfg);
    if (rc != RT_OK) { rt_engine_destroy(e); return rc; }    // whatever was allocated before the failure goes back
    *out = e;
    return RT_OK;
}

}  // extern "C"

namespace {
int init_engine(rt_engine* e, const rt_config* cfg) {
    e->cfg = *cfg;
    if (e->cfg.tile_w <= 0) e->cfg.tile_w = 64;
    if (e->cfg.tile_h <= 0) e->cfg.tile_h = 64;
    const int spp = cfg->samples_per_axis * cfg->samples_per_axis;
    // default: up to 2^27 paths per batch (about 30 GB of queues at 8 bounces) — a 1080p, 64 spp frame is one batch, and
    // every pass is then a single pair of launches with one short tail
    long long want = e->cfg.max_batch_paths > 0 ? e->cfg.max_batch_paths : (1ll << 27);
    const long long min_paths = (long long)spp * 64;
    if (want < min_paths) want = min_paths;
    e->win[0] = 0; e->win[1] = 0; e->win[2] = cfg->width; e->win[3] = cfg->height;
    const long long frame_paths = (long long)tiles_x(e) * tiles_y(e) * e->cfg.tile_w * e->cfg.tile_h * spp;
    if (want > frame_paths) want = frame_paths;
    const long long batch_pixels = want / spp;   // a batch holds whole pixels
    e->q.capacity = (int)(batch_pixels * spp);
    e->cfg.max_batch_paths = e->q.capacity;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    e->n_cus = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreate(&e->stream));
    HIP_TRY(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void**)&e->d_cancel, 256));
    HIP_TRY(hipMemset(e->d_cancel, 0, 256));
    HIP_TRY(hipHostMalloc((void**)&e->h_serial, 256, hipHostMallocDefault));
    // the engine's device is current (rt_engine_create): the address of this GPU's copy of the staging array
    HIP_TRY(hipGetSymbolAddress((void**)&e->d_scratch, HIP_SYMBOL(scratch)));
    HIP_TRY(hipHostMalloc((void**)&e->h_stage, sizeof(scratch), hipHostMallocDefault));
    const size_t cap = (size_t)e->q.capacity;
    for (int k = 0; k < 2; k++) {
        // one allocation, b behind a: the camera rays' pass also uses the input queue of the pass after it, empty until
        // then, as `cap` records of 32 bytes (walker_special, SAVE)
        HIP_TRY(hipMalloc((void**)&e->q.a[k], 2 * cap * sizeof(float4)));
        e->q.b[k] = e->q.a[k] + cap;
    }
    HIP_TRY(hipMalloc((void**)&e->q.hit, cap * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&e->q.hit_ctr, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&e->q.exc, cap * sizeof(uint32_t)));      // (every ray of a pass may be handed over: a caller's own rays)
    for (int k = 0; k < 2; k++) HIP_TRY(hipMalloc((void**)&e->q.c[k], cap * sizeof(uint32_t)));
    e->q.stack_depth = std::max(1, cfg->bounces);
    HIP_TRY(hipMalloc((void**)&e->q.stack, cap * sizeof(uint32_t) * (size_t)e->q.stack_depth));
    HIP_TRY(hipMalloc((void**)&e->q.radiance, cap * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&e->d_frame, (size_t)cfg->width * cfg->height * 4));
    HIP_TRY(hipMemset(e->d_frame, 0, (size_t)cfg->width * cfg->height * 4));
    e->n_counts = cfg->bounces + 2;
    // pass counters for the largest number of batches a render can need (one rank rendering the whole frame), so that
    // rendering never allocates
    // (COUNT_WORDS words, 135 KB, per batch: a small max_batch_paths on a large frame would ask for gigabytes of counters,
    // and for more batches than an int counts words of)
    const long long max_batches = (frame_paths / spp + batch_pixels - 1) / batch_pixels;
    if (max_batches > MAX_BATCHES)
        return fail(RT_E_INVALID, "max_batch_paths is too small for this frame: more than " + std::to_string(MAX_BATCHES) + " batches");
    e->counts_capacity = (size_t)max_batches * COUNT_WORDS;
    HIP_TRY(hipMalloc((void**)&e->d_counts_all, sizeof(uint32_t) * e->counts_capacity));
    // the staged variants of the intersect kernel may use up to a whole CU's LDS; the limit is a property of the kernel,
    // not of an engine or a scene, so every engine sets the same value
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_STAGED_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_STAGED_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<false, true, false, 768, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_STAGED_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<true, true, false, 768, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_STAGED_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<true, false, false, OBLOCK, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_ORD_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<false, false, false, OBLOCK, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_ORD_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<true, false, false, OBLOCK, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_ORD_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<false, false, false, OBLOCK, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_ORD_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_STAGED_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_STAGED_BYTES));
#ifdef RT_PROFILE
    HIP_TRY(hipMalloc((void**)&e->dbg.prof, sizeof(unsigned long long) * 32 * 16));
    HIP_TRY(hipMemset(e->dbg.prof, 0, sizeof(unsigned long long) * 32 * 16));
#endif
    if (const char* v = getenv("RT_REFILL_AT")) e->tune.refill_at = e->refill_at_wide = std::max(1, std::min(64, atoi(v)));
    if (const char* v = getenv("RT_REFILL_AT_BOUNCE")) e->tune.refill_at_bounce = std::max(1, std::min(64, atoi(v)));
    if (const char* v = getenv("RT_QUORUM")) e->tune.quorum = e->quorum_mem = std::max(1, std::min(65, atoi(v)));
    if (const char* v = getenv("RT_CHUNK")) e->tune.chunk = std::max(64, atoi(v) & ~63);
    if (const char* v = getenv("RT_MIN_CHUNK")) e->tune.min_chunk = std::max(8, atoi(v) & ~7);
    if (const char* v = getenv("RT_GUIDE")) e->tune.guide = std::max(1, std::min(64, atoi(v)));
    if (const char* v = getenv("RT_FINISH_QUORUM")) e->tune.finish_quorum = std::max(1, std::min(65, atoi(v)));
    if (const char* v = getenv("RT_WIDE_BLOCKS")) e->wide_blocks = atoi(v) ? 1 : 0;
    if (const char* v = getenv("RT_WIDE_MAX_PASS")) e->wide_max_pass = atoi(v);
    if (const char* v = getenv("RT_WIDE_FIRST")) e->wide_first = atoi(v);
    if (const char* v = getenv("RT_PRIM_QUORUM")) e->tune.prim_quorum = std::max(1, std::min(65, atoi(v)));
    return RT_OK;
}
}  // namespace

extern "C" {

void rt_engine_destroy(rt_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    (void)wait_all(e);
    free_scene(e);
    for (int k = 0; k < 2; k++) (void)hipFree(e->q.a[k]);   // (b[k] is its second half)
    (void)hipFree(e->q.hit); (void)hipFree(e->q.hit_ctr); (void)hipFree(e->q.exc);
    (void)hipFree(e->q.c[0]); (void)hipFree(e->q.c[1]); (void)hipFree(e->q.stack); (void)hipFree(e->q.radiance); (void)hipFree(e->d_frame); (void)hipFree(e->d_counts_all);
    (void)hipFree(e->dbg.primary_id); (void)hipFree(e->dbg.primary_t); (void)hipFree(e->dbg.rng_draws);
    for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
    if (e->side_stream) { (void)hipStreamSynchronize(e->side_stream); (void)hipStreamDestroy(e->side_stream); }
    (void)hipFree(e->d_cancel);
    if (e->h_serial) (void)hipHostFree(e->h_serial);
    if (e->h_frame) (void)hipHostFree(e->h_frame);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int rt_engine_get_config(const rt_engine* e, rt_config* out) {
    if (!e || !out) return fail(RT_E_INVALID, "null argument");
    *out = e->cfg;
    return RT_OK;
}

// Device entries of one layout (rt_flat.h): 8 dwords each.  A leaf carries one more number in a dword its kind leaves
// free (sphere, cube: dword 7; triangle: dword 0; a fog volume has none, and scenes with fog need none): its rank in the
// reference's visiting order (what settles equal distances), or, in the reference's layout when the engine renders from
// the regrouped one, the leaf's entry there (`leaf_peer`: what the walk of an exceptional ray reports).
static std::vector<rt::F4> device_entries(const rt_layout& L, const std::vector<int32_t>* leaf_peer, bool& has_fog) {
    std::vector<rt::F4> inter(L.ent_lo.size() * 2);
    for (size_t i = 0; i < L.ent_lo.size(); i++) {
        const rt::F4 lo = L.ent_lo[i], hi = L.ent_hi[i];
        const int kind = rt::bits_i(hi.w);
        auto tag = [](int k, int payload) { return rt::bits_f((int32_t)(((uint32_t)k << rt::SK_SHIFT) | (uint32_t)payload)); };
        const float extra = rt::bits_f(leaf_peer ? (*leaf_peer)[i] : (int32_t)L.ent_rank[i]);
        rt::F4 a{0, 0, 0, 0}, b{0, 0, tag(kind, 0), 0};
        switch (kind) {
        case rt::ENT_INNER:
            a = {lo.x, hi.x, lo.y, hi.y};
            if (lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z) b = {lo.z, hi.z, lo.w, L.ent_entry_box[i] ? rt::bits_f((int32_t)rt::ENTRY_BOX_MARK) : L.ent_rho[i]};
            else b = {lo.z, hi.z, tag(rt::ENT_INNER_INV, 0), lo.w};
            break;
        case rt::ENT_SPHERE: a = lo; b.z = tag(kind, rt::bits_i(hi.x)); b.w = extra; break;
        case rt::ENT_CUBE: a = {lo.x, lo.y, lo.z, hi.x}; b = {hi.y, hi.z, tag(kind, rt::bits_i(lo.w)), extra}; break;
        case rt::ENT_FOG: a = {lo.x, lo.y, lo.z, hi.x}; b = {hi.y, hi.z, tag(kind, rt::bits_i(lo.w)), rt::bits_f(-1)}; has_fog = true; break;
        case rt::ENT_TRI: a.x = extra; b.z = tag(kind, rt::bits_i(lo.x)); break;
        case rt::ENT_INSTANCE: b.z = tag(kind, rt::bits_i(lo.x)); break;
        default: break;
        }
        inter[2 * i] = a; inter[2 * i + 1] = b;
    }
    // a triangle or fog entry names the instance it belongs to (every instance has its own range of entries)
    for (size_t k = 0; k + rt::RT_INST_F4 <= L.inst.size(); k += rt::RT_INST_F4)
        for (int i = rt::bits_i(L.inst[k + 2].z); i < rt::bits_i(L.inst[k + 2].w); i++) {
            const int kind = rt::bits_i(L.ent_hi[i].w);
            if (kind == rt::ENT_TRI || kind == rt::ENT_FOG) inter[2 * (size_t)i + 1].w = rt::bits_f((int)(k / rt::RT_INST_F4));
        }
    return inter;
}
// Device instance records of one layout: the host's, with the alpha mask of the mesh's image texture (so that entering
// an instance needs no look-up in the object table) and, in place of the end of the instance's range, the entry behind
// the instance's own entry in the top-level list, where a ray goes on when it leaves the instance.
static int device_instances(const rt_layout& L, const rt_scene_flat* f, const std::vector<int>& alpha_at, std::vector<rt::F4>& inst_rec) {
    inst_rec = L.inst;
    for (size_t k = 0; k + rt::RT_INST_F4 <= inst_rec.size(); k += rt::RT_INST_F4) {
        const rt::F4 ob = f->objects[(size_t)rt::bits_i(inst_rec[k + 3].x) * rt::RT_OBJ_F4];
        int at = -1;
        uint32_t wh = 0;
        if (rt::bits_i(ob.y) == RT_TEX_IMAGE) {
            const int img = rt::bits_i(ob.z);
            if (alpha_at[img] < 0) return fail(RT_E_INVALID, "image textures must be RGBA and at most 65535 texels wide and high");
            at = alpha_at[img];
            wh = (f->images[img].h << 16) | f->images[img].w;
        }
        inst_rec[k + 3].z = rt::bits_f(at);
        inst_rec[k + 3].w = rt::bits_f((int32_t)wh);
    }
    for (int i = 0; i < L.top_end; i++)
        if (rt::bits_i(L.ent_hi[i].w) == rt::ENT_INSTANCE) {
            const size_t k = (size_t)rt::bits_i(L.ent_lo[i].x) * rt::RT_INST_F4;
            if (k + rt::RT_INST_F4 <= inst_rec.size()) inst_rec[k + 2].w = rt::bits_f(i + 1);
        }
    return RT_OK;
}

static int rt_engine_set_scene_impl(rt_engine* e, const rt_scene* s) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    const rt_scene_flat* f = rt_scene_get_flat(s);
    if (!f) return fail(RT_E_INVALID, "scene is not built");
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = wait_all(e)) return rc;
    free_scene(e);
    // the layout the kernels walk: the regrouped one when the scene has it, else the reference's
    const bool use_fast = f->has_fast && !(e->cfg.flags & RT_FLAG_NO_REGROUP);
    // ... with wide nodes when the scene will be held on chip (the kernels that read it from memory walk pairs of children)
    auto staged_need = [&](const rt_layout& X) { return (2 * X.ent_lo.size() + f->tris.size() + X.inst.size() + f->planes.size()) * 16; };
    const bool use_wide = use_fast && f->has_wide && !(e->cfg.flags & (RT_FLAG_NO_LDS | RT_FLAG_NO_WIDE)) && staged_need(f->wide) <= (size_t)MAX_STAGED_BYTES;
    const rt_layout& L = use_wide ? f->wide : use_fast ? f->fast : static_cast<const rt_layout&>(*f);
    Scene sc{};
    float4 *ent, *tris, *inst, *objects, *planes;
    rt::ImageRec* images;
    uint8_t* texels;
    // the traversal addresses entries with 32-bit byte offsets (32 bytes each) and texels with 32-bit offsets
    if (L.ent_lo.size() >= (1u << 27) || f->objects.size() / rt::RT_OBJ_F4 > rt::SK_MASK) return fail(RT_E_INVALID, "scene too large: 2^27 entries or more, or more than 2^28 objects");
    // (a bounce records where its colour is in 29 bits: an object id, or a texel's byte offset / 4)
    if (f->texels.size() >= ((size_t)1 << 31)) return fail(RT_E_INVALID, "scene too large: 2 GiB of texels or more");
    bool has_fog = false;
    std::vector<rt::F4> inter = device_entries(L, nullptr, has_fog);
    if (int rc = upload(&ent, inter.data(), inter.size() * 16)) return rc;
    e->scene_allocs.push_back(ent);
    // device triangles carry the two edges Triangle::Intersect starts from (Object.cpp:503-504: the same float
    // subtractions, done once here); the corners v1 and v2, needed only once the ray is known to reach the triangle,
    // move to the spare lanes of the records
    std::vector<rt::F4> dtris(f->tris);
    for (size_t k = 0; k + rt::RT_TRI_F4 <= dtris.size(); k += rt::RT_TRI_F4) {
        rt::F4* T = &dtris[k];
        const rt::F4 v0 = T[0], v1 = T[1], v2 = T[2];
        T[0] = rt::F4{v0.x, v0.y, v0.z, v1.x};
        T[1] = rt::F4{v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.y};
        T[2] = rt::F4{v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v1.z};
        T[5] = rt::F4{T[5].x, v2.x, v2.y, v2.z};
    }
    if (int rc = upload(&tris, dtris.data(), dtris.size() * 16)) return rc;
    e->scene_allocs.push_back(tris);
    // one bit per texel of every image: "alpha byte > 0.01f" (DataUtility.cpp:123-127).  The padding row of the texel
    // pool (all zero) is covered too, because uv == 1 indexes one row past the image.
    std::vector<uint32_t> alpha_bits;
    std::vector<int> alpha_at(f->images.size(), -1);
    for (size_t k = 0; k < f->images.size(); k++) {
        const rt::ImageRec& im = f->images[k];
        if (im.channels != 4 || im.w > 0xFFFFu || im.h > 0xFFFFu) continue;
        alpha_at[k] = (int)alpha_bits.size();
        const size_t texels = (size_t)im.w * (im.h + 1) + 32;
        alpha_bits.resize(alpha_bits.size() + (texels + 31) / 32, 0u);
        for (size_t t = 0; t < (size_t)im.w * (im.h + 1); t++) {
            const size_t byte = (size_t)im.offset + 4 * t + 3;
            if (byte < f->texels.size() && f->texels[byte] >= 1) alpha_bits[(size_t)alpha_at[k] + (t >> 5)] |= 1u << (t & 31);
        }
    }
    std::vector<rt::F4> inst_rec;
    if (int rc = device_instances(L, f, alpha_at, inst_rec)) return rc;
    uint32_t* d_alpha;
    if (int rc = upload(&d_alpha, alpha_bits.data(), alpha_bits.size() * 4)) return rc;
    e->scene_allocs.push_back(d_alpha);
    sc.alpha_bits = d_alpha;
    if (int rc = upload(&inst, inst_rec.data(), inst_rec.size() * 16)) return rc;
    e->scene_allocs.push_back(inst);
    if (int rc = upload(&objects, f->objects.data(), f->objects.size() * 16)) return rc;
    e->scene_allocs.push_back(objects);
    if (int rc = upload(&planes, f->planes.data(), f->planes.size() * 16)) return rc;
    e->scene_allocs.push_back(planes);
    if (int rc = upload(&images, f->images.data(), f->images.size() * sizeof(rt::ImageRec))) return rc;
    e->scene_allocs.push_back(images);
    if (int rc = upload(&texels, f->texels.data(), f->texels.size())) return rc;
    e->scene_allocs.push_back(texels);
    sc.ent = ent; sc.tris = tris; sc.inst = inst; sc.objects = objects; sc.planes = planes;
    sc.images = images; sc.texels = texels;
    sc.sky_off = f->sky.offset; sc.sky_w = f->sky.w; sc.sky_h = f->sky.h;
    sc.top_end = L.top_end; sc.stale_entry = L.stale_entry; sc.n_planes = (int)f->planes.size();
    sc.n_entries = (int)L.ent_lo.size(); sc.n_tris = (int)f->tris.size() / rt::RT_TRI_F4; sc.n_inst = (int)L.inst.size() / rt::RT_INST_F4; sc.fast_div_ok = f->fast_div_ok ? 1 : 0;
    sc.idle_entry = L.idle_entry;
    sc.has_fog = has_fog ? 1 : 0;
    sc.ranked = has_fog ? 0 : 1;
    sc.has_ref = use_fast ? 1 : 0;
    sc.entry_guard = f->entry_guard;
    sc.cull_delta = f->cull_delta; sc.cull_cabs = f->cull_cabs; sc.cull_2rmax = f->cull_2rmax;
    sc.cull_on = (e->cfg.flags & RT_FLAG_NO_CULL) ? 0 : 1;
    // The reference's own layout beside a regrouped one: rays that the regrouped hierarchy is not proven for (a zero
    // direction component, in the scene's frame or in a mesh instance's) are walked through it by exceptions_kernel; its
    // leaves name their entries in the regrouped layout, which is what hit records hold.
    e->sc_ref = Scene{};
    if (use_fast) {
        bool fog_ref = false;
        std::vector<rt::F4> inter_ref = device_entries(*f, use_wide ? &f->ent_peer_wide : &f->ent_peer, fog_ref);
        std::vector<rt::F4> inst_ref;
        if (int rc = device_instances(*f, f, alpha_at, inst_ref)) return rc;
        float4 *d_ent_ref, *d_inst_ref;
        if (int rc = upload(&d_ent_ref, inter_ref.data(), inter_ref.size() * 16)) return rc;
        e->scene_allocs.push_back(d_ent_ref);
        if (int rc = upload(&d_inst_ref, inst_ref.data(), inst_ref.size() * 16)) return rc;
        e->scene_allocs.push_back(d_inst_ref);
        Scene r = sc;
        r.ent = d_ent_ref; r.inst = d_inst_ref;
        r.top_end = f->top_end; r.stale_entry = f->stale_entry; r.n_entries = (int)f->ent_lo.size(); r.n_inst = (int)f->inst.size() / rt::RT_INST_F4;
        r.idle_entry = f->idle_entry;
        r.ranked = 0; r.has_ref = 0;      // met in the reference's order: the first of two equal distances is the one to keep
        e->sc_ref = r;
    }
    // stage the whole scene in LDS when two 512-thread blocks per CU can each hold it (160 KiB per CU)
    const size_t need = (inter.size() + f->tris.size() + L.inst.size() + f->planes.size()) * 16;
    e->staged_bytes = (need <= (size_t)MAX_STAGED_BYTES && !(e->cfg.flags & RT_FLAG_NO_LDS)) ? need : 0;
    // Otherwise, without fog: the ordered traversal, if a ray's stack fits (layout of its entries and table: rt_flat.h)
    e->ord_bytes = 0;
    e->os = OrdScene{};
    const size_t stack_bytes = (size_t)L.stack_need * OBLOCK * 4;
    if (!e->staged_bytes && !has_fog && !(e->cfg.flags & RT_FLAG_NO_ORDER) && L.ent_lo.size() < rt::LINK_LDS && stack_bytes + 4096 <= (size_t)MAX_ORD_BYTES) {
        const size_t n = L.ent_lo.size();
        auto is_plain_inner = [&](size_t i) { return rt::bits_i(L.ent_hi[i].w) == rt::ENT_INNER && rt::bits_i(inter[2 * i + 1].z) >> rt::SK_SHIFT == rt::ENT_INNER; };
        // the upper levels of the top-level tree, level by level, as many whole levels as fit beside the stacks
        std::vector<int32_t> table, level;               // entry indices: in the table, in the level being tried
        if (is_plain_inner(0)) level.push_back(0);
        const size_t max_records = std::min<size_t>(((size_t)MAX_ORD_BYTES - stack_bytes) / (rt::TOP_REC_DWORDS * 4), (size_t)rt::LINK_INDEX);
        while (!level.empty() && table.size() + level.size() <= max_records) {
            table.insert(table.end(), level.begin(), level.end());
            std::vector<int32_t> next;
            for (int32_t i : level)
                for (int32_t c : {i + 1, L.ent_right[i]})
                    if (c >= 0 && is_plain_inner((size_t)c)) next.push_back(c);
            level.swap(next);
        }
        std::vector<int32_t> record_of(n, -1);
        for (size_t k = 0; k < table.size(); k++) record_of[table[k]] = (int32_t)k;
        auto link_of = [&](int32_t entry) { return entry < 0 ? rt::LINK_NONE : record_of[entry] >= 0 ? (rt::LINK_LDS | (uint32_t)record_of[entry]) : (uint32_t)entry; };
        // rho rounded up to a multiple of four units in the last place, the axis in the two bits that frees: never below
        // the bound's own value; +inf ("never cull") becomes a NaN, which no comparison finds smaller than anything
        auto rho_axis = [&](size_t i) {
            uint32_t b = (uint32_t)rt::bits_i(inter[2 * i + 1].w);
            if (b < 0x7F800000u) b = (b + 3u) & ~3u;
            return (b & ~3u) | (uint32_t)L.ent_axis[i];
        };
        std::vector<rt::F4> ord(inter);
        for (size_t i = 0; i < n; i++) {
            if (rt::bits_i(L.ent_hi[i].w) != rt::ENT_INNER) continue;
            const uint32_t second = link_of(L.ent_right[i]);
            if (is_plain_inner(i)) { ord[2 * i + 1].z = rt::bits_f((int32_t)second); ord[2 * i + 1].w = rt::bits_f((int32_t)rho_axis(i)); }
            else ord[2 * i + 1].w = rt::bits_f((int32_t)second);          // INNER_INV: kind in dword 6, the link in dword 7
        }
        std::vector<uint32_t> tab(table.size() * rt::TOP_REC_DWORDS, 0u);
        for (size_t k = 0; k < table.size(); k++) {
            const size_t i = (size_t)table[k];
            uint32_t* r = &tab[k * rt::TOP_REC_DWORDS];
            memcpy(r, &ord[2 * i], 32);
            r[8] = link_of((int32_t)i + 1);
        }
        std::vector<int32_t> inst_end(std::max<size_t>(1, L.inst.size() / rt::RT_INST_F4), 0);
        for (size_t k = 0; k + rt::RT_INST_F4 <= L.inst.size(); k += rt::RT_INST_F4) inst_end[k / rt::RT_INST_F4] = rt::bits_i(L.inst[k + 2].w);
        float4* d_ord; uint32_t* d_tab; int32_t* d_end;
        if (int rc = upload(&d_ord, ord.data(), ord.size() * 16)) return rc;
        e->scene_allocs.push_back(d_ord);
        if (int rc = upload(&d_tab, tab.data(), tab.size() * 4)) return rc;
        e->scene_allocs.push_back(d_tab);
        if (int rc = upload(&d_end, inst_end.data(), inst_end.size() * 4)) return rc;
        e->scene_allocs.push_back(d_end);
        e->os.ent_ord = d_ord; e->os.top_tab = d_tab; e->os.inst_end = d_end;
        e->os.n_top_tab = (int)table.size(); e->os.stack_slots = L.stack_need; e->os.root_link = link_of(0);
        e->ord_bytes = stack_bytes + tab.size() * 4;
        // Pair steps (rt_device.h, "pair steps"; records: rt_flat.h): in place of the above unless switched off
        if (!(e->cfg.flags & RT_FLAG_NO_PAIRS)) {
            auto skip_of = [&](size_t i) { return (size_t)rt::bits_i(L.ent_lo[i].w); };
            auto has_pair = [&](size_t i) { return is_plain_inner(i) && L.ent_right[i] >= 0 && is_plain_inner(i + 1) && is_plain_inner((size_t)L.ent_right[i]); };
            // a node whose children are leaves (no instance, no box among them): its leaves are chained, the node is skipped
            auto leaf_list = [&](size_t i) {
                if (!is_plain_inner(i) || L.ent_entry_box[i]) return false;
                for (size_t c = i + 1; c < skip_of(i); c++) { const int k = rt::bits_i(L.ent_hi[c].w); if (k < rt::ENT_SPHERE || k > rt::ENT_TRI) return false; }
                return skip_of(i) > i + 1;
            };
            // the table in LDS: whole levels of nodes that have records, from the root down
            std::vector<int32_t> ptable, plevel;
            if (has_pair(0)) plevel.push_back(0);
            const size_t pmax = std::min<size_t>(((size_t)MAX_ORD_BYTES - stack_bytes) / (rt::PAIR_REC_DWORDS * 4), (size_t)rt::LINK_INDEX);
            while (!plevel.empty() && ptable.size() + plevel.size() <= pmax) {
                ptable.insert(ptable.end(), plevel.begin(), plevel.end());
                std::vector<int32_t> next;
                for (int32_t i : plevel)
                    for (int32_t c : {i + 1, L.ent_right[i]})
                        if (has_pair((size_t)c)) next.push_back(c);
                plevel.swap(next);
            }
            std::vector<int32_t> prec(n, -1);
            for (size_t k = 0; k < ptable.size(); k++) prec[ptable[k]] = (int32_t)k;
            auto pair_link = [&](size_t i) { return rt::LINK_PAIR | (prec[i] >= 0 ? (rt::LINK_LDS | (uint32_t)prec[i]) : (uint32_t)i); };
            // where a step goes for a child: its record; its first leaf, when it is a node of chained leaves; or its entry
            auto child_link = [&](size_t c) { return has_pair(c) ? pair_link(c) : (leaf_list(c) ? (uint32_t)c + 1u : (uint32_t)c); };
            auto never_less = [](float rho) { uint32_t b = (uint32_t)rt::bits_i(rho); if (b == 0x7F800000u) b = 0x7FC00000u; return b; };   // +inf: a NaN, which culls nothing
            std::vector<rt::F4> ent2(inter);
            std::vector<rt::F4> pairs(n * 4, rt::F4{0, 0, 0, 0});
            for (size_t i = 0; i < n; i++) {
                const int kind = rt::bits_i(L.ent_hi[i].w);
                if (kind == rt::ENT_INNER) {
                    const uint32_t second = L.ent_right[i] < 0 || leaf_list(i) ? rt::LINK_NONE : (uint32_t)L.ent_right[i];
                    if (is_plain_inner(i)) ent2[2 * i + 1].z = rt::bits_f((int32_t)(has_pair(i) ? (pair_link(i) & ~rt::LINK_PAIR) | rt::ENTRY_HAS_PAIR : second));
                    else ent2[2 * i + 1].w = rt::bits_f((int32_t)second);      // INNER_INV: kind in dword 6, the link in dword 7
                    if (leaf_list(i))
                        for (size_t c = i + 1; c + 1 < skip_of(i); c++) ent2[2 * c + 1].w = rt::bits_f((int32_t)((uint32_t)rt::bits_i(ent2[2 * c + 1].w) | rt::LEAF_MORE));
                }
                if (has_pair(i)) {
                    const size_t a = i + 1, b = (size_t)L.ent_right[i];
                    const rt::F4 a0 = inter[2 * a], a1 = inter[2 * a + 1], b0 = inter[2 * b], b1 = inter[2 * b + 1];
                    pairs[4 * i + 0] = a0;
                    pairs[4 * i + 1] = {a1.x, a1.y, b0.x, b0.y};
                    pairs[4 * i + 2] = {b0.z, b0.w, b1.x, b1.y};
                    pairs[4 * i + 3] = {rt::bits_f((int32_t)child_link(a)), rt::bits_f((int32_t)child_link(b)), rt::bits_f((int32_t)never_less(a1.w)), rt::bits_f((int32_t)never_less(b1.w))};
                }
            }
            // (leaf entries: dword 7 of a sphere or cube holds its rank, of a triangle its instance; neither reaches 2^31)
            std::vector<rt::F4> ptab(ptable.size() * 4);
            for (size_t k = 0; k < ptable.size(); k++) memcpy(&ptab[4 * k], &pairs[4 * (size_t)ptable[k]], 64);
            std::vector<uint32_t> inst_start(std::max<size_t>(1, L.inst.size() / rt::RT_INST_F4), 0u);
            for (size_t k = 0; k + rt::RT_INST_F4 <= L.inst.size(); k += rt::RT_INST_F4) {
                const size_t b = (size_t)rt::bits_i(L.inst[k + 2].z);
                inst_start[k / rt::RT_INST_F4] = has_pair(b) ? pair_link(b) : (uint32_t)b;
            }
            float4 *d_ent2, *d_pairs; uint32_t *d_ptab, *d_start;
            if (int rc = upload(&d_ent2, ent2.data(), ent2.size() * 16)) return rc;
            e->scene_allocs.push_back(d_ent2);
            if (int rc = upload(&d_pairs, pairs.data(), pairs.size() * 16)) return rc;
            e->scene_allocs.push_back(d_pairs);
            if (int rc = upload(&d_ptab, ptab.data(), ptab.size() * 16)) return rc;
            e->scene_allocs.push_back(d_ptab);
            if (int rc = upload(&d_start, inst_start.data(), inst_start.size() * 4)) return rc;
            e->scene_allocs.push_back(d_start);
            e->os.ent_ord = d_ent2; e->os.top_tab = d_ptab; e->os.pair = d_pairs; e->os.inst_start = d_start;
            e->os.n_top_tab = (int)ptable.size(); e->os.root_link = has_pair(0) ? pair_link(0) : 0u;
            e->os.pairs = 1;
            e->ord_bytes = stack_bytes + ptab.size() * 16;
        }
    }
    e->sc = sc;
    e->have_scene = true;
    return RT_OK;
}

int rt_engine_set_camera(rt_engine* e, const float c[12]) {
    if (!e || !c) return fail(RT_E_INVALID, "null argument");
    e->cam.pos = V3{c[0], c[1], c[2]}; e->cam.fwd = V3{c[3], c[4], c[5]}; e->cam.u = V3{c[6], c[7], c[8]}; e->cam.v = V3{c[9], c[10], c[11]};
    e->have_camera = true;
    return RT_OK;
}

int rt_engine_set_window(rt_engine* e, int x0, int y0, int x1, int y1) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    if (x0 < 0 || y0 < 0 || x1 > e->cfg.width || y1 > e->cfg.height || x0 >= x1 || y0 >= y1) return fail(RT_E_INVALID, "window must be a non-empty rectangle inside the frame");
    e->win[0] = x0; e->win[1] = y0; e->win[2] = x1 - x0; e->win[3] = y1 - y0;
    e->debug_valid = false;
    return RT_OK;
}

int64_t rt_engine_tile_bytes(const rt_engine* e, int n_ranks) {
    if (!e || n_ranks <= 0) return -1;
    return (int64_t)tiles_per_rank(e, n_ranks) * e->cfg.tile_w * e->cfg.tile_h * 4;
}

int rt_engine_render(rt_engine* e, uint32_t seed, int rank, int n_ranks, uint8_t* d_rgba, void* stream) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    if (!e->have_scene || !e->have_camera) return fail(RT_E_INVALID, "set a scene and a camera before rendering");
    if (n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(RT_E_INVALID, "rank must be within [0, n_ranks)");
    HIP_TRY(hipSetDevice(e->cfg.device));
    // the engine's own frame buffer holds width * height pixels; a rank's packed tiles can be more (partial tiles are
    // padded to whole ones), so a split render must be given its destination
    if (n_ranks > 1 && !d_rgba) return fail(RT_E_INVALID, "a render split across ranks needs d_rgba (rt_engine_tile_bytes() bytes)");
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    e->last_stream = stream ? (hipStream_t)stream : nullptr;
    uint8_t* out = d_rgba ? d_rgba : e->d_frame;
    const rt_config& c = e->cfg;
    const Cfg g = make_cfg(c);
    const int spp = g.S * g.S;
    const int per_tile = c.tile_w * c.tile_h;
    const long long my_pixels = (long long)tiles_per_rank(e, n_ranks) * per_tile;
    const int batch_pixels = e->q.capacity / spp;
    const int n_batches = (int)((my_pixels + batch_pixels - 1) / batch_pixels);
    const int per_batch = COUNT_WORDS;
    if ((size_t)n_batches * per_batch > e->counts_capacity) return fail(RT_E_INVALID, "render needs more batches than the engine was created for");
    HIP_TRY(hipMemsetAsync(e->d_counts_all, 0, sizeof(uint32_t) * (size_t)n_batches * per_batch, st));
    if (n_ranks > 1) HIP_TRY(hipMemsetAsync(out, 0, (size_t)my_pixels * 4, st));
    e->ev_used = 0;
    e->last_batches = n_batches;
    e->serial = e->serial + 1u ? e->serial + 1u : 1u;   // never 0: the cancel word starts as 0
    if (out == e->d_frame) e->frame_dirty = true;
    {   // samples of the pixels this rank owns
        uint64_t px = 0;
        const int n_tiles = tiles_x(e) * tiles_y(e);
        for (int t = rank; t < n_tiles; t += n_ranks) {
            const int ty = t / tiles_x(e), tx = t % tiles_x(e);
            px += (uint64_t)std::min(c.tile_w, e->win[2] - tx * c.tile_w) * (uint64_t)std::min(c.tile_h, e->win[3] - ty * c.tile_h);
        }
        e->last_paths = px * (uint64_t)spp;
    }
    e->debug_valid = (n_ranks == 1 && n_batches == 1);
    for (int b = 0; b < n_batches; b++) {
        Batch bt = make_batch(e, rank, n_ranks, seed);
        bt.first_pixel = b * batch_pixels;
        bt.n_pixels = (int)std::min<long long>(batch_pixels, my_pixels - (long long)b * batch_pixels);
        uint32_t* counts = e->d_counts_all + (size_t)b * per_batch;
        if (int rc = run_passes(e, bt, st, counts, false, c.bounces)) return rc;
        {   // pixels per block and samples per step so that a step's records fill the staging tile (padded rows of rch + 1)
            const int rpb = std::max(1, std::min(256, RTILE / (spp + 1)));
            const int rch = std::min(spp, RTILE / rpb - 1);
            hipLaunchKernelGGL(resolve_kernel, dim3((bt.n_pixels + rpb - 1) / rpb), dim3(256), 0, st, g, bt, e->q.radiance, out, n_ranks > 1 ? 1 : 0,
                               rpb, rch, make_fastdiv((uint32_t)rch));
        }
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_engine_assemble(rt_engine* e, const uint8_t* d_gathered, int n_ranks, uint8_t* d_rgba, void* stream) {
    if (!e || !d_gathered || n_ranks <= 0) return fail(RT_E_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    e->last_stream = stream ? (hipStream_t)stream : nullptr;
    uint8_t* out = d_rgba ? d_rgba : e->d_frame;
    if (out == e->d_frame) e->frame_dirty = true;
    const long long per_rank = rt_engine_tile_bytes(e, n_ranks) / 4;
    const long long total = per_rank * n_ranks;
    Batch bt = make_batch(e, 0, n_ranks, 0);
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, make_cfg(e->cfg), bt,
                       reinterpret_cast<const uint32_t*>(d_gathered), per_rank, reinterpret_cast<uint32_t*>(out));
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_engine_sync(rt_engine* e) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    HIP_TRY(hipSetDevice(e->cfg.device));
    return wait_all(e);
}

int rt_engine_poll(rt_engine* e) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    HIP_TRY(hipSetDevice(e->cfg.device));
    for (hipStream_t st : {e->stream, e->last_stream}) {
        if (st == nullptr) continue;
        const hipError_t rc = hipStreamQuery(st);
        if (rc == hipErrorNotReady) return 0;
        HIP_TRY(rc);
    }
    return 1;
}

int rt_engine_read_frame(rt_engine* e, uint8_t* rgba_out) {
    if (!e || !rgba_out) return fail(RT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = wait_all(e)) return rc;
    HIP_TRY(hipMemcpy(rgba_out, e->d_frame, (size_t)e->cfg.width * e->cfg.height * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

// The pixels [x0,x1) x [y0,y1) of the engine's own frame buffer, rows of (x1 - x0) * 4 bytes one after another: what a
// caller of World::CalculateColorForScreenPosition (World.cpp:68-102) for a few pixels needs back, without the frame.
int rt_engine_read_rect(rt_engine* e, int x0, int y0, int x1, int y1, uint8_t* rgba_out) {
    if (!e || !rgba_out) return fail(RT_E_INVALID, "null argument");
    if (x0 < 0 || y0 < 0 || x1 > e->cfg.width || y1 > e->cfg.height || x0 >= x1 || y0 >= y1) return fail(RT_E_INVALID, "rectangle must be non-empty and inside the frame");
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = wait_all(e)) return rc;
    const size_t row = (size_t)(x1 - x0) * 4, pitch = (size_t)e->cfg.width * 4;
    HIP_TRY(hipMemcpy2D(rgba_out, row, e->d_frame + (size_t)y0 * pitch + (size_t)x0 * 4, pitch, row, (size_t)(y1 - y0), hipMemcpyDeviceToHost));
    return RT_OK;
}

// FrameManager::TerminateAllThreads without the wait (FrameManager.cpp:31-38: state = Quit): the remaining launches of the
// render in flight leave at once.  The store travels on a stream of its own, so it overtakes the
// render; a render queued later carries another serial and is not affected, whichever reaches the GPU first.
int rt_engine_cancel(rt_engine* e) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipStreamSynchronize(e->side_stream));      // the pinned source word is free again
    *e->h_serial = e->serial;
    HIP_TRY(hipMemcpyAsync(e->d_cancel, e->h_serial, 4, hipMemcpyHostToDevice, e->side_stream));
    return RT_OK;
}
// memset(texturePixels, 0) of FrameManager::StartNewFrame (FrameManager.cpp:13), in stream order
int rt_engine_clear_frame(rt_engine* e, void* stream) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    e->last_stream = stream ? (hipStream_t)stream : nullptr;
    HIP_TRY(hipMemsetAsync(e->d_frame, 0, (size_t)e->cfg.width * e->cfg.height * 4, st));
    e->frame_dirty = true;
    return RT_OK;
}
// FrameManager::GetTexturePixelsToPresent (FrameManager.cpp:60-71) without a second copy: the engine's frame buffer in
// page-locked host memory.  wait != 0: after everything queued has finished (the complete frame).  wait == 0: as the
// frame buffer stands now, while the render goes on — batches that have finished show their pixels, the rest what the
// buffer held before (zero after rt_engine_clear_frame), which is how the reference presents a frame in progress.
int rt_engine_map_frame(rt_engine* e, int wait, const uint8_t** rgba) {
    if (!e || !rgba) return fail(RT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t bytes = (size_t)e->cfg.width * e->cfg.height * 4;
    if (!e->h_frame) HIP_TRY(hipHostMalloc((void**)&e->h_frame, bytes, hipHostMallocDefault));
    if (wait) {
        if (e->frame_dirty) {
            // on the stream that carries the render, right behind it
            hipStream_t st = e->last_stream ? e->last_stream : e->stream;
            HIP_TRY(hipStreamSynchronize(e->side_stream));          // a snapshot may still be writing the same host buffer
            HIP_TRY(hipMemcpyAsync(e->h_frame, e->d_frame, bytes, hipMemcpyDeviceToHost, st));
            if (int rc = wait_all(e)) return rc;
            e->frame_dirty = false;
        }
    } else {
        HIP_TRY(hipMemcpyAsync(e->h_frame, e->d_frame, bytes, hipMemcpyDeviceToHost, e->side_stream));
        HIP_TRY(hipStreamSynchronize(e->side_stream));
    }
    *rgba = e->h_frame;
    return RT_OK;
}

int rt_engine_keep_samples(rt_engine* e, int enable) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = wait_all(e)) return rc;
    if (enable && !e->dbg.primary_id) {
        const size_t cap = (size_t)e->q.capacity;
        HIP_TRY(hipMalloc((void**)&e->dbg.primary_id, cap * 4));
        HIP_TRY(hipMalloc((void**)&e->dbg.primary_t, cap * 4));
        HIP_TRY(hipMalloc((void**)&e->dbg.rng_draws, cap * 4));
    } else if (!enable && e->dbg.primary_id) {
        (void)hipFree(e->dbg.primary_id); (void)hipFree(e->dbg.primary_t); (void)hipFree(e->dbg.rng_draws);
        e->dbg = Debug{nullptr, nullptr, nullptr, nullptr, e->dbg.prof};
    }
    e->debug_valid = false;
    return RT_OK;
}

static int rt_engine_read_samples_impl(rt_engine* e, float* radiance3, int32_t* primary_id, float* primary_t, uint32_t* rng_draws) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = wait_all(e)) return rc;
    if (!e->dbg.primary_id) return fail(RT_E_INVALID, "per-sample outputs are off: call rt_engine_keep_samples(e, 1) before rendering");
    if (!e->debug_valid) return fail(RT_E_INVALID, "per-sample outputs need a single-rank render that fits one batch");
    const rt_config& c = e->cfg;
    const int spp = c.samples_per_axis * c.samples_per_axis;
    const int tx_n = tiles_x(e), per_tile = c.tile_w * c.tile_h, ww = e->win[2], wh = e->win[3];
    const size_t slots = (size_t)tx_n * tiles_y(e) * per_tile * spp;
    // slots are enumerated tile by tile; hand them back in row-major pixel order
    std::vector<float4> rad;
    std::vector<int32_t> pid;
    std::vector<float> pt;
    std::vector<uint32_t> draws;
    if (radiance3) { rad.resize(slots); HIP_TRY(hipMemcpy(rad.data(), e->q.radiance, slots * 16, hipMemcpyDeviceToHost)); }
    if (primary_id) { pid.resize(slots); HIP_TRY(hipMemcpy(pid.data(), e->dbg.primary_id, slots * 4, hipMemcpyDeviceToHost)); }
    if (primary_t) { pt.resize(slots); HIP_TRY(hipMemcpy(pt.data(), e->dbg.primary_t, slots * 4, hipMemcpyDeviceToHost)); }
    if (rng_draws) { draws.resize(slots); HIP_TRY(hipMemcpy(draws.data(), e->dbg.rng_draws, slots * 4, hipMemcpyDeviceToHost)); }
    for (int py = 0; py < wh; py++)
        for (int px = 0; px < ww; px++) {
            const int tile = (py / c.tile_h) * tx_n + px / c.tile_w;
            const size_t lp = (size_t)tile * per_tile + (size_t)(py % c.tile_h) * c.tile_w + (px % c.tile_w);
            const size_t dst = ((size_t)py * ww + px) * spp;
            for (int s = 0; s < spp; s++) {
                const size_t src = lp * spp + s;
                if (radiance3) { radiance3[3 * (dst + s)] = rad[src].x; radiance3[3 * (dst + s) + 1] = rad[src].y; radiance3[3 * (dst + s) + 2] = rad[src].z; }
                if (primary_id) primary_id[dst + s] = pid[src];
                if (primary_t) primary_t[dst + s] = pt[src];
                if (rng_draws) rng_draws[dst + s] = draws[src];
            }
        }
    return RT_OK;
}

int rt_engine_trace(rt_engine* e, int n, const float* o3, const float* d3, int32_t* hit_id, float* rec9) {
    if (!e || !o3 || !d3 || !hit_id || !rec9 || n < 0) return fail(RT_E_INVALID, "bad arguments");
    if (!e->have_scene) return fail(RT_E_INVALID, "no scene set");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = wait_all(e)) return rc;
    float *d_o = nullptr, *d_d = nullptr, *d_rec = nullptr;
    int32_t* d_hit = nullptr;
    int rc = RT_OK;
    if ((rc = upload(&d_o, o3, (size_t)n * 12)) == RT_OK && (rc = upload(&d_d, d3, (size_t)n * 12)) == RT_OK) {
        if (hipMalloc((void**)&d_hit, (size_t)n * 4) != hipSuccess || hipMalloc((void**)&d_rec, (size_t)n * 36) != hipSuccess)
            rc = fail(RT_E_NOMEM, "device allocation failed");
        else {
            if (e->sc.has_fog) hipLaunchKernelGGL(trace_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->sc, e->sc_ref, n, d_o, d_d, d_hit, d_rec);
            else hipLaunchKernelGGL(trace_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->sc, e->sc_ref, n, d_o, d_d, d_hit, d_rec);
            hipError_t he = hipStreamSynchronize(e->stream);
            if (he != hipSuccess) rc = fail(RT_E_DEVICE, hipGetErrorString(he));
            else {
                (void)hipMemcpy(hit_id, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(rec9, d_rec, (size_t)n * 36, hipMemcpyDeviceToHost);
            }
        }
    }
    (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_hit); (void)hipFree(d_rec);
    return rc;
}

// rt_engine_trace for callers that must not allocate: the rays and the results are staged in this GPU's static array
// instead of four fresh allocations, and cross the bus through the engine's page-locked mirror of that array (allocated
// with the engine), so that the copies are plain DMA transfers.  No hipMalloc / hipFree / allocating library call here.
int rt_engine_trace_static(rt_engine* e, int n, const float* o3, const float* d3, int32_t* hit_id, float* rec9) {
    if (!e || !o3 || !d3 || !hit_id || !rec9 || n < 0) return fail(RT_E_INVALID, "bad arguments");
    if (!e->have_scene) return fail(RT_E_INVALID, "no scene set");
    if (n > RT_TRACE_STATIC_MAX_RAYS) return fail(RT_E_INVALID, "rt_engine_trace_static takes at most RT_TRACE_STATIC_MAX_RAYS (32768) rays per call");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (!e->d_scratch || !e->h_stage) return fail(RT_E_DEVICE, "the static staging array was not resolved on this device");
    if (int rc = wait_all(e)) return rc;   // like rt_engine_trace: a render on a stream the caller brought has finished
    // fixed sections of the array, sized for the maximum: origins | directions | hit ids | records.  Every byte a call
    // reads was written by the same call (inputs by the copies below, outputs by the kernel for rays 0..n-1), so what an
    // earlier, larger call left behind is never looked at and nothing needs clearing.
    constexpr size_t AT_D = (size_t)RT_TRACE_STATIC_MAX_RAYS * 12, AT_HIT = (size_t)RT_TRACE_STATIC_MAX_RAYS * 24, AT_REC = (size_t)RT_TRACE_STATIC_MAX_RAYS * 28;
    unsigned char* base = e->d_scratch;
    unsigned char* host = e->h_stage;
    float* d_o = reinterpret_cast<float*>(base);
    float* d_d = reinterpret_cast<float*>(base + AT_D);
    int32_t* d_hit = reinterpret_cast<int32_t*>(base + AT_HIT);
    float* d_rec = reinterpret_cast<float*>(base + AT_REC);
    memcpy(host, o3, (size_t)n * 12);
    memcpy(host + AT_D, d3, (size_t)n * 12);
    HIP_TRY(hipMemcpyAsync(d_o, host, (size_t)n * 12, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d_d, host + AT_D, (size_t)n * 12, hipMemcpyHostToDevice, e->stream));
    if (e->sc.has_fog) hipLaunchKernelGGL(trace_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->sc, e->sc_ref, n, d_o, d_d, d_hit, d_rec);
    else hipLaunchKernelGGL(trace_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->sc, e->sc_ref, n, d_o, d_d, d_hit, d_rec);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host + AT_HIT, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(host + AT_REC, d_rec, (size_t)n * 36, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(hit_id, host + AT_HIT, (size_t)n * 4);
    memcpy(rec9, host + AT_REC, (size_t)n * 36);
    return RT_OK;
}

static int rt_engine_ray_color_impl(rt_engine* e, int n, const float* o3, const float* d3, int bounces, const uint32_t* keys, float* rgb3) {
    if (!e || !o3 || !d3 || !keys || !rgb3 || n < 0) return fail(RT_E_INVALID, "bad arguments");
    if (!e->have_scene) return fail(RT_E_INVALID, "no scene set");
    if (bounces < 0 || bounces > e->cfg.bounces) return fail(RT_E_INVALID, "bounces must be within the engine's configured bounce count");
    if (n > e->q.capacity) return fail(RT_E_INVALID, "more rays than the batch capacity");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = wait_all(e)) return rc;   // the queues may still belong to a render on a stream the caller brought
    float *d_o = nullptr, *d_d = nullptr;
    uint32_t *d_keys = nullptr, *d_counts = nullptr;
    int rc = RT_OK;
    if ((rc = upload(&d_o, o3, (size_t)n * 12)) == RT_OK && (rc = upload(&d_d, d3, (size_t)n * 12)) == RT_OK &&
        (rc = upload(&d_keys, keys, (size_t)n * 4)) == RT_OK) {
        const size_t words = COUNT_WORDS;
        if (hipMalloc((void**)&d_counts, sizeof(uint32_t) * words) != hipSuccess) rc = fail(RT_E_NOMEM, "device allocation failed");
        else {
            (void)hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * words, e->stream);
            // the rays enter pass 0 through queue (0+1)&1 = 1
            hipLaunchKernelGGL(load_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, d_o, d_d, e->q.a[1], e->q.b[1], d_counts);
            e->serial = e->serial + 1u ? e->serial + 1u : 1u;
            const Debug saved = e->dbg;
            const bool saved_timing = e->time_kernels;
            e->dbg = Debug{nullptr, nullptr, nullptr, d_keys, e->dbg.prof};
            e->time_kernels = false;
            Batch bt = make_batch(e, 0, 1, 0);
            bt.n_pixels = 0;
            rc = run_passes(e, bt, e->stream, d_counts, true, bounces);
            e->dbg = saved;
            e->time_kernels = saved_timing;
            hipError_t he = hipStreamSynchronize(e->stream);
            if (rc == RT_OK && he != hipSuccess) rc = fail(RT_E_DEVICE, hipGetErrorString(he));
            if (rc == RT_OK) {
                std::vector<float4> rad((size_t)n);
                (void)hipMemcpy(rad.data(), e->q.radiance, (size_t)n * 16, hipMemcpyDeviceToHost);
                for (int i = 0; i < n; i++) { rgb3[3 * i] = rad[i].x; rgb3[3 * i + 1] = rad[i].y; rgb3[3 * i + 2] = rad[i].z; }
            }
        }
    }
    (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_keys); (void)hipFree(d_counts);
    e->debug_valid = false;
    return rc;
}

static int rt_engine_get_stats_impl(rt_engine* e, rt_stats* out) {
    if (!e || !out) return fail(RT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = wait_all(e)) return rc;
    memset(out, 0, sizeof *out);
    const int nb = e->last_batches, nc = e->n_counts, per_batch = COUNT_WORDS;
    if (nb <= 0) return RT_OK;
    e->h_counts.resize((size_t)nb * per_batch);
    HIP_TRY(hipMemcpy(e->h_counts.data(), e->d_counts_all, sizeof(uint32_t) * (size_t)nb * per_batch, hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; b++) {
        for (int k = 1; k < nc - 1 && k < 33; k++) out->rays_at_depth[k] += e->h_counts[(size_t)b * per_batch + k];
        for (int k = 0; k < nc - 1 && k < 33; k++) {
            const uint32_t* w2 = &e->h_counts[(size_t)b * per_batch + BOX_COUNT_AT + 2 * k];
            out->box_tests += (uint64_t)w2[0] | ((uint64_t)w2[1] << 32);
        }
    }
#ifdef RT_PROFILE
    if (e->dbg.prof) {
        unsigned long long h[32 * 16];
        HIP_TRY(hipMemcpy(h, e->dbg.prof, sizeof h, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(e->dbg.prof, 0, sizeof h));
        fprintf(stderr, "pass waves cycles/wave rounds/wave box%% prim%% special%% refill%% prim_steps/wave lanes/prim_step specials/wave refills/wave (all renders since the last call)\n");
        for (int k = 0; k < 16; k++) {
            const unsigned long long* p = h + 32 * k;
            if (!p[0]) continue;
            const double w = (double)p[0], cyc = (double)p[1];
            const double iters = (double)p[11], lanes = (double)p[12];
            fprintf(stderr, "%2d %6llu %9.0f %7.1f %5.1f %5.1f %5.1f %5.1f %7.1f %5.1f %6.1f %6.1f  box iters/wave %.0f lanes/iter %.1f  tri finishes/wave %.1f lanes %.1f\n", k, p[0], cyc / w, p[2] / w, 100.0 * p[3] / cyc,
                    100.0 * p[4] / cyc, 100.0 * p[5] / cyc, 100.0 * p[6] / cyc, p[7] / w, p[7] ? (double)p[8] / p[7] : 0.0, p[9] / w, p[10] / w, iters / w, iters ? lanes / iters : 0.0, p[13] / w, p[13] ? (double)p[14] / p[13] : 0.0);
            fprintf(stderr, "     state changes per wave (lanes each): end of walk %.1f (%.1f), end of instance %.1f (%.1f), instance %.1f (%.1f), box as written %.1f (%.1f)\n",
                    p[16] / w, p[16] ? (double)p[20] / p[16] : 0.0, p[17] / w, p[17] ? (double)p[21] / p[17] : 0.0, p[18] / w, p[18] ? (double)p[22] / p[18] : 0.0, p[19] / w, p[19] ? (double)p[23] / p[19] : 0.0);
        }
    }
#endif
    out->paths = e->last_paths;
    out->rays_at_depth[0] = e->last_paths;
    for (int k = 0; k < 33; k++) out->rays += out->rays_at_depth[k];
    return RT_OK;
}

int rt_engine_time_kernels(rt_engine* e, int enable) {
    if (!e) return fail(RT_E_INVALID, "null engine");
    e->time_kernels = enable != 0;
    e->ev_used = 0;
    return RT_OK;
}
int rt_engine_kernel_times(rt_engine* e, double out2[2]) {
    if (!e || !out2) return fail(RT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    double total = 0;
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(e->ev[i + 1]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
        total += ms;
    }
    out2[0] = total;
    out2[1] = (double)(e->ev_used / 2);
    return RT_OK;
}

int rt_engine_eval_math(rt_engine* e, int n, const float* x, const float* y, float* sin_x, float* atan2_yx) {
    if (!e || !x || !y || !sin_x || !atan2_yx || n <= 0) return fail(RT_E_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(e->cfg.device));
    DevBuf dx, dy, ds, da;
    if (int rc = dx.from_host(x, (size_t)n * 4)) return rc;
    if (int rc = dy.from_host(y, (size_t)n * 4)) return rc;
    if (int rc = ds.alloc((size_t)n * 4)) return rc;
    if (int rc = da.alloc((size_t)n * 4)) return rc;
    hipLaunchKernelGGL(math_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, dx.as<float>(), dy.as<float>(), ds.as<float>(), da.as<float>());
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(sin_x, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(atan2_yx, da.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}
int rt_engine_eval_rng(rt_engine* e, int n, const uint32_t* key, const uint32_t* ctr, uint32_t* r31, float* ud) {
    if (!e || !key || !ctr || !r31 || !ud || n <= 0) return fail(RT_E_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(e->cfg.device));
    DevBuf dk, dc, dr, du;
    if (int rc = dk.from_host(key, (size_t)n * 4)) return rc;
    if (int rc = dc.from_host(ctr, (size_t)n * 4)) return rc;
    if (int rc = dr.alloc((size_t)n * 4)) return rc;
    if (int rc = du.alloc((size_t)n * 4)) return rc;
    hipLaunchKernelGGL(rng_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, dk.as<uint32_t>(), dc.as<uint32_t>(), dr.as<uint32_t>(), du.as<float>());
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(r31, dr.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ud, du.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}
// The slab test's reciprocal-based division against the hardware's IEEE division: the n given pairs plus
// threads x rounds x 4 generated pairs next to rounding boundaries.  out[0] = disagreeing pairs; first_bad4 = (a, b, got, want).
int rt_engine_check_division(rt_engine* e, int n, const float* a, const float* b, uint32_t seed, int threads, int rounds,
                             uint64_t* mismatches, float* first_bad4) {
    if (!e || !mismatches || !first_bad4 || n < 0 || threads <= 0 || rounds < 0 || (n > 0 && (!a || !b))) return fail(RT_E_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(e->cfg.device));
    DevBuf da, db, dm, dbad;
    if (int rc = da.from_host(a, (size_t)n * 4)) return rc;
    if (int rc = db.from_host(b, (size_t)n * 4)) return rc;
    if (int rc = dm.alloc(16)) return rc;
    if (int rc = dbad.alloc(16)) return rc;
    HIP_TRY(hipMemset(dm.p, 0, 16));
    HIP_TRY(hipMemset(dbad.p, 0, 16));
    const int total = std::max(n, threads);
    hipLaunchKernelGGL(div_check_kernel, dim3((total + 255) / 256), dim3(256), 0, e->stream, n, da.as<float>(), db.as<float>(), seed, rounds,
                       dm.as<unsigned long long>(), dbad.as<float>());
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned long long m[2];
    HIP_TRY(hipMemcpy(m, dm.p, 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(first_bad4, dbad.p, 16, hipMemcpyDeviceToHost));
    *mismatches = m[0];
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Several GPUs behind one handle, in one process: what a front end that owns a single FrameManager needs in order to use
// a whole node.  The scene is replicated; device k renders tiles k, k+n, ... into a packed buffer of its own; the display
// device (the first of the list) waits for each peer's event, pulls the peer's buffer over with one peer-to-peer copy and
// scatters all tiles into its frame buffer.  No host thread per device: everything is queued from the caller's thread,
// asynchronously.  A device may be listed more than once (each entry gets an engine of its own), which is how a one-GPU
// machine exercises the same code.
struct rt_multi {
    std::vector<rt_engine*> eng;
    std::vector<uint8_t*> d_tiles;      // per engine, on its device: rt_engine_tile_bytes
    std::vector<hipEvent_t> done;       // per engine: its tiles are complete
    std::vector<hipEvent_t> taken;      // per engine: the display device has copied its tiles (recorded on the display stream)
    bool rendered = false;              // `taken` events have been recorded at least once
    uint8_t* d_gathered = nullptr;      // on the display device: n packed buffers back to back
    size_t tile_bytes = 0;
};

void rt_multi_destroy(rt_multi* m) {
    if (!m) return;
    for (size_t k = 0; k < m->eng.size(); k++) {
        if (!m->eng[k]) continue;
        (void)hipSetDevice(m->eng[k]->cfg.device);
        (void)wait_all(m->eng[k]);
        if (k < m->d_tiles.size()) (void)hipFree(m->d_tiles[k]);
        if (k < m->done.size() && m->done[k]) (void)hipEventDestroy(m->done[k]);
        if (k < m->taken.size() && m->taken[k]) (void)hipEventDestroy(m->taken[k]);
    }
    if (!m->eng.empty() && m->eng[0]) { (void)hipSetDevice(m->eng[0]->cfg.device); (void)hipFree(m->d_gathered); }
    for (rt_engine* e : m->eng) rt_engine_destroy(e);
    delete m;
}

int rt_multi_create(const rt_config* cfg, const int32_t* device_ids, int n_dev, rt_multi** out) {
    if (!cfg || !device_ids || !out || n_dev <= 0 || n_dev > 64) return fail(RT_E_INVALID, "rt_multi_create: 1 to 64 devices");
    return rt::guard([&] {
        rt_multi* m = new rt_multi();
        m->eng.assign((size_t)n_dev, nullptr);
        m->d_tiles.assign((size_t)n_dev, nullptr);
        m->done.assign((size_t)n_dev, nullptr);
        m->taken.assign((size_t)n_dev, nullptr);
        for (int k = 0; k < n_dev; k++) {
            rt_config c = *cfg;
            c.device = device_ids[k];
            // an engine's queues hold its own share of the frame at most
            {
                const int tw = c.tile_w > 0 ? c.tile_w : 64, th = c.tile_h > 0 ? c.tile_h : 64;
                const long long tiles = (long long)((c.width + tw - 1) / tw) * ((c.height + th - 1) / th);
                const long long share = ((tiles + n_dev - 1) / n_dev) * tw * th * (long long)c.samples_per_axis * c.samples_per_axis;
                const long long cap = c.max_batch_paths > 0 ? c.max_batch_paths : (1ll << 27);
                c.max_batch_paths = (int)std::min(cap, share);
            }
            if (int rc = rt_engine_create(&c, &m->eng[k])) { rt_multi_destroy(m); return rc; }
            hipError_t he = hipEventCreateWithFlags(&m->done[k], hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&m->taken[k], hipEventDisableTiming);
            if (he == hipSuccess && n_dev > 1) {
                m->tile_bytes = (size_t)rt_engine_tile_bytes(m->eng[k], n_dev);
                he = hipMalloc((void**)&m->d_tiles[k], m->tile_bytes);
            }
            if (he != hipSuccess) { rt_multi_destroy(m); return fail(RT_E_DEVICE, hipGetErrorString(he)); }
        }
        if (n_dev > 1) {
            (void)hipSetDevice(m->eng[0]->cfg.device);
            if (hipMalloc((void**)&m->d_gathered, m->tile_bytes * (size_t)n_dev) != hipSuccess) { rt_multi_destroy(m); return fail(RT_E_NOMEM, "device allocation failed"); }
            for (int k = 1; k < n_dev; k++)   // direct access to the peers' buffers where the devices differ (xGMI); ignored if already on
                if (m->eng[k]->cfg.device != m->eng[0]->cfg.device) (void)hipDeviceEnablePeerAccess(m->eng[k]->cfg.device, 0);
            (void)hipGetLastError();
        }
        *out = m;
        return RT_OK;
    });
}
int rt_multi_count(const rt_multi* m) { return m ? (int)m->eng.size() : 0; }
rt_engine* rt_multi_engine(rt_multi* m, int k) { return (m && k >= 0 && k < (int)m->eng.size()) ? m->eng[k] : nullptr; }
int rt_multi_set_scene(rt_multi* m, const rt_scene* s) {
    if (!m) return fail(RT_E_INVALID, "null handle");
    for (rt_engine* e : m->eng) if (int rc = rt_engine_set_scene(e, s)) return rc;
    return RT_OK;
}
int rt_multi_set_camera(rt_multi* m, const float cam12[12]) {
    if (!m) return fail(RT_E_INVALID, "null handle");
    for (rt_engine* e : m->eng) if (int rc = rt_engine_set_camera(e, cam12)) return rc;
    return RT_OK;
}
// One frame into the display engine's own frame buffer (rt_multi_map_frame / rt_engine_read_frame on engine 0 hand it out).
// Frames may be queued one behind the other without a wait in between.  What keeps them apart: engine k's packed tiles
// are written by its render on its own stream and read by the copy on the display stream, so (1) the copy waits for the
// render's event `done[k]`, and (2) the NEXT render of engine k — which begins by zeroing that same buffer — waits for the
// event `taken[k]` recorded on the display stream behind the copy.  The gathered buffer and the frame buffer are only
// touched on the display stream (copies, then the scatter), which orders them by itself.
int rt_multi_render(rt_multi* m, uint32_t frame_seed) {
    if (!m) return fail(RT_E_INVALID, "null handle");
    const int n = (int)m->eng.size();
    if (n == 1) return rt_engine_render(m->eng[0], frame_seed, 0, 1, nullptr, nullptr);
    for (int k = 0; k < n; k++) {
        rt_engine* e = m->eng[k];
        HIP_TRY(hipSetDevice(e->cfg.device));
        if (m->rendered && k > 0) HIP_TRY(hipStreamWaitEvent(e->stream, m->taken[k], 0));   // (engine 0 renders on the display stream itself)
        if (int rc = rt_engine_render(e, frame_seed, k, n, m->d_tiles[k], nullptr)) return rc;
        HIP_TRY(hipEventRecord(m->done[k], e->stream));     // the engine's device is current after rt_engine_render
    }
    rt_engine* d = m->eng[0];
    HIP_TRY(hipSetDevice(d->cfg.device));
    for (int k = 0; k < n; k++) {
        rt_engine* e = m->eng[k];
        if (k > 0) HIP_TRY(hipStreamWaitEvent(d->stream, m->done[k], 0));
        uint8_t* dst = m->d_gathered + (size_t)k * m->tile_bytes;
        if (e->cfg.device == d->cfg.device) HIP_TRY(hipMemcpyAsync(dst, m->d_tiles[k], m->tile_bytes, hipMemcpyDeviceToDevice, d->stream));
        else HIP_TRY(hipMemcpyPeerAsync(dst, d->cfg.device, m->d_tiles[k], e->cfg.device, m->tile_bytes, d->stream));
        if (k > 0) HIP_TRY(hipEventRecord(m->taken[k], d->stream));
    }
    m->rendered = true;
    return rt_engine_assemble(d, m->d_gathered, n, nullptr, nullptr);
}
int rt_multi_poll(rt_multi* m) {
    if (!m) return fail(RT_E_INVALID, "null handle");
    for (rt_engine* e : m->eng) { const int rc = rt_engine_poll(e); if (rc != 1) return rc; }
    return 1;
}
int rt_multi_sync(rt_multi* m) {
    if (!m) return fail(RT_E_INVALID, "null handle");
    for (size_t k = m->eng.size(); k-- > 0;) if (int rc = rt_engine_sync(m->eng[k])) return rc;   // the display engine last: it waits for the others' tiles
    return RT_OK;
}
int rt_multi_cancel(rt_multi* m) {
    if (!m) return fail(RT_E_INVALID, "null handle");
    for (rt_engine* e : m->eng) if (int rc = rt_engine_cancel(e)) return rc;
    return RT_OK;
}
int rt_multi_clear_frame(rt_multi* m) { return m ? rt_engine_clear_frame(m->eng[0], nullptr) : fail(RT_E_INVALID, "null handle"); }
int rt_multi_map_frame(rt_multi* m, int wait, const uint8_t** rgba) {
    if (!m) return fail(RT_E_INVALID, "null handle");
    if (wait) for (size_t k = m->eng.size(); k-- > 1;) if (int rc = rt_engine_sync(m->eng[k])) return rc;
    return rt_engine_map_frame(m->eng[0], wait, rgba);
}

// The entry points above that allocate host memory, behind rt::guard: no exception crosses the C boundary.
int rt_engine_set_scene(rt_engine* e, const rt_scene* s) { return rt::guard([&] { return rt_engine_set_scene_impl(e, s); }); }
int rt_engine_read_samples(rt_engine* e, float* radiance3, int32_t* primary_id, float* primary_t, uint32_t* rng_draws) { return rt::guard([&] { return rt_engine_read_samples_impl(e, radiance3, primary_id, primary_t, rng_draws); }); }
int rt_engine_ray_color(rt_engine* e, int n, const float* o3, const float* d3, int bounces, const uint32_t* keys, float* rgb3) { return rt::guard([&] { return rt_engine_ray_color_impl(e, n, o3, d3, bounces, keys, rgb3); }); }
int rt_engine_get_stats(rt_engine* e, rt_stats* out) { return rt::guard([&] { return rt_engine_get_stats_impl(e, out); }); }

}  // extern "C"
