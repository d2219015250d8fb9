// Device code of the render loop: ray/box and ray/primitive tests, stack traversal of the flattened
// hierarchy, hit attributes, lights, Phong, cube map, and the iterative form of the reference's recursive
// colour evaluation.  Included by rt_kernels.hip only.
//
// One lane owns one ray at a time (wave64; 64 rays in flight per wave).  The traversal stack lives in LDS,
// laid out [entry][lane] so that lanes never collide on a bank whatever depth each of them is at.
#pragma once
#include "rt_amd.h"
#include "rt_layout.h"

namespace rt {

// Per-light constants the reference recomputes at every hit (reference src/light.cpp:151-155,188-194);
// they only depend on the light and the sampling tunables, so the host evaluates them once per frame with
// the same fp32 operations.
struct HitEntry;

struct LightDev {
    int32_t type;
    float v[21];
    float recA, recB;   // segment: sampleCountRec, segmentLengthRec; parallelogram: dirISamplesRec, dirJSamplesRec
    int32_t nA, nB;     // trip counts of the reference's sample loops (its float loop bounds evaluated on the host)
    float step[3];      // segment: sampleDistanceSegment
    int32_t draws;      // rand() calls one evaluation of this light consumes
    int32_t drawBase;   // rand() calls the lights before this one consume at a hit
    int32_t shadowBase; // index of this light's first sample among the hit's shadow-casting samples, or -1
};

enum { INLINE_LIGHTS = 16,  // lights that travel in the kernel argument block; more of them go through device memory
       MAX_LIGHTS = 4096 };  // (a bound on the table, not a design limit)

// Wavefront evaluation of the reference's recursion (src/render.cpp:18-80): the call tree of every pixel sample of a
// chunk is laid out as nodes, generation by generation (a generation = the calls at one recursion depth), each
// generation a contiguous range.  A node is one getFinalColor call: its ray, what the call found, and what it still
// waits for.
enum {
    WF_DEAD = 0,    // no call (a glossy sample below the surface, a lens ray of a pixel in focus): contributes nothing
    WF_PENDING = 1, // ray written, not traced yet
    WF_MISS = 2,    // value = sky box texel or black
    WF_VALUE = 3,   // value set at the hit (reflection map)
    WF_LO = 4,      // value = direct lighting Lo (written to coef by the shade pass)
    WF_BLEND = 5,   // value = a * Lo + (1 - a) * child          (render.cpp:43)
    WF_MIRROR = 6,  // value = ks * child                        (render.cpp:61)
    WF_GLOSSY = 7   // value = ks * (sum of children) / filter   (render.cpp:53-59)
};
struct WfRay {
    float o[3], d[3];
    uint32_t work;   // (sample - sampleBegin) * pixels + pixel of the pixel sample the call belongs to
    uint32_t parent; // node index of the calling node (unused for roots)
};
// (a node's meta word — kind | specular depth left << 4 | see-through nesting << 8 | WF_ROOT — and its value live in
// arrays of their own: the passes that only skip or sum nodes touch 4 or 12 bytes per node instead of the record)
struct WfNode {
    float t;
    int32_t slot;
    uint32_t firstChild;
    uint32_t randBase; // index on the rand() stream of the first draw this call's lighting consumes
    float coef[3];     // WF_MIRROR / WF_GLOSSY: ks; WF_BLEND: Lo (written by the shade pass)
    float a;           // WF_BLEND: the material's transparency
};
// what a glossy call hands to the lanes that generate its samples (src/gloss.cpp:14-51)
struct WfGlossy {
    float origin[3], dir[3], normal[3];
    float side;
    uint32_t node, key;
};
enum : uint32_t { WF_ROOT = 1u << 16 }; // the node of a pixel sample's own call
__host__ __device__ inline uint32_t wf_kind(uint32_t meta) { return meta & 15u; }
__host__ __device__ inline uint32_t wf_depth(uint32_t meta) { return (meta >> 4) & 15u; }
__host__ __device__ inline uint32_t wf_blend_depth(uint32_t meta) { return (meta >> 8) & 255u; }
__host__ __device__ inline uint32_t wf_meta(uint32_t kind, uint32_t depth, uint32_t blendDepth) { return kind | (depth << 4) | (blendDepth << 8); }

struct FrameParams {
    SceneView scene;
    quat camQ;
    float camPos[3], camUp[3], camLeft[3];
    float halfW, halfH;
    int32_t width, height;
    // The rows this call renders, as LOCAL rows 0..rowsLocal-1 in Screen order (top-most first): local row r is
    // Screen row (counted from the top) rowTop + (r / rowBlock) * blockStride * rowBlock + r % rowBlock.  One range
    // of rows is a single block; interleaved blocks (multi-GPU tiling) have blockStride = number of ranks.
    int32_t rowsLocal, rowTop, rowBlock, blockStride;
    rt_features f;
    rt_params p;
    LightDev lights[INLINE_LIGHTS]; // travels in the kernel argument block: read with scalar loads, no upload to order
    const LightDev* lightTable;     // null, or all nLights lights in device memory when there are more than INLINE_LIGHTS
    int32_t nLights;
    int32_t drawsPerHit; // rand() calls one computeLightContribution consumes
    // Output.  One lane evaluates one (pixel, sample) pair.  With a single sample per pixel the colour goes
    // straight to `out` (Screen order); otherwise to plane s of `samples` ([sample][row][x][rgb], rows in
    // Screen order) and a resolve pass sums the planes in the reference's order.
    float* out;
    float* samples;
    uint8_t* dofFlags;     // per pixel: 1 when the thin-lens samples take part (render.cpp:110)
    int32_t sampleBegin, sampleEnd, samplesPerPixel;
    unsigned long long* stats; // primary, shadow, bounce, probe, box, tri, overflow
    int32_t countTests;
    // queue passes (direct-lighting frames): the hits of the camera rays, one entry each
    struct HitEntry* queue;
    unsigned long long* queueCounters; // Q_COUNT counters of the current pass, see below: those of the lighting passes
    unsigned long long* primaryCounters; // ... and those of the primary pass (hit-queue length, rays that gave up, cursors);
                                       // the same block unless two lighting rounds of a pass run side by side
    uint32_t* gaveUpQueue;             // camera rays (sample * pixels + pixel) that gave up in k_primary
    int32_t shadowQueueReversed;       // shadow work items are stored from the END of shadowQueue downwards (the second
    uint32_t shadowQueueItems;         // lighting round shares the buffer with the first); its capacity in items
    uint32_t* shadowQueue;             // work items hit * shadowSamplesPerHit + j that need a shadow ray; during the
                                       // primary pass: the camera rays (sample * pixels + pixel) that gave up
    uint32_t* shadeList;               // hit-queue entries whose colour the shade pass computes (written by the cull pass:
                                       // the others are black and done); as many entries as the queue, filled from the
                                       // front, or from the back when shadowQueueReversed is set
    uint2* laneStacks;                 // when not null: traversal stacks in device memory, [wave][entry][lane], for
    uint32_t laneStackWaves;           // k_closest on the second stream, which then needs no LDS and fits beside the
                                       // shadow pass that fills every CU's LDS; the waves it has room for
    int32_t primaryBudget;             // inner nodes a camera ray may visit in the per-lane kernel before it gives up
    int32_t primaryStack;              // ... and the stack entries it may hold
    size_t queueCapacity;        // entries
    uint8_t* visibility;         // [hit][shadowSamplesPerHit]: 1 = light sample visible
    int32_t shadowSamplesPerHit; // light samples per hit that cast a shadow ray under the current features
    uint32_t logicalBlocks;      // (tile, sample) blocks of the current pass
    uint32_t primaryChunk;       // logical blocks a wave of k_primary takes per cursor grab
    int32_t pairStackLds;        // shadow pass: entries of a wave's pair stack kept in LDS; the rest of the bound spills
    uint32_t* pairSpill;         // [pairSpillWaves][pairSpillEntries] spill area of the pair stacks
    uint32_t pairSpillEntries;
    uint32_t pairSpillWaves;
    int32_t streamGroup;         // shadow pass: consecutive queries that share a stack entry (1, 2 or 4)
    // wavefront evaluation of bounce features (null / zero when the call does not use it)
    WfRay* wfRays;
    WfNode* wfNodes;
    uint32_t* wfMeta;
    float* wfValue;        // [node][rgb]
    uint32_t* wfHits;      // [node] calls in the node's subtree that evaluate the lighting
    uint32_t* wfHitNode;   // hit-queue entry -> node whose direct lighting it is
    WfGlossy* wfGlossy;    // glossy calls of the current generation
    uint32_t wfCapacity;   // nodes
    uint32_t wfGlossyCapacity;
    uint32_t wfBegin, wfEnd; // node range of the generation a kernel works on ...
    int32_t wfGen;           // ... or, when >= 0, the generation whose range [table[wfGen], table[wfGen + 1]) the kernel reads
                             // from the generation table on the device (Q_WF_GEN): the host does not know it yet
    unsigned long long* wfReport; // pinned host memory the device reports a chunk's node count, overflow flag and table to
    uint32_t wfRootBegin;    // work id of the chunk's first root
    int32_t wfLighting;      // the queue passes read their rays from the nodes and write Lo there
    int32_t wfRootsTurn;     // k_wf_combine: evaluate the pixel samples' own nodes (else: every other node)
};

// Light i of the frame.  The kernels of the lighting passes are compiled for either home of the lights (TABLE), so that
// the usual case reads them from the argument block with scalar loads; the others decide at run time.
template <bool TABLE>
__device__ __forceinline__ const LightDev& light_in(const FrameParams& P, int i)
{
    if (TABLE) return P.lightTable[i];
    return P.lights[i];
}
__device__ __forceinline__ const LightDev& light_at(const FrameParams& P, int i) { return P.lightTable ? P.lightTable[i] : P.lights[i]; }

// Device counters of a pass.  Each sits in a 128-byte line of its own: atomics on one line are served one after the
// other, and the kernels of a pass hammer different counters at the same time.
enum {
    Q_HITS = 0,              // hit-queue entries
    Q_SHADOW_PACKETS = 16,   // packets of shadow work items handed out
    Q_SHADOW_ITEMS = 32,     // shadow work items queued
    Q_GAVE_UP_PACKETS = 48,  // packets of given-up camera rays handed out
    Q_GAVE_UP = 64,          // camera rays that gave up
    Q_PRIMARY_CURSORS = 80,  // 16 cursors (one per 128-byte line) that hand out k_primary's chunks of logical blocks
    Q_WF_NODES = 80 + 16 * 16,     // wavefront: nodes allocated so far in the chunk
    Q_WF_GLOSSY = Q_WF_NODES + 16, // glossy calls of the current generation
    Q_WF_OVERFLOW = Q_WF_GLOSSY + 16, // set when a generation did not fit the node or glossy capacity
    Q_RANGE_BEGIN = Q_WF_OVERFLOW + 16, // lighting passes (cull / shadow / shade): the hit-queue entries they work on,
    Q_RANGE_END = Q_RANGE_BEGIN + 16,   // set by k_range once the entries exist
    Q_SHADOW_XCD = Q_RANGE_END + 16,    // 8 cursors (one per 128-byte line): shadow packets handed out per eighth of the queue
    Q_WF_GEN = Q_SHADOW_XCD + 8 * 16,   // wavefront: generation table, entry g = nodes allocated when generation g's trace began
    Q_SHADE_ITEMS = Q_WF_GEN + 64,      // hits of the round that the shade pass has to evaluate (the cull pass lists them)
    Q_COUNT = Q_SHADE_ITEMS + 16
};
enum { WF_MAX_GENERATIONS = 60 };        // (the table's size; ray_depth <= 8 and at most 24 see-through continuations)

struct Counts {
    uint32_t primary = 0, shadow = 0, bounce = 0, probe = 0, box = 0, tri = 0, overflow = 0;
};

struct RayD {
    vec3 o, d;
    // per-axis reciprocal and "direction component is zero" class of the slab test (reference
    // src/intersect.cpp:114-126); functions of the ray only, so computed once per ray
    vec3 inv;
    bool zx, zy, zz;
    // FAST CLASS of the slab test: no direction component in the zero class or non-finite, every origin component
    // within +-FAST_BOUND (see box_pair_fast)
    bool fast;
};
constexpr float FAST_BOUND = 1e30f;

__device__ __forceinline__ RayD make_ray(vec3 o, vec3 d)
{
    RayD r;
    r.o = o;
    r.d = d;
    r.inv = make3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.zx = is_zero(d.x); r.zy = is_zero(d.y); r.zz = is_zero(d.z);
    r.fast = !(r.zx || r.zy || r.zz) && fabsf(d.x) <= FLT_MAX && fabsf(d.y) <= FLT_MAX && fabsf(d.z) <= FLT_MAX
        && fabsf(o.x) <= FAST_BOUND && fabsf(o.y) <= FAST_BOUND && fabsf(o.z) <= FAST_BOUND;
    return r;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One 64-byte record as four 16-byte loads.
struct Rec64 {
    f32x4 a, b, c, d;
};
__device__ __forceinline__ Rec64 load_rec(const void* p)
{
    const f32x4* q = static_cast<const f32x4*>(p);
    Rec64 r;
    r.a = q[0];
    r.b = q[1];
    r.c = q[2];
    r.d = q[3];
    return r;
}

// One 64-byte record through the scalar data path, for addresses that are the same in every lane of the wave
// (packet traversal): one s_load instead of 64 lanes x 4 vector loads, and the values land in SGPRs that the
// box / triangle arithmetic reads as scalar operands.  Load only: the scalar cache is never written.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ Rec64 load_rec_uniform(const void* p)
{
    u32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    Rec64 r;
    r.a = f32x4 { __uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]) };
    r.b = f32x4 { __uint_as_float(v[4]), __uint_as_float(v[5]), __uint_as_float(v[6]), __uint_as_float(v[7]) };
    r.c = f32x4 { __uint_as_float(v[8]), __uint_as_float(v[9]), __uint_as_float(v[10]), __uint_as_float(v[11]) };
    r.d = f32x4 { __uint_as_float(v[12]), __uint_as_float(v[13]), __uint_as_float(v[14]), __uint_as_float(v[15]) };
    return r;
}

// Ray/box test of reference src/intersect.cpp:111-140 without the final comparison against ray.t: returns
// whether the box is entered at all and the parameter the reference would report (t_out when the origin is
// inside).  The caller compares against its own bound.
//
// The reference picks the near and far plane of each slab by the sign of the direction component.  Because
// lo <= hi and multiplication by the (same-signed) reciprocal is monotonic, the near value is the smaller
// and the far value the larger of the two products, so min/max select the same numbers; none of them can
// be NaN (the |d| <= 1e-6 case is overridden with -/+FLT_MAX exactly as in the reference), so the hardware
// min/max agree with glm's comparison-based ones.
__device__ __forceinline__ bool box_entry(const RayD& r, float lx, float ly, float lz, float hx, float hy, float hz, float& tBox)
{
    const float ax = (lx - r.o.x) * r.inv.x, bx = (hx - r.o.x) * r.inv.x;
    const float ay = (ly - r.o.y) * r.inv.y, by = (hy - r.o.y) * r.inv.y;
    const float az = (lz - r.o.z) * r.inv.z, bz = (hz - r.o.z) * r.inv.z;
    const float t0 = r.zx ? -FLT_MAX : fminf(ax, bx), t1 = r.zx ? FLT_MAX : fmaxf(ax, bx);
    const float t2 = r.zy ? -FLT_MAX : fminf(ay, by), t3 = r.zy ? FLT_MAX : fmaxf(ay, by);
    const float t4 = r.zz ? -FLT_MAX : fminf(az, bz), t5 = r.zz ? FLT_MAX : fmaxf(az, bz);
    const float tin = fmaxf(fmaxf(t0, t2), t4);
    const float tout = fminf(fminf(t1, t3), t5);
    if (tin > tout || tout <= 0 || is_zero(tin)) return false;
    tBox = tin < 0 ? tout : tin;
    return true;
}
__device__ __forceinline__ bool box_entry(const RayD& r, const float* lo, const float* hi, float& tBox)
{
    return box_entry(r, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], tBox);
}
// the two child boxes of a node record loaded with load_rec (field order of NodeRec)
// (a child box is stored lo.x hi.x lo.y hi.y lo.z hi.z)
__device__ __forceinline__ bool box_entry_left(const RayD& r, const Rec64& n, float& t) { return box_entry(r, n.a.x, n.a.z, n.b.x, n.a.y, n.a.w, n.b.y, t); }
__device__ __forceinline__ bool box_entry_right(const RayD& r, const Rec64& n, float& t) { return box_entry(r, n.b.z, n.c.x, n.c.z, n.b.w, n.c.y, n.c.w, t); }
// Both child boxes of a node record for a ray of the FAST CLASS in a scene whose root box (hence every box) lies within
// +-FAST_BOUND: every slab product (plane - o) * (1 / d) is then finite and below 2e30 * 1e6 in magnitude (|d| > 1e-6
// bounds the reciprocal), so none is a NaN, no zero-class override applies, and the parameter box_entry() reports is
// always below FLT_MAX (the comparison its callers append is always true).  A slab's two planes are adjacent in the
// record: one packed subtract and one packed multiply with the ray's scalars broadcast to both halves.
__device__ __forceinline__ void box_pair_fast(const RayD& r, const Rec64& n, bool& vL, float& tL, bool& vR, float& tR)
{
    const f32x2 oX { r.o.x, r.o.x }, oY { r.o.y, r.o.y }, oZ { r.o.z, r.o.z };
    const f32x2 iX { r.inv.x, r.inv.x }, iY { r.inv.y, r.inv.y }, iZ { r.inv.z, r.inv.z };
    const f32x2 p0 = (n.a.xy - oX) * iX, p1 = (n.a.zw - oY) * iY, p2 = (n.b.xy - oZ) * iZ; // left box: x, y, z slab
    const f32x2 p3 = (n.b.zw - oX) * iX, p4 = (n.c.xy - oY) * iY, p5 = (n.c.zw - oZ) * iZ; // right box
    const float tinL = fmaxf(fmaxf(fminf(p0.x, p0.y), fminf(p1.x, p1.y)), fminf(p2.x, p2.y));
    const float toutL = fminf(fminf(fmaxf(p0.x, p0.y), fmaxf(p1.x, p1.y)), fmaxf(p2.x, p2.y));
    const float tinR = fmaxf(fmaxf(fminf(p3.x, p3.y), fminf(p4.x, p4.y)), fminf(p5.x, p5.y));
    const float toutR = fminf(fminf(fmaxf(p3.x, p3.y), fmaxf(p4.x, p4.y)), fmaxf(p5.x, p5.y));
    vL = !(tinL > toutL) && !(toutL <= 0) && !is_zero(tinL);
    vR = !(tinR > toutR) && !(toutR <= 0) && !is_zero(tinR);
    tL = tinL < 0 ? toutL : tinL;
    tR = tinR < 0 ? toutR : tinR;
}
__device__ __forceinline__ int32_t node_ref_left(const Rec64& n) { return __float_as_int(n.d.x); }
__device__ __forceinline__ int32_t node_ref_right(const Rec64& n) { return __float_as_int(n.d.y); }

// Ray/triangle test of reference src/intersect.cpp:22-80 on a precomputed record.  `tmax` is the current
// closest distance; returns true and the new distance when the triangle is accepted.  The rejections are
// evaluated cheapest-first; each is a pure predicate, so the order does not change the outcome.
__device__ __forceinline__ bool triangle_hit(const RayD& r, const Rec64& p, float tmax, float& tHit)
{
    // field order of PrimRec: v0, v1, v2, n, invArea, more, mesh, prim
    const vec3 v0 = make3(p.a.x, p.a.y, p.a.z), v1 = make3(p.a.w, p.b.x, p.b.y), v2 = make3(p.b.z, p.b.w, p.c.x);
    const vec3 n = make3(p.c.y, p.c.z, p.c.w);
    const float invArea = p.d.x;
    const float dn = dot(n, r.d);
    if (is_zero(dn)) return false;
    const float D = dot(n, v0);
    const float t = (D - dot(r.o, n)) / dn;
    if (!(t > 0)) return false;
    if (!(t < tmax)) return false;
    const vec3 pt = r.o + r.d * t;
    if (!is_zero(dot(n, pt - v0))) return false;
    const float alpha = length(cross(pt - v2, v1 - v2)) * invArea;
    const float beta = length(cross(pt - v2, v0 - v2)) * invArea;
    const float gamma = length(cross(pt - v1, v0 - v1)) * invArea;
    if (alpha < 0 || beta < 0 || alpha + beta > 1.0f || alpha + gamma > 1.0f || beta + gamma > 1.0f) return false;
    tHit = t;
    return true;
}
__device__ __forceinline__ int32_t prim_more(const Rec64& p) { return __float_as_int(p.d.y); }
__device__ __forceinline__ int32_t prim_mesh(const Rec64& p) { return __float_as_int(p.d.z); }

// Ray/sphere test of reference src/intersect.cpp:84-107 on a PrimRec (v0 = centre, v1[0] = radius).  The
// reference's unqualified pow() and sqrt() are the C double functions, so the radius term and the two roots
// are evaluated in double and narrowed on assignment; reproduced here.  `wrote` reports that the reference
// would have overwritten HitInfo::normal / material (it does so even when the sphere is not the closest hit).
__device__ __forceinline__ bool sphere_hit(const RayD& r, const Rec64& p, float& tmax, bool& wrote)
{
    wrote = false;
    const vec3 c = make3(p.a.x, p.a.y, p.a.z);
    const float radius = p.a.w;
    const vec3 oc = r.o - c;
    const float a = dot(r.d, r.d);
    const float b = 2.0f * dot(r.d, oc);
    const float cc = float(double(dot(oc, oc)) - double(radius) * double(radius));
    const float D = b * b - 4 * a * cc;
    if (D < 0 || is_zero(cc)) return false;
    const double root = sqrt(double(D));
    const float t1 = float((double(-b) - root) / double(2 * a));
    const float t2 = float((double(-b) + root) / double(2 * a));
    if (t1 < 0 && t2 < 0) return false;
    const float newT = t1 * t2 < 0 ? fmaxf(t1, t2) : fminf(t1, t2);
    const bool valid = newT < tmax;
    if (valid) tmax = newT;
    wrote = true;
    return valid;
}

// Which primitive last wrote HitInfo's normal/material and the ray distance at that moment.  Without spheres
// this is simply the closest hit; a sphere test rewrites them even when it loses (see sphere_hit).
struct Attr {
    int slot;
    float t;
};

// LDS traversal stack of one lane.  Entry e of lane l sits at base[e * blockDim + l].
struct Stack {
    uint2* base;
    uint32_t stride;
    __device__ __forceinline__ void push(int sp, int32_t ref, float t) { base[uint32_t(sp) * stride] = make_uint2(uint32_t(ref), __float_as_uint(t)); }
    __device__ __forceinline__ void pop(int sp, int32_t& ref, float& t) const
    {
        uint2 e = base[uint32_t(sp) * stride];
        ref = int32_t(e.x);
        t = __uint_as_float(e.y);
    }
};

// Closest-hit (ANY = false) or any-hit (ANY = true) query, result-equivalent to the reference's recursive
// traversal (reference src/bounding_volume_hierarchy.cpp:293-449, 529-547):
//   - the whole query is rejected unless the root box is entered before the caller's bound `tmax`;
//   - with PRUNE (neither recursive nor transparency enabled) a node is skipped when its box parameter
//     exceeds the closest triangle distance found so far; without PRUNE every entered box is visited;
//   - both children are tested against an unbounded ray; the one with the smaller box parameter is visited
//     first, the right one on a tie;
//   - a triangle is accepted when strictly closer than the current distance, so ties keep the first found.
// Returns the slot of the accepted primitive (or -1) and updates tmax.  An any-hit query returns at the
// first accepted triangle: the reference's boolean is the same because acceptance is what sets it.
//
// BUDGET: a query that has visited `budget` inner nodes without accepting anything, or that needs more than `stackLimit`
// stack entries, gives up and returns TRAVERSE_GAVE_UP; the caller hands it to kernels that finish such rays (the
// first kind skims along the surface through thousands of boxes; the second kind is rare, and a short stack is what
// lets the kernel keep twice the waves resident).
constexpr int TRAVERSE_GAVE_UP = -2;
template <bool PRUNE, bool ANY, bool COUNT, bool SPH, bool BUDGET = false>
__device__ int traverse(const SceneView& S, const RayD& r, float& tmax, Stack st, Counts& cnt, Attr& attr, int budget = 0, int stackLimit = 0)
{
    attr.slot = -1;
    attr.t = tmax;
    float bestTri = FLT_MAX; // the reference's absoluteT: only triangle hits lower it (…hierarchy.cpp:344-347)
    if (S.numPrims == 0) return -1;
    float tRoot;
    if (COUNT) cnt.box++;
    if (!box_entry(r, S.rootLo, S.rootHi, tRoot) || !(tRoot < tmax)) return -1;
    constexpr int32_t DONE = INT32_MIN;
    int hit = -1;
    int sp = 0;
    int32_t cur = S.rootRef;
    float curT = tRoot;
    // the lanes that traverse together take the packed box test when all their rays are of the fast class
    const bool fast = S.bounded && __builtin_amdgcn_ballot_w64(!r.fast) == 0;
    // "while-while": all lanes of the wave walk inner nodes until each holds a leaf (or has run out), then
    // all of them run their primitive tests together.  The per-lane visiting order is unchanged.
    while (true) {
        while (cur >= 0) {
            bool descend = true;
            if (PRUNE) descend = !(curT > (SPH ? bestTri : (hit >= 0 ? tmax : FLT_MAX)));
            if (descend) {
                if (BUDGET && hit < 0 && --budget < 0) return TRAVERSE_GAVE_UP;
                const Rec64 nd = load_rec(&S.nodes[cur]);
                float tL, tR;
                if (COUNT) cnt.box += 2;
                bool vL, vR;
                if (fast) {
                    box_pair_fast(r, nd, vL, tL, vR, tR);
                } else {
                    vL = box_entry_left(r, nd, tL) && tL < FLT_MAX;
                    vR = box_entry_right(r, nd, tR) && tR < FLT_MAX;
                }
                const int32_t refL = node_ref_left(nd), refR = node_ref_right(nd);
                if (vL && vR) {
                    const bool leftFirst = tL < tR;
#ifdef RT_SP_HIST
                    if (BUDGET && sp + 1 > int(cnt.overflow)) cnt.overflow = uint32_t(sp + 1);
#endif
                    // (a budgeted query runs on a short stack: 99.98 % of camera rays never hold more than ten entries)
                    if (BUDGET && sp >= stackLimit) return TRAVERSE_GAVE_UP;
                    st.push(sp++, leftFirst ? refR : refL, leftFirst ? tR : tL);
                    cur = leftFirst ? refL : refR;
                    curT = leftFirst ? tL : tR;
                    continue;
                }
                if (vL || vR) {
                    cur = vR ? refR : refL;
                    curT = vR ? tR : tL;
                    continue;
                }
            }
            if (sp == 0) cur = DONE;
            else st.pop(--sp, cur, curT);
        }
        if (cur == DONE) break;
        bool test = true;
        if (PRUNE) test = !(curT > (SPH ? bestTri : (hit >= 0 ? tmax : FLT_MAX)));
        if (test) {
            int32_t slot = ~cur;
            int32_t more;
            do {
                const Rec64 p = load_rec(&S.prims[slot]);
                more = prim_more(p);
                if (COUNT) cnt.tri++;
                if (SPH && prim_mesh(p) < 0) {
                    bool wrote;
                    if (sphere_hit(r, p, tmax, wrote)) {
                        hit = slot;
                        if (ANY) return hit;
                    }
                    if (wrote) {
                        attr.slot = slot;
                        attr.t = tmax;
                    }
                } else {
                    float t;
                    if (triangle_hit(r, p, tmax, t)) {
                        tmax = t;
                        bestTri = t;
                        hit = slot;
                        attr.slot = slot;
                        attr.t = t;
                        if (ANY) return hit;
                    }
                }
                slot++;
            } while (more > 0);
        }
        if (sp == 0) break;
        st.pop(--sp, cur, curT);
    }
    return hit;
}

// Brute-force branch (reference src/bounding_volume_hierarchy.cpp:462-528): every triangle in mesh order.
template <bool ANY, bool COUNT, bool SPH>
__device__ int brute_force(const SceneView& S, const RayD& r, float& tmax, Counts& cnt, Attr& attr)
{
    int hit = -1;
    attr.slot = -1;
    attr.t = tmax;
    for (int i = 0; i < S.numPrims; i++) {
        const int slot = int(S.primOrder[i]);
        const Rec64 p = load_rec(&S.prims[slot]);
        if (COUNT) cnt.tri++;
        if (SPH && prim_mesh(p) < 0) {
            bool wrote;
            if (sphere_hit(r, p, tmax, wrote)) {
                hit = slot;
                if (ANY) return hit;
            }
            if (wrote) {
                attr.slot = slot;
                attr.t = tmax;
            }
        } else {
            float t;
            if (triangle_hit(r, p, tmax, t)) {
                tmax = t;
                hit = slot;
                attr.slot = slot;
                attr.t = t;
                if (ANY) return hit;
            }
        }
    }
    return hit;
}

template <bool PRUNE, bool ANY, bool COUNT, bool SPH = false>
__device__ __forceinline__ int intersect(const FrameParams& P, const RayD& r, float& tmax, Stack st, Counts& cnt, Attr& attr)
{
    if (!P.f.accel_structure) return brute_force<ANY, COUNT, SPH>(P.scene, r, tmax, cnt, attr);
    return traverse<PRUNE, ANY, COUNT, SPH>(P.scene, r, tmax, st, cnt, attr);
}
template <bool PRUNE, bool ANY, bool COUNT, bool SPH = false>
__device__ __forceinline__ int intersect(const FrameParams& P, const RayD& r, float& tmax, Stack st, Counts& cnt)
{
    Attr attr;
    return intersect<PRUNE, ANY, COUNT, SPH>(P, r, tmax, st, cnt, attr);
}

// ---------------------------------------------------------------------------------------------------
// Hit attributes
// ---------------------------------------------------------------------------------------------------
struct Surface {
    vec3 normal;   // HitInfo::normal
    vec3 bary;     // HitInfo::barycentricCoord
    vec3 kd, ks;
    float shininess, transparency;
};

__device__ __forceinline__ vec3 texel_u8(const uint8_t* texels, const ImageRec& im, int idx)
{
    const int n = im.width * im.height;
    idx = idx < 0 ? 0 : (idx >= n ? n - 1 : idx); // the reference reads out of bounds here; clamp instead
    const uint8_t* px = texels + im.offset + size_t(idx) * 3;
    return make3(float(px[0]) / 255.0f, float(px[1]) / 255.0f, float(px[2]) / 255.0f);
}

// std::round(double(x) - 0.5) for a float x, evaluated without doubles: x - 0.5 is exact in double, and
// rounding halves away from zero gives floor(x) from 0.5 upwards, 0 inside (0, 0.5), and ceil(x - 1) at and
// below zero (where the float subtraction can only round when the result is unaffected).
__device__ __forceinline__ float round_minus_half(float x)
{
    return x >= 0.5f ? floorf(x) : (x > 0.0f ? 0.0f : ceilf(x - 1.0f));
}

// reference src/texture.cpp:5-17: nearest texel, wrapped with fmod.  The reference's intermediates are
// double; every one of them is an integer-valued or exactly representable quantity, so the float
// evaluation below produces the same index.
__device__ __forceinline__ vec3 acquire_texel(const uint8_t* texels, const ImageRec& im, float u, float v)
{
    const float i = fmodf(round_minus_half(u * float(im.width)), float(im.width));
    const float j = fmodf(round_minus_half((1 - v) * float(im.height)), float(im.height));
    return texel_u8(texels, im, int(j * float(im.width) + i));
}

// Bilinear lookup.  The reference declares the switch (src/common.h:59) but its acquireTexel ignores it
// (src/texture.cpp:5-17): there is no reference behaviour to match.  Defined as the natural extension of the nearest
// lookup: same (1 - v) flip, texel centres at +0.5, repeat wrap, fp32 throughout, blended as
// (1-fy)*((1-fx)*T00 + fx*T10) + fy*((1-fx)*T01 + fx*T11).
__device__ __forceinline__ int wrap_texel(float c, int n)
{
    float m = fmodf(c, float(n));
    if (m < 0) m += float(n);
    const int i = int(m);
    return i >= n ? n - 1 : i; // m + n can round up to n for a tiny negative m
}
__device__ __forceinline__ vec3 acquire_texel_bilinear(const uint8_t* texels, const ImageRec& im, float u, float v)
{
    const float x = u * float(im.width) - 0.5f, y = (1 - v) * float(im.height) - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int i0 = wrap_texel(x0, im.width), i1 = wrap_texel(x0 + 1.0f, im.width);
    const int j0 = wrap_texel(y0, im.height), j1 = wrap_texel(y0 + 1.0f, im.height);
    const vec3 t00 = texel_u8(texels, im, j0 * im.width + i0), t10 = texel_u8(texels, im, j0 * im.width + i1);
    const vec3 t01 = texel_u8(texels, im, j1 * im.width + i0), t11 = texel_u8(texels, im, j1 * im.width + i1);
    const vec3 top = t00 * (1 - fx) + t10 * fx, bottom = t01 * (1 - fx) + t11 * fx;
    return top * (1 - fy) + bottom * fy;
}

// reference src/interpolate.cpp:4-11
__device__ __forceinline__ vec3 barycentric(vec3 v0, vec3 v1, vec3 v2, vec3 p)
{
    const float totalArea = length(cross(v0 - v2, v1 - v2)) / 2;
    const float alpha = length(cross(p - v2, v1 - v2)) / (2 * totalArea);
    const float beta = length(cross(p - v2, v0 - v2)) / (2 * totalArea);
    const float gamma = length(cross(p - v1, v0 - v1)) / (2 * totalArea);
    return make3(alpha, beta, gamma);
}

// What the reference leaves in HitInfo after an accepted triangle (intersect.cpp:74-78 and
// bounding_volume_hierarchy.cpp:351-396): geometric normal, barycentrics, material, texel, and the
// interpolated normal.  Only the final hit's values survive, so they are computed once, afterwards.
__device__ __forceinline__ Surface surface_at(const FrameParams& P, const RayD& r, float t, int slot)
{
    const SceneView& S = P.scene;
    const PrimRec& p = S.prims[slot];
    Surface s;
    if (p.mesh < 0) {
        // sphere: HitInfo::normal = point - centre, unnormalised (intersect.cpp:103); `t` is the ray distance
        // at the moment the sphere test wrote it
        const MaterialRec& m = S.mats[S.numMeshes + p.prim];
        s.normal = -load3(p.v0) + (r.d * t + r.o);
        s.bary = splat3(0.0f);
        s.kd = load3(m.kd);
        s.ks = load3(m.ks);
        s.shininess = m.shininess;
        s.transparency = m.transparency;
        return s;
    }
    const vec3 v0 = load3(p.v0), v1 = load3(p.v1), v2 = load3(p.v2);
    const vec3 point = r.o + r.d * t;
    s.normal = load3(p.n);
    s.bary = barycentric(v0, v1, v2, point);
    const MaterialRec& m = S.mats[p.mesh];
    s.kd = load3(m.kd);
    s.ks = load3(m.ks);
    s.shininess = m.shininess;
    s.transparency = m.transparency;
    if ((P.f.texture_mapping && m.texture >= 0) || P.f.normal_interp) {
        const AttrRec& a = S.attrs[slot];
        if (P.f.texture_mapping && m.texture >= 0) {
            const float tu = a.uv0[0] * s.bary.x + a.uv1[0] * s.bary.y + a.uv2[0] * s.bary.z; // interpolate.cpp:20-28
            const float tv = a.uv0[1] * s.bary.x + a.uv1[1] * s.bary.y + a.uv2[1] * s.bary.z;
            s.kd = P.f.bilinear_filtering ? acquire_texel_bilinear(S.texels, S.textures[m.texture], tu, tv)
                                          : acquire_texel(S.texels, S.textures[m.texture], tu, tv);
        }
        if (P.f.normal_interp) // interpolate.cpp:13-18
            s.normal = normalize(s.bary.x * load3(a.n0) + s.bary.y * load3(a.n1) + s.bary.z * load3(a.n2));
    }
    return s;
}

// ---------------------------------------------------------------------------------------------------
// Cube map, reference src/environment_mapping.cpp:9-49
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ vec3 skybox_color(const SceneView& S, vec3 dir)
{
    if (!S.hasSkybox) return splat3(0.0f);
    const vec3 v = normalize(dir);
    const float ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z);
    float maxA, ut, vt;
    int face;
    if (ax >= ay && ax >= az) {
        maxA = 2 * ax;
        ut = (v.x >= 0) ? -v.z : v.z;
        vt = -v.y;
        face = 1 - int(v.x < 0); // the x faces are swapped
    } else if (ay >= az) {
        maxA = 2 * ay;
        ut = v.x;
        vt = (v.y >= 0) ? v.z : -v.z;
        face = int(v.y < 0) + 2;
    } else {
        maxA = 2 * az;
        ut = (v.z >= 0) ? v.x : -v.x;
        vt = -v.y;
        face = int(v.z < 0) + 4;
    }
    // the reference adds the double literal 0.5 and narrows; for these magnitudes the float sum is identical
    const float tu = -(ut / maxA) + 0.5f;
    const float tv = -(vt / maxA) + 0.5f;
    const ImageRec& im = S.skybox[face];
    const int i = int(tu * float(im.width));
    float fj = (1.0f - tv) * float(im.height);
    const float hiJ = float(im.height) - 1.0f;
    fj = fj < 0.0f ? 0.0f : (hiJ < fj ? hiJ : fj); // std::clamp
    const int j = int(fj);
    return texel_u8(S.texels, im, j * im.width + i);
}

// ---------------------------------------------------------------------------------------------------
// Lights and shading
// ---------------------------------------------------------------------------------------------------
struct Rng {
    uint32_t pixelKey, sample, randCounter, glossCall;
    __device__ __forceinline__ int rand31() { return int(rng_u32(pixelKey, sample, RNG_STREAM_RAND, randCounter++) >> 1); }
    __device__ __forceinline__ double gauss(uint32_t idx) const
    {
        const uint32_t a = rng_u32(pixelKey, sample, RNG_STREAM_GLOSS, 2u * idx);
        const uint32_t b = rng_u32(pixelKey, sample, RNG_STREAM_GLOSS, 2u * idx + 1u);
        return sqrt(-2.0 * log(rng_unit(a))) * cos(6.283185307179586476925 * rng_unit(b));
    }
};

// Phong term, reference src/shading.cpp:6-39.  The reference's exponentiation resolves to C's double pow and is
// narrowed to float; powf here differs from that by at most a few units in the last place of a factor that is
// at most 1, i.e. below 1e-6 in the pixel value, and no ray decision depends on it.
__device__ __forceinline__ vec3 phong(vec3 lightPos, vec3 lightColor, const RayD& r, float t, const Surface& s)
{
    const vec3 P = r.o + (r.d * t);
    const vec3 N = normalize(s.normal);
    const vec3 L = normalize(lightPos - P);
    const float nl = dot(N, L);
    const float lambertian = nl > 0 ? nl : 0;
    const vec3 diffuse = lightColor * s.kd * lambertian;
    float specular = 0.0f;
    if (lambertian > 0) {
        const vec3 R = reflect(L, N);
        const vec3 V = normalize(-r.o + P);
        const float rv = dot(R, V);
        const float specAngle = rv > 0 ? rv : 0;
        specular = powf(specAngle, s.shininess);
    }
    return lightColor * s.ks * specular + diffuse;
}

// reference src/light.cpp:68-89
template <bool PRUNE, bool COUNT, bool SPH = false>
__device__ __forceinline__ float visibility(const FrameParams& P, vec3 samplePos, const RayD& r, float t, Stack st, Counts& cnt)
{
    const float epsilon = 0.00001f;
    const vec3 origin = r.o + (1 - epsilon) * t * r.d;
    const RayD shadow = make_ray(origin, samplePos - origin);
    float tmax = 1.0f;
    cnt.shadow++;
    return intersect<PRUNE, true, COUNT, SPH>(P, shadow, tmax, st, cnt) >= 0 ? 0.0f : 1.0f;
}

// Sample k of light L (reference src/light.cpp:22-64 with the loop bookkeeping of :160-167,195-201): position
// and colour.  `draw` is the index on the rand() stream of the first draw this sample consumes (a segment
// sample takes one draw, a parallelogram sample two, a point light none).
__device__ __forceinline__ void light_sample(const LightDev& L, int k, uint32_t pixelKey, uint32_t sample, uint32_t draw, vec3& position, vec3& color)
{
    if (L.type == RT_LIGHT_POINT) {
        position = load3(L.v);
        color = load3(L.v + 3);
    } else if (L.type == RT_LIGHT_SEGMENT) {
        const vec3 e0 = load3(L.v), e1 = load3(L.v + 3), c0 = load3(L.v + 6), c1 = load3(L.v + 9);
        const float randomUniform = float(int(rng_u32(pixelKey, sample, RNG_STREAM_RAND, draw) >> 1) % 50) / 50.0f;
        const vec3 randomAddition = load3(L.step) * randomUniform;
        const float alpha = float(k) * L.recA;
        position = (e0 * (1 - alpha) + e1 * alpha) + randomAddition;
        float p0 = 0.5f, p1 = 0.5f;
        if (L.recB != 0) {
            p1 = length(position - e0) * L.recB;
            p0 = 1 - p1;
        }
        color = p0 * c0 + p1 * c1;
    } else {
        const int i = k / L.nB, j = k - i * L.nB;
        const vec3 v0 = load3(L.v), e01 = load3(L.v + 3), e02 = load3(L.v + 6);
        const vec3 c0 = load3(L.v + 9), c1 = load3(L.v + 12), c2 = load3(L.v + 15), c3 = load3(L.v + 18);
        const float ri = float(int(rng_u32(pixelKey, sample, RNG_STREAM_RAND, draw) >> 1) % 50) / 50.0f + float(i);
        const float rj = float(int(rng_u32(pixelKey, sample, RNG_STREAM_RAND, draw + 1u) >> 1) % 50) / 50.0f + float(j);
        const float x = ri * L.recA, y = rj * L.recB;
        position = (v0 + e01 * x) + e02 * y;
        color = (1 - y) * (x * c1 + (1 - x) * c0) + y * (x * c3 + (1 - x) * c2);
    }
}
__device__ __forceinline__ uint32_t light_draws_per_sample(int type) { return type == RT_LIGHT_POINT ? 0u : (type == RT_LIGHT_SEGMENT ? 1u : 2u); }
__device__ __forceinline__ bool light_enabled(const FrameParams& P, int type) { return type == RT_LIGHT_POINT || P.f.soft_shadow || P.f.shading; }
__device__ __forceinline__ bool light_shadowed(const FrameParams& P, int type) { return type == RT_LIGHT_POINT ? P.f.hard_shadow != 0 : P.f.soft_shadow != 0; }
// weight the reference applies to one sample's shade (light.cpp:141-143,175-178,209-212)
__device__ __forceinline__ vec3 light_weight(const LightDev& L, vec3 shade)
{
    if (L.type == RT_LIGHT_SEGMENT) return shade * L.recA;
    if (L.type == RT_LIGHT_PARALLELOGRAM) return shade * L.recA * L.recB;
    return shade;
}

// reference src/light.cpp:124-220.  All light types run through one sample loop so the kernel holds a single
// copy of the shadow-ray traversal and of the Phong term.
template <bool PRUNE, bool COUNT, bool SPH = false>
__device__ vec3 light_contribution(const FrameParams& P, const RayD& r, float t, const Surface& s, Rng& rng, Stack st, Counts& cnt)
{
    vec3 shading = splat3(0.0f);
    for (int li = 0; li < P.nLights; li++) {
        const LightDev& L = light_at(P, li);
        if (!light_enabled(P, L.type)) continue;
        const bool shadowed = light_shadowed(P, L.type);
        const int total = L.nA * L.nB;
        const uint32_t per = light_draws_per_sample(L.type);
        for (int k = 0; k < total; k++) {
            vec3 position, color;
            light_sample(L, k, rng.pixelKey, rng.sample, rng.randCounter, position, color);
            rng.randCounter += per;
            const float lighted = shadowed ? visibility<PRUNE, COUNT, SPH>(P, position, r, t, st, cnt) : 1.0f;
            const vec3 shade = P.f.shading ? phong(position, color, r, t, s) : s.kd;
            shading = shading + light_weight(L, shade) * lighted;
        }
    }
    return shading;
}

// ---------------------------------------------------------------------------------------------------
// Colour of one ray: the reference's recursion (src/render.cpp:18-80) unrolled into a loop with an
// explicit frame stack, so results combine in exactly the order the recursion would.
//   MIRROR frame:  result = ks * child                              (render.cpp:61)
//   BLEND  frame:  result = a * Lo + (1 - a) * child                (render.cpp:43)
//   GLOSSY frame:  result = ks * (sum of children) / filterSize     (render.cpp:53-59), children generated
//                  one after the other by perturbing the reflection ray (src/gloss.cpp:14-51)
// ---------------------------------------------------------------------------------------------------
enum { FRAME_MIRROR = 0, FRAME_BLEND = 1, FRAME_GLOSSY = 2 };
enum { MAX_SPECULAR_DEPTH = 8, MAX_BLEND_DEPTH = 24, MAX_FRAMES = MAX_SPECULAR_DEPTH + MAX_BLEND_DEPTH };

struct Frame {
    float a;  // BLEND: transparency
    vec3 c;   // MIRROR/GLOSSY: ks; BLEND: Lo
    int32_t kind;
};

struct GlossyFrame {
    vec3 origin, dir, u, v, normal, sum;
    float side, offset;
    int32_t next, call, depth;
};

template <bool PRUNE, bool GLOSSY, bool COUNT, bool SPH>
__device__ vec3 final_color(const FrameParams& P, vec3 origin, vec3 direction, Rng& rng, Stack st, Counts& cnt, float& firstT)
{
    bool first = true;
    Frame frames[MAX_FRAMES];
    GlossyFrame gframes[GLOSSY ? MAX_SPECULAR_DEPTH : 1];
    int nFrames = 0, nGlossy = 0, blendDepth = 0;
    int depth = P.p.ray_depth; // bounce budget of the ray currently being traced / whose result is held
    vec3 result = splat3(0.0f);
    RayD r = make_ray(origin, direction);
    enum { TRACE, SPAWN, RETURN } state = TRACE;
    while (true) {
        if (state == TRACE) {
            float t = FLT_MAX;
            Attr attr;
            const int slot = intersect<PRUNE, false, COUNT, SPH>(P, r, t, st, cnt, attr);
            state = RETURN;
            if (first) {
                firstT = t;
                first = false;
            }
            if (slot < 0) {
                result = (P.f.environment_mapping && P.p.enabled_skybox) ? skybox_color(P.scene, r.d) : splat3(0.0f);
                continue;
            }
            const Surface s = surface_at(P, r, attr.t, attr.slot);
            const bool reflMap = P.f.environment_mapping && P.p.enabled_refl_map;
            const bool blend = !reflMap && P.f.transparency && s.transparency != 1.0f;
            const bool specular = !reflMap && !blend && (P.f.recursive || P.f.glossy) && depth > 0 && (s.ks.x != 0 || s.ks.y != 0 || s.ks.z != 0);
            // The reference evaluates the direct lighting at every hit and then drops it on the mirror /
            // glossy / reflection-map branches (render.cpp:25,27-34,48-63).  Those branches skip the work
            // here but still consume the same random draws, so later draws keep their indices.
            vec3 Lo = splat3(0.0f);
            if (reflMap || specular) rng.randCounter += uint32_t(P.drawsPerHit);
            else Lo = light_contribution<PRUNE, COUNT, SPH>(P, r, t, s, rng, st, cnt);
            if (reflMap) {
                result = skybox_color(P.scene, r.d);
            } else if (blend) {
                if (blendDepth >= P.p.max_transparency_depth) {
                    cnt.overflow++;
                    result = Lo;
                } else {
                    Frame& f = frames[nFrames++];
                    f.kind = FRAME_BLEND;
                    f.a = s.transparency;
                    f.c = Lo;
                    blendDepth++;
                    cnt.bounce++;
                    r = make_ray((t + 0.000001f) * r.d + r.o, r.d);
                    state = TRACE;
                }
            } else if (specular) {
                // computeReflectionRay, reference src/shading.cpp:42-51
                const float epsilon = 0.00001f;
                const vec3 rd = r.d - (2 * dot(r.d, s.normal)) * s.normal;
                const vec3 ro = t * r.d * (1 - epsilon) + r.o;
                Frame& f = frames[nFrames++];
                f.c = s.ks;
                f.a = 0.0f;
                if (GLOSSY && P.f.glossy) {
                    // basis and footprint of getRaySamples, reference src/gloss.cpp:16-43
                    f.kind = FRAME_GLOSSY;
                    GlossyFrame& g = gframes[nGlossy++];
                    const vec3 w = normalize(rd);
                    const float awx = fabsf(w.x), awy = fabsf(w.y), awz = fabsf(w.z);
                    const int minTerm = (awx <= awy && awx <= awz) ? 0 : ((awy <= awx && awy <= awz) ? 1 : 2);
                    vec3 tt = w;
                    if (minTerm == 0) tt.x = 1; else if (minTerm == 1) tt.y = 1; else tt.z = 1;
                    g.u = normalize(cross(tt, w));
                    g.v = cross(w, g.u);
                    g.side = P.p.glossy_sigma / (s.shininess * 3.0f);
                    g.offset = -g.side / 2.0f;
                    g.origin = ro;
                    g.dir = rd;
                    g.normal = s.normal;
                    g.sum = splat3(0.0f);
                    g.next = 0;
                    g.call = int32_t(rng_ray_key(ro, rd));
                    g.depth = depth;
                    state = SPAWN;
                } else {
                    f.kind = FRAME_MIRROR;
                    depth--;
                    cnt.bounce++;
                    r = make_ray(ro, rd);
                    state = TRACE;
                }
            } else {
                result = Lo;
            }
        } else if (GLOSSY && state == SPAWN) {
            // next surviving sample of the glossy frame on top (reference src/gloss.cpp:44-49), or its total
            GlossyFrame& g = gframes[nGlossy - 1];
            state = RETURN;
            while (g.next < P.p.glossy_filter_size) {
                const uint32_t k = uint32_t(g.call) + 2u * uint32_t(g.next);
                g.next++;
                const double n0 = rng.gauss(k), n1 = rng.gauss(k + 1u);
                const vec3 rp = (g.dir + float(double(g.offset) + n0 * double(g.side)) * g.u) + float(double(g.offset) + n1 * double(g.side)) * g.v;
                if (dot(rp, g.normal) > 0) {
                    depth = g.depth - 1;
                    cnt.bounce++;
                    r = make_ray(g.origin, normalize(rp));
                    state = TRACE;
                    break;
                }
            }
            if (state == RETURN) {
                const Frame& f = frames[--nFrames];
                result = f.c * g.sum / float(P.p.glossy_filter_size);
                depth = g.depth;
                nGlossy--;
            }
        } else {
            // RETURN: hand `result` to the innermost pending frame
            if (nFrames == 0) return result;
            const Frame& f = frames[nFrames - 1];
            if (f.kind == FRAME_MIRROR) {
                result = f.c * result;
                depth++;
                nFrames--;
            } else if (f.kind == FRAME_BLEND) {
                result = f.a * f.c + (1.0f - f.a) * result;
                blendDepth--;
                nFrames--;
            } else if (GLOSSY) {
                GlossyFrame& g = gframes[nGlossy - 1];
                g.sum = g.sum + result;
                state = SPAWN;
            } else {
                nFrames--;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// One pixel: reference src/render.cpp:93-122 with calculateColor (src/multipleRays.cpp:15-54) and
// getEyeFrame (src/dof.cpp:21-51) inlined.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void camera_ray(const FrameParams& P, float px, float py, vec3& o, vec3& d)
{
    // reference framework/src/trackball.cpp:101-110
    const vec3 c = normalize(make3(-px * P.halfW, py * P.halfH, 1.0f));
    o = load3(P.camPos);
    d = rotate(P.camQ, c);
}

// Lens ray k of a pixel whose centre ray is (o, d): getEyeFrame, reference src/dof.cpp:21-51.
__device__ __forceinline__ void lens_ray(const FrameParams& P, uint32_t pixelKey, int k, vec3& o, vec3& d)
{
    const vec3 w = normalize(d);
    const vec3 up = load3(P.camUp), left = load3(P.camLeft);
    const float side = P.p.dof_f / (P.p.dof_aperture * 6.0f);
    const float offset = -side / 2.0f;
    const vec3 focus = o + w * P.p.dof_f;
    const double u0 = rng_unit(rng_u32(pixelKey, 0, RNG_STREAM_DOF, uint32_t(2 * k)));
    const double u1 = rng_unit(rng_u32(pixelKey, 0, RNG_STREAM_DOF, uint32_t(2 * k + 1)));
    const vec3 eye = (o + float(double(offset) + u0 * double(side)) * up) + float(double(offset) + u1 * double(side)) * left;
    d = normalize(focus - eye);
    o = eye;
}

// Colour of sample `sidx` of pixel (x, y):
//   supersampling on  -> sample (i, j) = (sidx / n, sidx % n) of calculateColor's n x n loop;
//   depth of field on -> sample 0 is the centre ray (whose first hit distance doubles as the reference's
//                        probe query, render.cpp:106-110: same ray, same traversal), samples 1.. the lens rays;
//   otherwise         -> the single camera ray.
template <bool PRUNE, bool GLOSSY, bool COUNT, bool SPH>
__device__ vec3 render_sample(const FrameParams& P, int x, int y, int sidx, Stack st, Counts& cnt, bool& dofFlag)
{
    Rng rng;
    rng.pixelKey = rng_pixel_key(P.p.seed, uint32_t(y * P.width + x));
    rng.glossCall = 0;
    rng.sample = uint32_t(sidx);
    const float W = float(P.width), H = float(P.height);
    vec3 o, d;
    if (P.f.multiple_rays) {
        const int n = P.p.ray_multiplier;
        const int i = sidx / n, j = sidx - i * n;
        rng.randCounter = 0;
        const float randomx = float(rng.rand31() % 100) / 100.0f;
        const float randomy = float(rng.rand31() % 100) / 100.0f;
        const float px = (float(x) + (float(i) / float(n) + randomx)) / W * 2.0f - 1.0f;
        const float py = (float(y) + (float(j) / float(n) + randomy)) / H * 2.0f - 1.0f;
        camera_ray(P, px, py, o, d);
    } else {
        const float px = float(x) / W * 2.0f - 1.0f;
        const float py = float(y) / H * 2.0f - 1.0f;
        camera_ray(P, px, py, o, d);
        rng.randCounter = 2; // indices 0 and 1 belong to the pixel jitter
        if (P.f.depth_of_field && sidx > 0) lens_ray(P, rng.pixelKey, sidx - 1, o, d);
    }
    cnt.primary++;
    const float dlen = length(d);
    float firstT;
    const vec3 color = final_color<PRUNE, GLOSSY, COUNT, SPH>(P, o, d, rng, st, cnt, firstT);
    dofFlag = firstT != FLT_MAX && fabsf(P.p.dof_f - dlen * firstT) > P.p.dof;
    return color;
}

// ---------------------------------------------------------------------------------------------------
// Queue form of the same per-sample evaluation, used when a sample is one camera ray plus direct lighting (no
// mirror / glossy / transparency / reflection-map / depth-of-field branch can trigger).  Passes (rt_kernels.hip):
//   primary: camera ray -> closest hit; misses are finished on the spot, hits are queued (compacted: the later
//            passes run on full waves);
//   cull   : which light samples of a hit need a shadow ray at all;
//   shadow : the visibility bit of each of those;
//   shade  : one lane per queued hit -> surface attributes, light samples, Phong x visibility.
// ---------------------------------------------------------------------------------------------------
struct HitEntry {
    uint32_t work; // sample * pixelsInSlice + pixel (pixel = Screen-order index inside the rendered row slice)
    float t;
    int32_t slot;
    uint32_t pad; // 1 once k_cull has written the hit's colour (no light sample reaches it)
};

__device__ __forceinline__ void sample_camera_ray(const FrameParams& P, int x, int y, int sidx, Rng& rng, vec3& o, vec3& d)
{
    rng.pixelKey = rng_pixel_key(P.p.seed, uint32_t(y * P.width + x));
    rng.glossCall = 0;
    rng.sample = uint32_t(sidx);
    const float W = float(P.width), H = float(P.height);
    if (P.f.multiple_rays) {
        const int n = P.p.ray_multiplier;
        const int i = sidx / n, j = sidx - i * n;
        rng.randCounter = 0;
        const float randomx = float(rng.rand31() % 100) / 100.0f;
        const float randomy = float(rng.rand31() % 100) / 100.0f;
        const float px = (float(x) + (float(i) / float(n) + randomx)) / W * 2.0f - 1.0f;
        const float py = (float(y) + (float(j) / float(n) + randomy)) / H * 2.0f - 1.0f;
        camera_ray(P, px, py, o, d);
    } else {
        const float px = float(x) / W * 2.0f - 1.0f;
        const float py = float(y) / H * 2.0f - 1.0f;
        camera_ray(P, px, py, o, d);
        rng.randCounter = 2;
        if (P.f.depth_of_field && sidx > 0) lens_ray(P, rng.pixelKey, sidx - 1, o, d);
    }
}

// The reference's test whether a pixel gets its lens rays at all (render.cpp:110): the centre ray hit something at
// distance t and that point is further than P.p.dof from the focal distance.
__device__ __forceinline__ bool dof_flag(const FrameParams& P, vec3 d, float t)
{
    return t != FLT_MAX && fabsf(P.p.dof_f - length(d) * t) > P.p.dof;
}

// y coordinate (counted from the bottom, as the camera and the random streams use it) of a local row
__device__ __forceinline__ int local_row_y(const FrameParams& P, int localRow)
{
    const int block = localRow / P.rowBlock;
    return P.height - 1 - (P.rowTop + block * P.blockStride * P.rowBlock + (localRow - block * P.rowBlock));
}

__device__ __forceinline__ float* sample_slot(const FrameParams& P, size_t pix, int sidx)
{
    if (P.samplesPerPixel == 1) return P.out + pix * 3;
    return P.samples + (size_t(sidx) * size_t(P.rowsLocal) * size_t(P.width) + pix) * 3;
}

} // namespace rt
