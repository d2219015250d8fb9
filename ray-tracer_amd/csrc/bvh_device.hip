// The hierarchy build on the GPU: primitive boxes, the per-level sorts, node numbering, boxes, and the device tables
// (node, primitive, attribute records) without a round trip through host memory.
//
// The reference's builder (src/bounding_volume_hierarchy.cpp:106-208, median variant) sorts the primitives of a node by
// centroid coordinate on axis `level % 3` with std::sort and hands the lower half (count / 2 primitives) to the left
// child.  std::sort is not stable: where equal keys meet at a cut, which primitives go left is decided by the moves of
// libstdc++'s introsort, and those depend on the order the node's primitives arrive in, i.e. on every sort above it.
// The tree is therefore reproduced by running THE SAME ALGORITHM, move for move, as a data-parallel program:
//
//   * introsort's loop (median of three to the front, unguarded Hoare partition around it, both parts again until a
//     part has 16 elements or fewer) runs in ROUNDS; a round partitions every part of every node of the level at once.
//   * One Hoare partition as prefix sums.  With pivot p at the front of [first, last), the sequential loop pairs the
//     j-th element from the left that is >= p with the j-th element from the right that is <= p and swaps them as long
//     as the left one lies before the right one.  Position i holds the j-th ">=" from the left when j = number of
//     ">=" in (first, i), and it is swapped exactly when more than j "<=" lie to its right; symmetrically for the right
//     partners.  Two counts per position (one exclusive scan over packed pairs), a table "j -> position" for both
//     sides, a binary search for the number of swaps k per part, and the swaps themselves are independent.  The
//     returned cut is the smaller of the position of the (k+1)-th ">=" and that of the k-th right partner.
//   * The final insertion sort of introsort never moves an element out of its part (everything in the part before is
//     not greater), and an insertion sort is stable: it is a stable sort of each part of 16 or fewer elements, one
//     thread per part.
//   * A part that is still larger than 16 after 2 * floor(log2 n) partitions is heap-sorted by introsort (on meshes:
//     a part of some twenty elements now and then).  libstdc++'s make_heap / sort_heap are run on it as they are, by one
//     thread per such part.
//
// With the median cut the node ranges of every level follow from the primitive count alone (counts of a level differ
// by at most one), so node numbers (depth-first pre-order, as the reference appends them), leaf slots and inner-node
// ranks are closed-form from per-level tables of subtree sizes, and every node can be written by a thread of its own.
// A leaf's box is a segmented scan over its primitives' boxes in order; inner boxes are combined bottom-up, a level per
// launch, left before right and keeping the earlier value on ties — which is what the host builder's sequential
// fmin / fmax over the node's primitives yields (signed zeros included).
//
// The surface-area-heuristic variant (src/bounding_volume_hierarchy.cpp:133-190) runs on the same sorts; its part of
// this file says how the cuts are found and the nodes numbered when the tree's shape depends on the data.
#include "bvh_build.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <climits>
#include <cstring>
#include <vector>

namespace rt {

namespace {

constexpr int SORT_THRESHOLD = 16; // libstdc++'s _S_threshold
constexpr int TABLE_LEVELS = 34;

struct Keyed {
    float key;
    int index;
};
struct Seg { // one part of introsort's loop: [first, last) with `depth` partitions left before the limit
    int first, last, depth, pad;
};
struct SegOut {
    int k, cut, newL, newR;
};
struct Ctl {
    int nActive[2];
    int flags; // 1: a NaN key, 2: more or larger parts at introsort's depth limit than the heapsort kernel takes
    int pad;
    int nBlocks[TABLE_LEVELS];
    int nHeaps[TABLE_LEVELS];
};
constexpr int MAX_HEAP_PARTS = 1024; // parts per level that reach introsort's depth limit (observed: none to a few)
constexpr int MAX_HEAP_PART = 1 << 16; // ... and their size (observed: about 20 elements; a thread sorts one part)

// Per-level tables of the median-split tree over n primitives.  Every node of level l holds cLo[l] or cLo[l] + 1
// primitives; sub / leaves / inner = nodes, leaves and inner nodes of the subtree of such a node.
struct Tables {
    int n, depth, maxLevel; // depth = index of the deepest level
    int cLo[TABLE_LEVELS];
    int sub[TABLE_LEVELS][2], leaves[TABLE_LEVELS][2], inner[TABLE_LEVELS][2];
};

struct Geometry {
    const float* verts;       // 8 floats per vertex, meshes one after the other
    const uint32_t* tris;     // 3 indices per triangle (into the mesh's own vertices), meshes one after the other
    const int* primBase;      // [nMeshes + 1] first primitive number of every mesh
    const int* vertBase;      // [nMeshes] first vertex of every mesh
    const rt_sphere_desc* spheres;
    int nMeshes, nTriangles, nSpheres;
};

// A box as the segmented scans carry it, and their operator.
struct Box6 {
    float lo[3], hi[3];
};
struct GrowBox { // Builder::grow: the earlier value stays on ties
    __host__ __device__ Box6 operator()(const Box6& x, const Box6& y) const
    {
        Box6 r;
        for (int k = 0; k < 3; k++) {
            r.lo[k] = min_first(x.lo[k], y.lo[k]);
            r.hi[k] = max_first(x.hi[k], y.hi[k]);
        }
        return r;
    }
};
// Inputs of the box scans, made on the fly: position -> box of the primitive that stands there, and -> key of the
// position; reversed, both read position n - 1 - i (a scan from the end).
struct BoxAt {
    const Keyed* a;
    const float *lo, *hi;
    int n, reversed;
    __host__ __device__ Box6 operator()(int i) const
    {
        const size_t p = size_t(a[reversed ? n - 1 - i : i].index);
        Box6 b;
        for (int k = 0; k < 3; k++) {
            b.lo[k] = lo[3 * p + k];
            b.hi[k] = hi[3 * p + k];
        }
        return b;
    }
};
struct KeyAt {
    const int* key;
    int n, reversed;
    __host__ __device__ int operator()(int i) const { return key[reversed ? n - 1 - i : i]; }
};
using Positions = hipcub::CountingInputIterator<int>;
using BoxIterator = hipcub::TransformInputIterator<Box6, BoxAt, Positions>;
using KeyIterator = hipcub::TransformInputIterator<int, KeyAt, Positions>;

__device__ __forceinline__ int mesh_of(const Geometry& g, int prim)
{
    int lo = 0, hi = g.nMeshes; // primBase[lo] <= prim < primBase[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (g.primBase[mid] <= prim) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ void triangle_vertices(const Geometry& g, int prim, int& mesh, const float*& va, const float*& vb, const float*& vc)
{
    mesh = mesh_of(g, prim);
    const uint32_t* t = g.tris + size_t(prim) * 3;
    const float* base = g.verts + size_t(g.vertBase[mesh]) * 8;
    va = base + size_t(t[0]) * 8;
    vb = base + size_t(t[1]) * 8;
    vc = base + size_t(t[2]) * 8;
}

// Box and centroid of every primitive with the host builder's operations (build_bvh in bvh_build.cpp; reference
// src/bounding_volume_hierarchy.cpp:21-31,61-76), and the identity permutation.
__global__ void k_prim_bounds(const Geometry g, int n, float* __restrict__ lo, float* __restrict__ hi, float* __restrict__ centroid, Keyed* __restrict__ a, Ctl* ctl)
{
    const int p = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= n) return;
    vec3 l, h, c;
    if (p < g.nTriangles) {
        int mesh;
        const float *va, *vb, *vc;
        triangle_vertices(g, p, mesh, va, vb, vc);
        const vec3 A = load3(va), B = load3(vb), C = load3(vc);
        l = make3(min_first(min_first(A.x, B.x), C.x), min_first(min_first(A.y, B.y), C.y), min_first(min_first(A.z, B.z), C.z));
        h = make3(max_first(max_first(A.x, B.x), C.x), max_first(max_first(A.y, B.y), C.y), max_first(max_first(A.z, B.z), C.z));
        const vec3 sum = (A + B) + C;
        c = make3(sum.x / 3.0f, sum.y / 3.0f, sum.z / 3.0f);
    } else {
        const rt_sphere_desc& s = g.spheres[p - g.nTriangles];
        c = load3(s.center);
        l = c - splat3(s.radius);
        h = c + splat3(s.radius);
    }
    lo[3 * size_t(p)] = l.x; lo[3 * size_t(p) + 1] = l.y; lo[3 * size_t(p) + 2] = l.z;
    hi[3 * size_t(p)] = h.x; hi[3 * size_t(p) + 1] = h.y; hi[3 * size_t(p) + 2] = h.z;
    centroid[3 * size_t(p)] = c.x; centroid[3 * size_t(p) + 1] = c.y; centroid[3 * size_t(p) + 2] = c.z;
    if (c.x != c.x || c.y != c.y || c.z != c.z) ctl->flags = 1;
    a[p].index = p;
}

__device__ __forceinline__ int floor_log2(int v) { return 31 - __clz(v); }

// Start of a level's sorts: the key of every element on the level's axis, the part table (one part per node of the
// level that has more than 16 primitives, numbered by the node's path) and the nodes of 2..16 primitives as finished
// parts.  The node of position i follows from halving [0, n) `level` times.
__global__ void k_level_init(Keyed* __restrict__ a, const float* __restrict__ centroid, int* __restrict__ segOf, Seg* __restrict__ segs, int2* __restrict__ blocks, Ctl* ctl, int n, int level, int cur, int nActive)
{
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i == 0) ctl->nActive[cur] = nActive;
    if (i >= n) return;
    int first = 0, count = n;
    uint32_t path = 0;
    bool split = true; // the node is reached (no ancestor is a leaf)
    for (int l = 0; l < level; l++) {
        if (count <= 1) {
            split = false;
            break;
        }
        const int left = count / 2;
        path <<= 1;
        if (i < first + left) {
            count = left;
        } else {
            first += left;
            count -= left;
            path |= 1u;
        }
    }
    if (!split || count < 2) {
        segOf[i] = -1;
        return;
    }
    a[i].key = centroid[3 * size_t(a[i].index) + size_t(level % 3)];
    const bool active = count > SORT_THRESHOLD;
    segOf[i] = active ? int(path) : -1;
    if (i == first) {
        if (nActive > 0) segs[path] = active ? Seg { first, first + count, 2 * floor_log2(count), 0 } : Seg { 0, 0, 0, 0 };
        if (!active) blocks[atomicAdd(&ctl->nBlocks[level], 1)] = make_int2(first, first + count);
    }
}

// position of the median of a[first + 1], a[first + (last - first) / 2], a[last - 1] as __move_median_to_first picks it
__device__ __forceinline__ int median_position(const Keyed* a, int first, int last)
{
    const int A = first + 1, B = first + (last - first) / 2, C = last - 1;
    const float ka = a[A].key, kb = a[B].key, kc = a[C].key;
    if (ka < kb) {
        if (kb < kc) return B;
        if (ka < kc) return C;
        return A;
    }
    if (ka < kc) return A;
    if (kb < kc) return C;
    return B;
}

// Round, step 1: per position its two flags (">= pivot" in the high word, "<= pivot" in the low word) as seen AFTER the
// median has been swapped to the front (the swap itself is made in step 2, when nobody reads the keys).
__global__ void k_round_flags(const Keyed* __restrict__ a, const int* __restrict__ segOf, const Seg* __restrict__ segs, unsigned long long* __restrict__ flags, const Ctl* ctl, int n, int cur)
{
    if (ctl->nActive[cur] == 0) return;
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    unsigned long long f = 0;
    const int s = segOf[i];
    if (s >= 0) {
        const Seg sg = segs[s];
        if (i != sg.first) {
            const int m = median_position(a, sg.first, sg.last);
            const float pivot = a[m].key;
            const float key = i == m ? a[sg.first].key : a[i].key;
            f = (key >= pivot ? 1ull << 32 : 0ull) | (key <= pivot ? 1ull : 0ull);
        }
    }
    flags[i] = f;
}

// Round, step 2 (after the exclusive scan of the flags): the tables "j-th '>=' from the left -> position" and
// "j-th '<=' from the right -> position" of every part, kept in the part's own index range.
__global__ void k_round_ranks(Keyed* __restrict__ a, const int* __restrict__ segOf, const Seg* __restrict__ segs, const unsigned long long* __restrict__ scan, int* __restrict__ posL, int* __restrict__ posR, const Ctl* ctl, int n, int cur)
{
    if (ctl->nActive[cur] == 0) return;
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int s = segOf[i];
    if (s < 0) return;
    const Seg sg = segs[s];
    if (i == sg.first) {
        const int m = median_position(a, sg.first, sg.last);
        const Keyed t = a[sg.first];
        a[sg.first] = a[m];
        a[m] = t;
        return;
    }
    const unsigned long long here = scan[i], after = scan[i + 1], base = scan[sg.first + 1], end = scan[sg.last];
    const bool ge = (after >> 32) != (here >> 32), le = uint32_t(after) != uint32_t(here);
    if (ge) posL[sg.first + 1 + int((here >> 32) - (base >> 32))] = i;
    if (le) posR[sg.first + 1 + int(uint32_t(end) - uint32_t(after))] = i;
}

// Round, step 3: per part the number of swaps, the cut, and the two parts it leaves.
__global__ void k_round_cuts(const Seg* __restrict__ segs, Seg* __restrict__ next, SegOut* __restrict__ out, const unsigned long long* __restrict__ scan, const int* __restrict__ posL, const int* __restrict__ posR, int2* __restrict__ blocks, int2* __restrict__ heaps, Ctl* ctl, int level, int cur)
{
    const int s = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (s >= ctl->nActive[cur]) return;
    const Seg sg = segs[s];
    if (sg.last - sg.first <= SORT_THRESHOLD) return; // (a node of the level with exactly 16 primitives)
    const unsigned long long base = scan[sg.first + 1], end = scan[sg.last];
    const int nGe = int((end >> 32) - (base >> 32)), nLe = int(uint32_t(end) - uint32_t(base));
    int lo = 0, hi = nGe < nLe ? nGe : nLe;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (posL[sg.first + mid] < posR[sg.first + mid]) lo = mid; else hi = mid - 1;
    }
    const int k = lo;
    const int cutL = k < nGe ? posL[sg.first + 1 + k] : INT_MAX, cutR = k > 0 ? posR[sg.first + k] : INT_MAX;
    const int cut = cutL < cutR ? cutL : cutR;
    SegOut o { k, cut, -1, -1 };
    for (int c = 0; c < 2; c++) {
        const int f = c ? cut : sg.first, l = c ? sg.last : cut;
        int id = -1;
        if (l - f > SORT_THRESHOLD) {
            if (sg.depth - 1 == 0) {
                // introsort switches to heapsort here: the part is sorted by a thread of its own after the round
                const int h = atomicAdd(&ctl->nHeaps[level], 1);
                if (h < MAX_HEAP_PARTS && l - f <= MAX_HEAP_PART) heaps[level * MAX_HEAP_PARTS + h] = make_int2(f, l);
                else ctl->flags = 2;
            } else {
                id = atomicAdd(&ctl->nActive[cur ^ 1], 1);
                next[id] = Seg { f, l, sg.depth - 1, 0 };
            }
        } else {
            blocks[atomicAdd(&ctl->nBlocks[level], 1)] = make_int2(f, l);
        }
        if (c) o.newR = id; else o.newL = id;
    }
    out[s] = o;
}

// Round, step 4: the swaps (made by the left partner) and the part every position belongs to in the next round.
__global__ void k_round_swaps(Keyed* __restrict__ a, int* __restrict__ segOf, const Seg* __restrict__ segs, const SegOut* __restrict__ out, const unsigned long long* __restrict__ scan, const int* __restrict__ posR, Ctl* ctl, int n, int cur)
{
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i == 0) ctl->nActive[cur] = 0; // (this round's table becomes the next round's output table)
    if (i >= n) return;
    const int s = segOf[i];
    if (s < 0) return;
    const Seg sg = segs[s];
    const SegOut o = out[s];
    if (i != sg.first) {
        const unsigned long long here = scan[i], after = scan[i + 1], base = scan[sg.first + 1];
        const int rankL = int((here >> 32) - (base >> 32));
        if ((after >> 32) != (here >> 32) && rankL < o.k) {
            const int partner = posR[sg.first + 1 + rankL];
            const Keyed t = a[i];
            a[i] = a[partner];
            a[partner] = t;
        }
    }
    segOf[i] = i < o.cut ? o.newL : o.newR;
}

// The final insertion sort: a stable sort of every finished part (2..16 elements).
__global__ void k_sort_blocks(Keyed* __restrict__ a, const int2* __restrict__ blocks, const Ctl* ctl, int level)
{
    const int b = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= ctl->nBlocks[level]) return;
    const int2 r = blocks[b];
    for (int i = r.x + 1; i < r.y; i++) {
        const Keyed v = a[i];
        int j = i;
        while (j > r.x && v.key < a[j - 1].key) {
            a[j] = a[j - 1];
            j--;
        }
        a[j] = v;
    }
}


// libstdc++'s heap sort of a part that reached introsort's depth limit: __partial_sort(first, last, last) =
// __make_heap + __sort_heap, built on __adjust_heap / __push_heap (bits/stl_heap.h), with the same moves.
__device__ void adjust_heap(Keyed* first, int hole, int len, Keyed value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (first[child].key < first[child - 1].key) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    // __push_heap
    int parent = (hole - 1) / 2;
    while (hole > top && first[parent].key < value.key) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
__global__ void k_heap_sorts(Keyed* __restrict__ a, const int2* __restrict__ heaps, const Ctl* ctl, int level)
{
    const int h = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (h >= ctl->nHeaps[level] || h >= MAX_HEAP_PARTS) return;
    const int2 r = heaps[level * MAX_HEAP_PARTS + h];
    Keyed* first = a + r.x;
    const int len = r.y - r.x;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; parent--) { // __make_heap
        adjust_heap(first, parent, len, first[parent]);
        if (parent == 0) break;
    }
    for (int last = len - 1; last >= 1; last--) { // __sort_heap: __pop_heap(first, last, last)
        const Keyed value = first[last];
        first[last] = first[0];
        adjust_heap(first, 0, last, value);
    }
}

// ---- nodes ------------------------------------------------------------------------------------------------------
struct NodeAt {
    int first, count, level;
    int index;     // depth-first pre-order number
    int innerRank; // inner nodes before it in that order
    bool leaf;
};
__device__ __forceinline__ int table(const int (*t)[2], const Tables& T, int level, int count) { return t[level][count - T.cLo[level]]; }

// the node with path bits `path` on `level`; false when an ancestor is a leaf
__device__ __forceinline__ bool node_at(const Tables& T, int level, uint32_t path, NodeAt& nd)
{
    int first = 0, count = T.n, index = 0, innerRank = 0;
    for (int l = 0; l < level; l++) {
        if (count <= 1 || l == T.maxLevel) return false;
        const int left = count / 2;
        if (!((path >> (level - 1 - l)) & 1u)) {
            index += 1;
            innerRank += 1;
            count = left;
        } else {
            index += 1 + table(T.sub, T, l + 1, left);
            innerRank += 1 + table(T.inner, T, l + 1, left);
            first += left;
            count -= left;
        }
    }
    nd = NodeAt { first, count, level, index, innerRank, count <= 1 || level == T.maxLevel };
    return true;
}
__device__ __forceinline__ bool node_of_thread(const Tables& T, NodeAt& nd)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 1ull; // 1-based heap number
    if (t >> (T.depth + 1)) return false;
    const int level = 63 - __clzll((long long)t);
    return node_at(T, level, uint32_t(t - (1ull << level)), nd);
}

// The leaf of every position (halving [0, n) until a node has one primitive or stands on the level limit): the `more`
// count of the slot, and the leaf's first position as the key of the box scan below.
__global__ void k_leaf_of(const Tables T, int* __restrict__ more, int* __restrict__ key)
{
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= T.n) return;
    int first = 0, count = T.n;
    for (int l = 0; l < T.maxLevel && count > 1; l++) {
        const int left = count / 2;
        if (i < first + left) {
            count = left;
        } else {
            first += left;
            count -= left;
        }
    }
    more[i] = first + count - 1 - i;
    key[i] = first;
}
// Every node's record in the reference's order; a leaf's box is the scanned box at its last position.
__global__ void k_nodes(const Tables T, const Box6* __restrict__ pre, rt_node_record* __restrict__ nodes)
{
    NodeAt nd;
    if (!node_of_thread(T, nd)) return;
    rt_node_record rec;
    rec.is_leaf = nd.leaf ? 1 : 0;
    rec.level = nd.level;
    for (int k = 0; k < 3; k++) {
        rec.lower[k] = FLT_MAX;
        rec.upper[k] = -FLT_MAX;
    }
    if (nd.leaf) {
        rec.a = nd.first;
        rec.b = nd.count;
        if (nd.count > 0) {
            const Box6& b = pre[nd.first + nd.count - 1];
            for (int k = 0; k < 3; k++) {
                rec.lower[k] = b.lo[k];
                rec.upper[k] = b.hi[k];
            }
        }
    } else {
        rec.a = nd.index + 1;
        rec.b = nd.index + 1 + table(T.sub, T, nd.level + 1, nd.count / 2);
    }
    nodes[nd.index] = rec;
}

// Boxes of the inner nodes of one level from their children's.
__global__ void k_boxes(const Tables T, int level, rt_node_record* __restrict__ nodes)
{
    const uint32_t path = blockIdx.x * blockDim.x + threadIdx.x;
    if (path >> level) return;
    NodeAt nd;
    if (!node_at(T, level, path, nd) || nd.leaf) return;
    rt_node_record& rec = nodes[nd.index];
    const rt_node_record &L = nodes[rec.a], &R = nodes[rec.b];
    for (int k = 0; k < 3; k++) {
        rec.lower[k] = min_first(L.lower[k], R.lower[k]);
        rec.upper[k] = max_first(L.upper[k], R.upper[k]);
    }
}

// The device table of inner nodes (flatten_bvh in bvh_build.cpp).
__global__ void k_flatten(const Tables T, const rt_node_record* __restrict__ nodes, NodeRec* __restrict__ out)
{
    NodeAt nd;
    if (!node_of_thread(T, nd) || nd.leaf) return;
    const rt_node_record& rec = nodes[nd.index];
    const rt_node_record &L = nodes[rec.a], &R = nodes[rec.b];
    NodeRec d;
    for (int k = 0; k < 3; k++) {
        d.boxL[2 * k] = L.lower[k]; d.boxL[2 * k + 1] = L.upper[k];
        d.boxR[2 * k] = R.lower[k]; d.boxR[2 * k + 1] = R.upper[k];
    }
    d.refL = L.is_leaf ? ~L.a : nd.innerRank + 1;
    d.refR = R.is_leaf ? ~R.a : nd.innerRank + 1 + table(T.inner, T, nd.level + 1, nd.count / 2);
    d.level = nd.level;
    d.refNode = nd.index;
    out[nd.innerRank] = d;
}

// Primitive and attribute records in leaf order, the (triangle, mesh) list of the leaves and the brute-force order.
__global__ void k_prim_records(const Geometry g, int n, const Keyed* __restrict__ a, const int* __restrict__ more, PrimRec* __restrict__ prims, AttrRec* __restrict__ attrs, rt_prim_ref* __restrict__ refs, uint32_t* __restrict__ primOrder)
{
    const int slot = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (slot >= n) return;
    const int p = a[slot].index;
    PrimRec r;
    AttrRec at;
    memset(&r, 0, sizeof r);
    memset(&at, 0, sizeof at);
    r.more = more[slot];
    if (p < g.nTriangles) {
        int mesh;
        const float *va, *vb, *vc;
        triangle_vertices(g, p, mesh, va, vb, vc);
        const vec3 v0 = load3(va), v1 = load3(vb), v2 = load3(vc);
        r.mesh = mesh;
        r.prim = p - g.primBase[mesh];
        for (int k = 0; k < 3; k++) {
            r.v0[k] = va[k]; r.v1[k] = vb[k]; r.v2[k] = vc[k];
            at.n0[k] = va[3 + k]; at.n1[k] = vb[3 + k]; at.n2[k] = vc[3 + k];
        }
        for (int k = 0; k < 2; k++) {
            at.uv0[k] = va[6 + k]; at.uv1[k] = vb[6 + k]; at.uv2[k] = vc[6 + k];
        }
        // trianglePlane and pointInTriangle's total area, reference src/intersect.cpp:52-60,26
        const vec3 nn = cross(v1 - v0, v2 - v0);
        const vec3 normal = is_zero(length(nn)) ? make3(1.0f, 0.0f, 0.0f) : normalize(nn);
        r.n[0] = normal.x; r.n[1] = normal.y; r.n[2] = normal.z;
        r.invArea = 1.0f / length(cross(v0 - v2, v1 - v2));
    } else {
        const rt_sphere_desc& s = g.spheres[p - g.nTriangles];
        r.mesh = -1;
        r.prim = p - g.nTriangles;
        r.v0[0] = s.center[0]; r.v0[1] = s.center[1]; r.v0[2] = s.center[2];
        r.v1[0] = s.radius;
    }
    prims[slot] = r;
    attrs[slot] = at;
    refs[slot] = rt_prim_ref { r.prim, r.mesh };
    primOrder[p] = uint32_t(slot);
}

// ---- the surface-area-heuristic variant ------------------------------------------------------------------------------
// (reference src/bounding_volume_hierarchy.cpp:133-190; sahSplit in bvh_build.cpp.)  A node sorts its primitives on
// x, on y, on z — each sort starting from the arrangement the one before left — prices the cuts of every sorted order,
// and sorts once more on the axis of the cheapest cut.  The cuts depend on the data, so the tree's shape is found level
// by level: all nodes of a level are open at once, every one of the four sorts is one run of the rounds above over
// all of them, the boxes of all prefixes and suffixes of a sorted node are two segmented scans, and the cheapest cut of
// a node is one 64-bit minimum over (cost, axis, position) — the order in which the sequential search keeps the first
// of equal costs.  Nodes are created in level order with links; their depth-first numbers follow from subtree sizes
// (bottom-up) and offsets (top-down) afterwards.
struct SNode {
    int first, last, level, parent;
    int left, right, isLeaf, pad;
    unsigned long long best; // (bits of the cost) << 32 | axis << 30 | cut position; ~0: no cut had a comparable cost
    int sub, inner;          // nodes and inner nodes of the subtree
    int index, innerRank;    // depth-first pre-order number; inner nodes before it in that order
    float lo[3], hi[3];
};
struct SahCtl {
    int nNodes, nOpenNext, maxCountNext, flags; // flags 4: more nodes than the table holds, 8: a cost outside the key's range
};
constexpr unsigned long long NO_CUT = ~0ull;
constexpr int CUT_BITS = 30;

__device__ __forceinline__ float box_area(const float* lo, const float* hi)
{
    const vec3 e = make3(hi[0], hi[1], hi[2]) - make3(lo[0], lo[1], lo[2]);
    return dot(e, make3(e.y, e.z, e.x));
}

// Start of a level: the box of every node of the level (its primitives in the order they arrive in, as the builder's
// bounds() walks them) from the forward scan, and the `more` counts of the slots of its leaves.
__global__ void k_sah_node_boxes(const int* __restrict__ nodeOf, SNode* __restrict__ nodes, const Box6* __restrict__ pre, int* __restrict__ more, int n)
{
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int nd = nodeOf[i];
    if (nd < 0) return;
    SNode& N = nodes[nd];
    if (N.isLeaf) more[i] = N.last - 1 - i;
    if (i == N.last - 1)
        for (int k = 0; k < 3; k++) {
            N.lo[k] = pre[i].lo[k];
            N.hi[k] = pre[i].hi[k];
        }
}

__device__ __forceinline__ int best_axis(unsigned long long best) { return best == NO_CUT ? 0 : int((best >> CUT_BITS) & 3ull); }
__device__ __forceinline__ int best_cut(unsigned long long best) { return best == NO_CUT ? 0 : int(best & ((1ull << CUT_BITS) - 1ull)); }

// Start of one sort of a level: keys on `axis` (or, with axis < 0, on the axis of each node's cheapest cut) and the
// part table, one slot per node of the level (leaves and nodes of 16 primitives or fewer stay empty slots).
__global__ void k_sah_sort_init(Keyed* __restrict__ a, const float* __restrict__ centroid, const int* __restrict__ nodeOf, const SNode* __restrict__ nodes, int* __restrict__ segOf, Seg* __restrict__ segs, int2* __restrict__ blocks, Ctl* ctl, int n, int axis, int cur, int nActive, int levelStart)
{
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i == 0) ctl->nActive[cur] = nActive;
    if (i >= n) return;
    const int nd = nodeOf[i];
    if (nd < 0 || nodes[nd].isLeaf) {
        segOf[i] = -1;
        return;
    }
    const SNode& N = nodes[nd];
    const int count = N.last - N.first;
    const int ax = axis >= 0 ? axis : best_axis(N.best);
    a[i].key = centroid[3 * size_t(a[i].index) + size_t(ax)];
    const bool active = count > SORT_THRESHOLD;
    segOf[i] = active ? nd - levelStart : -1;
    if (i == N.first) {
        if (active) segs[nd - levelStart] = Seg { N.first, N.last, 2 * floor_log2(count), 0 };
        else blocks[atomicAdd(&ctl->nBlocks[0], 1)] = make_int2(N.first, N.last);
    }
}

// The cuts of one sorted order priced (Builder::sahSplit's `consider`).  Without binning the thread of position
// first + k - 1 prices cut k; with binning the first `bins` threads of a node take a bin each: the cut is the number of
// keys below the bin's boundary (at least 1), which is where the sequential search's running position stands, the
// boundaries being non-decreasing.
__global__ void k_sah_eval(const Keyed* __restrict__ a, const int* __restrict__ nodeOf, SNode* __restrict__ nodes, const Box6* __restrict__ pre, const Box6* __restrict__ sufR, SahCtl* sctl, int n, int axis, int bins)
{
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int nd = nodeOf[i];
    if (nd < 0) return;
    const SNode& N = nodes[nd];
    if (N.isLeaf) return;
    const int count = N.last - N.first;
    int k;
    if (size_t(bins) >= size_t(count)) {
        k = i - N.first + 1; // (cuts 0 and count have an empty side: infinite area times zero, never comparable)
        if (k >= count) return;
    } else {
        const int b = i - N.first;
        if (b >= bins) return;
        const float c0 = a[N.first].key;
        const float range = a[N.last - 1].key - c0;
        const float step = range / float(bins);
        const float boundary = c0 + float(b) * step;
        int lo = N.first, hi = N.last; // keys below the boundary: [first, lo)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a[mid].key < boundary) lo = mid + 1; else hi = mid;
        }
        k = lo - N.first;
        if (k < 1) k = 1;
        if (k >= count) { // (the sequential search would have run past the node's end)
            sctl->flags = 8;
            return;
        }
    }
    const Box6& P = pre[N.first + k - 1];
    const Box6& S = sufR[n - 1 - (N.first + k)];
    const float invParent = 1.0f / box_area(N.lo, N.hi);
    const float cost = (box_area(P.lo, P.hi) * float(k) + box_area(S.lo, S.hi) * float(size_t(count - k))) * invParent;
    if (!(cost < FLT_MAX)) return;
    const uint32_t bits = __float_as_uint(cost);
    if (bits >> 31) { // a negative cost (or negative zero) does not order by its bits
        sctl->flags = 8;
        return;
    }
    atomicMin(&nodes[nd].best, ((unsigned long long)bits << 32) | ((unsigned long long)axis << CUT_BITS) | (unsigned long long)k);
}

// End of a level: the two children of every open node.
__global__ void k_sah_children(SNode* __restrict__ nodes, SahCtl* sctl, int levelStart, int levelEnd, int maxLevel, int capacity)
{
    const int t = levelStart + int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= levelEnd) return;
    const SNode N = nodes[t];
    if (N.isLeaf) return;
    const int split = best_cut(N.best);
    const int id = atomicAdd(&sctl->nNodes, 2);
    if (id + 2 > capacity) {
        sctl->flags = 4;
        return;
    }
    for (int c = 0; c < 2; c++) {
        SNode C;
        C.first = c ? N.first + split : N.first;
        C.last = c ? N.last : N.first + split;
        C.level = N.level + 1;
        C.parent = t;
        C.left = C.right = -1;
        C.isLeaf = (C.level == maxLevel || C.last - C.first <= 1) ? 1 : 0;
        C.pad = 0;
        C.best = NO_CUT;
        C.sub = C.inner = C.index = C.innerRank = 0;
        for (int k = 0; k < 3; k++) {
            C.lo[k] = FLT_MAX;
            C.hi[k] = -FLT_MAX;
        }
        nodes[id + c] = C;
        if (!C.isLeaf) {
            atomicAdd(&sctl->nOpenNext, 1);
            atomicMax(&sctl->maxCountNext, C.last - C.first);
        }
    }
    nodes[t].left = id;
    nodes[t].right = id + 1;
}
__global__ void k_sah_assign(int* __restrict__ nodeOf, const SNode* __restrict__ nodes, int n)
{
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int nd = nodeOf[i];
    if (nd < 0) return;
    const SNode& N = nodes[nd];
    nodeOf[i] = N.isLeaf ? -1 : (i < N.first + best_cut(N.best) ? N.left : N.right);
}

// Subtree sizes of one level's nodes (deepest level first), then the numbers of one level's children (root first).
__global__ void k_sah_sizes(SNode* __restrict__ nodes, int levelStart, int levelEnd)
{
    const int t = levelStart + int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= levelEnd) return;
    SNode& N = nodes[t];
    N.sub = N.isLeaf ? 1 : 1 + nodes[N.left].sub + nodes[N.right].sub;
    N.inner = N.isLeaf ? 0 : 1 + nodes[N.left].inner + nodes[N.right].inner;
}
__global__ void k_sah_number(SNode* __restrict__ nodes, int levelStart, int levelEnd)
{
    const int t = levelStart + int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= levelEnd) return;
    const SNode& N = nodes[t];
    if (N.isLeaf) return;
    SNode &L = nodes[N.left], &R = nodes[N.right];
    L.index = N.index + 1;
    L.innerRank = N.innerRank + 1;
    R.index = N.index + 1 + L.sub;
    R.innerRank = N.innerRank + 1 + L.inner;
}

// Node records in the reference's order and the device table of inner nodes.
__global__ void k_sah_write(const SNode* __restrict__ nodes, int nNodes, rt_node_record* __restrict__ refNodes, NodeRec* __restrict__ out)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= nNodes) return;
    const SNode& N = nodes[t];
    rt_node_record rec;
    rec.is_leaf = N.isLeaf;
    rec.level = N.level;
    for (int k = 0; k < 3; k++) {
        rec.lower[k] = N.lo[k];
        rec.upper[k] = N.hi[k];
    }
    if (N.isLeaf) {
        rec.a = N.first;
        rec.b = N.last - N.first;
    } else {
        const SNode &L = nodes[N.left], &R = nodes[N.right];
        rec.a = L.index;
        rec.b = R.index;
        NodeRec d;
        for (int k = 0; k < 3; k++) {
            d.boxL[2 * k] = L.lo[k]; d.boxL[2 * k + 1] = L.hi[k];
            d.boxR[2 * k] = R.lo[k]; d.boxR[2 * k + 1] = R.hi[k];
        }
        d.refL = L.isLeaf ? ~L.first : L.innerRank;
        d.refR = R.isLeaf ? ~R.first : R.innerRank;
        d.level = N.level;
        d.refNode = N.index;
        out[N.innerRank] = d;
    }
    refNodes[N.index] = rec;
}

bool make_tables(int n, int maxLevel, Tables& T)
{
    std::memset(&T, 0, sizeof T);
    T.n = n;
    T.maxLevel = maxLevel;
    int depth = 0;
    while (depth < maxLevel && ((n + (1 << depth) - 1) >> depth) > 1) depth++; // deepest level: every node has <= 1, or the limit
    if (depth >= TABLE_LEVELS - 1) return false;
    T.depth = depth;
    for (int l = 0; l <= depth + 1; l++) T.cLo[l] = n >> l;
    for (int l = depth; l >= 0; l--)
        for (int k = 0; k < 2; k++) {
            const int c = T.cLo[l] + k;
            if (c <= 1 || l == maxLevel) {
                T.sub[l][k] = 1;
                T.leaves[l][k] = 1;
                T.inner[l][k] = 0;
            } else {
                const int left = c / 2, right = c - left;
                const int a = left - T.cLo[l + 1], b = right - T.cLo[l + 1];
                T.sub[l][k] = 1 + T.sub[l + 1][a] + T.sub[l + 1][b];
                T.leaves[l][k] = T.leaves[l + 1][a] + T.leaves[l + 1][b];
                T.inner[l][k] = 1 + T.inner[l + 1][a] + T.inner[l + 1][b];
            }
        }
    return true;
}

template <class T>
hipError_t own(T*& p, size_t count)
{
    void* q = nullptr;
    const hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
    p = static_cast<T*>(q);
    return e;
}

struct Scratch {
    std::vector<void*> all;
    ~Scratch()
    {
        for (void* p : all)
            if (p) (void)hipFree(p);
    }
    template <class T>
    hipError_t get(T*& p, size_t count)
    {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) all.push_back(q);
        p = static_cast<T*>(q);
        return e;
    }
};

} // namespace

void DeviceBvh::release()
{
    for (void** p : { (void**)&nodes, (void**)&prims, (void**)&attrs, (void**)&primOrder, (void**)&refNodes, (void**)&refPrims })
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
}

hipError_t device_build(const HostScene& scene, const rt_bvh_options& opt, DeviceBvh& out, bool& unsupported)
{
    unsupported = true;
    out = DeviceBvh {};
    size_t nTriangles = 0, nVertices = 0;
    for (const HostMesh& m : scene.meshes) {
        nTriangles += m.triangles.size() / 3;
        nVertices += m.vertices.size() / 8;
    }
    const size_t total = nTriangles + scene.spheres.size();
    if (total < 2 || total >= (size_t(1) << 28) || opt.max_level < 0 || opt.max_level >= TABLE_LEVELS - 1) return hipSuccess;
    const bool sah = opt.sah_binning != 0;
    if (sah && opt.max_level < 1) return hipSuccess; // (the root a leaf of everything)
    const int n = int(total);
    Tables T;
    if (!sah) {
        if (!make_tables(n, opt.max_level, T)) return hipSuccess;
    }
    hipError_t e;
#define RT_TRY(call) do { if ((e = (call)) != hipSuccess) return e; } while (0)
    Scratch tmp;
    // ---- geometry
    float* dVerts;
    uint32_t* dTris;
    int *dPrimBase, *dVertBase;
    rt_sphere_desc* dSpheres;
    RT_TRY(tmp.get(dVerts, nVertices * 8));
    RT_TRY(tmp.get(dTris, nTriangles * 3));
    RT_TRY(tmp.get(dPrimBase, scene.meshes.size() + 1));
    RT_TRY(tmp.get(dVertBase, scene.meshes.size() + 1));
    RT_TRY(tmp.get(dSpheres, scene.spheres.size()));
    std::vector<int> primBase(scene.meshes.size() + 1, 0), vertBase(scene.meshes.size() + 1, 0);
    for (size_t m = 0; m < scene.meshes.size(); m++) {
        const HostMesh& mesh = scene.meshes[m];
        if (!mesh.vertices.empty()) RT_TRY(hipMemcpyAsync(dVerts + size_t(vertBase[m]) * 8, mesh.vertices.data(), mesh.vertices.size() * sizeof(float), hipMemcpyHostToDevice, nullptr));
        if (!mesh.triangles.empty()) RT_TRY(hipMemcpyAsync(dTris + size_t(primBase[m]) * 3, mesh.triangles.data(), mesh.triangles.size() * sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));
        primBase[m + 1] = primBase[m] + int(mesh.triangles.size() / 3);
        vertBase[m + 1] = vertBase[m] + int(mesh.vertices.size() / 8);
    }
    RT_TRY(hipMemcpyAsync(dPrimBase, primBase.data(), primBase.size() * sizeof(int), hipMemcpyHostToDevice, nullptr));
    RT_TRY(hipMemcpyAsync(dVertBase, vertBase.data(), vertBase.size() * sizeof(int), hipMemcpyHostToDevice, nullptr));
    if (!scene.spheres.empty()) RT_TRY(hipMemcpyAsync(dSpheres, scene.spheres.data(), scene.spheres.size() * sizeof(rt_sphere_desc), hipMemcpyHostToDevice, nullptr));
    const Geometry geo { dVerts, dTris, dPrimBase, dVertBase, dSpheres, int(scene.meshes.size()), int(nTriangles), int(scene.spheres.size()) };
    // ---- sort state
    float *dLo, *dHi, *dCentroid;
    Keyed* a;
    int *segOf, *posL, *posR, *more;
    unsigned long long *flags, *scan;
    Seg* segs[2];
    SegOut* segOut;
    int2 *blocks, *heaps;
    Ctl* ctl;
    // part tables: the median variant numbers the parts of more than 16; the other has a slot per node of a level
    const size_t maxSegs = sah ? size_t(n) + 2 : size_t(n) / (SORT_THRESHOLD + 1) + 2;
    RT_TRY(tmp.get(dLo, size_t(n) * 3));
    RT_TRY(tmp.get(dHi, size_t(n) * 3));
    RT_TRY(tmp.get(dCentroid, size_t(n) * 3));
    RT_TRY(tmp.get(a, size_t(n)));
    RT_TRY(tmp.get(segOf, size_t(n)));
    RT_TRY(tmp.get(posL, size_t(n) + 1));
    RT_TRY(tmp.get(posR, size_t(n) + 1));
    RT_TRY(tmp.get(more, size_t(n)));
    RT_TRY(tmp.get(flags, size_t(n) + 1));
    RT_TRY(tmp.get(scan, size_t(n) + 1));
    RT_TRY(tmp.get(segs[0], maxSegs));
    RT_TRY(tmp.get(segs[1], maxSegs));
    RT_TRY(tmp.get(segOut, maxSegs));
    RT_TRY(tmp.get(blocks, size_t(n)));
    RT_TRY(tmp.get(heaps, size_t(TABLE_LEVELS) * MAX_HEAP_PARTS));
    RT_TRY(tmp.get(ctl, 1));
    RT_TRY(hipMemsetAsync(ctl, 0, sizeof(Ctl), nullptr));
    RT_TRY(hipMemsetAsync(flags + n, 0, sizeof(unsigned long long), nullptr));
    size_t scanBytes = 0;
    RT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, flags, scan, n + 1, nullptr));
    char* scanTemp;
    RT_TRY(tmp.get(scanTemp, scanBytes));
    const dim3 block(256), grid(uint32_t((size_t(n) + 255) / 256));
    hipLaunchKernelGGL(k_prim_bounds, grid, block, 0, nullptr, geo, n, dLo, dHi, dCentroid, a, ctl);
    int cur = 0;
    Ctl host;
    bool giveUp = false; // a NaN key, or too much for the heapsort kernel: the host builds
    // The rounds of one sort over the part table segs[cur] (counters of slot `slot`), then its heap sorts.  introsort
    // needs about 1.4 log2(count / 16) partitions of a part; the rounds are launched in batches and the number of parts
    // still active is read back after each.
    auto run_rounds = [&](int slot, int maxCount) -> hipError_t {
        int batch = 2;
        for (int c = maxCount; c > SORT_THRESHOLD; c >>= 1) batch += (batch & 1) ? 2 : 1;
        while (true) {
            for (int r = 0; r < batch; r++) {
                hipLaunchKernelGGL(k_round_flags, grid, block, 0, nullptr, a, segOf, segs[cur], flags, ctl, n, cur);
                RT_TRY(hipcub::DeviceScan::ExclusiveSum(scanTemp, scanBytes, flags, scan, n + 1, nullptr));
                hipLaunchKernelGGL(k_round_ranks, grid, block, 0, nullptr, a, segOf, segs[cur], scan, posL, posR, ctl, n, cur);
                hipLaunchKernelGGL(k_round_cuts, dim3(uint32_t((maxSegs + 255) / 256)), block, 0, nullptr, segs[cur], segs[cur ^ 1], segOut, scan, posL, posR, blocks, heaps, ctl, slot, cur);
                hipLaunchKernelGGL(k_round_swaps, grid, block, 0, nullptr, a, segOf, segs[cur], segOut, scan, posR, ctl, n, cur);
                cur ^= 1;
            }
            RT_TRY(hipMemcpyAsync(&host, ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, nullptr));
            RT_TRY(hipStreamSynchronize(nullptr));
            if (host.flags) {
                giveUp = true;
                return hipSuccess;
            }
            if (host.nActive[cur] == 0) break;
            batch = 3;
        }
        hipLaunchKernelGGL(k_heap_sorts, dim3(MAX_HEAP_PARTS / 256), block, 0, nullptr, a, heaps, ctl, slot);
        return hipSuccess;
    };
    struct Guard {
        DeviceBvh& d;
        bool keep = false;
        ~Guard() { if (!keep) d.release(); }
    } guard { out };
    out.numPrims = size_t(n);
    if (!sah) {
        // ---- the sorts, level by level
        for (int level = 0; level < T.depth; level++) {
            const int maxCount = T.cLo[level] + 1;
            const int nActive = maxCount > SORT_THRESHOLD ? (1 << level) : 0; // (then every node of the level exists)
            if (size_t(nActive) > maxSegs) return hipErrorInvalidValue;       // cannot happen: 2^level parts of more than 16
            hipLaunchKernelGGL(k_level_init, grid, block, 0, nullptr, a, dCentroid, segOf, segs[cur], blocks, ctl, n, level, cur, nActive);
            if (nActive > 0) {
                RT_TRY(run_rounds(level, maxCount));
                if (giveUp) return hipSuccess;
            }
            hipLaunchKernelGGL(k_sort_blocks, grid, block, 0, nullptr, a, blocks, ctl, level);
        }
        RT_TRY(hipMemcpyAsync(&host, ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, nullptr));
        RT_TRY(hipStreamSynchronize(nullptr));
        if (host.flags) return hipSuccess;
        // ---- nodes, boxes, device tables
        out.numRefNodes = size_t(T.sub[0][n - T.cLo[0]]);
        out.numInner = size_t(T.inner[0][n - T.cLo[0]]);
        out.numLeaves = T.leaves[0][n - T.cLo[0]];
        out.numLevels = T.depth + 1;
        RT_TRY(own(out.nodes, out.numInner));
        RT_TRY(own(out.refNodes, out.numRefNodes));
        // leaf boxes: a segmented scan over the primitives' boxes in their final order, one segment per leaf (the keys
        // take the place of the sorts' part numbers, which nobody needs any more)
        Box6* pre;
        RT_TRY(tmp.get(pre, size_t(n)));
        const KeyIterator keys(Positions(0), KeyAt { segOf, n, 0 });
        const BoxIterator boxes(Positions(0), BoxAt { a, dLo, dHi, n, 0 });
        size_t boxScanBytes = 0;
        RT_TRY(hipcub::DeviceScan::InclusiveScanByKey(nullptr, boxScanBytes, keys, boxes, pre, GrowBox(), n, hipcub::Equality(), nullptr));
        char* boxScanTemp;
        RT_TRY(tmp.get(boxScanTemp, boxScanBytes));
        hipLaunchKernelGGL(k_leaf_of, grid, block, 0, nullptr, T, more, segOf);
        RT_TRY(hipcub::DeviceScan::InclusiveScanByKey(boxScanTemp, boxScanBytes, keys, boxes, pre, GrowBox(), n, hipcub::Equality(), nullptr));
        const unsigned long long heap = (2ull << T.depth) - 1ull; // threads of the per-node kernels: every (level, path)
        const dim3 gridNodes(uint32_t((heap + 255) / 256));
        hipLaunchKernelGGL(k_nodes, gridNodes, block, 0, nullptr, T, pre, out.refNodes);
        for (int level = T.depth - 1; level >= 0; level--)
            hipLaunchKernelGGL(k_boxes, dim3(uint32_t(((1ull << level) + 255) / 256)), block, 0, nullptr, T, level, out.refNodes);
        hipLaunchKernelGGL(k_flatten, gridNodes, block, 0, nullptr, T, out.refNodes, out.nodes);
    } else {
        // ---- the tree, level by level
        // 2 n - 1 nodes, unless nodes without area cut at position 0 and leave an empty leaf behind on every level down to
        // the limit: a level has n nodes at most, which small scenes are given room for; beyond the table the host builds
        const long long worst = (long long)n * (opt.max_level + 1) + 1;
        const int capacity = int(std::max<long long>(4ll * n + 64, std::min<long long>(worst, 1ll << 18)));
        SNode* nodes;
        SahCtl* sctl;
        int* nodeOf;
        Box6 *pre, *sufR;
        RT_TRY(tmp.get(nodes, size_t(capacity)));
        RT_TRY(tmp.get(sctl, 1));
        RT_TRY(tmp.get(nodeOf, size_t(n)));
        RT_TRY(tmp.get(pre, size_t(n)));
        RT_TRY(tmp.get(sufR, size_t(n)));
        const KeyIterator keys(Positions(0), KeyAt { nodeOf, n, 0 }), keysR(Positions(0), KeyAt { nodeOf, n, 1 });
        const BoxIterator boxes(Positions(0), BoxAt { a, dLo, dHi, n, 0 }), boxesR(Positions(0), BoxAt { a, dLo, dHi, n, 1 });
        size_t boxScanBytes = 0;
        RT_TRY(hipcub::DeviceScan::InclusiveScanByKey(nullptr, boxScanBytes, keys, boxes, pre, GrowBox(), n, hipcub::Equality(), nullptr));
        char* boxScanTemp;
        RT_TRY(tmp.get(boxScanTemp, boxScanBytes));
        SNode root;
        std::memset(&root, 0, sizeof root);
        root.last = n;
        root.parent = root.left = root.right = -1;
        root.best = NO_CUT;
        for (int k = 0; k < 3; k++) {
            root.lo[k] = FLT_MAX;
            root.hi[k] = -FLT_MAX;
        }
        SahCtl sh { 1, 1, n, 0 };
        RT_TRY(hipMemcpyAsync(nodes, &root, sizeof root, hipMemcpyHostToDevice, nullptr));
        RT_TRY(hipMemcpyAsync(sctl, &sh, sizeof sh, hipMemcpyHostToDevice, nullptr));
        RT_TRY(hipMemsetAsync(nodeOf, 0, size_t(n) * sizeof(int), nullptr));
        std::vector<int> levelStart { 0, 1 }; // nodes of level l: [levelStart[l], levelStart[l + 1])
        for (int level = 0;; level++) {
            const int first = levelStart[size_t(level)], end = levelStart[size_t(level) + 1];
            const int nOpen = sh.nOpenNext, maxCount = sh.maxCountNext;
            RT_TRY(hipcub::DeviceScan::InclusiveScanByKey(boxScanTemp, boxScanBytes, keys, boxes, pre, GrowBox(), n, hipcub::Equality(), nullptr));
            hipLaunchKernelGGL(k_sah_node_boxes, grid, block, 0, nullptr, nodeOf, nodes, pre, more, n);
            if (nOpen == 0) break;
            if (size_t(end - first) > maxSegs) return hipErrorInvalidValue; // cannot happen: a level's nodes have n primitives
            const int nActive = maxCount > SORT_THRESHOLD ? end - first : 0;
            for (int pass = 0; pass < 4; pass++) { // x, y, z, and the axis of the cheapest cut
                RT_TRY(hipMemsetAsync(&ctl->nBlocks[0], 0, sizeof(int), nullptr));
                RT_TRY(hipMemsetAsync(&ctl->nHeaps[0], 0, sizeof(int), nullptr));
                if (nActive > 0) RT_TRY(hipMemsetAsync(segs[cur], 0, size_t(nActive) * sizeof(Seg), nullptr));
                hipLaunchKernelGGL(k_sah_sort_init, grid, block, 0, nullptr, a, dCentroid, nodeOf, nodes, segOf, segs[cur], blocks, ctl, n, pass < 3 ? pass : -1, cur, nActive, first);
                if (nActive > 0) {
                    RT_TRY(run_rounds(0, maxCount));
                    if (giveUp) return hipSuccess;
                }
                hipLaunchKernelGGL(k_sort_blocks, grid, block, 0, nullptr, a, blocks, ctl, 0);
                if (pass == 3) break;
                RT_TRY(hipcub::DeviceScan::InclusiveScanByKey(boxScanTemp, boxScanBytes, keys, boxes, pre, GrowBox(), n, hipcub::Equality(), nullptr));
                RT_TRY(hipcub::DeviceScan::InclusiveScanByKey(boxScanTemp, boxScanBytes, keysR, boxesR, sufR, GrowBox(), n, hipcub::Equality(), nullptr));
                hipLaunchKernelGGL(k_sah_eval, grid, block, 0, nullptr, a, nodeOf, nodes, pre, sufR, sctl, n, pass, opt.sah_bins);
            }
            RT_TRY(hipMemsetAsync(&sctl->nOpenNext, 0, 2 * sizeof(int), nullptr));
            hipLaunchKernelGGL(k_sah_children, dim3(uint32_t((size_t(end - first) + 255) / 256)), block, 0, nullptr, nodes, sctl, first, end, opt.max_level, capacity);
            hipLaunchKernelGGL(k_sah_assign, grid, block, 0, nullptr, nodeOf, nodes, n);
            RT_TRY(hipMemcpyAsync(&sh, sctl, sizeof sh, hipMemcpyDeviceToHost, nullptr));
            RT_TRY(hipMemcpyAsync(&host, ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, nullptr));
            RT_TRY(hipStreamSynchronize(nullptr));
            if (sh.flags || host.flags) return hipSuccess;
            levelStart.push_back(sh.nNodes);
        }
        RT_TRY(hipMemcpyAsync(&host, ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, nullptr));
        RT_TRY(hipStreamSynchronize(nullptr));
        if (host.flags) return hipSuccess;
        // ---- numbers and tables
        const int numLevels = int(levelStart.size()) - 1, nNodes = levelStart.back();
        auto launch_level = [&](void (*kernel)(SNode*, int, int), int level) {
            const int first = levelStart[size_t(level)], end = levelStart[size_t(level) + 1];
            if (end > first) hipLaunchKernelGGL(kernel, dim3(uint32_t((size_t(end - first) + 255) / 256)), block, 0, nullptr, nodes, first, end);
        };
        for (int level = numLevels - 1; level >= 0; level--) launch_level(k_sah_sizes, level);
        for (int level = 0; level < numLevels - 1; level++) launch_level(k_sah_number, level);
        RT_TRY(hipMemcpyAsync(&root, nodes, sizeof root, hipMemcpyDeviceToHost, nullptr));
        RT_TRY(hipStreamSynchronize(nullptr));
        if (root.sub != nNodes) return hipErrorInvalidValue;
        out.numRefNodes = size_t(nNodes);
        out.numInner = size_t(root.inner);
        out.numLeaves = nNodes - root.inner;
        out.numLevels = numLevels;
        RT_TRY(own(out.nodes, out.numInner));
        RT_TRY(own(out.refNodes, out.numRefNodes));
        hipLaunchKernelGGL(k_sah_write, dim3(uint32_t((size_t(nNodes) + 255) / 256)), block, 0, nullptr, nodes, nNodes, out.refNodes, out.nodes);
    }
    RT_TRY(own(out.prims, out.numPrims));
    RT_TRY(own(out.attrs, out.numPrims));
    RT_TRY(own(out.primOrder, out.numPrims));
    RT_TRY(own(out.refPrims, out.numPrims));
    hipLaunchKernelGGL(k_prim_records, grid, block, 0, nullptr, geo, n, a, more, out.prims, out.attrs, out.refPrims, out.primOrder);
    RT_TRY(hipGetLastError());
    rt_node_record rootRec;
    RT_TRY(hipMemcpyAsync(&rootRec, out.refNodes, sizeof rootRec, hipMemcpyDeviceToHost, nullptr));
    RT_TRY(hipStreamSynchronize(nullptr));
#undef RT_TRY
    std::memcpy(out.rootLo, rootRec.lower, 12);
    std::memcpy(out.rootHi, rootRec.upper, 12);
    out.rootRef = rootRec.is_leaf ? ~rootRec.a : 0;
    guard.keep = true;
    unsupported = false;
    return hipSuccess;
}

} // namespace rt
