#include "bvh_build.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <exception>
#include <system_error>
#include <thread>

namespace rt {

bool copy_scene(const rt_scene_desc& d, HostScene& out, std::string& error)
{
    if ((d.n_meshes && !d.meshes) || (d.n_textures && !d.textures) || (d.n_spheres && !d.spheres)) {
        error = "scene description has a null table with a non-zero count";
        return false;
    }
    out.meshes.resize(d.n_meshes);
    for (uint32_t m = 0; m < d.n_meshes; m++) {
        const rt_mesh_desc& md = d.meshes[m];
        if ((md.n_vertices && !md.vertices) || (md.n_triangles && !md.triangles)) {
            error = "mesh " + std::to_string(m) + " has a null vertex or triangle pointer";
            return false;
        }
        if (md.texture >= int32_t(d.n_textures) || md.texture < -1) {
            error = "mesh " + std::to_string(m) + " names texture " + std::to_string(md.texture) + " which does not exist";
            return false;
        }
        HostMesh& hm = out.meshes[m];
        hm.vertices.assign(md.vertices, md.vertices + size_t(md.n_vertices) * 8);
        hm.triangles.assign(md.triangles, md.triangles + size_t(md.n_triangles) * 3);
        for (uint32_t idx : hm.triangles)
            if (idx >= md.n_vertices) {
                error = "mesh " + std::to_string(m) + " has a triangle index beyond its vertex count";
                return false;
            }
        MaterialRec& mat = hm.material;
        std::memset(&mat, 0, sizeof mat);
        std::memcpy(mat.kd, md.kd, sizeof mat.kd);
        std::memcpy(mat.ks, md.ks, sizeof mat.ks);
        mat.shininess = md.shininess;
        mat.transparency = md.transparency;
        mat.texture = md.texture;
    }
    out.spheres.assign(d.spheres, d.spheres + d.n_spheres);
    auto copyImage = [&](const rt_image_desc& im, HostImage& hi, const char* what) {
        if (!im.rgb || im.width <= 0 || im.height <= 0) {
            error = std::string(what) + " image is empty";
            return false;
        }
        hi.width = im.width;
        hi.height = im.height;
        hi.rgb.assign(im.rgb, im.rgb + size_t(im.width) * size_t(im.height) * 3);
        return true;
    };
    out.textures.resize(d.n_textures);
    for (uint32_t i = 0; i < d.n_textures; i++)
        if (!copyImage(d.textures[i], out.textures[i], "texture")) return false;
    out.skybox.clear();
    if (d.skybox) {
        out.skybox.resize(6);
        for (int i = 0; i < 6; i++)
            if (!copyImage(d.skybox[i], out.skybox[size_t(i)], "skybox")) return false;
    }
    return true;
}

namespace {

struct PrimBounds {
    vec3 lo, hi, centroid;
    int32_t prim, mesh;
};

// One subtree in reference order (node before its left subtree before its right subtree, primitives in leaf
// order), with indices counted from the subtree's own first node / first primitive.
struct Subtree {
    std::vector<rt_node_record> nodes;
    std::vector<rt_prim_ref> prims;
    int numLevels = 0, numLeaves = 0;
};

struct Bounds {
    vec3 lo, hi;
};

struct Builder {
    const std::vector<PrimBounds>& prims;
    std::vector<int> order; // permutation of prims being partitioned in place; a subtree owns its range of it
    int maxLevel, sahBins;
    bool sah;
    int spawnLevels; // subtrees above this level are built by a thread of their own
    size_t spawnMinimum;

    static Bounds empty() { return { splat3(FLT_MAX), splat3(-FLT_MAX) }; }
    static void grow(Bounds& b, const PrimBounds& p)
    {
        b.lo = make3(min_first(b.lo.x, p.lo.x), min_first(b.lo.y, p.lo.y), min_first(b.lo.z, p.lo.z));
        b.hi = make3(max_first(b.hi.x, p.hi.x), max_first(b.hi.y, p.hi.y), max_first(b.hi.z, p.hi.z));
    }
    Bounds bounds(size_t first, size_t last) const
    {
        Bounds b = empty();
        for (size_t i = first; i < last; i++) grow(b, prims[size_t(order[i])]);
        return b;
    }
    static float area(const Bounds& b)
    {
        vec3 e = b.hi - b.lo;
        return dot(b.hi - b.lo, make3(e.y, e.z, e.x));
    }
    // std::sort of the range by centroid coordinate, as the reference does it on its primitive list.  The sort
    // runs on (coordinate, index) pairs: the algorithm sees the same comparison results as with an indirect
    // comparator, so it makes the same moves (equal coordinates included) without chasing an index per comparison.
    void sortAxis(size_t first, size_t last, int axis)
    {
        struct Keyed {
            float key;
            int index;
        };
        thread_local std::vector<Keyed> keyed;
        keyed.resize(last - first);
        for (size_t i = first; i < last; i++) keyed[i - first] = { comp(prims[size_t(order[i])].centroid, axis), order[i] };
        std::sort(keyed.begin(), keyed.end(), [](const Keyed& a, const Keyed& b) { return a.key < b.key; });
        for (size_t i = first; i < last; i++) order[i] = keyed[i - first].index;
    }

    // SAH split search over the three axes; returns the split offset and leaves the range sorted on the
    // winning axis (reference …hierarchy.cpp:133-190).  The reference recomputes both boxes for every candidate;
    // here the boxes of all prefixes and suffixes of the sorted range are made in two passes (min and max do not
    // depend on the order they are taken in).
    size_t sahSplit(size_t first, size_t last, const Bounds& whole)
    {
        const size_t count = last - first;
        float best = FLT_MAX;
        const float invParent = 1.0f / area(whole);
        int bestAxis = -1;
        size_t split = 0;
        thread_local std::vector<Bounds> prefix, suffix;
        prefix.resize(count + 1);
        suffix.resize(count + 1);
        for (int axis = 0; axis < 3; axis++) {
            sortAxis(first, last, axis);
            prefix[0] = empty();
            for (size_t k = 0; k < count; k++) {
                prefix[k + 1] = prefix[k];
                grow(prefix[k + 1], prims[size_t(order[first + k])]);
            }
            suffix[count] = empty();
            for (size_t k = count; k > 0; k--) {
                suffix[k - 1] = suffix[k];
                grow(suffix[k - 1], prims[size_t(order[first + k - 1])]);
            }
            auto consider = [&](size_t k) {
                float cost = (area(prefix[k]) * float(int(k)) + area(suffix[k]) * float(count - k)) * invParent;
                if (cost < best) {
                    best = cost;
                    bestAxis = axis;
                    split = k;
                }
            };
            if (size_t(sahBins) >= count) {
                for (size_t k = 0; k <= count; k++) consider(k);
            } else {
                const float c0 = comp(prims[size_t(order[first])].centroid, axis);
                const float range = comp(prims[size_t(order[last - 1])].centroid, axis) - c0;
                const float step = range / float(sahBins);
                size_t left = 1;
                for (int b = 0; b < sahBins; b++) {
                    const float boundary = c0 + float(b) * step;
                    while (comp(prims[size_t(order[first + left])].centroid, axis) < boundary) left++;
                    consider(left);
                }
            }
        }
        if (bestAxis < 0) bestAxis = 0; // every cost was NaN; the reference indexes centr[-1] here
        sortAxis(first, last, bestAxis);
        return split;
    }

    // appends `sub` behind the nodes and primitives `into` already has
    static int splice(Subtree& into, const Subtree& sub)
    {
        const int32_t nodeBase = int32_t(into.nodes.size()), primBase = int32_t(into.prims.size());
        for (rt_node_record rec : sub.nodes) {
            if (rec.is_leaf) {
                rec.a += primBase;
            } else {
                rec.a += nodeBase;
                rec.b += nodeBase;
            }
            into.nodes.push_back(rec);
        }
        into.prims.insert(into.prims.end(), sub.prims.begin(), sub.prims.end());
        into.numLevels = std::max(into.numLevels, sub.numLevels);
        into.numLeaves += sub.numLeaves;
        return nodeBase;
    }

    void node(size_t first, size_t last, int level, int parent, Subtree& out)
    {
        out.numLevels = std::max(out.numLevels, level + 1);
        const int self = int(out.nodes.size());
        rt_node_record rec;
        std::memset(&rec, 0, sizeof rec);
        rec.level = level;
        const Bounds box = bounds(first, last);
        rec.lower[0] = box.lo.x; rec.lower[1] = box.lo.y; rec.lower[2] = box.lo.z;
        rec.upper[0] = box.hi.x; rec.upper[1] = box.hi.y; rec.upper[2] = box.hi.z;
        if (parent >= 0) {
            rt_node_record& p = out.nodes[size_t(parent)];
            if (p.a < 0) p.a = self; else p.b = self;
        }
        const size_t count = last - first;
        if (level == maxLevel || count <= 1) {
            rec.is_leaf = 1;
            rec.a = int32_t(out.prims.size());
            rec.b = int32_t(count);
            for (size_t i = first; i < last; i++) {
                const PrimBounds& p = prims[size_t(order[i])];
                out.prims.push_back(rt_prim_ref { p.prim, p.mesh });
            }
            out.numLeaves++;
            out.nodes.push_back(rec);
            return;
        }
        rec.is_leaf = 0;
        rec.a = rec.b = -1;
        size_t split;
        if (sah) {
            split = sahSplit(first, last, box);
        } else {
            sortAxis(first, last, level % 3);
            split = count / 2;
        }
        out.nodes.push_back(rec);
        if (level < spawnLevels && count >= spawnMinimum) {
            // the two halves own disjoint ranges of `order`: the left one goes to another thread, and both are
            // put behind this node in the reference's order afterwards
            Subtree left, right;
            // If the system refuses another thread the left half is built here as well; the guard joins the
            // thread also when the right half throws (a joinable std::thread must not be destroyed).
            struct Joiner {
                std::thread t;
                std::exception_ptr error;
                ~Joiner() { if (t.joinable()) t.join(); }
            } other;
            bool spawned = true;
            try {
                other.t = std::thread([&] {
                    try {
                        node(first, first + split, level + 1, -1, left);
                    } catch (...) {
                        other.error = std::current_exception();
                    }
                });
            } catch (const std::system_error&) {
                spawned = false;
            }
            if (!spawned) node(first, first + split, level + 1, -1, left);
            node(first + split, last, level + 1, -1, right);
            if (other.t.joinable()) other.t.join();
            if (other.error) std::rethrow_exception(other.error);
            out.nodes[size_t(self)].a = splice(out, left);
            out.nodes[size_t(self)].b = splice(out, right);
            return;
        }
        node(first, first + split, level + 1, self, out);
        node(first + split, last, level + 1, self, out);
    }
};

} // namespace

void build_bvh(const HostScene& scene, const rt_bvh_options& opt, HostBvh& out)
{
    out = HostBvh {};
    std::vector<PrimBounds> prims;
    size_t total = scene.spheres.size();
    for (const HostMesh& m : scene.meshes) total += m.triangles.size() / 3;
    prims.reserve(total);
    for (size_t m = 0; m < scene.meshes.size(); m++) {
        const HostMesh& mesh = scene.meshes[m];
        for (size_t t = 0; t < mesh.triangles.size() / 3; t++) {
            const vec3 a = load3(&mesh.vertices[size_t(mesh.triangles[3 * t]) * 8]);
            const vec3 b = load3(&mesh.vertices[size_t(mesh.triangles[3 * t + 1]) * 8]);
            const vec3 c = load3(&mesh.vertices[size_t(mesh.triangles[3 * t + 2]) * 8]);
            PrimBounds p;
            p.lo = make3(min_first(min_first(a.x, b.x), c.x), min_first(min_first(a.y, b.y), c.y), min_first(min_first(a.z, b.z), c.z));
            p.hi = make3(max_first(max_first(a.x, b.x), c.x), max_first(max_first(a.y, b.y), c.y), max_first(max_first(a.z, b.z), c.z));
            const vec3 sum = (a + b) + c;
            p.centroid = make3(sum.x / 3.0f, sum.y / 3.0f, sum.z / 3.0f);
            p.prim = int32_t(t);
            p.mesh = int32_t(m);
            prims.push_back(p);
        }
    }
    for (size_t s = 0; s < scene.spheres.size(); s++) {
        const rt_sphere_desc& sp = scene.spheres[s];
        const vec3 c = load3(sp.center);
        prims.push_back(PrimBounds { c - splat3(sp.radius), c + splat3(sp.radius), c, int32_t(s), -1 });
    }
    // host threads: 2^spawnLevels subtrees at most run side by side (a GPU box gives a process 16 cores)
    unsigned threads = std::thread::hardware_concurrency();
    if (const char* env = std::getenv("RT_BUILD_THREADS")) threads = unsigned(std::max(1, std::atoi(env)));
    threads = std::min(threads < 1 ? 1u : threads, 16u);
    int spawnLevels = 0;
    while ((1u << (spawnLevels + 1)) <= threads) spawnLevels++;
    Builder b { prims, std::vector<int>(prims.size()), opt.max_level, opt.sah_bins, opt.sah_binning != 0, spawnLevels, size_t(4096) };
    std::iota(b.order.begin(), b.order.end(), 0);
    Subtree tree;
    tree.nodes.reserve(prims.size() * 2 + 1);
    tree.prims.reserve(prims.size());
    b.node(0, prims.size(), 0, -1, tree);
    out.nodes = std::move(tree.nodes);
    out.prims = std::move(tree.prims);
    out.numLevels = tree.numLevels;
    out.numLeaves = tree.numLeaves;
}

void flatten_bvh(const HostScene& scene, const HostBvh& bvh, FlatBvh& out)
{
    out = FlatBvh {};
    const size_t n = bvh.nodes.size();
    // device index of every inner node: its rank among inner nodes in reference order
    std::vector<int32_t> innerIndex(n, -1);
    int32_t inners = 0;
    for (size_t i = 0; i < n; i++)
        if (!bvh.nodes[i].is_leaf) innerIndex[i] = inners++;
    auto refOf = [&](int32_t node) -> int32_t {
        const rt_node_record& r = bvh.nodes[size_t(node)];
        return r.is_leaf ? ~r.a : innerIndex[size_t(node)];
    };
    out.nodes.resize(size_t(inners));
    for (size_t i = 0; i < n; i++) {
        const rt_node_record& r = bvh.nodes[i];
        if (r.is_leaf) continue;
        NodeRec& d = out.nodes[size_t(innerIndex[i])];
        const rt_node_record& L = bvh.nodes[size_t(r.a)];
        const rt_node_record& R = bvh.nodes[size_t(r.b)];
        for (int ax = 0; ax < 3; ax++) {
            d.boxL[2 * ax] = L.lower[ax]; d.boxL[2 * ax + 1] = L.upper[ax];
            d.boxR[2 * ax] = R.lower[ax]; d.boxR[2 * ax + 1] = R.upper[ax];
        }
        d.refL = refOf(r.a);
        d.refR = refOf(r.b);
        d.level = r.level;
        d.refNode = int32_t(i);
    }
    if (n) {
        std::memcpy(out.rootLo, bvh.nodes[0].lower, 12);
        std::memcpy(out.rootHi, bvh.nodes[0].upper, 12);
        out.rootRef = refOf(0);
    }
    // primitive and attribute records, in leaf order; `more` from the owning leaf
    out.prims.resize(bvh.prims.size());
    out.attrs.resize(bvh.prims.size());
    for (size_t i = 0; i < n; i++) {
        const rt_node_record& r = bvh.nodes[i];
        if (!r.is_leaf) continue;
        for (int32_t k = 0; k < r.b; k++) {
            const size_t slot = size_t(r.a + k);
            const rt_prim_ref& pr = bvh.prims[slot];
            PrimRec& p = out.prims[slot];
            AttrRec& a = out.attrs[slot];
            std::memset(&p, 0, sizeof p);
            std::memset(&a, 0, sizeof a);
            p.more = r.b - 1 - k;
            p.mesh = pr.mesh;
            p.prim = pr.triangle;
            if (pr.mesh < 0) {
                const rt_sphere_desc& s = scene.spheres[size_t(pr.triangle)];
                std::memcpy(p.v0, s.center, 12);
                p.v1[0] = s.radius;
                continue;
            }
            const HostMesh& mesh = scene.meshes[size_t(pr.mesh)];
            const float* va = &mesh.vertices[size_t(mesh.triangles[size_t(3 * pr.triangle)]) * 8];
            const float* vb = &mesh.vertices[size_t(mesh.triangles[size_t(3 * pr.triangle + 1)]) * 8];
            const float* vc = &mesh.vertices[size_t(mesh.triangles[size_t(3 * pr.triangle + 2)]) * 8];
            const vec3 v0 = load3(va), v1 = load3(vb), v2 = load3(vc);
            std::memcpy(p.v0, va, 12); std::memcpy(p.v1, vb, 12); std::memcpy(p.v2, vc, 12);
            // trianglePlane, reference src/intersect.cpp:52-60
            const vec3 nn = cross(v1 - v0, v2 - v0);
            const vec3 normal = is_zero(length(nn)) ? make3(1.0f, 0.0f, 0.0f) : normalize(nn);
            p.n[0] = normal.x; p.n[1] = normal.y; p.n[2] = normal.z;
            // pointInTriangle's totalArea, reference src/intersect.cpp:26
            p.invArea = 1.0f / length(cross(v0 - v2, v1 - v2));
            std::memcpy(a.n0, va + 3, 12); std::memcpy(a.n1, vb + 3, 12); std::memcpy(a.n2, vc + 3, 12);
            std::memcpy(a.uv0, va + 6, 8); std::memcpy(a.uv1, vb + 6, 8); std::memcpy(a.uv2, vc + 6, 8);
        }
    }
    // brute-force visiting order: meshes in order, triangles in order, then spheres
    std::vector<size_t> meshBase(scene.meshes.size() + 1, 0);
    for (size_t m = 0; m < scene.meshes.size(); m++) meshBase[m + 1] = meshBase[m] + scene.meshes[m].triangles.size() / 3;
    const size_t sphereBase = meshBase.back();
    out.primOrder.assign(bvh.prims.size(), 0);
    for (size_t slot = 0; slot < bvh.prims.size(); slot++) {
        const rt_prim_ref& pr = bvh.prims[slot];
        const size_t rank = pr.mesh < 0 ? sphereBase + size_t(pr.triangle) : meshBase[size_t(pr.mesh)] + size_t(pr.triangle);
        out.primOrder[rank] = uint32_t(slot);
    }
}

} // namespace rt
