// TEST INFRASTRUCTURE — the oracle.  Nothing in the product may include, link or call this file; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
//
// A scalar CPU restatement of the reference ray tracer's per-pixel render loop, written from the
// reference's algorithm (file:line citations are relative to the reference checkout) with the same fp32
// operation order as the reference's vendored glm 0.9.9.9, so that deterministic results are bit-identical
// to the compiled reference.  Parity status: PINNED — tests/test_oracle_golden.py checks every function
// here against the fixtures under tests/golden/, which oracle/gen_golden.py produced by running the
// reference's own translation units (oracle/ref_harness.cpp explains what was and was not reference code).
//
// Compile with -ffp-contract=off and without -ffast-math (oracle/Makefile).
#include "rt_amd.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------------
// Vector arithmetic in glm's operation order (glm/detail/func_geometric.inl:48-55,68-79,84-89,104-110).
// ---------------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float a) { return { a, a, a }; }
inline V3 v3(const float* p) { return { p[0], p[1], p[2] }; }
inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
inline V3 operator*(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
inline V3 operator/(V3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 x, V3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float distance(V3 a, V3 b) { return length(b - a); }
inline V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }
inline float gmax(float a, float b) { return (a < b) ? b : a; } // glm::max
inline float gmin(float a, float b) { return (b < a) ? b : a; } // glm::min
inline bool isZero(float a) { return std::fabs(a) <= 0.000001f; } // intersect.cpp:17-20

struct Ray {
    V3 o, d;
    float t;
};
struct Material {
    V3 kd, ks;
    float shininess, transparency;
    int texture;
};
struct HitInfo {
    V3 normal { 0, 0, 0 }, bary { 0, 0, 0 };
    Material material { { 0, 0, 0 }, { 0, 0, 0 }, 1.0f, 1.0f, -1 };
};
struct Vertex {
    V3 p, n;
    float u, v;
};
struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;
    V3 texel(int idx) const
    {
        long n = long(w) * long(h);
        if (idx < 0) idx = 0; // the reference indexes out of bounds here (undefined behaviour); clamp instead
        if (idx >= n) idx = int(n - 1);
        const uint8_t* p = &rgb[size_t(idx) * 3];
        return { p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f }; // framework/src/image.cpp:31
    }
};
struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<uint32_t> tris; // 3 per triangle
    Material material;
};
struct Sphere {
    V3 c;
    float r;
    Material material;
};
struct Node {
    bool leaf;
    int level;
    V3 lo, hi;
    std::vector<int> ids; // inner: 2 child indices; leaf: (triangle, mesh) pairs   (bounding_volume_hierarchy.h:15-23)
};
struct Prim {
    V3 mn, mx, centr;
    int t_id, m_id;
};

struct Counters {
    uint64_t primary = 0, shadow = 0, bounce = 0, probe = 0, box = 0, tri = 0, overflow = 0;
    uint64_t shadowBox = 0, shadowTri = 0; // the part of box / tri spent inside shadow queries
    // ... and the part of those spent on queries whose light sample the surface faces (Phong factor not exactly zero):
    // the only shadow queries whose outcome can reach the pixel
    uint64_t litShadow = 0, litShadowBox = 0, litShadowTri = 0;
    void add(const Counters& o)
    {
        primary += o.primary; shadow += o.shadow; bounce += o.bounce; probe += o.probe; box += o.box; tri += o.tri; overflow += o.overflow;
        shadowBox += o.shadowBox; shadowTri += o.shadowTri;
        litShadow += o.litShadow; litShadowBox += o.litShadowBox; litShadowTri += o.litShadowTri;
    }
};

} // namespace

struct orc_scene {
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Image> textures;
    std::vector<Image> skybox;
    // hierarchy
    std::vector<Node> nodes;
    int numLevels = 0, numLeaves = 0, maxLevel = 32;
};

namespace {

// ---------------------------------------------------------------------------------------------------
// Counter-based random stream (specification shared with the HIP kernels and oracle/ref_harness.cpp).
// ---------------------------------------------------------------------------------------------------
inline uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
inline uint32_t rng_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t index)
{
    uint32_t h = hash32(seed);
    h = hash32(h ^ pixel);
    h = hash32(h ^ (sample * 0x9e3779b9U + stream));
    return hash32(h ^ index);
}
enum { STREAM_RAND = 0, STREAM_DOF = 1, STREAM_GLOSS = 2 };
inline double u01(uint32_t u) { return (double(u) + 0.5) * (1.0 / 4294967296.0); }
inline uint32_t floatBits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
// Key of one glossy-reflection call: a hash of the bits of the reflection ray it perturbs.  Sample i of the call uses
// the Gaussian deviates key + 2i and key + 2i + 1 of the glossy stream.
inline uint32_t rayKey6(float ox, float oy, float oz, float dx, float dy, float dz)
{
    uint32_t h = hash32(floatBits(ox) ^ 0x9e3779b9U);
    h = hash32(h ^ floatBits(oy));
    h = hash32(h ^ floatBits(oz));
    h = hash32(h ^ floatBits(dx));
    h = hash32(h ^ floatBits(dy));
    return hash32(h ^ floatBits(dz));
}

struct Rng {
    uint32_t seed = 1, pixel = 0, sample = 0, randCounter = 0, glossCall = 0;
    int rand31() { return int(rng_u32(seed, pixel, sample, STREAM_RAND, randCounter++) >> 1); } // stands in for rand()
    double gauss(uint32_t idx) const
    {
        uint32_t a = rng_u32(seed, pixel, sample, STREAM_GLOSS, 2u * idx);
        uint32_t b = rng_u32(seed, pixel, sample, STREAM_GLOSS, 2u * idx + 1u);
        return std::sqrt(-2.0 * std::log(u01(a))) * std::cos(6.283185307179586476925 * u01(b));
    }
};

// ---------------------------------------------------------------------------------------------------
// Intersection tests.
// ---------------------------------------------------------------------------------------------------
// intersect.cpp:111-140
bool intersectAABB(V3 lo, V3 hi, Ray& ray)
{
    float inv[3] = { 1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z };
    float ts[6];
    for (int i = 0; i < 3; i++) {
        bool dir = ray.d[i] > 0;
        ts[2 * i] = isZero(ray.d[i]) ? -FLT_MAX : ((dir ? lo[i] : hi[i]) - ray.o[i]) * inv[i];
        ts[2 * i + 1] = isZero(ray.d[i]) ? FLT_MAX : ((dir ? hi[i] : lo[i]) - ray.o[i]) * inv[i];
    }
    float t_in = gmax(gmax(ts[0], ts[2]), ts[4]);
    float t_out = gmin(gmin(ts[1], ts[3]), ts[5]);
    if (t_in > t_out || t_out <= 0 || isZero(t_in))
        return false;
    float newT = t_in < 0 ? t_out : t_in;
    bool valid = newT < ray.t;
    ray.t = valid ? newT : ray.t;
    return valid;
}

// interpolate.cpp:4-11
V3 computeBarycentricCoord(V3 v0, V3 v1, V3 v2, V3 p)
{
    float totalArea = length(cross(v0 - v2, v1 - v2)) / 2;
    float alpha = length(cross(p - v2, v1 - v2)) / (2 * totalArea);
    float beta = length(cross(p - v2, v0 - v2)) / (2 * totalArea);
    float gamma = length(cross(p - v1, v0 - v1)) / (2 * totalArea);
    return { alpha, beta, gamma };
}
// interpolate.cpp:13-18
V3 interpolateNormal(V3 n0, V3 n1, V3 n2, V3 b) { return normalize(b.x * n0 + b.y * n1 + b.z * n2); }

// intersect.cpp:22-38
bool pointInTriangle(V3 v0, V3 v1, V3 v2, V3 n, V3 p)
{
    if (!isZero(dot(n, p - v0)))
        return false;
    float totalArea = 1.0f / length(cross(v0 - v2, v1 - v2));
    float alpha = length(cross(p - v2, v1 - v2)) * totalArea;
    float beta = length(cross(p - v2, v0 - v2)) * totalArea;
    float gamma = length(cross(p - v1, v0 - v1)) * totalArea;
    if (alpha < 0 || beta < 0 || alpha + beta > 1.0f || alpha + gamma > 1.0f || beta + gamma > 1.0f)
        return false;
    return true;
}

// intersect.cpp:40-80
bool intersectTriangle(V3 v0, V3 v1, V3 v2, Ray& ray, HitInfo& hit)
{
    V3 n = cross(v1 - v0, v2 - v0);
    V3 normal = isZero(length(n)) ? V3 { 1.0f, 0.0f, 0.0f } : normalize(n);
    float D = dot(normal, v0);
    float oldT = ray.t;
    if (isZero(dot(normal, ray.d)))
        return false;
    float t = (D - dot(ray.o, normal)) / dot(ray.d, normal);
    if (!(t > 0))
        return false;
    ray.t = t;
    bool pit = pointInTriangle(v0, v1, v2, normal, ray.o + ray.d * ray.t) && ray.t < oldT;
    ray.t = pit ? ray.t : oldT;
    if (pit) {
        hit.normal = normal;
        V3 point = ray.o + ray.d * ray.t;
        hit.bary = computeBarycentricCoord(v0, v1, v2, point);
    }
    return pit;
}

// intersect.cpp:84-107
bool intersectSphere(const Sphere& s, Ray& ray, HitInfo& hit)
{
    // The unqualified pow(...) and sqrt(...) of the reference resolve to the C double functions, so the
    // expressions they appear in are evaluated in double and narrowed on assignment.
    float a = dot(ray.d, ray.d);
    float b = 2.0f * dot(ray.d, ray.o - s.c);
    float c = float(double(dot(ray.o - s.c, ray.o - s.c)) - std::pow(double(s.r), double(2.0f)));
    float D = b * b - 4 * a * c;
    if (D < 0 || isZero(c))
        return false;
    float t1 = float((double(-b) - std::sqrt(double(D))) / double(2 * a));
    float t2 = float((double(-b) + std::sqrt(double(D))) / double(2 * a));
    if (t1 < 0 && t2 < 0)
        return false;
    float newT = t1 * t2 < 0 ? std::fmax(t1, t2) : std::fmin(t1, t2);
    bool valid = newT < ray.t;
    ray.t = valid ? newT : ray.t;
    hit.normal = -s.c + (ray.d * ray.t + ray.o);
    hit.material = s.material;
    return valid;
}

// texture.cpp:5-17
V3 acquireTexel(const Image& img, float u, float v)
{
    float i = float(std::fmod(std::round(double(u * float(img.w)) - 0.5), double(img.w)));
    float j = float(std::fmod(std::round(double((1 - v) * float(img.h)) - 0.5), double(img.h)));
    int idx = int(j * float(img.w) + i);
    return img.texel(idx);
}
// Bilinear lookup.  The reference declares the switch (common.h:59, enableBilinearTextureFiltering) but its
// acquireTexel ignores it (texture.cpp:5-17), so there is no reference behaviour to match: PARITY UNPINNED.  Defined
// here as the natural extension of the nearest lookup: same (1 - v) flip, texel centres at +0.5 (the nearest lookup
// rounds u*W - 0.5, this one interpolates at u*W - 0.5), repeat wrap, fp32 throughout, blended as
// (1-fy)*((1-fx)*T00 + fx*T10) + fy*((1-fx)*T01 + fx*T11).
V3 acquireTexelBilinear(const Image& img, float u, float v)
{
    const float x = u * float(img.w) - 0.5f, y = (1 - v) * float(img.h) - 0.5f;
    const float x0 = std::floor(x), y0 = std::floor(y);
    const float fx = x - x0, fy = y - y0;
    auto wrap = [](float c, int n) {
        float m = std::fmod(c, float(n));
        if (m < 0) m += float(n);
        int i = int(m);
        return i >= n ? n - 1 : i; // m + n can round up to n for a tiny negative m
    };
    const int i0 = wrap(x0, img.w), i1 = wrap(x0 + 1.0f, img.w), j0 = wrap(y0, img.h), j1 = wrap(y0 + 1.0f, img.h);
    const V3 t00 = img.texel(j0 * img.w + i0), t10 = img.texel(j0 * img.w + i1);
    const V3 t01 = img.texel(j1 * img.w + i0), t11 = img.texel(j1 * img.w + i1);
    const V3 top = t00 * (1 - fx) + t10 * fx, bottom = t01 * (1 - fx) + t11 * fx;
    return top * (1 - fy) + bottom * fy;
}
// environment_mapping.cpp:44-49
V3 acquireTexelClamp(const Image& img, float u, float v)
{
    int i = int(u * float(img.w));
    int j = int(std::clamp((1.0f - v) * float(img.h), 0.0f, float(img.h) - 1.0f));
    return img.texel(j * img.w + i);
}
// environment_mapping.cpp:9-42
V3 environment_lookup(V3 v)
{
    V3 a { std::fabs(v.x), std::fabs(v.y), std::fabs(v.z) };
    float maxA, ut, vt, idx;
    if (a.x >= a.y && a.x >= a.z) {
        maxA = 2 * a.x;
        ut = (v.x >= 0) ? -v.z : v.z;
        vt = -v.y;
        idx = float(v.x < 0);
    } else if (a.y >= a.z) {
        maxA = 2 * a.y;
        ut = v.x;
        vt = (v.y >= 0) ? v.z : -v.z;
        idx = float((v.y < 0) + 2);
    } else {
        maxA = 2 * a.z;
        ut = (v.z >= 0) ? v.x : -v.x;
        vt = -v.y;
        idx = float((v.z < 0) + 4);
    }
    if (idx <= 1)
        idx = 1 - idx;
    return { float(double(-(ut / maxA)) + 0.5), float(double(-(vt / maxA)) + 0.5), idx };
}

// ---------------------------------------------------------------------------------------------------
// BVH build, bounding_volume_hierarchy.cpp:21-208.
// ---------------------------------------------------------------------------------------------------
void calcAABB(const std::vector<Prim>& prims, const std::vector<int>& ids, int start, int end, V3& lo, V3& hi)
{
    lo = v3(FLT_MAX);
    hi = v3(-FLT_MAX);
    size_t stop = end == -1 ? ids.size() : size_t(end);
    for (size_t i = size_t(start); i < stop; i++) {
        const Prim& p = prims[size_t(ids[i])];
        lo = { std::fmin(lo.x, p.mn.x), std::fmin(lo.y, p.mn.y), std::fmin(lo.z, p.mn.z) };
        hi = { std::fmax(hi.x, p.mx.x), std::fmax(hi.y, p.mx.y), std::fmax(hi.z, p.mx.z) };
    }
}
float surfaceArea(V3 lo, V3 hi)
{
    V3 b = hi - lo;
    V3 b2 { b.y, b.z, b.x };
    return dot(hi - lo, b2);
}

void buildNode(orc_scene& sc, const std::vector<Prim>& prims, std::vector<int> ids, int level, int parent, int idx, bool sah, int sahBins)
{
    sc.numLevels = std::max(sc.numLevels, level + 1);
    Node cur;
    cur.level = level;
    calcAABB(prims, ids, 0, -1, cur.lo, cur.hi);
    if (level == sc.maxLevel || ids.size() <= 1) {
        cur.leaf = true;
        for (int i : ids) {
            cur.ids.push_back(prims[size_t(i)].t_id);
            cur.ids.push_back(prims[size_t(i)].m_id);
        }
        sc.numLeaves++;
        sc.nodes.push_back(cur);
        if (parent != -1)
            sc.nodes[size_t(parent)].ids.push_back(idx);
        return;
    }
    cur.leaf = false;
    int split = 0;
    if (sah) {
        float minCost = FLT_MAX;
        float rec = 1.0f / surfaceArea(cur.lo, cur.hi);
        int axis = -1;
        for (int a = 0; a < 3; a++) {
            std::sort(ids.begin(), ids.end(), [&](int i, int j) { return prims[size_t(i)].centr[a] < prims[size_t(j)].centr[a]; });
            if (size_t(sahBins) >= ids.size()) {
                for (int i = 0; size_t(i) <= ids.size(); i++) {
                    V3 llo, lhi, rlo, rhi;
                    calcAABB(prims, ids, 0, i, llo, lhi);
                    calcAABB(prims, ids, i, -1, rlo, rhi);
                    float cost = (surfaceArea(llo, lhi) * float(i) + surfaceArea(rlo, rhi) * float(ids.size() - size_t(i))) * rec;
                    if (cost < minCost) { minCost = cost; axis = a; split = i; }
                }
            } else {
                float range = prims[size_t(ids[ids.size() - 1])].centr[a] - prims[size_t(ids[0])].centr[a];
                float dist = range / float(sahBins);
                int countLeft = 1;
                for (int i = 0; i < sahBins; i++) {
                    float cs = prims[size_t(ids[0])].centr[a] + float(i) * dist;
                    while (prims[size_t(ids[size_t(countLeft)])].centr[a] < cs)
                        countLeft++;
                    V3 llo, lhi, rlo, rhi;
                    calcAABB(prims, ids, 0, countLeft, llo, lhi);
                    calcAABB(prims, ids, countLeft, -1, rlo, rhi);
                    float cost = (surfaceArea(llo, lhi) * float(countLeft) + surfaceArea(rlo, rhi) * float(ids.size() - size_t(countLeft))) * rec;
                    if (cost < minCost) { minCost = cost; axis = a; split = countLeft; }
                }
            }
        }
        if (axis < 0) axis = 0; // the reference would index centr[-1] here (undefined behaviour)
        std::sort(ids.begin(), ids.end(), [&](int i, int j) { return prims[size_t(i)].centr[axis] < prims[size_t(j)].centr[axis]; });
    } else {
        int a = level % 3;
        std::sort(ids.begin(), ids.end(), [&](int i, int j) { return prims[size_t(i)].centr[a] < prims[size_t(j)].centr[a]; });
        split = int(ids.size() / 2);
    }
    sc.nodes.push_back(cur);
    if (parent != -1)
        sc.nodes[size_t(parent)].ids.push_back(idx);
    buildNode(sc, prims, std::vector<int>(ids.begin(), ids.begin() + split), level + 1, idx, int(sc.nodes.size()), sah, sahBins);
    buildNode(sc, prims, std::vector<int>(ids.begin() + split, ids.end()), level + 1, idx, int(sc.nodes.size()), sah, sahBins);
}

void buildBvh(orc_scene& sc, bool sah, int maxLevel, int sahBins)
{
    sc.nodes.clear();
    sc.numLevels = 0;
    sc.numLeaves = 0;
    sc.maxLevel = maxLevel;
    std::vector<Prim> prims;
    for (size_t i = 0; i < sc.meshes.size(); i++) {
        const Mesh& m = sc.meshes[i];
        for (size_t j = 0; j < m.tris.size() / 3; j++) {
            V3 v0 = m.vertices[m.tris[3 * j]].p, v1 = m.vertices[m.tris[3 * j + 1]].p, v2 = m.vertices[m.tris[3 * j + 2]].p;
            Prim p;
            p.mn = { std::fmin(std::fmin(v0.x, v1.x), v2.x), std::fmin(std::fmin(v0.y, v1.y), v2.y), std::fmin(std::fmin(v0.z, v1.z), v2.z) };
            p.mx = { std::fmax(std::fmax(v0.x, v1.x), v2.x), std::fmax(std::fmax(v0.y, v1.y), v2.y), std::fmax(std::fmax(v0.z, v1.z), v2.z) };
            V3 s = (v0 + v1) + v2;
            p.centr = { s.x / 3.0f, s.y / 3.0f, s.z / 3.0f };
            p.t_id = int(j);
            p.m_id = int(i);
            prims.push_back(p);
        }
    }
    for (size_t i = 0; i < sc.spheres.size(); i++) {
        const Sphere& s = sc.spheres[i];
        prims.push_back({ s.c - v3(s.r), s.c + v3(s.r), s.c, int(i), -1 });
    }
    std::vector<int> ids(prims.size());
    std::iota(ids.begin(), ids.end(), 0);
    buildNode(sc, prims, ids, 0, -1, 0, sah, sahBins);
}

// ---------------------------------------------------------------------------------------------------
// Traversal, bounding_volume_hierarchy.cpp:293-548.
// ---------------------------------------------------------------------------------------------------
struct Trav {
    const orc_scene& sc;
    const rt_features& f;
    Counters& cnt;
    bool hit = false, sphereInt = false;
    float absoluteT = FLT_MAX;
    int finalMesh = -1, finalTri = -1;
};

void applyHitAttributes(const orc_scene& sc, const rt_features& f, const Mesh& mesh, int tri, const Ray& ray, HitInfo& h)
{
    // material (+ texel) and interpolated normal of an accepted triangle hit, …hierarchy.cpp:351-396 / 477-523
    const Vertex& a = mesh.vertices[mesh.tris[size_t(3 * tri)]];
    const Vertex& b = mesh.vertices[mesh.tris[size_t(3 * tri + 1)]];
    const Vertex& c = mesh.vertices[mesh.tris[size_t(3 * tri + 2)]];
    h.material = mesh.material;
    if (f.texture_mapping && mesh.material.texture >= 0) {
        float tu = a.u * h.bary.x + b.u * h.bary.y + c.u * h.bary.z; // interpolate.cpp:20-28
        float tv = a.v * h.bary.x + b.v * h.bary.y + c.v * h.bary.z;
        const Image& img = sc.textures[size_t(mesh.material.texture)];
        h.material.kd = f.bilinear_filtering ? acquireTexelBilinear(img, tu, tv) : acquireTexel(img, tu, tv);
    }
    if (f.normal_interp) {
        V3 point = ray.o + ray.d * ray.t;
        V3 bary = computeBarycentricCoord(a.p, b.p, c.p, point);
        h.normal = interpolateNormal(a.n, b.n, c.n, bary);
    }
}

void traverse(Trav& T, HitInfo& h, Ray& ray, int nodeIdx)
{
    const Node& node = T.sc.nodes[size_t(nodeIdx)];
    float oldT = ray.t;
    if (node.level == 0) {
        T.cnt.box++;
        bool ok = intersectAABB(node.lo, node.hi, ray);
        ray.t = oldT;
        if (!ok) return;
    }
    if (!T.f.recursive && !T.f.transparency) {
        ray.t = FLT_MAX;
        if (intersectAABB(node.lo, node.hi, ray)) {
            if (ray.t > T.absoluteT) {
                ray.t = oldT;
                return;
            }
            ray.t = oldT;
        }
        // (a failed test would leave ray.t = FLT_MAX in the reference; it cannot fail here because the
        //  parent — or the root gate above — already accepted this box with a bound at least as tight)
        ray.t = oldT;
    }
    if (node.leaf) {
        for (size_t i = 0; i < node.ids.size(); i += 2) {
            int tri = node.ids[i], meshId = node.ids[i + 1];
            if (meshId == -1) {
                T.cnt.tri++;
                if (intersectSphere(T.sc.spheres[size_t(tri)], ray, h)) {
                    h.material = T.sc.spheres[size_t(tri)].material;
                    T.hit = true;
                    T.sphereInt = true;
                    T.finalMesh = -1;
                    T.finalTri = tri;
                }
            } else {
                const Mesh& mesh = T.sc.meshes[size_t(meshId)];
                const V3 v0 = mesh.vertices[mesh.tris[size_t(3 * tri)]].p;
                const V3 v1 = mesh.vertices[mesh.tris[size_t(3 * tri + 1)]].p;
                const V3 v2 = mesh.vertices[mesh.tris[size_t(3 * tri + 2)]].p;
                T.cnt.tri++;
                if (intersectTriangle(v0, v1, v2, ray, h)) {
                    if (ray.t < T.absoluteT) T.absoluteT = ray.t;
                    T.finalMesh = meshId;
                    T.finalTri = tri;
                    T.hit = true;
                    applyHitAttributes(T.sc, T.f, mesh, tri, ray, h);
                }
            }
        }
        return;
    }
    int left = node.ids[0], right = node.ids[1];
    const Node& L = T.sc.nodes[size_t(left)];
    const Node& R = T.sc.nodes[size_t(right)];
    bool iL = false, iR = false;
    float tL = 0, tR = 0;
    ray.t = FLT_MAX;
    T.cnt.box++;
    if (intersectAABB(L.lo, L.hi, ray)) { tL = ray.t; iL = true; }
    ray.t = FLT_MAX;
    T.cnt.box++;
    if (intersectAABB(R.lo, R.hi, ray)) { tR = ray.t; iR = true; }
    ray.t = oldT;
    if (iL && iR) {
        if (tL < tR) { traverse(T, h, ray, left); traverse(T, h, ray, right); }
        else { traverse(T, h, ray, right); traverse(T, h, ray, left); }
    } else if (iR) traverse(T, h, ray, right);
    else if (iL) traverse(T, h, ray, left);
}

// BoundingVolumeHierarchy::intersect, …hierarchy.cpp:459-548
bool intersectScene(const orc_scene& sc, Ray& ray, HitInfo& h, const rt_features& f, Counters& cnt, int* meshOut = nullptr, int* triOut = nullptr)
{
    if (!f.accel_structure) {
        bool hit = false;
        int fm = -1, ft = -1;
        for (size_t m = 0; m < sc.meshes.size(); m++) {
            const Mesh& mesh = sc.meshes[m];
            for (size_t t = 0; t < mesh.tris.size() / 3; t++) {
                const V3 v0 = mesh.vertices[mesh.tris[3 * t]].p, v1 = mesh.vertices[mesh.tris[3 * t + 1]].p, v2 = mesh.vertices[mesh.tris[3 * t + 2]].p;
                cnt.tri++;
                if (intersectTriangle(v0, v1, v2, ray, h)) {
                    hit = true; fm = int(m); ft = int(t);
                    applyHitAttributes(sc, f, mesh, int(t), ray, h);
                }
            }
        }
        for (size_t s = 0; s < sc.spheres.size(); s++) {
            cnt.tri++;
            if (intersectSphere(sc.spheres[s], ray, h)) { hit = true; fm = -1; ft = int(s); }
        }
        if (meshOut) *meshOut = fm;
        if (triOut) *triOut = ft;
        return hit;
    }
    if (sc.nodes.empty()) return false;
    Trav T { sc, f, cnt };
    traverse(T, h, ray, 0);
    if (meshOut) *meshOut = T.finalMesh;
    if (triOut) *triOut = T.finalTri;
    return T.hit;
}

// ---------------------------------------------------------------------------------------------------
// Shading and lights.
// ---------------------------------------------------------------------------------------------------
// shading.cpp:6-39
V3 computeShading(V3 lightPos, V3 lightColor, const Ray& ray, const HitInfo& h)
{
    V3 P = ray.o + (ray.d * ray.t);
    V3 N = normalize(h.normal);
    V3 L = normalize(lightPos - P);
    float lambertian = dot(N, L) > 0 ? dot(N, L) : 0;
    V3 diffuse = lightColor * h.material.kd * lambertian;
    float specular = 0.0f;
    if (lambertian > 0) {
        V3 R = reflect(L, N);
        V3 V = normalize(-ray.o + P);
        float specAngle = dot(R, V) > 0 ? dot(R, V) : 0;
        specular = float(std::pow(double(specAngle), double(h.material.shininess))); // C pow(double,double), shading.cpp:31
    }
    V3 spec = lightColor * h.material.ks * specular;
    return spec + diffuse;
}
// shading.cpp:42-51
Ray computeReflectionRay(const Ray& ray, const HitInfo& h)
{
    Ray r;
    float epsilon = 0.00001f;
    r.d = ray.d - (2 * dot(ray.d, h.normal)) * h.normal;
    r.o = ray.t * ray.d * (1 - epsilon) + ray.o;
    r.t = FLT_MAX;
    return r;
}

struct Ctx {
    const orc_scene& sc;
    const rt_light* lights;
    int nLights;
    const rt_features& f;
    const rt_params& p;
    Counters cnt;
    Rng rng;
};

// light.cpp:68-89
float testVisibility(Ctx& C, V3 samplePos, const Ray& ray, const HitInfo& h)
{
    float epsilon = 0.00001f;
    if (C.f.hard_shadow || C.f.soft_shadow) {
        V3 P = ray.o + (1 - epsilon) * ray.t * ray.d;
        Ray light { P, samplePos - P, 1.0f };
        HitInfo copy = h;
        C.cnt.shadow++;
        const uint64_t box0 = C.cnt.box, tri0 = C.cnt.tri;
        const bool blocked = intersectScene(C.sc, light, copy, C.f, C.cnt);
        C.cnt.shadowBox += C.cnt.box - box0;
        C.cnt.shadowTri += C.cnt.tri - tri0;
        // bookkeeping only: does the surface face this light sample (the test computeShading makes, shading.cpp:14-21)
        const V3 point = ray.o + (ray.d * ray.t);
        if (!C.f.shading || dot(normalize(h.normal), normalize(samplePos - point)) > 0) {
            C.cnt.litShadow++;
            C.cnt.litShadowBox += C.cnt.box - box0;
            C.cnt.litShadowTri += C.cnt.tri - tri0;
        }
        if (blocked)
            return 0.0f;
    }
    return 1.0f;
}

// light.cpp:124-220 (with sampleSegmentLight :22-43 and sampleParallelogramLight :47-64 inlined)
V3 computeLightContribution(Ctx& C, const Ray& ray, const HitInfo& h)
{
    float lighted;
    V3 shading = v3(0.0f);
    for (int li = 0; li < C.nLights; li++) {
        const rt_light& L = C.lights[li];
        if (L.type == RT_LIGHT_POINT) {
            V3 pos = v3(L.v), col = v3(L.v + 3);
            lighted = C.f.hard_shadow ? testVisibility(C, pos, ray, h) : 1.0f;
            V3 shade = C.f.shading ? computeShading(pos, col, ray, h) : h.material.kd;
            shading = shading + shade * lighted;
        } else if (L.type == RT_LIGHT_SEGMENT) {
            if (C.f.soft_shadow || C.f.shading) {
                V3 e0 = v3(L.v), e1 = v3(L.v + 3), c0 = v3(L.v + 6), c1 = v3(L.v + 9);
                float segmentLengthRec = 1.0f / distance(e0, e1);
                float sampleCountRec = 1.0f / std::floor(std::fmax(float(C.p.samples_per_unit) / segmentLengthRec, 1.0f));
                V3 sampleDistance = (e1 - e0) * sampleCountRec;
                float sampleCount = 1.0f / sampleCountRec;
                for (int i = 0; float(i) < sampleCount; i++) {
                    float randomUniform = float(C.rng.rand31() % 50) / 50.0f;
                    V3 randomAddition = sampleDistance * randomUniform;
                    float alpha = float(i) * sampleCountRec;
                    V3 position = (e0 * (1 - alpha) + e1 * alpha) + randomAddition;
                    float p0 = 0.5f, p1 = 0.5f;
                    if (segmentLengthRec != 0) {
                        p1 = distance(e0, position) * segmentLengthRec;
                        p0 = 1 - p1;
                    }
                    V3 color = p0 * c0 + p1 * c1;
                    lighted = C.f.soft_shadow ? testVisibility(C, position, ray, h) : 1.0f;
                    V3 shade = C.f.shading ? computeShading(position, color, ray, h) * sampleCountRec : h.material.kd * sampleCountRec;
                    shading = shading + shade * lighted;
                }
            }
        } else {
            if (C.f.soft_shadow || C.f.shading) {
                V3 v0 = v3(L.v), e01 = v3(L.v + 3), e02 = v3(L.v + 6), c0 = v3(L.v + 9), c1 = v3(L.v + 12), c2 = v3(L.v + 15), c3 = v3(L.v + 18);
                float recI = 1.0f / std::floor(std::fmax(float(C.p.samples_per_unit_parallel) * length(e01), 1.0f));
                float recJ = 1.0f / std::floor(std::fmax(float(C.p.samples_per_unit_parallel) * length(e02), 1.0f));
                float countI = 1.0f / recI, countJ = 1.0f / recJ;
                for (int i = 0; float(i) < countI; i++)
                    for (int j = 0; float(j) < countJ; j++) {
                        float ri = float(C.rng.rand31() % 50) / 50.0f + float(i);
                        float rj = float(C.rng.rand31() % 50) / 50.0f + float(j);
                        float x = ri * recI, y = rj * recJ;
                        V3 position = (v0 + e01 * x) + e02 * y;
                        V3 color = (1 - y) * (x * c1 + (1 - x) * c0) + y * (x * c3 + (1 - x) * c2);
                        lighted = C.f.soft_shadow ? testVisibility(C, position, ray, h) : 1.0f;
                        V3 shade = C.f.shading ? computeShading(position, color, ray, h) * recI * recJ : h.material.kd * recI * recJ;
                        shading = shading + shade * lighted;
                    }
            }
        }
    }
    return shading;
}

V3 skyboxColor(const orc_scene& sc, V3 dir)
{
    V3 l = environment_lookup(normalize(dir));
    size_t face = size_t(l.z);
    if (sc.skybox.size() != 6) return v3(0.0f); // the reference would index an empty vector (scene.cpp:121-130)
    return acquireTexelClamp(sc.skybox[face], l.x, l.y);
}

// gloss.cpp:14-51
int glossySamples(Ctx& C, const HitInfo& h, const Ray& ref, std::vector<Ray>& out)
{
    V3 w = normalize(ref.d);
    int minTerm = std::fabs(w.x) <= std::fabs(w.y) && std::fabs(w.x) <= std::fabs(w.z) ? 0 : (std::fabs(w.y) <= std::fabs(w.x) && std::fabs(w.y) <= std::fabs(w.z) ? 1 : 2);
    V3 t = w;
    if (minTerm == 0) t.x = 1; else if (minTerm == 1) t.y = 1; else t.z = 1;
    V3 u = normalize(cross(t, w));
    V3 v = cross(w, u);
    float side = C.p.glossy_sigma / (h.material.shininess * 3.0f);
    float offset = -side / 2.0f;
    // the call's deviates are indexed from a hash of the reflection ray's bits (see rayKey6)
    uint32_t call = rayKey6(ref.o.x, ref.o.y, ref.o.z, ref.d.x, ref.d.y, ref.d.z);
    for (int i = 0; i < C.p.glossy_filter_size; i++) {
        uint32_t k = call + 2u * uint32_t(i);
        double n0 = C.rng.gauss(k), n1 = C.rng.gauss(k + 1u);
        V3 rPrime = (ref.d + float(double(offset) + n0 * double(side)) * u) + float(double(offset) + n1 * double(side)) * v;
        if (dot(rPrime, h.normal) > 0)
            out.push_back({ ref.o, normalize(rPrime), FLT_MAX });
    }
    return int(out.size());
}

// render.cpp:18-80
V3 getFinalColor(Ctx& C, Ray ray, int rayDepth, int transparencyDepth = 0)
{
    HitInfo h;
    if (intersectScene(C.sc, ray, h, C.f, C.cnt)) {
        V3 Lo = computeLightContribution(C, ray, h);
        if (C.f.environment_mapping && C.p.enabled_refl_map)
            return skyboxColor(C.sc, ray.d);
        if (C.f.transparency) {
            if (h.material.transparency != 1.0f) {
                if (transparencyDepth >= C.p.max_transparency_depth) {
                    C.cnt.overflow++;
                    return Lo;
                }
                Ray next { (ray.t + 0.000001f) * ray.d + ray.o, ray.d, FLT_MAX };
                C.cnt.bounce++;
                return h.material.transparency * Lo + (1.0f - h.material.transparency) * getFinalColor(C, next, rayDepth, transparencyDepth + 1);
            }
        }
        if (C.f.recursive || C.f.glossy) {
            if (rayDepth > 0 && (h.material.ks.x != 0 || h.material.ks.y != 0 || h.material.ks.z != 0)) {
                Ray reflection = computeReflectionRay(ray, h);
                if (C.f.glossy) {
                    V3 avg = v3(0.0f);
                    std::vector<Ray> rays;
                    glossySamples(C, h, reflection, rays);
                    for (const Ray& r : rays) {
                        C.cnt.bounce++;
                        avg = avg + getFinalColor(C, r, rayDepth - 1, transparencyDepth);
                    }
                    return h.material.ks * avg / float(C.p.glossy_filter_size);
                }
                C.cnt.bounce++;
                return h.material.ks * getFinalColor(C, reflection, rayDepth - 1, transparencyDepth);
            }
        }
        return Lo;
    }
    if (C.f.environment_mapping && C.p.enabled_skybox)
        return skyboxColor(C.sc, ray.d);
    return v3(0.0f);
}

// ---------------------------------------------------------------------------------------------------
// Camera, framework/src/trackball.cpp:23-31,71-74,101-126; quaternion per glm/detail/type_quat.inl:208-217,347-354.
// ---------------------------------------------------------------------------------------------------
struct Quat { float w, x, y, z; };
struct Camera {
    Quat q;
    V3 lookAt;
    float dist, halfH, halfW;
};
inline V3 rotate(Quat q, V3 v)
{
    V3 qv { q.x, q.y, q.z };
    V3 uv = cross(qv, v);
    V3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}
Camera makeCamera(const rt_camera& c)
{
    Camera cam;
    V3 e { c.rotation[0] * 0.5f, c.rotation[1] * 0.5f, c.rotation[2] * 0.5f };
    V3 co { std::cos(e.x), std::cos(e.y), std::cos(e.z) };
    V3 s { std::sin(e.x), std::sin(e.y), std::sin(e.z) };
    cam.q.w = co.x * co.y * co.z + s.x * s.y * s.z;
    cam.q.x = s.x * co.y * co.z - co.x * s.y * s.z;
    cam.q.y = co.x * s.y * co.z + s.x * co.y * s.z;
    cam.q.z = co.x * co.y * s.z - s.x * s.y * co.z;
    cam.lookAt = v3(c.look_at);
    cam.dist = c.distance;
    cam.halfH = std::tan(c.fovy / 2.0f);
    cam.halfW = c.aspect * cam.halfH;
    return cam;
}
inline V3 camPosition(const Camera& c) { return c.lookAt + rotate(c.q, V3 { 0, 0, -c.dist }); }
inline Ray camRay(const Camera& c, float px, float py)
{
    V3 d = normalize(V3 { -px * c.halfW, py * c.halfH, 1.0f });
    return { camPosition(c), rotate(c.q, d), FLT_MAX };
}

// renderRayTracing's per-pixel body, render.cpp:93-122, with calculateColor (multipleRays.cpp:15-54) and
// getEyeFrame (dof.cpp:21-51) inlined.
V3 renderPixel(Ctx& C, const Camera& cam, int x, int y, int W, int H)
{
    V3 color;
    C.rng.pixel = uint32_t(y * W + x);
    C.rng.seed = C.p.seed;
    if (C.f.multiple_rays) {
        color = v3(0.0f);
        int n = C.p.ray_multiplier;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                C.rng.sample = uint32_t(i * n + j);
                C.rng.randCounter = 0;
                C.rng.glossCall = 0;
                float randomx = float(C.rng.rand31() % 100) / 100.0f;
                float randomy = float(C.rng.rand31() % 100) / 100.0f;
                float px = (float(x) + (float(i) / float(n) + randomx)) / float(W) * 2.0f - 1.0f;
                float py = (float(y) + (float(j) / float(n) + randomy)) / float(H) * 2.0f - 1.0f;
                C.cnt.primary++;
                color = color + getFinalColor(C, camRay(cam, px, py), C.p.ray_depth);
            }
        color = color / float(std::pow(double(n), double(2.0f)));
        return color;
    }
    float px = float(x) / float(W) * 2.0f - 1.0f;
    float py = float(y) / float(H) * 2.0f - 1.0f;
    Ray cameraRay = camRay(cam, px, py);
    C.rng.sample = 0;
    C.rng.randCounter = 2;
    C.rng.glossCall = 0;
    if (C.f.depth_of_field) {
        Ray copy = cameraRay;
        HitInfo hh;
        C.cnt.probe++;
        intersectScene(C.sc, copy, hh, C.f, C.cnt);
        C.cnt.primary++;
        color = getFinalColor(C, cameraRay, C.p.ray_depth);
        if (copy.t != FLT_MAX && std::fabs(C.p.dof_f - length(copy.d) * copy.t) > C.p.dof) {
            V3 w = normalize(cameraRay.d);
            V3 up = rotate(cam.q, V3 { 0, 1, 0 });
            V3 left = rotate(cam.q, V3 { 1, 0, 0 });
            float side = C.p.dof_f / (C.p.dof_aperture * 6.0f);
            float offset = -side / 2.0f;
            V3 focus = cameraRay.o + w * C.p.dof_f;
            for (int k = 0; k < C.p.dof_samples; k++) {
                double u0 = u01(rng_u32(C.p.seed, C.rng.pixel, 0, STREAM_DOF, uint32_t(2 * k)));
                double u1 = u01(rng_u32(C.p.seed, C.rng.pixel, 0, STREAM_DOF, uint32_t(2 * k + 1)));
                V3 origin = (cameraRay.o + float(double(offset) + u0 * double(side)) * up) + float(double(offset) + u1 * double(side)) * left;
                Ray r { origin, normalize(focus - origin), FLT_MAX };
                C.rng.sample = uint32_t(k + 1);
                C.rng.randCounter = 2;
                C.rng.glossCall = 0;
                C.cnt.primary++;
                color = color + getFinalColor(C, r, C.p.ray_depth);
            }
            color = color / float(C.p.dof_samples + 1);
        }
        return color;
    }
    C.cnt.primary++;
    return getFinalColor(C, cameraRay, C.p.ray_depth);
}

// bloom.cpp:14-114
void addBloom(float* pixels, int width, int height, const rt_params& p)
{
    size_t n = size_t(width) * size_t(height);
    std::vector<V3> filt(n);
    V3* px = reinterpret_cast<V3*>(pixels);
    for (size_t i = 0; i < n; i++) {
        V3 c = px[i];
        filt[i] = (c.x > p.bloom_threshold || c.y > p.bloom_threshold || c.z > p.bloom_threshold) ? c : v3(0.0f);
    }
    int k = p.bloom_size, size = 2 * k + 1;
    std::vector<float> g(size_t(size) * size_t(size), 0.0f);
    if (p.bloom_gaussian) {
        float sd = float(std::pow(double(p.bloom_sigma), 2.0));
        const double spread = 1.0 / double(2 * sd);
        float denom = 0;
        for (int i = 0; i < size; i++) {
            const int x = i - k;
            float valx = float(std::exp(double(-x * x) * spread));
            for (int j = 0; j < size; j++) {
                const int y = j - k;
                float valy = float(std::exp(double(-y * y) * spread));
                g[size_t(i) * size_t(size) + size_t(j)] = valy * valx;
                denom += valy * valx;
            }
        }
        for (float& v : g) v /= denom;
    }
    const float boxDiv = float(std::pow(double(size), 2.0));
    for (size_t idx = 0; idx < n; idx++) {
        V3 avg = v3(0.0f);
        for (int i = 0; i < size; i++)
            for (int j = 0; j < size; j++) {
                int heightpos = (j - k) * height; // the reference strides by HEIGHT, bloom.cpp:66
                int widthpos = i - k;
                int x = widthpos + int(idx % size_t(width));
                long y = long(idx) + heightpos + widthpos;
                if (x >= 0 && x < width && y >= 0 && y < long(n)) {
                    if (p.bloom_gaussian) avg = avg + filt[size_t(y)] * g[size_t(j) * size_t(size) + size_t(i)];
                    else avg = avg + filt[size_t(y)] / boxDiv;
                }
            }
        if (p.bloom_debug) px[idx] = avg * p.bloom_scale;
        else px[idx] = px[idx] + avg * p.bloom_scale;
    }
}

Material toMaterial(const float kd[3], const float ks[3], float sh, float tr, int tex) { return { v3(kd), v3(ks), sh, tr, tex }; }

} // namespace

// =====================================================================================================
// C entry points (ctypes-friendly).
// =====================================================================================================
extern "C" {

orc_scene* orc_scene_create(const rt_scene_desc* d)
{
    orc_scene* s = new orc_scene;
    for (uint32_t m = 0; m < d->n_meshes; m++) {
        const rt_mesh_desc& md = d->meshes[m];
        Mesh mesh;
        mesh.vertices.resize(md.n_vertices);
        for (uint32_t i = 0; i < md.n_vertices; i++) {
            const float* v = md.vertices + size_t(i) * 8;
            mesh.vertices[i] = { v3(v), v3(v + 3), v[6], v[7] };
        }
        mesh.tris.assign(md.triangles, md.triangles + size_t(md.n_triangles) * 3);
        mesh.material = toMaterial(md.kd, md.ks, md.shininess, md.transparency, md.texture);
        s->meshes.push_back(std::move(mesh));
    }
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        const rt_sphere_desc& sd = d->spheres[i];
        s->spheres.push_back({ v3(sd.center), sd.radius, toMaterial(sd.kd, sd.ks, sd.shininess, sd.transparency, -1) });
    }
    auto copyImage = [](const rt_image_desc& im) {
        Image out;
        out.w = im.width;
        out.h = im.height;
        out.rgb.assign(im.rgb, im.rgb + size_t(im.width) * size_t(im.height) * 3);
        return out;
    };
    for (uint32_t i = 0; i < d->n_textures; i++) s->textures.push_back(copyImage(d->textures[i]));
    if (d->skybox)
        for (int i = 0; i < 6; i++) s->skybox.push_back(copyImage(d->skybox[i]));
    return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }

void orc_bvh_build(orc_scene* s, const rt_bvh_options* o) { buildBvh(*s, o->sah_binning != 0, o->max_level, o->sah_bins); }
int32_t orc_bvh_num_levels(const orc_scene* s) { return s->numLevels; }
int32_t orc_bvh_num_leaves(const orc_scene* s) { return s->numLeaves; }
int32_t orc_bvh_num_nodes(const orc_scene* s) { return int32_t(s->nodes.size()); }
int32_t orc_bvh_num_ids(const orc_scene* s)
{
    size_t n = 0;
    for (const Node& nd : s->nodes) n += nd.ids.size();
    return int32_t(n);
}
// meta: (isLeaf, level, idsBegin, idsCount) per node; box: lo3 hi3; ids: concatenated — the fixture layout
void orc_bvh_export(const orc_scene* s, int32_t* meta, float* box, int32_t* ids)
{
    int32_t at = 0;
    for (size_t i = 0; i < s->nodes.size(); i++) {
        const Node& n = s->nodes[i];
        meta[4 * i] = n.leaf ? 1 : 0;
        meta[4 * i + 1] = n.level;
        meta[4 * i + 2] = at;
        meta[4 * i + 3] = int32_t(n.ids.size());
        float b[6] = { n.lo.x, n.lo.y, n.lo.z, n.hi.x, n.hi.y, n.hi.z };
        std::memcpy(box + 6 * i, b, sizeof b);
        for (int v : n.ids) ids[at++] = v;
    }
}

void orc_intersect_batch(const orc_scene* s, const rt_ray* rays, uint64_t n, const rt_features* f, rt_hit* hits, uint64_t* boxTests, uint64_t* triTests)
{
    Counters total;
#pragma omp parallel
    {
        Counters cnt;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < int64_t(n); i++) {
            Ray r { v3(rays[i].origin), v3(rays[i].direction), rays[i].t };
            HitInfo h;
            int m = -1, t = -1;
            bool ok = intersectScene(*s, r, h, *f, cnt, &m, &t);
            rt_hit& o = hits[i];
            o.hit = ok ? 1 : 0;
            o.mesh = m;
            o.triangle = t;
            o.t = r.t;
            o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
            o.bary[0] = h.bary.x; o.bary[1] = h.bary.y; o.bary[2] = h.bary.z;
            o.kd[0] = h.material.kd.x; o.kd[1] = h.material.kd.y; o.kd[2] = h.material.kd.z;
        }
#pragma omp critical
        total.add(cnt);
    }
    if (boxTests) *boxTests = total.box;
    if (triTests) *triTests = total.tri;
}

void orc_visibility_batch(const orc_scene* s, const rt_ray* rays, uint64_t n, const float* samples, int32_t nSamples, const rt_features* f, uint8_t* out)
{
    rt_params p {};
    for (uint64_t i = 0; i < n; i++) {
        Ctx C { *s, nullptr, 0, *f, p, {}, {} };
        Ray r { v3(rays[i].origin), v3(rays[i].direction), rays[i].t };
        HitInfo h;
        for (int32_t k = 0; k < nSamples; k++)
            out[i * uint64_t(nSamples) + uint64_t(k)] = testVisibility(C, v3(samples + 3 * k), r, h) != 0.0f ? 1 : 0;
    }
}

void orc_camera_rays(const rt_camera* c, int32_t W, int32_t H, rt_ray* out)
{
    Camera cam = makeCamera(*c);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            Ray r = camRay(cam, float(x) / float(W) * 2.0f - 1.0f, float(y) / float(H) * 2.0f - 1.0f);
            rt_ray& o = out[size_t(y) * size_t(W) + size_t(x)];
            o.origin[0] = r.o.x; o.origin[1] = r.o.y; o.origin[2] = r.o.z;
            o.direction[0] = r.d.x; o.direction[1] = r.d.y; o.direction[2] = r.d.z;
            o.t = r.t;
        }
}

// Rows [row_begin,row_end) from the bottom, every row_step-th row only (row_step > 1 is used by the CPU
// baseline to bound its run time; skipped rows are left untouched).  `out` is the full-frame-sized slice
// for those rows: (row_end-row_begin)*W pixels, top-most row first.  threads <= 0 keeps OpenMP's default.
static uint64_t g_lastLit[3] = { 0, 0, 0 };
void orc_render(const orc_scene* s, const rt_camera* c, const rt_light* lights, int32_t nLights, const rt_features* f, const rt_params* p,
    int32_t W, int32_t H, int32_t rowBegin, int32_t rowEnd, int32_t rowStep, int32_t threads, float* out, rt_stats* stats)
{
    Camera cam = makeCamera(*c);
    Counters total;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    if (rowStep < 1) rowStep = 1;
    int nRows = (rowEnd - rowBegin + rowStep - 1) / rowStep;
#pragma omp parallel
    {
        Ctx C { *s, lights, nLights, *f, *p, {}, {} };
#pragma omp for schedule(guided)
        for (int k = 0; k < nRows; k++) {
            int y = rowBegin + k * rowStep;
            for (int x = 0; x < W; x++) {
                V3 col = renderPixel(C, cam, x, y, W, H);
                float* o = out + (size_t(rowEnd - 1 - y) * size_t(W) + size_t(x)) * 3;
                o[0] = col.x; o[1] = col.y; o[2] = col.z;
            }
        }
#pragma omp critical
        total.add(C.cnt);
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->primary_rays = total.primary; stats->shadow_rays = total.shadow; stats->bounce_rays = total.bounce; stats->probe_rays = total.probe;
        stats->box_tests = total.box; stats->triangle_tests = total.tri; stats->transparency_overflows = total.overflow;
        stats->shadow_box_tests = total.shadowBox; stats->shadow_triangle_tests = total.shadowTri;
    }
    g_lastLit[0] = total.litShadow; g_lastLit[1] = total.litShadowBox; g_lastLit[2] = total.litShadowTri;
}

// Counters of the last orc_render call of this process for the shadow queries whose light sample the surface faces:
// number of queries, their ray/box tests, their ray/triangle tests.
void orc_last_lit_shadow_counts(uint64_t out[3])
{
    for (int i = 0; i < 3; i++) out[i] = g_lastLit[i];
}

void orc_bloom(float* pixels, int32_t W, int32_t H, const rt_params* p) { addBloom(pixels, W, H, *p); }

// ---- known-answer entry points for the leaf functions (layouts match tests/golden/kats.npz)
void orc_kat_aabb(const float* in, int32_t n, float* out)
{
    for (int i = 0; i < n; i++) {
        const float* a = in + 13 * i;
        Ray r { v3(a + 6), v3(a + 9), a[12] };
        bool ok = intersectAABB(v3(a), v3(a + 3), r);
        out[2 * i] = ok ? 1.0f : 0.0f;
        out[2 * i + 1] = r.t;
    }
}
void orc_kat_tri(const float* in, int32_t n, float* out)
{
    for (int i = 0; i < n; i++) {
        const float* a = in + 16 * i;
        Ray r { v3(a + 9), v3(a + 12), a[15] };
        HitInfo h;
        bool ok = intersectTriangle(v3(a), v3(a + 3), v3(a + 6), r, h);
        float* o = out + 8 * i;
        o[0] = ok ? 1.0f : 0.0f; o[1] = r.t; o[2] = h.normal.x; o[3] = h.normal.y; o[4] = h.normal.z; o[5] = h.bary.x; o[6] = h.bary.y; o[7] = h.bary.z;
    }
}
void orc_kat_shade(const float* in, int32_t n, float* out)
{
    for (int i = 0; i < n; i++) {
        const float* a = in + 23 * i;
        Ray r { v3(a + 6), v3(a + 9), a[12] };
        HitInfo h;
        h.normal = v3(a + 13);
        h.material = { v3(a + 16), v3(a + 19), a[22], 1.0f, -1 };
        V3 c = computeShading(v3(a), v3(a + 3), r, h);
        Ray rf = computeReflectionRay(r, h);
        float* o = out + 10 * i;
        o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = rf.o.x; o[4] = rf.o.y; o[5] = rf.o.z; o[6] = rf.d.x; o[7] = rf.d.y; o[8] = rf.d.z; o[9] = rf.t;
    }
}
void orc_kat_env(const float* in, int32_t n, float* out)
{
    for (int i = 0; i < n; i++) {
        V3 l = environment_lookup(v3(in + 3 * i));
        out[3 * i] = l.x; out[3 * i + 1] = l.y; out[3 * i + 2] = l.z;
    }
}
void orc_kat_interp(const float* in, int32_t n, float* out)
{
    for (int i = 0; i < n; i++) {
        const float* a = in + 18 * i;
        V3 b = v3(a + 15);
        V3 nn = interpolateNormal(v3(a), v3(a + 3), v3(a + 6), b);
        float* o = out + 5 * i;
        o[0] = nn.x; o[1] = nn.y; o[2] = nn.z;
        o[3] = a[9] * b.x + a[11] * b.y + a[13] * b.z;
        o[4] = a[10] * b.x + a[12] * b.y + a[14] * b.z;
    }
}
void orc_kat_texel_bilinear(const orc_scene* s, int32_t texture, const float* uv, int32_t n, float* out)
{
    for (int i = 0; i < n; i++) {
        V3 c = acquireTexelBilinear(s->textures[size_t(texture)], uv[2 * i], uv[2 * i + 1]);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}
void orc_kat_texel(const orc_scene* s, int32_t texture, const float* uv, int32_t n, float* out)
{
    for (int i = 0; i < n; i++) {
        V3 c = acquireTexel(s->textures[size_t(texture)], uv[2 * i], uv[2 * i + 1]);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}
void orc_kat_clamp(const orc_scene* s, const float* uvf, int32_t n, float* out)
{
    for (int i = 0; i < n; i++) {
        V3 c = acquireTexelClamp(s->skybox[size_t(uvf[3 * i + 2])], uvf[3 * i], uvf[3 * i + 1]);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}
uint32_t orc_rng_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t index) { return rng_u32(seed, pixel, sample, stream, index); }

} // extern "C"
