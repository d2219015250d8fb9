// TEST INFRASTRUCTURE — never linked into or called by the product.
//
// Golden-vector generator.  This file is compiled together with the reference's own hot-path
// translation units (compiled where they lie under /root/reference, unmodified, see oracle/Makefile)
// and drives them to produce the fixtures committed under tests/golden/.  It only runs in the build
// container; nothing on the GPU box reads /root/reference or runs this program.
//
// What is reference code and what is harness code in the resulting binary:
//   * reference (unmodified): BVH build + traversal (bounding_volume_hierarchy.cpp, bvh_interface.cpp),
//     intersection tests (intersect.cpp, interpolate.cpp), lights/shading (light.cpp, shading.cpp),
//     textures / cube map (texture.cpp, environment_mapping.cpp), the Whitted recursion getFinalColor
//     (render.cpp), bloom (bloom.cpp), OBJ/MTL + image loading (scene.cpp, mesh.cpp, image.cpp,
//     tinyobjloader, stb_image), glm arithmetic.
//   * harness (this file): the per-pixel driver loop, camera-ray generation written with the
//     reference's vendored glm (framework/src/trackball.cpp includes <GL/glu.h>, which this image
//     lacks, so that file — like gloss.cpp, dof.cpp, screen.cpp — is NOT built; no stand-in header is
//     used), no-op bodies for the reference's debug-draw hooks (src/draw.h:18-28), the three globals
//     src/main.cpp owns, abort()-ing link stubs for Screen/Trackball members that render.cpp and
//     multipleRays.cpp mention but this harness never reaches, a replacement `rand()` that returns the
//     counter-based stream (so the reference's stochastic light/pixel jitter becomes reproducible),
//     and a glossy/DoF sampler restated from src/gloss.cpp:14-51 / src/dof.cpp:21-51 on that stream.
//
// Output: a directory of .npy arrays + .json manifests; oracle/gen_golden.py packs them into
// tests/golden/*.npz.
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <numeric>
#include <optional>
#include <span>
#include <stack>
#include <string>
#include <variant>
#include <vector>

#include <glm/glm.hpp>
#include <glm/gtc/quaternion.hpp>

// The BVH node vector is a private member; the harness needs it to pin the node order.
#define private public
#include "bounding_volume_hierarchy.h"
#include "bvh_interface.h"
#undef private
#include "bloom.h"
#include "config.h"
#include "dof.h"
#include "draw.h"
#include "environment_mapping.h"
#include "gloss.h"
#include "interpolate.h"
#include "intersect.h"
#include "light.h"
#include "scene.h"
#include "screen.h"
#include "render.h"
#include "multipleRays.h"
#include "shading.h"
#include "texture.h"
#include <framework/trackball.h>

// ------------------------------------------------------------------------------------------------
// Counter-based random stream shared (by specification, not by code) with oracle/cpu_ref.cpp and the
// HIP kernels: value = H(H(H(H(seed) ^ pixel) ^ (sample*0x9e3779b9 + stream)) ^ index).
// ------------------------------------------------------------------------------------------------
static inline uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static inline uint32_t rng_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t index)
{
    uint32_t h = hash32(seed);
    h = hash32(h ^ pixel);
    h = hash32(h ^ (sample * 0x9e3779b9U + stream));
    return hash32(h ^ index);
}
enum { STREAM_RAND = 0, STREAM_DOF = 1, STREAM_GLOSS = 2 };
static inline uint32_t float_bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
// Key of one getRaySamples call: a hash of the bits of the reflection ray it perturbs.  Sample i of the call uses the
// Gaussian deviates key + 2i and key + 2i + 1 of the glossy stream (same specification as the oracle and the kernels).
static inline uint32_t ray_key(const glm::vec3& o, const glm::vec3& d)
{
    uint32_t h = hash32(float_bits(o.x) ^ 0x9e3779b9U);
    h = hash32(h ^ float_bits(o.y));
    h = hash32(h ^ float_bits(o.z));
    h = hash32(h ^ float_bits(d.x));
    h = hash32(h ^ float_bits(d.y));
    return hash32(h ^ float_bits(d.z));
}

struct RngCtx {
    uint32_t seed = 1, pixel = 0, sample = 0;
    uint32_t randCounter = 0;  // next index on STREAM_RAND
    uint32_t glossCall = 0;    // number of getRaySamples calls so far for this pixel sample
};
static thread_local RngCtx g_rng;

// Replaces libc rand() for the whole program (the executable's definition wins at link time), so the
// reference's `rand() % 50` / `rand() % 100` draws (src/light.cpp:25,50-51, src/multipleRays.cpp:32-33)
// consume the counter-based stream in call order.  31-bit like glibc's RAND_MAX.
extern "C" int rand(void) noexcept
{
    uint32_t u = rng_u32(g_rng.seed, g_rng.pixel, g_rng.sample, STREAM_RAND, g_rng.randCounter++);
    return int(u >> 1);
}

// ------------------------------------------------------------------------------------------------
// Link-time hooks.
// ------------------------------------------------------------------------------------------------
bool enableDebugDraw = false;       // src/draw.cpp:25
int rayDepth = 4;                   // src/main.cpp:47
bool intersectedButNotTraversed = false; // src/main.cpp:45
int debugBVHLeafId = 0;             // src/main.cpp:46

// Tunables that live in the files this build cannot compile (src/gloss.cpp:9-10, src/dof.cpp:9-12).
int extr_glossy_filterSize = 64;
float extr_glossy_sigma = 1;
int extr_dof_samples = 3;
float extr_dof_aperture = 1.4f;
float extr_dof_f = 2.5f;
float extr_dof = 0.0f;

// Capture of the final-hit triangle: BoundingVolumeHierarchy::intersect calls drawTriangle with
// references into scene.meshes[m].vertices for every successful BVH query
// (src/bounding_volume_hierarchy.cpp:541-545).
struct TriCapture {
    bool armed = false;
    const Vertex* v[3] = { nullptr, nullptr, nullptr };
    int count = 0;
};
static thread_local TriCapture g_cap;
static thread_local uint64_t g_boxVisits = 0; // drawAABB calls = nodes visited (src/...hierarchy.cpp:313,318)

void drawExampleOfCustomVisualDebug() {}
void drawRay(const Ray&, const glm::vec3&) {}
void drawAABB(const AxisAlignedBox&, DrawMode, const glm::vec3&, float) { g_boxVisits++; }
void drawTriangle(const Vertex& v0, const Vertex& v1, const Vertex& v2)
{
    if (g_cap.armed) {
        g_cap.v[0] = &v0; g_cap.v[1] = &v1; g_cap.v[2] = &v2;
        g_cap.count++;
    }
}
void drawMesh(const Mesh&) {}
void drawSphere(const Sphere&) {}
void drawSphere(const glm::vec3&, float, const glm::vec3&) {}
void drawScene(const Scene&) {}

[[noreturn]] static void unreachable_stub(const char* what)
{
    std::fprintf(stderr, "ref_harness: link stub '%s' was called; it belongs to a reference file this build cannot compile\n", what);
    std::abort();
}
// Members of classes whose .cpp needs <GL/glu.h>.  Never reached: the harness drives getFinalColor
// itself and does not call renderRayTracing / calculateColor.
glm::ivec2 Screen::resolution() const { unreachable_stub("Screen::resolution"); }
void Screen::setPixel(int, int, const glm::vec3&) { unreachable_stub("Screen::setPixel"); }
std::vector<glm::vec3>& Screen::pixels() { unreachable_stub("Screen::pixels"); }
Ray Trackball::generateRay(const glm::vec2&) const { unreachable_stub("Trackball::generateRay"); }
std::vector<Ray> getEyeFrame(Ray&, const Trackball&) { unreachable_stub("getEyeFrame"); }

// Glossy sampler, restated from src/gloss.cpp:14-51 with the normal deviates drawn from the
// counter-based stream (Box–Muller in double) instead of a random_device-seeded mt19937.  Called by the
// reference's getFinalColor (src/render.cpp:55).
static inline double u01(uint32_t u) { return (double(u) + 0.5) * (1.0 / 4294967296.0); }
static double gauss(uint32_t idx)
{
    uint32_t a = rng_u32(g_rng.seed, g_rng.pixel, g_rng.sample, STREAM_GLOSS, 2u * idx);
    uint32_t b = rng_u32(g_rng.seed, g_rng.pixel, g_rng.sample, STREAM_GLOSS, 2u * idx + 1u);
    return std::sqrt(-2.0 * std::log(u01(a))) * std::cos(6.283185307179586476925 * u01(b));
}
std::vector<Ray> getRaySamples(HitInfo& hitInfo, Ray& ref)
{
    glm::vec3 w = glm::normalize(ref.direction);
    int minTerm = fabs(w.x) <= fabs(w.y) && fabs(w.x) <= fabs(w.z) ? 0 : (fabs(w.y) <= fabs(w.x) && fabs(w.y) <= fabs(w.z) ? 1 : 2);
    glm::vec3 t = w;
    t[minTerm] = 1;
    glm::vec3 u = glm::normalize(glm::cross(t, w));
    glm::vec3 v = glm::cross(w, u);
    float side = extr_glossy_sigma / (hitInfo.material.shininess * 3.0f);
    float offset = -side / 2.0f;
    std::vector<Ray> out;
    // the call's deviates are indexed from a hash of the reflection ray's bits (see ray_key)
    uint32_t call = ray_key(ref.origin, ref.direction);
    for (int i = 0; i < extr_glossy_filterSize; i++) {
        uint32_t k = call + 2u * uint32_t(i);
        double n0 = gauss(k), n1 = gauss(k + 1u);
        glm::vec3 rPrime = ref.direction + float(offset + n0 * side) * u + float(offset + n1 * side) * v;
        if (glm::dot(rPrime, hitInfo.normal) > 0)
            out.push_back({ ref.origin, glm::normalize(rPrime), std::numeric_limits<float>::max() });
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// NPY writer (format 1.0) — only what the fixtures need.
// ------------------------------------------------------------------------------------------------
static std::filesystem::path g_out;

static void write_npy(const std::string& name, const char* descr, size_t itemsize, const std::vector<size_t>& shape, const void* data)
{
    std::string dict = "{'descr': '" + std::string(descr) + "', 'fortran_order': False, 'shape': (";
    size_t count = 1;
    for (size_t i = 0; i < shape.size(); i++) {
        dict += std::to_string(shape[i]);
        if (shape.size() == 1 || i + 1 < shape.size()) dict += ",";
        if (i + 1 < shape.size()) dict += " ";
        count *= shape[i];
    }
    dict += "), }";
    size_t total = 10 + dict.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    dict += std::string(pad, ' ');
    dict += "\n";
    std::ofstream f(g_out / (name + ".npy"), std::ios::binary);
    const unsigned char magic[8] = { 0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0 };
    f.write(reinterpret_cast<const char*>(magic), 8);
    uint16_t hl = uint16_t(dict.size());
    f.write(reinterpret_cast<const char*>(&hl), 2);
    f.write(dict.data(), std::streamsize(dict.size()));
    f.write(reinterpret_cast<const char*>(data), std::streamsize(count * itemsize));
}
static void npy_f32(const std::string& n, const std::vector<size_t>& s, const float* d) { write_npy(n, "<f4", 4, s, d); }
static void npy_i32(const std::string& n, const std::vector<size_t>& s, const int32_t* d) { write_npy(n, "<i4", 4, s, d); }
static void npy_u32(const std::string& n, const std::vector<size_t>& s, const uint32_t* d) { write_npy(n, "<u4", 4, s, d); }
static void npy_u8(const std::string& n, const std::vector<size_t>& s, const uint8_t* d) { write_npy(n, "|u1", 1, s, d); }
static void npy_f64(const std::string& n, const std::vector<size_t>& s, const double* d) { write_npy(n, "<f8", 8, s, d); }

// ------------------------------------------------------------------------------------------------
// Camera (restates framework/src/trackball.cpp:23-31,71-74,101-126 with the vendored glm).
// ------------------------------------------------------------------------------------------------
struct Cam {
    float fovyDeg, dist;
    glm::vec3 lookAt, rotDeg;
    float fovy() const { return glm::radians(fovyDeg); }
    glm::vec3 euler() const { return glm::radians(rotDeg); }
};
static const Cam CAM_A { 50.0f, 3.0f, { 0.0f, 0.0f, 0.0f }, { 20.0f, 20.0f, 0.0f } }; // src/config.h:16-21
static const Cam CAM_B { 60.0f, 2.5f, { 0.1f, -0.05f, 0.05f }, { -15.0f, 140.0f, 0.0f } };
static const Cam CAM_S { 70.0f, 3.0f, { 0.0f, 0.0f, 4.0f }, { 10.0f, 15.0f, 0.0f } }; // looks down +z at the Spheres scene

struct CamState {
    glm::vec3 euler, lookAt;
    float dist, halfH, halfW;
};
static CamState make_cam(const Cam& c, int W, int H)
{
    CamState s;
    s.euler = c.euler();
    s.euler.z = 0; // trackball.cpp:30-32
    s.lookAt = c.lookAt;
    s.dist = c.dist;
    s.halfH = std::tan(c.fovy() / 2.0f);
    s.halfW = (float(W) / float(H)) * s.halfH; // window.cpp:379-384
    return s;
}
static glm::vec3 cam_position(const CamState& s) { return s.lookAt + glm::quat(s.euler) * glm::vec3(0, 0, -s.dist); }
static Ray cam_ray(const CamState& s, const glm::vec2& pixel)
{
    const glm::vec3 d = glm::normalize(glm::vec3(-pixel.x * s.halfW, pixel.y * s.halfH, 1.0f));
    Ray r;
    r.origin = cam_position(s);
    r.direction = glm::quat(s.euler) * d;
    r.t = std::numeric_limits<float>::max();
    return r;
}
static glm::vec2 ndc_of(int x, int y, int W, int H)
{
    return { float(x) / float(W) * 2.0f - 1.0f, float(y) / float(H) * 2.0f - 1.0f }; // render.cpp:99-102
}

// ------------------------------------------------------------------------------------------------
// Scene handling.
// ------------------------------------------------------------------------------------------------
// Small procedural cube map used instead of the 6 x 2048^2 JPGs so the same texels can be regenerated
// anywhere from a formula (tests/golden stores it as u8 inside each scene fixture).
static constexpr int SKY_N = 32;
static uint8_t sky_texel(int face, int x, int y, int c)
{
    uint32_t h = hash32(uint32_t(face) * 0x1000193u ^ hash32(uint32_t(y * SKY_N + x) * 3u + uint32_t(c) + 0x51ed27u));
    // smooth-ish gradient plus hash noise so neighbouring texels differ
    int base = (face * 37 + x * 5 + y * 3 + c * 60) & 0xff;
    return uint8_t((base + int(h & 31u)) & 0xff);
}
static void install_procedural_skybox(Scene& scene)
{
    // Image has no default constructor; reuse the loaded ones and overwrite their payload.
    for (int f = 0; f < 6; f++) {
        Image& img = scene.skybox[size_t(f)];
        img.width = SKY_N;
        img.height = SKY_N;
        img.pixels.clear();
        for (int y = 0; y < SKY_N; y++)
            for (int x = 0; x < SKY_N; x++)
                img.pixels.emplace_back(sky_texel(f, x, y, 0) / 255.0f, sky_texel(f, x, y, 1) / 255.0f, sky_texel(f, x, y, 2) / 255.0f);
    }
}

static uint8_t to_u8(float p)
{
    long v = std::lround(double(p) * 255.0);
    if (std::fabs(float(v) / 255.0f - p) != 0.0f) {
        std::fprintf(stderr, "texture value %g is not an exact u8/255\n", double(p));
        std::abort();
    }
    return uint8_t(v);
}

static void dump_scene(const std::string& prefix, const Scene& scene)
{
    std::vector<float> verts;
    std::vector<uint32_t> tris;
    std::vector<int32_t> meshTable; // vtxOff, vtxCnt, triOff, triCnt, textureId
    std::vector<float> materials;   // kd3 ks3 shininess transparency
    std::vector<const Image*> textures;
    for (const Mesh& m : scene.meshes) {
        int32_t texId = -1;
        if (m.material.kdTexture) {
            const Image* img = m.material.kdTexture.get();
            auto it = std::find(textures.begin(), textures.end(), img);
            if (it == textures.end()) { textures.push_back(img); texId = int32_t(textures.size() - 1); }
            else texId = int32_t(it - textures.begin());
        }
        meshTable.insert(meshTable.end(), { int32_t(verts.size() / 8), int32_t(m.vertices.size()), int32_t(tris.size() / 3), int32_t(m.triangles.size()), texId });
        for (const Vertex& v : m.vertices)
            verts.insert(verts.end(), { v.position.x, v.position.y, v.position.z, v.normal.x, v.normal.y, v.normal.z, v.texCoord.x, v.texCoord.y });
        for (const glm::uvec3& t : m.triangles)
            tris.insert(tris.end(), { t.x, t.y, t.z });
        materials.insert(materials.end(), { m.material.kd.x, m.material.kd.y, m.material.kd.z, m.material.ks.x, m.material.ks.y, m.material.ks.z, m.material.shininess, m.material.transparency });
    }
    npy_f32(prefix + "vertices", { verts.size() / 8, 8 }, verts.data());
    npy_u32(prefix + "triangles", { tris.size() / 3, 3 }, tris.data());
    npy_i32(prefix + "mesh_table", { meshTable.size() / 5, 5 }, meshTable.data());
    npy_f32(prefix + "materials", { materials.size() / 8, 8 }, materials.data());
    for (size_t i = 0; i < textures.size(); i++) {
        std::vector<uint8_t> px;
        for (const glm::vec3& p : textures[i]->pixels) { px.push_back(to_u8(p.x)); px.push_back(to_u8(p.y)); px.push_back(to_u8(p.z)); }
        npy_u8(prefix + "texture" + std::to_string(i), { size_t(textures[i]->height), size_t(textures[i]->width), 3 }, px.data());
    }
    std::vector<float> sph;
    for (const Sphere& s : scene.spheres)
        sph.insert(sph.end(), { s.center.x, s.center.y, s.center.z, s.radius, s.material.kd.x, s.material.kd.y, s.material.kd.z, s.material.ks.x, s.material.ks.y, s.material.ks.z, s.material.shininess, s.material.transparency });
    npy_f32(prefix + "spheres", { sph.size() / 12, 12 }, sph.data());
    // lights: type (0 point, 1 segment, 2 parallelogram) + 21 floats
    std::vector<float> lights;
    for (const auto& l : scene.lights) {
        float rec[22] = { 0 };
        auto put = [&](int at, const glm::vec3& v) { rec[at] = v.x; rec[at + 1] = v.y; rec[at + 2] = v.z; };
        if (std::holds_alternative<PointLight>(l)) {
            const auto& p = std::get<PointLight>(l);
            rec[0] = 0; put(1, p.position); put(4, p.color);
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& s = std::get<SegmentLight>(l);
            rec[0] = 1; put(1, s.endpoint0); put(4, s.endpoint1); put(7, s.color0); put(10, s.color1);
        } else {
            const auto& p = std::get<ParallelogramLight>(l);
            rec[0] = 2; put(1, p.v0); put(4, p.edge01); put(7, p.edge02); put(10, p.color0); put(13, p.color1); put(16, p.color2); put(19, p.color3);
        }
        lights.insert(lights.end(), rec, rec + 22);
    }
    npy_f32(prefix + "lights", { lights.size() / 22, 22 }, lights.data());
    std::vector<uint8_t> sky;
    for (int f = 0; f < 6; f++)
        for (const glm::vec3& p : scene.skybox[size_t(f)].pixels) { sky.push_back(to_u8(p.x)); sky.push_back(to_u8(p.y)); sky.push_back(to_u8(p.z)); }
    npy_u8(prefix + "skybox", { 6, size_t(scene.skybox[0].height), size_t(scene.skybox[0].width), 3 }, sky.data());
}

static void dump_bvh(const std::string& prefix, BvhInterface& bvh)
{
    const std::vector<BVHNode>& nodes = bvh.m_impl->nodes;
    std::vector<int32_t> meta;   // isLeaf, level, idsBegin, idsCount
    std::vector<float> boxes;    // lo3 hi3
    std::vector<int32_t> ids;    // concatenated ids vectors
    for (const BVHNode& n : nodes) {
        meta.insert(meta.end(), { n.isLeafNode ? 1 : 0, n.level, int32_t(ids.size()), int32_t(n.ids.size()) });
        boxes.insert(boxes.end(), { n.box.lower.x, n.box.lower.y, n.box.lower.z, n.box.upper.x, n.box.upper.y, n.box.upper.z });
        ids.insert(ids.end(), n.ids.begin(), n.ids.end());
    }
    npy_i32(prefix + "node_meta", { nodes.size(), 4 }, meta.data());
    npy_f32(prefix + "node_box", { nodes.size(), 6 }, boxes.data());
    npy_i32(prefix + "node_ids", { ids.size() }, ids.data());
    int32_t counts[2] = { bvh.numLevels(), bvh.numLeaves() };
    npy_i32(prefix + "levels_leaves", { 2 }, counts);
}

// Recover (mesh, triangle) from the vertex references the reference passes to drawTriangle.
static void resolve_capture(const Scene& scene, int32_t& meshOut, int32_t& triOut)
{
    meshOut = -1; triOut = -1;
    if (g_cap.count == 0) return;
    for (size_t m = 0; m < scene.meshes.size(); m++) {
        const Mesh& mesh = scene.meshes[m];
        if (mesh.vertices.empty()) continue;
        const Vertex* b = mesh.vertices.data();
        const Vertex* e = b + mesh.vertices.size();
        if (g_cap.v[0] >= b && g_cap.v[0] < e) {
            glm::uvec3 want { unsigned(g_cap.v[0] - b), unsigned(g_cap.v[1] - b), unsigned(g_cap.v[2] - b) };
            // The reference reports the triangle by id; ids with identical vertex triples would be
            // indistinguishable here, so take the first match and let the test compare vertex triples.
            for (size_t t = 0; t < mesh.triangles.size(); t++)
                if (mesh.triangles[t] == want) { meshOut = int32_t(m); triOut = int32_t(t); return; }
        }
    }
}

struct FeatureSet {
    const char* name;
    Features f;
};
static Features feat(bool shading, bool recursive, bool hard, bool soft, bool nrm, bool tex, bool accel = true)
{
    Features f;
    f.enableShading = shading; f.enableRecursive = recursive; f.enableHardShadow = hard; f.enableSoftShadow = soft;
    f.enableNormalInterp = nrm; f.enableTextureMapping = tex; f.enableAccelStructure = accel;
    return f;
}

// Primary-ray hit records through the reference's BvhInterface::intersect.
static void dump_hits(const std::string& prefix, Scene& scene, BvhInterface& bvh, const Cam& cam, int W, int H, const Features& f, bool full, bool storeRays)
{
    CamState cs = make_cam(cam, W, H);
    size_t N = size_t(W) * size_t(H);
    std::vector<float> rays(N * 7), t(N), nrm(N * 3), bary(N * 3), kd(N * 3);
    std::vector<uint8_t> hit(N);
    std::vector<int32_t> mesh(N), tri(N);
    uint64_t visits = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = size_t(y) * size_t(W) + size_t(x);
            Ray r = cam_ray(cs, ndc_of(x, y, W, H));
            float* rr = &rays[i * 7];
            rr[0] = r.origin.x; rr[1] = r.origin.y; rr[2] = r.origin.z; rr[3] = r.direction.x; rr[4] = r.direction.y; rr[5] = r.direction.z; rr[6] = r.t;
            HitInfo h {};
            g_cap = TriCapture {};
            g_cap.armed = f.enableAccelStructure;
            g_boxVisits = 0;
            bool ok = bvh.intersect(r, h, f);
            visits += g_boxVisits;
            g_cap.armed = false;
            hit[i] = ok ? 1 : 0;
            t[i] = r.t;
            mesh[i] = -1; tri[i] = -1;
            if (ok) {
                resolve_capture(scene, mesh[i], tri[i]);
                nrm[i * 3] = h.normal.x; nrm[i * 3 + 1] = h.normal.y; nrm[i * 3 + 2] = h.normal.z;
                bary[i * 3] = h.barycentricCoord.x; bary[i * 3 + 1] = h.barycentricCoord.y; bary[i * 3 + 2] = h.barycentricCoord.z;
                kd[i * 3] = h.material.kd.x; kd[i * 3 + 1] = h.material.kd.y; kd[i * 3 + 2] = h.material.kd.z;
            }
        }
    if (storeRays) npy_f32(prefix + "rays", { N, 7 }, rays.data());
    npy_u8(prefix + "hit", { N }, hit.data());
    npy_f32(prefix + "t", { N }, t.data());
    npy_i32(prefix + "mesh", { N }, mesh.data());
    npy_i32(prefix + "tri", { N }, tri.data());
    if (full) {
        npy_f32(prefix + "normal", { N, 3 }, nrm.data());
        npy_f32(prefix + "bary", { N, 3 }, bary.data());
        npy_f32(prefix + "kd", { N, 3 }, kd.data());
    }
    double v = double(visits);
    npy_f64(prefix + "node_visits", { 1 }, &v);
}

// Shadow-ray visibility through the reference's testVisibilityLightSample (src/light.cpp:68-89) for
// every primary hit and a list of sample positions.
static void dump_visibility(const std::string& prefix, Scene& scene, BvhInterface& bvh, const Cam& cam, int W, int H, Features f, const std::vector<glm::vec3>& samples)
{
    CamState cs = make_cam(cam, W, H);
    size_t N = size_t(W) * size_t(H);
    std::vector<uint8_t> vis(N * samples.size(), 255);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = size_t(y) * size_t(W) + size_t(x);
            Ray r = cam_ray(cs, ndc_of(x, y, W, H));
            HitInfo h {};
            if (!bvh.intersect(r, h, f)) continue;
            for (size_t s = 0; s < samples.size(); s++)
                vis[i * samples.size() + s] = testVisibilityLightSample(samples[s], glm::vec3(1.0f), bvh, f, r, h) != 0.0f ? 1 : 0;
        }
    npy_u8(prefix + "vis", { N, samples.size() }, vis.data());
    std::vector<float> sp;
    for (auto& s : samples) sp.insert(sp.end(), { s.x, s.y, s.z });
    npy_f32(prefix + "vis_samples", { samples.size(), 3 }, sp.data());
}

struct RenderOpts {
    int rayMultiplier = 1;   // used when features.extra.enableMultipleRaysPerPixel
    uint32_t seed = 1;
};

// The per-pixel loop of renderRayTracing (src/render.cpp:89-124) and calculateColor
// (src/multipleRays.cpp:26-50), driving the reference's getFinalColor.  Output is in Screen order
// (top row first, src/screen.cpp:41).
static std::vector<glm::vec3> render_image(Scene& scene, BvhInterface& bvh, const Cam& cam, int W, int H, const Features& f, const RenderOpts& o)
{
    CamState cs = make_cam(cam, W, H);
    std::vector<glm::vec3> img(size_t(W) * size_t(H), glm::vec3(0.0f));
    const glm::vec3 up = glm::quat(cs.euler) * glm::vec3(0, 1, 0);   // trackball.cpp:117-120
    const glm::vec3 left = glm::quat(cs.euler) * glm::vec3(1, 0, 0); // trackball.cpp:122-126
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            glm::vec3 color;
            g_rng = RngCtx {};
            g_rng.seed = o.seed;
            g_rng.pixel = uint32_t(y * W + x);
            if (f.extra.enableMultipleRaysPerPixel) {
                color = glm::vec3 { 0 };
                int n = o.rayMultiplier;
                for (int i = 0; i < n; i++)
                    for (int j = 0; j < n; j++) {
                        g_rng.sample = uint32_t(i * n + j);
                        g_rng.randCounter = 0;
                        g_rng.glossCall = 0;
                        float randomx = (float)((rand() % 100) / 100.0f);
                        float randomy = (float)((rand() % 100) / 100.0f);
                        glm::vec2 ndc {
                            float(float(x) + ((i) / float(n) + randomx)) / float(W) * 2.0f - 1.0f,
                            float(float(y) + ((j) / float(n) + randomy)) / float(H) * 2.0f - 1.0f
                        };
                        Ray r = cam_ray(cs, ndc);
                        color += getFinalColor(scene, bvh, r, f);
                    }
                color /= pow(n, 2.0f);
            } else {
                Ray cameraRay = cam_ray(cs, ndc_of(x, y, W, H));
                g_rng.sample = 0;
                g_rng.randCounter = 2; // indices 0,1 are reserved for the pixel jitter
                if (f.extra.enableDepthOfField) {
                    Ray copy = { cameraRay.origin, cameraRay.direction, cameraRay.t };
                    HitInfo h = {};
                    bvh.intersect(copy, h, f);
                    color = getFinalColor(scene, bvh, cameraRay, f);
                    if (copy.t != std::numeric_limits<float>::max() && fabs(extr_dof_f - glm::length(copy.direction) * copy.t) > extr_dof) {
                        // getEyeFrame, src/dof.cpp:21-51, on STREAM_DOF
                        glm::vec3 w = glm::normalize(cameraRay.direction);
                        float side = extr_dof_f / (extr_dof_aperture * 6.0f);
                        float offset = -side / 2.0f;
                        glm::vec3 focusPoint = cameraRay.origin + w * extr_dof_f;
                        for (int k = 0; k < extr_dof_samples; k++) {
                            double u0 = u01(rng_u32(o.seed, g_rng.pixel, 0, STREAM_DOF, uint32_t(2 * k)));
                            double u1 = u01(rng_u32(o.seed, g_rng.pixel, 0, STREAM_DOF, uint32_t(2 * k + 1)));
                            glm::vec3 origin = cameraRay.origin + float(offset + u0 * side) * up + float(offset + u1 * side) * left;
                            Ray r { origin, glm::normalize(focusPoint - origin) };
                            g_rng.sample = uint32_t(k + 1);
                            g_rng.randCounter = 2;
                            g_rng.glossCall = 0;
                            color += getFinalColor(scene, bvh, r, f);
                        }
                        color /= float(extr_dof_samples + 1);
                    }
                } else
                    color = getFinalColor(scene, bvh, cameraRay, f);
            }
            img[size_t(H - 1 - y) * size_t(W) + size_t(x)] = color;
        }
    if (f.extra.enableBloomEffect)
        addBloom(img, W, H);
    return img;
}

static void dump_image(const std::string& name, const std::vector<glm::vec3>& img, int W, int H)
{
    npy_f32(name, { size_t(H), size_t(W), 3 }, &img[0].x);
}

// FNV-1a 64 over the raw bytes; enough to pin a full-size buffer without committing it.
static uint64_t fnv1a(const void* p, size_t n)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// ------------------------------------------------------------------------------------------------
// Known-answer vectors for the leaf functions.
// ------------------------------------------------------------------------------------------------
static float frand(uint32_t& s, float lo, float hi)
{
    s = hash32(s + 0x9e3779b9u);
    return lo + (hi - lo) * float(s >> 8) * (1.0f / 16777216.0f);
}

static void dump_kats()
{
    uint32_t s = 12345;
    // --- ray / AABB (src/intersect.cpp:111-140)
    {
        std::vector<float> in, out;
        auto run = [&](glm::vec3 lo, glm::vec3 hi, glm::vec3 o, glm::vec3 d, float t) {
            Ray r { o, d, t };
            AxisAlignedBox b { lo, hi };
            bool ok = intersectRayWithShape(b, r);
            in.insert(in.end(), { lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, o.x, o.y, o.z, d.x, d.y, d.z, t });
            out.insert(out.end(), { ok ? 1.0f : 0.0f, r.t });
        };
        const float FM = std::numeric_limits<float>::max();
        for (int i = 0; i < 400; i++) {
            glm::vec3 lo { frand(s, -1, 0.5f), frand(s, -1, 0.5f), frand(s, -1, 0.5f) };
            glm::vec3 hi = lo + glm::vec3 { frand(s, 0, 1), frand(s, 0, 1), frand(s, 0, 1) };
            glm::vec3 o { frand(s, -2, 2), frand(s, -2, 2), frand(s, -2, 2) };
            glm::vec3 d { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) };
            int mode = i % 8;
            if (mode == 1) d.x = 0.0f;
            if (mode == 2) d.y = 5e-7f;
            if (mode == 3) { d.z = -1e-6f; d.x = 0.0f; }
            if (mode == 4) o = (lo + hi) * 0.5f;             // origin inside
            if (mode == 5) { o = lo; o.x = lo.x; d = glm::vec3(1, 0.3f, 0.2f); } // origin on a corner
            if (mode == 6) { o = glm::vec3(lo.x, (lo.y + hi.y) * 0.5f, (lo.z + hi.z) * 0.5f); d = glm::vec3(1, 0, 0); } // on a face, pointing in
            float t = (i % 3 == 0) ? 1.0f : FM;
            if (mode == 7) t = frand(s, 0.0f, 3.0f);
            run(lo, hi, o, d, t);
        }
        npy_f32("kat_aabb_in", { in.size() / 13, 13 }, in.data());
        npy_f32("kat_aabb_out", { out.size() / 2, 2 }, out.data());
    }
    // --- ray / triangle (src/intersect.cpp:22-80) incl. barycentrics (src/interpolate.cpp:4-11)
    {
        std::vector<float> in, out;
        auto run = [&](glm::vec3 v0, glm::vec3 v1, glm::vec3 v2, glm::vec3 o, glm::vec3 d, float t) {
            Ray r { o, d, t };
            HitInfo h {};
            h.normal = glm::vec3(0); h.barycentricCoord = glm::vec3(0);
            bool ok = intersectRayWithTriangle(v0, v1, v2, r, h);
            in.insert(in.end(), { v0.x, v0.y, v0.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z, o.x, o.y, o.z, d.x, d.y, d.z, t });
            out.insert(out.end(), { ok ? 1.0f : 0.0f, r.t, h.normal.x, h.normal.y, h.normal.z, h.barycentricCoord.x, h.barycentricCoord.y, h.barycentricCoord.z });
        };
        const float FM = std::numeric_limits<float>::max();
        for (int i = 0; i < 600; i++) {
            glm::vec3 v0 { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) };
            glm::vec3 v1 { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) };
            glm::vec3 v2 { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) };
            glm::vec3 o { frand(s, -3, 3), frand(s, -3, 3), frand(s, -3, 3) };
            // aim at a point near/inside the triangle most of the time
            float a = frand(s, -0.2f, 1.2f), b = frand(s, -0.2f, 1.2f);
            glm::vec3 target = v0 + a * (v1 - v0) + b * (1.0f - a) * (v2 - v0);
            int mode = i % 10;
            if (mode == 1) target = v0;                              // vertex hit
            if (mode == 2) target = (v0 + v1) * 0.5f;                // edge hit
            if (mode == 3) v2 = v0 + (v1 - v0) * 2.0f;               // degenerate (collinear)
            if (mode == 4) v1 = v0;                                  // degenerate (repeated vertex)
            glm::vec3 d = target - o;
            if (mode == 5) d = glm::normalize(d);
            if (mode == 6) d = glm::cross(v1 - v0, glm::cross(v1 - v0, v2 - v0)); // parallel to the plane
            if (mode == 7) d = -d;                                   // behind the origin
            float t = (mode == 8) ? 0.5f : ((mode == 9) ? 1.0f : FM);
            run(v0, v1, v2, o, d, t);
        }
        npy_f32("kat_tri_in", { in.size() / 16, 16 }, in.data());
        npy_f32("kat_tri_out", { out.size() / 8, 8 }, out.data());
    }
    // --- Phong + reflection ray (src/shading.cpp:6-51)
    {
        std::vector<float> in, out;
        Features f;
        for (int i = 0; i < 300; i++) {
            glm::vec3 lp { frand(s, -2, 2), frand(s, -2, 2), frand(s, -2, 2) };
            glm::vec3 lc { frand(s, 0, 1), frand(s, 0, 1), frand(s, 0, 1) };
            Ray r { { frand(s, -2, 2), frand(s, -2, 2), frand(s, -2, 2) }, { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) }, frand(s, 0.1f, 4.0f) };
            HitInfo h {};
            h.normal = { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) };
            if (i % 3 == 0) h.normal = glm::normalize(h.normal);
            h.material.kd = { frand(s, 0, 1), frand(s, 0, 1), frand(s, 0, 1) };
            h.material.ks = { frand(s, 0, 1), frand(s, 0, 1), frand(s, 0, 1) };
            const float shin[5] = { 1.0f, 4.0f, 10.000002f, 250.0f, 37.5f };
            h.material.shininess = shin[i % 5];
            glm::vec3 c = computeShading(lp, lc, f, r, h);
            Ray refl = computeReflectionRay(r, h);
            in.insert(in.end(), { lp.x, lp.y, lp.z, lc.x, lc.y, lc.z, r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z, r.t,
                                  h.normal.x, h.normal.y, h.normal.z, h.material.kd.x, h.material.kd.y, h.material.kd.z, h.material.ks.x, h.material.ks.y, h.material.ks.z, h.material.shininess });
            out.insert(out.end(), { c.x, c.y, c.z, refl.origin.x, refl.origin.y, refl.origin.z, refl.direction.x, refl.direction.y, refl.direction.z, refl.t });
        }
        npy_f32("kat_shade_in", { in.size() / 23, 23 }, in.data());
        npy_f32("kat_shade_out", { out.size() / 10, 10 }, out.data());
    }
    // --- cube-map lookup (src/environment_mapping.cpp:9-49) on the procedural skybox size
    {
        std::vector<float> in, out;
        Features f;
        for (int i = 0; i < 400; i++) {
            glm::vec3 v { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) };
            if (i % 7 == 1) v = glm::vec3(1, 1, 0.3f);
            if (i % 7 == 2) v = glm::vec3(-0.5f, 0.5f, 0.5f);
            if (i % 7 == 3) v = glm::vec3(0, 0, -1);
            if (i % 7 == 4) v = glm::vec3(0.2f, -1, 1);
            glm::vec3 n = glm::normalize(v);
            glm::vec3 l = environment_lookup(n);
            in.insert(in.end(), { n.x, n.y, n.z });
            out.insert(out.end(), { l.x, l.y, l.z });
        }
        npy_f32("kat_env_in", { in.size() / 3, 3 }, in.data());
        npy_f32("kat_env_out", { out.size() / 3, 3 }, out.data());
    }
    // --- interpolation helpers (src/interpolate.cpp:13-28)
    {
        std::vector<float> in, out;
        for (int i = 0; i < 200; i++) {
            glm::vec3 n0 { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) }, n1 { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) }, n2 { frand(s, -1, 1), frand(s, -1, 1), frand(s, -1, 1) };
            glm::vec2 t0 { frand(s, 0, 1), frand(s, 0, 1) }, t1 { frand(s, 0, 1), frand(s, 0, 1) }, t2 { frand(s, 0, 1), frand(s, 0, 1) };
            glm::vec3 b { frand(s, 0, 1), frand(s, 0, 1), frand(s, 0, 1) };
            glm::vec3 n = interpolateNormal(n0, n1, n2, b);
            glm::vec2 t = interpolateTexCoord(t0, t1, t2, b);
            in.insert(in.end(), { n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, n2.x, n2.y, n2.z, t0.x, t0.y, t1.x, t1.y, t2.x, t2.y, b.x, b.y, b.z });
            out.insert(out.end(), { n.x, n.y, n.z, t.x, t.y });
        }
        npy_f32("kat_interp_in", { in.size() / 18, 18 }, in.data());
        npy_f32("kat_interp_out", { out.size() / 5, 5 }, out.data());
    }
}

// Texture fetch KATs need an Image; use the scene's.
static void dump_texture_kats(const Scene& texScene)
{
    const Image* img = nullptr;
    for (const Mesh& m : texScene.meshes) if (m.material.kdTexture) { img = m.material.kdTexture.get(); break; }
    if (!img) return;
    uint32_t s = 777;
    Features f;
    std::vector<float> in, out;
    for (int i = 0; i < 400; i++) {
        glm::vec2 uv { frand(s, 0, 1), frand(s, 0, 1) };
        if (i % 9 == 1) uv = { 0.0f, 0.0f };
        if (i % 9 == 2) uv = { 1.0f, 1.0f };
        if (i % 9 == 3) uv = { 0.999999f, 0.5f };
        if (i % 9 == 4) uv = { 0.5f, 1.0f };
        if (i % 9 == 5) uv = { frand(s, 1, 2), frand(s, 0, 1) }; // wraps (u > 1)
        if (i % 9 == 6) uv = { (float(i % 128) + 0.5f) / 128.0f, 0.25f }; // exact texel centres (round-half cases)
        if (i % 9 == 7) uv = { float(i % 128) / 128.0f, 0.75f };          // texel edges
        glm::vec3 c = acquireTexel(*img, uv, f);
        in.insert(in.end(), { uv.x, uv.y });
        out.insert(out.end(), { c.x, c.y, c.z });
    }
    npy_f32("kat_texel_in", { in.size() / 2, 2 }, in.data());
    npy_f32("kat_texel_out", { out.size() / 3, 3 }, out.data());
    // clamp fetch on the procedural skybox face 0
    std::vector<float> cin, cout;
    for (int i = 0; i < 300; i++) {
        glm::vec2 uv { frand(s, 0, 1), frand(s, 0, 1) };
        if (i % 5 == 1) uv = { 0.0f, 0.0f };
        if (i % 5 == 2) uv = { 0.999f, 1.0f };
        if (i % 5 == 3) uv = { 0.5f, 0.0f };
        glm::vec3 c = acquireTexelClamp(texScene.skybox[size_t(i % 6)], uv, f);
        cin.insert(cin.end(), { uv.x, uv.y, float(i % 6) });
        cout.insert(cout.end(), { c.x, c.y, c.z });
    }
    npy_f32("kat_clamp_in", { cin.size() / 3, 3 }, cin.data());
    npy_f32("kat_clamp_out", { cout.size() / 3, 3 }, cout.data());
}

// ------------------------------------------------------------------------------------------------
struct SceneSpec {
    const char* name;
    SceneType type;
};

static std::vector<glm::vec3> light_sample_positions(const Scene& scene)
{
    std::vector<glm::vec3> out;
    for (const auto& l : scene.lights) {
        if (std::holds_alternative<PointLight>(l)) out.push_back(std::get<PointLight>(l).position);
        else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& s = std::get<SegmentLight>(l);
            for (int i = 0; i < 5; i++) out.push_back(s.endpoint0 + (s.endpoint1 - s.endpoint0) * (float(i) / 4.0f));
        } else {
            const auto& p = std::get<ParallelogramLight>(l);
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++)
                    out.push_back(p.v0 + p.edge01 * ((float(i) + 0.5f) / 4.0f) + p.edge02 * ((float(j) + 0.5f) / 4.0f));
        }
    }
    // a point well outside every scene's root box so the root gate does not mask the result
    out.push_back(glm::vec3(0.3f, 4.0f, -0.2f));
    return out;
}

int main(int argc, char** argv)
{
    if (argc == 5 && std::string(argv[1]) == "--obj") {
        // ref_dump --obj <file.obj> <0|1: centre and normalise> <output prefix>: the reference's loadMesh on any OBJ
        // file (used to pin the facade's own loader on OBJ files written for this repository's tests)
        std::vector<Mesh> meshes = loadMesh(argv[2], std::atoi(argv[3]) != 0);
        std::vector<float> verts, materials;
        std::vector<uint32_t> tris;
        std::vector<int32_t> meshTable; // vtxOff, vtxCnt, triOff, triCnt, has texture
        for (const Mesh& m : meshes) {
            meshTable.insert(meshTable.end(), { int32_t(verts.size() / 8), int32_t(m.vertices.size()), int32_t(tris.size() / 3), int32_t(m.triangles.size()), m.material.kdTexture ? 1 : 0 });
            for (const Vertex& v : m.vertices)
                verts.insert(verts.end(), { v.position.x, v.position.y, v.position.z, v.normal.x, v.normal.y, v.normal.z, v.texCoord.x, v.texCoord.y });
            for (const glm::uvec3& t : m.triangles)
                tris.insert(tris.end(), { t.x, t.y, t.z });
            materials.insert(materials.end(), { m.material.kd.x, m.material.kd.y, m.material.kd.z, m.material.ks.x, m.material.ks.y, m.material.ks.z, m.material.shininess, m.material.transparency });
        }
        g_out = "";
        const std::string prefix = argv[4];
        npy_f32(prefix + "vertices", { verts.size() / 8, 8 }, verts.data());
        npy_u32(prefix + "triangles", { tris.size() / 3, 3 }, tris.data());
        npy_i32(prefix + "mesh_table", { meshTable.size() / 5, 5 }, meshTable.data());
        npy_f32(prefix + "materials", { materials.size() / 8, 8 }, materials.data());
        return 0;
    }
    if (argc == 3 && std::string(argv[1]) == "--config") {
        // ref_dump --config <file.toml>: the reference's readConfigFile and its printer
        std::cout << readConfigFile(argv[2]);
        return 0;
    }
    if (argc == 4 && std::string(argv[1]) == "--image") {
        // ref_dump --image <file> <out.npy>: the reference's Image(path) (stb_image, three channels) as height x width x 3
        // bytes; exit status 3 when the reference throws (missing or undecodable file)
        try {
            Image img(argv[2]);
            std::vector<uint8_t> bytes;
            bytes.reserve(img.pixels.size() * 3);
            for (const glm::vec3& p : img.pixels)
                for (int c = 0; c < 3; c++) bytes.push_back(uint8_t(std::lround(p[c] * 255.0f)));
            g_out = "";
            std::string out = argv[3];
            if (out.size() > 4 && out.substr(out.size() - 4) == ".npy") out.resize(out.size() - 4);
            npy_u8(out, { size_t(img.height), size_t(img.width), 3 }, bytes.data());
        } catch (const std::exception&) {
            return 3;
        }
        return 0;
    }
    if (argc >= 2 && std::string(argv[1]) == "--time") {
        // ref_dump --time [data dir]: wall time of the compiled reference's per-pixel evaluation on one thread (its
        // lighting code keeps scratch in globals, so the harness does not run it on several), for the ratio between the
        // reference as it is and the oracle on the same frames.
        std::filesystem::path dir = argc > 2 ? argv[2] : DATA_DIR;
        struct Job { const char* name; SceneType type; int W, H; bool soft; int n; };
        for (const Job& j : { Job { "cornell_para", CornellBoxParallelogramLight, 256, 256, true, 1 }, Job { "cornell_para_4spp", CornellBoxParallelogramLight, 128, 128, true, 2 },
                 Job { "teapot", Teapot, 128, 128, false, 1 }, Job { "monkey", Monkey, 256, 256, false, 1 } }) {
            Scene scene = loadScenePrebuilt(j.type, dir);
            Features f = feat(true, false, false, j.soft, false, false);
            f.extra.enableMultipleRaysPerPixel = j.n > 1;
            BvhInterface bvh(&scene, f);
            RenderOpts o; o.rayMultiplier = j.n;
            rayMultiplier = j.n;
            auto t0 = std::chrono::steady_clock::now();
            auto img = render_image(scene, bvh, CAM_A, j.W, j.H, f, o);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            rayMultiplier = 1;
            std::printf("%s %d %d %d %.1f\n", j.name, j.W, j.H, j.n, ms);
            std::fflush(stdout);
        }
        return 0;
    }
    if (argc == 6 && std::string(argv[1]) == "--time-obj") {
        // ref_dump --time-obj <file.obj> <width> <height> <n>: the same wall time for any mesh under the benchmark's
        // parallelogram light (the reference's Cornell light, src/scene.cpp:45-53, lifted to y = 1.2), shading + soft
        // shadows, n x n samples per pixel, default camera.  Prints "<triangles> <width> <height> <n> <ms>".
        Scene scene;
        scene.meshes = loadMesh(argv[2], false);
        scene.lights.emplace_back(ParallelogramLight { glm::vec3 { -0.2f, 1.2f, -0.2f }, glm::vec3 { 0.4f, 0.0f, 0.0f }, glm::vec3 { 0.0f, 0.0f, 0.4f },
            glm::vec3 { 1, 0, 0 }, glm::vec3 { 0, 1, 0 }, glm::vec3 { 0, 0, 1 }, glm::vec3 { 1, 1, 1 } });
        const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), n = std::atoi(argv[5]);
        size_t triangles = 0;
        for (const Mesh& m : scene.meshes) triangles += m.triangles.size();
        Features f = feat(true, false, false, true, false, false);
        f.extra.enableMultipleRaysPerPixel = n > 1;
        BvhInterface bvh(&scene, f);
        RenderOpts o; o.rayMultiplier = n;
        rayMultiplier = n;
        auto t0 = std::chrono::steady_clock::now();
        auto img = render_image(scene, bvh, CAM_A, W, H, f, o);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%zu %d %d %d %.1f\n", triangles, W, H, n, ms);
        return 0;
    }
    if (argc < 2) { std::fprintf(stderr, "usage: ref_dump <outdir> [data dir] | --time [data dir] | --time-obj <obj> <width> <height> <n>\n"); return 2; }
    g_out = argv[1];
    std::filesystem::create_directories(g_out);
    std::filesystem::path dataDir = argc > 2 ? argv[2] : DATA_DIR;

    const int WA = 72, HA = 48, WB = 48, HB = 48;

    const SceneSpec scenes[] = {
        { "cube_segment", Cube }, { "cube_textured", CubeTextured }, { "cornell", CornellBox },
        { "cornell_para", CornellBoxParallelogramLight }, { "monkey", Monkey }, { "teapot", Teapot },
        { "quad_textured", QuadTextured }, { "transparent_cubes", TransparentCubes }, { "transparent_cornell", TransparentCornell },
        { "single_triangle", SingleTriangle }, { "spheres", Spheres },
    };

    dump_kats();

    for (const SceneSpec& spec : scenes) {
        std::string nm = spec.name;
        std::fprintf(stderr, "[ref_dump] scene %s\n", spec.name);
        Scene scene = loadScenePrebuilt(spec.type, dataDir);
        install_procedural_skybox(scene);
        dump_scene(nm + ".scene.", scene);
        if (nm == "cube_textured") dump_texture_kats(scene);

        Features base = feat(false, false, false, false, false, false);
        BvhInterface bvh(&scene, base);
        dump_bvh(nm + ".bvh.", bvh);

        // --- hit records.  "prune" = neither recursive nor transparency (traversal prunes by best hit).
        Features fFull = feat(false, false, false, false, true, true);
        Features fNoPrune = feat(false, true, false, false, false, false);
        bool big = nm == "teapot";
        dump_hits(nm + ".A.prune.", scene, bvh, CAM_A, WA, HA, fFull, true, nm == "cornell");
        dump_hits(nm + ".A.noprune.", scene, bvh, CAM_A, WA, HA, fNoPrune, false, false);
        if (nm == "cornell" || nm == "monkey" || nm == "teapot" || nm == "transparent_cubes") {
            dump_hits(nm + ".B.prune.", scene, bvh, CAM_B, WB, HB, fFull, true, nm == "cornell");
            dump_hits(nm + ".B.noprune.", scene, bvh, CAM_B, WB, HB, fNoPrune, false, false);
        }
        if (!big) {
            // brute-force path for cross-checking (src/bounding_volume_hierarchy.cpp:462-528)
            Features fBrute = feat(false, false, false, false, true, true, false);
            dump_hits(nm + ".A.brute.", scene, bvh, CAM_A, WA, HA, fBrute, true, false);
        }
        // --- shadow visibility (both pruning modes)
        {
            std::vector<glm::vec3> samples = light_sample_positions(scene);
            Features fv = feat(true, false, true, true, false, false);
            dump_visibility(nm + ".A.prune.", scene, bvh, CAM_A, WA, HA, fv, samples);
            Features fv2 = fv; fv2.enableRecursive = true;
            dump_visibility(nm + ".A.noprune.", scene, bvh, CAM_A, WA, HA, fv2, samples);
        }
        // --- images
        auto img = [&](const char* tag, const Cam& cam, int W, int H, Features f, RenderOpts o = {}) {
            dump_image(nm + ".img." + tag, render_image(scene, bvh, cam, W, H, f, o), W, H);
        };
        extr_enabledSkyBox = false; extr_enabledReflMap = false;
        rayMultiplier = 1;
        extr_glossy_filterSize = 64; extr_glossy_sigma = 1; extr_dof_samples = 3; extr_dof_aperture = 1.4f; extr_dof_f = 2.5f; extr_dof = 0.0f;
        threshold = 0.7f; bloomsize = 2; sigma = 1.0f; gaussian = 1; scale = 1.0f; debugBloom = false;
        samplesPerUnit = 10; samplesPerUnitParallel = 10;

        img("A.unshaded", CAM_A, WA, HA, feat(false, false, false, false, false, false));
        img("A.shading", CAM_A, WA, HA, feat(true, false, false, false, false, false));
        if (!big) {
            img("A.hard", CAM_A, WA, HA, feat(true, false, true, false, false, false));
            img("A.recursive", CAM_A, WA, HA, feat(true, true, false, false, false, false));
            img("A.all_det", CAM_A, WA, HA, feat(true, true, true, false, true, true));
            img("B.all_det", CAM_B, WB, HB, feat(true, true, true, false, true, true));
            img("A.soft", CAM_A, WA, HA, feat(true, false, true, true, false, false));
            img("A.soft_recursive", CAM_A, WA, HA, feat(true, true, true, true, true, true));
            {
                Features f = feat(true, false, false, false, false, true);
                f.extra.enableEnvironmentMapping = true;
                extr_enabledSkyBox = true;
                img("A.sky", CAM_A, WA, HA, f);
                extr_enabledReflMap = true;
                img("A.sky_refl", CAM_A, WA, HA, f);
                extr_enabledSkyBox = false; extr_enabledReflMap = false;
            }
            {
                Features f = feat(true, false, true, false, false, false);
                f.extra.enableTransparency = true;
                img("A.transparency", CAM_A, WA, HA, f);
                f.enableRecursive = true;
                img("A.transparency_recursive", CAM_A, WA, HA, f);
            }
            {
                Features f = feat(true, false, false, true, false, false);
                f.extra.enableMultipleRaysPerPixel = true;
                RenderOpts o; o.rayMultiplier = 2;
                rayMultiplier = 2;
                img("A.multi2_soft", CAM_A, WA, HA, f, o);
                rayMultiplier = 1;
            }
            {
                Features f = feat(true, true, false, false, false, false);
                f.extra.enableBloomEffect = true;
                img("A.bloom_gauss", CAM_A, WA, HA, f);
                gaussian = 0; bloomsize = 3; threshold = 0.5f; scale = 1.5f;
                img("A.bloom_box", CAM_A, WA, HA, f);
                gaussian = 1; bloomsize = 2; threshold = 0.7f; scale = 1.0f;
                debugBloom = true;
                img("B.bloom_debug", CAM_B, WB, HB, f);
                debugBloom = false;
            }
            {
                Features f = feat(true, false, false, false, false, false);
                f.extra.enableDepthOfField = true;
                extr_dof_f = 3.0f; extr_dof = 0.05f; extr_dof_samples = 3;
                img("A.dof", CAM_A, WA, HA, f);
                extr_dof_f = 2.5f; extr_dof = 0.0f;
            }
        }
        if (nm == "cornell" || nm == "transparent_cornell" || nm == "monkey") {
            // glossy: 64^k fan-out, keep the frame tiny and the filter small
            Features f = feat(true, false, false, false, false, false);
            f.extra.enableGlossyReflection = true;
            extr_glossy_filterSize = 8;
            dump_image(nm + ".img.G.glossy8", render_image(scene, bvh, CAM_A, 36, 24, f, {}), 36, 24);
            if (nm == "transparent_cornell") {
                f.extra.enableTransparency = true;
                dump_image(nm + ".img.G.glossy8_transparency", render_image(scene, bvh, CAM_A, 36, 24, f, {}), 36, 24);
            }
            extr_glossy_filterSize = 64;
        }
        // --- the benchmark configurations' feature sets at fixture size (group "configs" of the golden vectors)
        if (!big) {
            // Config 2: textures + sky box on a miss + 2 x 2 supersampling
            Features f = feat(true, false, false, false, false, true);
            f.extra.enableEnvironmentMapping = true;
            f.extra.enableMultipleRaysPerPixel = true;
            extr_enabledSkyBox = true;
            RenderOpts o; o.rayMultiplier = 2;
            rayMultiplier = 2;
            dump_image("configs." + nm + ".img.A.c2", render_image(scene, bvh, CAM_A, WA, HA, f, o), WA, HA);
            dump_image("configs." + nm + ".img.B.c2", render_image(scene, bvh, CAM_B, WB, HB, f, o), WB, HB);
            rayMultiplier = 1;
            extr_enabledSkyBox = false;
        }
        if (nm == "transparent_cornell" || nm == "transparent_cubes") {
            // Config 4: transparency + glossy reflection with the reference's default 64 samples per bounce,
            // 2 x 2 supersampling
            Features f = feat(true, false, false, false, false, false);
            f.extra.enableGlossyReflection = true;
            f.extra.enableTransparency = true;
            f.extra.enableMultipleRaysPerPixel = true;
            RenderOpts o; o.rayMultiplier = 2;
            rayMultiplier = 2;
            extr_glossy_filterSize = 64;
            dump_image("configs." + nm + ".img.A.c4", render_image(scene, bvh, CAM_A, 48, 32, f, o), 48, 32);
            rayMultiplier = 1;
        }
        if (nm == "cornell") {
            // Config 1 at full size: 512x512, shading + recursive + BVH, 1 spp.  Only digests are kept.
            auto full = render_image(scene, bvh, CAM_A, 512, 512, feat(true, true, false, false, false, false), {});
            uint64_t h = fnv1a(full.data(), full.size() * sizeof(glm::vec3));
            uint32_t parts[2] = { uint32_t(h & 0xffffffffu), uint32_t(h >> 32) };
            npy_u32(nm + ".c1_fnv1a64", { 2 }, parts);
            std::vector<double> rowsum(512 * 3, 0.0);
            for (int y = 0; y < 512; y++)
                for (int x = 0; x < 512; x++)
                    for (int c = 0; c < 3; c++) rowsum[size_t(y) * 3 + size_t(c)] += double(full[size_t(y) * 512 + size_t(x)][c]);
            npy_f64(nm + ".c1_rowsum", { 512, 3 }, rowsum.data());
        }
    }

    // SAH + binning build (src/bounding_volume_hierarchy.cpp:133-190): pin node order on two scenes.
    for (const SceneSpec& spec : { SceneSpec { "cornell", CornellBox }, SceneSpec { "monkey", Monkey } }) {
        Scene scene = loadScenePrebuilt(spec.type, dataDir);
        Features f = feat(false, false, false, false, false, false);
        f.extra.enableBvhSahBinning = true;
        BvhInterface bvh(&scene, f);
        dump_bvh(std::string(spec.name) + ".bvh_sah.", bvh);
    }
    std::fprintf(stderr, "[ref_dump] done\n");
    return 0;
}
