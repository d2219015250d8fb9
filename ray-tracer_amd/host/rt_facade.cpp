#include "rt_facade.h"
#include <string>
#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace rtamd {

int samplesPerUnit = 10;
int samplesPerUnitParallel = 10;
int rayMultiplier = 1;
int extr_glossy_filterSize = 64;
float extr_glossy_sigma = 1;
int extr_dof_samples = 3;
float extr_dof_aperture = 1.4f, extr_dof_f = 2.5f, extr_dof = 0.0f;
float threshold = 0.7f;
int bloomsize = 2;
float sigma = 1.0f;
bool debugBloom = false;
int gaussian = 1;
float scale = 1.0f;
bool extr_enabledSkyBox = false, extr_enabledReflMap = false;
bool tiledGatherOnDevice = false;
int extr_max_level = 32, extr_sah_bins = 64;
unsigned renderSeed = 1;

namespace {

thread_local rt_stats g_lastStats {};

void put3(float* dst, const vec3& v)
{
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

rt_features to_abi(const Features& f)
{
    rt_features o;
    std::memset(&o, 0, sizeof o);
    o.shading = f.enableShading;
    o.recursive = f.enableRecursive;
    o.hard_shadow = f.enableHardShadow;
    o.soft_shadow = f.enableSoftShadow;
    o.normal_interp = f.enableNormalInterp;
    o.texture_mapping = f.enableTextureMapping;
    o.accel_structure = f.enableAccelStructure;
    o.environment_mapping = f.extra.enableEnvironmentMapping;
    o.bvh_sah_binning = f.extra.enableBvhSahBinning;
    o.motion_blur = f.extra.enableMotionBlur;
    o.bloom = f.extra.enableBloomEffect;
    o.bilinear_filtering = f.extra.enableBilinearTextureFiltering;
    o.mipmap_filtering = f.extra.enableMipmapTextureFiltering;
    o.multiple_rays = f.extra.enableMultipleRaysPerPixel;
    o.glossy = f.extra.enableGlossyReflection;
    o.transparency = f.extra.enableTransparency;
    o.depth_of_field = f.extra.enableDepthOfField;
    return o;
}

// snapshot of the tunable globals, taken at each call like the reference reads them
rt_params snapshot_params()
{
    rt_params p;
    rt_default_params(&p);
    p.samples_per_unit = samplesPerUnit;
    p.samples_per_unit_parallel = samplesPerUnitParallel;
    p.ray_multiplier = rayMultiplier;
    p.glossy_filter_size = extr_glossy_filterSize;
    p.glossy_sigma = extr_glossy_sigma;
    p.dof_samples = extr_dof_samples;
    p.dof_aperture = extr_dof_aperture;
    p.dof_f = extr_dof_f;
    p.dof = extr_dof;
    p.bloom_threshold = threshold;
    p.bloom_size = bloomsize;
    p.bloom_sigma = sigma;
    p.bloom_debug = debugBloom;
    p.bloom_gaussian = gaussian;
    p.bloom_scale = scale;
    p.enabled_skybox = extr_enabledSkyBox;
    p.enabled_refl_map = extr_enabledReflMap;
    p.seed = renderSeed;
    return p;
}

std::vector<rt_light> to_abi(const Scene& scene)
{
    std::vector<rt_light> out;
    for (const auto& l : scene.lights) {
        rt_light r;
        std::memset(&r, 0, sizeof r);
        if (std::holds_alternative<PointLight>(l)) {
            const auto& p = std::get<PointLight>(l);
            r.type = RT_LIGHT_POINT;
            put3(r.v, p.position);
            put3(r.v + 3, p.color);
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& s = std::get<SegmentLight>(l);
            r.type = RT_LIGHT_SEGMENT;
            put3(r.v, s.endpoint0);
            put3(r.v + 3, s.endpoint1);
            put3(r.v + 6, s.color0);
            put3(r.v + 9, s.color1);
        } else {
            const auto& p = std::get<ParallelogramLight>(l);
            r.type = RT_LIGHT_PARALLELOGRAM;
            put3(r.v, p.v0);
            put3(r.v + 3, p.edge01);
            put3(r.v + 6, p.edge02);
            put3(r.v + 9, p.color0);
            put3(r.v + 12, p.color1);
            put3(r.v + 15, p.color2);
            put3(r.v + 18, p.color3);
        }
        out.push_back(r);
    }
    return out;
}

// glm's operation order (quaternion from Euler angles, q * v), as in the camera code of the library
struct Quat { float w, x, y, z; };
vec3 cross(vec3 a, vec3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
Quat quat_of(vec3 e)
{
    const float cx = std::cos(e.x * 0.5f), cy = std::cos(e.y * 0.5f), cz = std::cos(e.z * 0.5f);
    const float sx = std::sin(e.x * 0.5f), sy = std::sin(e.y * 0.5f), sz = std::sin(e.z * 0.5f);
    return { cx * cy * cz + sx * sy * sz, sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz };
}
vec3 rotate(Quat q, vec3 v)
{
    const vec3 qv { q.x, q.y, q.z };
    const vec3 uv = cross(qv, v);
    const vec3 uuv = cross(qv, uv);
    return { v.x + ((uv.x * q.w) + uuv.x) * 2.0f, v.y + ((uv.y * q.w) + uuv.y) * 2.0f, v.z + ((uv.z * q.w) + uuv.z) * 2.0f };
}

} // namespace

// ---------------------------------------------------------------------------------------------------
BvhInterface::BvhInterface(Scene* pScene, const Features& features)
    : m_pScene(pScene)
{
    std::vector<rt_mesh_desc> meshes(pScene->meshes.size());
    std::vector<const Image*> textures;
    for (size_t i = 0; i < pScene->meshes.size(); i++) {
        const Mesh& m = pScene->meshes[i];
        rt_mesh_desc& d = meshes[i];
        static_assert(sizeof(Vertex) == 32 && sizeof(uvec3) == 12, "Vertex / triangle layout must match the ABI");
        d.vertices = reinterpret_cast<const float*>(m.vertices.data());
        d.triangles = reinterpret_cast<const uint32_t*>(m.triangles.data());
        d.n_vertices = uint32_t(m.vertices.size());
        d.n_triangles = uint32_t(m.triangles.size());
        put3(d.kd, m.material.kd);
        put3(d.ks, m.material.ks);
        d.shininess = m.material.shininess;
        d.transparency = m.material.transparency;
        d.texture = -1;
        if (m.material.kdTexture) {
            const Image* img = m.material.kdTexture.get();
            auto it = std::find(textures.begin(), textures.end(), img);
            if (it == textures.end()) {
                textures.push_back(img);
                d.texture = int32_t(textures.size() - 1);
            } else {
                d.texture = int32_t(it - textures.begin());
            }
        }
    }
    std::vector<rt_image_desc> tex;
    for (const Image* img : textures) tex.push_back({ img->rgb.data(), img->width, img->height });
    std::vector<rt_sphere_desc> spheres;
    for (const Sphere& s : pScene->spheres) {
        rt_sphere_desc d;
        put3(d.center, s.center);
        d.radius = s.radius;
        put3(d.kd, s.material.kd);
        put3(d.ks, s.material.ks);
        d.shininess = s.material.shininess;
        d.transparency = s.material.transparency;
        spheres.push_back(d);
    }
    std::vector<rt_image_desc> sky;
    if (pScene->skybox.size() == 6)
        for (const Image& f : pScene->skybox) sky.push_back({ f.rgb.data(), f.width, f.height });
    rt_scene_desc desc;
    desc.meshes = meshes.data();
    desc.n_meshes = uint32_t(meshes.size());
    desc.textures = tex.data();
    desc.n_textures = uint32_t(tex.size());
    desc.spheres = spheres.data();
    desc.n_spheres = uint32_t(spheres.size());
    desc.skybox = sky.empty() ? nullptr : sky.data();
    if (rt_scene_create(&desc, &m_scene) != RT_OK) {
        std::fprintf(stderr, "BvhInterface: %s\n", rt_last_error());
        return;
    }
    rt_bvh_options opt { features.extra.enableBvhSahBinning ? 1 : 0, extr_max_level, extr_sah_bins };
    // built on the calling thread's current GPU (same tree, tens of times faster on large meshes); without a usable GPU,
    // on the host (the tables are then uploaded by the first call that renders)
    if (rt_bvh_build_device(m_scene, &opt, &m_bvh) == RT_OK) return;
    if (rt_bvh_build(m_scene, &opt, &m_bvh) != RT_OK) std::fprintf(stderr, "BvhInterface: %s\n", rt_last_error());
}

BvhInterface::~BvhInterface()
{
    rt_bvh_destroy(m_bvh);
    rt_scene_destroy(m_scene);
}

int BvhInterface::numLevels() const { return rt_bvh_num_levels(m_bvh); }
int BvhInterface::numLeaves() const { return rt_bvh_num_leaves(m_bvh); }
void BvhInterface::debugDrawLevel(int) {}
void BvhInterface::debugDrawLeaf(int) {}

bool BvhInterface::intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const
{
    if (!m_bvh) return false;
    rt_ray r;
    put3(r.origin, ray.origin);
    put3(r.direction, ray.direction);
    r.t = ray.t;
    rt_hit h;
    const rt_features f = to_abi(features);
    if (rt_intersect_batch(m_bvh, &r, 1, &f, &h, 0, nullptr) != RT_OK) {
        std::fprintf(stderr, "BvhInterface::intersect: %s\n", rt_last_error());
        return false;
    }
    if (!h.hit) return false;
    ray.t = h.t;
    hitInfo.normal = { h.normal[0], h.normal[1], h.normal[2] };
    hitInfo.barycentricCoord = { h.bary[0], h.bary[1], h.bary[2] };
    if (h.mesh >= 0) hitInfo.material = m_pScene->meshes[size_t(h.mesh)].material;
    else hitInfo.material = m_pScene->spheres[size_t(h.triangle)].material;
    hitInfo.material.kd = { h.kd[0], h.kd[1], h.kd[2] };
    return true;
}

// ---------------------------------------------------------------------------------------------------
Trackball::Trackball(float aspect, float fovy, const vec3& lookAt, float distFromLookAt, float rotationX, float rotationY)
    : m_aspect(aspect), m_fovy(fovy), m_lookAt(lookAt), m_distanceFromLookAt(distFromLookAt), m_rotationEulerAngles(rotationX, rotationY, 0)
{
}
void Trackball::setCamera(const vec3 lookAt, const vec3 rotations, const float dist)
{
    m_lookAt = lookAt;
    m_rotationEulerAngles = rotations;
    m_distanceFromLookAt = dist;
}
vec3 Trackball::left() const { return rotate(quat_of(m_rotationEulerAngles), { 1, 0, 0 }); }
vec3 Trackball::up() const { return rotate(quat_of(m_rotationEulerAngles), { 0, 1, 0 }); }
vec3 Trackball::forward() const { return rotate(quat_of(m_rotationEulerAngles), { 0, 0, 1 }); }
vec3 Trackball::position() const
{
    const vec3 p = rotate(quat_of(m_rotationEulerAngles), { 0, 0, -m_distanceFromLookAt });
    return { m_lookAt.x + p.x, m_lookAt.y + p.y, m_lookAt.z + p.z };
}
Ray Trackball::generateRay(const vec2& pixel) const
{
    const float halfH = std::tan(m_fovy / 2.0f), halfW = m_aspect * halfH;
    vec3 c { -pixel.x * halfW, pixel.y * halfH, 1.0f };
    const float inv = 1.0f / std::sqrt((c.x * c.x + c.y * c.y) + c.z * c.z);
    c = { c.x * inv, c.y * inv, c.z * inv };
    Ray r;
    r.origin = position();
    r.direction = rotate(quat_of(m_rotationEulerAngles), c);
    r.t = FLT_MAX;
    return r;
}
rt_camera Trackball::abi() const
{
    rt_camera c;
    c.fovy = m_fovy;
    c.aspect = m_aspect;
    put3(c.look_at, m_lookAt);
    c.distance = m_distanceFromLookAt;
    put3(c.rotation, m_rotationEulerAngles);
    return c;
}

// ---------------------------------------------------------------------------------------------------
Screen::Screen(const ivec2& resolution, bool)
    : m_resolution(resolution), m_textureData(size_t(resolution.x) * size_t(resolution.y), vec3(0.0f))
{
}
void Screen::clear(const vec3& color) { std::fill(m_textureData.begin(), m_textureData.end(), color); }
void Screen::setPixel(int x, int y, const vec3& color) { m_textureData[size_t(indexAt(x, y))] = color; }

// 32-bit BGRA bitmap, top row first in the buffer; channel = clamp(c, 0, 1) * 255 truncated (reference
// src/screen.cpp:45-56).
void Screen::writeBitmapToFile(const std::filesystem::path& filePath)
{
    const int w = m_resolution.x, h = m_resolution.y;
    const uint32_t dataBytes = uint32_t(w) * uint32_t(h) * 4u;
    unsigned char header[54] = { 'B', 'M' };
    auto put32 = [&](int at, uint32_t v) { std::memcpy(header + at, &v, 4); };
    put32(2, 54u + dataBytes);
    put32(10, 54u);
    put32(14, 40u);
    put32(18, uint32_t(w));
    put32(22, uint32_t(h)); // positive height: rows stored bottom-up
    header[26] = 1;
    header[28] = 32;
    put32(34, dataBytes);
    std::ofstream f(filePath, std::ios::binary);
    f.write(reinterpret_cast<const char*>(header), 54);
    std::vector<unsigned char> row(size_t(w) * 4);
    for (int y = h - 1; y >= 0; y--) {
        for (int x = 0; x < w; x++) {
            const vec3& c = m_textureData[size_t(y) * size_t(w) + size_t(x)];
            auto q = [](float v) { return static_cast<unsigned char>(std::min(std::max(v, 0.0f), 1.0f) * 255.0f); };
            row[size_t(x) * 4 + 0] = q(c.z);
            row[size_t(x) * 4 + 1] = q(c.y);
            row[size_t(x) * 4 + 2] = q(c.x);
            row[size_t(x) * 4 + 3] = 255;
        }
        f.write(reinterpret_cast<const char*>(row.data()), std::streamsize(row.size()));
    }
}

// ---------------------------------------------------------------------------------------------------
void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features)
{
    if (!bvh.handle()) return;
    const ivec2 res = screen.resolution();
    const rt_features f = to_abi(features);
    const rt_params p = snapshot_params();
    const rt_camera cam = camera.abi();
    const std::vector<rt_light> lights = to_abi(scene);
    static_assert(sizeof(vec3) == 12, "vec3 must be three packed floats");
    float* out = reinterpret_cast<float*>(screen.pixels().data());
    rt_stats stats;
    if (rt_render(bvh.handle(), &cam, lights.data(), int32_t(lights.size()), &f, &p, res.x, res.y, 0, res.y, out, 0, nullptr, &stats) != RT_OK) {
        std::fprintf(stderr, "renderRayTracing: %s\n", rt_last_error());
        return;
    }
    g_lastStats = stats;
    if (features.extra.enableBloomEffect && rt_bloom(out, res.x, res.y, &p, 0, nullptr) != RT_OK)
        std::fprintf(stderr, "renderRayTracing (bloom): %s\n", rt_last_error());
}

void renderRayTracingTiled(const Scene& scene, const Trackball& camera, const std::vector<RenderShare>& shares, Screen& screen, const Features& features)
{
    const int n = int(shares.size());
    if (n == 0) return;
    const ivec2 res = screen.resolution();
    const rt_features f = to_abi(features);
    const rt_params p = snapshot_params();
    const rt_camera cam = camera.abi();
    const std::vector<rt_light> lights = to_abi(scene);
    constexpr int rowBlock = 16;
    const int nBlocks = (res.y + rowBlock - 1) / rowBlock;
    std::vector<std::vector<float>> parts { size_t(n) };
    std::vector<std::string> errors { size_t(n) };
    std::vector<std::thread> threads;
    // GPU-to-GPU gather: a communicator per share, which needs every share on a device of its own
    std::vector<rt_comm*> comms(size_t(n), nullptr);
    struct CommGuard {
        std::vector<rt_comm*>& comms;
        ~CommGuard()
        {
            for (rt_comm* c : comms) rt_comm_destroy(c);
        }
    } commGuard { comms };
    bool onDevice = tiledGatherOnDevice;
    for (int i = 0; i < n && onDevice; i++)
        for (int j = 0; j < i; j++)
            if (shares[size_t(i)].device == shares[size_t(j)].device) onDevice = false;
    if (onDevice) {
        std::vector<int32_t> devices;
        for (const RenderShare& sh : shares) devices.push_back(sh.device);
        if (rt_comm_init_all(n, devices.data(), comms.data()) != RT_OK) {
            std::fprintf(stderr, "renderRayTracingTiled: no GPU-to-GPU gather (%s); the shares come back through host memory\n", rt_last_error());
            onDevice = false;
        }
    }
    float* const screenPixels = reinterpret_cast<float*>(screen.pixels().data());
    // (whatever happens below, every thread that was started is joined before this function is left)
    struct Joiner {
        std::vector<std::thread>& threads;
        ~Joiner()
        {
            for (std::thread& t : threads)
                if (t.joinable()) t.join();
        }
    } joiner { threads };
    auto share_job = [&](int i) {
            if (!shares[size_t(i)].bvh || !shares[size_t(i)].bvh->handle()) {
                errors[size_t(i)] = "share " + std::to_string(i) + " has no hierarchy";
                return;
            }
            if (onDevice) {
                // every rank of the communicator takes part, also one without rows; rank 0 receives the frame
                if (rt_init(shares[size_t(i)].device) != RT_OK ||
                    rt_render_blocks_gather(shares[size_t(i)].bvh->handle(), &cam, lights.data(), int32_t(lights.size()), &f, &p, res.x, res.y, rowBlock,
                        comms[size_t(i)], 0, i == 0 ? screenPixels : nullptr, 0, nullptr, nullptr) != RT_OK)
                    errors[size_t(i)] = rt_last_error();
                return;
            }
            int rows = 0;
            for (int b = i; b < nBlocks; b += n) rows += std::min(res.y, (b + 1) * rowBlock) - b * rowBlock;
            parts[size_t(i)].assign(size_t(rows) * size_t(res.x) * 3, 0.0f);
            if (rows == 0) return;
            if (rt_init(shares[size_t(i)].device) != RT_OK ||
                rt_render_blocks(shares[size_t(i)].bvh->handle(), &cam, lights.data(), int32_t(lights.size()), &f, &p, res.x, res.y, rowBlock, i, n,
                    parts[size_t(i)].data(), 0, nullptr, nullptr) != RT_OK)
                errors[size_t(i)] = rt_last_error(); // (the error text belongs to the thread that made the call)
    };
    try {
        for (int i = 0; i < n; i++) threads.emplace_back(share_job, i);
    } catch (const std::exception& e) {
        for (std::thread& t : threads) t.join();
        std::fprintf(stderr, "renderRayTracingTiled: cannot start a thread per share: %s\n", e.what());
        return;
    }
    for (std::thread& t : threads) t.join();
    for (const std::string& e : errors)
        if (!e.empty()) {
            std::fprintf(stderr, "renderRayTracingTiled: %s\n", e.c_str());
            return;
        }
    float* out = screenPixels;
    for (int i = 0; i < n && !onDevice; i++) {
        size_t at = 0;
        for (int b = i; b < nBlocks; b += n) {
            const int top = b * rowBlock, bottom = std::min(res.y, top + rowBlock);
            const size_t count = size_t(bottom - top) * size_t(res.x) * 3;
            std::memcpy(out + size_t(top) * size_t(res.x) * 3, parts[size_t(i)].data() + at, count * sizeof(float));
            at += count;
        }
    }
    if (features.extra.enableBloomEffect && rt_bloom(out, res.x, res.y, &p, 0, nullptr) != RT_OK)
        std::fprintf(stderr, "renderRayTracingTiled (bloom): %s\n", rt_last_error());
}

const rt_stats& lastRenderStats() { return g_lastStats; }

// ---------------------------------------------------------------------------------------------------
Scene loadSceneBinary(const std::filesystem::path& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::exception(); // the reference's loaders throw a bare std::exception (framework/src/mesh.cpp:54-57)
    auto rd = [&](void* dst, size_t n) {
        f.read(static_cast<char*>(dst), std::streamsize(n));
        if (!f) throw std::exception();
    };
    char magic[8];
    rd(magic, 8);
    if (std::memcmp(magic, "RTSCENE1", 8) != 0) throw std::exception();
    uint32_t head[8];
    rd(head, sizeof head);
    const uint32_t nMeshes = head[0], nTextures = head[1], nSpheres = head[2], nLights = head[3], hasSky = head[4];
    Scene scene;
    std::vector<int32_t> meshTexture(nMeshes);
    for (uint32_t m = 0; m < nMeshes; m++) {
        uint32_t counts[2];
        rd(counts, sizeof counts);
        float mat[8];
        rd(mat, sizeof mat);
        rd(&meshTexture[m], 4);
        Mesh mesh;
        mesh.vertices.resize(counts[0]);
        mesh.triangles.resize(counts[1]);
        rd(mesh.vertices.data(), size_t(counts[0]) * sizeof(Vertex));
        rd(mesh.triangles.data(), size_t(counts[1]) * sizeof(uvec3));
        mesh.material.kd = { mat[0], mat[1], mat[2] };
        mesh.material.ks = { mat[3], mat[4], mat[5] };
        mesh.material.shininess = mat[6];
        mesh.material.transparency = mat[7];
        scene.meshes.push_back(std::move(mesh));
    }
    auto readImage = [&]() {
        int32_t wh[2];
        rd(wh, sizeof wh);
        std::vector<uint8_t> px((size_t(wh[0]) * size_t(wh[1]) * 3 + 3) / 4 * 4);
        rd(px.data(), px.size());
        px.resize(size_t(wh[0]) * size_t(wh[1]) * 3);
        return Image(wh[0], wh[1], std::move(px));
    };
    std::vector<std::shared_ptr<Image>> textures;
    for (uint32_t t = 0; t < nTextures; t++) textures.push_back(std::make_shared<Image>(readImage()));
    for (uint32_t m = 0; m < nMeshes; m++)
        if (meshTexture[m] >= 0) scene.meshes[m].material.kdTexture = textures[size_t(meshTexture[m])];
    for (uint32_t s = 0; s < nSpheres; s++) {
        float v[12];
        rd(v, sizeof v);
        Sphere sp;
        sp.center = { v[0], v[1], v[2] };
        sp.radius = v[3];
        sp.material.kd = { v[4], v[5], v[6] };
        sp.material.ks = { v[7], v[8], v[9] };
        sp.material.shininess = v[10];
        sp.material.transparency = v[11];
        scene.spheres.push_back(sp);
    }
    for (uint32_t l = 0; l < nLights; l++) {
        float v[22];
        rd(v, sizeof v);
        auto at = [&](int i) { return vec3(v[i], v[i + 1], v[i + 2]); };
        if (int(v[0]) == RT_LIGHT_POINT) scene.lights.emplace_back(PointLight { at(1), at(4) });
        else if (int(v[0]) == RT_LIGHT_SEGMENT) scene.lights.emplace_back(SegmentLight { at(1), at(4), at(7), at(10) });
        else scene.lights.emplace_back(ParallelogramLight { at(1), at(4), at(7), at(10), at(13), at(16), at(19) });
    }
    if (hasSky)
        for (int i = 0; i < 6; i++) scene.skybox.push_back(readImage());
    return scene;
}

} // namespace rtamd
