// C++ host facade over the C ABI of librt_amd (include/rt_amd.h): the reference ray tracer's own surface for
// the render path — Scene, Features, BvhInterface, Trackball, Screen, renderRayTracing and the tunable
// globals — with the same names, argument meaning and (absence of) error behaviour, so that code written
// against the reference's headers reads the same against this one.  Everything lives in namespace rtamd
// because glm is not available here; the vector types are plain structs with the layout of glm's.
//
//   reference                                   here
//   src/common.h:14-77   HitInfo, lights,       same names and fields
//                        Features/ExtraFeatures
//   framework mesh.h:14-43, image.h:11-18       Vertex, Material, Mesh, Image (8-bit texels instead of float)
//   src/scene.h:31-37    Scene                  same fields
//   src/bvh_interface.h  BvhInterface           same members; frees what it owns (the reference leaks)
//   framework trackball.h Trackball             camera part only; takes the aspect ratio instead of a Window*
//   src/screen.h         Screen                 same members except draw() (OpenGL)
//   src/render.h:17      renderRayTracing       same signature
//   tunable globals      light.cpp:11,16  multipleRays.cpp:11  gloss.cpp:9-10  dof.cpp:9-12  bloom.cpp:5-11
//                        environment_mapping.cpp:5-6  bounding_volume_hierarchy.cpp:16-17
#pragma once
#include "rt_amd.h"

#include <cfloat>
#include <cstdint>
#include <filesystem>
#include <memory>
#include <string>
#include <variant>
#include <vector>

namespace rtamd {

struct vec2 { float x = 0, y = 0; };
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float a) : x(a), y(a), z(a) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct ivec2 { int x = 0, y = 0; };
struct uvec3 { uint32_t x = 0, y = 0, z = 0; };

struct Ray {
    vec3 origin { 0.0f };
    vec3 direction { 0.0f, 0.0f, -1.0f };
    float t { FLT_MAX };
};

struct Image {
    Image(int w, int h, std::vector<uint8_t> rgbBytes) : width(w), height(h), rgb(std::move(rgbBytes)) {}
    // PNG or JPEG file -> the bytes stb_image 2.27 gives the reference for three channels (reference
    // framework/src/image.cpp:13-35).  Throws std::exception when the file is missing or cannot be decoded.
    explicit Image(const std::filesystem::path& filePath);
    int width, height;
    std::vector<uint8_t> rgb; // row-major RGB bytes; texel value = byte / 255.0f
};

struct Vertex {
    vec3 position, normal;
    vec2 texCoord;
};
struct Material {
    vec3 kd;
    vec3 ks { 0.0f };
    float shininess { 1.0f };
    float transparency { 1.0f };
    std::shared_ptr<Image> kdTexture;
    std::string kdTextureFile; // no counterpart: map_Kd file name as loadMesh read it
};
struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<uvec3> triangles;
    Material material;
};
struct HitInfo {
    vec3 normal, barycentricCoord;
    vec2 texCoord;
    Material material;
};
struct Sphere {
    vec3 center { 0.0f };
    float radius = 1.0f;
    Material material;
};
struct PointLight { vec3 position, color; };
struct SegmentLight { vec3 endpoint0, endpoint1, color0, color1; };
struct ParallelogramLight { vec3 v0, edge01, edge02, color0, color1, color2, color3; };

struct ExtraFeatures {
    bool enableEnvironmentMapping = false;
    bool enableBvhSahBinning = false;
    bool enableMotionBlur = false;
    bool enableBloomEffect = false;
    bool enableBilinearTextureFiltering = false;
    bool enableMipmapTextureFiltering = false;
    bool enableMultipleRaysPerPixel = false;
    bool enableGlossyReflection = false;
    bool enableTransparency = false;
    bool enableDepthOfField = false;
};
struct Features {
    bool enableShading = false;
    bool enableRecursive = false;
    bool enableHardShadow = false;
    bool enableSoftShadow = false;
    bool enableNormalInterp = false;
    bool enableTextureMapping = false;
    bool enableAccelStructure = false;
    ExtraFeatures extra = {};
};

enum SceneType { SingleTriangle, Cube, CubeTextured, CornellBox, CornellBoxParallelogramLight, Monkey, Teapot, Dragon, Spheres, Custom, QuadTextured, TransparentCubes, TransparentCornell };

struct Scene {
    SceneType type = Custom;
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>> lights;
    std::vector<Image> skybox; // empty, or right, left, top, bottom, front, back
};

// Tunables the reference keeps as globals and its GUI edits between frames; renderRayTracing snapshots them.
extern int samplesPerUnit;
extern int samplesPerUnitParallel;
extern int rayMultiplier;
extern int extr_glossy_filterSize;
extern float extr_glossy_sigma;
extern int extr_dof_samples;
extern float extr_dof_aperture, extr_dof_f, extr_dof;
extern float threshold;
extern int bloomsize;
extern float sigma;
extern bool debugBloom;
extern int gaussian;
extern float scale;
extern bool extr_enabledSkyBox, extr_enabledReflMap;
extern int extr_max_level, extr_sah_bins;
extern unsigned renderSeed; // no counterpart: seed of the counter-based random stream
extern bool tiledGatherOnDevice; // no counterpart: renderRayTracingTiled gathers the shares GPU to GPU (RCCL), see there

class BvhInterface {
public:
    // Builds the hierarchy in the reference's node order and prepares the device copies.  The scene must
    // outlive the object and its geometry must not change afterwards; lights are read at render time.
    BvhInterface(Scene* pScene, const Features& features);
    ~BvhInterface();
    BvhInterface(const BvhInterface&) = delete;
    BvhInterface& operator=(const BvhInterface&) = delete;

    [[nodiscard]] int numLevels() const;
    [[nodiscard]] int numLeaves() const;
    void debugDrawLevel(int level); // OpenGL debug drawing in the reference; no-op here
    void debugDrawLeaf(int leafIdx);
    // Closest hit closer than ray.t; updates ray.t and hitInfo like the reference and returns whether anything was hit.
    bool intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const;

    [[nodiscard]] const rt_bvh* handle() const { return m_bvh; }
    [[nodiscard]] const Scene* scene() const { return m_pScene; }

private:
    Scene* m_pScene;
    rt_scene* m_scene = nullptr;
    rt_bvh* m_bvh = nullptr;
};

class Trackball {
public:
    // fovy in radians, rotations in radians; aspect = window width / height
    Trackball(float aspect, float fovy, const vec3& lookAt, float distFromLookAt = 4.0f, float rotationX = 0.0f, float rotationY = 0.0f);
    void setCamera(const vec3 lookAt, const vec3 rotations, const float dist);
    [[nodiscard]] vec3 left() const;
    [[nodiscard]] vec3 up() const;
    [[nodiscard]] vec3 forward() const;
    [[nodiscard]] vec3 position() const;
    [[nodiscard]] vec3 lookAt() const { return m_lookAt; }
    [[nodiscard]] Ray generateRay(const vec2& pixel) const;
    [[nodiscard]] rt_camera abi() const;

private:
    float m_aspect, m_fovy;
    vec3 m_lookAt;
    float m_distanceFromLookAt;
    vec3 m_rotationEulerAngles;
};

class Screen {
public:
    explicit Screen(const ivec2& resolution, bool presentable = false);
    void clear(const vec3& color);
    void setPixel(int x, int y, const vec3& color);
    void writeBitmapToFile(const std::filesystem::path& filePath);
    [[nodiscard]] ivec2 resolution() const { return m_resolution; }
    [[nodiscard]] int indexAt(int x, int y) const { return (m_resolution.y - 1 - y) * m_resolution.x + x; }
    [[nodiscard]] const std::vector<vec3>& pixels() const { return m_textureData; }
    [[nodiscard]] std::vector<vec3>& pixels() { return m_textureData; }

private:
    ivec2 m_resolution;
    std::vector<vec3> m_textureData;
};

// The whole frame, bloom included when features.extra.enableBloomEffect is set.  Throws nothing; a failure of
// the device path is reported on stderr and leaves the screen untouched (the reference has no error path).
void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features);

// renderRayTracing with the frame's rows dealt to several hierarchies, normally one per GPU: the reference hands rows
// to OpenMP threads (src/render.cpp:86-89); here the frame's blocks of 16 Screen rows are dealt round-robin to
// `shares.size()` shares, share i is rendered by a host thread of its own with one rt_render_blocks call on
// shares[i].bvh after rt_init(shares[i].device), the shares are put together in `screen` and bloom runs on the
// assembled frame.  Every hierarchy must have been built from `scene`; several shares may use the same hierarchy
// (they then take turns on it).  Random streams are keyed by pixel, so the frame does not depend on the shares.
// By default every share comes back through host memory.  With `tiledGatherOnDevice` set and every share on a device of
// its own, the shares are gathered GPU to GPU instead (rt_render_blocks_gather: grouped RCCL sends / receives into the
// first share's device, one device-to-host copy of the assembled frame); with a single share the same path runs on a
// one-rank communicator.  A hierarchy must have been built on its share's device (BvhInterface builds on the GPU that
// is current at construction).
struct RenderShare {
    const BvhInterface* bvh;
    int device;
};
void renderRayTracingTiled(const Scene& scene, const Trackball& camera, const std::vector<RenderShare>& shares, Screen& screen, const Features& features);

// Last rt_stats of renderRayTracing on this thread (no counterpart; the reference only prints wall time).
const rt_stats& lastRenderStats();

// Wavefront OBJ (+ MTL) file -> one Mesh per material run, as the reference's loadMesh does it (reference
// framework/src/mesh.cpp:52-149 on tinyobjloader 2.0.0): same vertices in the same order, same triangles, same
// materials; centerAndNormalize applies the reference's centring and scaling to the unit sphere (:151-176).
// A material's map_Kd file is decoded into kdTexture (:131-133; the file name is kept in kdTextureFile).  Throws
// std::exception like the reference when the OBJ file or a texture file is missing or malformed.
std::vector<Mesh> loadMesh(const std::filesystem::path& file, bool centerAndNormalize = false);

// The same decoder on bytes in memory (no counterpart).
Image decodeImage(const uint8_t* data, size_t size);

// The reference's table of prebuilt scenes (reference src/scene.cpp:5-119): OBJ file, unit-sphere normalisation
// and lights per SceneType, and the six sky box faces right/left/top/bottom/front/back.jpg.  The reference reads
// the sky box from its compile-time DATA_DIR; here it comes from dataDir as well, and withSkybox = false leaves it
// out (a scene without sky box renders black where the reference samples the cube map).
Scene loadScenePrebuilt(SceneType type, const std::filesystem::path& dataDir, bool withSkybox = true);
// One OBJ file with the caller's lights (reference src/scene.cpp:121-130).
Scene loadSceneFromFile(const std::filesystem::path& path, const std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>>& lights);

// Flat binary scene file written by tests (meshes, materials, textures, spheres, lights, skybox); an extension
// used where the reference would call loadScenePrebuilt with its OBJ files.
Scene loadSceneBinary(const std::filesystem::path& path);

} // namespace rtamd
