// The reference's prebuilt scenes and its one-file scene loader (reference src/scene.cpp:5-130): which OBJ file,
// whether it is normalised to the unit sphere, and the lights, per SceneType.
#include <stdexcept>
#include "rt_facade.h"

namespace rtamd {
namespace {

struct Entry {
    SceneType type;
    const char* file;   // under the data directory; nullptr = no mesh
    bool normalise;
};

constexpr Entry kTable[] = {
    { SingleTriangle, "triangle.obj", false },
    { Cube, "cube.obj", false },
    { CubeTextured, "cube-textured.obj", false },
    { CornellBox, "CornellBox-Mirror-Rotated.obj", true },
    { CornellBoxParallelogramLight, "CornellBox-Mirror-Rotated.obj", true },
    { Monkey, "monkey.obj", true },
    { Teapot, "teapot.obj", true },
    { Dragon, "dragon.obj", true },
    { Spheres, nullptr, false },
    { Custom, "custom.obj", false },
    { QuadTextured, "quadtextured.obj", true },
    { TransparentCubes, "transparent-cubes.obj", true },
    { TransparentCornell, "cornelltransparent.obj", true },
};

Sphere sphere(vec3 center, float radius, vec3 kd)
{
    Sphere s;
    s.center = center;
    s.radius = radius;
    s.material.kd = kd;
    return s;
}

void add_lights(Scene& scene)
{
    auto point = [&scene](vec3 position, vec3 color) { scene.lights.emplace_back(PointLight { position, color }); };
    switch (scene.type) {
    case SingleTriangle:
        if (scene.meshes.empty()) throw std::runtime_error("triangle.obj holds no mesh");
        scene.meshes[0].material.kd = vec3(1.0f);
        point({ -1, 1, -1 }, vec3(1));
        break;
    case Cube:
        scene.lights.emplace_back(SegmentLight { { 1.5f, 0.5f, -0.6f }, { -1, 0.5f, -0.5f }, { 0.9f, 0.2f, 0.1f }, { 0.2f, 1, 0.3f } });
        break;
    case CubeTextured: point({ -1.0f, 1.5f, -1.0f }, vec3(1)); break;
    case CornellBox: point({ 0, 0.58f, 0 }, vec3(1)); break;
    case CornellBoxParallelogramLight:
        scene.lights.emplace_back(ParallelogramLight { { -0.2f, 0.5f, 0 }, { 0.4f, 0, 0 }, { 0.0f, 0.0f, 0.4f }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 0, 1, 1 } });
        break;
    case Monkey:
        point({ -1, 1, -1 }, vec3(1));
        point({ 1, -1, -1 }, vec3(1));
        break;
    case Teapot:
    case Dragon:
    case Custom: point({ -1, 1, -1 }, vec3(1)); break;
    case Spheres:
        scene.spheres.push_back(sphere({ 3.0f, -2.0f, 10.2f }, 1.0f, { 0.8f, 0.2f, 0.2f }));
        scene.spheres.push_back(sphere({ -2.0f, 2.0f, 4.0f }, 2.0f, { 0.6f, 0.8f, 0.2f }));
        scene.spheres.push_back(sphere({ 0.0f, 0.0f, 6.0f }, 0.75f, { 0.2f, 0.2f, 0.8f }));
        point({ 3, 0, 3 }, vec3(15));
        break;
    case QuadTextured: point({ -0.3f, 0.58f, 0 }, vec3(1)); break;
    case TransparentCubes: point({ -0.8f, 0.58f, 0 }, vec3(1)); break;
    case TransparentCornell: point({ 0, 0.58f, 0 }, vec3(1)); break;
    }
}

} // namespace

Scene loadScenePrebuilt(SceneType type, const std::filesystem::path& dataDir, bool withSkybox)
{
    Scene scene;
    scene.type = type;
    for (const Entry& e : kTable)
        if (e.type == type && e.file) scene.meshes = loadMesh(dataDir / e.file, e.normalise);
    add_lights(scene);
    if (withSkybox)
        for (const char* face : { "right.jpg", "left.jpg", "top.jpg", "bottom.jpg", "front.jpg", "back.jpg" }) scene.skybox.emplace_back(dataDir / face);
    return scene;
}

Scene loadSceneFromFile(const std::filesystem::path& path, const std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>>& lights)
{
    Scene scene;
    scene.lights = lights;
    scene.meshes = loadMesh(path);
    return scene;
}

} // namespace rtamd
