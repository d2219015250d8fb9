// Small driver over the C++ facade, used by the tests: loads a binary scene, renders one frame through
// BvhInterface + renderRayTracing exactly the way the reference's CLI mode does (src/main.cpp:523-545), and
// writes the float frame (and optionally a BMP).
//   facade_render <scene.rtscene> <out.f32> <width> <height> <feature flags, comma separated> [out.bmp]
// Second form, no rendering: reads a Wavefront OBJ file with loadMesh and writes the meshes as a binary scene.
//   facade_render --obj <file.obj> <0|1: centre and normalise> <out.rtscene>
// Third and fourth form: an image file through Image(path); a prebuilt scene through loadScenePrebuilt.
//   facade_render --image <file.png|jpg> <out.bin>
//   facade_render --table <scene type number> <data dir> <0|1: sky box> <out.rtscene>
#include "rt_facade.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <variant>

using namespace rtamd;

static int write_meshes(const std::vector<Mesh>& meshes, const char* outFile);

static int convert_obj(const char* objFile, bool normalize, const char* outFile)
{
    std::vector<Mesh> meshes;
    try {
        meshes = loadMesh(objFile, normalize);
    } catch (const std::exception&) {
        return 1;
    }
    return write_meshes(meshes, outFile);
}

// Image(path): int32 width, int32 height, then the RGB bytes.  Exit status 3 when the file cannot be decoded.
static int convert_image(const char* imageFile, const char* outFile)
{
    try {
        const Image img { std::filesystem::path(imageFile) };
        std::ofstream out(outFile, std::ios::binary);
        const int32_t size[2] = { img.width, img.height };
        out.write(reinterpret_cast<const char*>(size), sizeof size);
        out.write(reinterpret_cast<const char*>(img.rgb.data()), std::streamsize(img.rgb.size()));
        return out ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
}

// loadScenePrebuilt: the meshes as a binary scene, lights and spheres as text lines (the 22 / 12 numbers per record
// that the golden scene arrays hold), texture sizes.
static int convert_table(int type, const char* dataDir, bool withSkybox, const char* outFile)
{
    Scene scene;
    try {
        scene = loadScenePrebuilt(SceneType(type), dataDir, withSkybox);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    for (const auto& l : scene.lights) {
        float rec[22] = { 0 };
        auto put = [&](int at, const vec3& v) { rec[at] = v.x; rec[at + 1] = v.y; rec[at + 2] = v.z; };
        if (std::holds_alternative<PointLight>(l)) {
            const auto& p = std::get<PointLight>(l);
            put(1, p.position); put(4, p.color);
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& g = std::get<SegmentLight>(l);
            rec[0] = 1; put(1, g.endpoint0); put(4, g.endpoint1); put(7, g.color0); put(10, g.color1);
        } else {
            const auto& p = std::get<ParallelogramLight>(l);
            rec[0] = 2; put(1, p.v0); put(4, p.edge01); put(7, p.edge02); put(10, p.color0); put(13, p.color1); put(16, p.color2); put(19, p.color3);
        }
        std::printf("light");
        for (float v : rec) std::printf(" %.9g", double(v));
        std::printf("\n");
    }
    for (const Sphere& sp : scene.spheres)
        std::printf("sphere %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", double(sp.center.x), double(sp.center.y), double(sp.center.z), double(sp.radius),
            double(sp.material.kd.x), double(sp.material.kd.y), double(sp.material.kd.z), double(sp.material.ks.x), double(sp.material.ks.y), double(sp.material.ks.z),
            double(sp.material.shininess), double(sp.material.transparency));
    for (const Mesh& m : scene.meshes)
        if (m.material.kdTexture) std::printf("texture %d %d\n", m.material.kdTexture->width, m.material.kdTexture->height);
    for (const Image& face : scene.skybox) std::printf("skybox %d %d\n", face.width, face.height);
    return write_meshes(scene.meshes, outFile);
}

static int write_meshes(const std::vector<Mesh>& meshes, const char* outFile)
{
    std::ofstream out(outFile, std::ios::binary);
    out.write("RTSCENE1", 8);
    const uint32_t counts[8] = { uint32_t(meshes.size()), 0, 0, 0, 0, 0, 0, 0 };
    out.write(reinterpret_cast<const char*>(counts), sizeof counts);
    for (const Mesh& m : meshes) {
        const uint32_t n[2] = { uint32_t(m.vertices.size()), uint32_t(m.triangles.size()) };
        out.write(reinterpret_cast<const char*>(n), sizeof n);
        const float mat[8] = { m.material.kd.x, m.material.kd.y, m.material.kd.z, m.material.ks.x, m.material.ks.y, m.material.ks.z, m.material.shininess, m.material.transparency };
        out.write(reinterpret_cast<const char*>(mat), sizeof mat);
        const int32_t texture = m.material.kdTextureFile.empty() ? -1 : 0;
        out.write(reinterpret_cast<const char*>(&texture), sizeof texture);
        out.write(reinterpret_cast<const char*>(m.vertices.data()), std::streamsize(m.vertices.size() * sizeof(Vertex)));
        out.write(reinterpret_cast<const char*>(m.triangles.data()), std::streamsize(m.triangles.size() * sizeof(uvec3)));
    }
    std::printf("meshes %zu\n", meshes.size());
    return out ? 0 : 1;
}

int main(int argc, char** argv)
{
    if (argc == 5 && std::strcmp(argv[1], "--obj") == 0) return convert_obj(argv[2], std::atoi(argv[3]) != 0, argv[4]);
    if (argc == 4 && std::strcmp(argv[1], "--image") == 0) return convert_image(argv[2], argv[3]);
    if (argc == 6 && std::strcmp(argv[1], "--table") == 0) return convert_table(std::atoi(argv[2]), argv[3], std::atoi(argv[4]) != 0, argv[5]);
    if (argc < 6) {
        std::fprintf(stderr, "usage: facade_render scene out.f32 W H flags [out.bmp]\n");
        return 2;
    }
    Scene scene = loadSceneBinary(argv[1]);
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
    Features f;
    f.enableAccelStructure = true;
    std::string flags = argv[5];
    auto has = [&](const char* name) { return ("," + flags + ",").find(std::string(",") + name + ",") != std::string::npos; };
    f.enableShading = has("shading");
    f.enableRecursive = has("recursive");
    f.enableHardShadow = has("hard");
    f.enableSoftShadow = has("soft");
    f.enableNormalInterp = has("normals");
    f.enableTextureMapping = has("textures");
    f.extra.enableBloomEffect = has("bloom");
    f.extra.enableTransparency = has("transparency");
    f.extra.enableMultipleRaysPerPixel = has("multi2");
    if (has("multi2")) rayMultiplier = 2;
    BvhInterface bvh(&scene, f);
    std::printf("levels %d leaves %d\n", bvh.numLevels(), bvh.numLeaves());
    Screen screen({ W, H });
    // the reference's default camera, src/config.h:16-21 (degrees converted like glm::radians)
    const float rad = 0.01745329251994329576923690768489f;
    Trackball camera(float(W) / float(H), 50.0f * rad, vec3(0.0f), 3.0f);
    camera.setCamera(vec3(0.0f), vec3(20.0f * rad, 20.0f * rad, 0.0f), 3.0f);
    renderRayTracing(scene, camera, bvh, screen, f);
    // one debug ray through the pixel in the middle, like the GUI's ray inspector (src/main.cpp:155-158)
    Ray ray = camera.generateRay({ 0.0f, 0.0f });
    HitInfo hit;
    const bool ok = bvh.intersect(ray, hit, f);
    std::printf("centre ray hit %d t %.9g\n", int(ok), double(ray.t));
    std::ofstream out(argv[2], std::ios::binary);
    out.write(reinterpret_cast<const char*>(screen.pixels().data()), std::streamsize(screen.pixels().size() * sizeof(vec3)));
    if (argc > 6) screen.writeBitmapToFile(argv[6]);
    const rt_stats& st = lastRenderStats();
    std::printf("rays %llu kernel_ms %.3f\n", (unsigned long long)(st.primary_rays + st.shadow_rays + st.bounce_rays), double(st.kernel_ms));
    return 0;
}
