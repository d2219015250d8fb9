// Command-line renderer: the reference's "command_line_rendering" mode (reference src/main.cpp:499-556).
//   rt_cli <config.toml> [--gpus N] [--tiles T] [--gather-on-device] [--print-config]
// Reads the configuration, prints it, loads the scene (a prebuilt one, or an OBJ file with the configured lights),
// builds the hierarchy and renders one bitmap per configured camera into the output directory, named
// <scene>_<start time>_cam_<index>.bmp like the reference's.  The reference gives every camera a CPU thread; here
// the cameras are dealt round-robin to N GPUs (default 1), one host thread and one copy of the hierarchy per GPU.
// With --tiles T (> 1) the cameras are rendered one after the other instead, each frame's rows dealt in blocks of 16 to T
// shares spread over the N GPUs (renderRayTracingTiled: the reference's row parallelism across GPUs; T may exceed N).
// --gather-on-device (with T = N, a share per GPU): the shares are gathered GPU to GPU by RCCL instead of through host
// memory.
// --print-config stops after printing (used by the tests).  The interactive window of the reference is not part
// of this program: a configuration with command_line_rendering = false is refused.
#include "rt_config.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>

using namespace rtamd;

int main(int argc, char** argv)
{
    const char* configFile = nullptr;
    int gpus = 1, tiles = 1;
    bool printOnly = false, gatherOnDevice = false;
    for (int i = 1; i < argc; i++) {
        if (std::strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (std::strcmp(argv[i], "--tiles") == 0 && i + 1 < argc) tiles = std::atoi(argv[++i]);
        else if (std::strcmp(argv[i], "--print-config") == 0) printOnly = true;
        else if (std::strcmp(argv[i], "--gather-on-device") == 0) gatherOnDevice = true;
        else if (!configFile) configFile = argv[i];
        else configFile = "";
    }
    if (!configFile || !*configFile || gpus < 1 || tiles < 1) {
        std::fprintf(stderr, "usage: rt_cli <config.toml> [--gpus N] [--tiles T] [--gather-on-device] [--print-config]\n");
        return 2;
    }
    Config config;
    try {
        config = readConfigFile(configFile);
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    std::cout << config;
    if (printOnly) return 0;
    if (!config.cliRenderingEnabled) {
        std::cerr << "command_line_rendering = false asks for the interactive window, which this program does not have" << std::endl;
        return 2;
    }
    if (config.windowSize.x <= 0 || config.windowSize.y <= 0) {
        std::cerr << "window_size must be two positive integers" << std::endl;
        return 1;
    }
    config.tunables.apply();

    Scene scene;
    std::string sceneName;
    try {
        if (const std::filesystem::path* path = std::get_if<std::filesystem::path>(&config.scene)) {
            scene = loadSceneFromFile(*path, config.lights);
            sceneName = path->stem().string();
        } else {
            const SceneType type = std::get<SceneType>(config.scene);
            // the sky box is read only when something can show it (the reference always reads it)
            scene = loadScenePrebuilt(type, config.dataPath, config.features.extra.enableEnvironmentMapping);
            sceneName = serialize(type);
        }
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }

    if (!std::filesystem::exists(config.outputDir)) std::filesystem::create_directories(config.outputDir);
    using clock = std::chrono::high_resolution_clock;
    const auto start = clock::now();
    char startTime[64];
    const std::time_t now = std::time(nullptr);
    std::strftime(startTime, sizeof startTime, "%Y-%m-%d-%H:%M:%S", std::localtime(&now));

    const float rad = 0.01745329251994329576923690768489f; // glm::radians
    std::mutex printLock;
    bool failed = false;
    auto worker = [&](int gpu) {
        if (rt_init(gpu) != RT_OK) {
            std::lock_guard<std::mutex> g(printLock);
            std::cerr << "GPU " << gpu << ": " << rt_last_error() << std::endl;
            failed = true;
            return;
        }
        BvhInterface bvh { &scene, config.features };
        for (size_t index = size_t(gpu); index < config.cameras.size(); index += size_t(gpus)) {
            const CameraConfig& c = config.cameras[index];
            Screen screen { config.windowSize, false };
            screen.clear(vec3(0.0f));
            Trackball camera { float(config.windowSize.x) / float(config.windowSize.y), c.fieldOfView * rad, vec3(0.0f), c.distanceFromLookAt };
            camera.setCamera(c.lookAt, vec3(c.rotation.x * rad, c.rotation.y * rad, c.rotation.z * rad), c.distanceFromLookAt);
            renderRayTracing(scene, camera, bvh, screen, config.features);
            const std::filesystem::path file = config.outputDir / (sceneName + "_" + startTime + "_cam_" + std::to_string(index) + ".bmp");
            {
                std::lock_guard<std::mutex> g(printLock);
                std::cout << "Image " << index << " saved to " << file.string() << std::endl;
            }
            screen.writeBitmapToFile(file);
        }
    };
    if (tiles > 1 || gatherOnDevice) {
        tiledGatherOnDevice = gatherOnDevice;
        // one hierarchy per GPU (built by a thread that has selected it), every camera's frame dealt to `tiles` shares
        std::vector<std::unique_ptr<BvhInterface>> bvhs;
        bvhs.resize(size_t(gpus));
        std::vector<std::thread> builders;
        for (int g = 0; g < gpus; g++)
            builders.emplace_back([&, g] {
                if (rt_init(g) != RT_OK) {
                    std::lock_guard<std::mutex> lock(printLock);
                    std::cerr << "GPU " << g << ": " << rt_last_error() << std::endl;
                    failed = true;
                    return;
                }
                bvhs[size_t(g)] = std::make_unique<BvhInterface>(&scene, config.features);
            });
        for (std::thread& b : builders) b.join();
        if (failed) return 1;
        std::vector<RenderShare> shares;
        for (int t = 0; t < tiles; t++) shares.push_back(RenderShare { bvhs[size_t(t % gpus)].get(), t % gpus });
        for (size_t index = 0; index < config.cameras.size(); index++) {
            const CameraConfig& c = config.cameras[index];
            Screen screen { config.windowSize, false };
            screen.clear(vec3(0.0f));
            Trackball camera { float(config.windowSize.x) / float(config.windowSize.y), c.fieldOfView * rad, vec3(0.0f), c.distanceFromLookAt };
            camera.setCamera(c.lookAt, vec3(c.rotation.x * rad, c.rotation.y * rad, c.rotation.z * rad), c.distanceFromLookAt);
            renderRayTracingTiled(scene, camera, shares, screen, config.features);
            const std::filesystem::path file = config.outputDir / (sceneName + "_" + startTime + "_cam_" + std::to_string(index) + ".bmp");
            std::cout << "Image " << index << " saved to " << file.string() << std::endl;
            screen.writeBitmapToFile(file);
        }
    } else {
        std::vector<std::thread> workers;
        for (int g = 0; g < gpus; g++) workers.emplace_back(worker, g);
        for (std::thread& w : workers) w.join();
    }
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - start).count();
    std::cout << "Rendering took " << ms << " ms, " << config.cameras.size() << " images rendered." << std::endl;
    return failed ? 1 : 0;
}
