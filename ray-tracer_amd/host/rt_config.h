// Configuration file of the command-line renderer: the reference's Config / CameraConfig / readConfigFile /
// serialize / deserialize (reference src/config.h:16-40, src/config.cpp:71-431) over the facade's types.
#pragma once
#include "rt_facade.h"

#include <iosfwd>
#include <optional>

namespace rtamd {

struct CameraConfig {
    float fieldOfView = 50.0f; // degrees
    float distanceFromLookAt = 3.0f;
    vec3 lookAt { 0.0f, 0.0f, 0.0f };
    vec3 rotation { 20.0f, 20.0f, 0.0f }; // degrees
};

// Values for the tunable globals of rt_facade.h that a configuration file may set ([tunables] table; no counterpart:
// the reference changes them only through its GUI).  An unset entry leaves the global alone.
struct Tunables {
    std::optional<int> samplesPerUnit, samplesPerUnitParallel, rayMultiplier, glossyFilterSize, dofSamples, bloomSize, maxLevel, sahBins;
    std::optional<float> glossySigma, dofAperture, dofFocalLength, bloomThreshold, bloomSigma, bloomScale;
    std::optional<bool> skyBox, reflectionMap;
    std::optional<unsigned> seed;
    void apply() const;
};

struct Config {
    Features features = {};
    bool cliRenderingEnabled = false;
    ivec2 windowSize { 800, 800 };
    std::filesystem::path dataPath;
    std::variant<SceneType, std::filesystem::path> scene = SceneType::SingleTriangle;
    std::filesystem::path outputDir = "";
    std::vector<CameraConfig> cameras;
    std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>> lights;
    Tunables tunables; // no counterpart
};

// The reference's compile-time DATA_DIR: here the environment variable RT_DATA_DIR, or "data".
std::filesystem::path defaultDataDir();

std::ostream& operator<<(std::ostream& os, const Config& config);

// Reads a TOML configuration with the reference's keys and defaults.  Keys the reference dereferences without a
// check (command_line_rendering, the six [features] switches, field_of_view / distance_from_look_at of a camera)
// crash it when absent; here they fall back to the default the reference names in its value_or() and a warning is
// printed.  Like there, a missing window_size reads as (0, 0) and a missing look_at / rotation of a listed camera as
// (0, 0, 0).  Additional keys: features.enable_soft_shadow, features.extra.enable_transparency,
// features.extra.enable_bvh_sah_binning and the [tunables] table.  Throws std::runtime_error on a file that
// cannot be read or parsed, or when the scene names neither a prebuilt scene nor a file under data_path.
Config readConfigFile(const std::filesystem::path& config_path);

std::string serialize(const SceneType& sceneType);
std::optional<SceneType> deserialize(const std::string& sceneTypeStr);

} // namespace rtamd
