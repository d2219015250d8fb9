// readConfigFile and friends (reference src/config.cpp) on the TOML reader of toml_lite.h.
#include "rt_config.h"

#include "toml_lite.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

namespace rtamd {
namespace {

using toml::Value;

const char* scene_type_name(SceneType t)
{
    // the reference's printer knows the first ten scene types and prints nothing for the later ones (src/config.cpp:16-61)
    switch (t) {
    case SingleTriangle: return "SceneType::SingleTriangle";
    case Cube: return "SceneType::Cube";
    case CubeTextured: return "SceneType::CubeTextured";
    case CornellBox: return "SceneType::CornellBox";
    case CornellBoxParallelogramLight: return "SceneType::CornellBoxParallelogramLight";
    case Monkey: return "SceneType::Monkey";
    case Teapot: return "SceneType::Teapot";
    case Dragon: return "SceneType::Dragon";
    case Spheres: return "SceneType::Spheres";
    case Custom: return "SceneType::Custom";
    default: return "";
    }
}

std::ostream& operator<<(std::ostream& os, const vec3& v) { return os << "[" << v.x << ", " << v.y << ", " << v.z << "]"; }

// Up to three numbers of an array; whatever is missing (the whole array included) stays zero: the reference's helper
// always yields a value, so the fallbacks written at its call sites never apply (src/config.cpp:148-168).
vec3 to_vec3(const Value& v)
{
    float out[3] = { 0.0f, 0.0f, 0.0f };
    if (v.kind == Value::Array) {
        int i = 0;
        for (const Value& e : v.array) {
            if (e.is_number()) {
                if (i > 2) continue;
                out[i++] = float(e.number());
            } else {
                std::cerr << "Error: Expected a number in array" << std::endl;
            }
        }
    }
    return { out[0], out[1], out[2] };
}

ivec2 to_ivec2(const Value& v)
{
    int out[2] = { 0, 0 };
    if (v.kind == Value::Array) {
        int i = 0;
        for (const Value& e : v.array) {
            if (e.kind == Value::Int) {
                if (i > 1) continue;
                out[i++] = int(e.integer);
            } else {
                std::cerr << "Error: Expected an integer in array" << std::endl;
            }
        }
    }
    return { out[0], out[1] };
}

const Value& element(const Value& array, size_t i)
{
    static const Value none;
    return array.kind == Value::Array && i < array.array.size() ? array.array[i] : none;
}

bool required_bool(const Value& v, const char* name, bool fallback)
{
    if (v.kind == Value::Bool) return v.boolean;
    std::cerr << "WARN: " << name << " is missing or not a boolean; using " << (fallback ? "true" : "false") << std::endl;
    return fallback;
}

float required_float(const Value& v, const char* name, float fallback)
{
    if (v.is_number()) return float(v.number());
    std::cerr << "WARN: " << name << " is missing or not a number; using " << fallback << std::endl;
    return fallback;
}

template <class T>
void optional_number(const Value& v, std::optional<T>& out)
{
    if (v.is_number()) out = T(v.number());
}

} // namespace

std::filesystem::path defaultDataDir()
{
    const char* env = std::getenv("RT_DATA_DIR");
    return env && *env ? std::filesystem::path(env) : std::filesystem::path("data");
}

void Tunables::apply() const
{
    if (samplesPerUnit) rtamd::samplesPerUnit = *samplesPerUnit;
    if (samplesPerUnitParallel) rtamd::samplesPerUnitParallel = *samplesPerUnitParallel;
    if (rayMultiplier) rtamd::rayMultiplier = *rayMultiplier;
    if (glossyFilterSize) extr_glossy_filterSize = *glossyFilterSize;
    if (glossySigma) extr_glossy_sigma = *glossySigma;
    if (dofSamples) extr_dof_samples = *dofSamples;
    if (dofAperture) extr_dof_aperture = *dofAperture;
    if (dofFocalLength) extr_dof_f = *dofFocalLength;
    if (bloomThreshold) threshold = *bloomThreshold;
    if (bloomSize) bloomsize = *bloomSize;
    if (bloomSigma) sigma = *bloomSigma;
    if (bloomScale) scale = *bloomScale;
    if (skyBox) extr_enabledSkyBox = *skyBox;
    if (reflectionMap) extr_enabledReflMap = *reflectionMap;
    if (maxLevel) extr_max_level = *maxLevel;
    if (sahBins) extr_sah_bins = *sahBins;
    if (seed) renderSeed = *seed;
}

std::ostream& operator<<(std::ostream& os, const Config& config)
{
    os << "Final Project Configurations: " << std::endl
       << std::boolalpha << "  + command_line_rendering: " << config.cliRenderingEnabled << std::endl
       << "  + window_size: " << config.windowSize.x << ", " << config.windowSize.y << std::endl
       << "  + data_path: " << config.dataPath << std::endl
       << "  + scene: ";
    if (std::holds_alternative<SceneType>(config.scene)) os << scene_type_name(std::get<SceneType>(config.scene)) << std::endl;
    else os << std::get<std::filesystem::path>(config.scene) << std::endl;
    const Features& f = config.features;
    os << "  + output_filepath: " << config.outputDir << std::endl
       << "  + features: " << std::endl
       << "    - enable_shading: " << f.enableShading << std::endl
       << "    - enable_recursive: " << f.enableRecursive << std::endl
       << "    - enable_hard_shadow: " << f.enableHardShadow << std::endl
       << "    - enable_soft_shadow: " << f.enableSoftShadow << std::endl
       << "    - enable_normal_interp: " << f.enableNormalInterp << std::endl
       << "    - enable_texture_mapping: " << f.enableTextureMapping << std::endl
       << "    - enable_accel_structure: " << f.enableAccelStructure << std::endl
       << "  + extra_features: " << std::endl
       << "    - enable_bloom_effect: " << f.extra.enableBloomEffect << std::endl
       << "    - enable_multiple_rays_per_pixel: " << f.extra.enableMultipleRaysPerPixel << std::endl
       << "    - enable_motion_blur: " << f.extra.enableMotionBlur << std::endl
       << "    - enable_depth_of_field: " << f.extra.enableDepthOfField << std::endl
       << "    - enable_glossy_reflection: " << f.extra.enableGlossyReflection << std::endl
       << "    - enable_transparency: " << f.extra.enableTransparency << std::endl
       << "    - enable_bvh_sah_binning: " << f.extra.enableBvhSahBinning << std::endl
       << "    - enable_environment_mapping: " << f.extra.enableEnvironmentMapping << std::endl
       << "    - enable_bilinear_texture_filtering: " << f.extra.enableBilinearTextureFiltering << std::endl
       << "    - enable_mipmap_texture_filtering: " << f.extra.enableMipmapTextureFiltering << std::endl;
    os << "  + cameras: " << std::endl;
    for (const CameraConfig& camera : config.cameras) {
        os << "    - field_of_view: " << camera.fieldOfView << std::endl
           << "      distance_from_look_at: " << camera.distanceFromLookAt << std::endl
           << "      look_at: " << camera.lookAt << std::endl
           << "      rotation: " << camera.rotation << std::endl;
    }
    os << "  + lights: " << std::endl;
    for (const auto& light : config.lights) {
        if (const PointLight* p = std::get_if<PointLight>(&light)) {
            os << "    - type: point" << std::endl << "      position: " << p->position << ", color: " << p->color << std::endl;
        } else if (const SegmentLight* s = std::get_if<SegmentLight>(&light)) {
            os << "    - type: segment" << std::endl
               << "      endpoint0: " << s->endpoint0 << ", endpoint1: " << s->endpoint1 << std::endl
               << "      color0: " << s->color0 << ", color1: " << s->color1 << std::endl;
        } else if (const ParallelogramLight* g = std::get_if<ParallelogramLight>(&light)) {
            os << "    - type: parallelogram" << std::endl
               << "      v0: " << g->v0 << std::endl
               << "      edge01: " << g->edge01 << ", edge02: " << g->edge02 << std::endl
               << "      color0: " << g->color0 << ", color1: " << g->color1 << std::endl
               << "      color2: " << g->color2 << ", color3: " << g->color3 << std::endl;
        }
    }
    return os;
}

Config readConfigFile(const std::filesystem::path& config_path)
{
    Config config = {};
    std::ifstream file(config_path, std::ios::binary);
    if (!file) throw std::runtime_error("Failed reading " + config_path.string());
    std::stringstream text;
    text << file.rdbuf();
    Value table;
    try {
        table = toml::parse(text.str());
    } catch (const std::exception& e) {
        throw std::runtime_error("Failed parsing " + config_path.string() + ":\n" + e.what());
    }

    config.cliRenderingEnabled = required_bool(table["command_line_rendering"], "command_line_rendering", true);
    config.windowSize = to_ivec2(table["window_size"]);

    std::string dataPath = table["data_path"].kind == Value::String ? table["data_path"].string : std::string("default");
    config.dataPath = dataPath == "default" ? defaultDataDir() : std::filesystem::path(dataPath);

    const Value& scene = table["scene"];
    if (scene.is_number()) {
        config.scene = SceneType(int(scene.number()));
    } else {
        const std::string name = scene.kind == Value::String ? scene.string : std::string("none");
        if (const std::optional<SceneType> type = deserialize(name)) {
            config.scene = *type;
        } else {
            const std::filesystem::path path = config.dataPath / name;
            if (!std::filesystem::exists(path)) throw std::runtime_error("Error: Scene file " + path.string() + " does not exist.");
            config.scene = path;
        }
    }

    std::string outputDir = table["output_dir"].kind == Value::String ? table["output_dir"].string : std::string();
    if (outputDir.empty()) {
        std::cout << "Warning: No output directory specified, using current directory." << std::endl;
        config.outputDir = std::filesystem::current_path();
    } else {
        const char* home = std::getenv("HOME");
        if (home && outputDir[0] == '~') outputDir.replace(0, 1, home);
        if (home && outputDir.rfind("$HOME", 0) == 0) outputDir.replace(0, 5, home);
        config.outputDir = std::filesystem::absolute(std::filesystem::path(outputDir));
    }

    const Value& features = table["features"];
    config.features.enableShading = required_bool(features["enable_shading"], "features.enable_shading", false);
    config.features.enableRecursive = required_bool(features["enable_recursive"], "features.enable_recursive", false);
    config.features.enableHardShadow = required_bool(features["enable_hard_shadow"], "features.enable_hard_shadow", false);
    config.features.enableNormalInterp = required_bool(features["enable_normal_interp"], "features.enable_normal_interp", false);
    config.features.enableTextureMapping = required_bool(features["enable_texture_mapping"], "features.enable_texture_mapping", false);
    config.features.enableAccelStructure = required_bool(features["enable_accel_structure"], "features.enable_accel_structure", false);
    if (features["enable_soft_shadow"].kind == Value::Bool) config.features.enableSoftShadow = features["enable_soft_shadow"].boolean;

    const Value& extra = features["extra"];
    auto optional_switch = [&extra](const char* key, bool& out) {
        if (extra[key].kind == Value::Bool) out = extra[key].boolean;
    };
    optional_switch("enable_bloom_effect", config.features.extra.enableBloomEffect);
    optional_switch("enable_multiple_rays_per_pixel", config.features.extra.enableMultipleRaysPerPixel);
    optional_switch("enable_motion_blur", config.features.extra.enableMotionBlur);
    optional_switch("enable_depth_of_field", config.features.extra.enableDepthOfField);
    optional_switch("enable_glossy_reflection", config.features.extra.enableGlossyReflection);
    optional_switch("enable_environment_mapping", config.features.extra.enableEnvironmentMapping);
    optional_switch("enable_bilinear_texture_filtering", config.features.extra.enableBilinearTextureFiltering);
    optional_switch("enable_mipmap_texture_filtering", config.features.extra.enableMipmapTextureFiltering);
    optional_switch("enable_transparency", config.features.extra.enableTransparency);
    optional_switch("enable_bvh_sah_binning", config.features.extra.enableBvhSahBinning);

    const Value& cameras = table["cameras"];
    if (cameras.kind == Value::Array) {
        for (const Value& camera : cameras.array) {
            CameraConfig c;
            c.fieldOfView = required_float(camera["field_of_view"], "cameras.field_of_view", 50.0f);
            c.distanceFromLookAt = required_float(camera["distance_from_look_at"], "cameras.distance_from_look_at", 3.0f);
            c.lookAt = to_vec3(camera["look_at"]);
            c.rotation = to_vec3(camera["rotation"]);
            config.cameras.push_back(c);
        }
    }

    const Value& lights = table["lights"];
    if (lights.kind == Value::Array) {
        for (const Value& light : lights.array) {
            const std::string type = light["type"].kind == Value::String ? light["type"].string : std::string("none");
            if (type == "point") {
                config.lights.emplace_back(PointLight { to_vec3(light["position"]), to_vec3(light["color"]) });
            } else if (type == "segment") {
                config.lights.emplace_back(SegmentLight { to_vec3(element(light["endpoints"], 0)), to_vec3(element(light["endpoints"], 1)), to_vec3(element(light["colors"], 0)),
                    to_vec3(element(light["colors"], 1)) });
            } else if (type == "parallelogram") {
                config.lights.emplace_back(ParallelogramLight { to_vec3(light["corner"]), to_vec3(element(light["edges"], 0)), to_vec3(element(light["edges"], 1)),
                    to_vec3(element(light["colors"], 0)), to_vec3(element(light["colors"], 1)), to_vec3(element(light["colors"], 2)), to_vec3(element(light["colors"], 3)) });
            } else {
                std::cerr << "Unknown light type: " << type << " -- Skip" << std::endl;
            }
        }
    } else {
        std::cerr << "WARN: No lights found in config file.\n";
    }

    const Value& tun = table["tunables"];
    Tunables& t = config.tunables;
    optional_number(tun["samples_per_unit"], t.samplesPerUnit);
    optional_number(tun["samples_per_unit_parallel"], t.samplesPerUnitParallel);
    optional_number(tun["ray_multiplier"], t.rayMultiplier);
    optional_number(tun["glossy_filter_size"], t.glossyFilterSize);
    optional_number(tun["glossy_sigma"], t.glossySigma);
    optional_number(tun["dof_samples"], t.dofSamples);
    optional_number(tun["dof_aperture"], t.dofAperture);
    optional_number(tun["dof_focal_length"], t.dofFocalLength);
    optional_number(tun["bloom_threshold"], t.bloomThreshold);
    optional_number(tun["bloom_size"], t.bloomSize);
    optional_number(tun["bloom_sigma"], t.bloomSigma);
    optional_number(tun["bloom_scale"], t.bloomScale);
    optional_number(tun["max_level"], t.maxLevel);
    optional_number(tun["sah_bins"], t.sahBins);
    optional_number(tun["seed"], t.seed);
    if (tun["sky_box"].kind == Value::Bool) t.skyBox = tun["sky_box"].boolean;
    if (tun["reflection_map"].kind == Value::Bool) t.reflectionMap = tun["reflection_map"].boolean;
    return config;
}

std::string serialize(const SceneType& sceneType)
{
    switch (sceneType) {
    case SingleTriangle: return "single_triangle";
    case Cube: return "cube";
    case CubeTextured: return "cube_textured";
    case CornellBox: return "cornell_box";
    case CornellBoxParallelogramLight: return "cornell_box_parallelogram_light";
    case Monkey: return "monkey";
    case Teapot: return "teapot";
    case Dragon: return "dragon";
    case Spheres: return "spheres";
    case Custom: return "custom";
    default: return "unknown";
    }
}

std::optional<SceneType> deserialize(const std::string& sceneTypeStr)
{
    std::string s;
    std::transform(sceneTypeStr.begin(), sceneTypeStr.end(), std::back_inserter(s), [](unsigned char c) { return char(std::tolower(c)); });
    // each name with '_', with '-' and without a separator, as far as the reference accepts them (src/config.cpp:404-431)
    struct Name {
        const char* spelling;
        SceneType type;
    };
    static const Name names[] = {
        { "single_triangle", SingleTriangle }, { "singletriangle", SingleTriangle }, { "single-triangle", SingleTriangle }, { "cube", Cube },
        { "cube-textured", CubeTextured }, { "cube_textured", CubeTextured }, { "cubetextured", CubeTextured }, { "cornell_box", CornellBox },
        { "cornellbox", CornellBox }, { "cornell-box", CornellBox }, { "cornell_box_parallelogram_light", CornellBoxParallelogramLight },
        { "cornellboxparallelogramlight", CornellBoxParallelogramLight }, { "cornell-box-parallelogram-light", CornellBoxParallelogramLight }, { "monkey", Monkey },
        { "teapot", Teapot }, { "dragon", Dragon }, { "spheres", Spheres }, { "custom", Custom },
    };
    for (const Name& n : names)
        if (s == n.spelling) return n.type;
    return std::nullopt;
}

} // namespace rtamd
