// Wavefront OBJ / MTL ingestion for the C++ facade: loadMesh() with the behaviour of the reference's loader
// (reference framework/src/mesh.cpp:52-176), which sits on tinyobjloader 2.0.0
// (reference framework/third_party/tinyobjloader/include/tinyobjloader/tiny_obj_loader.h).  The reference's
// scenes are what this has to reproduce bit for bit — same vertices in the same order, same triangles, same
// materials — so the parts of tinyobjloader's behaviour that decide those are restated here:
//   * decimal numbers are converted by its own routine (tiny_obj_loader.h:866-996), not strtod: digits are
//     accumulated in a double, fraction digits through a power table, a decimal exponent through
//     ldexp(m * 5^e, e); the result is narrowed to float;
//   * indices: 1-based, negative = relative to the elements read so far (:800-821, :1136-1187);
//   * a quad is cut along its shorter diagonal, [0,1,2][0,2,3] if |v2-v0|^2 < |v3-v1|^2 else [0,1,3][1,2,3]
//     (:1429-1532); faces with more than four corners are not supported here;
//   * `g` and `o` start a new shape, `usemtl` only changes the material of the faces that follow (:2700ff);
//   * MTL: Kd, Ks, Ns, d, Tr (= 1 - value, ignored once d was seen), map_Kd (a texture sets Kd to 0.6 unless
//     a Kd line has been seen — in this or any earlier material of the file); defaults Kd = Ks = 0, Ns = 1,
//     d = 1 (:1334-1379, :1930ff).
// On top of that, the reference's own post-processing: shapes cut into one Mesh per run of equal material
// (with its off-by-one at the last triangle), vertices created lazily and shared when position, normal and
// texture coordinate are all equal, the geometric normal where the file has none, optional centring and
// scaling to the unit sphere in fp32.
//
// A material's texture file (map_Kd) is decoded by Image(path) and attached like the reference does (:131-133).
#include "rt_facade.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <unordered_map>

namespace rtamd {

namespace {

bool is_digit(char c) { return c >= '0' && c <= '9'; }
bool is_space(char c) { return c == ' ' || c == '\t'; }

// tiny_obj_loader.h:866-996
bool try_parse_double(const char* s, const char* s_end, double* result)
{
    if (s >= s_end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', exp_sign = '+';
    const char* curr = s;
    int read = 0;
    bool end_not_reached = false, leading_decimal_dots = false;

    if (*curr == '+' || *curr == '-') {
        sign = *curr;
        curr++;
        if (curr != s_end && *curr == '.') leading_decimal_dots = true;
    } else if (is_digit(*curr)) {
    } else if (*curr == '.') {
        leading_decimal_dots = true;
    } else {
        return false;
    }
    end_not_reached = curr != s_end;
    if (!leading_decimal_dots) {
        while (end_not_reached && is_digit(*curr)) {
            mantissa *= 10;
            mantissa += int(*curr - '0');
            curr++;
            read++;
            end_not_reached = curr != s_end;
        }
        if (read == 0) return false;
    }
    auto assemble = [&]() {
        *result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
        return true;
    };
    if (!end_not_reached) return assemble();
    if (*curr == '.') {
        curr++;
        read = 1;
        end_not_reached = curr != s_end;
        while (end_not_reached && is_digit(*curr)) {
            static const double pow_lut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
            const int lut_entries = int(sizeof pow_lut / sizeof pow_lut[0]);
            mantissa += int(*curr - '0') * (read < lut_entries ? pow_lut[read] : std::pow(10.0, -read));
            read++;
            curr++;
            end_not_reached = curr != s_end;
        }
    } else if (*curr == 'e' || *curr == 'E') {
    } else {
        return assemble();
    }
    if (!end_not_reached) return assemble();
    if (*curr == 'e' || *curr == 'E') {
        curr++;
        end_not_reached = curr != s_end;
        if (end_not_reached && (*curr == '+' || *curr == '-')) {
            exp_sign = *curr;
            curr++;
        } else if (is_digit(*curr)) {
        } else {
            return false;
        }
        read = 0;
        end_not_reached = curr != s_end;
        while (end_not_reached && is_digit(*curr)) {
            if (exponent > 2147483647 / 10) return false;
            exponent *= 10;
            exponent += int(*curr - '0');
            curr++;
            read++;
            end_not_reached = curr != s_end;
        }
        exponent *= exp_sign == '+' ? 1 : -1;
        if (read == 0) return false;
    }
    return assemble();
}

float parse_real(const char** token, double default_value = 0.0)
{
    *token += std::strspn(*token, " \t");
    const char* end = *token + std::strcspn(*token, " \t\r");
    double val = default_value;
    try_parse_double(*token, end, &val);
    *token = end;
    return float(val);
}

std::string parse_string(const char** token)
{
    *token += std::strspn(*token, " \t");
    const size_t e = std::strcspn(*token, " \t\r");
    std::string s(*token, *token + e);
    *token += e;
    return s;
}

struct Corner {
    int v = -1, vt = -1, vn = -1;
};

bool fix_index(int idx, int n, int* ret)
{
    if (idx > 0) {
        *ret = idx - 1;
        return true;
    }
    if (idx == 0) return false;
    *ret = n + idx;
    return true;
}

// i, i/j, i//k, i/j/k (tiny_obj_loader.h:1136-1187)
bool parse_corner(const char** token, int vsize, int vnsize, int vtsize, Corner* ret)
{
    Corner c;
    if (!fix_index(std::atoi(*token), vsize, &c.v)) return false;
    *token += std::strcspn(*token, "/ \t\r");
    if ((*token)[0] != '/') {
        *ret = c;
        return true;
    }
    (*token)++;
    if ((*token)[0] == '/') {
        (*token)++;
        if (!fix_index(std::atoi(*token), vnsize, &c.vn)) return false;
        *token += std::strcspn(*token, "/ \t\r");
        *ret = c;
        return true;
    }
    if (!fix_index(std::atoi(*token), vtsize, &c.vt)) return false;
    *token += std::strcspn(*token, "/ \t\r");
    if ((*token)[0] != '/') {
        *ret = c;
        return true;
    }
    (*token)++;
    if (!fix_index(std::atoi(*token), vnsize, &c.vn)) return false;
    *token += std::strcspn(*token, "/ \t\r");
    *ret = c;
    return true;
}

struct ObjMaterial {
    std::string name;
    float diffuse[3] = { 0, 0, 0 }, specular[3] = { 0, 0, 0 };
    float shininess = 1.0f, dissolve = 1.0f;
    std::string diffuseTexture;
};

void load_mtl(std::istream& in, std::vector<ObjMaterial>& materials, std::map<std::string, int>& byName)
{
    ObjMaterial m;
    bool first = true, has_d = false, has_kd = false;
    std::string linebuf;
    auto flush = [&]() {
        if (!m.name.empty() || !first) {
            byName.insert({ m.name, int(materials.size()) });
            materials.push_back(m);
        }
    };
    while (std::getline(in, linebuf)) {
        while (!linebuf.empty() && (linebuf.back() == '\n' || linebuf.back() == '\r')) linebuf.pop_back();
        if (linebuf.empty()) continue;
        const char* token = linebuf.c_str();
        token += std::strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#') continue;
        if (std::strncmp(token, "newmtl", 6) == 0 && is_space(token[6])) {
            flush();
            first = false;
            m = ObjMaterial();
            has_d = false; // (has_kd is NOT reset by tinyobjloader 2.0.0, :1992-1993: one Kd anywhere earlier in the file counts)
            token += 7;
            m.name = parse_string(&token);
            continue;
        }
        if (token[0] == 'K' && token[1] == 'd' && is_space(token[2])) {
            token += 2;
            for (float& c : m.diffuse) c = parse_real(&token);
            has_kd = true;
            continue;
        }
        if (token[0] == 'K' && token[1] == 's' && is_space(token[2])) {
            token += 2;
            for (float& c : m.specular) c = parse_real(&token);
            continue;
        }
        if (token[0] == 'N' && token[1] == 's' && is_space(token[2])) {
            token += 2;
            m.shininess = parse_real(&token);
            continue;
        }
        if (token[0] == 'd' && is_space(token[1])) {
            token += 1;
            m.dissolve = parse_real(&token);
            has_d = true;
            continue;
        }
        if (token[0] == 'T' && token[1] == 'r' && is_space(token[2])) {
            token += 2;
            if (!has_d) m.dissolve = 1.0f - parse_real(&token);
            continue;
        }
        if (std::strncmp(token, "map_Kd", 6) == 0 && is_space(token[6])) {
            token += 7;
            // options (-o, -s, -blendu, ...) come first; the file name is the last word
            std::string rest(token), word, last;
            std::istringstream words(rest);
            while (words >> word) last = word;
            m.diffuseTexture = last;
            if (!has_kd) m.diffuse[0] = m.diffuse[1] = m.diffuse[2] = 0.6f;
            continue;
        }
    }
    flush();
}

struct ObjShape {
    std::vector<Corner> corners;   // three per triangle
    std::vector<int> materialIds;  // one per triangle
};

struct ObjFile {
    std::vector<float> v, vn, vt;
    std::vector<ObjShape> shapes;
    std::vector<ObjMaterial> materials;
};

[[noreturn]] void fail(const std::string& what)
{
    std::cerr << what << std::endl;
    throw std::exception();
}

ObjFile parse_obj(const std::filesystem::path& file)
{
    std::ifstream in(file);
    if (!in) fail("Failed to load mesh " + file.string());
    const std::filesystem::path baseDir = file.parent_path();
    ObjFile obj;
    std::map<std::string, int> materialByName;
    ObjShape shape;
    std::vector<std::vector<Corner>> faces; // faces read since the last export
    int material = -1;

    // tiny_obj_loader.h:1400-1640, triangulating form
    auto export_faces = [&]() {
        for (const std::vector<Corner>& face : faces) {
            const size_t n = face.size();
            if (n < 3) continue;
            auto bad = [&](const Corner& c) { return size_t(3 * c.v + 2) >= obj.v.size(); };
            if (n == 4) {
                if (bad(face[0]) || bad(face[1]) || bad(face[2]) || bad(face[3])) continue;
                const float* p0 = &obj.v[size_t(face[0].v) * 3];
                const float* p1 = &obj.v[size_t(face[1].v) * 3];
                const float* p2 = &obj.v[size_t(face[2].v) * 3];
                const float* p3 = &obj.v[size_t(face[3].v) * 3];
                const float e02x = p2[0] - p0[0], e02y = p2[1] - p0[1], e02z = p2[2] - p0[2];
                const float e13x = p3[0] - p1[0], e13y = p3[1] - p1[1], e13z = p3[2] - p1[2];
                const float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
                const float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                const int order[2][6] = { { 0, 1, 2, 0, 2, 3 }, { 0, 1, 3, 1, 2, 3 } };
                const int* o = order[sqr02 < sqr13 ? 0 : 1];
                for (int k = 0; k < 6; k++) shape.corners.push_back(face[size_t(o[k])]);
                shape.materialIds.push_back(material);
                shape.materialIds.push_back(material);
            } else if (n == 3) {
                for (const Corner& c : face) shape.corners.push_back(c);
                shape.materialIds.push_back(material);
            } else {
                fail("faces with more than four corners are not supported: " + file.string());
            }
        }
        faces.clear();
    };
    auto end_shape = [&]() {
        export_faces();
        if (!shape.corners.empty()) obj.shapes.push_back(shape);
        shape = ObjShape();
    };

    std::string linebuf;
    while (std::getline(in, linebuf)) {
        while (!linebuf.empty() && (linebuf.back() == '\n' || linebuf.back() == '\r')) linebuf.pop_back();
        if (linebuf.empty()) continue;
        const char* token = linebuf.c_str();
        token += std::strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#') continue;
        if (token[0] == 'v' && is_space(token[1])) {
            token += 2;
            for (int k = 0; k < 3; k++) obj.v.push_back(parse_real(&token));
            continue;
        }
        if (token[0] == 'v' && token[1] == 'n' && is_space(token[2])) {
            token += 3;
            for (int k = 0; k < 3; k++) obj.vn.push_back(parse_real(&token));
            continue;
        }
        if (token[0] == 'v' && token[1] == 't' && is_space(token[2])) {
            token += 3;
            for (int k = 0; k < 2; k++) obj.vt.push_back(parse_real(&token));
            continue;
        }
        if (token[0] == 'f' && is_space(token[1])) {
            token += 2;
            token += std::strspn(token, " \t");
            std::vector<Corner> face;
            while (token[0] != '\0' && token[0] != '\r' && token[0] != '\n') {
                Corner c;
                if (!parse_corner(&token, int(obj.v.size() / 3), int(obj.vn.size() / 3), int(obj.vt.size() / 2), &c))
                    fail("Failed to parse `f' line in " + file.string());
                face.push_back(c);
                token += std::strspn(token, " \t\r");
            }
            faces.push_back(std::move(face));
            continue;
        }
        if (std::strncmp(token, "usemtl", 6) == 0 && is_space(token[6])) {
            token += 7;
            const std::string name = parse_string(&token);
            const auto it = materialByName.find(name);
            const int newId = it != materialByName.end() ? it->second : -1;
            if (newId != material) {
                export_faces();
                material = newId;
            }
            continue;
        }
        if (std::strncmp(token, "mtllib", 6) == 0 && is_space(token[6])) {
            token += 7;
            std::istringstream names { std::string(token) };
            std::string name;
            while (names >> name) {
                std::ifstream mtl(baseDir / name);
                if (mtl) {
                    load_mtl(mtl, obj.materials, materialByName);
                    break;
                }
            }
            continue;
        }
        if ((token[0] == 'g' || token[0] == 'o') && is_space(token[1])) {
            end_shape();
            continue;
        }
        // smoothing groups, tags, lines and points do not reach the meshes
    }
    end_shape();
    // Every index a face names must exist once the whole file is read (tinyobjloader and the reference index the
    // arrays unchecked; a malformed file must fail here instead of reading outside them).  Relative indices further
    // back than the first element arrive here negative.
    const long nV = long(obj.v.size() / 3), nVn = long(obj.vn.size() / 3), nVt = long(obj.vt.size() / 2);
    for (const ObjShape& sh : obj.shapes)
        for (const Corner& c : sh.corners) {
            if (c.v < 0 || c.v >= nV) fail("face names vertex " + std::to_string(c.v + 1) + " but the file has " + std::to_string(nV) + ": " + file.string());
            // (a file without any vn / vt lines may still name them, like the reference's triangle.obj: those are ignored)
            if (c.vn != -1 && nVn > 0 && (c.vn < 0 || c.vn >= nVn)) fail("face names a normal that does not exist: " + file.string());
            if (c.vt != -1 && nVt > 0 && (c.vt < 0 || c.vt >= nVt)) fail("face names a texture coordinate that does not exist: " + file.string());
        }
    return obj;
}

// glm's fp32 forms (func_geometric.inl): cross as written there, normalize = v * (1 / sqrt(dot)), length = sqrt(dot)
vec3 sub(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
float dot3(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
vec3 cross3(vec3 x, vec3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
vec3 normalize3(vec3 v)
{
    const float s = 1.0f / std::sqrt(dot3(v, v));
    return { v.x * s, v.y * s, v.z * s };
}

struct VertexKey {
    float f[8];
    bool operator==(const VertexKey& o) const
    {
        for (int i = 0; i < 8; i++)
            if (!(f[i] == o.f[i])) return false; // float equality like the reference's Vertex::operator==
        return true;
    }
};
struct VertexKeyHash {
    size_t operator()(const VertexKey& k) const
    {
        size_t seed = 0;
        for (float x : k.f) seed ^= std::hash<float>()(x) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};

// reference framework/src/mesh.cpp:151-176
void center_and_scale_to_unit(std::vector<Mesh>& meshes)
{
    vec3 sum(0.0f);
    size_t n = 0;
    for (const Mesh& m : meshes)
        for (const Vertex& v : m.vertices) {
            sum = { sum.x + v.position.x, sum.y + v.position.y, sum.z + v.position.z };
            n++;
        }
    const float fn = float(n);
    const vec3 center { sum.x / fn, sum.y / fn, sum.z / fn };
    float maxD = 0.0f;
    for (const Mesh& m : meshes)
        for (const Vertex& v : m.vertices) {
            const vec3 d = sub(v.position, center);
            maxD = std::max(std::sqrt(dot3(d, d)), maxD);
        }
    for (Mesh& m : meshes)
        for (Vertex& v : m.vertices) {
            const vec3 d = sub(v.position, center);
            v.position = { d.x / maxD, d.y / maxD, d.z / maxD };
        }
}

} // namespace

std::vector<Mesh> loadMesh(const std::filesystem::path& file, bool centerAndNormalize)
{
    if (!std::filesystem::exists(file)) fail("File " + file.string() + " does not exist.");
    const ObjFile obj = parse_obj(file);
    const std::filesystem::path baseDir = file.parent_path();
    auto position = [&](int i) { return vec3(obj.v[size_t(i) * 3], obj.v[size_t(i) * 3 + 1], obj.v[size_t(i) * 3 + 2]); };

    std::vector<Mesh> out;
    for (const ObjShape& shape : obj.shapes) {
        const size_t nTri = shape.corners.size() / 3;
        size_t startTriangle = 0;
        int prevMaterial = shape.materialIds[0];
        // one Mesh per run of equal material, with the reference's treatment of the last triangle (mesh.cpp:78-86)
        for (size_t endTriangle = 0; endTriangle < nTri; ++endTriangle) {
            if (endTriangle == nTri - 1)
                ++endTriangle;
            else if (shape.materialIds[endTriangle] == prevMaterial)
                continue;
            else
                prevMaterial = shape.materialIds[endTriangle];

            Mesh mesh;
            std::unordered_map<VertexKey, uint32_t, VertexKeyHash> cache;
            for (size_t i = startTriangle * 3; i != endTriangle * 3; i += 3) {
                const vec3 v0 = position(shape.corners[i].v), v1 = position(shape.corners[i + 1].v), v2 = position(shape.corners[i + 2].v);
                const vec3 geometricNormal = normalize3(cross3(sub(v1, v0), sub(v2, v0)));
                uint32_t tri[3];
                for (unsigned j = 0; j < 3; j++) {
                    const Corner& c = shape.corners[i + j];
                    Vertex vertex;
                    vertex.position = position(c.v);
                    vertex.normal = (c.vn != -1 && !obj.vn.empty()) ? vec3(obj.vn[size_t(c.vn) * 3], obj.vn[size_t(c.vn) * 3 + 1], obj.vn[size_t(c.vn) * 3 + 2]) : geometricNormal;
                    if (c.vt != -1 && !obj.vt.empty()) vertex.texCoord = { obj.vt[size_t(c.vt) * 2], obj.vt[size_t(c.vt) * 2 + 1] };
                    const VertexKey key { { vertex.position.x, vertex.position.y, vertex.position.z, vertex.normal.x, vertex.normal.y, vertex.normal.z, vertex.texCoord.x, vertex.texCoord.y } };
                    const auto it = cache.find(key);
                    if (it != cache.end()) {
                        tri[j] = it->second;
                    } else {
                        tri[j] = uint32_t(mesh.vertices.size());
                        cache.insert({ key, tri[j] });
                        mesh.vertices.push_back(vertex);
                    }
                }
                mesh.triangles.push_back({ tri[0], tri[1], tri[2] });
            }

            const int materialId = shape.materialIds[startTriangle];
            if (materialId == -1) {
                mesh.material.kd = vec3(1.0f);
                mesh.material.ks = vec3(0.0f);
                mesh.material.shininess = 1.0f;
            } else {
                const ObjMaterial& m = obj.materials[size_t(materialId)];
                mesh.material.kd = { m.diffuse[0], m.diffuse[1], m.diffuse[2] };
                mesh.material.kdTextureFile = m.diffuseTexture;
                if (!m.diffuseTexture.empty()) mesh.material.kdTexture = std::make_shared<Image>(baseDir / m.diffuseTexture);
                mesh.material.ks = { m.specular[0], m.specular[1], m.specular[2] };
                mesh.material.shininess = m.shininess;
                mesh.material.transparency = m.dissolve;
            }
            out.push_back(std::move(mesh));
            startTriangle = endTriangle;
        }
    }
    if (centerAndNormalize) center_and_scale_to_unit(out);
    return out;
}

} // namespace rtamd
