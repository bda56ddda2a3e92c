/* pbrt_scene.h -- semantic scene produced by the build's own PBRT-v3 subset loader.
 *
 * Plays the role of pbrt::Scene (/root/reference/PBRTParser/include/pbrtParser/Scene.h) for the
 * directives the reference's scenes use.  The vector/affine helpers restate
 * pbrtParser/math.h:159-191 with the same operation order so that world-space vertices, normals
 * and the camera frame come out bit-identical to the reference parser's (pinned by
 * tests/golden/*.scene.bin, dumped by oracle/_ref/pbrt_dump).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace tbhost {

struct Vec2 { float x = 0, y = 0; };
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    explicit Vec3(float v) : x(v), y(v), z(v) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct Mat3 { Vec3 vx{1, 0, 0}, vy{0, 1, 0}, vz{0, 0, 1}; };
struct Affine { Mat3 l; Vec3 p{0, 0, 0}; };

inline Vec3 operator-(const Vec3& a) { return Vec3(-a.x, -a.y, -a.z); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator*(const Vec3& a, float b) { return Vec3(a.x * b, a.y * b, a.z * b); }
inline Vec3 operator*(float a, const Vec3& b) { return Vec3(a * b.x, a * b.y, a * b.z); }
inline Mat3 operator*(const Mat3& a, float b) { Mat3 m; m.vx = a.vx * b; m.vy = a.vy * b; m.vz = a.vz * b; return m; }
inline Vec3 operator*(const Mat3& a, const Vec3& b) { return a.vx * b.x + a.vy * b.y + a.vz * b.z; }
inline Mat3 operator*(const Mat3& a, const Mat3& b) { Mat3 m; m.vx = a * b.vx; m.vy = a * b.vy; m.vz = a * b.vz; return m; }
inline Vec3 operator*(const Affine& a, const Vec3& b) { return a.l * b + a.p; }
inline Affine operator*(const Affine& a, const Affine& b) { Affine r; r.l = a.l * b.l; r.p = a.l * b.p + a.p; return r; }
inline float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3& a, const Vec3& b) { return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline Vec3 normalize(const Vec3& a) { return a * (1 / sqrtf(dot(a, a))); }
inline Mat3 transpose(const Mat3& a) { Mat3 m; m.vx = Vec3(a.vx.x, a.vy.x, a.vz.x); m.vy = Vec3(a.vx.y, a.vy.y, a.vz.y); m.vz = Vec3(a.vx.z, a.vy.z, a.vz.z);
    return m; }
inline float determinant(const Mat3& a) { return dot(a.vx, cross(a.vy, a.vz)); }
inline Mat3 adjoint_transpose(const Mat3& a) { Mat3 m; m.vx = cross(a.vy, a.vz); m.vy = cross(a.vz, a.vx); m.vz = cross(a.vx, a.vy); return m; }
inline Mat3 inverse_transpose(const Mat3& a) { return adjoint_transpose(a) * (1 / determinant(a)); }
inline Mat3 inverse(const Mat3& a) { return transpose(inverse_transpose(a)); }
inline Affine inverse(const Affine& a) { Affine r; r.l = inverse(a.l); r.p = -(r.l * a.p); return r; }
inline Vec3 xfmPoint(const Affine& m, const Vec3& p) { return m * p; }
inline Vec3 xfmVector(const Affine& m, const Vec3& v) { return m.l * v; }
inline Vec3 xfmNormal(const Affine& m, const Vec3& n) { return inverse_transpose(m.l) * n; }

struct PbrtTexture {
    std::string name, kind /* "imagemap" | "checkerboard" | "scale" | "constant" | other */;
    std::string fileName;            /* imagemap (already made global) */
    float uscale = 1, vscale = 1;    /* checkerboard */
    Vec3 tex1{0, 0, 0}, tex2{1, 1, 1};
    std::shared_ptr<PbrtTexture> scaleTex1, scaleTex2; /* scale */
    Vec3 scale1{1, 1, 1}, scale2{1, 1, 1};
    Vec3 value{1, 1, 1};             /* constant */
};
typedef std::shared_ptr<PbrtTexture> PbrtTextureSP;

struct PbrtMaterial {
    std::string type, name;
    /* union of the parameters TracerBoy's CreateMaterial reads (TracerBoy.cpp:273-505); defaults per
     * type are filled in by the loader from pbrtParser/Scene.h:425-738 */
    Vec3 kd, ks, kr, kt, opacity{1, 1, 1}, color{0.5f, 0.5f, 0.5f}, eta3, amount{0.5f, 0.5f, 0.5f};
    float roughness = 0, uRoughness = 0, vRoughness = 0, index = 1.5f, sigma = 0, eta = 1.5f, metallic = 0, specTrans = 0;
    PbrtTextureSP map_kd, map_normal, map_emissive, map_specular;
    std::shared_ptr<PbrtMaterial> material0, material1;
};
typedef std::shared_ptr<PbrtMaterial> PbrtMaterialSP;

struct PbrtMesh {
    std::vector<Vec3> vertex, normal, tangents;
    std::vector<Vec2> texcoord;
    std::vector<uint32_t> index; /* 3 per triangle */
    PbrtMaterialSP material;
    bool hasAreaLight = false;
    Vec3 areaLightL{0, 0, 0};
    std::map<std::string, PbrtTextureSP> textures; /* e.g. "alpha" */
    bool reverseOrientation = false;
};
typedef std::shared_ptr<PbrtMesh> PbrtMeshSP;

struct PbrtObject;
struct PbrtInstance { Affine xfm; std::shared_ptr<PbrtObject> object; };
struct PbrtObject {
    std::string name;
    std::vector<PbrtMeshSP> shapes;
    std::vector<PbrtInstance> instances;
};

struct PbrtLight {
    enum Kind { Infinite, Distant } kind = Infinite;
    Affine transform;
    std::string mapName;  /* as written in the scene file (what the reference parser records) */
    std::string mapFile;  /* resolved against the scene directory */
    Vec3 L{1, 1, 1}, scale{1, 1, 1}, from{0, 0, 0}, to{0, 0, 1};
};

struct PbrtScene {
    bool hasCamera = false;
    Affine cameraFrame; /* inverse(CTM at Camera), Camera.cpp:102 */
    float fov = 30.0f;
    int filmWidth = 0, filmHeight = 0;
    PbrtObject world;
    std::vector<PbrtLight> lights;
    std::string basePath;
    size_t numSkippedShapes = 0;
};

/* Throws std::runtime_error (like pbrt::importPBRT, impl/semantic/importPBRT.cpp:26-42). */
std::shared_ptr<PbrtScene> importPBRT(const std::string& fileName);
/* `.pbf`, the reference parser's binary scene format (pbf_loader.cpp; <-> pbrt::Scene::loadFrom, TracerBoy.cpp:1210-1223) */
std::shared_ptr<PbrtScene> importPBF(const std::string& fileName);
/* by extension: ".pbf" -> importPBF, anything else -> importPBRT (TracerBoy.cpp:1189-1224) */
inline std::shared_ptr<PbrtScene> importScene(const std::string& fileName)
{
    const size_t n = fileName.size();
    return (n >= 4 && fileName.compare(n - 4, 4, ".pbf") == 0) ? importPBF(fileName) : importPBRT(fileName);
}

/* Binary PLY reader (triangles only; any other face arity throws, Geometry.cpp:46-66). */
void readPly(const std::string& fileName, std::vector<Vec3>& pos, std::vector<Vec3>& nor, std::vector<Vec2>& uv, std::vector<uint32_t>& idx);

} // namespace tbhost
