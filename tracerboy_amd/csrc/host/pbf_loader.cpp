/* pbf_loader.cpp -- reader for the `.pbf` binary scene files of the reference's parser.
 *
 * TracerBoy::LoadScene accepts `<scene>.pbf` next to `.pbrt` (TracerBoy.cpp:1210-1223, pbrt::Scene::loadFrom); the format
 * is pbrt-parser's semantic serialisation, format tag 9 (PBRTParser/impl/semantic/BinaryFileFormat.cpp:34-46):
 *
 *   int32 formatTag, then entities in dependency order:  uint64 payloadBytes, int32 typeTag, payload
 *   a reference to another entity is its int32 index in file order (-1 = null); the LAST entity is the Scene.
 *   payload fields are written raw in declaration order: float, vec2/3 as floats, affine3f = 12 floats (vx, vy, vz, p),
 *   bool = 1 byte, std::string = int32 length + bytes, std::vector<T> = uint64 count + elements,
 *   std::map<string, ref> = int32 count + (string, ref) pairs   (BinaryFileFormat.cpp:151-252).
 *
 * This is an independent reader of that format into the build's own PbrtScene (pbrt_scene.h); entity types the hot path
 * does not use (quad meshes, spheres, curves, spot / point lights, samplers ...) are skipped by their size.  Pinned by a
 * `.pbf` written by the reference's own parser (oracle/_ref/pbrt_dump --save-pbf, fixture tests/golden/cornell-box.pbf):
 * loading it must give exactly the scene the `.pbrt` loader gives. */
#include "pbrt_scene.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>

#include <sys/stat.h>

namespace tbhost {
namespace {

enum { /* BinaryFileFormat.cpp:48-109 */
    T_SCENE = 1, T_OBJECT, T_SHAPE, T_INSTANCE, T_CAMERA, T_FILM, T_SPECTRUM, T_SAMPLER, T_INTEGRATOR,
    T_MATERIAL = 10, T_DISNEY, T_UBER, T_MIX, T_GLASS, T_MIRROR, T_MATTE, T_SUBSTRATE, T_SUBSURFACE, T_FOURIER, T_METAL, T_PLASTIC, T_TRANSLUCENT, T_HAIR,
    T_TEXTURE = 30, T_IMAGE_TEXTURE, T_SCALE_TEXTURE, T_PTEX_TEXTURE, T_CONSTANT_TEXTURE, T_CHECKER_TEXTURE, T_WINDY, T_FBM, T_MARBLE, T_MIX_TEXTURE,
        T_WRINKLED,
    T_TRIANGLE_MESH = 50, T_QUAD_MESH, T_SPHERE, T_DISK, T_CURVE,
    T_AREALIGHT_BB = 60, T_AREALIGHT_RGB,
    T_INFINITE_LIGHT = 70, T_DISTANT_LIGHT, T_SPOT_LIGHT, T_POINT_LIGHT,
};

struct AreaLight { bool rgb = false; Vec3 L; };
struct Camera { float fov = 30.f; Affine frame; };
struct Film { int w = 0, h = 0; };
struct SceneRec { int film = -1, world = -1; std::vector<int> cameras; };

struct Entity {
    int tag = 0;
    PbrtTextureSP tex; PbrtMaterialSP mat; PbrtMeshSP mesh; std::shared_ptr<PbrtObject> obj;
    std::shared_ptr<PbrtInstance> inst; std::shared_ptr<PbrtLight> light; std::shared_ptr<AreaLight> area;
    std::shared_ptr<Camera> cam; std::shared_ptr<Film> film; std::shared_ptr<SceneRec> scene;
    bool unsupportedShape = false;
};

struct Cursor {
    const uint8_t* p; size_t n, at = 0;
    void bytes(void* dst, size_t k) { if (at + k > n) throw std::runtime_error("pbf: entity payload too short"); memcpy(dst, p + at, k); at += k; }
    template <class T> T get() { T v; bytes(&v, sizeof v); return v; }
    float f() { return get<float>(); }
    int32_t i32() { return get<int32_t>(); }
    bool b() { return get<uint8_t>() != 0; }
    Vec3 v3() { Vec3 v; v.x = f(); v.y = f(); v.z = f(); return v; }
    Affine affine() { Affine a; a.l.vx = v3(); a.l.vy = v3(); a.l.vz = v3(); a.p = v3(); return a; }
    std::string str() { int32_t k = i32(); if (k < 0) throw std::runtime_error("pbf: negative string length"); std::string s((size_t)k, ' ');
        bytes(&s[0], (size_t)k); return s; }
    uint64_t count(size_t elemBytes) { uint64_t c = get<uint64_t>();
        if (elemBytes && c > (n - at) / elemBytes) throw std::runtime_error("pbf: vector longer than its entity"); return c; }
    void skipSpectrum() { uint64_t c = count(8); at += (size_t)c * 8; } /* Spectrum::spd, vector<pair<float,float>> */
};

struct Reader {
    std::vector<Entity> ents;
    std::string dir;

    template <class T> std::shared_ptr<T> ref(int id, std::shared_ptr<T> Entity::*member)
    {
        if (id == -1) return nullptr;
        if (id < 0 || id >= (int)ents.size()) throw std::runtime_error("pbf: reference to an entity that has not been read");
        return ents[(size_t)id].*member; /* null when the entity is of a type this reader skips */
    }
    PbrtTextureSP tex(Cursor& c) { return ref(c.i32(), &Entity::tex); }
    PbrtMaterialSP mat(Cursor& c) { return ref(c.i32(), &Entity::mat); }
    std::string global(const std::string& f) const { return (f.empty() || f[0] == '/') ? f : dir + f; }

    void material(Entity& e, Cursor& c, const char* type)
    {
        e.mat = std::make_shared<PbrtMaterial>();
        e.mat->type = type; e.mat->name = c.str(); /* Material::readFrom */
    }

    void read(Entity& e, Cursor& c)
    {
        switch (e.tag) {
        case T_TEXTURE: case T_WINDY: case T_FBM: case T_WRINKLED: case T_MARBLE: case T_PTEX_TEXTURE: case T_MIX_TEXTURE:
            e.tex = std::make_shared<PbrtTexture>(); e.tex->kind = "other"; break;
        case T_CONSTANT_TEXTURE: e.tex = std::make_shared<PbrtTexture>(); e.tex->kind = "constant"; e.tex->value = c.v3(); break;
        case T_CHECKER_TEXTURE:
            e.tex = std::make_shared<PbrtTexture>(); e.tex->kind = "checkerboard";
            e.tex->uscale = c.f(); e.tex->vscale = c.f(); e.tex->tex1 = c.v3(); e.tex->tex2 = c.v3(); break;
        case T_IMAGE_TEXTURE:
            e.tex = std::make_shared<PbrtTexture>(); e.tex->kind = "imagemap"; e.tex->fileName = global(c.str()); break; /* uscale, vscale follow: unused */
        case T_SCALE_TEXTURE:
            e.tex = std::make_shared<PbrtTexture>(); e.tex->kind = "scale";
            e.tex->scaleTex1 = tex(c); e.tex->scaleTex2 = tex(c); e.tex->scale1 = c.v3(); e.tex->scale2 = c.v3(); break;

        case T_MATERIAL: material(e, c, "none"); break;
        case T_FOURIER: material(e, c, "fourier"); break;
        case T_SUBSURFACE: material(e, c, "subsurface"); break;
        case T_HAIR: material(e, c, "hair"); break;
        case T_DISNEY: {
            material(e, c, "disney"); PbrtMaterial& m = *e.mat;
            c.f(); c.f(); c.f(); m.color = c.v3(); c.f(); m.eta = c.f(); c.f(); m.metallic = c.f(); m.roughness = c.f(); c.f(); c.f(); m.specTrans = c.f();
            break; }
        case T_UBER: {
            material(e, c, "uber"); PbrtMaterial& m = *e.mat;
            m.kd = c.v3(); m.map_kd = tex(c); m.ks = c.v3(); tex(c); m.kr = c.v3(); tex(c); m.kt = c.v3(); tex(c); m.opacity = c.v3(); tex(c);
            c.f(); tex(c); c.f(); tex(c); m.index = c.f(); m.roughness = c.f(); /* map_roughness, map_bump follow; u/vRoughness are not serialised */
            m.uRoughness = 0.f; m.vRoughness = 0.f;
            break; }
        case T_SUBSTRATE: {
            material(e, c, "substrate"); PbrtMaterial& m = *e.mat;
            m.kd = c.v3(); m.map_kd = tex(c); m.ks = c.v3(); tex(c); tex(c); m.uRoughness = c.f(); tex(c); m.vRoughness = c.f();
            break; }
        case T_MIX: {
            material(e, c, "mix"); PbrtMaterial& m = *e.mat;
            m.material0 = mat(c); m.material1 = mat(c); tex(c); m.amount = c.v3();
            break; }
        case T_TRANSLUCENT: { material(e, c, "translucent"); PbrtMaterial& m = *e.mat; m.map_kd = tex(c); c.v3(); c.v3(); m.kd = c.v3(); break; }
        case T_GLASS: { material(e, c, "glass"); PbrtMaterial& m = *e.mat; m.kr = c.v3(); m.kt = c.v3(); m.index = c.f(); break; }
        case T_MATTE: { material(e, c, "matte"); PbrtMaterial& m = *e.mat; m.map_kd = tex(c); m.kd = c.v3(); m.sigma = c.f(); break; }
        case T_METAL: {
            material(e, c, "metal"); PbrtMaterial& m = *e.mat;
            m.roughness = c.f(); m.uRoughness = c.f(); m.vRoughness = c.f(); c.b(); c.skipSpectrum(); c.skipSpectrum(); m.eta3 = c.v3();
            break; }
        case T_MIRROR: { material(e, c, "mirror"); tex(c); e.mat->kr = c.v3(); break; }
        case T_PLASTIC: { material(e, c, "plastic"); PbrtMaterial& m = *e.mat; m.map_kd = tex(c); tex(c); m.kd = c.v3(); m.ks = c.v3(); m.roughness = c.f();
            break; }

        case T_AREALIGHT_RGB: e.area = std::make_shared<AreaLight>(); e.area->rgb = true; e.area->L = c.v3(); break;
        case T_AREALIGHT_BB: e.area = std::make_shared<AreaLight>(); e.area->rgb = false; break;
        case T_INFINITE_LIGHT: {
            e.light = std::make_shared<PbrtLight>(); PbrtLight& l = *e.light; l.kind = PbrtLight::Infinite;
            l.mapName = c.str(); l.mapFile = l.mapName.empty() ? std::string() : global(l.mapName); l.transform = c.affine(); l.L = c.v3(); l.scale = c.v3();
            break; }
        case T_DISTANT_LIGHT: {
            e.light = std::make_shared<PbrtLight>(); PbrtLight& l = *e.light; l.kind = PbrtLight::Distant;
            l.from = c.v3(); l.to = c.v3(); l.L = c.v3(); l.scale = c.v3(); l.transform = c.affine();
            break; }

        case T_TRIANGLE_MESH: case T_QUAD_MESH: case T_SPHERE: case T_DISK: case T_CURVE: case T_SHAPE: {
            /* Shape::readFrom: material, textures, areaLight, reverseOrientation, alpha */
            PbrtMeshSP mesh = std::make_shared<PbrtMesh>();
            mesh->material = mat(c);
            const int32_t nt = c.i32();
            for (int32_t i = 0; i < nt; i++) { std::string name = c.str(); mesh->textures[name] = tex(c); }
            std::shared_ptr<AreaLight> al = ref(c.i32(), &Entity::area);
            mesh->reverseOrientation = c.get<int8_t>() != 0;
            c.f(); /* alpha */
            if (al) {
                if (!al->rgb) throw std::runtime_error("pbf: blackbody area lights are not supported (TracerBoy.cpp:257 VERIFY)");
                mesh->hasAreaLight = true; mesh->areaLightL = al->L;
            }
            if (e.tag != T_TRIANGLE_MESH) { e.unsupportedShape = true; break; }
            uint64_t n = c.count(12); mesh->vertex.resize((size_t)n); for (Vec3& v : mesh->vertex) v = c.v3();
            n = c.count(12); mesh->normal.resize((size_t)n); for (Vec3& v : mesh->normal) v = c.v3();
            n = c.count(8); mesh->texcoord.resize((size_t)n); for (Vec2& v : mesh->texcoord) { v.x = c.f(); v.y = c.f(); }
            n = c.count(12); mesh->index.resize((size_t)n * 3); for (uint32_t& v : mesh->index) v = (uint32_t)c.i32();
            e.mesh = mesh;
            break; }
        case T_INSTANCE: {
            e.inst = std::make_shared<PbrtInstance>(); e.inst->xfm = c.affine(); e.inst->object = ref(c.i32(), &Entity::obj);
            break; }
        case T_OBJECT: {
            e.obj = std::make_shared<PbrtObject>(); e.obj->name = c.str();
            int32_t n = c.i32();
            for (int32_t i = 0; i < n; i++) { int id = c.i32();
                if (id >= 0 && id < (int)ents.size()) { if (ents[(size_t)id].mesh) e.obj->shapes.push_back(ents[(size_t)id].mesh);
                else if (ents[(size_t)id].unsupportedShape) skipped++; } }
            n = c.i32();
            for (int32_t i = 0; i < n; i++) { std::shared_ptr<PbrtLight> l = ref(c.i32(), &Entity::light); if (l) objectLights[e.obj.get()].push_back(*l); }
            n = c.i32();
            for (int32_t i = 0; i < n; i++) { std::shared_ptr<PbrtInstance> in = ref(c.i32(), &Entity::inst);
                if (in && in->object) e.obj->instances.push_back(*in); }
            break; }
        case T_CAMERA: { e.cam = std::make_shared<Camera>(); e.cam->fov = c.f(); c.f(); c.f(); e.cam->frame = c.affine(); break; }
        case T_FILM: { e.film = std::make_shared<Film>(); e.film->w = c.i32(); e.film->h = c.i32(); break; }
        case T_SCENE: {
            e.scene = std::make_shared<SceneRec>(); e.scene->film = c.i32();
            const uint64_t n = c.count(4);
            for (uint64_t i = 0; i < n; i++) e.scene->cameras.push_back(c.i32());
            e.scene->world = c.i32();
            break; }
        default: break; /* spectrum, sampler, integrator, pixel filter, spot / point lights: not used by the path */
        }
    }

    std::map<const PbrtObject*, std::vector<PbrtLight>> objectLights;
    size_t skipped = 0;
};

} // namespace

std::shared_ptr<PbrtScene> importPBF(const std::string& fileName)
{
    FILE* f = fopen(fileName.c_str(), "rb");
    if (!f) throw std::runtime_error("could not open '" + fileName + "'");
    std::vector<uint8_t> data;
    { struct stat st; if (fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode)) { fclose(f);
        throw std::runtime_error("'" + fileName + "' is not a regular file"); } }
    { fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); data.resize(n > 0 ? (size_t)n : 0);
        if (!data.empty() && fread(data.data(), 1, data.size(), f) != data.size()) { fclose(f); throw std::runtime_error("short read from '" + fileName + "'");
        } }
    fclose(f);
    if (data.size() < 4) throw std::runtime_error("pbf: file too short");
    int32_t tag; memcpy(&tag, data.data(), 4);
    if ((tag >> 16) != 0 || tag < 6 || tag > 9) throw std::runtime_error("pbf: unsupported format tag " + std::to_string(tag) + " (this reader follows tag 9)");
    Reader r;
    { size_t s = fileName.find_last_of('/'); r.dir = s == std::string::npos ? std::string() : fileName.substr(0, s + 1); }
    size_t at = 4;
    while (at + 12 <= data.size()) { /* BinaryFileFormat.cpp:132-147 */
        uint64_t size; int32_t type; memcpy(&size, &data[at], 8); memcpy(&type, &data[at + 8], 4); at += 12;
        if (size > data.size() - at) throw std::runtime_error("pbf: truncated entity");
        Entity e; e.tag = type;
        Cursor c{data.data() + at, (size_t)size};
        r.read(e, c);
        r.ents.push_back(std::move(e));
        at += (size_t)size;
    }
    if (r.ents.empty() || !r.ents.back().scene) throw std::runtime_error("error in Scene::load - no entities"); /* Scene::loadFrom :1672-1680 */
    const SceneRec& sr = *r.ents.back().scene;
    auto scene = std::make_shared<PbrtScene>();
    scene->basePath = r.dir;
    if (sr.film >= 0 && sr.film < (int)r.ents.size() && r.ents[(size_t)sr.film].film) { scene->filmWidth = r.ents[(size_t)sr.film].film->w;
        scene->filmHeight = r.ents[(size_t)sr.film].film->h; }
    if (!sr.cameras.empty() && sr.cameras[0] >= 0 && sr.cameras[0] < (int)r.ents.size() && r.ents[(size_t)sr.cameras[0]].cam) {
        scene->hasCamera = true; scene->cameraFrame = r.ents[(size_t)sr.cameras[0]].cam->frame; scene->fov = r.ents[(size_t)sr.cameras[0]].cam->fov;
    }
    if (sr.world < 0 || sr.world >= (int)r.ents.size() || !r.ents[(size_t)sr.world].obj) throw std::runtime_error("pbf: scene without a world object");
    scene->world = *r.ents[(size_t)sr.world].obj;
    auto it = r.objectLights.find(r.ents[(size_t)sr.world].obj.get());
    if (it != r.objectLights.end()) scene->lights = it->second;
    scene->numSkippedShapes = r.skipped;
    return scene;
}

} // namespace tbhost
