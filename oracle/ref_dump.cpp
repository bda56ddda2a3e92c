/* ref_dump.cpp -- dumps what the REFERENCE's own PBRT parser (pbrt::importPBRT,
 * /root/reference/PBRTParser) produces for a scene, so the build's independent loader
 * (tracerboy_amd/csrc/host/pbrt_loader.cpp) can be pinned against it.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by `make -C oracle ref` into oracle/_ref/pbrt_dump, compiling the
 * reference's parser sources where they lie (nothing of the reference is copied into this repo).
 * Runs only in the build container; its output is turned into small fixtures under tests/golden/ by
 * oracle/make_golden.py.
 *
 * Output (text): one record per line, `name n v0 v1 ...`; floats are printed as their IEEE-754 bit
 * patterns (hex) so the fixture is exact.
 */
#include "pbrtParser/Scene.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <typeinfo>

using namespace pbrt;

static void putf(FILE* f, const char* name, const float* v, size_t n)
{
    fprintf(f, "%s %zu", name, n);
    for (size_t i = 0; i < n; i++) { uint32_t u; memcpy(&u, v + i, 4); fprintf(f, " %08x", u); }
    fprintf(f, "\n");
}
static void puti(FILE* f, const char* name, const int* v, size_t n)
{
    fprintf(f, "%s %zu", name, n);
    for (size_t i = 0; i < n; i++) fprintf(f, " %d", v[i]);
    fprintf(f, "\n");
}

static const char* materialType(const Material::SP& m)
{
    if (!m) return "null";
    if (std::dynamic_pointer_cast<DisneyMaterial>(m)) return "disney";
    if (std::dynamic_pointer_cast<UberMaterial>(m)) return "uber";
    if (std::dynamic_pointer_cast<MixMaterial>(m)) return "mix";
    if (std::dynamic_pointer_cast<MirrorMaterial>(m)) return "mirror";
    if (std::dynamic_pointer_cast<MetalMaterial>(m)) return "metal";
    if (std::dynamic_pointer_cast<SubstrateMaterial>(m)) return "substrate";
    if (std::dynamic_pointer_cast<GlassMaterial>(m)) return "glass";
    if (std::dynamic_pointer_cast<FourierMaterial>(m)) return "fourier";
    if (std::dynamic_pointer_cast<MatteMaterial>(m)) return "matte";
    if (std::dynamic_pointer_cast<PlasticMaterial>(m)) return "plastic";
    if (std::dynamic_pointer_cast<SubSurfaceMaterial>(m)) return "subsurface";
    if (std::dynamic_pointer_cast<TranslucentMaterial>(m)) return "translucent";
    return "other";
}

static void dumpMaterial(FILE* f, const Material::SP& m)
{
    fprintf(f, "material_type 1 %s\n", materialType(m));
    if (!m) return;
    fprintf(f, "material_ptr 1 %p\n", (void*)m.get());
    if (auto p = std::dynamic_pointer_cast<MatteMaterial>(m)) { putf(f, "kd", &p->kd.x, 3); putf(f, "sigma", &p->sigma, 1);
        fprintf(f, "map_kd 1 %d\n", p->map_kd ? 1 : 0); }
    if (auto p = std::dynamic_pointer_cast<SubstrateMaterial>(m)) { putf(f, "kd", &p->kd.x, 3); putf(f, "ks", &p->ks.x, 3);
        putf(f, "uroughness", &p->uRoughness, 1); putf(f, "vroughness", &p->vRoughness, 1); }
    if (auto p = std::dynamic_pointer_cast<PlasticMaterial>(m)) { putf(f, "kd", &p->kd.x, 3); putf(f, "ks", &p->ks.x, 3);
        putf(f, "roughness", &p->roughness, 1); }
    if (auto p = std::dynamic_pointer_cast<UberMaterial>(m)) { putf(f, "kd", &p->kd.x, 3); putf(f, "ks", &p->ks.x, 3); putf(f, "kt", &p->kt.x, 3);
        putf(f, "opacity", &p->opacity.x, 3); putf(f, "index", &p->index, 1); putf(f, "roughness", &p->roughness, 1); putf(f, "uroughness", &p->uRoughness, 1);
        }
    if (auto p = std::dynamic_pointer_cast<MirrorMaterial>(m)) { putf(f, "kr", &p->kr.x, 3); }
    if (auto p = std::dynamic_pointer_cast<MetalMaterial>(m)) { putf(f, "eta", &p->eta.x, 3); putf(f, "roughness", &p->roughness, 1);
        putf(f, "uroughness", &p->uRoughness, 1); }
    if (auto p = std::dynamic_pointer_cast<GlassMaterial>(m)) { putf(f, "index", &p->index, 1); }
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: pbrt_dump scene.pbrt out.txt | pbrt_dump --save-pbf scene.pbrt out.pbf | pbrt_dump scene.pbf out.txt\n"); return 2;
        }
    if (!strcmp(argv[1], "--save-pbf")) { /* the reference parser's own binary writer (pbrt::Scene::saveTo), for the .pbf reader's fixture */
        if (argc < 4) return 2;
        try { Scene::SP s = importPBRT(argv[2]); s->saveTo(argv[3]); }
        catch (const std::exception& e) { fprintf(stderr, "save-pbf threw: %s\n", e.what()); return 1; }
        return 0;
    }
    Scene::SP scene;
    try {
        const std::string in = argv[1];
        scene = (in.size() > 4 && in.compare(in.size() - 4, 4, ".pbf") == 0) ? Scene::loadFrom(in) : importPBRT(in);
    }
    catch (const std::exception& e) { fprintf(stderr, "importPBRT threw: %s\n", e.what()); return 1; }
    FILE* f = fopen(argv[2], "w");
    if (!f) return 3;
    fprintf(f, "num_cameras 1 %zu\n", scene->cameras.size());
    if (!scene->cameras.empty()) {
        auto& c = scene->cameras[0];
        putf(f, "camera_fov", &c->fov, 1);
        putf(f, "camera_frame_vx", &c->frame.l.vx.x, 3); putf(f, "camera_frame_vy", &c->frame.l.vy.x, 3);
        putf(f, "camera_frame_vz", &c->frame.l.vz.x, 3); putf(f, "camera_frame_p", &c->frame.p.x, 3);
    }
    if (scene->film) { int r[2] = {scene->film->resolution.x, scene->film->resolution.y}; puti(f, "film", r, 2); }
    fprintf(f, "num_shapes 1 %zu\n", scene->world->shapes.size());
    fprintf(f, "num_instances 1 %zu\n", scene->world->instances.size());
    fprintf(f, "num_lights 1 %zu\n", scene->world->lightSources.size());
    size_t si = 0;
    for (auto& shape : scene->world->shapes) {
        fprintf(f, "shape 1 %zu\n", si++);
        TriangleMesh::SP mesh = std::dynamic_pointer_cast<TriangleMesh>(shape);
        if (!mesh) { fprintf(f, "shape_kind 1 other\n"); continue; }
        fprintf(f, "shape_kind 1 trianglemesh\n");
        putf(f, "vertex", mesh->vertex.empty() ? nullptr : &mesh->vertex[0].x, mesh->vertex.size() * 3);
        putf(f, "normal", mesh->normal.empty() ? nullptr : &mesh->normal[0].x, mesh->normal.size() * 3);
        putf(f, "texcoord", mesh->texcoord.empty() ? nullptr : &mesh->texcoord[0].x, mesh->texcoord.size() * 2);
        puti(f, "index", mesh->index.empty() ? nullptr : &mesh->index[0].x, mesh->index.size() * 3);
        dumpMaterial(f, mesh->material);
        DiffuseAreaLightRGB::SP al = std::dynamic_pointer_cast<DiffuseAreaLightRGB>(mesh->areaLight);
        if (al) putf(f, "area_light_L", &al->L.x, 3);
        for (auto& t : mesh->textures) fprintf(f, "shape_texture 1 %s\n", t.first.c_str());
    }
    for (auto& l : scene->world->lightSources) {
        if (auto inf = std::dynamic_pointer_cast<InfiniteLightSource>(l)) {
            fprintf(f, "light_infinite 1 %s\n", inf->mapName.c_str());
            putf(f, "light_transform_vx", &inf->transform.l.vx.x, 3); putf(f, "light_transform_vy", &inf->transform.l.vy.x, 3);
            putf(f, "light_transform_vz", &inf->transform.l.vz.x, 3); putf(f, "light_scale", &inf->scale.x, 3);
        } else if (auto d = std::dynamic_pointer_cast<DistantLightSource>(l)) {
            putf(f, "light_distant_from", &d->from.x, 3); putf(f, "light_distant_to", &d->to.x, 3); putf(f, "light_distant_L", &d->L.x, 3);
        } else fprintf(f, "light_other 1 1\n");
    }
    fclose(f);
    return 0;
}
