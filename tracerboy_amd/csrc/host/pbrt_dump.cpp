/* pbrt_dump.cpp -- writes the build's own parse of a PBRT scene in the record format of
 * oracle/ref_dump.cpp (which dumps the reference parser's result), so the two can be diffed.
 * Host-only diagnostic entry point of the C ABI: tb_host_pbrt_dump(). */
#include "pbrt_scene.h"
#include "../../../include/tracerboy_hip.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

using namespace tbhost;

namespace {
void putf(FILE* f, const char* name, const float* v, size_t n)
{
    fprintf(f, "%s %zu", name, n);
    for (size_t i = 0; i < n; i++) { uint32_t u; memcpy(&u, v + i, 4); fprintf(f, " %08x", u); }
    fprintf(f, "\n");
}
void dumpMaterial(FILE* f, const PbrtMaterialSP& m)
{
    std::string t = m ? m->type : std::string("null");
    static const char* known[] = {"disney", "uber", "mix", "mirror", "metal", "substrate", "glass", "fourier", "matte", "plastic", "subsurface", "translucent"};
    bool ok = !m; for (const char* k : known) if (t == k) ok = true;
    fprintf(f, "material_type 1 %s\n", ok ? t.c_str() : "other");
    if (!m) return;
    if (t == "matte") { putf(f, "kd", &m->kd.x, 3); putf(f, "sigma", &m->sigma, 1); fprintf(f, "map_kd 1 %d\n", m->map_kd ? 1 : 0); }
    if (t == "substrate") { putf(f, "kd", &m->kd.x, 3); putf(f, "ks", &m->ks.x, 3); putf(f, "uroughness", &m->uRoughness, 1);
        putf(f, "vroughness", &m->vRoughness, 1); }
    if (t == "plastic") { putf(f, "kd", &m->kd.x, 3); putf(f, "ks", &m->ks.x, 3); putf(f, "roughness", &m->roughness, 1); }
    if (t == "uber") { putf(f, "kd", &m->kd.x, 3); putf(f, "ks", &m->ks.x, 3); putf(f, "kt", &m->kt.x, 3); putf(f, "opacity", &m->opacity.x, 3);
        putf(f, "index", &m->index, 1); putf(f, "roughness", &m->roughness, 1); putf(f, "uroughness", &m->uRoughness, 1); }
    if (t == "mirror") putf(f, "kr", &m->kr.x, 3);
    if (t == "metal") { putf(f, "eta", &m->eta3.x, 3); putf(f, "roughness", &m->roughness, 1); putf(f, "uroughness", &m->uRoughness, 1); }
    if (t == "glass") putf(f, "index", &m->index, 1);
}
} // namespace

extern "C" int tb_host_pbrt_dump(const char* pbrt_path, const char* out_path, char* err, uint32_t errLen)
{
    try {
        std::shared_ptr<PbrtScene> s = importScene(pbrt_path);
        FILE* f = fopen(out_path, "w");
        if (!f) throw std::runtime_error(std::string("could not open '") + out_path + "' for writing");
        fprintf(f, "num_cameras 1 %d\n", s->hasCamera ? 1 : 0);
        if (s->hasCamera) {
            putf(f, "camera_fov", &s->fov, 1);
            putf(f, "camera_frame_vx", &s->cameraFrame.l.vx.x, 3); putf(f, "camera_frame_vy", &s->cameraFrame.l.vy.x, 3);
            putf(f, "camera_frame_vz", &s->cameraFrame.l.vz.x, 3); putf(f, "camera_frame_p", &s->cameraFrame.p.x, 3);
        }
        if (s->filmWidth || s->filmHeight) fprintf(f, "film 2 %d %d\n", s->filmWidth, s->filmHeight);
        fprintf(f, "num_shapes 1 %zu\n", s->world.shapes.size() + s->numSkippedShapes);
        fprintf(f, "num_instances 1 %zu\n", s->world.instances.size());
        fprintf(f, "num_lights 1 %zu\n", s->lights.size());
        size_t si = 0;
        for (const PbrtMeshSP& m : s->world.shapes) {
            fprintf(f, "shape 1 %zu\n", si++);
            fprintf(f, "shape_kind 1 trianglemesh\n");
            putf(f, "vertex", m->vertex.empty() ? nullptr : &m->vertex[0].x, m->vertex.size() * 3);
            putf(f, "normal", m->normal.empty() ? nullptr : &m->normal[0].x, m->normal.size() * 3);
            putf(f, "texcoord", m->texcoord.empty() ? nullptr : &m->texcoord[0].x, m->texcoord.size() * 2);
            fprintf(f, "index %zu", m->index.size());
            for (uint32_t i : m->index) fprintf(f, " %d", (int)i);
            fprintf(f, "\n");
            dumpMaterial(f, m->material);
            if (m->hasAreaLight) putf(f, "area_light_L", &m->areaLightL.x, 3);
            for (auto& t : m->textures) fprintf(f, "shape_texture 1 %s\n", t.first.c_str());
        }
        for (const PbrtLight& l : s->lights) {
            if (l.kind == PbrtLight::Infinite) {
                fprintf(f, "light_infinite 1 %s\n", l.mapName.c_str());
                putf(f, "light_transform_vx", &l.transform.l.vx.x, 3); putf(f, "light_transform_vy", &l.transform.l.vy.x, 3);
                putf(f, "light_transform_vz", &l.transform.l.vz.x, 3); putf(f, "light_scale", &l.scale.x, 3);
            } else { putf(f, "light_distant_from", &l.from.x, 3); putf(f, "light_distant_to", &l.to.x, 3); putf(f, "light_distant_L", &l.L.x, 3); }
        }
        fclose(f);
        return TB_OK;
    } catch (const std::exception& e) {
        if (err && errLen) { strncpy(err, e.what(), errLen - 1); err[errLen - 1] = 0; }
        return TB_E_PARSE;
    }
}
