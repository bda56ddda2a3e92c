/* procedural.cpp -- deterministic stand-ins for the benchmark scenes that the reference tree does
 * not contain (dragon: 4 of 16 PLYs missing, vw-van: 1 PLY + env map missing, no Bistro-class scene at
 * all -- /root/reference/.MISSING_LARGE_BLOBS).  Geometry is a function of
 * (kind, targetTriangles, seed) only, built with integer hashing and the host libm-free tb_math.h so
 * that every machine generates identical bits.
 *   kind 0  "dragon-class": one displaced closed blob + ground quad, all matte, white environment,
 *           camera fov 20.1143 (dragon/scene.pbrt:2-6)
 *   kind 1  "van-class": 5x3 blobs in matte / mirror / glass / plastic / metal + emissive quad
 *   kind 2  "bistro-class": 8x6 blobs, 40 materials, 4 emissive quads
 */
#include "host_scene.h"
#include "../../../include/tb_vec.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace tbhost {

namespace {

inline uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
inline float urand(uint32_t& st) { st = pcg(st); return (float)(st >> 8) * (1.0f / 16777216.0f); }

struct MeshOut { std::vector<float> pos, nrm, uv; std::vector<uint32_t> idx; };

/* displaced UV sphere: rings x segs quads */
void makeBlob(MeshOut& m, tb3 c, float radius, uint32_t rings, uint32_t segs, uint32_t seed, float bump)
{
    uint32_t st = seed * 9781u + 17u;
    float ph[6], fr[6];
    for (int i = 0; i < 6; i++) { ph[i] = urand(st) * 6.2831853f; fr[i] = 2.0f + tb_floor(urand(st) * 7.0f); }
    auto radial = [&](float theta, float phi) {
        float d = tb_sin(fr[0] * theta + ph[0]) * tb_sin(fr[1] * phi + ph[1]) + 0.5f * tb_sin(fr[2] * theta + fr[3] * phi + ph[2]) + 0.25f * tb_sin(3.0f *
            fr[4] * phi + ph[3]) * tb_cos(2.0f * fr[5] * theta + ph[4]);
        return radius * (1.0f + bump * d);
    };
    for (uint32_t r = 0; r <= rings; r++) for (uint32_t s = 0; s <= segs; s++) {
        float v = (float)r / (float)rings, u = (float)s / (float)segs;
        float theta = v * 3.14159265f, phi = (s == segs ? 0.0f : u) * 6.2831853f; /* close the seam exactly */
        float st_ = tb_sin(theta), ct = tb_cos(theta), sp = tb_sin(phi), cp = tb_cos(phi);
        float rad = (r == 0 || r == rings) ? radius : radial(theta, phi);
        tb3 dir = tb3_make(st_ * cp, ct, st_ * sp);
        tb3 p = c + dir * rad;
        m.pos.push_back(p.x); m.pos.push_back(p.y); m.pos.push_back(p.z);
        m.nrm.push_back(dir.x); m.nrm.push_back(dir.y); m.nrm.push_back(dir.z); /* refined below */
        m.uv.push_back(u); m.uv.push_back(v);
    }
    uint32_t stride = segs + 1;
    for (uint32_t r = 0; r < rings; r++) for (uint32_t s = 0; s < segs; s++) {
        uint32_t a = r * stride + s, b = a + 1, d = a + stride, e = d + 1;
        if (r != 0) { m.idx.push_back(a); m.idx.push_back(b); m.idx.push_back(e); }
        if (r != rings - 1) { m.idx.push_back(a); m.idx.push_back(e); m.idx.push_back(d); }
    }
    /* smooth normals from face normals (area weighted), seam vertices share */
    std::vector<float> acc(m.pos.size(), 0.0f);
    auto V = [&](uint32_t i) { return tb3_make(m.pos[3 * i], m.pos[3 * i + 1], m.pos[3 * i + 2]); };
    for (size_t t = 0; t + 2 < m.idx.size(); t += 3) {
        uint32_t i0 = m.idx[t], i1 = m.idx[t + 1], i2 = m.idx[t + 2];
        tb3 n = tb3_cross(V(i1) - V(i0), V(i2) - V(i0));
        for (uint32_t i : {i0, i1, i2}) { acc[3 * i] += n.x; acc[3 * i + 1] += n.y; acc[3 * i + 2] += n.z; }
    }
    for (uint32_t r = 0; r <= rings; r++) { /* merge the seam column */
        uint32_t a = r * stride, b = a + segs;
        for (int k = 0; k < 3; k++) { float sum = acc[3 * a + k] + acc[3 * b + k]; acc[3 * a + k] = acc[3 * b + k] = sum; }
    }
    for (size_t i = 0; i < m.pos.size() / 3; i++) {
        tb3 n = tb3_make(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]);
        float l = tb3_length(n);
        tb3 o = tb3_make(m.nrm[3 * i], m.nrm[3 * i + 1], m.nrm[3 * i + 2]);
        if (l > 0.0f) { n = n / l; if (tb3_dot(n, o) < 0.0f) n = -n; } else n = o;
        m.nrm[3 * i] = n.x; m.nrm[3 * i + 1] = n.y; m.nrm[3 * i + 2] = n.z;
    }
}

void makeQuad(MeshOut& m, tb3 p0, tb3 p1, tb3 p2, tb3 p3, tb3 n)
{
    tb3 p[4] = {p0, p1, p2, p3}; const float uv[8] = {0, 0, 1, 0, 1, 1, 0, 1};
    for (int i = 0; i < 4; i++) { m.pos.push_back(p[i].x); m.pos.push_back(p[i].y); m.pos.push_back(p[i].z); m.nrm.push_back(n.x); m.nrm.push_back(n.y);
        m.nrm.push_back(n.z); m.uv.push_back(uv[2 * i]); m.uv.push_back(uv[2 * i + 1]); }
    const uint32_t ix[6] = {0, 1, 2, 0, 2, 3}; m.idx.insert(m.idx.end(), ix, ix + 6);
}

TbMaterial baseMaterial()
{
    TbMaterial m; memset(&m, 0, sizeof m);
    m.IOR = 1.5f; m.albedoIndex = m.alphaIndex = m.normalMapIndex = m.emissiveIndex = m.specularMapIndex = TB_INVALID_TEXTURE;
    m.Flags = TB_MAT_NO_ALPHA;
    return m;
}
TbMaterial matte(float r, float g, float b) { TbMaterial m = baseMaterial(); m.albedo = {r, g, b}; m.Flags |= TB_MAT_NO_SPECULAR; return m; }
TbMaterial mirror() { TbMaterial m = baseMaterial(); m.albedo = {0.9f, 0.9f, 0.9f}; m.SpecularCoef = 1.0f; m.Flags |= TB_MAT_METALLIC; return m; }
TbMaterial metal(float rough) { TbMaterial m = baseMaterial(); m.albedo = {1, 1, 1}; m.IOR = 0.7434f; m.roughness = rough; m.Flags |= TB_MAT_METALLIC;
    return m; }
TbMaterial glass() { TbMaterial m = baseMaterial(); m.IOR = 1.5f; m.Flags |= TB_MAT_SUBSURFACE_SCATTER; return m; }
TbMaterial plastic(float r, float g, float b, float rough) { TbMaterial m = baseMaterial(); m.albedo = {r, g, b}; m.roughness = rough; m.SpecularCoef = 0.25f;
    m.IOR = 3.0f; return m; }
TbMaterial emitter(float r, float g, float b) { TbMaterial m = matte(0, 0, 0); m.emissive = {r, g, b}; m.Flags |= TB_MAT_LIGHT; return m; }

void addMesh(HostScene& s, const MeshOut& m, uint32_t materialIndex, bool isLight, tb3 L)
{
    const uint32_t firstVertex = (uint32_t)(s.positions.size() / 3);
    const uint32_t vbOff = (uint32_t)(s.vertexBuffer.size() * 4);
    for (size_t v = 0; v < m.pos.size() / 3; v++) {
        const float vert[8] = {m.nrm[3 * v], m.nrm[3 * v + 1], m.nrm[3 * v + 2], m.uv[2 * v], m.uv[2 * v + 1], 0, 0, 1};
        s.vertexBuffer.insert(s.vertexBuffer.end(), vert, vert + 8);
        s.positions.push_back(m.pos[3 * v]); s.positions.push_back(m.pos[3 * v + 1]); s.positions.push_back(m.pos[3 * v + 2]);
    }
    while (s.indexBuffer.size() % 4) s.indexBuffer.push_back(0);
    const uint32_t ibOff = (uint32_t)(s.indexBuffer.size() * 4);
    const uint32_t geom = (uint32_t)s.hitGroups.size();
    for (size_t t = 0; t + 2 < m.idx.size(); t += 3) {
        for (int k = 0; k < 3; k++) { s.indexBuffer.push_back(m.idx[t + k]); s.triVertexIndex.push_back(firstVertex + m.idx[t + k]); }
        s.triGeometry.push_back(geom); s.triPrimitive.push_back((uint32_t)(t / 3)); s.triFlags.push_back(1u);
        if (isLight) {
            TbLight l; memset(&l, 0, sizeof l); l.LightType = TB_LIGHT_TYPE_AREA; l.LightColor = {L.x, L.y, L.z};
            auto V = [&](uint32_t i) { return tb3_make(m.pos[3 * i], m.pos[3 * i + 1], m.pos[3 * i + 2]); };
            auto Nn = [&](uint32_t i) { return TbFloat3{m.nrm[3 * i], m.nrm[3 * i + 1], m.nrm[3 * i + 2]}; };
            tb3 p0 = V(m.idx[t]), p1 = V(m.idx[t + 1]), p2 = V(m.idx[t + 2]);
            l.SurfaceArea = 0.5f * tb3_length(tb3_cross(p1 - p0, p2 - p0));
            l.P0 = {p0.x, p0.y, p0.z}; l.P1 = {p1.x, p1.y, p1.z}; l.P2 = {p2.x, p2.y, p2.z};
            l.N0 = Nn(m.idx[t]); l.N1 = Nn(m.idx[t + 1]); l.N2 = Nn(m.idx[t + 2]);
            s.lights.push_back(l);
        }
    }
    TbHitGroupRecord rec; memset(&rec, 0, sizeof rec);
    rec.GeometryIndex = geom; rec.MaterialIndex = materialIndex; rec.VertexBufferOffset = vbOff; rec.IndexBufferOffset = ibOff;
    s.hitGroups.push_back(rec);
}

void setCamera(HostScene& s, tb3 eye, tb3 target, float fovDeg)
{
    tb3 view = tb3_normalize(target - eye), up0 = tb3_make(0, 1, 0);
    tb3 right = tb3_normalize(tb3_cross(up0, view)), up = tb3_cross(view, right);
    s.camera.LensHeight = 2.0f;
    s.camera.FocalDistance = (float)(1.0 / tan((double)fovDeg * M_PI / 360.0));
    tb3 pos = eye + view * (s.camera.FocalDistance + 0.01f), look = pos + view;
    const tb3 v[4] = {pos, look, right, up}; float* d[4] = {s.camera.Position, s.camera.LookAt, s.camera.Right, s.camera.Up};
    for (int i = 0; i < 4; i++) { d[i][0] = v[i].x; d[i][1] = v[i].y; d[i][2] = v[i].z; }
}

} // namespace

void MakeProceduralScene(HostScene& s, int kind, uint32_t targetTriangles, uint32_t seed)
{
    if (kind < 0 || kind > 2) throw std::runtime_error("unknown procedural scene kind");
    s = HostScene();
    s.filmWidth = 1920; s.filmHeight = 1080;
    uint32_t gx = kind == 0 ? 1 : (kind == 1 ? 5 : 8), gz = kind == 0 ? 1 : (kind == 1 ? 3 : 6);
    uint32_t blobs = gx * gz;
    if (targetTriangles < 64 * blobs) targetTriangles = 64 * blobs;
    /* 2*segs*(rings-1) triangles per blob with segs = 2*rings */
    double perBlob = (double)targetTriangles / blobs;
    uint32_t rings = (uint32_t)floor(sqrt(perBlob / 4.0) + 0.5); if (rings < 3) rings = 3;
    uint32_t segs = 2 * rings;

    if (kind == 0) { s.materials.push_back(matte(0.725f, 0.71f, 0.68f)); s.materials.push_back(matte(0.6f, 0.55f, 0.4f)); }
    else if (kind == 1) {
        s.materials.push_back(matte(0.5f, 0.5f, 0.5f)); s.materials.push_back(matte(0.63f, 0.065f, 0.05f)); s.materials.push_back(mirror());
        s.materials.push_back(glass()); s.materials.push_back(plastic(0.14f, 0.45f, 0.091f, 0.1f)); s.materials.push_back(metal(0.2f));
    } else {
        s.materials.push_back(matte(0.5f, 0.5f, 0.5f));
        uint32_t st = seed ^ 0x9e3779b9u;
        for (int i = 0; i < 39; i++) {
            float r = 0.1f + 0.8f * urand(st), g = 0.1f + 0.8f * urand(st), b = 0.1f + 0.8f * urand(st);
            switch (i % 5) { case 0: case 1: s.materials.push_back(matte(r, g, b)); break;
                case 2: s.materials.push_back(plastic(r, g, b, 0.05f + 0.3f * urand(st))); break;
                case 3: s.materials.push_back(metal(0.02f + 0.3f * urand(st))); break; default: s.materials.push_back(i % 10 == 4 ? glass() : mirror()); break;
                    }
        }
    }
    const uint32_t numSurfaceMaterials = (uint32_t)s.materials.size();

    float spacing = 2.4f, width = spacing * gx, depth = spacing * gz;
    { MeshOut g; float hx = width * 0.5f + 4.0f, hz = depth * 0.5f + 4.0f;
      makeQuad(g, tb3_make(-hx, 0, -hz), tb3_make(-hx, 0, hz), tb3_make(hx, 0, hz), tb3_make(hx, 0, -hz), tb3_make(0, 1, 0));
          addMesh(s, g, 0, false, tb3_splat(0)); }
    uint32_t st = seed * 2654435761u + 12345u;
    for (uint32_t iz = 0; iz < gz; iz++) for (uint32_t ix = 0; ix < gx; ix++) {
        uint32_t b = iz * gx + ix;
        float rad = kind == 0 ? 1.0f : 0.7f + 0.3f * urand(st);
        tb3 c = tb3_make(((float)ix + 0.5f) * spacing - width * 0.5f, rad * 1.05f + 0.02f, ((float)iz + 0.5f) * spacing - depth * 0.5f);
        MeshOut m; makeBlob(m, c, rad, rings, segs, seed + 31u * b, kind == 0 ? 0.12f : 0.08f);
        uint32_t mat = kind == 0 ? 1u : 1u + (b % (numSurfaceMaterials - 1));
        addMesh(s, m, mat, false, tb3_splat(0));
    }
    if (kind != 0) {
        uint32_t nl = kind == 1 ? 1 : 4;
        for (uint32_t i = 0; i < nl; i++) {
            float cx = nl == 1 ? 0.0f : (((float)(i % 2) - 0.5f) * width * 0.5f), cz = nl == 1 ? 0.0f : (((float)(i / 2) - 0.5f) * depth * 0.5f), y = 5.0f,
                h = 1.0f;
            MeshOut q;
                makeQuad(q, tb3_make(cx - h, y, cz - h), tb3_make(cx + h, y, cz - h), tb3_make(cx + h, y, cz + h), tb3_make(cx - h, y, cz + h), tb3_make(0, -1,
                0));
            s.materials.push_back(emitter(17.0f, 12.0f, 4.0f));
            addMesh(s, q, (uint32_t)s.materials.size() - 1, true, tb3_make(17.0f, 12.0f, 4.0f));
        }
    }
    /* constant white environment for kind 0, dim sky otherwise */
    s.envMap.assign(1, TbFloat4{1, 1, 1, 1}); s.envWidth = s.envHeight = 1;
    memset(&s.config, 0, sizeof s.config);
    s.config.EnvMapTransformVx = {1, 0, 0, 0}; s.config.EnvMapTransformVy = {0, 1, 0, 0}; s.config.EnvMapTransformVz = {0, 0, 1, 0};
    float sky = kind == 0 ? 1.0f : 0.05f;
    s.config.EnvironmentMapColorScale = {sky, sky, sky};
    float dist = kind == 0 ? 9.0f : (kind == 1 ? 16.0f : 26.0f);
    setCamera(s, tb3_make(0.0f, dist * 0.35f, dist), tb3_make(0.0f, kind == 0 ? 1.0f : 0.8f, 0.0f), kind == 0 ? 20.1143f : 35.0f);
    s.config.CameraLensHeight = s.camera.LensHeight;
    tb3 mn = tb3_splat(3.4e38f), mx = tb3_splat(-3.4e38f);
    for (size_t i = 0; i + 2 < s.positions.size(); i += 3) { tb3 p = tb3_make(s.positions[i], s.positions[i + 1], s.positions[i + 2]); mn = tb3_min(mn, p);
        mx = tb3_max(mx, p); }
    s.sceneMin[0] = mn.x; s.sceneMin[1] = mn.y; s.sceneMin[2] = mn.z; s.sceneMax[0] = mx.x; s.sceneMax[1] = mx.y; s.sceneMax[2] = mx.z;
}

} // namespace tbhost
