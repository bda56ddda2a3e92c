/* host_scene.cpp -- the build's restatement of the data-producing half of TracerBoy::LoadScene
 * (/root/reference/TracerBoy/TracerBoy.cpp): camera derivation :1243-1272, the per-shape loop
 * :1356-1835 (area lights :1526-1576, CreateMaterial :273-505, vertex/index packing :1595-1802,
 * HitGroupShaderRecord :1804-1817), non-area lights + environment map :1896-1934, and the
 * OutputSettings -> PerFrameConstants mapping :2808-2851.  D3D12 resource plumbing is replaced by
 * flat std::vectors that the context uploads once.
 */
#include "host_scene.h"

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <unordered_map>

namespace tbhost {

namespace {

inline TbFloat3 F3(const Vec3& v) { TbFloat3 r = {v.x, v.y, v.z}; return r; }
inline float ChannelAverage(const Vec3& v) { return (float)((v.x + v.y + v.z) / 3.0); }                     /* :118-121 */
inline float ConvertSpecularToIOR(float s) { return (float)((sqrt((double)s) + 1.0) / (1.0 - sqrt((double)s))); } /* :123-126 */

struct TextureAllocator { /* TracerBoy.cpp:177-251 */
    HostScene& scene;
    struct CachedImage { uint32_t index; bool normalized, hasAlpha; };
    std::unordered_map<std::string, CachedImage> imageCache;
    explicit TextureAllocator(HostScene& s) : scene(s) {}

    uint32_t CreateTexture(const PbrtTextureSP& tex, bool gammaCorrect, bool* hasAlpha)
    {
        if (!tex) return TB_INVALID_TEXTURE;
        TbTextureData td; memset(&td, 0, sizeof td);
        if (tex->kind == "imagemap") {
            td.TextureType = TB_TEXTURE_TYPE_IMAGE;
            bool normalized = false, alpha = false;
            auto it = imageCache.find(tex->fileName);
            if (it != imageCache.end()) { td.DescriptorHeapIndex = it->second.index; normalized = it->second.normalized; alpha = it->second.hasAlpha; }
            else {
                std::vector<TbFloat4> texels; uint32_t w = 0, h = 0; std::string err;
                if (!LoadImageRGBA32F(tex->fileName, texels, w, h, normalized, err, &alpha)) throw std::runtime_error(err);
                TbImageDesc d; d.width = w; d.height = h; d.texelOffset = scene.texelPool.size();
                scene.texelPool.insert(scene.texelPool.end(), texels.begin(), texels.end());
                td.DescriptorHeapIndex = (uint32_t)scene.images.size();
                scene.images.push_back(d);
                imageCache[tex->fileName] = CachedImage{td.DescriptorHeapIndex, normalized, alpha};
            }
            td.TextureFlags = 0;
            if (gammaCorrect && normalized) td.TextureFlags |= TB_TEXTURE_FLAG_NEEDS_GAMMA;
            if (hasAlpha) *hasAlpha = alpha; /* !scratchImage.IsAlphaAllOpaque(), TracerBoy.cpp:2229-2232 */
        } else if (tex->kind == "checkerboard") {
            td.TextureType = TB_TEXTURE_TYPE_CHECKER;
            td.UScale = tex->uscale; td.VScale = tex->vscale;
            td.CheckerColor1 = F3(tex->tex1); td.CheckerColor2 = F3(tex->tex2);
        } else if (tex->kind == "scale") {
            if ((tex->scaleTex1 && tex->scaleTex1->kind == "scale") || (tex->scaleTex2 && tex->scaleTex2->kind == "scale"))
                throw std::runtime_error("scale texture referring to a scale texture is not supported (TracerBoy.cpp:230-231)");
            bool a1 = false, a2 = false;
            td.TextureType = TB_TEXTURE_TYPE_SCALE;
            td.TextureIndex1 = CreateTexture(tex->scaleTex1, gammaCorrect, &a1);
            td.TextureIndex2 = CreateTexture(tex->scaleTex2, gammaCorrect, &a2);
            td.ScaleColor1 = F3(tex->scale1); td.ScaleColor2 = F3(tex->scale2);
            if (hasAlpha) *hasAlpha = a1 || a2;
        } else {
            throw std::runtime_error("unsupported texture type '" + tex->kind + "' (VERIFY(false) at TracerBoy.cpp:246)");
        }
        scene.textureData.push_back(td);
        return (uint32_t)scene.textureData.size() - 1;
    }
};

struct MaterialTracker { /* TracerBoy.h:130-156 */
    std::vector<TbMaterial>& list;
    std::unordered_map<const PbrtMaterial*, uint32_t> index;
    explicit MaterialTracker(std::vector<TbMaterial>& l) : list(l) {}
    bool Exists(const PbrtMaterial* m) const { return index.find(m) != index.end(); }
    uint32_t Add(const PbrtMaterial* m, const TbMaterial& v) { uint32_t i = (uint32_t)list.size(); index[m] = i; list.push_back(v); return i; }
};

/* TracerBoy.cpp:273-505 */
TbMaterial CreateMaterial(const PbrtMaterialSP& pm, const PbrtTextureSP* alphaTexture, const Vec3& emissive, MaterialTracker& tracker, TextureAllocator& tex)
{
    TbMaterial m; memset(&m, 0, sizeof m);
    m.IOR = 1.5f;
    m.albedoIndex = m.alphaIndex = m.normalMapIndex = m.emissiveIndex = m.specularMapIndex = TB_INVALID_TEXTURE;
    m.emissive = F3(emissive);
    m.Flags = ChannelAverage(emissive) > 0.0 ? TB_MAT_LIGHT : TB_MAT_DEFAULT;
    bool bHasAlpha = false;
    if (alphaTexture && *alphaTexture) { m.alphaIndex = tex.CreateTexture(*alphaTexture, false, nullptr); bHasAlpha = true; }
    auto albedoMap = [&](const PbrtTextureSP& t, bool gamma) { if (t) { bool a = false; m.albedoIndex = tex.CreateTexture(t, gamma, &a); bHasAlpha |= a; } };
    const std::string type = pm ? pm->type : std::string();
    if (!pm) {
        /* null material: defaults */
    } else if (type == "disney") { /* :309-330 */
        m.albedo = F3(pm->color);
        if (m.albedo.x > 0.7) { m.albedo.x = m.albedo.y = m.albedo.z = 0.2f; }
        m.roughness = pm->roughness; m.IOR = pm->eta;
        if (pm->metallic > 0.5) m.Flags |= TB_MAT_METALLIC;
        if (pm->specTrans > 0.001) { m.Flags |= TB_MAT_SUBSURFACE_SCATTER; m.absorption = {0, 0, 0}; m.roughness = 0; }
    } else if (type == "uber") { /* :331-364 */
        albedoMap(pm->map_kd, true);
        if (pm->map_normal) m.normalMapIndex = tex.CreateTexture(pm->map_normal, false, nullptr);
        if (pm->map_emissive) m.emissiveIndex = tex.CreateTexture(pm->map_emissive, false, nullptr);
        if (pm->map_specular) m.specularMapIndex = tex.CreateTexture(pm->map_specular, false, nullptr);
        m.albedo = F3(pm->kd);
        if (pm->uRoughness != pm->vRoughness) throw std::runtime_error("uber material with uroughness != vroughness (VERIFY at TracerBoy.cpp:354)");
        m.roughness = pm->uRoughness > 0.0 ? pm->uRoughness : pm->roughness;
        if (ChannelAverage(pm->opacity) < 1.0) {
            m.Flags |= TB_MAT_SUBSURFACE_SCATTER | TB_MAT_SINGLE_SIDED;
            m.IOR = pm->index; m.absorption = F3(pm->kt);
        }
    } else if (type == "mix") { /* :365-373 */
        uint32_t i0 = tracker.Add(pm->material0.get(), CreateMaterial(pm->material0, nullptr, emissive, tracker, tex));
        uint32_t i1 = tracker.Add(pm->material1.get(), CreateMaterial(pm->material1, nullptr, emissive, tracker, tex));
        m.Flags = TB_MAT_MIX;
        m.albedo = {(float)i0, (float)i1, ChannelAverage(pm->amount)};
    } else if (type == "mirror") { /* :374-380 */
        m.albedo = F3(pm->kr); m.SpecularCoef = 1.0f; m.roughness = 0.0f; m.Flags |= TB_MAT_METALLIC;
    } else if (type == "metal") { /* :381-390 */
        m.albedo = {1.0f, 1.0f, 1.0f}; m.IOR = ChannelAverage(pm->eta3);
        if (pm->uRoughness != pm->vRoughness) throw std::runtime_error("metal material with uroughness != vroughness (VERIFY at TracerBoy.cpp:387)");
        m.roughness = pm->uRoughness; m.Flags |= TB_MAT_METALLIC;
    } else if (type == "substrate") { /* :391-406 */
        albedoMap(pm->map_kd, false);
        m.albedo = F3(pm->kd);
        m.IOR = ConvertSpecularToIOR(ChannelAverage(pm->ks));
        m.SpecularCoef = ChannelAverage(pm->ks);
        if (pm->uRoughness != pm->vRoughness) throw std::runtime_error("substrate material with uroughness != vroughness (VERIFY at TracerBoy.cpp:404)");
        m.roughness = pm->uRoughness;
    } else if (type == "glass") { /* :407-416 */
        m.albedo = {0, 0, 0}; m.absorption = {0, 0, 0}; m.IOR = pm->index; m.Flags |= TB_MAT_SUBSURFACE_SCATTER;
    } else if (type == "fourier") { /* :417-422 */
        m.albedo = {0.6f, 0.6f, 0.6f}; m.roughness = 0.2f;
    } else if (type == "matte") { /* :423-435 */
        m.roughness = pm->sigma;
        albedoMap(pm->map_kd, false);
        m.albedo = F3(pm->kd); m.Flags |= TB_MAT_NO_SPECULAR;
    } else if (type == "plastic") { /* :436-453 */
        m.roughness = pm->roughness;
        albedoMap(pm->map_kd, false);
        m.albedo = F3(pm->kd);
        m.IOR = ConvertSpecularToIOR(ChannelAverage(pm->ks));
        m.SpecularCoef = ChannelAverage(pm->ks);
    } else if (type == "subsurface") { /* :454-476 HANDLE_FAILURE() */
        throw std::runtime_error("subsurface material is not supported (HANDLE_FAILURE at TracerBoy.cpp:456)");
    } else if (type == "translucent") { /* :477-491 */
        if (pm->map_kd) albedoMap(pm->map_kd, false);
        else { m.albedo = {0, 0, 0}; m.absorption = {0.001f, 0.001f, 0.001f}; m.Flags |= TB_MAT_SUBSURFACE_SCATTER; }
    } else { /* :492-497 */
        m.albedo = {(float)(153.0 / 255.0f), (float)(102.0f / 255.0), (float)(58.0f / 255.0f)}; m.roughness = 0.2f;
    }
    if (!bHasAlpha) m.Flags |= TB_MAT_NO_ALPHA;
    return m;
}

struct FlatShape { PbrtMeshSP mesh; Affine xfm; bool baked; };

void Flatten(const PbrtObject& obj, const Affine& xfm, bool isWorldLevel, bool allShapes, std::vector<FlatShape>& out, int depth)
{
    if (depth > 16) throw std::runtime_error("instance nesting too deep");
    if (out.size() > 0x00ffffffu) throw std::runtime_error("more than 2^24-1 instanced shapes (nested instancing multiplies)");
    if (!isWorldLevel) {
        size_t n = allShapes ? obj.shapes.size() : (obj.shapes.empty() ? 0 : 1);
        for (size_t i = 0; i < n; i++) { FlatShape f; f.mesh = obj.shapes[i]; f.xfm = xfm; f.baked = false; out.push_back(f); }
    }
    for (const PbrtInstance& inst : obj.instances) if (inst.object) Flatten(*inst.object, xfm * inst.xfm, false, allShapes, out, depth + 1);
}

/* every path into the conversion asks this first, so it is also where a mesh whose arrays do not fit together is refused (the
 * reference would read past its vertex buffer): an index beyond the vertices, per-vertex normals / texture coordinates that are fewer */
inline bool usable(const PbrtMesh& mesh)
{
    if (mesh.index.empty() || mesh.vertex.empty()) return false;
    const size_t nv = mesh.vertex.size();
    for (uint32_t i : mesh.index) if (i >= nv) throw std::runtime_error("mesh index " + std::to_string(i) + " is beyond its " + std::to_string(nv) +
        " vertices");
    if (!mesh.normal.empty() && mesh.normal.size() < nv) throw std::runtime_error("mesh has fewer normals than vertices");
    if (!mesh.texcoord.empty() && mesh.texcoord.size() < nv) throw std::runtime_error("mesh has fewer texture coordinates than vertices");
    return true;
}

/* area lights, one per triangle (TracerBoy.cpp:1526-1576); instanced emitters get their world positions */
void AppendAreaLights(HostScene& out, const PbrtMesh& mesh, const Affine& xf)
{
    const Vec3 emissive = mesh.areaLightL;
    const uint32_t numTris = (uint32_t)(mesh.index.size() / 3);
    for (uint32_t i = 0; i < numTris; i++) {
        TbLight light; memset(&light, 0, sizeof light);
        light.LightType = TB_LIGHT_TYPE_AREA;
        light.LightColor = F3(emissive);
        Vec3 p0 = xf * mesh.vertex[mesh.index[3 * i]], p1 = xf * mesh.vertex[mesh.index[3 * i + 1]], p2 = xf * mesh.vertex[mesh.index[3 * i + 2]];
        {
            Vec3 v0 = p1 - p0, v1 = p2 - p0;
            float v0Length = sqrtf(dot(v0, v0)), v1Length = sqrtf(dot(v1, v1));
            float angle = acosf(dot(v0, v1) / (v0Length * v1Length));
            light.SurfaceArea = (float)(v0Length * v1Length * sinf(angle) / 2.0);
        }
        light.P0 = F3(p0); light.P1 = F3(p1); light.P2 = F3(p2);
        if (!mesh.normal.empty()) {
            light.N0 = F3(xfmNormal(xf, mesh.normal[mesh.index[3 * i]]));
            light.N1 = F3(xfmNormal(xf, mesh.normal[mesh.index[3 * i + 1]]));
            light.N2 = F3(xfmNormal(xf, mesh.normal[mesh.index[3 * i + 2]]));
        } else {
            Vec3 n = normalize(cross(p1 - p0, p2 - p0));
            light.N0 = light.N1 = light.N2 = F3(n);
        }
        out.lights.push_back(light);
    }
}

/* material of a shape (TracerBoy.cpp:1578-1593) */
uint32_t MaterialOf(const PbrtMesh& mesh, MaterialTracker& tracker, TextureAllocator& textures)
{
    if (mesh.material && tracker.Exists(mesh.material.get())) return tracker.index[mesh.material.get()];
    auto a = mesh.textures.find("alpha");
    const PbrtTextureSP* alpha = a != mesh.textures.end() ? &a->second : nullptr;
    const Vec3 emissive = mesh.hasAreaLight ? mesh.areaLightL : Vec3(0.0f);
    TbMaterial created = CreateMaterial(mesh.material, alpha, emissive, tracker, textures);
    return tracker.Add(mesh.material.get(), created);
}

struct GeometrySlot { uint32_t vertexBufferOffset, indexBufferOffset, materialIndex; };

/* vertex / position / index buffers and the builder's per-triangle arrays (TracerBoy.cpp:1595-1802).  xf is baked into the
 * positions and attributes (identity for the geometry of an instanced structure: bBakeTransformIntoVertexBuffer, :1623-1624);
 * geometryIndexForTris is what a triangle reports as GeometryContributionToHitGroupIndex. */
GeometrySlot AppendGeometry(HostScene& out, const PbrtMesh& mesh, const Affine& xf, uint32_t geometryIndexForTris, uint32_t materialIndex)
{
    const uint32_t numTris = (uint32_t)(mesh.index.size() / 3);
    const uint32_t firstVertex = (uint32_t)(out.positions.size() / 3);
    const uint32_t vertexBufferOffset = (uint32_t)(out.vertexBuffer.size() * 4);
    const bool bNormalsProvided = !mesh.normal.empty();
    for (size_t v = 0; v < mesh.vertex.size(); v++) {
        Vec3 P = xf * mesh.vertex[v];
        Vec3 N(0, 1, 0), T(0, 0, 1);
        if (bNormalsProvided && v < mesh.normal.size()) N = normalize(xfmNormal(xf, mesh.normal[v]));
        if (v < mesh.tangents.size()) T = normalize(xfmNormal(xf, mesh.tangents[v]));
        Vec2 uv; if (v < mesh.texcoord.size()) uv = mesh.texcoord[v];
        const float vert[8] = {N.x, N.y, N.z, uv.x, uv.y, T.x, T.y, T.z};
        out.vertexBuffer.insert(out.vertexBuffer.end(), vert, vert + 8);
        out.positions.push_back(P.x); out.positions.push_back(P.y); out.positions.push_back(P.z);
    }
    while (out.indexBuffer.size() % 4) out.indexBuffer.push_back(0); /* D3D12_RAW_UAV_SRV_BYTE_ALIGNMENT (:1694) */
    const uint32_t indexBufferOffset = (uint32_t)(out.indexBuffer.size() * 4);
    const uint32_t geometryFlag = (out.materials[materialIndex].Flags & TB_MAT_NO_ALPHA) ? 1u : 0u; /* USE_ANYHIT 1, :1756-1760 */
    for (uint32_t i = 0; i < numTris; i++) {
        uint32_t ix = mesh.index[3 * i], iy = mesh.index[3 * i + 1], iz = mesh.index[3 * i + 2];
        if (ix >= mesh.vertex.size() || iy >= mesh.vertex.size() || iz >= mesh.vertex.size()) throw std::runtime_error("triangle index out of range");
        out.indexBuffer.push_back(ix); out.indexBuffer.push_back(iy); out.indexBuffer.push_back(iz);
        if (!bNormalsProvided) { /* flat normals :1710-1729 */
            Vec3 edge1 = mesh.vertex[iz] - mesh.vertex[ix], edge2 = mesh.vertex[iz] - mesh.vertex[iy];
            Vec3 normal = cross(edge1, edge2);
            if (dot(normal, normal) <= 0.0000000001f) normal = Vec3(0, 1, 0);
            else normal = normalize(xfmNormal(xf, normal));
            for (uint32_t vi : {ix, iy, iz}) { float* p = &out.vertexBuffer[(size_t)vertexBufferOffset / 4 + 8 * (size_t)vi]; p[0] = normal.x; p[1] = normal.y;
                p[2] = normal.z; }
        }
        out.triVertexIndex.push_back(firstVertex + ix); out.triVertexIndex.push_back(firstVertex + iy); out.triVertexIndex.push_back(firstVertex + iz);
        out.triGeometry.push_back(geometryIndexForTris); out.triPrimitive.push_back(i); out.triFlags.push_back(geometryFlag);
    }
    return GeometrySlot{vertexBufferOffset, indexBufferOffset, materialIndex};
}

void AppendHitGroup(HostScene& out, const GeometrySlot& g)
{
    TbHitGroupRecord rec; memset(&rec, 0, sizeof rec); /* :1804-1817 */
    rec.GeometryIndex = (uint32_t)out.hitGroups.size();
    rec.MaterialIndex = g.materialIndex;
    rec.VertexBufferIndex = 0; rec.VertexBufferOffset = g.vertexBufferOffset;
    rec.IndexBufferIndex = 0; rec.IndexBufferOffset = g.indexBufferOffset;
    out.hitGroups.push_back(rec);
}

/* rows of a 3x4 matrix as D3D12_RAYTRACING_INSTANCE_DESC::Transform stores them (ConvertAffine3f, TracerBoy.cpp:2048) */
void Rows(const Affine& a, float r[12])
{
    const float m[12] = {a.l.vx.x, a.l.vy.x, a.l.vz.x, a.p.x, a.l.vx.y, a.l.vy.y, a.l.vz.y, a.p.y, a.l.vx.z, a.l.vy.z, a.l.vz.z, a.p.z};
    memcpy(r, m, sizeof m);
}

float Determinant(const float t[12]) /* RayTracingHelper.hlsli:287-295 */
{
#define M(r, c) t[(r) * 4 + (c)]
    return M(0, 0) * M(1, 1) * M(2, 2) - M(0, 0) * M(2, 1) * M(1, 2) - M(1, 0) * M(0, 1) * M(2, 2) + M(1, 0) * M(2, 1) * M(0, 2) + M(2, 0) * M(0, 1) * M(1,
        2) - M(2, 0) * M(1, 1) * M(0, 2);
}

/* InverseAffineTransform, RayTracingHelper.hlsli:297-316 -- the fallback layer inverts ObjectToWorld on the GPU in fp32
 * (TopLevelLoadAABBs.hlsli:83-87); the same expressions term by term (products with the literal 0 / 1 of the implied fourth row kept) */
void InverseAffineTransform(const float t[12], float o[12])
{
    const float invDet = 1.0f / Determinant(t);
#define O(r, c) o[(r) * 4 + (c)]
    O(0, 0) = invDet * (M(1, 1) * (M(2, 2) * 1.0f - 0.0f * M(2, 3)) + M(2, 1) * (0.0f * M(1, 3) - M(1, 2) * 1.0f) + 0.0f * (M(1, 2) * M(2, 3) - M(2, 2) * M(1,
        3)));
    O(1, 0) = invDet * (M(1, 2) * (M(2, 0) * 1.0f - 0.0f * M(2, 3)) + M(2, 2) * (0.0f * M(1, 3) - M(1, 0) * 1.0f) + 0.0f * (M(1, 0) * M(2, 3) - M(2, 0) * M(1,
        3)));
    O(2, 0) = invDet * (M(1, 3) * (M(2, 0) * 0.0f - 0.0f * M(2, 1)) + M(2, 3) * (0.0f * M(1, 1) - M(1, 0) * 0.0f) + 1.0f * (M(1, 0) * M(2, 1) - M(2, 0) * M(1,
        1)));
    O(0, 1) = invDet * (M(2, 1) * (M(0, 2) * 1.0f - 0.0f * M(0, 3)) + 0.0f * (M(2, 2) * M(0, 3) - M(0, 2) * M(2, 3)) + M(0, 1) * (0.0f * M(2, 3) - M(2,
        2) * 1.0f));
    O(1, 1) = invDet * (M(2, 2) * (M(0, 0) * 1.0f - 0.0f * M(0, 3)) + 0.0f * (M(2, 0) * M(0, 3) - M(0, 0) * M(2, 3)) + M(0, 2) * (0.0f * M(2, 3) - M(2,
        0) * 1.0f));
    O(2, 1) = invDet * (M(2, 3) * (M(0, 0) * 0.0f - 0.0f * M(0, 1)) + 1.0f * (M(2, 0) * M(0, 1) - M(0, 0) * M(2, 1)) + M(0, 3) * (0.0f * M(2, 1) - M(2,
        0) * 0.0f));
    O(0, 2) = invDet * (0.0f * (M(0, 2) * M(1, 3) - M(1, 2) * M(0, 3)) + M(0, 1) * (M(1, 2) * 1.0f - 0.0f * M(1, 3)) + M(1, 1) * (0.0f * M(0, 3) - M(0,
        2) * 1.0f));
    O(1, 2) = invDet * (0.0f * (M(0, 0) * M(1, 3) - M(1, 0) * M(0, 3)) + M(0, 2) * (M(1, 0) * 1.0f - 0.0f * M(1, 3)) + M(1, 2) * (0.0f * M(0, 3) - M(0,
        0) * 1.0f));
    O(2, 2) = invDet * (1.0f * (M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1)) + M(0, 3) * (M(1, 0) * 0.0f - 0.0f * M(1, 1)) + M(1, 3) * (0.0f * M(0, 1) - M(0,
        0) * 0.0f));
    O(0, 3) = invDet * (M(0, 1) * (M(2, 2) * M(1, 3) - M(1, 2) * M(2, 3)) + M(1, 1) * (M(0, 2) * M(2, 3) - M(2, 2) * M(0, 3)) + M(2, 1) * (M(1, 2) * M(0,
        3) - M(0, 2) * M(1, 3)));
    O(1, 3) = invDet * (M(0, 2) * (M(2, 0) * M(1, 3) - M(1, 0) * M(2, 3)) + M(1, 2) * (M(0, 0) * M(2, 3) - M(2, 0) * M(0, 3)) + M(2, 2) * (M(1, 0) * M(0,
        3) - M(0, 0) * M(1, 3)));
    O(2, 3) = invDet * (M(0, 3) * (M(2, 0) * M(1, 1) - M(1, 0) * M(2, 1)) + M(1, 3) * (M(0, 0) * M(2, 1) - M(2, 0) * M(0, 1)) + M(2, 3) * (M(1, 0) * M(0,
        1) - M(0, 0) * M(1, 1)));
#undef O
#undef M
}

} // namespace

void ConvertScene(const PbrtScene& in, HostScene& out, const ConvertOptions& opt)
{
    out = HostScene();
    out.filmWidth = in.filmWidth; out.filmHeight = in.filmHeight;
    if (!in.hasCamera) throw std::runtime_error("scene has no camera (assert at TracerBoy.cpp:1243)");
    { /* camera: TracerBoy.cpp:1246-1272 */
        Vec3 CameraPosition(0.f), CameraView(0.0f, 0.0f, 1.0f), CameraRight(1.0f, 0.0f, 0.0f), CameraUp(0.0f, 1.0f, 0.0f);
        CameraPosition = in.cameraFrame * CameraPosition;
        CameraView = normalize(xfmVector(in.cameraFrame, CameraView));
        CameraRight = normalize(xfmVector(in.cameraFrame, CameraRight));
        CameraUp = xfmVector(in.cameraFrame, CameraUp);
        out.camera.LensHeight = (float)(2.0 * sqrtf(dot(CameraUp, CameraUp)));
        CameraUp = normalize(CameraUp);
        float FOVAngle = (float)(in.fov * M_PI / 180.0);
        out.camera.FocalDistance = (float)((out.camera.LensHeight / 2.0) / tan(FOVAngle / 2.0));
        CameraPosition = CameraPosition + (out.camera.FocalDistance + 0.01f) * CameraView;
        Vec3 look = CameraPosition + CameraView;
        float* dst[4] = {out.camera.Position, out.camera.LookAt, out.camera.Right, out.camera.Up};
        const Vec3* src[4] = {&CameraPosition, &look, &CameraRight, &CameraUp};
        for (int i = 0; i < 4; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    }

    TextureAllocator textures(out);
    MaterialTracker tracker(out.materials);
    Vec3 smin(3.402823466e+38f), smax(-3.402823466e+38f);
    auto growWorld = [&](const PbrtMesh& mesh, const Affine& xf) {
        for (const Vec3& v : mesh.vertex) {
            const Vec3 P = xf * v;
            smin = Vec3(fminf(smin.x, P.x), fminf(smin.y, P.y), fminf(smin.z, P.z));
            smax = Vec3(fmaxf(smax.x, P.x), fmaxf(smax.y, P.y), fmaxf(smax.z, P.z));
        }
    };

    const bool twoLevel = !opt.flattenInstances && !in.world.instances.empty();
    if (!twoLevel) {
        /* world shapes first, then instances (TracerBoy.cpp:1356-1376) */
        std::vector<FlatShape> shapes;
        for (const PbrtMeshSP& m : in.world.shapes) { FlatShape f; f.mesh = m; f.baked = true; shapes.push_back(f); }
        if (opt.flattenInstances) Flatten(in.world, Affine(), true, true, shapes, 0);
        for (const FlatShape& fs : shapes) {
            const PbrtMesh& mesh = *fs.mesh;
            if (!usable(mesh)) continue;
            const Affine xf = fs.baked ? Affine() : fs.xfm;
            if (mesh.hasAreaLight) AppendAreaLights(out, mesh, xf);
            const uint32_t materialIndex = MaterialOf(mesh, tracker, textures);
            const GeometrySlot g = AppendGeometry(out, mesh, xf, (uint32_t)out.hitGroups.size(), materialIndex);
            growWorld(mesh, xf);
            AppendHitGroup(out, g);
        }
    } else {
        /* Two-level: the structure the reference hands its hardware path.  Structure 0 = every world-level shape under an identity
         * instance ("global BLAS", TracerBoy.cpp:1363-1368); every ObjectInstance -- nested ones with their transforms composed --
         * becomes an instance of its object's structure (all of the object's shapes: the reference takes shapes[0], :1374).
         * Geometry is stored once per structure, in object space; hit-group records exist once per (instance, geometry) and an
         * instance's first record is its InstanceContributionToHitGroupIndex.  Vertex attributes are NOT transformed for
         * instanced geometry (bBakeTransformIntoVertexBuffer is false, :1623-1624): shading normals of instances are the
         * object's, exactly as in the reference. */
        struct BlasDef { const PbrtObject* key; std::vector<PbrtMeshSP> meshes; std::vector<GeometrySlot> slots; };
        struct InstDef { uint32_t blas; Affine xfm; };
        std::vector<BlasDef> defs; std::vector<InstDef> insts;
        { BlasDef world; world.key = nullptr; for (const PbrtMeshSP& m : in.world.shapes) if (m && usable(*m)) world.meshes.push_back(m);
          if (!world.meshes.empty()) { defs.push_back(world); insts.push_back(InstDef{0, Affine()}); } }
        std::unordered_map<const PbrtObject*, uint32_t> blasOf;
        struct Walker {
            std::vector<BlasDef>& defs; std::vector<InstDef>& insts; std::unordered_map<const PbrtObject*, uint32_t>& blasOf; uint32_t visited = 0;
            void visit(const PbrtObject& obj, const Affine& xfm, int depth)
            {
                if (depth > 16) throw std::runtime_error("instance nesting too deep");
                for (const PbrtInstance& inst : obj.instances) {
                    if (!inst.object) continue;
                    const Affine X = xfm * inst.xfm;
                    const PbrtObject* o = inst.object.get();
                    auto it = blasOf.find(o);
                    if (it == blasOf.end()) {
                        BlasDef d; d.key = o; for (const PbrtMeshSP& m : o->shapes) if (m && usable(*m)) d.meshes.push_back(m);
                        uint32_t idx = 0xffffffffu;
                        if (!d.meshes.empty()) { idx = (uint32_t)defs.size(); defs.push_back(d); }
                        it = blasOf.emplace(o, idx).first;
                    }
                    if (it->second != 0xffffffffu) insts.push_back(InstDef{it->second, X});
                    if (insts.size() > 0x00ffffffu || ++visited > 0x04000000u)
                        throw std::runtime_error("more than 2^24-1 instances (nested instancing multiplies)");
                    visit(*o, X, depth + 1);
                }
            }
        } walker{defs, insts, blasOf};
        walker.visit(in.world, Affine(), 0);
        if (insts.size() > 0x00ffffffu) throw std::runtime_error("more than 2^24-1 instances");
        for (BlasDef& d : defs) {
            HostScene::Blas b; b.firstTri = (uint32_t)out.triGeometry.size();
            for (const PbrtMeshSP& m : d.meshes) d.slots.push_back(AppendGeometry(out, *m, Affine(), (uint32_t)d.slots.size(), MaterialOf(*m, tracker,
                textures)));
            b.numTris = (uint32_t)out.triGeometry.size() - b.firstTri;
            out.blas.push_back(b);
        }
        for (const InstDef& i : insts) {
            HostScene::Instance hi; hi.blas = i.blas; hi.hitGroupBase = (uint32_t)out.hitGroups.size();
            Rows(i.xfm, hi.objectToWorld); InverseAffineTransform(hi.objectToWorld, hi.worldToObject);
            const BlasDef& d = defs[i.blas];
            for (size_t g = 0; g < d.slots.size(); g++) {
                AppendHitGroup(out, d.slots[g]);
                if (d.meshes[g]->hasAreaLight) AppendAreaLights(out, *d.meshes[g], i.xfm);
                growWorld(*d.meshes[g], i.xfm);
            }
            out.instances.push_back(hi);
        }
    }
    if (out.triGeometry.empty()) throw std::runtime_error("scene has no triangles");
    out.sceneMin[0] = smin.x; out.sceneMin[1] = smin.y; out.sceneMin[2] = smin.z;
    out.sceneMax[0] = smax.x; out.sceneMax[1] = smax.y; out.sceneMax[2] = smax.z;

    /* non-area lights + environment (:1896-1934) */
    Mat3 envL; Vec3 envScale(1, 1, 1);
    for (const PbrtLight& l : in.lights) {
        if (l.kind == PbrtLight::Infinite) {
            if (!l.mapFile.empty()) {
                bool normalized; std::string err;
                if (!LoadImageRGBA32F(l.mapFile, out.envMap, out.envWidth, out.envHeight, normalized, err)) throw std::runtime_error(err);
            }
            envL = l.transform.l; envScale = l.scale;
        } else {
            TbLight light; memset(&light, 0, sizeof light);
            light.LightColor = F3(l.L); light.LightType = TB_LIGHT_TYPE_DIRECTIONAL;
            light.Direction = F3(normalize(l.to - l.from));
            out.lights.push_back(light);
        }
    }
    /* UpdateConfigConstants (:3372-3384) */
    memset(&out.config, 0, sizeof out.config);
    out.config.CameraLensHeight = out.camera.LensHeight;
    out.config.FlipTextureUVs = opt.flipTextureUVs ? 1u : 0u; /* m_flipTextureUVs: true for .pbrt / .pbf scenes (TracerBoy.cpp:1208,1222,3376) */
    out.config.EnvMapTransformVx = {envL.vx.x, envL.vx.y, envL.vx.z, 0.0f};
    out.config.EnvMapTransformVy = {envL.vy.x, envL.vy.y, envL.vy.z, 0.0f};
    out.config.EnvMapTransformVz = {envL.vz.x, envL.vz.y, envL.vz.z, 0.0f};
    out.config.EnvironmentMapColorScale = F3(envScale);
}

void DefaultOutputSettings(tb_output_settings& s) /* TracerBoy.h:290-360 */
{
    memset(&s, 0, sizeof s);
    s.OutputType = TB_OUTPUT_TYPE_LIT;
    s.EnableNormalMaps = 0;
    s.RenderModeRealTime = 0;
    s.DebugValue = 1.0f; s.DebugValue2 = 1.0f;
    s.DOFFocalDistance = 0.0f; s.ApertureWidth = 0.075f; s.FilterType = TB_FILTER_TYPE_BOX; s.FilterWidth = 1.0f;
    s.FireflyClampValue = 0.0f; s.MaxZ = 10000.0f;
    s.ConvergencePercentage = 0.001f;
    s.EnableBlueNoise = 1; s.EnableNextEventEstimation = 1; s.EnableSamplingImportanceResampling = 0;
    s.MaxBounces = 6; s.SampleTarget = 256;
}

void MakeFrameConstants(const HostScene& scene, const tb_camera& cam, const tb_output_settings& s, uint32_t frame, float timeSeed,
                        uint32_t selX, uint32_t selY, TbPerFrameConstants& c) /* TracerBoy.cpp:2808-2851 */
{
    memset(&c, 0, sizeof c);
    c.CameraPosition = {cam.Position[0], cam.Position[1], cam.Position[2]};
    c.CameraLookAt = {cam.LookAt[0], cam.LookAt[1], cam.LookAt[2]};
    c.CameraRight = {cam.Right[0], cam.Right[1], cam.Right[2]};
    c.CameraUp = {cam.Up[0], cam.Up[1], cam.Up[2]};
    c.LightCount = (uint32_t)scene.lights.size();
    c.Time = timeSeed;
    c.EnableNormalMaps = s.EnableNormalMaps;
    c.FocalDistance = cam.FocalDistance;
    c.DOFFocusDistance = s.DOFFocalDistance;
    c.DOFApertureWidth = s.ApertureWidth;
    c.InvalidateHistory = frame == 0;
    c.FireflyClampValue = s.FireflyClampValue;
    c.GlobalFrameCount = frame;
    c.MinConvergence = s.ConvergencePercentage;
    c.UseBlueNoise = s.EnableBlueNoise;
    c.EnableNextEventEstimation = s.EnableNextEventEstimation;
    c.MaxBounces = (uint32_t)(s.MaxBounces < 0 ? 0 : s.MaxBounces);
    c.EnableSamplingImportanceResampling = s.EnableSamplingImportanceResampling;
    c.IsRealTime = s.RenderModeRealTime;
    c.OutputMode = s.OutputType;
    c.FilterWidth = s.FilterWidth;
    c.FilterType = s.FilterType;
    c.DebugValue = s.DebugValue; c.DebugValue2 = s.DebugValue2;
    c.SelectedPixelX = selX; c.SelectedPixelY = selY;
    c.FixedPixelOffset = {-1.0f, -1.0f};
    c.MaxZ = s.MaxZ;
}

} // namespace tbhost
