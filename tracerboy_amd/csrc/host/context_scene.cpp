/* context_scene.cpp -- the device half of LoadScene (/root/reference/TracerBoy/TracerBoy.cpp:1065-2161): BVH builds on the GPU, storage
 * order of the nodes, the compact node layout, uploads in the kernels' 16-B aligned device forms, the whole-scene-in-LDS image. */
#include "context_internal.h"

namespace tbctx {

void releaseScene(tb_context* c)
{
    for (DevBuf& b : c->sceneBufs) b.release();
    c->sceneBufs.clear();
    memset(&c->ds, 0, sizeof c->ds);
}

uint32_t sceneFeatureMask(const HostScene& s)
{
    uint32_t f = 0;
    if (!s.envMap.empty()) f |= PT_FEAT_ENV;
    for (const TbMaterial& m : s.materials) {
        if ((m.Flags & TB_MAT_NO_SPECULAR) == 0 && (m.Flags & TB_MAT_MIX) == 0) f |= PT_FEAT_SPECULAR;
        if (m.albedoIndex != TB_INVALID_TEXTURE || m.emissiveIndex != TB_INVALID_TEXTURE || m.specularMapIndex != TB_INVALID_TEXTURE ||
            m.normalMapIndex != TB_INVALID_TEXTURE) f |= PT_FEAT_TEXTURES | PT_FEAT_SPECULAR;
        if (m.Flags & TB_MAT_SUBSURFACE_SCATTER) f |= PT_FEAT_SSS;
        if (m.Flags & TB_MAT_MIX) f |= PT_FEAT_MIX;
    }
    for (const TbLight& l : s.lights) if (l.LightType != TB_LIGHT_TYPE_AREA) f |= PT_FEAT_EXT;
    /* two-level scenes: renderImpl picks a kernel copy that walks instances (the frame-group kernels of the higher-occupancy copies,
     * else the full feature set) */
    return f;
}

uint32_t settingsFeatureMask(const tb_context* c, const tb_output_settings& s, bool aov)
{
    bool ext = s.EnableSamplingImportanceResampling || s.DOFFocalDistance > 0.0f || s.FilterType != TB_FILTER_TYPE_BOX ||
               s.FireflyClampValue != 0.0f || s.RenderModeRealTime || s.OutputType == TB_OUTPUT_TYPE_HEATMAP || aov ||
               (c->selX != 0xffffffffu) || c->ds.alphaTest != 0;
    return ext ? PT_FEAT_EXT : 0u;
}

/* Storage order of the layout-B nodes (results do not depend on it).  order 0: breadth-first, the top of the tree is one
 * contiguous prefix; order 1: depth-first pre-order, a node's left child follows it (same 128-B line every other step of
 * a descent); order 2: breadth-first for the top `topLevels` levels, depth-first below (cached top + local subtrees);
 * order 3: blocks of `topLevels` levels stored breadth-first, the blocks themselves depth-first (van Emde Boas style: a
 * descent of `topLevels` steps stays inside one contiguous block); order 4: depth-first by SIBLING PAIRS -- the two inner children
 * of a node lie side by side in one aligned 128-B line (a dummy node pads where needed), so fetching the near child brings the
 * far child's record along for when it is popped; order 5: order 4 below a breadth-first top of `topLevels` levels. */
void reorderNodes(HostScene& s, int order_, uint32_t topLevels)
{
    const uint32_t n = (uint32_t)s.nodesB.size();
    if (s.rootRefB & TB_BVH_LEAF_FLAG) return;
    constexpr uint32_t PAD = 0xffffffffu;
    std::vector<uint32_t> order; order.reserve(n + n / 4);
    std::vector<uint32_t> newIndex(n, 0);
    auto inner = [](uint32_t ref) { return !(ref & TB_BVH_LEAF_FLAG); };
    auto pairDfs = [&](uint32_t root) { /* `root` itself is already placed */
        std::vector<uint32_t> st; st.push_back(root);
        while (!st.empty()) {
            const uint32_t x = st.back(); st.pop_back();
            const TbNodeB& nd = s.nodesB[x];
            const bool li = inner(nd.left), ri = inner(nd.right);
            if (li && ri && (order.size() & 1u)) order.push_back(PAD);
            if (li) order.push_back(nd.left);
            if (ri) order.push_back(nd.right);
            if (ri) st.push_back(nd.right);
            if (li) st.push_back(nd.left);
        }
    };
    auto dfs = [&](uint32_t root) {
        std::vector<uint32_t> st; st.push_back(root);
        while (!st.empty()) {
            uint32_t x = st.back(); st.pop_back(); order.push_back(x);
            const TbNodeB& nd = s.nodesB[x];
            if (inner(nd.right)) st.push_back(nd.right);
            if (inner(nd.left)) st.push_back(nd.left);
        }
    };
    if (order_ == 1) dfs(s.rootRefB);
    else if (order_ == 4) { order.push_back(s.rootRefB); pairDfs(s.rootRefB); }
    else if (order_ == 3) {
        const uint32_t h = topLevels ? topLevels : 2;
        std::vector<uint32_t> blocks; blocks.push_back(s.rootRefB);
        std::vector<uint32_t> level, next, below;
        while (!blocks.empty()) {
            level.assign(1, blocks.back()); blocks.pop_back(); below.clear();
            for (uint32_t d = 0; d < h && !level.empty(); d++) {
                next.clear();
                for (uint32_t x : level) { order.push_back(x); const TbNodeB& nd = s.nodesB[x]; if (inner(nd.left)) next.push_back(nd.left);
                    if (inner(nd.right)) next.push_back(nd.right); }
                level.swap(next);
            }
            for (size_t i = level.size(); i-- > 0;) blocks.push_back(level[i]); /* leftmost block below comes next */
        }
    } else {
        std::vector<uint32_t> level; level.push_back(s.rootRefB);
        uint32_t depth = 0;
        while (!level.empty() && (order_ == 0 || depth < topLevels)) { /* orders 0, 2, 5 */
            std::vector<uint32_t> next;
            for (uint32_t x : level) { order.push_back(x); const TbNodeB& nd = s.nodesB[x]; if (inner(nd.left)) next.push_back(nd.left);
                if (inner(nd.right)) next.push_back(nd.right); }
            level.swap(next); depth++;
        }
        if (order_ == 5) { if ((order.size() & 1u) && !level.empty()) order.push_back(PAD); for (uint32_t x : level) order.push_back(x);
            for (uint32_t x : level) pairDfs(x); }
        else for (uint32_t x : level) dfs(x); /* order 2: the subtrees hanging below the breadth-first top */
    }
    for (uint32_t i = 0; i < (uint32_t)order.size(); i++) if (order[i] != PAD) newIndex[order[i]] = i;
    std::vector<TbNodeB> out(order.size());
    for (uint32_t i = 0; i < (uint32_t)order.size(); i++) {
        if (order[i] == PAD) { memset(&out[i], 0, sizeof(TbNodeB)); out[i].left = out[i].right = TB_BVH_LEAF_FLAG; continue; }
        TbNodeB nd = s.nodesB[order[i]];
        if (inner(nd.left)) nd.left = newIndex[nd.left];
        if (inner(nd.right)) nd.right = newIndex[nd.right];
        out[i] = nd;
    }
    s.nodesB.swap(out);
    s.rootRefB = 0;
}

/* option "bvh_builder" = 2 / 4: the LBVH of builder 0 / the LBVH + treelet passes of builder 3 constructed on the GPU
 * (bvh_kernels.hip); the host copies are filled
 * from the device result so that every host-side consumer (oracle view, layout queries) sees the same tree */
void BuildBvhGpu(tb_context* c, HostScene& s, uint32_t treeletPasses)
{
    const uint64_t N64 = s.triGeometry.size();
    if (N64 == 0) throw std::runtime_error("BuildBvh: no triangles");
    if (N64 > 0x00ffffffull) throw std::runtime_error("BuildBvh: more than 2^24-1 triangles does not fit the 24-bit node indices of the reference layout");
    if (s.blueNoise0.empty()) LoadBlueNoiseTiles(s);
    const uint32_t N = (uint32_t)N64;
    const uint64_t nodes = 2ull * N - 1, total = 16 + 32 * nodes + 52ull * N;
    if (total > 0xffffffffull) throw std::runtime_error("BuildBvh: BVH image exceeds 4 GiB");
    DevBuf dPos, dIdx, dGeo, dPrim, dFlag, dA, dNodes, dTris, dScratch, dHeight;
    auto up = [&](DevBuf& b, const void* p, size_t bytes) { ensure(b, bytes); HIP_TRY(hipMemcpyAsync(b.p, p, bytes, hipMemcpyHostToDevice, c->stream)); };
    try {
        up(dPos, s.positions.data(), s.positions.size() * 4); up(dIdx, s.triVertexIndex.data(), s.triVertexIndex.size() * 4);
        up(dGeo, s.triGeometry.data(), 4ull * N); up(dPrim, s.triPrimitive.data(), 4ull * N); up(dFlag, s.triFlags.data(), 4ull * N);
        const size_t nB = N > 1 ? N - 1 : 1, scratchBytes = bvh_gpu_scratch_bytes(N);
        ensure(dA, total); ensure(dNodes, nB * sizeof(TbNodeB)); ensure(dTris, (size_t)N * sizeof(TbTriB)); ensure(dScratch, scratchBytes); ensure(dHeight, 4);
        HIP_TRY(hipMemsetAsync(dNodes.p, 0, nB * sizeof(TbNodeB), c->stream));
        HIP_TRY(bvh_gpu_build(c->stream, (const float*)dPos.p, (const uint32_t*)dIdx.p, (const uint32_t*)dGeo.p, (const uint32_t*)dPrim.p,
            (const uint32_t*)dFlag.p, N,
                              treeletPasses, (uint8_t*)dScratch.p, scratchBytes, (uint8_t*)dA.p, (TbNodeB*)dNodes.p, (TbTriB*)dTris.p, (uint32_t*)dHeight.p));
        s.bvhA.resize((size_t)total); s.nodesB.resize(nB); s.trisB.resize(N);
        HIP_TRY(hipMemcpy(s.bvhA.data(), dA.p, total, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(s.nodesB.data(), dNodes.p, nB * sizeof(TbNodeB), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(s.trisB.data(), dTris.p, (size_t)N * sizeof(TbTriB), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&s.bvhMaxDepth, dHeight.p, 4, hipMemcpyDeviceToHost));
        s.rootRefB = N == 1 ? TB_BVH_LEAF_FLAG : 0u;
    } catch (...) {
        for (DevBuf* b : {&dPos, &dIdx, &dGeo, &dPrim, &dFlag, &dA, &dNodes, &dTris, &dScratch, &dHeight}) b->release();
        throw;
    }
    for (DevBuf* b : {&dPos, &dIdx, &dGeo, &dPrim, &dFlag, &dA, &dNodes, &dTris, &dScratch, &dHeight}) b->release();
}

/* Layout C (tb_abi.h TbNodeC): the layout-B nodes, same order, boxes rounded outward onto a 16-bit grid over the root box.
 * A quantised box [c - h, c + h] contains its layout-B box with at least an eighth of a cell to spare on every side, which is
 * what covers the different rounding of the two slab computations (the kernel evaluates q * (cell * inv) - (o - origin) * inv
 * where layout B evaluates c * inv - o * inv: errors of a few ulp of |c * inv| + |o * inv|, i.e. below 2^-6 cells while ray origin
 * and box lie within a few scene extents of each other). */
void buildCompactNodes(const HostScene& s, std::vector<TbNodeC>& out, TbQuantFrame& q, uint32_t nodeUnits)
{
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    auto grow = [&](const float* cc, const float* hh, int k) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)cc[a * 2 + k] - hh[a * 2 + k]);
        hi[a] = std::max(hi[a], (double)cc[a * 2 + k] + hh[a * 2 + k]); } };
    auto boxOf = [](const TbNodeB& n, float* cc, float* hh) { /* [axis * 2 + child] */
        cc[0] = n.cx[0]; cc[1] = n.cx[1]; cc[2] = n.cy[0]; cc[3] = n.cy[1]; cc[4] = n.cz[0]; cc[5] = n.cz[1];
        hh[0] = n.hx[0]; hh[1] = n.hx[1]; hh[2] = n.hy[0]; hh[3] = n.hy[1]; hh[4] = n.hz[0]; hh[5] = n.hz[1];
    };
    auto isPad = [](const TbNodeB& n) { return n.left == TB_BVH_LEAF_FLAG && n.right == TB_BVH_LEAF_FLAG && n.hx[0] == 0.0f && n.hx[1] == 0.0f && n.cx[0] ==
        0.0f && n.cx[1] == 0.0f; };
    for (const TbNodeB& n : s.nodesB) { if (isPad(n)) continue; float cc[6], hh[6]; boxOf(n, cc, hh); grow(cc, hh, 0); grow(cc, hh, 1); }
    double ext = 0; for (int a = 0; a < 3; a++) ext = std::max(ext, hi[a] - lo[a]);
    if (!(ext > 0)) ext = 1.0;
    for (int a = 0; a < 3; a++) {
        const double e = std::max(hi[a] - lo[a], ext * 1e-6); /* flat scenes: keep the cell finite */
        q.cell[a] = (float)(e * 1.004 / 65535.0);
        q.origin[a] = (float)(lo[a] - 0.002 * e);
        /* the origin is an fp32 number: step it DOWN until two cells of margin are really there (a scene far from the coordinate origin has
         * ulps larger than the margin; if they are larger than the grid can absorb, the box test below refuses the layout and the render
         * stays with layout B) */
        for (int guard = 0; guard < 64 && !((double)q.origin[a] + 2.0 * (double)q.cell[a] <= lo[a]); guard++) q.origin[a] = std::nextafter(q.origin[a],
            -std::numeric_limits<float>::infinity());
    }
    out.assign(s.nodesB.size(), TbNodeC{});
    auto ref = [nodeUnits](uint32_t r) { return (r & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((r & ~TB_BVH_LEAF_FLAG) * 3u)) : r * nodeUnits; };
    for (size_t i = 0; i < s.nodesB.size(); i++) {
        const TbNodeB& n = s.nodesB[i]; TbNodeC& o = out[i];
        o.left = ref(n.left); o.right = ref(n.right);
        if (isPad(n)) continue;
        float cc[6], hh[6]; boxOf(n, cc, hh);
        for (int a = 0; a < 3; a++) for (int k = 0; k < 2; k++) {
            const double bl = ((double)cc[a * 2 + k] - hh[a * 2 + k] - q.origin[a]) / q.cell[a],
                bh = ((double)cc[a * 2 + k] + hh[a * 2 + k] - q.origin[a]) / q.cell[a];
            long ql = (long)std::floor(bl - 0.125), qh = (long)std::ceil(bh + 0.125);
            if (ql < 0 || qh > 65535 || !(bl == bl) || !(bh == bh)) throw std::runtime_error("compact nodes: a box lies outside the quantisation grid");
            const long cq = (ql + qh) >> 1, hq = qh - cq; /* cq - hq <= ql and cq + hq == qh */
            o.c[a][k] = (uint16_t)cq; o.h[a][k] = (uint16_t)hq;
        }
    }
}

/* Layout C for the loaded scene, on first demand (option "node_layout" = 1 at a render or a trace): +32 B per node of device memory and a
 * host pass, paid only by who asks.  A scene the 16-bit grid cannot hold (coordinates far from the origin relative to the extent, boxes
 * with NaN or infinite bounds) keeps layout B: the failure is remembered, not thrown (it used to abort tb_load_scene). */
void ensureCompactNodes(tb_context* c)
{
    if (c->compactTried || c->ds.nodesC) return;
    c->compactTried = true;
    const HostScene& s = c->scene;
    if (!s.instances.empty() || (s.rootRefB & TB_BVH_LEAF_FLAG)) return;
    try {
        std::vector<TbNodeC> compact;
        buildCompactNodes(s, compact, c->ds.quant, 2u);
        c->ds.nodesC = upload(c, compact);
    } catch (const std::exception&) { c->ds.nodesC = nullptr; }
}

/* the top level of a two-level scene on the GPU (bvh_gpu_build_tlas): same bytes as bvh_build.cpp BuildTlas / the oracle's tbo_build_tlas */
void BuildTlasGpu(tb_context* c, HostScene& s, const std::vector<float>& blasBoxes, std::vector<TbNodeB>& top, uint32_t& rootRef, uint32_t& depth)
{
    const uint32_t M = (uint32_t)s.instances.size();
    std::vector<float> o2w(12ull * M), w2o(12ull * M); std::vector<uint32_t> blas(M), base(M);
    for (uint32_t i = 0; i < M; i++) { memcpy(&o2w[12ull * i], s.instances[i].objectToWorld, 48); memcpy(&w2o[12ull * i], s.instances[i].worldToObject, 48);
        blas[i] = s.instances[i].blas; base[i] = s.instances[i].hitGroupBase; }
    const size_t total = 16 + 32 * (2ull * M - 1) + 116ull * M, scratchBytes = bvh_gpu_tlas_scratch_bytes(M);
    DevBuf dO, dW, dB, dH, dBox, dScratch, dA, dTop, dWords;
    auto up = [&](DevBuf& b, const void* p, size_t bytes) { ensure(b, bytes); HIP_TRY(hipMemcpyAsync(b.p, p, bytes, hipMemcpyHostToDevice, c->stream)); };
    try {
        up(dO, o2w.data(), o2w.size() * 4); up(dW, w2o.data(), w2o.size() * 4); up(dB, blas.data(), 4ull * M); up(dH, base.data(), 4ull * M);
            up(dBox, blasBoxes.data(), blasBoxes.size() * 4);
        ensure(dScratch, scratchBytes); ensure(dA, total); ensure(dTop, std::max<size_t>(1, M - 1) * sizeof(TbNodeB)); ensure(dWords, 8);
        HIP_TRY(hipMemsetAsync(dA.p, 0, total, c->stream));
        HIP_TRY(bvh_gpu_build_tlas(c->stream, M, (const float*)dO.p, (const float*)dW.p, (const uint32_t*)dB.p, (const uint32_t*)dH.p, (const float*)dBox.p,
            (uint8_t*)dScratch.p, scratchBytes,
                                   (uint8_t*)dA.p, (TbNodeB*)dTop.p, (uint32_t*)dWords.p, (uint32_t*)dWords.p + 1));
        s.tlasA.resize(total); top.assign(M > 1 ? M - 1 : 0, TbNodeB{});
        uint32_t words[2];
        HIP_TRY(hipMemcpy(s.tlasA.data(), dA.p, total, hipMemcpyDeviceToHost));
        if (M > 1) HIP_TRY(hipMemcpy(top.data(), dTop.p, (size_t)(M - 1) * sizeof(TbNodeB), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(words, dWords.p, 8, hipMemcpyDeviceToHost));
        rootRef = words[0]; depth = words[1];
    } catch (...) {
        for (DevBuf* b : {&dO, &dW, &dB, &dH, &dBox, &dScratch, &dA, &dTop, &dWords}) b->release();
        throw;
    }
    for (DevBuf* b : {&dO, &dW, &dB, &dH, &dBox, &dScratch, &dA, &dTop, &dWords}) b->release();
}

/* TbDeviceScene::textureUse of the scene as it stands (finalizeScene, tb_set_material) */
uint32_t sceneTextureUse(const tb_context* c)
{
    auto it = c->options.find("texture_use_hint");
    if (it != c->options.end() && it->second == 0) return 3u; /* 0: fetch whole vertices whatever the materials say */
    uint32_t use = 0;
    for (const TbMaterial& m : c->scene.materials) {
        if (m.albedoIndex != TB_INVALID_TEXTURE || m.emissiveIndex != TB_INVALID_TEXTURE || m.specularMapIndex != TB_INVALID_TEXTURE ||
            m.alphaIndex != TB_INVALID_TEXTURE) use |= 1u;
        if (m.normalMapIndex != TB_INVALID_TEXTURE) use |= 2u;
    }
    return use;
}

void finalizeScene(tb_context* c, bool build)
{
    HostScene& s = c->scene;
    c->sceneGeneration++;
    {   /* share of the triangles whose material sends a path on an interior walk (the pre-pass policy in renderImpl) */
        uint64_t walks = 0;
        if (s.instances.empty())
            for (uint32_t g : s.triGeometry) { if (g < s.hitGroups.size()) { const uint32_t m = s.hitGroups[g].MaterialIndex;
                if (m < s.materials.size() && (s.materials[m].Flags & TB_MAT_SUBSURFACE_SCATTER)) walks++; } }
        c->interiorWalkTriangleShare = s.triGeometry.empty() ? 0.0f : (float)((double)walks / (double)s.triGeometry.size());
    }
    {   /* compact hit records (pt_scene.h): the widths of the two index fields */
        uint32_t maxPrim = 0;
        for (uint32_t q : s.triPrimitive) maxPrim = std::max(maxPrim, q);
        auto bitsFor = [](uint64_t maxValue) { uint32_t b = 1; while (b < 32 && (maxValue >> b)) b++; return b; };
        c->hitPrimBits = bitsFor(maxPrim); c->hitGeomBits = bitsFor(s.hitGroups.empty() ? 0 : s.hitGroups.size() - 1);
    }
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    const int64_t builder = opt("bvh_builder", 0);
    const bool twoLevel = !s.instances.empty();
    s.reinsertionPasses = (int)opt("reinsertion_passes", -1); s.reinsertionShare = (int)opt("reinsertion_share", 100);
    s.presplitPercent = (int)std::max<int64_t>(0, std::min<int64_t>(400, opt("presplit", 0)));
    if (build) {
    /* every bottom-level structure and the top level on the GPU (GpuBVH2Builder.cpp:498-501: the same passes, no treelets at the top) */
    if (twoLevel && (builder == 2 || builder == 4))
        BuildBvhWith(s, [&](HostScene& one) { BuildBvhGpu(c, one, builder == 4 ? 3u : 0u); },
                     [&](HostScene& all, const std::vector<float>& boxes, std::vector<TbNodeB>& top, uint32_t& rootRef, uint32_t& depth) { BuildTlasGpu(c, all,
                         boxes, top, rootRef, depth); });
    else if (twoLevel) BuildBvh(s, (int)builder);
    else if (builder == 2 || builder == 4) BuildBvhGpu(c, s, builder == 4 ? 3u : 0u);
    else BuildBvh(s, (int)builder);
    /* measured on the 870 k scene: 0 -> 2258, 1 -> 2283, 2 (10 levels) -> 2300 Msamples/s */
    if (!twoLevel) reorderNodes(s, (int)opt("node_order", 2), (uint32_t)opt("node_order_top_levels", 10));
    }
    c->camera = s.camera;
    releaseScene(c);
    TbDeviceScene& d = c->ds;
    if (s.nodesB.size() > 0x7fffffffull / 5 || s.trisB.size() > 0x7fffffffull / 3) throw std::runtime_error("scene too large for 31-bit device child refs");
    /* device child refs: offsets in 16-B units (pt_scene.h) */
    auto deviceRef = [](uint32_t ref,
        uint32_t nodeUnits) { return (ref & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((ref & ~TB_BVH_LEAF_FLAG) * 3u)) : ref * nodeUnits; };
    {
        std::vector<TbNodeB> dev(s.nodesB);
        /* two-level scenes: the first M - 1 nodes are the top level, whose leaf refs address 64-B instance records (4 units) */
        const size_t topNodes = s.instances.size() > 1 ? s.instances.size() - 1 : 0;
        auto topRef = [](uint32_t ref) { return (ref & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((ref & ~TB_BVH_LEAF_FLAG) * 4u)) : ref * 4u; };
        for (size_t i = 0; i < dev.size(); i++) {
            TbNodeB& nd = dev[i];
            if (i < topNodes) { nd.left = topRef(nd.left); nd.right = topRef(nd.right); } else { nd.left = deviceRef(nd.left, 4);
                nd.right = deviceRef(nd.right, 4); }
        }
        d.nodes = upload(c, dev);
    }
    d.tris = upload(c, s.trisB);
    d.nodesC = nullptr; c->compactTried = false; /* layout C is built when a render or trace first asks for it (ensureCompactNodes) */
    d.rootRef = twoLevel ? ((s.rootRefB & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((s.rootRefB & ~TB_BVH_LEAF_FLAG) * 4u)) : s.rootRefB * 4u) :
        deviceRef(s.rootRefB, 4); /* 0 or LEAF|0: the same in both images */ d.numNodes = (uint32_t)s.nodesB.size(); d.numTris = (uint32_t)s.trisB.size();
    { const TbAabbNode* root = (const TbAabbNode*)((twoLevel ? s.tlasA.data() : s.bvhA.data()) + 16); memcpy(d.rootCenter, root->center, 12);
        memcpy(d.rootHalf, root->halfDim, 12); }
    {   /* instances in their device form: the bottom-level root as a device child ref */
        std::vector<TbInstanceB> devInst(s.instancesB);
        for (TbInstanceB& ib : devInst) ib.blasRootRef = deviceRef(ib.blasRootRef, 4);
        d.instances = upload(c, devInst); d.numInstances = (uint32_t)devInst.size();
    }
    /* shading records in their 16-B aligned device form (pt_scene.h) */
    std::vector<TbDevHitGroup> devHit(s.hitGroups.size());
    for (size_t i = 0; i < devHit.size(); i++) {
        if (s.hitGroups[i].VertexBufferOffset % 32 || s.hitGroups[i].IndexBufferOffset % 4)
            throw std::runtime_error("hit group buffer offsets must be vertex-/index-aligned");
        devHit[i] = TbDevHitGroup{s.hitGroups[i].MaterialIndex, s.hitGroups[i].VertexBufferOffset / 4, s.hitGroups[i].IndexBufferOffset / 4, 0};
    }
    std::vector<TbDevMaterial> devMat(s.materials.size());
    for (size_t i = 0; i < devMat.size(); i++) { memset(&devMat[i], 0, sizeof(TbDevMaterial)); devMat[i].m = s.materials[i]; }
    std::vector<TbDevLight> devLight(s.lights.size());
    for (size_t i = 0; i < devLight.size(); i++) { memset(&devLight[i], 0, sizeof(TbDevLight)); devLight[i].l = s.lights[i]; }
    d.hitGroups = upload(c, devHit); d.numHitGroups = (uint32_t)s.hitGroups.size();
    d.indexBuffer = upload(c, s.indexBuffer); d.numIndices = (uint32_t)s.indexBuffer.size();
    d.vertexBuffer = upload(c, s.vertexBuffer); d.numVertexFloats = (uint32_t)s.vertexBuffer.size();
    d.materials = upload(c, devMat); d.numMaterials = (uint32_t)s.materials.size();
    d.textureData = upload(c, s.textureData); d.numTextureData = (uint32_t)s.textureData.size();
    d.lights = upload(c, devLight); d.numLights = (uint32_t)s.lights.size();
    d.images = upload(c, s.images); d.numImages = (uint32_t)s.images.size();
    d.texelPool = upload(c, s.texelPool);
    d.envMap = upload(c, s.envMap); d.envWidth = s.envWidth; d.envHeight = s.envHeight;
    d.blueNoise0 = upload(c, s.blueNoise0); d.blueNoise1 = upload(c, s.blueNoise1);
    d.config = s.config;
    /* a root-to-leaf path of bvhMaxDepth nodes has bvhMaxDepth - 1 inner nodes, each of which parks at most one far child: the
     * walk never holds more than bvhMaxDepth - 1 entries (one spare) */
    d.stackDepth = s.bvhMaxDepth < 2 ? 2 : s.bvhMaxDepth;
    d.alphaTest = opt("alpha_test", 0) ? 1u : 0u;
    d.textureUse = sceneTextureUse(c);
    /* whole-scene LDS image */
    {
        std::vector<uint8_t> blob;
        auto put = [&](const void* p, size_t bytes) { while (blob.size() % 16) blob.push_back(0); uint32_t off = (uint32_t)blob.size();
            const uint8_t* b = (const uint8_t*)p; blob.insert(blob.end(), b, b + bytes); return off; };
        auto ldsRef = [](uint32_t ref) { return (ref & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((ref & ~TB_BVH_LEAF_FLAG) * 3u * TB_LDS_TRI_COPIES)) : ref *
            (TB_LDS_NODE_STRIDE / 16); };
        {   /* nodes TB_LDS_NODE_STRIDE apart (pt_scene.h) */
            std::vector<uint8_t> padded(s.nodesB.size() * TB_LDS_NODE_STRIDE, 0);
            for (size_t i = 0; i < s.nodesB.size(); i++) {
                TbNodeB nd = s.nodesB[i]; nd.left = ldsRef(nd.left); nd.right = ldsRef(nd.right);
                memcpy(padded.data() + i * TB_LDS_NODE_STRIDE, &nd, sizeof nd);
            }
            d.offNodes = put(padded.data(), padded.size());
        }
        {   /* six axis-permuted copies per triangle (pt_scene.h): copy = kz * 2 + swapped, (kx, ky) = the two axes after kz, swapped when d[kz] < 0 */
            std::vector<TbTriB> perm(s.trisB.size() * TB_LDS_TRI_COPIES);
            for (size_t i = 0; i < s.trisB.size(); i++)
                for (int kz = 0; kz < 3; kz++)
                    for (int sw = 0; sw < 2; sw++) {
                        int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
                        if (sw) std::swap(kx, ky);
                        const TbTriB& t = s.trisB[i]; TbTriB q = t;
                        const float* src[3] = {t.v0, t.v1, t.v2}; float* dst[3] = {q.v0, q.v1, q.v2};
                        for (int v = 0; v < 3; v++) { dst[v][0] = src[v][kx]; dst[v][1] = src[v][ky]; dst[v][2] = src[v][kz]; }
                        perm[i * TB_LDS_TRI_COPIES + (size_t)(kz * 2 + sw)] = q;
                    }
            d.offTris = put(perm.data(), perm.size() * sizeof(TbTriB));
        }
        d.offHitGroups = put(devHit.data(), devHit.size() * sizeof(TbDevHitGroup));
        d.offIndices = put(s.indexBuffer.data(), s.indexBuffer.size() * 4);
        d.offVertices = put(s.vertexBuffer.data(), s.vertexBuffer.size() * 4);
        d.offMaterials = put(devMat.data(), devMat.size() * sizeof(TbDevMaterial));
        d.offLights = put(devLight.data(), devLight.size() * sizeof(TbDevLight));
        while (blob.size() % 16) blob.push_back(0);
        size_t budget = (size_t)opt("lds_scene_budget", 40 * 1024);
        c->sceneInLds = blob.size() + (size_t)d.stackDepth * 256 * 4 <= budget && opt("scene_in_lds", 1) != 0 && !twoLevel;
        if (c->sceneInLds) { d.ldsBlob = upload(c, blob); d.ldsBlobBytes = (uint32_t)blob.size(); }
        else { d.ldsBlob = nullptr; d.ldsBlobBytes = 0; }
        /* measured on MI355X: LDS-resident scenes are nearly insensitive (at five waves per SIMD 1-2 is best: 6 745 / 6 730 against
         * 6 680 at 4, 6 230 at 12), scenes fetched through the caches gain ~5 % from a late switch to the leaf phase (16-24) */
        d.parkMin = (uint32_t)std::max<int64_t>(1, opt("park_min", c->sceneInLds ? 2 : 24));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sceneFeatures = sceneFeatureMask(s);
    c->hasScene = true;
    c->samplesRendered = 0;
}

} // namespace tbctx
