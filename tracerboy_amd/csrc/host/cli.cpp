/* cli.cpp -- headless replacement of the reference's Win32 shell (WinMain/WinMain.cpp, D3D12App.cpp):
 *   tracerboy-hip scene.pbrt [--width W] [--height H] [--spp N] [--depth D] [--seed-time T] [--device I]
 *                 [--builder lbvh|sah|lbvh-gpu|treelets|treelets-gpu] [--blue-noise 0|1] [--tonemap 0..7] [--exposure E|auto]
 *                 [--out frame.png|frame.pfm|frame.exr]
 *                 [--ranks N]
 * Uses only the C ABI (include/tracerboy_hip.h), the way an embedding application would.
 *
 * --ranks N (N > 1): the frame tiled across N GPUs of the node, natively.  The process starts N copies of itself -- before it
 * has made a single HIP call -- one per GPU (rank r -> device r); each loads the scene, takes its tiles (tb_set_tile_assignment:
 * tile t -> rank t % N), renders, packs (tb_pack_owned_device) and the packed HDR buffers travel to rank 0 over RCCL / xGMI as ONE
 * grouped exchange (ncclGroupStart; rank 0: N - 1 x ncclRecv, the others: one ncclSend; ncclGroupEnd -- SURVEY.md 8e), on the
 * context's stream; rank 0 un-permutes them on the device straight into its own accumulation surface
 * (tb_unpack_gathered_device -> tb_accum_device_ptr) and writes the picture exactly like the one-GPU path.  librccl is loaded with
 * dlopen, so the tool still runs where it is absent; the unique id goes from rank 0 to the others through a file.
 * TB_CLI_FORCE_RCCL=1 runs the same sequence with one rank (communicator of size 1, self-gather): the test of the plumbing on a
 * one-GPU machine.
 * Output by extension: .png = what the reference presents (auto exposure + PostProcessCS tonemap, 8-bit back buffer,
 * tb_post_process); .pfm = linear radiance sum(rgb*w)/sum(w), the value PostProcessCS divides out before tonemapping
 * (PostProcessCS.hlsl:23-47), RGB float32, bottom row first; .exr = the same radiance as OpenEXR (RGBA float32, A = 1). */
#include "../../../include/tracerboy_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int fail(tb_context* c, const char* what, int rc)
{
    fprintf(stderr, "tracerboy-hip: %s failed (%d): %s\n", what, rc, tb_last_error(c));
    if (c) tb_destroy(c);
    return 1;
}

extern char** environ;

/* ---- RCCL, loaded at run time: the handful of entry points the gather needs (rccl.h) ---- */
struct RcclId { char internal[128]; };                 /* ncclUniqueId */
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, RcclId id, int rank) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int dtype, int peer, void* comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int dtype, int peer, void* comm, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load()
    {
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) { fprintf(stderr, "tracerboy-hip: cannot load librccl: %s\n", dlerror()); return false; }
#define SYM(field, name) if (!((*(void**)&field) = dlsym(lib, name))) { fprintf(stderr, "tracerboy-hip: librccl lacks %s\n", name); return false; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
        SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(CommDestroy, "ncclCommDestroy") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        return true;
    }
};
static const int kNcclFloat = 7; /* ncclFloat32 */

/* start `world` copies of this program, one per rank, and wait for them; nothing here touches HIP */
static int spawnRanks(int argc, char** argv, int world)
{
    /* The RCCL unique id travels from rank 0 to the others through a file in a directory only this user can enter (mkdtemp: mode
     * 0700, unpredictable name): nobody else can pre-create the file, plant a symlink where rank 0 writes, or read the id. */
    char idDir[] = "/tmp/tracerboy-hip-rccl-XXXXXX";
    if (!mkdtemp(idDir)) { perror("tracerboy-hip: mkdtemp"); return 1; }
    const std::string idFile = std::string(idDir) + "/id", idTmp = idFile + ".tmp";
    auto cleanup = [&]() { unlink(idFile.c_str()); unlink(idTmp.c_str()); rmdir(idDir); };
    std::vector<pid_t> kids;
    for (int r = 0; r < world; r++) {
        std::vector<std::string> envs;
        for (char** e = environ; *e; e++) if (strncmp(*e, "TB_CLI_", 7)) envs.push_back(*e);
        envs.push_back("TB_CLI_RANK=" + std::to_string(r)); envs.push_back("TB_CLI_WORLD=" + std::to_string(world)); envs.push_back("TB_CLI_ID_FILE=" + idFile);
        bool ipc = false; for (const std::string& e : envs) ipc |= e.rfind("HSA_ENABLE_IPC_MODE_LEGACY=", 0) == 0;
        if (!ipc) envs.push_back("HSA_ENABLE_IPC_MODE_LEGACY=0"); /* this pool's driver only supports dmabuf IPC */
        std::vector<char*> envp; for (std::string& e : envs) envp.push_back(&e[0]); envp.push_back(nullptr);
        pid_t pid = 0;
        if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, argv, envp.data()) != 0) {
            perror("tracerboy-hip: posix_spawn");
            for (pid_t k : kids) kill(k, SIGTERM);
            for (size_t i = 0; i < kids.size(); i++) { int st = 0; (void)waitpid(kids[i], &st, 0); }
            cleanup(); return 1;
        }
        kids.push_back(pid);
    }
    /* a rank that fails (no such device, scene error ...) would leave the others waiting in the communicator forever: the first
     * non-zero exit ends the ranks still running, and ITS status is what the tool returns (the others then die of the SIGTERM) */
    int first = 0;
    while (!kids.empty()) {
        int st = 0; const pid_t k = waitpid(-1, &st, 0);
        if (k < 0) break;
        auto it = std::find(kids.begin(), kids.end(), k);
        if (it == kids.end()) continue;                 /* not one of ours */
        kids.erase(it);                                 /* reaped: its pid may be reused, never signal it again */
        const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (rc != 0 && first == 0) { first = rc; for (pid_t o : kids) kill(o, SIGTERM); }
    }
    cleanup();
    (void)argc;
    return first;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr,
        "usage: tracerboy-hip scene.pbrt [--width W --height H --spp N --depth D --seed-time T --device I --builder lbvh|sah|lbvh-gpu|treelets|treelets-gpu --blue-noise 0|1 --tonemap 0..7 --exposure E|auto --out f.png|f.pfm|f.exr]\n"); return 2; }
    std::string scene = argv[1], out = "frame.png";
    tb_post_settings post; tb_default_post_settings(&post);
    uint32_t W = 0, H = 0, spp = 64; int depth = -1, device = 0, builder = 0, blue = -1, ranks = 1; float t = 0.0f;
    for (int i = 2; i + 1 < argc; i += 2) {
        std::string k = argv[i]; const char* v = argv[i + 1];
        if (k == "--width") W = (uint32_t)atoi(v); else if (k == "--height") H = (uint32_t)atoi(v); else if (k == "--spp") spp = (uint32_t)atoi(v);
        else if (k == "--depth") depth = atoi(v); else if (k == "--seed-time") t = (float)atof(v); else if (k == "--device") device = atoi(v);
        else if (k == "--builder") builder = !strcmp(v, "sah") ? 1 : !strcmp(v, "lbvh-gpu") ? 2 : !strcmp(v, "treelets") ? 3 : !strcmp(v,
            "treelets-gpu") ? 4 : 0; /* tb_set_option "bvh_builder" */ else if (k == "--blue-noise") blue = atoi(v); else if (k == "--out") out = v;
            else if (k == "--ranks") ranks = atoi(v);
        else if (k == "--tonemap") post.TonemapType = (uint32_t)atoi(v);
        else if (k == "--exposure") { if (!strcmp(v, "auto")) post.EnableAutoExposure = 1; else { post.EnableAutoExposure = 0;
            post.ExposureMultiplier = (float)atof(v); } }
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    /* multi-GPU: the parent only starts the ranks; a rank knows itself from the environment */
    const char* envRank = getenv("TB_CLI_RANK");
    const bool forceRccl = getenv("TB_CLI_FORCE_RCCL") && atoi(getenv("TB_CLI_FORCE_RCCL")) != 0;
    if (ranks < 1) { fprintf(stderr, "--ranks must be at least 1\n"); return 2; }
    if (ranks > 1 && !envRank) return spawnRanks(argc, argv, ranks);
    if (envRank && (!getenv("TB_CLI_WORLD") || atoi(getenv("TB_CLI_WORLD")) < 1 || atoi(envRank) < 0 || atoi(envRank) >= atoi(getenv("TB_CLI_WORLD")) ||
                    (atoi(getenv("TB_CLI_WORLD")) > 1 && !getenv("TB_CLI_ID_FILE")))) {
        fprintf(stderr,
            "tracerboy-hip: TB_CLI_RANK needs TB_CLI_WORLD (rank < world) and, for more than one rank, TB_CLI_ID_FILE -- these are set by --ranks, not by hand\n"); return 2;
    }
    const int rank = envRank ? atoi(envRank) : 0, world = envRank ? atoi(getenv("TB_CLI_WORLD")) : 1;
    if (world > 1) device = rank; /* one process per GPU */
    tb_context* ctx = nullptr;
    int rc = tb_create(&ctx, device);
    if (rc) return fail(nullptr, "tb_create", rc);
    tb_set_option(ctx, "bvh_builder", builder);
    auto t0 = std::chrono::steady_clock::now();
    if ((rc = tb_load_scene(ctx, scene.c_str()))) return fail(ctx, "tb_load_scene", rc);
    double loadS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    tb_scene_info info; tb_scene_info_get(ctx, &info);
    if (!W) W = info.filmWidth ? info.filmWidth : 1920;
    if (!H) H = info.filmHeight ? info.filmHeight : 1080;
    tb_output_settings s; tb_default_output_settings(&s);
    if (depth >= 0) s.MaxBounces = depth;
    if (blue >= 0) s.EnableBlueNoise = (uint32_t)blue;
    const uint32_t TILE = 64;
    if (world > 1 && (rc = tb_set_tile_assignment(ctx, (uint32_t)rank, (uint32_t)world, TILE, TILE))) return fail(ctx, "tb_set_tile_assignment", rc);
    auto r0 = std::chrono::steady_clock::now();
    if ((rc = tb_render(ctx, W, H, spp, &s, t))) return fail(ctx, "tb_render", rc);
    float ms = tb_last_render_ms(ctx);
    if (world > 1 || forceRccl) {
        /* ---- the gather: packed tiles of every rank -> rank 0's accumulation surface ---- */
        Rccl nccl; if (!nccl.load()) { tb_destroy(ctx); return 1; }
#define NCCL_TRY(x) do { int e_ = (x); if (e_) { fprintf(stderr, "tracerboy-hip: rank %d: %s: %s\n", rank, #x, nccl.GetErrorString(e_)); tb_destroy(ctx); \
    return 1; } } while (0)
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "tracerboy-hip: rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); \
    tb_destroy(ctx); return 1; } } while (0)
        RcclId id; memset(&id, 0, sizeof id);
        const char* idFile = getenv("TB_CLI_ID_FILE");
        if (rank == 0) {
            NCCL_TRY(nccl.GetUniqueId(&id));
            if (world > 1) { /* publish atomically: write beside, rename */
                if (!idFile) { fprintf(stderr, "tracerboy-hip: TB_CLI_ID_FILE is not set\n"); return 1; }
                const std::string tmp = std::string(idFile) + ".tmp";
                /* never through a link, never over an existing file */
                const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
                if (fd < 0 || write(fd, &id, sizeof id) != (ssize_t)sizeof id) { perror("tracerboy-hip: unique id file"); if (fd >= 0) close(fd); return 1; }
                close(fd); if (rename(tmp.c_str(), idFile)) { perror("tracerboy-hip: rename"); return 1; }
            }
        } else {
            bool got = false;
            for (int tries = 0; tries < 6000 && !got; tries++) { /* up to 10 minutes: rank 0 may still be building its BVH */
                FILE* f = idFile ? fopen(idFile, "rb") : nullptr;
                if (f) { got = fread(&id, sizeof id, 1, f) == 1; fclose(f); }
                if (!got) usleep(100 * 1000);
            }
            if (!got) { fprintf(stderr, "tracerboy-hip: rank %d: no unique id from rank 0\n", rank); tb_destroy(ctx); return 1; }
        }
        HIP_OK(hipSetDevice(device));
        void* comm = nullptr;
        NCCL_TRY(nccl.CommInitRank(&comm, world, id, rank));
        hipStream_t stream = (hipStream_t)tb_stream(ctx);
        const uint32_t tilesX = (W + TILE - 1) / TILE, tilesY = (H + TILE - 1) / TILE, tiles = tilesX * tilesY;
        const uint64_t capacity = (uint64_t)((tiles + world - 1) / world) * TILE * TILE; /* pixels per rank buffer: every rank pads to the largest owner */
        float* packed = nullptr; HIP_OK(hipMalloc((void**)&packed, capacity * 16)); HIP_OK(hipMemsetAsync(packed, 0, capacity * 16, stream));
        if ((rc = tb_pack_owned_device_async(ctx, packed))) return fail(ctx, "tb_pack_owned_device_async", rc);
        float* gathered = nullptr;
        if (rank == 0) HIP_OK(hipMalloc((void**)&gathered, capacity * 16 * (uint64_t)world));
        NCCL_TRY(nccl.GroupStart());
        if (rank == 0) { for (int r = 1; r < world; r++) NCCL_TRY(nccl.Recv(gathered + (size_t)r * capacity * 4, capacity * 4, kNcclFloat, r, comm, stream)); }
        else NCCL_TRY(nccl.Send(packed, capacity * 4, kNcclFloat, 0, comm, stream));
        NCCL_TRY(nccl.GroupEnd());
        if (rank == 0) {
            HIP_OK(hipMemcpyAsync(gathered, packed, capacity * 16, hipMemcpyDeviceToDevice, stream));
            void *surface = nullptr, *jit = nullptr;
            if ((rc = tb_accum_device_ptr(ctx, &surface, &jit))) return fail(ctx, "tb_accum_device_ptr", rc);
            if ((rc = tb_unpack_gathered_device(ctx, stream, gathered, capacity, W, H, (uint32_t)world, TILE, TILE, surface))) return fail(ctx,
                "tb_unpack_gathered_device", rc);
        }
        if ((rc = tb_sync(ctx))) return fail(ctx, "tb_sync", rc);
        NCCL_TRY(nccl.CommDestroy(comm));
        (void)hipFree(packed); if (gathered) (void)hipFree(gathered);
        if (rank != 0) { tb_destroy(ctx); return 0; } /* the picture is rank 0's to write */
        /* render + gather + un-permute, rank 0's wall clock */
        ms = (float)(std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count() * 1e3);
#undef NCCL_TRY
#undef HIP_OK
    }
    const bool png = out.size() >= 4 && out.compare(out.size() - 4, 4, ".png") == 0;
    if (png) {
        std::vector<uint8_t> img((size_t)W * H * 4);
        if ((rc = tb_post_process(ctx, &post, TB_OUTPUT_TYPE_LIT, nullptr, img.data()))) return fail(ctx, "tb_post_process", rc);
        if ((rc = tb_write_image_rgba8(out.c_str(), W, H, img.data()))) return fail(ctx, "tb_write_image_rgba8", rc);
    } else {
        std::vector<float> acc((size_t)W * H * 4);
        if ((rc = tb_read_accum(ctx, acc.data(), nullptr))) return fail(ctx, "tb_read_accum", rc);
        for (size_t i = 0; i < (size_t)W * H; i++) { float w = acc[4 * i + 3], inv = w > 0 ? 1.0f / w : 0.0f; acc[4 * i] *= inv; acc[4 * i + 1] *= inv;
            acc[4 * i + 2] *= inv; acc[4 * i + 3] = w > 0 ? 1.0f : 0.0f; }
        if ((rc = tb_write_image_f32(out.c_str(), W, H, acc.data()))) return fail(ctx, "tb_write_image_f32 (use .png, .pfm or .exr)", rc);
    }
    printf("%s: %u triangles, %ux%u x %u spp, depth %d, %d GPU%s: %.2f ms (%.1f Msamples/s), scene load + BVH %.2f s -> %s\n",
           scene.c_str(), info.numTriangles, W, H, spp, s.MaxBounces, world, world > 1 ? "s (tiles gathered over RCCL)" : "", ms,
               (double)W * H * spp / (ms * 1e3), loadS, out.c_str());
    tb_destroy(ctx);
    return 0;
}
