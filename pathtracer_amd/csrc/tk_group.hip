// tk_group.hip — one SPPM render over several GPUs of one process.
//
// The CPU renderer spreads an iteration over the threads of one machine
// (TkSpawn / TkWait in src/Threads.cpp:19-56, called from
// src/tracer/TkSppmTracer.cpp:411-438, :456-496).  Here the unit is a GPU: a
// group holds one context per GPU and one host thread per context, and an
// iteration is
//
//   every rank       light paths of its share (Halton index = rank mod world) and, beside them,
//                    the camera paths of its pixel rows;
//   barrier          every rank has filled its quota of both maps;
//   every rank       pulls the other ranks' photons into its own concatenated arrays with direct
//                    peer copies (xGMI is point to point: world-1 independent copies per GPU,
//                    none of them relayed) and installs them as the two maps;
//   barrier          everything has been pulled, the ranks' own arrays may be overwritten;
//   every rank       on request the light paths of the NEXT iteration are enqueued (they depend on
//                    nothing this iteration still computes); the trees are built; on request the
//                    camera paths of the next iteration are started (into a second set of query
//                    buffers); the rank's queries are gathered.
//
// No collective library is involved and the host never reads a photon: the
// counts a rank publishes are the ones its photon pass already holds.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tk_common.h"

namespace tkd {

namespace {

// A barrier that can be broken: a rank that fails wakes the others, which then fail too
// instead of waiting for it for ever.
class Barrier {
    std::mutex m_;
    std::condition_variable cv_;
    int n_, waiting_ = 0;
    uint64_t round_ = 0;
    bool broken_ = false;

public:
    explicit Barrier(int n) : n_(n) {}
    bool arrive() {
        std::unique_lock<std::mutex> lk(m_);
        if (broken_) return false;
        if (++waiting_ == n_) {
            waiting_ = 0;
            ++round_;
            cv_.notify_all();
            return true;
        }
        const uint64_t mine = round_;
        cv_.wait(lk, [&] { return round_ != mine || broken_; });
        return !broken_;
    }
    void breakup() {
        std::lock_guard<std::mutex> lk(m_);
        broken_ = true;
        cv_.notify_all();
    }
    void reset() {
        std::lock_guard<std::mutex> lk(m_);
        broken_ = false;
        waiting_ = 0;
    }
};

__global__ void k_merge_rows(float* dst, const float* src, int width, int height, int world, int rank) {
    const int64_t np = (int64_t)width * height;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= np * 13) return;
    const int y = (int)((i % np) / width);
    if ((y / 8) % world == rank) dst[i] = src[i];
}

}  // namespace

}  // namespace tkd

using namespace tkd;

struct tkhip_group {
    int n = 0;
    std::vector<tkhip_ctx*> ctx;
    std::vector<int> device;
    struct Local {
        const void *pos = nullptr, *pow = nullptr, *dir = nullptr;
        int64_t n = 0, shot = 0;
    };
    std::vector<Local> local[2];                 // what every rank's last photon pass stored, per map
    struct Bufs {
        DevBuf pos[2], pow[2], dir[2], planes, incoming;
    };
    std::vector<Bufs> bufs;
    int width = 0, height = 0, next_begun = -1;  // next_begun: iteration whose light paths are already in flight
    bool rendering = false;
    // workers
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t generation = 0;
    int pending = 0;
    bool quit = false;
    std::function<int(int)> job;
    std::vector<int> rc;
    std::vector<std::string> err;
    Barrier* bar = nullptr;

    void worker(int r) {
        (void)hipSetDevice(device[r]);
        uint64_t seen = 0;
        for (;;) {
            std::function<int(int)> fn;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_go.wait(lk, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
                fn = job;
            }
            int code = fn(r);
            std::string msg = code != TKHIP_OK ? tkhip_last_error() : "";
            if (code != TKHIP_OK) bar->breakup();
            {
                std::lock_guard<std::mutex> lk(m);
                rc[r] = code;
                err[r] = msg;
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }

    // fn(rank) on every rank's thread at once; the first failure (by rank) is the group's error.
    int run(std::function<int(int)> fn) {
        bar->reset();
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(fn);
            pending = n;
            ++generation;
        }
        cv_go.notify_all();
        {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        int first = -1;   // the rank that failed on its own, not one that was woken by its failure
        for (int r = n - 1; r >= 0; --r)
            if (rc[r] != TKHIP_OK && err[r].find("aborted:") != 0) first = r;
        for (int r = n - 1; r >= 0 && first < 0; --r)
            if (rc[r] != TKHIP_OK) first = r;
        if (first < 0) return TKHIP_OK;
        set_error("GPU %d of the group (device %d): %s", first, device[first], err[first].c_str());
        return rc[first];
    }
};

namespace {

int aborted() {
    set_error("aborted: another GPU of the group failed");
    return TKHIP_E_STATE;
}

// The exchange and the rest of iteration `iter` on rank r; the light paths are in flight or done.
int rank_iteration(tkhip_group* g, int r, int iter, int begin_next, TkIterStats* st) {
    tkhip_ctx* ctx = g->ctx[r];
    TK_HIP(hipSetDevice(g->device[r]));
    if (g->next_begun != iter) TK_TRY(tkhip_sppm_photon_begin(ctx, iter));
    TK_TRY(tkhip_sppm_camera_begin(ctx, iter));
    TK_TRY(tkhip_sppm_photon_pass(ctx, iter, st));
    for (int w = 0; w < 2; ++w) {
        tkhip_group::Local& L = g->local[w][r];
        TK_TRY(sppm_local_photons(ctx, w, &L.pos, &L.pow, &L.dir, &L.n, &L.shot));
    }
    if (!g->bar->arrive()) return aborted();
    int64_t total[2] = {0, 0}, shot[2] = {0, 0};
    tkhip_group::Bufs& B = g->bufs[r];
    for (int w = 0; w < 2; ++w) {
        for (int q = 0; q < g->n; ++q) { total[w] += g->local[w][q].n; shot[w] += g->local[w][q].shot; }
        const size_t cap = (size_t)std::max<int64_t>(total[w], 1);
        TK_TRY(B.pos[w].ensure(cap * 12));
        TK_TRY(B.pow[w].ensure(cap * 12));
        TK_TRY(B.dir[w].ensure(cap * 2));
        int64_t off = 0;
        for (int q = 0; q < g->n; ++q) {
            const tkhip_group::Local& L = g->local[w][q];
            if (L.n == 0) continue;
            TK_HIP(hipMemcpyPeerAsync((char*)B.pos[w].p + off * 12, g->device[r], L.pos, g->device[q], (size_t)L.n * 12, ctx->stream));
            TK_HIP(hipMemcpyPeerAsync((char*)B.pow[w].p + off * 12, g->device[r], L.pow, g->device[q], (size_t)L.n * 12, ctx->stream));
            TK_HIP(hipMemcpyPeerAsync((char*)B.dir[w].p + off * 2, g->device[r], L.dir, g->device[q], (size_t)L.n * 2, ctx->stream));
            off += L.n;
        }
    }
    TK_HIP(hipStreamSynchronize(ctx->stream));
    if (!g->bar->arrive()) return aborted();
    for (int w = 0; w < 2; ++w)
        TK_TRY(tkhip_sppm_set_map_photons(ctx, w, B.pos[w].as<float>(), B.pow[w].as<float>(), B.dir[w].as<uint16_t>(), total[w],
                                          shot[w]));
    if (begin_next >= 1) TK_TRY(tkhip_sppm_photon_begin(ctx, iter + 1));
    TK_TRY(tkhip_sppm_build_maps(ctx, st));
    // begin_next 2: the camera paths of iteration iter + 1 are started as well, before this iteration's gathers
    TK_TRY(begin_next >= 2 ? tkhip_sppm_camera_pass_ahead(ctx, iter, st) : tkhip_sppm_camera_pass(ctx, iter, st));
    return TKHIP_OK;
}

}  // namespace

extern "C" {

int tkhip_group_create(const int* devices, int n, tkhip_group** out) {
    TK_REQUIRE(devices && out && n >= 1 && n <= 64, TKHIP_E_ARG, "tkhip_group_create: 1 to 64 devices");
    tkhip_group* g = new tkhip_group();
    g->n = n;
    g->device.assign(devices, devices + n);
    g->ctx.assign((size_t)n, nullptr);
    g->local[0].resize((size_t)n);
    g->local[1].resize((size_t)n);
    g->bufs.resize((size_t)n);
    g->rc.assign((size_t)n, 0);
    g->err.assign((size_t)n, "");
    g->bar = new Barrier(n);
    int rc = TKHIP_OK;
    for (int r = 0; r < n && rc == TKHIP_OK; ++r) rc = tkhip_create(devices[r], &g->ctx[(size_t)r]);
    // direct access between every pair of distinct GPUs (already enabled is fine)
    for (int r = 0; r < n && rc == TKHIP_OK; ++r)
        for (int q = 0; q < n; ++q) {
            if (devices[q] == devices[r]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[r], devices[q]) != hipSuccess || !can) continue;
            (void)hipSetDevice(devices[r]);
            hipError_t e = hipDeviceEnablePeerAccess(devices[q], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // copies are staged then
        }
    if (rc != TKHIP_OK) {
        for (tkhip_ctx* c : g->ctx) tkhip_destroy(c);
        delete g->bar;
        delete g;
        return rc;
    }
    for (int r = 0; r < n; ++r) g->threads.emplace_back([g, r] { g->worker(r); });
    *out = g;
    return TKHIP_OK;
}

void tkhip_group_destroy(tkhip_group* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->quit = true;
    }
    g->cv_go.notify_all();
    for (std::thread& t : g->threads) t.join();
    for (int r = 0; r < g->n; ++r) {
        (void)hipSetDevice(g->device[(size_t)r]);
        tkhip_group::Bufs& B = g->bufs[(size_t)r];
        for (int w = 0; w < 2; ++w) { B.pos[w].release(); B.pow[w].release(); B.dir[w].release(); }
        B.planes.release();
        B.incoming.release();
        tkhip_destroy(g->ctx[(size_t)r]);
    }
    delete g->bar;
    delete g;
}

int tkhip_group_size(const tkhip_group* g) { return g ? g->n : 0; }

tkhip_ctx* tkhip_group_ctx(tkhip_group* g, int rank) {
    if (!g || rank < 0 || rank >= g->n) return nullptr;
    return g->ctx[(size_t)rank];
}

int tkhip_group_set_scene(tkhip_group* g, const TkObject* objects, int n_objects, const TkTriangle* tris, int n_tris,
                          const TkSphere* spheres, int n_spheres, const TkMaterial* materials, int n_materials,
                          const TkLight* lights, int n_lights) {
    TK_REQUIRE(g, TKHIP_E_ARG, "tkhip_group_set_scene: null group");
    return g->run([=](int r) {
        return tkhip_set_scene(g->ctx[(size_t)r], objects, n_objects, tris, n_tris, spheres, n_spheres, materials, n_materials,
                               lights, n_lights);
    });
}

int tkhip_group_set_camera(tkhip_group* g, const float c2w[16], const float raster2cam[16], int width, int height) {
    TK_REQUIRE(g, TKHIP_E_ARG, "tkhip_group_set_camera: null group");
    TK_TRY(g->run([=](int r) { return tkhip_set_camera(g->ctx[(size_t)r], c2w, raster2cam, width, height); }));
    g->width = width;
    g->height = height;
    return TKHIP_OK;
}

int tkhip_group_sppm_begin(tkhip_group* g, const TkSppmParam* param, int max_depth, float russian_roulette, uint64_t seed) {
    TK_REQUIRE(g && param, TKHIP_E_ARG, "tkhip_group_sppm_begin: null argument");
    g->next_begun = -1;
    TK_TRY(g->run([=](int r) { return tkhip_sppm_begin(g->ctx[(size_t)r], param, max_depth, russian_roulette, seed, r, g->n); }));
    g->rendering = true;
    return TKHIP_OK;
}

int tkhip_group_sppm_iteration(tkhip_group* g, int iter, int begin_next, TkIterStats* stats) {
    TK_REQUIRE(g && g->rendering, TKHIP_E_STATE, "tkhip_group_sppm_iteration: call tkhip_group_sppm_begin first");
    if (g->n == 1) return tkhip_sppm_iteration(g->ctx[0], iter, stats);
    TK_REQUIRE(g->next_begun < 0 || g->next_begun == iter, TKHIP_E_STATE,
               "tkhip_group_sppm_iteration: the light paths of iteration %d are in flight, not those of %d", g->next_begun, iter);
    if (stats) memset(stats, 0, sizeof(TkIterStats) * (size_t)g->n);
    int rc = g->run([=](int r) { return rank_iteration(g, r, iter, begin_next, stats ? stats + r : nullptr); });
    g->next_begun = rc == TKHIP_OK && begin_next ? iter + 1 : -1;
    return rc;
}

int tkhip_group_sppm_merge(tkhip_group* g) {
    TK_REQUIRE(g && g->rendering, TKHIP_E_STATE, "tkhip_group_sppm_merge: call tkhip_group_sppm_begin first");
    if (g->n == 1) return TKHIP_OK;
    const size_t bytes = (size_t)g->width * (size_t)g->height * 13 * sizeof(float);
    // Every rank but rank 0 snapshots its planes (camera paths begun ahead with begin_next 2 are waited for:
    // the snapshot then holds their direct light, as the rank's own planes do).  Rank 0 only waits.
    TK_TRY(g->run([=](int r) {
        TK_HIP(hipSetDevice(g->device[(size_t)r]));
        if (r == 0) return sppm_wait_camera_lane(g->ctx[0]);
        TK_TRY(g->bufs[(size_t)r].planes.ensure(bytes));
        return tkhip_sppm_export_planes(g->ctx[(size_t)r], g->bufs[(size_t)r].planes.as<float>());
    }));
    // The other ranks' rows go straight into rank 0's planes; its own rows are not rewritten, so whatever its
    // camera paths begun ahead have added to them stays, and the render goes on from the merged state.
    tkhip_ctx* c0 = g->ctx[0];
    tkhip_group::Bufs& B0 = g->bufs[0];
    TK_HIP(hipSetDevice(g->device[0]));
    TK_TRY(B0.incoming.ensure(bytes));
    float* live = sppm_planes_device(c0);
    TK_REQUIRE(live, TKHIP_E_STATE, "tkhip_group_sppm_merge: rank 0 holds no render");
    const int64_t words = (int64_t)(bytes / 4);
    for (int q = 1; q < g->n; ++q) {
        TK_HIP(hipMemcpyPeerAsync(B0.incoming.p, g->device[0], g->bufs[(size_t)q].planes.p, g->device[(size_t)q], bytes, c0->stream));
        k_merge_rows<<<divup(words, 256), 256, 0, c0->stream>>>(live, B0.incoming.as<float>(), g->width, g->height, g->n, q);
        TK_LAUNCH_CHECK();
    }
    TK_HIP(hipStreamSynchronize(c0->stream));
    return TKHIP_OK;
}

}  // extern "C"
