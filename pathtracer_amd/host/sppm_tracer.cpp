// sppm_tracer.cpp — host-side tracer classes driving the device backend.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ray_tracer.h"
#include "tkhost.h"

namespace tk {

void setHostError(const std::string& e);

RayTracer::RayTracer(int spp, int maxDepth, int numThreads, float russianRoulette)
    : mNumThreads(numThreads), mMaxDepth(maxDepth), mRussianRoulette(russianRoulette) {
    // spp rounded up to a power of two (src/core/TkRayTracer.cpp:20-26)
    spp--;
    spp |= spp >> 1; spp |= spp >> 2; spp |= spp >> 4; spp |= spp >> 8; spp |= spp >> 16;
    mSpp = spp + 1;
}

RayTracer::~RayTracer() { releaseDevice(); }

void RayTracer::releaseDevice() {
    if (mGroup) tkhip_group_destroy(mGroup);   // destroys the contexts of all ranks, mCtx among them
    else if (mCtx) tkhip_destroy(mCtx);
    mGroup = nullptr;
    mCtx = nullptr;
}

bool RayTracer::fail(const char* what) {
    mError = std::string(what) + ": " + tkhip_last_error();
    fprintf(stderr, "\n[Tracer] %s\n", mError.c_str());
    return false;
}

// src/core/TkRayTracer.cpp:36-41
void RayTracer::setScene(const SceneData* scene) {
    if (mState != INIT) return;
    mScene = scene;
}

// src/core/TkRayTracer.cpp:43-52
void RayTracer::setCamera(const Camera* cam) {
    if (mState != INIT && mState != READY) return;
    mCamera = cam;
    mWidth = cam->width;
    mHeight = cam->height;
    mFrameBuffer.assign(size_t(mWidth) * size_t(mHeight), 0u);
    mState = READY;
}

bool RayTracer::uploadScene() {
    if (!mScene || !mCamera) {
        mError = "setScene and setCamera must be called before rendering";
        return false;
    }
    const SceneData& s = *mScene;
    if (mGpus > 1) {
        const int present = tkhip_device_count();
        if (present < 1) {
            mError = "no GPU";
            return false;
        }
        std::vector<int> devices;
        for (int r = 0; r < mGpus; ++r) devices.push_back((mDevice + r) % present);
        if (mGpus > present)
            fprintf(stderr, "\r[Tracer] %d ranks on %d GPU(s): ranks share GPUs\n", mGpus, present);
        if (!mGroup && tkhip_group_create(devices.data(), mGpus, &mGroup) != 0) return fail("tkhip_group_create");
        mCtx = tkhip_group_ctx(mGroup, 0);
        if (tkhip_group_set_scene(mGroup, s.objects.data(), int(s.objects.size()), s.tris.data(), int(s.tris.size()),
                                  s.spheres.data(), int(s.spheres.size()), s.materials.data(), int(s.materials.size()),
                                  s.lights.data(), int(s.lights.size())) != 0)
            return fail("tkhip_group_set_scene");
        if (tkhip_group_set_camera(mGroup, mCamera->c2w, mCamera->rasterToCamera, mWidth, mHeight) != 0)
            return fail("tkhip_group_set_camera");
        return true;
    }
    if (!mCtx && tkhip_create(mDevice, &mCtx) != 0) return fail("tkhip_create");
    if (tkhip_set_scene(mCtx, s.objects.data(), int(s.objects.size()), s.tris.data(), int(s.tris.size()),
                        s.spheres.data(), int(s.spheres.size()), s.materials.data(), int(s.materials.size()),
                        s.lights.data(), int(s.lights.size())) != 0)
        return fail("tkhip_set_scene");
    if (tkhip_set_camera(mCtx, mCamera->c2w, mCamera->rasterToCamera, mWidth, mHeight) != 0)
        return fail("tkhip_set_camera");
    return true;
}

// src/core/TkRayTracer.cpp:54-73 without the GL blit: advances the render by
// one step and refreshes the preview buffer.
void RayTracer::updateScreen() {
    switch (mState) {
    case INIT:
    case READY:
    case DONE: break;
    case VISUALIZE: visualize(); break;
    case RENDERING: rendering(); break;
    }
}

void RayTracer::stop() {   // src/core/TkRayTracer.cpp:75-91
    switch (mState) {
    case INIT:
    case READY: break;
    case VISUALIZE:
    case RENDERING:
    case DONE: mState = READY; break;
    }
}

void RayTracer::clear() {}   // src/core/TkRayTracer.cpp:93-100 (a no-op there too)

void RayTracer::startVisualizing() {
    if (mState != READY) return;
    mState = VISUALIZE;
}

void RayTracer::startRaytracing() {   // src/core/TkRayTracer.cpp:109-122
    if (mState != READY) return;
    std::fill(mFrameBuffer.begin(), mFrameBuffer.end(), 0u);
    mState = RENDERING;
}

void RayTracer::saveImage(std::string) {}

// ------------------------------------------------------------------- SPPM
SppmTracer::SppmTracer(int spp, int maxDepth, int numThreads, float russianRoulette, TkSppmParam param)
    : RayTracer(spp, maxDepth, numThreads, russianRoulette), mParam(param) {
    int x = 0, y = 0;
    while (++x * ++y < param.numRecurse) {}   // src/tracer/TkSppmTracer.cpp:18-20
    numRecurse = x * y;
}

// src/tracer/TkSppmTracer.cpp:385-398
void SppmTracer::startRaytracing() {
    if (mState != READY) return;
    curRecurse = 0;
    mStats.clear();
    releaseDevice();   // a fresh context per render keeps the state machine simple
    if (!uploadScene()) return;
    if (mGroup) {
        if (tkhip_group_sppm_begin(mGroup, &mParam, mMaxDepth, mRussianRoulette, mSeed) != 0) {
            fail("tkhip_group_sppm_begin");
            return;
        }
    } else if (tkhip_sppm_begin(mCtx, &mParam, mMaxDepth, mRussianRoulette, mSeed, 0, 1) != 0) {
        fail("tkhip_sppm_begin");
        return;
    }
    fprintf(stderr, "\r[Tracer] device %d, %d GPU rank(s), %dx%d, %d iterations\n", mDevice, mGpus, mWidth, mHeight, numRecurse);
    // The windowed path shows every iteration; the frame buffer is refreshed on the device (with several
    // ranks: when somebody asks for it, after their rows have been merged).
    if (!mGroup && tkhip_sppm_set_preview(mCtx, 1) != 0) {
        fail("tkhip_sppm_set_preview");
        return;
    }
    mPreviewStale = false;
    RayTracer::startRaytracing();
}

bool SppmTracer::mergeRanks() {
    if (mGroup && tkhip_group_sppm_merge(mGroup) != 0) return fail("tkhip_group_sppm_merge");
    return true;
}

bool SppmTracer::step(TkIterStats* st) {
    if (mGroup) {
        std::vector<TkIterStats> per_rank(st ? size_t(mGpus) : 0);
        if (tkhip_group_sppm_iteration(mGroup, curRecurse, curRecurse + 1 < numRecurse, st ? per_rank.data() : nullptr) != 0)
            return fail("tkhip_group_sppm_iteration");
        if (st) *st = per_rank[0];
        return true;
    }
    if (tkhip_sppm_iteration(mCtx, curRecurse, st) != 0) return fail("tkhip_sppm_iteration");
    return true;
}

bool SppmTracer::resolve(std::vector<uint8_t>* rgb) {
    if (!mergeRanks()) return false;
    rgb->resize(size_t(mWidth) * size_t(mHeight) * 3);
    if (tkhip_sppm_resolve_rgb8(mCtx, numRecurse, rgb->data()) != 0) return fail("tkhip_sppm_resolve_rgb8");
    return true;
}

// One round of the Generate -> Rendering state machine
// (src/tracer/TkSppmTracer.cpp:43-102), i.e. one SPPM iteration, followed by
// the tone-mapped preview of the current estimate.
void SppmTracer::rendering() {
    if (mState != RENDERING) return;
    TkIterStats st;
    memset(&st, 0, sizeof(st));
    if (!step(&st)) {
        mState = READY;
        return;
    }
    mStats.push_back(st);
    fprintf(stderr, "\r[Tracer] Shoot global photons %lld\n", (long long)st.shoot_global);
    fprintf(stderr, "\r[Tracer] Shoot caustic photons %lld\n", (long long)st.shoot_caustic);
    fprintf(stderr, "\r[Tracer] Iteration %d... 100%%  (%.2f ms: photons %.2f, build %.2f, camera %.2f, gather %.2f)\n",
            curRecurse, st.ms_total, st.ms_photon, st.ms_build, st.ms_camera, st.ms_gather);
    ++curRecurse;
    // The tone-mapped frame buffer stays on the device (the library refreshed it behind the pass);
    // the host copy is fetched when somebody asks for it (frameBuffer(), saveImage).
    mPreviewStale = true;
    if (curRecurse >= numRecurse) {
        fprintf(stderr, "\r[Tracer] Rendering Done!\n");
        mState = DONE;
    }
}

const uint32_t* SppmTracer::frameBuffer() {
    if (mPreviewStale && mCtx) {
        int shown = -1;
        if (mGroup && (!mergeRanks() || tkhip_sppm_preview(mCtx, 1, nullptr, nullptr) != 0)) fail("tkhip_sppm_preview");
        else if (tkhip_sppm_read_preview(mCtx, mFrameBuffer.data(), &shown) != 0) fail("tkhip_sppm_read_preview");
        mPreviewStale = false;
    }
    return mFrameBuffer.data();
}

// Mid-render (key 's' of the windowed path, src/TkRenderer.cpp:220-222) the image is the current frame
// buffer, rows flipped as the final PPM has them; after the last iteration it is the resolved image.
void SppmTracer::saveImage(std::string filename) {
    if (mState != DONE && mState != RENDERING) return;
    if (filename.empty()) filename = "binary.ppm";
    std::vector<uint8_t> rgb;
    if (mState == RENDERING) {
        int shown = -1;
        std::vector<uint32_t> fb(size_t(mWidth) * size_t(mHeight));
        if (!mergeRanks() || tkhip_sppm_preview(mCtx, 1, nullptr, nullptr) != 0 ||
            tkhip_sppm_read_preview(mCtx, fb.data(), &shown) != 0) {
            fail("tkhip_sppm_read_preview");
            return;
        }
        rgb.resize(fb.size() * 3);
        for (int y = 0; y < mHeight; ++y)
            for (int x = 0; x < mWidth; ++x) {
                const uint32_t v = fb[size_t(y) * mWidth + x];
                uint8_t* p = &rgb[(size_t(mHeight - 1 - y) * mWidth + x) * 3];
                p[0] = uint8_t(v); p[1] = uint8_t(v >> 8); p[2] = uint8_t(v >> 16);
            }
    } else if (!resolve(&rgb)) return;
    fprintf(stderr, "\n[Tracer] Saving to file: %s... ", filename.c_str());
    if (tkhost_write_ppm(filename.c_str(), mWidth, mHeight, rgb.data()) == 0) fprintf(stderr, "Done!\n");
    else fprintf(stderr, "FAILED\n");
}

// The headless loop (src/tracer/TkSppmTracer.cpp:447-544).  Like the reference's
// it neither tone-maps between iterations nor times their phases (that is what
// rendering(), the step of the windowed path, does): the image is resolved once,
// at the end.  TKHOST_ITER_STATS=1 in the environment switches to the timed,
// previewed steps.
void SppmTracer::render(std::string filename) {
    startRaytracing();
    const char* want_stats = getenv("TKHOST_ITER_STATS");
    if (want_stats && atoi(want_stats) != 0) {
        while (mState == RENDERING) rendering();
    } else {
        if (mState == RENDERING && !mGroup && tkhip_sppm_set_preview(mCtx, 0) != 0) fail("tkhip_sppm_set_preview");
        int64_t shot[2] = {0, 0};
        while (mState == RENDERING) {
            if (!step(nullptr)) {
                mState = READY;
                return;
            }
            int64_t total[2] = {0, 0};
            tkhip_sppm_counters(mCtx, &total[0], &total[1], nullptr);
            fprintf(stderr, "\r[Tracer] Iteration %d... 100%%  (shot %lld global, %lld caustic light paths)\n", curRecurse,
                    (long long)(total[0] - shot[0]), (long long)(total[1] - shot[1]));
            shot[0] = total[0];
            shot[1] = total[1];
            if (++curRecurse >= numRecurse) {
                fprintf(stderr, "\r[Tracer] Rendering Done!\n");
                mState = DONE;
            }
        }
    }
    if (mState == DONE) saveImage(filename);
}

// ------------------------------------------------------ hash-grid SPPM
SPPM::SPPM(int spp, int maxDepth, int numThreads, float russianRoulette, TkSppmParam param)
    : RayTracer(spp, maxDepth, numThreads, russianRoulette), mParam(param) {
    int x = 0, y = 0;
    while (++x * ++y < param.numRecurse) {}   // src/TkSPPM.cpp:120-122
    numIteration = x * y;
}

void SPPM::startRaytracing() {   // src/TkSPPM.cpp:274-282
    if (mState != READY) return;
    curIteration = 0;
    releaseDevice();
    if (mGpus > 1) fprintf(stderr, "\r[Tracer] the hash-grid tracer renders on one GPU; --gpus ignored\n");
    mGpus = 1;
    if (!uploadScene()) return;
    if (tkhip_hsppm_begin(mCtx, &mParam, mSpp, mMaxDepth, mRussianRoulette, mSeed) != 0) {
        fail("tkhip_hsppm_begin");
        return;
    }
    RayTracer::startRaytracing();
}

// One round of ray pass, photon pass and renew pass (the three states of src/TkSPPM.cpp:347-431).
void SPPM::rendering() {
    if (mState != RENDERING) return;
    TkIterStats st;
    if (tkhip_hsppm_iteration(mCtx, curIteration, &st) != 0) {
        fail("tkhip_hsppm_iteration");
        mState = READY;
        return;
    }
    fprintf(stderr, "\r[Tracer] Iteration %d... Done!  (%.2f ms: ray pass %.2f, grid %.2f, photon pass %.2f; %lld visible points)\n",
            curIteration, st.ms_total, st.ms_camera, st.ms_build, st.ms_photon, (long long)st.gathers_caustic);
    if (++curIteration >= numIteration) {
        fprintf(stderr, "\r[Tracer] Rendering Done!\n");
        mState = DONE;
    }
}

void SPPM::saveImage(std::string filename) {
    if (mState != DONE && mState != RENDERING) return;
    if (filename.empty()) filename = "binary.ppm";
    std::vector<uint8_t> rgb(size_t(mWidth) * size_t(mHeight) * 3);
    if (tkhip_hsppm_resolve_rgb8(mCtx, rgb.data()) != 0) {
        fail("tkhip_hsppm_resolve_rgb8");
        return;
    }
    fprintf(stderr, "\n[Tracer] Saving to file: %s... ", filename.c_str());
    if (tkhost_write_ppm(filename.c_str(), mWidth, mHeight, rgb.data()) == 0) fprintf(stderr, "Done!\n");
    else fprintf(stderr, "FAILED\n");
}

void SPPM::render(std::string filename) {   // src/TkSPPM.cpp:347-465
    startRaytracing();
    while (mState == RENDERING) rendering();
    if (mState == DONE) saveImage(filename);
}

// ------------------------------------------------------------ path tracer
PathTracer::PathTracer(int spp, int maxDepth, int numThreads, float russianRoulette)
    : RayTracer(spp, maxDepth, numThreads, russianRoulette) {}

void PathTracer::startRaytracing() {
    if (mState != READY) return;
    mDone = 0;
    releaseDevice();
    if (mGpus > 1) fprintf(stderr, "\r[Tracer] the path tracer renders on one GPU; --gpus ignored\n");
    mGpus = 1;
    if (!uploadScene()) return;
    if (tkhip_pt_begin(mCtx, mSpp, mMaxDepth, mRussianRoulette, mSeed) != 0) {
        fail("tkhip_pt_begin");
        return;
    }
    RayTracer::startRaytracing();
}

void PathTracer::rendering() {   // src/tracer/TkPathTracer.cpp:114-125: progress, done when all work is handed out
    if (mState != RENDERING) return;
    if (tkhip_pt_render_samples(mCtx, 1, nullptr) != 0) {
        fail("tkhip_pt_render_samples");
        mState = READY;
        return;
    }
    ++mDone;
    fprintf(stderr, "\r[Tracer] Rendering...... %02d%%", std::min(mDone * 100 / mSpp, 100));
    if (mDone >= mSpp) {
        fprintf(stderr, "\n");
        mState = DONE;
    }
}

void PathTracer::saveImage(std::string filename) {   // RayTracer::saveImage -> Film::writeImage
    if (mState != DONE && mState != RENDERING) return;
    if (filename.empty()) filename = "binary.ppm";
    std::vector<uint8_t> rgb(size_t(mWidth) * size_t(mHeight) * 3);
    if (tkhip_pt_resolve_rgb8(mCtx, rgb.data()) != 0) {
        fail("tkhip_pt_resolve_rgb8");
        return;
    }
    fprintf(stderr, "\n[Tracer] Saving to file: %s... ", filename.c_str());
    if (tkhost_write_ppm(filename.c_str(), mWidth, mHeight, rgb.data()) == 0) fprintf(stderr, "Done!\n");
    else fprintf(stderr, "FAILED\n");
}

void PathTracer::render(std::string filename) {   // RayTracer::render, src/core/TkRayTracer.cpp:142-190
    startRaytracing();
    if (mState == RENDERING) {
        if (tkhip_pt_render_samples(mCtx, mSpp, nullptr) != 0) {
            fail("tkhip_pt_render_samples");
            mState = READY;
            return;
        }
        mDone = mSpp;
        fprintf(stderr, "\r[Tracer] Rendering...... 100%%\n");
        mState = DONE;
    }
    if (mState == DONE) saveImage(filename);
}

std::unique_ptr<RayTracer> createTracer(const TkTracerConfig& cfg, std::string* error) {
    std::string type = cfg.type;
    if (type == "pt")
        return std::unique_ptr<RayTracer>(new PathTracer(cfg.spp, cfg.max_depth, cfg.num_threads, cfg.russian_roulette));
    if (type == "sppm_hash")
        return std::unique_ptr<RayTracer>(new SPPM(cfg.spp, cfg.max_depth, cfg.num_threads, cfg.russian_roulette, cfg.sppm));
    if (type == "sppm")
        return std::unique_ptr<RayTracer>(
            new SppmTracer(cfg.spp, cfg.max_depth, cfg.num_threads, cfg.russian_roulette, cfg.sppm));
    if (error) *error = "tracer type '" + type + "' has no device path ('sppm' and 'pt' are accelerated)";
    return nullptr;
}

}  // namespace tk

extern "C" int tkhost_render(tkhost_scene* s, int device, int width, int height, const char* out_ppm, uint64_t seed,
                             double* seconds_out) {
    if (!s) {
        tk::setHostError("tkhost_render: null scene");
        return -1;
    }
    tk::Config& cfg = s->config;
    int w = width > 0 ? width : cfg.width, h = height > 0 ? height : cfg.height;
    std::string err;
    std::unique_ptr<tk::RayTracer> tracer = tk::createTracer(cfg.tracer, &err);
    if (!tracer) {
        tk::setHostError(err);
        return -1;
    }
    tk::Camera camera(cfg.camera, w, h);
    tracer->setDevice(device);
    tracer->setGpuCount(cfg.gpus);
    tracer->setSeed(seed);
    tracer->setScene(&cfg.scene);
    tracer->setCamera(&camera);
    auto start = std::chrono::system_clock::now();
    tracer->render(out_ppm ? out_ppm : cfg.outfilename);
    auto stop = std::chrono::system_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(stop - start).count();
    if (!tracer->done()) {
        tk::setHostError(tracer->error().empty() ? "render did not finish" : tracer->error());
        return -1;
    }
    return 0;
}

// The windowed flow without a window (src/TkRenderer.cpp:186-222): startRaytracing(), then one
// updateScreen() per frame; after `save_after` steps the 's' key, i.e. saveImage(mid_ppm), and at
// the end the final image.  frame_rgba (width*height words, may be null) receives the frame buffer
// as it stood when the mid-render image was saved.
extern "C" int tkhost_render_windowed(tkhost_scene* s, int device, int width, int height, const char* out_ppm,
                                      uint64_t seed, int save_after, const char* mid_ppm, uint32_t* frame_rgba) {
    if (!s) {
        tk::setHostError("tkhost_render_windowed: null scene");
        return -1;
    }
    tk::Config& cfg = s->config;
    int w = width > 0 ? width : cfg.width, h = height > 0 ? height : cfg.height;
    std::string err;
    std::unique_ptr<tk::RayTracer> tracer = tk::createTracer(cfg.tracer, &err);
    if (!tracer) {
        tk::setHostError(err);
        return -1;
    }
    tk::Camera camera(cfg.camera, w, h);
    tracer->setDevice(device);
    tracer->setGpuCount(cfg.gpus);
    tracer->setSeed(seed);
    tracer->setScene(&cfg.scene);
    tracer->setCamera(&camera);
    tracer->startRaytracing();
    for (int step = 0; !tracer->done() && tracer->error().empty(); ++step) {
        if (step == save_after) {
            if (mid_ppm) tracer->saveImage(mid_ppm);
            if (frame_rgba) memcpy(frame_rgba, tracer->frameBuffer(), size_t(w) * size_t(h) * 4);
        }
        tracer->updateScreen();
    }
    if (!tracer->done()) {
        tk::setHostError(tracer->error().empty() ? "render did not finish" : tracer->error());
        return -1;
    }
    tracer->saveImage(out_ppm ? out_ppm : cfg.outfilename);
    return 0;
}
