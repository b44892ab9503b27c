// ray_tracer.h — the tracer interface of the host program.
//
// Same public surface as the reference's abstract tk::RayTracer
// (src/core/TkRayTracer.h:11-60) and its SPPM subclass
// (src/tracer/TkSppmTracer.h:21-53): construct from the <tracer> block, give it
// a scene and a camera, call render(filename).  What differs is behind it: the
// scene is the flat POD scene of the loader and every phase of an iteration is
// a call into the device backend (include/tkhip.h) instead of worker threads.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "scene_loader.h"
#include "tkhip.h"

namespace tk {

// What RayTracer::setCamera needs from the reference's Camera + Film pair
// (src/core/camera.cpp:10-20, src/main.cpp:39-46).
struct Camera {
    float c2w[16];
    float rasterToCamera[16];
    int width = 0, height = 0;
    Camera(const TkCameraConfig& info, int w, int h) : width(w), height(h) {
        cameraMatrices(info, w, h, c2w, rasterToCamera);
    }
};

class RayTracer {
protected:
    enum eState { INIT, READY, VISUALIZE, RENDERING, DONE };
    eState mState = INIT;
    const SceneData* mScene = nullptr;
    const Camera* mCamera = nullptr;
    std::vector<uint32_t> mFrameBuffer;   // RGBA8 preview, y down
    int mWidth = 0, mHeight = 0;
    int mNumThreads, mSpp, mMaxDepth;
    float mRussianRoulette;
    int mDevice = 0;
    uint64_t mSeed = 0x5EED;
    tkhip_ctx* mCtx = nullptr;            // the context of rank 0 when mGroup is set (owned by the group then)
    tkhip_group* mGroup = nullptr;        // more than one GPU: one context per rank
    int mGpus = 1;
    std::string mError;

    bool fail(const char* what);          // records tkhip_last_error(), prints it, returns false
    bool uploadScene();
    void releaseDevice();
    virtual void visualize() {}
    virtual void rendering() = 0;         // one step of the render state machine
public:
    RayTracer(int spp, int maxDepth, int numThreads, float russianRoulette);
    virtual ~RayTracer();
    void setDevice(int device) { mDevice = device; }
    void setSeed(uint64_t seed) { mSeed = seed; }
    // Ranks the render is spread over, on GPUs mDevice, mDevice + 1, ... (modulo the GPUs present).
    void setGpuCount(int gpus) { mGpus = gpus < 1 ? 1 : gpus; }
    void setScene(const SceneData* scene);
    void setCamera(const Camera* camera);
    void updateScreen();
    void stop();
    void clear();
    virtual void startVisualizing();
    virtual void startRaytracing();
    virtual void keyPress(int /*key*/) {}
    virtual void saveImage(std::string filename = "");
    virtual void render(std::string filename) = 0;
    bool done() const { return mState == DONE; }
    const std::string& error() const { return mError; }
    // RGBA8, rows in raster order (the buffer updateScreen hands to glDrawPixels in the reference)
    virtual const uint32_t* frameBuffer() { return mFrameBuffer.data(); }
};

class SppmTracer : public RayTracer {
protected:
    TkSppmParam mParam;
    int numRecurse = 0;
    int curRecurse = 0;
    std::vector<TkIterStats> mStats;
    bool mPreviewStale = false;           // the device frame buffer is newer than mFrameBuffer
    void rendering() override;
    bool resolve(std::vector<uint8_t>* rgb);
    bool step(TkIterStats* st);           // one iteration on the context or on the group
    bool mergeRanks();                    // group: every rank's pixel rows into rank 0's context
public:
    SppmTracer(int spp, int maxDepth, int numThreads, float russianRoulette, TkSppmParam param);
    void startRaytracing() override;
    void saveImage(std::string filename = "") override;
    void render(std::string filename) override;
    const uint32_t* frameBuffer() override;
    // Progressive display: the device-resident frame buffer is refreshed after every n-th step of
    // updateScreen() / rendering() (call after startRaytracing; the windowed path uses 1).
    bool setPreviewInterval(int n) { return mCtx && !mGroup && tkhip_sppm_set_preview(mCtx, n) == 0; }
    const std::vector<TkIterStats>& stats() const { return mStats; }
};

// The reference's second SPPM class (src/TkSPPM.h:34-64): visible point per pixel, spatial hash
// grid, photons deposited into the pixels they reach.  Same constructor, same three calls.
class SPPM : public RayTracer {
    TkSppmParam mParam;
    int numIteration = 0;
    int curIteration = 0;
    void rendering() override;
public:
    SPPM(int spp, int maxDepth, int numThreads, float russianRoulette, TkSppmParam param);
    void startRaytracing() override;
    void saveImage(std::string filename = "") override;
    void render(std::string filename) override;
};

// <tracer type="pt">: the reference's path tracer (src/tracer/TkPathTracer.h:9-19) on the path
// kernels.  One step of the state machine adds one sample per pixel to the film (the reference
// advances tile by tile); the BVH visualisation of the windowed program is not carried over.
class PathTracer : public RayTracer {
    int mDone = 0;                        // samples per pixel in the film so far
    void rendering() override;
public:
    PathTracer(int spp, int maxDepth, int numThreads, float russianRoulette);
    void startRaytracing() override;
    void saveImage(std::string filename = "") override;
    void render(std::string filename) override;
};

// parse_tracer's dispatch on the type attribute (src/TkLoader.cpp:238-262).
// "sppm" (tk::SppmTracer), "sppm_hash" (tk::SPPM, chosen by <hashgrid i="1"/>) and "pt"
// (tk::PathTracer) have a device path; "bdpt" and "light" return null with a message in *error.
std::unique_ptr<RayTracer> createTracer(const TkTracerConfig& cfg, std::string* error);

}  // namespace tk
