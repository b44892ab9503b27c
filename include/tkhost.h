/*
 * tkhost.h — C view of the host side: the .scene/OBJ/PLY loader and the
 * headless render driver.  The C++ classes behind it (pathtracer_amd/host/)
 * mirror the reference's loader and tracer interface; this flat API exists so
 * that tools without a C++ compiler (the Python tests and bench) can use the
 * same loader.  Citations refer to the upstream CPU renderer (src/...).
 */
#ifndef TKHOST_H
#define TKHOST_H

#include "tkhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tkhost_scene tkhost_scene;

/* Tracer block of a .scene file: parse_tracer, src/TkLoader.cpp:228-263. */
typedef struct TkTracerConfig {
    char type[16];          /* "sppm", "pt", "bdpt", "light" */
    int32_t spp;            /* default 16 */
    int32_t max_depth;      /* default 32 */
    int32_t num_threads;    /* default 4  */
    float russian_roulette; /* default 0.8 */
    TkSppmParam sppm;       /* valid when type == "sppm" */
} TkTracerConfig;

/* Camera block: parse_camera, src/TkLoader.cpp:213-226. */
typedef struct TkCameraConfig {
    float fov_degrees, near_clip, far_clip;
    float theta, phi, r;
} TkCameraConfig;

const char* tkhost_last_error(void);

/* load_scene, src/TkLoader.cpp:364-464.  Mesh file names are resolved
 * relative to the directory of the .scene file first, then to the working
 * directory (the reference only does the latter).  Returns 0 on success. */
int tkhost_load_scene(const char* scene_path, tkhost_scene** out);
void tkhost_free_scene(tkhost_scene* s);

/* A procedural scene: Cornell shell + glass and mirror spheres + a diffuse
 * height field of quads x quads cells replacing the floor
 * (y = -2.2 + 0.15 sin(7x) cos(5z) + 0.02 (pcg32(seed,i) - 0.5)).
 * model_dir must hold cornellbox/{floor,left,right,light}.obj. */
int tkhost_make_synthetic(const char* model_dir, int quads, uint32_t seed, tkhost_scene** out);

const TkObject* tkhost_objects(const tkhost_scene* s, int* n);
const TkTriangle* tkhost_triangles(const tkhost_scene* s, int* n);
const TkSphere* tkhost_spheres(const tkhost_scene* s, int* n);
const TkMaterial* tkhost_materials(const tkhost_scene* s, int* n);
const TkLight* tkhost_lights(const tkhost_scene* s, int* n);

void tkhost_tracer_config(const tkhost_scene* s, TkTracerConfig* out);
void tkhost_camera_config(const tkhost_scene* s, TkCameraConfig* out);
void tkhost_image_config(const tkhost_scene* s, int* width, int* height, int* windows,
                         char* outfilename, size_t outfilename_cap);

/* GPUs one render of this scene is spread over: the optional <gpus i="N"/>
 * element of the scene root (an addition to the format parsed at
 * src/TkLoader.cpp:387-390; default 1), or what tkhost_set_gpu_count set.
 * With more ranks than GPUs the ranks share the GPUs round-robin. */
int tkhost_gpu_count(const tkhost_scene* s);
void tkhost_set_gpu_count(tkhost_scene* s, int gpus);

/* CameraInfo::rotate(0,0) + setTranslation + perspective + Camera ctor
 * (src/TkRenderer.cpp:333-376, src/main.cpp:39-46, src/core/camera.cpp:10-20):
 * the two row-major matrices tkhip_set_camera takes. */
void tkhost_camera_matrices(const TkCameraConfig* cam, int width, int height, float c2w[16],
                            float raster2cam[16]);

/* Headless render of a loaded scene on one GPU, as src/main.cpp:37-58 does:
 * runs numRecurse SPPM iterations through libtkhip and writes a binary PPM.
 * width/height <= 0 keep the scene's values.  seconds_out may be null. */
int tkhost_render(tkhost_scene* s, int device, int width, int height, const char* out_ppm,
                  uint64_t seed, double* seconds_out);

/* The windowed flow without a window (src/TkRenderer.cpp:186-222 driving
 * RayTracer::updateScreen, src/core/TkRayTracer.cpp:54-73): startRaytracing(),
 * one updateScreen() per frame, the frame buffer refreshed on the device after
 * every iteration.  After save_after steps (0 <= save_after < iterations) the
 * image is saved mid-render to mid_ppm (may be null) and the RGBA8 frame buffer
 * copied to frame_rgba (width*height words, may be null); the final image goes
 * to out_ppm. */
int tkhost_render_windowed(tkhost_scene* s, int device, int width, int height, const char* out_ppm,
                           uint64_t seed, int save_after, const char* mid_ppm, uint32_t* frame_rgba);

/* Writes a binary P6 PPM (src/tracer/TkSppmTracer.cpp:530-535). */
int tkhost_write_ppm(const char* path, int width, int height, const uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif
