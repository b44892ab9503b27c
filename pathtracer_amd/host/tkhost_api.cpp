// tkhost_api.cpp — the flat C view of the loader (include/tkhost.h).
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>

#include "scene_loader.h"
#include "tkhost.h"

namespace {
thread_local std::string g_error;
}

extern "C" {

const char* tkhost_last_error(void) { return g_error.c_str(); }

int tkhost_load_scene(const char* scene_path, tkhost_scene** out) {
    if (!scene_path || !out) {
        g_error = "tkhost_load_scene: null argument";
        return -1;
    }
    tkhost_scene* s = new tkhost_scene();
    try {
        tk::loadScene(&s->config, scene_path);
    } catch (const std::exception& e) {
        g_error = e.what();
        delete s;
        return -1;
    }
    *out = s;
    return 0;
}

int tkhost_make_synthetic(const char* model_dir, int quads, uint32_t seed, tkhost_scene** out) {
    if (!model_dir || !out || quads < 1 || quads > 8192) {
        g_error = "tkhost_make_synthetic: bad argument";
        return -1;
    }
    tkhost_scene* s = new tkhost_scene();
    try {
        tk::makeSyntheticScene(&s->config, model_dir, quads, seed);
    } catch (const std::exception& e) {
        g_error = e.what();
        delete s;
        return -1;
    }
    *out = s;
    return 0;
}

void tkhost_free_scene(tkhost_scene* s) { delete s; }

const TkObject* tkhost_objects(const tkhost_scene* s, int* n) {
    *n = int(s->config.scene.objects.size());
    return s->config.scene.objects.data();
}
const TkTriangle* tkhost_triangles(const tkhost_scene* s, int* n) {
    *n = int(s->config.scene.tris.size());
    return s->config.scene.tris.data();
}
const TkSphere* tkhost_spheres(const tkhost_scene* s, int* n) {
    *n = int(s->config.scene.spheres.size());
    return s->config.scene.spheres.data();
}
const TkMaterial* tkhost_materials(const tkhost_scene* s, int* n) {
    *n = int(s->config.scene.materials.size());
    return s->config.scene.materials.data();
}
const TkLight* tkhost_lights(const tkhost_scene* s, int* n) {
    *n = int(s->config.scene.lights.size());
    return s->config.scene.lights.data();
}

void tkhost_tracer_config(const tkhost_scene* s, TkTracerConfig* out) { *out = s->config.tracer; }
void tkhost_camera_config(const tkhost_scene* s, TkCameraConfig* out) { *out = s->config.camera; }
int tkhost_gpu_count(const tkhost_scene* s) { return s ? s->config.gpus : 0; }
void tkhost_set_gpu_count(tkhost_scene* s, int gpus) {
    if (s && gpus >= 1 && gpus <= 64) s->config.gpus = gpus;
}

void tkhost_image_config(const tkhost_scene* s, int* width, int* height, int* windows, char* outfilename,
                         size_t cap) {
    if (width) *width = s->config.width;
    if (height) *height = s->config.height;
    if (windows) *windows = s->config.isWindows;
    if (outfilename && cap) snprintf(outfilename, cap, "%s", s->config.outfilename.c_str());
}

void tkhost_camera_matrices(const TkCameraConfig* cam, int width, int height, float c2w[16],
                            float raster2cam[16]) {
    tk::cameraMatrices(*cam, width, height, c2w, raster2cam);
}

int tkhost_write_ppm(const char* path, int width, int height, const uint8_t* rgb) {
    FILE* fp = fopen(path, "wb");
    if (!fp) {
        g_error = std::string("cannot write ") + path;
        return -1;
    }
    fprintf(fp, "P6\n%d %d\n255\n", width, height);
    fwrite(rgb, 1, size_t(width) * size_t(height) * 3, fp);
    fclose(fp);
    return 0;
}

}  // extern "C"

namespace tk {
void setHostError(const std::string& e) { g_error = e; }
}
