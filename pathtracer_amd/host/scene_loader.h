// scene_loader.h — .scene / OBJ / PLY loader producing the POD scene the
// device backend consumes.  Same XML schema and defaults as the reference's
// load_scene (src/TkLoader.cpp:364-464); the scene graph classes of the
// reference (Scene, Object, MeshTriangle, Material, AreaLight) are replaced by
// flat arrays because their parameters are private there and the GPU needs
// them contiguous.
#pragma once
#include <string>
#include <vector>

#include "tkhost.h"

namespace tk {

struct SceneData {
    std::vector<TkObject> objects;
    std::vector<TkTriangle> tris;
    std::vector<TkSphere> spheres;
    std::vector<TkMaterial> materials;
    std::vector<TkLight> lights;
};

// Config of the reference (src/TkConfigure.h:39-48) with the scene flattened.
struct Config {
    SceneData scene;
    TkCameraConfig camera{};
    TkTracerConfig tracer{};
    int isWindows = 0;
    std::string outfilename;
    int width = 784;
    int height = 784;
    int gpus = 1;   // <gpus i="N"/> in the scene root: GPUs one render is spread over (not in the reference's format)
};

// Row-major 4x4, m[row*4+col] (src/Matrix4.h:11-15).
struct Mat4 {
    float m[16];
};

Mat4 makeTransform(const float translate[3], const float scale[3], const float euler[3]);
void transformPos(const Mat4& m, const float p[3], float out[3]);

// Appends the triangles of an OBJ or PLY file, transformed by `xf`.
// Throws std::runtime_error on I/O or format errors.
void loadMeshFile(const std::string& path, const Mat4& xf, std::vector<TkTriangle>* out);

// Throws std::runtime_error with a "file:line: reason" message.
void loadScene(Config* config, const std::string& filename);

void makeSyntheticScene(Config* config, const std::string& modelDir, int quads, uint32_t seed);

void cameraMatrices(const TkCameraConfig& cam, int width, int height, float c2w[16], float raster2cam[16]);

}  // namespace tk

// The opaque scene handle of the C view (include/tkhost.h).
struct tkhost_scene {
    tk::Config config;
};
