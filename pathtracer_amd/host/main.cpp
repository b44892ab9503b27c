// main.cpp — headless command line of the renderer: raytracer <file.scene>.
// Same flow as the reference's main (src/main.cpp:16-64) without the window
// branch: load the scene, build camera and tracer from it, time render().
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "ray_tracer.h"

using namespace tk;

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "Please input the scene file!\n"
                        "usage: raytracer <file.scene> [--device N] [--gpus N] [--width W] [--height H] [--seed S] [--out file.ppm]\n");
        return 0;
    }
    std::string inputfile = argv[1];
    int device = 0, width = 0, height = 0, gpus = 0;
    uint64_t seed = 0x5EED;
    std::string out;
    for (int i = 2; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--device")) device = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--gpus")) gpus = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--width")) width = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--height")) height = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--seed")) seed = strtoull(argv[i + 1], nullptr, 0);
        else if (!strcmp(argv[i], "--out")) out = argv[i + 1];
        else {
            fprintf(stderr, "unknown option %s\n", argv[i]);
            return 2;
        }
    }
    Config config;
    try {
        loadScene(&config, inputfile);
    } catch (const std::exception& e) {
        std::cout << e.what() << "\n";
        return 1;
    }
    if (config.isWindows) fprintf(stderr, "[Tracer] windows=1 ignored: this build renders headless\n");
    if (width <= 0) width = config.width;
    if (height <= 0) height = config.height;
    std::string err;
    std::unique_ptr<RayTracer> tracer = createTracer(config.tracer, &err);
    if (!tracer) {
        std::cout << err << "\n";
        return 1;
    }
    Camera camera(config.camera, width, height);
    tracer->setDevice(device);
    tracer->setGpuCount(gpus > 0 ? gpus : config.gpus);   // --gpus overrides the scene's <gpus i="N"/>
    tracer->setSeed(seed);
    tracer->setScene(&config.scene);
    tracer->setCamera(&camera);

    auto start = std::chrono::system_clock::now();
    tracer->render(out.empty() ? config.outfilename : out);
    auto stop = std::chrono::system_clock::now();
    if (!tracer->done()) return 1;

    std::cout << "Render complete: \n";
    std::cout << "Time taken: " << std::chrono::duration_cast<std::chrono::hours>(stop - start).count() << " hours\n";
    std::cout << "          : " << std::chrono::duration_cast<std::chrono::minutes>(stop - start).count() << " minutes\n";
    std::cout << "          : " << std::chrono::duration<double>(stop - start).count() << " seconds\n";
    return 0;
}
