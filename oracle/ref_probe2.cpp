// ref_probe2.cpp — TEST INFRASTRUCTURE, build container only.
//
// Second prober over the upstream CPU renderer's own translation units (see
// oracle/build_ref.sh): the parts of the SPPM path that draw unseeded random
// numbers upstream, run with those numbers fed in, and whole renders.
//
//   ref_probe2 paths  <in_dir> <out_dir> <scene> <tag>   SppmTracer::tracePhoton / traceCausticPhoton per Halton index
//   ref_probe2 li     <in_dir> <out_dir> <scene> <tag>   PathTracer::Li for one unjittered sample of every pixel
//   ref_probe2 hsppm  <in_dir> <out_dir> <scene> <tag>   tk::SPPM (src/TkSPPM.cpp): hash grid, ray / photon / renew passes
//   ref_probe2 render <in_dir> <out_dir> <scene> <tag> <W> <H> <recurse> <gsize> <csize> <runs> <threads>
//                                                        SppmTracer::render, the float image of every run
//
// Feeding: build_ref.sh patches get_random_float (src/TkMath.h:312-319) in its
// scratch copy so that it returns tk::probe_draw_hook() when that pointer is
// set and runs its own body otherwise.  "paths", "li" and "hsppm" set the hook
// to hand out, call by call, the numbers of <in_dir>/draws_*.f32; "render"
// leaves it unset (upstream's own std::random_device draws).
// cwd must be assets/scenes (the scene files' "../models/..." paths).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <algorithm>
#include <array>
#include <cmath>
#include <condition_variable>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <unistd.h>
#include <fcntl.h>

#define private public
#define protected public
#include "Photonmap.h"
#include "TkSppmTracer.h"
#include "TkPathTracer.h"
#include "Scene.hpp"
#include "camera.h"
#include "sampler.h"
#include "Material.hpp"
#include "AreaLight.hpp"
#include "Object.hpp"
#include "Sphere.hpp"
#include "Triangle.hpp"
// the second SPPM class is not part of upstream's build: its translation unit is taken in whole, which also
// makes the file-local SpatialHashGrids visible here
#include "TkSPPM.cpp"
#undef private
#undef protected
#include "TkLoader.h"
#include "TkFilm.h"
#include "TkFilter.h"
#include "TkLowdiscrepancy.h"

using namespace tk;

namespace tk { float (*probe_draw_hook)() = nullptr; }

// ------------------------------------------------------------------ npy out
static std::string g_out;
static void save_npy(const std::string& name, const char* descr, const std::vector<size_t>& shape, const void* data, size_t bytes) {
    std::string path = g_out + "/" + name + ".npy";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    std::string sh = "(";
    for (size_t i = 0; i < shape.size(); ++i) sh += std::to_string(shape[i]) + ",";
    sh += ")";
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + sh + ", }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    uint16_t hl = (uint16_t)hdr.size();
    fwrite(magic, 1, 8, f);
    fwrite(&hl, 2, 1, f);
    fwrite(hdr.data(), 1, hdr.size(), f);
    fwrite(data, 1, bytes, f);
    fclose(f);
}
template <class T> struct Descr;
template <> struct Descr<float> { static const char* s() { return "<f4"; } };
template <> struct Descr<double> { static const char* s() { return "<f8"; } };
template <> struct Descr<int32_t> { static const char* s() { return "<i4"; } };
template <> struct Descr<int64_t> { static const char* s() { return "<i8"; } };
template <> struct Descr<uint64_t> { static const char* s() { return "<u8"; } };
template <> struct Descr<uint8_t> { static const char* s() { return "|u1"; } };
template <class T>
static void save(const std::string& name, const std::vector<T>& v, std::vector<size_t> shape) {
    size_t n = 1;
    for (size_t s : shape) n *= s;
    if (n != v.size()) { fprintf(stderr, "%s: shape does not match %zu values\n", name.c_str(), v.size()); exit(2); }
    save_npy(name, Descr<T>::s(), shape, v.data(), v.size() * sizeof(T));
}
template <class T>
static std::vector<T> load_raw(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)sz / sizeof(T));
    if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) { fprintf(stderr, "short read %s\n", path.c_str()); exit(2); }
    fclose(f);
    return v;
}
static void push3(std::vector<float>& o, const Vector3f& v) { o.push_back(v.x); o.push_back(v.y); o.push_back(v.z); }
static void pushs(std::vector<float>& o, const Spectrum& s) { o.push_back(s.r); o.push_back(s.g); o.push_back(s.b); }

static Matrix4 mat_from(const float* m) {
    return Matrix4(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
}
static Config* load(const char* file) {
    Config* cfg = new Config();
    cfg->scene = new Scene();
    if (!load_scene(cfg, file)) { fprintf(stderr, "load_scene(%s) failed\n", file); exit(3); }
    return cfg;
}

// ------------------------------------------------------------- fed draws
static const float* g_draws = nullptr;
static int g_draw_n = 0, g_draw_at = 0, g_draw_overrun = 0;
static float next_draw() {
    if (g_draw_at < g_draw_n) return g_draws[g_draw_at++];
    ++g_draw_overrun;
    ++g_draw_at;
    return 2.0f;          // beyond the table: every roulette test fails (and the table was too short: reported)
}
static void feed(const float* row, int n) { g_draws = row; g_draw_n = n; g_draw_at = 0; probe_draw_hook = next_draw; }

// -------------------------------------------------- light paths, one by one
// SppmTracer::tracePhoton(H) / traceCausticPhoton(H) for H = 0 .. n-1 with the unseeded draws of a path
// (the triangle pick inside AreaLight::sample_Le, then one Russian-roulette number per bounce) taken from
// row H of draws_<kind>.f32; what each call stored, in call order.
static void probe_paths(const std::string& in, const char* scene_file, const std::string& tag) {
    for (int kind = 0; kind < 2; ++kind) {
        const std::string dn = in + "/draws_" + tag + (kind == 0 ? "_global.f32" : "_caustic.f32");
        std::vector<float> draws = load_raw<float>(dn);
        const int per = 65;                                  // pick + up to 64 bounces
        const int n = (int)(draws.size() / per);
        if (n == 0) continue;
        Config* cfg = load(scene_file);
        delete cfg->tracer;
        SPPMParam prm;
        prm.numRecurse = 4; prm.globalSample = 100; prm.causticSample = 50; prm.radius2 = 0.08f;
        prm.globalSize = kind == 0 ? n * 20 + 1 : 1;         // never full: the loop under test does not stop early
        prm.causticSize = kind == 1 ? n + 1 : 1;
        SppmTracer* tr = new SppmTracer(1, 20, 1, 0.8f, prm);
        tr->setScene(cfg->scene);
        tr->storedGlobalPhotons = tr->storedCausticPhotons = 0;
        PhotonMap& map = kind == 0 ? tr->globalPhotonMap : tr->causticPhotonMap;
        s32& stored = kind == 0 ? tr->storedGlobalPhotons : tr->storedCausticPhotons;
        std::vector<int32_t> count, used;
        std::vector<float> pos, pow;
        std::vector<uint8_t> dir;
        for (int h = 0; h < n; ++h) {
            feed(&draws[(size_t)h * per], per);
            const s32 before = stored;
            if (kind == 0) tr->tracePhoton((u64)h);
            else tr->traceCausticPhoton((u64)h);
            count.push_back(stored - before);
            used.push_back(g_draw_at);
            for (s32 i = before; i < stored; ++i) {
                const Photon& p = map.all_raw_photons[i];
                push3(pos, p.pos); pushs(pow, p.power);
                dir.push_back(p.theta); dir.push_back(p.phi);
            }
        }
        probe_draw_hook = nullptr;
        if (g_draw_overrun) { fprintf(stderr, "paths: draw table too short (%d calls beyond it)\n", g_draw_overrun); exit(4); }
        const std::string t = "paths_" + tag + (kind == 0 ? "_global" : "_caustic");
        const size_t np = dir.size() / 2;
        save(t + "_count", count, {(size_t)n}); save(t + "_draws_used", used, {(size_t)n});
        save(t + "_pos", pos, {np, 3}); save(t + "_power", pow, {np, 3}); save(t + "_dir", dir, {np, 2});
        save(t + "_draws", draws, {(size_t)n, (size_t)per});
        delete tr;
        delete cfg->scene;
        delete cfg;
    }
}

// ------------------------------------------------------------ PathTracer::Li
// One sample (index 0) of every pixel of a W x W image with an unjittered sampler of spp strata, the
// Russian-roulette number of bounce b of pixel p taken from draws_<tag>_li.f32[p][b].
static void probe_li(const std::string& in, const char* scene_file, const std::string& tag, const std::vector<float>& cam) {
    const int W = 32, spp = 4, depth = 20;
    std::vector<float> draws = load_raw<float>(in + "/draws_" + tag + "_li.f32");
    const int per = depth;
    if ((int)draws.size() != W * W * per) { fprintf(stderr, "li: draws_%s_li.f32 must hold %d numbers\n", tag.c_str(), W * W * per); exit(2); }
    Config* cfg = load(scene_file);
    delete cfg->tracer;
    PathTracer* tr = new PathTracer(spp, depth, 1, 0.8f);
    Matrix4 c2w = mat_from(&cam[0]), c2s = mat_from(&cam[16]);
    BoxFilter filter(0.5, 0.5);
    Film film(Point2i(W, W), Vector2f(0), Vector2f(1.0f, 1.0f), &filter);
    Camera camera(&c2w, c2s, 0.00, 2.0, &film);
    tr->setScene(cfg->scene);
    tr->setCamera(&camera);
    // the strata of the tracer's own sampler (PathTracer ctor), without jitter
    StratifiedSampler* own = static_cast<StratifiedSampler*>(tr->mSampler);
    StratifiedSampler sampler(own->xPixelSamples, own->yPixelSamples, false, 0);
    std::vector<float> L, sample;
    std::vector<int32_t> used;
    for (int y = 0; y < W; ++y)
        for (int x = 0; x < W; ++x) {
            const int p = x + y * W;
            feed(&draws[(size_t)p * per], per);
            sampler.startPixelSample(Point2i(x, y), 0, 0);
            Vector2f cs = sampler.get2D() + Vector2f(x, y);
            Ray r;
            camera.generateRay(cs, sampler.get2D(), &r);
            Spectrum li = tr->Li(r, sampler, 0);
            pushs(L, li);
            sample.push_back(cs.x); sample.push_back(cs.y);
            used.push_back(g_draw_at);
        }
    probe_draw_hook = nullptr;
    if (g_draw_overrun) { fprintf(stderr, "li: draw table too short\n"); exit(4); }
    const std::string t = "li_" + tag;
    save(t + "_L", L, {(size_t)W * W, 3}); save(t + "_sample", sample, {(size_t)W * W, 2}); save(t + "_draws_used", used, {(size_t)W * W});
    save(t + "_draws", draws, {(size_t)W * W, (size_t)per});
    std::vector<int32_t> meta = {W, spp, depth, own->xPixelSamples, own->yPixelSamples};
    save(t + "_meta", meta, {meta.size()});
}

// ------------------------------------------------------------------ tk::SPPM
// The hash-grid variant pass by pass, single-threaded, on a W x W image: ray pass with an unjittered sampler,
// grid rebuild, photon pass (one path per call so that its draws can be fed: row i of
// draws_<tag>_hsppm_it<k>.f32), renew pass.
static void probe_hsppm(const std::string& in, const char* scene_file, const std::string& tag, const std::vector<float>& cam) {
    const int W = 40, iters = 2, per = 21;
    std::vector<float> prm_in = load_raw<float>(in + "/hsppm_" + tag + ".f32");   // photons per iteration, initial radius
    const int photons = (int)prm_in[0];
    Config* cfg = load(scene_file);
    delete cfg->tracer;
    SPPMParam prm;
    prm.numRecurse = 4; prm.globalSize = photons; prm.globalSample = 100; prm.causticSize = 0; prm.causticSample = 50;
    prm.radius2 = prm_in[1];
    SPPM* tr = new SPPM(1, 20, 1, 0.8f, prm);
    Matrix4 c2w = mat_from(&cam[0]), c2s = mat_from(&cam[16]);
    BoxFilter filter(0.5, 0.5);
    Film film(Point2i(W, W), Vector2f(0), Vector2f(1.0f, 1.0f), &filter);
    Camera camera(&c2w, c2s, 0.00, 2.0, &film);
    tr->setScene(cfg->scene);
    tr->setCamera(&camera);
    // SPPM::startRaytracing without the worker threads (src/TkSPPM.cpp:274-282)
    tr->mPixels.resize((size_t)W * W, PixelData(tr->radius));
    tr->mHashGrids = new SpatialHashGrids(W * W);
    tr->curIteration = 0;
    int xs = 0, ys = 0;
    while (++xs * ++ys < prm.numRecurse) {}
    StratifiedSampler sampler(xs, ys, false, 0);
    for (int it = 0; it < iters; ++it) {
        const std::string t = "hsppm_" + tag + "_it" + std::to_string(it);
        tr->curIteration = it;
        tr->traceTile(Point2i(0, 0), Point2i(W, W), sampler);
        std::vector<float> vp, thr, ld;
        for (const PixelData& p : tr->mPixels) { push3(vp, p.isect.p); pushs(thr, p.throughput); pushs(ld, p.Ld); }
        save(t + "_visible_p", vp, {(size_t)W * W, 3}); save(t + "_throughput", thr, {(size_t)W * W, 3});
        save(t + "_Ld", ld, {(size_t)W * W, 3});
        SpatialHashGrids* g = tr->mHashGrids;
        g->rebuild(tr->mPixels);
        std::vector<float> gb;
        push3(gb, g->mGridsBounds.pMin); push3(gb, g->mGridsBounds.pMax);
        std::vector<int32_t> res = {g->mGridsRes[0], g->mGridsRes[1], g->mGridsRes[2]}, start(1, 0), items;
        for (size_t b = 0; b < g->mGrids.size(); ++b) {
            for (PixelData* p : g->mGrids[b]) items.push_back((int32_t)(p - &tr->mPixels[0]));
            start.push_back((int32_t)items.size());
        }
        save(t + "_grid_bounds", gb, {2, 3}); save(t + "_grid_res", res, {3});
        save(t + "_bucket_start", start, {start.size()}); save(t + "_bucket_items", items, {items.size()});
        if (it == 0) {   // hash and worldToGrid on points around the grid
            std::mt19937 rg(77);
            auto u = [&]() { return (float)(rg() >> 8) * (1.0f / 16777216.0f); };
            std::vector<float> pts;
            std::vector<int32_t> cell, inb, hv;
            const Vector3f lo = g->mGridsBounds.pMin, hi = g->mGridsBounds.pMax;
            for (int i = 0; i < 2048; ++i) {
                Vector3f p(lo.x + (hi.x - lo.x) * (1.2f * u() - 0.1f), lo.y + (hi.y - lo.y) * (1.2f * u() - 0.1f),
                           lo.z + (hi.z - lo.z) * (1.2f * u() - 0.1f));
                s32 pi[3];
                const bool ok = g->worldToGrid(p, pi);
                push3(pts, p);
                cell.push_back(pi[0]); cell.push_back(pi[1]); cell.push_back(pi[2]);
                inb.push_back(ok ? 1 : 0);
                hv.push_back(g->hash(pi[0], pi[1], pi[2]));
            }
            save(t + "_w2g_points", pts, {2048, 3}); save(t + "_w2g_cell", cell, {2048, 3});
            save(t + "_w2g_inside", inb, {2048}); save(t + "_w2g_hash", hv, {2048});
        }
        std::vector<float> draws = load_raw<float>(in + "/draws_" + tag + "_hsppm_it" + std::to_string(it) + ".f32");
        if ((int)draws.size() != photons * per) { fprintf(stderr, "hsppm: draws of iteration %d must hold %d numbers\n", it, photons * per); exit(2); }
        std::vector<int32_t> used;
        for (int i = 0; i < photons; ++i) {
            feed(&draws[(size_t)i * per], per);
            tr->tracePhoton(i, i + 1);
            used.push_back(g_draw_at);
        }
        probe_draw_hook = nullptr;
        if (g_draw_overrun) { fprintf(stderr, "hsppm: draw table too short\n"); exit(4); }
        std::vector<float> phi;
        std::vector<int32_t> m;
        for (const PixelData& p : tr->mPixels) { pushs(phi, p.Phi); m.push_back(p.M); }
        save(t + "_Phi", phi, {(size_t)W * W, 3}); save(t + "_M", m, {(size_t)W * W});
        save(t + "_draws", draws, {(size_t)photons, (size_t)per}); save(t + "_draws_used", used, {(size_t)photons});
        tr->updatePixel(0, W * W);
        std::vector<float> tau, nn, rad;
        for (const PixelData& p : tr->mPixels) { pushs(tau, p.Tau); nn.push_back(p.N); rad.push_back(p.radius); }
        save(t + "_Tau", tau, {(size_t)W * W, 3}); save(t + "_N", nn, {(size_t)W * W}); save(t + "_radius", rad, {(size_t)W * W});
    }
    std::vector<int32_t> meta = {W, iters, photons, per, tr->numIteration, tr->mSpp};
    save("hsppm_" + tag + "_meta", meta, {meta.size()});
    std::vector<float> r0 = {tr->radius};
    save("hsppm_" + tag + "_radius0", r0, {1});
}

// -------------------------------------------------------- SppmTracer::render
// Whole renders with upstream's own threads and its own unseeded random numbers: every run's image before
// gamma and quantisation (the Spectrum expression of src/tracer/TkSppmTracer.cpp:522-524), as floats.
static void probe_render(const char* scene_file, const std::string& tag, const std::vector<float>& cam, int W, int H, int recurse,
                         int gsize, int csize, int runs, int threads) {
    std::vector<float> images;
    std::vector<int64_t> counters;
    for (int run = 0; run < runs; ++run) {
        Config* cfg = load(scene_file);
        delete cfg->tracer;
        SPPMParam prm;
        prm.numRecurse = recurse; prm.globalSize = gsize; prm.globalSample = 100; prm.causticSize = csize; prm.causticSample = 50;
        prm.radius2 = 0.08f;
        SppmTracer* tr = new SppmTracer(1, 20, threads, 0.8f, prm);
        Matrix4 c2w = mat_from(&cam[0]), c2s = mat_from(&cam[16]);
        BoxFilter filter(0.5, 0.5);
        Film film(Point2i(W, H), Vector2f(0), Vector2f(1.0f, 1.0f), &filter);
        Camera camera(&c2w, c2s, 0.00, 2.0, &film);
        tr->setScene(cfg->scene);
        tr->setCamera(&camera);
        // render() prints a line per photon to stderr: not to the terminal
        fflush(stderr);
        const int keep = dup(2), nul = open("/dev/null", O_WRONLY);
        dup2(nul, 2);
        const std::string ppm = g_out + "/render_" + tag + "_run" + std::to_string(run) + ".ppm";
        tr->render(ppm);
        fflush(stderr);
        dup2(keep, 2);
        close(keep); close(nul);
        for (const SPPMPixel& p : tr->pp) {
            Spectrum c = p.Ld / tr->numRecurse + (tr->storedCausticPhotons > 0 ?
                p.causticFlux / (Math::pi * p.causticRadius2 * tr->shootCausticPhotons) : Spectrum::black) +
                p.globalFlux / (Math::pi * p.globalRadius2 * tr->shootGlobalPhotons);
            pushs(images, c);
        }
        counters.push_back(tr->shootGlobalPhotons); counters.push_back(tr->shootCausticPhotons); counters.push_back(tr->numRecurse);
        fprintf(stderr, "render %s run %d: shoot %d / %d\n", tag.c_str(), run, tr->shootGlobalPhotons, tr->shootCausticPhotons);
        remove(ppm.c_str());
        delete tr;
        delete cfg->scene;
        delete cfg;
    }
    save("render_" + tag + "_images", images, {(size_t)runs, (size_t)H, (size_t)W, 3});
    save("render_" + tag + "_counters", counters, {(size_t)runs, 3});
    std::vector<int32_t> meta = {W, H, recurse, gsize, csize, runs, threads};
    save("render_" + tag + "_meta", meta, {meta.size()});
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: ref_probe2 <paths|li|hsppm|render> <in_dir> <out_dir> <scene> <tag> [...]\n"); return 1; }
    const std::string mode = argv[1], in = argv[2];
    g_out = argv[3];
    const char* scene = argv[4];
    const std::string tag = argv[5];
    std::vector<float> cam = load_raw<float>(in + "/camera.f32");
    if (cam.size() != 32) { fprintf(stderr, "camera.f32 must hold 32 floats\n"); return 1; }
    if (mode == "paths") probe_paths(in, scene, tag);
    else if (mode == "li") probe_li(in, scene, tag, cam);
    else if (mode == "hsppm") probe_hsppm(in, scene, tag, cam);
    else if (mode == "render") {
        if (argc < 13) { fprintf(stderr, "render: W H recurse gsize csize runs threads\n"); return 1; }
        probe_render(scene, tag, cam, atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), atoi(argv[11]), atoi(argv[12]));
    } else { fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 1; }
    fprintf(stderr, "ref_probe2 %s %s: done\n", mode.c_str(), tag.c_str());
    return 0;
}
