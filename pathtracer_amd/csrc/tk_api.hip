// tk_api.hip — context lifecycle, scene / camera upload and the small
// kernel-level entry points of the C ABI (include/tkhip.h).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "photon_map.h"
#include "tk_bvh_device.h"

namespace tkd {

static thread_local char g_error[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

// PhotonMap ctor tables, src/tracer/Photonmap.cpp:26-33: computed in double,
// stored as float.
int upload_direction_tables(tkhip_ctx* ctx) {
    if (ctx->b_dirtab.p) return TKHIP_OK;
    TK_TRY(ctx->b_dirtab.ensure(1024 * 4));
    float tab[1024];
    const float pi = float(4.0 * std::atan(1.0));
    for (int i = 0; i < 256; ++i) {
        double radians = double(i) * pi / 255.0;
        tab[i] = float(std::cos(radians));
        tab[256 + i] = float(std::sin(radians));
        tab[512 + i] = float(std::cos(2.0 * radians));
        tab[768 + i] = float(std::sin(2.0 * radians));
    }
    TK_HIP(hipMemcpy(ctx->b_dirtab.p, tab, sizeof(tab), hipMemcpyHostToDevice));
    return TKHIP_OK;
}

__global__ void __launch_bounds__(256) k_trace_rays(DScene sc, int64_t n, const float* org, const float* dir,
                                                    const float* tmax, int32_t* prim, float* t, float* p, float* nrm,
                                                    float* perr) {
    __shared__ int s_stack[kStackDepth * kTraceBlock];
    LdsStack stk{&s_stack[threadIdx.x]};
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 o = ld3(org + 3 * i), d = ld3(dir + 3 * i);
    double tm = tmax ? (double)tmax[i] : 1.7976931348623157e308;
    Hit h = trace_ray(sc, o, d, tm, false, stk);
    prim[i] = h.prim;
    SurfacePoint sp{};
    if (h.prim >= 0) sp = surface_at(sc, h, o, d);
    if (t) t[i] = h.prim >= 0 ? (float)h.t : 0.f;
    if (p) { p[3 * i] = sp.p.x; p[3 * i + 1] = sp.p.y; p[3 * i + 2] = sp.p.z; }
    if (nrm) { nrm[3 * i] = sp.n.x; nrm[3 * i + 1] = sp.n.y; nrm[3 * i + 2] = sp.n.z; }
    if (perr) { perr[3 * i] = sp.perr.x; perr[3 * i + 1] = sp.perr.y; perr[3 * i + 2] = sp.perr.z; }
}

// The same through the wide walk (eight lanes per ray, eight rays of a wave at a time).
__global__ void __launch_bounds__(256) k_trace_rays_wide(DScene sc, int64_t n, const float* org, const float* dir,
                                                         const float* tmax, int32_t* prim, float* t, float* p, float* nrm,
                                                         float* perr) {
    __shared__ int s_stack[kStackDepth * kTraceBlock];
    LdsStack stk{&s_stack[threadIdx.x]};
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool want = i < n;
    F3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
    double tm = 1.7976931348623157e308;
    if (want) { o = ld3(org + 3 * i); d = ld3(dir + 3 * i); if (tmax) tm = (double)tmax[i]; }
    Hit h = walk_wide(sc, want, o, d, tm, false, stk);
    if (!want) return;
    prim[i] = h.prim;
    SurfacePoint sp{};
    if (h.prim >= 0) sp = surface_at(sc, h, o, d);
    if (t) t[i] = h.prim >= 0 ? (float)h.t : 0.f;
    if (p) { p[3 * i] = sp.p.x; p[3 * i + 1] = sp.p.y; p[3 * i + 2] = sp.p.z; }
    if (nrm) { nrm[3 * i] = sp.n.x; nrm[3 * i + 1] = sp.n.y; nrm[3 * i + 2] = sp.n.z; }
    if (perr) { perr[3 * i] = sp.perr.x; perr[3 * i + 1] = sp.perr.y; perr[3 * i + 2] = sp.perr.z; }
}

// Scene::uniformSampleOneLight (src/core/Scene.cpp:43-61) at the first hit of each ray, with the four
// numbers the sampler would deliver given by the caller: the hit, the light sample, its shadow ray.
__global__ void __launch_bounds__(256) k_sample_one_light(DScene sc, int64_t n, const float* org, const float* dir,
                                                          const float* u, int32_t* prim, float* ld) {
    __shared__ int s_stack[kStackDepth * kTraceBlock];
    LdsStack stk{&s_stack[threadIdx.x]};
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F3 o = ld3(org + 3 * i), d = ld3(dir + 3 * i);
    const Hit h = trace_ray(sc, o, d, 1.7976931348623157e308, false, stk);
    prim[i] = h.prim;
    F3 out = mk3(0, 0, 0);
    if (h.prim >= 0) {
        const SurfacePoint sp = surface_at(sc, h, o, d);
        const LightSample ls = sample_one_light(sc, sp, sc.mats[sp.mat], u[4 * i], u[4 * i + 1], u[4 * i + 2], u[4 * i + 3]);
        if (ls.want_shadow) {
            const Hit sh = trace_ray(sc, ls.o, ls.d, (double)kOneMinusEps, true, stk);
            if (sh.prim < 0) out = ls.ld;
        }
    }
    ld[3 * i] = out.x; ld[3 * i + 1] = out.y; ld[3 * i + 2] = out.z;
}

__global__ void k_radical_inverse(int64_t n, const int32_t* dim, const uint64_t* index, float* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = radical_inverse(dim[i], index[i]);
}

__global__ void k_camera_rays(DCamera cam, int64_t n, const float* raster, float* org, float* dir) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 o, d;
    camera_ray(cam, raster[2 * i], raster[2 * i + 1], &o, &d);
    org[3 * i] = o.x; org[3 * i + 1] = o.y; org[3 * i + 2] = o.z;
    dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
}

template <class T>
static int upload(DevBuf& b, const std::vector<T>& v, hipStream_t s) {
    TK_TRY(b.ensure(std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) TK_HIP(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return TKHIP_OK;
}

}  // namespace tkd

using namespace tkd;

extern "C" {

const char* tkhip_last_error(void) { return tkd::g_error; }

const char* tkhip_version(void) { return "tkhip 0.1 gfx950"; }

int tkhip_create(int device, tkhip_ctx** out) {
    TK_REQUIRE(out, TKHIP_E_ARG, "tkhip_create: null output pointer");
    int count = 0;
    TK_HIP(hipGetDeviceCount(&count));
    TK_REQUIRE(device >= 0 && device < count, TKHIP_E_ARG, "tkhip_create: device %d of %d", device, count);
    TK_HIP(hipSetDevice(device));
    tkhip_ctx* ctx = new tkhip_ctx();
    ctx->device = device;
    if (const char* e = getenv("TKHIP_WIDE_RAYS")) ctx->wide_rays = std::min(std::max(atoi(e), 0), 64);   // tuning aid
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("tkhip_create: hipStreamCreate: %s", hipGetErrorString(e));
        delete ctx;
        return TKHIP_E_HIP;
    }
    int rc = upload_direction_tables(ctx);
    if (!rc) rc = static_scratch_init(device);
    if (rc) { tkhip_destroy(ctx); return rc; }
    *out = ctx;
    return TKHIP_OK;
}

int tkhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void tkhip_destroy(tkhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    destroy_sppm(ctx);
    DevBuf* bufs[] = {&ctx->b_wide, &ctx->b_nodes, &ctx->b_tri_isect, &ctx->b_tri_shade, &ctx->b_spheres, &ctx->b_prim_mat,
                      &ctx->b_mats, &ctx->b_lights, &ctx->b_cdf, &ctx->b_dirtab, &ctx->scratch_a, &ctx->scratch_b,
                      &ctx->scratch_c, &ctx->scratch_d, &ctx->scratch_e, &ctx->scratch_f, &ctx->prim_temp};
    for (DevBuf* b : bufs) b->release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int tkhip_set_bvh_builder(tkhip_ctx* ctx, int mode) {
    TK_REQUIRE(ctx, TKHIP_E_ARG, "tkhip_set_bvh_builder: null context");
    TK_REQUIRE(mode >= 0 && mode <= 2, TKHIP_E_ARG, "tkhip_set_bvh_builder: mode %d (0 by size, 1 host, 2 device)", mode);
    ctx->bvh_builder = mode;
    return TKHIP_OK;
}

int tkhip_set_wide_walk(tkhip_ctx* ctx, int max_rays) {
    TK_REQUIRE(ctx, TKHIP_E_ARG, "tkhip_set_wide_walk: null context");
    TK_REQUIRE(max_rays >= 0 && max_rays <= 64, TKHIP_E_ARG, "tkhip_set_wide_walk: %d outside 0..64", max_rays);
    ctx->wide_rays = max_rays;
    ctx->scene.wide_rays = max_rays;
    return TKHIP_OK;
}

int tkhip_bvh_info(tkhip_ctx* ctx, int32_t* n_nodes, int32_t* depth, int32_t* built_on_device, float* build_ms) {
    TK_REQUIRE(ctx && ctx->has_scene, TKHIP_E_STATE, "tkhip_bvh_info: no scene");
    if (n_nodes) *n_nodes = ctx->scene.n_nodes;
    if (depth) *depth = ctx->bvh_depth;
    if (built_on_device) *built_on_device = ctx->bvh_on_device;
    if (build_ms) *build_ms = ctx->bvh_ms;
    return TKHIP_OK;
}

int tkhip_set_scene(tkhip_ctx* ctx, const TkObject* objects, int n_objects, const TkTriangle* tris, int n_tris,
                    const TkSphere* spheres, int n_spheres, const TkMaterial* materials, int n_materials,
                    const TkLight* lights, int n_lights) {
    TK_REQUIRE(ctx, TKHIP_E_ARG, "tkhip_set_scene: null context");
    TK_REQUIRE(n_objects >= 0 && n_tris >= 0 && n_spheres >= 0 && n_materials >= 1 && n_lights >= 0, TKHIP_E_ARG,
               "tkhip_set_scene: negative count or no material");
    TK_REQUIRE((n_objects == 0 || objects) && (n_tris == 0 || tris) && (n_spheres == 0 || spheres) && materials &&
                   (n_lights == 0 || lights),
               TKHIP_E_ARG, "tkhip_set_scene: null array");
    TK_REQUIRE((int64_t)n_tris + n_spheres < (int64_t)1 << 28, TKHIP_E_LIMIT, "tkhip_set_scene: too many primitives");
    TK_REQUIRE(n_materials < 65536 && n_lights < 32767, TKHIP_E_LIMIT, "tkhip_set_scene: too many materials or lights");
    TK_REQUIRE(ctx->sppm == nullptr, TKHIP_E_STATE, "tkhip_set_scene: a render is in progress on this context");
    TK_HIP(hipSetDevice(ctx->device));
    // per-primitive material / light from the owning object
    std::vector<int32_t> prim_mat((size_t)n_tris + n_spheres, 0);
    std::vector<char> covered((size_t)n_tris + n_spheres, 0);
    for (int o = 0; o < n_objects; ++o) {
        const TkObject& ob = objects[o];
        TK_REQUIRE(ob.material >= 0 && ob.material < n_materials, TKHIP_E_ARG, "object %d: material %d out of range", o, ob.material);
        TK_REQUIRE(ob.light >= -1 && ob.light < n_lights, TKHIP_E_ARG, "object %d: light %d out of range", o, ob.light);
        int lim = ob.kind == 0 ? n_tris : n_spheres;
        TK_REQUIRE((ob.kind == 0 || ob.kind == 1) && ob.first >= 0 && ob.count >= 0 && ob.first + ob.count <= lim,
                   TKHIP_E_ARG, "object %d: primitive range out of bounds", o);
        for (int i = 0; i < ob.count; ++i) {
            size_t id = ob.kind == 0 ? (size_t)(ob.first + i) : (size_t)(n_tris + ob.first + i);
            prim_mat[id] = ob.material | ((ob.light + 1) << 16);
            covered[id] = 1;
        }
    }
    for (size_t i = 0; i < covered.size(); ++i)
        TK_REQUIRE(covered[i], TKHIP_E_ARG, "primitive %zu belongs to no object", i);
    for (int m = 0; m < n_materials; ++m)
        TK_REQUIRE(materials[m].type == TK_DIFFUSE || materials[m].type == TK_MIRROR || materials[m].type == TK_GLASS,
                   TKHIP_E_ARG, "material %d: unsupported type %d", m, materials[m].type);

    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (getenv("TKHIP_VERBOSE"))
            fprintf(stderr, "[tkhip] set_scene: %s at %.1f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    };
    const size_t nprim = (size_t)n_tris + n_spheres;
    hipStream_t s = ctx->stream;
    const bool on_device = nprim >= 3 && (ctx->bvh_builder == 2 || (ctx->bvh_builder == 0 && nprim >= (size_t)kDeviceBuildMin));
    int n_nodes = 0, depth = 0;
    float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
    if (on_device) {
        DevBuf d_tris, d_sph;
        struct Cleanup {
            DevBuf *a, *b;
            ~Cleanup() { a->release(); b->release(); }
        } cleanup{&d_tris, &d_sph};
        TK_TRY(d_tris.ensure(std::max<size_t>((size_t)n_tris, 1) * sizeof(TkTriangle)));
        TK_TRY(d_sph.ensure(std::max<size_t>((size_t)n_spheres, 1) * sizeof(TkSphere)));
        if (n_tris) TK_HIP(hipMemcpyAsync(d_tris.p, tris, (size_t)n_tris * sizeof(TkTriangle), hipMemcpyHostToDevice, s));
        if (n_spheres) TK_HIP(hipMemcpyAsync(d_sph.p, spheres, (size_t)n_spheres * sizeof(TkSphere), hipMemcpyHostToDevice, s));
        TK_TRY(build_bvh_device(s, ctx->prim_temp, d_tris.as<TkTriangle>(), n_tris, d_sph.as<TkSphere>(), n_spheres, kMaxLeaf,
                                kStackDepth, ctx->b_nodes, ctx->b_tri_isect, &n_nodes, &depth, bmin, bmax));
        lap("tree built on the device");
        // the same bound as for the host-built tree: a ray's stack holds one entry per level it has passed
        TK_REQUIRE(depth < kStackDepth - 1, TKHIP_E_LIMIT, "scene BVH depth %d (built on the device) exceeds the traversal stack", depth);
    } else {
        HostBvh bvh;
        build_bvh(tris, n_tris, spheres, n_spheres, kMaxLeaf, kStackDepth, &bvh);
        lap("tree built");
        TK_REQUIRE(bvh.max_depth < kStackDepth - 1, TKHIP_E_LIMIT, "scene BVH depth %d exceeds the traversal stack", bvh.max_depth);
        std::vector<TriIsect> isect(nprim);
        host_parallel_for(nprim, [&](size_t begin, size_t end) {
            for (size_t s = begin; s < end; ++s) {
                int id = bvh.order[s];
                TriIsect& t = isect[s];
                memset(&t, 0, sizeof(t));
                t.prim = id;
                if (id < n_tris) {
                    for (int a = 0; a < 3; ++a) {
                        t.v0[a] = tris[id].v0[a];
                        t.e1[a] = tris[id].v1[a] - tris[id].v0[a];   // src/Triangle.hpp:21-22
                        t.e2[a] = tris[id].v2[a] - tris[id].v0[a];
                    }
                }
            }
        });
        TK_TRY(upload(ctx->b_nodes, bvh.nodes, s));
        TK_TRY(upload(ctx->b_tri_isect, isect, s));
        TK_HIP(hipStreamSynchronize(s));   // the vectors go out of scope
        n_nodes = (int)bvh.nodes.size();
        depth = bvh.max_depth;
        for (int a = 0; a < 3; ++a) { bmin[a] = bvh.bmin[a]; bmax[a] = bvh.bmax[a]; }
    }
    // the eight-children form of the tree for the wide walk: every ray stack must fit its LDS columns
    ctx->wide_depth = 0;
    bool have_wide = false;
    if (n_nodes > 0 && n_nodes <= kWideMaxNodes) {
        std::vector<BvhNode> host_nodes((size_t)n_nodes);
        TK_HIP(hipMemcpyAsync(host_nodes.data(), ctx->b_nodes.p, (size_t)n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost, s));
        TK_HIP(hipStreamSynchronize(s));
        std::vector<WideNode> wide;
        const int wd = collapse_bvh(host_nodes.data(), n_nodes, &wide);
        if (wd * 7 <= kWideStack) {
            TK_TRY(upload(ctx->b_wide, wide, s));
            TK_HIP(hipStreamSynchronize(s));
            ctx->wide_depth = wd;
            have_wide = true;
        }
        lap("wide nodes");
    }
    ctx->bvh_on_device = on_device ? 1 : 0;
    ctx->bvh_depth = depth;
    ctx->bvh_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count();
    std::vector<TriShade> shade((size_t)std::max(n_tris, 1));
    std::vector<float> area((size_t)std::max(n_tris, 1), 0.f);
    host_parallel_for((size_t)n_tris, [&](size_t begin, size_t end) {
        for (size_t i = begin; i < end; ++i) {
            TriShade& t = shade[(size_t)i];
            float e1[3], e2[3];
            for (int a = 0; a < 3; ++a) {
                t.v0[a] = tris[i].v0[a]; t.v1[a] = tris[i].v1[a]; t.v2[a] = tris[i].v2[a];
                e1[a] = tris[i].v1[a] - tris[i].v0[a];
                e2[a] = tris[i].v2[a] - tris[i].v0[a];
            }
            // normal = normalize(e1 x e2), area = |e1 x e2| / 2 (src/Triangle.hpp:23-24, src/Vector.hpp:80-103)
            float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            float mag2 = cx * cx + cy * cy + cz * cz;
            if (mag2 > 0) {
                float inv = 1 / sqrtf(mag2);
                t.nx = cx * inv; t.ny = cy * inv; t.nz = cz * inv;
            } else { t.nx = cx; t.ny = cy; t.nz = cz; }
            area[(size_t)i] = sqrtf(cx * cx + cy * cy + cz * cz) * 0.5f;
        }
    });
    std::vector<float4> sph((size_t)std::max(n_spheres, 1));
    for (int j = 0; j < n_spheres; ++j)
        sph[(size_t)j] = make_float4(spheres[j].center[0], spheres[j].center[1], spheres[j].center[2], spheres[j].radius);
    // lights: Distribution1D over the triangle areas of the emitting mesh
    // (src/Triangle.cpp:170-178, src/core/sampling.h:57-66)
    std::vector<DLight> dl((size_t)std::max(n_lights, 1));
    std::vector<float> cdf;
    for (int l = 0; l < n_lights; ++l) {
        TK_REQUIRE(lights[l].object >= 0 && lights[l].object < n_objects, TKHIP_E_ARG, "light %d: object out of range", l);
        const TkObject& ob = objects[lights[l].object];
        TK_REQUIRE(ob.kind == 0 && ob.count > 0, TKHIP_E_ARG, "light %d: the emitting object must be a non-empty mesh", l);
        DLight& d = dl[(size_t)l];
        memcpy(d.Le, lights[l].Le, sizeof(d.Le));
        d.two_sided = lights[l].two_sided;
        d.tri_first = ob.first; d.tri_count = ob.count;
        d.cdf_first = (int32_t)cdf.size();
        size_t base = cdf.size();
        cdf.resize(base + (size_t)ob.count + 1);
        cdf[base] = 0;
        float total = 0;
        for (int i = 1; i < ob.count + 1; ++i) {
            float f = area[(size_t)(ob.first + i - 1)];
            total += f;
            cdf[base + (size_t)i] = cdf[base + (size_t)i - 1] + f / ob.count;
        }
        float c = cdf[base + (size_t)ob.count];
        for (int i = 1; i < ob.count + 1; ++i) cdf[base + (size_t)i] /= c;
        d.area = total;
    }
    std::vector<TkMaterial> mats(materials, materials + n_materials);
    ctx->h_mats = mats;
    lap("arrays laid out");
    TK_TRY(upload(ctx->b_tri_shade, shade, s));
    TK_TRY(upload(ctx->b_spheres, sph, s));
    TK_TRY(upload(ctx->b_prim_mat, prim_mat, s));
    TK_TRY(upload(ctx->b_mats, mats, s));
    TK_TRY(upload(ctx->b_lights, dl, s));
    TK_TRY(upload(ctx->b_cdf, cdf, s));
    TK_HIP(hipStreamSynchronize(s));
    lap("uploaded");
    DScene& sc = ctx->scene;
    sc.nodes = ctx->b_nodes.as<BvhNode>();
    sc.tri_isect = ctx->b_tri_isect.as<TriIsect>();
    sc.tri_shade = ctx->b_tri_shade.as<TriShade>();
    sc.spheres = ctx->b_spheres.as<float4>();
    sc.prim_mat = ctx->b_prim_mat.as<int32_t>();
    sc.mats = ctx->b_mats.as<TkMaterial>();
    sc.lights = ctx->b_lights.as<DLight>();
    sc.light_cdf = ctx->b_cdf.as<float>();
    sc.n_tris = n_tris; sc.n_spheres = n_spheres; sc.n_lights = n_lights;
    sc.n_nodes = n_nodes;
    sc.wide = have_wide ? ctx->b_wide.as<WideNode>() : nullptr;
    sc.wide_rays = ctx->wide_rays;
    for (int a = 0; a < 3; ++a) { sc.bmin[a] = bmin[a]; sc.bmax[a] = bmax[a]; }
    // bounds of the specular objects, padded by a ten-thousandth of the scene's extent
    sc.n_spec_boxes = 0;
    float pad = 0.f;
    for (int a = 0; a < 3; ++a) pad = std::max(pad, (bmax[a] - bmin[a]) * 1e-4f);
    for (int o = 0; o < n_objects && sc.n_spec_boxes >= 0; ++o) {
        const TkObject& ob = objects[o];
        if (materials[ob.material].type == TK_DIFFUSE || ob.count == 0) continue;
        if (sc.n_spec_boxes == kMaxSpecBoxes) { sc.n_spec_boxes = -1; break; }
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < ob.count; ++i)
            for (int a = 0; a < 3; ++a) {
                if (ob.kind == 0) {
                    const TkTriangle& t = tris[ob.first + i];
                    lo[a] = std::min(lo[a], std::min(t.v0[a], std::min(t.v1[a], t.v2[a])));
                    hi[a] = std::max(hi[a], std::max(t.v0[a], std::max(t.v1[a], t.v2[a])));
                } else {
                    const TkSphere& sp = spheres[ob.first + i];
                    lo[a] = std::min(lo[a], sp.center[a] - sp.radius);
                    hi[a] = std::max(hi[a], sp.center[a] + sp.radius);
                }
            }
        float* b = sc.spec_box[sc.n_spec_boxes++];
        for (int a = 0; a < 3; ++a) { b[a] = lo[a] - pad; b[3 + a] = hi[a] + pad; }
    }
    ctx->has_scene = true;
    return TKHIP_OK;
}

namespace tkd {
__global__ void k_sample_f(TkMaterial m, int mode, int64_t n, const float* wo, const float* nrm, const float* u, float* wi,
                           float* f, float* pdf) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 w;
    float p;
    F3 fv = mat_sample_f(m, mode != 0, ld3(wo + 3 * i), &w, ld3(nrm + 3 * i), u[2 * i], u[2 * i + 1], &p);
    wi[3 * i] = w.x; wi[3 * i + 1] = w.y; wi[3 * i + 2] = w.z;
    f[3 * i] = fv.x; f[3 * i + 1] = fv.y; f[3 * i + 2] = fv.z;
    pdf[i] = p;
}
__global__ void k_sample_le(DScene sc, int li, int64_t n, const float* u1, const float* u2, const float* upick, float* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LightEmission e = light_sample_le(sc, sc.lights[li], u1[2 * i], u1[2 * i + 1], u2[2 * i], u2[2 * i + 1], upick[i]);
    float* o = out + 14 * i;
    o[0] = e.o.x; o[1] = e.o.y; o[2] = e.o.z; o[3] = e.d.x; o[4] = e.d.y; o[5] = e.d.z;
    o[6] = e.n.x; o[7] = e.n.y; o[8] = e.n.z; o[9] = e.Le.x; o[10] = e.Le.y; o[11] = e.Le.z;
    o[12] = e.pdf_pos; o[13] = e.pdf_dir;
}
// StratifiedSampler::get1D / get2D (src/core/sampler.cpp:46-64) for n (pixel, sample) pairs, each
// drawing the same sequence of 1-D and 2-D samples from dimension 0 on.
__global__ void k_stratified(int64_t n, const int32_t* px, const int32_t* py, const int32_t* idx, const int32_t* xs,
                             const int32_t* ys, int n_draws, const uint8_t* draws, int floats_per, int jitter, uint32_t seed,
                             float* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sampler s;
    s.px = px[i]; s.py = py[i]; s.sample_idx = idx[i]; s.xs = xs[i]; s.ys = ys[i]; s.spp = xs[i] * ys[i]; s.dim = 0;
    s.rng_seed = seed; s.rng_pixel = (uint32_t)(px[i] + py[i] * 65536); s.rng_iter = (uint32_t)idx[i];
    s.no_jitter = jitter == 0;
    float* o = out + i * floats_per;
    for (int d = 0; d < n_draws; ++d) {
        if (draws[d] == 1) *o++ = sampler_get1d(s);
        else { sampler_get2d(s, o, o + 1); o += 2; }
    }
}
// Intersection::spawnRay / spawnRayTo(pos) (src/Intersection.hpp:25-38, src/Ray.hpp:43-57):
// in  = p, pError, n, w, target (5 x 3 floats); out = origin along w, origin towards target, direction to target.
__global__ void k_spawn_ray(int64_t n, const float* in, float* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = in + 15 * i;
    const F3 p = ld3(a), pe = ld3(a + 3), nn = ld3(a + 6), w = ld3(a + 9), tg = ld3(a + 12);
    const F3 o1 = offset_ray_origin(p, pe, nn, w);
    const F3 o2 = offset_ray_origin(p, pe, nn, sub3(tg, p));
    const F3 d2 = sub3(tg, o2);
    float* o = out + 9 * i;
    o[0] = o1.x; o[1] = o1.y; o[2] = o1.z; o[3] = o2.x; o[4] = o2.y; o[5] = o2.z; o[6] = d2.x; o[7] = d2.y; o[8] = d2.z;
}
__global__ void k_encode_direction(int64_t n, const float* wo, uint8_t* theta, uint8_t* phi) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = encode_direction(ld3(wo + 3 * i));
    theta[i] = (uint8_t)(b & 255u);
    phi[i] = (uint8_t)(b >> 8);
}
}  // namespace tkd

int tkhip_sample_f(tkhip_ctx* ctx, const TkMaterial* material, int mode, int64_t n, const float* wo, const float* nrm,
                   const float* u, float* wi, float* f, float* pdf) {
    TK_REQUIRE(ctx && material && wo && nrm && u && wi && f && pdf, TKHIP_E_ARG, "tkhip_sample_f: null argument");
    TK_REQUIRE(n >= 0, TKHIP_E_ARG, "tkhip_sample_f: negative count");
    TK_REQUIRE(material->type == TK_DIFFUSE || material->type == TK_MIRROR || material->type == TK_GLASS, TKHIP_E_ARG,
               "tkhip_sample_f: unsupported material type %d", material->type);
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TK_TRY(ctx->scratch_a.ensure((size_t)n * 8 * 4));    // wo 3, n 3, u 2
    TK_TRY(ctx->scratch_b.ensure((size_t)n * 7 * 4));    // wi 3, f 3, pdf 1
    float* in = ctx->scratch_a.as<float>();
    float* out = ctx->scratch_b.as<float>();
    TK_HIP(hipMemcpyAsync(in, wo, (size_t)n * 12, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(in + 3 * n, nrm, (size_t)n * 12, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(in + 6 * n, u, (size_t)n * 8, hipMemcpyHostToDevice, s));
    k_sample_f<<<divup(n, 256), 256, 0, s>>>(*material, mode, n, in, in + 3 * n, in + 6 * n, out, out + 3 * n, out + 6 * n);
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(wi, out, (size_t)n * 12, hipMemcpyDeviceToHost, s));
    TK_HIP(hipMemcpyAsync(f, out + 3 * n, (size_t)n * 12, hipMemcpyDeviceToHost, s));
    TK_HIP(hipMemcpyAsync(pdf, out + 6 * n, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    return TKHIP_OK;
}

int tkhip_sample_le(tkhip_ctx* ctx, int light, int64_t n, const float* u1, const float* u2, const float* upick, float* org,
                    float* dir, float* nrm, float* le, float* pdf_pos, float* pdf_dir) {
    TK_REQUIRE(ctx && u1 && u2 && upick && org && dir && nrm && le && pdf_pos && pdf_dir, TKHIP_E_ARG,
               "tkhip_sample_le: null argument");
    TK_REQUIRE(ctx->has_scene, TKHIP_E_STATE, "tkhip_sample_le: set the scene first");
    TK_REQUIRE(light >= 0 && light < ctx->scene.n_lights, TKHIP_E_ARG, "tkhip_sample_le: light %d out of range", light);
    TK_REQUIRE(n >= 0, TKHIP_E_ARG, "tkhip_sample_le: negative count");
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TK_TRY(ctx->scratch_a.ensure((size_t)n * 5 * 4));
    TK_TRY(ctx->scratch_b.ensure((size_t)n * 14 * 4));
    float* in = ctx->scratch_a.as<float>();
    TK_HIP(hipMemcpyAsync(in, u1, (size_t)n * 8, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(in + 2 * n, u2, (size_t)n * 8, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(in + 4 * n, upick, (size_t)n * 4, hipMemcpyHostToDevice, s));
    k_sample_le<<<divup(n, 256), 256, 0, s>>>(ctx->scene, light, n, in, in + 2 * n, in + 4 * n, ctx->scratch_b.as<float>());
    TK_LAUNCH_CHECK();
    std::vector<float> host((size_t)n * 14);
    TK_HIP(hipMemcpyAsync(host.data(), ctx->scratch_b.p, host.size() * 4, hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i) {
        const float* o = &host[(size_t)i * 14];
        for (int a = 0; a < 3; ++a) { org[3 * i + a] = o[a]; dir[3 * i + a] = o[3 + a]; nrm[3 * i + a] = o[6 + a]; le[3 * i + a] = o[9 + a]; }
        pdf_pos[i] = o[12]; pdf_dir[i] = o[13];
    }
    return TKHIP_OK;
}

int tkhip_stratified_samples(tkhip_ctx* ctx, int64_t n, const int32_t* px, const int32_t* py, const int32_t* sample_idx,
                             const int32_t* xs, const int32_t* ys, int n_draws, const uint8_t* draws, int jitter,
                             uint64_t seed, float* out) {
    TK_REQUIRE(ctx && (n == 0 || (px && py && sample_idx && xs && ys && out)) && draws, TKHIP_E_ARG,
               "tkhip_stratified_samples: null argument");
    TK_REQUIRE(n >= 0 && n_draws >= 1 && n_draws <= 256, TKHIP_E_LIMIT, "tkhip_stratified_samples: 1..256 draws per sample");
    int floats_per = 0;
    for (int d = 0; d < n_draws; ++d) {
        TK_REQUIRE(draws[d] == 1 || draws[d] == 2, TKHIP_E_ARG, "tkhip_stratified_samples: a draw is 1 (get1D) or 2 (get2D)");
        floats_per += draws[d];
    }
    for (int64_t i = 0; i < n; ++i)
        TK_REQUIRE(xs[i] >= 1 && ys[i] >= 1 && sample_idx[i] >= 0 && sample_idx[i] < xs[i] * ys[i], TKHIP_E_ARG,
                   "tkhip_stratified_samples: sample %lld: index %d outside %d x %d strata", (long long)i, sample_idx[i], xs[i], ys[i]);
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t q = (size_t)n;
    TK_TRY(ctx->scratch_a.ensure(q * 20 + 256));
    TK_TRY(ctx->scratch_b.ensure(q * (size_t)floats_per * 4));
    int32_t* d = ctx->scratch_a.as<int32_t>();
    uint8_t* dd = reinterpret_cast<uint8_t*>(d + 5 * q);
    const int32_t* host[5] = {px, py, sample_idx, xs, ys};
    for (int a = 0; a < 5; ++a) TK_HIP(hipMemcpyAsync(d + a * q, host[a], q * 4, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(dd, draws, (size_t)n_draws, hipMemcpyHostToDevice, s));
    k_stratified<<<divup(n, 256), 256, 0, s>>>(n, d, d + q, d + 2 * q, d + 3 * q, d + 4 * q, n_draws, dd, floats_per, jitter,
                                               (uint32_t)seed, ctx->scratch_b.as<float>());
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(out, ctx->scratch_b.p, q * (size_t)floats_per * 4, hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    return TKHIP_OK;
}

int tkhip_spawn_rays(tkhip_ctx* ctx, int64_t n, const float* p, const float* perr, const float* nrm, const float* w,
                     const float* target, float* org, float* org_to, float* dir_to) {
    TK_REQUIRE(ctx && (n == 0 || (p && perr && nrm && w && target && org && org_to && dir_to)), TKHIP_E_ARG,
               "tkhip_spawn_rays: null argument");
    TK_REQUIRE(n >= 0, TKHIP_E_ARG, "tkhip_spawn_rays: negative count");
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t q = (size_t)n;
    std::vector<float> in(q * 15), out(q * 9);
    const float* src[5] = {p, perr, nrm, w, target};
    for (size_t i = 0; i < q; ++i)
        for (int a = 0; a < 5; ++a)
            for (int c = 0; c < 3; ++c) in[i * 15 + (size_t)a * 3 + c] = src[a][i * 3 + c];
    TK_TRY(ctx->scratch_a.ensure(q * 60));
    TK_TRY(ctx->scratch_b.ensure(q * 36));
    TK_HIP(hipMemcpyAsync(ctx->scratch_a.p, in.data(), q * 60, hipMemcpyHostToDevice, s));
    k_spawn_ray<<<divup(n, 256), 256, 0, s>>>(n, ctx->scratch_a.as<float>(), ctx->scratch_b.as<float>());
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(out.data(), ctx->scratch_b.p, q * 36, hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < q; ++i)
        for (int c = 0; c < 3; ++c) {
            org[i * 3 + c] = out[i * 9 + c];
            org_to[i * 3 + c] = out[i * 9 + 3 + c];
            dir_to[i * 3 + c] = out[i * 9 + 6 + c];
        }
    return TKHIP_OK;
}

int tkhip_encode_directions(tkhip_ctx* ctx, int64_t n, const float* wo, uint8_t* theta, uint8_t* phi) {
    TK_REQUIRE(ctx && (n == 0 || (wo && theta && phi)), TKHIP_E_ARG, "tkhip_encode_directions: null argument");
    TK_REQUIRE(n >= 0, TKHIP_E_ARG, "tkhip_encode_directions: negative count");
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t q = (size_t)n;
    TK_TRY(ctx->scratch_a.ensure(q * 12));
    TK_TRY(ctx->scratch_b.ensure(q * 2));
    uint8_t* d = ctx->scratch_b.as<uint8_t>();
    TK_HIP(hipMemcpyAsync(ctx->scratch_a.p, wo, q * 12, hipMemcpyHostToDevice, s));
    k_encode_direction<<<divup(n, 256), 256, 0, s>>>(n, ctx->scratch_a.as<float>(), d, d + q);
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(theta, d, q, hipMemcpyDeviceToHost, s));
    TK_HIP(hipMemcpyAsync(phi, d + q, q, hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    return TKHIP_OK;
}

int tkhip_set_camera(tkhip_ctx* ctx, const float c2w[16], const float raster2cam[16], int width, int height) {
    TK_REQUIRE(ctx && c2w && raster2cam, TKHIP_E_ARG, "tkhip_set_camera: null argument");
    TK_REQUIRE(width >= 1 && height >= 1 && (int64_t)width * height < (int64_t)1 << 30, TKHIP_E_LIMIT,
               "tkhip_set_camera: image size %dx%d out of range", width, height);
    TK_REQUIRE(ctx->sppm == nullptr, TKHIP_E_STATE, "tkhip_set_camera: a render is in progress on this context");
    memcpy(ctx->cam.c2w, c2w, sizeof(float) * 16);
    memcpy(ctx->cam.r2c, raster2cam, sizeof(float) * 16);
    ctx->cam.w = width;
    ctx->cam.h = height;
    ctx->has_camera = true;
    return TKHIP_OK;
}

int tkhip_trace_rays(tkhip_ctx* ctx, int64_t n, const float* org, const float* dir, const float* tmax, int32_t* prim,
                     float* t, float* p, float* nrm, float* perr) {
    TK_REQUIRE(ctx && (n == 0 || (org && dir && prim)), TKHIP_E_ARG, "tkhip_trace_rays: null argument");
    TK_REQUIRE(ctx->has_scene, TKHIP_E_STATE, "tkhip_trace_rays: no scene set");
    TK_REQUIRE(n >= 0 && n < (int64_t)1 << 31, TKHIP_E_LIMIT, "tkhip_trace_rays: ray count out of range");
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    const size_t q = (size_t)n;
    DevBuf in, out;
    int rc = in.ensure(q * 28);
    if (!rc) rc = out.ensure(q * 44);
    if (rc) { in.release(); out.release(); return rc; }
    hipStream_t s = ctx->stream;
    float* dorg = in.as<float>();
    float* ddir = dorg + 3 * q;
    float* dtm = ddir + 3 * q;
    int32_t* dprim = out.as<int32_t>();
    float* dt = out.as<float>() + q;
    float* dp = dt + q;
    float* dn = dp + 3 * q;
    float* de = dn + 3 * q;
    hipError_t e = hipMemcpyAsync(dorg, org, q * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ddir, dir, q * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && tmax) e = hipMemcpyAsync(dtm, tmax, q * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        if (ctx->scene.wide && ctx->scene.wide_rays >= 64)
            k_trace_rays_wide<<<divup(n, kTraceBlock), kTraceBlock, 0, s>>>(ctx->scene, n, dorg, ddir, tmax ? dtm : nullptr, dprim,
                                                                          dt, dp, dn, de);
        else
            k_trace_rays<<<divup(n, kTraceBlock), kTraceBlock, 0, s>>>(ctx->scene, n, dorg, ddir, tmax ? dtm : nullptr, dprim, dt,
                                                                     dp, dn, de);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(prim, dprim, q * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && t) e = hipMemcpy(t, dt, q * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && p) e = hipMemcpy(p, dp, q * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && nrm) e = hipMemcpy(nrm, dn, q * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && perr) e = hipMemcpy(perr, de, q * 12, hipMemcpyDeviceToHost);
    in.release();
    out.release();
    if (e != hipSuccess) { set_error("tkhip_trace_rays: %s", hipGetErrorString(e)); return TKHIP_E_HIP; }
    return TKHIP_OK;
}

int tkhip_sample_one_light(tkhip_ctx* ctx, int64_t n, const float* org, const float* dir, const float* u, int32_t* prim,
                           float* ld) {
    TK_REQUIRE(ctx && (n == 0 || (org && dir && u && prim && ld)), TKHIP_E_ARG, "tkhip_sample_one_light: null argument");
    TK_REQUIRE(ctx->has_scene, TKHIP_E_STATE, "tkhip_sample_one_light: no scene set");
    TK_REQUIRE(n >= 0 && n < (int64_t)1 << 28, TKHIP_E_LIMIT, "tkhip_sample_one_light: ray count out of range");
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t q = (size_t)n;
    TK_TRY(ctx->scratch_a.ensure(q * 40));
    TK_TRY(ctx->scratch_b.ensure(q * 16));
    float* dorg = ctx->scratch_a.as<float>();
    float* ddir = dorg + 3 * q;
    float* du = ddir + 3 * q;
    int32_t* dprim = ctx->scratch_b.as<int32_t>();
    float* dld = ctx->scratch_b.as<float>() + q;
    TK_HIP(hipMemcpyAsync(dorg, org, q * 12, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(ddir, dir, q * 12, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(du, u, q * 16, hipMemcpyHostToDevice, s));
    k_sample_one_light<<<divup(n, kTraceBlock), kTraceBlock, 0, s>>>(ctx->scene, n, dorg, ddir, du, dprim, dld);
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(prim, dprim, q * 4, hipMemcpyDeviceToHost, s));
    TK_HIP(hipMemcpyAsync(ld, dld, q * 12, hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    return TKHIP_OK;
}

int tkhip_radical_inverse(tkhip_ctx* ctx, int64_t n, const int32_t* dim, const uint64_t* index, float* out) {
    TK_REQUIRE(ctx && (n == 0 || (dim && index && out)), TKHIP_E_ARG, "tkhip_radical_inverse: null argument");
    if (n == 0) return TKHIP_OK;
    for (int64_t i = 0; i < n; ++i)
        TK_REQUIRE(dim[i] >= 0 && dim[i] < 16, TKHIP_E_LIMIT, "tkhip_radical_inverse: dimension %d outside 0..15", dim[i]);
    TK_HIP(hipSetDevice(ctx->device));
    const size_t q = (size_t)n;
    DevBuf b;
    TK_TRY(b.ensure(q * 16));
    uint64_t* di = b.as<uint64_t>();
    int32_t* dd = (int32_t*)(di + q);
    float* dout = (float*)(dd + q);
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemcpyAsync(di, index, q * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dd, dim, q * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        k_radical_inverse<<<divup(n, 256), 256, 0, s>>>(n, dd, di, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(out, dout, q * 4, hipMemcpyDeviceToHost);
    b.release();
    if (e != hipSuccess) { set_error("tkhip_radical_inverse: %s", hipGetErrorString(e)); return TKHIP_E_HIP; }
    return TKHIP_OK;
}

int tkhip_camera_rays(tkhip_ctx* ctx, int64_t n, const float* raster, float* org, float* dir) {
    TK_REQUIRE(ctx && (n == 0 || (raster && org && dir)), TKHIP_E_ARG, "tkhip_camera_rays: null argument");
    TK_REQUIRE(ctx->has_camera, TKHIP_E_STATE, "tkhip_camera_rays: no camera set");
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    const size_t q = (size_t)n;
    DevBuf b;
    TK_TRY(b.ensure(q * 32));
    float* dr = b.as<float>();
    float* dorg = dr + 2 * q;
    float* ddir = dorg + 3 * q;
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemcpyAsync(dr, raster, q * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        k_camera_rays<<<divup(n, 256), 256, 0, s>>>(ctx->cam, n, dr, dorg, ddir);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(org, dorg, q * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(dir, ddir, q * 12, hipMemcpyDeviceToHost);
    b.release();
    if (e != hipSuccess) { set_error("tkhip_camera_rays: %s", hipGetErrorString(e)); return TKHIP_E_HIP; }
    return TKHIP_OK;
}

}  // extern "C"
