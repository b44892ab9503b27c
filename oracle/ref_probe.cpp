// ref_probe.cpp — TEST INFRASTRUCTURE, build container only.
//
// Links the upstream CPU renderer's own translation units (compiled by
// oracle/build_ref.sh from /root/reference) and calls its components directly
// on seeded inputs, writing inputs and outputs as .npy files.  The arrays are
// packed into tests/golden/ref_*.npz by oracle/make_ref_golden.py; the CPU
// oracle and the HIP kernels are then checked against them.  Nothing here is
// part of the product and nothing of the upstream sources is copied: this file
// only includes their headers and calls their functions.
//
// Usage (cwd must be a directory from which the scene files' "../models/..."
// paths resolve, i.e. assets/scenes):
//   ref_probe <in_dir> <out_dir>
// in_dir holds raw little-endian arrays written by make_ref_golden.py:
//   camera.f32        32 floats: cameraToWorld, cameraToScreen (row-major)
//   g_pos.f32 g_pow.f32 g_dir.u8   photons of the global map  (xyz, rgb, theta/phi)
//   c_pos.f32 c_pow.f32 c_dir.u8   photons of the caustic map
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

#define private public
#define protected public
#include "Photonmap.h"
#include "TkSppmTracer.h"
#include "Scene.hpp"
#include "camera.h"
#include "sampler.h"
#include "Material.hpp"
#include "AreaLight.hpp"
#include "Object.hpp"
#include "Sphere.hpp"
#include "Triangle.hpp"
#undef private
#undef protected
#include "TkLoader.h"
#include "TkFilm.h"
#include "TkFilter.h"
#include "TkLowdiscrepancy.h"
#include "TkHash.h"
#include "sampling.h"

using namespace tk;

// build_ref.sh turns get_random_float into a dispatcher over this pointer (for ref_probe2.cpp); it stays unset
// here, so every call runs upstream's own body.
namespace tk { float (*probe_draw_hook)() = nullptr; }

// ------------------------------------------------------------------ npy out
static std::string g_out;
static void save_npy(const std::string& name, const char* descr, const std::vector<size_t>& shape, const void* data,
                     size_t bytes) {
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
template <> struct Descr<uint32_t> { static const char* s() { return "<u4"; } };
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

// Seeded inputs: 24-bit uniforms from a Mersenne Twister (the same on every platform).
struct Rng {
    std::mt19937 g;
    explicit Rng(uint32_t s) : g(s) {}
    float u() { return (float)(g() >> 8) * (1.0f / 16777216.0f); }
    float range(float a, float b) { return a + (b - a) * u(); }
    uint32_t below(uint32_t n) { return (uint32_t)(g() % n); }
    Vector3f unit() {
        for (;;) {
            Vector3f v(range(-1, 1), range(-1, 1), range(-1, 1));
            float l2 = v.x * v.x + v.y * v.y + v.z * v.z;
            if (l2 > 1e-4f && l2 <= 1.f) return normalize(v);
        }
    }
};
static void push3(std::vector<float>& o, const Vector3f& v) { o.push_back(v.x); o.push_back(v.y); o.push_back(v.z); }
static void pushs(std::vector<float>& o, const Spectrum& s) { o.push_back(s.r); o.push_back(s.g); o.push_back(s.b); }

// ---------------------------------------------------- RadicalInverse, sampler
static void probe_lowdiscrepancy() {
    std::vector<int32_t> dim;
    std::vector<uint64_t> idx;
    std::vector<float> val;
    const uint64_t extra[] = {1000003ull, 123456789ull, 4294967295ull, 4294967296ull, 1099511627776ull,
                              987654321987ull, 72057594037927935ull};
    for (int d = 0; d < 7; ++d) {
        for (uint64_t i = 0; i < 4096; ++i) { dim.push_back(d); idx.push_back(i); val.push_back(RadicalInverse(d, i)); }
        for (uint64_t e : extra) { dim.push_back(d); idx.push_back(e); val.push_back(RadicalInverse(d, e)); }
    }
    save("radinv_dim", dim, {dim.size()});
    save("radinv_index", idx, {idx.size()});
    save("radinv_value", val, {val.size()});

    // Hash(Point2i, dim, seed) and Math::PermutationElement
    Rng r(11);
    std::vector<int32_t> hx, hy, hd, hs;
    std::vector<uint64_t> hv;
    for (int i = 0; i < 4096; ++i) {
        int x = (int)r.below(4096), y = (int)r.below(4096), d = (int)r.below(64), s = (int)r.below(9);
        hx.push_back(x); hy.push_back(y); hd.push_back(d); hs.push_back(s);
        hv.push_back(Hash(Point2i(x, y), d, s));
    }
    save("hash_x", hx, {hx.size()}); save("hash_y", hy, {hy.size()}); save("hash_dim", hd, {hd.size()});
    save("hash_seed", hs, {hs.size()}); save("hash_value", hv, {hv.size()});
    std::vector<uint32_t> pi, pl, pp;
    std::vector<int32_t> pv;
    const uint32_t lens[] = {1, 4, 9, 16, 25, 64, 100, 1024};
    for (int i = 0; i < 8192; ++i) {
        uint32_t l = lens[r.below(8)], ii = r.below(l), p = r.g();
        pi.push_back(ii); pl.push_back(l); pp.push_back(p);
        pv.push_back(Math::PermutationElement(ii, l, p));
    }
    save("perm_i", pi, {pi.size()}); save("perm_l", pl, {pl.size()}); save("perm_p", pp, {pp.size()});
    save("perm_value", pv, {pv.size()});

    // StratifiedSampler without jitter: the dimension sequence of one camera path
    // (get2D, get2D, then get1D, get2D, get1D, get2D per vertex)
    std::vector<int32_t> sx, sy, si, sxs, sys;
    std::vector<float> sv;
    const int strata[][2] = {{1, 1}, {2, 2}, {3, 3}, {4, 4}, {8, 8}};
    for (int i = 0; i < 1024; ++i) {
        const int* st = strata[r.below(5)];
        StratifiedSampler s(st[0], st[1], false, 0);
        int x = (int)r.below(1024), y = (int)r.below(1024), id = (int)r.below((uint32_t)(st[0] * st[1]));
        s.startPixelSample(Point2i(x, y), id, 0);
        sx.push_back(x); sy.push_back(y); si.push_back(id); sxs.push_back(st[0]); sys.push_back(st[1]);
        Vector2f a = s.get2D(), b = s.get2D();
        sv.push_back(a.x); sv.push_back(a.y); sv.push_back(b.x); sv.push_back(b.y);
        for (int v = 0; v < 2; ++v) {
            sv.push_back(s.get1D());
            Vector2f c = s.get2D();
            sv.push_back(c.x); sv.push_back(c.y);
            sv.push_back(s.get1D());
            Vector2f d = s.get2D();
            sv.push_back(d.x); sv.push_back(d.y);
        }
    }
    save("strat_x", sx, {sx.size()}); save("strat_y", sy, {sy.size()}); save("strat_idx", si, {si.size()});
    save("strat_xs", sxs, {sxs.size()}); save("strat_ys", sys, {sys.size()});
    save("strat_values", sv, {sx.size(), 16});

    // concentricSampleDisk / cosineSampleHemisphere / uniformSampleTriangle
    std::vector<float> u, disk, hemi, tri;
    for (int i = 0; i < 8192; ++i) {
        Vector2f uu(r.u(), r.u());
        if (i < 8) uu = Vector2f(i & 1 ? 0.5f : 0.f, i & 2 ? 0.5f : (i & 4 ? 1.f - 0x1p-24f : 0.25f));
        u.push_back(uu.x); u.push_back(uu.y);
        Vector2f d = concentricSampleDisk(uu);
        disk.push_back(d.x); disk.push_back(d.y);
        push3(hemi, cosineSampleHemisphere(uu));
        Vector2f t = uniformSampleTriangle(uu);
        tri.push_back(t.x); tri.push_back(t.y);
    }
    save("sampling_u", u, {u.size() / 2, 2}); save("sampling_disk", disk, {disk.size() / 2, 2});
    save("sampling_hemisphere", hemi, {hemi.size() / 3, 3}); save("sampling_triangle", tri, {tri.size() / 2, 2});
}

// ------------------------------------------------------------- photon map
struct Cloud {
    std::vector<float> pos, pow;
    std::vector<uint8_t> th, ph;
    size_t n() const { return th.size(); }
};
// A floor, a wall and some volume scatter, like the photons of a Cornell box.
static Cloud surface_cloud(Rng& r, int n) {
    Cloud c;
    for (int i = 0; i < n; ++i) {
        float x = r.range(-2, 2), y = r.range(-2, 2), z = r.range(-2, 2), u = r.u();
        if (u < 0.45f) y = -2.0f;
        else if (u < 0.8f) x = 2.0f;
        c.pos.push_back(x); c.pos.push_back(y); c.pos.push_back(z);
        for (int k = 0; k < 3; ++k) c.pow.push_back(r.u());
        c.th.push_back((uint8_t)r.below(256)); c.ph.push_back((uint8_t)r.below(256));
    }
    return c;
}
static void fill_map(PhotonMap& m, const Cloud& c) {
    s32 num = 0;
    for (size_t i = 0; i < c.n(); ++i)
        m.storePhoton(Vector3f(c.pos[3 * i], c.pos[3 * i + 1], c.pos[3 * i + 2]),
                      Spectrum(c.pow[3 * i], c.pow[3 * i + 1], c.pow[3 * i + 2]), c.th[i], c.ph[i], num);
    m.buildKdTree();
}

// radiance_estimate + the neighbour set of neighbor_search for nq queries.
static void run_queries(const std::string& tag, const Cloud& c, int k, const std::vector<float>& qp,
                        const std::vector<float>& qn, const std::vector<float>& qwo, const std::vector<float>& qR,
                        const std::vector<float>& qr2) {
    PhotonMap map((s32)c.n(), k);
    fill_map(map, c);
    const size_t nq = qr2.size();
    std::vector<float> flux, r2out, d2;
    std::vector<int32_t> found, idx(nq * (size_t)k, -1);
    for (size_t q = 0; q < nq; ++q) {
        Intersection it;
        it.p = Vector3f(qp[3 * q], qp[3 * q + 1], qp[3 * q + 2]);
        it.n = Vector3f(qn[3 * q], qn[3 * q + 1], qn[3 * q + 2]);
        it.wo = Vector3f(qwo[3 * q], qwo[3 * q + 1], qwo[3 * q + 2]);
        std::shared_ptr<Material> mat(new DiffuseMaterial(Spectrum(qR[3 * q], qR[3 * q + 1], qR[3 * q + 2])));
        Object obj(std::shared_ptr<Shape>(), std::shared_ptr<AreaLight>(), mat);
        it.obj = &obj;
        float r2 = qr2[q];
        int nf = 0;
        Spectrum s = map.radiance_estimate(it, r2, nf);
        pushs(flux, s);
        r2out.push_back(r2);
        found.push_back(nf);
        // the neighbour set itself: the same search, called directly
        NearestPhotons np;
        np.dstPos = it.p;
        np.found = 0;
        np.neighbor_photons.resize(k, 0);
        np.dist2 = (float*)malloc(sizeof(float) * (k + 1));
        np.dist2[k] = qr2[q];
        neighbor_search(&np, map.root, k);
        std::vector<int32_t> ids;
        for (int i = 0; i < np.found; ++i) ids.push_back((int32_t)(np.neighbor_photons[i] - &map.all_raw_photons[0]));
        std::sort(ids.begin(), ids.end());
        for (int i = 0; i < np.found; ++i) idx[q * (size_t)k + i] = ids[i];
        // sorted squared distances of the set (for fixtures with ties, where the set is not unique)
        std::vector<float> dd;
        for (int i = 0; i < np.found; ++i) {
            const Photon* p = np.neighbor_photons[i];
            float dx = p->pos.x - it.p.x, dy = p->pos.y - it.p.y, dz = p->pos.z - it.p.z;
            dd.push_back(dx * dx + dy * dy + dz * dz);
        }
        std::sort(dd.begin(), dd.end());
        dd.resize(k, -1.f);
        d2.insert(d2.end(), dd.begin(), dd.end());
        free(np.dist2);
    }
    save(tag + "_pos", c.pos, {c.n(), 3}); save(tag + "_power", c.pow, {c.n(), 3});
    save(tag + "_theta", c.th, {c.n()}); save(tag + "_phi", c.ph, {c.n()});
    save(tag + "_q_p", qp, {nq, 3}); save(tag + "_q_n", qn, {nq, 3}); save(tag + "_q_wo", qwo, {nq, 3});
    save(tag + "_q_R", qR, {nq, 3}); save(tag + "_q_radius2", qr2, {nq});
    save(tag + "_flux", flux, {nq, 3}); save(tag + "_found", found, {nq}); save(tag + "_radius2_out", r2out, {nq});
    save(tag + "_idx", idx, {nq, (size_t)k}); save(tag + "_dist2", d2, {nq, (size_t)k});
    std::vector<int32_t> kk(1, k);
    save(tag + "_k", kk, {1});
}

static void probe_photon_map() {
    {   // direction tables (PhotonMap::PhotonMap)
        PhotonMap m(1, 1);
        std::vector<float> t;
        t.insert(t.end(), m.cosTheta, m.cosTheta + 256); t.insert(t.end(), m.sinTheta, m.sinTheta + 256);
        t.insert(t.end(), m.cosPhi, m.cosPhi + 256); t.insert(t.end(), m.sinPhi, m.sinPhi + 256);
        save("dirtab", t, {4, 256});
    }
    Rng r(5);
    for (int pass = 0; pass < 2; ++pass) {
        const int k = pass == 0 ? 100 : 50, n = pass == 0 ? 20000 : 6000, nq = 1500;
        Cloud c = surface_cloud(r, n);
        std::vector<float> qp, qn, qwo, qR, qr2;
        const float radii[3] = {0.08f, 0.8f, 0.005f};
        for (int q = 0; q < nq; ++q) {
            float x = r.range(-2, 2), y = r.range(-2, 2), z = r.range(-2, 2);
            Vector3f nn(0, 1, 0);
            if (q < nq / 2) y = -2.0f;
            else nn = Vector3f(-1, 0, 0);
            if (q >= nq / 2 && q < 3 * nq / 4) x = 2.0f;
            qp.push_back(x); qp.push_back(y); qp.push_back(z);
            push3(qn, nn);
            push3(qwo, r.unit());
            for (int j = 0; j < 3; ++j) qR.push_back(r.u());
            qr2.push_back(radii[r.below(3)]);
        }
        // a few degenerate radii: zero, and one that holds the whole map
        qr2[0] = 0.f; qr2[1] = 1e30f;
        run_queries(pass == 0 ? "knn100" : "knn50", c, k, qp, qn, qwo, qR, qr2);
    }
    {   // ties: lattice positions (many equal distances), duplicates, photons at exactly radius2
        Cloud c;
        for (int i = 0; i < 4096; ++i) {
            int gx = i & 15, gy = (i >> 4) & 15, gz = (i >> 8) & 15;
            c.pos.push_back(0.25f * gx); c.pos.push_back(0.25f * gy); c.pos.push_back(0.25f * (gz >> 1));   // z-duplicates
            for (int k = 0; k < 3; ++k) c.pow.push_back(1.0f);   // equal powers: the estimate does not depend on which tie wins
            c.th.push_back(64); c.ph.push_back(0);
        }
        std::vector<float> qp, qn, qwo, qR, qr2;
        Rng q(9);
        for (int i = 0; i < 256; ++i) {
            float x = 0.25f * (float)q.below(16), y = 0.25f * (float)q.below(16), z = 0.25f * (float)q.below(8);
            if (i & 1) { x += 0.125f; }
            qp.push_back(x); qp.push_back(y); qp.push_back(z);
            push3(qn, Vector3f(0, 0, 1)); push3(qwo, Vector3f(0, 0, 1));
            for (int j = 0; j < 3; ++j) qR.push_back(0.5f);
            const float rr[4] = {0.0625f, 0.25f, 0.5f, 4.0f};   // 0.0625 = 0.25^2: lattice neighbours sit exactly on the radius
            qr2.push_back(rr[q.below(4)]);
        }
        run_queries("knnties", c, 20, qp, qn, qwo, qR, qr2);
    }
}

// A sampler that hands out four given numbers: what Scene::uniformSampleOneLight draws, in its order
// (get1D: the light, get2D: the point on it, get1D: the triangle of its mesh).
struct FixedSampler : public Sampler {
    float v[4];
    int at;
    FixedSampler() : Sampler(1), at(0) { v[0] = v[1] = v[2] = v[3] = 0.f; }
    float get1D() override { return v[at++ & 3]; }
    Vector2f get2D() override { float a = v[at++ & 3], b = v[at++ & 3]; return Vector2f(a, b); }
    std::unique_ptr<Sampler> clone(int) override { return std::unique_ptr<Sampler>(); }
};

// ------------------------------------------------------------------- rays
static Config* load(const char* file) {
    Config* cfg = new Config();
    cfg->scene = new Scene();
    if (!load_scene(cfg, file)) { fprintf(stderr, "load_scene(%s) failed\n", file); exit(3); }
    return cfg;
}

static void probe_rays(const char* file, const std::string& tag, int n_rays, uint32_t seed) {
    Config* cfg = load(file);
    const Scene* sc = cfg->scene;
    Rng r(seed);
    std::vector<float> org, dir, t, p, nrm, perr, wo;
    std::vector<int32_t> hit, objidx;
    const ObjectPtrVec& objs = sc->get_objects();
    // Scene::uniformSampleOneLight at the hits of the first rays
    const int n_nee = n_rays < 8192 ? n_rays : 8192;
    std::vector<float> nee_u, nee_ld;
    Rng r4(seed + 3);
    Vector3f cur_o(0, 0, -6.5f);
    for (int i = 0; i < n_rays; ++i) {
        Vector3f o, d;
        const int kind = i % 4;
        if (kind == 0) {   // from the eye into the box
            o = Vector3f(0, 0, -6.5f);
            d = normalize(Vector3f(r.range(-2.2f, 2.2f), r.range(-2.2f, 2.2f), r.range(0.f, 4.4f)) - o);
        } else if (kind == 1) {   // from inside the box, any direction
            o = Vector3f(r.range(-2.1f, 2.1f), r.range(-2.1f, 2.1f), r.range(0.1f, 4.3f));
            d = r.unit();
        } else if (kind == 2) {   // continue from the last hit point: secondary-ray origins on surfaces
            o = cur_o;
            d = r.unit();
        } else {   // axis-parallel and unnormalised directions (exact zeros in the slab test)
            o = Vector3f(r.range(-2.f, 2.f), r.range(-2.f, 2.f), r.range(0.2f, 4.f));
            int a = (int)r.below(3);
            float s = r.u() < 0.5f ? -1.f : 1.f;
            d = Vector3f(a == 0 ? s : 0.f, a == 1 ? s : 0.f, a == 2 ? s : 0.f);
            if (r.u() < 0.3f) d = d * 2.5f;
        }
        Ray ray(o, d);
        Intersection is;
        is.obj = nullptr;
        bool h = sc->intersect(ray, &is);
        push3(org, o); push3(dir, d);
        hit.push_back(h ? 1 : 0);
        if (i < n_nee) {
            FixedSampler fs;
            for (int c = 0; c < 4; ++c) { fs.v[c] = r4.u(); nee_u.push_back(fs.v[c]); }
            Spectrum L = h ? sc->uniformSampleOneLight(is, fs) : Spectrum(0.f, 0.f, 0.f);
            pushs(nee_ld, L);
        }
        if (h) {
            t.push_back((float)ray.t_max);
            push3(p, is.p); push3(nrm, is.n); push3(perr, is.pError); push3(wo, is.wo);
            int oi = -1;
            for (size_t j = 0; j < objs.size(); ++j) if (objs[j] == is.obj) oi = (int)j;
            objidx.push_back(oi);
            cur_o = is.spawnRay(r.unit()).origin;
        } else {
            t.push_back(0.f);
            push3(p, Vector3f(0)); push3(nrm, Vector3f(0)); push3(perr, Vector3f(0)); push3(wo, Vector3f(0));
            objidx.push_back(-1);
        }
    }
    const size_t n = (size_t)n_rays;
    save(tag + "_org", org, {n, 3}); save(tag + "_dir", dir, {n, 3}); save(tag + "_hit", hit, {n});
    save(tag + "_t", t, {n}); save(tag + "_p", p, {n, 3}); save(tag + "_n", nrm, {n, 3});
    save(tag + "_perr", perr, {n, 3}); save(tag + "_wo", wo, {n, 3}); save(tag + "_obj", objidx, {n});
    // "rays_<scene>" -> "nee_<scene>": the light samples go into a fixture of their own
    const std::string ntag = "nee_" + tag.substr(5);
    save(ntag + "_u", nee_u, {(size_t)n_nee, 4}); save(ntag + "_ld", nee_ld, {(size_t)n_nee, 3});

    // Intersection::spawnRay / spawnRayTo on the hits just found (offsetRayOrigin)
    std::vector<float> sp_in, sp_out;
    Rng r2(seed + 1);
    for (size_t i = 0; i < n && sp_in.size() < 15 * 2048; ++i) {
        if (!hit[i]) continue;
        Intersection is(Vector3f(p[3 * i], p[3 * i + 1], p[3 * i + 2]), Vector3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]),
                        Vector2f(0), Vector3f(perr[3 * i], perr[3 * i + 1], perr[3 * i + 2]), Vector3f(0));
        Vector3f w = r2.unit();
        Vector3f target(r2.range(-2, 2), r2.range(-2, 2), r2.range(0, 4));
        push3(sp_in, is.p); push3(sp_in, is.pError); push3(sp_in, is.n); push3(sp_in, w); push3(sp_in, target);
        Ray a = is.spawnRay(w), b = is.spawnRayTo(target);
        push3(sp_out, a.origin); push3(sp_out, b.origin); push3(sp_out, b.direction);
    }
    save(tag + "_spawn_in", sp_in, {sp_in.size() / 15, 5, 3});
    save(tag + "_spawn_out", sp_out, {sp_out.size() / 9, 3, 3});

    // AreaLight::sample_Le (the triangle of the light mesh is picked with an unseeded random
    // number inside the call: the output is one of the mesh's triangles' samples)
    if (!sc->get_lights().empty()) {
        std::vector<float> u1, u2, lo, ld, ln, le, pdfs;
        Rng r3(seed + 2);
        for (int i = 0; i < 2048; ++i) {
            Vector2f a(r3.u(), r3.u()), b(r3.u(), r3.u());
            Ray ray;
            Vector3f nl;
            Real pp = 0, pd = 0;
            Spectrum L = sc->get_lights()[0]->sample_Le(a, b, 0, &ray, &nl, &pp, &pd);
            u1.push_back(a.x); u1.push_back(a.y); u2.push_back(b.x); u2.push_back(b.y);
            push3(lo, ray.origin); push3(ld, ray.direction); push3(ln, nl); pushs(le, L);
            pdfs.push_back(pp); pdfs.push_back(pd);
        }
        save(tag + "_le_u1", u1, {2048, 2}); save(tag + "_le_u2", u2, {2048, 2});
        save(tag + "_le_org", lo, {2048, 3}); save(tag + "_le_dir", ld, {2048, 3}); save(tag + "_le_n", ln, {2048, 3});
        save(tag + "_le_L", le, {2048, 3}); save(tag + "_le_pdf", pdfs, {2048, 2});
    }
    delete cfg->tracer;
    delete cfg->scene;
    delete cfg;
}

// -------------------------------------------------------------- materials
static void probe_materials() {
    Rng r(21);
    const int n = 4096;
    std::vector<float> wo, nn, u;
    for (int i = 0; i < n; ++i) {
        Vector3f N = r.unit(), w = r.unit();
        if (i % 5 == 0) w = normalize(N + w * 0.05f);            // near-normal incidence
        if (i % 7 == 0) w = normalize(w - N * dotProduct(w, N) + N * 0.01f);   // grazing
        push3(wo, w); push3(nn, N);
        u.push_back(r.u()); u.push_back(r.u());
    }
    save("mat_wo", wo, {(size_t)n, 3}); save("mat_n", nn, {(size_t)n, 3}); save("mat_u", u, {(size_t)n, 2});
    struct M { const char* tag; Material* m; };
    DiffuseMaterial diff(Spectrum(0.725f, 0.71f, 0.68f));
    MirrorMaterial mirr(Spectrum(1.f, 0.9f, 0.8f));
    GlassMaterial glass(Spectrum(1.f, 1.f, 1.f), Spectrum(0.9f, 1.f, 0.95f), 1.0f, 1.75496f);
    M mats[3] = {{"diffuse", &diff}, {"mirror", &mirr}, {"glass", &glass}};
    for (const M& mm : mats)
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<float> wi, f, pdf, fe;
            mm.m->setTransportMode(mode == 0 ? Radiance : Importance);
            for (int i = 0; i < n; ++i) {
                Vector3f w(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), N(nn[3 * i], nn[3 * i + 1], nn[3 * i + 2]), o(0);
                float p = 0;
                Spectrum s = mm.m->sample_f(w, &o, N, Vector2f(u[2 * i], u[2 * i + 1]), &p);
                push3(wi, o); pushs(f, s); pdf.push_back(p);
                // f() for an independent direction pair (the second direction: wo of the next sample)
                const int j = (i + 1) % n;
                pushs(fe, mm.m->f(w, Vector3f(wo[3 * j], wo[3 * j + 1], wo[3 * j + 2]), N));
            }
            std::string t = std::string("mat_") + mm.tag + (mode == 0 ? "_radiance" : "_importance");
            save(t + "_wi", wi, {(size_t)n, 3}); save(t + "_f", f, {(size_t)n, 3}); save(t + "_pdf", pdf, {(size_t)n});
            save(t + "_feval", fe, {(size_t)n, 3});
        }
}

// ------------------------------------------------------------------ camera
static Matrix4 mat_from(const float* m) {
    return Matrix4(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
}
static void probe_camera(const std::vector<float>& cam) {
    Matrix4 c2w = mat_from(&cam[0]), c2s = mat_from(&cam[16]);
    BoxFilter filter(0.5, 0.5);
    const int sizes[3] = {512, 784, 1024};
    for (int si = 0; si < 3; ++si) {
        const int W = sizes[si];
        Film film(Point2i(W, W), Vector2f(0), Vector2f(1.0f, 1.0f), &filter);
        Camera camera(&c2w, c2s, 0.00, 2.0, &film);   // src/main.cpp:46
        std::vector<float> r2c(camera.rasterToCamera.m, camera.rasterToCamera.m + 16), raster, org, dir;
        Rng r(31 + si);
        for (int i = 0; i < 2048; ++i) {
            Vector2f f(r.range(0, (float)W), r.range(0, (float)W));
            if (i < 4) f = Vector2f(i & 1 ? (float)W : 0.f, i & 2 ? (float)W : 0.f);
            else if (i < 64) f = Vector2f((float)r.below(W) + 0.5f, (float)r.below(W) + 0.5f);
            Ray ray;
            camera.generateRay(f, Vector2f(0.5f, 0.5f), &ray);
            raster.push_back(f.x); raster.push_back(f.y);
            push3(org, ray.origin); push3(dir, ray.direction);
        }
        std::string t = "camera" + std::to_string(W);
        save(t + "_raster2cam", r2c, {16}); save(t + "_raster", raster, {2048, 2});
        save(t + "_org", org, {2048, 3}); save(t + "_dir", dir, {2048, 3});
    }
}

// --------------------------------------------------------------- traceTile
// SppmTracer::traceTile on given photon maps with an unjittered sampler: the
// camera paths, next-event estimation, both radiance estimates and the
// SPPMPixel update of two iterations, with nothing random left in them.
static void probe_trace_tile(const std::string& in, const std::vector<float>& cam) {
    const int W = 48, H = 48, kg = 100, kc = 50, recurse = 4;
    std::vector<float> gpos = load_raw<float>(in + "/g_pos.f32"), gpow = load_raw<float>(in + "/g_pow.f32");
    std::vector<float> cpos = load_raw<float>(in + "/c_pos.f32"), cpow = load_raw<float>(in + "/c_pow.f32");
    std::vector<uint8_t> gdir = load_raw<uint8_t>(in + "/g_dir.u8"), cdir = load_raw<uint8_t>(in + "/c_dir.u8");
    const int ng = (int)gdir.size() / 2, nc = (int)cdir.size() / 2;
    Config* cfg = load("cornell_sphere.scene");
    delete cfg->tracer;
    SPPMParam prm;
    prm.numRecurse = recurse; prm.globalSize = ng; prm.globalSample = kg; prm.causticSize = nc; prm.causticSample = kc;
    prm.radius2 = 0.08f;
    SppmTracer* tr = new SppmTracer(1, 20, 1, 0.8f, prm);
    Matrix4 c2w = mat_from(&cam[0]), c2s = mat_from(&cam[16]);
    BoxFilter filter(0.5, 0.5);
    Film film(Point2i(W, H), Vector2f(0), Vector2f(1.0f, 1.0f), &filter);
    Camera camera(&c2w, c2s, 0.00, 2.0, &film);
    tr->setScene(cfg->scene);
    tr->setCamera(&camera);
    // what startRaytracing sets, without starting worker threads (src/tracer/TkSppmTracer.cpp:385-398)
    tr->shootGlobalPhotons = tr->shootCausticPhotons = 0;
    tr->curRecurse = 0;
    tr->pp.clear();
    tr->pp.resize((size_t)W * H, SPPMPixel(tr->radius2));
    s32 sg = 0, scs = 0;
    for (int i = 0; i < ng; ++i)
        tr->globalPhotonMap.storePhoton(Vector3f(gpos[3 * i], gpos[3 * i + 1], gpos[3 * i + 2]),
                                        Spectrum(gpow[3 * i], gpow[3 * i + 1], gpow[3 * i + 2]), gdir[2 * i], gdir[2 * i + 1], sg);
    for (int i = 0; i < nc; ++i)
        tr->causticPhotonMap.storePhoton(Vector3f(cpos[3 * i], cpos[3 * i + 1], cpos[3 * i + 2]),
                                         Spectrum(cpow[3 * i], cpow[3 * i + 1], cpow[3 * i + 2]), cdir[2 * i], cdir[2 * i + 1], scs);
    tr->storedGlobalPhotons = sg; tr->storedCausticPhotons = scs;
    tr->globalPhotonMap.buildKdTree();
    tr->causticPhotonMap.buildKdTree();
    int xs = 0, ys = 0;
    while (++xs * ++ys < recurse) {}   // the strata of the tracer's sampler (src/tracer/TkSppmTracer.cpp:18-21)
    StratifiedSampler sampler(xs, ys, false, 0);
    for (int it = 0; it < 2; ++it) {
        tr->curRecurse = it;
        tr->traceTile(Point2i(0, 0), Point2i(W, H), sampler);
        std::vector<float> px;
        for (const SPPMPixel& p : tr->pp) {
            pushs(px, p.Ld); pushs(px, p.causticFlux); pushs(px, p.globalFlux);
            px.push_back(p.causticPhotons); px.push_back(p.globalPhotons);
            px.push_back(p.causticRadius2); px.push_back(p.globalRadius2);
        }
        save("tile_pixels_iter" + std::to_string(it), px, {(size_t)W * H, 13});
    }
    std::vector<int32_t> meta = {W, H, kg, kc, recurse, ng, nc};
    save("tile_meta", meta, {meta.size()});
    std::vector<float> num = {(float)tr->numRecurse};
    save("tile_num_recurse", num, {1});
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_probe <in_dir> <out_dir>\n"); return 1; }
    const std::string in = argv[1];
    g_out = argv[2];
    std::vector<float> cam = load_raw<float>(in + "/camera.f32");
    if (cam.size() != 32) { fprintf(stderr, "camera.f32 must hold 32 floats\n"); return 1; }
    probe_lowdiscrepancy();
    probe_photon_map();
    probe_materials();
    probe_camera(cam);
    probe_rays("cornell_box.scene", "rays_box", 20000, 101);
    probe_rays("cornell_sphere.scene", "rays_sphere", 20000, 202);
    probe_rays("cornell_dragon.scene", "rays_dragon", 40000, 303);
    probe_trace_tile(in, cam);
    fprintf(stderr, "ref_probe: done\n");
    return 0;
}
