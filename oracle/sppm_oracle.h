/*
 * sppm_oracle.h — CPU restatement of the reference SPPM path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libtkhip.so, libtkhost.so) never does.
 *
 * PINNED: the reference ships no tests or fixtures of its own, but the part
 * of it this path uses compiles without a window (oracle/build_ref.sh: its own
 * translation units and its own vendored GL/glew.h, nothing stood in).  A
 * prober (oracle/ref_probe.cpp) calls the reference's functions on seeded
 * inputs; the outputs are committed as tests/golden/ref_*.npz
 * (oracle/make_ref_golden.py) and tests/test_ref_pinned_cpu.py holds this
 * restatement to them: bit-exact for hits, indices, radii and counts.
 * Not pinned, because no reference output can exist for them: the camera
 * matrices of CameraInfo (src/TkRenderer.cpp needs GL/glu.h), the photon
 * loops as a whole (unseeded random numbers) and the rule for photons tied at
 * the k-th distance.
 *
 * The scene/parameter/pixel structs are the data formats of include/tkhip.h.
 */
#ifndef SPPM_ORACLE_H
#define SPPM_ORACLE_H
#include "../include/tkhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;
typedef struct orc_map orc_map;
typedef struct orc_sppm orc_sppm;

/* --- scalar pieces --- */
float orc_radical_inverse(int32_t base_index, uint64_t a);
uint64_t orc_hash_pixel(int32_t x, int32_t y, int32_t dim, int32_t seed);
int32_t orc_permutation_element(uint32_t i, uint32_t l, uint32_t p);
float orc_rng_u01(uint32_t seed, uint32_t a, uint32_t b, uint32_t c);
/* what a light path / a PathTracer::Li sample draws where the reference calls get_random_float(), in its call order */
void orc_photon_draws(uint32_t seed, int kind, uint64_t H0, int64_t n, int per, float* out);
void orc_pt_draws(uint32_t seed, int sample, int64_t n, int per, float* out);
void orc_direction_tables(float* cos_theta, float* sin_theta, float* cos_phi, float* sin_phi);
void orc_encode_direction(const float wo[3], uint8_t* theta, uint8_t* phi);
void orc_pixel_update(int64_t n, float* flux, float* photons, float* radius2,
                      const float* lindir, const int32_t* found, const float* radius2_ret);
void orc_resolve_rgb8(const TkSPPMPixel* pp, int w, int h, int num_recurse,
                      int64_t shoot_global, int64_t shoot_caustic, int caustic_stored,
                      uint8_t* out);

/* --- scene, rays, camera --- */
orc_scene* orc_scene_create(const TkObject* objects, int n_objects, const TkTriangle* tris,
                            int n_tris, const TkSphere* spheres, int n_spheres,
                            const TkMaterial* materials, int n_materials,
                            const TkLight* lights, int n_lights);
void orc_scene_destroy(orc_scene* s);
void orc_scene_set_camera(orc_scene* s, const float c2w[16], const float raster2cam[16],
                          int w, int h);
/* faithful = 1: two-level BVH, far child first, no t_max culling (src/BVH.cpp:245-288).
 * faithful = 0: same BVHs, near child first with t_max culling.  Same hits. */
void orc_trace_rays(orc_scene* s, int64_t n, const float* org, const float* dir,
                    const float* tmax, int32_t* prim, float* t, float* p, float* nrm,
                    float* perr, int faithful);
/* Scene::uniformSampleOneLight (src/core/Scene.cpp:43-61) at the first hit of each ray; u: per ray
 * the four numbers the call takes from its sampler (light, point on it x2, triangle of its mesh). */
void orc_sample_one_light(orc_scene* s, int64_t n, const float* org, const float* dir, const float* u,
                          int32_t* prim, float* ld, int faithful);
void orc_camera_rays(orc_scene* s, int64_t n, const float* raster, float* org, float* dir);
/* Material::sample_f for n samples: mode 0 Radiance, 1 Importance. */
void orc_sample_f(const TkMaterial* m, int mode, int64_t n, const float* wo, const float* nrm,
                  const float* u, float* wi, float* f, float* pdf);
/* AreaLight::sample_Le on light `li` with explicit triangle-pick random number. */
void orc_sample_le(orc_scene* s, int li, int64_t n, const float* u1, const float* u2,
                   const float* upick, float* org, float* dir, float* nrm, float* le,
                   float* pdf_pos, float* pdf_dir);

/* --- photon map --- */
orc_map* orc_map_create(int64_t capacity, int k);
void orc_map_destroy(orc_map* m);
void orc_map_reset(orc_map* m);
int64_t orc_map_store(orc_map* m, const float* pos, const float* power, const uint8_t* theta,
                      const uint8_t* phi, int64_t n);
/* split_mode 0: reference behaviour, every interior node splits on z.
 * split_mode 1: widest axis of the node's photon bounds. */
void orc_map_build(orc_map* m, int split_mode);
/* radiance_estimate through neighbor_search (reference traversal order). */
void orc_map_gather(orc_map* m, int64_t nq, const float* p, const float* n, const float* wo,
                    const float* R, const float* radius2, float* flux, int32_t* found,
                    float* radius2_out, int32_t* idx);
/* The same estimate defined by exhaustive search with the (dist2, index) total
 * order: the k smallest among dist2 < radius2; ties by lower photon index. */
void orc_map_gather_brute(orc_map* m, int64_t nq, const float* p, const float* n,
                          const float* wo, const float* R, const float* radius2, float* flux,
                          int32_t* found, float* radius2_out, int32_t* idx);
int64_t orc_map_split_histogram(orc_map* m, int64_t hist[5]);
/* by_index 1: among photons tied at the k-th distance orc_map_gather keeps the
 * lowest indices (the GPU path's rule) instead of the ones the traversal met
 * first (the reference's, src/tracer/neighbor.cpp:31-83).  Count and radius are
 * the same either way. */
void orc_map_set_tie_rule(orc_map* m, int by_index);

/* --- full SPPM --- */
/* rng_cost 0: counter-based RNG only.  rng_cost 1: additionally seeds and
 * twists a 624-word Mersenne Twister per random number, the cost the
 * reference pays in get_random_float (src/TkMath.h:312-319).
 * split_mode / faithful_bvh as above.  threads = worker count. */
orc_sppm* orc_sppm_begin(orc_scene* s, const TkSppmParam* param, int max_depth, float rr,
                         uint64_t seed, int rank, int world, int threads, int split_mode,
                         int faithful_bvh, int rng_cost);
void orc_sppm_destroy(orc_sppm* r);
/* jitter 0: stratum centres, as StratifiedSampler(x, y, jitter = false) (src/core/sampler.cpp:51, :61-62). */
void orc_sppm_set_jitter(orc_sppm* r, int jitter);
/* Traversal work of the passes so far: rays, box tests, primitive tests of the photon passes
 * (out[0..2]) and of the camera passes (out[3..5]); src/BVH.cpp:245-288 counts one box test per
 * node popped and one primitive test per Object::intersect of a leaf. */
void orc_sppm_trace_counters(orc_sppm* r, int64_t out[6]);
void orc_sppm_set_tie_rule(orc_sppm* r, int by_index);   /* orc_map_set_tie_rule for both maps */
void orc_sppm_photon_pass(orc_sppm* r, int iter, TkIterStats* st);
void orc_sppm_build_maps(orc_sppm* r, TkIterStats* st);
/* y0,y1: restrict the camera pass to rows [y0,y1) (bounded CPU baseline sample);
 * pass 0,height for the full image. */
void orc_sppm_camera_pass(orc_sppm* r, int iter, int y0, int y1, TkIterStats* st);
void orc_sppm_iteration(orc_sppm* r, int iter, TkIterStats* st);
void orc_sppm_read_pixels(orc_sppm* r, TkSPPMPixel* out);
void orc_sppm_write_pixels(orc_sppm* r, const TkSPPMPixel* in);
void orc_sppm_resolve(orc_sppm* r, uint8_t* out);
int64_t orc_sppm_read_photons(orc_sppm* r, int which, float* pos, float* power, uint8_t* theta,
                              uint8_t* phi, int64_t* shot_local);
void orc_sppm_set_map_photons(orc_sppm* r, int which, const float* pos, const float* power,
                              const uint8_t* theta, const uint8_t* phi, int64_t n,
                              int64_t shot_total);
void orc_sppm_counters(orc_sppm* r, int64_t* shoot_global, int64_t* shoot_caustic,
                       int32_t* num_recurse);
/* One camera path for pixel (x,y) of iteration iter with its gather queries
 * reported instead of executed: used to compare hitpoints with the GPU. */
typedef struct OrcQuery {
    float p[3], n[3], wo[3], coef[3], R[3];
    int32_t pixel;
    int32_t which; /* 0 global, 1 caustic */
} OrcQuery;
int orc_sppm_camera_path(orc_sppm* r, int iter, int x, int y, float ld_out[3], OrcQuery q[2]);

/* --- hash-grid variant, tk::SPPM (src/TkSPPM.cpp) on an orc_sppm begun with causticSize 0.
 * Not part of the reference's build, so unpinned beyond the shared building blocks.  Pixel fields:
 * Ld = Ld, causticFlux = Tau, causticPhotons = N, causticRadius2 = radius (not squared).
 * stats: visible points, grid entries, photon deposits. */
void orc_hsppm_iteration(orc_sppm* r, int spp, int iter, int64_t stats[3]);
void orc_hsppm_resolve(orc_sppm* r, uint8_t* out);
void orc_hg_cells(const float bounds[6], const int32_t res[3], uint32_t buckets, int64_t n, const float* points,
                  int32_t* cell, int32_t* inside, int32_t* hash);

/* --- path tracer, PathTracer::Li + the headless program's film (box filter of radius 0.5) on the
 * scene, depth, roulette probability and seed of an orc_sppm.  contrib: 3 floats per pixel,
 * wsum: 1 (Pixel::contribSum / filterWeightSum); samples [s0, s1) of spp. */
void orc_pt_render(orc_sppm* r, int spp, int s0, int s1, float* contrib, float* wsum);
void orc_pt_resolve(const float* contrib, const float* wsum, int w, int h, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
