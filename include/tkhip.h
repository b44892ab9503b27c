/*
 * tkhip.h — C ABI of the MI355X (gfx950) SPPM backend.
 *
 * Plain C types only: POD structs, pointers and sizes.  No exception crosses
 * this boundary; every function returns 0 on success and a negative TKHIP_E_*
 * code on failure, with a human-readable reason available from
 * tkhip_last_error().  Citations "file:line" refer to the upstream CPU
 * renderer (tohkachan/PathTracer, src/...) whose interface each entry point
 * replaces.
 *
 * Pointer conventions
 *   "host"   : ordinary process memory, read or written synchronously.
 *   "device" : memory of the context's GPU (hipMalloc / torch CUDA tensor
 *              data_ptr).  Calls that take device pointers enqueue work on the
 *              context's stream and synchronise before returning unless the
 *              function's comment says otherwise.
 */
#ifndef TKHIP_H
#define TKHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ errors */
#define TKHIP_OK 0
#define TKHIP_E_ARG (-1)     /* bad argument (null pointer, size out of range) */
#define TKHIP_E_STATE (-2)   /* call order violated (e.g. no scene set)        */
#define TKHIP_E_HIP (-3)     /* a HIP runtime call or kernel launch failed     */
#define TKHIP_E_NOMEM (-4)   /* allocation failed                              */
#define TKHIP_E_LIMIT (-5)   /* size beyond a documented limit                 */

/* Material type flags: src/core/Material.hpp:14 */
#define TK_DIFFUSE 0x1
#define TK_MIRROR 0x2
#define TK_GLASS 0x4

/* Largest kNN sample count a photon map accepts (gsample/csample). */
#define TKHIP_MAX_K 448

/* ------------------------------------------------------------- scene PODs */

/* One triangle in world space (vertices already transformed):
 * src/Triangle.hpp:12-25 (v0,v1,v2; e1,e2,normal,area are derived). */
typedef struct TkTriangle {
    float v0[3];
    float v1[3];
    float v2[3];
} TkTriangle;

/* src/Sphere.hpp:15-16 */
typedef struct TkSphere {
    float center[3];
    float radius;
} TkSphere;

/* src/core/Material.hpp:106-146.  type is one TK_* flag.
 * diffuse: R.  mirror: R.  glass: R, T, etaA (outside, 1.0), etaB (inside). */
typedef struct TkMaterial {
    int32_t type;
    float R[3];
    float T[3];
    float etaA;
    float etaB;
} TkMaterial;

/* src/AreaLight.hpp:13-37: an area light bound to one mesh object. */
typedef struct TkLight {
    float Le[3];
    int32_t two_sided; /* the loader always passes 0: src/TkLoader.cpp:336 */
    int32_t object;    /* index of the mesh object that emits */
} TkLight;

/* src/core/Object.hpp: shape + material + optional light.
 * kind 0: mesh  = triangles [first, first+count) of the triangle array.
 * kind 1: sphere = spheres  [first, first+1)     of the sphere array. */
typedef struct TkObject {
    int32_t kind;
    int32_t first;
    int32_t count;
    int32_t material;
    int32_t light; /* -1 if the object does not emit */
} TkObject;

/* src/TkConfigure.h:29-37.  radius2 is used directly as r^2
 * (src/tracer/TkSppmTracer.cpp:14, :395). */
typedef struct TkSppmParam {
    int32_t numRecurse;
    int32_t globalSize;
    int32_t globalSample;
    int32_t causticSize;
    int32_t causticSample;
    float radius2;
} TkSppmParam;

/* src/tracer/TkSppmTracer.h:9-19, same field order, 13 floats = 52 bytes. */
typedef struct TkSPPMPixel {
    float Ld[3];
    float causticFlux[3];
    float globalFlux[3];
    float causticPhotons;
    float globalPhotons;
    float causticRadius2;
    float globalRadius2;
} TkSPPMPixel;

/* Counters of one iteration (or of one pass when the passes are called
 * separately); also the per-phase device time in milliseconds. */
typedef struct TkIterStats {
    int64_t global_paths;     /* light paths traced for the global map (this call)  */
    int64_t caustic_paths;    /* light paths traced for the caustic map (this call) */
    int64_t global_stored;    /* photons in the global map                          */
    int64_t caustic_stored;   /* photons in the caustic map                         */
    int64_t shoot_global;     /* cumulative shootGlobalPhotons (TkSppmTracer.h:31)  */
    int64_t shoot_caustic;    /* cumulative shootCausticPhotons (TkSppmTracer.h:32) */
    int64_t rays;             /* closest-hit + shadow queries (this call)           */
    int64_t gathers_global;   /* radiance_estimate calls on the global map          */
    int64_t gathers_caustic;  /* radiance_estimate calls on the caustic map         */
    int64_t gather_photons;   /* sum of photons found over all gathers              */
    int64_t overflow_global;  /* global gathers that met >= k photons and went to   */
    int64_t overflow_caustic; /*   the per-query selection kernel (same, caustic)   */
    int64_t group_staged;     /* photons the grouped gather kernel staged, summed   */
                              /*   over its waves (x 64 lanes = distance tests)     */
    float ms_photon;          /* photon passes + compaction                         */
    float ms_build;           /* both KD-tree builds                                */
    float ms_camera;          /* camera paths incl. NEE, without gathers            */
    float ms_gather;          /* all gather kernels incl. query sort + pixel update */
    float ms_gather_kernel;   /* the dominant (grouped) gather kernel only          */
    float ms_total;
    int64_t rays_camera;      /* of `rays`: those of the camera paths (the rest are light paths) */
    /* Filled only while tkhip_sppm_set_counting is on: what those rays cost on the flattened BVH that is   */
    /* walked (BVHAccel::intersect, src/BVH.cpp:245-288): nodes fetched (a node holds both child boxes, 64  */
    /* bytes; a node of the eight-children copy counts as four) and primitives tested (48 bytes each).      */
    int64_t node_visits_camera;
    int64_t prim_tests_camera;
    int64_t node_visits_light;
    int64_t prim_tests_light;
} TkIterStats;

typedef struct tkhip_ctx tkhip_ctx;
typedef struct tkhip_map tkhip_map;

/* --------------------------------------------------------------- lifecycle */

/* Thread-local text of the last failure in this thread. */
const char* tkhip_last_error(void);

/* Library build tag (used by loaders to check that the right .so is loaded). */
const char* tkhip_version(void);

/* One context per GPU.  Owns a stream and every device buffer created through
 * it.  Replaces the tracer object construction in parse_tracer,
 * src/TkLoader.cpp:228-251. */
int tkhip_create(int device, tkhip_ctx** out);
/* GPUs this process can see (0 when there is none or the runtime fails). */
int tkhip_device_count(void);
void tkhip_destroy(tkhip_ctx* ctx);

/* --------------------------------------------------------- scene and camera */

/* Flattens the scene into one linear BVH over every triangle and sphere and
 * uploads it.  Replaces Scene::buildBVH + the per-mesh BVHAccel
 * (src/core/Scene.cpp:18-20, src/Triangle.cpp:179, src/BVH.cpp:62-243) and
 * RayTracer::setScene (src/core/TkRayTracer.cpp:36-41).  Host pointers. */
int tkhip_set_scene(tkhip_ctx* ctx,
                    const TkObject* objects, int n_objects,
                    const TkTriangle* tris, int n_tris,
                    const TkSphere* spheres, int n_spheres,
                    const TkMaterial* materials, int n_materials,
                    const TkLight* lights, int n_lights);

/* Who builds the scene hierarchy (buildBVH, src/BVH.cpp:62-243, called per
 * mesh from src/Triangle.cpp:103-180 and for the scene from
 * src/core/Scene.cpp:18-20): 0 (default) the GPU for scenes of 4096 primitives
 * or more and host threads below, 1 host threads (binned SAH), 2 the GPU
 * (Morton order, SAH over sixteen cuts of each run).  Takes effect at the next
 * tkhip_set_scene.  The closest hit of a ray does not depend on the choice.
 * tkhip_bvh_info reports the tree of the current scene and the wall time of its
 * construction (any pointer may be null). */
int tkhip_set_bvh_builder(tkhip_ctx* ctx, int mode);
/* How the last rays of a pass are walked through the BVH (no counterpart in the reference, whose
 * BVHAccel::intersect, src/BVH.cpp:245-288, has one walk).  A wave of a path kernel that is left with at most
 * max_rays rays walks them eight lanes to a ray in a copy of the tree with eight children per node (built by
 * tkhip_set_scene for trees of up to 2^18 nodes): the same boxes, the same primitive tests, the same hit, a third
 * of the dependent memory round trips.  Default 12; 0: never; 64: every ray begun once the pass's queue is empty, and
 * every ray of tkhip_trace_rays (what the parity tests use).  Takes effect with the next launch. */
int tkhip_set_wide_walk(tkhip_ctx* ctx, int max_rays);
int tkhip_bvh_info(tkhip_ctx* ctx, int32_t* n_nodes, int32_t* depth, int32_t* built_on_device,
                   float* build_ms);

/* Row-major 4x4 matrices as the reference stores them (src/Matrix4.h:11-15):
 * c2w = CameraInfo::c2w after rotate(0,0)+setTranslation (src/main.cpp:39-40),
 * raster2cam = Camera::rasterToCamera (src/core/camera.cpp:15-20).
 * Replaces RayTracer::setCamera (src/core/TkRayTracer.cpp:43-52). */
int tkhip_set_camera(tkhip_ctx* ctx, const float c2w[16], const float raster2cam[16],
                     int width, int height);

/* ------------------------------------------------------------ SPPM render */

/* SppmTracer ctor + startRaytracing (src/tracer/TkSppmTracer.cpp:11-29,
 * :385-398): rounds numRecurse up to x*y, allocates both photon maps and the
 * SPPMPixel array (every pixel's radius2 = param->radius2), zeroes counters.
 * rank/world partition the work of one iteration over `world` processes, one
 * GPU each: rank r emits the photon paths with Halton index = r (mod world)
 * and fills a quota of size/world photons per map, and owns the pixel rows
 * y with (y / 8) % world == r.  world = 1 is the single-GPU case. */
int tkhip_sppm_begin(tkhip_ctx* ctx, const TkSppmParam* param, int max_depth,
                     float russian_roulette, uint64_t seed, int rank, int world);

/* Photon phase of iteration `iter` (tracePhoton / traceCausticPhoton loops,
 * src/tracer/TkSppmTracer.cpp:104-217, :456-470): fills this rank's quota of
 * both maps.  stats may be null. */
int tkhip_sppm_photon_pass(tkhip_ctx* ctx, int iter, TkIterStats* stats);

/* Copies this rank's freshly traced photons of map `which` (0 global,
 * 1 caustic) into caller-owned DEVICE buffers for the exchange between ranks:
 * pos and power are float[3*cap] (xyz / rgb interleaved), dir is uint16[cap]
 * (theta | phi << 8), cap >= this rank's quota.  *n = photons copied,
 * *shot = light paths this rank traced to fill its quota. */
int tkhip_sppm_export_local_photons(tkhip_ctx* ctx, int which, float* pos, float* power,
                                    uint16_t* dir, int64_t cap, int64_t* n, int64_t* shot);

/* Installs the photons of ALL ranks (device pointers, same layout) as map
 * `which` for the coming camera pass and adds shot_total to the cumulative
 * shoot counter.  With world = 1 this is done by tkhip_sppm_photon_pass. */
int tkhip_sppm_set_map_photons(tkhip_ctx* ctx, int which, const float* pos, const float* power,
                               const uint16_t* dir, int64_t n, int64_t shot_total);

/* The same exchange without the host in it, for a driver that moves ONE buffer per rank with one collective
 * (an all-gather): no counterpart in the reference, whose threads share one PhotonMap behind a mutex
 * (src/tracer/TkSppmTracer.cpp:139-141).  A rank's buffer is *words_per_rank 32-bit words (the same on every
 * rank): a header with its photon counts and shot counters, then both maps' arrays.
 * tkhip_sppm_export_exchange writes this rank's buffer (DEVICE memory) on the caller's stream (a hipStream_t,
 * 0 = the null stream) and does not wait: the collective the caller enqueues behind it on that stream is
 * ordered, and the rank's next photon pass waits for the copies.
 * tkhip_sppm_import_exchange takes all ranks' buffers back to back (world * words_per_rank words, DEVICE, as
 * the all-gather delivers them on `stream`), waits until `stream` has got that far and installs both maps on
 * the device: the host reads nothing of the buffer, the counts come from the headers on the device, the host
 * assumes that every rank met its quota and enqueues the builds at once;
 * tkhip_sppm_build_maps then checks that assumption against what the device summed (and rebuilds a map that
 * came out smaller), and adds the ranks' shot counters to the cumulative ones. */
int tkhip_sppm_exchange_words(tkhip_ctx* ctx, int64_t* words_per_rank);
int tkhip_sppm_export_exchange(tkhip_ctx* ctx, int32_t* device_buf, void* stream);
int tkhip_sppm_import_exchange(tkhip_ctx* ctx, const int32_t* device_gathered, void* stream);

/* PhotonMap::buildKdTree for both maps (src/tracer/Photonmap.cpp:96-103,
 * :132-179): the search tree of each map, here a balanced binary tree over the
 * photons in Hilbert-curve order with tight boxes (a gather's result does not
 * depend on the shape of the tree). */
int tkhip_sppm_build_maps(tkhip_ctx* ctx, TkIterStats* stats);

/* tkhip_map_set_tree for both maps of the render. */
int tkhip_sppm_set_tree(tkhip_ctx* ctx, int mode);

/* Material::sample_f for n (wo, N, u) triples on one material
 * (src/core/Material.cpp:22-95; DiffuseMaterial, MirrorMaterial, GlassMaterial).
 * mode 0 = TransportMode Radiance, 1 = Importance.  Host buffers: wo, nrm, wi, f
 * 3 floats per sample, u 2, pdf 1.  For parity tests. */
int tkhip_sample_f(tkhip_ctx* ctx, const TkMaterial* material, int mode, int64_t n, const float* wo,
                   const float* nrm, const float* u, float* wi, float* f, float* pdf);

/* AreaLight::sample_Le on light `light` of the scene (src/Light.cpp:41-75) with
 * the triangle pick of MeshTriangle::Sample (src/Triangle.cpp:190-197) drawn
 * from upick: u1 places the point, u2 the direction.  Returns the spawned ray
 * (offset origin, direction), the light normal, Le and the two pdfs.  Host
 * buffers.  For parity tests. */
int tkhip_sample_le(tkhip_ctx* ctx, int light, int64_t n, const float* u1, const float* u2,
                    const float* upick, float* org, float* dir, float* nrm, float* le, float* pdf_pos,
                    float* pdf_dir);

/* Running totals of the grouped gather kernel (the kernel that answers
 * radiance_estimate, src/tracer/Photonmap.cpp:105-130, for the queries that
 * meet fewer than k photons) over all tkhip_sppm_* gathers since the last
 * reset: every launch is bracketed by events on its stream.  For benchmarks. */
typedef struct TkGatherTotals {
    int64_t launches;            /* grouped-kernel launches (one per map and gather) */
    int64_t queries;             /* queries given to them */
    int64_t queries_handed_on;   /* of those, finished by the per-query kernel instead */
    int64_t photons_found;       /* sum of `found` over all queries (both kernels) */
    int64_t photons_staged;      /* photons the grouped kernel staged, summed over waves */
    int64_t photons_found_k[2];      /* per map: global, caustic */
    int64_t queries_handed_on_k[2];
    double ms_group_kernel;      /* summed event-timed durations of the launches */
    double ms_group_side_by_side;    /* the two maps' launches of a step run side by side: first begin to last end, summed over the steps */
    int64_t steps_side_by_side;      /* steps in that sum */
} TkGatherTotals;
int tkhip_sppm_gather_totals(tkhip_ctx* ctx, TkGatherTotals* out, int reset);

/* Starts the photon passes of iteration `iter` (the first batch of light paths
 * of both maps) and returns at once; tkhip_sppm_photon_pass(iter) then picks
 * them up.  Optional, like tkhip_sppm_camera_begin below: a multi-GPU driver
 * calls this first, then tkhip_sppm_camera_begin, so that the light paths (which
 * the exchange waits for) are ahead of the camera paths in the GPU's queues.
 * No counterpart in the reference, whose photon pass and camera pass follow
 * each other (src/tracer/TkSppmTracer.cpp:411-421, :480-496). */
int tkhip_sppm_photon_begin(tkhip_ctx* ctx, int iter);

/* The camera sampler of the render: jitter 1 (default) jitters inside the
 * strata, jitter 0 takes the stratum centres, the reference's
 * StratifiedSampler(x, y, jitter = false) (src/core/sampler.h:37-46,
 * src/core/sampler.cpp:51, :61-62).  With jitter 0 nothing random is left in a
 * camera pass. */
int tkhip_sppm_set_jitter(tkhip_ctx* ctx, int jitter);

/* on != 0: from the next pass on the path kernels run as counting instances (the same code with two
 * more counters per lane) and TkIterStats reports node_visits_* / prim_tests_*, the figures the ray
 * roofline is priced with.  Results are unchanged; the passes get slower, so timings are taken with
 * counting off.  Counts what BVHAccel::intersect does per ray (src/BVH.cpp:245-288) on this tree. */
int tkhip_sppm_set_counting(tkhip_ctx* ctx, int on);

/* Starts the camera paths of iteration `iter` (the part of traceTile that does
 * not read the photon maps, src/tracer/TkSppmTracer.cpp:296-321, :341-371) and
 * returns at once; tkhip_sppm_camera_pass(iter) then picks them up.  Optional:
 * lets a multi-GPU driver trace camera paths while photons are still being
 * traced and exchanged. */
int tkhip_sppm_camera_begin(tkhip_ctx* ctx, int iter);

/* traceTile over this rank's pixels (src/tracer/TkSppmTracer.cpp:290-374):
 * camera paths with next-event estimation, the caustic and global gathers
 * (radiance_estimate, src/tracer/Photonmap.cpp:105-130) and the per-pixel
 * flux / radius / photon-count update. */
int tkhip_sppm_camera_pass(tkhip_ctx* ctx, int iter, TkIterStats* stats);

/* tkhip_sppm_camera_pass(iter) that also starts the camera paths of iteration
 * iter + 1 before it gathers, into a second set of query buffers: on a rank of
 * several, whose share of the paths leaves most of its GPU idle, the next camera
 * paths (a chain of up to `depth` rays one after the other) run under this
 * iteration's gathers.  The pixels' radii those paths put into their queries are
 * stale by one update and are read again when the queries are gathered, so the
 * result is that of tkhip_sppm_camera_pass(iter) followed by
 * tkhip_sppm_camera_begin(iter + 1).  The caller commits to iteration iter + 1:
 * the direct light of its camera paths is already being added to the pixels.
 * Worker threads of the reference overlap tiles the same way
 * (src/tracer/TkSppmTracer.cpp:411-438). */
int tkhip_sppm_camera_pass_ahead(tkhip_ctx* ctx, int iter, TkIterStats* stats);

/* photon_pass + build_maps + camera_pass for world = 1: one state-machine
 * round of SppmTracer::render (src/tracer/TkSppmTracer.cpp:447-541). */
int tkhip_sppm_iteration(tkhip_ctx* ctx, int iter, TkIterStats* stats);

/* Copies the SPPMPixel array (width*height entries, row-major, y down as the
 * reference indexes pp[x + y*width]) to host memory. */
int tkhip_sppm_read_pixels(tkhip_ctx* ctx, TkSPPMPixel* host_out);

/* The SPPMPixel array as 13 planes of width*height floats (plane order =
 * field order of TkSPPMPixel) copied to / from a caller-owned DEVICE buffer,
 * for the image gather between ranks.  Rows not owned by a rank hold their
 * initial values there. */
int tkhip_sppm_export_planes(tkhip_ctx* ctx, float* device_dst);
int tkhip_sppm_import_planes(tkhip_ctx* ctx, const float* device_src);

/* Final colour, gamma 0.6, RGB8, rows written bottom-up
 * (src/tracer/TkSppmTracer.cpp:514-529).  host_out holds width*height*3 bytes.
 * iterations = the divisor of Ld (numRecurse in the reference). */
int tkhip_sppm_resolve_rgb8(tkhip_ctx* ctx, int iterations, uint8_t* host_out);

/* Progressive display (SppmTracer::rendering, src/tracer/TkSppmTracer.cpp:43-102,
 * the frame buffer RayTracer::updateScreen hands to glDrawPixels,
 * src/core/TkRayTracer.cpp:54-73).  The frame buffer lives on the device: one
 * RGBA8 word per pixel (red in the low byte, alpha 255), in the order of the
 * SPPMPixel array (row y of the raster, no flip), tone-mapped as the final image
 * with Ld divided by the rounded-up iteration count.
 * tkhip_sppm_set_preview(every_n): from now on tkhip_sppm_iteration /
 * tkhip_sppm_camera_pass refresh it after every every_n-th iteration (0: never),
 * enqueued behind the pass and not waited for: no host round trip per step.
 * tkhip_sppm_preview: waits for a pending refresh (refresh != 0: makes one now)
 * and returns the device pointer (valid until the render ends) and the last
 * iteration it shows (-1: none yet).  tkhip_sppm_read_preview copies it to
 * host memory (width*height words), e.g. to save an image mid-render. */
int tkhip_sppm_set_preview(tkhip_ctx* ctx, int every_n);
int tkhip_sppm_preview(tkhip_ctx* ctx, int refresh, const uint32_t** device_rgba, int* iteration);
int tkhip_sppm_read_preview(tkhip_ctx* ctx, uint32_t* host_rgba, int* iteration);

/* ------------------------------------------------ hash-grid variant: tk::SPPM
 * The reference's second SPPM class (src/TkSPPM.h:10-64, src/TkSPPM.cpp): per
 * iteration a ray pass that leaves one visible point per pixel (traceTile,
 * :136-179), a spatial hash grid over the visible points (SpatialHashGrids,
 * :19-112), a photon pass of photonsPerIt = param->globalSize light paths that
 * deposit into the pixels they reach (tracePhoton, :181-250) and the per-pixel
 * radius reduction (updatePixel, :252-272).  It is selected by swapping the
 * commented line at src/TkLoader.cpp:249-250 and is not in the reference's
 * build.  param->radius2 is the initial RADIUS here (SPPM's constructor reads it
 * so, :117), causticSize / the sample counts are unused; spp is RayTracer's spp
 * (a power of two), by which every direct-light term is divided.
 * tkhip_sppm_read_pixels then reports: Ld = Ld, causticFlux = Tau,
 * causticPhotons = N, causticRadius2 = the radius (not squared).
 * stats: rays, global_paths, shoot_global, gathers_caustic = visible points. */
int tkhip_hsppm_begin(tkhip_ctx* ctx, const TkSppmParam* param, int spp, int max_depth,
                      float russian_roulette, uint64_t seed);
int tkhip_hsppm_iteration(tkhip_ctx* ctx, int iter, TkIterStats* stats);
/* Final image of SPPM::render (src/TkSPPM.cpp:436-450): RGB8, rows bottom-up. */
int tkhip_hsppm_resolve_rgb8(tkhip_ctx* ctx, uint8_t* host_out);

/* ------------------------------------------------ path tracer: tk::PathTracer
 * <tracer type="pt"> of the reference (src/tracer/TkPathTracer.cpp): spp samples
 * per pixel from a stratified sampler of x * y strata (ctor, :13-24), each a
 * path with next-event estimation at every hit, emitted light seen on the
 * first hit and after specular bounces, Russian roulette (Li, :149-204), summed
 * into the film of the headless program: box filter of radius 0.5
 * (src/main.cpp:41, FilmTile::addSample, src/core/TkFilm.cpp:14-42), i.e. a
 * sample counts for its own pixel and, when it sits exactly on a pixel edge,
 * for the neighbour as well.  The reference renders tile by tile, all samples
 * of a tile at once (traceTile, :127-147); here a call renders the next
 * `count` samples of EVERY pixel, so an image can be shown after any call.
 * spp is RayTracer's (a power of two).  tkhip_sppm_read_pixels then reports
 * Ld = Pixel::contribSum and causticPhotons = Pixel::filterWeightSum. */
int tkhip_pt_begin(tkhip_ctx* ctx, int spp, int max_depth, float russian_roulette, uint64_t seed);
int tkhip_pt_render_samples(tkhip_ctx* ctx, int count, TkIterStats* stats);
/* Film::writeImage (src/core/TkFilm.cpp:124-153): RGB8, rows bottom-up. */
int tkhip_pt_resolve_rgb8(tkhip_ctx* ctx, uint8_t* host_out);

/* ------------------------------------------------ several GPUs, one process
 * The reference spreads an iteration over the threads of one machine (TkSpawn /
 * TkWait, src/Threads.cpp:19-56, called from src/tracer/TkSppmTracer.cpp:411-438
 * and :456-496); a group spreads it over the GPUs of one process instead: one
 * context and one host thread per GPU, work split as tkhip_sppm_begin(rank,
 * world) describes, the photons of the other ranks pulled with direct peer
 * copies between the two barriers of an iteration.  devices may name a GPU more
 * than once (a rehearsal of the multi-GPU path on fewer GPUs).
 * set_scene / set_camera / sppm_begin: the call of the same name on every
 * context at once.  sppm_iteration: one iteration on all GPUs; begin_next 1
 * also enqueues the light paths of iteration iter + 1 behind the exchange, under
 * this iteration's gathers; begin_next 2 starts its camera paths as well
 * (tkhip_sppm_camera_pass_ahead: the caller commits to iteration iter + 1, and
 * the pixels are in between two iterations until it has run); pass 0 on the last
 * iteration; stats is null or one TkIterStats per rank.  sppm_merge copies every rank's pixel rows into the
 * context of rank 0 (tkhip_group_ctx(g, 0)), on which tkhip_sppm_resolve_rgb8,
 * tkhip_sppm_read_pixels and tkhip_sppm_preview then see the whole image; the
 * render can go on afterwards. */
typedef struct tkhip_group tkhip_group;
int tkhip_group_create(const int* devices, int n, tkhip_group** out);
void tkhip_group_destroy(tkhip_group* g);
int tkhip_group_size(const tkhip_group* g);
tkhip_ctx* tkhip_group_ctx(tkhip_group* g, int rank);
int tkhip_group_set_scene(tkhip_group* g,
                          const TkObject* objects, int n_objects,
                          const TkTriangle* tris, int n_tris,
                          const TkSphere* spheres, int n_spheres,
                          const TkMaterial* materials, int n_materials,
                          const TkLight* lights, int n_lights);
int tkhip_group_set_camera(tkhip_group* g, const float c2w[16], const float raster2cam[16],
                           int width, int height);
int tkhip_group_sppm_begin(tkhip_group* g, const TkSppmParam* param, int max_depth,
                           float russian_roulette, uint64_t seed);
int tkhip_group_sppm_iteration(tkhip_group* g, int iter, int begin_next, TkIterStats* stats);
int tkhip_group_sppm_merge(tkhip_group* g);

/* Cumulative shoot counters and the rounded-up iteration count. */
int tkhip_sppm_counters(tkhip_ctx* ctx, int64_t* shoot_global, int64_t* shoot_caustic,
                        int32_t* num_recurse);

/* ------------------------------------------- kernel-level entry points
 * Used by the parity tests and by hosts that drive the phases themselves. */

/* A photon map of fixed capacity with sample count k
 * (PhotonMap::PhotonMap, src/tracer/Photonmap.cpp:20-34). */
int tkhip_map_create(tkhip_ctx* ctx, int64_t capacity, int k, tkhip_map** out);
void tkhip_map_destroy(tkhip_map* map);

/* storePhoton x n (src/tracer/Photonmap.cpp:44-53) from host arrays:
 * pos float[3n], power float[3n], theta u8[n], phi u8[n].  Photon index i is
 * its position in these arrays; kNN ties are broken by this index. */
int tkhip_map_store(tkhip_map* map, const float* pos, const float* power,
                    const uint8_t* theta, const uint8_t* phi, int64_t n);

/* How buildKdTree / balance (src/tracer/Photonmap.cpp:96-103, :132-179) lays
 * the photons out.  mode 1: median splits on the widest axis of the clipped
 * node box, level by level, one 64-bit radix sort per level (the tree balance
 * builds, with the axis choice it meant to make).  mode 0 (default): one sort
 * along a Hilbert curve; same leaf size, same implicit complete tree, twenty
 * times faster to build and as cheap to search.  A gather returns the same
 * photons either way.  Takes effect at the next build. */
int tkhip_map_set_tree(tkhip_map* map, int mode);

/* buildKdTree (src/tracer/Photonmap.cpp:96-103). */
int tkhip_map_build(tkhip_map* map);

/* Tree introspection for tests: number of leaves, photons per leaf (max), and
 * a copy of the photon permutation (leaf order -> original index), host out
 * int32[n]; perm may be null. */
int tkhip_map_tree_info(tkhip_map* map, int32_t* n_leaves, int32_t* leaf_capacity,
                        int32_t* depth, int32_t* perm);

/* radiance_estimate for nq queries (src/tracer/Photonmap.cpp:105-130 with
 * neighbor_search, src/tracer/neighbor.cpp:5-85).  Host arrays:
 *   in : p float[3nq], n float[3nq], wo float[3nq], R float[3nq] (diffuse
 *        reflectance of the surface hit), radius2 float[nq]
 *   out: flux float[3nq] (sum of power*|wi.n|*f, no division by pi r^2),
 *        found int32[nq], radius2_out float[nq] (= dist2[k] on return),
 *        idx int32[nq*k] or null: the found photon indices, ascending, rest -1.
 * mode 0: production path (grouped kernel + per-query kernel for overflow);
 * mode 1: per-query kernel only. */
int tkhip_map_gather(tkhip_map* map, int64_t nq, const float* p, const float* n, const float* wo,
                     const float* R, const float* radius2, float* flux, int32_t* found,
                     float* radius2_out, int32_t* idx, int mode);

/* tkhip_map_gather (production path, mode 0) for callers that cannot allocate
 * device memory: tight serving loops, harnesses that forbid hipMalloc, code
 * that must not disturb an allocator another library owns.  (Not for use while
 * a stream is being captured into a HIP graph: the call synchronises its
 * stream before it returns.)  No counterpart in the reference.  Every array is a
 * DEVICE pointer (same shapes as tkhip_map_gather; idx may be null); the query
 * order, the hand-over list between the two gather kernels and the sort's
 * temporary storage live in one static device array of
 * TKHIP_STATIC_SCRATCH_BYTES that the library holds on every GPU it runs on, so
 * the call allocates and frees nothing.  At most TKHIP_STATIC_MAX_QUERIES
 * queries per call (TKHIP_E_LIMIT beyond).  All calls on one GPU share that
 * array: calls that overlap on one GPU race and must be serialised by the
 * caller.  Work is enqueued on the map's stream; the call returns after it has
 * finished.  It takes no part in the timing totals of tkhip_sppm_gather_totals. */
#define TKHIP_STATIC_MAX_QUERIES 65536
#define TKHIP_STATIC_SCRATCH_BYTES (12 << 20)
int tkhip_map_gather_static(tkhip_map* map, int64_t nq, const float* p, const float* n, const float* wo,
                            const float* R, const float* radius2, float* flux, int32_t* found,
                            float* radius2_out, int32_t* idx);

/* Scene::intersect for n rays (src/core/Scene.cpp:22-25 -> src/BVH.cpp:245-288,
 * src/Triangle.cpp:15-52, src/Sphere.cpp:14-55).  Host arrays:
 *   in : org float[3n], dir float[3n], tmax float[n]
 *   out: prim int32[n] (-1 miss; triangles 0..n_tris-1, spheres n_tris+j),
 *        t float[n], p float[3n], nrm float[3n], perr float[3n] */
int tkhip_trace_rays(tkhip_ctx* ctx, int64_t n, const float* org, const float* dir,
                     const float* tmax, int32_t* prim, float* t, float* p, float* nrm,
                     float* perr);

/* The flux/radius/photon-count update rule of traceTile
 * (src/tracer/TkSppmTracer.cpp:327-340, :346-359) applied to n independent
 * (flux rgb, photons, radius2) records with gathered (Lindir rgb, found M,
 * returned radius2).  Host arrays, updated in place. */
int tkhip_pixel_update(tkhip_ctx* ctx, int64_t n, float* flux, float* photons, float* radius2,
                       const float* lindir, const int32_t* found, const float* radius2_ret);

/* RadicalInverse(baseIndex, a) for n (dim, index) pairs
 * (src/core/TkLowdiscrepancy.h:19-32).  Host arrays. */
int tkhip_radical_inverse(tkhip_ctx* ctx, int64_t n, const int32_t* dim, const uint64_t* index,
                          float* out);

/* StratifiedSampler::get1D / get2D (src/core/sampler.cpp:40-64, with Hash,
 * src/TkHash.h:111-119, and Math::PermutationElement, src/TkMath.h:206-234) for
 * n (pixel, sample index) pairs with xs*ys strata each: every pair calls
 * startPixelSample(pixel, index) and then makes the n_draws draws listed in
 * `draws` (1 = get1D, 2 = get2D).  jitter 0 is the reference's jitter = false
 * (stratum centres); otherwise the jitter comes from the counter-based generator
 * keyed by seed.  out holds sum(draws) floats per pair.  Host arrays. */
int tkhip_stratified_samples(tkhip_ctx* ctx, int64_t n, const int32_t* px, const int32_t* py,
                             const int32_t* sample_idx, const int32_t* xs, const int32_t* ys, int n_draws,
                             const uint8_t* draws, int jitter, uint64_t seed, float* out);

/* Intersection::spawnRay(w) and Intersection::spawnRayTo(target)
 * (src/Intersection.hpp:25-38) with offsetRayOrigin (src/Ray.hpp:43-57) for n
 * surface points (p, pError, n): org = origin of the ray along w, org_to /
 * dir_to = origin and direction of the ray towards target.  Host arrays, 3
 * floats per entry. */
int tkhip_spawn_rays(tkhip_ctx* ctx, int64_t n, const float* p, const float* perr, const float* nrm,
                     const float* w, const float* target, float* org, float* org_to, float* dir_to);

/* Scene::uniformSampleOneLight (src/core/Scene.cpp:43-61: AreaLight::sample_Li,
 * src/Light.cpp:24-39, the shadow ray of Intersection::spawnRayTo(other),
 * src/Intersection.hpp:40-48, Scene::intersectP, Material::f) at the first hit
 * of each of n rays.  u holds, per ray, the four numbers the call takes from its
 * sampler, in the order it takes them: get1D (which light), get2D (point on the
 * light), get1D (which triangle of the light's mesh).  prim is the hit as in
 * tkhip_trace_rays, ld the returned spectrum (black for a miss).  Host arrays:
 * org, dir float[3n], u float[4n] -> prim int32[n], ld float[3n]. */
int tkhip_sample_one_light(tkhip_ctx* ctx, int64_t n, const float* org, const float* dir, const float* u,
                           int32_t* prim, float* ld);

/* The two direction bytes a photon is stored with, for n world-space
 * directions wo (src/tracer/TkSppmTracer.cpp:135-140, :193-198).  Host arrays. */
int tkhip_encode_directions(tkhip_ctx* ctx, int64_t n, const float* wo, uint8_t* theta, uint8_t* phi);

/* Camera::generateRay for n raster positions (src/core/camera.cpp:51-71).
 * Host arrays: raster float[2n] -> org float[3n], dir float[3n]. */
int tkhip_camera_rays(tkhip_ctx* ctx, int64_t n, const float* raster, float* org, float* dir);

/* The photons of render map `which` (0 global, 1 caustic) after a photon pass,
 * in storage (path) order, to host arrays sized by the map capacity; any out
 * pointer may be null.  Returns the stored count in *n. */
int tkhip_sppm_read_photons(tkhip_ctx* ctx, int which, float* pos, float* power,
                            uint8_t* theta, uint8_t* phi, int64_t* n);

#ifdef __cplusplus
}
#endif
#endif /* TKHIP_H */
