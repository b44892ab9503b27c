// sppm_render.hip — one SPPM iteration on the device: photon passes, map
// builds, camera pass with gathers, resolve.
//
// Shape of the work: a pass has two stages.  The first traces the first
// segment of every path, one path per lane in plain full waves (these rays are
// alike and cheap); paths that go on are queued.  The second is one launch
// whose grid fills the GPU once: each lane walks a queued path to its end and
// takes the next queued path when its own ends.  Its rays are the resumable
// walks of tk_bvh_device.h: a wave does not wait for its slowest ray before it
// goes on, the lanes whose rays have finished shade and start their next rays
// while the slow rays continue:
//   photon pass : k_photon_first<KIND> -> k_photon_paths<KIND> -> scan -> k_photon_scatter
//                 (tracePhoton / traceCausticPhoton, src/tracer/TkSppmTracer.cpp:104-217)
//   camera pass : k_camera_first -> k_camera_paths -> gathers
//                 (traceTile, src/tracer/TkSppmTracer.cpp:290-374)
//   resolve     : k_resolve (src/tracer/TkSppmTracer.cpp:514-529)
//
// "Emit until the map holds N photons" (src/tracer/TkSppmTracer.cpp:411-421) is
// sequential in the CPU code.  Here light paths are traced in batches, each
// path keeps its photons in a private staging slot, and a prefix scan in path
// order cuts the batch where the map is full: the map holds the first N
// photons in path order and the shoot counter is the index of the path that
// filled it, which is what a single CPU thread produces.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "photon_map.h"
#include "tk_bvh_device.h"

// TKHIP_TRACE=1: the host-side steps of a pass on stderr (to see where a pass is waiting)
#define TK_TRACE(...) do { static const bool tr_ = getenv("TKHIP_TRACE") != nullptr; if (tr_) { fprintf(stderr, "tkhip trace: " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

namespace tkd {

// Counters of a pass.  Those that many waves hit with atomics at the same time
// (the work cursor of the path kernels, the two query appends, the two ends of
// the path queue) sit in 128-byte lines of their own: sharing one line cost
// 4 % of the iteration.  Everything before kCounterBatchInts is cleared for
// every batch of a pass; the 64-bit totals behind it run over the whole pass.
enum { C_QA = 0 /* work cursor of the path kernels */, C_QB = 32, C_QD = 64, C_NGQ = 96, C_NCQ = 128, C_TOTAL = 160, C_USED = 161,
       C_RAYS = 192 /* u64 */, C_NODES = 194 /* u64 */, C_PRIMS = 196 /* u64 */,   // the last two: counting instances only
       C_WALK_SECTIONS = 198 /* 4 x u64: node_turns, node_busy, leaf_turns, leaf_lanes of the counting instances */ };
constexpr int kCounterBatchInts = 192;
constexpr int kCounterBytes = 256 * 4;

// Everything one concurrently running pass needs: its stream, path state,
// queues, counters and scratch.  Lane 0 traces the global photons and builds
// the global map, lane 1 does the same for the caustic map, lane 2 traces the
// camera paths.  The three run side by side in tkhip_sppm_iteration: all of
// them are bound by the latency of single rays, not by throughput.
// The three streams of a render, kept for the life of the process and handed from one render to the
// next: a stream created after another one has been destroyed does not get a hardware queue of its
// own back, and the lanes of every render after the first in a process ran one behind the other
// (8.1 ms per iteration instead of 4.8).  Renders that exist at the same time get sets of their own.
struct StreamSet {
    int device;
    hipStream_t stream[3];
};
static std::mutex g_stream_mutex;
static std::vector<StreamSet> g_free_streams;

static int take_streams(int device, hipStream_t out[3]) {
    {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        for (size_t i = 0; i < g_free_streams.size(); ++i) {
            if (g_free_streams[i].device != device) continue;
            for (int l = 0; l < 3; ++l) out[l] = g_free_streams[i].stream[l];
            g_free_streams.erase(g_free_streams.begin() + (long)i);
            return TKHIP_OK;
        }
    }
    // One priority level per lane: streams of different priority never share a hardware queue,
    // so the three lanes cannot end up serialised behind one another.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int prio[3] = {prio_greatest, (prio_least + prio_greatest) / 2, prio_least};
    for (int l = 0; l < 3; ++l) {
        if (hipStreamCreateWithPriority(&out[l], hipStreamNonBlocking, prio[l]) != hipSuccess) {
            for (int k = 0; k < l; ++k) (void)hipStreamDestroy(out[k]);
            set_error("tkhip_sppm_begin: hipStreamCreate failed");
            return TKHIP_E_HIP;
        }
    }
    return TKHIP_OK;
}
static void give_streams(int device, hipStream_t s[3]) {
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    StreamSet set;
    set.device = device;
    for (int l = 0; l < 3; ++l) set.stream[l] = s[l];
    g_free_streams.push_back(set);
}

struct Lane {
    hipStream_t stream = nullptr;
    DevBuf counters;
    DevBuf q_o, q_d, q_w;            // path records between the two stages of a pass
    int32_t* h_counters = nullptr;   // pinned host copy of the counters
    int64_t queued_last = -1, batch_last = 0;   // paths the first stage of the last batch queued, and its size
    DevBuf stage_pos, stage_pow, rec_count, rec_offset, prim_temp;
    void release() {
        DevBuf* bufs[] = {&counters, &q_o, &q_d, &q_w, &stage_pos, &stage_pow, &rec_count, &rec_offset, &prim_temp};
        for (DevBuf* b : bufs) b->release();
        stream = nullptr;   // back to the pool, see destroy_sppm
        if (h_counters) (void)hipHostFree(h_counters);
        h_counters = nullptr;
    }
};

struct SppmState {
    TkSppmParam prm{};
    int num_recurse = 1, xs = 1, ys = 1, max_depth = 20;
    float rr = 0.8f;
    uint32_t seed = 0;
    int no_jitter = 0;               // tkhip_sppm_set_jitter
    bool counting = false;           // tkhip_sppm_set_counting: the passes run the counting instances of their kernels
    int rank = 0, world = 1;
    tkhip_map* map[2] = {nullptr, nullptr};
    int64_t quota[2] = {0, 0};
    DevBuf loc_pos[2], loc_pow[2], loc_dir[2];
    int64_t n_loc[2] = {0, 0}, last_shot[2] = {0, 0}, local_shot[2] = {0, 0}, shoot[2] = {0, 0};
    int64_t stored_c = 0;
    int64_t last_gather_photons[2] = {0, 0}, last_gather_staged[2] = {0, 0};   // of the last timed gather
    // the grouped kernels of the two maps side by side: from the first one's begin to the last one's end, summed over the steps
    double group_union_ms_total = 0.0;
    int64_t group_union_steps = 0;
    bool photon_begun = false;       // tkhip_sppm_photon_begin has enqueued the first batches
    int64_t begun_batch[2] = {0, 0};
    int camera_in_flight = -1;       // iteration whose camera paths have been launched and not yet waited for
    int64_t caustic_hoff = 0;        // global paths shot before the current iteration
    double yield[2] = {2.0, 0.05};
    // pixels
    DevBuf planes, rows;
    int64_t n_pixels = 0;
    int n_rows = 0;
    int64_t n_slots = 0;
    Lane lane[3];
    int photon_loads = 5;            // see photon_launch_batch
    int camera_share = 16;           // sixteenths of the wave slots the camera path kernel takes (steered with photon_loads)
    bool steer_settled = false;   // the two chains have ended within 5 % of each other once
    // world > 1: sixteenths of the GPU's wave slots the path kernel of each lane takes.  The three run side
    // by side from the start (whole grids of each would queue up behind one another) and one sixteenth
    // stays free for the small kernels between the passes; measured at rank 0 of 2: 4.55 -> 3.6 ms.
    int share[3] = {2, 3, 10};
    int path_waves[3] = {0, 0, 0};   // resident waves of the path kernel of each lane when it has the GPU to itself
    // Gather queries, global / caustic.  Two sets: a camera pass begun ahead (tkhip_sppm_camera_pass_ahead)
    // writes the one while the gathers of the iteration before still read the other; the second set is
    // allocated when it is first needed.
    DevBuf qbuf[2][2];
    int camera_set = 0;              // the set the camera pass launched last writes
    int gather_set = 0;              // the set of the camera pass waited for last: what the gathers read
    bool stale_radius[2] = {false, false};   // the set was written before the gathers of the iteration before had updated the radii
    int64_t n_queries[2] = {0, 0};   // global / caustic queries of the camera pass waited for last
    bool pt_mode = false;            // path tracer (tkhip_pt_*): the pixel planes are the film
    int pt_next_sample = 0;
    // hash-grid variant (tkhip_hsppm_*)
    bool hashgrid = false;
    int hg_spp = 1;
    DevBuf hg_prm, hg_bounds, hg_count, hg_start, hg_cursor, hg_items, hg_phi, hg_m;
    // photon exchange between ranks through one buffer per rank (tkhip_sppm_export_exchange / import_exchange)
    hipEvent_t ev_exchange = nullptr, ev_preview = nullptr;
    long long* h_exchange = nullptr;         // pinned: per map the photon count and the light paths shot, summed over the ranks
    bool exchange_pending = false;           // an import has installed maps whose sizes the host has only assumed so far
    bool export_pending = false;             // an export's copies of this rank's photon arrays may still be running
    bool preview_pending = false;            // a preview kernel may still be reading the pixel planes
    int64_t exchange_expected[2] = {0, 0};
    DevBuf preview;                  // RGBA8 frame buffer on the device (tkhip_sppm_set_preview)
    int preview_every = 0, preview_iter = -1;
};

// Paths that outlive the first stage of a pass wait here for the second one,
// three float4 per path.  Paths leaving a specular surface are appended from
// the front (counter C_QB), the others from the back (counter C_QD): the
// second stage takes the specular ones first.  They are the long and costly
// paths (chains of reflections inside glass), so they start early and do not
// share waves with the many paths that need one more ray.
struct PathQueue {
    float4* o;
    float4* d;
    float4* w;
    int cap;
};
TKD int queue_push(const PathQueue& q, int32_t* counters, bool want, bool specular) {   // wave-wide
    const int is = wave_append(&counters[C_QB], want && specular);
    const int id = wave_append(&counters[C_QD], want && !specular);
    if (!want) return -1;
    return specular ? is : q.cap - 1 - id;
}
TKD int queue_index(const PathQueue& q, int w, int n_front) { return w < n_front ? w : q.cap - 1 - (w - n_front); }

// ------------------------------------------------- hash grid of visible points
// SpatialHashGrids of the reference's second SPPM class (src/TkSPPM.cpp:19-112): the visible point of
// every pixel is listed in all the cells its search box overlaps, cells are hashed into as many
// buckets as the image has pixels.  Here the buckets are one array in bucket order (start / items).
struct HashGridParams {
    float bmin[3], bmax[3];
    int32_t res[3];
    int32_t valid;            // 0: no visible point this iteration
};
struct HashGridView {
    const HashGridParams* prm;
    const int32_t* start;     // buckets + 1
    const int32_t* items;     // visible-point numbers, bucket by bucket
    const float4* vp0;        // position, radius
    const float4* vp1;        // normal, dot(wo, n)
    const float4* vp3;        // R / pi of the surface
    float* phi;               // 3 planes over the visible points
    int32_t* m;               // photons that reached each visible point
    int64_t n_vis;
    uint32_t buckets;
};
TKD uint32_t hg_hash(int x, int y, int z, uint32_t buckets) {      // src/TkSPPM.cpp:109-112
    const int32_t h = (int32_t)(((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349663u) ^ ((uint32_t)z * 83492791u));
    return (uint32_t)((unsigned long long)(long long)h % (unsigned long long)buckets);   // s32 widened to size_t, as there
}
TKD bool hg_world_to_grid(const HashGridParams& g, F3 p, int pi[3]) {   // src/TkSPPM.cpp:88-99, Bounds3::Offset
    float o[3] = {p.x - g.bmin[0], p.y - g.bmin[1], p.z - g.bmin[2]};
    bool in = true;
    for (int i = 0; i < 3; ++i) {
        if (g.bmax[i] > g.bmin[i]) o[i] /= g.bmax[i] - g.bmin[i];
        pi[i] = (int)((float)g.res[i] * o[i]);
        in = in && pi[i] >= 0 && pi[i] < g.res[i];
        pi[i] = pi[i] < 0 ? 0 : (pi[i] > g.res[i] - 1 ? g.res[i] - 1 : pi[i]);
    }
    return in;
}
// A photon of power `w` arriving at x from direction wo: every visible point of x's cell within its
// radius counts it and takes f * w (src/TkSPPM.cpp:211-233).
TKD void hg_deposit(const HashGridView& g, F3 x, F3 wo, F3 w) {
    const HashGridParams P = *g.prm;
    if (!P.valid) return;
    int pi[3];
    if (!hg_world_to_grid(P, x, pi)) return;
    const uint32_t b = hg_hash(pi[0], pi[1], pi[2], g.buckets);
    const int e1 = g.start[b + 1];
    for (int e = g.start[b]; e < e1; ++e) {
        const int v = g.items[e];
        const float4 vp = g.vp0[v];
        const F3 d = sub3(mk3(vp.x, vp.y, vp.z), x);
        if (dot3(d, d) <= vp.w * vp.w) {
            atomicAdd(&g.m[v], 1);
            const float4 vn = g.vp1[v];
            if (vn.w * dot3(wo, mk3(vn.x, vn.y, vn.z)) > 0) {       // DiffuseMaterial::f
                const float4 r = g.vp3[v];
                atomicAdd(&g.phi[v], r.x * w.x);
                atomicAdd(&g.phi[g.n_vis + v], r.y * w.y);
                atomicAdd(&g.phi[2 * g.n_vis + v], r.z * w.z);
            }
        }
    }
}

// The same deposit by the whole wave: the lanes that have one take turns, and the 64 lanes walk the list of
// the photon's cell together (a list holds hundreds of visible points where the search radii are large; lane
// by lane the wave waited for its longest list: 14.5 ms per iteration of cornell_dragon at 1024^2).
struct HgPending {
    bool want;
    F3 x, wo, w;
};
TKD void hg_deposit_wave(const HashGridView& g, const HgPending& d) {
    const HashGridParams P = *g.prm;
    int e0 = 0, e1 = 0;
    if (d.want && P.valid) {
        int pi[3];
        if (hg_world_to_grid(P, d.x, pi)) {
            const uint32_t b = hg_hash(pi[0], pi[1], pi[2], g.buckets);
            e0 = g.start[b];
            e1 = g.start[b + 1];
        }
    }
    unsigned long long todo = __ballot(e1 > e0);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int s0 = __shfl(e0, src), s1 = __shfl(e1, src);
        const F3 x = mk3(__shfl(d.x.x, src), __shfl(d.x.y, src), __shfl(d.x.z, src));
        const F3 wo = mk3(__shfl(d.wo.x, src), __shfl(d.wo.y, src), __shfl(d.wo.z, src));
        const F3 w = mk3(__shfl(d.w.x, src), __shfl(d.w.y, src), __shfl(d.w.z, src));
        for (int e = s0 + lane; e < s1; e += 64) {
            const int v = g.items[e];
            const float4 vp = g.vp0[v];
            const F3 dv = sub3(mk3(vp.x, vp.y, vp.z), x);
            if (dot3(dv, dv) <= vp.w * vp.w) {
                atomicAdd(&g.m[v], 1);
                const float4 vn = g.vp1[v];
                if (vn.w * dot3(wo, mk3(vn.x, vn.y, vn.z)) > 0) {       // DiffuseMaterial::f
                    const float4 r = g.vp3[v];
                    atomicAdd(&g.phi[v], r.x * w.x);
                    atomicAdd(&g.phi[g.n_vis + v], r.y * w.y);
                    atomicAdd(&g.phi[2 * g.n_vis + v], r.z * w.z);
                }
            }
        }
    }
}

struct PhotonArgs {
    DScene sc;
    HashGridView hg;     // KIND 2 only
    PathQueue q;
    int32_t* rec_count;
    float4* stage_pos;   // xyz + direction bits
    float4* stage_pow;   // rgb
    int64_t m0;          // local index of the first path of the batch
    int64_t hoff;        // Halton offset (caustic paths: + paths shot for the global map)
    int world, rank;
    uint32_t seed;
    float rr;
    int slots;           // staging records per path
    int depth;
    int defer_below;     // k_photon_first<1>: a wave with at most this many paths left after the cull hands them on untraced
};

TKD uint64_t halton_index(const PhotonArgs& a, int slot) {
    return (uint64_t)(a.m0 + slot + a.hoff) * (uint64_t)a.world + (uint64_t)a.rank;
}

struct PhotonPath {
    F3 o, d, power;
    float ubx, uby;
    int flags;
};

// Light choice, AreaLight::sample_Le and the initial power of light path i of
// the batch (src/tracer/TkSppmTracer.cpp:106-122, src/Light.cpp:41-75,
// src/Triangle.cpp:190-197).  False when the path carries no power.
template <int KIND>
TKD bool photon_emit(const PhotonArgs& a, int i, PhotonPath& P) {
    const DScene& sc = a.sc;
    const uint64_t H = halton_index(a, i);
    const int nl = sc.n_lights;
    if (nl <= 0) return false;
    int li = (int)(radical_inverse(0, H) * nl);
    li = li > nl - 1 ? nl - 1 : li;
    const float pdf = 1.0f / nl;
    const float u0x = radical_inverse(1, H), u0y = radical_inverse(2, H);
    const float u1x = radical_inverse(3, H), u1y = radical_inverse(4, H);
    const DLight L = sc.lights[li];
    const float upick = photon_rng(a.seed, H, KIND, 0, 0);
    const LightEmission e = light_sample_le(sc, L, u0x, u0y, u1x, u1y, upick);
    if (e.pdf_pos == 0 || e.pdf_dir == 0) return false;
    P.power = divs3(scale3(e.Le, absdot3(e.d, e.n)), e.pdf_pos * e.pdf_dir * pdf);
    P.o = e.o;
    P.d = e.d;
    P.flags = 0;
    return true;
}
// The two numbers of the path's BSDF samples (the same two at every bounce, src/tracer/TkSppmTracer.cpp:149, :209);
// drawn once it is known that the path is traced at all.
TKD void photon_bsdf_numbers(const PhotonArgs& a, int i, PhotonPath& P) {
    const uint64_t H = halton_index(a, i);
    P.ubx = radical_inverse(5, H);
    P.uby = radical_inverse(6, H);
}

// One bounce of a light path: closest hit, photon store at diffuse surfaces,
// Russian roulette, BSDF sample (src/tracer/TkSppmTracer.cpp:124-156 for the
// global map, :179-216 for the caustic map).  Returns true while the path lives.
template <int KIND>
TKD bool photon_on_hit(const PhotonArgs& a, int slot, int bounce, PhotonPath& P, const Hit& h, bool* specular,
                       HgPending* deposit = nullptr) {
    const DScene& sc = a.sc;
    *specular = false;
    if (h.prim < 0) return false;
    SurfacePoint sp = surface_at(sc, h, P.o, P.d);
    F3 wo = neg3(P.d);
    const TkMaterial m = sc.mats[sp.mat];
    bool diffuse = m.type == TK_DIFFUSE;
    *specular = !diffuse;
    if (KIND == 2) {
        if (bounce > 0) {   // the caller makes the deposit with the whole wave (hg_deposit_wave) when it asks for it here
            if (deposit) { deposit->want = true; deposit->x = sp.p; deposit->wo = wo; deposit->w = P.power; }
            else hg_deposit(a.hg, sp.p, wo, P.power);
        }
    } else if (KIND == 0) {
        if (diffuse) {
            int c = P.flags;   // photons this path has staged so far
            int64_t at = (int64_t)slot * a.slots + c;
            a.stage_pos[at] = make_float4(sp.p.x, sp.p.y, sp.p.z, __uint_as_float(encode_direction(wo)));
            a.stage_pow[at] = make_float4(P.power.x, P.power.y, P.power.z, 0.f);
            P.flags = c + 1;
        }
    } else {
        if (!diffuse) P.flags |= 1;
        if (diffuse) {
            if (P.flags & 1) {
                a.stage_pos[slot] = make_float4(sp.p.x, sp.p.y, sp.p.z, __uint_as_float(encode_direction(wo)));
                a.stage_pow[slot] = make_float4(P.power.x, P.power.y, P.power.z, 0.f);
                a.rec_count[slot] = 1;
            }
            return false;
        }
    }
    uint64_t H = halton_index(a, slot);
    if (photon_rng(a.seed, H, KIND, bounce, 1) > a.rr) return false;
    F3 wi;
    float pdf;
    F3 f = mat_sample_f(m, true, wo, &wi, sp.n, P.ubx, P.uby, &pdf);
    if (pdf == 0 || isblack(f)) return false;
    P.power = divs3(scale3(mul3(P.power, f), absdot3(wi, sp.n)), pdf * a.rr);
    P.o = offset_ray_origin(sp.p, sp.perr, sp.n, wi);
    P.d = wi;
    return bounce + 1 < a.depth;
}
// Whether a ray can meet a specular object at all: slab test against the padded bounds of each (a ray
// parallel to a slab and outside of it yields an empty or NaN interval and counts as passing by only
// when it is empty beyond doubt).
TKD bool may_hit_specular(const DScene& sc, F3 o, F3 d) {
    if (sc.n_spec_boxes < 0) return true;
    const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    for (int b = 0; b < sc.n_spec_boxes; ++b) {
        const float* B = sc.spec_box[b];
        float t0 = 0.f, t1 = INFINITY;
        bool miss = false;
        {
            const float a0 = (B[0] - o.x) * ix, a1 = (B[3] - o.x) * ix;
            if (d.x != 0.f) { t0 = fmaxf(t0, fminf(a0, a1)); t1 = fminf(t1, fmaxf(a0, a1)); }
            else miss = miss || o.x < B[0] || o.x > B[3];
        }
        {
            const float a0 = (B[1] - o.y) * iy, a1 = (B[4] - o.y) * iy;
            if (d.y != 0.f) { t0 = fmaxf(t0, fminf(a0, a1)); t1 = fminf(t1, fmaxf(a0, a1)); }
            else miss = miss || o.y < B[1] || o.y > B[4];
        }
        {
            const float a0 = (B[2] - o.z) * iz, a1 = (B[5] - o.z) * iz;
            if (d.z != 0.f) { t0 = fmaxf(t0, fminf(a0, a1)); t1 = fminf(t1, fmaxf(a0, a1)); }
            else miss = miss || o.z < B[2] || o.z > B[5];
        }
        if (!miss && t0 <= t1 * 1.0001f) return true;
    }
    return false;
}
template <int KIND, bool COUNT, class STACK>
TKD bool photon_step(const PhotonArgs& a, int slot, int bounce, PhotonPath& P, bool* specular, STACK stk, WalkCount* wc) {
    const Hit h = trace_ray_t<COUNT>(a.sc, P.o, P.d, 1.7976931348623157e308, false, stk, wc);
    return photon_on_hit<KIND>(a, slot, bounce, P, h, specular);
}

constexpr int kPathBlock = 64;   // one wave per block: the kernels below use wave-wide votes as barriers
// The path kernels go on (walk_rays returns) once all but 1/8 of the rays a
// wave has in flight have finished: measured best among 1/2 .. 1/32.
#ifndef TK_WALK_REST_SHIFT
#define TK_WALK_REST_SHIFT 3            // (a build flag for sweeps)
#endif
constexpr int kWalkRestShift = TK_WALK_REST_SHIFT;

// Hands the idle lanes of a wave the next unstarted work items: returns this
// lane's item or -1.  *exhausted turns true (for the whole wave) once the
// cursor has passed n.  Every lane of the wave must call it.
TKD int wave_fetch_work(int32_t* cursor, bool idle, int64_t n, bool* exhausted) {
    const unsigned long long m = __ballot(idle);
    const int want = __popcll(m);
    int base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(cursor, want);
    base = __shfl(base, 0);
    const int mine = base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
    if ((int64_t)base + want >= n) *exhausted = true;
    return (idle && mine < n) ? mine : -1;
}

// A wave runs as long as its slowest lane, and the more paths a wave walks at
// once the longer that is.  When a queue holds fewer paths than the grid has
// lanes, only the first lanes_in_use(n) lanes of every wave take paths: the
// queue spreads over all the waves of the grid and is done sooner.
TKD int lanes_in_use(int n) {
    const int per_wave = (int)(((long long)n + gridDim.x - 1) / gridDim.x);
    return per_wave < 1 ? 1 : (per_wave > 64 ? 64 : per_wave);
}

TKD void count_wave_rays(int32_t* counters, int nrays) {
    float wr = (float)nrays;
    for (int o2 = 32; o2 > 0; o2 >>= 1) wr += __shfl_xor(wr, o2);
    if ((threadIdx.x & 63) == 0 && wr > 0)
        atomicAdd(reinterpret_cast<unsigned long long*>(&counters[C_RAYS]), (unsigned long long)wr);
}
// Counting instances of the kernels (tkhip_sppm_set_counting): nodes fetched and primitives tested by the wave.
TKD void count_wave_walk(int32_t* counters, const WalkCount& wc) {
    int n = wc.nodes, p = wc.prims;
    for (int o2 = 32; o2 > 0; o2 >>= 1) { n += __shfl_xor(n, o2); p += __shfl_xor(p, o2); }
    int sec[4] = {wc.node_turns, wc.node_busy, wc.leaf_turns, wc.leaf_lanes};
    for (int k = 0; k < 4; ++k)
        for (int o2 = 32; o2 > 0; o2 >>= 1) sec[k] += __shfl_xor(sec[k], o2);
    if ((threadIdx.x & 63) == 0) {
        if (n > 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[C_NODES]), (unsigned long long)n);
        if (p > 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[C_PRIMS]), (unsigned long long)p);
        for (int k = 0; k < 4; ++k)
            if (sec[k] > 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[C_WALK_SECTIONS + 2 * k]), (unsigned long long)sec[k]);
    }
}

// First stage of the photon pass of a batch of B light paths: emission and the
// first segment of every path, one path per lane.  These rays all leave the
// lights, cost about the same and most caustic paths end here (at a diffuse
// surface), so plain full waves are the efficient shape.  Paths that go on are
// queued for the second stage.
constexpr int kUntraced = 0x80;    // flag of a queued light path whose first ray has not been traced yet
template <int KIND, bool COUNT = false>
__global__ void __launch_bounds__(kTraceBlock) k_photon_first(PhotonArgs a, int64_t B, int32_t* counters) {
    __shared__ int s_stack[kStackDepth * kTraceBlock];
    LdsStack stk{&s_stack[threadIdx.x]};
    WalkCount wc{};
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool alive = false, specular = false;
    int traced = 0;
    PhotonPath P{};
    if (i < B) {
        if (KIND == 1) a.rec_count[i] = 0;
        alive = photon_emit<KIND>(a, (int)i, P);
        // a path for the caustic map whose first ray passes by every specular object ends at a diffuse
        // surface or nowhere and stores nothing: it is counted as shot and not traced
        if (KIND == 1 && alive && !may_hit_specular(a.sc, P.o, P.d)) alive = false;
        if (alive) photon_bsdf_numbers(a, (int)i, P);
    }
    // Where that leaves a wave with a few paths only (a small specular object in a large scene: 3 % of the
    // paths of the 10 M-triangle field), tracing them here would run the walk for a handful of lanes: they
    // go to the second stage untraced instead (kUntraced in the flags), where the waves fill up.
    const bool defer = KIND == 1 && __popcll(__ballot(alive)) <= a.defer_below;
    if (i < B) {
        if (alive && !defer) {
            traced = 1;
            alive = photon_step<KIND, COUNT>(a, (int)i, 0, P, &specular, stk, &wc);
        }
        if (alive && defer) P.flags |= kUntraced;
        if (KIND == 0 && !alive) a.rec_count[i] = P.flags;
    }
    const int at = queue_push(a.q, counters, alive, specular);
    if (alive) {
        a.q.o[at] = make_float4(P.o.x, P.o.y, P.o.z, P.ubx);
        a.q.d[at] = make_float4(P.d.x, P.d.y, P.d.z, P.uby);
        a.q.w[at] = make_float4(P.power.x, P.power.y, P.power.z, __int_as_float((int)i | (P.flags << 24)));
    }
    count_wave_rays(counters, traced);
    if (COUNT) count_wave_walk(counters, wc);
}

// Second stage: the queued paths from their second segment on.  The grid is
// sized to fill the GPU once; every lane walks one path at a time, and a lane
// whose path has ended takes the next queued path, so that waves stay full
// until the queue runs out.  A turn of the loop advances the rays in flight
// (walk_rays) until all but an eighth of them have finished; the lanes whose
// ray is over take the bounce and start their next ray, the others keep
// walking theirs in the next turn.
template <int KIND, bool COUNT = false>
__global__ void __launch_bounds__(kPathBlock, 4) k_photon_paths(PhotonArgs a, int32_t* counters) {
    __shared__ int s_stack[kStackDepth * kPathBlock];
    LdsStackT<kPathBlock> stk{&s_stack[threadIdx.x]};
    WalkCount wc{};
    const int n_front = counters[C_QB], n = n_front + counters[C_QD];
    // A queue shorter than the grid is spread over all waves (lanes_in_use); the paths at its front (from
    // specular surfaces: the long ones) are then dealt thinly too, a few to every wave with the wave's
    // first fetch, when that leaves a wave few enough of them for the walk with eight lanes to a ray.
    bool first_fetch = n_front > 0 && a.sc.wide && lanes_in_use(n) < 64 && lanes_in_use(n_front) <= a.sc.wide_rays;
    bool takes = (int)(threadIdx.x & 63) < (first_fetch ? lanes_in_use(n_front) : lanes_in_use(n));
    bool alive = false, exhausted = false, in_flight = false;
    int slot = 0, bounce = 0, traced = 0;
    PhotonPath P{};
    RayWalk w;
    w.cur = kWalkOver;
    while (true) {
        if (!exhausted && __any(!alive && takes)) {
            const int got = wave_fetch_work(&counters[C_QA], !alive && takes, n, &exhausted);
            if (first_fetch) { first_fetch = false; takes = (int)(threadIdx.x & 63) < lanes_in_use(n); }
            if (got >= 0) {
                const int at = queue_index(a.q, got, n_front);
                const float4 ro = a.q.o[at], rd = a.q.d[at], rw = a.q.w[at];
                P.o = mk3(ro.x, ro.y, ro.z); P.ubx = ro.w;
                P.d = mk3(rd.x, rd.y, rd.z); P.uby = rd.w;
                P.power = mk3(rw.x, rw.y, rw.z);
                const int packed = __float_as_int(rw.w);
                slot = packed & 0xffffff;
                P.flags = (packed >> 24) & 0x7f;
                bounce = ((packed >> 24) & kUntraced) ? 0 : 1;   // untraced: the first stage left the path's first ray to this one
                alive = true;
                in_flight = false;
            }
        }
        if (!__any(alive)) {
            if (exhausted) break;
            continue;
        }
        const int n_alive = __popcll(__ballot(alive));
        const bool last = exhausted && a.sc.wide && n_alive <= a.sc.wide_rays;
        if (last && !__any(in_flight)) break;   // the last paths: below
        if (alive && !in_flight && !last) {
            walk_begin(w, a.sc, P.d, 1.7976931348623157e308);
            in_flight = true;
            ++traced;
        }
        walk_rays_t<COUNT>(a.sc, w, P.o, P.d, false, stk, last ? 0 : n_alive >> kWalkRestShift, &wc);   // last: the rays under way are finished, none begun
        HgPending dep{};
        if (alive && in_flight && w.cur == kWalkOver) {   // this lane's ray has finished: the bounce, then its next ray
            bool specular;
            alive = photon_on_hit<KIND>(a, slot, bounce, P, w.h, &specular, KIND == 2 ? &dep : nullptr);
            in_flight = false;
            ++bounce;
            if (KIND == 0 && !alive) a.rec_count[slot] = P.flags;
        }
        if (KIND == 2) hg_deposit_wave(a.hg, dep);
    }
    // the few paths left once the queue is empty: eight lanes to a ray in the wide tree (see k_camera_paths)
    while (__any(alive)) {
        const Hit h = walk_wide_t<COUNT>(a.sc, alive, P.o, P.d, 1.7976931348623157e308, false, stk, &wc);
        HgPending dep{};
        if (alive) {
            ++traced;
            bool specular;
            alive = photon_on_hit<KIND>(a, slot, bounce, P, h, &specular, KIND == 2 ? &dep : nullptr);
            ++bounce;
            if (KIND == 0 && !alive) a.rec_count[slot] = P.flags;
        }
        if (KIND == 2) hg_deposit_wave(a.hg, dep);
    }
    count_wave_rays(counters, traced);
    if (COUNT) count_wave_walk(counters, wc);
}

// Appends the staged photons of the batch in path order behind the `base`
// photons already stored, up to `quota`; reports how many photons the batch
// produced and which path filled the map.
__global__ void k_photon_scatter(const int32_t* rec_count, const int32_t* rec_offset, const float4* stage_pos,
                                 const float4* stage_pow, int slots, int64_t B, int64_t base, int64_t quota,
                                 float* out_pos, float* out_pow, uint16_t* out_dir, int32_t* counters) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= B) return;
    int c = rec_count[i];
    int64_t off = rec_offset[i];
    int64_t need = quota - base;
    for (int r = 0; r < c; ++r) {
        int64_t g = off + r;
        if (g >= need) break;
        float4 p = stage_pos[i * slots + r], w = stage_pow[i * slots + r];
        int64_t at = base + g;
        out_pos[3 * at] = p.x; out_pos[3 * at + 1] = p.y; out_pos[3 * at + 2] = p.z;
        out_pow[3 * at] = w.x; out_pow[3 * at + 1] = w.y; out_pow[3 * at + 2] = w.z;
        out_dir[at] = (uint16_t)__float_as_uint(p.w);
    }
    if (off < need && off + c >= need) counters[C_USED] = (int32_t)(i + 1);
    if (i == B - 1) counters[C_TOTAL] = (int32_t)(off + c);
}

// ------------------------------------------------------------ camera pass
struct CameraArgs {
    DScene sc;
    DCamera cam;
    const int32_t* rows;
    float* planes;
    int64_t n_pixels;
    int iter, spp, xs, ys, depth;
    uint32_t seed;
    int no_jitter;
    int caustic_stored;
    int visible_only;    // hash-grid variant: the path ends at its first diffuse hit, which becomes the pixel's visible point
    float ld_scale;      // 1 / spp there (src/TkSPPM.cpp:160-161), 1 otherwise
    int pt_mode;         // PathTracer::Li (src/tracer/TkPathTracer.cpp:149-204): no gathers, Russian roulette, film sums
    float rr;
    float4* gq;          // global queries: 4 arrays of capacity n_slots, back to back
    float4* cq;          // caustic queries
    int64_t qcap;
    PathQueue q;
};

TKD Sampler make_sampler(const CameraArgs& a, int x, int y, int dim) {
    Sampler s;
    s.px = x; s.py = y; s.sample_idx = a.iter; s.spp = a.spp; s.xs = a.xs; s.ys = a.ys; s.dim = dim;
    s.rng_seed = a.seed; s.rng_pixel = (uint32_t)(x + y * a.cam.w); s.rng_iter = (uint32_t)a.iter;
    s.no_jitter = a.no_jitter != 0;
    return s;
}

// State of one camera path between two rays.  A lane traces one ray per turn
// of the kernel's loop: the path's next segment, or the shadow ray of the
// light sample taken at the last diffuse hit.  While the shadow ray is out,
// everything the hit already decided waits in the `next_*` / `ld_*` fields.
struct CameraPath {
    F3 o, d;       // the ray to trace now
    F3 coef;       // throughput of the path up to the ray's origin
    int flags;     // bit0: only specular surfaces so far; bits 1..: sampler dimension
    bool shadow;   // the ray is a shadow ray
    F3 ld_seen;    // emitted radiance seen at the hit the shadow ray left from
    F3 ld_light;   // light sample's contribution, added if the shadow ray arrives
    F3 next_o, next_d, next_coef;
    int next_flags;
    bool next_alive;
};

// The camera sample of a pixel (src/tracer/TkSppmTracer.cpp:296-304).
TKD void camera_start(const CameraArgs& a, int x, int y, CameraPath& P) {
    Sampler sm = make_sampler(a, x, y, 0);
    float cx, cy, lx, ly;
    sampler_get2d(sm, &cx, &cy);
    cx = cx + (float)x; cy = cy + (float)y;
    sampler_get2d(sm, &lx, &ly);   // lens sample: drawn and unused with a pinhole camera
    camera_ray(a.cam, cx, cy, &P.o, &P.d);
    P.coef = mk3(1.f, 1.f, 1.f);
    P.flags = 1 | (sm.dim << 1);
    P.shadow = false;
    if (a.pt_mode) {
        // FilmTile::addSample with the box filter of radius 0.5 (src/core/TkFilm.cpp:14-42, src/main.cpp:41):
        // the pixels [p0, p1) the sample lands in are its own and, when the sample sits exactly on
        // a pixel edge, a neighbour.  Bits 27..30 of the flags name the neighbours that also take it.
        const float dx = cx - 0.5f, dy = cy - 0.5f;
        const int p0x = (int)ceilf(dx - 0.5f), p1x = (int)floorf(dx + 0.5f + 1);
        const int p0y = (int)ceilf(dy - 0.5f), p1y = (int)floorf(dy + 0.5f + 1);
        if (p0x < x && x - 1 >= 0) P.flags |= 1 << 27;
        if (p1x > x + 1 && x + 1 < a.cam.w) P.flags |= 1 << 28;
        if (p0y < y && y - 1 >= 0) P.flags |= 1 << 29;
        if (p1y > y + 1 && y + 1 < a.cam.h) P.flags |= 1 << 30;
    }
}
constexpr int kPtFootprintMask = 0xf << 27;
// Calls fn(pixel) for every pixel a path-tracer sample lands in.
template <class F>
TKD void pt_footprint(const CameraArgs& a, int pixel, int flags, F fn) {
    const int W = a.cam.w;
    const int xlo = (flags >> 27) & 1 ? -1 : 0, xhi = (flags >> 28) & 1 ? 1 : 0;
    const int ylo = (flags >> 29) & 1 ? -1 : 0, yhi = (flags >> 30) & 1 ? 1 : 0;
    for (int dy = ylo; dy <= yhi; ++dy)
        for (int dx = xlo; dx <= xhi; ++dx) fn(pixel + dx + dy * W);
}

TKD void emit_query(float4* qbase, int64_t qcap, int dst, const SurfacePoint& sp, F3 coef, const TkMaterial& m,
                    float radius2, int pixel) {
    qbase[dst] = make_float4(sp.p.x, sp.p.y, sp.p.z, radius2);
    qbase[qcap + dst] = make_float4(sp.n.x, sp.n.y, sp.n.z, dot3(sp.wo, sp.n));
    qbase[2 * qcap + dst] = make_float4(coef.x, coef.y, coef.z, __int_as_float(pixel));
    qbase[3 * qcap + dst] = make_float4(m.R[0] * kInvPi, m.R[1] * kInvPi, m.R[2] * kInvPi, 0.f);
}

TKD void add_direct(const CameraArgs& a, int pixel, F3 Ld, int flags = 0) {
    if (a.pt_mode) {   // Pixel::contribSum (filter weight 1); a neighbour's sample may land here too: atomics
        if (isblack(Ld)) return;
        pt_footprint(a, pixel, flags, [&](int px) {
            atomicAdd(&a.planes[px], Ld.x);
            atomicAdd(&a.planes[a.n_pixels + px], Ld.y);
            atomicAdd(&a.planes[2 * a.n_pixels + px], Ld.z);
        });
        return;
    }
    if (!isblack(Ld)) {
        Ld = scale3(Ld, a.ld_scale);   // a power of two: exact
        a.planes[pixel] += Ld.x;
        a.planes[a.n_pixels + pixel] += Ld.y;
        a.planes[2 * a.n_pixels + pixel] += Ld.z;
    }
}

// What a hit of a path segment leads to (loop body of traceTile,
// src/tracer/TkSppmTracer.cpp:306-371, with Scene::uniformSampleOneLight,
// src/core/Scene.cpp:43-61, inlined).  Gather queries are reported through
// want_cq / want_gq together with the surface they belong to.  Returns true
// while the lane has another ray to trace (P.o, P.d, P.shadow).
struct CameraStepOut {
    bool want_cq, want_gq;
    SurfacePoint sp;
    F3 coef;
    TkMaterial m;
};
TKD bool camera_on_hit(const CameraArgs& a, int x, int y, int pixel, int bounce, const Hit& h, CameraPath& P,
                       CameraStepOut& out) {
    const DScene& sc = a.sc;
    bool has_glossy = P.flags & 1;
    const bool pt = a.pt_mode != 0;
    Sampler sm = make_sampler(a, x, y, (P.flags & ~kPtFootprintMask) >> 1);
    const SurfacePoint sp = surface_at(sc, h, P.o, P.d);
    const TkMaterial m = sc.mats[sp.mat];
    const bool diffuse = m.type == TK_DIFFUSE;
    const F3 wo = neg3(P.d);
    const F3 coef = P.coef;
    F3 Ld = mk3(0, 0, 0);
    // `Ld += coef * Le` and `Ld += coef * uniformSampleOneLight(...)` run at every such hit, with a black
    // second factor where nothing is emitted or nothing arrives: that adds nothing unless the throughput
    // has gone non-finite (a reflection at exactly grazing incidence: f = R / 0), in which case the pixel's
    // direct light turns NaN there as well.  The products with black below keep that.
    const F3 black = mk3(0.f, 0.f, 0.f);
    if (has_glossy) Ld = add3(Ld, mul3(coef, sp.light >= 0 ? light_L(sc.lights[sp.light], sp.n, wo) : black));
    bool want_shadow = false;
    F3 sh_o = mk3(0, 0, 0), sh_d = mk3(0, 0, 0), ld_light = mk3(0, 0, 0);
    if ((pt || has_glossy || !diffuse) && sc.n_lights > 0) {
        // uniformSampleOneLight: the sampler dimensions advance for every material
        const float uc = sampler_get1d(sm);
        float ulx, uly;
        sampler_get2d(sm, &ulx, &uly);
        const float u0 = sampler_get1d(sm);
        const LightSample ls = sample_one_light(sc, sp, m, uc, ulx, uly, u0);
        if (ls.want_shadow) {
            sh_o = ls.o; sh_d = ls.d;
            ld_light = mul3(coef, ls.ld);
            want_shadow = true;
        } else {
            Ld = add3(Ld, mul3(coef, black));
        }
    }
    // what follows the light sample does not depend on the shadow ray's answer
    bool go_on = true;
    if (pt) {   // Russian roulette before the BSDF sample (src/tracer/TkPathTracer.cpp:165-166)
        if (rng_u01(a.seed, (uint32_t)pixel, (uint32_t)a.iter, 0x10000u + (uint32_t)bounce) > a.rr) go_on = false;
    } else if (diffuse) {
        out.sp = sp; out.coef = coef; out.m = m;
        if (a.visible_only) { out.want_cq = true; go_on = false; }   // src/TkSPPM.cpp:162-167
        else {
            if (has_glossy && a.caustic_stored > 0) out.want_cq = true;
            if (!has_glossy) { out.want_gq = true; go_on = false; }
        }
    }
    F3 n_o = P.o, n_d = P.d, n_coef = coef;
    int n_flags = P.flags;
    if (go_on) {
        has_glossy = pt ? !diffuse : (has_glossy && !diffuse);   // pt: `test`, set by the surface just left (:172)
        float ux, uy;
        sampler_get2d(sm, &ux, &uy);
        F3 wi;
        float pdf;
        F3 f = mat_sample_f(m, false, wo, &wi, sp.n, ux, uy, &pdf);
        if (isblack(f) || pdf == 0 || !(bounce + 1 < a.depth)) go_on = false;
        else {
            if (pt) n_coef = mul3(coef, divs3(scale3(f, absdot3(wi, sp.n)), pdf * a.rr));   // beta *= f * |cos| / (pdf * rr)
            else n_coef = divs3(scale3(mul3(coef, f), absdot3(wi, sp.n)), pdf);
            n_o = offset_ray_origin(sp.p, sp.perr, sp.n, wi);
            n_d = wi;
            n_flags = (has_glossy ? 1 : 0) | (sm.dim << 1) | (P.flags & kPtFootprintMask);
        }
    }
    if (want_shadow) {
        P.ld_seen = Ld; P.ld_light = ld_light;
        P.next_o = n_o; P.next_d = n_d; P.next_coef = n_coef; P.next_flags = n_flags; P.next_alive = go_on;
        P.o = sh_o; P.d = sh_d; P.shadow = true;
        return true;
    }
    add_direct(a, pixel, Ld, P.flags);
    P.o = n_o; P.d = n_d; P.coef = n_coef; P.flags = n_flags;
    return go_on;
}
TKD void camera_emit_queries(const CameraArgs& a, int32_t* counters, const CameraStepOut& o, int pixel) {
    int dc = wave_append(&counters[C_NCQ], o.want_cq);
    if (o.want_cq) emit_query(a.cq, a.qcap, dc, o.sp, o.coef, o.m, a.planes[11 * a.n_pixels + pixel], pixel);
    int dg = wave_append(&counters[C_NGQ], o.want_gq);
    if (o.want_gq) emit_query(a.gq, a.qcap, dg, o.sp, o.coef, o.m, a.planes[12 * a.n_pixels + pixel], pixel);
}
// One turn of a camera path: traces the lane's ray (a path segment or the
// pending shadow ray) and takes what follows from it.  Returns true while the
// lane has another ray to trace.
TKD bool camera_after_ray(const CameraArgs& a, int x, int y, int pixel, int& bounce, CameraPath& P, CameraStepOut& so,
                          const Hit& h) {
    if (P.shadow) {
        F3 Ld = P.ld_seen;
        Ld = add3(Ld, h.prim < 0 ? P.ld_light : mul3(P.ld_light, mk3(0.f, 0.f, 0.f)));   // blocked: coef * black
        add_direct(a, pixel, Ld, P.flags);
        P.o = P.next_o; P.d = P.next_d; P.coef = P.next_coef; P.flags = P.next_flags;
        P.shadow = false;
        ++bounce;
        return P.next_alive;
    }
    if (h.prim < 0) return false;
    const bool go_on = camera_on_hit(a, x, y, pixel, bounce, h, P, so);
    if (!P.shadow) ++bounce;
    return go_on;
}
template <bool COUNT, class STACK>
TKD bool camera_turn(const CameraArgs& a, int x, int y, int pixel, int& bounce, CameraPath& P, CameraStepOut& so,
                     STACK stk, WalkCount* wc) {
    const Hit h = trace_ray_t<COUNT>(a.sc, P.o, P.d, P.shadow ? (double)kOneMinusEps : 1.7976931348623157e308, P.shadow, stk, wc);
    return camera_after_ray(a, x, y, pixel, bounce, P, so, h);
}

// First stage of the camera pass: the camera ray of every pixel of this rank,
// what its hit leads to, and the shadow ray of the light sample taken there,
// one pixel per lane.  Camera rays are coherent and cheap and so are the shadow
// rays that follow them; paths that go on are queued for the second stage.
template <bool COUNT = false>
__global__ void __launch_bounds__(kTraceBlock) k_camera_first(CameraArgs a, int64_t n_slots, int32_t* counters) {
    __shared__ int s_stack[kStackDepth * kTraceBlock];
    LdsStack stk{&s_stack[threadIdx.x]};
    WalkCount wc{};
    const int64_t slot = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool alive = slot < n_slots;
    int pixel = 0, x = 0, y = 0, bounce = 0, nrays = 0;
    CameraPath P{};
    CameraStepOut so{};
    so.want_cq = so.want_gq = false;
    if (alive) {
        const int W = a.cam.w;
        y = a.rows[slot / W]; x = (int)(slot % W);
        pixel = x + y * W;
        camera_start(a, x, y, P);
        if (a.pt_mode)   // Pixel::filterWeightSum
            pt_footprint(a, pixel, P.flags, [&](int px) { atomicAdd(&a.planes[9 * a.n_pixels + px], 1.0f); });
        nrays = 1;
        alive = camera_turn<COUNT>(a, x, y, pixel, bounce, P, so, stk, &wc);
    }
    camera_emit_queries(a, counters, so, pixel);   // wave-wide appends: every lane takes part
    if (alive && P.shadow) {
        ++nrays;
        CameraStepOut none{};
        alive = camera_turn<COUNT>(a, x, y, pixel, bounce, P, none, stk, &wc);
    }
    const int at = queue_push(a.q, counters, alive, (P.flags & 1) != 0);
    if (alive) {
        a.q.o[at] = make_float4(P.o.x, P.o.y, P.o.z, __int_as_float(pixel));
        a.q.d[at] = make_float4(P.d.x, P.d.y, P.d.z, __int_as_float(P.flags));
        a.q.w[at] = make_float4(P.coef.x, P.coef.y, P.coef.z, __int_as_float(bounce));
    }
    count_wave_rays(counters, nrays);
    if (COUNT) count_wave_walk(counters, wc);
}

// Second stage: the queued paths from their second segment on, organised like
// k_photon_paths; path segments and shadow rays are in flight side by side.
// spread: see lanes_in_use; otherwise the queue fills whole waves.
template <bool COUNT = false>
__global__ void __launch_bounds__(kPathBlock, 4) k_camera_paths(CameraArgs a, int spread, int32_t* counters) {
    __shared__ int s_stack[kStackDepth * kPathBlock];
    LdsStackT<kPathBlock> stk{&s_stack[threadIdx.x]};
    WalkCount wc{};
    const int n_front = counters[C_QB], n = n_front + counters[C_QD];
    if (!spread && (int64_t)blockIdx.x * 64 >= n) return;   // the waves before this one take the whole queue
    // spread: the paths at the front of the queue (from specular surfaces: the long ones) are dealt thinly
    // too, a few to every wave with the wave's first fetch, instead of filling the first waves with them
    // (when that leaves a wave few enough of them to end up walking them eight lanes to a ray soon: up to
    // twice the limit of that walk; rank 0 of 4: 3.05 -> 2.98 ms per iteration, of 8: 2.65 -> 2.39 ms)
    bool first_fetch = spread && n_front > 0 && a.sc.wide && lanes_in_use(n_front) <= 2 * a.sc.wide_rays;
    bool takes = !spread || (int)(threadIdx.x & 63) < (first_fetch ? lanes_in_use(n_front) : lanes_in_use(n));
    bool alive = false, exhausted = false, in_flight = false;
    int pixel = 0, x = 0, y = 0, bounce = 0, nrays = 0;
    CameraPath P{};
    RayWalk w;
    w.cur = kWalkOver;
    const int W = a.cam.w;
#ifdef TK_CP_STAMPS
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
    int st_main_turns = 0, st_wide_turns = 0, st_first_alive = -1, st_at_exhaust = -1;
#endif
    while (true) {
#ifdef TK_CP_STAMPS
        ++st_main_turns;
        if (exhausted && st_at_exhaust < 0) st_at_exhaust = __popcll(__ballot(alive));
#endif
        if (!exhausted && __any(!alive && takes)) {
            const int got = wave_fetch_work(&counters[C_QA], !alive && takes, n, &exhausted);
            if (first_fetch) { first_fetch = false; takes = (int)(threadIdx.x & 63) < lanes_in_use(n); }
            if (got >= 0) {
                const int at = queue_index(a.q, got, n_front);
                const float4 ro = a.q.o[at], rd = a.q.d[at], rw = a.q.w[at];
                P.o = mk3(ro.x, ro.y, ro.z); P.d = mk3(rd.x, rd.y, rd.z); P.coef = mk3(rw.x, rw.y, rw.z);
                pixel = __float_as_int(ro.w);
                P.flags = __float_as_int(rd.w);
                bounce = __float_as_int(rw.w);
                P.shadow = false;
                y = pixel / W; x = pixel - y * W;
                alive = true;
                in_flight = false;
            }
        }
        if (!__any(alive)) {
            if (exhausted) break;
            continue;
        }
        const int n_alive = __popcll(__ballot(alive));
        const bool last = exhausted && a.sc.wide && n_alive <= a.sc.wide_rays;
        if (last && !__any(in_flight)) break;   // the last paths: below
        if (alive && !in_flight && !last) {
            walk_begin(w, a.sc, P.d, P.shadow ? (double)kOneMinusEps : 1.7976931348623157e308);
            in_flight = true;
            ++nrays;
        }
        walk_rays_t<COUNT>(a.sc, w, P.o, P.d, P.shadow, stk, last ? 0 : n_alive >> kWalkRestShift, &wc);   // last: the rays under way are finished, none begun
        CameraStepOut so{};
        so.want_cq = so.want_gq = false;
        if (alive && in_flight && w.cur == kWalkOver) {   // this lane's ray has finished: what follows from it, then its next ray
            alive = camera_after_ray(a, x, y, pixel, bounce, P, so, w.h);
            in_flight = false;
        }
        camera_emit_queries(a, counters, so, pixel);   // wave-wide appends: every lane takes part
    }
    // The few paths a wave is left with once the queue is empty (none with a ray under way): eight lanes to
    // a ray in the wide tree.  A loop of its own, so that the loop above keeps its registers.
#ifdef TK_CP_STAMPS
    const unsigned long long st_mid = __builtin_amdgcn_s_memtime();
    st_first_alive = __popcll(__ballot(alive));
#endif
    while (__any(alive)) {
#ifdef TK_CP_STAMPS
        ++st_wide_turns;
#endif
        const Hit h = walk_wide_t<COUNT>(a.sc, alive, P.o, P.d, P.shadow ? (double)kOneMinusEps : 1.7976931348623157e308, P.shadow, stk, &wc);
        CameraStepOut so{};
        so.want_cq = so.want_gq = false;
        if (alive) {
            ++nrays;
            alive = camera_after_ray(a, x, y, pixel, bounce, P, so, h);
        }
        camera_emit_queries(a, counters, so, pixel);
    }
#ifdef TK_CP_STAMPS
    if ((threadIdx.x & 63) == 0) {   // profiling build: the slowest wave's times (100 MHz ticks) and turns
        const unsigned long long st_end = __builtin_amdgcn_s_memtime();
        unsigned long long* u = reinterpret_cast<unsigned long long*>(&counters[224]);
        atomicMax(&u[0], st_mid - st_begin); atomicMax(&u[1], st_end - st_mid);
        atomicMax(&u[2], (unsigned long long)st_main_turns); atomicMax(&u[3], (unsigned long long)st_wide_turns);
        atomicMax(&u[4], (unsigned long long)st_first_alive); atomicMax(&u[5], (unsigned long long)(st_at_exhaust + 1));
        atomicAdd(&u[6], st_end - st_begin); atomicAdd(&u[7], 1ull);
    }
#endif
    count_wave_rays(counters, nrays);
    if (COUNT) count_wave_walk(counters, wc);
}

// The search radius of every query of one map, read again from its pixel (plane = the map's radius plane).
__global__ void k_query_radius(float4* q0, const float4* q2, int64_t nq, const float* radius2) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    q0[i].w = radius2[__float_as_int(q2[i].w)];
}

// Final colour + gamma + RGB8, rows bottom-up (src/tracer/TkSppmTracer.cpp:514-529)
__global__ void k_resolve(const float* planes, int64_t np, int W, int H, int num_recurse, float shoot_g,
                          float shoot_c, int caustic_stored, uint8_t* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= np) return;
    int x = (int)(i % W), y = (int)(i / W);
    float cr2 = planes[11 * np + i], gr2 = planes[12 * np + i];
    for (int ch = 0; ch < 3; ++ch) {
        float v = planes[ch * np + i] / (float)num_recurse;
        if (caustic_stored > 0) v = v + planes[(3 + ch) * np + i] / (kPi * cr2 * shoot_c);
        else v = v + 0.f;
        v = v + planes[(6 + ch) * np + i] / (kPi * gr2 * shoot_g);
        float cl = sse_max(sse_min(v, 1.0f), 0.0f);
        out[((int64_t)(H - 1 - y) * W + x) * 3 + ch] = (uint8_t)(255 * powf(cl, 0.6f));
    }
}

// The tone-mapped frame buffer of the windowed path (SppmTracer::rendering,
// src/tracer/TkSppmTracer.cpp:85-97): one RGBA8 word per pixel, red in the low byte, alpha 255, in
// the order of the SPPMPixel array (no row flip: glDrawPixels draws bottom-up).
__global__ void k_preview(const float* planes, int64_t np, int num_recurse, float shoot_g, float shoot_c,
                          int caustic_stored, uint32_t* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= np) return;
    float cr2 = planes[11 * np + i], gr2 = planes[12 * np + i];
    uint32_t v = 0xFF000000u;
    for (int ch = 0; ch < 3; ++ch) {
        float c = planes[ch * np + i] / (float)num_recurse;
        if (caustic_stored > 0) c = c + planes[(3 + ch) * np + i] / (kPi * cr2 * shoot_c);
        else c = c + 0.f;
        c = c + planes[(6 + ch) * np + i] / (kPi * gr2 * shoot_g);
        float cl = sse_max(sse_min(c, 1.0f), 0.0f);
        v |= (uint32_t)(255 * powf(cl, 0.6f)) << (8 * ch);
    }
    out[i] = v;
}

__global__ void k_init_planes(float* planes, int64_t np, float radius2) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= np) return;
    for (int pl = 0; pl < 11; ++pl) planes[pl * np + i] = 0.f;
    planes[11 * np + i] = radius2;
    planes[12 * np + i] = radius2;
}

__global__ void k_planes_to_aos(const float* planes, int64_t np, float* aos) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= np) return;
    for (int pl = 0; pl < 13; ++pl) aos[13 * i + pl] = planes[pl * np + i];
}

__global__ void k_unpack_dir(const uint16_t* dir, uint8_t* th, uint8_t* ph, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    th[i] = (uint8_t)(dir[i] & 255u);
    ph[i] = (uint8_t)(dir[i] >> 8);
}

// ------------------------------------------------ photon exchange between ranks
// One buffer of 32-bit words per rank: a header (per map: photons stored, light paths shot as 31 + 33 bits),
// then per map the positions (3 words per photon), the powers (3) and the direction byte pairs (two photons
// to a word), every array sized for the largest quota of any rank.  All ranks' buffers back to back are what
// an all-gather delivers; k_exchange_unpack makes one map's arrays of them without the host reading a word.
constexpr int kExchangeHeaderWords = 8;
struct ExchangeLayout {
    int64_t cap[2], pos[2], pow[2], dir[2], words;
};
__global__ void k_exchange_header(int32_t* buf, long long n0, long long shot0, long long n1, long long shot1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    buf[0] = (int32_t)n0; buf[1] = (int32_t)(shot0 & 0x7fffffffll); buf[2] = (int32_t)(shot0 >> 31);
    buf[3] = (int32_t)n1; buf[4] = (int32_t)(shot1 & 0x7fffffffll); buf[5] = (int32_t)(shot1 >> 31);
    buf[6] = 0; buf[7] = 0;
}
__global__ void k_exchange_unpack(const int32_t* gathered, int64_t words, int world, int which, int64_t cap, int64_t o_pos,
                                  int64_t o_pow, int64_t o_dir, float* pos, float* pow, uint16_t* dir, int64_t limit,
                                  long long* totals_host) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i == 0) {
        long long n = 0, shot = 0;
        for (int q = 0; q < world; ++q) {
            const int32_t* h = gathered + q * words + 3 * which;
            n += h[0];
            shot += (long long)h[1] + ((long long)h[2] << 31);
        }
        totals_host[2 * which] = n;
        totals_host[2 * which + 1] = shot;
    }
    if (i >= (int64_t)world * cap) return;
    const int r = (int)(i / cap);
    const int64_t j = i - (int64_t)r * cap;
    const int32_t* mine = gathered + r * words;
    if (j >= mine[3 * which]) return;
    int64_t at = j;                                   // photons of the ranks before this one come first
    for (int q = 0; q < r; ++q) at += gathered[q * words + 3 * which];
    if (at >= limit) return;                          // never beyond the map's arrays, whatever the headers say
    const float* sp = reinterpret_cast<const float*>(mine + o_pos) + 3 * j;
    const float* sw = reinterpret_cast<const float*>(mine + o_pow) + 3 * j;
    pos[3 * at] = sp[0]; pos[3 * at + 1] = sp[1]; pos[3 * at + 2] = sp[2];
    pow[3 * at] = sw[0]; pow[3 * at + 1] = sw[1]; pow[3 * at + 2] = sw[2];
    dir[at] = reinterpret_cast<const uint16_t*>(mine + o_dir)[j];
}

// ----------------------------------------------------------------- host side
static ExchangeLayout exchange_layout(const SppmState* s) {
    ExchangeLayout L{};
    int64_t off = kExchangeHeaderWords;
    for (int w = 0; w < 2; ++w) {
        const int64_t size = w == 0 ? s->prm.globalSize : s->prm.causticSize;
        const int64_t cap = std::max<int64_t>(size / s->world + (size % s->world ? 1 : 0), 1);   // rank 0's quota: the largest
        L.cap[w] = cap; L.pos[w] = off; L.pow[w] = off + 3 * cap; L.dir[w] = off + 6 * cap;
        off += 6 * cap + (cap + 1) / 2;
    }
    L.words = off;
    return L;
}

void destroy_sppm(tkhip_ctx* ctx) {
    SppmState* s = ctx->sppm;
    if (!s) return;
    // passes begun ahead (tkhip_sppm_photon_begin, tkhip_sppm_camera_begin, tkhip_sppm_camera_pass_ahead) may
    // still be running: nothing is freed under them
    for (Lane& l : s->lane)
        if (l.stream) (void)hipStreamSynchronize(l.stream);
    (void)hipStreamSynchronize(ctx->stream);
    for (int w = 0; w < 2; ++w) {
        if (s->map[w]) map_destroy(s->map[w]);
        s->loc_pos[w].release(); s->loc_pow[w].release(); s->loc_dir[w].release(); s->qbuf[0][w].release(); s->qbuf[1][w].release();
    }
    s->planes.release();
    s->rows.release();
    s->preview.release();
    if (s->ev_exchange) (void)hipEventDestroy(s->ev_exchange);
    if (s->ev_preview) (void)hipEventDestroy(s->ev_preview);
    if (s->h_exchange) (void)hipHostFree(s->h_exchange);
    for (DevBuf* b : {&s->hg_prm, &s->hg_bounds, &s->hg_count, &s->hg_start, &s->hg_cursor, &s->hg_items, &s->hg_phi, &s->hg_m})
        b->release();
    if (s->lane[0].stream) {
        hipStream_t three[3] = {s->lane[0].stream, s->lane[1].stream, s->lane[2].stream};
        for (hipStream_t st : three) (void)hipStreamSynchronize(st);
        give_streams(ctx->device, three);
    }
    for (Lane& l : s->lane) l.release();
    delete s;
    ctx->sppm = nullptr;
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Rays, nodes fetched and primitives tested (C_RAYS, C_NODES, C_PRIMS: consecutive) of a lane's pass; cleared.
static int read_walk_counters(Lane& l, int64_t out[3], bool counting) {
    unsigned long long v[7] = {0, 0, 0, 0, 0, 0, 0};
    int32_t* c = l.counters.as<int32_t>();
    TK_HIP(hipMemcpyAsync(v, &c[C_RAYS], sizeof v, hipMemcpyDeviceToHost, l.stream));
    TK_HIP(hipStreamSynchronize(l.stream));
    TK_HIP(hipMemsetAsync(&c[C_RAYS], 0, sizeof v, l.stream));
    for (int i = 0; i < 3; ++i) out[i] = (int64_t)v[i];
    static const bool verbose = getenv("TKHIP_VERBOSE") != nullptr;
    if (verbose && counting && v[3] > 0)   // counting instances: what share of the lanes the sections of the walk keep busy
        fprintf(stderr, "tkhip: walks of this lane's pass: node loop %.3g lane-turns, %.1f %% with a ray under way, %.1f %% visiting a node; "
                "leaf phase %.3g lane-turns, %.1f %% testing a leaf\n", (double)v[3], 100.0 * v[4] / v[3], 100.0 * v[1] / v[3],
                (double)v[5], v[5] ? 100.0 * v[6] / v[5] : 0.0);
    return TKHIP_OK;
}

// Largest batch of light paths: 16 M, and no more than 4 GiB of photon staging (32 bytes per slot;
// a scene that yields few photons per path would otherwise ask for batches of 10 GB and more).
static int64_t photon_batch_cap(int slots) {
    return std::min<int64_t>((int64_t)1 << 24, ((int64_t)1 << 32) / (32 * (int64_t)std::max(slots, 1)));
}

// Enqueues one batch of light paths of a photon pass on its lane: the path
// kernel, the scan over the per-path photon counts, the scatter into the local
// store and the read-back of the batch counters.  Nothing is waited for.
template <int KIND>
static int photon_launch_batch(tkhip_ctx* ctx, int64_t stored, int64_t shot, int64_t* B_out) {
    SppmState* s = ctx->sppm;
    Lane& ln = s->lane[KIND];
    hipStream_t stream = ln.stream;
    const int64_t quota = s->quota[KIND];
    const int slots = KIND == 0 ? s->max_depth : 1;
    const int depth = KIND == 0 ? s->max_depth : 64;
    int32_t* counters = ln.counters.as<int32_t>();
    const int64_t need = quota - stored;
    int64_t B = (int64_t)((double)need / s->yield[KIND] * 1.02) + 256;
    B = std::min<int64_t>(B, photon_batch_cap(slots));
    TK_TRY(ln.stage_pos.ensure((size_t)B * slots * 16));
    TK_TRY(ln.stage_pow.ensure((size_t)B * slots * 16));
    TK_TRY(ln.rec_count.ensure((size_t)B * 4));
    TK_TRY(ln.rec_offset.ensure((size_t)B * 4));
    PhotonArgs a;
    a.sc = ctx->scene;
    static const int defer_below = getenv("TKHIP_DEFER_BELOW") ? std::min(std::max(atoi(getenv("TKHIP_DEFER_BELOW")), 0), 64) : 16;   // tuning aid
    a.defer_below = defer_below;
    TK_TRY(ln.q_o.ensure((size_t)B * 16));
    TK_TRY(ln.q_d.ensure((size_t)B * 16));
    TK_TRY(ln.q_w.ensure((size_t)B * 16));
    a.q = PathQueue{ln.q_o.as<float4>(), ln.q_d.as<float4>(), ln.q_w.as<float4>(), (int)B};
    a.rec_count = ln.rec_count.as<int32_t>();
    a.stage_pos = ln.stage_pos.as<float4>();
    a.stage_pow = ln.stage_pow.as<float4>();
    a.m0 = s->local_shot[KIND] + shot;
    // caustic index = j + shootGlobalPhotons (TkSppmTracer.cpp:419), the global count taken at the
    // start of the iteration so that the two passes do not depend on each other
    a.hoff = KIND == 0 ? 0 : s->caustic_hoff;
    a.world = s->world; a.rank = s->rank; a.seed = s->seed; a.rr = s->rr; a.slots = slots; a.depth = depth;
    TK_HIP(hipMemsetAsync(counters, 0, kCounterBatchInts * 4, stream));
    if (s->counting) k_photon_first<KIND, true><<<divup(B, kTraceBlock), kTraceBlock, 0, stream>>>(a, B, counters);
    else k_photon_first<KIND, false><<<divup(B, kTraceBlock), kTraceBlock, 0, stream>>>(a, B, counters);
    // Second-stage grid: enough waves for `photon_loads` path loads each of what the queue held last
    // time (the long paths need the refill to keep their waves full), at most what fits the GPU.  The
    // photon passes share the GPU with the camera pass: every wave slot a light-path kernel does not
    // hold goes to the camera paths.  tkhip_sppm_iteration steers photon_loads so that the photon chains
    // (pass, then tree build) end shortly before the camera paths do: with the dragon it settles at 5-6
    // (197 -> 208 iterations/s against a fixed 2; at 7 the caustic pass would be the longest chain).
    int64_t want = divup(B, kPathBlock);
    if (ln.queued_last >= 0) want = std::max<int64_t>(divup(ln.queued_last * B / std::max<int64_t>(ln.batch_last, 1), s->photon_loads * kPathBlock), 256);
    // A rank of several traces a share of the paths and leaves much of its GPU idle: there a fixed
    // share of the wave slots, over which the kernel spreads the queue thinly, is the faster shape.
    const int waves = s->world > 1 ? std::max(s->path_waves[KIND] * s->share[KIND] / 16, 64) : (int)std::min<int64_t>(want, s->path_waves[KIND]);
    if (s->counting) k_photon_paths<KIND, true><<<waves, kPathBlock, 0, stream>>>(a, counters);
    else k_photon_paths<KIND, false><<<waves, kPathBlock, 0, stream>>>(a, counters);
    TK_LAUNCH_CHECK();
    TK_TRY(exclusive_scan_i32(stream, ln.prim_temp, ln.rec_count.as<int32_t>(), ln.rec_offset.as<int32_t>(), (size_t)B));
    k_photon_scatter<<<divup(B, 256), 256, 0, stream>>>(ln.rec_count.as<int32_t>(), ln.rec_offset.as<int32_t>(),
                                                        ln.stage_pos.as<float4>(), ln.stage_pow.as<float4>(), slots, B, stored,
                                                        quota, s->loc_pos[KIND].as<float>(), s->loc_pow[KIND].as<float>(),
                                                        s->loc_dir[KIND].as<uint16_t>(), counters);
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(ln.h_counters, counters, kCounterBytes, hipMemcpyDeviceToHost, stream));
    *B_out = B;
    return TKHIP_OK;
}

// Start of a photon pass: clears the ray counter and enqueues the first batch.
template <int KIND>
static int photon_pass_begin(tkhip_ctx* ctx, int64_t* B_out) {
    SppmState* s = ctx->sppm;
    Lane& ln = s->lane[KIND];
    *B_out = 0;
    if (s->export_pending) {   // tkhip_sppm_export_exchange is still copying the arrays this pass overwrites
        TK_HIP(hipEventSynchronize(s->ev_exchange));
        s->export_pending = false;
    }
    if (s->quota[KIND] <= 0) return TKHIP_OK;
    TK_HIP(hipMemsetAsync(&ln.counters.as<int32_t>()[C_RAYS], 0, 56, ln.stream));
    return photon_launch_batch<KIND>(ctx, 0, 0, B_out);
}

// Rest of the pass: takes the result of the batch in flight and enqueues
// further batches until the quota is met.  Returns with the lane's stream idle.
template <int KIND>
static int photon_pass_finish(tkhip_ctx* ctx, int64_t B) {
    SppmState* s = ctx->sppm;
    Lane& ln = s->lane[KIND];
    const int64_t quota = s->quota[KIND];
    int64_t stored = 0, shot = 0;
    // "Emit until the map is full" never ends in a scene that cannot take photons (no diffuse surface
    // in reach for the global map, nothing specular for the caustic map): both passes give up after
    // 4000 paths per photon asked for.
    const int64_t emitted_cap = quota * 4000 + 100000;
    while (B > 0) {
        TK_HIP(hipStreamSynchronize(ln.stream));
        const int64_t need = quota - stored;
        ln.queued_last = (int64_t)ln.h_counters[C_QB] + ln.h_counters[C_QD];
        ln.batch_last = B;
        const int64_t total = ln.h_counters[C_TOTAL];
        const int64_t got = std::min<int64_t>(total, need);
        const int64_t used = total >= need ? ln.h_counters[C_USED] : B;
        stored += got;
        shot += used;
        if (used > 0 && got > 0) s->yield[KIND] = 0.5 * s->yield[KIND] + 0.5 * ((double)got / (double)used);
        else if (got == 0) s->yield[KIND] *= 0.25;
        s->yield[KIND] = std::max(s->yield[KIND], 1e-4);
        B = 0;
        if (stored < quota && shot < emitted_cap) TK_TRY(photon_launch_batch<KIND>(ctx, stored, shot, &B));
    }
    s->n_loc[KIND] = stored;
    s->last_shot[KIND] = shot;
    s->local_shot[KIND] += shot;
    TK_REQUIRE(!(KIND == 0 && stored == 0 && quota > 0), TKHIP_E_STATE,
               "photon pass: %lld light paths stored no photon for the global map (no diffuse surface in reach of the lights)",
               (long long)shot);
    return TKHIP_OK;
}

template <int KIND>
static int photon_pass_kind(tkhip_ctx* ctx) {
    int64_t B = 0;
    TK_TRY(photon_pass_begin<KIND>(ctx, &B));
    return photon_pass_finish<KIND>(ctx, B);
}

static int install_map(tkhip_ctx* ctx, int which, const float* pos, const float* pow, const uint16_t* dir, int64_t n,
                       int64_t shot_total) {
    SppmState* s = ctx->sppm;
    TK_TRY(map_set_photons_device(s->map[which], pos, pow, dir, n));
    s->shoot[which] += shot_total;
    if (which == 1) s->stored_c = n;
    return TKHIP_OK;
}
static int install_local(tkhip_ctx* ctx, int which) {
    SppmState* s = ctx->sppm;
    return install_map(ctx, which, s->loc_pos[which].as<float>(), s->loc_pow[which].as<float>(),
                       s->loc_dir[which].as<uint16_t>(), s->n_loc[which], s->last_shot[which]);
}

// Camera paths of this rank's pixels on lane 2: direct light into the pixel
// planes, gather queries into the two query buffers.  The photon maps are not
// read here, so this runs beside the photon passes.  expect_caustic: emit
// caustic queries (they are dropped later if the caustic map stays empty).
static int camera_trace_launch(tkhip_ctx* ctx, int iter, bool expect_caustic, bool ahead = false) {
    SppmState* s = ctx->sppm;
    s->stale_radius[s->camera_set] = ahead;   // ahead: the gathers of the iteration before have not updated the radii yet
    Lane& ln = s->lane[2];
    hipStream_t stream = ln.stream;
    int32_t* counters = ln.counters.as<int32_t>();
    s->camera_in_flight = iter;
    if (s->preview_pending) {   // a preview of the iteration before still reads the planes (it is a kernel of microseconds)
        TK_HIP(hipEventSynchronize(s->ev_preview));
        s->preview_pending = false;
    }
    if (s->n_slots == 0) return TKHIP_OK;
    CameraArgs a;
    a.sc = ctx->scene; a.cam = ctx->cam;
    a.rows = s->rows.as<int32_t>();
    a.planes = s->planes.as<float>();
    a.n_pixels = s->n_pixels;
    a.iter = iter; a.spp = s->num_recurse; a.xs = s->xs; a.ys = s->ys; a.depth = s->max_depth;
    a.seed = s->seed;
    a.no_jitter = s->no_jitter;
    a.caustic_stored = expect_caustic ? 1 : 0;
    a.visible_only = s->hashgrid ? 1 : 0;
    a.ld_scale = s->hashgrid ? 1.0f / (float)s->hg_spp : 1.0f;
    a.pt_mode = s->pt_mode ? 1 : 0;
    a.rr = s->rr;
    a.gq = s->qbuf[s->camera_set][0].as<float4>();
    a.cq = s->qbuf[s->camera_set][1].as<float4>();
    a.qcap = s->n_slots;
    a.q = PathQueue{ln.q_o.as<float4>(), ln.q_d.as<float4>(), ln.q_w.as<float4>(), (int)s->n_slots};
    TK_HIP(hipMemsetAsync(counters, 0, kCounterBatchInts * 4, stream));
    TK_HIP(hipMemsetAsync(&counters[C_RAYS], 0, 56, stream));
#ifdef TK_CP_STAMPS
    TK_HIP(hipMemsetAsync(&counters[224], 0, 64, stream));
#endif
    if (s->counting) k_camera_first<true><<<divup(s->n_slots, kTraceBlock), kTraceBlock, 0, stream>>>(a, s->n_slots, counters);
    else k_camera_first<false><<<divup(s->n_slots, kTraceBlock), kTraceBlock, 0, stream>>>(a, s->n_slots, counters);
    const int waves = s->world > 1 ? std::max(s->path_waves[2] * s->share[2] / 16, 64)
                                   : (int)std::min<int64_t>(divup(s->n_slots, kPathBlock), std::max(s->path_waves[2] * s->camera_share / 16, 64));
    if (s->counting) k_camera_paths<true><<<waves, kPathBlock, 0, stream>>>(a, s->world > 1 ? 1 : 0, counters);
    else k_camera_paths<false><<<waves, kPathBlock, 0, stream>>>(a, s->world > 1 ? 1 : 0, counters);
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(ln.h_counters, counters, kCounterBytes, hipMemcpyDeviceToHost, stream));
    return TKHIP_OK;
}
// Waits for the camera paths launched last and takes their query counts.
static int camera_trace_wait(tkhip_ctx* ctx) {
    SppmState* s = ctx->sppm;
    Lane& ln = s->lane[2];
    s->camera_in_flight = -1;
    s->gather_set = s->camera_set;
    s->n_queries[0] = s->n_queries[1] = 0;
    if (s->n_slots == 0) return TKHIP_OK;
    TK_HIP(hipStreamSynchronize(ln.stream));
    TK_TRACE("camera paths done");
    s->n_queries[0] = ln.h_counters[C_NGQ];
    s->n_queries[1] = ln.h_counters[C_NCQ];
    if (getenv("TKHIP_DEBUG_STAMPS")) {   // a profiling build (TK_CP_STAMPS) of k_camera_paths leaves its slowest wave's figures here
        const unsigned long long* u = reinterpret_cast<const unsigned long long*>(&ln.h_counters[224]);
        fprintf(stderr, "tkhip: camera paths: slowest main loop %.1f us, wide loop %.1f us, turns %llu / %llu, paths into the wide loop %llu, "
                "at queue end %lld; mean wave %.1f us of %llu waves\n", u[0] * 0.01, u[1] * 0.01, u[2], u[3], u[4], (long long)u[5] - 1,
                u[7] ? u[6] * 0.01 / u[7] : 0.0, u[7]);
    }
    return TKHIP_OK;
}
static int camera_trace(tkhip_ctx* ctx, int iter, bool expect_caustic) {
    TK_TRY(camera_trace_launch(ctx, iter, expect_caustic));
    return camera_trace_wait(ctx);
}

// radiance_estimate + SPPMPixel update for the queries of both maps (the two maps update disjoint planes).
// The grouped kernel of the two gathers is ONE launch (map_gather_pair): the query order of each map is made
// on its own stream, the launch takes both maps' groups, and each map's per-query kernel follows on its own
// stream again.  Everything is enqueued from this thread; both streams have drained when it returns.
static int camera_gather_both(tkhip_ctx* ctx, bool timed) {
    SppmState* s = ctx->sppm;
    TK_TRACE("gather_both: nq %lld %lld stored_c %lld timed %d", (long long)s->n_queries[0], (long long)s->n_queries[1], (long long)s->stored_c, (int)timed);
    int64_t nq[2] = {s->n_queries[0], s->n_queries[1]};
    if (s->stored_c == 0) nq[1] = 0;   // src/tracer/TkSppmTracer.cpp:323
    QueryView qv[2];
    GatherOut out[2];
    unsigned long long before[2][2] = {{0, 0}, {0, 0}};
    for (int which = 0; which < 2; ++which) {
        s->n_queries[which] = nq[which];
        tkhip_map* m = s->map[which];
        m->last_group_ms = 0.f;
        m->last_overflow = 0;
        if (nq[which] == 0) continue;
        float4* q = s->qbuf[s->gather_set][which].as<float4>();
        qv[which] = QueryView{q, q + s->n_slots, q + 2 * s->n_slots, q + 3 * s->n_slots};
        if (s->stale_radius[s->gather_set]) {
            // queries written while the gathers of the iteration before were still to come: their search
            // radii are the pixels' radii of now (this stream ran those gathers)
            k_query_radius<<<divup(nq[which], 256), 256, 0, m->stream>>>(q, q + 2 * s->n_slots, nq[which], s->planes.as<float>() +
                                                                         (which == 0 ? 12 : 11) * s->n_pixels);
            TK_LAUNCH_CHECK();
        }
        GatherOut o{};
        o.planes = s->planes.as<float>();
        o.n_pixels = s->n_pixels;
        o.which = which;
        // running totals of the map (read by tkhip_sppm_gather_totals and, per call, by the stats below)
        unsigned long long* totals = m->totals.as<unsigned long long>();
        o.photon_sum = totals;
        o.cand_sum = totals + 1;
        o.overflow_sum = totals + 2;
        out[which] = o;
        if (timed) {
            TK_HIP(hipMemcpyAsync(before[which], totals, 16, hipMemcpyDeviceToHost, m->stream));
            TK_HIP(hipStreamSynchronize(m->stream));
        }
    }
    if (nq[0] > 0 && nq[1] > 0) TK_TRY(map_gather_pair(s->map[0], nq[0], qv[0], out[0], s->map[1], nq[1], qv[1], out[1], timed));
    else
        for (int which = 0; which < 2; ++which)
            if (nq[which] > 0) TK_TRY(map_gather_device(s->map[which], nq[which], qv[which], out[which], 0, timed));
    TK_TRACE("gather_both: enqueued");
    for (int which = 0; which < 2; ++which) { TK_HIP(hipStreamSynchronize(s->map[which]->stream)); TK_TRACE("gather_both: stream %d drained", which); }
    if (timed) {
        for (int which = 0; which < 2; ++which) {
            if (nq[which] == 0) continue;
            unsigned long long after[2] = {0, 0};
            TK_HIP(hipMemcpy(after, s->map[which]->totals.p, 16, hipMemcpyDeviceToHost));
            s->last_gather_photons[which] = (int64_t)(after[0] - before[which][0]);
            s->last_gather_staged[which] = (int64_t)(after[1] - before[which][1]);
        }
    }
    return TKHIP_OK;
}

// Runs fn on a worker thread bound to the context's device; its error text is
// carried back to the caller's thread.
struct Worker {
    std::thread th;
    int rc = TKHIP_OK;
    std::string err;
    template <class F>
    void start(tkhip_ctx* ctx, F fn) {
        th = std::thread([this, ctx, fn]() {
            if (hipSetDevice(ctx->device) != hipSuccess) { rc = TKHIP_E_HIP; err = "hipSetDevice failed in worker"; return; }
            rc = fn();
            if (rc != TKHIP_OK) err = tkhip_last_error();
        });
    }
    int join() {
        if (th.joinable()) th.join();
        if (rc != TKHIP_OK) set_error("%s", err.c_str());
        return rc;
    }
};

static int build_both(tkhip_ctx* ctx, double* ms) {
    SppmState* s = ctx->sppm;
    double t0 = now_ms();
    Worker w;
    const bool caustic = s->stored_c > 0;   // src/tracer/TkSppmTracer.cpp:477-478
    if (caustic) w.start(ctx, [s]() { TK_TRY(map_build(s->map[1])); TK_HIP(hipStreamSynchronize(s->map[1]->stream)); return TKHIP_OK; });
    int rc = map_build(s->map[0]);
    if (rc == TKHIP_OK && hipStreamSynchronize(s->map[0]->stream) != hipSuccess) { set_error("build: stream sync failed"); rc = TKHIP_E_HIP; }
    int rc1 = w.join();
    if (ms) *ms += now_ms() - t0;
    return rc != TKHIP_OK ? rc : rc1;
}

static int gather_both(tkhip_ctx* ctx, TkIterStats* st) {
    SppmState* s = ctx->sppm;
    double t0 = now_ms();
    TK_TRY(camera_gather_both(ctx, st != nullptr));
    if (st) {
        st->ms_gather += (float)(now_ms() - t0);
        st->ms_gather_kernel += s->map[0]->last_group_ms + s->map[1]->last_group_ms;
        st->overflow_global += s->map[0]->last_overflow;
        st->overflow_caustic += s->map[1]->last_overflow;
        st->gathers_global += s->n_queries[0];
        st->gathers_caustic += s->n_queries[1];
        for (int which = 0; which < 2; ++which) {
            if (s->n_queries[which] == 0) continue;
            st->gather_photons += s->last_gather_photons[which];
            st->group_staged += s->last_gather_staged[which];
        }
    }
    return TKHIP_OK;
}

static int collect_rays(tkhip_ctx* ctx, int lane, TkIterStats* st) {
    if (!st) return TKHIP_OK;
    int64_t v[3] = {0, 0, 0};
    TK_TRY(read_walk_counters(ctx->sppm->lane[lane], v, ctx->sppm->counting));
    st->rays += v[0];
    if (lane == 2) { st->rays_camera += v[0]; st->node_visits_camera += v[1]; st->prim_tests_camera += v[2]; }
    else { st->node_visits_light += v[1]; st->prim_tests_light += v[2]; }
    return TKHIP_OK;
}

// After the camera pass of iteration `iter`: refreshes the device-resident preview when one is due.
// Enqueued on the context's stream, not waited for.
static int maybe_preview(tkhip_ctx* ctx, int iter, bool force) {
    SppmState* s = ctx->sppm;
    if (!s->preview.p) return TKHIP_OK;
    if (!force && (s->preview_every <= 0 || (iter + 1) % s->preview_every != 0)) return TKHIP_OK;
    k_preview<<<divup(s->n_pixels, 256), 256, 0, ctx->stream>>>(s->planes.as<float>(), s->n_pixels, s->num_recurse,
                                                               (float)s->shoot[0], (float)s->shoot[1], (int)s->stored_c,
                                                               s->preview.as<uint32_t>());
    TK_LAUNCH_CHECK();
    s->preview_iter = iter;
    // the next camera pass adds into the planes this kernel reads: its launch waits for the preview (camera_trace_launch)
    TK_HIP(hipEventRecord(s->ev_preview, ctx->stream));
    s->preview_pending = true;
    return TKHIP_OK;
}

static void fill_photon_stats(SppmState* s, TkIterStats* st) {
    st->global_paths = s->last_shot[0]; st->caustic_paths = s->last_shot[1];
    st->global_stored = s->n_loc[0]; st->caustic_stored = s->n_loc[1];
    st->shoot_global = s->shoot[0]; st->shoot_caustic = s->shoot[1];
}

}  // namespace tkd

using namespace tkd;

// Last part of tkhip_sppm_begin: the first uploads and warm-up launches.  On failure the caller
// synchronises and destroys the state.
static int sppm_begin_launches(tkhip_ctx* ctx, SppmState* s, const std::vector<int32_t>& rows, float radius2) {
    if (!rows.empty()) TK_HIP(hipMemcpyAsync(s->rows.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    k_init_planes<<<divup(s->n_pixels, 256), 256, 0, ctx->stream>>>(s->planes.as<float>(), s->n_pixels, radius2);
    TK_LAUNCH_CHECK();
    // One launch of each path kernel on an empty queue (the counters are zero): the first launch of a
    // kernel loads its code and sets up the scratch memory its register spills need, milliseconds
    // that would otherwise fall into the first iteration.
    {
        PhotonArgs pa{};
        pa.sc = ctx->scene;
        CameraArgs ca{};
        ca.sc = ctx->scene; ca.cam = ctx->cam;
        k_photon_paths<0><<<s->path_waves[0], kPathBlock, 0, s->lane[0].stream>>>(pa, s->lane[0].counters.as<int32_t>());
        k_photon_paths<1><<<s->path_waves[1], kPathBlock, 0, s->lane[1].stream>>>(pa, s->lane[1].counters.as<int32_t>());
        k_camera_paths<false><<<s->path_waves[2], kPathBlock, 0, s->lane[2].stream>>>(ca, 1, s->lane[2].counters.as<int32_t>());
        TK_LAUNCH_CHECK();
        // likewise the scan and the sort of the library the passes use (their first call is slow)
        for (int w = 0; w < 2; ++w) {
            Lane& ln = s->lane[w];
            if (s->quota[w] <= 0) continue;
            TK_HIP(hipMemsetAsync(ln.rec_count.p, 0, 256 * 4, ln.stream));   // the staging holds at least 256 paths
            TK_TRY(exclusive_scan_i32(ln.stream, ln.prim_temp, ln.rec_count.as<int32_t>(), ln.rec_offset.as<int32_t>(), 256));
            tkhip_map* m = s->map[w];
            if (m->cap >= 2048) {
                TK_HIP(hipMemsetAsync(m->key_a.p, 0, 2048 * 4, m->stream));
                TK_TRY(sort_pairs_u32(m->stream, m->prim_temp, m->key_a.as<uint32_t>(), m->key_b.as<uint32_t>(), m->idx_a.as<uint32_t>(),
                                      m->idx_b.as<uint32_t>(), 2048, 30));
            }
        }
    }
    TK_HIP(hipStreamSynchronize(ctx->stream));
    for (Lane& l : s->lane) TK_HIP(hipStreamSynchronize(l.stream));
    return TKHIP_OK;
}

namespace tkd {
// ------------------------------------------------ hash-grid SPPM (src/TkSPPM.cpp)
TKD uint32_t hg_f2ord(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
TKD float hg_ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// Bounds of the search boxes of the visible points and their largest radius (src/TkSPPM.cpp:49-62).
// bounds[0..2] min, [3..5] max, [6] largest radius, all as order-preserving integers.
__global__ void __launch_bounds__(256) k_hg_bounds(const float4* vp0, int64_t n, uint32_t* bounds) {
    // every thread takes a strided share, a block reduces to one set of seven atomics (one set per wave of a
    // thread per point was 115 k atomics on seven addresses: 1.1 ms for a million points)
    __shared__ float s_part[4][7];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, r = -INFINITY;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = vp0[i];
        lo[0] = fminf(lo[0], v.x - v.w); lo[1] = fminf(lo[1], v.y - v.w); lo[2] = fminf(lo[2], v.z - v.w);
        hi[0] = fmaxf(hi[0], v.x + v.w); hi[1] = fmaxf(hi[1], v.y + v.w); hi[2] = fmaxf(hi[2], v.z + v.w);
        r = fmaxf(r, v.w);
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
        r = fmaxf(r, __shfl_xor(r, off));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; ++a) { s_part[wave][a] = lo[a]; s_part[wave][3 + a] = hi[a]; }
        s_part[wave][6] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], s_part[w][a]); hi[a] = fmaxf(hi[a], s_part[w][3 + a]); }
            r = fmaxf(r, s_part[w][6]);
        }
        if (r > -INFINITY) {
            for (int a = 0; a < 3; ++a) {
                atomicMin(&bounds[a], hg_f2ord(lo[a]));
                atomicMax(&bounds[3 + a], hg_f2ord(hi[a]));
            }
            atomicMax(&bounds[6], hg_f2ord(r));
        }
    }
}
// Grid resolution (src/TkSPPM.cpp:64-68).
__global__ void k_hg_setup(const uint32_t* bounds, int64_t n, HashGridParams* out) {
    HashGridParams g;
    g.valid = n > 0 ? 1 : 0;
    float diag[3];
    for (int a = 0; a < 3; ++a) {
        g.bmin[a] = n > 0 ? hg_ord2f(bounds[a]) : 0.f;
        g.bmax[a] = n > 0 ? hg_ord2f(bounds[3 + a]) : 0.f;
        diag[a] = g.bmax[a] - g.bmin[a];
        g.res[a] = 1;
    }
    if (n > 0) {
        const float max_radius = hg_ord2f(bounds[6]);
        const int ext = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);   // Bounds3::maxExtent
        const float max_diag = diag[ext];
        const int base = (int)(max_diag / max_radius);
        for (int a = 0; a < 3; ++a) {
            const int r = (int)((float)base * diag[a] / max_diag);
            g.res[a] = r > 1 ? r : 1;
        }
    }
    *out = g;
}
// Every visible point into all the cells its search box overlaps (src/TkSPPM.cpp:70-85): FILL = false
// counts per bucket, FILL = true writes the lists.
template <bool FILL>
__global__ void k_hg_insert(const float4* vp0, int64_t n, const HashGridParams* prm, uint32_t buckets, int32_t* count,
                            const int32_t* start, int32_t* items) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const HashGridParams g = *prm;
    const float4 v = vp0[i];
    int lo[3], hi[3];
    hg_world_to_grid(g, mk3(v.x - v.w, v.y - v.w, v.z - v.w), lo);
    hg_world_to_grid(g, mk3(v.x + v.w, v.y + v.w, v.z + v.w), hi);
    for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
            for (int x = lo[0]; x <= hi[0]; ++x) {
                const uint32_t b = hg_hash(x, y, z, buckets);
                const int at = atomicAdd(&count[b], 1);
                if (FILL) items[start[b] + at] = (int)i;
            }
}
// updatePixel (src/TkSPPM.cpp:252-272) for the pixels that have a visible point this iteration.
// Planes: 3..5 Tau, 9 N, 11 radius.
__global__ void k_hg_update(const float4* vp2, const float* phi, const int32_t* m, int64_t n, float* planes, int64_t np) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int M = m[i];
    if (M <= 0) return;
    const float4 t = vp2[i];   // throughput, pixel
    const int64_t px = __float_as_int(t.w);
    const float alpha = 0.7f;
    const float N = planes[9 * np + px], radius = planes[11 * np + px];
    const float newN = N + alpha * (float)M;
    const float newR = radius * sqrtf(newN / (N + (float)M));
    const float thr[3] = {t.x, t.y, t.z};
    for (int c = 0; c < 3; ++c) {
        const float tau = planes[(3 + c) * np + px];
        planes[(3 + c) * np + px] = (tau + thr[c] * phi[c * n + i]) * (newR * newR) / (radius * radius);
    }
    planes[9 * np + px] = newN;
    planes[11 * np + px] = newR;
}
// Film::writeImage (src/core/TkFilm.cpp:124-153): contribSum / filterWeightSum, gamma 0.6, rows bottom-up.
// Planes: 0..2 contribSum, 9 filterWeightSum.
__global__ void k_pt_resolve(const float* planes, int64_t np, int W, int H, uint8_t* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= np) return;
    const int x = (int)(i % W), y = (int)(i / W);
    const float wsum = planes[9 * np + i];
    uint8_t* o = out + ((int64_t)(H - 1 - y) * W + x) * 3;
    for (int c = 0; c < 3; ++c) {
        float v = planes[c * np + i];
        if (wsum != 0) v = v / wsum;
        const float cl = sse_max(sse_min(v, 1.0f), 0.0f);
        o[c] = (uint8_t)(255 * powf(cl, 0.6f));
    }
}
// The final image (src/TkSPPM.cpp:436-450), rows bottom-up.
__global__ void k_hg_resolve(const float* planes, int64_t np, int W, int H, int iterations, int photons_per_it, uint8_t* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= np) return;
    const int x = (int)(i % W), y = (int)(i / W);
    const float radius = planes[11 * np + i];
    const float denom = (float)(iterations * photons_per_it) * kPi * radius * radius;
    uint8_t* o = out + ((int64_t)(H - 1 - y) * W + x) * 3;
    for (int c = 0; c < 3; ++c) {
        const float v = planes[c * np + i] / (float)iterations + planes[(3 + c) * np + i] / denom;
        const float cl = sse_max(sse_min(v, 1.0f), 0.0f);
        o[c] = (uint8_t)(255 * powf(cl, 0.6f));
    }
}
}  // namespace tkd

namespace tkd {
int sppm_local_photons(tkhip_ctx* ctx, int which, const void** pos, const void** pow, const void** dir, int64_t* n,
                       int64_t* shot) {
    TK_REQUIRE(ctx && ctx->sppm && (which == 0 || which == 1), TKHIP_E_STATE, "sppm_local_photons: no render on this context");
    SppmState* s = ctx->sppm;
    *pos = s->loc_pos[which].p;
    *pow = s->loc_pow[which].p;
    *dir = s->loc_dir[which].p;
    *n = s->n_loc[which];
    *shot = s->last_shot[which];
    return TKHIP_OK;
}
// Camera paths begun ahead (tkhip_sppm_camera_pass_ahead, tkhip_sppm_camera_begin) add direct light into the
// pixel planes on their own stream: whoever reads the planes on another stream waits for them first.  The
// pass stays "in flight" for the iteration that picks it up.
int sppm_wait_camera_lane(tkhip_ctx* ctx) {
    SppmState* s = ctx->sppm;
    if (s && s->camera_in_flight >= 0 && s->lane[2].stream) TK_HIP(hipStreamSynchronize(s->lane[2].stream));
    return TKHIP_OK;
}
float* sppm_planes_device(tkhip_ctx* ctx) { return ctx->sppm ? ctx->sppm->planes.as<float>() : nullptr; }
}  // namespace tkd

#define TK_CTX_SPPM(ctx, fn)                                                             \
    TK_REQUIRE((ctx) != nullptr, TKHIP_E_ARG, fn ": null context");                      \
    TK_REQUIRE((ctx)->sppm != nullptr, TKHIP_E_STATE, fn ": call tkhip_sppm_begin first"); \
    TK_HIP(hipSetDevice((ctx)->device))

extern "C" {

int tkhip_sppm_begin(tkhip_ctx* ctx, const TkSppmParam* param, int max_depth, float rr, uint64_t seed, int rank,
                     int world) {
    TK_REQUIRE(ctx && param, TKHIP_E_ARG, "tkhip_sppm_begin: null argument");
    TK_REQUIRE(ctx->has_scene && ctx->has_camera, TKHIP_E_STATE, "tkhip_sppm_begin: set the scene and the camera first");
    TK_REQUIRE(world >= 1 && rank >= 0 && rank < world, TKHIP_E_ARG, "tkhip_sppm_begin: bad rank %d / world %d", rank, world);
    TK_REQUIRE(max_depth >= 1 && max_depth <= 64, TKHIP_E_LIMIT, "tkhip_sppm_begin: depth %d outside 1..64", max_depth);
    TK_REQUIRE(param->globalSize >= 1 && param->causticSize >= 0, TKHIP_E_ARG, "tkhip_sppm_begin: bad photon map sizes");
    TK_REQUIRE(param->numRecurse >= 1, TKHIP_E_ARG, "tkhip_sppm_begin: recurse must be >= 1");
    TK_HIP(hipSetDevice(ctx->device));
    destroy_sppm(ctx);
    SppmState* s = new SppmState();
    ctx->sppm = s;
    s->prm = *param;
    s->max_depth = max_depth; s->rr = rr; s->seed = (uint32_t)seed; s->rank = rank; s->world = world;
    int x = 0, y = 0;
    while (++x * ++y < param->numRecurse) {}   // src/tracer/TkSppmTracer.cpp:18-21
    s->num_recurse = x * y; s->xs = x; s->ys = y;
    int rc = TKHIP_OK;
    {
        hipStream_t three[3] = {nullptr, nullptr, nullptr};
        rc = take_streams(ctx->device, three);
        if (rc) { destroy_sppm(ctx); return rc; }
        for (int l = 0; l < 3; ++l) s->lane[l].stream = three[l];
    }
    for (Lane& l : s->lane) {
        rc = l.counters.ensure(kCounterBytes);
        if (rc) break;
        if (hipHostMalloc(reinterpret_cast<void**>(&l.h_counters), kCounterBytes, hipHostMallocDefault) != hipSuccess) { set_error("tkhip_sppm_begin: pinned allocation failed"); rc = TKHIP_E_NOMEM; break; }
        std::memset(l.h_counters, 0, kCounterBytes);
        if (hipMemsetAsync(l.counters.p, 0, kCounterBytes, l.stream) != hipSuccess) { set_error("tkhip_sppm_begin: memset failed"); rc = TKHIP_E_HIP; break; }
    }
    if (!rc && (hipEventCreateWithFlags(&s->ev_exchange, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s->ev_preview, hipEventDisableTiming) != hipSuccess ||
                hipHostMalloc(reinterpret_cast<void**>(&s->h_exchange), 4 * sizeof(long long), hipHostMallocDefault) != hipSuccess)) {
        set_error("tkhip_sppm_begin: exchange set-up failed");
        rc = TKHIP_E_HIP;
    }
    if (!rc) {
        hipDeviceProp_t prop{};
        int per_cu[3] = {0, 0, 0};
        if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[0], k_photon_paths<0>, kPathBlock, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[1], k_photon_paths<1>, kPathBlock, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[2], k_camera_paths<false>, kPathBlock, 0) != hipSuccess) {
            set_error("tkhip_sppm_begin: occupancy query failed");
            rc = TKHIP_E_HIP;
        } else {
            if (const char* sv = getenv("TKHIP_SHARE")) sscanf(sv, "%d %d %d", &s->share[0], &s->share[1], &s->share[2]);   // tuning aid
            const char* ov = getenv("TKHIP_PATH_WAVES_PER_CU");   // tuning aid
            for (int l = 0; l < 3; ++l) {
                if (ov && atoi(ov) > 0) per_cu[l] = atoi(ov);
                s->path_waves[l] = std::max(per_cu[l], 1) * std::max(prop.multiProcessorCount, 1);
            }
            if (getenv("TKHIP_VERBOSE"))
                fprintf(stderr, "tkhip: path kernels resident waves per CU: %d %d %d on %d CUs\n", per_cu[0], per_cu[1], per_cu[2],
                        prop.multiProcessorCount);
        }
    }
    if (!rc) rc = map_create(ctx, param->globalSize, param->globalSample, &s->map[0]);
    if (!rc) rc = map_create(ctx, std::max(param->causticSize, 1), param->causticSample, &s->map[1]);
    if (rc) { destroy_sppm(ctx); return rc; }
    s->map[0]->stream = s->lane[0].stream;
    s->map[1]->stream = s->lane[1].stream;
    s->quota[0] = param->globalSize / world + (rank < param->globalSize % world ? 1 : 0);
    s->quota[1] = param->causticSize / world + (rank < param->causticSize % world ? 1 : 0);
    for (int w = 0; w < 2 && !rc; ++w) {
        size_t c = (size_t)std::max<int64_t>(s->quota[w], 1);
        rc = s->loc_pos[w].ensure(c * 12);
        if (!rc) rc = s->loc_pow[w].ensure(c * 12);
        if (!rc) rc = s->loc_dir[w].ensure(c * 2);
    }
    const int W = ctx->cam.w, H = ctx->cam.h;
    s->n_pixels = (int64_t)W * H;
    std::vector<int32_t> rows;
    for (int yy = 0; yy < H; ++yy)
        if ((yy / 8) % world == rank) rows.push_back(yy);
    s->n_rows = (int)rows.size();
    s->n_slots = (int64_t)s->n_rows * W;
    if (!rc) rc = s->planes.ensure((size_t)s->n_pixels * 13 * 4);
    if (!rc) rc = s->rows.ensure(std::max<size_t>(rows.size(), 1) * 4);
    for (int w = 0; w < 2 && !rc; ++w) rc = s->qbuf[0][w].ensure((size_t)std::max<int64_t>(s->n_slots, 1) * 64);
    for (DevBuf* b : {&s->lane[2].q_o, &s->lane[2].q_d, &s->lane[2].q_w})
        if (!rc) rc = b->ensure((size_t)std::max<int64_t>(s->n_slots, 1) * 16);
    // Everything an iteration would otherwise allocate on first use: the tree and gather scratch of
    // both maps, and the staging of a first batch of light paths (sized as photon_launch_batch does).
    for (int w = 0; w < 2 && !rc; ++w) {
        rc = map_reserve_build(s->map[w]);
        if (!rc) rc = map_reserve_queries(s->map[w], std::max<int64_t>(s->n_slots, 1));
        if (s->quota[w] <= 0) continue;
        Lane& ln = s->lane[w];
        const size_t slots = w == 0 ? (size_t)max_depth : 1;
        const size_t B = (size_t)std::min<int64_t>((int64_t)((double)s->quota[w] / s->yield[w] * 1.02) + 256, photon_batch_cap((int)slots));
        for (DevBuf* b : {&ln.stage_pos, &ln.stage_pow})
            if (!rc) rc = b->ensure(B * slots * 16);
        for (DevBuf* b : {&ln.rec_count, &ln.rec_offset})
            if (!rc) rc = b->ensure(B * 4);
        for (DevBuf* b : {&ln.q_o, &ln.q_d, &ln.q_w})
            if (!rc) rc = b->ensure(B * 16);
    }
    if (rc) { destroy_sppm(ctx); return rc; }
    rc = sppm_begin_launches(ctx, s, rows, param->radius2);
    if (rc != TKHIP_OK) {
        // nothing half-initialised stays behind: wait for what was enqueued, then drop the state
        (void)hipStreamSynchronize(ctx->stream);
        for (Lane& l : s->lane) if (l.stream) (void)hipStreamSynchronize(l.stream);
        destroy_sppm(ctx);
        return rc;
    }
    return TKHIP_OK;
}

int tkhip_sppm_photon_begin(tkhip_ctx* ctx, int iter) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_photon_begin");
    (void)iter;
    SppmState* s = ctx->sppm;
    TK_REQUIRE(!s->photon_begun, TKHIP_E_STATE, "tkhip_sppm_photon_begin: a photon pass is already in flight");
    s->caustic_hoff = s->local_shot[0];
    TK_TRY(photon_pass_begin<0>(ctx, &s->begun_batch[0]));      // both passes are in flight together
    int rc1 = photon_pass_begin<1>(ctx, &s->begun_batch[1]);
    if (rc1 != TKHIP_OK) {   // no batch stays in flight behind an error
        (void)hipStreamSynchronize(s->lane[0].stream);
        (void)hipStreamSynchronize(s->lane[1].stream);
        return rc1;
    }
    s->photon_begun = true;
    return TKHIP_OK;
}

int tkhip_sppm_photon_pass(tkhip_ctx* ctx, int iter, TkIterStats* st) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_photon_pass");
    SppmState* s = ctx->sppm;
    double t0 = now_ms();
    if (!s->photon_begun) TK_TRY(tkhip_sppm_photon_begin(ctx, iter));
    s->photon_begun = false;
    const int64_t Bg = s->begun_batch[0], Bc = s->begun_batch[1];
    TK_TRY(photon_pass_finish<0>(ctx, Bg));
    TK_TRY(photon_pass_finish<1>(ctx, Bc));
    if (s->world == 1) {
        TK_TRY(install_local(ctx, 0));
        TK_TRY(install_local(ctx, 1));
        TK_HIP(hipStreamSynchronize(s->lane[0].stream));
        TK_HIP(hipStreamSynchronize(s->lane[1].stream));
    }
    if (st) {
        TK_TRY(collect_rays(ctx, 0, st));
        TK_TRY(collect_rays(ctx, 1, st));
        st->ms_photon += (float)(now_ms() - t0);
        fill_photon_stats(s, st);
    }
    return TKHIP_OK;
}

int tkhip_sppm_export_local_photons(tkhip_ctx* ctx, int which, float* pos, float* power, uint16_t* dir, int64_t cap,
                                    int64_t* n, int64_t* shot) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_export_local_photons");
    TK_REQUIRE(which == 0 || which == 1, TKHIP_E_ARG, "tkhip_sppm_export_local_photons: which must be 0 or 1");
    SppmState* s = ctx->sppm;
    int64_t cnt = s->n_loc[which];
    TK_REQUIRE(cnt <= cap, TKHIP_E_LIMIT, "tkhip_sppm_export_local_photons: %lld photons, buffer holds %lld", (long long)cnt,
               (long long)cap);
    if (cnt > 0) {
        TK_REQUIRE(pos && power && dir, TKHIP_E_ARG, "tkhip_sppm_export_local_photons: null buffer");
        TK_HIP(hipMemcpyAsync(pos, s->loc_pos[which].p, (size_t)cnt * 12, hipMemcpyDeviceToDevice, ctx->stream));
        TK_HIP(hipMemcpyAsync(power, s->loc_pow[which].p, (size_t)cnt * 12, hipMemcpyDeviceToDevice, ctx->stream));
        TK_HIP(hipMemcpyAsync(dir, s->loc_dir[which].p, (size_t)cnt * 2, hipMemcpyDeviceToDevice, ctx->stream));
        TK_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (n) *n = cnt;
    if (shot) *shot = s->last_shot[which];
    return TKHIP_OK;
}

int tkhip_sppm_set_map_photons(tkhip_ctx* ctx, int which, const float* pos, const float* power, const uint16_t* dir,
                               int64_t n, int64_t shot_total) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_set_map_photons");
    TK_REQUIRE(which == 0 || which == 1, TKHIP_E_ARG, "tkhip_sppm_set_map_photons: which must be 0 or 1");
    TK_REQUIRE(n == 0 || (pos && power && dir), TKHIP_E_ARG, "tkhip_sppm_set_map_photons: null buffer");
    TK_TRY(install_map(ctx, which, pos, power, dir, n, shot_total));
    TK_HIP(hipStreamSynchronize(ctx->sppm->map[which]->stream));
    return TKHIP_OK;
}

int tkhip_sppm_exchange_words(tkhip_ctx* ctx, int64_t* words_per_rank) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_exchange_words");
    TK_REQUIRE(words_per_rank, TKHIP_E_ARG, "tkhip_sppm_exchange_words: null argument");
    *words_per_rank = exchange_layout(ctx->sppm).words;
    return TKHIP_OK;
}

int tkhip_sppm_export_exchange(tkhip_ctx* ctx, int32_t* device_buf, void* stream_handle) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_export_exchange");
    TK_REQUIRE(device_buf, TKHIP_E_ARG, "tkhip_sppm_export_exchange: null buffer");
    SppmState* s = ctx->sppm;
    TK_REQUIRE(!s->photon_begun, TKHIP_E_STATE, "tkhip_sppm_export_exchange: finish the photon pass first (tkhip_sppm_photon_pass)");
    hipStream_t stream = static_cast<hipStream_t>(stream_handle);
    const ExchangeLayout L = exchange_layout(s);
    // the photon pass has waited for its lanes: the rank's arrays are complete and can be read on any stream
    k_exchange_header<<<1, 64, 0, stream>>>(device_buf, (long long)s->n_loc[0], (long long)s->last_shot[0], (long long)s->n_loc[1],
                                            (long long)s->last_shot[1]);
    TK_LAUNCH_CHECK();
    for (int w = 0; w < 2; ++w) {
        const size_t n = (size_t)s->n_loc[w];
        if (n == 0) continue;
        TK_HIP(hipMemcpyAsync(device_buf + L.pos[w], s->loc_pos[w].p, n * 12, hipMemcpyDeviceToDevice, stream));
        TK_HIP(hipMemcpyAsync(device_buf + L.pow[w], s->loc_pow[w].p, n * 12, hipMemcpyDeviceToDevice, stream));
        TK_HIP(hipMemcpyAsync(device_buf + L.dir[w], s->loc_dir[w].p, n * 2, hipMemcpyDeviceToDevice, stream));
    }
    // the next photon pass overwrites the rank's arrays: it waits for these copies first (photon_pass_begin)
    TK_HIP(hipEventRecord(s->ev_exchange, stream));
    s->export_pending = true;
    return TKHIP_OK;
}

int tkhip_sppm_import_exchange(tkhip_ctx* ctx, const int32_t* device_gathered, void* stream_handle) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_import_exchange");
    TK_REQUIRE(device_gathered, TKHIP_E_ARG, "tkhip_sppm_import_exchange: null buffer");
    SppmState* s = ctx->sppm;
    TK_REQUIRE(s->world > 1, TKHIP_E_STATE, "tkhip_sppm_import_exchange: a single rank installs its photons in tkhip_sppm_photon_pass");
    hipStream_t stream = static_cast<hipStream_t>(stream_handle);
    const ExchangeLayout L = exchange_layout(s);
    // Whatever filled the buffer on the caller's stream (the all-gather) has to be over before the maps' streams
    // read it.  The host waits for that point of the caller's stream; it reads nothing and decides nothing there.
    // (A device-side wait of the maps' streams on the event would spare the host even that, but waits between
    // streams of different priority were seen to stall for good once a process holds more streams than the GPU
    // has hardware queues.)
    TK_HIP(hipEventRecord(s->ev_exchange, stream));
    TK_HIP(hipEventSynchronize(s->ev_exchange));
    s->export_pending = false;
    for (int w = 0; w < 2; ++w) {
        tkhip_map* m = s->map[w];
        const int64_t size = w == 0 ? s->prm.globalSize : s->prm.causticSize;
        s->exchange_expected[w] = size;
        s->h_exchange[2 * w] = -1;
        m->n = size;          // every rank fills its quota as a rule; tkhip_sppm_build_maps checks the sum the device found
        m->built = false;
        if (w == 1) s->stored_c = size;
        if (size == 0) { s->h_exchange[2 * w] = 0; s->h_exchange[2 * w + 1] = 0; continue; }
        k_exchange_unpack<<<divup((int64_t)s->world * L.cap[w], 256), 256, 0, m->stream>>>(
            device_gathered, L.words, s->world, w, L.cap[w], L.pos[w], L.pow[w], L.dir[w], m->in_pos.as<float>(), m->in_pow.as<float>(),
            m->in_dir.as<uint16_t>(), m->cap, s->h_exchange);
        TK_LAUNCH_CHECK();
    }
    s->exchange_pending = true;
    return TKHIP_OK;
}

// After the builds that follow an import (their streams have been waited for): the totals the device summed
// from the ranks' headers.  A rank that could not fill its quota (a scene that takes no caustic photons) makes
// a map smaller than assumed: its tree is built again over the photons there are.
static int exchange_settle(tkhip_ctx* ctx) {
    SppmState* s = ctx->sppm;
    s->exchange_pending = false;
    for (int w = 0; w < 2; ++w) {
        tkhip_map* m = s->map[w];
        TK_HIP(hipStreamSynchronize(m->stream));
        const long long n = s->h_exchange[2 * w], shot = s->h_exchange[2 * w + 1];
        TK_REQUIRE(n >= 0 && n <= s->exchange_expected[w], TKHIP_E_STATE, "photon exchange: the ranks report %lld photons for a map of %lld",
                   n, (long long)s->exchange_expected[w]);
        s->shoot[w] += shot;
        if (n == s->exchange_expected[w]) continue;
        m->n = n;
        if (w == 1) s->stored_c = n;
        if (w == 0 || n > 0) TK_TRY(map_build(m));
        TK_HIP(hipStreamSynchronize(m->stream));
    }
    return TKHIP_OK;
}

int tkhip_sppm_build_maps(tkhip_ctx* ctx, TkIterStats* st) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_build_maps");
    double ms = 0;
    TK_TRY(build_both(ctx, &ms));
    if (ctx->sppm->exchange_pending) TK_TRY(exchange_settle(ctx));
    if (st) st->ms_build += (float)ms;
    return TKHIP_OK;
}

int tkhip_sppm_set_tree(tkhip_ctx* ctx, int mode) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_set_tree");
    TK_TRY(tkhip_map_set_tree(ctx->sppm->map[0], mode));
    return tkhip_map_set_tree(ctx->sppm->map[1], mode);
}

int tkhip_sppm_set_counting(tkhip_ctx* ctx, int on) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_set_counting");
    ctx->sppm->counting = on != 0;
    return TKHIP_OK;
}

int tkhip_sppm_set_jitter(tkhip_ctx* ctx, int jitter) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_set_jitter");
    ctx->sppm->no_jitter = jitter ? 0 : 1;
    return TKHIP_OK;
}

int tkhip_sppm_camera_begin(tkhip_ctx* ctx, int iter) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_camera_begin");
    SppmState* s = ctx->sppm;
    if (s->camera_in_flight == iter) return TKHIP_OK;   // begun ahead by tkhip_sppm_camera_pass_ahead
    if (s->camera_in_flight >= 0) TK_TRY(camera_trace_wait(ctx));
    return camera_trace_launch(ctx, iter, s->prm.causticSize > 0);
}

static int camera_pass(tkhip_ctx* ctx, int iter, bool next_ahead, TkIterStats* st) {
    SppmState* s = ctx->sppm;
    double t0 = now_ms();
    if (s->camera_in_flight == iter) TK_TRY(camera_trace_wait(ctx));   // started by tkhip_sppm_camera_begin or ahead
    else {
        if (s->camera_in_flight >= 0) TK_TRY(camera_trace_wait(ctx));
        TK_TRY(camera_trace(ctx, iter, s->prm.causticSize > 0));
    }
    if (st) st->ms_camera += (float)(now_ms() - t0);
    TK_TRY(collect_rays(ctx, 2, st));
    if (next_ahead) {
        // The camera paths of the next iteration into the other set of query buffers, before this
        // iteration's gathers: they read nothing the gathers write except the pixels' radii, which
        // their queries take again when they are gathered.
        const int other = s->camera_set ^ 1;
        for (int w = 0; w < 2; ++w) TK_TRY(s->qbuf[other][w].ensure((size_t)std::max<int64_t>(s->n_slots, 1) * 64));
        s->camera_set = other;
        const int rc = camera_trace_launch(ctx, iter + 1, s->prm.causticSize > 0, true);
        if (rc != TKHIP_OK) {   // nothing of the next iteration stays behind: this one can still be gathered by a plain call
            (void)hipStreamSynchronize(s->lane[2].stream);
            s->camera_in_flight = -1;
            s->camera_set = s->gather_set;
            return rc;
        }
    }
    TK_TRY(gather_both(ctx, st));
    return maybe_preview(ctx, iter, false);
}

int tkhip_sppm_camera_pass(tkhip_ctx* ctx, int iter, TkIterStats* st) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_camera_pass");
    TK_REQUIRE(ctx->sppm->map[0]->built, TKHIP_E_STATE, "tkhip_sppm_camera_pass: build the photon maps first");
    return camera_pass(ctx, iter, false, st);
}

int tkhip_sppm_camera_pass_ahead(tkhip_ctx* ctx, int iter, TkIterStats* st) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_camera_pass_ahead");
    SppmState* s = ctx->sppm;
    TK_REQUIRE(s->map[0]->built, TKHIP_E_STATE, "tkhip_sppm_camera_pass_ahead: build the photon maps first");
    TK_REQUIRE(!s->hashgrid && !s->pt_mode, TKHIP_E_STATE, "tkhip_sppm_camera_pass_ahead: only for the SPPM tracer with photon maps");
    return camera_pass(ctx, iter, true, st);
}

// One iteration on a single GPU.  The camera paths do not read the photon
// maps until they gather, so three things run side by side: the global photon
// pass followed by the global tree build, the caustic photon pass followed by
// the caustic tree build, and the camera paths.  The gathers of both maps
// follow, again side by side.  The per-phase times of the stats therefore
// overlap.
int tkhip_sppm_iteration(tkhip_ctx* ctx, int iter, TkIterStats* st) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_iteration");
    SppmState* s = ctx->sppm;
    TK_REQUIRE(s->world == 1, TKHIP_E_STATE,
               "tkhip_sppm_iteration: with world > 1 drive the passes and the photon exchange separately");
    TK_REQUIRE(!s->photon_begun, TKHIP_E_STATE, "tkhip_sppm_iteration: finish the photon pass begun with tkhip_sppm_photon_begin first");
    const double t0 = now_ms();
    double t_photon_end[2] = {t0, t0}, build_ms[2] = {0, 0};
    // Enqueue in the order of the critical path: the global pass (the longest chain: pass, then
    // the larger tree), the caustic pass, the camera paths.  Each then fills what the others leave.
    // (The camera paths first, which the steering below makes the longer chain in steady state:
    // no change there, 4.71 ms per iteration of cornell_dragon either way, and 2 ms more for the
    // first iterations of a 16-iteration render.)
    s->caustic_hoff = s->local_shot[0];
    int64_t Bg = 0, Bc = 0;
    TK_TRY(photon_pass_begin<0>(ctx, &Bg));
    {
        int rc1 = photon_pass_begin<1>(ctx, &Bc);
        if (rc1 != TKHIP_OK) {   // no batch stays in flight behind an error
            (void)hipStreamSynchronize(s->lane[0].stream);
            (void)hipStreamSynchronize(s->lane[1].stream);
            return rc1;
        }
    }
    Worker wg, wc;
    wg.start(ctx, [&, ctx, s]() {
        TK_TRY(photon_pass_finish<0>(ctx, Bg));
        TK_TRY(install_local(ctx, 0));
        t_photon_end[0] = now_ms();
        double b0 = now_ms();
        TK_TRY(map_build(s->map[0]));
        TK_HIP(hipStreamSynchronize(s->map[0]->stream));
        build_ms[0] = now_ms() - b0;
        return TKHIP_OK;
    });
    wc.start(ctx, [&, ctx, s]() {
        TK_TRY(photon_pass_finish<1>(ctx, Bc));
        TK_TRY(install_local(ctx, 1));
        TK_HIP(hipStreamSynchronize(s->map[1]->stream));
        t_photon_end[1] = now_ms();
        if (s->stored_c > 0) {   // src/tracer/TkSppmTracer.cpp:477-478
            double b0 = now_ms();
            TK_TRY(map_build(s->map[1]));
            TK_HIP(hipStreamSynchronize(s->map[1]->stream));
            build_ms[1] = now_ms() - b0;
        }
        return TKHIP_OK;
    });
    int rc_cam = camera_trace(ctx, iter, s->quota[1] > 0);
    const double t_cam = now_ms();
    int rc_g = wg.join();
    int rc_c = wc.join();
    {
        // the maps are ready at photon_done, the gather queries at camera_done: the later of the two
        // sets the iteration, and wave slots move from the one to the other through photon_loads
        const double photon_done = std::max(t_photon_end[0] + build_ms[0], t_photon_end[1] + build_ms[1]) - t0;
        const double camera_done = t_cam - t0;
        // (not in the first two iterations, whose photon chains run late for reasons of their own)
        // Once the light-path grids are as large as they get (photon_loads 1: renders with millions of photons,
        // whose light paths are the longer chain whatever their grids), the camera path kernel gives up wave
        // slots instead, a sixteenth at a time, down to three sixteenths.
        // The further apart the two are, the larger the step (two from 10 % on, three from 18 %), so that a
        // render of the usual sixteen iterations spends most of them near the balance: one step moves either
        // chain by a per cent or two.
        const double ratio = photon_done / std::max(camera_done, 1e-3);
        const bool balanced = ratio <= 1.05 && ratio >= 0.95;
        if (iter >= 2 && balanced) s->steer_settled = true;   // from here on single steps: one late iteration must not undo the balance
        const int step = s->steer_settled ? 1 : (ratio > 1.18 || ratio < 1.0 / 1.18) ? 3 : ((ratio > 1.10 || ratio < 1.0 / 1.10) ? 2 : 1);
        if (iter < 2) {}
        else if (photon_done > 1.05 * camera_done) {
            for (int i = 0; i < step; ++i) {
                if (s->photon_loads > 1) --s->photon_loads;
                else s->camera_share = std::max(3, s->camera_share - 1);
            }
        } else if (photon_done < 0.95 * camera_done) {
            for (int i = 0; i < step; ++i) {
                if (s->camera_share < 16) ++s->camera_share;
                else s->photon_loads = std::min(8, s->photon_loads + 1);
            }
        }
        static const bool verbose = getenv("TKHIP_VERBOSE") != nullptr;
        if (verbose) fprintf(stderr, "tkhip: iteration %d: maps ready %.2f ms, queries ready %.2f ms -> photon_loads %d, camera share %d/16\n", iter, photon_done, camera_done, s->photon_loads, s->camera_share);
    }
    if (rc_g != TKHIP_OK) return rc_g;
    if (rc_c != TKHIP_OK) return rc_c;
    if (rc_cam != TKHIP_OK) return rc_cam;
    TK_TRY(gather_both(ctx, st));
    if (st) {
        for (int l = 0; l < 3; ++l) TK_TRY(collect_rays(ctx, l, st));
        st->ms_photon += (float)(std::max(t_photon_end[0], t_photon_end[1]) - t0);
        st->ms_build += (float)(build_ms[0] + build_ms[1]);
        st->ms_camera += (float)(t_cam - t0);
        st->ms_total += (float)(now_ms() - t0);
        fill_photon_stats(s, st);
    }
    return maybe_preview(ctx, iter, false);
}

int tkhip_sppm_set_preview(tkhip_ctx* ctx, int every_n) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_set_preview");
    TK_REQUIRE(every_n >= 0, TKHIP_E_ARG, "tkhip_sppm_set_preview: interval must be >= 0");
    SppmState* s = ctx->sppm;
    s->preview_every = every_n;
    if (every_n > 0 && !s->preview.p) {
        TK_TRY(s->preview.ensure((size_t)s->n_pixels * 4));
        TK_HIP(hipMemsetAsync(s->preview.p, 0, (size_t)s->n_pixels * 4, ctx->stream));
        s->preview_iter = -1;
    }
    return TKHIP_OK;
}

int tkhip_sppm_preview(tkhip_ctx* ctx, int refresh, const uint32_t** device_rgba, int* iteration) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_preview");
    SppmState* s = ctx->sppm;
    if (!s->preview.p) {
        TK_TRY(s->preview.ensure((size_t)s->n_pixels * 4));
        TK_HIP(hipMemsetAsync(s->preview.p, 0, (size_t)s->n_pixels * 4, ctx->stream));
        s->preview_iter = -1;
    }
    if (refresh) TK_TRY(maybe_preview(ctx, s->preview_iter < 0 ? 0 : s->preview_iter, true));
    TK_HIP(hipStreamSynchronize(ctx->stream));
    if (device_rgba) *device_rgba = s->preview.as<uint32_t>();
    if (iteration) *iteration = s->preview_iter;
    return TKHIP_OK;
}

int tkhip_sppm_read_preview(tkhip_ctx* ctx, uint32_t* host_rgba, int* iteration) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_read_preview");
    TK_REQUIRE(host_rgba, TKHIP_E_ARG, "tkhip_sppm_read_preview: null buffer");
    const uint32_t* dev = nullptr;
    TK_TRY(tkhip_sppm_preview(ctx, 0, &dev, iteration));
    TK_HIP(hipMemcpy(host_rgba, dev, (size_t)ctx->sppm->n_pixels * 4, hipMemcpyDeviceToHost));
    return TKHIP_OK;
}

int tkhip_sppm_gather_totals(tkhip_ctx* ctx, TkGatherTotals* out, int reset) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_gather_totals");
    SppmState* s = ctx->sppm;
    TkGatherTotals t{};
    for (int which = 0; which < 2; ++which) {
        tkhip_map* m = s->map[which];
        TK_HIP(hipStreamSynchronize(m->stream));
        TK_TRY(map_flush_timing(m));
        unsigned long long dev[16] = {0};
        TK_HIP(hipMemcpy(dev, m->totals.p, sizeof(dev), hipMemcpyDeviceToHost));
        if (getenv("TKHIP_DEBUG_STAMPS"))   // section cycle sums of a profiling build of the grouped gather kernel
            fprintf(stderr, "tkhip: map %d stamps: launches %lld wait-index %llu load %llu descent %llu steps %llu finish %llu\n", which,
                    (long long)m->group_launches, dev[4], dev[5], dev[6], dev[7], dev[8]);
        if (getenv("TKHIP_DEBUG_GW"))       // per-query kernel of a profiling build (TK_GW_STAMPS): clock ticks per query by the way it took
            fprintf(stderr, "tkhip: map %d per-query kernel: from group leaves max %llu (%llu leaves, %llu cuts) mean %.0f n %llu; descent max %llu (%llu leaves, %llu cuts) mean %.0f n %llu; "
                    "depth first max %llu mean %.0f n %llu\n",
                    which, dev[4] >> 20, (dev[4] >> 10) & 1023, dev[4] & 1023, dev[10] ? (double)dev[7] / dev[10] : 0.0, dev[10],
                    dev[5] >> 20, (dev[5] >> 10) & 1023, dev[5] & 1023, dev[11] ? (double)dev[8] / dev[11] : 0.0, dev[11],
                    dev[6] >> 20, dev[12] ? (double)dev[9] / dev[12] : 0.0, dev[12]);
        if (getenv("TKHIP_DEBUG_GW") && dev[10])
            fprintf(stderr, "tkhip: map %d per-query kernel, from group leaves, mean ticks: leaf list %.0f, steps %.0f, last cut %.0f, sum + update %.0f\n",
                    which, (double)dev[13] / dev[10], (double)dev[14] / dev[10], (double)dev[15] / dev[10],
                    ((double)dev[7] - dev[13] - dev[14] - dev[15]) / dev[10]);
        if (getenv("TKHIP_DEBUG_TURNS"))    // phase-B turn counts of a profiling build (TK_GG_TURNS) of the same kernel
            fprintf(stderr, "tkhip: map %d turns: as-is %llu per-step %llu per-pair %llu per-group %llu steps %llu groups %llu accepted %llu\n",
                    which, dev[9], dev[10], dev[11], dev[12], dev[13], dev[14], dev[15]);
        t.launches += m->group_launches;
        t.ms_group_kernel += m->group_ms_total;
        t.queries += m->group_queries;
        t.photons_found += (int64_t)dev[0];
        t.photons_staged += (int64_t)dev[1];
        t.queries_handed_on += (int64_t)dev[2];
        t.photons_found_k[which] = (int64_t)dev[0];
        t.queries_handed_on_k[which] = (int64_t)dev[2];
        if (reset) {
            TK_HIP(hipMemset(m->totals.p, 0, 128));
            m->group_launches = 0; m->group_queries = 0; m->group_ms_total = 0.0;
        }
    }
    t.ms_group_side_by_side = s->group_union_ms_total;
    t.steps_side_by_side = s->group_union_steps;
    if (reset) { s->group_union_ms_total = 0.0; s->group_union_steps = 0; }
    if (out) *out = t;
    return TKHIP_OK;
}

int tkhip_sppm_read_pixels(tkhip_ctx* ctx, TkSPPMPixel* host_out) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_read_pixels");
    TK_REQUIRE(host_out, TKHIP_E_ARG, "tkhip_sppm_read_pixels: null buffer");
    SppmState* s = ctx->sppm;
    TK_TRY(ctx->scratch_a.ensure((size_t)s->n_pixels * 52));
    k_planes_to_aos<<<divup(s->n_pixels, 256), 256, 0, ctx->stream>>>(s->planes.as<float>(), s->n_pixels,
                                                                     ctx->scratch_a.as<float>());
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(host_out, ctx->scratch_a.p, (size_t)s->n_pixels * 52, hipMemcpyDeviceToHost, ctx->stream));
    TK_HIP(hipStreamSynchronize(ctx->stream));
    return TKHIP_OK;
}

int tkhip_sppm_export_planes(tkhip_ctx* ctx, float* device_dst) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_export_planes");
    TK_REQUIRE(device_dst, TKHIP_E_ARG, "tkhip_sppm_export_planes: null buffer");
    SppmState* s = ctx->sppm;
    TK_TRY(sppm_wait_camera_lane(ctx));   // a camera pass begun ahead is still adding direct light
    TK_HIP(hipMemcpyAsync(device_dst, s->planes.p, (size_t)s->n_pixels * 52, hipMemcpyDeviceToDevice, ctx->stream));
    TK_HIP(hipStreamSynchronize(ctx->stream));
    return TKHIP_OK;
}

int tkhip_sppm_import_planes(tkhip_ctx* ctx, const float* device_src) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_import_planes");
    TK_REQUIRE(device_src, TKHIP_E_ARG, "tkhip_sppm_import_planes: null buffer");
    SppmState* s = ctx->sppm;
    // the direct light a camera pass begun ahead has added since the planes were exported would be overwritten
    TK_REQUIRE(s->camera_in_flight < 0, TKHIP_E_STATE,
               "tkhip_sppm_import_planes: the camera paths of iteration %d are in flight; run that iteration first", s->camera_in_flight);
    TK_HIP(hipMemcpyAsync(s->planes.p, device_src, (size_t)s->n_pixels * 52, hipMemcpyDeviceToDevice, ctx->stream));
    TK_HIP(hipStreamSynchronize(ctx->stream));
    return TKHIP_OK;
}

int tkhip_sppm_resolve_rgb8(tkhip_ctx* ctx, int iterations, uint8_t* host_out) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_resolve_rgb8");
    TK_REQUIRE(host_out && iterations >= 1, TKHIP_E_ARG, "tkhip_sppm_resolve_rgb8: bad argument");
    SppmState* s = ctx->sppm;
    TK_TRY(ctx->scratch_a.ensure((size_t)s->n_pixels * 3));
    k_resolve<<<divup(s->n_pixels, 256), 256, 0, ctx->stream>>>(s->planes.as<float>(), s->n_pixels, ctx->cam.w, ctx->cam.h,
                                                               iterations, (float)s->shoot[0],
                                                               (float)s->shoot[1], (int)s->stored_c,
                                                               ctx->scratch_a.as<uint8_t>());
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(host_out, ctx->scratch_a.p, (size_t)s->n_pixels * 3, hipMemcpyDeviceToHost, ctx->stream));
    TK_HIP(hipStreamSynchronize(ctx->stream));
    return TKHIP_OK;
}

int tkhip_sppm_counters(tkhip_ctx* ctx, int64_t* shoot_global, int64_t* shoot_caustic, int32_t* num_recurse) {
    TK_REQUIRE(ctx && ctx->sppm, TKHIP_E_STATE, "tkhip_sppm_counters: call tkhip_sppm_begin first");
    if (shoot_global) *shoot_global = ctx->sppm->shoot[0];
    if (shoot_caustic) *shoot_caustic = ctx->sppm->shoot[1];
    if (num_recurse) *num_recurse = ctx->sppm->num_recurse;
    return TKHIP_OK;
}

int tkhip_sppm_read_photons(tkhip_ctx* ctx, int which, float* pos, float* power, uint8_t* theta, uint8_t* phi,
                            int64_t* n) {
    TK_CTX_SPPM(ctx, "tkhip_sppm_read_photons");
    TK_REQUIRE(which == 0 || which == 1, TKHIP_E_ARG, "tkhip_sppm_read_photons: which must be 0 or 1");
    SppmState* s = ctx->sppm;
    tkhip_map* m = s->map[which];
    int64_t cnt = m->n;
    if (n) *n = cnt;
    if (cnt == 0) return TKHIP_OK;
    hipStream_t st = ctx->stream;
    if (pos) TK_HIP(hipMemcpyAsync(pos, m->in_pos.p, (size_t)cnt * 12, hipMemcpyDeviceToHost, st));
    if (power) TK_HIP(hipMemcpyAsync(power, m->in_pow.p, (size_t)cnt * 12, hipMemcpyDeviceToHost, st));
    if (theta || phi) {
        TK_TRY(ctx->scratch_b.ensure((size_t)cnt * 2));
        uint8_t* d = ctx->scratch_b.as<uint8_t>();
        k_unpack_dir<<<divup(cnt, 256), 256, 0, st>>>(m->in_dir.as<uint16_t>(), d, d + cnt, cnt);
        TK_LAUNCH_CHECK();
        if (theta) TK_HIP(hipMemcpyAsync(theta, d, (size_t)cnt, hipMemcpyDeviceToHost, st));
        if (phi) TK_HIP(hipMemcpyAsync(phi, d + cnt, (size_t)cnt, hipMemcpyDeviceToHost, st));
    }
    TK_HIP(hipStreamSynchronize(st));
    return TKHIP_OK;
}

/* ---- hash-grid variant ---- */
int tkhip_hsppm_begin(tkhip_ctx* ctx, const TkSppmParam* param, int spp, int max_depth, float rr, uint64_t seed) {
    TK_REQUIRE(ctx && param, TKHIP_E_ARG, "tkhip_hsppm_begin: null argument");
    TK_REQUIRE(spp >= 1 && (spp & (spp - 1)) == 0, TKHIP_E_ARG, "tkhip_hsppm_begin: spp must be a power of two (RayTracer rounds it up)");
    TK_REQUIRE(param->radius2 > 0.f, TKHIP_E_ARG, "tkhip_hsppm_begin: the initial radius must be positive");
    TkSppmParam p = *param;
    p.causticSize = 0;                 // one photon budget per iteration (photonsPerIt = globalSize)
    TK_TRY(tkhip_sppm_begin(ctx, &p, max_depth, rr, seed, 0, 1));
    SppmState* s = ctx->sppm;
    s->hashgrid = true;
    s->hg_spp = spp;
    TK_TRY(s->hg_prm.ensure(sizeof(HashGridParams)));
    TK_TRY(s->hg_bounds.ensure(8 * 4));
    TK_TRY(s->hg_count.ensure((size_t)(s->n_pixels + 1) * 4));
    TK_TRY(s->hg_start.ensure((size_t)(s->n_pixels + 1) * 4));
    return TKHIP_OK;
}

int tkhip_hsppm_iteration(tkhip_ctx* ctx, int iter, TkIterStats* st) {
    TK_CTX_SPPM(ctx, "tkhip_hsppm_iteration");
    SppmState* s = ctx->sppm;
    TK_REQUIRE(s->hashgrid, TKHIP_E_STATE, "tkhip_hsppm_iteration: the render was not begun with tkhip_hsppm_begin");
    const double t0 = now_ms();
    if (st) memset(st, 0, sizeof(*st));
    // ray pass: the visible point of every pixel (src/TkSPPM.cpp:136-179)
    TK_TRY(camera_trace(ctx, iter, true));
    if (st) {
        TK_TRY(collect_rays(ctx, 2, st));
        st->ms_camera = (float)(now_ms() - t0);
    }
    const int64_t nv = s->n_queries[1];
    const uint32_t buckets = (uint32_t)s->n_pixels;
    hipStream_t stream = ctx->stream;
    const float4* q = s->qbuf[s->gather_set][1].as<float4>();
    const float4 *vp0 = q, *vp1 = q + s->n_slots, *vp2 = q + 2 * s->n_slots, *vp3 = q + 3 * s->n_slots;
    // grid rebuild (src/TkSPPM.cpp:45-86)
    const double t1 = now_ms();
    const uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    TK_HIP(hipMemcpyAsync(s->hg_bounds.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    TK_HIP(hipMemsetAsync(s->hg_count.p, 0, (size_t)(buckets + 1) * 4, stream));
    if (nv > 0) {
        k_hg_bounds<<<std::min<int64_t>(divup(nv, 256), 1024), 256, 0, stream>>>(vp0, nv, s->hg_bounds.as<uint32_t>());
        TK_LAUNCH_CHECK();
    }
    k_hg_setup<<<1, 1, 0, stream>>>(s->hg_bounds.as<uint32_t>(), nv, s->hg_prm.as<HashGridParams>());
    TK_LAUNCH_CHECK();
    int32_t total = 0;
    if (nv > 0) {
        k_hg_insert<false><<<divup(nv, 256), 256, 0, stream>>>(vp0, nv, s->hg_prm.as<HashGridParams>(), buckets,
                                                               s->hg_count.as<int32_t>(), nullptr, nullptr);
        TK_LAUNCH_CHECK();
    }
    TK_TRY(exclusive_scan_i32(stream, ctx->prim_temp, s->hg_count.as<int32_t>(), s->hg_start.as<int32_t>(), (size_t)buckets + 1));
    TK_HIP(hipMemcpyAsync(&total, s->hg_start.as<int32_t>() + buckets, 4, hipMemcpyDeviceToHost, stream));
    TK_HIP(hipStreamSynchronize(stream));
    TK_TRY(s->hg_items.ensure((size_t)std::max(total, 1) * 4));
    TK_TRY(s->hg_phi.ensure((size_t)std::max<int64_t>(nv, 1) * 3 * 4));
    TK_TRY(s->hg_m.ensure((size_t)std::max<int64_t>(nv, 1) * 4));
    TK_HIP(hipMemsetAsync(s->hg_phi.p, 0, (size_t)std::max<int64_t>(nv, 1) * 3 * 4, stream));
    TK_HIP(hipMemsetAsync(s->hg_m.p, 0, (size_t)std::max<int64_t>(nv, 1) * 4, stream));
    if (nv > 0) {
        TK_HIP(hipMemsetAsync(s->hg_count.p, 0, (size_t)(buckets + 1) * 4, stream));
        k_hg_insert<true><<<divup(nv, 256), 256, 0, stream>>>(vp0, nv, s->hg_prm.as<HashGridParams>(), buckets,
                                                              s->hg_count.as<int32_t>(), s->hg_start.as<int32_t>(),
                                                              s->hg_items.as<int32_t>());
        TK_LAUNCH_CHECK();
    }
    TK_HIP(hipStreamSynchronize(stream));
    if (st) st->ms_build = (float)(now_ms() - t1);
    // photon pass (src/TkSPPM.cpp:181-250): photonsPerIt light paths, Halton index iter * photonsPerIt + i
    const double t2 = now_ms();
    Lane& ln = s->lane[0];
    const int64_t per_it = s->prm.globalSize;
    int32_t* counters = ln.counters.as<int32_t>();
    TK_HIP(hipMemsetAsync(&counters[C_RAYS], 0, 56, ln.stream));
    for (int64_t done = 0; done < per_it;) {
        const int64_t B = std::min<int64_t>(per_it - done, (int64_t)1 << 22);
        PhotonArgs a{};
        a.sc = ctx->scene;
        TK_TRY(ln.q_o.ensure((size_t)B * 16));
        TK_TRY(ln.q_d.ensure((size_t)B * 16));
        TK_TRY(ln.q_w.ensure((size_t)B * 16));
        a.q = PathQueue{ln.q_o.as<float4>(), ln.q_d.as<float4>(), ln.q_w.as<float4>(), (int)B};
        a.hg = HashGridView{s->hg_prm.as<HashGridParams>(), s->hg_start.as<int32_t>(), s->hg_items.as<int32_t>(), vp0, vp1, vp3,
                            s->hg_phi.as<float>(), s->hg_m.as<int32_t>(), nv, buckets};
        a.m0 = done;
        a.hoff = (int64_t)iter * per_it;
        a.world = 1; a.rank = 0; a.seed = s->seed; a.rr = s->rr; a.slots = 1; a.depth = s->max_depth;
        TK_HIP(hipMemsetAsync(counters, 0, kCounterBatchInts * 4, ln.stream));
        k_photon_first<2><<<divup(B, kTraceBlock), kTraceBlock, 0, ln.stream>>>(a, B, counters);
        const int waves = (int)std::min<int64_t>(divup(B, kPathBlock), std::max(s->path_waves[0], 64));
        k_photon_paths<2><<<waves, kPathBlock, 0, ln.stream>>>(a, counters);
        TK_LAUNCH_CHECK();
        done += B;
    }
    TK_HIP(hipStreamSynchronize(ln.stream));
    s->shoot[0] += per_it;
    if (st) {
        TK_TRY(collect_rays(ctx, 0, st));
        st->ms_photon = (float)(now_ms() - t2);
        st->global_paths = per_it;
        st->shoot_global = s->shoot[0];
        st->gathers_caustic = nv;
    }
    // renew pass (src/TkSPPM.cpp:252-272)
    if (nv > 0) {
        k_hg_update<<<divup(nv, 256), 256, 0, stream>>>(vp2, s->hg_phi.as<float>(), s->hg_m.as<int32_t>(), nv, s->planes.as<float>(),
                                                        s->n_pixels);
        TK_LAUNCH_CHECK();
    }
    TK_HIP(hipStreamSynchronize(stream));
    if (st) st->ms_total = (float)(now_ms() - t0);
    return TKHIP_OK;
}

int tkhip_hsppm_resolve_rgb8(tkhip_ctx* ctx, uint8_t* host_out) {
    TK_CTX_SPPM(ctx, "tkhip_hsppm_resolve_rgb8");
    SppmState* s = ctx->sppm;
    TK_REQUIRE(s->hashgrid && host_out, TKHIP_E_STATE, "tkhip_hsppm_resolve_rgb8: no hash-grid render, or null buffer");
    const size_t bytes = (size_t)s->n_pixels * 3;
    TK_TRY(ctx->scratch_a.ensure(bytes));
    k_hg_resolve<<<divup(s->n_pixels, 256), 256, 0, ctx->stream>>>(s->planes.as<float>(), s->n_pixels, ctx->cam.w, ctx->cam.h,
                                                                  s->num_recurse, s->prm.globalSize, ctx->scratch_a.as<uint8_t>());
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(host_out, ctx->scratch_a.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TK_HIP(hipStreamSynchronize(ctx->stream));
    return TKHIP_OK;
}

/* ---- path tracer ---- */
int tkhip_pt_begin(tkhip_ctx* ctx, int spp, int max_depth, float rr, uint64_t seed) {
    TK_REQUIRE(ctx, TKHIP_E_ARG, "tkhip_pt_begin: null context");
    TK_REQUIRE(spp >= 1 && (spp & (spp - 1)) == 0, TKHIP_E_ARG, "tkhip_pt_begin: spp must be a power of two (RayTracer rounds it up)");
    TkSppmParam p{};
    p.numRecurse = 1; p.globalSize = 1; p.globalSample = 1; p.causticSize = 0; p.causticSample = 1; p.radius2 = 1.f;
    TK_TRY(tkhip_sppm_begin(ctx, &p, max_depth, rr, seed, 0, 1));
    SppmState* s = ctx->sppm;
    s->pt_mode = true;
    s->pt_next_sample = 0;
    // strata of the pixel sampler (PathTracer ctor, src/tracer/TkPathTracer.cpp:16-22)
    int x = 1, y = spp / x;
    while (y != x && y / x != 2) { x <<= 1; y = spp / x; }
    s->xs = x; s->ys = y; s->num_recurse = spp;
    // the film starts empty: contribSum and filterWeightSum zero (Film::clear)
    TK_HIP(hipMemsetAsync(s->planes.p, 0, (size_t)s->n_pixels * 13 * 4, ctx->stream));
    TK_HIP(hipStreamSynchronize(ctx->stream));
    return TKHIP_OK;
}

int tkhip_pt_render_samples(tkhip_ctx* ctx, int count, TkIterStats* st) {
    TK_CTX_SPPM(ctx, "tkhip_pt_render_samples");
    SppmState* s = ctx->sppm;
    TK_REQUIRE(s->pt_mode, TKHIP_E_STATE, "tkhip_pt_render_samples: the render was not begun with tkhip_pt_begin");
    TK_REQUIRE(count >= 0 && s->pt_next_sample + count <= s->num_recurse, TKHIP_E_ARG,
               "tkhip_pt_render_samples: %d samples asked, %d of %d left", count, s->num_recurse - s->pt_next_sample, s->num_recurse);
    const double t0 = now_ms();
    if (st) memset(st, 0, sizeof(*st));
    for (int i = 0; i < count; ++i) {
        TK_TRY(camera_trace(ctx, s->pt_next_sample, false));
        if (st) TK_TRY(collect_rays(ctx, 2, st));
        ++s->pt_next_sample;
    }
    if (st) st->ms_camera = st->ms_total = (float)(now_ms() - t0);
    return TKHIP_OK;
}

int tkhip_pt_resolve_rgb8(tkhip_ctx* ctx, uint8_t* host_out) {
    TK_CTX_SPPM(ctx, "tkhip_pt_resolve_rgb8");
    SppmState* s = ctx->sppm;
    TK_REQUIRE(s->pt_mode && host_out, TKHIP_E_STATE, "tkhip_pt_resolve_rgb8: no path-tracer render, or null buffer");
    const size_t bytes = (size_t)s->n_pixels * 3;
    TK_TRY(ctx->scratch_a.ensure(bytes));
    k_pt_resolve<<<divup(s->n_pixels, 256), 256, 0, ctx->stream>>>(s->planes.as<float>(), s->n_pixels, ctx->cam.w, ctx->cam.h,
                                                                  ctx->scratch_a.as<uint8_t>());
    TK_LAUNCH_CHECK();
    TK_HIP(hipMemcpyAsync(host_out, ctx->scratch_a.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TK_HIP(hipStreamSynchronize(ctx->stream));
    return TKHIP_OK;
}

}  // extern "C"
