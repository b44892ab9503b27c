// photon_map.h — device photon map: storage, search tree, kNN gather.
#pragma once
#include "tk_common.h"

namespace tkd {

constexpr int kLeafCap = 32;   // photons per tree leaf (at most; at least half)

// Queries of one gather launch, struct-of-arrays on the device.
struct QueryView {
    const float4* q0;   // p.xyz, search radius^2
    const float4* q1;   // n.xyz, dot(wo, n)
    const float4* q2;   // path throughput rgb, pixel index (int bits)
    const float4* q3;   // BRDF value R/pi rgb, unused
};

// Where results go.  Test outputs are indexed by query; the render output is
// the fused per-pixel flux / photon count / radius update.
struct GatherOut {
    float* flux;        // 3 per query or null
    int32_t* found;     // or null
    float* r2out;       // or null
    int32_t* idx;       // k per query or null
    float* planes;      // SPPMPixel planes (13 x n_pixels) or null
    int64_t n_pixels;
    int32_t which;      // 0 global, 1 caustic: selects the planes to update
    unsigned long long* photon_sum;   // optional: sum of found over queries
    unsigned long long* cand_sum;     // optional: photons staged by the grouped kernel, summed over waves (x64 = tests)
    unsigned long long* overflow_sum; // optional: queries handed to the per-query kernel
};

// Device scratch of one gather of up to nq queries: sort keys and values of the query order, the
// hand-over list between the two kernels, and the sort's temporary storage.
struct GatherScratch {
    uint32_t *q_keys_a, *q_keys_b, *q_vals_a, *q_order;
    float4* q_sorted;            // 2 x nq: q0 and q1 of the queries in the order of the curve
    int32_t *overflow, *overflow_group, *group_leaves, *group_leaf_count;
    int32_t *sel_groups, *sel_cnt;   // groups left to the selection kernel; per query slot: photons met
    float* q_r2s;                    // per query slot: the radius^2 to search with (the query's own or a smaller bound)
    int heavy;                       // many gathers of the map's last launch met k photons (or no launch yet): radii are cut first
    int track;                       // count the gathers of this launch that meet k photons (counter[24]) for the next decision
    int untimed;                     // no events around the grouped kernel and nothing added to the map's running totals
    int32_t* counter;            // 32 ints, cleared per gather: [0] the hand-over count, [16] the selection kernel's groups, [24] gathers that met k photons
    void* sort_temp;
    size_t sort_temp_bytes;
};

struct MapView {
    const float4* pos4;      // leaf order: x, y, z, original index (int bits)
    const float4* pow4;      // leaf order: r, g, b, theta | phi << 8 (int bits)
    const float4* node_box;  // heap order, 2 float4 per node: (mnx,mny,mnz,mxx) (mxy,mxz,-,-)
    const float* dirtab;     // cosTheta[256], sinTheta[256], cosPhi[256], sinPhi[256]
    const uint32_t* keys;    // leaf order: the photons' cells along the Hilbert curve (ascending), or null (median-split tree)
    int32_t n, nl, depth, k;
};

}  // namespace tkd

struct tkhip_map {
    tkhip_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;   // where this map's copies, build and gathers are enqueued
    tkd::DevBuf prim_temp;          // sort scratch of that stream
    int64_t cap = 0, n = 0;
    int k = 0;
    bool built = false;
    int nl = 0, depth = 0;
    tkd::DevBuf in_pos, in_pow, in_dir;            // storage order: float3, float3, u16
    tkd::DevBuf pos4, pow4, node_box, node_axis, clip_box;
    int32_t* met_k_host = nullptr;  // pinned: gathers of the last launch that met k photons or more
    int64_t met_k_queries = 0;      // ... of this many
    bool met_k_pending = false;
    hipEvent_t ev_met_k = nullptr;  // recorded behind the copy into met_k_host
    int tree_mode = 0;              // 0: one sort along a Hilbert curve; 1: median splits, one sort per level
    tkd::DevBuf idx_a, idx_b, key_a, key_b, root_box;
    tkd::DevBuf q_order, q_keys_a, q_keys_b, q_vals_a, q_sorted, overflow, counters;
    tkd::DevBuf overflow_group, group_leaves, group_leaf_count;   // what the grouped kernel hands on with a query
    tkd::DevBuf sel_groups, sel_cnt, q_r2s;                      // ... and leaves for the selection kernel
    float last_group_ms = 0.f;   // duration of the last grouped gather kernel (timed gathers only)
    int64_t last_overflow = 0;   // queries the last timed gather handed to the per-query kernel
    // Every grouped-kernel launch is bracketed by this pair of events on the map's stream (recording
    // costs nothing worth naming); the pair of the launch before is read back when the next one is
    // recorded or when the totals are asked for.
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool ev_pending = false;
    double group_ms_total = 0.0;
    int64_t group_launches = 0, group_queries = 0;
    tkd::DevBuf totals;          // 3 x u64 on the device: photons found, photons staged, queries handed on
    tkd::MapView view() const;
};

namespace tkd {

int map_create(tkhip_ctx* ctx, int64_t capacity, int k, tkhip_map** out);
void map_destroy(tkhip_map* m);
// Device pointers, storage order.  Asynchronous copy on the context stream.
int map_set_photons_device(tkhip_map* m, const float* pos, const float* power, const uint16_t* dir, int64_t n);
int map_build(tkhip_map* m);
// nq queries on the device -> out.  mode 0: grouped kernel then per-query
// kernel for the overflow; mode 1: per-query kernel only.  timed: record the
// grouped kernel's duration with events (synchronises).
// Sizes the scratch of gathers of up to nq queries (so that later gathers do not reallocate).
int map_reserve_queries(tkhip_map* m, int64_t nq);
// Sizes what map_build needs for a full map.
int map_reserve_build(tkhip_map* m);
// Adds the pending event pair to the running totals (waits for the launch it brackets).
int map_flush_timing(tkhip_map* m);
int map_gather_device(tkhip_map* m, int64_t nq, const QueryView& qv, const GatherOut& out, int mode, bool timed);
// The same with the scratch given by the caller (nothing is allocated).
int map_gather_in(tkhip_map* m, int64_t nq, const QueryView& qv, const GatherOut& out, int mode, bool timed,
                  const GatherScratch& sc);
// Scratch of a production gather from the map's own buffers (grown on demand).
int map_gather_scratch(tkhip_map* m, int64_t nq, GatherScratch* out_sc);
// Production gathers of two maps (both with queries) whose grouped kernel is one launch; the launch is timed
// under m0.  Work is left enqueued on the two maps' streams.
int map_gather_pair(tkhip_map* m0, int64_t nq0, const QueryView& qv0, const GatherOut& out0, tkhip_map* m1, int64_t nq1,
                    const QueryView& qv1, const GatherOut& out1, bool timed);

}  // namespace tkd
