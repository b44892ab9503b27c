// photon_map.hip — the photon map on the device.
//
//  * storage  : PhotonMap::storePhoton (src/tracer/Photonmap.cpp:44-53)
//  * build    : the search structure PhotonMap::buildKdTree / balance make
//               (src/tracer/Photonmap.cpp:96-103, :132-179), here a balanced
//               binary tree over the photons in Hilbert-curve order: one radix
//               sort, leaves are equal runs of at most kLeafCap consecutive
//               photons, the tree above them is implicit (complete binary
//               tree, photon ranges by arithmetic) and every node carries the
//               tight box of its photons.  A gather's result does not depend
//               on the shape of the tree.
//  * gather   : neighbor_search + radiance_estimate (src/tracer/neighbor.cpp:5-85,
//               src/tracer/Photonmap.cpp:105-130) with the fused per-pixel update of
//               traceTile (src/tracer/TkSppmTracer.cpp:327-340, :346-359)
//
// Result contract of a gather (what the CPU code computes, stated without its
// traversal order): among the photons with dist2 < radius2 (strict) keep the k
// smallest by (dist2, photon index); found = how many were kept; the returned
// radius2 is the largest kept dist2 when k were kept and the input radius2
// otherwise; flux = sum over the kept photons of power * |wi.n| * f.
// dist2 is ((dx*dx + dy*dy) + dz*dz) in fp32 with separate roundings.
#include "photon_map.h"

#include <algorithm>
#include <mutex>

#include "tk_device.h"

namespace tkd {

// ------------------------------------------------------------ small helpers
TKD uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
TKD float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// First photon of leaf j: the n photons are dealt evenly to the 2^depth leaves.  A shift, not a
// division: the gather kernels evaluate this for every leaf they touch.
TKD int64_t leaf_start(int64_t j, int64_t n, int depth) { return (j * n) >> depth; }

TKD float dist2_exact(float px, float py, float pz, F3 q) {
    float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
    return dx * dx + dy * dy + dz * dz;
}

TKD float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
TKD float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
TKD float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
TKD unsigned long long wave_max_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}
// Wave-wide reductions through DPP row shifts and row broadcasts (no LDS crossbar round trips):
// lane 63 ends up with the result, which is returned wave-uniform.
template <int CTRL, int ROW_MASK>
TKD float dpp_keep(float v) {   // lanes without a source lane, or in a masked row, keep their own value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
TKD float dpp_zero(float v) {   // ... read zero
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
TKD float wave_min_all(float v) {
    v = fminf(v, dpp_keep<0x111, 0xf>(v)); v = fminf(v, dpp_keep<0x112, 0xf>(v));
    v = fminf(v, dpp_keep<0x114, 0xf>(v)); v = fminf(v, dpp_keep<0x118, 0xf>(v));     // lane 15 of a row: the row
    v = fminf(v, dpp_keep<0x142, 0xa>(v));                                              // row_bcast:15 into rows 1, 3
    v = fminf(v, dpp_keep<0x143, 0xc>(v));                                              // row_bcast:31 into rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
TKD float wave_max_all(float v) {
    v = fmaxf(v, dpp_keep<0x111, 0xf>(v)); v = fmaxf(v, dpp_keep<0x112, 0xf>(v));
    v = fmaxf(v, dpp_keep<0x114, 0xf>(v)); v = fmaxf(v, dpp_keep<0x118, 0xf>(v));
    v = fmaxf(v, dpp_keep<0x142, 0xa>(v));
    v = fmaxf(v, dpp_keep<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
TKD float wave_sum_all(float v) {
    v += dpp_zero<0x111, 0xf>(v); v += dpp_zero<0x112, 0xf>(v);
    v += dpp_zero<0x114, 0xf>(v); v += dpp_zero<0x118, 0xf>(v);                        // inclusive scan inside each row
    v += dpp_zero<0x142, 0xa>(v);
    v += dpp_zero<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
TKD float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
TKD int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
TKD float lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// ---------------------------------------------------------------- build
// Bounds of the photons: block-wide min / max, then one atomic per block and
// component on order-preserving integer images of the floats.
__global__ void k_bounds_init(uint32_t* box6) {
    if (threadIdx.x < 3) box6[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) box6[threadIdx.x] = 0u;
}
__global__ void __launch_bounds__(256) k_bounds(const float* pos, int64_t n, uint32_t* box6) {
    __shared__ float s_red[6][4];
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            float x = pos[3 * i + a];
            v[a] = fminf(v[a], x);
            v[3 + a] = fmaxf(v[3 + a], x);
        }
    for (int a = 0; a < 3; ++a) { v[a] = wave_min(v[a]); v[3 + a] = wave_max(v[3 + a]); }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 6; ++a) s_red[a][threadIdx.x >> 6] = v[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float r = s_red[a][0];
        for (int w = 1; w < 4; ++w) r = a < 3 ? fminf(r, s_red[a][w]) : fmaxf(r, s_red[a][w]);
        if (a < 3) atomicMin(&box6[a], f2ord(r));
        else atomicMax(&box6[a], f2ord(r));
    }
}

// Position of a query along a 3-D Hilbert curve through the photon bounds
// (10 bits per axis, Skilling's transpose form).  Unlike a Morton curve a
// Hilbert curve has no jumps: consecutive queries in this order are always
// neighbours in space, so the search boxes of the 64 queries of a wave overlap.
TKD uint32_t hilbert30(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t X[3] = {x, y, z};
    const uint32_t M = 1u << 9;
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        uint32_t P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    uint32_t key = 0;
#pragma unroll
    for (int b = 9; b >= 0; --b) {
        key = (key << 3) | (((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u);
    }
    return key;
}
// Hilbert key of every photon inside the photon bounds (10 bits per axis).
__global__ void k_photon_curve(const float* pos, int64_t n, const uint32_t* box6, uint32_t* keys, uint32_t* vals) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float q[3];
    for (int a = 0; a < 3; ++a) {
        float lo = ord2f(box6[a]), hi = ord2f(box6[3 + a]);
        float e = hi - lo;
        float inv = e > 0 ? 1023.f / e : 0.f;
        q[a] = fminf(fmaxf((pos[3 * i + a] - lo) * inv, 0.f), 1023.f);
    }
    keys[i] = hilbert30((uint32_t)q[0], (uint32_t)q[1], (uint32_t)q[2]);
    vals[i] = (uint32_t)i;
}

// ---- the median-split build (PhotonMap::balance, src/tracer/Photonmap.cpp:132-179, level by level):
// every level sorts the photons by (node, coordinate on the node's split axis); a node's photons are
// then contiguous and its median is the first photon of its right half.  The split axis is the
// widest axis of the node's box, the parent's box clipped at the split planes above
// (Bounds3::maxExtent, src/Bounds3.hpp:32-41; :149-152, :167-170).
__global__ void k_iota(uint32_t* a, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
// clip_box: 6 floats per heap node (min xyz, max xyz); node 1 is the root.
__global__ void k_root_clip(const uint32_t* box6, float* clip_box) {
    if (threadIdx.x < 6) clip_box[6 + threadIdx.x] = ord2f(box6[threadIdx.x]);
}
__global__ void k_level_axis(const float* clip_box, int32_t* node_axis, int level) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int count = 1 << level;
    if (j >= count) return;
    int node = count + j;
    const float* b = clip_box + 6 * (int64_t)node;
    float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    node_axis[node] = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);
}
__global__ void k_level_keys(const float* pos, const uint32_t* idx, const int32_t* node_axis, uint64_t* keys,
                             int64_t n, int nl, int depth, int level) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int leaf = (int)(((i + 1) * nl - 1) / n);   // the leaf position i falls into: inverse of leaf_start
    const int j = leaf >> (depth - level);
    const int axis = node_axis[(1 << level) + j];
    const float v = pos[3 * (int64_t)idx[i] + axis];
    keys[i] = ((uint64_t)(uint32_t)j << 32) | f2ord(v);
}
__global__ void k_level_split(const float* pos, const uint32_t* idx, float* clip_box, const int32_t* node_axis,
                              int64_t n, int depth, int level) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int count = 1 << level;
    if (j >= count) return;
    int node = count + j;
    int axis = node_axis[node];
    int mid_leaf = (2 * j + 1) << (depth - level - 1);
    int64_t mid = leaf_start(mid_leaf, n, depth);
    float split = pos[3 * (int64_t)idx[mid] + axis];
    const float* b = clip_box + 6 * (int64_t)node;
    float* l = clip_box + 6 * (int64_t)(2 * node);
    float* r = clip_box + 6 * (int64_t)(2 * node + 1);
    for (int a = 0; a < 6; ++a) { l[a] = b[a]; r[a] = b[a]; }
    l[3 + axis] = split;
    r[axis] = split;
}

__global__ void k_permute(const float* pos, const float* pow, const uint16_t* dir, const uint32_t* idx,
                          float4* pos4, float4* pow4, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = idx[i];
    pos4[i] = make_float4(pos[3 * (int64_t)s], pos[3 * (int64_t)s + 1], pos[3 * (int64_t)s + 2], __uint_as_float(s));
    pow4[i] = make_float4(pow[3 * (int64_t)s], pow[3 * (int64_t)s + 1], pow[3 * (int64_t)s + 2],
                          __uint_as_float((uint32_t)dir[s]));
}

// Tight box of every leaf: kLeafCap lanes per leaf, one photon per lane (coalesced), then a
// butterfly reduction inside the kLeafCap lanes.
static_assert(kLeafCap == 32, "k_leaf_boxes reduces over 32-lane halves of a wave");
__global__ void __launch_bounds__(256) k_leaf_boxes(const float4* pos4, float4* node_box, int64_t n, int nl, int depth) {
    const int64_t j = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kLeafCap;
    const int ph = threadIdx.x % kLeafCap;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (j < nl) {
        const int64_t s = leaf_start(j, n, depth), e = leaf_start(j + 1, n, depth);
        if (s + ph < e) {
            const float4 p = pos4[s + ph];
            mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z;
        }
    }
    for (int o = kLeafCap / 2; o > 0; o >>= 1)
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    if (j < nl && ph == 0) {
        const int64_t node = nl + j;
        node_box[2 * node] = make_float4(mn[0], mn[1], mn[2], mx[0]);
        node_box[2 * node + 1] = make_float4(mx[1], mx[2], 0.f, 0.f);
    }
}
// Boxes of the inner nodes, level by level from the leaves' parents to the root.  A level of
// more than kWideLevel nodes gets a launch of its own; the narrow levels above are done by one
// block, which needs no launch per level.
constexpr int kWideLevel = 4096;
TKD void inner_box(float4* node_box, int64_t node) {
    float4 a0 = node_box[4 * node], a1 = node_box[4 * node + 1];
    float4 b0 = node_box[4 * node + 2], b1 = node_box[4 * node + 3];
    node_box[2 * node] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), fmaxf(a0.w, b0.w));
    node_box[2 * node + 1] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), 0.f, 0.f);
}
__global__ void __launch_bounds__(256) k_inner_level(float4* node_box, int level) {
    const int count = 1 << level;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) inner_box(node_box, (int64_t)count + j);
}
__global__ void __launch_bounds__(1024) k_inner_boxes(float4* node_box, int top_level) {
    for (int level = top_level; level >= 0; --level) {
        const int count = 1 << level;
        for (int j = threadIdx.x; j < count; j += blockDim.x) inner_box(node_box, (int64_t)count + j);
        __syncthreads();
    }
}

// --------------------------------------------------------------- queries
struct QLoad {
    F3 p, n, coef, f;
    float r2, won;
    int pixel;
};
TKD QLoad load_query(const QueryView& qv, int64_t q) {
    float4 a = qv.q0[q], b = qv.q1[q], c = qv.q2[q], d = qv.q3[q];
    QLoad Q;
    Q.p = mk3(a.x, a.y, a.z); Q.r2 = a.w;
    Q.n = mk3(b.x, b.y, b.z); Q.won = b.w;
    Q.coef = mk3(c.x, c.y, c.z); Q.pixel = __float_as_int(c.w);
    Q.f = mk3(d.x, d.y, d.z);
    return Q;
}

// The flux / photon count / radius rule of traceTile
// (src/tracer/TkSppmTracer.cpp:327-340 caustic, :346-359 global).
TKD void pixel_update(float* fr, float* fg, float* fb, float* photons, float* radius2, F3 lindir, int found,
                      float r2ret) {
    if (*photons == 0) {
        *radius2 = r2ret;
        *photons = (float)found;
        *fr = lindir.x; *fg = lindir.y; *fb = lindir.z;
    } else {
        float new_photons = (float)((double)*photons + 0.7 * (double)found);
        float new_radius2 = *radius2 * new_photons / (*photons + (float)found);
        *fr = (*fr + lindir.x) * new_radius2 / *radius2;
        *fg = (*fg + lindir.y) * new_radius2 / *radius2;
        *fb = (*fb + lindir.z) * new_radius2 / *radius2;
        *photons = new_photons;
        *radius2 = new_radius2;
    }
}

TKD void finish_query(const GatherOut& out, int64_t q, const QLoad& Q, int found, float r2ret, F3 S) {
    F3 est = mul3(S, Q.f);
    if (out.flux) { out.flux[3 * q] = est.x; out.flux[3 * q + 1] = est.y; out.flux[3 * q + 2] = est.z; }
    if (out.found) out.found[q] = found;
    if (out.r2out) out.r2out[q] = r2ret;
    if (out.planes) {
        F3 lind = mul3(Q.coef, est);
        const int64_t np = out.n_pixels, px = Q.pixel;
        const int fl = out.which == 0 ? 6 : 3, pn = out.which == 0 ? 10 : 9, pr = out.which == 0 ? 12 : 11;
        float* P = out.planes;
        float r = P[fl * np + px], g = P[(fl + 1) * np + px], b = P[(fl + 2) * np + px];
        float photons = P[pn * np + px], radius2 = P[pr * np + px];
        pixel_update(&r, &g, &b, &photons, &radius2, lind, found, r2ret);
        P[fl * np + px] = r; P[(fl + 1) * np + px] = g; P[(fl + 2) * np + px] = b;
        P[pn * np + px] = photons; P[pr * np + px] = radius2;
    }
}

// wi decoded from the two direction bytes (src/tracer/Photonmap.cpp:122-123)
TKD F3 decode_dir(const float* tab, uint32_t bits) {
    uint32_t th = bits & 255u, ph = (bits >> 8) & 255u;
    float st = tab[256 + th];
    return mk3(st * tab[512 + ph], st * tab[768 + ph], tab[th]);
}

// Conservative squared distance from a point to a node box (never above the
// distance of any photon inside the box).
TKD float box_dist2(const MapView& mv, int node, F3 p) {
    float4 b0 = mv.node_box[2 * (int64_t)node], b1 = mv.node_box[2 * (int64_t)node + 1];
    float dx = fmaxf(fmaxf(b0.x - p.x, p.x - b0.w), 0.f);
    float dy = fmaxf(fmaxf(b0.y - p.y, p.y - b1.x), 0.f);
    float dz = fmaxf(fmaxf(b0.z - p.z, p.z - b1.y), 0.f);
    return (dx * dx + dy * dy + dz * dz) * 0.999999f;
}

// A query whose search sphere does not reach the root box of the tree finds nothing.  Clamped into the box such
// queries would pile up in the cells of its faces (a caustic map covers a corner of the scene, the queries all of
// it: millions in one bucket); they get buckets of their own behind the 2^bits of the curve, dealt by query number:
// their result does not depend on who their neighbours are.
TKD bool sphere_misses_box(float4 b0, float4 b1, float px, float py, float pz, float r2) {
    const float dx = fmaxf(fmaxf(b0.x - px, px - b0.w), 0.f);
    const float dy = fmaxf(fmaxf(b0.y - py, py - b1.x), 0.f);
    const float dz = fmaxf(fmaxf(b0.z - pz, pz - b1.y), 0.f);
    return !((dx * dx + dy * dy + dz * dz) * 0.999999f < r2);
}

// Squared distance from a point to the nearest and to the farthest point of a node box, rounded like a
// photon's dist2 (no photon of the box is farther than the latter).
TKD float box_near2(float4 b0, float4 b1, F3 p) {
    const float dx = fmaxf(fmaxf(b0.x - p.x, p.x - b0.w), 0.f);
    const float dy = fmaxf(fmaxf(b0.y - p.y, p.y - b1.x), 0.f);
    const float dz = fmaxf(fmaxf(b0.z - p.z, p.z - b1.y), 0.f);
    return dx * dx + dy * dy + dz * dz;
}
TKD float box_far2(float4 b0, float4 b1, F3 p) {
    const float dx = fmaxf(fabsf(b0.x - p.x), fabsf(b0.w - p.x));
    const float dy = fmaxf(fabsf(b0.y - p.y), fabsf(b1.x - p.y));
    const float dz = fmaxf(fabsf(b0.z - p.z), fabsf(b1.y - p.z));
    return dx * dx + dy * dy + dz * dz;
}

// ------------------------------------------- per-query kernel (one wave each)
// The wave keeps the candidates of its query in an LDS buffer of 64*EPL
// (dist2, index) keys and, whenever the buffer could overflow, cuts it back to
// the k smallest with a radix select over the key bits; the k-th key then
// becomes the new search bound.  This is the bounded max-heap of
// neighbor_search (src/tracer/neighbor.cpp:31-83) restated for 64 lanes.

template <int EPL>
TKD unsigned long long wave_select_radix(unsigned long long* keys, uint32_t* pay, int* count_io, int k, int lane) {
    int count = *count_io;
    const int nchunk = (count + 63) >> 6;
    // k-th smallest high word (dist2 bits; non-negative floats order as integers)
    uint32_t prefix = 0;
    int kk = k;
    for (int bit = 30; bit >= 0; --bit) {
        const uint32_t above = ~((2u << bit) - 1u);
        int c0 = 0;
        for (int ch = 0; ch < nchunk; ++ch) {
            int e = ch * 64 + lane;
            bool t = false;
            if (e < count) {
                uint32_t hi = (uint32_t)(keys[e] >> 32);
                t = ((hi & above) == prefix) && !((hi >> bit) & 1u);
            }
            c0 += __popcll(__ballot(t));
        }
        if (kk > c0) { kk -= c0; prefix |= 1u << bit; }
    }
    const uint32_t tau = prefix;
    int less = 0, eq = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        int e = ch * 64 + lane;
        uint32_t hi = e < count ? (uint32_t)(keys[e] >> 32) : 0xffffffffu;
        less += __popcll(__ballot(e < count && hi < tau));
        eq += __popcll(__ballot(e < count && hi == tau));
    }
    uint32_t tau_lo = 0xffffffffu;
    if (less + eq > k) {
        // more photons tie at the k-th distance than fit: lower index wins
        int need = k - less;
        uint32_t pl = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t above = bit == 31 ? 0u : ~((2u << bit) - 1u);
            int c0 = 0;
            for (int ch = 0; ch < nchunk; ++ch) {
                int e = ch * 64 + lane;
                bool t = false;
                if (e < count) {
                    unsigned long long key = keys[e];
                    uint32_t lo = (uint32_t)key;
                    t = ((uint32_t)(key >> 32) == tau) && ((lo & above) == pl) && !((lo >> bit) & 1u);
                }
                c0 += __popcll(__ballot(t));
            }
            if (need > c0) { need -= c0; pl |= 1u << bit; }
        }
        tau_lo = pl;
    }
    const unsigned long long cut = ((unsigned long long)tau << 32) | tau_lo;
    // compact in place: chunk c only writes below what it has read
    int w = 0;
    unsigned long long kmax = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        int e = ch * 64 + lane;
        unsigned long long key = 0;
        uint32_t p = 0;
        bool keep = false;
        if (e < count) { key = keys[e]; p = pay[e]; keep = key <= cut; }
        unsigned long long m = __ballot(keep);
        int dst = w + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) { keys[dst] = key; pay[dst] = p; kmax = key > kmax ? key : kmax; }
        w += __popcll(m);
    }
    *count_io = w;
    return wave_max_u64(kmax);
}

// The same cut, found faster in the common case: the keys are counted into 64
// equal bins of dist2 over [0, limit_d2] (no key lies beyond), a scan of the
// counts names the bin the k-th smallest key falls in, and the few keys of that
// bin are ranked against each other.  bin(dist2) is monotone, so the cut is the
// exact k-th smallest key.  More than 64 keys in that bin (many photons at one
// distance): the radix select above takes over.
//   hist: 64 ints, cand: 64 keys, both LDS of this wave.
template <int EPL>
TKD unsigned long long wave_select_prune(unsigned long long* keys, uint32_t* pay, int* count_io, int k, int lane,
                                         float limit_d2, int* hist, unsigned long long* cand) {
    const int count = *count_io;
    const int nchunk = (count + 63) >> 6;
    const float scale = 64.f / limit_d2;
    auto bin_of = [&](unsigned long long key) {
        const int b = (int)(__uint_as_float((uint32_t)(key >> 32)) * scale);
        return b > 63 ? 63 : (b < 0 ? 0 : b);
    };
    hist[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int e = ch * 64 + lane;
        if (e < count) atomicAdd(&hist[bin_of(keys[e])], 1);
    }
    __builtin_amdgcn_wave_barrier();
    const int mine = hist[lane];
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const unsigned long long reach = __ballot(incl >= k);
    if (reach == 0ull) return wave_select_radix<EPL>(keys, pay, count_io, k, lane);   // count < k: not expected
    const int bstar = __ffsll((long long)reach) - 1;
    const int in_bin = __shfl(mine, bstar);
    const int need = k - (__shfl(incl, bstar) - in_bin);
    if (in_bin > 64) return wave_select_radix<EPL>(keys, pay, count_io, k, lane);
    int nb = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int e = ch * 64 + lane;
        unsigned long long key = 0;
        bool t = false;
        if (e < count) { key = keys[e]; t = bin_of(key) == bstar; }
        const unsigned long long m = __ballot(t);
        if (t) cand[nb + __popcll(m & ((1ull << lane) - 1ull))] = key;
        nb += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long my_key = lane < nb ? cand[lane] : ~0ull;
    int rank = 0;
    for (int j = 0; j < nb; ++j) rank += cand[j] < my_key ? 1 : 0;
    const unsigned long long at = __ballot(lane < nb && rank == need - 1);
    const int src = __ffsll((long long)at) - 1;
    const unsigned long long cut = ((unsigned long long)(uint32_t)__shfl((int)(my_key >> 32), src) << 32) |
                                   (unsigned long long)(uint32_t)__shfl((int)(uint32_t)my_key, src);
    __builtin_amdgcn_wave_barrier();
    // compact in place: chunk c only writes below what it has read
    int w = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int e = ch * 64 + lane;
        unsigned long long key = 0;
        uint32_t p = 0;
        bool keep = false;
        if (e < count) { key = keys[e]; p = pay[e]; keep = key <= cut; }
        const unsigned long long m = __ballot(keep);
        const int dst = w + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) { keys[dst] = key; pay[dst] = p; }
        w += __popcll(m);
    }
    *count_io = w;
    return cut;
}

// Appends the photons of leaf node `node` that lie inside the current bound to
// the candidate buffer.  Lanes [lane0, lane0+32) of the wave serve this leaf.
static_assert(64 % kLeafCap == 0, "a step of a gather kernel serves 64 / kLeafCap leaves, one photon per lane");
constexpr int kLeavesPerStep = 64 / kLeafCap;
constexpr int kSortFrontierFrom = 20;   // the per-query kernel orders its leaves by distance from this many on

// What the grouped kernel hands to the per-query kernel with a query: the group it was in, and per
// group the leaves under the union of its search boxes (at most kMaxGroupLeaves <= 64).
constexpr int kMaxGroupLeaves = 24 * kLeavesPerStep;
constexpr int kGroupLeafStride = 64 * kLeavesPerStep;   // longer lists (up to this) are for the selection kernel only
struct GroupLeaves {
    int32_t* query_group;   // per entry of the hand-over list: group, or -1 (search from the root)
    int32_t* leaves;        // kGroupLeafStride node numbers per group
    int32_t* count;
};
// What the grouped kernel leaves for the selection kernel: the groups in which many lanes met k photons or more.
struct SelectWork {
    int32_t* groups;      // group numbers
    int32_t* count;       // how many
    int32_t* slot_cnt;    // per query slot (curve order) of such a group: photons inside the search radius, -1: nothing to select
    int32_t* met_k;       // counts the queries of this launch that met k photons or more (or null)
    int min_groups;       // fewer groups than this cannot fill the GPU: their lanes go to the per-query kernel
    int heavy;            // many gathers of the map's last launch met k photons: radii have been cut, long leaf lists are taken
};

template <int EPL>
__global__ void __launch_bounds__(256) k_gather_wave(MapView mv, QueryView qv, const int32_t* list,
                                                     const int32_t* list_count, int64_t nq, GatherOut out, GroupLeaves gl) {
    constexpr int CAP = 64 * EPL;
    constexpr int FCAP = 128 * kLeavesPerStep;   // frontier entries of the level-by-level descent
    __shared__ unsigned long long s_keys[4][CAP];
    __shared__ uint32_t s_pay[4][CAP];
    __shared__ int s_node[4][64];
    __shared__ float s_nd2[4][64];
    __shared__ int s_front[4][2][FCAP];
    __shared__ float s_tab[1024];
    __shared__ int s_hist[4][64];
    __shared__ unsigned long long s_cand[4][64];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_tab[i] = mv.dirtab[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long* keys = s_keys[wave];
    uint32_t* pay = s_pay[wave];
    int* st_node = s_node[wave];
    float* st_d2 = s_nd2[wave];
    const int64_t total = list ? (int64_t)*list_count : nq;
    const int k = mv.k;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long found_sum = 0;
    for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < total; w += (int64_t)gridDim.x * 4) {
        const int64_t q = list ? (int64_t)list[w] : w;
#ifdef TK_GW_STAMPS
        const unsigned long long st_q0 = __builtin_amdgcn_s_memtime();
        unsigned long long st_a = st_q0;
        int st_path = 0, st_fcount = 0, st_prunes = 0;
#endif
        const QLoad Q = load_query(qv, q);
        unsigned long long bound = (unsigned long long)__float_as_uint(Q.r2) << 32;
        float bound_d2 = Q.r2;
        int count = 0;
        const bool searchable = mv.n > 0 && Q.r2 > 0.f;

        // The photons of up to kLeavesPerStep leaves (nodes[0 .. count), wave-uniform) against the
        // bound, one photon per lane; survivors join the buffer.
        auto take_leaves = [&](const int* nodes, int count_nodes) {
            const int slot = lane / kLeafCap, ph = lane % kLeafCap;
            bool acc = false;
            unsigned long long key = 0;
            uint32_t at = 0;
            if (slot < count_nodes) {
                const int leaf = nodes[slot] - mv.nl;
                const int64_t s0 = leaf_start(leaf, mv.n, mv.depth), e0 = leaf_start(leaf + 1, mv.n, mv.depth);
                if (ph < (int)(e0 - s0)) {
                    at = (uint32_t)(s0 + ph);
                    float4 pp = mv.pos4[at];
                    float d2 = dist2_exact(pp.x, pp.y, pp.z, Q.p);
                    key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)__float_as_uint(pp.w);
                    acc = key < bound;
                }
            }
            unsigned long long m = __ballot(acc);
            if (m) {
                int dst = count + __popcll(m & lt_mask);
                if (acc) { keys[dst] = key; pay[dst] = at; }
                count += __popcll(m);
                if (count > CAP - 64) {
                    bound = wave_select_prune<EPL>(keys, pay, &count, k, lane, bound_d2, s_hist[wave], s_cand[wave]);
                    // a photon exactly at the k-th distance can still win on its index
                    bound_d2 = __uint_as_float((uint32_t)(bound >> 32) + 1u);
                }
            }
        };

        // 1. Level-by-level descent: the lanes test the children of a whole frontier of nodes at
        //    once, so a level costs one memory round trip instead of one per node.  The tree is
        //    complete, every leaf sits at the same depth.
        bool done = !searchable;
        if (searchable) {
            int (*front)[FCAP] = s_front[wave];
            int cur = 0, fcount = 0;
            const int l0 = mv.depth < 6 ? mv.depth : 6;
            const int n0 = 1 << l0;
            bool fits = true;
            // A query handed on by the grouped kernel comes with the leaves under its group's search
            // boxes: those within its own reach are the leaves the descent below would end with.
            const int gid = (list && gl.query_group) ? gl.query_group[w] : -1;
            if (gid >= 0) {
                const int have = gl.count[gid];
                const int node = lane < have ? gl.leaves[(int64_t)gid * kGroupLeafStride + lane] : 0;
                const bool keep = lane < have && box_dist2(mv, node, Q.p) < bound_d2;
                const unsigned long long m = __ballot(keep);
                if (keep) front[cur][__popcll(m & lt_mask)] = node;
                fcount = __popcll(m);
            }
#ifdef TK_GW_STAMPS
            if (gid < 0) st_path = 1;
#endif
            for (int base = 0; base < n0 && gid < 0; base += 64) {
                const int node = n0 + base + lane;
                const bool keep = base + lane < n0 && box_dist2(mv, node, Q.p) < bound_d2;
                unsigned long long m = __ballot(keep);
                if (keep) front[cur][fcount + __popcll(m & lt_mask)] = node;
                fcount += __popcll(m);
            }
            for (int level = l0; level < mv.depth && fits && gid < 0; ++level) {
                int ncount = 0;
                for (int e0 = 0; e0 < 2 * fcount; e0 += 64) {
                    const int e = e0 + lane;
                    const bool in = e < 2 * fcount;
                    const int node = in ? 2 * front[cur][e >> 1] + (e & 1) : 0;
                    const bool keep = in && box_dist2(mv, node, Q.p) < bound_d2;
                    unsigned long long m = __ballot(keep);
                    const int add = __popcll(m);
                    if (ncount + add > FCAP) { fits = false; break; }
                    if (keep) front[cur ^ 1][ncount + __popcll(m & lt_mask)] = node;
                    ncount += add;
                }
                cur ^= 1;
                fcount = ncount;
            }
#ifdef TK_GW_STAMPS
            st_a = __builtin_amdgcn_s_memtime();
            st_fcount = fcount;
#endif
            if (fits) {
                // The leaves of the frontier, 64 at a time, nearest first (every lane ranks its leaf's box
                // distance among the others'), so that the first cut brings the bound close to its final
                // value and the leaves beyond it are never loaded.  The photons of a step are requested
                // one step ahead of their tests.
                int* fr = front[cur];
                int* sorted = front[cur ^ 1];
                auto fetch = [&](const int* nodes, int count_nodes, float4* pp, uint32_t* at) {
                    const int slot = lane / kLeafCap, ph = lane % kLeafCap;
                    bool has = false;
                    if (slot < count_nodes) {
                        const int leaf = nodes[slot] - mv.nl;
                        const int64_t s0 = leaf_start(leaf, mv.n, mv.depth), e0 = leaf_start(leaf + 1, mv.n, mv.depth);
                        if (ph < (int)(e0 - s0)) {
                            *at = (uint32_t)(s0 + ph);
                            *pp = mv.pos4[*at];
                            has = true;
                        }
                    }
                    return has;
                };
                auto test = [&](bool has, float4 pp, uint32_t at) {
                    bool acc = false;
                    unsigned long long key = 0;
                    if (has) {
                        const float d2 = dist2_exact(pp.x, pp.y, pp.z, Q.p);
                        key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)__float_as_uint(pp.w);
                        acc = key < bound;
                    }
                    const unsigned long long m = __ballot(acc);
                    if (m) {
                        const int dst = count + __popcll(m & lt_mask);
                        if (acc) { keys[dst] = key; pay[dst] = at; }
                        count += __popcll(m);
                        if (count > CAP - 64) {
#ifdef TK_GW_STAMPS
                            ++st_prunes;
#endif
                            bound = wave_select_prune<EPL>(keys, pay, &count, k, lane, bound_d2, s_hist[wave], s_cand[wave]);
                            bound_d2 = __uint_as_float((uint32_t)(bound >> 32) + 1u);
                        }
                    }
                };
                for (int c0 = 0; c0 < fcount; c0 += 64) {
                    const int nc = fcount - c0 < 64 ? fcount - c0 : 64;
                    const int node = lane < nc ? fr[c0 + lane] : 0;
                    const float d = lane < nc ? box_dist2(mv, node, Q.p) : INFINITY;
                    int rank = lane;
                    if (fcount > kSortFrontierFrom) {   // a short list is scanned whole anyway: tree order will do
                        rank = 0;
                        for (int j = 0; j < nc; ++j) {
                            const float dj = lane_f(d, j);
                            rank += (dj < d || (dj == d && j < lane)) ? 1 : 0;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < nc) { sorted[rank] = node; st_d2[rank] = d; }
                    __builtin_amdgcn_wave_barrier();
                    // how many leaves of step i are still within the bound: 0 ends the chunk
                    const bool by_distance = fcount > kSortFrontierFrom;
                    auto within = [&](int i) {
                        if (i >= nc) return 0;
                        if (!by_distance) return i + 1 < nc ? 2 : 1;
                        if (!(uniform_f(st_d2[i]) < uniform_f(bound_d2))) return 0;
                        return (i + 1 < nc && uniform_f(st_d2[i + 1]) < uniform_f(bound_d2)) ? 2 : 1;
                    };
                    float4 pp_next = make_float4(0, 0, 0, 0);
                    uint32_t at_next = 0;
                    int n_next = within(0);
                    bool has_next = n_next > 0 && fetch(&sorted[0], n_next, &pp_next, &at_next);
                    for (int i = 0; n_next > 0; i += kLeavesPerStep) {
                        const float4 pp = pp_next;
                        const uint32_t at = at_next;
                        const bool has = has_next;
                        n_next = within(i + kLeavesPerStep);
                        has_next = n_next > 0 && fetch(&sorted[i + kLeavesPerStep], n_next, &pp_next, &at_next);
                        test(has, pp, at);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                done = true;
            }
        }
        // 2. A search region too wide for the frontier (huge radii): depth-first, near child first,
        //    pruning against the bound as it tightens.
        if (!done) {
#ifdef TK_GW_STAMPS
            st_path = 2;
#endif
            count = 0;
            bound = (unsigned long long)__float_as_uint(Q.r2) << 32;
            bound_d2 = Q.r2;
            int sp = 1;
            if (lane == 0) { st_node[0] = 1; st_d2[0] = 0.f; }
            while (sp > 0) {
                --sp;
                const int node = uniform_i(st_node[sp]);
                const float nd2 = uniform_f(st_d2[sp]);
                if (!(nd2 < bound_d2)) continue;
                if (node < mv.nl) {
                    const int c0 = 2 * node, c1 = c0 + 1;
                    const float d0 = box_dist2(mv, c0, Q.p), d1 = box_dist2(mv, c1, Q.p);
                    const bool near0 = d0 <= d1;
                    const int nn = near0 ? c0 : c1, fn = near0 ? c1 : c0;
                    const float dn = near0 ? d0 : d1, df = near0 ? d1 : d0;
                    if (df < bound_d2) { if (lane == 0) { st_node[sp] = fn; st_d2[sp] = df; } ++sp; }
                    if (dn < bound_d2) { if (lane == 0) { st_node[sp] = nn; st_d2[sp] = dn; } ++sp; }
                } else {
                    if (lane == 0) st_node[63] = node;    // the stack never reaches its last entry
                    take_leaves(&st_node[63], 1);
                }
            }
        }
#ifdef TK_GW_STAMPS
        const unsigned long long st_b = __builtin_amdgcn_s_memtime();
#endif
        float r2ret = Q.r2;
        if (count >= k) {
            unsigned long long kmax;
            if (count > k) kmax = wave_select_prune<EPL>(keys, pay, &count, k, lane, bound_d2, s_hist[wave], s_cand[wave]);
            else {
                unsigned long long mine = 0;
                for (int e = lane; e < count; e += 64) mine = keys[e] > mine ? keys[e] : mine;
                kmax = wave_max_u64(mine);
            }
            r2ret = __uint_as_float((uint32_t)(kmax >> 32));
        }
#ifdef TK_GW_STAMPS
        const unsigned long long st_c = __builtin_amdgcn_s_memtime();
#endif
        float sx = 0, sy = 0, sz = 0;
        for (int e = lane; e < count; e += 64) {
            float4 pw = mv.pow4[pay[e]];
            F3 wi = decode_dir(s_tab, __float_as_uint(pw.w));
            float c = wi.x * Q.n.x + wi.y * Q.n.y + wi.z * Q.n.z;
            if (Q.won * c > 0) {
                float ac = fabsf(c);
                sx += pw.x * ac; sy += pw.y * ac; sz += pw.z * ac;
            }
            if (out.idx) out.idx[q * k + e] = (int32_t)(uint32_t)keys[e];
        }
        if (out.idx) for (int e = count + lane; e < k; e += 64) out.idx[q * k + e] = -1;
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (lane == 0) finish_query(out, q, Q, count, r2ret, mk3(sx, sy, sz));
        found_sum += (unsigned long long)count;
#ifdef TK_GW_STAMPS
        if (out.photon_sum && lane == 0) {   // profiling build: per way a query took (group leaves / descent / depth first) the slowest, the sum, the count
            const unsigned long long st_d = __builtin_amdgcn_s_memtime();
            const unsigned long long dt = st_d - st_q0;
            // the slowest query with its leaf count and the cuts it made on the way, packed behind its ticks
            atomicMax(out.photon_sum + 4 + st_path, (dt << 20) | ((unsigned long long)(st_fcount < 1023 ? st_fcount : 1023) << 10) |
                                                        (unsigned long long)(st_prunes < 1023 ? st_prunes : 1023));
            atomicAdd(out.photon_sum + 7 + st_path, dt);
            atomicAdd(out.photon_sum + 10 + st_path, 1ull);
            if (st_path == 0) {
                atomicAdd(out.photon_sum + 13, st_a - st_q0);
                atomicAdd(out.photon_sum + 14, st_b - st_a);
                atomicAdd(out.photon_sum + 15, st_c - st_b);
            }
        }
#endif
    }
    if (out.photon_sum && lane == 0 && found_sum) atomicAdd(out.photon_sum, found_sum);
    if (out.overflow_sum && list && blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(out.overflow_sum, (unsigned long long)*list_count);
}

// ---------------------------------------- grouped kernel (64 queries per wave)
// Queries arrive sorted along a Hilbert curve, so the 64 of a wave are
// neighbours in space.  The wave finds the leaves that overlap the union of
// their search boxes by descending the tree level by level (the lanes test a
// whole frontier of nodes at once: one memory round trip per level), then
// streams those leaves through LDS two at a time: the lanes that loaded the
// photons stage them, every lane reads them back as broadcasts and tests them
// against its own query with a handful of branch-free vector instructions,
// while the loads of the next two leaves are already in flight.
// A query that meets fewer than k photons inside its radius is finished here
// (no selection is needed: found < k keeps them all and leaves the radius
// unchanged); the rest go to the per-query kernel.

// What the grouped kernel reads of every query at the start of a group, in the order of the curve.
struct SortedQueries {
    const float4* q0;     // p.xyz, radius^2
    const float4* q1;     // n.xyz, dot(wo, n)
    const float* pixel;   // pixel index (int bits)
    const float* r2s;     // radius^2 to search with: the query's own, or one cut by k_query_bound
};
__global__ void k_sort_queries(QueryView qv, const uint32_t* order, int64_t nq, float4* q0s, float4* q1s, float* pixel,
                               float* r2s) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int64_t q = order[i];
    const float4 a = qv.q0[q];
    q0s[i] = a;
    q1s[i] = qv.q1[q];
    pixel[i] = qv.q2[q].w;
    r2s[i] = a.w;
}

// Radii far larger than the photon spacing (the first iterations of a render, dense maps) make every gather
// wade through thousands of photons for its k nearest.  This kernel cuts such a radius beforehand: a lane
// looks up a node of at least k photons near its query (by the query's place on the curve the photons are
// sorted along, or, in a median-split tree, by walking down to the nearer child at every level); the
// k-th smallest dist2 among the photons of that node bounds the dist2 of the k-th nearest photon of the
// whole map from above, and the k nearest inside a sphere that holds k photons or more are the k nearest
// inside any larger one.  The k-th smallest is taken to the upper edge of its bin among kSelBins bins of
// dist2 up to the radius (or the node's far corner when that is nearer).  The host launches it while many
// gathers of the map's last launch met k photons; otherwise radii are small already.
constexpr int kSelBins = 32;        // histogram bins of a lane (two 16-bit counters to a word: kSelBins / 2 words)
__global__ void __launch_bounds__(256) k_query_bound(MapView mv, int64_t nq, const float4* q0s, float* r2s,
                                                     int level) {
    __shared__ uint32_t s_hist[kSelBins / 2][256];
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float4 a = q0s[i];
    const F3 p = mk3(a.x, a.y, a.z);
    const float r2 = a.w;
    if (!(r2 > 0.f) || sphere_misses_box(mv.node_box[2], mv.node_box[3], p.x, p.y, p.z, r2)) return;
    int node = 1;
    if (mv.keys) {
        // the photons lie in the order of their cells along the curve: the node around the place of the query's cell
        const float4 b0 = mv.node_box[2], b1 = mv.node_box[3];
        const float ex = b0.w - b0.x, ey = b1.x - b0.y, ez = b1.y - b0.z;
        const float fx = fminf(fmaxf((p.x - b0.x) * (ex > 0 ? 1023.f / ex : 0.f), 0.f), 1023.f);
        const float fy = fminf(fmaxf((p.y - b0.y) * (ey > 0 ? 1023.f / ey : 0.f), 0.f), 1023.f);
        const float fz = fminf(fmaxf((p.z - b0.z) * (ez > 0 ? 1023.f / ez : 0.f), 0.f), 1023.f);
        const uint32_t key = hilbert30((uint32_t)fx, (uint32_t)fy, (uint32_t)fz);
        int64_t lo = 0, hi = mv.n;          // first photon whose key is not below the query's
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (mv.keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        const int64_t at = lo < mv.n ? lo : (int64_t)mv.n - 1;
        const int64_t leaf = ((at + 1) * mv.nl - 1) / mv.n;     // the leaf photon `at` lies in: inverse of leaf_start
        node = (1 << level) + (int)(leaf >> (mv.depth - level));
    } else {
        for (int l = 0; l < level; ++l) {   // the nearer child at every level
            const float4 a0 = mv.node_box[4 * (int64_t)node], a1 = mv.node_box[4 * (int64_t)node + 1];
            const float4 c0 = mv.node_box[4 * (int64_t)node + 2], c1 = mv.node_box[4 * (int64_t)node + 3];
            const float n0 = box_near2(a0, a1, p), n1 = box_near2(c0, c1, p);
            const bool right = n1 < n0 || (n1 == n0 && box_far2(c0, c1, p) < box_far2(a0, a1, p));
            node = 2 * node + (right ? 1 : 0);
        }
    }
    // no photon of the node is farther than its far corner (dist2 and this are rounded alike, term by term)
    const float far2 = box_far2(mv.node_box[2 * (int64_t)node], mv.node_box[2 * (int64_t)node + 1], p) * 1.00001f;
    const float top = fminf(far2, r2);
    const int h = mv.depth - level;
    const int64_t j = node - (1 << level);
    const int64_t s0 = leaf_start(j << h, mv.n, mv.depth), e0 = leaf_start((j + 1) << h, mv.n, mv.depth);
    const float scale = ((float)kSelBins - 0.5f) / top;
#pragma unroll
    for (int r = 0; r < kSelBins / 2; ++r) s_hist[r][threadIdx.x] = 0u;
    for (int64_t at = s0; at < e0; ++at) {
        const float4 pp = mv.pos4[at];
        const int b = (int)fminf(dist2_exact(pp.x, pp.y, pp.z, p) * scale, (float)(kSelBins - 1));   // the last bin: beyond, too
        atomicAdd(&s_hist[b >> 1][threadIdx.x], 1u << ((b & 1) * 16));   // the lane's own column
    }
    int cum = 0, bstar = kSelBins - 1;
    bool hit = false;
#pragma unroll
    for (int r = 0; r < kSelBins / 2; ++r) {
        const uint32_t v = s_hist[r][threadIdx.x];
        const int c0 = (int)(v & 0xffffu), c1 = (int)(v >> 16);
        if (!hit) { cum += c0; if (cum >= mv.k) { hit = true; bstar = 2 * r; } }
        if (!hit) { cum += c1; if (cum >= mv.k) { hit = true; bstar = 2 * r + 1; } }
    }
    // every photon counted up to bin bstar < kSelBins - 1 has int(dist2 * scale) <= bstar: dist2 below this edge
    const float edge = (hit && bstar < kSelBins - 1) ? (float)(bstar + 1) / scale * 1.00001f : far2;
    const float cut = fminf(far2, edge);
    if (cut > 0.f && cut < r2) r2s[i] = cut;
}

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// Selection of the k nearest inside the grouped kernel (groups in which many queries meet k photons
// or more: the first iterations of a render, large maps).  Every such lane narrows a window of
// dist2 that holds its k-th nearest photon: the photons of the group's leaves are streamed again
// and counted into kSelBins equal bins of dist2 per lane; the bin of the k-th is refined the same
// way while it holds more than kSelStash photons, then collected: everything below the bin is in,
// the few photons of the bin are ranked by (dist2, photon index).  bin = int(dist2 * scale) - off
// with scale a power-of-two multiple of the first window's, so that the bins of a refinement nest
// exactly in the bin they refine and every pass classifies a photon the same way.
constexpr int kSelStash = 8;        // photons of the final bin a lane can rank (<= kSelBins / 2: same words)
constexpr int kSelMinLanes = 48;    // a group with fewer lanes at k or above hands them to the per-query kernel
constexpr int kSelMinLanesLarge = 32;      // ... in a map of more than kSelLargeMap photons (it does not fit the caches, and
constexpr int64_t kSelLargeMap = 1 << 20;  //     the per-query kernel pays for that query by query)
constexpr int kSelMinGroups = 2048; // ... and so do all groups when they are fewer than this (half the waves a GPU holds)
constexpr int kSelMaxPasses = 5;
constexpr int kSelHeavyPercent = 75;       // share of a map's last launch that met k photons from which radii are cut and
constexpr int kSelHeavyPercentLarge = 25;  // long leaf lists taken; in a map of more than kSelLargeMap photons
static_assert(kSelStash <= kSelBins / 2, "the stash of a lane reuses the words of its histogram");

// Everything the grouped kernel needs of one map's gather.  A launch takes one or two of them: the gathers of
// the two maps of a render (radiance_estimate on the caustic and on the global map, src/tracer/TkSppmTracer.cpp:323,
// :342) are independent and their groups are dealt to the waves of ONE grid, the first job's groups first.
struct GroupJob {
    MapView mv;
    QueryView qv;
    SortedQueries qs;
    const uint32_t* order;
    int64_t nq;
    GatherOut out;
    int32_t* overflow;
    int32_t* overflow_count;
    GroupLeaves gl;
    SelectWork sw;
    int sel_min_lanes;
};
template <bool WANT_IDX>
__global__ void __launch_bounds__(256, 5) k_gather_group(GroupJob job0, GroupJob job1, int64_t ngroups0, int64_t ngroups_all) {
    constexpr int FCAP = kGroupLeafStride;
    __shared__ float s_tab[1024];
    // candidates of the current step, struct of arrays so that four of them load as one float4
    __shared__ __align__(16) float s_x[4][64];
    __shared__ __align__(16) float s_y[4][64];
    __shared__ __align__(16) float s_z[4][64];
    __shared__ float4 s_rec[4][64][2];   // (wi.xyz, power.r) (power.g, power.b, original index, -)
    __shared__ int s_front[4][2][FCAP];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_tab[i] = job0.mv.dirtab[i];   // one table for every map of a context
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int leaf_slot = lane / kLeafCap, ph = lane % kLeafCap;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int64_t ngroups = ngroups_all;
    int (*front)[FCAP] = s_front[wave];
    // per job: photons found by the queries finished here, photons staged, queries that met k photons or more
    unsigned long long found_sum[2] = {0, 0}, staged_sum[2] = {0, 0};
    int flagged_sum[2] = {0, 0};
    // Groups are handed out one at a time: the waves of the grid (sized to what the GPU holds at once)
    // stay busy until the last group is taken.  Two group indices are kept ahead of the one being worked
    // on, and the queries of the next group are loaded while the current one is worked on, so that a
    // group does not start with a chain of memory round trips.
    struct Pre { float4 a, b; int q, pixel; float r2s; bool valid; };
    auto prefetch = [&](int64_t g_all) {
        Pre P;
        const bool second = g_all >= ngroups0;
        const SortedQueries& qs = second ? job1.qs : job0.qs;
        const uint32_t* order = second ? job1.order : job0.order;
        const int64_t nq = second ? job1.nq : job0.nq;
        const int64_t slot = (g_all - (second ? ngroups0 : 0)) * 64 + lane;
        P.valid = g_all < ngroups && slot < nq;
        P.a = make_float4(0, 0, 0, -1.f); P.b = make_float4(0, 0, 0, 0); P.q = 0; P.pixel = 0; P.r2s = -1.f;
        if (P.valid) {
            P.a = qs.q0[slot]; P.b = qs.q1[slot];          // in the order of the curve: coalesced
            P.q = (int)order[slot];
            P.pixel = __float_as_int(qs.pixel[slot]);
            P.r2s = qs.r2s[slot];
        }
        return P;
    };
    // Only lane 0's value is ever read (through readfirstlane); the other lanes' is left as it is (an
    // empty asm "defines" it): merging with a constant would make the wave wait for the atomic where
    // it is issued instead of where its result is needed.
    // Groups are dealt round-robin to the waves of the grid (what the GPU holds at once).  A wave knows
    // its groups without asking: a cursor with an atomic per group made every group start with a
    // contended round trip (0.31 against 0.23 ms per launch).
    const int n_waves = (int)gridDim.x * 4;
    int next_group = (int)blockIdx.x * 4 + wave;
    auto next_index = [&]() { const int v = next_group; next_group += n_waves; return v; };
    int gi0 = next_index(), gi1 = next_index();
#ifdef TK_GG_STAMPS
    unsigned long long st_sum[5] = {0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
#define TK_STAMP(i) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_sum[i] += now_ - st_t; st_t = now_; }
#else
#define TK_STAMP(i)
#endif
#ifdef TK_GG_TURNS
    // turns of phase B as it is (two halves per step), per whole step, per pair of steps, per group; steps; groups; accepted pairs
    unsigned long long tn_half = 0, tn_step = 0, tn_pair = 0, tn_group = 0, tn_steps = 0, tn_groups = 0, tn_pairs_total = 0;
    float tn_pair_acc = 0.f, tn_group_acc = 0.f;
#endif
    Pre cur_q = prefetch((int64_t)uniform_i(gi0));
    while (true) {
        const int64_t g_all = (int64_t)uniform_i(gi0);
        TK_STAMP(0)
        if (g_all >= ngroups) break;
        // the job this group belongs to (wave-uniform): its arguments are read from the kernel's argument block
        const int jb = g_all >= ngroups0 ? 1 : 0;
        const GroupJob& J = jb ? job1 : job0;
        const int64_t g = g_all - (jb ? ngroups0 : 0);
        const MapView mv = J.mv;
        const QueryView qv = J.qv;
        const int64_t nq = J.nq;
        const GatherOut out = J.out;
        int32_t* const overflow = J.overflow;
        int32_t* const overflow_count = J.overflow_count;
        const GroupLeaves gl = J.gl;
        const SelectWork sw = J.sw;
        const int sel_min_lanes = J.sel_min_lanes;
        const int k = mv.k;
        const bool heavy = sw.heavy != 0;
        const float4 root0 = mv.node_box[2], root1 = mv.node_box[3];   // box of the whole map (heap node 1)
        const int gi2 = next_index();
        const Pre next_q = prefetch((int64_t)uniform_i(gi1));
        const bool valid = cur_q.valid;
        const int64_t q = cur_q.q;
        // what the search needs of the query; the rest is read when the query is finished
        struct { F3 p, n; float r2, won; } Q;
        Q.p = mk3(cur_q.a.x, cur_q.a.y, cur_q.a.z); Q.r2 = cur_q.a.w;
        Q.n = mk3(cur_q.b.x, cur_q.b.y, cur_q.b.z); Q.won = cur_q.b.w;
        const float r2_in = Q.r2;
        const int pixel = cur_q.pixel;
        // a radius cut by k_query_bound holds k photons or more: such a lane ends up selected or handed on
        const bool shrunk = cur_q.r2s < r2_in;
        Q.r2 = cur_q.r2s;
        // a query that is not one, or whose search sphere does not reach the photons at all, finds nothing:
        // it takes no part in the search (its box would stretch the union over empty space)
        if (!valid || !(Q.r2 > 0.f) || sphere_misses_box(root0, root1, Q.p.x, Q.p.y, Q.p.z, Q.r2)) Q.r2 = -1.f;
        const float rr = Q.r2 > 0.f ? sqrtf(Q.r2) * 1.00001f : -INFINITY;
        // union of the search boxes of the wave
        const float lox = wave_min_all(Q.r2 > 0.f ? Q.p.x - rr : INFINITY);
        const float loy = wave_min_all(Q.r2 > 0.f ? Q.p.y - rr : INFINITY);
        const float loz = wave_min_all(Q.r2 > 0.f ? Q.p.z - rr : INFINITY);
        const float hix = wave_max_all(Q.r2 > 0.f ? Q.p.x + rr : -INFINITY);
        const float hiy = wave_max_all(Q.r2 > 0.f ? Q.p.y + rr : -INFINITY);
        const float hiz = wave_max_all(Q.r2 > 0.f ? Q.p.z + rr : -INFINITY);
        // a wave whose queries are spread out (ends of the curve, isolated surfaces) would test far
        // more photons than its queries need: it hands all of them to the per-query kernel instead
        const float own_sum = wave_sum_all(Q.r2 > 0.f ? 8.f * rr * rr * rr : 0.f);
        const int n_live = __popcll(__ballot(Q.r2 > 0.f));
        bool spread = n_live > 0 && (hix - lox) * (hiy - loy) * (hiz - loz) * (float)n_live > 64.f * own_sum;
        int cnt = 0;
        int cur = 0, fcount = 0;
        float sx = 0, sy = 0, sz = 0;
        auto overlaps = [&](int node) {
            const float4 b0 = mv.node_box[2 * (int64_t)node], b1 = mv.node_box[2 * (int64_t)node + 1];
            return b0.x <= hix && b0.w >= lox && b0.y <= hiy && b1.x >= loy && b0.z <= hiz && b1.y >= loz;
        };
        TK_STAMP(1)
        if (mv.n > 0 && lox <= hix && !spread) {
            // 1. leaves under the union box.  The descent starts with every node of level 6 (one round
            //    trip) and then goes down up to three levels at a time: the descendants of a frontier
            //    node three levels below are tested directly (a box that overlaps lies in boxes that
            //    overlap, so skipping the levels in between loses nothing), which costs a few more box
            //    tests and saves two of three dependent round trips.
            const int l0 = mv.depth < 6 ? mv.depth : 6;
            const int n0 = 1 << l0;
            {
                const int node = n0 + lane;
                const bool keep = lane < n0 && overlaps(node);
                unsigned long long m = __ballot(keep);
                if (keep) front[cur][__popcll(m & lt_mask)] = node;
                fcount = __popcll(m);
            }
            for (int level = l0; level < mv.depth && !spread;) {
                int jump = mv.depth - level < 3 ? mv.depth - level : 3;
                while (jump > 1 && (fcount << jump) > 64) --jump;
                const int total = fcount << jump;
                int ncount = 0;
                for (int e0 = 0; e0 < total; e0 += 64) {
                    const int e = e0 + lane;
                    const bool in = e < total;
                    const int node = in ? (front[cur][e >> jump] << jump) + (e & ((1 << jump) - 1)) : 0;
                    const bool keep = in && overlaps(node);
                    unsigned long long m = __ballot(keep);
                    const int add = __popcll(m);
                    if (ncount + add > FCAP) { spread = true; break; }
                    if (keep) front[cur ^ 1][ncount + __popcll(m & lt_mask)] = node;
                    ncount += add;
                }
                cur ^= 1;
                fcount = ncount;
                level += jump;
            }
            // A wave that would have to stream far more leaves than its neighbours (its queries straddle
            // two surfaces, or one of them still has a large radius) would hold up the whole launch.  While
            // many gathers meet k photons (first iterations, dense maps) long lists are the rule and are taken.
            if (fcount > (heavy ? FCAP : kMaxGroupLeaves)) spread = true;
        }
        TK_STAMP(2)
        if (mv.n > 0 && lox <= hix && !spread && fcount > 0) {
            // 2. the photons of those leaves, a step of kLeavesPerStep leaves at a time (one photon per
            //    lane, lane = leaf_slot * kLeafCap + photon), the next step's loads in flight.  A step has
            //    two phases.  A: every lane runs all staged positions against its own query with the exact
            //    distance rule and keeps the outcome as a bit mask: the positions come four at a time as
            //    LDS broadcasts and the arithmetic is on 4-vectors (packed instructions), the sign of
            //    dist2 - radius2 is shifted into the mask.  B: every lane walks the set bits of its mask
            //    and adds those photons' contributions: a tenth of the candidates, so the per-photon
            //    shading work is done for the accepted ones only.
            auto fetch = [&](int i, float4* pp, float4* pw) {
                bool has = false;
                if (i + leaf_slot < fcount) {
                    const int leaf = front[cur][i + leaf_slot] - mv.nl;
                    const int64_t s0 = leaf_start(leaf, mv.n, mv.depth), e0 = leaf_start(leaf + 1, mv.n, mv.depth);
                    has = ph < (int)(e0 - s0);
                    if (has) { *pp = mv.pos4[s0 + ph]; *pw = mv.pow4[s0 + ph]; }
                }
                return has;
            };
            // the photons a step has loaded go to LDS, packed to the front; returns how many
            auto stage = [&](bool has, float4 pp, float4 pw) {
                const unsigned long long staged = __ballot(has);
                const int n_in = __popcll(staged);
                const int n4 = (n_in + 3) & ~3;
                __builtin_amdgcn_wave_barrier();      // the previous step has been read by every lane
                if (has) {
                    const int at = __popcll(staged & lt_mask);
                    F3 wi = decode_dir(s_tab, __float_as_uint(pw.w));
                    s_x[wave][at] = pp.x; s_y[wave][at] = pp.y; s_z[wave][at] = pp.z;
                    s_rec[wave][at][0] = make_float4(wi.x, wi.y, wi.z, pw.x);
                    s_rec[wave][at][1] = make_float4(pw.y, pw.z, pp.w, 0.f);
                } else {
                    // the slots up to the next multiple of four hold a position no query accepts
                    const int at = n_in + __popcll(~staged & lt_mask);
                    if (at < n4) { s_x[wave][at] = INFINITY; s_y[wave][at] = 0.f; s_z[wave][at] = 0.f; }
                }
                __builtin_amdgcn_wave_barrier();
                return n_in;
            };
            float4 pp = make_float4(0, 0, 0, 0), pw = pp;
            bool has = fetch(0, &pp, &pw);
            // negated query position and radius as pairs: dist2 - radius2 is sums and products of pairs
            const f2v nqx = {-Q.p.x, -Q.p.x}, nqy = {-Q.p.y, -Q.p.y}, nqz = {-Q.p.z, -Q.p.z}, nr2 = {-Q.r2, -Q.r2};
            // The normal turned to the side wo is on: f() is non-zero when wi and wo are on one side
            // of the surface, which then reads wi . nw > 0, and |wi . n| = wi . nw there.
            const float sgn = Q.won > 0.f ? 1.f : (Q.won < 0.f ? -1.f : 0.f);
            const float nwx = Q.n.x * sgn, nwy = Q.n.y * sgn, nwz = Q.n.z * sgn;
            const float* const lx = s_x[wave];
            const float* const ly = s_y[wave];
            const float* const lz = s_z[wave];
            const float4* const rec = &s_rec[wave][0][0];
            for (int i = 0; i < fcount; i += kLeavesPerStep) {
                const int n_in = stage(has, pp, pw);
                const int n4 = (n_in + 3) & ~3;
                has = i + kLeavesPerStep < fcount ? fetch(i + kLeavesPerStep, &pp, &pw) : false;
                staged_sum[jb] += (unsigned long long)n_in;
                // A: candidate c ends up at bit n4 - 1 - c of the 64-bit mask (m1 : m0)
                uint32_t m0 = 0, m1 = 0;
                for (int c = 0; c < n4; c += 4) {
                    const f4v X = *reinterpret_cast<const f4v*>(lx + c);
                    const f4v Y = *reinterpret_cast<const f4v*>(ly + c);
                    const f4v Z = *reinterpret_cast<const f4v*>(lz + c);
                    const f2v dx0 = X.xy + nqx, dy0 = Y.xy + nqy, dz0 = Z.xy + nqz;
                    const f2v dx1 = X.zw + nqx, dy1 = Y.zw + nqy, dz1 = Z.zw + nqz;
                    const f2v t0 = (dx0 * dx0 + dy0 * dy0 + dz0 * dz0) + nr2;   // negative: inside (the exact rule, strict)
                    const f2v t1 = (dx1 * dx1 + dy1 * dy1 + dz1 * dz1) + nr2;
                    m1 = __builtin_amdgcn_alignbit(m1, m0, 28);
                    m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(t0.x), 31);
                    m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(t0.y), 31);
                    m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(t1.x), 31);
                    m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(t1.y), 31);
                }
                int wr = cnt;
                cnt += __popc(m0) + __popc(m1);
                if (!WANT_IDX && cnt >= k) { m0 = 0; m1 = 0; }   // selected later or handed on: this sum is not needed
#ifdef TK_GG_TURNS
                {
                    const float c1 = (float)__popc(m1), c0 = (float)__popc(m0);
                    tn_half += (unsigned long long)(wave_max_all(n4 > 32 ? c1 : 0.f) + wave_max_all(c0));
                    tn_step += (unsigned long long)wave_max_all(c0 + c1);
                    tn_pair_acc += c0 + c1;
                    if (((i / kLeavesPerStep) & 1) || i + kLeavesPerStep >= fcount) { tn_pair += (unsigned long long)wave_max_all(tn_pair_acc); tn_pair_acc = 0.f; }
                    tn_group_acc += c0 + c1;
                    tn_steps += 1;
                }
#endif
                // B: the set bits of m1 (the first n4 - 32 candidates of a step of more than 32), then of
                // m0, highest first, i.e. in staging order.  A lane without bits left reads slot 0 and
                // adds nothing.
                auto drain = [&](uint32_t w, int n_half, int first) {
                    while (__any(w != 0)) {
                        const bool live = w != 0;
                        const int lzc = __clz(w);
                        w &= ~(0x80000000u >> (lzc & 31));
                        const int at = live ? first + lzc + n_half - 32 : 0;   // bit b is candidate n_half - 1 - b of this half
                        const float4 D = rec[2 * at];
                        const float2 Wp = *reinterpret_cast<const float2*>(&rec[2 * at + 1]);
                        const float cs = fmaf(D.x, nwx, fmaf(D.y, nwy, D.z * nwz));
                        const float wgt = live ? fmaxf(cs, 0.f) : 0.f;
                        sx = fmaf(D.w, wgt, sx); sy = fmaf(Wp.x, wgt, sy); sz = fmaf(Wp.y, wgt, sz);
                        if (WANT_IDX) {
                            if (live && wr < k) out.idx[q * k + wr] = __float_as_int(rec[2 * at + 1].z);
                            wr += live ? 1 : 0;
                        }
                    }
                };
                if (n4 > 32) { drain(m1, n4 - 32, 0); drain(m0, 32, n4 - 32); }
                else drain(m0, n4, 0);
            }
        }
        TK_STAMP(3)
#ifdef TK_GG_TURNS
        tn_group += (unsigned long long)wave_max_all(tn_group_acc);
        tn_pairs_total += (unsigned long long)wave_sum_all(tn_group_acc);
        tn_group_acc = 0.f; tn_pair_acc = 0.f;
        tn_groups += 1;
#endif
        // Lanes that met k photons or more need a selection.  When they are many (first iterations, dense maps)
        // the whole group is left to the selection kernel, which streams the same leaves again; when they are
        // few they go one by one to the per-query kernel.
        const bool streamed = mv.n > 0 && lox <= hix && !spread && fcount > 0;
        const bool flagged = streamed && Q.r2 > 0.f && cnt >= k;
        const bool defer_group = __popcll(__ballot(flagged)) >= sel_min_lanes;
        const bool deferred = defer_group && flagged;
        // a radius that was cut holds k photons or more; should a lane find fewer all the same, the
        // per-query kernel searches its full radius
        const bool over = valid && !deferred && (cnt >= k || (spread && Q.r2 > 0.f) || (shrunk && Q.r2 > 0.f));
        // Loads of the finished queries first, then the slot in the hand-over list (an atomic with a
        // result, which the wave waits for): the round trips run side by side instead of one after another.
        const bool fin = valid && !over && !deferred;
        float4 c4 = make_float4(0, 0, 0, 0), d4 = c4;
        float pl[5] = {0, 0, 0, 0, 0};
        const int64_t np = out.n_pixels;
        const int fl = out.which == 0 ? 6 : 3, pn = out.which == 0 ? 10 : 9, pr = out.which == 0 ? 12 : 11;
        if (fin) {
            c4 = qv.q2[q]; d4 = qv.q3[q];
            if (out.planes) {
                const float* P = out.planes;
                pl[0] = P[fl * np + pixel]; pl[1] = P[(fl + 1) * np + pixel]; pl[2] = P[(fl + 2) * np + pixel];
                pl[3] = P[pn * np + pixel]; pl[4] = P[pr * np + pixel];
            }
        }
        const int dst = wave_append(overflow_count, over);
        if (fin) {
            const F3 est = mul3(mk3(sx, sy, sz), mk3(d4.x, d4.y, d4.z));
            if (out.flux) { out.flux[3 * q] = est.x; out.flux[3 * q + 1] = est.y; out.flux[3 * q + 2] = est.z; }
            if (out.found) out.found[q] = cnt;
            if (out.r2out) out.r2out[q] = r2_in;
            if (out.planes) {
                const F3 lind = mul3(mk3(c4.x, c4.y, c4.z), est);
                pixel_update(&pl[0], &pl[1], &pl[2], &pl[3], &pl[4], lind, cnt, r2_in);
                float* P = out.planes;
                P[fl * np + pixel] = pl[0]; P[(fl + 1) * np + pixel] = pl[1]; P[(fl + 2) * np + pixel] = pl[2];
                P[pn * np + pixel] = pl[3]; P[pr * np + pixel] = pl[4];
            }
        }
        if (over) overflow[dst] = (int32_t)q;
        // the per-query kernel starts from this group's leaves instead of the root (a list it has room for)
        const bool with_leaves = !spread && mv.n > 0 && lox <= hix && fcount <= kMaxGroupLeaves;
        if (over) gl.query_group[dst] = with_leaves ? (int32_t)g : -1;
        if ((with_leaves && __any(over)) || defer_group) {
            for (int e = lane; e < fcount; e += 64) gl.leaves[g * kGroupLeafStride + e] = front[cur][e];
            if (lane == 0) gl.count[g] = fcount;
        }
        flagged_sum[jb] += __popcll(__ballot(valid && Q.r2 > 0.f && (cnt >= k || spread)));
        if (defer_group) {
            const int64_t slot = g * 64 + lane;
            if (slot < nq) sw.slot_cnt[slot] = deferred ? cnt : -1;
            if (lane == 0) sw.groups[atomicAdd(sw.count, 1)] = (int32_t)g;
        }
        cur_q = next_q; gi0 = gi1; gi1 = gi2;
        found_sum[jb] += (unsigned long long)wave_sum_all(fin ? (float)cnt : 0.f);
        TK_STAMP(4)
    }
    for (int jb = 0; jb < 2; ++jb) {
        const GroupJob& J = jb ? job1 : job0;
        if (J.out.photon_sum && lane == 0 && found_sum[jb]) atomicAdd(J.out.photon_sum, found_sum[jb]);
        if (J.out.cand_sum && lane == 0 && staged_sum[jb]) atomicAdd(J.out.cand_sum, staged_sum[jb]);
        if (J.sw.met_k && lane == 0 && flagged_sum[jb]) atomicAdd(J.sw.met_k, flagged_sum[jb]);
    }
#if defined(TK_GG_TURNS) || defined(TK_GG_STAMPS)
    const GatherOut& out = job0.out;   // the profiling builds below report through the first job
#endif
#ifdef TK_GG_TURNS
    if (out.photon_sum && lane == 0) {
        atomicAdd(out.photon_sum + 9, tn_half); atomicAdd(out.photon_sum + 10, tn_step); atomicAdd(out.photon_sum + 11, tn_pair);
        atomicAdd(out.photon_sum + 12, tn_group); atomicAdd(out.photon_sum + 13, tn_steps); atomicAdd(out.photon_sum + 14, tn_groups);
        atomicAdd(out.photon_sum + 15, tn_pairs_total);
    }
#endif
#ifdef TK_GG_STAMPS
    // cycles per section, summed over waves: [4] wait for the group index, [5] query load + union box,
    // [6] descent, [7] candidate steps, [8] finish + hand-over (profiling builds only)
    if (out.photon_sum && lane == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(out.photon_sum + 4 + i, st_sum[i]);
#endif
}

// ------------------------------- selection kernel (groups with many lanes at k or above)
// Works on the groups the grouped kernel has listed (SelectWork): the same 64 queries, the same leaves.
// Two passes over the leaves as a rule: a counting pass (photons per bin of the lane's window of dist2,
// photons below the window), a collecting pass (everything below the bin of the k-th is summed, the
// photons of that bin are stashed, at most kSelStash, and ranked); a bin with more photons is first
// refined by another counting pass, as long as enough lanes of the group need one.  Lanes that do not
// come to an end this way (a k-th outside the window, many photons at one distance) go to the
// per-query kernel like the handed-over lanes of the grouped kernel.
template <bool WANT_IDX>
__global__ void __launch_bounds__(256) k_gather_select(MapView mv, QueryView qv, SortedQueries qs, const uint32_t* order,
                                                       int64_t nq, GatherOut out, int32_t* overflow, int32_t* overflow_count,
                                                       GroupLeaves gl, SelectWork sw, int sel_min_lanes) {
    __shared__ float s_tab[1024];
    __shared__ __align__(16) float s_x[4][64];
    __shared__ __align__(16) float s_y[4][64];
    __shared__ __align__(16) float s_z[4][64];
    __shared__ float4 s_rec[4][64][2];   // (wi.xyz, power.r) (power.g, power.b, original index, position in the tree arrays)
    __shared__ int s_leaves[4][kGroupLeafStride];
    __shared__ uint32_t s_sel[4][kSelBins / 2][64];   // per lane (column): histogram of a counting pass, or the stash
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_tab[i] = mv.dirtab[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int leaf_slot = lane / kLeafCap, ph = lane % kLeafCap;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int k = mv.k;
    const int n_groups = *sw.count;
    int* const leaves = s_leaves[wave];
    uint32_t (*sel)[64] = s_sel[wave];
    const float* const lx = s_x[wave];
    const float* const ly = s_y[wave];
    const float* const lz = s_z[wave];
    const float4* const rec = &s_rec[wave][0][0];
    unsigned long long found_sum = 0, staged_sum = 0;
    for (int w = (int)blockIdx.x * 4 + wave; w < n_groups; w += (int)gridDim.x * 4) {
        const int64_t g = sw.groups[w];
        const int64_t slot = g * 64 + lane;
        const bool valid = slot < nq;
        float4 qa = make_float4(0, 0, 0, -1.f), qb = make_float4(0, 0, 0, 0);
        int64_t q = 0;
        int pixel = 0, cnt = -1;
        float r2s = -1.f;
        if (valid) {
            qa = qs.q0[slot]; qb = qs.q1[slot];
            q = order[slot];
            pixel = __float_as_int(qs.pixel[slot]);
            cnt = sw.slot_cnt[slot];
            r2s = qs.r2s[slot];
        }
        const F3 qp = mk3(qa.x, qa.y, qa.z);
        const int fcount = gl.count[g];
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < fcount; e += 64) leaves[e] = gl.leaves[g * kGroupLeafStride + e];
        __builtin_amdgcn_wave_barrier();
        const bool flagged = valid && cnt >= k && r2s > 0.f;
        // 1 counting into bins, 2 collecting, 3 given up (handed on), 4 done; a handful of groups would
        // keep a handful of waves busy for a long time: those lanes are handed on at once
        int sel_mode = flagged ? (n_groups < sw.min_groups ? 3 : 1) : 0;
        float sel_r2 = 0.f, ssx = 0, ssy = 0, ssz = 0;
        auto fetch = [&](int i, float4* pp, float4* pw, uint32_t* pat) {
            bool has = false;
            if (i + leaf_slot < fcount) {
                const int leaf = leaves[i + leaf_slot] - mv.nl;
                const int64_t s0 = leaf_start(leaf, mv.n, mv.depth), e0 = leaf_start(leaf + 1, mv.n, mv.depth);
                has = ph < (int)(e0 - s0);
                if (has) { *pp = mv.pos4[s0 + ph]; *pw = mv.pow4[s0 + ph]; *pat = (uint32_t)(s0 + ph); }
            }
            return has;
        };
        auto stage = [&](bool has, float4 pp, float4 pw, uint32_t pat, bool with_rec) {
            const unsigned long long staged = __ballot(has);
            const int n_in = __popcll(staged);
            const int n4 = (n_in + 3) & ~3;
            __builtin_amdgcn_wave_barrier();      // the previous step has been read by every lane
            if (has) {
                const int at = __popcll(staged & lt_mask);
                s_x[wave][at] = pp.x; s_y[wave][at] = pp.y; s_z[wave][at] = pp.z;
                if (with_rec) {
                    F3 wi = decode_dir(s_tab, __float_as_uint(pw.w));
                    s_rec[wave][at][0] = make_float4(wi.x, wi.y, wi.z, pw.x);
                    s_rec[wave][at][1] = make_float4(pw.y, pw.z, pp.w, __uint_as_float(pat));
                }
            } else {
                // the slots up to the next multiple of four hold a position no query accepts
                const int at = n_in + __popcll(~staged & lt_mask);
                if (at < n4) { s_x[wave][at] = INFINITY; s_y[wave][at] = 0.f; s_z[wave][at] = 0.f; }
            }
            __builtin_amdgcn_wave_barrier();
            return n_in;
        };
        const f2v nqx = {-qp.x, -qp.x}, nqy = {-qp.y, -qp.y}, nqz = {-qp.z, -qp.z};
        const float sgn = qb.w > 0.f ? 1.f : (qb.w < 0.f ? -1.f : 0.f);     // as in the grouped kernel
        const float nwx = qb.x * sgn, nwy = qb.y * sgn, nwz = qb.z * sgn;
        // First window: around the dist2 the k-th would have if the photons were spread evenly over the
        // disc, from 0.4 to 1.4 times that; what lies below the window is only counted.
        const float est = r2s * (float)k / (float)(cnt > 0 ? cnt : 1);
        const float w_hi = fminf(1.4f * est, r2s);
        const float w_lo = w_hi < r2s ? 0.4f * est : 0.f;
        // bin = int(dist2 * scale) - off.  A refinement multiplies scale by kSelBins (a power of two: the
        // product rounds the same way), so its bins nest exactly in the bin they refine.
        float scale = flagged ? ((float)kSelBins - 0.5f) / (w_hi - w_lo) : 1.f;
        int off = (int)(w_lo * scale), width = kSelBins, ns = 0, wr2 = 0;
        for (int pass = 0; pass < kSelMaxPasses; ++pass) {
            // (the lanes a third pass would be for are few as a rule: they go to the per-query kernel instead)
            const int n_active = __popcll(__ballot(sel_mode == 1 || sel_mode == 2));
            if (n_active == 0 || (pass >= 2 && n_active < sel_min_lanes)) break;
            const bool any_hist = __any(sel_mode == 1), any_coll = __any(sel_mode == 2);
            // candidates beyond this cannot fall into the window (and the bin number stays a small integer)
            const float lim = fminf(r2s, (float)(off + width) / scale * 1.000002f);
            const float lim_hist = sel_mode == 1 ? lim : -1.f, lim_coll = sel_mode == 2 ? lim : -1.f;
#pragma unroll
            for (int r = 0; r < kSelBins / 2; ++r) sel[r][lane] = 0u;
            ns = 0;
            int below = 0;        // candidates below the window (counting pass) / below the bin (collecting pass)
            float4 pp = make_float4(0, 0, 0, 0), pw = pp;
            uint32_t pat = 0;
            bool has = fetch(0, &pp, &pw, &pat);
            for (int i = 0; i < fcount; i += kLeavesPerStep) {
                const int n_in = stage(has, pp, pw, pat, any_coll);
                const int n4 = (n_in + 3) & ~3;
                has = i + kLeavesPerStep < fcount ? fetch(i + kLeavesPerStep, &pp, &pw, &pat) : false;
                staged_sum += (unsigned long long)n_in;
                for (int c = 0; c < n4; c += 4) {
                    const f4v X = *reinterpret_cast<const f4v*>(lx + c);
                    const f4v Y = *reinterpret_cast<const f4v*>(ly + c);
                    const f4v Z = *reinterpret_cast<const f4v*>(lz + c);
                    const f2v dx0 = X.xy + nqx, dy0 = Y.xy + nqy, dz0 = Z.xy + nqz;
                    const f2v dx1 = X.zw + nqx, dy1 = Y.zw + nqy, dz1 = Z.zw + nqz;
                    const f2v t0 = dx0 * dx0 + dy0 * dy0 + dz0 * dz0;   // dist2, rounded as everywhere
                    const f2v t1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1;
                    const float d2v[4] = {t0.x, t0.y, t1.x, t1.y};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float d2 = d2v[u];
                        const bool in_h = d2 < lim_hist, in_c = d2 < lim_coll;
                        const int rel = (in_h || in_c) ? (int)(d2 * scale) - off : 0x40000000;
                        below += rel < 0 ? 1 : 0;
                        if (any_hist && in_h && (unsigned)rel < (unsigned)kSelBins)
                            atomicAdd(&sel[rel >> 1][lane], 1u << ((rel & 1) * 16));   // the lane's own column
                        if (any_coll && __any(in_c)) {
                            const float4 D = rec[2 * (c + u)];
                            const float4 E = rec[2 * (c + u) + 1];
                            const bool inc = in_c && rel < 0;
                            const float cs = fmaf(D.x, nwx, fmaf(D.y, nwy, D.z * nwz));
                            const float wgt = inc ? fmaxf(cs, 0.f) : 0.f;
                            ssx = fmaf(D.w, wgt, ssx); ssy = fmaf(E.x, wgt, ssy); ssz = fmaf(E.y, wgt, ssz);
                            if (WANT_IDX && inc) { if (wr2 < k) out.idx[q * k + wr2] = __float_as_int(E.z); ++wr2; }
                            if (in_c && rel == 0) {
                                if (ns < kSelStash) sel[ns][lane] = __float_as_uint(E.w);
                                ++ns;
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (sel_mode == 1) {
                // the bin that holds the k-th nearest: the (k - below)-th photon of the window
                const int need = k - below;
                int cum = 0, bstar = -1, cstar = 0;
#pragma unroll
                for (int r = 0; r < kSelBins / 2; ++r) {
                    const uint32_t v = sel[r][lane];
                    const int c0 = (int)(v & 0xffffu), c1 = (int)(v >> 16);
                    if (bstar < 0) { if (cum + c0 >= need) { bstar = 2 * r; cstar = c0; } else cum += c0; }
                    if (bstar < 0) { if (cum + c1 >= need) { bstar = 2 * r + 1; cstar = c1; } else cum += c1; }
                }
                if (need < 1 || bstar < 0) sel_mode = 3;       // the k-th lies outside the window
                else if (cstar <= kSelStash) { off += bstar; width = 1; sel_mode = 2; }
                else {
                    const long long noff = ((long long)off + bstar) * kSelBins;
                    if (noff + kSelBins >= (1ll << 30)) sel_mode = 3;   // photons at one distance
                    else { off = (int)noff; scale *= (float)kSelBins; }
                }
            } else if (sel_mode == 2) {
                const int need = k - below;      // of the bin's photons
                if (ns > kSelStash || need < 1 || ns < need) sel_mode = 3;   // (the counting pass saw them: does not happen)
                else {
                    // the photons of the bin, ranked by (dist2, photon index); the need smallest are in
                    uint32_t pidx[kSelStash];
#pragma unroll
                    for (int e = 0; e < kSelStash; ++e) {
                        pidx[e] = 0xffffffffu;
                        if (e < ns) {
                            const float4 p4 = mv.pos4[sel[e][lane]];
                            sel[kSelStash + e][lane] = __float_as_uint(dist2_exact(p4.x, p4.y, p4.z, qp));
                            pidx[e] = __float_as_uint(p4.w);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < kSelStash; ++e) {
                        if (e < ns) {
                            const uint32_t de = sel[kSelStash + e][lane];
                            int rank = 0;
#pragma unroll
                            for (int j = 0; j < kSelStash; ++j) {
                                if (j < ns && j != e) {
                                    const uint32_t dj = sel[kSelStash + j][lane];
                                    rank += (dj < de || (dj == de && pidx[j] < pidx[e])) ? 1 : 0;
                                }
                            }
                            if (rank < need) {
                                const float4 w4 = mv.pow4[sel[e][lane]];
                                const F3 wi = decode_dir(s_tab, __float_as_uint(w4.w));
                                const float cs = fmaf(wi.x, nwx, fmaf(wi.y, nwy, wi.z * nwz));
                                const float wgt = fmaxf(cs, 0.f);
                                ssx = fmaf(w4.x, wgt, ssx); ssy = fmaf(w4.y, wgt, ssy); ssz = fmaf(w4.z, wgt, ssz);
                                sel_r2 = fmaxf(sel_r2, __uint_as_float(de));
                                if (WANT_IDX) { if (wr2 < k) out.idx[q * k + wr2] = (int32_t)pidx[e]; ++wr2; }
                            }
                        }
                    }
                    sel_mode = 4;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        const bool fin = sel_mode == 4;
        const bool over = flagged && !fin;
        float4 c4 = make_float4(0, 0, 0, 0), d4 = c4;
        float pl[5] = {0, 0, 0, 0, 0};
        const int64_t np = out.n_pixels;
        const int fl = out.which == 0 ? 6 : 3, pn = out.which == 0 ? 10 : 9, pr = out.which == 0 ? 12 : 11;
        if (fin) {
            c4 = qv.q2[q]; d4 = qv.q3[q];
            if (out.planes) {
                const float* P = out.planes;
                pl[0] = P[fl * np + pixel]; pl[1] = P[(fl + 1) * np + pixel]; pl[2] = P[(fl + 2) * np + pixel];
                pl[3] = P[pn * np + pixel]; pl[4] = P[pr * np + pixel];
            }
        }
        const int dst = wave_append(overflow_count, over);
        if (fin) {
            const F3 est3 = mul3(mk3(ssx, ssy, ssz), mk3(d4.x, d4.y, d4.z));
            if (out.flux) { out.flux[3 * q] = est3.x; out.flux[3 * q + 1] = est3.y; out.flux[3 * q + 2] = est3.z; }
            if (out.found) out.found[q] = k;
            if (out.r2out) out.r2out[q] = sel_r2;
            if (out.planes) {
                const F3 lind = mul3(mk3(c4.x, c4.y, c4.z), est3);
                pixel_update(&pl[0], &pl[1], &pl[2], &pl[3], &pl[4], lind, k, sel_r2);
                float* P = out.planes;
                P[fl * np + pixel] = pl[0]; P[(fl + 1) * np + pixel] = pl[1]; P[(fl + 2) * np + pixel] = pl[2];
                P[pn * np + pixel] = pl[3]; P[pr * np + pixel] = pl[4];
            }
        }
        if (over) {
            overflow[dst] = (int32_t)q;
            gl.query_group[dst] = fcount <= kMaxGroupLeaves ? (int32_t)g : -1;
        }
        found_sum += (unsigned long long)wave_sum_all(fin ? (float)k : 0.f);
    }
    if (out.photon_sum && lane == 0 && found_sum) atomicAdd(out.photon_sum, found_sum);
    if (out.cand_sum && lane == 0 && staged_sum) atomicAdd(out.cand_sum, staged_sum);
}

// Hilbert key of every query inside the root box of the tree.  Consecutive
// queries in this order are neighbours in space (a Hilbert curve has no jumps,
// unlike a Morton curve), so the search boxes of the 64 queries of a wave overlap.
// The order of the queries along the curve does not have to be exact: what the grouped kernel needs is that
// the 64 queries of a wave are neighbours in space.  So instead of a full sort the queries are dealt into
// 2^bits buckets by the leading bits of their curve index (a count, a prefix sum, a placement): the buckets
// are in curve order; inside a bucket (a handful of queries) the atomics' order is replaced by the order of
// (pixel, query number) - every query counts the smaller ones in its bucket - so the grouping is the same on
// every run even though the camera pass appends its queries in whatever order its waves finish: which kernel
// finishes a query, and with it the last bits of its flux sum, depends on its group.
constexpr int kQueryBucketBitsMax = 25;
constexpr int kQueryBucketRankCap = 4096;   // larger buckets (every query in one spot) keep the atomics' order
// Buckets of a gather of nq queries: about two per query (surfaces fill few of the cells), 2^bits of them.
static int query_bucket_bits(int64_t nq) {
    int bits = 6;
    while (bits < kQueryBucketBitsMax && ((int64_t)1 << bits) <= 2 * nq) ++bits;
    return bits;
}
// ... and a quarter as many behind them for the queries that cannot reach the map (k_query_bucket)
static size_t query_bucket_count(int64_t nq) {
    const int bits = query_bucket_bits(nq);
    return ((size_t)1 << bits) + ((size_t)1 << (bits - 2));
}
__global__ void k_query_bucket(QueryView qv, int64_t nq, const float4* root_box, int bits, int32_t* bucket_count,
                               uint32_t* bucket_of, uint32_t* rank_in) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float4 b0 = root_box[0], b1 = root_box[1];
    const float ex = b0.w - b0.x, ey = b1.x - b0.y, ez = b1.y - b0.z;
    float4 a = qv.q0[i];
    uint32_t b;
    if (sphere_misses_box(b0, b1, a.x, a.y, a.z, a.w)) {
        b = (1u << bits) + ((uint32_t)i & ((1u << (bits - 2)) - 1u));
    } else {
        float fx = fminf(fmaxf((a.x - b0.x) * (ex > 0 ? 1023.f / ex : 0.f), 0.f), 1023.f);
        float fy = fminf(fmaxf((a.y - b0.y) * (ey > 0 ? 1023.f / ey : 0.f), 0.f), 1023.f);
        float fz = fminf(fmaxf((a.z - b0.z) * (ez > 0 ? 1023.f / ez : 0.f), 0.f), 1023.f);
        b = hilbert30((uint32_t)fx, (uint32_t)fy, (uint32_t)fz) >> (30 - bits);
    }
    bucket_of[i] = b;
    rank_in[i] = (uint32_t)atomicAdd(&bucket_count[b], 1);
}
__global__ void k_query_place(QueryView qv, int64_t nq, const uint32_t* bucket_of, const uint32_t* rank_in,
                              const int32_t* bucket_start, uint32_t* placed_pixel, uint32_t* placed_index) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint32_t at = (uint32_t)bucket_start[bucket_of[i]] + rank_in[i];
    placed_pixel[at] = __float_as_uint(qv.q2[i].w);
    placed_index[at] = (uint32_t)i;
}
__global__ void k_query_rank(QueryView qv, int64_t nq, const uint32_t* bucket_of, const uint32_t* rank_in,
                             const int32_t* bucket_start, const uint32_t* placed_pixel, const uint32_t* placed_index,
                             uint32_t* order) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint32_t b = bucket_of[i];
    const int s0 = bucket_start[b], e0 = bucket_start[b + 1];
    if (e0 - s0 > kQueryBucketRankCap) {   // the bucket stays as the atomics placed it
        order[(uint32_t)s0 + rank_in[i]] = (uint32_t)i;
        return;
    }
    const uint32_t mine = __float_as_uint(qv.q2[i].w);
    int rank = 0;
    for (int p = s0; p < e0; ++p) {
        const uint32_t px = placed_pixel[p];
        if (px < mine) ++rank;
        else if (px == mine && placed_index[p] < (uint32_t)i) ++rank;   // stand-alone gathers: every query of pixel 0
    }
    order[s0 + rank] = (uint32_t)i;
}

// ------------------------------------------------------------ host side
int map_create(tkhip_ctx* ctx, int64_t capacity, int k, tkhip_map** out) {
    TK_REQUIRE(ctx && out, TKHIP_E_ARG, "map_create: null argument");
    TK_REQUIRE(capacity >= 0 && capacity < (int64_t)1 << 31, TKHIP_E_LIMIT, "map_create: capacity %lld out of range",
               (long long)capacity);
    TK_REQUIRE(k >= 1 && k <= TKHIP_MAX_K, TKHIP_E_LIMIT, "map_create: sample count %d outside 1..%d", k, TKHIP_MAX_K);
    tkhip_map* m = new tkhip_map();
    m->ctx = ctx;
    m->stream = ctx->stream;
    m->cap = capacity;
    m->k = k;
    size_t c = (size_t)(capacity > 0 ? capacity : 1);
    int rc = m->in_pos.ensure(c * 12);
    if (!rc) rc = m->in_pow.ensure(c * 12);
    if (!rc) rc = m->in_dir.ensure(c * 2);
    if (!rc) rc = m->counters.ensure(512);
    if (!rc) rc = m->totals.ensure(128);
    if (!rc && hipMemset(m->totals.p, 0, 128) != hipSuccess) { set_error("map_create: memset failed"); rc = TKHIP_E_HIP; }
    // where the device reports how many gathers of a launch met k photons (read when the next launch is set up)
    if (!rc && (hipHostMalloc((void**)&m->met_k_host, 64) != hipSuccess || hipEventCreate(&m->ev_met_k) != hipSuccess)) {
        set_error("map_create: pinned counter set-up failed");
        rc = TKHIP_E_HIP;
    }
    if (!rc && (hipEventCreate(&m->ev_begin) != hipSuccess || hipEventCreate(&m->ev_end) != hipSuccess)) {
        set_error("map_create: hipEventCreate failed");
        rc = TKHIP_E_HIP;
    }
    if (!rc) rc = upload_direction_tables(ctx);
    if (rc) { map_destroy(m); return rc; }
    *out = m;
    return TKHIP_OK;
}

void map_destroy(tkhip_map* m) {
    if (!m) return;
    DevBuf* bufs[] = {&m->in_pos, &m->in_pow, &m->in_dir, &m->pos4, &m->pow4, &m->node_box, &m->node_axis, &m->clip_box, &m->idx_a, &m->idx_b, &m->key_a, &m->key_b, &m->root_box, &m->q_order,
                      &m->q_keys_a, &m->q_keys_b, &m->q_vals_a, &m->q_sorted, &m->overflow, &m->counters, &m->prim_temp,
                      &m->overflow_group, &m->group_leaves, &m->group_leaf_count, &m->sel_groups, &m->sel_cnt, &m->q_r2s};
    for (DevBuf* b : bufs) b->release();
    m->totals.release();
    if (m->ev_begin) (void)hipEventDestroy(m->ev_begin);
    if (m->ev_end) (void)hipEventDestroy(m->ev_end);
    if (m->ev_met_k) (void)hipEventDestroy(m->ev_met_k);
    if (m->met_k_host) (void)hipHostFree(m->met_k_host);
    delete m;
}

int map_set_photons_device(tkhip_map* m, const float* pos, const float* power, const uint16_t* dir, int64_t n) {
    TK_REQUIRE(n >= 0 && n <= m->cap, TKHIP_E_LIMIT, "map_set_photons: %lld photons exceed capacity %lld", (long long)n,
               (long long)m->cap);
    hipStream_t s = m->stream;
    if (n > 0) {
        TK_HIP(hipMemcpyAsync(m->in_pos.p, pos, (size_t)n * 12, hipMemcpyDeviceToDevice, s));
        TK_HIP(hipMemcpyAsync(m->in_pow.p, power, (size_t)n * 12, hipMemcpyDeviceToDevice, s));
        TK_HIP(hipMemcpyAsync(m->in_dir.p, dir, (size_t)n * 2, hipMemcpyDeviceToDevice, s));
    }
    m->n = n;
    m->built = false;
    return TKHIP_OK;
}

int map_reserve_build(tkhip_map* m) {
    const int64_t n = std::max<int64_t>(m->cap, 1);
    int64_t want = (n + kLeafCap - 1) / kLeafCap;
    int nl = 1;
    while (nl < want) nl <<= 1;
    TK_TRY(m->pos4.ensure((size_t)n * 16));
    TK_TRY(m->pow4.ensure((size_t)n * 16));
    TK_TRY(m->node_box.ensure((size_t)nl * 2 * 32));
    TK_TRY(m->idx_a.ensure((size_t)n * 4));
    TK_TRY(m->idx_b.ensure((size_t)n * 4));
    TK_TRY(m->key_a.ensure((size_t)n * 4));
    TK_TRY(m->key_b.ensure((size_t)n * 4));
    TK_TRY(m->root_box.ensure(64));
    return sort_pairs_u32_reserve(m->prim_temp, (size_t)n, 30);
}

int map_build(tkhip_map* m) {
    hipStream_t s = m->stream;
    const int64_t n = m->n;
    m->built = false;
    if (n == 0) {
        // an empty map still answers gathers (nothing found): give it a degenerate root box
        m->nl = 0; m->depth = 0;
        TK_TRY(m->node_box.ensure(4 * 16));
        TK_HIP(hipMemsetAsync(m->node_box.p, 0, 4 * 16, s));
        m->built = true;
        return TKHIP_OK;
    }
    int64_t want = (n + kLeafCap - 1) / kLeafCap;
    int nl = 1, depth = 0;
    while (nl < want) { nl <<= 1; ++depth; }
    m->nl = nl;
    m->depth = depth;
    TK_TRY(m->pos4.ensure((size_t)n * 16));
    TK_TRY(m->pow4.ensure((size_t)n * 16));
    TK_TRY(m->node_box.ensure((size_t)nl * 2 * 32));
    TK_TRY(m->idx_a.ensure((size_t)n * 4));
    TK_TRY(m->idx_b.ensure((size_t)n * 4));
    TK_TRY(m->key_a.ensure((size_t)n * 4));
    TK_TRY(m->key_b.ensure((size_t)n * 4));
    TK_TRY(m->root_box.ensure(64));
    const float* pos = m->in_pos.as<float>();
    const int B = 256;
    uint32_t* box6 = m->root_box.as<uint32_t>();
    k_bounds_init<<<1, 64, 0, s>>>(box6);
    k_bounds<<<std::min(divup(n, B * 8), 256), B, 0, s>>>(pos, n, box6);
    if (m->tree_mode == 1 && depth > 0) {
        // median splits, one sort per level: the layout PhotonMap::balance produces (with the widest
        // axis the reference meant to take); some 6 launches per level
        TK_TRY(m->key_a.ensure((size_t)n * 8));
        TK_TRY(m->key_b.ensure((size_t)n * 8));
        TK_TRY(m->node_axis.ensure((size_t)nl * 2 * 4));
        TK_TRY(m->clip_box.ensure((size_t)nl * 2 * 24));
        uint32_t* idx = m->idx_a.as<uint32_t>();
        uint32_t* idx_alt = m->idx_b.as<uint32_t>();
        k_iota<<<divup(n, B), B, 0, s>>>(idx, n);
        k_root_clip<<<1, 64, 0, s>>>(box6, m->clip_box.as<float>());
        TK_LAUNCH_CHECK();
        for (int level = 0; level < depth; ++level) {
            const int nodes = 1 << level;
            k_level_axis<<<divup(nodes, B), B, 0, s>>>(m->clip_box.as<float>(), m->node_axis.as<int32_t>(), level);
            k_level_keys<<<divup(n, B), B, 0, s>>>(pos, idx, m->node_axis.as<int32_t>(), m->key_a.as<uint64_t>(), n, nl,
                                                   depth, level);
            TK_LAUNCH_CHECK();
            TK_TRY(sort_pairs_u64(s, m->prim_temp, m->key_a.as<uint64_t>(), m->key_b.as<uint64_t>(), idx, idx_alt, (size_t)n,
                                  32 + level));
            std::swap(idx, idx_alt);
            k_level_split<<<divup(nodes, B), B, 0, s>>>(pos, idx, m->clip_box.as<float>(), m->node_axis.as<int32_t>(), n,
                                                        depth, level);
            TK_LAUNCH_CHECK();
        }
        if (idx != m->idx_a.as<uint32_t>()) std::swap(m->idx_a, m->idx_b);   // idx_a: the leaf order
    } else {
        k_photon_curve<<<divup(n, B), B, 0, s>>>(pos, n, box6, m->key_a.as<uint32_t>(), m->idx_b.as<uint32_t>());
        TK_LAUNCH_CHECK();
        TK_TRY(sort_pairs_u32(s, m->prim_temp, m->key_a.as<uint32_t>(), m->key_b.as<uint32_t>(), m->idx_b.as<uint32_t>(),
                              m->idx_a.as<uint32_t>(), (size_t)n, 30));   // idx_a: the leaf order
    }
    k_permute<<<divup(n, B), B, 0, s>>>(pos, m->in_pow.as<float>(), m->in_dir.as<uint16_t>(), m->idx_a.as<uint32_t>(),
                                        m->pos4.as<float4>(), m->pow4.as<float4>(), n);
    k_leaf_boxes<<<divup((int64_t)nl * kLeafCap, 256), 256, 0, s>>>(m->pos4.as<float4>(), m->node_box.as<float4>(), n, nl, depth);
    int level = depth - 1;
    for (; level >= 0 && (1 << level) > kWideLevel; --level)
        k_inner_level<<<divup(1 << level, 256), 256, 0, s>>>(m->node_box.as<float4>(), level);
    if (level >= 0) k_inner_boxes<<<1, 1024, 0, s>>>(m->node_box.as<float4>(), level);
    TK_LAUNCH_CHECK();
    m->built = true;
    return TKHIP_OK;
}

}  // namespace tkd

tkd::MapView tkhip_map::view() const {
    tkd::MapView v;
    v.pos4 = pos4.as<float4>();
    v.pow4 = pow4.as<float4>();
    v.node_box = node_box.as<float4>();
    v.dirtab = ctx->b_dirtab.as<float>();
    v.keys = (tree_mode == 0 && n > 0) ? key_b.as<uint32_t>() : nullptr;
    v.n = (int32_t)n;
    v.nl = nl;
    v.depth = depth;
    v.k = k;
    return v;
}

namespace tkd {

template <int EPL>
static void launch_wave(tkhip_map* m, const QueryView& qv, const int32_t* list, const int32_t* list_count,
                        int64_t nq, int64_t work_hint, const GatherOut& out, const GroupLeaves& gl) {
    int blocks = (int)std::min<int64_t>(std::max<int64_t>(divup(work_hint, 4), 1), 256 * 16);
    k_gather_wave<EPL><<<blocks, 256, 0, m->stream>>>(m->view(), qv, list, list_count, nq, out, gl);
}

int map_reserve_queries(tkhip_map* m, int64_t nq) {
    TK_TRY(m->q_keys_a.ensure((size_t)nq * 4));
    TK_TRY(m->q_keys_b.ensure((size_t)nq * 4));
    TK_TRY(m->q_vals_a.ensure((size_t)nq * 4));
    TK_TRY(m->q_order.ensure((size_t)nq * 4));
    TK_TRY(m->q_sorted.ensure((size_t)nq * 32));
    TK_TRY(m->overflow.ensure((size_t)nq * 4));
    TK_TRY(m->overflow_group.ensure((size_t)nq * 4));
    TK_TRY(m->group_leaves.ensure((size_t)((nq + 63) / 64) * kGroupLeafStride * 4));
    TK_TRY(m->group_leaf_count.ensure((size_t)((nq + 63) / 64) * 4));
    TK_TRY(m->sel_groups.ensure((size_t)((nq + 63) / 64) * 4));
    TK_TRY(m->sel_cnt.ensure((size_t)nq * 4));
    TK_TRY(m->q_r2s.ensure((size_t)nq * 4));
    const size_t nb = query_bucket_count(nq);
    size_t scan_bytes = 0;
    TK_TRY(exclusive_scan_i32_temp_bytes(nb + 1, &scan_bytes));
    TK_TRY(m->prim_temp.ensure(2 * ((nb + 1) * 4 + 256) + scan_bytes + 256));
    return TKHIP_OK;
}

int map_flush_timing(tkhip_map* m) {
    if (!m->ev_pending) return TKHIP_OK;
    float ms = 0.f;
    TK_HIP(hipEventSynchronize(m->ev_end));
    TK_HIP(hipEventElapsedTime(&ms, m->ev_begin, m->ev_end));
    m->group_ms_total += ms;
    m->group_launches += 1;
    m->ev_pending = false;
    return TKHIP_OK;
}

// The scratch of a production gather of nq queries on a map's own buffers, and the decision the map's last
// launch leaves for this one.
int map_gather_scratch(tkhip_map* m, int64_t nq, GatherScratch* out_sc) {
    GatherScratch sc{};
    TK_TRY(map_reserve_queries(m, nq));
    sc.q_keys_a = m->q_keys_a.as<uint32_t>(); sc.q_keys_b = m->q_keys_b.as<uint32_t>();
    sc.q_vals_a = m->q_vals_a.as<uint32_t>(); sc.q_order = m->q_order.as<uint32_t>();
    sc.overflow = m->overflow.as<int32_t>(); sc.overflow_group = m->overflow_group.as<int32_t>();
    sc.group_leaves = m->group_leaves.as<int32_t>(); sc.group_leaf_count = m->group_leaf_count.as<int32_t>();
    sc.sel_groups = m->sel_groups.as<int32_t>(); sc.sel_cnt = m->sel_cnt.as<int32_t>(); sc.q_r2s = m->q_r2s.as<float>();
    // Many gathers of the map's last launch met k photons (or there was none yet): radii are cut first and long
    // leaf lists taken.  The count of that launch has arrived long ago; the wait is for form.
    static const int heavy_pct_env = getenv("TKHIP_HEAVY_PCT") ? atoi(getenv("TKHIP_HEAVY_PCT")) : 0;
    const int heavy_pct = heavy_pct_env > 0 ? heavy_pct_env : (m->n > kSelLargeMap ? kSelHeavyPercentLarge : kSelHeavyPercent);
    sc.heavy = 1;
    if (m->met_k_pending) {
        TK_HIP(hipEventSynchronize(m->ev_met_k));
        sc.heavy = (int64_t)m->met_k_host[0] * 100 >= m->met_k_queries * heavy_pct ? 1 : 0;
    }
    sc.track = 1;
    sc.counter = m->counters.as<int32_t>();
    sc.q_sorted = m->q_sorted.as<float4>();
    sc.sort_temp = m->prim_temp.p; sc.sort_temp_bytes = m->prim_temp.bytes;
    *out_sc = sc;
    return TKHIP_OK;
}

int map_gather_device(tkhip_map* m, int64_t nq, const QueryView& qv, const GatherOut& out, int mode, bool timed) {
    TK_REQUIRE(m->built, TKHIP_E_STATE, "gather: the KD-tree has not been built");
    if (nq == 0) return TKHIP_OK;
    TK_REQUIRE(nq < (int64_t)1 << 31, TKHIP_E_LIMIT, "gather: too many queries");
    GatherScratch sc{};
    if (mode == 0) TK_TRY(map_gather_scratch(m, nq, &sc));
    return map_gather_in(m, nq, qv, out, mode, timed, sc);
}

// Blocks of the grouped kernel that one GPU holds at once (its waves fetch their groups from a cursor).
static int group_grid_blocks(int device) {
    static std::mutex mu;
    static int cached[64] = {0};
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 64) return 256 * 8;
    if (cached[device] == 0) {
        hipDeviceProp_t prop{};
        int per_cu = 0;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gather_group<false>, 256, 0) != hipSuccess || per_cu < 1)
            cached[device] = 256 * 8;
        else
            cached[device] = std::max(prop.multiProcessorCount, 1) * per_cu;
    }
    return cached[device];
}

static int select_grid_blocks(int device) {
    static std::mutex mu;
    static int cached[64] = {0};
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 64) return 256 * 4;
    if (cached[device] == 0) {
        hipDeviceProp_t prop{};
        int per_cu = 0;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gather_select<false>, 256, 0) != hipSuccess || per_cu < 1)
            cached[device] = 256 * 4;
        else
            cached[device] = std::max(prop.multiProcessorCount, 1) * per_cu;
    }
    return cached[device];
}

// A gather in three parts, so that the grouped kernel of two maps can be one launch:
//   gather_prepare  (the map's stream)  query order along the curve, radii cut where they hold many times k photons;
//   gather_group    (one stream)        the grouped kernel over the jobs of one or two maps;
//   gather_finish   (the map's stream)  the selection kernel (while many gathers meet k photons) and the
//                                       per-query kernel for what was handed on.
static int gather_prepare(tkhip_map* m, int64_t nq, const QueryView& qv, const GatherOut& out, const GatherScratch& sc,
                          GroupJob* job) {
    hipStream_t s = m->stream;
    // Curve order inside the root box of the tree (heap node 1), to the bucket
    const int bits = query_bucket_bits(nq);
    const size_t nb = query_bucket_count(nq);
    const size_t arr = ((nb + 1) * 4 + 255) & ~(size_t)255;
    TK_REQUIRE(sc.sort_temp_bytes > 2 * arr, TKHIP_E_LIMIT, "gather: %zu bytes of scratch for the query buckets, %zu given", 2 * arr,
               sc.sort_temp_bytes);
    int32_t* bucket_count = static_cast<int32_t*>(sc.sort_temp);
    int32_t* bucket_start = reinterpret_cast<int32_t*>(static_cast<char*>(sc.sort_temp) + arr);
    TK_HIP(hipMemsetAsync(bucket_count, 0, (nb + 1) * 4, s));
    k_query_bucket<<<divup(nq, 256), 256, 0, s>>>(qv, nq, m->node_box.as<float4>() + 2, bits, bucket_count, sc.q_keys_a,
                                                  sc.q_vals_a);
    TK_LAUNCH_CHECK();
    TK_TRY(exclusive_scan_i32_in(s, static_cast<char*>(sc.sort_temp) + 2 * arr, sc.sort_temp_bytes - 2 * arr, bucket_count,
                                 bucket_start, nb + 1));
    uint32_t* placed_index = reinterpret_cast<uint32_t*>(sc.q_sorted);   // free until k_sort_queries fills it
    k_query_place<<<divup(nq, 256), 256, 0, s>>>(qv, nq, sc.q_keys_a, sc.q_vals_a, bucket_start, sc.q_keys_b, placed_index);
    k_query_rank<<<divup(nq, 256), 256, 0, s>>>(qv, nq, sc.q_keys_a, sc.q_vals_a, bucket_start, sc.q_keys_b, placed_index,
                                                sc.q_order);
    TK_LAUNCH_CHECK();
    k_sort_queries<<<divup(nq, 256), 256, 0, s>>>(qv, sc.q_order, nq, sc.q_sorted, sc.q_sorted + nq, reinterpret_cast<float*>(sc.q_keys_a),
                                                  sc.q_r2s);
    TK_LAUNCH_CHECK();
    // radii far larger than the photon spacing are cut first; level: the deepest whose nodes all hold k photons
    // or more (a node of level L holds at least n >> L)
    if (sc.heavy && m->n >= m->k) {
        int level = 0;
        while (level < m->depth && (m->n >> (level + 1)) >= m->k) ++level;
        level = std::max(level - 1, 0);   // one level higher (2 k photons): the k nearest of a larger node are nearer
        k_query_bound<<<divup(nq, 256), 256, 0, s>>>(m->view(), nq, sc.q_sorted, sc.q_r2s, level);
        TK_LAUNCH_CHECK();
    }
    int32_t* ocount = sc.counter;          // the hand-over count
    int32_t* scount = sc.counter + 16;     // groups left to the selection kernel
    TK_HIP(hipMemsetAsync(ocount, 0, 128, s));
    // (the two thresholds can be forced from the environment: tests run every gather through the selection kernel)
    static const int sel_min_env = getenv("TKHIP_SEL_MIN") ? atoi(getenv("TKHIP_SEL_MIN")) : 0;
    // (while few gathers meet k photons no group is left to the selection kernel: it is not even launched)
    const int sel_min = !sc.heavy ? 65 : sel_min_env > 0 ? sel_min_env : (m->n > kSelLargeMap ? kSelMinLanesLarge : kSelMinLanes);
    static const int sel_groups = getenv("TKHIP_SEL_GROUPS") ? atoi(getenv("TKHIP_SEL_GROUPS")) : kSelMinGroups;
    job->mv = m->view();
    job->qv = qv;
    job->qs = SortedQueries{sc.q_sorted, sc.q_sorted + nq, reinterpret_cast<const float*>(sc.q_keys_a), sc.q_r2s};   // the keys are done with
    job->order = sc.q_order;
    job->nq = nq;
    job->out = out;
    job->overflow = sc.overflow;
    job->overflow_count = ocount;
    job->gl = GroupLeaves{sc.overflow_group, sc.group_leaves, sc.group_leaf_count};
    job->sw = SelectWork{sc.sel_groups, scount, sc.sel_cnt, sc.track ? sc.counter + 24 : nullptr, sel_groups, sc.heavy};
    job->sel_min_lanes = sel_min;
    return TKHIP_OK;
}

// The grouped kernel over one or two jobs on `s`, bracketed by the events of `mt` (whose running totals take
// the launch) unless untimed.  The grid: as many waves as the GPU holds at once (fewer waves with the same
// number of groups each were no faster: 0.24 against 0.23 ms per launch).
static int gather_group(tkhip_map* mt, hipStream_t s, const GroupJob* jobs, int n_jobs, bool untimed) {
    const int64_t g0 = (jobs[0].nq + 63) / 64, g1 = n_jobs > 1 ? (jobs[1].nq + 63) / 64 : 0;
    const int64_t groups = g0 + g1;
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(divup(groups, 4), 1), group_grid_blocks(mt->ctx->device));
    if (!untimed) {
        TK_TRY(map_flush_timing(mt));
        TK_HIP(hipEventRecord(mt->ev_begin, s));
    }
    const GroupJob& second = jobs[n_jobs > 1 ? 1 : 0];
    if (jobs[0].out.idx || second.out.idx) k_gather_group<true><<<blocks, 256, 0, s>>>(jobs[0], second, g0, groups);
    else k_gather_group<false><<<blocks, 256, 0, s>>>(jobs[0], second, g0, groups);
    TK_LAUNCH_CHECK();
    if (!untimed) {
        TK_HIP(hipEventRecord(mt->ev_end, s));
        mt->ev_pending = true;
        mt->group_queries += jobs[0].nq + (n_jobs > 1 ? jobs[1].nq : 0);
    }
    return TKHIP_OK;
}

static int gather_finish(tkhip_map* m, int64_t nq, const QueryView& qv, const GatherOut& out, bool timed, const GatherScratch& sc,
                         const GroupJob& job) {
    hipStream_t s = m->stream;
    const int epl = m->k <= 64 ? 2 : (m->k <= 192 ? 4 : 8);
    const int64_t groups = (nq + 63) / 64;
    // the groups with many lanes at k or above (their number is known to the device only: the waves of a
    // grid that fills the GPU once share them out)
    const int sel_blocks = (int)std::min<int64_t>(std::max<int64_t>(divup(groups, 4), 1), select_grid_blocks(m->ctx->device));
    if (sc.heavy) {
        if (out.idx)
            k_gather_select<true><<<sel_blocks, 256, 0, s>>>(job.mv, qv, job.qs, job.order, nq, out, job.overflow, job.overflow_count, job.gl, job.sw, job.sel_min_lanes);
        else
            k_gather_select<false><<<sel_blocks, 256, 0, s>>>(job.mv, qv, job.qs, job.order, nq, out, job.overflow, job.overflow_count, job.gl, job.sw, job.sel_min_lanes);
        TK_LAUNCH_CHECK();
    }
    if (epl == 2) launch_wave<2>(m, qv, job.overflow, job.overflow_count, nq, nq, out, job.gl);
    else if (epl == 4) launch_wave<4>(m, qv, job.overflow, job.overflow_count, nq, nq, out, job.gl);
    else launch_wave<8>(m, qv, job.overflow, job.overflow_count, nq, nq, out, job.gl);
    TK_LAUNCH_CHECK();
    if (sc.track) {
        TK_HIP(hipMemcpyAsync(m->met_k_host, sc.counter + 24, 4, hipMemcpyDeviceToHost, s));
        TK_HIP(hipEventRecord(m->ev_met_k, s));
        m->met_k_pending = true;
        m->met_k_queries = nq;
    }
    if (timed) {
        int32_t oc = 0;
        TK_HIP(hipMemcpyAsync(&oc, job.overflow_count, 4, hipMemcpyDeviceToHost, s));
        TK_HIP(hipStreamSynchronize(s));
        m->last_overflow = oc;
    }
    return TKHIP_OK;
}

int map_gather_in(tkhip_map* m, int64_t nq, const QueryView& qv, const GatherOut& out, int mode, bool timed,
                  const GatherScratch& sc) {
    TK_REQUIRE(m->built, TKHIP_E_STATE, "gather: the KD-tree has not been built");
    if (nq == 0) return TKHIP_OK;
    TK_REQUIRE(nq < (int64_t)1 << 31, TKHIP_E_LIMIT, "gather: too many queries");
    m->last_group_ms = 0.f;
    m->last_overflow = 0;
    if (mode == 1) {
        const int epl = m->k <= 64 ? 2 : (m->k <= 192 ? 4 : 8);
        const GroupLeaves none{nullptr, nullptr, nullptr};
        if (epl == 2) launch_wave<2>(m, qv, nullptr, nullptr, nq, nq, out, none);
        else if (epl == 4) launch_wave<4>(m, qv, nullptr, nullptr, nq, nq, out, none);
        else launch_wave<8>(m, qv, nullptr, nullptr, nq, nq, out, none);
        TK_LAUNCH_CHECK();
        return TKHIP_OK;
    }
    GroupJob job{};
    TK_TRY(gather_prepare(m, nq, qv, out, sc, &job));
    TK_TRY(gather_group(m, m->stream, &job, 1, sc.untimed != 0));
    if (timed && !sc.untimed) {
        TK_HIP(hipEventSynchronize(m->ev_end));
        TK_HIP(hipEventElapsedTime(&m->last_group_ms, m->ev_begin, m->ev_end));
    }
    return gather_finish(m, nq, qv, out, timed, sc, job);
}

// The gathers of the two maps of a render with ONE launch of the grouped kernel: both maps' queries are put in
// order on their own streams, the launch (on the first map's stream, timed under the first map) waits for both,
// and each map's selection and per-query kernels follow on its own stream again.
int map_gather_pair(tkhip_map* m0, int64_t nq0, const QueryView& qv0, const GatherOut& out0, tkhip_map* m1, int64_t nq1,
                    const QueryView& qv1, const GatherOut& out1, bool timed) {
    TK_REQUIRE(m0->built && m1->built, TKHIP_E_STATE, "gather: the KD-trees have not been built");
    TK_REQUIRE(nq0 > 0 && nq1 > 0 && nq0 < (int64_t)1 << 31 && nq1 < (int64_t)1 << 31, TKHIP_E_ARG, "gather: bad query counts");
    tkhip_map* ms[2] = {m0, m1};
    const int64_t nqs[2] = {nq0, nq1};
    const QueryView* qvs[2] = {&qv0, &qv1};
    const GatherOut* outs[2] = {&out0, &out1};
    GatherScratch sc[2];
    GroupJob jobs[2];
    for (int w = 1; w >= 0; --w) {
        ms[w]->last_group_ms = 0.f;
        ms[w]->last_overflow = 0;
        TK_TRY(map_gather_scratch(ms[w], nqs[w], &sc[w]));
        TK_TRY(gather_prepare(ms[w], nqs[w], *qvs[w], *outs[w], sc[w], &jobs[w]));
    }
    // The two streams are ordered through the host: it waits for the second map's query order (the first map's
    // is being made meanwhile), enqueues the launch and the first map's remaining kernels behind it, and gives
    // the second map's remaining kernels to their stream once the launch has ended.  (Event waits between the
    // two streams instead were seen to stall for good when several renders share a GPU: the streams of a
    // process outnumber the hardware queues then.)
    TK_HIP(hipStreamSynchronize(m1->stream));
    TK_TRY(gather_group(m0, m0->stream, jobs, 2, false));
    TK_TRY(gather_finish(m0, nq0, qv0, out0, false, sc[0], jobs[0]));
    TK_HIP(hipEventSynchronize(m0->ev_end));
    if (timed) TK_HIP(hipEventElapsedTime(&m0->last_group_ms, m0->ev_begin, m0->ev_end));
    TK_TRY(gather_finish(m1, nq1, qv1, out1, timed, sc[1], jobs[1]));
    if (timed) {
        int32_t oc = 0;
        TK_HIP(hipMemcpyAsync(&oc, jobs[0].overflow_count, 4, hipMemcpyDeviceToHost, m0->stream));
        TK_HIP(hipStreamSynchronize(m0->stream));
        m0->last_overflow = oc;
    }
    return TKHIP_OK;
}

__global__ void k_pack_store(const float* pos, const float* pow, const uint8_t* th, const uint8_t* ph, float* dpos,
                             float* dpow, uint16_t* ddir, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int a = 0; a < 3; ++a) { dpos[3 * i + a] = pos[3 * i + a]; dpow[3 * i + a] = pow[3 * i + a]; }
    ddir[i] = (uint16_t)th[i] | ((uint16_t)ph[i] << 8);
}

__global__ void k_pack_queries(const float* p, const float* n, const float* wo, const float* R, const float* r2,
                               float4* q0, float4* q1, float4* q2, float4* q3, int64_t nq) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    F3 nn = ld3(n + 3 * i), w = ld3(wo + 3 * i);
    q0[i] = make_float4(p[3 * i], p[3 * i + 1], p[3 * i + 2], r2[i]);
    q1[i] = make_float4(nn.x, nn.y, nn.z, dot3(w, nn));
    q2[i] = make_float4(1.f, 1.f, 1.f, __int_as_float(0));
    q3[i] = make_float4(R[3 * i] * kInvPi, R[3 * i + 1] * kInvPi, R[3 * i + 2] * kInvPi, 0.f);
}

__global__ void k_pixel_update(int64_t n, float* flux, float* photons, float* radius2, const float* lindir,
                               const int32_t* found, const float* r2ret) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    pixel_update(&flux[3 * i], &flux[3 * i + 1], &flux[3 * i + 2], &photons[i], &radius2[i], ld3(lindir + 3 * i),
                 found[i], r2ret[i]);
}

}  // namespace tkd


// ------------------------------------------------- gather without allocation
// Scratch of tkhip_map_gather_static: one array per GPU, part of the library's
// code object (each GPU the library runs on holds its copy until the process
// ends).  Host code never uses the array's name as an address: the address on a
// device is looked up once (static_scratch_init, called by tkhip_create with
// that device current) and kept per device.
static __device__ __align__(256) unsigned char g_static_scratch[TKHIP_STATIC_SCRATCH_BYTES];
static std::mutex g_static_mutex;
static unsigned char* g_static_ptr[64] = {nullptr};

namespace tkd {
int static_scratch_init(int device) {
    TK_REQUIRE(device >= 0 && device < 64, TKHIP_E_LIMIT, "static scratch: device %d outside 0..63", device);
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (g_static_ptr[device]) return TKHIP_OK;
    void* ptr = nullptr;
    TK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_static_scratch)));
    TK_REQUIRE(ptr != nullptr, TKHIP_E_HIP, "static scratch: no device address on device %d", device);
    g_static_ptr[device] = static_cast<unsigned char*>(ptr);
    return TKHIP_OK;
}
}  // namespace tkd

// ------------------------------------------------------------------ C ABI
using namespace tkd;

extern "C" {

int tkhip_map_create(tkhip_ctx* ctx, int64_t capacity, int k, tkhip_map** out) {
    TK_REQUIRE(ctx && out, TKHIP_E_ARG, "tkhip_map_create: null argument");
    TK_HIP(hipSetDevice(ctx->device));
    return map_create(ctx, capacity, k, out);
}

void tkhip_map_destroy(tkhip_map* map) {
    if (!map) return;
    (void)hipSetDevice(map->ctx->device);
    (void)hipStreamSynchronize(map->stream);
    map_destroy(map);
}

int tkhip_map_store(tkhip_map* m, const float* pos, const float* power, const uint8_t* theta, const uint8_t* phi,
                    int64_t n) {
    TK_REQUIRE(m && (n == 0 || (pos && power && theta && phi)), TKHIP_E_ARG, "tkhip_map_store: null argument");
    TK_REQUIRE(n >= 0 && m->n + n <= m->cap, TKHIP_E_LIMIT, "tkhip_map_store: %lld + %lld photons exceed capacity %lld",
               (long long)m->n, (long long)n, (long long)m->cap);
    if (n == 0) return TKHIP_OK;
    tkhip_ctx* ctx = m->ctx;
    TK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = m->stream;
    TK_TRY(ctx->scratch_a.ensure((size_t)n * 12));
    TK_TRY(ctx->scratch_b.ensure((size_t)n * 12));
    TK_TRY(ctx->scratch_c.ensure((size_t)n * 2));
    TK_HIP(hipMemcpyAsync(ctx->scratch_a.p, pos, (size_t)n * 12, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(ctx->scratch_b.p, power, (size_t)n * 12, hipMemcpyHostToDevice, s));
    uint8_t* dth = ctx->scratch_c.as<uint8_t>();
    TK_HIP(hipMemcpyAsync(dth, theta, (size_t)n, hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(dth + n, phi, (size_t)n, hipMemcpyHostToDevice, s));
    k_pack_store<<<divup(n, 256), 256, 0, s>>>(ctx->scratch_a.as<float>(), ctx->scratch_b.as<float>(), dth, dth + n,
                                               m->in_pos.as<float>() + 3 * m->n, m->in_pow.as<float>() + 3 * m->n,
                                               m->in_dir.as<uint16_t>() + m->n, n);
    TK_LAUNCH_CHECK();
    TK_HIP(hipStreamSynchronize(s));
    m->n += n;
    m->built = false;
    return TKHIP_OK;
}

int tkhip_map_set_tree(tkhip_map* m, int mode) {
    TK_REQUIRE(m != nullptr, TKHIP_E_ARG, "tkhip_map_set_tree: null map");
    TK_REQUIRE(mode == 0 || mode == 1, TKHIP_E_ARG, "tkhip_map_set_tree: mode must be 0 (curve order) or 1 (median splits)");
    m->tree_mode = mode;
    m->built = false;
    return TKHIP_OK;
}

int tkhip_map_build(tkhip_map* m) {
    TK_REQUIRE(m, TKHIP_E_ARG, "tkhip_map_build: null map");
    TK_HIP(hipSetDevice(m->ctx->device));
    TK_TRY(map_build(m));
    TK_HIP(hipStreamSynchronize(m->stream));
    return TKHIP_OK;
}

int tkhip_map_tree_info(tkhip_map* m, int32_t* n_leaves, int32_t* leaf_capacity, int32_t* depth, int32_t* perm) {
    TK_REQUIRE(m && m->built, TKHIP_E_STATE, "tkhip_map_tree_info: map not built");
    if (n_leaves) *n_leaves = m->nl;
    if (leaf_capacity) *leaf_capacity = kLeafCap;
    if (depth) *depth = m->depth;
    if (perm && m->n > 0) {
        TK_HIP(hipSetDevice(m->ctx->device));
        TK_HIP(hipMemcpy(perm, m->idx_a.p, (size_t)m->n * 4, hipMemcpyDeviceToHost));
    }
    return TKHIP_OK;
}

int tkhip_map_gather(tkhip_map* m, int64_t nq, const float* p, const float* n, const float* wo, const float* R,
                     const float* radius2, float* flux, int32_t* found, float* radius2_out, int32_t* idx, int mode) {
    TK_REQUIRE(m && (nq == 0 || (p && n && wo && R && radius2)), TKHIP_E_ARG, "tkhip_map_gather: null argument");
    TK_REQUIRE(mode == 0 || mode == 1, TKHIP_E_ARG, "tkhip_map_gather: mode must be 0 or 1");
    TK_REQUIRE(m->built, TKHIP_E_STATE, "tkhip_map_gather: call tkhip_map_build first");
    if (nq == 0) return TKHIP_OK;
    tkhip_ctx* ctx = m->ctx;
    TK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = m->stream;
    const size_t q = (size_t)nq;
    DevBuf in, qb, ob;
    int rc = in.ensure(q * 52);
    if (!rc) rc = qb.ensure(q * 64);
    if (!rc) rc = ob.ensure(q * (12 + 4 + 4) + (idx ? q * (size_t)m->k * 4 : 0) + 64);
    auto cleanup = [&]() { in.release(); qb.release(); ob.release(); };
    if (rc) { cleanup(); return rc; }
    float* d_in = in.as<float>();
    float *dp = d_in, *dn = d_in + 3 * q, *dw = d_in + 6 * q, *dR = d_in + 9 * q, *dr2 = d_in + 12 * q;
    hipError_t e = hipMemcpyAsync(dp, p, q * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dn, n, q * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dw, wo, q * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dR, R, q * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dr2, radius2, q * 4, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) { set_error("tkhip_map_gather: upload failed: %s", hipGetErrorString(e)); cleanup(); return TKHIP_E_HIP; }
    float4* q4 = qb.as<float4>();
    QueryView qv{q4, q4 + q, q4 + 2 * q, q4 + 3 * q};
    k_pack_queries<<<divup(nq, 256), 256, 0, s>>>(dp, dn, dw, dR, dr2, q4, q4 + q, q4 + 2 * q, q4 + 3 * q, nq);
    GatherOut out{};
    char* o = ob.as<char>();
    out.flux = (float*)o;
    out.found = (int32_t*)(o + q * 12);
    out.r2out = (float*)(o + q * 16);
    out.idx = idx ? (int32_t*)(o + q * 20) : nullptr;
    if (idx) (void)hipMemsetAsync(out.idx, 0xff, q * (size_t)m->k * 4, s);
    rc = map_gather_device(m, nq, qv, out, mode, false);
    if (!rc) {
        e = hipStreamSynchronize(s);
        if (e == hipSuccess && flux) e = hipMemcpy(flux, out.flux, q * 12, hipMemcpyDeviceToHost);
        if (e == hipSuccess && found) e = hipMemcpy(found, out.found, q * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && radius2_out) e = hipMemcpy(radius2_out, out.r2out, q * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && idx) e = hipMemcpy(idx, out.idx, q * (size_t)m->k * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("tkhip_map_gather: %s", hipGetErrorString(e)); rc = TKHIP_E_HIP; }
    }
    (void)hipStreamSynchronize(s);
    cleanup();
    return rc;
}

int tkhip_map_gather_static(tkhip_map* m, int64_t nq, const float* p, const float* n, const float* wo, const float* R,
                            const float* radius2, float* flux, int32_t* found, float* radius2_out, int32_t* idx) {
    TK_REQUIRE(m && (nq == 0 || (p && n && wo && R && radius2 && flux && found && radius2_out)), TKHIP_E_ARG,
               "tkhip_map_gather_static: null argument");
    TK_REQUIRE(m->built, TKHIP_E_STATE, "tkhip_map_gather_static: call tkhip_map_build first");
    TK_REQUIRE(nq >= 0 && nq <= TKHIP_STATIC_MAX_QUERIES, TKHIP_E_LIMIT,
               "tkhip_map_gather_static: %lld queries, the static scratch holds at most %d", (long long)nq,
               TKHIP_STATIC_MAX_QUERIES);
    if (nq == 0) return TKHIP_OK;
    tkhip_ctx* ctx = m->ctx;
    TK_HIP(hipSetDevice(ctx->device));
    unsigned char* base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_static_mutex);
        base = ctx->device >= 0 && ctx->device < 64 ? g_static_ptr[ctx->device] : nullptr;
    }
    TK_REQUIRE(base != nullptr, TKHIP_E_STATE, "tkhip_map_gather_static: no static scratch on device %d", ctx->device);
    // carve the array: everything sized for the maximum, so the layout does not depend on nq
    const size_t Q = (size_t)TKHIP_STATIC_MAX_QUERIES;
    size_t at = 0;
    auto take = [&](size_t bytes) { unsigned char* r = base + at; at += (bytes + 255) & ~(size_t)255; return r; };
    float4* q4 = reinterpret_cast<float4*>(take(Q * 64));
    GatherScratch sc{};
    sc.q_keys_a = reinterpret_cast<uint32_t*>(take(Q * 4));
    sc.q_keys_b = reinterpret_cast<uint32_t*>(take(Q * 4));
    sc.q_vals_a = reinterpret_cast<uint32_t*>(take(Q * 4));
    sc.q_order = reinterpret_cast<uint32_t*>(take(Q * 4));
    sc.q_sorted = reinterpret_cast<float4*>(take(Q * 32));
    sc.overflow = reinterpret_cast<int32_t*>(take(Q * 4));
    sc.overflow_group = reinterpret_cast<int32_t*>(take(Q * 4));
    sc.group_leaves = reinterpret_cast<int32_t*>(take((Q / 64) * kGroupLeafStride * 4));
    sc.group_leaf_count = reinterpret_cast<int32_t*>(take((Q / 64) * 4));
    sc.sel_groups = reinterpret_cast<int32_t*>(take((Q / 64) * 4));
    sc.sel_cnt = reinterpret_cast<int32_t*>(take(Q * 4));
    sc.q_r2s = reinterpret_cast<float*>(take(Q * 4));
    sc.heavy = 1;      // the array is shared by every map of the GPU: no history, the radii are always bounded
    sc.track = 0;
    sc.untimed = 1;    // the map's running totals (tkhip_sppm_gather_totals) are those of its render's gathers
    sc.counter = reinterpret_cast<int32_t*>(take(512));
    TK_REQUIRE(at < (size_t)TKHIP_STATIC_SCRATCH_BYTES, TKHIP_E_LIMIT, "tkhip_map_gather_static: scratch layout exceeds the array");
    sc.sort_temp = base + at;
    sc.sort_temp_bytes = (size_t)TKHIP_STATIC_SCRATCH_BYTES - at;
    hipStream_t s = m->stream;
    const size_t q = (size_t)nq;
    QueryView qv{q4, q4 + q, q4 + 2 * q, q4 + 3 * q};
    k_pack_queries<<<divup(nq, 256), 256, 0, s>>>(p, n, wo, R, radius2, q4, q4 + q, q4 + 2 * q, q4 + 3 * q, nq);
    TK_LAUNCH_CHECK();
    GatherOut out{};
    out.flux = flux;
    out.found = found;
    out.r2out = radius2_out;
    out.idx = idx;
    if (idx) TK_HIP(hipMemsetAsync(idx, 0xff, q * (size_t)m->k * 4, s));
    TK_TRY(map_gather_in(m, nq, qv, out, 0, false, sc));
    TK_HIP(hipStreamSynchronize(s));
    return TKHIP_OK;
}

int tkhip_pixel_update(tkhip_ctx* ctx, int64_t n, float* flux, float* photons, float* radius2, const float* lindir,
                       const int32_t* found, const float* radius2_ret) {
    TK_REQUIRE(ctx && (n == 0 || (flux && photons && radius2 && lindir && found && radius2_ret)), TKHIP_E_ARG,
               "tkhip_pixel_update: null argument");
    if (n == 0) return TKHIP_OK;
    TK_HIP(hipSetDevice(ctx->device));
    const size_t q = (size_t)n;
    DevBuf b;
    TK_TRY(b.ensure(q * (12 + 4 + 4 + 12 + 4 + 4)));
    float* d = b.as<float>();
    float *dflux = d, *dph = d + 3 * q, *dr2 = d + 4 * q, *dl = d + 5 * q, *drr = d + 8 * q;
    int32_t* dfound = (int32_t*)(d + 9 * q);
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemcpyAsync(dflux, flux, q * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dph, photons, q * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dr2, radius2, q * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dl, lindir, q * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(drr, radius2_ret, q * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dfound, found, q * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        k_pixel_update<<<divup(n, 256), 256, 0, s>>>(n, dflux, dph, dr2, dl, dfound, drr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(flux, dflux, q * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(photons, dph, q * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(radius2, dr2, q * 4, hipMemcpyDeviceToHost);
    b.release();
    if (e != hipSuccess) { set_error("tkhip_pixel_update: %s", hipGetErrorString(e)); return TKHIP_E_HIP; }
    return TKHIP_OK;
}

}  // extern "C"
