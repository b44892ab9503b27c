// bvh_build_device.hip — the scene BVH built on the GPU.
//
// The CPU renderer builds its hierarchies on one host thread by recursive SAH
// partitioning over buckets (src/BVH.cpp:62-243, one BVH per mesh from
// src/Triangle.cpp:103-180).  Here the same recursion runs level by level, every
// node of a level at once, and no primitive is touched by the host.  A node is
// a run [a, b) of a primitive array; one level is
//
//   k_clear / k_cbounds   extent of the box centres of every run (integer atomics on
//                         order-preserving float keys, one per wave where a wave lies
//                         inside one run);
//   k_bins                sixteen buckets along the widest axis of that extent, each a
//                         count and a box (LDS buckets for a block that lies inside
//                         one run, global atomics otherwise);
//   k_eval                per run: cheapest bucket boundary by the surface-area heuristic,
//                         or "leaf" for a run of up to kMaxLeaf primitives that is cheaper
//                         to test than to split: the decisions of the host builder
//                         (bvh_build.hip), taken with the same arithmetic;
//   k_emit / k_next       node numbers (a prefix sum, so the numbering is the order of the
//                         runs), the 64-byte node with both child boxes (unions of buckets),
//                         the link from the parent, the runs of the next level;
//   k_flag / k_scatter    stable partition of every run about its boundary (one prefix sum
//                         over the primitive array), carrying box and id along.
//
// Min, max and integer sums do not depend on the order of the atomics, so the
// tree is the same on every run.  Depth is bounded as in the host builder: from
// a depth limit on runs are cut in the middle, and a tree that still comes out
// too deep for the traversal stack is rebuilt from middle cuts only.
#include <cmath>
#include <cstdlib>

#include "tk_common.h"

namespace tkd {

namespace {

constexpr int kBins = 16;
constexpr int kBinWords = 7;                    // count, ordered min xyz, ordered max xyz
constexpr int kRunWords = kBins * kBinWords;    // 112 words of buckets per run

struct Run {
    int32_t a, b;         // slots [a, b)
    int32_t parent, side; // node that links to this run, 0 left / 1 right; parent -1: the root
};

struct Eval {             // what k_eval decided for a run
    float cmin, k1;       // bucket of a centre c: int(k1 * (c[axis] - cmin)), clamped
    int32_t axis;
    int32_t best;         // primitives of buckets <= best go left
    int32_t lc;           // how many those are
    int32_t interior;     // 0: the run is a leaf
    int32_t next_l, next_r;   // runs of the children on the next level, -1 for a leaf child
    float lbox[6], rbox[6];
};

struct Rec {              // a primitive while it is being moved about
    float4 lo;            // box min, w = primitive id (bits)
    float4 hi;            // box max
};

// floats as unsigned integers that sort the same way
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// The box the host builder gives a primitive: the vertices' extent, or the sphere's padded by a
// few ulps because the hit point is re-projected onto the sphere.
__global__ void k_prim_recs(const TkTriangle* tris, int n_tris, const TkSphere* spheres, int n, Rec* recs, int32_t* run_of) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rec r;
    if (i < n_tris) {
        const TkTriangle& t = tris[i];
        r.lo.x = fminf(fminf(t.v0[0], t.v1[0]), t.v2[0]); r.hi.x = fmaxf(fmaxf(t.v0[0], t.v1[0]), t.v2[0]);
        r.lo.y = fminf(fminf(t.v0[1], t.v1[1]), t.v2[1]); r.hi.y = fmaxf(fmaxf(t.v0[1], t.v1[1]), t.v2[1]);
        r.lo.z = fminf(fminf(t.v0[2], t.v1[2]), t.v2[2]); r.hi.z = fmaxf(fmaxf(t.v0[2], t.v1[2]), t.v2[2]);
    } else {
        const TkSphere& s = spheres[i - n_tris];
        float rad = s.radius * 1.00001f;
        r.lo.x = s.center[0] - rad; r.hi.x = s.center[0] + rad;
        r.lo.y = s.center[1] - rad; r.hi.y = s.center[1] + rad;
        r.lo.z = s.center[2] - rad; r.hi.z = s.center[2] + rad;
    }
    r.lo.w = __int_as_float(i);
    r.hi.w = 0.f;
    recs[i] = r;
    run_of[i] = 0;
}

__device__ __forceinline__ void centre_of(const Rec& r, float c[3]) {
    c[0] = 0.5f * r.lo.x + 0.5f * r.hi.x;
    c[1] = 0.5f * r.lo.y + 0.5f * r.hi.y;
    c[2] = 0.5f * r.lo.z + 0.5f * r.hi.z;
}

__global__ void k_clear(int count, uint32_t* cb, uint32_t* bins) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count * 6) cb[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
    if (i < count * kRunWords) {
        int w = i % kBinWords;
        bins[i] = w == 0 ? 0u : (w < 4 ? 0xffffffffu : 0u);
    }
}

__global__ void k_cbounds(int n, const int32_t* run_of, const Rec* recs, uint32_t* cb) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = i < n ? run_of[i] : -1;
    float c[3] = {0, 0, 0};
    if (j >= 0) centre_of(recs[i], c);
    const int j0 = __shfl(j, 0);
    if (__all(j == j0)) {
        if (j0 < 0) return;
        float lo[3] = {c[0], c[1], c[2]}, hi[3] = {c[0], c[1], c[2]};
        for (int a = 0; a < 3; ++a)
            for (int off = 32; off > 0; off >>= 1) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
            }
        if ((threadIdx.x & 63) == 0)
            for (int a = 0; a < 3; ++a) {
                atomicMin(&cb[j0 * 6 + a], f2ord(lo[a]));
                atomicMax(&cb[j0 * 6 + 3 + a], f2ord(hi[a]));
            }
    } else if (j >= 0) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&cb[j * 6 + a], f2ord(c[a]));
            atomicMax(&cb[j * 6 + 3 + a], f2ord(c[a]));
        }
    }
}

// How a run is cut into buckets: along the widest axis of its centres (the first of equals), or,
// when the centres coincide or the depth limit says so, into a lower and an upper half by position.
struct BinRule {
    float cmin, k1, ext;
    int axis;
    bool by_position;
};
__device__ __forceinline__ BinRule bin_rule(const uint32_t* cb6, bool force_middle) {
    float mn[3] = {ord2f(cb6[0]), ord2f(cb6[1]), ord2f(cb6[2])};
    float ext[3] = {ord2f(cb6[3]) - mn[0], ord2f(cb6[4]) - mn[1], ord2f(cb6[5]) - mn[2]};
    BinRule r;
    r.axis = 0;
    if (ext[1] > ext[r.axis]) r.axis = 1;
    if (ext[2] > ext[r.axis]) r.axis = 2;
    r.ext = ext[r.axis];
    r.cmin = mn[r.axis];
    r.by_position = force_middle || !(r.ext > 0.f);
    r.k1 = r.by_position ? 0.f : kBins * (1.0f - 1e-6f) / r.ext;
    return r;
}
__device__ __forceinline__ int bin_of(const BinRule& r, const Run& run, int slot, const float c[3]) {
    if (r.by_position) return slot - run.a < (run.b - run.a) / 2 ? 0 : kBins - 1;
    int bi = (int)(r.k1 * (c[r.axis] - r.cmin));
    return min(kBins - 1, max(0, bi));
}

__global__ void __launch_bounds__(256)
k_bins(int n, const int32_t* run_of, const Rec* recs, const Run* runs, const uint32_t* cb, int force_middle, uint32_t* bins) {
    __shared__ uint32_t s_bins[kRunWords];
    __shared__ int s_one_run;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x == 0) {
        int first = run_of[blockIdx.x * blockDim.x];
        int last_slot = min(n, (int)((blockIdx.x + 1) * blockDim.x)) - 1;
        s_one_run = first >= 0 && run_of[last_slot] == first ? 1 : 0;   // runs are contiguous: equal ends, equal between
    }
    if (threadIdx.x < kRunWords) {
        int w = threadIdx.x % kBinWords;
        s_bins[threadIdx.x] = w == 0 ? 0u : (w < 4 ? 0xffffffffu : 0u);
    }
    __syncthreads();
    const bool local = s_one_run != 0;
    const int j = i < n ? run_of[i] : -1;
    if (j >= 0) {
        const Rec rec = recs[i];
        float c[3];
        centre_of(rec, c);
        const Run run = runs[j];
        const BinRule rule = bin_rule(cb + j * 6, force_middle != 0);
        const int bi = bin_of(rule, run, i, c);
        uint32_t* b = local ? s_bins + bi * kBinWords : bins + (size_t)j * kRunWords + bi * kBinWords;
        atomicAdd(&b[0], 1u);
        atomicMin(&b[1], f2ord(rec.lo.x)); atomicMin(&b[2], f2ord(rec.lo.y)); atomicMin(&b[3], f2ord(rec.lo.z));
        atomicMax(&b[4], f2ord(rec.hi.x)); atomicMax(&b[5], f2ord(rec.hi.y)); atomicMax(&b[6], f2ord(rec.hi.z));
    }
    if (!local) return;
    __syncthreads();
    if (threadIdx.x < kRunWords) {
        const int jr = run_of[blockIdx.x * blockDim.x];
        const int w = threadIdx.x % kBinWords;
        const uint32_t v = s_bins[threadIdx.x];
        uint32_t* g = bins + (size_t)jr * kRunWords + threadIdx.x;
        if (w == 0) { if (v) atomicAdd(g, v); }
        else if (w < 4) { if (v != 0xffffffffu) atomicMin(g, v); }
        else if (v != 0u) atomicMax(g, v);
    }
}

struct HBox {
    float mn[3], mx[3];
    __device__ void reset() { for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; } }
    __device__ void grow(const HBox& o) { for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], o.mn[a]); mx[a] = fmaxf(mx[a], o.mx[a]); } }
    __device__ double area() const {
        double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0;
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
};

// One thread per run.  A leaf links itself into its parent here; an interior run is numbered by k_emit.
__global__ void k_eval(int count, const Run* runs, const uint32_t* cb, const uint32_t* bins, int force_middle, int max_leaf,
                       BvhNode* nodes, Eval* evals, int32_t* interior, int32_t* bad) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const Run run = runs[r];
    const int n = run.b - run.a;
    const BinRule rule = bin_rule(cb + r * 6, force_middle != 0);
    const uint32_t* bw = bins + (size_t)r * kRunWords;
    int cnt[kBins];
    HBox bb[kBins];
    for (int i = 0; i < kBins; ++i) {
        cnt[i] = (int)bw[i * kBinWords];
        for (int a = 0; a < 3; ++a) {
            bb[i].mn[a] = cnt[i] ? ord2f(bw[i * kBinWords + 1 + a]) : INFINITY;
            bb[i].mx[a] = cnt[i] ? ord2f(bw[i * kBinWords + 4 + a]) : -INFINITY;
        }
    }
    double la[kBins - 1], ra[kBins - 1];
    int lc[kBins - 1], rc[kBins - 1];
    HBox acc;
    acc.reset();
    int c = 0;
    for (int i = 0; i < kBins - 1; ++i) { acc.grow(bb[i]); c += cnt[i]; la[i] = acc.area(); lc[i] = c; }
    acc.reset();
    c = 0;
    for (int i = kBins - 1; i > 0; --i) { acc.grow(bb[i]); c += cnt[i]; ra[i - 1] = acc.area(); rc[i - 1] = c; }
    acc.grow(bb[0]);   // acc: the run's own box
    double best = INFINITY;
    int best_i = -1;
    for (int i = 0; i < kBins - 1; ++i) {
        if (lc[i] == 0 || rc[i] == 0) continue;
        double cost = lc[i] * la[i] + rc[i] * ra[i];
        if (cost < best) { best = cost; best_i = i; }
    }
    bool leaf = false;
    if (n <= max_leaf) {
        if (rule.by_position) leaf = true;   // coincident centres, or below the depth limit
        else {
            const double c_trav = 1.0, c_isect = 1.5;
            double parent_area = fmax(acc.area(), 1e-30);
            double split_cost = c_trav + c_isect * best / parent_area;
            leaf = best_i < 0 || split_cost >= c_isect * n;
        }
    }
    Eval e;
    e.cmin = rule.cmin; e.k1 = rule.k1; e.axis = rule.axis;
    e.best = best_i; e.lc = best_i >= 0 ? lc[best_i] : 0;
    e.interior = leaf ? 0 : 1;
    e.next_l = e.next_r = -1;
    HBox L, R;
    L.reset(); R.reset();
    for (int i = 0; i < kBins; ++i) (i <= best_i ? L : R).grow(bb[i]);
    for (int a = 0; a < 3; ++a) { e.lbox[a] = L.mn[a]; e.lbox[3 + a] = L.mx[a]; e.rbox[a] = R.mn[a]; e.rbox[3 + a] = R.mx[a]; }
    evals[r] = e;
    interior[r] = leaf ? 0 : 1;
    if (!leaf && best_i < 0) atomicAdd(bad, 1);   // cannot happen: two centres that differ fill two buckets
    if (leaf) {
        const int32_t code = ~(run.a * 8 + (n - 1));
        if (run.side == 0) nodes[run.parent].left = code;
        else nodes[run.parent].right = code;
    }
}

__global__ void k_emit(int count, const Run* runs, Eval* evals, const int32_t* interior, const int32_t* rank, int base,
                       BvhNode* nodes, int32_t* kids, int32_t* totals) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    if (r == count - 1) totals[0] = rank[r] + interior[r];   // interior runs of this level
    if (!interior[r]) {
        kids[r] = 0;
        return;
    }
    const Run run = runs[r];
    const Eval e = evals[r];
    const int id = base + rank[r];
    const int lc = e.lc, rc = (run.b - run.a) - e.lc;
    BvhNode o;
    for (int a = 0; a < 3; ++a) {
        o.lmin[a] = e.lbox[a]; o.lmax[a] = e.lbox[3 + a];
        o.rmin[a] = e.rbox[a]; o.rmax[a] = e.rbox[3 + a];
    }
    o.left = lc == 1 ? ~(run.a * 8) : 0;                 // longer runs link themselves on the next level
    o.right = rc == 1 ? ~((run.a + lc) * 8) : 0;
    o.pad0 = o.pad1 = 0;
    nodes[id] = o;
    if (run.parent >= 0) {
        if (run.side == 0) nodes[run.parent].left = id;
        else nodes[run.parent].right = id;
    }
    kids[r] = (lc >= 2 ? 1 : 0) + (rc >= 2 ? 1 : 0);
    evals[r].interior = id + 1;   // from here on: node number + 1
}

__global__ void k_next(int count, const Run* runs, Eval* evals, const int32_t* kids, const int32_t* kid_rank, Run* next_runs,
                       int32_t* totals) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    if (r == count - 1) totals[1] = kid_rank[r] + kids[r];   // runs of the next level
    const int id1 = evals[r].interior;
    if (id1 == 0) return;
    const Run run = runs[r];
    const int lc = evals[r].lc, rc = (run.b - run.a) - lc;
    int at = kid_rank[r];
    if (lc >= 2) {
        next_runs[at] = Run{run.a, run.a + lc, id1 - 1, 0};
        evals[r].next_l = at++;
    }
    if (rc >= 2) {
        next_runs[at] = Run{run.a + lc, run.b, id1 - 1, 1};
        evals[r].next_r = at;
    }
}

__global__ void k_flag(int n, const int32_t* run_of, const Rec* recs, const Run* runs, const Eval* evals, int force_middle,
                       int32_t* flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = run_of[i];
    int left = 0;
    if (j >= 0 && evals[j].interior) {
        const Eval& e = evals[j];
        BinRule rule;
        rule.cmin = e.cmin; rule.k1 = e.k1; rule.axis = e.axis; rule.ext = 0.f;
        rule.by_position = e.k1 == 0.f;
        float c[3];
        centre_of(recs[i], c);
        left = bin_of(rule, runs[j], i, c) <= e.best ? 1 : 0;
    }
    flags[i] = left;
}

// Stable partition of every interior run about its boundary; everything else stays where it is.
__global__ void k_scatter(int n, const int32_t* run_of, const Rec* recs, const Run* runs, const Eval* evals, const int32_t* flags,
                          const int32_t* before, Rec* recs_out, int32_t* run_of_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = run_of[i];
    int dst = i, next = -1;
    if (j >= 0 && evals[j].interior) {
        const Run run = runs[j];
        const Eval& e = evals[j];
        const int lrank = before[i] - before[run.a];
        if (flags[i]) { dst = run.a + lrank; next = e.next_l; }
        else { dst = run.a + e.lc + (i - run.a - lrank); next = e.next_r; }
    }
    recs_out[dst] = recs[i];
    run_of_out[dst] = next;
}

__global__ void k_make_isect(const TkTriangle* tris, int n_tris, int n, const Rec* recs, TriIsect* isect) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int id = __float_as_int(recs[s].lo.w);
    TriIsect t;
    memset(&t, 0, sizeof(t));
    t.prim = id;
    if (id < n_tris) {
        const TkTriangle& T = tris[id];
        for (int a = 0; a < 3; ++a) {
            t.v0[a] = T.v0[a];
            t.e1[a] = T.v1[a] - T.v0[a];   // src/Triangle.hpp:21-22
            t.e2[a] = T.v2[a] - T.v0[a];
        }
    }
    isect[s] = t;
}

}  // namespace

// Builds the tree over `n_tris` triangles followed by `n_spheres` spheres (device arrays) into
// `nodes` (BvhNode) and `isect` (TriIsect in leaf order).  Needs more than max_leaf primitives (the
// root is a node).  *depth is the depth of the deepest leaf, which the caller checks against the
// traversal stack.
int build_bvh_device(hipStream_t stream, DevBuf& prim_temp, const TkTriangle* d_tris, int n_tris, const TkSphere* d_spheres,
                     int n_spheres, int max_leaf, int max_stack_depth, DevBuf& nodes, DevBuf& isect, int* n_nodes, int* depth,
                     float bmin[3], float bmax[3]) {
    const int n = n_tris + n_spheres;
    TK_REQUIRE(n > max_leaf && n >= 2, TKHIP_E_ARG, "build_bvh_device: needs more than %d primitives", max_leaf);
    const int max_runs = n / 2 + 1;   // a run holds at least two primitives
    DevBuf b_recs_a, b_recs_b, b_of_a, b_of_b, b_runs_a, b_runs_b, b_cb, b_bins, b_evals, b_int, b_rank, b_kids, b_krank, b_flags,
        b_before, b_small;
    DevBuf* all[] = {&b_recs_a, &b_recs_b, &b_of_a, &b_of_b, &b_runs_a, &b_runs_b, &b_cb, &b_bins, &b_evals, &b_int, &b_rank,
                     &b_kids, &b_krank, &b_flags, &b_before, &b_small};
    struct Cleanup {
        DevBuf** bufs;
        size_t count;
        ~Cleanup() { for (size_t i = 0; i < count; ++i) bufs[i]->release(); }
    } cleanup{all, sizeof(all) / sizeof(all[0])};
    TK_TRY(b_recs_a.ensure((size_t)n * sizeof(Rec)));
    TK_TRY(b_recs_b.ensure((size_t)n * sizeof(Rec)));
    TK_TRY(b_of_a.ensure((size_t)n * 4));
    TK_TRY(b_of_b.ensure((size_t)n * 4));
    TK_TRY(b_runs_a.ensure((size_t)max_runs * sizeof(Run)));
    TK_TRY(b_runs_b.ensure((size_t)max_runs * sizeof(Run)));
    TK_TRY(b_cb.ensure((size_t)max_runs * 6 * 4));
    TK_TRY(b_bins.ensure((size_t)max_runs * kRunWords * 4));
    TK_TRY(b_evals.ensure((size_t)max_runs * sizeof(Eval)));
    TK_TRY(b_int.ensure((size_t)max_runs * 4));
    TK_TRY(b_rank.ensure((size_t)max_runs * 4));
    TK_TRY(b_kids.ensure((size_t)max_runs * 4));
    TK_TRY(b_krank.ensure((size_t)max_runs * 4));
    TK_TRY(b_flags.ensure((size_t)n * 4));
    TK_TRY(b_before.ensure((size_t)n * 4));
    TK_TRY(b_small.ensure(64));
    TK_TRY(nodes.ensure((size_t)n * sizeof(BvhNode)));
    TK_TRY(isect.ensure((size_t)n * sizeof(TriIsect)));
    int32_t* totals = b_small.as<int32_t>();       // [0] interior runs, [1] next runs, [2] impossible cases
    const int depth_limit = max_stack_depth - 6;    // from here down: middle cuts
    Rec* recs = nullptr;

    for (int attempt = 0; attempt < 2; ++attempt) {
        recs = b_recs_a.as<Rec>();
        Rec* recs_other = b_recs_b.as<Rec>();
        int32_t* run_of = b_of_a.as<int32_t>();
        int32_t* run_of_other = b_of_b.as<int32_t>();
        Run* runs = b_runs_a.as<Run>();
        Run* runs_next = b_runs_b.as<Run>();
        k_prim_recs<<<divup(n, 256), 256, 0, stream>>>(d_tris, n_tris, d_spheres, n, recs, run_of);
        TK_LAUNCH_CHECK();
        const Run root{0, n, -1, 0};
        TK_HIP(hipMemcpyAsync(runs, &root, sizeof(root), hipMemcpyHostToDevice, stream));
        TK_HIP(hipMemsetAsync(totals, 0, 16, stream));
        int count = 1, base = 0, level = 0;
        while (count > 0) {
            TK_REQUIRE(count <= max_runs && base + count <= n, TKHIP_E_HIP, "build_bvh_device: node count overran the primitive count");
            const int middle = attempt == 1 || level >= depth_limit;
            k_clear<<<divup((int64_t)count * kRunWords, 256), 256, 0, stream>>>(count, b_cb.as<uint32_t>(), b_bins.as<uint32_t>());
            TK_LAUNCH_CHECK();
            k_cbounds<<<divup(n, 256), 256, 0, stream>>>(n, run_of, recs, b_cb.as<uint32_t>());
            TK_LAUNCH_CHECK();
            k_bins<<<divup(n, 256), 256, 0, stream>>>(n, run_of, recs, runs, b_cb.as<uint32_t>(), middle, b_bins.as<uint32_t>());
            TK_LAUNCH_CHECK();
            k_eval<<<divup(count, 64), 64, 0, stream>>>(count, runs, b_cb.as<uint32_t>(), b_bins.as<uint32_t>(), middle, max_leaf,
                                                        nodes.as<BvhNode>(), b_evals.as<Eval>(), b_int.as<int32_t>(), totals + 2);
            TK_LAUNCH_CHECK();
            TK_TRY(exclusive_scan_i32(stream, prim_temp, b_int.as<int32_t>(), b_rank.as<int32_t>(), (size_t)count));
            k_emit<<<divup(count, 256), 256, 0, stream>>>(count, runs, b_evals.as<Eval>(), b_int.as<int32_t>(), b_rank.as<int32_t>(),
                                                          base, nodes.as<BvhNode>(), b_kids.as<int32_t>(), totals);
            TK_LAUNCH_CHECK();
            TK_TRY(exclusive_scan_i32(stream, prim_temp, b_kids.as<int32_t>(), b_krank.as<int32_t>(), (size_t)count));
            k_next<<<divup(count, 256), 256, 0, stream>>>(count, runs, b_evals.as<Eval>(), b_kids.as<int32_t>(),
                                                          b_krank.as<int32_t>(), runs_next, totals);
            TK_LAUNCH_CHECK();
            k_flag<<<divup(n, 256), 256, 0, stream>>>(n, run_of, recs, runs, b_evals.as<Eval>(), middle, b_flags.as<int32_t>());
            TK_LAUNCH_CHECK();
            TK_TRY(exclusive_scan_i32(stream, prim_temp, b_flags.as<int32_t>(), b_before.as<int32_t>(), (size_t)n));
            k_scatter<<<divup(n, 256), 256, 0, stream>>>(n, run_of, recs, runs, b_evals.as<Eval>(), b_flags.as<int32_t>(),
                                                         b_before.as<int32_t>(), recs_other, run_of_other);
            TK_LAUNCH_CHECK();
            int32_t got[3] = {0, 0, 0};
            TK_HIP(hipMemcpyAsync(got, totals, sizeof(got), hipMemcpyDeviceToHost, stream));
            TK_HIP(hipStreamSynchronize(stream));
            TK_REQUIRE(got[2] == 0, TKHIP_E_HIP, "build_bvh_device: a run with distinct centres filled one bucket only");
            base += got[0];
            count = got[1];
            std::swap(recs, recs_other);
            std::swap(run_of, run_of_other);
            std::swap(runs, runs_next);
            ++level;
            if (count > 0 && level + 1 >= max_stack_depth - 1) break;   // too deep: rebuild from middle cuts
        }
        *n_nodes = base;
        *depth = level;   // leaves hang off the last level's nodes
        if (count == 0) break;
        TK_REQUIRE(attempt == 0, TKHIP_E_LIMIT, "build_bvh_device: %d primitives need a tree deeper than the traversal stack", n);
    }
    k_make_isect<<<divup(n, 256), 256, 0, stream>>>(d_tris, n_tris, n, recs, isect.as<TriIsect>());
    TK_LAUNCH_CHECK();
    // scene bounds: the two child boxes of the root
    BvhNode root_node;
    TK_HIP(hipMemcpyAsync(&root_node, nodes.p, sizeof(root_node), hipMemcpyDeviceToHost, stream));
    TK_HIP(hipStreamSynchronize(stream));
    for (int a = 0; a < 3; ++a) {
        bmin[a] = std::min(root_node.lmin[a], root_node.rmin[a]);
        bmax[a] = std::max(root_node.lmax[a], root_node.rmax[a]);
    }
    return TKHIP_OK;
}

}  // namespace tkd
