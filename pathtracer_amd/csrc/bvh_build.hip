// bvh_build.hip — host-side construction of the flat scene BVH.
//
// Replaces the CPU renderer's per-object hierarchy (top-level BVH over objects
// plus one BVH per mesh, SAH with 12 buckets, one primitive per leaf:
// src/BVH.cpp:62-243, src/core/Scene.cpp:18-20, src/Triangle.cpp:179) by a
// single binned-SAH tree over every triangle and sphere, laid out for the GPU:
// 64-byte nodes that carry both child boxes, leaves of up to four primitives
// referenced straight from the parent, primitives reordered into leaf order.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <system_error>
#include <thread>

#include "tk_common.h"

namespace tkd {

namespace {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -mn[a]; }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); }
    }
    void grow(const float p[3]) {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
    }
    double area() const {
        double dx = double(mx[0]) - mn[0], dy = double(mx[1]) - mn[1], dz = double(mx[2]) - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0;
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
};

struct Prim {
    Box b;
    float c[3];
    int32_t id;
};

struct TmpNode {
    Box b;
    int32_t left = -1, right = -1;   // TmpNode indices
    int32_t first = 0, count = 0;    // leaf: slot range
};

// The two halves of a large range are built by two threads.  The result does not depend on how the
// threads interleave: a range of primitives only ever moves within itself, a leaf's slots are the
// positions of its range, and the final node numbers come from a walk of the finished tree.
constexpr int kTaskPrims = 2048;   // a half smaller than this stays on the thread that reached it
constexpr int kMaxThreads = 16;

struct Builder {
    std::vector<Prim> prims;
    std::vector<TmpNode> nodes;          // 2n slots, handed out by next_node
    std::atomic<int32_t> next_node{0};
    int max_leaf = 4;
    int depth_limit = 32;
    bool median_only = false;
    std::atomic<int> max_depth{0};
    std::atomic<int> threads{1};

    int32_t build(int start, int end, int depth) {
        int32_t me = next_node.fetch_add(1, std::memory_order_relaxed);
        Box b, cb;
        b.reset();
        cb.reset();
        for (int i = start; i < end; ++i) { b.grow(prims[i].b); cb.grow(prims[i].c); }
        nodes[me].b = b;
        int n = end - start;
        auto make_leaf = [&]() {
            nodes[me].first = start;
            nodes[me].count = n;
            int seen = max_depth.load(std::memory_order_relaxed);
            while (depth > seen && !max_depth.compare_exchange_weak(seen, depth, std::memory_order_relaxed)) {}
            return me;
        };
        if (n == 1) return make_leaf();
        int axis = 0;
        float ext[3] = {cb.mx[0] - cb.mn[0], cb.mx[1] - cb.mn[1], cb.mx[2] - cb.mn[2]};
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        int mid = -1;
        bool force_median = median_only || depth >= depth_limit;
        if (ext[axis] <= 0) {
            if (n <= max_leaf) return make_leaf();
            mid = (start + end) / 2;   // coincident centroids: split by count
        } else if (!force_median) {
            constexpr int NB = 16;
            int cnt[NB] = {0};
            Box bb[NB];
            for (auto& x : bb) x.reset();
            float k1 = NB * (1.0f - 1e-6f) / ext[axis];
            for (int i = start; i < end; ++i) {
                int bi = std::min(NB - 1, std::max(0, int(k1 * (prims[i].c[axis] - cb.mn[axis]))));
                cnt[bi]++;
                bb[bi].grow(prims[i].b);
            }
            double la[NB - 1], ra[NB - 1];
            int lc[NB - 1], rc[NB - 1];
            Box acc;
            acc.reset();
            int c = 0;
            for (int i = 0; i < NB - 1; ++i) { acc.grow(bb[i]); c += cnt[i]; la[i] = acc.area(); lc[i] = c; }
            acc.reset();
            c = 0;
            for (int i = NB - 1; i > 0; --i) { acc.grow(bb[i]); c += cnt[i]; ra[i - 1] = acc.area(); rc[i - 1] = c; }
            double best = std::numeric_limits<double>::infinity();
            int best_i = -1;
            for (int i = 0; i < NB - 1; ++i) {
                if (lc[i] == 0 || rc[i] == 0) continue;
                double cost = lc[i] * la[i] + rc[i] * ra[i];
                if (cost < best) { best = cost; best_i = i; }
            }
            const double c_trav = 1.0, c_isect = 1.5;
            double parent_area = std::max(b.area(), 1e-30);
            double split_cost = c_trav + c_isect * best / parent_area;
            if (n <= max_leaf && (best_i < 0 || split_cost >= c_isect * n)) return make_leaf();
            if (best_i >= 0) {
                auto it = std::partition(prims.begin() + start, prims.begin() + end, [&](const Prim& p) {
                    int bi = std::min(NB - 1, std::max(0, int(k1 * (p.c[axis] - cb.mn[axis]))));
                    return bi <= best_i;
                });
                mid = int(it - prims.begin());
            }
        }
        if (mid <= start || mid >= end) {
            if (n <= max_leaf && force_median) return make_leaf();
            mid = (start + end) / 2;
            std::nth_element(prims.begin() + start, prims.begin() + mid, prims.begin() + end,
                             [axis](const Prim& x, const Prim& y) { return x.c[axis] < y.c[axis]; });
        }
        int32_t l = -1, r = -1;
        bool forked = false;
        if (std::min(mid - start, end - mid) >= kTaskPrims) {
            if (threads.fetch_add(1, std::memory_order_relaxed) < kMaxThreads) {
                try {
                    std::thread other([&]() { l = build(start, mid, depth + 1); });
                    forked = true;
                    r = build(mid, end, depth + 1);
                    other.join();
                } catch (const std::system_error&) {   // no thread to be had: carry on alone
                }
            }
            threads.fetch_sub(1, std::memory_order_relaxed);
        }
        if (!forked) {
            l = build(start, mid, depth + 1);
            r = build(mid, end, depth + 1);
        }
        nodes[me].left = l;
        nodes[me].right = r;
        return me;
    }
};

int32_t encode_leaf(const TmpNode& n) { return ~(n.first * 8 + (n.count - 1)); }

}  // namespace

void build_bvh(const TkTriangle* tris, int n_tris, const TkSphere* spheres, int n_spheres, int max_leaf,
               int max_stack_depth, HostBvh* out) {
    out->nodes.clear();
    out->order.clear();
    out->max_depth = 0;
    for (int a = 0; a < 3; ++a) { out->bmin[a] = 0; out->bmax[a] = 0; }
    int n = n_tris + n_spheres;
    if (n == 0) return;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (getenv("TKHIP_VERBOSE"))
            fprintf(stderr, "[tkhip] build_bvh: %s at %.1f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    };
    Builder bld;
    bld.max_leaf = std::min(max_leaf, 8);
    bld.prims.resize(size_t(n));
    bld.nodes.resize(size_t(n) * 2);
    auto fill_prims = [&]() {
        host_parallel_for(size_t(n_tris), [&](size_t begin, size_t end) {
            for (size_t i = begin; i < end; ++i) {
                Prim& p = bld.prims[i];
                p.b.reset();
                p.b.grow(tris[i].v0);
                p.b.grow(tris[i].v1);
                p.b.grow(tris[i].v2);
                p.id = int32_t(i);
                for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * p.b.mn[a] + 0.5f * p.b.mx[a];
            }
        });
        for (int j = 0; j < n_spheres; ++j) {
            Prim& p = bld.prims[size_t(n_tris + j)];
            // pad by a few ulps: the hit point is re-projected onto the sphere
            float r = spheres[j].radius * 1.00001f;
            for (int a = 0; a < 3; ++a) { p.b.mn[a] = spheres[j].center[a] - r; p.b.mx[a] = spheres[j].center[a] + r; }
            p.id = n_tris + j;
            for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * p.b.mn[a] + 0.5f * p.b.mx[a];
        }
    };
    // a second attempt, with median splits only, if the SAH tree comes out too deep for the stack
    for (int attempt = 0; attempt < 2; ++attempt) {
        fill_prims();
        if (attempt == 1) std::fill(bld.nodes.begin(), bld.nodes.end(), TmpNode());
        bld.next_node = 0;
        bld.max_depth = 0;
        bld.median_only = attempt == 1;
        bld.depth_limit = max_stack_depth - 6;
        lap("boxes set up");
        bld.build(0, n, 0);
        lap("tree built");
        if (bld.max_depth < max_stack_depth - 1) break;
    }
    out->max_depth = bld.max_depth;
    out->order.resize(size_t(n));
    host_parallel_for(size_t(n), [&](size_t begin, size_t end) {
        for (size_t i = begin; i < end; ++i) out->order[i] = bld.prims[i].id;
    });
    const TmpNode& root = bld.nodes[0];
    for (int a = 0; a < 3; ++a) { out->bmin[a] = root.b.mn[a]; out->bmax[a] = root.b.mx[a]; }
    if (root.left < 0) {
        // The whole scene fits one leaf: give the traversal a root node all the same.  Its left
        // child is that leaf; the right child repeats it behind a box no ray reaches.
        BvhNode o{};
        for (int a = 0; a < 3; ++a) {
            o.lmin[a] = root.b.mn[a]; o.lmax[a] = root.b.mx[a];
            o.rmin[a] = 3.0e38f; o.rmax[a] = 3.0e38f;
        }
        o.left = o.right = encode_leaf(root);
        o.pad0 = o.pad1 = 0;
        out->nodes.push_back(o);
        return;
    }
    // number the interior nodes in depth-first order
    bld.nodes.resize(size_t(bld.next_node.load()));
    std::vector<int32_t> flat(bld.nodes.size(), -1);
    int32_t count = 0;
    std::vector<int32_t> stack{0};
    while (!stack.empty()) {
        int32_t t = stack.back();
        stack.pop_back();
        const TmpNode& tn = bld.nodes[size_t(t)];
        if (tn.left < 0) continue;
        flat[size_t(t)] = count++;
        stack.push_back(tn.right);
        stack.push_back(tn.left);
    }
    out->nodes.resize(size_t(count));
    host_parallel_for(bld.nodes.size(), [&](size_t begin, size_t end) {
        for (size_t t = begin; t < end; ++t) {
            if (flat[t] < 0) continue;
            const TmpNode& tn = bld.nodes[t];
            const TmpNode& L = bld.nodes[size_t(tn.left)];
            const TmpNode& R = bld.nodes[size_t(tn.right)];
            BvhNode& o = out->nodes[size_t(flat[t])];
            for (int a = 0; a < 3; ++a) {
                o.lmin[a] = L.b.mn[a]; o.lmax[a] = L.b.mx[a];
                o.rmin[a] = R.b.mn[a]; o.rmax[a] = R.b.mx[a];
            }
            o.left = L.left < 0 ? encode_leaf(L) : flat[size_t(tn.left)];
            o.right = R.left < 0 ? encode_leaf(R) : flat[size_t(tn.right)];
            o.pad0 = o.pad1 = 0;
        }
    });
    lap("nodes laid out");
}

// Wide nodes, breadth first: a node's children start as the two of its two-children node; the inner child
// with the largest surface is replaced by its own two until there are eight or none is inner.
int collapse_bvh(const BvhNode* nodes, int n_nodes, std::vector<WideNode>* out) {
    out->clear();
    if (n_nodes <= 0) return 0;
    struct Item { int node, wide, depth; };
    std::vector<Item> todo;
    todo.reserve((size_t)n_nodes / 2 + 1);
    out->reserve((size_t)n_nodes / 2 + 1);
    out->emplace_back();
    todo.push_back({0, 0, 1});
    int max_depth = 1;
    auto area = [](const WideChild& c) {
        const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto children_of = [&](int node, WideChild* l, WideChild* r) {
        const BvhNode& nd = nodes[node];
        for (int a = 0; a < 3; ++a) { l->lo[a] = nd.lmin[a]; l->hi[a] = nd.lmax[a]; r->lo[a] = nd.rmin[a]; r->hi[a] = nd.rmax[a]; }
        l->ref = nd.left; r->ref = nd.right;
        l->pad = r->pad = 0;
    };
    for (size_t at = 0; at < todo.size(); ++at) {
        const Item it = todo[at];
        max_depth = std::max(max_depth, it.depth);
        WideChild ch[8];
        int n = 2;
        children_of(it.node, &ch[0], &ch[1]);
        while (n < 8) {
            int pick = -1;
            float best = -1.f;
            for (int i = 0; i < n; ++i)
                if (ch[i].ref >= 0 && area(ch[i]) > best) { best = area(ch[i]); pick = i; }
            if (pick < 0) break;
            const int node = ch[pick].ref;
            children_of(node, &ch[pick], &ch[n]);
            ++n;
        }
        WideNode w;
        for (int i = 0; i < 8; ++i) {
            if (i < n) {
                w.c[i] = ch[i];
                if (ch[i].ref >= 0) {
                    w.c[i].ref = (int32_t)out->size();
                    out->emplace_back();
                    todo.push_back({ch[i].ref, w.c[i].ref, it.depth + 1});
                }
            } else {
                for (int a = 0; a < 3; ++a) { w.c[i].lo[a] = INFINITY; w.c[i].hi[a] = -INFINITY; }
                w.c[i].ref = kWideEmpty;
                w.c[i].pad = 0;
            }
        }
        (*out)[(size_t)it.wide] = w;
    }
    return max_depth;
}

}  // namespace tkd
