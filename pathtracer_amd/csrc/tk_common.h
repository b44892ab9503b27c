// tk_common.h — host-side plumbing shared by the translation units of
// libtkhip.so: error reporting, device buffers, the context object.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "tkhip.h"

namespace tkd {

void set_error(const char* fmt, ...);

#define TK_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess) {                                                           \
            tkd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
            return TKHIP_E_HIP;                                                           \
        }                                                                                 \
    } while (0)

#define TK_TRY(call)                 \
    do {                             \
        int _rc = (call);            \
        if (_rc != TKHIP_OK) return _rc; \
    } while (0)

#define TK_REQUIRE(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            tkd::set_error(__VA_ARGS__); \
            return (code);               \
        }                                \
    } while (0)

// Checks the launch itself (bad grid, missing code object); execution faults
// surface at the next synchronisation.
#define TK_LAUNCH_CHECK() TK_HIP(hipGetLastError())

// A device allocation that grows on demand and is freed with its owner.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return TKHIP_OK;
        if (p) {
            (void)hipFree(p);
            p = nullptr;
            bytes = 0;
        }
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return TKHIP_E_NOMEM;
        }
        bytes = want;
        return TKHIP_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

inline int divup(int64_t a, int64_t b) { return int((a + b - 1) / b); }

// ------------------------------------------------------------ device scene
// 64-byte BVH node: both child boxes live in the parent so one fetch decides
// both descents.  child < 0 encodes a leaf: ~child = first_slot * 8 + (count-1).
struct BvhNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// 256-byte wide node: up to eight descendants of a node of the tree above whose boxes partition its
// primitives (the node's two children with the largest of them replaced by its own two children, over and
// over, until there are eight or only leaves are left).  ref >= 0: a wide node; < 0: a leaf as in BvhNode;
// kWideEmpty: no child.  Eight lanes test the eight boxes of a visit side by side (walk_wide).
struct WideChild {
    float lo[3], hi[3];
    int32_t ref, pad;
};
struct WideNode {
    WideChild c[8];
};
static_assert(sizeof(WideNode) == 256, "WideNode must be 256 bytes");
constexpr int32_t kWideEmpty = INT32_MIN;
constexpr int kWideRaysDefault = 12;  // measured: 8 .. 16 alike, 24 slower
// Larger trees are not collapsed: their passes are bound by throughput, not by their last rays, and the collapse
// (on the host, 3.5 ms for the 76 k nodes of the dragon scene) would take a third of the set-up of a 1 M-triangle scene.
constexpr int kWideMaxNodes = 1 << 18;

// Per-slot triangle record for the intersection test (leaf order).
struct TriIsect {
    float v0[3], e1[3], e2[3];
    int32_t prim;       // original primitive id (>= n_tris: sphere n_tris + j)
    int32_t pad[2];
};
static_assert(sizeof(TriIsect) == 48, "TriIsect must be 48 bytes");

// Per-primitive record read once for the winning hit (original order).
struct TriShade {
    float v0[3], nx;
    float v1[3], ny;
    float v2[3], nz;
};
static_assert(sizeof(TriShade) == 48, "TriShade must be 48 bytes");

struct DLight {
    float Le[3];
    int32_t two_sided;
    int32_t tri_first, tri_count;   // triangle range of the emitting mesh
    int32_t cdf_first;              // offset into the light cdf array (tri_count+1 entries)
    float area;
};

struct DScene {
    const BvhNode* nodes;
    const TriIsect* tri_isect;
    const TriShade* tri_shade;
    const float4* spheres;          // center.xyz, radius
    const int32_t* prim_mat;        // material | (light+1) << 16, per original primitive id
    const TkMaterial* mats;
    const DLight* lights;
    const float* light_cdf;
    int32_t n_tris, n_spheres, n_lights, n_nodes;
    const WideNode* wide;           // the same tree with eight children per node, or null (not built for this scene)
    int32_t wide_rays;              // a wave of a path kernel with at most this many rays left walks them in the wide tree
    float bmin[3], bmax[3];
    // Padded bounds of the objects with a specular material (min xyz, max xyz), n_spec_boxes of them; -1:
    // more such objects than kMaxSpecBoxes, no statement.  A light path for the caustic map that misses all
    // of them cannot meet a specular surface first and so stores nothing.
    int32_t n_spec_boxes;
    float spec_box[8][6];
};
constexpr int kMaxSpecBoxes = 8;

struct DCamera {
    float c2w[16];
    float r2c[16];
    int32_t w, h;
};

struct HostBvh {
    std::vector<BvhNode> nodes;
    std::vector<int32_t> order;     // slot -> original primitive id
    int max_depth = 0;
    float bmin[3], bmax[3];
};

// Builds the flat two-children-per-node BVH over triangles followed by spheres.
void build_bvh(const TkTriangle* tris, int n_tris, const TkSphere* spheres, int n_spheres,
               int max_leaf, int max_stack_depth, HostBvh* out);

// The eight-children form of a two-children tree (WideNode).  Returns the depth of the wide tree.
int collapse_bvh(const BvhNode* nodes, int n_nodes, std::vector<WideNode>* out);

// The same kind of tree built on the GPU from device arrays (bvh_build_device.hip): nodes and the
// per-slot triangle records are left in `nodes` / `isect`.
int build_bvh_device(hipStream_t stream, DevBuf& prim_temp, const TkTriangle* d_tris, int n_tris, const TkSphere* d_spheres,
                     int n_spheres, int max_leaf, int max_stack_depth, DevBuf& nodes, DevBuf& isect, int* n_nodes, int* depth,
                     float bmin[3], float bmax[3]);

// fn(begin, end) over [0, n) in contiguous pieces on up to 16 host threads; small n runs on the caller.
template <class F>
void host_parallel_for(size_t n, F fn) {
    constexpr size_t kPiece = 65536;
    size_t pieces = std::min<size_t>((n + kPiece - 1) / kPiece, 16);
    if (pieces <= 1) {
        fn(size_t(0), n);
        return;
    }
    std::vector<std::thread> pool;
    size_t done = 0;
    for (size_t t = 0; t + 1 < pieces; ++t) {
        size_t end = n * (t + 1) / pieces;
        try {
            pool.emplace_back([=]() { fn(done, end); });
        } catch (const std::system_error&) {   // no thread to be had: the caller takes the rest
            break;
        }
        done = end;
    }
    fn(done, n);
    for (auto& t : pool) t.join();
}

struct SppmState;

}  // namespace tkd

struct tkhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool has_scene = false, has_camera = false;
    int bvh_builder = 0;            // 0: by scene size, 1: host threads, 2: GPU (tkhip_set_bvh_builder)
    int bvh_on_device = 0, bvh_depth = 0;
    float bvh_ms = 0.f;             // wall time of the last build, uploads of its inputs included
    tkd::DScene scene{};
    tkd::DCamera cam{};
    tkd::DevBuf b_wide;
    int wide_rays = tkd::kWideRaysDefault;
    int wide_depth = 0;
    tkd::DevBuf b_nodes, b_tri_isect, b_tri_shade, b_spheres, b_prim_mat, b_mats, b_lights, b_cdf;
    tkd::DevBuf b_dirtab;           // 4 x 256 floats: cosTheta, sinTheta, cosPhi, sinPhi
    tkd::DevBuf scratch_a, scratch_b, scratch_c, scratch_d, scratch_e, scratch_f;
    tkd::DevBuf prim_temp;          // rocPRIM temporary storage
    tkd::SppmState* sppm = nullptr;
    std::vector<TkMaterial> h_mats;
};

namespace tkd {

// rocPRIM wrappers (tk_prims.hip).  Asynchronous on `stream`; `temp` is the
// caller's scratch allocation (one per stream, grown on demand).
int sort_pairs_u64(hipStream_t stream, DevBuf& temp, uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout,
                   size_t n, int end_bit);
int sort_pairs_u32(hipStream_t stream, DevBuf& temp, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout,
                   size_t n, int end_bit);
// The same sort in caller-provided scratch of temp_bytes (no allocation; TKHIP_E_LIMIT if too small),
// and the scratch size a sort of n pairs needs.
int sort_pairs_u32_in(hipStream_t stream, void* temp, size_t temp_bytes, uint32_t* kin, uint32_t* kout, uint32_t* vin,
                      uint32_t* vout, size_t n, int end_bit);
int sort_pairs_u32_temp_bytes(size_t n, int end_bit, size_t* bytes);
// Grows `temp` to what sort_pairs_u32 of n pairs needs, without sorting.
int sort_pairs_u32_reserve(DevBuf& temp, size_t n, int end_bit);
int exclusive_scan_i32(hipStream_t stream, DevBuf& temp, const int32_t* in, int32_t* out, size_t n);
// The same scan in caller-provided scratch (no allocation; TKHIP_E_LIMIT if too small), and the size it needs.
int exclusive_scan_i32_in(hipStream_t stream, void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n);
int exclusive_scan_i32_temp_bytes(size_t n, size_t* bytes);

int upload_direction_tables(tkhip_ctx* ctx);
// Looks up the address of the library's static gather scratch on `device` (which must be current).
int static_scratch_init(int device);
void destroy_sppm(tkhip_ctx* ctx);
// The photons this rank's last photon pass stored for map `which` (arrays on the context's GPU:
// 3 floats, 3 floats, 2 bytes per photon) and the light paths it shot for them.
int sppm_local_photons(tkhip_ctx* ctx, int which, const void** pos, const void** pow, const void** dir, int64_t* n,
                       int64_t* shot);
// Waits for camera paths that were begun ahead and are still adding direct light to the pixel planes (they stay
// "in flight" for the iteration that picks them up); the planes of the render on the context's GPU.
int sppm_wait_camera_lane(tkhip_ctx* ctx);
float* sppm_planes_device(tkhip_ctx* ctx);

}  // namespace tkd
