// tk_scan.hip — device-wide exclusive prefix sum of 32-bit counts.
//
// Used between the stages of the passes (photons per light path -> place in the store, queries per
// bucket -> first slot of the bucket, ...), i.e. while the persistent path kernels of the other lanes
// hold the LDS of every CU.  A library scan built on LDS tiles then waits for a whole workgroup slot
// (measured: 0.5 - 1.4 ms for a scan of 260 k counts); these kernels use registers and cross-lane
// moves only, so their waves fit into the wave slots the path kernels leave.  For millions of counts
// that pays (10 M-triangle field, 16 M light paths per batch: iteration 84 -> 81 ms).  Below
// kOwnScanFrom counts the library's scan is kept: there its wait delays nothing that matters, while
// these kernels, running at once, take issue slots from the path kernels (cornell_dragon: 4.90
// against 4.86 ms per iteration).
//
// Every wave scans a chunk of kScanChunk counts (16 consecutive counts per lane, a wave-wide scan of
// the lane sums) and leaves the chunk's total; one wave scans the totals; the chunks add their offset.
#include "tk_common.h"

namespace tkd {

namespace {

constexpr int kScanPerLane = 16;
constexpr int kScanChunk = 64 * kScanPerLane;

__device__ __forceinline__ int wave_exclusive_scan(int v, int lane, int* total) {
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

__global__ void __launch_bounds__(64) k_scan_chunks(const int32_t* in, int32_t* out, int64_t n, int32_t* totals) {
    const int lane = threadIdx.x & 63;
    const int64_t chunk = blockIdx.x;
    const int64_t base = chunk * kScanChunk + (int64_t)lane * kScanPerLane;
    if (chunk * kScanChunk >= n) return;      // whole waves only: the cross-lane moves below need every lane
    int v[kScanPerLane];
    int sum = 0;
#pragma unroll
    for (int t = 0; t < kScanPerLane; ++t) {
        v[t] = base + t < n ? in[base + t] : 0;
        sum += v[t];
    }
    int total;
    int run = wave_exclusive_scan(sum, lane, &total);
#pragma unroll
    for (int t = 0; t < kScanPerLane; ++t) {
        if (base + t < n) out[base + t] = run;
        run += v[t];
    }
    if (lane == 0) totals[chunk] = total;
}

// one wave: exclusive scan of the chunk totals, in place
__global__ void __launch_bounds__(64) k_scan_totals(int32_t* totals, int64_t n_chunks) {
    const int lane = threadIdx.x;
    int carry = 0;
    for (int64_t b0 = 0; b0 < n_chunks; b0 += kScanChunk) {
        const int64_t base = b0 + (int64_t)lane * kScanPerLane;
        int v[kScanPerLane];
        int sum = 0;
#pragma unroll
        for (int t = 0; t < kScanPerLane; ++t) {
            v[t] = base + t < n_chunks ? totals[base + t] : 0;
            sum += v[t];
        }
        int total;
        int run = carry + wave_exclusive_scan(sum, lane, &total);
#pragma unroll
        for (int t = 0; t < kScanPerLane; ++t) {
            if (base + t < n_chunks) totals[base + t] = run;
            run += v[t];
        }
        carry += total;
    }
}

__global__ void __launch_bounds__(64) k_scan_offsets(int32_t* out, int64_t n, const int32_t* totals) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x + kScanChunk;   // chunk 0 has no offset
    if (i < n) out[i] += totals[i / kScanChunk];
}

}  // namespace

int library_scan_i32_temp_bytes(size_t n, size_t* bytes);
int library_scan_i32_in(hipStream_t stream, void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n);
constexpr size_t kOwnScanFrom = (size_t)1 << 22;

int exclusive_scan_i32_temp_bytes(size_t n, size_t* bytes) {
    if (n < kOwnScanFrom) return library_scan_i32_temp_bytes(n, bytes);
    *bytes = (((n + kScanChunk - 1) / kScanChunk + 1) * 4 + 255) & ~(size_t)255;
    return TKHIP_OK;
}

int exclusive_scan_i32_in(hipStream_t stream, void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n) {
    if (n == 0) return TKHIP_OK;
    if (n < kOwnScanFrom) return library_scan_i32_in(stream, temp, temp_bytes, in, out, n);
    size_t need = 0;
    TK_TRY(exclusive_scan_i32_temp_bytes(n, &need));
    TK_REQUIRE(need <= temp_bytes, TKHIP_E_LIMIT, "scan of %zu values needs %zu bytes of scratch, %zu given", n, need, temp_bytes);
    int32_t* totals = static_cast<int32_t*>(temp);
    const int64_t n_chunks = (int64_t)((n + kScanChunk - 1) / kScanChunk);
    // blocks of one wave: they fit wherever a single wave slot is free
    k_scan_chunks<<<(unsigned)n_chunks, 64, 0, stream>>>(in, out, (int64_t)n, totals);
    if (n_chunks > 1) {
        k_scan_totals<<<1, 64, 0, stream>>>(totals, n_chunks);
        k_scan_offsets<<<(unsigned)divup((int64_t)n - kScanChunk, 64), 64, 0, stream>>>(out, (int64_t)n, totals);
    }
    TK_LAUNCH_CHECK();
    return TKHIP_OK;
}

int exclusive_scan_i32(hipStream_t stream, DevBuf& temp, const int32_t* in, int32_t* out, size_t n) {
    if (n == 0) return TKHIP_OK;
    size_t need = 0;
    TK_TRY(exclusive_scan_i32_temp_bytes(n, &need));
    TK_TRY(temp.ensure(need));
    return exclusive_scan_i32_in(stream, temp.p, temp.bytes, in, out, n);
}

}  // namespace tkd
