// tk_prims.hip — device-wide radix sort and (for small inputs) prefix scan of
// rocPRIM, kept in a translation unit of their own because the headers are
// heavy to compile.  The prefix sum the passes call is in tk_scan.hip.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "tk_common.h"

namespace tkd {

template <class K>
static int sort_pairs_impl(hipStream_t stream, DevBuf& temp, K* kin, K* kout, uint32_t* vin, uint32_t* vout, size_t n, int end_bit) {
    if (n == 0) return TKHIP_OK;
    size_t bytes = 0;
    TK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, end_bit, stream));
    TK_TRY(temp.ensure(bytes));
    TK_HIP(rocprim::radix_sort_pairs(temp.p, bytes, kin, kout, vin, vout, n, 0, end_bit, stream));
    return TKHIP_OK;
}

int sort_pairs_u64(hipStream_t stream, DevBuf& temp, uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, size_t n,
                   int end_bit) {
    return sort_pairs_impl<uint64_t>(stream, temp, kin, kout, vin, vout, n, end_bit);
}

int sort_pairs_u32(hipStream_t stream, DevBuf& temp, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout, size_t n,
                   int end_bit) {
    return sort_pairs_impl<uint32_t>(stream, temp, kin, kout, vin, vout, n, end_bit);
}

int sort_pairs_u32_temp_bytes(size_t n, int end_bit, size_t* bytes) {
    *bytes = 0;
    if (n == 0) return TKHIP_OK;
    uint32_t* none = nullptr;
    TK_HIP(rocprim::radix_sort_pairs(nullptr, *bytes, none, none, none, none, n, 0, end_bit, (hipStream_t) nullptr));
    return TKHIP_OK;
}

int sort_pairs_u32_in(hipStream_t stream, void* temp, size_t temp_bytes, uint32_t* kin, uint32_t* kout, uint32_t* vin,
                      uint32_t* vout, size_t n, int end_bit) {
    if (n == 0) return TKHIP_OK;
    size_t bytes = 0;
    TK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, end_bit, stream));
    TK_REQUIRE(bytes <= temp_bytes, TKHIP_E_LIMIT, "sort of %zu pairs needs %zu bytes of scratch, %zu given", n, bytes, temp_bytes);
    TK_HIP(rocprim::radix_sort_pairs(temp, bytes, kin, kout, vin, vout, n, 0, end_bit, stream));
    return TKHIP_OK;
}

int sort_pairs_u32_reserve(DevBuf& temp, size_t n, int end_bit) {
    if (n == 0) return TKHIP_OK;
    size_t bytes = 0;
    uint32_t* none = nullptr;
    TK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, none, none, none, none, n, 0, end_bit, (hipStream_t) nullptr));
    return temp.ensure(bytes);
}

// The library's scan, for the sizes tk_scan.hip leaves to it.
int library_scan_i32_temp_bytes(size_t n, size_t* bytes) {
    *bytes = 0;
    if (n == 0) return TKHIP_OK;
    const int32_t* none = nullptr;
    int32_t* out = nullptr;
    TK_HIP(rocprim::exclusive_scan(nullptr, *bytes, none, out, 0, n, rocprim::plus<int32_t>(), (hipStream_t) nullptr));
    return TKHIP_OK;
}

int library_scan_i32_in(hipStream_t stream, void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n) {
    if (n == 0) return TKHIP_OK;
    size_t bytes = 0;
    TK_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int32_t>(), stream));
    TK_REQUIRE(bytes <= temp_bytes, TKHIP_E_LIMIT, "scan of %zu values needs %zu bytes of scratch, %zu given", n, bytes, temp_bytes);
    TK_HIP(rocprim::exclusive_scan(temp, bytes, in, out, 0, n, rocprim::plus<int32_t>(), stream));
    return TKHIP_OK;
}

}  // namespace tkd
