// tk_prims.hip — device-wide radix sort (rocPRIM), kept in a translation unit
// of its own because the headers are heavy to compile.  (The prefix sum is in
// tk_scan.hip.)
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

}  // namespace tkd
