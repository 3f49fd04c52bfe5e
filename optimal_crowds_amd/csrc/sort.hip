// Radix sort of (coarse-cell key, agent index) pairs: the spatial order used for tile culling.
// rocPRIM is used for this utility sort only (stable, deterministic); the hot kernels are hand-written.
#include "oc_internal.h"
#include <rocprim/rocprim.hpp>

int oc_sort_pairs(oc_handle* h, uint32_t* keys_in, uint32_t* keys_out, int32_t* vals_in, int32_t* vals_out, int n,
                  int end_bit, hipStream_t s) {
    size_t bytes = 0;
    OC_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0,
                                           (unsigned)end_bit, s));
    if (bytes > h->sort_tmp_bytes) {
        if (h->sort_tmp) OC_HIP_CHECK(hipFree(h->sort_tmp));
        h->sort_tmp = nullptr;
        OC_HIP_CHECK(hipMalloc(&h->sort_tmp, bytes));
        h->sort_tmp_bytes = bytes;
    }
    OC_HIP_CHECK(rocprim::radix_sort_pairs(h->sort_tmp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0,
                                           (unsigned)end_bit, s));
    return OC_OK;
}
