// sort_rocprim.hip -- stable (key, value) radix sort used by the voxel ranking.
// Round-1 implementation: rocPRIM's LSD radix sort (AMD's native primitive library, stable by
// construction), restricted to the low `bits` of the key.  Kept in its own translation unit so
// the hand-written kernels do not pay its template compile time.
#include "rt.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

size_t fbbev_rt_sort_pairs_temp_bytes(size_t n, int bits) {
    size_t bytes = 0;
    uint32_t* k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, n, 0u, (unsigned)bits, (hipStream_t)0);
    return bytes;
}

int fbbev_rt_sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                        const uint32_t* vals_in, uint32_t* vals_out, size_t n, int bits,
                        fbbev_rt_stream stream) {
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n,
                                             0u, (unsigned)bits, stream);
    return (int)e;
}
