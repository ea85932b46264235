// Reference timing only (not part of the product): rocPRIM's radix_sort_pairs on 1 M (u32, u32) pairs, 22 key bits.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main() {
    const size_t n = 1 << 20;
    std::vector<unsigned> hk(n), hv(n);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; ++i) { hk[i] = rng() & ((1u << 21) - 1); hv[i] = (unsigned)i; }
    unsigned *k, *v, *k2, *v2;
    hipMalloc(&k, n * 4); hipMalloc(&v, n * 4); hipMalloc(&k2, n * 4); hipMalloc(&v2, n * 4);
    hipMemcpy(k, hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), n * 4, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0; void* tmp = nullptr;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, k, k2, v, v2, n, 0, 22);
    hipMalloc(&tmp, tmp_bytes);
    for (int i = 0; i < 5; ++i) rocprim::radix_sort_pairs(tmp, tmp_bytes, k, k2, v, v2, n, 0, 22);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) rocprim::radix_sort_pairs(tmp, tmp_bytes, k, k2, v, v2, n, 0, 22);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("rocprim radix_sort_pairs 1M pairs 22 bits: %.1f us per sort (temp %zu bytes)\n", ms / 50 * 1000, tmp_bytes);
    return 0;
}
