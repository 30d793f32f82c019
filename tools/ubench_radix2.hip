// rocPRIM onesweep with a wider digit: 19 key bits in 2 passes (10 + 9) instead of 3 (8 + 8 + 3)?
// hipcc -O3 --offload-arch=gfx950 tools/ubench_radix2.hip -o tools/_bin/ubench_radix2
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>

template <class Config>
float run(const char* name, size_t n, int bits) {
    std::vector<uint32_t> hk(n), hv(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = (uint32_t)s & ((1u << bits) - 1); hv[i] = (uint32_t)i; }
    uint32_t *k0, *k1, *v0, *v1;
    hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
    size_t tb = 0;
    rocprim::radix_sort_pairs<Config>(nullptr, tb, k0, k1, v0, v1, n, 0u, (unsigned)bits, (hipStream_t)0);
    void* tmp; hipMalloc(&tmp, tb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) rocprim::radix_sort_pairs<Config>(tmp, tb, k0, k1, v0, v1, n, 0u, (unsigned)bits, (hipStream_t)0);
    hipEventRecord(a, 0);
    for (int rep = 0; rep < 5; rep++) rocprim::radix_sort_pairs<Config>(tmp, tb, k0, k1, v0, v1, n, 0u, (unsigned)bits, (hipStream_t)0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> ok(n);
    hipMemcpy(ok.data(), k1, n * 4, hipMemcpyDeviceToHost);
    bool sorted = true;
    for (size_t i = 1; i < n; i++) if (ok[i - 1] > ok[i]) { sorted = false; break; }
    printf("%-28s n=%zu bits=%d: %.3f ms per sort, %.2f G pairs/s, sorted=%d, temp %zu MB\n", name, n, bits, ms / 5, n / (ms / 5) / 1e6, (int)sorted, tb >> 20);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
    return ms / 5;
}

using namespace rocprim;
template <unsigned BS, unsigned IPT, unsigned RB>
using OS = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<BS, IPT>, RB, block_radix_rank_algorithm::match>>;

int main() {
    const size_t n = 13631462;
    run<default_config>("default", n, 19);
    run<OS<1024, 8, 8>>("onesweep 1024x8 r8", n, 19);
    run<OS<1024, 8, 10>>("onesweep 1024x8 r10", n, 19);
    run<OS<1024, 6, 10>>("onesweep 1024x6 r10", n, 19);
    run<OS<1024, 4, 10>>("onesweep 1024x4 r10", n, 19);
    run<OS<512, 12, 9>>("onesweep 512x12 r9", n, 18);
    run<OS<1024, 8, 10>>("onesweep 1024x8 r10 (20 bits)", n, 20);
    run<default_config>("default 3n", 3 * n, 19);
    run<OS<1024, 8, 10>>("onesweep 1024x8 r10 3n", 3 * n, 19);
    return 0;
}
