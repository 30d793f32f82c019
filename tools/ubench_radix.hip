// How fast is rocPRIM's radix sort on (20-bit key, 32-bit value) pairs at the MSM sort's size (13.6 M entries)?
// Decides whether the counting sort of msm.hip (L2 / memory-side atomics) should be replaced.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
int main() {
    for (size_t n : {(size_t)13631462, (size_t)(1 << 24)}) {
        for (int bits : {19, 20, 32}) {
            std::vector<uint32_t> hk(n), hv(n);
            uint64_t s = 88172645463325252ull;
            for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = (uint32_t)s & (bits == 32 ? 0xffffffffu : ((1u << bits) - 1)); hv[i] = (uint32_t)i; }
            uint32_t *k0, *k1, *v0, *v1;
            hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
            hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
            hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
            size_t tb = 0;
            hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k0, k1, v0, v1, (int)n, 0, bits, 0);
            void* tmp; hipMalloc(&tmp, tb);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; rep++) hipcub::DeviceRadixSort::SortPairs(tmp, tb, k0, k1, v0, v1, (int)n, 0, bits, 0);
            hipEventRecord(a, 0);
            for (int rep = 0; rep < 5; rep++) hipcub::DeviceRadixSort::SortPairs(tmp, tb, k0, k1, v0, v1, (int)n, 0, bits, 0);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("n=%zu bits=%d: %.3f ms per sort, %.2f G pairs/s, temp %zu MB\n", n, bits, ms / 5, n / (ms / 5) / 1e6, tb >> 20);
            hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
        }
    }
    return 0;
}
