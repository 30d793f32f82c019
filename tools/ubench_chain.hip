// Micro-benchmark: v_mad_u64_u32 issue rate as a function of the number of INDEPENDENT accumulator chains per wave and the
// number of waves per SIMD (gfx950).  The Montgomery products of fp29.cuh are one dependent chain per product.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_chain.hip -o tools/_bin/ubench_chain && tools/_bin/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 8192;

template <int CH>
__global__ void __launch_bounds__(256) k_chain(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS / CH; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CH; c++)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
int run(uint64_t* d, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;    // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<CH>, blocks, 256, 0, 0, d, 3u, 5u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chain<CH>, blocks, 256, 0, 0, d, 3u, 5u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mads = (double)blocks * 256 * (ITERS / CH) * 8 * CH;
    printf("chains/wave %d  waves/SIMD %d : %7.2f T mad/s   (%.1f cycles per wave-instruction per SIMD at 2.4 GHz)\n", CH, waves_per_simd,
           mads / (ms * 1e-3) / 1e12, 2.4e9 * (ms * 1e-3) / ((double)waves_per_simd * (ITERS / CH) * 8 * CH));
    return 0;
}

int main() {
    uint64_t* d;
    CHECK(hipMalloc(&d, (size_t)256 * 8 * 256 * 8));
    for (int w : {1, 2, 3, 4, 8}) {
        if (run<1>(d, w)) return 1;
        if (run<2>(d, w)) return 1;
        if (run<4>(d, w)) return 1;
    }
    return 0;
}
