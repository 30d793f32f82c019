// Micro-benchmark: issue rate of the integer / fp64 multiply instructions the field
// arithmetic is built from (gfx950).  Prints ops/s per instruction kind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 8;  // independent chains per thread

__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mad64_addc(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH]; uint32_t hi[CH];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) { acc[c] = threadIdx.x + c; hi[c] = c; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                         : "+v"(acc[c]), "+v"(hi[c]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s += acc[c] + hi[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mullo(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[CH];
    uint32_t y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) acc[c] = a + threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(y));
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mulhi(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[CH];
    uint32_t y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) acc[c] = a + threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(y));
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mad24(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[CH];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"(y));
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_addc(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[CH];
    uint32_t y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) acc[c] = a + threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(acc[c]) : "v"(y) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_dfma(uint64_t* out, double a, double b) {
    double acc[CH];
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"(y));
    }
    double s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

__global__ void k_ffma(uint64_t* out, float a, float b) {
    float acc[CH];
    float x = a + threadIdx.x * 1e-9f, y = b;
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"(y));
    }
    float s = 0;
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}


#define MAKE_K(NAME, ASM, ...) \
__global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) { \
    uint32_t acc[CH]; uint32_t x = a + threadIdx.x, y = b + blockIdx.x; \
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c; \
    for (int i = 0; i < ITERS; i++) { _Pragma("unroll") for (int c = 0; c < CH; c++) asm volatile(ASM : "+v"(acc[c]) : __VA_ARGS__); } \
    uint64_t s = 0; for (int c = 0; c < CH; c++) s += acc[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s; }
MAKE_K(k_add_u32, "v_add_u32 %0, %0, %1", "v"(y))
MAKE_K(k_add3, "v_add3_u32 %0, %0, %1, %2", "v"(x), "v"(y))
MAKE_K(k_and, "v_and_b32 %0, %0, %1", "v"(y))
MAKE_K(k_lshr, "v_lshrrev_b32 %0, 3, %0", "v"(y))
MAKE_K(k_ashr, "v_ashrrev_i32 %0, 3, %0", "v"(y))
MAKE_K(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29", "v"(y))
MAKE_K(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1", "v"(y))
MAKE_K(k_and_or, "v_and_or_b32 %0, %0, %1, %2", "v"(x), "v"(y))
MAKE_K(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", "v"(y))
MAKE_K(k_sub_u32, "v_sub_u32 %0, %0, %1", "v"(y))
MAKE_K(k_bfe, "v_bfe_u32 %0, %0, 3, 29", "v"(y))
MAKE_K(k_bfi, "v_bfi_b32 %0, %1, %0, %2", "v"(x), "v"(y))
MAKE_K(k_xor, "v_xor_b32 %0, %0, %1", "v"(y))
__global__ void k_cnd64(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[CH]; uint32_t y = b + blockIdx.x;
    unsigned long long m = 0x5555555555555555ull ^ a;
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(y), "s"(m));
    }
    uint64_t s = 0; for (int c = 0; c < CH; c++) s += acc[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cmp_cnd(uint64_t* out, uint32_t a, uint32_t b) {   // what the compiler emits: v_cmp + v_cndmask pairs
    uint32_t acc[CH]; uint32_t y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[c]) : "v"(y) : "vcc");
    }
    uint64_t s = 0; for (int c = 0; c < CH; c++) s += acc[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lshr64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH];
    for (int c = 0; c < CH; c++) acc[c] = ((uint64_t)(threadIdx.x + c) << 40) + a + b;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[c]));
    }
    uint64_t s = 0; for (int c = 0; c < CH; c++) s += acc[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lshl_add64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH]; uint64_t y = b + blockIdx.x;
    for (int c = 0; c < CH; c++) acc[c] = threadIdx.x + c + a;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(y));
    }
    uint64_t s = 0; for (int c = 0; c < CH; c++) s += acc[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0 * 1e-3;
}

int main() {
    int blocks = 256 * 8, threads = 256;
    uint64_t* out; CHECK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads));
    double n = (double)blocks * threads * ITERS * CH;
    double t;
    t = timeit([&] { k_ffma<<<blocks, threads>>>(out, 1.0f, 1.0f); });      printf("v_fma_f32          %8.2f Tinst/s\n", n / t / 1e12);
    t = timeit([&] { k_dfma<<<blocks, threads>>>(out, 1.0, 1.0); });        printf("v_fma_f64          %8.2f Tinst/s\n", n / t / 1e12);
    t = timeit([&] { k_addc<<<blocks, threads>>>(out, 3, 5); });            printf("v_add_co_u32       %8.2f Tinst/s\n", n / t / 1e12);
    t = timeit([&] { k_mad24<<<blocks, threads>>>(out, 3, 5); });           printf("v_mad_u32_u24      %8.2f Tinst/s\n", n / t / 1e12);
    t = timeit([&] { k_mullo<<<blocks, threads>>>(out, 3, 5); });           printf("v_mul_lo_u32       %8.2f Tinst/s\n", n / t / 1e12);
    t = timeit([&] { k_mulhi<<<blocks, threads>>>(out, 3, 5); });           printf("v_mul_hi_u32       %8.2f Tinst/s\n", n / t / 1e12);
    t = timeit([&] { k_mad64<<<blocks, threads>>>(out, 3, 5); });           printf("v_mad_u64_u32      %8.2f Tinst/s\n", n / t / 1e12);
    t = timeit([&] { k_mad64_addc<<<blocks, threads>>>(out, 3, 5); });      printf("mad_u64_u32+addc   %8.2f Tpair/s\n", n / t / 1e12);

#define RUN(K, LABEL) t = timeit([&] { K<<<blocks, threads>>>(out, 3, 5); }); printf("%-18s %8.2f Tinst/s\n", LABEL, n / t / 1e12);
    RUN(k_add_u32, "v_add_u32") RUN(k_sub_u32, "v_sub_u32") RUN(k_add3, "v_add3_u32") RUN(k_and, "v_and_b32") RUN(k_lshr, "v_lshrrev_b32")
    RUN(k_ashr, "v_ashrrev_i32") RUN(k_alignbit, "v_alignbit_b32") RUN(k_lshl_or, "v_lshl_or_b32") RUN(k_and_or, "v_and_or_b32")
    RUN(k_cndmask, "v_cndmask_b32") RUN(k_cnd64, "v_cndmask_e64 sgpr") RUN(k_cmp_cnd, "v_cmp+v_cndmask") RUN(k_bfi, "v_bfi_b32") RUN(k_xor, "v_xor_b32") RUN(k_bfe, "v_bfe_u32") RUN(k_lshr64, "v_lshrrev_b64") RUN(k_lshl_add64, "v_lshl_add_u64")
    hipFree(out);
    return 0;
}
