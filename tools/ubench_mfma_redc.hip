// ubench_mfma_redc.hip -- the ingredients of "Montgomery reduction on the matrix cores" (north_star: "MFMA only if Montgomery
// limb-matrix contraction proves dense"), measured on gfx950, one row each:
//
//   mad      the VALU work to replace: the 168 reduction multiply-adds of one Fq product (fp29.cuh::fp_mul_lazy: m_i * p_j columns),
//            as 168 v_mad_u64_u32 in 8 independent chains per lane                                   -> ns per element
//   mfma     36 x v_mfma_i32_16x16x64_i8 per wave = the constant-matrix products m = t_lo q' mod R (3 output tiles of 16 columns) and
//            m q (6 tiles) for 4 batches of 16 elements (K = 64 >= 48 operand bytes)                  -> ns per element
//   mfma+mad both streams in one wave (the matrix cores run beside the VALU): is the overlap real?   -> ns per element
//   bperm    the lane transposition the matrix cores need: element i lives in lane i (curve arithmetic), the A operand wants its
//            48 bytes in lanes i, i+16, i+32, i+48, and the 16 x 16 i32 result tiles come back 4 rows x 1 column per lane:
//            16 + 144 ds_bpermute_b32 per wave and reduction                                          -> ns per element
//   slice    14 limbs of 29 bits -> 12 packed byte words and 27 x (and, shift, add) carry steps back  -> ns per element
//
// Verdict (DESIGN 6 "The MFMA question"): mfma + bperm + slice must beat `mad` by 1.3x on the WHOLE product (337 multiply-adds, of
// which only these 168 move) to be worth a new data layout.   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_redc.hip -o tools/_bin/ubench_mfma_redc
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int ITERS = 512;          // reductions per lane (mad / slice) or per wave batch of 64 elements (mfma / bperm)

__global__ void k_mad(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int c = 0; c < 8; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 21; k++)                    // 21 x 8 = 168
#pragma unroll
            for (int c = 0; c < 8; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < 8; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool WITH_MAD>
__global__ void k_mfma(uint64_t* out, uint32_t a, uint32_t b) {
    v4i A = {(int)(a + threadIdx.x), (int)b, (int)(a ^ threadIdx.x), 7}, B = {(int)b, 3, (int)a, 5};
    v4i acc[9];
    for (int t = 0; t < 9; t++) acc[t] = v4i{0, 0, 0, (int)threadIdx.x};
    uint64_t m[8];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int c = 0; c < 8; c++) m[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int batch = 0; batch < 4; batch++) {
#pragma unroll
            for (int t = 0; t < 9; t++) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[t], 0, 0, 0);
            if (WITH_MAD) {
                // the other half of the product (the 169 operand multiply-adds of 16 elements' worth of lanes: a quarter of a wave
                // per batch) keeps the VALU busy beside the matrix cores: 42 per batch x 4 = 168 per wave iteration
#pragma unroll
                for (int k = 0; k < 42; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[k & 7]) : "v"(x), "v"(y) : "vcc");
            }
        }
    }
    uint64_t s = 0;
    for (int t = 0; t < 9; t++) s += (uint32_t)(acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3]);
    for (int c = 0; c < 8; c++) s += m[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_bperm(uint64_t* out, uint32_t a) {
    uint32_t v[8];
    for (int c = 0; c < 8; c++) v[c] = a + threadIdx.x * (c + 1);
    const int addr = ((threadIdx.x & 15) * 4 + (threadIdx.x >> 4)) << 2;          // a 16 x 4 transposition of the wave
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 20; k++)                     // 160 = 16 (operands in) + 144 (result columns out)
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[c]) + c;
    }
    uint64_t s = 0;
    for (int c = 0; c < 8; c++) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_slice(uint64_t* out, uint32_t a) {
    uint32_t l[14], w[12];
    for (int k = 0; k < 14; k++) l[k] = (a + threadIdx.x * (k + 3)) & 0x1fffffffu;
    uint64_t s = 0;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 12; j++) {                   // fp_pack: 29-bit limbs -> 32-bit words (the byte digits of the A operand)
            const int bit = 32 * j, i0 = bit / 29, o = bit - 29 * i0;
            uint32_t v = l[i0] >> o;
            if (i0 + 1 < 14) v |= l[i0 + 1] << (29 - o);
            if (i0 + 2 < 14 && 58 - o < 32) v |= l[i0 + 2] << (58 - o);
            w[j] = v;
        }
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < 27; k++) {                   // 27 result columns of 8-bit-digit sums back into 29-bit limbs
            const uint32_t t = w[k % 12] + carry + (uint32_t)k;
            if (k < 14) l[k] = (t ^ l[k]) & 0x1fffffffu;
            carry = t >> 29;
        }
        s += carry;
    }
    for (int k = 0; k < 14; k++) s += l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static float run(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    const int blocks = 256 * 8, threads = 256;                        // 8 waves per SIMD-quad: two per SIMD, like the accumulate kernel
    uint64_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
    const double lanes = (double)blocks * threads, waves = lanes / 64;
    float t;
    t = run([&] { hipLaunchKernelGGL(k_mad, blocks, threads, 0, 0, out, 3u, 5u); });
    const double ns_mad = t * 1e6 / (lanes * ITERS);
    printf("mad       168 v_mad_u64_u32 per element                       : %8.4f ns / element  (%.2f T mad/s)\n", ns_mad, 168.0 / ns_mad / 1e3);
    t = run([&] { hipLaunchKernelGGL(k_mfma<false>, blocks, threads, 0, 0, out, 3u, 5u); });
    const double ns_mfma = t * 1e6 / (waves * ITERS * 64);
    printf("mfma      36 v_mfma_i32_16x16x64_i8 per 64 elements            : %8.4f ns / element  (%.1f T i8-MAC/s)\n", ns_mfma,
           36.0 * 16384 / 64 / ns_mfma / 1e3);
    t = run([&] { hipLaunchKernelGGL(k_mfma<true>, blocks, threads, 0, 0, out, 3u, 5u); });
    const double ns_both = t * 1e6 / (waves * ITERS * 64);
    printf("mfma+mad  the same beside 168 v_mad_u64_u32 per WAVE iteration : %8.4f ns / element  (the two streams alone: %.4f + %.4f)\n", ns_both,
           ns_mfma, ns_mad / 64.0);
    t = run([&] { hipLaunchKernelGGL(k_bperm, blocks, threads, 0, 0, out, 3u); });
    const double ns_bperm = t * 1e6 / (waves * ITERS * 64);
    printf("bperm     160 ds_bpermute_b32 per 64 elements                  : %8.4f ns / element\n", ns_bperm);
    t = run([&] { hipLaunchKernelGGL(k_slice, blocks, threads, 0, 0, out, 3u); });
    const double ns_slice = t * 1e6 / (lanes * ITERS);
    printf("slice     limb <-> byte-word conversion + 27 carry steps       : %8.4f ns / element\n", ns_slice);
    const double redc_mfma = ns_mfma + ns_bperm + ns_slice, other = ns_mad * 169.0 / 168.0;
    printf("reduction on the matrix cores, serial   : %.4f ns vs %.4f on the VALU  -> whole product %.2fx\n", redc_mfma, ns_mad,
           (ns_mad + other) / (redc_mfma + other));
    const double overl = (ns_bperm + ns_slice + other > ns_mfma ? ns_bperm + ns_slice + other : ns_mfma);
    printf("... with the matrix instructions fully hidden under the VALU stream: whole product %.2fx (kill criterion: < 1.3x)\n",
           (ns_mad + other) / overl);
    return 0;
}
