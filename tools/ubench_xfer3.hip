// ubench_xfer3.hip -- where does the staged path lose time?  Host-side copy rates between a pageable buffer and a page-locked ring,
// per allocation flag of the ring and per thread count, without any DMA; then the DMA alone on the same rings.
// Build: hipcc -O2 --offload-arch=gfx950 tools/ubench_xfer3.hip -o tools/_bin/ubench_xfer3 -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>
#include <sched.h>
#include <string>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

static void par_copy(char* dst, const char* src, size_t bytes, int T) {
    std::vector<std::thread> th;
    const size_t per = (bytes + T - 1) / T;
    for (int t = 0; t < T; t++) th.emplace_back([=] { size_t lo = std::min(bytes, t * per), hi = std::min(bytes, lo + per); memcpy(dst + lo, src + lo, hi - lo); });
    for (auto& x : th) x.join();
}

int main() {
    const size_t bytes = (size_t)32 << 20;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    void* dev;
    CK(hipMalloc(&dev, bytes));
    printf("{\"cpu_of_main_thread\": %d}\n", sched_getcpu());
    struct { const char* name; unsigned flags; } kinds[] = {{"default", hipHostMallocDefault}, {"numa_user", hipHostMallocNumaUser}, {"noncoherent", hipHostMallocNonCoherent},
                                                            {"coherent", hipHostMallocCoherent}, {"portable", hipHostMallocPortable}};
    char* host = (char*)malloc(bytes);
    memset(host, 5, bytes);
    char* host2 = (char*)malloc(bytes);
    memset(host2, 6, bytes);
    for (int T : {1, 2, 4, 8}) {
        std::vector<double> v;
        for (int r = 0; r < 7; r++) { double t0 = now(); par_copy(host2, host, bytes, T); v.push_back((now() - t0) * 1e3); }
        printf("{\"copy\": \"pageable->pageable\", \"threads\": %d, \"ms\": %.3f, \"GBps\": %.1f}\n", T, med(v), bytes / med(v) / 1e6);
    }
    for (auto& k : kinds) {
        char* ring = nullptr;
        if (hipHostMalloc((void**)&ring, bytes, k.flags) != hipSuccess) { printf("{\"ring\": \"%s\", \"error\": \"alloc\"}\n", k.name); (void)hipGetLastError(); continue; }
        memset(ring, 1, bytes);
        for (int T : {1, 2, 4, 8}) {
            std::vector<double> in, out;
            for (int r = 0; r < 7; r++) {
                double t0 = now(); par_copy(ring, host, bytes, T); double t1 = now(); par_copy(host, ring, bytes, T); double t2 = now();
                in.push_back((t1 - t0) * 1e3); out.push_back((t2 - t1) * 1e3);
            }
            printf("{\"ring\": \"%s\", \"threads\": %d, \"fill_ms\": %.3f, \"fill_GBps\": %.1f, \"drain_ms\": %.3f, \"drain_GBps\": %.1f}\n", k.name, T, med(in), bytes / med(in) / 1e6,
                   med(out), bytes / med(out) / 1e6);
        }
        std::vector<double> h, d;
        for (int r = 0; r < 7; r++) {
            double t0 = now(); CK(hipMemcpyAsync(dev, ring, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
            double t1 = now(); CK(hipMemcpyAsync(ring, dev, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double t2 = now();
            h.push_back((t1 - t0) * 1e3); d.push_back((t2 - t1) * 1e3);
        }
        printf("{\"ring\": \"%s\", \"dma_h2d_ms\": %.3f, \"dma_d2h_ms\": %.3f}\n", k.name, med(h), med(d));
        if (std::string(k.name) == "default") {
            // (a) the same DMA while 8 threads copy unrelated memory; (b) the DMA cut into 8 chunks issued at once; (c) 8 chunks, each
            // issued by a thread right after it filled it (what a staged upload does)
            std::vector<double> a, b, c, c2;
            for (int r = 0; r < 7; r++) {
                std::thread load([&] { par_copy(host2, host, bytes, 8); par_copy(host, host2, bytes, 8); });
                double t0 = now(); CK(hipMemcpyAsync(dev, ring, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); a.push_back((now() - t0) * 1e3);
                load.join();
                t0 = now();
                for (int q = 0; q < 8; q++) CK(hipMemcpyAsync((char*)dev + q * (bytes / 8), ring + q * (bytes / 8), bytes / 8, hipMemcpyHostToDevice, st));
                CK(hipStreamSynchronize(st)); b.push_back((now() - t0) * 1e3);
                t0 = now();
                {
                    std::vector<std::thread> th;
                    for (int q = 0; q < 8; q++) th.emplace_back([&, q] { CK(hipSetDevice(0)); memcpy(ring + q * (bytes / 8), host + q * (bytes / 8), bytes / 8);
                                                                         CK(hipMemcpyAsync((char*)dev + q * (bytes / 8), ring + q * (bytes / 8), bytes / 8, hipMemcpyHostToDevice, st)); });
                    for (auto& x : th) x.join();
                }
                double t1 = now();
                CK(hipStreamSynchronize(st)); c.push_back((now() - t0) * 1e3); c2.push_back((t1 - t0) * 1e3);
            }
            printf("{\"dma_h2d_under_cpu_copy_load_ms\": %.3f, \"dma_h2d_8_chunks_at_once_ms\": %.3f, \"fill_then_dma_per_thread_total_ms\": %.3f, \"of_which_until_all_issued_ms\": %.3f}\n",
                   med(a), med(b), med(c), med(c2));
        }
        CK(hipHostFree(ring));
    }
    return 0;
}
