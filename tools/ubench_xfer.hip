// ubench_xfer.hip -- how fast can a host slice the caller owns (pageable, freshly allocated: a Rust Vec) reach HBM and come back?
// The trait-shaped entry points (zk_fr_fft_in_place, zk_msm_g1, ...) move 32 MB - 100 MB per call; this measures, per size:
//   a) hipMemcpy from / to the pageable buffer (the runtime's own staging)
//   b) hipHostRegister + hipMemcpyAsync + hipHostUnregister, the three timed separately
//   c) a ring of page-locked chunks filled / drained by T host threads while the DMA of the neighbouring chunks runs
// Build: hipcc -O2 --offload-arch=gfx950 tools/ubench_xfer.hip -o tools/_bin/ubench_xfer -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t sizes[] = {(size_t)1 << 20, (size_t)32 << 20, (size_t)100 << 20};
    const int reps = 5;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipStream_t st2;
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    void* dev;
    CK(hipMalloc(&dev, (size_t)128 << 20));
    const size_t CH = (size_t)4 << 20, NSLOT = 32;
    char* ring;
    CK(hipHostMalloc((void**)&ring, CH * NSLOT, hipHostMallocDefault));
    std::vector<hipEvent_t> ev(NSLOT);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (size_t bytes : sizes) {
        for (int rep = 0; rep < reps; rep++) {
            char* host = (char*)malloc(bytes);                       // fresh pages every time, as a Vec is
            memset(host, rep + 1, bytes);
            double t0 = now();
            CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
            double t1 = now();
            CK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
            double t2 = now();
            printf("{\"bytes\": %zu, \"mode\": \"pageable\", \"h2d_ms\": %.3f, \"d2h_ms\": %.3f}\n", bytes, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
            // b) register
            t0 = now();
            CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
            t1 = now();
            CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            t2 = now();
            CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double t3 = now();
            CK(hipHostUnregister(host));
            double t4 = now();
            printf("{\"bytes\": %zu, \"mode\": \"register\", \"register_ms\": %.3f, \"h2d_ms\": %.3f, \"d2h_ms\": %.3f, \"unregister_ms\": %.3f}\n", bytes,
                   (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
            // c) ring with T threads
            for (int T : {1, 2, 4, 8, 16}) {
                const size_t nch = (bytes + CH - 1) / CH;
                t0 = now();
                {
                    std::atomic<size_t> next{0};
                    std::vector<std::thread> th;
                    for (int t = 0; t < T; t++)
                        th.emplace_back([&] {
                            CK(hipSetDevice(0));
                            for (;;) {
                                const size_t k = next.fetch_add(1);
                                if (k >= nch) break;
                                const size_t slot = k % NSLOT, off = k * CH, len = std::min(CH, bytes - off);
                                if (k >= NSLOT) CK(hipEventSynchronize(ev[slot]));
                                memcpy(ring + slot * CH, host + off, len);
                                CK(hipMemcpyAsync((char*)dev + off, ring + slot * CH, len, hipMemcpyHostToDevice, st));
                                CK(hipEventRecord(ev[slot], st));
                            }
                        });
                    for (auto& x : th) x.join();
                }
                const double tq = now();
                CK(hipStreamSynchronize(st));
                t1 = now();
                // d2h: all chunk copies enqueued up front (bytes <= ring), threads drain them as their events fire
                {
                    for (size_t k = 0; k < nch && k < NSLOT; k++) {
                        const size_t off = k * CH, len = std::min(CH, bytes - off);
                        CK(hipMemcpyAsync(ring + k * CH, (char*)dev + off, len, hipMemcpyDeviceToHost, st));
                        CK(hipEventRecord(ev[k], st));
                    }
                    std::atomic<size_t> next{0};
                    std::vector<std::thread> th;
                    for (int t = 0; t < T; t++)
                        th.emplace_back([&] {
                            CK(hipSetDevice(0));
                            for (;;) {
                                const size_t k = next.fetch_add(1);
                                if (k >= nch || k >= NSLOT) break;
                                const size_t off = k * CH, len = std::min(CH, bytes - off);
                                CK(hipEventSynchronize(ev[k]));
                                memcpy(host + off, ring + k * CH, len);
                            }
                        });
                    for (auto& x : th) x.join();
                }
                t2 = now();
                printf("{\"bytes\": %zu, \"mode\": \"ring\", \"threads\": %d, \"h2d_ms\": %.3f, \"h2d_host_released_ms\": %.3f, \"d2h_ms\": %.3f}\n", bytes, T,
                       (t1 - t0) * 1e3, (tq - t0) * 1e3, (t2 - t1) * 1e3);
            }
            // d) pinned to device directly (the DMA roof)
            t0 = now();
            CK(hipMemcpyAsync(dev, ring, std::min(bytes, CH * NSLOT), hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            t1 = now();
            CK(hipMemcpyAsync(ring, dev, std::min(bytes, CH * NSLOT), hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            t2 = now();
            printf("{\"bytes\": %zu, \"mode\": \"pinned\", \"h2d_ms\": %.3f, \"d2h_ms\": %.3f}\n", bytes, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
            free(host);
        }
    }
    // persistent worker cost: thread creation is in the ring numbers above; report it alone
    double t0 = now();
    { std::vector<std::thread> th; for (int t = 0; t < 8; t++) th.emplace_back([] {}); for (auto& x : th) x.join(); }
    printf("{\"mode\": \"spawn8\", \"ms\": %.3f}\n", (now() - t0) * 1e3);
    return 0;
}
