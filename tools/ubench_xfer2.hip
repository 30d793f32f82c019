// ubench_xfer2.hip -- the copies of the host-slice entry points at the sizes where they went wrong: zk_fr_fft_in_place at 2^16 elements
// (2 MB) took 2.5 - 4 ms per call in a steady-state caller, 0.25 ms at 2^10 and 1.7 ms at 2^20 (profiles/r5_trait_path_first.jsonl).
// Per size and per kind of host buffer (fresh malloc each time -- mmap'ed above the allocator's threshold -- or one heap buffer
// reused, or memory from the brk heap after the threshold has grown): hipMemcpy, hipMemcpyAsync + sync, each direction.
// Build: hipcc -O2 --offload-arch=gfx950 tools/ubench_xfer2.hip -o tools/_bin/ubench_xfer2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    void* dev;
    CK(hipMalloc(&dev, (size_t)64 << 20));
    char* pinned;
    CK(hipHostMalloc((void**)&pinned, (size_t)64 << 20, hipHostMallocDefault));
    const size_t sizes[] = {(size_t)32 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20, (size_t)32 << 20};
    // make the allocator's mmap threshold grow as it does in a long-lived caller: free one large block
    { void* p = malloc((size_t)40 << 20); memset(p, 1, (size_t)40 << 20); free(p); }
    for (size_t bytes : sizes) {
        for (int kind = 0; kind < 3; kind++) {         // 0: fresh malloc per rep, 1: one buffer reused, 2: page-locked staging + memcpy
            std::vector<double> hs, ds, ha, da;
            char* keep = (char*)malloc(bytes);
            memset(keep, 3, bytes);
            for (int rep = 0; rep < 9; rep++) {
                char* host = kind == 0 ? (char*)malloc(bytes) : keep;
                if (kind == 0) memset(host, rep + 1, bytes);
                double t0 = now();
                if (kind == 2) { memcpy(pinned, host, bytes); CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
                else CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
                double t1 = now();
                if (kind == 2) { CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); memcpy(host, pinned, bytes); }
                else CK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
                double t2 = now();
                CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
                CK(hipStreamSynchronize(st));
                double t3 = now();
                CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                double t4 = now();
                hs.push_back((t1 - t0) * 1e3); ds.push_back((t2 - t1) * 1e3); ha.push_back((t3 - t2) * 1e3); da.push_back((t4 - t3) * 1e3);
                if (kind == 0) free(host);
            }
            free(keep);
            printf("{\"bytes\": %zu, \"buffer\": \"%s\", \"h2d_%s_ms\": %.3f, \"d2h_%s_ms\": %.3f, \"h2d_async_ms\": %.3f, \"d2h_async_ms\": %.3f, \"d2h_async_max_ms\": %.3f}\n",
                   bytes, kind == 0 ? "fresh" : kind == 1 ? "reused" : "reused", kind == 2 ? "staged" : "sync", med(hs), kind == 2 ? "staged" : "sync", med(ds),
                   med(ha), med(da), *std::max_element(da.begin(), da.end()));
        }
    }
    return 0;
}
