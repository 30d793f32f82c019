// host cost of the event / stream API calls a small proof makes ~40 of: hipcc -O2 tools/ubench_event.hip -o tools/_bin/ubench_event
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_nop() {}
int main() {
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int N = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    hipEvent_t e[N];
    auto t0 = now();
    for (int i = 0; i < N; i++) hipEventCreateWithFlags(&e[i], hipEventDisableTiming);
    auto t1 = now();
    for (int i = 0; i < N; i++) hipEventRecord(e[i], s);
    auto t2 = now();
    for (int i = 0; i < N; i++) hipStreamWaitEvent(s2, e[i], 0);
    auto t3 = now();
    hipDeviceSynchronize();
    auto t4 = now();
    for (int i = 0; i < N; i++) hipEventDestroy(e[i]);
    auto t5 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_nop, 1, 64, 0, s);
    auto t6 = now();
    hipDeviceSynchronize();
    auto t7 = now();
    for (int i = 0; i < 200; i++) { hipLaunchKernelGGL(k_nop, 1, 64, 0, s); hipStreamSynchronize(s); }
    auto t8 = now();
    printf("{\"event_create_us\": %.2f, \"event_record_us\": %.2f, \"stream_wait_event_us\": %.2f, \"event_destroy_us\": %.2f, \"launch_us\": %.2f, \"launch_plus_sync_us\": %.2f}\n",
           us(t0, t1) / N, us(t1, t2) / N, us(t2, t3) / N, us(t4, t5) / N, us(t5, t6) / N, us(t7, t8) / 200);
    return 0;
}
