// How much CPU time does a host thread spend waiting for the GPU?  (a) hipStreamSynchronize, (b) hipEventSynchronize on an
// event made with hipEventBlockingSync, (c) a pinned-memory read-back + (b), (d) hipStreamSynchronize after
// hipSetDeviceFlags(hipDeviceScheduleBlockingSync).  tools/exp: measurement only, not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#include <ctime>
__global__ void busy(long long cycles, int* out) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) { } if (out) *out = 1; }
static double cpu_s() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static double wall_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t eb, en; hipEventCreateWithFlags(&eb, hipEventDisableTiming | hipEventBlockingSync); hipEventCreateWithFlags(&en, hipEventDisableTiming);
    int* d; hipMalloc((void**)&d, 4); int* h; hipHostMalloc((void**)&h, 4, hipHostMallocDefault); int pageable = 0;
    const long long cyc = 30000000;   // 0.3 s at 100 MHz
    busy<<<1, 64, 0, s>>>(1000, d); hipStreamSynchronize(s);
    for (int mode = 0; mode < 9; ++mode) {
        if (mode == 5) { hipSetDeviceFlags(hipDeviceScheduleBlockingSync); hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
        if (mode == 7) { hipSetDeviceFlags(hipDeviceScheduleSpin); }
        double c0 = cpu_s(), w0 = wall_s();
        busy<<<1, 64, 0, s>>>(cyc, d);
        switch (mode) {
        case 0: hipStreamSynchronize(s); break;
        case 1: hipEventRecord(eb, s); hipEventSynchronize(eb); break;
        case 2: hipEventRecord(en, s); hipEventSynchronize(en); break;
        case 3: hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(eb, s); hipEventSynchronize(eb); break;
        case 4: hipMemcpyAsync(&pageable, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(eb, s); hipEventSynchronize(eb); break;
        case 5: hipStreamSynchronize(s); break;
        case 6: hipEventRecord(en, s); hipEventSynchronize(en); break;
        case 7: hipStreamSynchronize(s); break;
        case 8: hipEventRecord(eb, s); hipEventSynchronize(eb); break;
        }
        const char* names[] = { "hipStreamSynchronize", "blocking event", "plain event", "pinned copy + blocking event", "pageable copy + blocking event", "hipStreamSynchronize, stream made after hipDeviceScheduleBlockingSync", "plain event on that stream", "that stream after hipDeviceScheduleSpin again", "blocking event on that stream, flags Spin" };
        printf("%-58s wall %.3f s  cpu %.3f s\n", names[mode], wall_s() - w0, cpu_s() - c0);
    }
    return 0;
}
