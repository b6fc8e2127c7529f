// Minimal form of the round-2 "hipFree never returns" report.
// Under hipDeviceScheduleBlockingSync, hipFree's implicit device synchronise (ihipFree -> Device::SyncAllStreams ->
// HostQueue::finish) waits for the LAST COMMAND of every stream.  When that command is a marker without a hardware signal --
// an event record on the idle legacy NULL stream, or a stream-wait on an event that had already completed -- the runtime
// takes its "No HW event ... await command completion" branch and sleeps on a condition variable that only a completion
// callback signals.  This program builds exactly that tail: blocking sync, a side stream ordered behind the NULL stream by
// event record + stream wait, a kernel, a synchronise, then hipMalloc / hipFree.  It prints which step it reached; an
// alarm ends it if a step does not return (exit by SIGALRM = the hang reproduced).
//   ./blocking_sync_free [blocking=1] [markers=1] [rounds=50]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
__global__ void k(float* p) { p[threadIdx.x] += 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
int main(int argc, char** argv)
{
    const int blocking = argc > 1 ? atoi(argv[1]) : 1, markers = argc > 2 ? atoi(argv[2]) : 1, rounds = argc > 3 ? atoi(argv[3]) : 50;
    alarm(40);
    if (blocking) CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    hipStream_t s;
    hipEvent_t ev, done;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&done, hipEventBlockingSync | hipEventDisableTiming));
    float* buf;
    CK(hipMalloc(&buf, 1 << 20));
    for (int r = 0; r < rounds; ++r) {
        if (markers) {
            CK(hipEventRecord(ev, nullptr));       // marker at the tail of the (idle) NULL stream
            CK(hipStreamWaitEvent(s, ev, 0));      // marker on the side stream, its dependency already complete
        }
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, buf);
        CK(hipEventRecord(done, s));
        CK(hipEventSynchronize(done));
        float* tmp;
        CK(hipMalloc(&tmp, 1 << 22));
        printf("round %d: hipFree ...\n", r);
        fflush(stdout);
        CK(hipFree(tmp));                          // implicit device synchronise
    }
    printf("blocking=%d markers=%d: %d rounds of hipFree returned\n", blocking, markers, rounds);
    return 0;
}
