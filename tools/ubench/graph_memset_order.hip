// Round-2 anomaly: a hipMemsetAsync captured inside a compress() body "did not clear its buffer on replay".
// The bodies reuse workspace: the buffer that is cleared had been scratch of an earlier kernel of the SAME body.  This
// program captures exactly that shape -- kernel A scribbles on buf, memset(buf, 0), kernel B copies buf to out -- as a stream
// capture, replays the graph several times and checks out == 0.  A non-zero count on replay means the memset node does not
// take its place in the captured order between A and B (it ran before A, or not at all); variants: small / large fills,
// eager stream order as the control.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
__global__ void scribble(unsigned* p, size_t n, unsigned v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i;
}
__global__ void copy(const unsigned* a, unsigned* b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
static int run(size_t n, bool use_graph)
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *buf, *out;
    CK(hipMalloc(&buf, n * 4));
    CK(hipMalloc(&out, n * 4));
    std::vector<unsigned> h(n);
    hipGraphExec_t ex = nullptr;
    hipGraph_t g = nullptr;
    long bad_total = 0;
    for (int rep = 0; rep < 6; ++rep) {
        if (use_graph && rep == 1) CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        if (!use_graph || rep <= 1) {
            hipLaunchKernelGGL(scribble, dim3(256), dim3(256), 0, s, buf, n, 0xA5000000u + rep);
            CK(hipMemsetAsync(buf, 0, n * 4, s));
            hipLaunchKernelGGL(copy, dim3(256), dim3(256), 0, s, (const unsigned*)buf, out, n);
        }
        if (use_graph && rep == 1) {
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        }
        if (use_graph && rep >= 1) CK(hipGraphLaunch(ex, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (size_t i = 0; i < n; ++i) bad += h[i] != 0;
        printf("  n=%zu %s rep %d (%s): %ld of %zu words not cleared\n", n, use_graph ? "graph" : "eager", rep,
               !use_graph || rep == 0 ? "stream order" : (rep == 1 ? "captured + first launch" : "replay"), bad, n);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
int main()
{
    alarm(60);
    int rc = 0;
    for (size_t n : {(size_t)1, (size_t)64, (size_t)4096, (size_t)1 << 20}) {
        rc |= run(n, false);
        rc |= run(n, true);
    }
    printf(rc ? "memset nodes lost their place on replay\n" : "every fill cleared its buffer\n");
    return rc;
}
