// Host-side launch throughput: T threads, each with its own stream, launching a near-empty kernel N times.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
__global__ void tiny(float* p, int n) { if (threadIdx.x == 0 && n < 0) p[blockIdx.x] = 1.f; }
int main()
{
    float* d; hipMalloc(&d, 4096);
    const int N = 20000;
    for (int T : {1, 2, 4, 8, 12, 16}) {
        std::vector<hipStream_t> ss(T);
        for (auto& s : ss) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, ss[t], d, i);
                hipStreamSynchronize(ss[t]);
            });
        for (auto& x : th) x.join();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %2d: %.0f launches/s total (%.2f us per launch per thread)\n", T, T * N / sec, sec / N * 1e6);
        for (auto& s : ss) hipStreamDestroy(s);
    }
    return 0;
}
