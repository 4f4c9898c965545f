// Does a hipGraph replay of a dependent kernel chain cost less per kernel than plain launches?  600 kernels (the launches of one
// 100-step sampling pass) of the two shapes, empty and with one dependent load + store, launched plainly and as one graph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
#include <chrono>
__global__ void k_empty(float *a, const float *b) {}
__global__ void k_load1(float *a, const float *b) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    a[t] = b[t] + 1.f;
}
typedef void (*kern_t)(float *, const float *);
int main() {
    const int n = 493 * 256, NK = 600;
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    struct { const char *name; kern_t k; } ks[] = {{"empty", k_empty}, {"load1", k_load1}};
    for (auto &k : ks) {
        auto chain = [&]() {
            float *x = a, *y = b;
            for (int i = 0; i < NK; i++) {
                if (i % 6 == 1 || i % 6 == 3 || i % 6 == 5) hipLaunchKernelGGL(k.k, dim3(47), dim3(512), 0, s, x, y);
                else hipLaunchKernelGGL(k.k, dim3(493), dim3(256), 0, s, x, y);
                std::swap(x, y);
            }
        };
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms;
        chain(); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < 5; r++) chain();
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-6s plain launches: %.2f us per kernel\n", k.name, ms * 1e3 / (5 * NK));
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        chain();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < 5; r++) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-6s graph replay:   %.2f us per kernel\n", k.name, ms * 1e3 / (5 * NK));
    }
    // a SHORT graph (the six launches of one network evaluation) replayed 100 times, and what capture + instantiate cost
    for (int nk : {6, 60, 600}) {
        auto t0 = std::chrono::steady_clock::now();
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        float *x = a, *y = b;
        for (int i = 0; i < nk; i++) {
            if (i % 6 == 1 || i % 6 == 3 || i % 6 == 5) hipLaunchKernelGGL(k_load1, dim3(47), dim3(512), 0, s, x, y);
            else hipLaunchKernelGGL(k_load1, dim3(493), dim3(256), 0, s, x, y);
            std::swap(x, y);
        }
        hipStreamEndCapture(s, &g);
        auto t1 = std::chrono::steady_clock::now();
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        auto t2 = std::chrono::steady_clock::now();
        const int reps = 600 / nk;
        for (int r = 0; r < reps; r++) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto h0 = std::chrono::steady_clock::now();
        hipEventRecord(e0, s);
        for (int p = 0; p < 5; p++)
            for (int r = 0; r < reps; r++) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        auto h1 = std::chrono::steady_clock::now();
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("graph of %3d kernels: capture %.0f us, instantiate %.0f us; replayed %3d x: %.2f us per kernel on the GPU, %.1f us of host time per hipGraphLaunch\n",
               nk, std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count(), reps,
               ms * 1e3 / (5 * 600), std::chrono::duration<double, std::micro>(h1 - h0).count() / (5 * reps));
    }
    return 0;
}
