// Microbenchmark: N small dependent kernels on one stream, launched one by one against one hipGraphLaunch of the captured sequence.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/graph_launch tools/micro/graph_launch.hip && /tmp/graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void step(float* p, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 0.5f + 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int n = 60800 * 5;
    float* p; hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int N : {8, 40, 70}) {
        for (int i = 0; i < 200; ++i) step<<<(n + 255) / 256, 256, 0, s>>>(p, n);
        hipStreamSynchronize(s);
        const int reps = 50;
        double t0 = now();
        for (int r = 0; r < reps; ++r) for (int i = 0; i < N; ++i) step<<<(n + 255) / 256, 256, 0, s>>>(p, n);
        double t_enq = now() - t0;
        hipStreamSynchronize(s);
        double t_eager = now() - t0;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < N; ++i) step<<<(n + 255) / 256, 256, 0, s>>>(p, n);
        hipStreamEndCapture(s, &g);
        double tc = now();
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        double t_inst = now() - tc;
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t0 = now();
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
        double g_enq = now() - t0;
        hipStreamSynchronize(s);
        double t_graph = now() - t0;
        printf("N=%d kernels: eager %.1f us per sequence (host enqueue %.1f), graph replay %.1f us (host enqueue %.1f), instantiate %.0f us\n", N,
               t_eager / reps * 1e6, t_enq / reps * 1e6, t_graph / reps * 1e6, g_enq / reps * 1e6, t_inst * 1e6);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
