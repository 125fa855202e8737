// dev: host cost and end-to-end time of a chain of N small dependent kernels, launched one by one or as a hipGraph
// (two streams, fork/join).   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_launch tools/ubench/graph_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 64, n = 1 << 16;
    float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    auto body = [&]() {
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s1, a, n);
        hipEventRecord(e1, s1); hipStreamWaitEvent(s2, e1, 0);
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s1, a, n);
        for (int i = 0; i < N - 16; i++) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s2, b, n);
        hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0);
    };
    for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        double t0 = now(); body(); double t1 = now(); CK(hipStreamSynchronize(s1)); double t2 = now();
        printf("launches: enqueue %.0f us, total %.0f us\n", t1 - t0, t2 - t0);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal)); body(); CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 5; rep++) {
        CK(hipDeviceSynchronize());
        double t0 = now(); CK(hipGraphLaunch(ge, s1)); double t1 = now(); CK(hipStreamSynchronize(s1)); double t2 = now();
        printf("graph: launch %.0f us, total %.0f us\n", t1 - t0, t2 - t0);
    }
    // parameter update cost
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    int upd = 0; double t0 = now();
    for (size_t i = 0; i < nn && upd < 4; i++) {
        hipGraphNodeType ty; CK(hipGraphNodeGetType(nodes[i], &ty));
        if (ty != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams kp; CK(hipGraphKernelNodeGetParams(nodes[i], &kp));
        CK(hipGraphExecKernelNodeSetParams(ge, nodes[i], &kp)); upd++;
    }
    printf("%d node updates: %.1f us (%zu nodes)\n", upd, now() - t0, nn);
    { CK(hipDeviceSynchronize()); double t0 = now(); CK(hipGraphLaunch(ge, s1)); double t1 = now(); CK(hipStreamSynchronize(s1)); printf("graph after update: launch %.0f us, total %.0f us\n", t1 - t0, now() - t0); }
    return 0;
}
