// microbenchmark: what does a kernel boundary + a minimal GEMV-shaped kernel cost in a graph?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void k_empty(const int* step, float* out) { if (threadIdx.x == 0 && blockIdx.x == 0 && step[0] < 0) out[0] = 1.f; }

// each wave: R rows x 1 KiB loads, dot with constant, wave reduce, store
template <int R>
__global__ __launch_bounds__(256) void k_stream(const int* step, const uint4* __restrict__ w, float* out, int rows)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wid) * R;
    uint4 q[R];
#pragma unroll
    for (int j = 0; j < R; j++) q[j] = w[(size_t)(r0 + j < rows ? r0 + j : 0) * 64 + lane];
    const int n = step[0];
#pragma unroll
    for (int j = 0; j < R; j++) {
        float acc = (float)(int)(q[j].x ^ q[j].y ^ q[j].z ^ q[j].w) * (float)n;
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0 && r0 + j < rows) out[r0 + j] = acc;
    }
}

// same + an LDS prologue with two barriers and ~N dependent flops per thread
template <int R, int WORK>
__global__ __launch_bounds__(256) void k_stream_pro(const int* step, const uint4* __restrict__ w, const float* __restrict__ x, float* out, int rows)
{
    __shared__ float sh[2048 + 16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wid) * R;
    const float4 xa = ((const float4*)x)[threadIdx.x * 2], xb = ((const float4*)x)[threadIdx.x * 2 + 1];
    uint4 q[R];
#pragma unroll
    for (int j = 0; j < R; j++) q[j] = w[(size_t)(r0 + j < rows ? r0 + j : 0) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    float v[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
    float ss = 0.f;
    for (int it = 0; it < WORK; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) { v[i] = v[i] * 1.0001f + 0.5f; ss += v[i] * v[i]; }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) sh[2048 + wid] = ss;
    __syncthreads();
    ss = sh[2048] + sh[2049] + sh[2050] + sh[2051];
#pragma unroll
    for (int i = 0; i < 8; i++) sh[threadIdx.x * 8 + i] = v[i] * ss;
    __syncthreads();
    const float a0 = sh[lane * 32];
    const int n = step[0];
#pragma unroll
    for (int j = 0; j < R; j++) {
        float acc = (float)(int)(q[j].x ^ q[j].y ^ q[j].z ^ q[j].w) * (float)n * a0;
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0 && r0 + j < rows) out[r0 + j] = acc;
    }
}

template <typename F>
static double time_graph(hipStream_t st, int reps, F enqueue)
{
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.0 / reps;
}

int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int* step; float* out; float* x; CK(hipMalloc(&step, 64)); CK(hipMemset(step, 0, 64)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&x, 1 << 16));
    CK(hipMemset(x, 0, 1 << 16));
    // 600 MB of "weights": 22 layers x several matrices so that nothing is cache resident
    const size_t wbytes = 600ull << 20;
    uint4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
    const int K = 132;
    auto rep = [&](const char* name, double us) { printf("%-44s %8.1f us/graph  %6.2f us/kernel\n", name, us, us / K); };
    rep("empty 256 WG", time_graph(st, 20, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, step, out); }));
    rep("empty 1 WG", time_graph(st, 20, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, step, out); }));
    // stream 2560 rows x 1 KiB (= 2.6 MB) per kernel, different slice each kernel (cold)
    const int rows = 2560;
    const size_t stride = (size_t)rows * 64;   // uint4 per kernel
    rep("stream 2.6MB R=2 (320 WG)", time_graph(st, 10, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL((k_stream<2>), dim3(rows / 8), dim3(256), 0, st, step, w + stride * i, out, rows); }));
    rep("stream 2.6MB R=1 (640 WG)", time_graph(st, 10, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL((k_stream<1>), dim3(rows / 4), dim3(256), 0, st, step, w + stride * i, out, rows); }));
    rep("stream 2.6MB R=8 (80 WG)", time_graph(st, 10, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL((k_stream<8>), dim3(rows / 32), dim3(256), 0, st, step, w + stride * i, out, rows); }));
    const int rows2 = 11264;
    const size_t stride2 = (size_t)rows2 * 64;
    rep("stream 11.5MB R=8 (352 WG) x44", time_graph(st, 10, [&] { for (int i = 0; i < 44; i++) hipLaunchKernelGGL((k_stream<8>), dim3(rows2 / 32), dim3(256), 0, st, step, w + stride2 * i, out, rows2); }) * 3);
    rep("stream 11.5MB R=4 (704 WG) x44", time_graph(st, 10, [&] { for (int i = 0; i < 44; i++) hipLaunchKernelGGL((k_stream<4>), dim3(rows2 / 16), dim3(256), 0, st, step, w + stride2 * i, out, rows2); }) * 3);
    rep("stream+pro(work 0) 2.6MB R=2", time_graph(st, 10, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL((k_stream_pro<2, 0>), dim3(rows / 8), dim3(256), 0, st, step, w + stride * i, x, out, rows); }));
    rep("stream+pro(work 8 = 128 fma) 2.6MB R=2", time_graph(st, 10, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL((k_stream_pro<2, 8>), dim3(rows / 8), dim3(256), 0, st, step, w + stride * i, x, out, rows); }));
    rep("stream+pro(work 32 = 512 fma) 2.6MB R=2", time_graph(st, 10, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL((k_stream_pro<2, 32>), dim3(rows / 8), dim3(256), 0, st, step, w + stride * i, x, out, rows); }));
    rep("stream+pro(work 64 = 1024 fma) 2.6MB R=2", time_graph(st, 10, [&] { for (int i = 0; i < K; i++) hipLaunchKernelGGL((k_stream_pro<2, 64>), dim3(rows / 8), dim3(256), 0, st, step, w + stride * i, x, out, rows); }));
    return 0;
}
