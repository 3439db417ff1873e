// microbenchmark: a chain of dependent GEMV-shaped launches, (A) serialised by the stream, against
// (B) the same launches alternating between two streams with the dependency carried by an arrival
// counter in device memory, so that launch k+1 is resident and has requested its weights while launch k
// is still computing.  Every spin is bounded; a stall sets `abort_flag` and the remaining launches bail.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/flagchain tools/microbench_flag_chain.hip && /tmp/flagchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s (line %d)\n",#x,hipGetErrorString(e),__LINE__); exit(1);} }while(0)

struct Phase {
    const uint4* w;          // rows x 1 KiB
    const float* x;          // 2048 inputs (the previous launch's outputs)
    float* out;              // rows outputs (rows >= 2048)
    int rows;
    unsigned* my_flag;       // arrival counter of this launch slot (monotonic)
    const unsigned* prev_flag;
    unsigned my_wgs, prev_wgs;
    int prev_is_older;       // 1: the awaited launch belongs to the previous replay (first launch of a replay)
    unsigned* abort_flag;
};

template <int R, bool FLAGGED>
__global__ __launch_bounds__(256) void k_phase(const Phase p)
{
    __shared__ float sh[2048 + 16];
    __shared__ unsigned s_go;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wid) * R;
    uint4 q[R];
#pragma unroll
    for (int j = 0; j < R; j++) q[j] = p.w[(size_t)(r0 + j < p.rows ? r0 + j : 0) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    float v[8];
    if (FLAGGED) {
        if (threadIdx.x == 0) {
            // my own counter is quiescent: the previous launch of this slot finished before this one began
            const unsigned epoch = __hip_atomic_load(p.my_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / p.my_wgs + 1;
            const unsigned want = (epoch - (unsigned)p.prev_is_older) * p.prev_wgs;
            unsigned spins = 0, ok = 1;
            while (__hip_atomic_load(p.prev_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
                if (spins > (1u << 21)) { __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            }
            s_go = ok;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __hip_atomic_load(p.x + threadIdx.x * 8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const float4 xa = ((const float4*)p.x)[threadIdx.x * 2], xb = ((const float4*)p.x)[threadIdx.x * 2 + 1];
        v[0] = xa.x; v[1] = xa.y; v[2] = xa.z; v[3] = xa.w; v[4] = xb.x; v[5] = xb.y; v[6] = xb.z; v[7] = xb.w;
    }
    float ss = 0.f;
    for (int it = 0; it < 4; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) { v[i] = v[i] * 1.0001f + 0.5f; ss += v[i] * v[i]; }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) sh[2048 + wid] = ss;
    __syncthreads();
    ss = sh[2048] + sh[2049] + sh[2050] + sh[2051];
#pragma unroll
    for (int i = 0; i < 8; i++) sh[threadIdx.x * 8 + i] = v[i] * (1.0f / (1.0f + ss));
    __syncthreads();
    const float a0 = sh[lane * 32];
#pragma unroll
    for (int j = 0; j < R; j++) {
        float acc = (float)(int)((q[j].x ^ q[j].y ^ q[j].z ^ q[j].w) & 0xff) * a0;
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0 && r0 + j < p.rows) {
            if (FLAGGED) __hip_atomic_store(p.out + r0 + j, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else p.out[r0 + j] = acc;
        }
    }
    if (FLAGGED) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(p.my_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv)
{
    const int K = 132, R = 8;
    const int rows = argc > 1 ? atoi(argv[1]) : 8192;           // 8 MiB of weights per launch
    const int wgs = (rows + 4 * R - 1) / (4 * R);
    CK(hipSetDevice(0));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    // distinct weights per launch so that nothing is an L2 hit: K x rows x 1 KiB
    uint4* w; CK(hipMalloc(&w, (size_t)K * rows * 1024));
    CK(hipMemset(w, 1, (size_t)K * rows * 1024));
    float* buf[2]; CK(hipMalloc(&buf[0], rows * 4)); CK(hipMalloc(&buf[1], rows * 4));
    CK(hipMemset(buf[0], 0, rows * 4)); CK(hipMemset(buf[1], 0, rows * 4));
    unsigned* flags; CK(hipMalloc(&flags, (K + 1) * 64)); CK(hipMemset(flags, 0, (K + 1) * 64));
    unsigned* abort_flag = flags + K * 16;
    std::vector<Phase> ph(K);
    for (int k = 0; k < K; k++) {
        Phase& p = ph[k];
        p.w = w + (size_t)k * rows * 64; p.x = buf[k & 1]; p.out = buf[(k + 1) & 1]; p.rows = rows;
        p.my_flag = flags + k * 16; p.prev_flag = flags + ((k + K - 1) % K) * 16;
        p.my_wgs = wgs; p.prev_wgs = wgs; p.prev_is_older = (k == 0); p.abort_flag = abort_flag;
    }
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int reps = 20;
    auto report = [&](const char* name, float ms) {
        printf("%-44s %8.3f us per launch   (%7.1f us per %d-launch replay, %.0f GB/s)\n", name, ms * 1e3 / (reps * K), ms * 1e3 / reps, K,
               (double)K * rows * 1024 / (ms / reps * 1e-3) / 1e9);
    };

    // (A) one stream, plain launches, graph
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; k++) hipLaunchKernelGGL((k_phase<R, false>), dim3(wgs), dim3(256), 0, s0, ph[k]);
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report("A  serial graph, stream order", ms);
    }
    // (B) two streams, eager, dependency by arrival counters
    {
        auto run = [&]() { for (int k = 0; k < K; k++) hipLaunchKernelGGL((k_phase<R, true>), dim3(wgs), dim3(256), 0, (k & 1) ? s1 : s0, ph[k]); };
        for (int i = 0; i < 3; i++) run();
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; i++) run();
        CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report("B  two streams eager, arrival counters", ms);
        unsigned ab; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost)); printf("   abort flag = %u\n", ab);
    }
    // (C) the same as a graph with two parallel chains
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
        for (int k = 0; k < K; k++) hipLaunchKernelGGL((k_phase<R, true>), dim3(wgs), dim3(256), 0, (k & 1) ? s1 : s0, ph[k]);
        CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report("C  two-chain graph, arrival counters", ms);
        unsigned ab; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost)); printf("   abort flag = %u\n", ab);
    }
    return 0;
}
