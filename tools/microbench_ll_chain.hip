// microbenchmark: a chain of dependent GEMV-shaped launches (the shape of the batch-1 decode step: every launch needs the
// WHOLE output vector of the one before), three ways:
//   A  one stream, stream order (what the decoder's graph does today);
//   L  two streams / two parallel graph chains, launch k on chain k % 2, so launch k+1 is resident and has requested its
//      weights while launch k still computes; the vector is handed over IN THE DATA: every element travels as an 8-byte
//      {value, tag} pair written with an agent-scope store and polled with agent-scope loads (the "LL" protocol of
//      collective libraries) -- no counter, no fence, one store + one load on the critical path;
//   (microbench_flag_chain.hip measured the arrival-counter form: 12.4 us against 5.6 us per launch.)
// Every spin is bounded; a stall sets `abort_flag` and every later poll gives up at once, so the grid always drains.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/llchain tools/microbench_ll_chain.hip && /tmp/llchain [rows] [wg_threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s (line %d)\n",#x,hipGetErrorString(e),__LINE__); exit(1);} }while(0)

constexpr int D = 2048;      // vector length handed from launch to launch

struct Phase {
    const uint4* w;                      // rows x 1 KiB
    const float* x;  float* out;         // plain form
    const unsigned long long* xl;        // LL form: D pairs {value bits, tag}
    unsigned long long* outl;
    int rows, k, K;
    const unsigned* epoch;               // replay number (written before the chains fork)
    unsigned* abort_flag;
};

__global__ void k_epoch(unsigned* epoch) { if (threadIdx.x == 0) epoch[0] += 1; }

template <int R, int NT, bool LL>
__global__ __launch_bounds__(NT) void k_phase(const Phase p)
{
    constexpr int PER = D / NT;                               // elements of x per thread
    __shared__ float sh[D + 16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * (NT / 64) + wid) * R;
    uint4 q[R];
#pragma unroll
    for (int j = 0; j < R; j++) q[j] = p.w[(size_t)(r0 + j < p.rows ? r0 + j : 0) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    float v[PER];
    unsigned tag_out = 0;
    if (LL) {
        const unsigned e = p.epoch[0];
        const unsigned tag_in = e * (unsigned)p.K + (unsigned)p.k - 1u;
        tag_out = tag_in + 1u;
        unsigned spins = 0;
        for (;;) {
            unsigned long long pr[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) pr[i] = __hip_atomic_load(p.xl + i * NT + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int i = 0; i < PER; i++) { ok = ok && ((unsigned)(pr[i] >> 32) == tag_in); v[i] = __uint_as_float((unsigned)pr[i]); }
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (spins > (1u << 20)) { __hip_atomic_store(p.abort_flag, 1u + (unsigned)p.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = p.x[i * NT + threadIdx.x];
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) ss += v[i] * v[i];
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) sh[D + wid] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) ss += sh[D + i];
    const float sc = 1.0f / (1.0f + ss);
#pragma unroll
    for (int i = 0; i < PER; i++) sh[i * NT + threadIdx.x] = v[i] * sc;
    __syncthreads();
    float a[4];
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = sh[lane * 4 + i] + sh[256 + lane * 4 + i];
#pragma unroll
    for (int j = 0; j < R; j++) {
        float acc = (float)(int)(q[j].x & 0xff) * a[0] + (float)(int)(q[j].y & 0xff) * a[1] + (float)(int)(q[j].z & 0xff) * a[2] + (float)(int)(q[j].w & 0xff) * a[3];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        acc = acc * 1e-3f + 0.25f;
        if (lane == 0 && r0 + j < p.rows) {
            if (LL) __hip_atomic_store(p.outl + r0 + j, ((unsigned long long)tag_out << 32) | __float_as_uint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else p.out[r0 + j] = acc;
        }
    }
}

template <int R, int NT>
static void run_all(int rows, int K)
{
    const int wgs = (rows + (NT / 64) * R - 1) / ((NT / 64) * R);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    uint4* w; CK(hipMalloc(&w, (size_t)K * rows * 1024));       // distinct weights per launch: nothing is a cache hit
    CK(hipMemset(w, 1, (size_t)K * rows * 1024));
    float* buf[2]; unsigned long long* bufl[2];
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&buf[i], rows * 4)); CK(hipMalloc(&bufl[i], rows * 8)); }
    unsigned* words; CK(hipMalloc(&words, 256)); CK(hipMemset(words, 0, 256));
    unsigned* epoch = words, *abort_flag = words + 32;
    std::vector<Phase> ph(K);
    for (int k = 0; k < K; k++) {
        Phase& p = ph[k];
        p.w = w + (size_t)k * rows * 64; p.x = buf[k & 1]; p.out = buf[(k + 1) & 1]; p.xl = bufl[k & 1]; p.outl = bufl[(k + 1) & 1];
        p.rows = rows; p.k = k; p.K = K; p.epoch = epoch; p.abort_flag = abort_flag;
    }
    auto seed = [&]() {                                          // the vector the first launch of replay 1 reads
        std::vector<float> x(rows, 0.5f);
        std::vector<unsigned long long> xl(rows);
        for (int i = 0; i < rows; i++) { unsigned b; memcpy(&b, &x[i], 4); xl[i] = ((unsigned long long)(1u * K - 1u) << 32) | b; }
        CK(hipMemcpy(buf[0], x.data(), rows * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bufl[0], xl.data(), rows * 8, hipMemcpyHostToDevice));
        CK(hipMemset(buf[1], 0, rows * 4)); CK(hipMemset(bufl[1], 0, rows * 8));
        CK(hipMemset(words, 0, 256));
    };
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int reps = 20, warm = 3;
    auto report = [&](const char* name, float ms) {
        printf("%-52s %7.3f us per launch  (%7.1f us per %d-launch replay, %5.0f GB/s)\n", name, ms * 1e3 / (reps * K), ms * 1e3 / reps, K,
               (double)K * rows * 1024 / (ms / reps * 1e-3) / 1e9);
    };
    printf("rows %d (%.1f MB of weights per launch), %d threads x %d workgroups, %d launches per replay\n", rows, rows / 1024.0, NT, wgs, K);
    std::vector<float> ref(D), got(D);
    {   // A
        seed();
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_epoch, dim3(1), dim3(64), 0, s0, epoch);
        for (int k = 0; k < K; k++) hipLaunchKernelGGL((k_phase<R, NT, false>), dim3(wgs), dim3(NT), 0, s0, ph[k]);
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < warm; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report("A  one chain, stream order", ms);
        CK(hipMemcpy(ref.data(), buf[K & 1], D * 4, hipMemcpyDeviceToHost));
    }
    auto check = [&]() {
        std::vector<unsigned long long> o(D);
        CK(hipMemcpy(o.data(), bufl[K & 1], D * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < D; i++) { unsigned b = (unsigned)o[i]; float f; memcpy(&f, &b, 4); if (f != ref[i]) bad++; }
        unsigned ab; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
        printf("   abort flag = %u, %d of %d final values differ from A's\n", ab, bad, D);
    };
    {   // L, graph with two parallel chains
        seed();
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_epoch, dim3(1), dim3(64), 0, s0, epoch);
        CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
        for (int k = 0; k < K; k++) hipLaunchKernelGGL((k_phase<R, NT, true>), dim3(wgs), dim3(NT), 0, (k & 1) ? s1 : s0, ph[k]);
        CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < warm; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report("L  two-chain graph, {value, tag} pairs", ms);
        check();
    }
    {   // L, one chain: the cost of the pairs alone (no overlap)
        seed();
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_epoch, dim3(1), dim3(64), 0, s0, epoch);
        for (int k = 0; k < K; k++) hipLaunchKernelGGL((k_phase<R, NT, true>), dim3(wgs), dim3(NT), 0, s0, ph[k]);
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < warm; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report("L1 one chain, {value, tag} pairs (no overlap)", ms);
        check();
    }
    {   // L, eager on two streams
        seed();
        auto run = [&]() {
            hipLaunchKernelGGL(k_epoch, dim3(1), dim3(64), 0, s0, epoch);
            CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
            for (int k = 0; k < K; k++) hipLaunchKernelGGL((k_phase<R, NT, true>), dim3(wgs), dim3(NT), 0, (k & 1) ? s1 : s0, ph[k]);
            CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
        };
        for (int i = 0; i < warm; i++) run();
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; i++) run();
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report("L2 two streams eager, {value, tag} pairs", ms);
        check();
    }
    CK(hipFree(w)); for (int i = 0; i < 2; i++) { CK(hipFree(buf[i])); CK(hipFree(bufl[i])); } CK(hipFree(words));
    CK(hipStreamDestroy(s0)); CK(hipStreamDestroy(s1));
}

int main(int argc, char** argv)
{
    CK(hipSetDevice(0));
    const int K = 112;
    run_all<2, 512>(2048, K);        // o-projection sized: 2 MB per launch, 128 workgroups
    run_all<2, 512>(2560, K);        // q|k|v sized
    run_all<8, 512>(11264, K);       // gate|up sized: 11.5 MB per launch, 176 workgroups
    run_all<2, 256>(2048, K);
    return 0;
}
