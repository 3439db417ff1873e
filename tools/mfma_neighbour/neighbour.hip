// Test / experiment only -- NOT part of libgten_hip.so.  "Neighbour" kernels on a private stream, to see what running
// beside another stream's work does to a kernel's RESULTS (round 2: profiles/README.md "packed f32 beside MFMA").
//   kind 0  k_lds      64 KB of LDS filled with a NaN pattern and re-read
//   kind 1  k_mfma     v_mfma_f32_16x16x32_f16 on registers only, eight independent accumulators per wave (no memory at all)
//   kind 2  k_regs     192 VGPRs holding a NaN pattern
//   kind 3  k_valu     a dependent chain of v_fma_f32
//   kind 5  k_dot      v_dot4_i32_i8 chains
//   kind 6  k_mfma_i8  v_mfma_i32_16x16x64_i8 on registers only
//   kind 7  k_mfma with 64-thread workgroups (one wave)
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libneighbour.so neighbour.hip ; ctypes: nb_init(), nb_run(kind, launches,
// grid, spin), nb_sync().  tests/test_zz_neighbour_gpu.py builds and uses it.
#include <hip/hip_runtime.h>
#include <cstdint>
static hipStream_t g_s;
static float* g_out;
extern __shared__ unsigned g_l[];
__global__ __launch_bounds__(256) void k_lds(float* out, unsigned pattern, int words, int spin)
{
    for (int i = threadIdx.x; i < words; i += 256) g_l[i] = pattern;
    __syncthreads();
    unsigned acc = 0;
    for (int r = 0; r < spin; r++)
        for (int i = threadIdx.x; i < words; i += 256) { acc += g_l[i]; g_l[i] = pattern + (acc & 0); }
    if (acc == 12345u) out[0] = 1.f;
}
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void k_mfma(float* out, int spin)
{
    f4 c[8];
    for (int i = 0; i < 8; i++) c[i] = f4{0, 0, 0, 0};
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    for (int r = 0; r < spin; r++)
        for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mfma_i8(float* out, int spin)
{
    i4 c[8];
    for (int i = 0; i < 8; i++) c[i] = i4{0, 0, 0, 0};
    i4 a = i4{(int)threadIdx.x, 1, 2, 3}, b = i4{4, 5, 6, 7};
    for (int r = 0; r < spin; r++)
        for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    int s = 0;
    for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 12345) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_regs(float* out, unsigned pattern, int spin)
{
    float v[192];
    for (int i = 0; i < 192; i++) v[i] = __uint_as_float(pattern);
    for (int r = 0; r < spin; r++)
        for (int i = 0; i < 192; i++) v[i] = __builtin_fmaf(v[i], 1.0f, (float)(r & 0));
    float s = 0;
    for (int i = 0; i < 192; i++) s += v[i];
    if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_valu(float* out, int spin)
{
    float v = threadIdx.x;
    for (int r = 0; r < spin * 64; r++) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    if (v == 12345.f) out[0] = v;
}
__global__ __launch_bounds__(256) void k_dot(float* out, int spin)
{
    int acc[8];
    for (int i = 0; i < 8; i++) acc[i] = i;
    const int a = threadIdx.x * 0x01010101, b = 0x01020304;
    for (int r = 0; r < spin * 16; r++)
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_sdot4(a, b, acc[i], false);
    int s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    if (s == 12345) out[0] = 1.f;
}
extern "C" int nb_init()
{
    if (g_s) return 0;
    if (hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking) != hipSuccess) return 1;
    return hipMalloc(&g_out, 64) != hipSuccess;
}
extern "C" int nb_run(int kind, int launches, int grid, int spin)
{
    for (int i = 0; i < launches; i++) {
        if (kind == 0) k_lds<<<grid, 256, 65536, g_s>>>(g_out, 0x7fc00000u, 16384, spin);
        else if (kind == 1) k_mfma<<<grid, 256, 0, g_s>>>(g_out, spin * 8);
        else if (kind == 2) k_regs<<<grid, 256, 0, g_s>>>(g_out, 0x7fc00000u, spin);
        else if (kind == 3) k_valu<<<grid, 256, 0, g_s>>>(g_out, spin);
        else if (kind == 5) k_dot<<<grid, 256, 0, g_s>>>(g_out, spin);
        else if (kind == 6) k_mfma_i8<<<grid, 256, 0, g_s>>>(g_out, spin * 8);
        else if (kind == 7) k_mfma<<<grid, 64, 0, g_s>>>(g_out, spin * 8);
        else return 2;
    }
    return hipGetLastError() != hipSuccess;
}
extern "C" int nb_sync() { return hipStreamSynchronize(g_s) != hipSuccess; }
