// Standalone reproducer (round 3; DESIGN.md 3.6, profiles/r03_packed_f32/): a kernel's RESULT depends on another stream's
// matrix-core kernel.  No library code.  What it shows on MI355X (gfx950, ROCm 7.2), 40 of 40 launches, three GPUs:
//
//   a packed-f32 VALU instruction whose SRC1 reads its register pair with the halves SWAPPED
//   (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[0,1] op_sel_hi:[1,0]) returns a wrong value in lanes 48..63
//   -- only there, the instruction's last pass -- while another wave runs v_mfma_i32_16x16x64_i8 on the chip.
//
//   * never quiet, never beside a plain VALU loop, not (any more) beside a v_mfma_f32_16x16x32_f16 loop;
//   * 1, 2, 4 or 8 wait states (s_nop / v_nop) between the producing and the reading instruction change nothing:
//     not a missing-wait-state hazard the compiler could pad;
//   * the same instructions without op_sel, with the swap on SRC0, or with src1's low half used twice
//     (op_sel_hi:[1,0], the broadcast form) are clean; so is every packed f16 form, including the library's
//     `v_pk_add_f16 v, v, s op_sel_hi:[1,0]` and f16 forms with swapped src1 halves.
//
// hipcc's SLP vectorizer makes exactly this instruction from a balanced sum of four squares -- (v0^2 + v1^2) + (v2^2 + v3^2)
// becomes v_pk_mul, v_pk_mul, v_pk_add ... op_sel:[0,1] op_sel_hi:[1,0] -- which is how the fused decoder's RMSNorm
// prologue came to lose v0^2 in a quarter of its lanes whenever a prompt's int8 score MFMAs ran beside it (round 2).
// build.py keeps packed-f32 instructions out of every kernel (they are half rate on gfx950 anyway);
// tests/test_no_packed_f32_cpu.py reads the generated code.
//
// Victims (repro_victim.inc, compiled twice: repro_pk.hip / repro_scalar.hip):
//   k_victim_*        the prologue's shape in plain C++ (no op_sel form comes out of hipcc here: clean)
//   k_victim_*_loads  the decoder's own instruction sequence and seven variations of it, eight 16-byte loads in flight
//   k_probe_ops       eight packed forms, each fed from registers
//   k_ld_*            a register holding a SENTINEL is overwritten by a global load and read by the very next instruction
//                     behind s_waitcnt vmcnt(0): no sentinel ever comes back -- it is not a load that is read too early
// Build + run: bash tools/mfma_neighbour/run_repro.sh [launches per case]   (on the GPU box; ~3 s)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" __global__ void k_victim_pk(const float*, const float*, float*, float*, float*);
extern "C" __global__ void k_victim_scalar(const float*, const float*, float*, float*, float*);
extern "C" __global__ void k_victim_pk_loads(const float*, const float*, const float4*, size_t, float*, float*);
extern "C" __global__ void k_victim_scalar_loads(const float*, const float*, const float4*, size_t, float*, float*);

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define SENT 3.0f
// one dwordx2 per lane, consumed by v_pk_mul_f32 right behind the wait
__global__ __launch_bounds__(512) void k_ld_pk(const float* __restrict__ x, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    const float* p = x + 2 * (size_t)threadIdx.x;              // every workgroup reads the same 4 KB (L2 / L1 hits, like the decoder's rows)
    f2 d = {SENT, SENT}, r;
    asm volatile("global_load_dwordx2 %0, %2, off\n\ts_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %1, %0, %0"
                 : "+v"(d), "=v"(r) : "v"(p) : "memory");
    out[2 * i] = r.x; out[2 * i + 1] = r.y;
}
__global__ __launch_bounds__(512) void k_ld_mul(const float* __restrict__ x, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    const float* p = x + threadIdx.x;
    float d = SENT, r;
    asm volatile("global_load_dword %0, %2, off\n\ts_waitcnt vmcnt(0)\n\tv_mul_f32 %1, %0, %0"
                 : "+v"(d), "=v"(r) : "v"(p) : "memory");
    out[i] = r;
}
// dwordx4 per lane (the decoder's own load width), squares formed by the compiler right behind the wait
__global__ __launch_bounds__(512) void k_ld4(const float* __restrict__ x, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    const float* p = x + 4 * (size_t)threadIdx.x;
    f4 d = {SENT, SENT, SENT, SENT};
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(d) : "v"(p) : "memory");
    const f4 r = d * d;
    ((f4*)out)[i] = r;
}


// Which packed instructions are affected?  Eight forms, each fed from registers, each result stored raw (8 dwords per
// thread: two per form would double the stores -- the low and the high half are folded into one word by xor).
__global__ __launch_bounds__(512) void k_probe_ops(const float* __restrict__ raw, const float* __restrict__ res, const float4* __restrict__ big,
                                                   size_t big_stride, float* __restrict__ out, float* __restrict__ out_sink)
{
    const int t = threadIdx.x;
    const float4 a = ((const float4*)raw)[t], b = ((const float4*)res)[t];
    float4 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = big[((size_t)blockIdx.x * 8 + k) * big_stride + t];
    __builtin_amdgcn_sched_barrier(0);
    const f2 x01 = {b.x + a.x, b.y + a.y}, x23 = {b.z + a.z, b.w + a.w};
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h01 = {(_Float16)x01.x, (_Float16)x01.y}, h23 = {(_Float16)x23.x, (_Float16)x23.y};
    unsigned r[8];
    const unsigned sconst = 0x3c003800u;                      // halves (0.5, 1.0) in an SGPR
    f2 o; h2 oh;
#define F32OP(K, TXT) asm volatile(TXT "\n\ts_nop 1" : "=&v"(o) : "v"(x01), "v"(x23)); r[K] = __float_as_uint(o.x) ^ (__float_as_uint(o.y) * 3u)
#define F16OP(K, TXT) asm volatile(TXT "\n\ts_nop 1" : "=&v"(oh) : "v"(h01), "v"(h23), "s"(sconst)); r[K] = __builtin_bit_cast(unsigned, oh)
    F16OP(0, "v_pk_add_f16 %0, %1, %3 op_sel_hi:[1,0]");                     // the library's form: constant in an SGPR, its low half for both
    F16OP(1, "v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]");
    F16OP(2, "v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]");
    F16OP(3, "v_pk_mul_f16 %0, %1, %2");
    F32OP(4, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]");
    F32OP(5, "v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]");
    F32OP(6, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]");                     // src1's low half for both results (a broadcast)
    F32OP(7, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]");        // the failing form, without the multiplies in front
#pragma unroll
    for (int k = 0; k < 8; k++) out[((size_t)blockIdx.x * 512 + t) * 8 + k] = __uint_as_float(r[k]);
    float sink = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) sink += w[k].x + w[k].y + w[k].z + w[k].w;
    if (sink == 12345.f) out_sink[0] = sink;
}

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
typedef int i4 __attribute__((ext_vector_type(4)));
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
__global__ __launch_bounds__(256) void k_valu(float* out, int spin)
{
    float v = threadIdx.x;
    for (int r = 0; r < spin * 64; r++) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    if (v == 12345.f) out[0] = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static hipStream_t s_victim, s_nb;
static float* d_nb;
static void neighbour(int kind)
{
    for (int i = 0; i < 12; i++) {
        if (kind == 1) k_mfma<<<512, 256, 0, s_nb>>>(d_nb, 320);
        else if (kind == 2) k_mfma_i8<<<512, 256, 0, s_nb>>>(d_nb, 320);
        else if (kind == 3) k_valu<<<512, 256, 0, s_nb>>>(d_nb, 40);
    }
}

struct Case { const char* name; int words_per_thread; void (*launch)(const float*, const float*, float*, int); };
static const int G = 160;
static float *d_raw, *d_res;
static void l_vpk(const float* a, const float* b, float* o, int) { k_victim_pk<<<G, 512, 0, s_victim>>>(a, b, o, o + G * 512, o + G * 520); }
static void l_vsc(const float* a, const float* b, float* o, int) { k_victim_scalar<<<G, 512, 0, s_victim>>>(a, b, o, o + G * 512, o + G * 520); }
static float4* d_big; static const size_t BIG_STRIDE = 1 << 20;   // float4s between a lane's loads: 16 MB apart, 160 x 8 rows = 20 GB of address range? no: see main
static int g_launch;
static void l_vpkl(const float* a, const float* b, float* o, int) { k_victim_pk_loads<<<G, 512, 0, s_victim>>>(a, b, d_big + (size_t)(g_launch++ % 4) * 512, 2048, o, o + (size_t)G * 512 * 8); }
static void l_vscl(const float* a, const float* b, float* o, int) { k_victim_scalar_loads<<<G, 512, 0, s_victim>>>(a, b, d_big + (size_t)(g_launch++ % 4) * 512, 2048, o, o + (size_t)G * 512 * 8); }
static void l_ops(const float* a, const float* b, float* o, int) { k_probe_ops<<<G, 512, 0, s_victim>>>(a, b, d_big + (size_t)(g_launch++ % 4) * 512, 2048, o, o + (size_t)G * 512 * 8); }
static void l_ldpk(const float* a, const float*, float* o, int) { k_ld_pk<<<G, 512, 0, s_victim>>>(a, o); }
static void l_ldmul(const float* a, const float*, float* o, int) { k_ld_mul<<<G, 512, 0, s_victim>>>(a, o); }
static void l_ld4(const float* a, const float*, float* o, int) { k_ld4<<<G, 512, 0, s_victim>>>(a, o); }

int main(int argc, char** argv)
{
    const int trials = argc > 1 ? atoi(argv[1]) : 40;
    CK(hipStreamCreateWithFlags(&s_victim, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_nb, hipStreamNonBlocking));
    CK(hipMalloc(&d_nb, 64));
    std::vector<float> raw(2048), res(2048);
    unsigned s = 12345u;
    for (int i = 0; i < 2048; i++) { s = s * 1664525u + 1013904223u; raw[i] = ((s >> 8) & 0xffff) / 65536.0f * 4.f - 2.f; s = s * 1664525u + 1013904223u; res[i] = ((s >> 8) & 0xffff) / 65536.0f * 8.f - 4.f; }
    CK(hipMalloc(&d_raw, 8192)); CK(hipMalloc(&d_res, 8192));
    CK(hipMemcpy(d_raw, raw.data(), 8192, hipMemcpyHostToDevice)); CK(hipMemcpy(d_res, res.data(), 8192, hipMemcpyHostToDevice));
    const size_t out_words = (size_t)G * 512 * 8 + 4096;
    float* d_out; CK(hipMalloc(&d_out, out_words * 4));
    std::vector<float> ref(out_words), got(out_words);
    CK(hipMalloc(&d_big, ((size_t)G * 8 * 2048 + 4096) * 16));        // 160 x 8 rows of 2048 float4 (42 MB: more than the L2s hold)
    CK(hipMemset(d_big, 0, ((size_t)G * 8 * 2048 + 4096) * 16));
    const Case cases[] = {{"victim, packed f32 (plain -O3)   ", 1, l_vpk}, {"victim, scalar f32               ", 1, l_vsc},
                          {"victim + loads in flight, packed ", 8, l_vpkl}, {"victim + loads in flight, scalar ", 8, l_vscl},
                          {"probe: eight packed forms        ", 8, l_ops},
                          {"probe: dwordx2 -> v_pk_mul_f32   ", 2, l_ldpk}, {"probe: dword   -> v_mul_f32      ", 1, l_ldmul},
                          {"probe: dwordx4 -> squares (hipcc)", 4, l_ld4}};
    int any = 0;
    for (const Case& c : cases) {
        // expected values: a quiet run (and, for the probes, no sentinel anywhere)
        CK(hipMemsetAsync(d_out, 0, out_words * 4, s_victim));
        c.launch(d_raw, d_res, d_out, 0);
        CK(hipStreamSynchronize(s_victim));
        CK(hipMemcpy(ref.data(), d_out, out_words * 4, hipMemcpyDeviceToHost));
        for (int nbk = 0; nbk <= 3; nbk++) {
            int bad_launches = 0; long bad_words = 0, by_q[4] = {0, 0, 0, 0}, sentinels = 0, by_k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = 0; t < trials; t++) {
                CK(hipMemsetAsync(d_out, 0, out_words * 4, s_victim));
                CK(hipStreamSynchronize(s_victim));
                if (nbk) neighbour(nbk);
                c.launch(d_raw, d_res, d_out, 0);
                CK(hipStreamSynchronize(s_victim));
                CK(hipStreamSynchronize(s_nb));
                CK(hipMemcpy(got.data(), d_out, out_words * 4, hipMemcpyDeviceToHost));
                long w = 0;
                const size_t per_thread_words = (size_t)G * 512 * c.words_per_thread;
                for (size_t i = 0; i < per_thread_words; i++)
                    if (memcmp(&got[i], &ref[i], 4)) {
                        w++;
                        by_q[((i / c.words_per_thread) & 63) >> 4]++;
                        by_k[i % c.words_per_thread]++;
                        if (got[i] == SENT * SENT) sentinels++;
                    }
                for (size_t i = per_thread_words; i < out_words; i++) if (memcmp(&got[i], &ref[i], 4)) w++;
                bad_words += w; bad_launches += w != 0;
            }
            printf("%s %-24s: %2d of %d launches differ; %ld words; per-thread values by lane quarter [0-15 16-31 32-47 48-63] = [%ld %ld %ld %ld]; sentinel^2 seen %ld times\n",
                   c.name, nbk == 0 ? "quiet" : nbk == 1 ? "beside an f16 MFMA loop" : nbk == 2 ? "beside an i8 MFMA loop" : "beside a VALU loop", bad_launches, trials, bad_words, by_q[0], by_q[1], by_q[2], by_q[3], sentinels);
            if (c.launch == l_ops && bad_words)
                printf("      by form [0 pk_add_f16 v,v,s op_sel_hi:[1,0] | 1 pk_add_f16 src1 swapped | 2 pk_mul_f16 src1 swapped | 3 pk_mul_f16 plain | 4 pk_mul_f32 src1 swapped | 5 pk_fma_f32 src1 swapped | 6 pk_add_f32 src1 low half twice | 7 pk_add_f32 src1 swapped] = [%ld %ld %ld %ld %ld %ld %ld %ld]\n",
                       by_k[0], by_k[1], by_k[2], by_k[3], by_k[4], by_k[5], by_k[6], by_k[7]);
            else if (c.words_per_thread == 8 && bad_words)
                printf("      by variant [0 exact | 1 two wait states | 2 four | 3 no op_sel | 4 v_nop | 5 op_sel on src0 | 6 eight | 7 exact] = [%ld %ld %ld %ld %ld %ld %ld %ld]\n",
                       by_k[0], by_k[1], by_k[2], by_k[3], by_k[4], by_k[5], by_k[6], by_k[7]);
            if (nbk && bad_launches) any = 1;
        }
    }
    printf(any ? "RESULT: a kernel's result depended on the neighbour stream\n" : "RESULT: every launch returned the quiet run's bits\n");
    return 0;
}
