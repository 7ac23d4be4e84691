// Micro-benchmark: issue cost of the instructions the front-end kernels lean on (gfx950, wave64).
//
// Every measured instruction is an `asm volatile` statement, so the compiler can neither fold, fuse, hoist nor
// re-select it (the round-1 version used C expressions; several of its rows were partly folded and one exceeded the
// chip's physical rate).  A wave runs ITERS x 32 copies of ONE instruction on 16 independent destination registers
// (dependent-issue latency is hidden: a register is rewritten every 16th instruction) between two s_memtime stamps.
// With W waves resident per SIMD (256 CUs x 4 SIMDs x W waves, all co-resident) the SIMD's issue interval for that
// instruction is  cycles(one wave) / (ITERS x 32) / W ... reported both ways, together with the wall-clock rate in
// T lane-ops/s (64 lanes per wave-instruction; a packed instruction still counts as ONE instruction).
// The physical ceiling if every wave64 instruction issued in 2 cycles: 256 x 4 x 32 x 2.4 GHz = 78.6 T lane-ops/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define ITERS 2048
#define PER_ITER 32

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* cyc, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x * 2654435761u;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
    __syncthreads();
    uint32_t a = seed ^ 0x9e3779b9u, b = seed * 7u + 1u;
    uint32_t addr = (threadIdx.x * 4u) & 0x3ffcu;                       // conflict-free dword address
    uint32_t baddr = (threadIdx.x * 37u + (threadIdx.x >> 3)) & 0x3fffu; // scattered byte address (FAST ring reads)
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITERS; it++) {
#define ONE(i, str) asm volatile(str : "+v"(r[i]) : "v"(a), "v"(b), "v"(r[(i + 5) & 15]), "v"(addr), "v"(baddr));
#define BODY(str) ONE(0, str) ONE(1, str) ONE(2, str) ONE(3, str) ONE(4, str) ONE(5, str) ONE(6, str) ONE(7, str) \
                  ONE(8, str) ONE(9, str) ONE(10, str) ONE(11, str) ONE(12, str) ONE(13, str) ONE(14, str) ONE(15, str)
        if (OP == 0)  { BODY("v_xor_b32 %0, %0, %1") BODY("v_xor_b32 %0, %0, %2") }
        if (OP == 1)  { BODY("v_add_u32 %0, %0, %1") BODY("v_add_u32 %0, %0, %2") }
        if (OP == 2)  { BODY("v_pk_sub_i16 %0, %0, %1") BODY("v_pk_sub_i16 %0, %0, %2") }
        if (OP == 3)  { BODY("v_pk_min_i16 %0, %0, %3") BODY("v_pk_min_i16 %0, %0, %3") }
        if (OP == 4)  { BODY("v_pk_max_i16 %0, %0, %3") BODY("v_pk_max_i16 %0, %0, %3") }
        if (OP == 5)  { BODY("v_perm_b32 %0, %0, %1, %2") BODY("v_perm_b32 %0, %0, %3, %2") }
        if (OP == 6)  { BODY("v_alignbit_b32 %0, %0, %3, 16") BODY("v_alignbit_b32 %0, %0, %3, 16") }
        if (OP == 7)  { BODY("v_alignbyte_b32 %0, %0, %3, 1") BODY("v_alignbyte_b32 %0, %0, %3, 1") }
        if (OP == 8)  { BODY("v_dot4_u32_u8 %0, %3, %1, %0") BODY("v_dot4_u32_u8 %0, %3, %2, %0") }
        if (OP == 9)  { BODY("v_dot2_u32_u16 %0, %3, %1, %0") BODY("v_dot2_u32_u16 %0, %3, %2, %0") }
        if (OP == 10) { BODY("v_lshl_or_b32 %0, %0, 16, %3") BODY("v_lshl_or_b32 %0, %0, 16, %3") }
        if (OP == 11) { BODY("v_and_or_b32 %0, %0, %1, %3") BODY("v_and_or_b32 %0, %0, %1, %3") }
        if (OP == 12) { BODY("v_bcnt_u32_b32 %0, %3, %0") BODY("v_bcnt_u32_b32 %0, %3, %0") }
        if (OP == 13) { BODY("v_min_u32 %0, %0, %3") BODY("v_max_u32 %0, %0, %3") }
        if (OP == 14) { BODY("v_mad_u32_u24 %0, %0, %1, %3") BODY("v_mad_u32_u24 %0, %0, %1, %3") }
        if (OP == 15) { BODY("v_mbcnt_lo_u32_b32 %0, %1, %0") BODY("v_mbcnt_hi_u32_b32 %0, %2, %0") }
        if (OP == 16) { BODY("v_cndmask_b32 %0, %0, %3, vcc") BODY("v_cndmask_b32 %0, %0, %3, vcc") }
        if (OP == 17) { BODY("v_pk_mul_lo_u16 %0, %0, %3") BODY("v_pk_mul_lo_u16 %0, %0, %3") }
        if (OP == 18) { BODY("v_pk_add_u16 %0, %0, %3") BODY("v_pk_add_u16 %0, %0, %3") }
        if (OP == 19) { BODY("v_bfe_u32 %0, %3, 8, 8") BODY("v_bfe_u32 %0, %3, 16, 8") }
        if (OP == 20) { BODY("v_mov_b32 %0, %3") BODY("v_mov_b32 %0, %3") }
        if (OP == 21) { BODY("v_add_f32 %0, %0, %3") BODY("v_add_f32 %0, %0, %3") }
        if (OP == 22) { BODY("v_fma_f32 %0, %0, %1, %3") BODY("v_fma_f32 %0, %0, %1, %3") }
        if (OP == 23) { BODY("v_pk_fma_f32 %0, %0, %1, %3") }              // 64-bit operands: handled below
        if (OP == 24) { BODY("ds_read_b32 %0, %4") BODY("ds_read_b32 %0, %4 offset:256") asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == 25) { BODY("ds_read_u8 %0, %5") BODY("ds_read_u8 %0, %5 offset:160") asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == 26) { BODY("v_max3_i32 %0, %0, %1, %3") BODY("v_min3_i32 %0, %0, %1, %3") }
        if (OP == 27) { BODY("v_sad_u8 %0, %0, %1, %3") BODY("v_sad_u8 %0, %0, %1, %3") }
        if (OP == 28) { BODY("v_cmp_gt_u32 vcc, %0, %3") BODY("v_cmp_gt_i32 vcc, %0, %3") }
        if (OP == 29) { BODY("v_pk_lshrrev_b16 %0, 3, %0") BODY("v_pk_ashrrev_i16 %0, 3, %0") }
        if (OP == 30) { BODY("v_add3_u32 %0, %0, %1, %3") BODY("v_or3_b32 %0, %0, %1, %3") }
        if (OP == 31) { BODY("v_cmp_gt_u16_sdwa vcc, %0, %3 src0_sel:BYTE_0 src1_sel:BYTE_1") BODY("v_and_b32_sdwa %0, %0, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD") }
        if (OP == 32) { BODY("v_pk_min_f16 %0, %0, %3") BODY("v_pk_max_f16 %0, %0, %3") }
        if (OP == 33) { BODY("v_pk_add_f16 %0, %0, %3") BODY("v_pk_add_f16 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]") }
        if (OP == 34) { BODY("v_pk_minimum3_f16 %0, %0, %1, %3") BODY("v_pk_maximum3_f16 %0, %0, %1, %3") }
        if (OP == 35) { BODY("v_min_f16 %0, %0, %3") BODY("v_max_f16 %0, %0, %3") }
        if (OP == 36) { BODY("v_min_f32 %0, %0, %3") BODY("v_max_f32 %0, %0, %3") }
        if (OP == 37) { BODY("v_min3_f32 %0, %0, %1, %3") BODY("v_max3_f32 %0, %0, %1, %3") }
        if (OP == 38) { BODY("v_and_b32 %0, %0, %3") BODY("v_or_b32 %0, %0, %3") }
        if (OP == 39) { BODY("v_lshlrev_b32 %0, 3, %0") BODY("v_lshrrev_b32 %0, 3, %0") }
        if (OP == 40) { BODY("v_pk_fma_f16 %0, %0, %1, %3") BODY("v_pk_mul_f16 %0, %0, %3") }
        if (OP == 41) { BODY("v_sub_u32 %0, %0, %3") BODY("v_subrev_u32 %0, %0, %3") }
        if (OP == 42) { BODY("v_min3_f16 %0, %0, %1, %3") BODY("v_max3_f16 %0, %0, %1, %3") }
        if (OP == 43) { BODY("v_pk_min_u16 %0, %0, %3") BODY("v_pk_max_u16 %0, %0, %3") }
#undef BODY
#undef ONE
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// v_pk_fma_f32 needs 64-bit register pairs: its own kernel
__global__ __launch_bounds__(256) void k_pkfma(float* out, unsigned long long* cyc, float seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = (f2){seed * (i + 1), seed + threadIdx.x};
    f2 a = {1.0001f, 0.9999f}, b = {1e-3f, -1e-3f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    f2 acc = {0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) acc += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static uint32_t* g_out; static unsigned long long* g_cyc;

template <typename L>
static void measure(const char* name, L launch)
{
    for (int w : {1, 2, 4, 8}) {
        const int blocks = 256 * w;                                    // 256-thread blocks: one wave per SIMD each
        launch(blocks);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        launch(blocks);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks * 4);
        hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        const double n = (double)ITERS * PER_ITER;
        const double med = (double)c[c.size() / 2];
        printf("%-34s W=%d  %6.2f clk/instr seen by a wave  %5.2f clk SIMD issue interval  %7.3f ms  %6.2f Tlane-op/s\n",
               name, w, med / n, med / n / w, ms, n * 64.0 * blocks * 4 / (ms * 1e-3) / 1e12);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

#define RUN(op, name) measure(name, [](int blocks) { hipLaunchKernelGGL(k<op>, dim3(blocks), dim3(256), 0, 0, g_out, g_cyc, 12345u); })

int main()
{
    hipMalloc(&g_out, 256 * 8 * 256 * sizeof(uint32_t)); hipMalloc(&g_cyc, 256 * 8 * 4 * 8);
    RUN(0, "v_xor_b32"); RUN(1, "v_add_u32"); RUN(20, "v_mov_b32"); RUN(13, "v_min_u32 / v_max_u32"); RUN(26, "v_max3_i32 / v_min3_i32");
    RUN(30, "v_add3_u32 / v_or3_b32"); RUN(10, "v_lshl_or_b32"); RUN(11, "v_and_or_b32"); RUN(19, "v_bfe_u32");
    RUN(5, "v_perm_b32"); RUN(6, "v_alignbit_b32"); RUN(7, "v_alignbyte_b32"); RUN(12, "v_bcnt_u32_b32"); RUN(15, "v_mbcnt_lo / v_mbcnt_hi");
    RUN(14, "v_mad_u32_u24"); RUN(27, "v_sad_u8"); RUN(8, "v_dot4_u32_u8"); RUN(9, "v_dot2_u32_u16");
    RUN(2, "v_pk_sub_i16"); RUN(18, "v_pk_add_u16"); RUN(3, "v_pk_min_i16"); RUN(4, "v_pk_max_i16"); RUN(17, "v_pk_mul_lo_u16"); RUN(29, "v_pk_lshrrev_b16 / v_pk_ashrrev_i16");
    RUN(16, "v_cndmask_b32 (vcc)"); RUN(28, "v_cmp_gt_u32 / v_cmp_gt_i32 -> vcc"); RUN(31, "v_cmp_gt_u16_sdwa / v_and_b32_sdwa");
    RUN(21, "v_add_f32"); RUN(22, "v_fma_f32");
    RUN(38, "v_and_b32 / v_or_b32"); RUN(39, "v_lshlrev_b32 / v_lshrrev_b32"); RUN(41, "v_sub_u32 / v_subrev_u32");
    RUN(36, "v_min_f32 / v_max_f32"); RUN(37, "v_min3_f32 / v_max3_f32"); RUN(35, "v_min_f16 / v_max_f16"); RUN(42, "v_min3_f16 / v_max3_f16");
    RUN(32, "v_pk_min_f16 / v_pk_max_f16"); RUN(33, "v_pk_add_f16 (+ neg modifier)"); RUN(34, "v_pk_minimum3_f16 / v_pk_maximum3_f16");
    RUN(40, "v_pk_fma_f16 / v_pk_mul_f16"); RUN(43, "v_pk_min_u16 / v_pk_max_u16");
    measure("v_pk_fma_f32", [](int blocks) { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, (float*)g_out, g_cyc, 1.5f); });
    RUN(24, "ds_read_b32 (conflict-free)"); RUN(25, "ds_read_u8 (scattered bytes)");
    return 0;
}
