// Micro-benchmark: wave64 issue rate of the VALU instructions the front-end kernels lean on (gfx950).
// Each kernel runs a long unrolled chain of ONE instruction kind on independent registers; cycles per
// wave-instruction = clock delta / count, measured with 1 and with 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITERS 512
#define UNROLL 16
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(uint32_t* out, uint32_t seed, long long* cyc)
{
    uint32_t r[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) r[i] = seed * (i + 1) + threadIdx.x;
    uint32_t a = seed ^ 0x9e3779b9u, b = seed * 7u + 1u;
    long long t0 = clock64();
    for (int it = 0; it < N_ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) r[i] = r[i] ^ r[(i + 5) & 15];                          // v_xor_b32 (register operands)
            if (OP == 1) r[i] = __popc(r[i]) + b;                                // v_bcnt_u32_b32 (accumulate form)
            if (OP == 2) r[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, r[i]) - __builtin_bit_cast(s16x2, a));  // v_pk_sub_i16
            if (OP == 3) r[i] = __builtin_amdgcn_perm(r[i], a, 0x0c050c03u);     // v_perm_b32
            if (OP == 4) r[i] = __builtin_amdgcn_udot4(r[i], a, b, false);       // v_dot4_u32_u8
            if (OP == 5) r[i] = __builtin_amdgcn_alignbyte(r[i], a, 1);          // v_alignbyte_b32
            if (OP == 6) r[i] = min(r[i], a) + 1;                                // v_min_u32 + v_add (2 instrs)
            if (OP == 7) r[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, r[i]), __builtin_bit_cast(s16x2, a)));  // v_pk_min_i16
            if (OP == 8) r[i] = r[i] * 18u + b;                                  // v_mad_u32_u24 / mul_lo
            if (OP == 9) r[i] = (r[i] >> 3) + r[(i + 3) & 15];                   // v_lshrrev + v_add (or v_lshl_add)
            if (OP == 10) r[i] = (r[i] & a) | r[(i + 7) & 15];                   // v_and_or_b32 (1 instr)
            if (OP == 11) r[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, r[i]), __builtin_bit_cast(s16x2, r[(i + 1) & 15])));  // v_pk_max_i16
            if (OP == 12) { typedef unsigned short u16x2 __attribute__((ext_vector_type(2))); r[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, r[i]) >> (unsigned short)3) + 1u; }  // v_pk_lshrrev_b16 + add
            if (OP == 13) r[i] = r[i] > a ? r[(i + 1) & 15] : b;                  // v_cmp + v_cndmask
            if (OP == 14) r[i] = __builtin_amdgcn_sad_u8(r[i], a, b);            // v_sad_u8
            if (OP == 15) r[i] = __builtin_amdgcn_ubfe(r[i], 8, 8) + b;          // v_bfe_u32 + add
        }
    }
    long long t1 = clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int instrs_per_op)
{
    uint32_t* out; long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4 * sizeof(uint32_t)); hipMalloc(&cyc, 8);
    for (int waves_per_simd : {1, 8}) {
        // 256 CUs x 4 SIMDs: blocks of 256 threads (1 wave per SIMD each); `waves_per_simd` blocks per CU
        int blocks = 256 * waves_per_simd;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double n = (double)N_ITERS * UNROLL * instrs_per_op;
        printf("%-28s waves/SIMD %d: %6.2f clk per wave-instr (in-kernel), %7.3f ms, %6.2f Tlane-op/s\n", name, waves_per_simd,
               c / n, ms, n * 64.0 * blocks * 4 / (ms * 1e-3) / 1e12);
    }
}

int main()
{
    run<0>("v_xor_b32", 1); run<1>("v_bcnt_u32_b32 (+acc)", 1); run<2>("v_pk_sub_i16", 1); run<3>("v_perm_b32", 1);
    run<4>("v_dot4_u32_u8", 1); run<5>("v_alignbyte_b32", 1); run<6>("v_min_u32 + v_add", 2); run<7>("v_pk_min_i16", 1);
    run<8>("v_mad_u32_u24 / mul+add", 1); run<9>("v_lshrrev + v_add", 2); run<10>("v_and_or_b32", 1);
    run<11>("v_pk_max_i16", 1); run<12>("v_pk_lshrrev_b16 + v_add", 2); run<13>("v_cmp + v_cndmask", 2);
    run<14>("v_sad_u8", 1); run<15>("v_bfe_u32 + v_add", 2);
    return 0;
}
