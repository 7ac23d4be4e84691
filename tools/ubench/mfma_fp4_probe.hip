// Probe: operand layout and exactness of v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (e2m1) operands on gfx950.
// Hypothesis under test: lane l of A holds row l & 15, k = 32 (l >> 4) + n for nibble n (low nibble of byte 0 first) of its
// first four dwords; B likewise with column l & 15; C/D: col = lane & 15, row = 4 (lane >> 4) + reg.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const uint32_t* A, const uint32_t* B, float* C, int scale_a, int scale_b)
{
    const int l = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4; q++) { a[q] = (int)A[l * 4 + q]; b[q] = (int)B[l * 4 + q]; }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 4; r++) C[l * 4 + r] = c[r];
}

int main()
{
    const int codes[3] = {0x2, 0xA, 0x0};                      // +1, -1, 0 in e2m1
    const float vals[3] = {1.f, -1.f, 0.f};
    std::vector<float> Am(16 * 128), Bm(128 * 16);
    std::vector<uint32_t> Ar(64 * 4, 0), Br(64 * 4, 0);
    srand(7);
    for (int l = 0; l < 64; l++)
        for (int n = 0; n < 32; n++) {
            const int ca = rand() % 3, cb = rand() % 3;
            const int k = 32 * (l >> 4) + n;
            Am[(l & 15) * 128 + k] = vals[ca]; Bm[k * 16 + (l & 15)] = vals[cb];
            Ar[l * 4 + n / 8] |= (uint32_t)codes[ca] << (4 * (n % 8));
            Br[l * 4 + n / 8] |= (uint32_t)codes[cb] << (4 * (n % 8));
        }
    uint32_t *dA, *dB; float* dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024);
    hipMemcpy(dA, Ar.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, Br.data(), 1024, hipMemcpyHostToDevice);
    for (int sa = 127; sa <= 140; sa += 13) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, 127);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                float e = 0;
                for (int kk = 0; kk < 128; kk++) e += Am[row * 128 + kk] * Bm[kk * 16 + col];
                e *= (float)(1 << (sa - 127));
                if (C[l * 4 + r] != e) { if (bad < 5) printf("  mismatch lane %d reg %d: got %g want %g\n", l, r, C[l * 4 + r], e); bad++; }
            }
        printf("scale_a E8M0 %d (x%d): %d of 256 outputs differ from the hypothesis\n", sa, 1 << (sa - 127), bad);
    }
    return 0;
}
