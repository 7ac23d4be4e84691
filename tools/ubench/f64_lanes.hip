// Does a wave64 whose EXEC mask covers only 16 (or 32) lanes issue f64 / f32 vector instructions faster than a full wave?
// One wavefront per SIMD (workgroups of 256 threads, one per CU), 8 independent FMA chains per lane, lanes >= ACTIVE idle.
// Build: hipcc -O3 --offload-arch=gfx950 f64_lanes.hip -o /tmp/f64_lanes   (GPU box; the binary is not committed)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <typename T>
__global__ __launch_bounds__(256) void k(T* out, int active, T a, T b)
{
    const int lane = threadIdx.x & 63;
    T x[8];
    for (int i = 0; i < 8; i++) x[i] = (T)(threadIdx.x + i);
    if (lane < active) {
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (sizeof(T) == 8) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
            }
        }
    }
    T s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename T> void run(const char* name)
{
    T* out; (void)hipMalloc(&out, 256 * 256 * sizeof(T));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int active : {64, 64, 32, 16, 8, 64}) {                 // the first line of each type warms the clocks up
        hipLaunchKernelGGL(k<T>, dim3(256), dim3(256), 0, 0, out, active, (T)1.0000001, (T)0.5);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<T>, dim3(256), dim3(256), 0, 0, out, active, (T)1.0000001, (T)0.5);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double clk = ms * 1e-3 * 2.4e9 / (ITERS * 8.0);
        printf("%s  active lanes %2d: %.3f ms  -> %.2f clk per wave-instruction (one wave per SIMD)\n", name, active, ms, clk);
    }
}
int main() { run<double>("v_fma_f64"); run<float>("v_fma_f32"); return 0; }
