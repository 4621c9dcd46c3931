// Microbenchmark (round 5): what one SIMD sustains on the rotation sequence of the exact walk -- v_mul_f64 / v_add_f64 with the
// factors in SGPR pairs (as the walk has them) or in VGPRs -- at 1, 2, 4, 8 waves per SIMD.  Prints cycles per FP64 instruction per
// SIMD (s_memtime over the loop).   hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip && ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ROT2S                                  \
    "v_mul_f64 %[t0], %[c], %[x0]\n\t"        \
    "v_mul_f64 %[t1], %[s], %[y0]\n\t"        \
    "v_mul_f64 %[t2], %[c], %[y0]\n\t"        \
    "v_mul_f64 %[t3], %[s], %[x0]\n\t"        \
    "v_add_f64 %[x0], %[t0], -%[t1]\n\t"      \
    "v_mul_f64 %[t0], %[c], %[x1]\n\t"        \
    "v_mul_f64 %[t1], %[s], %[y1]\n\t"        \
    "v_add_f64 %[y0], %[t2], %[t3]\n\t"       \
    "v_mul_f64 %[t2], %[c], %[y1]\n\t"        \
    "v_mul_f64 %[t3], %[s], %[x1]\n\t"        \
    "v_add_f64 %[x1], %[t0], -%[t1]\n\t"      \
    "v_add_f64 %[y1], %[t2], %[t3]\n\t"

template <int MODE>      // 0: c, s in SGPRs; 1: c, s in VGPRs; 2: only v_mul_f64 (independent); 3: only v_add_f64; 4: v_fma_f64
__global__ void k(double *out, long long *cyc, double c, double s, int iters)
{
    double x0 = threadIdx.x * 1e-3, y0 = 0.5, x1 = 0.25, y1 = 0.125, t0, t1, t2, t3;
    double cv = c + threadIdx.x * 0.0, sv = s + threadIdx.x * 0.0;
    long long t_begin = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile(ROT2S ROT2S ROT2S ROT2S : [x0] "+v"(x0), [y0] "+v"(y0), [x1] "+v"(x1), [y1] "+v"(y1), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) : [c] "s"(c), [s] "s"(s));
        } else if (MODE == 1) {
            asm volatile(ROT2S ROT2S ROT2S ROT2S : [x0] "+v"(x0), [y0] "+v"(y0), [x1] "+v"(x1), [y1] "+v"(y1), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) : [c] "v"(cv), [s] "v"(sv));
        } else if (MODE == 2) {
            asm volatile("v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         "v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         "v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) : [x0] "v"(x0), [y0] "v"(y0), [x1] "v"(x1), [y1] "v"(y1), [c] "s"(c));
            asm volatile("v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         "v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         "v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) : [x0] "v"(x0), [y0] "v"(y0), [x1] "v"(x1), [y1] "v"(y1), [c] "s"(c));
            asm volatile("v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         "v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         "v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) : [x0] "v"(x0), [y0] "v"(y0), [x1] "v"(x1), [y1] "v"(y1), [c] "s"(c));
            asm volatile("v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         "v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         "v_mul_f64 %[t0], %[c], %[x0]\n\t v_mul_f64 %[t1], %[c], %[y0]\n\t v_mul_f64 %[t2], %[c], %[x1]\n\t v_mul_f64 %[t3], %[c], %[y1]\n\t"
                         : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) : [x0] "v"(x0), [y0] "v"(y0), [x1] "v"(x1), [y1] "v"(y1), [c] "s"(c));
            x0 += t0; y0 += t1; x1 += t2; y1 += t3;
        } else if (MODE == 3) {
#define ADD12 "v_add_f64 %[t0], %[t0], %[x0]\n\t v_add_f64 %[t1], %[t1], %[y0]\n\t v_add_f64 %[t2], %[t2], %[x1]\n\t v_add_f64 %[t3], %[t3], %[y1]\n\t" \
              "v_add_f64 %[t0], %[t0], %[x0]\n\t v_add_f64 %[t1], %[t1], %[y0]\n\t v_add_f64 %[t2], %[t2], %[x1]\n\t v_add_f64 %[t3], %[t3], %[y1]\n\t" \
              "v_add_f64 %[t0], %[t0], %[x0]\n\t v_add_f64 %[t1], %[t1], %[y0]\n\t v_add_f64 %[t2], %[t2], %[x1]\n\t v_add_f64 %[t3], %[t3], %[y1]\n\t"
            t0 = t1 = t2 = t3 = 0.0;
            asm volatile(ADD12 ADD12 ADD12 ADD12 : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3) : [x0] "v"(x0), [y0] "v"(y0), [x1] "v"(x1), [y1] "v"(y1));
            x0 += t0; y0 += t1; x1 += t2; y1 += t3;
        } else {
#define FMA12 "v_fma_f64 %[t0], %[c], %[x0], %[t0]\n\t v_fma_f64 %[t1], %[c], %[y0], %[t1]\n\t v_fma_f64 %[t2], %[c], %[x1], %[t2]\n\t v_fma_f64 %[t3], %[c], %[y1], %[t3]\n\t" \
              "v_fma_f64 %[t0], %[c], %[x0], %[t0]\n\t v_fma_f64 %[t1], %[c], %[y0], %[t1]\n\t v_fma_f64 %[t2], %[c], %[x1], %[t2]\n\t v_fma_f64 %[t3], %[c], %[y1], %[t3]\n\t" \
              "v_fma_f64 %[t0], %[c], %[x0], %[t0]\n\t v_fma_f64 %[t1], %[c], %[y0], %[t1]\n\t v_fma_f64 %[t2], %[c], %[x1], %[t2]\n\t v_fma_f64 %[t3], %[c], %[y1], %[t3]\n\t"
            t0 = t1 = t2 = t3 = 0.0;
            asm volatile(FMA12 FMA12 FMA12 FMA12 : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3) : [x0] "v"(x0), [y0] "v"(y0), [x1] "v"(x1), [y1] "v"(y1), [c] "s"(c));
            x0 += t0; y0 += t1; x1 += t2; y1 += t3;
        }
    }
    long long t_end = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + y0 + x1 + y1;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t_end - t_begin;
}

template <int MODE>
static void run(const char *name, int waves_per_simd)
{
    const int iters = 2000, block = 64 * 4 * waves_per_simd, grid = 256;      // one workgroup per CU, waves_per_simd waves on each SIMD
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * grid * block); hipMalloc(&cyc, sizeof(long long) * grid * block / 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, out, cyc, 0.8, 0.6, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, out, cyc, 0.8, 0.6, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * block / 64);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double avg = 0; for (long long v : h) avg += (double)v; avg /= h.size();
    const double per_wave = 48.0 * iters;                      // FP64 instructions per wave
    // s_memtime ticks at 100 MHz? (constant clock) -- report both the tick-based and the wall-based figure
    const double insts_per_simd = per_wave * waves_per_simd;
    printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per FP64 instruction per SIMD (= %.2f cycles at 2.4 GHz), memtime ticks per instr per SIMD %.3f\n",
           name, waves_per_simd, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4, avg / insts_per_simd);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4, 8}) run<0>("rotations, c/s in SGPRs", w);
    for (int w : {1, 2, 4, 8}) run<1>("rotations, c/s in VGPRs", w);
    for (int w : {1, 2, 4}) run<2>("v_mul_f64 only", w);
    for (int w : {1, 2, 4}) run<3>("v_add_f64 only", w);
    for (int w : {1, 2, 4}) run<4>("v_fma_f64 only", w);
    return 0;
}
