// Microbenchmark (round 5): what bounds a memory-bound chained pass (the fused Hadamard sweep's 5.4 TB/s) -- the access pattern or
// the workgroup shell around it?  No gates: a pass only moves tiles of 2^12 amplitudes, read as whole contiguous 64-KiB tiles and
// stored either contiguously or, like a chained pass, in runs of 2^c amplitudes under the layout the next pass would read whole
// (out = [t high][e high][t low][e low c]).  Shells:
//   A  what k_fused_q3 does: one tile per workgroup of 512 threads, LDS-DMA fill, barrier, 8 ds_read_b128 + 8 nontemporal stores per
//      thread, 64 KiB of LDS (two workgroups per CU)
//   B  one PERSISTENT workgroup per CU with TWO tile buffers (128 KiB): the fill of tile i+1 is issued before tile i is stored,
//      s_waitcnt vmcnt(K) leaves the newer stores in flight (gfx9 returns vector memory operations in order)
//   C  no LDS at all: 8 global loads per thread into registers, then the stores (what a pass without gates could do at best)
// hipcc --offload-arch=gfx950 -O3 -o tile_shell tile_shell.hip && ./tile_shell 30
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef double2 amp_t;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr unsigned TT = 12, TS = 1u << TT;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// c = 0: contiguous (identity); c > 0: runs of 2^c amplitudes, the tile's other bits go above the tile, the tile number's low bits below
__device__ __host__ inline uint64_t out_index(uint64_t t, unsigned e, unsigned c)
{
    if (c == 0) return (t << TT) | e;
    const unsigned hot = TT - c;
    return ((t >> hot) << (TT + hot)) | ((uint64_t)(e >> c) << TT) | ((t & ((1u << hot) - 1u)) << c) | (e & ((1u << c) - 1u));
}

__global__ void k_init(amp_t *a, uint64_t dim)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (uint64_t)gridDim.x * blockDim.x) {
        amp_t v; v.x = (double)i; v.y = -(double)i; a[i] = v;
    }
}

__global__ void k_check(const amp_t *out, uint64_t ntiles, unsigned c, unsigned long long *bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntiles * TS; i += (uint64_t)gridDim.x * blockDim.x) {
        const amp_t v = out[out_index(i >> TT, (unsigned)(i & (TS - 1)), c)];
        if (v.x != (double)i || v.y != -(double)i) atomicAdd(bad, 1ull);
    }
}

template <int BLOCK>
__device__ inline void fill_tile(const amp_t *g, amp_t *buf)
{
    const unsigned wbase = (threadIdx.x >> 6) * 64;
#pragma unroll
    for (unsigned k = 0; k < TS / BLOCK; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + k * BLOCK + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)(buf + k * BLOCK + wbase), 16, 0, 2);
}

template <int BLOCK>
__device__ inline void store_tile(const amp_t *buf, amp_t *out, uint64_t t, unsigned c, bool nt)
{
    constexpr unsigned K = TS / BLOCK;
    u4 v[K];
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) amp_t *)buf + 16u * threadIdx.x;
#pragma unroll
    for (unsigned k = 0; k < K; k++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(lds0 + 16u * k * BLOCK));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (unsigned k = 0; k < K; k++) {
        u4 *p = reinterpret_cast<u4 *>(out + out_index(t, k * BLOCK + threadIdx.x, c));
        if (nt) __builtin_nontemporal_store(v[k], p); else *p = v[k];
    }
}

// the per-gate kernels deal their tiles as 2^s interleaved streams (h_plan): workgroup b takes tile (b mod 2^s) * ntiles / 2^s + b / 2^s
__constant__ unsigned g_slog, g_pos1;       // g_pos1 = position of the stream number in the tile number + 1 (0: on top) -- fuse_stream_tile of qcx_kernels.h
__device__ inline uint64_t stream_tile(uint64_t b, uint64_t ntiles)
{
    const unsigned tl = 63u - (unsigned)__builtin_clzll(ntiles), sl = g_slog < tl ? g_slog : tl;
    const unsigned pos = g_pos1 ? (g_pos1 - 1u < tl - sl ? g_pos1 - 1u : tl - sl) : tl - sl;
    const uint64_t r = b >> sl, sb = b & ((1u << sl) - 1u);
    return (r & (((uint64_t)1 << pos) - 1u)) | (sb << pos) | ((r >> pos) << (pos + sl));
}

#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")     // (__syncthreads() also waits for the stores to land)

// shell A: one tile per workgroup (LDSB: the barrier behind the stores does not wait for them -- matters to persistent grids only)
template <int BLOCK, bool LDSB = false>
__global__ __launch_bounds__(BLOCK) void k_shell_a(const amp_t *in, amp_t *out, uint64_t ntiles, unsigned c, unsigned nt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(lds_raw);
    for (uint64_t b = blockIdx.x; b < ntiles; b += gridDim.x) {
        const uint64_t t = stream_tile(b, ntiles);
        fill_tile<BLOCK>(in + (t << TT), tile);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        store_tile<BLOCK>(tile, out, t, c, nt);
        if (LDSB) LDS_BARRIER(); else __syncthreads();
    }
}

// shell B: persistent, two buffers
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_shell_b(const amp_t *in, amp_t *out, uint64_t ntiles, unsigned c, unsigned nt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(lds_raw);
    constexpr unsigned K = TS / BLOCK;
    uint64_t t = blockIdx.x;
    if (t < ntiles) fill_tile<BLOCK>(in + (t << TT), tile);
    for (unsigned i = 0; t < ntiles; t += gridDim.x, i++) {
        // in flight, oldest first: fill(i), stores(i-1)
        if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (K == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        LDS_BARRIER();                      // tile i is in LDS; every wave is through with buffer (i+1)&1 (it read tile i-1 before it stored it)
        const uint64_t tn = t + gridDim.x;
        if (tn < ntiles) fill_tile<BLOCK>(in + (tn << TT), tile + ((i + 1u) & 1u) * TS);
        store_tile<BLOCK>(tile + (i & 1u) * TS, out, t, c, nt);
    }
}

// shell C: registers only
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_shell_c(const amp_t *in, amp_t *out, uint64_t ntiles, unsigned c, unsigned nt)
{
    constexpr unsigned K = TS / BLOCK;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        u4 v[K];
#pragma unroll
        for (unsigned k = 0; k < K; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(in + (t << TT) + k * BLOCK + threadIdx.x));
#pragma unroll
        for (unsigned k = 0; k < K; k++) {
            u4 *p = reinterpret_cast<u4 *>(out + out_index(t, k * BLOCK + threadIdx.x, c));
            if (nt) __builtin_nontemporal_store(v[k], p); else *p = v[k];
        }
    }
}

// shell D: the per-gate kernels' form -- one wave per workgroup, two amplitudes per lane, no LDS ("ntiles" workgroups of 64 threads
// per 2^12 amplitudes: 32 waves per tile)
__global__ __launch_bounds__(64) void k_shell_d(const amp_t *in, amp_t *out, uint64_t ntiles, unsigned c, unsigned nt)
{
    const uint64_t nchunks = ntiles * (TS / 128);
    for (uint64_t b = blockIdx.x; b < nchunks; b += gridDim.x) {
        const uint64_t w = stream_tile(b, nchunks);
        const uint64_t i0 = w * 128 + threadIdx.x, i1 = i0 + 64;
        const u4 v0 = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(in + i0));
        const u4 v1 = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(in + i1));
        u4 *p0 = reinterpret_cast<u4 *>(out + out_index(i0 >> TT, (unsigned)(i0 & (TS - 1)), c));
        u4 *p1 = reinterpret_cast<u4 *>(out + out_index(i1 >> TT, (unsigned)(i1 & (TS - 1)), c));
        if (nt) { __builtin_nontemporal_store(v0, p0); __builtin_nontemporal_store(v1, p1); } else { *p0 = v0; *p1 = v1; }
    }
}

// shell E: stores only (what a store-bound pass -- k_basis_front, the expanding store -- can hope for); E1: a workgroup of 512 threads
// per 64-KiB tile; E2: a wave per 32 KiB (k_basis_front's form: 4 waves per workgroup, 1 KiB per store instruction)
__global__ __launch_bounds__(512) void k_shell_e1(const amp_t *in, amp_t *out, uint64_t ntiles, unsigned c, unsigned nt)
{
    (void)in;
    for (uint64_t b = blockIdx.x; b < ntiles; b += gridDim.x) {
        const uint64_t t = stream_tile(b, ntiles);
#pragma unroll
        for (unsigned k = 0; k < 8; k++) {
            const unsigned e = k * 512 + threadIdx.x;
            const uint64_t i = (t << TT) | e;
            d2 v; v.x = (double)i; v.y = -(double)i;
            d2 *p = reinterpret_cast<d2 *>(out + out_index(t, e, c));
            if (nt) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}
__global__ __launch_bounds__(256) void k_shell_e2(const amp_t *in, amp_t *out, uint64_t ntiles, unsigned c, unsigned nt)
{
    (void)in; (void)c;
    const uint64_t nchunks = ntiles * 2;                    // 32 KiB = 2^11 amplitudes per wave
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
    for (uint64_t b = wave; b < nchunks; b += nwaves) {
        const uint64_t w = stream_tile(b, nchunks);
        for (unsigned j = 0; j < 32; j++) {
            const uint64_t i = (w << 11) + j * 64 + (threadIdx.x & 63u);
            d2 v; v.x = (double)i; v.y = -(double)i;
            d2 *p = reinterpret_cast<d2 *>(out + i);
            if (nt) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}
// shell R: reads only (measure_state's scan): a workgroup of 512 threads per 64-KiB tile, 8 loads per thread in flight
__global__ __launch_bounds__(512) void k_shell_r(const amp_t *in, amp_t *out, uint64_t ntiles, unsigned c, unsigned nt)
{
    (void)c;
    double acc = 0.0;
    for (uint64_t b = blockIdx.x; b < ntiles; b += gridDim.x) {
        const uint64_t t = stream_tile(b, ntiles);
        u4 v[8];
#pragma unroll
        for (unsigned k = 0; k < 8; k++) {
            const u4 *p = reinterpret_cast<const u4 *>(in + (t << TT) + k * 512 + threadIdx.x);
            v[k] = nt ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (unsigned k = 0; k < 8; k++) acc += (double)(v[k].x ^ v[k].w);
    }
    if (acc == 1.2345) out[0].x = acc;                      // (keeps the loads)
}

typedef void (*kern_t)(const amp_t *, amp_t *, uint64_t, unsigned, unsigned);

static void run(const char *name, kern_t kfn, unsigned block, unsigned grid, size_t lds, const amp_t *in, amp_t *out, uint64_t ntiles, unsigned c,
                unsigned nt, unsigned long long *bad_d)
{
    if (lds > 65536) CK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kfn, dim3(grid), dim3(block), lds, 0, in, out, ntiles, c, nt);
    CK(hipGetLastError());
    const int reps = 10;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kfn, dim3(grid), dim3(block), lds, 0, in, out, ntiles, c, nt);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    CK(hipMemset(bad_d, 0, 8));
    hipLaunchKernelGGL(k_check, dim3(8192), dim3(256), 0, 0, out, ntiles, c, bad_d);
    unsigned long long bad; CK(hipMemcpy(&bad, bad_d, 8, hipMemcpyDeviceToHost));
    const double bytes = 32.0 * (double)ntiles * TS;
    unsigned slog_h; CK(hipMemcpyFromSymbol(&slog_h, HIP_SYMBOL(g_slog), 4));
    printf("%-44s s=%u c=%u nt=%u grid=%-7u  %7.3f ms  %7.1f GB/s  %s\n", name, slog_h, c, nt, grid, ms, bytes / ms * 1e-6, bad ? "WRONG" : "ok");
    fflush(stdout);
    CK(hipMemset(out, 0xff, ntiles * TS * sizeof(amp_t)));          // (the next variant has to write for itself)
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
    const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 28;
    if (n < 22 || n > 31) { printf("n = 22..31\n"); return 1; }
    const uint64_t dim = (uint64_t)1 << n, ntiles = dim >> TT;
    amp_t *in, *out; unsigned long long *bad_d;
    CK(hipMalloc(&in, dim * sizeof(amp_t))); CK(hipMalloc(&out, dim * sizeof(amp_t))); CK(hipMalloc(&bad_d, 8));
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, in, dim);
    CK(hipDeviceSynchronize());
    printf("# n = %u: 2^%u tiles of 2^12 amplitudes, %.1f GB moved per pass (read + written)\n", n, n - TT, 32.0 * dim * 1e-9);
    const unsigned G = (unsigned)ntiles;
    if (argc > 2 && argv[2][0] == 'e') {          // store-only and read-only passes over streams x position
        for (unsigned sl : {0u, 1u, 2u, 3u, 4u}) {
            for (unsigned pos1 : {0u, 2u, 4u}) {
                if (sl == 0 && pos1) continue;
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_slog), &sl, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pos1), &pos1, 4));
                printf("pos1=%u ", pos1); run("E1 stores only, 512 thr per 64-KiB tile", k_shell_e1, 512, G, 0, in, out, ntiles, 0, 1, bad_d);
                printf("pos1=%u ", pos1); run("E1 stores only, plain stores", k_shell_e1, 512, G, 0, in, out, ntiles, 0, 0, bad_d);
                printf("pos1=%u ", pos1); run("E2 stores only, a wave per 32 KiB", k_shell_e2, 256, G / 2, 0, in, out, ntiles, 0, 1, bad_d);
                printf("pos1=%u ", pos1); run("R  reads only (bytes counted as 32/amp: x2)", k_shell_r, 512, G, 0, in, out, ntiles, 0, 1, bad_d);
                printf("pos1=%u ", pos1); run("R  reads only, plain loads", k_shell_r, 512, G, 0, in, out, ntiles, 0, 0, bad_d);
            }
        }
        return 0;
    }
    if (argc > 2) {             // stream sweep only
        for (unsigned sl = 0; sl <= 6; sl++) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_slog), &sl, 4));
            for (unsigned c : {0u, 3u, 4u}) {
                run("A one tile per workgroup, 512 thr, 64 KiB", k_shell_a<512>, 512, G, 65536, in, out, ntiles, c, 1, bad_d);
                run("D one wave per workgroup, 2 amplitudes/lane", k_shell_d, 64, G * 32, 0, in, out, ntiles, c, 1, bad_d);
            }
        }
        hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, out, dim);
        for (unsigned sl = 0; sl <= 6; sl++) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_slog), &sl, 4));
            run("A one tile per workgroup, IN PLACE", k_shell_a<512>, 512, G, 65536, out, out, ntiles, 0, 1, bad_d);
            hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, out, dim);
            run("D one wave per workgroup, IN PLACE", k_shell_d, 64, G * 32, 0, out, out, ntiles, 0, 1, bad_d);
            hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, out, dim);
        }
        return 0;
    }
    for (unsigned c : {0u, 3u, 4u}) {
        for (unsigned nt : {1u, 0u}) {
            if (nt == 0 && c == 4) continue;
            run("A one tile per workgroup, 512 thr, 64 KiB", k_shell_a<512>, 512, G, 65536, in, out, ntiles, c, nt, bad_d);
            run("A persistent, 512 workgroups", k_shell_a<512>, 512, 512, 65536, in, out, ntiles, c, nt, bad_d);
            run("A persistent, 512 wg, stores not awaited", k_shell_a<512, true>, 512, 512, 65536, in, out, ntiles, c, nt, bad_d);
            run("B two buffers, 256 x 512 thr, 128 KiB", k_shell_b<512>, 512, 256, 131072, in, out, ntiles, c, nt, bad_d);
            run("B two buffers, 256 x 1024 thr, 128 KiB", k_shell_b<1024>, 1024, 256, 131072, in, out, ntiles, c, nt, bad_d);
            run("C registers only, 512 thr per tile", k_shell_c<512>, 512, G, 0, in, out, ntiles, c, nt, bad_d);
            run("C registers only, 256 thr per tile", k_shell_c<256>, 256, G, 0, in, out, ntiles, c, nt, bad_d);
            run("D one wave per workgroup, 2 amplitudes/lane", k_shell_d, 64, G * 32, 0, in, out, ntiles, c, nt, bad_d);
        }
    }
    // the same shells IN PLACE (out = in, contiguous): what the per-gate kernels' 6.4-6.8 TB/s have that a ping-pong pass has not
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, out, dim);
    CK(hipDeviceSynchronize());
    for (unsigned nt : {1u, 0u}) {
        run("A one tile per workgroup, IN PLACE", k_shell_a<512>, 512, G, 65536, out, out, ntiles, 0, nt, bad_d);
        hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, out, dim);
        run("C registers only, 512 thr, IN PLACE", k_shell_c<512>, 512, G, 0, out, out, ntiles, 0, nt, bad_d);
        hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, out, dim);
        run("C registers only, 256 thr, IN PLACE", k_shell_c<256>, 256, G, 0, out, out, ntiles, 0, nt, bad_d);
        hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, out, dim);
        run("D one wave per workgroup, IN PLACE", k_shell_d, 64, G * 32, 0, out, out, ntiles, 0, nt, bad_d);
        hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, out, dim);
    }
    return 0;
}
