// tools/microbench.hip -- hardware rates that bound the assembly kernels on MI355X:
// HBM stream bandwidth, fp64 global atomic-add throughput (streaming / scattered / conflicting),
// LDS fp64 atomic-add throughput, fp64 FMA rate and fp64 MFMA (v_mfma_f64_16x16x4_f64) rate.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_read(const double2 *__restrict__ a, double *__restrict__ out, size_t n) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345) out[0] = s;
}
// mode 0: streaming (lane i -> address i); 1: scattered via index array; 
__global__ void k_atomic(double *__restrict__ dst, const int *__restrict__ idx, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        atomicAdd(&dst[idx ? idx[i] : i], 1.0);
}
__global__ void k_store(double *__restrict__ dst, const int *__restrict__ idx, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[idx ? idx[i] : i] = 1.0;
}
__global__ void k_gather(const double *__restrict__ src, const int *__restrict__ idx, double *__restrict__ out, size_t n) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += src[idx[i]];
    if (s == 1.2345) out[0] = s;
}
__global__ void k_lds_atomic(double *out, int iters, int span) {
    extern __shared__ double s[];
    for (int i = threadIdx.x; i < span; i += blockDim.x) s[i] = 0;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        atomicAdd(&s[(h >> 8) % span], 1.0);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s[0] == -1.0) out[0] = s[1];
}
__global__ void k_lds_read(double *out, int iters, int span) {
    extern __shared__ double s[];
    for (int i = threadIdx.x; i < span; i += blockDim.x) s[i] = i;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        acc += s[(h >> 8) % span];
    }
    if (acc == -1.0) out[0] = acc;
}
__global__ void k_fma(double *out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = a0 * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c;
        a4 = a4 * b + c; a5 = a5 * b + c; a6 = a6 * b + c; a7 = a7 * b + c;
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == -1.0) out[0] = a0;
}
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double *out, int iters) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3;
    if (s[0] + s[1] + s[2] + s[3] == -1.0) out[0] = s[0];
}

template <class F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t N = 1ull << 27;  // 128M doubles = 1 GiB
    double *a, *b, *out; int *idx;
    CK(hipMalloc(&a, N * 8)); CK(hipMalloc(&b, N * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&idx, N * 4));
    CK(hipMemset(a, 0, N * 8)); CK(hipMemset(b, 0, N * 8));
    int grid = 256 * 8;
    float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, N / 2); });
    printf("copy 1GiB->1GiB          : %.3f ms  %.0f GB/s (r+w)\n", ms, 2.0 * N * 8 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const double2 *)a, out, N / 2); });
    printf("read 1GiB                : %.3f ms  %.0f GB/s\n", ms, 1.0 * N * 8 / ms / 1e6);
    ms = timeit([&] { CK(hipMemsetAsync(b, 0, N * 8, 0)); });
    printf("memset 1GiB              : %.3f ms  %.0f GB/s\n", ms, 1.0 * N * 8 / ms / 1e6);
    // atomics
    const size_t NA = 1ull << 26;  // 64M atomics
    ms = timeit([&] { hipLaunchKernelGGL(k_atomic, dim3(grid * 4), dim3(256), 0, 0, b, (const int *)nullptr, NA); });
    printf("atomic f64 streaming 64M : %.3f ms  %.1f Gatom/s\n", ms, NA / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_store, dim3(grid * 4), dim3(256), 0, 0, b, (const int *)nullptr, NA); });
    printf("store  f64 streaming 64M : %.3f ms  %.1f Gstore/s\n", ms, NA / ms / 1e6);
    std::vector<int> h(NA);
    struct Case { const char *name; size_t span; int dup; };
    Case cases[] = {{"random in 80MB (10M dbl)", 10000000, 1}, {"random in 8MB (L2-ish)", 1000000, 1},
                    {"random in 256KB", 32768, 1}, {"near-sorted dup4 80MB", 10000000, 4}};
    for (auto &c : cases) {
        unsigned s = 12345;
        if (c.dup == 1) for (size_t i = 0; i < NA; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 4) % c.span); }
        else for (size_t i = 0; i < NA; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)(((i / c.dup) + (s >> 28)) % c.span); }
        CK(hipMemcpy(idx, h.data(), NA * 4, hipMemcpyHostToDevice));
        ms = timeit([&] { hipLaunchKernelGGL(k_atomic, dim3(grid * 4), dim3(256), 0, 0, b, (const int *)idx, NA); });
        printf("atomic f64 %-26s: %.3f ms  %.1f Gatom/s\n", c.name, ms, NA / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_gather, dim3(grid * 4), dim3(256), 0, 0, (const double *)b, (const int *)idx, out, NA); });
        printf("gather f64 %-26s: %.3f ms  %.1f Gload/s\n", c.name, ms, NA / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_store, dim3(grid * 4), dim3(256), 0, 0, b, (const int *)idx, NA); });
        printf("store  f64 %-26s: %.3f ms  %.1f Gstore/s\n", c.name, ms, NA / ms / 1e6);
    }
    // LDS
    for (int span : {256, 1024, 4096}) {
        int iters = 4096;
        ms = timeit([&] { hipLaunchKernelGGL(k_lds_atomic, dim3(256 * 8), dim3(256), span * 8, 0, out, iters, span); });
        printf("LDS atomic f64 span %-5d : %.3f ms  %.1f Gatom/s chip\n", span, ms, 256.0 * 8 * 256 * iters / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_lds_read, dim3(256 * 8), dim3(256), span * 8, 0, out, iters, span); });
        printf("LDS read   f64 span %-5d : %.3f ms  %.1f Gread/s chip\n", span, ms, 256.0 * 8 * 256 * iters / ms / 1e6);
    }
    {
        int iters = 1 << 16;
        ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(256 * 8), dim3(256), 0, 0, out, iters); });
        printf("fp64 FMA                 : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 8 * iters * 256.0 * 8 * 256 / ms / 1e9);
        iters = 1 << 14;
        ms = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(256 * 4), dim3(256), 0, 0, out, iters); });
        printf("fp64 MFMA 16x16x4        : %.3f ms  %.1f TFLOP/s\n", ms, 4.0 * iters * 2048.0 * (256.0 * 4 * 4) / ms / 1e9);
    }
    return 0;
}
