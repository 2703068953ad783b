// tools/microbench_lds.hip -- LDS fp64 gather / atomic-add / plain RMW rates with register-resident random indices
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: ds_add_f64 (no return)  1: gather ds_read_b64  2: plain read+add+write  3: ds_add_f32  4: u64 int atomic add
__global__ __launch_bounds__(256) void k(double *out, int iters, int span, int pat) {
    extern __shared__ double s[];
    for (int i = threadIdx.x; i < span; i += blockDim.x) s[i] = 0;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    int idx[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        h = h * 1664525u + 1013904223u;
        // pat 0: random; 1: lane-consecutive (conflict-free); 2: groups of 6 consecutive lanes share one address;
        // 3: lane-consecutive with stride 15 (CSR rows of 15 doubles, same column)
        idx[q] = pat == 0 ? (h >> 8) % span : pat == 1 ? (threadIdx.x + q * 256) % span
               : pat == 2 ? (threadIdx.x / 6 + q * 64) % span : pat == 3 ? (threadIdx.x * 15 + q) % span
               // 4/5/6: lanes l and l+16 / l+8 / l+32 hit the same BANK at different addresses (conflict window probe)
               : pat == 4 ? ((threadIdx.x & 15) + 32 * ((threadIdx.x >> 4) + 16 * q)) % span
               : pat == 5 ? ((threadIdx.x & 7) + 32 * ((threadIdx.x >> 3) + 32 * q)) % span
               : pat == 6 ? ((threadIdx.x & 31) + 32 * ((threadIdx.x >> 5) + 8 * q)) % span
               // 200/201/202: the whole-entity P1 wrapper's windows: rows of 15 doubles at row positions {0..6, 8..14, 16, 17} (two lines of
               // seven cells of an 8-node-wide box and two cells of a third line: ONE pair of lanes per bank collides twice) / the same
               // with lanes 14 and 15 of the window idle / 16 consecutive row positions
               : pat == 200 || pat == 201 ? (15 * ((threadIdx.x & 15) + ((threadIdx.x & 15) >= 7) + ((threadIdx.x & 15) >= 14)) + 1024 * (threadIdx.x >> 4) % 3072 + q) % span
               : pat == 202 ? (15 * (threadIdx.x & 15) + 1024 * (threadIdx.x >> 4) % 3072 + q) % span
               // 100+s: lane l of every 16-lane window accesses (l & 15) * s (+ window and q offsets that keep the bank): bank function probe
               : (((threadIdx.x & 15) * (pat - 100)) + 1024 * (threadIdx.x >> 4) % 3072 + 32 * q) % span;
    }
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 0) { if (pat != 201 || (threadIdx.x & 15) < 14) atomicAdd(&s[idx[q]], 1.0); }
            else if (MODE == 1) acc += s[idx[q]];
            else if (MODE == 2) s[idx[q]] += 1.0;
            else if (MODE == 3) atomicAdd(((float *)s) + idx[q], 1.0f);
            else if (MODE == 4) atomicAdd(((unsigned long long *)s) + idx[q], 1ull);
            else if (MODE == 5) { double2 v = *(const double2 *)&s[idx[q] & ~1]; acc += v.x + v.y; }
            else if (MODE == 6) { const double *r = &s[(idx[q] % (span / 6)) * 6]; double2 a = *(const double2 *)r, b = *(const double2 *)(r + 2), c = *(const double2 *)(r + 4); acc += a.x + a.y + b.x + b.y + c.x; (void)c.y; }
            else { const double *r = &s[(idx[q] % (span / 5)) * 5]; acc += r[0] + r[1] + r[2] + r[3] + r[4]; }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) idx[q] = (idx[q] + 17) < span ? idx[q] + 17 : idx[q] + 17 - span;
    }
    __syncthreads();
    if (threadIdx.x == 0 && (s[0] == -1.0 || acc == -1.0)) out[0] = s[1] + acc;
}

template <int MODE> void run(const char *name, int span, int pat = 0) {
    double *out; CK(hipMalloc(&out, 64));
    int iters = 256, grid = 256 * 8;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), span * 8, 0, out, iters, span, pat); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), span * 8, 0, out, iters, span, pat);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)grid * 256 * iters * 16;
    printf("%-28s pat %d span %5d : %.3f ms  %.0f Gop/s chip  (%.2f lanes/clk/CU @2.1GHz)\n", name, pat, span, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.1);
    CK(hipFree(out));
}

int main() {
    for (int span : {512, 2048, 5472}) {
        run<0>("ds_add_f64 (atomic)", span);
        run<1>("ds_read_b64 gather", span);
        run<2>("plain read+add+write f64", span);
        run<3>("ds_add_f32 (atomic)", span);
        run<4>("ds_add_u64 (atomic)", span);
        run<5>("ds_read_b128 gather", span);
        run<6>("record 48B: 3x b128", span);
        run<7>("record 40B: 5x b64", span);
    }
    for (int pat : {1, 2, 3, 4, 5, 6}) {
        run<0>("ds_add_f64 (atomic)", 4096, pat);
        run<1>("ds_read_b64 gather", 4096, pat);
        run<2>("plain read+add+write f64", 4096, pat);
    }
    for (int pat : {200, 201, 202}) run<0>("ds_add_f64 P1 window", 4096, pat);
    for (int st : {1, 2, 3, 4, 8, 16, 32, 64}) {
        run<0>("ds_add_f64 stride", 4096, 100 + st);
        run<1>("ds_read_b64 stride", 4096, 100 + st);
    }
    return 0;
}
