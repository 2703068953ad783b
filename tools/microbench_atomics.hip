// tools/microbench_atomics.hip -- rate of global fp64 atomics (and plain stores) for the address patterns of the matrix scatters:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o /tmp/mba tools/microbench_atomics.hip && /tmp/mba
// Every lane performs ONE operation per pass on an array of 2^27 doubles (1 GiB); the patterns differ in which addresses the
// 64 lanes of a wavefront instruction touch:
//   stream      lane l -> element base + l                          (the staged flush: 8 cache lines of 64 B per instruction)
//   runs R      groups of R consecutive lanes on R consecutive elements, groups at random places
//               (tensor-product scatter: 5 consecutive vertical nodes of a Q4 row = runs of 5)
//   rows 4xR    4 rows x 16 lanes, inside a row runs of R, rows 4 KB apart (the MFMA accumulator layout: 4 rows x 16 columns)
//   stride S    lane l -> base + S*l                                 ((Q_k)^3 blocks: entries 3 doubles apart)
//   random      every lane on its own random element
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool ATOMIC> __global__ void k(double *__restrict__ a, const unsigned *__restrict__ idx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned p = idx[i];
        if (ATOMIC) atomicAdd(&a[p], 1.0); else a[p] = 1.0;
    }
}

int main() {
    const long N = 1L << 27, OPS = 1L << 26;
    double *a; unsigned *idx;
    CK(hipMalloc(&a, N * 8)); CK(hipMalloc(&idx, OPS * 4)); CK(hipMemset(a, 0, N * 8));
    std::vector<unsigned> h(OPS);
    auto rnd = [s = 88172645463325252ull]() mutable { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name) {
        CK(hipMemcpy(idx, h.data(), OPS * 4, hipMemcpyHostToDevice));
        for (int atomic = 1; atomic >= 0; --atomic) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (atomic) hipLaunchKernelGGL(k<true>, dim3(256 * 16), dim3(256), 0, 0, a, idx, OPS);
                else hipLaunchKernelGGL(k<false>, dim3(256 * 16), dim3(256), 0, 0, a, idx, OPS);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
            }
            printf("%-14s %-7s %8.3f ms  %7.1f G ops/s\n", name, atomic ? "atomic" : "store", best, OPS / best / 1e6);
        }
    };
    for (long i = 0; i < OPS; ++i) h[i] = (unsigned)i;
    run("stream");
    for (int R : {16, 8, 5, 3, 2}) {
        for (long g = 0; g < OPS; g += R) { const unsigned b = (unsigned)(rnd() % (N - 64)); for (int r = 0; r < R && g + r < OPS; ++r) h[g + r] = b + r; }
        char nm[32]; snprintf(nm, 32, "runs %d", R); run(nm);
    }
    for (int R : {16, 5}) {
        for (long w = 0; w < OPS; w += 64) {
            const unsigned b = (unsigned)(rnd() % (N - 4 * 512 - 64));
            unsigned off[16]; unsigned cur = 0;
            for (int c = 0; c < 16; ++c) { if (c % R == 0 && c) cur += 8 + (unsigned)(rnd() % 24); off[c] = cur++; }
            for (int l = 0; l < 64; ++l) h[w + l] = b + (l >> 4) * 512 + off[l & 15];
        }
        char nm[32]; snprintf(nm, 32, "rows 4x%d", R); run(nm);
    }
    for (int S : {3, 9}) {
        for (long w = 0; w < OPS; w += 64) { const unsigned b = (unsigned)(rnd() % (N - 64 * S)); for (int l = 0; l < 64; ++l) h[w + l] = b + S * l; }
        char nm[32]; snprintf(nm, 32, "stride %d", S); run(nm);
    }
    for (long i = 0; i < OPS; ++i) h[i] = (unsigned)(rnd() % N);
    run("random");
    return 0;
}
