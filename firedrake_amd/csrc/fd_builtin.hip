// fd_builtin.hip -- wrapper kernels compiled ahead of time into libfdhip.so and reachable
// through fd_kernel_builtin().  (The JIT route -- codegen.py + hipcc --genco -- produces
// the same kind of kernel for arbitrary local kernels.)
#include "fd_common.h"
#include "fd_wrapper.h"

namespace fd { int register_builtin(const char *name, const void *fn); }
#define FD_REGISTER(sym) static int _fd_reg_##sym = fd::register_builtin(#sym, (const void *)sym)

// Smoke kernel: y[i] += a * x[i] over [start, end) -- used by the CPU-side ABI test (symbol
// presence) and the first GPU sanity check of launch marshalling.
extern "C" __global__ void wrap_fd_axpy(int start, int end, double *y, const double *x, const double *a) {
    for (int i = start + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x)
        y[i] += a[0] * x[i];
}
FD_REGISTER(wrap_fd_axpy);

// ---- PMC calibration kernels (bench.py --traffic, tools/pmc_calibrate.sh): known byte counts in the access widths the
// wrapper kernels use, so the FETCH_SIZE / WRITE_SIZE readings of rocprofv3 can be converted to bytes for THOSE patterns
// (MI355X_MICROARCH.md, HBM section: only 16 B/lane streaming reads are calibrated there).  Element i of [start, end) is
// read (or written) once, consecutive lanes touch consecutive elements; `sink` receives one value per workgroup.
template <class T> __device__ __forceinline__ double calib_val(const T &v) { return (double)v; }
template <> __device__ __forceinline__ double calib_val<uint4>(const uint4 &v) { return (double)(v.x ^ v.y ^ v.z ^ v.w); }
template <> __device__ __forceinline__ double calib_val<uint2>(const uint2 &v) { return (double)(v.x ^ v.y); }

template <class T> __device__ __forceinline__ void calib_read(int start, int end, const T *__restrict__ src, double *__restrict__ sink) {
    double acc = 0.0;
    for (long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        acc += calib_val<T>(src[i]);
    acc = fdw::wave_reduce<double, fdw::OpAdd<double>>(acc);
    if ((threadIdx.x & 63) == 0 && acc == 1.2345e300) sink[blockIdx.x] = acc;      // keeps the loads alive, (almost) never stores
}
extern "C" __global__ void wrap_fd_calib_read2(int start, int end, const unsigned short *src, double *sink) { calib_read<unsigned short>(start, end, src, sink); }
extern "C" __global__ void wrap_fd_calib_read4(int start, int end, const unsigned int *src, double *sink) { calib_read<unsigned int>(start, end, src, sink); }
extern "C" __global__ void wrap_fd_calib_read8(int start, int end, const uint2 *src, double *sink) { calib_read<uint2>(start, end, src, sink); }
extern "C" __global__ void wrap_fd_calib_read16(int start, int end, const uint4 *src, double *sink) { calib_read<uint4>(start, end, src, sink); }
// random 8-byte gathers through an index array (the pattern of the node-row staging): idx read coalesced, src gathered
extern "C" __global__ void wrap_fd_calib_gather8(int start, int end, const double *src, const int *idx, double *sink) {
    double acc = 0.0;
    for (long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        acc += src[idx[i]];
    acc = fdw::wave_reduce<double, fdw::OpAdd<double>>(acc);
    if ((threadIdx.x & 63) == 0 && acc == 1.2345e300) sink[blockIdx.x] = acc;
}
extern "C" __global__ void wrap_fd_calib_write8(int start, int end, double *dst) {
    for (long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (double)i;
}
extern "C" __global__ void wrap_fd_calib_atomic8(int start, int end, double *dst) {       // streaming fp64 atomics (the staged flush)
    for (long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        atomicAdd(&dst[i], 1.0);
}
FD_REGISTER(wrap_fd_calib_read2);
FD_REGISTER(wrap_fd_calib_read4);
FD_REGISTER(wrap_fd_calib_read8);
FD_REGISTER(wrap_fd_calib_read16);
FD_REGISTER(wrap_fd_calib_gather8);
FD_REGISTER(wrap_fd_calib_write8);
FD_REGISTER(wrap_fd_calib_atomic8);

// =====================================================================================
// Config C3: Helmholtz stiffness+mass on Q4 hexahedra (extruded), element matrix by fp64 MFMA.
//
//   a(u, v) = int grad(u).grad(v) + u v dx,   Q4 = CG4 (x) CG4 (x) CG4, 125 DoFs/cell,
//   dx(degree=8): 5 Gauss points per axis = 125 quadrature points.
//
// TSFC emits a sum-factorised scalar kernel for this form (tsfc/spectral.py:157-191); wrapping that
// per-lane would need a 125 KB private element tensor.  Here the element matrix is computed as the dense
// contraction it is:      A_e = sum_q  Phi_q^T  W_q  Phi_q
// with Phi_q (4 x 125): the three reference-gradient components and the value of every basis function at
// quadrature point q (cell independent, product of 1-D tables), and W_q (4 x 4) = [G_q 0; 0 m_q],
// G_q = w_q |J| J^-1 J^-T, m_q = w_q |J| (cell dependent).  K = 4 per quadrature point is exactly the K of
// v_mfma_f64_16x16x4_f64: one MFMA updates a 16x16 tile of A_e with one quadrature point.
//
// Two workgroups (4 wavefronts each) per cell; wavefront w of half h owns the 16-row panel i in
// [16(4h+w), 16(4h+w)+16) of the (padded 128 x 128) element matrix: 1 x 8 tiles = 64 accumulator registers, so
// three wavefronts fit per SIMD and one wavefront's operand preparation / scatter overlaps another's MFMAs.
// Per quadrature point a lane builds 1 A-operand ((W Phi)^T for its row tile) and 8 B-operands (Phi for the eight
// column tiles) from LDS-resident 1-D tables and the per-cell W (7 doubles/point, computed by the workgroup),
// then issues 8 MFMAs.  Finally each lane scatters its 32 accumulator values with fp64 atomics through the
// element->nonzero table (MatSetValuesLocal ADD_VALUES of the 125 x 125 block, builder.py:573-625).
// Roofline: fp64 MFMA; 2*128*128*4*125 = 16.4 MFLOP per cell.
// =====================================================================================
typedef double fd_d4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int Q4_ND = 125, Q4_NQ1 = 5;

__device__ __forceinline__ void inv3(const double J[3][3], double K[3][3], double &det) {
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double id = 1.0 / det;
    K[0][0] = c00 * id; K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    K[1][0] = c01 * id; K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    K[2][0] = c02 * id; K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
}
}  // namespace

// tables: L[5][5] (value of 1-D basis i at Gauss point q: L[q*5+i]), DL[5][5], QP[5], QW[5]  (60 doubles)
// args after (start, end): layers, values, coords, map_q4 (unused: positions come from elemtab), map_q1, elemtab, tables
extern "C" __global__ __launch_bounds__(256, 3)
void wrap_helmholtz_q4_hex_jacobian(int start, int end, const int *__restrict__ layers, double *__restrict__ vals,
                                    const double *__restrict__ coords, const int *__restrict__ map_q4,
                                    const int *__restrict__ map_q1, const int *__restrict__ elemtab,
                                    const double *__restrict__ tables) {
    __shared__ double sL[25], sDL[25], sQP[5], sQW[5];
    __shared__ double sX[24];
    __shared__ double sW[125][8];          // G00 G01 G02 G11 G12 G22 m pad
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = layers[1] - 1 - layers[0];
    const int cellid = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int col = start + cellid / nl;
    const int layer = layers[0] + cellid % nl;
    if (col >= end) return;
    if (tid < 25) { sL[tid] = tables[tid]; sDL[tid] = tables[25 + tid]; }
    if (tid < 5) { sQP[tid] = tables[50 + tid]; sQW[tid] = tables[55 + tid]; }
    if (tid < 24) {
        const int v = tid / 3, c = tid - 3 * v;
        const int node = map_q1[(size_t)col * 8 + v] + (layer - layers[0]);      // offset 1 per layer (Q1)
        sX[tid] = coords[(size_t)node * 3 + c];
    }
    __syncthreads();
    // ---- per-cell geometry at the 125 quadrature points
    if (tid < 125) {
        const int q1 = tid / 25, q2 = (tid / 5) % 5, q3 = tid % 5;
        const double t[3] = {sQP[q1], sQP[q2], sQP[q3]};
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int a = v >> 2, b = (v >> 1) & 1, c = v & 1;
            const double Na = a ? t[0] : 1.0 - t[0], Nb = b ? t[1] : 1.0 - t[1], Nc = c ? t[2] : 1.0 - t[2];
            const double da = a ? 1.0 : -1.0, db = b ? 1.0 : -1.0, dc = c ? 1.0 : -1.0;
            const double g0 = da * Nb * Nc, g1 = Na * db * Nc, g2 = Na * Nb * dc;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double xv = sX[v * 3 + r];
                J[r][0] += xv * g0; J[r][1] += xv * g1; J[r][2] += xv * g2;
            }
        }
        double K[3][3], det;
        inv3(J, K, det);
        const double w = sQW[q1] * sQW[q2] * sQW[q3] * fabs(det);
        double G[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) G[a][b] = w * (K[a][0] * K[b][0] + K[a][1] * K[b][1] + K[a][2] * K[b][2]);
        sW[tid][0] = G[0][0]; sW[tid][1] = G[0][1]; sW[tid][2] = G[0][2];
        sW[tid][3] = G[1][1]; sW[tid][4] = G[1][2]; sW[tid][5] = G[2][2]; sW[tid][6] = w; sW[tid][7] = 0.0;
    }
    __syncthreads();
    // ---- lane roles
    const int r16 = lane & 15, kk = lane >> 4;             // row/col inside a tile, MFMA k index (channel)
    const int itile = half * 4 + wave;
    int i1, i2, i3; bool iv;
    {
        const int i = itile * 16 + r16;
        iv = i < Q4_ND;
        const int ic = iv ? i : 0;
        i1 = ic / 25; i2 = (ic / 5) % 5; i3 = ic % 5;
    }
    int j1[8], j2[8], j3[8]; bool jv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = t * 16 + r16;
        jv[t] = j < Q4_ND;
        const int jc = jv[t] ? j : 0;
        j1[t] = jc / 25; j2[t] = (jc / 5) % 5; j3[t] = jc % 5;
    }
    // B operand of lane (kk, j): Phi[kk][j] = X[q1][j1] * Y[q2][j2] * Z[q3][j3], derivative table on axis kk
    const double *tabx = kk == 0 ? sDL : sL, *taby = kk == 1 ? sDL : sL, *tabz = kk == 2 ? sDL : sL;
    // index of W_q entries forming row kk of G: (kk,0) (kk,1) (kk,2)
    const int g0 = kk == 0 ? 0 : (kk == 1 ? 1 : 2), g1 = kk == 0 ? 1 : (kk == 1 ? 3 : 4), g2 = kk == 0 ? 2 : (kk == 1 ? 4 : 5);
    fd_d4 acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = fd_d4{0.0, 0.0, 0.0, 0.0};

#pragma unroll 1
    for (int q1 = 0; q1 < Q4_NQ1; ++q1) {
#pragma unroll 1
        for (int q2 = 0; q2 < Q4_NQ1; ++q2) {
            double bxy[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) bxy[t] = jv[t] ? tabx[q1 * 5 + j1[t]] * taby[q2 * 5 + j2[t]] : 0.0;
            const double lx = sL[q1 * 5 + i1], dx = sDL[q1 * 5 + i1];
            const double ly = sL[q2 * 5 + i2], dy = sDL[q2 * 5 + i2];
            const double ax = iv ? dx * ly : 0.0;      // d/dxi1 part
            const double ay = iv ? lx * dy : 0.0;      // d/dxi2 part
            const double axy = iv ? lx * ly : 0.0;     // value in (xi1, xi2)
#pragma unroll 1
            for (int q3 = 0; q3 < Q4_NQ1; ++q3) {
                const int q = (q1 * 5 + q2) * 5 + q3;
                const double w0 = sW[q][g0], w1 = sW[q][g1], w2 = sW[q][g2], wm = sW[q][6];
                const double lz = sL[q3 * 5 + i3], dz = sDL[q3 * 5 + i3];
                const double d0 = ax * lz, d1 = ay * lz, d2 = axy * dz, ph = axy * lz;
                const double aop = kk < 3 ? (w0 * d0 + w1 * d1 + w2 * d2) : wm * ph;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const double bop = bxy[t] * tabz[q3 * 5 + j3[t]];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- scatter: C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg
    const int *tab = elemtab + ((size_t)(col - start) * nl + (layer - layers[0])) * (Q4_ND * Q4_ND);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int i = itile * 16 + kk + 4 * g;
        if (i >= Q4_ND) continue;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = t * 16 + r16;
            if (j >= Q4_ND) continue;
            const int pos = tab[i * Q4_ND + j];
            if (pos >= 0) atomicAdd(&vals[pos], acc[t][g]);
        }
    }
}
FD_REGISTER(wrap_helmholtz_q4_hex_jacobian);
