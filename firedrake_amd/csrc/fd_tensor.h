// fd_tensor.h -- device templates for tensor-product (hexahedral Q_k) local kernels on extruded meshes.
//
// TSFC emits, for a form on a tensor-product element, a sum-factorised scalar kernel (tsfc/spectral.py:157-191); wrapped
// one-lane-per-cell the way builder.py:702-1008 does it, a Q4 element MATRIX would need a 125 KB private tensor per lane.
// The backend therefore recognises such kernels by a descriptor (kernel.TensorProductLocalKernel: degree, quadrature
// points per axis, and a per-quadrature-point 4 x 4 weight callback) and emits wrappers built from the templates below:
//
//   matrix:   A_e = sum_q  Phi_q^T  W_q  Phi_q          (MatSetValuesLocal ADD_VALUES of the 125 x 125 block)
//   action:   y_e = sum_q  Phi_q^T  W_q  (Phi_q u_e)     (INC into a Dat)
//
// Phi_q (4 x nd): reference-gradient components and value of every basis function at quadrature point q (products of the
// 1-D tables, cell independent), W_q (4 x 4): the point weight the callback computes from the cell geometry.  For the
// matrix K = 4 per quadrature point is exactly the K of v_mfma_f64_16x16x4_f64: one MFMA updates a 16 x 16 tile of A_e
// with one quadrature point -- a genuine dense contraction on the fp64 matrix cores.  The action is sum-factorised
// (O(k^4) per cell), one lane per line of the index cube, intermediates in LDS between the axis passes.
//
// Arguments follow the reference's positional order for an extruded loop (builder.py:962-981): start, end, layers, one
// pointer per Dat/Mat, one per distinct Map; backend-private tables follow.
#pragma once
#include "fd_wrapper.h"

namespace fdt {

constexpr int Q4_ND = 125, Q4_NQ1 = 5;

// geometry of a trilinear hexahedron (Q1 vertices, index a*4 + b*2 + c) at reference point t: J[r][s] = dx_r / dxi_s
__device__ __forceinline__ void hex_jacobian(const double *__restrict__ sX, const double t[3], double J[3][3], double X[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) { J[r][0] = J[r][1] = J[r][2] = 0.0; X[r] = 0.0; }
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const int a = v >> 2, b = (v >> 1) & 1, c = v & 1;
        const double Na = a ? t[0] : 1.0 - t[0], Nb = b ? t[1] : 1.0 - t[1], Nc = c ? t[2] : 1.0 - t[2];
        const double da = a ? 1.0 : -1.0, db = b ? 1.0 : -1.0, dc = c ? 1.0 : -1.0;
        const double g0 = da * Nb * Nc, g1 = Na * db * Nc, g2 = Na * Nb * dc, n = Na * Nb * Nc;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double xv = sX[v * 3 + r];
            J[r][0] += xv * g0; J[r][1] += xv * g1; J[r][2] += xv * g2; X[r] += xv * n;
        }
    }
}

// K = J^-1 and det J: helper for weight callbacks
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

// ------------------------------------------------------------------------------------------------------------------
// Q4 element matrix by fp64 MFMA.  Two workgroups (4 wavefronts each) per cell; wavefront w of half h owns the 16-row
// panel [16(4h+w), +16) of the padded 128 x 128 matrix: 1 x 8 tiles = 64 accumulator registers, three wavefronts per SIMD.
// Per quadrature point a lane builds 1 A-operand ((Phi^T W) for its row and k) and 8 B-operands (Phi for the eight
// column tiles) from LDS-resident 1-D tables and the per-cell W, then issues 8 MFMAs.  Scatter: fp64 atomics at
// rowptr[row] + offtab[column variant][i][j], BC rows/columns dropped through the lgmaps (parloop.py:279-302).
//   tables: L[5][5] (value of 1-D basis i at Gauss point q: L[q*5+i]), DL[5][5], QP[5], QW[5]  (60 doubles)
//   offtab: uint16 [ncol][3][125*125]: position of entry (i, j) inside its CSR row for the bottom / interior / top cell
//           of a column (interior layers of an extruded column are translates of each other)
// ------------------------------------------------------------------------------------------------------------------
template <class WF>
__device__ __forceinline__ void hex_q4_matrix(int start, int end, const int *__restrict__ layers, double *__restrict__ vals,
                                              const double *__restrict__ coords, const int *__restrict__ map_q4,
                                              const int *__restrict__ map_q1, const int *__restrict__ rowptr,
                                              const unsigned short *__restrict__ offtab, const int *__restrict__ rlg,
                                              const int *__restrict__ clg, const double *__restrict__ tables, WF weights) {
    __shared__ double sL[25], sDL[25], sQP[5], sQW[5];
    __shared__ double sX[24];
    __shared__ double sW[125][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = layers[1] - 1 - layers[0];
    const int cellid = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int col = start + cellid / nl;
    const int lrel = cellid % nl;                       // layer relative to the bottom
    if (col >= end) return;
    if (tid < 25) { sL[tid] = tables[tid]; sDL[tid] = tables[25 + tid]; }
    if (tid < 5) { sQP[tid] = tables[50 + tid]; sQW[tid] = tables[55 + tid]; }
    if (tid < 24) {
        const int v = tid / 3, c = tid - 3 * v;
        const int node = map_q1[(size_t)col * 8 + v] + lrel;       // offset 1 per layer (Q1)
        sX[tid] = coords[(size_t)node * 3 + c];
    }
    __syncthreads();
    if (tid < 125) {
        const int q1 = tid / 25, q2 = (tid / 5) % 5, q3 = tid % 5;
        const double t[3] = {sQP[q1], sQP[q2], sQP[q3]};
        double J[3][3], X[3], W[16];
        hex_jacobian(sX, t, J, X);
        weights(J, X, sQW[q1] * sQW[q2] * sQW[q3], W);
#pragma unroll
        for (int k = 0; k < 16; ++k) sW[tid][k] = W[k];
    }
    __syncthreads();
    const int r16 = lane & 15, kk = lane >> 4;             // row/col inside a tile, MFMA k index
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
    // B operand of lane (kk, j): Phi[kk][j] = X[q1][j1] * Y[q2][j2] * Z[q3][j3], derivative table on axis kk (kk = 3: value)
    const double *tabx = kk == 0 ? sDL : sL, *taby = kk == 1 ? sDL : sL, *tabz = kk == 2 ? sDL : sL;
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
                const double lz = sL[q3 * 5 + i3], dz = sDL[q3 * 5 + i3];
                // A operand: (Phi^T W)[i][kk] = sum_l Phi[l][i] W[l][kk]
                const double aop = ax * lz * sW[q][kk] + ay * lz * sW[q][4 + kk] + axy * dz * sW[q][8 + kk] + axy * lz * sW[q][12 + kk];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const double bop = bxy[t] * tabz[q3 * 5 + j3[t]];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- scatter: C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg
    const int variant = (lrel == 0) ? 0 : ((lrel == nl - 1) ? 2 : 1);
    const unsigned short *tab = offtab + ((size_t)(col - start) * 3 + variant) * (Q4_ND * Q4_ND);
    const int *mrow = map_q4 + (size_t)col * Q4_ND;
    int cn[8]; bool cok[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = t * 16 + r16;
        cok[t] = j < Q4_ND;
        cn[t] = cok[t] ? mrow[j] + 4 * lrel : 0;
        if (cok[t] && clg) cok[t] = clg[cn[t]] >= 0;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int i = itile * 16 + kk + 4 * g;
        if (i >= Q4_ND) continue;
        const int rn = mrow[i] + 4 * lrel;
        if (rlg && rlg[rn] < 0) continue;
        const size_t r0 = (size_t)rowptr[rn];
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (cok[t]) atomicAdd(&vals[r0 + tab[i * Q4_ND + t * 16 + r16]], acc[t][g]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Q4 operator action y_e += sum_q Phi_q^T W_q (Phi_q u_e), sum-factorised, ONE LANE PER LINE of the 5 x 5 x 5 index cube:
// a contraction along one axis reads the 5 entries of a line once and produces the line's 5 outputs in registers
// (25 FMAs per table), so a cell costs 25 lanes per stage and ~100 LDS accesses per lane in total -- against 135 reads per
// lane and 125 lanes per cell when every lane owns one output (LDS-bound: the first version of this template ran at the
// ds_read_b64 rate).  A 128-lane workgroup carries Q4_ACT_CELLS = 5 consecutive cells of the (column, layer) space; lane
// t -> cell slot t / 25, line t % 25 = (p, r).
//   stage 1  (lines along i1, u straight from the lane's own global loads): V = L_1 u, D1 = DL_1 u        -> A  (q1, i2, i3)
//   stage 2  (lines along i2):  VV = L_2 V,  D1V = L_2 D1,  D2 = DL_2 V                                    -> B  (q1, q2, i3)
//   stage 3+4 (lines along i3, all in registers): values and reference gradient at the line's 5 Gauss points, the
//            point weights (geometry of a trilinear hexahedron is affine along a line: 6 vectors per line, one
//            interpolation per point), F = W g, then the transposed contraction q3 -> i3                  -> A  (q1, q2, i3)
//   stage 5  (q2 -> i2), stage 6 (q1 -> i1) and one fp64 atomic per DoF (extruded addressing map + 4*layer, builder.py:94-124).
// ------------------------------------------------------------------------------------------------------------------
constexpr int Q4_ACT_CELLS = 5;

template <class WF>
__device__ __forceinline__ void hex_q4_action(int start, int end, const int *__restrict__ layers, double *__restrict__ y,
                                              const double *__restrict__ coords, const double *__restrict__ u,
                                              const int *__restrict__ map_q4, const int *__restrict__ map_q1,
                                              const double *__restrict__ tables, WF weights) {
    __shared__ double sA[Q4_ACT_CELLS][3][Q4_ND], sB[Q4_ACT_CELLS][3][Q4_ND], sX[Q4_ACT_CELLS][24];
    const int t = threadIdx.x;
    const int nl = layers[1] - 1 - layers[0];
    const int ncell = (end - start) * nl;
    const int first = (int)blockIdx.x * Q4_ACT_CELLS;
    {
        const int ks = t / 24, e = t - 24 * ks;         // 120 lanes fetch the 5 x 8 vertices
        const int cs = first + ks;
        if (ks < Q4_ACT_CELLS && cs < ncell) {
            const int v = e / 3, cc = e - 3 * v;
            sX[ks][e] = coords[(size_t)(map_q1[(size_t)(start + cs / nl) * 8 + v] + cs % nl) * 3 + cc];
        }
    }
    const int k = t / 25, l = t - 25 * k, p = l / 5, r = l - 5 * p;
    const bool on = k < Q4_ACT_CELLS && first + k < ncell;
    const int ka = on ? k : 0;
    double (*A)[Q4_ND] = sA[ka], (*B)[Q4_ND] = sB[ka];
    double L[25], DL[25];                               // wavefront-uniform: scalar registers
#pragma unroll
    for (int i = 0; i < 25; ++i) { L[i] = tables[i]; DL[i] = tables[25 + i]; }
    int node[5];
    if (on) {
        const int cell = first + k;
        const int *mrow = map_q4 + (size_t)(start + cell / nl) * Q4_ND;
        const int lrel4 = 4 * (cell % nl);
        double uv[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { node[i] = mrow[i * 25 + l] + lrel4; uv[i] = u[node[i]]; }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = 0; i < 5; ++i) { s0 += L[q * 5 + i] * uv[i]; s1 += DL[q * 5 + i] * uv[i]; }
            A[0][q * 25 + l] = s0; A[1][q * 25 + l] = s1;
        }
    }
    __syncthreads();
    if (on) {                                           // stage 2: (p, r) = (q1, i3)
        double v[5], d[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { v[i] = A[0][p * 25 + i * 5 + r]; d[i] = A[1][p * 25 + i * 5 + r]; }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < 5; ++i) { s0 += L[q * 5 + i] * v[i]; s1 += L[q * 5 + i] * d[i]; s2 += DL[q * 5 + i] * v[i]; }
            const int o = (p * 5 + q) * 5 + r;
            B[0][o] = s0; B[1][o] = s1; B[2][o] = s2;
        }
    }
    __syncthreads();
    if (on) {                                           // stages 3 + 4: (p, r) = (q1, q2); the line is index l*5 + i3
        double vv[5], d1[5], d2[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { vv[i] = B[0][l * 5 + i]; d1[i] = B[1][l * 5 + i]; d2[i] = B[2][l * 5 + i]; }
        // geometry along the line: with (t0, t1) fixed, dx/dt0, dx/dt1 and x are affine in t2 and dx/dt2 is constant
        const double t0 = tables[50 + p], t1 = tables[50 + r], w01 = tables[55 + p] * tables[55 + r];
        const double *X8 = sX[ka];
        double G0[3][2], G1[3][2], P[3][2];             // [component][bottom / top face]
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const double x00 = X8[(0 + f) * 3 + c], x01 = X8[(2 + f) * 3 + c], x10 = X8[(4 + f) * 3 + c], x11 = X8[(6 + f) * 3 + c];
                G0[c][f] = (1.0 - t1) * (x10 - x00) + t1 * (x11 - x01);
                G1[c][f] = (1.0 - t0) * (x01 - x00) + t0 * (x11 - x10);
                P[c][f] = (1.0 - t0) * ((1.0 - t1) * x00 + t1 * x01) + t0 * ((1.0 - t1) * x10 + t1 * x11);
            }
        double p0[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, p1[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, sv[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double g[4] = {0.0, 0.0, 0.0, 0.0};         // d1 u, d2 u, d3 u, u at the Gauss point (p, r, q)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                g[0] += L[q * 5 + i] * d1[i]; g[1] += L[q * 5 + i] * d2[i]; g[2] += DL[q * 5 + i] * vv[i]; g[3] += L[q * 5 + i] * vv[i];
            }
            const double t2 = tables[50 + q];
            double J[3][3], X[3], W[16], F[4];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                J[c][0] = G0[c][0] + t2 * (G0[c][1] - G0[c][0]);
                J[c][1] = G1[c][0] + t2 * (G1[c][1] - G1[c][0]);
                J[c][2] = P[c][1] - P[c][0];
                X[c] = P[c][0] + t2 * (P[c][1] - P[c][0]);
            }
            weights(J, X, w01 * tables[55 + q], W);
#pragma unroll
            for (int m = 0; m < 4; ++m) F[m] = W[m * 4 + 0] * g[0] + W[m * 4 + 1] * g[1] + W[m * 4 + 2] * g[2] + W[m * 4 + 3] * g[3];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                p0[i] += L[q * 5 + i] * F[0]; p1[i] += L[q * 5 + i] * F[1]; sv[i] += DL[q * 5 + i] * F[2] + L[q * 5 + i] * F[3];
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) { A[0][l * 5 + i] = p0[i]; A[1][l * 5 + i] = p1[i]; A[2][l * 5 + i] = sv[i]; }
    }
    __syncthreads();
    if (on) {                                           // stage 5: (p, r) = (q1, i3), q2 -> i2
        double a0[5], a1[5], a2[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { const int o = (p * 5 + q) * 5 + r; a0[q] = A[0][o]; a1[q] = A[1][o]; a2[q] = A[2][o]; }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            double r0 = 0.0, r1 = 0.0;
#pragma unroll
            for (int q = 0; q < 5; ++q) { r0 += L[q * 5 + i] * a0[q]; r1 += DL[q * 5 + i] * a1[q] + L[q * 5 + i] * a2[q]; }
            B[0][p * 25 + i * 5 + r] = r0; B[1][p * 25 + i * 5 + r] = r1;
        }
    }
    __syncthreads();
    if (on) {                                           // stage 6: line l = (i2, i3), q1 -> i1
        double r0[5], r1[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { r0[q] = B[0][q * 25 + l]; r1[q] = B[1][q * 25 + l]; }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            double yv = 0.0;
#pragma unroll
            for (int q = 0; q < 5; ++q) yv += DL[q * 5 + i] * r0[q] + L[q * 5 + i] * r1[q];
            atomicAdd(&y[node[i]], yv);
        }
    }
}

}  // namespace fdt
