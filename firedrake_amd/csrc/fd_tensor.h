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
// (O(k^4) per cell) with all intermediates in LDS and is bound by HBM/LDS, not by arithmetic.
//
// Arguments follow the reference's positional order for an extruded loop (builder.py:962-981): start, end, layers, one
// pointer per Dat/Mat, one per distinct Map; backend-private tables follow.
#pragma once
#include "fd_wrapper.h"

typedef double fd_d4 __attribute__((ext_vector_type(4)));

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
// Q4 operator action y_e += sum_q Phi_q^T W_q (Phi_q u_e), sum-factorised: one 128-lane workgroup per cell, lane t owns
// the tensor index (a, b, c) = (t/25, t/5 % 5, t%5); every contraction is one 5-term dot product per lane per array
// with its operands in LDS.  Forward: u -> (d1 u, d2 u, d3 u, u) at the 125 Gauss points; point weights; backward: the
// transposed contractions; scatter with one fp64 atomic per DoF (extruded addressing map + 4*layer, builder.py:94-124).
// ------------------------------------------------------------------------------------------------------------------
template <class WF>
__device__ __forceinline__ void hex_q4_action(int start, int end, const int *__restrict__ layers, double *__restrict__ y,
                                              const double *__restrict__ coords, const double *__restrict__ u,
                                              const int *__restrict__ map_q4, const int *__restrict__ map_q1,
                                              const double *__restrict__ tables, WF weights) {
    __shared__ double sL[25], sDL[25], sQP[5], sQW[5], sX[24];
    __shared__ double b0[125], b1[125], b2[125], b3[125], b4[125];
    const int t = threadIdx.x;
    const int nl = layers[1] - 1 - layers[0];
    const int col = start + (int)(blockIdx.x / nl), lrel = (int)(blockIdx.x % nl);
    if (col >= end) return;
    const bool on = t < Q4_ND;
    const int a = on ? t / 25 : 0, b = on ? (t / 5) % 5 : 0, c = on ? t % 5 : 0;
    if (t < 25) { sL[t] = tables[t]; sDL[t] = tables[25 + t]; }
    if (t < 5) { sQP[t] = tables[50 + t]; sQW[t] = tables[55 + t]; }
    if (t < 24) {
        const int v = t / 3, cc = t - 3 * v;
        sX[t] = coords[(size_t)(map_q1[(size_t)col * 8 + v] + lrel) * 3 + cc];
    }
    int node = 0;
    if (on) { node = map_q4[(size_t)col * Q4_ND + t] + 4 * lrel; b0[t] = u[node]; }
    __syncthreads();
    // stage 1: contract the first index with L / DL:  b1 = L_1 u, b2 = DL_1 u   (output index (q1, i2, i3))
    if (on) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) { const double v = b0[(i * 5 + b) * 5 + c]; s0 += sL[a * 5 + i] * v; s1 += sDL[a * 5 + i] * v; }
        b1[t] = s0; b2[t] = s1;
    }
    __syncthreads();
    // stage 2: second index:  b0 = L_2 b1, b3 = L_2 b2, b4 = DL_2 b1        (output index (q1, q2, i3))
    if (on) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double v1 = b1[(a * 5 + i) * 5 + c], v2 = b2[(a * 5 + i) * 5 + c];
            s0 += sL[b * 5 + i] * v1; s1 += sL[b * 5 + i] * v2; s2 += sDL[b * 5 + i] * v1;
        }
        b0[t] = s0; b3[t] = s1; b4[t] = s2;
    }
    __syncthreads();
    // stage 3: third index -> values at the Gauss point (q1, q2, q3) = (a, b, c); point weights in registers
    double F[4] = {0.0, 0.0, 0.0, 0.0};
    if (on) {
        double g[4] = {0.0, 0.0, 0.0, 0.0};            // d1 u, d2 u, d3 u, u
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double l = sL[c * 5 + i], d = sDL[c * 5 + i];
            const double v0 = b0[(a * 5 + b) * 5 + i];
            g[3] += l * v0; g[2] += d * v0; g[0] += l * b3[(a * 5 + b) * 5 + i]; g[1] += l * b4[(a * 5 + b) * 5 + i];
        }
        const double tq[3] = {sQP[a], sQP[b], sQP[c]};
        double J[3][3], X[3], W[16];
        hex_jacobian(sX, tq, J, X);
        weights(J, X, sQW[a] * sQW[b] * sQW[c], W);
#pragma unroll
        for (int l = 0; l < 4; ++l) F[l] = W[l * 4 + 0] * g[0] + W[l * 4 + 1] * g[1] + W[l * 4 + 2] * g[2] + W[l * 4 + 3] * g[3];
    }
    __syncthreads();                                    // all reads of b0/b3/b4 done
    if (on) { b0[t] = F[0]; b1[t] = F[1]; b2[t] = F[2]; b3[t] = F[3]; }
    __syncthreads();
    // stage 4: q3 -> i3:  P0 = L_3^T F0, P1 = L_3^T F1, S = DL_3^T F2 + L_3^T F3     (index (q1, q2, i3))
    double p0 = 0.0, p1 = 0.0, sv = 0.0;
    if (on) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double l = sL[q * 5 + c], d = sDL[q * 5 + c];
            const int o = (a * 5 + b) * 5 + q;
            p0 += l * b0[o]; p1 += l * b1[o]; sv += d * b2[o] + l * b3[o];
        }
    }
    __syncthreads();
    if (on) { b0[t] = p0; b1[t] = p1; b2[t] = sv; }
    __syncthreads();
    // stage 5: q2 -> i2:  R0 = L_2^T P0, R1 = DL_2^T P1 + L_2^T S                     (index (q1, i2, i3))
    double r0 = 0.0, r1 = 0.0;
    if (on) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double l = sL[q * 5 + b], d = sDL[q * 5 + b];
            const int o = (a * 5 + q) * 5 + c;
            r0 += l * b0[o]; r1 += d * b1[o] + l * b2[o];
        }
    }
    __syncthreads();
    if (on) { b3[t] = r0; b4[t] = r1; }
    __syncthreads();
    // stage 6: q1 -> i1:  y = DL_1^T R0 + L_1^T R1
    if (on) {
        double yv = 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int o = (q * 5 + b) * 5 + c;
            yv += sDL[q * 5 + a] * b3[o] + sL[q * 5 + a] * b4[o];
        }
        atomicAdd(&y[node], yv);
    }
}

}  // namespace fdt
