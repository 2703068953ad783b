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
#include <type_traits>

namespace fdt {

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

// values of the 8 trilinear vertex functions (index a*4 + b*2 + c) at reference point t: coefficient arguments on the Q1 map
__device__ __forceinline__ void hex_shape(const double t[3], double N[8]) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const int a = v >> 2, b = (v >> 1) & 1, c = v & 1;
        N[v] = (a ? t[0] : 1.0 - t[0]) * (b ? t[1] : 1.0 - t[1]) * (c ? t[2] : 1.0 - t[2]);
    }
}

// ... and their reference gradients dN[v][a] = dN_v / dxi_a (coefficient gradients of Q1 fields)
__device__ __forceinline__ void hex_shape_grad(const double t[3], double dN[8][3]) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const int a = v >> 2, b = (v >> 1) & 1, c = v & 1;
        const double Na = a ? t[0] : 1.0 - t[0], Nb = b ? t[1] : 1.0 - t[1], Nc = c ? t[2] : 1.0 - t[2];
        dN[v][0] = (a ? 1.0 : -1.0) * Nb * Nc; dN[v][1] = Na * (b ? 1.0 : -1.0) * Nc; dN[v][2] = Na * Nb * (c ? 1.0 : -1.0);
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
// Q_k element matrix by fp64 MFMA.  K1 = k + 1 nodes per axis (ND = K1^3 basis functions), Q1 Gauss points per axis.
// The ND x ND matrix is padded to NT = ceil(ND / 16) tiles per side; a wavefront owns one 16-row panel = 1 x NT tiles
// (4 NT accumulator registers), a workgroup carries WPB = tp_waves(NT) panels and a cell takes NT / WPB workgroups
// (Q4: ND = 125, NT = 8, two workgroups of four wavefronts, 64 accumulator registers -> three wavefronts per SIMD).
// Per quadrature point a lane builds 1 A-operand ((Phi^T W) for its row and k) and NT B-operands (Phi for the column
// tiles) from LDS-resident 1-D tables and the per-cell W, then issues NT MFMAs.  Scatter: fp64 atomics at
// rowptr[row] + offtab[column variant][i][j], BC rows/columns dropped through the lgmaps (parloop.py:279-302).
//   tables: L[Q1][K1] (value of 1-D basis i at Gauss point q: L[q*K1+i]), DL[Q1][K1], QP[Q1], QW[Q1]
//   offtab: uint16 [ncol][3][ND*ND]: position of entry (i, j) inside its CSR row for the bottom / interior / top cell
//           of a column (interior layers of an extruded column are translates of each other)
// ------------------------------------------------------------------------------------------------------------------
constexpr int tp_tiles(int k1) { return (k1 * k1 * k1 + 15) / 16; }
constexpr int tp_waves(int nt) { return nt % 4 == 0 ? 4 : (nt % 2 == 0 ? 2 : 1); }     // wavefronts per workgroup: divides NT
// Q5 and beyond (NT = 14, 22, 32, 46 tiles per side): a full 16-row panel is 4 NT accumulator registers -- one wavefront per SIMD for
// Q5, more than a lane has from Q6 on -- so the panel is cut into tp_col_splits(NT) column chunks of tp_col_tiles(NT) <= 8 tiles (the
// 64 accumulator registers of Q4) and a wavefront owns one (panel, chunk): NT * splits wavefront items per cell, tp_waves(items) of
// them per workgroup.  A chunk's wavefront builds the panel's A operand again (one of NTC + 1 operands per point, as for Q4).  When
// the chunk count does not divide NT the first NT % splits chunks hold one tile more (Q6: 22 = 8 + 7 + 7): the chunk body of
// hex_qk_matrix is instantiated for both sizes and picked per wavefront -- no MFMA on padding tiles (a run-time guard inside one
// instantiation cost the accumulators their static register indices: Q6 18.4 -> 21.8 ms).  Measured (profiles/r6s_ab_tensor_panels.txt):
// Q5 whole panels 9.09 ms = 0.46 of the MFMA peak, 2 x 7 tiles 6.52 ms = 0.64; Q4 (8 tiles) whole 0.685, 2 x 4 0.638: whole up to 8.
// (the three thresholds are tuning constants of firedrake_amd/configuration.py -- tp_max_panel_tiles, tp_chunk_tiles, tp_weight_lds --
// that codegen.generate_tensor_wrapper defines ahead of this header when they differ from the defaults below; the host-sim tests
// lower them to run both mechanisms on small elements)
#ifndef FD_TP_MAX_PANEL_TILES
#define FD_TP_MAX_PANEL_TILES 8
#endif
#ifndef FD_TP_CHUNK_TILES
#define FD_TP_CHUNK_TILES 8
#endif
#ifndef FD_TP_WEIGHT_LDS
#define FD_TP_WEIGHT_LDS (48 * 1024)
#endif
constexpr int tp_col_splits(int nt) { return nt <= FD_TP_MAX_PANEL_TILES ? 1 : (nt + FD_TP_CHUNK_TILES - 1) / FD_TP_CHUNK_TILES; }
constexpr int tp_col_tiles(int nt) { return (nt + tp_col_splits(nt) - 1) / tp_col_splits(nt); }
// the per-point weights of ALL Gauss points of a cell stay in LDS up to 48 KB (Q5 with 7 points per axis: 43 KB); beyond, the
// matrix template computes them one q1-slab at a time (Q1^2 points, two more barriers per slab)
constexpr bool tp_weight_slabs(int q1, int np_) { return q1 * q1 * q1 * 16 * np_ * 8 > FD_TP_WEIGHT_LDS; }
constexpr bool tp_fused(int nt, int d) { return d > 1 && 4 * nt * d * d <= 96; }         // all D^2 component pairs in one workgroup

// Coefficient arguments (NC > 0): READ Dats on the Q_k map -- what TSFC passes as w_k to a variable-coefficient or linearised
// nonlinear form (tsfc/kernel_interface/firedrake_loopy.py:432-522; evaluated at the quadrature points in tsfc/fem.py:742-805).
// Their values at the NQ Gauss points of the cell are computed sum-factorised (three 1-D contractions through LDS, ~2 K1 NQ
// FMAs per coefficient against 2 ND^2 NQ 4 for the element matrix) and handed to the weight callback as C[0..NC).
// GRAD: the REFERENCE GRADIENT of every coefficient at the points as well (the linearisation of a form nonlinear in grad(u0)
// needs it: tsfc/fem.py:742-805 tabulates the derivative tables for it) -- the derivative table on one axis at a time, three more
// results out of the same three passes: sC[m*4 + 0] the value, sC[m*4 + 1 + a] d/dxi_a.
template <int K1, int Q1, int NC, int NTHR, bool GRAD>
__device__ __forceinline__ void hex_qk_coefficients(const double *const (&cf)[NC > 0 ? NC : 1], const int *__restrict__ mrow, int lo,
                                                    const double *sL, const double *sDL, double (*sC)[Q1 * Q1 * Q1]) {
    constexpr int ND = K1 * K1 * K1, NQ = Q1 * Q1 * Q1, G1 = GRAD ? 2 : 1, G2 = GRAD ? 3 : 1, CW = GRAD ? 4 : 1;
    constexpr int N1 = Q1 * K1 * K1, N2 = Q1 * Q1 * K1;
    __shared__ double sU[ND], sT1[G1][N1], sT2[G2][N2];
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int m = 0; m < NC; ++m) {
        for (int i = tid; i < ND; i += NTHR) sU[i] = cf[m][mrow[i] + lo];
        __syncthreads();
        for (int o = tid; o < N1; o += NTHR) {                    // (q1, i2, i3) <- sum over i1
            const int q1 = o / (K1 * K1), r = o - q1 * (K1 * K1);
            double v = 0.0, d = 0.0;
#pragma unroll
            for (int i = 0; i < K1; ++i) { v += sL[q1 * K1 + i] * sU[i * K1 * K1 + r]; if (GRAD) d += sDL[q1 * K1 + i] * sU[i * K1 * K1 + r]; }
            sT1[0][o] = v;
            if (GRAD) sT1[G1 - 1][o] = d;
        }
        __syncthreads();
        for (int o = tid; o < N2; o += NTHR) {                    // (q1, q2, i3) <- sum over i2
            const int q1 = o / (Q1 * K1), q2 = (o / K1) % Q1, i3 = o % K1;
            double v = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int i = 0; i < K1; ++i) {
                const double tv = sT1[0][(q1 * K1 + i) * K1 + i3];
                v += sL[q2 * K1 + i] * tv;
                if (GRAD) { d1 += sL[q2 * K1 + i] * sT1[G1 - 1][(q1 * K1 + i) * K1 + i3]; d2 += sDL[q2 * K1 + i] * tv; }
            }
            sT2[0][o] = v;
            if (GRAD) { sT2[G2 > 1 ? 1 : 0][o] = d1; sT2[G2 - 1][o] = d2; }
        }
        __syncthreads();
        for (int q = tid; q < NQ; q += NTHR) {                    // (q1, q2, q3) <- sum over i3
            const int q12 = q / Q1, q3 = q - q12 * Q1;
            double v = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
            for (int i = 0; i < K1; ++i) {
                const double tv = sT2[0][q12 * K1 + i];
                v += sL[q3 * K1 + i] * tv;
                if (GRAD) { g0 += sL[q3 * K1 + i] * sT2[G2 > 1 ? 1 : 0][q12 * K1 + i]; g1 += sL[q3 * K1 + i] * sT2[G2 - 1][q12 * K1 + i]; g2 += sDL[q3 * K1 + i] * tv; }
            }
            sC[m * CW][q] = v;
            if (GRAD) { sC[m * CW + 1][q] = g0; sC[m * CW + 2][q] = g1; sC[m * CW + 3][q] = g2; }
        }
        __syncthreads();
    }
}

// N1 further coefficient arguments live on the Q1 map of the coordinates (a piecewise-trilinear diffusivity on a Q4 problem): their 8
// vertex values per cell are staged next to the coordinates and interpolated at the point; the callback sees C = [Q_k ..., Q1 ...]
// and, with GRAD, DC[3*m + a] = d C[m] / d xi_a in the same order.
//
// D > 1: a VECTOR-VALUED space (Q_k)^D -- Mat dims (D, D), element tensor t[(i*D + p)][(j*D + r)] (MatSetValuesBlockedLocal,
// builder.py:573-625).  The callback fills the point's 4D x 4D block weight, W[((p*4 + l) * 4D) + r*4 + k] coupling reference
// component l of test component p with reference component k of trial component r; the (p, r) blocks are D^2 independent scalar
// contractions sum_q Phi_q^T W_q^{pr} Phi_q, so a cell takes D^2 times the workgroups (the pair index fastest: the workgroups of a
// cell run together and share its coordinates in the caches) and each keeps only its own 4 x 4 slice of W per point in LDS.  The
// slice is picked with COMPILE-TIME indices (one instantiation of the weights phase per pair, selected by a switch), so the
// callback's other 16 (D^2 - 1) results are dead code in each instantiation and W never has to exist in full.
// CSR: scalar row (node, p) starts at node_rowptr[node]*D*D + p*rowlen*D, column (k-th node of the row, r) sits at k*D + r.
// Small elements (tp_fused: 4 NT D^2 <= 96 accumulator registers -- (Q1)^3, (Q2)^3, (Q3)^2) keep ALL D^2 blocks in one workgroup
// instead: the geometry, the point weights and the B operands are computed once per cell and every Gauss point feeds D^2 NT MFMAs
// instead of NT -- and the whole element matrix can leave through LDS transposed (below: the scatter of the accumulator layout is what
// bounds a vector-valued matrix -- entries D doubles apart, ~3x the cache lines per wavefront instruction of the scalar case:
// (Q2)^3 elasticity, n = 24, 2.10 ms = 0.12 of the fp64 MFMA peak, fused or not; profiles/r5k_tensor_forms.txt).
template <int K1, int Q1, int NC, int N1, bool GRAD, int D, int P, int R, int NTHR, int SWW, class WF>
__device__ __forceinline__ void hex_qk_point_weights(const double *sX, const double *sQP, const double *sQW,
                                                     const double (*sC)[Q1 * Q1 * Q1], const double (*sV1)[8],
                                                     double (*sW)[SWW], WF weights, int q_begin = 0, int q_count = Q1 * Q1 * Q1) {
    constexpr int CW = GRAD ? 4 : 1, NCT = NC + N1 > 0 ? NC + N1 : 1;
    for (int q = q_begin + (int)threadIdx.x; q < q_begin + q_count; q += NTHR) {
        const int q1 = q / (Q1 * Q1), q2 = (q / Q1) % Q1, q3 = q % Q1;
        const double t[3] = {sQP[q1], sQP[q2], sQP[q3]};
        double J[3][3], X[3], W[16 * D * D], C[NCT], DC[3 * NCT];
        hex_jacobian(sX, t, J, X);
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            C[m] = sC[m * CW][q];
            if (GRAD) { DC[3 * m] = sC[m * CW + 1][q]; DC[3 * m + 1] = sC[m * CW + 2][q]; DC[3 * m + 2] = sC[m * CW + 3][q]; }
        }
        if constexpr (N1 > 0) {
            double N[8], dN[8][3];
            hex_shape(t, N);
            if (GRAD) hex_shape_grad(t, dN);
#pragma unroll
            for (int m = 0; m < N1; ++m) {
                double v = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v += N[k] * sV1[m][k];
                    if (GRAD) { g0 += dN[k][0] * sV1[m][k]; g1 += dN[k][1] * sV1[m][k]; g2 += dN[k][2] * sV1[m][k]; }
                }
                C[NC + m] = v;
                if (GRAD) { DC[3 * (NC + m)] = g0; DC[3 * (NC + m) + 1] = g1; DC[3 * (NC + m) + 2] = g2; }
            }
        }
        weights(J, X, sQW[q1] * sQW[q2] * sQW[q3], C, DC, W);
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int k = 0; k < 4; ++k) sW[q - q_begin][(SWW > 16 ? (P * D + R) * 16 : 0) + l * 4 + k] = W[((P * 4 + l) * 4 * D) + R * 4 + k];
    }
}

template <int K1, int Q1, int NC, int N1, bool GRAD, int D, class WF>
__device__ __forceinline__ void hex_qk_matrix(int start, int end, const int *__restrict__ layers, double *__restrict__ vals,
                                              const double *__restrict__ coords, const double *const (&cf)[NC > 0 ? NC : 1],
                                              const double *const (&c1)[N1 > 0 ? N1 : 1],
                                              const int *__restrict__ map_qk,
                                              const int *__restrict__ map_q1, const fd_nnz_t *__restrict__ rowptr,
                                              const unsigned short *__restrict__ offtab, const int *__restrict__ rlg,
                                              const int *__restrict__ clg, const double *__restrict__ tables, WF weights) {
    constexpr int ND = K1 * K1 * K1, NQ = Q1 * Q1 * Q1, NT = tp_tiles(K1), NCS = tp_col_splits(NT);
    constexpr int NI = NT * NCS, WPB = tp_waves(NI), WGC = NI / WPB, NTAB = Q1 * K1;     // wavefront items (panel, column chunk) per cell
    constexpr int CW = GRAD ? 4 : 1;
    constexpr bool FUSED = tp_fused(NT, D) && NCS == 1 && !tp_weight_slabs(Q1, D * D);     // (small elements, all their weights in LDS)
    constexpr int NP = FUSED ? D * D : 1, DG = FUSED ? 1 : D * D;      // pairs per workgroup, workgroups per (cell, panel group)
    constexpr bool SLAB = tp_weight_slabs(Q1, NP);                     // point weights one q1-slab at a time
    constexpr int NQW = SLAB ? Q1 * Q1 : NQ;
    static_assert(D >= 1 && D <= 3, "vector-valued Q_k spaces of up to three components");
    __shared__ double sL[NTAB], sDL[NTAB], sQP[Q1], sQW[Q1];
    __shared__ double sX[24];
    __shared__ double sW[NQW][16 * NP];
    __shared__ double sC[NC > 0 ? NC * CW : 1][NQ];
    __shared__ double sV1[N1 > 0 ? N1 : 1][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = layers[1] - 1 - layers[0];
    // (unsigned, like blockIdx: the compiler then knows every index below is non-negative and keeps the table pointers in scalar
    // registers with 32-bit lane offsets -- with signed ids it fell back to 64-bit address arithmetic per access, 20 % of the kernel)
    const unsigned wg = blockIdx.x / (unsigned)DG, pr = blockIdx.x - wg * (unsigned)DG;   // (test, trial) component pair
    const int cp = pr / (unsigned)D, cr = pr - cp * (unsigned)D;
    const int cellid = wg / (unsigned)WGC, part = wg - cellid * (unsigned)WGC;
    const int col = start + cellid / nl;
    const int lrel = cellid % nl;                       // layer relative to the bottom
    if (col >= end) return;
    for (int i = tid; i < NTAB; i += WPB * 64) { sL[i] = tables[i]; sDL[i] = tables[NTAB + i]; }
    if (tid < Q1) { sQP[tid] = tables[2 * NTAB + tid]; sQW[tid] = tables[2 * NTAB + Q1 + tid]; }
    if (tid < 24) {
        const int v = tid / 3, c = tid - 3 * v;
        const int node = map_q1[(size_t)col * 8 + v] + lrel;       // offset 1 per layer (Q1)
        sX[tid] = coords[(size_t)node * 3 + c];
    }
    if constexpr (N1 > 0)
        for (int o = tid; o < 8 * N1; o += WPB * 64) sV1[o >> 3][o & 7] = c1[o >> 3][map_q1[(size_t)col * 8 + (o & 7)] + lrel];
    __syncthreads();
    if constexpr (NC > 0)
        hex_qk_coefficients<K1, Q1, NC, WPB * 64, GRAD>(cf, map_qk + (size_t)col * ND, (K1 - 1) * lrel, sL, sDL, sC);
    // the point weights of the Gauss points [qb, qb + qc) into sW[0 .. qc): every point of the cell at once, or -- SLAB -- the
    // Q1^2 points of one q1 plane ahead of that plane's MFMAs
    auto fill_weights = [&](int qb, int qc) {
#define FD_TP_PAIR(P, R)                                                                                                            \
    case (P) * 3 + (R):                                                                                                             \
        if constexpr ((P) < D && (R) < D && !FUSED)                                                                                 \
            hex_qk_point_weights<K1, Q1, NC, N1, GRAD, D, (P), (R), WPB * 64, 16>(sX, sQP, sQW, sC, sV1, sW, weights, qb, qc);      \
        break;
        if constexpr (D == 1) {
            hex_qk_point_weights<K1, Q1, NC, N1, GRAD, 1, 0, 0, WPB * 64, 16>(sX, sQP, sQW, sC, sV1, sW, weights, qb, qc);
        } else if constexpr (FUSED) {
            // (one pass per pair, each with its own compile-time slice: filling all D^2 slices from one callback evaluation keeps the
            // whole 4D x 4D weight live -- 288 registers for D = 3)
#define FD_TP_SLICE(P, R)                                                                                                           \
            if constexpr ((P) < D && (R) < D)                                                                                       \
                hex_qk_point_weights<K1, Q1, NC, N1, GRAD, D, (P), (R), WPB * 64, 16 * NP>(sX, sQP, sQW, sC, sV1, sW, weights, qb, qc);
            FD_TP_SLICE(0, 0) FD_TP_SLICE(0, 1) FD_TP_SLICE(0, 2) FD_TP_SLICE(1, 0) FD_TP_SLICE(1, 1) FD_TP_SLICE(1, 2)
            FD_TP_SLICE(2, 0) FD_TP_SLICE(2, 1) FD_TP_SLICE(2, 2)
#undef FD_TP_SLICE
        } else {
            switch (cp * 3 + cr) {
                FD_TP_PAIR(0, 0) FD_TP_PAIR(0, 1) FD_TP_PAIR(0, 2) FD_TP_PAIR(1, 0) FD_TP_PAIR(1, 1) FD_TP_PAIR(1, 2)
                FD_TP_PAIR(2, 0) FD_TP_PAIR(2, 1) FD_TP_PAIR(2, 2)
            default: break;
            }
        }
#undef FD_TP_PAIR
    };
    if constexpr (!SLAB) {
        fill_weights(0, NQ);
        __syncthreads();
    }
    const int r16 = lane & 15, kk = lane >> 4;             // row/col inside a tile, MFMA k index
    const int item = part * WPB + wave;
    const int itile = NCS == 1 ? item : item / NCS;        // 16-row panel
    // column chunks of unequal size when NCS does not divide NT (Q6: 22 = 8 + 7 + 7, Q8: 46 = 4 x 8 + 2 x 7): the first NT % NCS chunks
    // hold one tile more -- two instantiations of the chunk body, picked per wavefront, and no MFMA is issued on a padding tile
    constexpr int CB = NT / NCS, CR = NT % NCS;
    const int csplit = NCS == 1 ? 0 : fdw::wave_uniform(item - itile * NCS);
    auto chunk = [&](auto ntc_c, const int tc0) {
    constexpr int NTC = decltype(ntc_c)::value;
    int i1, i2, i3; bool iv;
    {
        const int i = itile * 16 + r16;
        iv = i < ND;
        const int ic = iv ? i : 0;
        i1 = ic / (K1 * K1); i2 = (ic / K1) % K1; i3 = ic % K1;
    }
    int j1[NTC], j2[NTC], j3[NTC]; bool jv[NTC];
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
        const int j = (tc0 + t) * 16 + r16;
        jv[t] = j < ND;
        const int jc = jv[t] ? j : 0;
        j1[t] = jc / (K1 * K1); j2[t] = (jc / K1) % K1; j3[t] = jc % K1;
    }
    // B operand of lane (kk, j): Phi[kk][j] = X[q1][j1] * Y[q2][j2] * Z[q3][j3], derivative table on axis kk (kk = 3: value)
    const double *tabx = kk == 0 ? sDL : sL, *taby = kk == 1 ? sDL : sL, *tabz = kk == 2 ? sDL : sL;
    fd_d4 acc[NP][NTC];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int b = 0; b < NTC; ++b) acc[a][b] = fd_d4{0.0, 0.0, 0.0, 0.0};

#pragma unroll 1
    for (int q1 = 0; q1 < Q1; ++q1) {
        if constexpr (SLAB) {
            __syncthreads();                                // (the previous plane's MFMAs have read their weights)
            fill_weights(q1 * Q1 * Q1, Q1 * Q1);
            __syncthreads();
        }
#pragma unroll 1
        for (int q2 = 0; q2 < Q1; ++q2) {
            double bxy[NTC];
#pragma unroll
            for (int t = 0; t < NTC; ++t) bxy[t] = jv[t] ? tabx[q1 * K1 + j1[t]] * taby[q2 * K1 + j2[t]] : 0.0;
            const double lx = sL[q1 * K1 + i1], dx = sDL[q1 * K1 + i1];
            const double ly = sL[q2 * K1 + i2], dy = sDL[q2 * K1 + i2];
            const double ax = iv ? dx * ly : 0.0;      // d/dxi1 part
            const double ay = iv ? lx * dy : 0.0;      // d/dxi2 part
            const double axy = iv ? lx * ly : 0.0;     // value in (xi1, xi2)
#pragma unroll 1
            for (int q3 = 0; q3 < Q1; ++q3) {
                const int q = SLAB ? q2 * Q1 + q3 : (q1 * Q1 + q2) * Q1 + q3;       // the point's slot in sW
                const double lz = sL[q3 * K1 + i3], dz = sDL[q3 * K1 + i3];
                // A operand: (Phi^T W)[i][kk] = sum_l Phi[l][i] W[l][kk]
                const double a0 = ax * lz, a1 = ay * lz, a2 = axy * dz, a3 = axy * lz;
                // (requesting the NT table values of the B operands together ahead of the MFMAs instead of one by one between them --
                // back-to-back MFMAs in the ISA, 16 more registers -- measured the same 9.80 ms at n = 32: three wavefronts per SIMD
                // already keep the matrix pipe fed; profiles/r5j_ab_c3_tree.txt)
                if constexpr (NP == 1) {
                    const double aop = a0 * sW[q][kk] + a1 * sW[q][4 + kk] + a2 * sW[q][8 + kk] + a3 * sW[q][12 + kk];
#pragma unroll
                    for (int t = 0; t < NTC; ++t) {
                        const double bop = bxy[t] * tabz[q3 * K1 + j3[t]];
                        acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[0][t], 0, 0, 0);
                    }
                } else {
                    double bop[NTC];
#pragma unroll
                    for (int t = 0; t < NTC; ++t) bop[t] = bxy[t] * tabz[q3 * K1 + j3[t]];
#pragma unroll
                    for (int a = 0; a < NP; ++a) {
                        const double *w = &sW[q][a * 16];
                        const double aop = a0 * w[kk] + a1 * w[4 + kk] + a2 * w[8 + kk] + a3 * w[12 + kk];
#pragma unroll
                        for (int t = 0; t < NTC; ++t) acc[a][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop[t], acc[a][t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- scatter: C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg
    constexpr int OFF = K1 - 1;                          // node offset per layer (CG_k on the vertical interval)
    const int variant = (lrel == 0) ? 0 : ((lrel == nl - 1) ? 2 : 1);
    const unsigned short *tab = offtab + ((size_t)(col - start) * 3 + variant) * (ND * ND);
    const int *mrow = map_qk + (size_t)col * ND;
    if constexpr (FUSED) {
        // All D^2 blocks of the cell are in this workgroup's registers: the element matrix goes through LDS a few node rows at a time
        // (the buffer of the point weights, free now) and leaves it TRANSPOSED -- consecutive lanes hold consecutive column nodes of
        // one scalar row, each with its D trial components, i.e. contiguous runs of the CSR row.  In the accumulator layout a
        // wavefront instruction touched 4 rows x 16 column nodes D doubles apart (~3x the cache lines of the scalar case); here it
        // touches one row's runs.  Measured on (Q2)^3 elasticity, n = 24: 2.10 -> 2.00 ms -- the kernel runs at the rate of its 90.7 M
        // fp64 atomics (45 G/s; the scalar Q4 matrix kernel, MFMA-bound, issues 52 G/s) whatever lines they fall on
        // (profiles/r5q_tensor_forms.txt).
        constexpr int RL = ND * D, CAP = NQ * 16 * NP, NR = (CAP / (D * RL)) < ND ? (CAP / (D * RL)) : ND, NTH = WPB * 64, U = 4;
        static_assert(NR >= 1, "the weight buffer holds at least one node row of the element matrix");
        __shared__ fd_nnz_t sRowP[ND];
        __shared__ int sRowL[ND];
        __shared__ unsigned char sRowOk[ND], sColOk[ND];
        double *sE = &sW[0][0];
        if (tid < ND) {
            const int nd_ = mrow[tid] + OFF * lrel;
            sRowOk[tid] = !(rlg && rlg[nd_] < 0);
            sColOk[tid] = !(clg && clg[nd_] < 0);
            const fd_nnz_t a_ = rowptr[nd_];
            sRowP[tid] = a_;
            sRowL[tid] = (int)(rowptr[nd_ + 1] - a_);
        }
#pragma unroll 1
        for (int i0 = 0; i0 < ND; i0 += NR) {
            const int i1 = i0 + NR < ND ? i0 + NR : ND;
            __syncthreads();                            // (the MFMA loop's reads of sW / the previous round's reads of sE are done)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = itile * 16 + kk + 4 * g;
                if (i < i0 || i >= i1) continue;
#pragma unroll
                for (int a = 0; a < NP; ++a)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int j = t * 16 + r16;
                        if (j < ND) sE[((i - i0) * D + a / D) * RL + j * D + a % D] = acc[a][t][g];
                    }
            }
            __syncthreads();
            const int items = (i1 - i0) * D * ND;       // (node row, test component, column node): D contiguous entries each
#pragma unroll 1
            for (int o0 = tid; o0 < items; o0 += U * NTH) {
                unsigned short ps[U];
#pragma unroll
                for (int q = 0; q < U; ++q) {           // the places first (loads and atomics share one counter), then the atomics
                    const int o = o0 + q * NTH, oc = o < items ? o : 0;
                    const int il = oc / (D * ND), j = oc % ND;
                    ps[q] = tab[(i0 + il) * ND + j];
                }
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const int o = o0 + q * NTH;
                    if (o >= items) continue;
                    const int il = o / (D * ND), rem = o - il * (D * ND), pc = rem / ND, j = rem - pc * ND, i = i0 + il;
                    if (!sRowOk[i] || !sColOk[j]) continue;
                    const size_t base = (size_t)sRowP[i] * (D * D) + (size_t)pc * sRowL[i] * D + (size_t)ps[q] * D;
                    const double *src = sE + (il * D + pc) * RL + j * D;
#pragma unroll
                    for (int c = 0; c < D; ++c) atomicAdd(&vals[base + c], src[c]);
                }
            }
        }
        return;
    }
    int cn[NTC]; bool cok[NTC];
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
        const int j = (tc0 + t) * 16 + r16;
        cok[t] = j < ND;
        cn[t] = mrow[cok[t] ? j : 0] + OFF * lrel;
    }
    if (clg) {
        int cl[NTC];
#pragma unroll
        for (int t = 0; t < NTC; ++t) cl[t] = clg[cn[t]];
#pragma unroll
        for (int t = 0; t < NTC; ++t) cok[t] = cok[t] && cl[t] >= 0;
    }
    // every row takes fire-and-forget atomics (storing the rows one cell owns alone -- the cell-interior nodes, 22 % of the entries
    // of Q4 -- and zeroing only the shared ones was built and measured 1 % slower: profiles/r4m_c3_single_rows.txt, r4n_c3_single_rows.txt).
    // ALL the index reads of the lane's 4 NT entries (row node -> lgmap -> row start, entry positions) are requested before the
    // first atomic: on gfx9 atomics and loads share one counter, so a position read issued between two atomics waits for every
    // atomic before it -- 4 NT serialised round trips per lane, during which the wavefront feeds no MFMA either.
    int rn[4]; bool rok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int i = itile * 16 + kk + 4 * g;
        rok[g] = i < ND;
        rn[g] = mrow[rok[g] ? i : 0] + OFF * lrel;
    }
    if (rlg) {
        int rl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) rl[g] = rlg[rn[g]];
#pragma unroll
        for (int g = 0; g < 4; ++g) rok[g] = rok[g] && rl[g] >= 0;
    }
    fd_nnz_t rp[4]; int rlen[4];
    unsigned short pos[4][NTC];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        rp[g] = rowptr[rn[g]];                             // (unconditional reads of clamped indices: no branch, no wait in between)
        rlen[g] = D > 1 ? (int)(rowptr[rn[g] + 1] - rp[g]) : 0;
        const int i = itile * 16 + kk + 4 * g;
#pragma unroll
        for (int t = 0; t < NTC; ++t) pos[g][t] = tab[(i < ND && (tc0 + t) * 16 + r16 < ND) ? i * ND + (tc0 + t) * 16 + r16 : 0];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (!rok[g]) continue;
        if constexpr (D == 1) {
            const size_t r0 = (size_t)rp[g];
#pragma unroll
            for (int t = 0; t < NTC; ++t)
                if (cok[t]) atomicAdd(&vals[r0 + pos[g][t]], acc[0][t][g]);
        } else {
#pragma unroll
            for (int a = 0; a < NP; ++a) {
                const int p = FUSED ? a / D : cp, r = FUSED ? a % D : cr;
                const size_t r0 = (size_t)rp[g] * (D * D) + (size_t)p * rlen[g] * D + r;
#pragma unroll
                for (int t = 0; t < NTC; ++t)
                    if (cok[t]) atomicAdd(&vals[r0 + (unsigned)pos[g][t] * D], acc[a][t][g]);
            }
        }
    }
    };
    if constexpr (CR == 0) {
        chunk(std::integral_constant<int, CB>{}, csplit * CB);
    } else {
        if (csplit < CR) chunk(std::integral_constant<int, CB + 1>{}, csplit * (CB + 1));
        else chunk(std::integral_constant<int, CB>{}, CR * (CB + 1) + (csplit - CR) * CB);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Q_k operator action y_e += sum_q Phi_q^T W_q (Phi_q u_e), sum-factorised, ONE LANE PER LINE of the index cube: a
// contraction along one axis reads the entries of a line once and produces the line's outputs in registers, so a cell
// costs M^2 lanes per pass (M = max(K1, Q1)) and ~100 LDS accesses per lane in total for Q4 -- against 135 reads per lane
// and 125 lanes per cell when every lane owns one output (LDS-bound: the first version of this template ran at the
// ds_read_b64 rate).  A 128-lane workgroup carries CPW = 128 / M^2 consecutive cells of the (column, layer) space (Q4: 5);
// lane t -> cell slot t / M^2, line t % M^2 = (p, r).  Index cubes are stored with extent M on every axis.
//   pass 1  (lines along i1, u straight from the lane's own global loads): V = L_1 u, D1 = DL_1 u        -> A  (q1, i2, i3)
//   pass 2  (lines along i2):  VV = L_2 V,  D1V = L_2 D1,  D2 = DL_2 V                                    -> B  (q1, q2, i3)
//   pass 3+4 (lines along i3, all in registers): values and reference gradient at the line's Gauss points, the
//            point weights (geometry of a trilinear hexahedron is affine along a line: 6 vectors per line, one
//            interpolation per point), F = W g, then the transposed contraction q3 -> i3                  -> B  (q1, q2, i3), in place
//   pass 5  (q2 -> i2) -> A, pass 6 (q1 -> i1) and one fp64 atomic per DoF (extruded addressing map + k*layer, builder.py:94-124).
// ------------------------------------------------------------------------------------------------------------------
constexpr int tp_action_cells(int k1, int q1) { return 128 / ((k1 > q1 ? k1 : q1) * (k1 > q1 ? k1 : q1)); }

// GRAD / D as in the matrix template: with GRAD every coefficient rides the passes like u itself (value and the derivative-table
// contractions: A two cubes, B three) and the callback gets DC[3*m + a]; a vector-valued unknown (Dat dim D) takes D times the cubes
// of u and a 4D x 4D point weight, F[p][l] = sum_{r,k} W[(p*4 + l)*4D + r*4 + k] g[r][k].
template <int K1, int Q1, int NC, int N1, bool GRAD, int D, class WF>
__device__ __forceinline__ void hex_qk_action(int start, int end, const int *__restrict__ layers, double *__restrict__ y,
                                              const double *__restrict__ coords, const double *__restrict__ u,
                                              const double *const (&cf)[NC > 0 ? NC : 1], const double *const (&c1)[N1 > 0 ? N1 : 1],
                                              const int *__restrict__ map_qk, const int *__restrict__ map_q1,
                                              const double *__restrict__ tables, WF weights) {
    constexpr int M = K1 > Q1 ? K1 : Q1, M2 = M * M, M3 = M2 * M, CPW = tp_action_cells(K1, Q1), ND = K1 * K1 * K1, NTAB = Q1 * K1;
    constexpr int OFF = K1 - 1, CA = GRAD ? 2 : 1, CB = GRAD ? 3 : 1, NCT = NC + N1 > 0 ? NC + N1 : 1;
    static_assert(CPW >= 1, "one cell needs at most 128 lines");
    static_assert(D >= 1 && D <= 3, "vector-valued Q_k spaces of up to three components");
    // A: two index cubes per cell and component (passes 1 and 5 write two), B: three (pass 2 writes three; passes 3 + 4 put their three
    // results back IN PLACE -- a lane reads and writes only its own line) -- 5 cubes of 1000 B per cell for scalar Q4, six workgroups
    // per CU.  The NC coefficient arguments ride through passes 1 and 2 in further slots (A: 2D.., B: 3D..) and are evaluated at the
    // line's points.
    __shared__ double sA[CPW][2 * D + NC * CA][M3], sB[CPW][3 * D + NC * CB][M3], sX[CPW][24];
    __shared__ double sV1[CPW][N1 > 0 ? N1 : 1][8];       // vertex values of the coefficient arguments on the Q1 map
    const int t = threadIdx.x;
    const int nl = layers[1] - 1 - layers[0];
    const int ncell = (end - start) * nl;
    const int first = (int)blockIdx.x * CPW;
    {
        const int ks = t / 24, e = t - 24 * ks;         // 24 lanes per cell fetch its 8 vertices, five cells at a time
        for (int kc = ks; ks < 128 / 24 && kc < CPW; kc += 128 / 24) {
            const int cs = first + kc;
            if (cs < ncell) {
                const int v = e / 3, cc = e - 3 * v;
                sX[kc][e] = coords[(size_t)(map_q1[(size_t)(start + cs / nl) * 8 + v] + cs % nl) * 3 + cc];
            }
        }
    }
    if constexpr (N1 > 0)
        for (int o = t; o < CPW * N1 * 8; o += 128) {
            const int kc = o / (N1 * 8), m = (o / 8) % N1, v = o & 7, cs = first + kc;
            if (cs < ncell) sV1[kc][m][v] = c1[m][map_q1[(size_t)(start + cs / nl) * 8 + v] + cs % nl];
        }
    const int k = t / M2, l = t - M2 * k, p = l / M, r = l - M * p;
    const bool in = k < CPW && first + k < ncell;
    const int ka = in ? k : 0;
    double (*A)[M3] = sA[ka], (*B)[M3] = sB[ka];
    // The 1-D tables are wavefront-uniform and live in scalar registers -- of which a wavefront has ~100: both tables of Q4 in full
    // (2 x 25 doubles) overflow them, and the compiler then parks the excess in the lanes of a vector register and fetches every
    // operand back with v_readlane (768 of the 2400 instructions of the Q4 kernel, a third of its issue slots; Q5: half).  Nodes
    // and points are symmetric about 1/2, so L[NTAB-1-m] = L[m] and DL[NTAB-1-m] = -DL[m]: only the first halves are kept
    // (Parloop._tp_tables checks the symmetry), the sign rides on the FMA's source modifier.  Measured at n = 64 (profiles/
    // r4n_action_variants.txt): on its own the shorter kernel is 10 % SLOWER (0.588 against 0.533 ms -- the kernel waits on its two
    // dependent rounds of global loads and its barriers, not on issue slots, and this form happens to need 172 registers: two
    // wavefronts per SIMD); compiled for three wavefronts per SIMD (codegen: __launch_bounds__(128, 3)) it is the fastest, 0.52 ms.
    constexpr int NH = (NTAB + 1) / 2;
    double Lh[NH], DLh[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) { Lh[i] = tables[i]; DLh[i] = tables[NTAB + i]; }
    auto L = [&](int m) { return m < NH ? Lh[m] : Lh[NTAB - 1 - m]; };
    auto DL = [&](int m) { return m < NH ? DLh[m] : -DLh[NTAB - 1 - m]; };
    // pass 1: line (i2, i3) = (p, r) of the K1^3 coefficient cube
    const bool on1 = in && p < K1 && r < K1;
    int node[K1];
    if (on1) {
        const int cell = first + k;
        const int *mrow = map_qk + (size_t)(start + cell / nl) * ND;
        const int lrel = OFF * (cell % nl);
#pragma unroll
        for (int i = 0; i < K1; ++i) node[i] = mrow[(i * K1 + p) * K1 + r] + lrel;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double uv[K1];
#pragma unroll
            for (int i = 0; i < K1; ++i) uv[i] = u[(size_t)node[i] * D + c];
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = 0; i < K1; ++i) { s0 += L(q * K1 + i) * uv[i]; s1 += DL(q * K1 + i) * uv[i]; }
                A[2 * c][q * M2 + l] = s0; A[2 * c + 1][q * M2 + l] = s1;
            }
        }
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            double cv[K1];
#pragma unroll
            for (int i = 0; i < K1; ++i) cv[i] = cf[m][node[i]];
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = 0; i < K1; ++i) { s0 += L(q * K1 + i) * cv[i]; if (GRAD) s1 += DL(q * K1 + i) * cv[i]; }
                A[2 * D + m * CA][q * M2 + l] = s0;
                if (GRAD) A[2 * D + m * CA + CA - 1][q * M2 + l] = s1;
            }
        }
    }
    __syncthreads();
    if (in && p < Q1 && r < K1) {                       // pass 2: (p, r) = (q1, i3), contract i2
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double v[K1], d[K1];
#pragma unroll
            for (int i = 0; i < K1; ++i) { v[i] = A[2 * c][p * M2 + i * M + r]; d[i] = A[2 * c + 1][p * M2 + i * M + r]; }
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int i = 0; i < K1; ++i) { s0 += L(q * K1 + i) * v[i]; s1 += L(q * K1 + i) * d[i]; s2 += DL(q * K1 + i) * v[i]; }
                const int o = (p * M + q) * M + r;
                B[3 * c][o] = s0; B[3 * c + 1][o] = s1; B[3 * c + 2][o] = s2;
            }
        }
#pragma unroll
        for (int m = 0; m < NC; ++m) {
            double cv[K1], cd[K1];
#pragma unroll
            for (int i = 0; i < K1; ++i) { cv[i] = A[2 * D + m * CA][p * M2 + i * M + r]; if (GRAD) cd[i] = A[2 * D + m * CA + CA - 1][p * M2 + i * M + r]; }
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int i = 0; i < K1; ++i) {
                    s0 += L(q * K1 + i) * cv[i];
                    if (GRAD) { s1 += L(q * K1 + i) * cd[i]; s2 += DL(q * K1 + i) * cv[i]; }
                }
                const int o = (p * M + q) * M + r;
                B[3 * D + m * CB][o] = s0;
                if (GRAD) { B[3 * D + m * CB + (CB > 1 ? 1 : 0)][o] = s1; B[3 * D + m * CB + CB - 1][o] = s2; }
            }
        }
    }
    __syncthreads();
    if (in && p < Q1 && r < Q1) {                       // passes 3 + 4: (p, r) = (q1, q2); the line is index l*M + i3
        double vv[D][K1], d1[D][K1], d2[D][K1];
#pragma unroll
        for (int c = 0; c < D; ++c)
#pragma unroll
            for (int i = 0; i < K1; ++i) { vv[c][i] = B[3 * c][l * M + i]; d1[c][i] = B[3 * c + 1][l * M + i]; d2[c][i] = B[3 * c + 2][l * M + i]; }
        // geometry along the line: with (t0, t1) fixed, dx/dt0, dx/dt1 and x are affine in t2 and dx/dt2 is constant
        const double t0 = tables[2 * NTAB + p], t1 = tables[2 * NTAB + r], w01 = tables[2 * NTAB + Q1 + p] * tables[2 * NTAB + Q1 + r];
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
        // Q1 coefficients: bilinear in (t0, t1) on the bottom / top face, affine along the line -- and so are their t0 / t1 derivatives
        double P1[N1 > 0 ? N1 : 1][2], P1a[N1 > 0 ? N1 : 1][2], P1b[N1 > 0 ? N1 : 1][2];
        if constexpr (N1 > 0) {
#pragma unroll
            for (int m = 0; m < N1; ++m)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const double *V = sV1[ka][m];
                    P1[m][f] = (1.0 - t0) * ((1.0 - t1) * V[0 + f] + t1 * V[2 + f]) + t0 * ((1.0 - t1) * V[4 + f] + t1 * V[6 + f]);
                    if (GRAD) {
                        P1a[m][f] = (1.0 - t1) * (V[4 + f] - V[0 + f]) + t1 * (V[6 + f] - V[2 + f]);
                        P1b[m][f] = (1.0 - t0) * (V[2 + f] - V[0 + f]) + t0 * (V[6 + f] - V[4 + f]);
                    }
                }
        }
        double p0[D][K1], p1[D][K1], sv[D][K1];
#pragma unroll
        for (int c = 0; c < D; ++c)
#pragma unroll
            for (int i = 0; i < K1; ++i) { p0[c][i] = 0.0; p1[c][i] = 0.0; sv[c][i] = 0.0; }
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            double g[D][4];                             // d1 u, d2 u, d3 u, u (per component) at the Gauss point (p, r, q)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                g[c][0] = g[c][1] = g[c][2] = g[c][3] = 0.0;
#pragma unroll
                for (int i = 0; i < K1; ++i) {
                    g[c][0] += L(q * K1 + i) * d1[c][i]; g[c][1] += L(q * K1 + i) * d2[c][i];
                    g[c][2] += DL(q * K1 + i) * vv[c][i]; g[c][3] += L(q * K1 + i) * vv[c][i];
                }
            }
            const double t2 = tables[2 * NTAB + q];
            double J[3][3], X[3], W[16 * D * D], C[NCT], DC[3 * NCT];
#pragma unroll
            for (int m = 0; m < NC; ++m) {
                double cq = 0.0, c0 = 0.0, c1v = 0.0, c2 = 0.0;
#pragma unroll
                for (int i = 0; i < K1; ++i) {
                    const double bv = B[3 * D + m * CB][l * M + i];
                    cq += L(q * K1 + i) * bv;
                    if (GRAD) {
                        c0 += L(q * K1 + i) * B[3 * D + m * CB + (CB > 1 ? 1 : 0)][l * M + i];
                        c1v += L(q * K1 + i) * B[3 * D + m * CB + CB - 1][l * M + i];
                        c2 += DL(q * K1 + i) * bv;
                    }
                }
                C[m] = cq;
                if (GRAD) { DC[3 * m] = c0; DC[3 * m + 1] = c1v; DC[3 * m + 2] = c2; }
            }
            if constexpr (N1 > 0) {
#pragma unroll
                for (int m = 0; m < N1; ++m) {
                    C[NC + m] = P1[m][0] + t2 * (P1[m][1] - P1[m][0]);
                    if (GRAD) {
                        DC[3 * (NC + m)] = P1a[m][0] + t2 * (P1a[m][1] - P1a[m][0]);
                        DC[3 * (NC + m) + 1] = P1b[m][0] + t2 * (P1b[m][1] - P1b[m][0]);
                        DC[3 * (NC + m) + 2] = P1[m][1] - P1[m][0];
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                J[c][0] = G0[c][0] + t2 * (G0[c][1] - G0[c][0]);
                J[c][1] = G1[c][0] + t2 * (G1[c][1] - G1[c][0]);
                J[c][2] = P[c][1] - P[c][0];
                X[c] = P[c][0] + t2 * (P[c][1] - P[c][0]);
            }
            weights(J, X, w01 * tables[2 * NTAB + Q1 + q], C, DC, W);
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double F[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    double f = 0.0;
#pragma unroll
                    for (int cr = 0; cr < D; ++cr)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) f += W[((c * 4 + m) * 4 * D) + cr * 4 + kk] * g[cr][kk];
                    F[m] = f;
                }
#pragma unroll
                for (int i = 0; i < K1; ++i) {
                    p0[c][i] += L(q * K1 + i) * F[0]; p1[c][i] += L(q * K1 + i) * F[1]; sv[c][i] += DL(q * K1 + i) * F[2] + L(q * K1 + i) * F[3];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < D; ++c)
#pragma unroll
            for (int i = 0; i < K1; ++i) { B[3 * c][l * M + i] = p0[c][i]; B[3 * c + 1][l * M + i] = p1[c][i]; B[3 * c + 2][l * M + i] = sv[c][i]; }
    }
    __syncthreads();
    if (in && p < Q1 && r < K1) {                       // pass 5: (p, r) = (q1, i3), q2 -> i2
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double a0[Q1], a1[Q1], a2[Q1];
#pragma unroll
            for (int q = 0; q < Q1; ++q) { const int o = (p * M + q) * M + r; a0[q] = B[3 * c][o]; a1[q] = B[3 * c + 1][o]; a2[q] = B[3 * c + 2][o]; }
#pragma unroll
            for (int i = 0; i < K1; ++i) {
                double r0 = 0.0, r1 = 0.0;
#pragma unroll
                for (int q = 0; q < Q1; ++q) { r0 += L(q * K1 + i) * a0[q]; r1 += DL(q * K1 + i) * a1[q] + L(q * K1 + i) * a2[q]; }
                A[2 * c][p * M2 + i * M + r] = r0; A[2 * c + 1][p * M2 + i * M + r] = r1;
            }
        }
    }
    __syncthreads();
    if (on1) {                                          // pass 6: line l = (i2, i3), q1 -> i1
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double r0[Q1], r1[Q1];
#pragma unroll
            for (int q = 0; q < Q1; ++q) { r0[q] = A[2 * c][q * M2 + l]; r1[q] = A[2 * c + 1][q * M2 + l]; }
#pragma unroll
            for (int i = 0; i < K1; ++i) {
                double yv = 0.0;
#pragma unroll
                for (int q = 0; q < Q1; ++q) yv += DL(q * K1 + i) * r0[q] + L(q * K1 + i) * r1[q];
                atomicAdd(&y[(size_t)node[i] * D + c], yv);
            }
        }
    }
}

}  // namespace fdt
