// Packed per-knot linearization table: the HBM / LDS layout of the constant blocks the
// reference keeps in RLin / RZLin / RthLin / Schur objects
// (/root/reference/src/controller/linearized_solver.jl:15-65, 188-222, 311-323;
//  src/solver/schur.jl:13-26).
//
// One table = LinLayout::size doubles, contiguous, 16-byte aligned.  Every matrix is
// stored "lane strided": element (row i, col k) of an operator applied as
//   y_i = sum_k A[i,k] x_k   (lane i owns row i, x_k broadcast from lane k)
// sits at  base + k*G + i , so that the G lanes of a group read G consecutive doubles
// (coalesced from HBM when the table is staged, conflict-free from LDS afterwards).
// The Schur matrix W = Ry1 - C A^-1 B is stored transposed-strided (row i at i*ldw + j)
// because the QR keeps one COLUMN j per lane; adj = 1: ldw = G + 1, so that the lanes read a column (i*ldw + j over i) as free
// of bank conflicts as a row - the adjoint sensitivity pass factorizes W^T from the same block.
#pragma once

namespace cimpc {

struct LinLayout {
    int nx, ny, nth, G, nths, adj, ldw, gst;
    // offsets in doubles
    int oW, oCAi, oAi, oDy1, oDx, oRx, oRy1, oRthDyn, oRthRst, oGs, oK0, oAiB, oVec, oTh0, size;
    // 64-lane models (round 6: ny > 32, one problem per wavefront): at lane stride 64 the whole table (208 KB for centroidal_quadruped_wall)
    // does not fit a CU's LDS.  Only the operators of the interior-point ITERATION are staged - [oW, oRthDyn) and the vectors, `hot`
    // doubles, the vectors at oVecL - ; the blocks a solve touches once (r_theta at the pull; Gs, K0, A^-1 B in the sensitivity
    // pass; theta0) are read from the table in global memory (L2).  Other models: hot = size, oVecL = oVec.
    int hot, oVecL;
    // oGs (compiled lane-group models; nths = 0: absent): the right-hand sides of the sensitivity pass as the QR sees them,
    //   Gs[:, c] = CAi * rthdyn[:, c] - rthrst[:, c],  c = 0 .. nths-1  (schur_solve!, schur.jl:93-110, on column c of r_theta,
    //   linearized_solver.jl:451-479) - a constant of the knot that every converged solve used to recompute for each of its
    //   nths columns; formed by cimpc_set_linearization with the kernel's own multiply-add chain (bit-identical columns).
    //   gst = 1 (adj = 1, 16-lane groups): the block is stored by ROW, Gs[k, c] at k*nths + c - the adjoint pass of those models
    //   keeps one COLUMN c per lane (ip_kernel_impl.h: sensitivities) and reads row k with consecutive addresses
    // oK0, oAiB (adj = 1: models whose sensitivity pass runs in the adjoint form, ip_kernel_impl.h: sensitivities): the constants
    //   of dx/dtheta = A^-1 rthdyn + (A^-1 B) M^-1 Gs  (the x rows of schur_solve! applied to every column at once),
    //   K0[i, c] = (A^-1 rthdyn[:, c])_i at c*nx + i  and  AiB[i, k] = (A^-1 B)[i, k] at i*ny + k  (packed, no lane padding:
    //   lanes read them contiguously either way; every KB counts - a sweep workgroup of the quadruped shares its CU's 160 KB with a
    //   second one or with a workgroup of the KKT kernel, the centroidal table sits next to sixteen problems)
    // oVec holds 8 lane-strided vectors:
    enum { V_RY2 = 0, V_RY1D, V_CAIBD, V_RDYN0, V_RRST0, V_X0, V_Y10, V_Y20, V_COUNT };

#ifndef CIMPC_ADJ_COLS
#define CIMPC_ADJ_COLS 1      // (0: diagnostic builds - the row-per-lane product on 16-lane groups too)
#endif
    __host__ __device__ constexpr LinLayout(int nx_, int ny_, int nth_, int G_, int nths_ = 0, int adj_ = 0)
        : nx(nx_), ny(ny_), nth(nth_), G(G_), nths(nths_), adj(adj_), ldw(adj_ ? G_ + 1 : G_), gst(CIMPC_ADJ_COLS && adj_ && G_ == 16 ? 1 : 0),
          oW(0),
          oCAi((oW + ny_ * (adj_ ? G_ + 1 : G_) + 1) & ~1),
          oAi(oCAi + nx_ * G_),
          oDy1(oAi + nx_ * G_),
          oDx(oDy1 + ny_ * G_),
          oRx(oDx + nx_ * G_),
          oRy1(oRx + nx_ * G_),
          oRthDyn(oRy1 + ny_ * G_),
          oRthRst(oRthDyn + nth_ * G_),
          oGs(oRthRst + nth_ * G_),
          oK0((oGs + ((CIMPC_ADJ_COLS && adj_ && G_ == 16) ? ny_ * nths_ : nths_ * G_) + 1) & ~1),
          oAiB(oK0 + (adj_ ? nths_ * nx_ : 0)),
          oVec((oAiB + (adj_ ? nx_ * ny_ : 0) + 1) & ~1),
          oTh0(oVec + V_COUNT * G_),
          size(((oTh0 + nth_) + 1) & ~1),
          hot(G_ == 64 ? oRthDyn + V_COUNT * G_ : size),
          oVecL(G_ == 64 ? oRthDyn : oVec) {}
};

}  // namespace cimpc
