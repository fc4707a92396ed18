// Micro-benchmark of the interior-point kernel's phases (not part of the product):
// cycles per phase for one wave (latency) and wall time with the machine full (throughput).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I contactimplicitmpc/jl_amd/csrc scripts/ubench_ip.hip -o scripts/build/ubench_ip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ip_kernel_impl.h"

using namespace cimpc;
using M = Model<11, 8, 2, 4, 8, 0>;

template <int PHASE>
__global__ __launch_bounds__(256) void ub(double* out, long long* cyc, int reps) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    for (int k = tid; k < L.size; k += blockDim.x) {
        unsigned h = (unsigned)k * 2654435761u;
        smem[k] = 0.02 * ((double)((h >> 8) & 1023) / 1023.0 - 0.5);
    }
    __syncthreads();
    for (int k = tid; k < M::G; k += blockDim.x) {
        smem[L.oVec + LinLayout::V_RY1D * M::G + k] = 3.0;
        smem[L.oVec + LinLayout::V_RY2 * M::G + k] = 1.0;
        smem[L.oVec + LinLayout::V_Y10 * M::G + k] = 0.5;
        smem[L.oVec + LinLayout::V_Y20 * M::G + k] = 0.5;
    }
    for (int k = tid; k < M::NX; k += blockDim.x) smem[L.oAi + k * M::G + k] = 1.0;
    __syncthreads();
    const int grp = tid / M::G, l = tid % M::G;
    double* Rst = smem + L.size + (size_t)grp * M::LDS_GROUP;
    IpSolver<M> S;
    S.bind(smem, Rst, l);
    S.x = 0.1 * l; S.y1 = 1.0; S.y2 = 1.0;
    S.residual(0.0);
    S.factorize(0.0);
    S.linear_solve();
    double sink = 0.0;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if constexpr (PHASE == 0) { S.residual(1e-4 * r); sink += S.rdyn + S.rrst; S.x += 1e-9 * sink; }
        if constexpr (PHASE == 1) { S.factorize(1e-6 * r); sink += S.Qc[3] + S.Rr[5]; S.y1 += 1e-12 * sink; }
        if constexpr (PHASE == 2) { S.linear_solve(); sink += S.Dx_ + S.Dy2_; S.rdyn += 1e-9 * sink; }
        if constexpr (PHASE == 3) { double t = S.qr_solve(S.rrst + 1e-9 * sink); sink += t; }
        if constexpr (PHASE == 4) { double a = S.step_length(0.99); double m = IpSolver<M>::LG::all_sum(S.y1 * S.y2 + 1e-9 * sink); sink += a + m + S.r_violation(); S.Dy1_ += 1e-9 * sink; }
        if constexpr (PHASE == 5) { cimpc_ip_opts o; o.r_tol = 1e-8; o.kappa_tol = 2e-4; o.undercut = 5; o.gamma_reg = 0.1; o.kappa_reg = 1e-3; o.eps_min = 0.05; o.ls_scale = 0.5; o.max_iter = 8; o.max_ls = 3; o.stall_alpha = 1e-13;
                                    int it = 0; double rg = 0, rv, kv; S.x = 0.1 * l; S.y1 = 1.0; S.y2 = 1.0; S.residual(0.0); rv = S.r_violation(); kv = S.k_violation(); S.solve(o, it, rg, rv, kv, 1000); sink += S.x + it; }
    }
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[(size_t)blockIdx.x * blockDim.x + tid] = sink;
}

template <int PHASE>
void run(const char* name, int reps) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G);
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 4096 * 256);
    hipMalloc(&cyc, sizeof(long long) * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { int wgs, thr; } cfgs[] = {{1, 64}, {1, 256}, {256, 256}, {512, 256}, {2048, 256}};
    printf("%-14s", name);
    for (auto c : cfgs) {
        const size_t lds = (size_t)(L.size + (c.thr / M::G) * M::LDS_GROUP) * sizeof(double);
        hipLaunchKernelGGL(ub<PHASE>, dim3(c.wgs), dim3(c.thr), lds, 0, out, cyc, 2);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(ub<PHASE>, dim3(c.wgs), dim3(c.thr), lds, 0, out, cyc, reps);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        long long cy; hipMemcpy(&cy, cyc, sizeof(cy), hipMemcpyDeviceToHost);
        const double probs = (double)c.wgs * (c.thr / M::G);
        printf(" | %4dx%-3d %7.0f cyc/rep %8.1f ns/prob-rep", c.wgs, c.thr, (double)cy / reps, 1e6 * ms / reps / probs);
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("residual", 200);
    run<1>("factorize", 100);
    run<2>("linear_solve", 200);
    run<3>("qr_solve", 200);
    run<4>("reductions", 200);
    run<5>("ip_solve(<=8)", 20);
    return 0;
}
