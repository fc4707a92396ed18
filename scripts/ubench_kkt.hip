// Micro-benchmark of the KKT kernel with per-phase cycle counters (not part of the product).
#define CIMPC_KKT_PROF 1
#include "newton_kernels.hip"
#include <cstdio>
#include <vector>
#include <random>
using namespace cimpc;
int main() {
    const int nq = 11, nu = 8, H = 40;
    for (int B : {1, 512}) {
        NewtonDev S{};
        S.dm = cimpc_dims{nq, nu, 2, 4, 8, 0, 60, H, B};
        S.nd = nq; S.nr = nq + nu; S.nth = 34; S.nths = 30; S.N = H * 30; S.kappa = 2e-4;
        std::mt19937 g(1); std::normal_distribution<double> nd(0, 0.3);
        std::vector<double> dz((size_t)B * CS * H * 30 * 11), r((size_t)B * S.N), Qi((size_t)H * 121, 0.0), Ri((size_t)H * 64, 0.0);
        for (auto& v : dz) v = nd(g);
        for (auto& v : r) v = nd(g);
        for (int i = 0; i < H; ++i) { for (int k = 0; k < 11; ++k) Qi[i * 121 + k * 12] = 2.0; for (int k = 0; k < 8; ++k) Ri[i * 64 + k * 9] = 3.0; }
        double *d_dz, *d_r, *d_delta, *d_Qi, *d_Ri, *d_ws; long long* d_stats;
        hipMalloc(&d_dz, dz.size() * 8); hipMalloc(&d_r, r.size() * 8); hipMalloc(&d_delta, r.size() * 8);
        hipMalloc(&d_Qi, Qi.size() * 8); hipMalloc(&d_Ri, Ri.size() * 8); hipMalloc(&d_ws, (size_t)B * H * (3 * 121 + 11) * 8);
        hipMalloc(&d_stats, 32 * 8); hipMemset(d_stats, 0, 32 * 8);
        hipMemcpy(d_dz, dz.data(), dz.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_r, r.data(), r.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(d_Qi, Qi.data(), Qi.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_Ri, Ri.data(), Ri.size() * 8, hipMemcpyHostToDevice);
        S.dz = d_dz; S.Qinv = d_Qi; S.Rinv = d_Ri; S.kkt_ws = d_ws; S.stats = d_stats; S.b0 = 0; S.nb_launch = B;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        launch_kkt_raw(S, d_r, 10.0, d_delta, 0); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int rep = 0; rep < 10; ++rep) launch_kkt_raw(S, d_r, 10.0, d_delta, 0);
        hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        long long pt[32]; hipMemcpy(pt, d_stats, sizeof(pt), hipMemcpyDeviceToHost);
        std::vector<double> delta(r.size()); hipMemcpy(delta.data(), d_delta, r.size() * 8, hipMemcpyDeviceToHost);
        printf("B=%d: %.1f us per launch; delta[0..2]=%g %g %g\n  phase cycles (total over %d steps):", B, 1e3 * ms / 10, delta[0], delta[1], delta[2], H);
        long long tot = 0; for (int j = 0; j < 12; ++j) tot += pt[8 + j];
        for (int j = 0; j < 12; ++j) printf(" p%d=%lld", j, pt[8 + j]);
        printf("  total=%lld\n", tot);
    }
    return 0;
}
