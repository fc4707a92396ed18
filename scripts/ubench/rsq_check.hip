// Accuracy of v_rsq_f64 and of the two refinements built on it (lane_group.h: fast_rsqrt, fast_rsqrt3) against 1/sqrt(x) in long double:
//   hipcc --offload-arch=gfx950 -O3 -I contactimplicitmpc/jl_amd/csrc scripts/ubench/rsq_check.hip -o /tmp/rsq_check && /tmp/rsq_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "lane_group.h"
namespace cimpc {
// the candidate: one third-order step on the hardware seed (five dependent instructions; the two Newton steps of fast_rsqrt take seven)
__device__ __forceinline__ double fast_rsqrt3(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(y * e, p, y);
}
}  // namespace cimpc
__global__ void k(const double* x, double* seed, double* two, double* three, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    seed[i] = __builtin_amdgcn_rsq(x[i]);
    two[i] = cimpc::fast_rsqrt(x[i]);
    three[i] = cimpc::fast_rsqrt3(x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> m(1.0, 4.0), e(-60.0, 60.0);
    for (int i = 0; i < n; ++i) x[i] = m(g) * std::exp2(std::floor(e(g)));
    double *dx, *da, *db, *dc;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8); (void)hipMalloc(&dc, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dc, n);
    (void)hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
    long double ea = 0, eb = 0, ec = 0; int diff = 0;
    for (int i = 0; i < n; ++i) {
        const long double r = 1.0L / sqrtl((long double)x[i]);
        ea = fmaxl(ea, fabsl(((long double)a[i] - r) / r)); eb = fmaxl(eb, fabsl(((long double)b[i] - r) / r)); ec = fmaxl(ec, fabsl(((long double)c[i] - r) / r));
        diff += b[i] != c[i];
    }
    printf("max relative error over %d samples: seed %.3Le (2^%.1Lf)  two Newton steps %.3Le (%.2Lf ulp)  one third-order step %.3Le (%.2Lf ulp); results differ on %d samples\n",
           n, ea, log2l(ea), eb, eb / 1.1102230246251565e-16L, ec, ec / 1.1102230246251565e-16L, diff);
    return 0;
}
