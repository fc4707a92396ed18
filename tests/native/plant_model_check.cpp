// Host harness for contactimplicitmpc/jl_amd/csrc/plant_model.h (the header is host- and device-compilable): reads
// "model nz nth kappa z... th..." from stdin, prints the residual and the dual-number Jacobian dr/dz (row-major).
#include <cstdio>
#include <vector>
#include "../../contactimplicitmpc/jl_amd/csrc/plant_model.h"
int main() {
    int model; double kappa;
    if (scanf("%d %lf", &model, &kappa) != 2) return 1;
    const cimpc::PlantModel M = model == 0 ? cimpc::plant_quadruped() : model == 1 ? cimpc::plant_flamingo() : model == 2 ? cimpc::plant_hopper_2d()
                                : model == 5 ? cimpc::plant_particle() : cimpc::plant_centroidal(model == 3);
    const int nz = M.nz(), nth = M.nth();
    std::vector<double> z(nz), th(nth), r(nz);
    for (auto& v : z) if (scanf("%lf", &v) != 1) return 1;
    for (auto& v : th) if (scanf("%lf", &v) != 1) return 1;
    cimpc::plant_residual<double>(M, z.data(), th.data(), kappa, r.data());
    printf("%d %d\n", nz, nth);
    for (double v : r) printf("%.17g ", v);
    printf("\n");
    std::vector<cimpc::Dual> zd(nz), rd(nz);
    std::vector<double> J((size_t)nz * nz);
    for (int j = 0; j < nz; ++j) {
        for (int i = 0; i < nz; ++i) zd[i] = {z[i], i == j ? 1.0 : 0.0};
        cimpc::plant_residual<cimpc::Dual>(M, zd.data(), th.data(), kappa, rd.data());
        for (int i = 0; i < nz; ++i) J[(size_t)i * nz + j] = rd[i].d;
    }
    for (double v : J) printf("%.17g ", v);
    printf("\n");
    return 0;
}
