// Issue interval and dependent latency of the fp64 instructions the lane-group solver is made of, on ONE wavefront (gfx950):
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_lat.hip -o /tmp/valu_lat && /tmp/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
__global__ void k(long long* out, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5, c = 1e-9;
    long long t[16];
    int n = 0;
#define T() t[n++] = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T()
    // 1: dependent v_fma_f64 chain (64)
    asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
    T()
    // 2: 8 independent chains (64 instructions)
    asm volatile(REP4(REP4("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n") )
                 REP4(REP4("v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n") )
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    T()
    // 3: dependent v_fmac_f64_dpp chain (row_newbcast)
    asm volatile("s_nop 1\n" REP64("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a0) : "v"(b), "v"(c));
    T()
    // 4: 4 independent accumulators, dpp (64 instructions) - the dot product's shape
    asm volatile("s_nop 1\n" REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    T()
    // 5: 8 independent dpp fmacs (the update's shape: every destination its own register)
    asm volatile("s_nop 1\n" REP4(REP4("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"))
                 REP4(REP4("v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"))
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    T()
    // 6: dependent v_rsq_f64 chain (16)
    asm volatile(REP16("v_rsq_f64 %0, %0\n") : "+v"(a0));
    T()
    // 7: dependent v_mul_f64 chain (64)
    asm volatile(REP64("v_mul_f64 %0, %0, %1\n") : "+v"(a1) : "v"(b));
    T()
    // 8: dependent v_add_f64 chain (64)
    asm volatile(REP64("v_add_f64 %0, %0, %1\n") : "+v"(a2) : "v"(c));
    T()
    // 9: dependent v_mov_b64_dpp row_newbcast (the broadcast) alternating with an add (32 pairs)
    asm volatile(REP16("s_nop 1\n v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_add_f64 %1, %0, %2\n s_nop 1\n v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_add_f64 %1, %0, %2\n") : "+v"(a3), "+v"(a4) : "v"(c));
    T()
    // 10: 64 independent v_fma_f64 writing 8 registers round robin with a DIFFERENT source (pure issue rate, no dependence at all)
    asm volatile(REP4(REP4("v_fma_f64 %0, %8, %8, %9\n v_fma_f64 %1, %8, %8, %9\n v_fma_f64 %2, %8, %8, %9\n v_fma_f64 %3, %8, %8, %9\n"))
                 REP4(REP4("v_fma_f64 %4, %8, %8, %9\n v_fma_f64 %5, %8, %8, %9\n v_fma_f64 %6, %8, %8, %9\n v_fma_f64 %7, %8, %8, %9\n"))
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    T()
    // 11: v_cndmask_b32 pairs dependent (64)
    {
        int x = (int)seed, y = 3;
        asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(x) : "v"(y));
        a5 += x;
    }
    T()
    if (threadIdx.x == 0) { for (int i = 0; i + 1 < n; ++i) out[i] = t[i + 1] - t[i]; out[15] = (long long)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); }
}
int main() {
    long long* d; hipMalloc(&d, 16 * sizeof(long long));
    long long h[16];
    const char* names[] = {"64 dependent v_fma_f64", "64 v_fma_f64, 8 chains", "64 dependent v_fmac_f64_dpp", "64 v_fmac_f64_dpp, 4 accumulators", "64 v_fmac_f64_dpp, 8 accumulators",
                           "16 dependent v_rsq_f64", "64 dependent v_mul_f64", "64 dependent v_add_f64", "32 x (s_nop 1 + v_mov_b64_dpp + dependent v_add_f64)", "64 independent v_fma_f64", "64 dependent v_cndmask_b32"};
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.0 + rep); hipDeviceSynchronize(); }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 11; ++i) printf("%-56s %6lld clocks\n", names[i], h[i]);
    return 0;
}
