// Issue interval and dependent latency of the fp64 instructions the lane-group solver is made of, on ONE wavefront (gfx950):
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_lat.hip -o /tmp/valu_lat && /tmp/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP2X(x) x x
__global__ void k(long long* out, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5, c = 1e-9;
    long long t[20];
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
    // 12: 64 x (dependent v_fma_f64 + two scalar instructions)
    {
        unsigned long long m = 0x0001000100010001ull, r;
        asm volatile(REP64("v_fma_f64 %0, %0, %2, %3\n s_lshl_b64 %1, %4, 3\n s_and_b64 %1, %1, exec\n") : "+v"(a0), "=&s"(r) : "v"(b), "v"(c), "s"(m) : "scc");
        a6 += (double)r;
    }
    T()
    // 13: 64 x (dependent v_fma_f64 + exec-masked v_mov_b64: s_and_b64 exec / v_mov / restore)
    {
        unsigned long long m = 0x0001000100010001ull, sv;
        asm volatile("s_mov_b64 %1, exec\n" REP64("v_fma_f64 %0, %0, %3, %4\n s_and_b64 exec, %1, %5\n v_mov_b64 %2, %0\n s_mov_b64 exec, %1\n") : "+v"(a0), "=&s"(sv), "+v"(a7) : "v"(b), "v"(c), "s"(m) : "scc");
    }
    T()
    // 14: 64 x (dependent v_fma_f64 + two v_cndmask_b32: the 64-bit select)
    {
        int x = (int)seed, y = 3;
        asm volatile(REP64("v_fma_f64 %0, %0, %3, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n") : "+v"(a0), "+v"(x), "+v"(y) : "v"(b), "v"(c));
        a5 += x + y;
    }
    T()
    // 15: the back-substitution step, 32 x (v_mul_f64 ; s_nop 1 ; dependent v_fmac_f64_dpp)
    asm volatile(REP16("v_mul_f64 %1, %0, %2\n s_nop 1\n v_fmac_f64_dpp %0, %1, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mul_f64 %1, %0, %2\n s_nop 1\n v_fmac_f64_dpp %0, %1, %3 row_newbcast:6 row_mask:0xf bank_mask:0xf\n") : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
    T()
    // 16: four such chains interleaved, 8 x (4 v_mul_f64 ; 4 v_fmac_f64_dpp) x 4 = the same 32 steps per chain... 32 steps in total here (8 per chain)
    asm volatile(REP4(REP2X("v_mul_f64 %4, %0, %8\n v_mul_f64 %5, %1, %8\n v_mul_f64 %6, %2, %8\n v_mul_f64 %7, %3, %8\n v_fmac_f64_dpp %0, %4, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %5, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %6, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %7, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"))
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    T()
    if (threadIdx.x == 0) { for (int i = 0; i + 1 < n; ++i) out[i] = t[i + 1] - t[i]; out[19] = (long long)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); }
}
int main() {
    long long* d; hipMalloc(&d, 20 * sizeof(long long));
    long long h[20];
    const char* names[] = {"64 dependent v_fma_f64", "64 v_fma_f64, 8 chains", "64 dependent v_fmac_f64_dpp", "64 v_fmac_f64_dpp, 4 accumulators", "64 v_fmac_f64_dpp, 8 accumulators",
                           "16 dependent v_rsq_f64", "64 dependent v_mul_f64", "64 dependent v_add_f64", "32 x (s_nop 1 + v_mov_b64_dpp + dependent v_add_f64)", "64 independent v_fma_f64", "64 dependent v_cndmask_b32", "64 x (v_fma_f64 + 2 SALU)", "64 x (v_fma_f64 + exec-masked v_mov_b64)", "64 x (v_fma_f64 + 2 v_cndmask_b32)", "32 back-substitution steps (mul, s_nop 1, fmac_dpp), one chain", "32 back-substitution steps, four chains interleaved (8 each), no s_nop"};
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.0 + rep); hipDeviceSynchronize(); }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("%-56s %6lld clocks\n", names[i], h[i]);
    return 0;
}
