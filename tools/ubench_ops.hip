// Issue cost of the instructions the Viterbi chain is made of, one wave on an otherwise idle CU (s_memtime around 512
// back-to-back instructions; "dep" = each instruction consumes the previous result, "ind" = 8 independent streams).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_ops.hip -o tools/_build/ubench_ops && tools/_build/ubench_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int K>
__global__ void bench(long long *out, double *sink, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double b = seed * 0.5;
    int i0 = 1, i1 = 2;
    __builtin_amdgcn_s_waitcnt(0);
    long long t0 = clock64();
    if (K == 0) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(b));) } }
    if (K == 1) { for (int r = 0; r < 8; ++r) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) } }
    if (K == 2) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_max_f64 %0, %0, %1" : "+v"(a0) : "v"(b));) } }
    if (K == 3) { for (int r = 0; r < 8; ++r) { REP8(asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) } }
    if (K == 4) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a0), "v"(b) : "vcc");) } }
    if (K == 5) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(i0) : "v"(a0), "v"(b), "v"(i1) : "vcc");) } }
    if (K == 6) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_add_f64 %1, %1, %3\n v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %4, vcc\n v_max_f64 %2, %2, %1" : "+v"(i0), "+v"(a1), "+v"(a0) : "v"(b), "v"(i1) : "vcc");) } }
    if (K == 7) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(b));) } }
    if (K == 8) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a0), "v"(b) : "vcc");) } }
    if (K == 9) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_min_u32 %0, %0, %1" : "+v"(i0) : "v"(i1));) } }
    if (K == 10) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(i0) : "v"(i1));) } }
    if (K == 11) { for (int r = 0; r < 8; ++r) { REP64(asm volatile("v_cmp_gt_f64 s[20:21], %0, %1" : : "v"(a0), "v"(b) : "s20", "s21");) } }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = clock64();
    if (threadIdx.x == 0) out[K] = t1 - t0;
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0;
}

int main() {
    long long *d, h[16] = {0};
    double *sink;
    hipMalloc(&d, 16 * 8); hipMalloc(&sink, 64 * 8);
    hipMemset(d, 0, 128);
#define RUN(K) hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64), 0, 0, d, sink, -1234.5); hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64), 0, 0, d, sink, -1234.5);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    hipDeviceSynchronize();
    hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    const char *names[] = {"v_add_f64 dep", "v_add_f64 ind x8", "v_max_f64 dep", "v_max_f64 ind x8", "v_cmp_gt_f64 -> vcc", "v_cmp_gt_f64 + v_cndmask_b32",
                           "chain step (add, cmp, cndmask, max)", "v_fma_f64 dep", "v_cmp_lt_u64 -> vcc", "v_min_u32 dep", "v_add_f32 dep", "v_cmp_gt_f64 -> sgpr pair"};
    const int per[] = {512, 512 * 8 / 8 * 8 / 8, 512, 512, 512, 512, 512, 512, 512, 512, 512, 512};
    for (int k = 0; k < 12; ++k) {
        const int n = (k == 1 || k == 3) ? 8 * 8 * 8 : 512;       // instructions (k==1,3: 8 reps x 8 x 8 instr)
        const int groups = (k == 5) ? 512 : (k == 6 ? 512 : n);
        printf("%-40s %8lld ticks / %4d = %6.2f per %s\n", names[k], h[k], groups, (double)h[k] / groups, (k == 5 || k == 6) ? "group" : "instr");
    }
    (void)per;
    return 0;
}
