// Issue cost of instruction patterns for ONE wavefront alone on its SIMD (gfx950): what a lone, in-order wave pays for
// dependent vector instructions, taken branches, vector->scalar hand-overs and lane reads.  Diagnostic; prints the time of
// each pattern per repetition relative to a chain of independent fp32 FMAs (one issue slot each).
//   hipcc --offload-arch=gfx950 -O2 -o issue_latency issue_latency.hip && ./issue_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 4096
#define TICK(t) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")

__global__ void k(unsigned long long* out, float* sink, double* dsink) {
    unsigned long long t0, t1;
    float a = sink[threadIdx.x], b = 1.0001f, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    double da = dsink[threadIdx.x], db = 1.0001, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    int idx = 0;
    // 0: independent fp32 FMAs (4 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 1: dependent fp32 FMAs (4 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0" : "+v"(c0) : "v"(b));
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 2: independent fp64 FMAs (4 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db));
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 3: dependent fp64 FMAs (4 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0" : "+v"(d0) : "v"(db));
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 4: four taken branches per rep (plus 4 independent FMAs)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_fma_f32 %0, %4, %5, %0\n s_branch 1f\n 1: v_fma_f32 %1, %4, %5, %1\n s_branch 2f\n 2: v_fma_f32 %2, %4, %5, %2\n s_branch 3f\n 3: v_fma_f32 %3, %4, %5, %3\n s_branch 4f\n 4:" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 5: four not-taken conditional branches per rep (plus 4 FMAs)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_fma_f32 %0, %4, %5, %0\n s_cbranch_execz 1f\n v_fma_f32 %1, %4, %5, %1\n s_cbranch_execz 1f\n v_fma_f32 %2, %4, %5, %2\n s_cbranch_execz 1f\n v_fma_f32 %3, %4, %5, %3\n s_cbranch_execz 1f\n 1:" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 6: compare -> saveexec -> masked op -> restore (2 per rep), the shape of a divergent `if`
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n s_and_saveexec_b64 s[40:41], vcc\n v_fma_f32 %0, %2, %3, %0\n s_mov_b64 exec, s[40:41]\n v_cmp_lt_f32 vcc, %1, %2\n s_and_saveexec_b64 s[40:41], vcc\n v_fma_f32 %1, %2, %3, %1\n s_mov_b64 exec, s[40:41]" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b) : "vcc", "s40", "s41", "scc");
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 7: readlane -> scalar op -> vector use (2 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_readlane_b32 s40, %0, 3\n s_add_u32 s40, s40, 1\n v_add_u32 %0, s40, %0\n v_readlane_b32 s41, %1, 5\n s_add_u32 s41, s41, 1\n v_add_u32 %1, s41, %1" : "+v"(c0), "+v"(c1) :: "s40", "s41", "scc");
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 8: four independent scalar ops per rep
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1" ::: "s40", "s41", "s42", "s43", "scc");
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 9: LDS read -> use, dependent (2 per rep)
    __shared__ float lds[256];
    lds[threadIdx.x] = 0.0f;
    __syncthreads();
    TICK(t0);
    { unsigned addr = threadIdx.x * 4;
      for (int i = 0; i < REP; i++) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "+v"(c0), "+v"(addr)); }
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 10: v_mov chain dependent (4 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1" : "+v"(c0), "+v"(c1));
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 11: v_cmp (to vcc) -> v_cndmask dependent pairs (2 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %3, vcc\n v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b) : "vcc");
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 12: fp64 reciprocal, dependent (2 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0" : "+v"(d0));
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    // 13: ballot-style: v_cmp to sgpr pair -> s_bcnt1 -> s_cmp -> s_cbranch not taken (2 per rep)
    TICK(t0);
    for (int i = 0; i < REP; i++) asm volatile("v_cmp_lt_f32 s[40:41], %0, %2\n s_bcnt1_i32_b64 s42, s[40:41]\n s_cmp_eq_u32 s42, 99\n s_cbranch_scc1 1f\n v_fma_f32 %0, %2, %3, %0\n v_cmp_lt_f32 s[40:41], %1, %2\n s_bcnt1_i32_b64 s42, s[40:41]\n s_cmp_eq_u32 s42, 99\n s_cbranch_scc1 1f\n v_fma_f32 %1, %2, %3, %1\n 1:" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b) : "s40", "s41", "s42", "scc");
    TICK(t1); if (threadIdx.x == 0) out[idx] = t1 - t0; idx++;
    sink[threadIdx.x] = c0 + c1 + c2 + c3; dsink[threadIdx.x] = d0 + d1 + d2 + d3;
}

int main() {
    unsigned long long* out; float* sink; double* dsink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 64 * 4); hipMalloc(&dsink, 64 * 8);
    hipMemset(sink, 0, 64 * 4); hipMemset(dsink, 0, 64 * 8);
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, dsink); hipDeviceSynchronize(); }
    unsigned long long h[16];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"4 independent fp32 fma", "4 dependent fp32 fma", "4 independent fp64 fma", "4 dependent fp64 fma", "4 taken branches + 4 fma", "4 not-taken cbranch + 4 fma",
                           "2 x (cmp, saveexec, fma, restore)", "2 x (readlane, s_add, v_add)", "4 scalar adds", "2 x dependent LDS read", "4 dependent v_mov",
                           "2 x (v_cmp vcc, v_cndmask)", "2 dependent v_rcp_f64", "2 x (v_cmp sgpr, s_bcnt, s_cmp, cbranch, fma)"};
    const double unit = (double)h[0] / 4.0;   // ticks per independent fma over REP repetitions: one issue slot
    for (int i = 0; i < 14; i++) printf("%-48s %8llu ticks  = %6.2f issue slots per rep\n", names[i], h[i], (double)h[i] / unit);
    return 0;
}
