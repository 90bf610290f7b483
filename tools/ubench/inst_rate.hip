// Micro-benchmark: issue cost of single gfx950 vector instructions relative to v_fma_f64, measured as time per instruction
// of a kernel that issues 8 independent streams of the instruction per lane (4 waves per SIMD: latency hidden, the vector
// unit's issue rate is what is timed).  What the lattice loop's non-FMA instructions cost in FMA slots.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/inst_rate.hip -o tools/ubench/inst_rate && tools/ubench/inst_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define OP8(STR)                                                                                         \
    asm volatile(STR(0) "\n" STR(1) "\n" STR(2) "\n" STR(3) "\n" STR(4) "\n" STR(5) "\n" STR(6) "\n" STR(7) \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c))

#define FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9"
#define MUL(i) "v_mul_f64 %" #i ", %" #i ", %8"
#define ADD(i) "v_add_f64 %" #i ", %" #i ", %9"
#define RCP(i) "v_rcp_f64 %" #i ", %" #i
#define RSQ(i) "v_rsq_f64 %" #i ", %" #i
#define SQRT(i) "v_sqrt_f64 %" #i ", %" #i
#define RND(i) "v_rndne_f64 %" #i ", %" #i
#define FLR(i) "v_floor_f64 %" #i ", %" #i
#define FRC(i) "v_fract_f64 %" #i ", %" #i
#define LDX(i) "v_ldexp_f64 %" #i ", %" #i ", 1"
#define FRM(i) "v_frexp_mant_f64 %" #i ", %" #i
#define MIN(i) "v_min_f64 %" #i ", %" #i ", %8"
#define MOV(i) "v_mov_b64 %" #i ", %" #i
#define CMP(i) "v_cmp_lt_f64 vcc, %" #i ", %8"
#define CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"

template <int WHICH>
__global__ __launch_bounds__(256) void k(double* out, int iters, double b, double c) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
        if (WHICH == 0) OP8(FMA);
        if (WHICH == 1) OP8(MUL);
        if (WHICH == 2) OP8(ADD);
        if (WHICH == 3) OP8(RCP);
        if (WHICH == 4) OP8(RSQ);
        if (WHICH == 5) OP8(SQRT);
        if (WHICH == 6) OP8(RND);
        if (WHICH == 7) OP8(FLR);
        if (WHICH == 8) OP8(FRC);
        if (WHICH == 9) OP8(LDX);
        if (WHICH == 10) OP8(FRM);
        if (WHICH == 11) OP8(MIN);
        if (WHICH == 12) OP8(MOV);
        if (WHICH == 13) OP8(CMP);
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WHICH>
__global__ __launch_bounds__(256) void k32(double* out, int iters, double b, double c) {
    // 32-bit instructions of the loop: v_cndmask_b32, v_cvt_i32_f64, v_cvt_f64_i32, v_mbcnt, v_rcp_f32, v_cvt_f32_f64
    int a[8];
    double d[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i; d[i] = 1.0 + 0.001 * (threadIdx.x + i); }
    const int bi = (int)b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (WHICH == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(bi));
            if (WHICH == 1) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            if (WHICH == 2) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            if (WHICH == 3) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(bi));
            if (WHICH == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (WHICH == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            if (WHICH == 6) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            if (WHICH == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(bi));
            if (WHICH == 8) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(bi));
            if (WHICH == 9) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(bi) : "s20", "s21");
            if (WHICH == 10) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a[i]) : "v"(bi), "v"(a[(i + 1) & 7]) : "s20", "s21");
            if (WHICH == 11) asm volatile("v_cndmask_b32_e64 %0, 0, %0, s[20:21]" : "+v"(a[i]) : : "s20", "s21");
            if (WHICH == 12) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(bi));
            if (WHICH == 13) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(bi));
            if (WHICH == 14) asm volatile("v_cmp_lt_f64 vcc, %1, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(d[i]), "v"(bi) : "vcc");
            if (WHICH == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(bi));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_fma_ns = 0;

template <class F>
void run(const char* name, F launch) {
    const int blocks = 256 * 4, iters = 40000;     // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, blocks, 50);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        launch(out, blocks, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave-instructions per SIMD: 4 waves x iters x 8; time per wave-instruction on one SIMD
    const double ns = best * 1e6 / (4.0 * iters * 8);
    if (g_fma_ns == 0) g_fma_ns = ns;
    printf("%-22s %7.3f ms  %6.2f ns per wave-instruction  = %5.2f x v_fma_f64\n", name, best, ns, ns / g_fma_ns);
    hipFree(out);
}

#define RUN64(N, NAME) run(NAME, [](double* o, int b, int it) { hipLaunchKernelGGL((k<N>), dim3(b), dim3(256), 0, 0, o, it, 1.0000001, 1e-9); })
#define RUN32(N, NAME) run(NAME, [](double* o, int b, int it) { hipLaunchKernelGGL((k32<N>), dim3(b), dim3(256), 0, 0, o, it, 3.0, 1e-9); })

int main() {
    {   // clocks up first: ~0.5 s of FMA work (a GPU that has idled starts at low clocks and the first kernels would be priced high)
        double* out;
        hipMalloc(&out, sizeof(double) * 1024 * 256);
        for (int i = 0; i < 150; i++) hipLaunchKernelGGL((k<0>), dim3(1024), dim3(256), 0, 0, out, 40000, 1.0000001, 1e-9);
        hipDeviceSynchronize();
        hipFree(out);
    }
    RUN64(0, "v_fma_f64");
    RUN64(1, "v_mul_f64");
    RUN64(2, "v_add_f64");
    RUN64(3, "v_rcp_f64");
    RUN64(4, "v_rsq_f64");
    RUN64(5, "v_sqrt_f64");
    RUN64(6, "v_rndne_f64");
    RUN64(7, "v_floor_f64");
    RUN64(8, "v_fract_f64");
    RUN64(9, "v_ldexp_f64");
    RUN64(10, "v_frexp_mant_f64");
    RUN64(11, "v_min_f64");
    RUN64(12, "v_mov_b64");
    RUN64(13, "v_cmp_lt_f64");
    RUN32(0, "v_cndmask_b32");
    RUN32(1, "v_cvt_i32_f64");
    RUN32(2, "v_cvt_f64_i32");
    RUN32(3, "v_mbcnt_lo_u32_b32");
    RUN32(4, "v_rcp_f32");
    RUN32(5, "v_cvt_f32_f64");
    RUN32(6, "v_cvt_f64_f32");
    RUN32(7, "v_add_u32");
    RUN32(8, "v_lshl_add_u32");
    RUN32(9, "v_cndmask e64 sgpr");
    RUN32(10, "v_cndmask e64 indep");
    RUN32(11, "v_cndmask e64 const0");
    RUN32(12, "v_bfi_b32");
    RUN32(13, "v_and_b32");
    RUN32(14, "v_cmp_f64 + v_cndmask");
    RUN32(15, "v_mov_b32");
    RUN64(0, "v_fma_f64 (again)");
    RUN64(3, "v_rcp_f64 (again)");
    return 0;
}
