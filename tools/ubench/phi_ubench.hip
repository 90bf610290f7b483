// Micro-benchmark: throughput of the (Phi, Phi^-1) chain of the lattice integrand in isolation, NC independent chains
// per lane, to separate the cost of the arithmetic itself from the kernel's bookkeeping (compaction, LDS, scalar work).
//   hipcc --offload-arch=gfx950 -O3 -I ital_amd/csrc -I include tools/ubench/phi_ubench.hip -o /tmp/phi_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "device_math.h"
using namespace ital;

template <int NC, int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0) {
    double y[NC], acc[NC];
    for (int c = 0; c < NC; c++) { y[c] = 0.001 * (threadIdx.x + 7 * c) - 0.1; acc[c] = 0; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const double ph = mvn_phi(a0 - 0.3 * y[c]);
            const double w = 1.0 - ph;
            acc[c] += w;
            double p = fma(0.37 + 0.001 * c, w, ph);
            if (MODE == 0) y[c] = phinv_central(p);        // central branch only
            else y[c] = mvn_phinv(p);                      // with the divergent tail branch
        }
    }
    double s = 0;
    for (int c = 0; c < NC; c++) s += acc[c] + y[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NC, int MODE>
void run(const char* name, int blocks) {
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NC, MODE>), dim3(blocks), dim3(256), 0, 0, out, 10, 0.2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NC, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.2);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double pairs = (double)blocks * 256 * NC * iters;
    printf("%-28s blocks %5d  %8.3f ms  %8.2f G pairs/s\n", name, blocks, ms, pairs / ms / 1e6);
    hipFree(out);
}

int main() {
    for (int blocks : {256 * 2, 256 * 3, 256 * 4, 256 * 8}) {
        run<1, 0>("1 chain central", blocks);
        run<2, 0>("2 chains central", blocks);
        run<4, 0>("4 chains central", blocks);
        run<8, 0>("8 chains central", blocks);
        run<4, 1>("4 chains with tail branch", blocks);
    }
    return 0;
}
