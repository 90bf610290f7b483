// Accuracy of the short FP64 routines of device_math.h against the host's long-double libm, in ulps of the exact value:
// fast_div, exp_neg, log_pos, sqrt_pos, mvn_phi (against erfc), phinv round trip.
//   hipcc --offload-arch=gfx950 -O3 -I ital_amd/csrc -I include tools/ubench/math_accuracy.hip -o /tmp/math_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "device_math.h"
using namespace ital;

__global__ void k(const double* a, const double* b, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[0 * n + i] = fast_div(a[i], b[i]);
    out[1 * n + i] = exp_neg(-fabs(a[i]) * 40.0);
    out[2 * n + i] = log_pos(b[i]);
    out[3 * n + i] = sqrt_pos(b[i]);
    out[4 * n + i] = mvn_phi(a[i] * 4.0);
    out[5 * n + i] = mvn_phi(mvn_phinv(0.5 * (a[i] * 0.999 + 1.0)));
}

static double ulps(double got, long double want) {
    if (want == 0) return got == 0 ? 0 : INFINITY;
    int e;
    frexpl(want, &e);
    return (double)(fabsl((long double)got - want) / ldexpl(1.0L, e - 53));
}

int main() {
    const int n = 1 << 20;
    std::vector<double> a(n), b(n), out(6 * (size_t)n);
    srand48(7);
    for (int i = 0; i < n; i++) {
        a[i] = 2 * drand48() - 1;                       // (-1, 1)
        b[i] = exp((2 * drand48() - 1) * 6.9);          // [1e-3, 1e3]
    }
    double *da, *db, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, n, dout);
    hipMemcpy(out.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
    double worst[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        worst[0] = fmax(worst[0], ulps(out[0 * (size_t)n + i], (long double)a[i] / (long double)b[i]));
        worst[1] = fmax(worst[1], ulps(out[1 * (size_t)n + i], expl(-fabsl((long double)a[i]) * 40.0L)));
        worst[2] = fmax(worst[2], ulps(out[2 * (size_t)n + i], logl((long double)b[i])));
        worst[3] = fmax(worst[3], ulps(out[3 * (size_t)n + i], sqrtl((long double)b[i])));
        worst[4] = fmax(worst[4], ulps(out[4 * (size_t)n + i], 0.5L * erfcl(-(long double)a[i] * 4.0L / sqrtl(2.0L))));
        const double p = 0.5 * (a[i] * 0.999 + 1.0);
        worst[5] = fmax(worst[5], fabs(out[5 * (size_t)n + i] - p) / (p < 1 - p ? p : 1 - p) / 1.1e-16);
    }
    const char* names[6] = {"fast_div", "exp_neg", "log_pos (ulps of the result; near 1 the result is tiny)", "sqrt_pos", "mvn_phi",
                            "Phi(Phi^-1(p)) relative to min(p, 1-p), in units of 1.1e-16"};
    for (int j = 0; j < 6; j++) printf("%-80s max %.2f\n", names[j], worst[j]);
    return 0;
}
