// A host that is not Python: one whole fetch_unlabelled(k) of the reference (ital/ital.py:84-134) on the golden USPS
// fixture, driven through libital_hip.so from C++ alone -- include/ital_hip.h for the declarations, the HIP runtime for the
// device memory, nothing else.  Built and run by tests/test_gpu_host_cpp.py (-m gpu):
//
//     host_gpu_driver <X.f64> <n> <d> <length_scale> <var> <noise> <query> <k> <expected picks ...>
//
// X.f64: the n x d feature matrix as raw little-endian doubles (the test writes it from tests/golden/usps500.npz, which the
// real reference produced).  Exit code 0: the picks are the reference's; 1: they differ; 2: a call failed.
// The sequence is INTEGRATION.md section B: update({query: +1}) = stage + Cholesky append + whitening sweep
// (gp.py:164-200), then per greedy step ital_score_step -> ital_select_fused -> ital_cross_cov_cols, the stream position of
// SciPy's mvndst advanced with ital_mvn_advance exactly as the serial reference consumes it.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ital_hip.h"

#define HIP_OK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return 2;                                                                         \
        }                                                                                     \
    } while (0)
#define ITAL_OK(call)                                                                         \
    do {                                                                                      \
        int rc_ = (call);                                                                     \
        if (rc_ != 0) {                                                                       \
            fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, ital_last_error()); \
            return 2;                                                                         \
        }                                                                                     \
    } while (0)

template <class T>
static T* dev_zeros(size_t count) {
    T* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T) + 64) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, count * sizeof(T) + 64) != hipSuccess) return nullptr;
    return p;
}

// The same round through the context-style layer (ital_ctx_*: the library owns the device buffers; csrc/ctx.hip).
static int run_ctx(const std::vector<double>& X, int64_t n, int d, double ls, double var, double noise, int query, int k,
                   char** want) {
    ital_ctx* ctx = nullptr;
    ITAL_OK(ital_ctx_create(n, d, ls, var, noise, 64, 0, 1, nullptr, &ctx));
    ITAL_OK(ital_ctx_fit(ctx, X.data(), 0, nullptr));
    const int64_t q = query;
    const double one = 1.0;
    ITAL_OK(ital_ctx_update(ctx, &q, &one, 1, nullptr));
    std::vector<int64_t> picks(k);
    const int got = ital_ctx_fetch(ctx, k, picks.data(), nullptr);
    if (got != k) {
        fprintf(stderr, "ital_ctx_fetch -> %d: %s\n", got, ital_last_error());
        return 2;
    }
    int bad = 0;
    printf("picks (context API):");
    for (int t = 0; t < k; t++) {
        printf(" %lld", (long long)picks[t]);
        if (picks[t] != atoll(want[t])) bad = 1;
    }
    printf("  %s\n", bad ? "MISMATCH" : "ok (the reference's batch)");
    ITAL_OK(ital_ctx_destroy(ctx));
    return bad;
}

int main(int argc, char** argv) {
    bool use_ctx = false;
    if (argc > 1 && strcmp(argv[1], "--ctx") == 0) {      // host_gpu_driver --ctx X.f64 ...: the context-style layer
        use_ctx = true;
        argv++;
        argc--;
    }
    if (argc < 10) {
        fprintf(stderr, "usage: %s X.f64 n d length_scale var noise query k picks...\n", argv[0]);
        return 2;
    }
    const int64_t n = atoll(argv[2]);
    const int d = atoi(argv[3]);
    const double ls = atof(argv[4]), var = atof(argv[5]), noise = atof(argv[6]);
    const int query = atoi(argv[7]), k = atoi(argv[8]);
    if (argc != 9 + k || k < 1 || k > ITAL_MAX_T) {
        fprintf(stderr, "expected %d picks\n", k);
        return 2;
    }
    std::vector<double> X((size_t)n * d);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(X.data(), sizeof(double), X.size(), f) != X.size()) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    fclose(f);
    if (use_ctx) return run_ctx(X, n, d, ls, var, noise, query, k, argv + 9);
    hipStream_t st = nullptr;
    HIP_OK(hipStreamCreate(&st));

    // ---- data and GP state: features padded to a multiple of 16, squared norms, Cholesky factor, whitened block
    const int ldx = (d + 15) / 16 * 16, cap = 16, kmax = 4 > k ? 4 : k;
    const int64_t ldv = (n + 15) / 16 * 16;
    double* Xd = dev_zeros<double>((size_t)n * ldx);
    double* xn = dev_zeros<double>(n);
    double* Lc = dev_zeros<double>((size_t)cap * cap);
    double* alpha = dev_zeros<double>(cap);
    double* XT = dev_zeros<double>((size_t)cap * ldx);
    double* XTn = dev_zeros<double>(cap);
    double* V = dev_zeros<double>((size_t)cap * ldv);
    double* mu = dev_zeros<double>(n);
    double* s2 = dev_zeros<double>(n);
    double* ybuf = dev_zeros<double>(16);
    int* status = dev_zeros<int>(1);
    if (!Xd || !xn || !Lc || !alpha || !XT || !XTn || !V || !mu || !s2 || !ybuf || !status) return 2;
    HIP_OK(hipMemcpy2D(Xd, (size_t)ldx * sizeof(double), X.data(), (size_t)d * sizeof(double), (size_t)d * sizeof(double), n,
                       hipMemcpyHostToDevice));
    {
        std::vector<double> v(n, var);
        HIP_OK(hipMemcpy(s2, v.data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    ITAL_OK(ital_row_norms(Xd, n, ldx, xn, st));
    ital_label_batch lb;
    memset(&lb, 0, sizeof(lb));
    lb.c = 1;
    lb.slot[0] = query;
    lb.y[0] = 1.0;
    ITAL_OK(ital_stage_labelled(Xd, ldx, lb, XT, XTn, ybuf, st));
    ITAL_OK(ital_chol_append(XT, XTn, ldx, Lc, cap, alpha, ybuf, 0, 1, var, ls, noise, status, st));
    ITAL_OK(ital_whiten_append(Xd, xn, n, ldx, XT, XTn, 1, Lc, cap, Lc, alpha, V, ldv, 0, var, ls, mu, s2, st));
    const int m = 1;

    // ---- candidate list (ascending, without the labelled sample: retrieval_base.py:78-87) and batch state
    std::vector<int32_t> cand_h;
    for (int64_t i = 0; i < n; i++)
        if (i != query) cand_h.push_back((int32_t)i);
    const int64_t nc = (int64_t)cand_h.size();
    int32_t* cand = dev_zeros<int32_t>(nc);
    uint8_t* alive = dev_zeros<uint8_t>(nc);
    double* mi = dev_zeros<double>(nc);
    HIP_OK(hipMemcpy(cand, cand_h.data(), nc * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(alive, 1, nc));
    ital_batch batch;
    memset(&batch, 0, sizeof(batch));
    batch.kmax = kmax;
    batch.ldx = ldx;
    batch.ldw = cap;
    batch.bidx = dev_zeros<int64_t>(kmax);
    batch.bgpos = dev_zeros<int64_t>(kmax);
    batch.bsort = dev_zeros<int32_t>(kmax);
    batch.bmu = dev_zeros<double>(kmax);
    batch.sig = dev_zeros<double>((size_t)kmax * kmax);
    batch.XB = dev_zeros<double>((size_t)kmax * ldx);
    batch.XBn = dev_zeros<double>(kmax);
    batch.VB = dev_zeros<double>((size_t)kmax * cap);
    double* C = dev_zeros<double>((size_t)kmax * ldv);
    int64_t* ret = dev_zeros<int64_t>(kmax + 1);
    double* rec = dev_zeros<double>(ital_record_len(ldx, cap, kmax));     // the size helper instead of the header's prose

    // ---- the stream of SciPy's mvndst: state and tables come from the library
    int state[6];
    ITAL_OK(ital_mvn_seed(state));
    int64_t n_alive = nc;
    for (int t = 1; t <= k; t++) {
        ital_score_desc desc;
        memset(&desc, 0, sizeof(desc));
        desc.t = t;
        desc.n_cand = nc;
        desc.cand = cand;
        desc.alive = alive;
        desc.mu = mu;
        desc.s2 = s2;
        desc.C = C;
        desc.ldc = ldv;
        desc.batch = batch;
        desc.noise = noise;
        desc.eps = 1e-12;
        desc.mi = mi;
        desc.status = status;
        if (t >= 3) {
            std::vector<long long> jump((size_t)ITAL_JUMP_BITS * 18), pat((size_t)(1 << t) * 18);
            std::vector<double> vk(t - 1);
            ITAL_OK(ital_mvn_tables(t, jump.data(), pat.data(), vk.data()));
            long long* jd = dev_zeros<long long>(jump.size());
            long long* pd = dev_zeros<long long>(pat.size());
            double* vd = dev_zeros<double>(vk.size());
            HIP_OK(hipMemcpy(jd, jump.data(), jump.size() * sizeof(long long), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(pd, pat.data(), pat.size() * sizeof(long long), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(vd, vk.data(), vk.size() * sizeof(double), hipMemcpyHostToDevice));
            const int64_t wd = ital_round_workspace(t, nc, 0);
            desc.jump = jd;
            desc.jumppat = pd;
            desc.vk = vd;
            desc.work = dev_zeros<double>(wd);
            desc.work_doubles = wd;
            if (!desc.work) return 2;
            for (int j = 0; j < 6; j++) desc.seed[j] = state[j];
        }
        ITAL_OK(ital_score_step(&desc, st));
        ITAL_OK(ital_select_fused(mi, cand, alive, nc, 0, nullptr, 0, 0, 0, mu, s2, Xd, xn, ldx, V, ldv, m, cap, C, ldv, t - 1, t - 1,
                                  batch, status, rec, ret, st));
        if (t < k) {
            const int slot = t - 1;
            ITAL_OK(ital_cross_cov_cols(Xd, xn, n, ldx, batch.XB + (size_t)slot * ldx, batch.XBn + slot, 1,
                                        batch.VB + (size_t)slot * cap, cap, V, ldv, m, var, ls, C + (size_t)slot * ldv, ldv, st));
        }
        // the serial reference has now made 2 * 2^t mvndst calls per live candidate (ital.py:191-206)
        ITAL_OK(ital_mvn_advance(state, n_alive * (int64_t)(2 << t) * ital_mvn_draws_per_call(t)));
        n_alive--;
    }
    std::vector<int64_t> host(kmax + 1);
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(host.data(), ret, (kmax + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    int status_h = 0;
    HIP_OK(hipMemcpy(&status_h, status, sizeof(int), hipMemcpyDeviceToHost));
    int bad = (host[kmax] != 0 || status_h != 0) ? 1 : 0;
    printf("picks:");
    for (int t = 0; t < k; t++) {
        printf(" %lld", (long long)host[t]);
        if (host[t] != atoll(argv[9 + t])) bad = 1;
    }
    printf("  status %lld / %d  %s\n", (long long)host[kmax], status_h, bad ? "MISMATCH" : "ok (the reference's batch)");
    return bad;
}
