// Host-side exerciser of libital_hip.so for the AddressSanitizer / UBSan build (tools/asan_host.sh): everything the C ABI
// does BEFORE a kernel launch -- argument validation, descriptor handling, error strings -- and the pure host code (stream
// bookkeeping of SciPy's mvndst, the walker of numpy's legacy generator, the RCCL lookup).  Runs without a GPU: every call
// below either is host-only or is refused by the validation in front of the launch.  Prints OK and exits 0, or the failed
// check; the sanitizers abort on their own findings.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ital_hip.h"

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                                      \
        }                                                                    \
    } while (0)

// MVNUNI stepped naively (L'Ecuyer 1996): the reference for ital_mvn_advance
static void mrg_step(long long s[6]) {
    const long long m1 = 2147483647LL, m2 = 2145483479LL;
    long long p1 = (63308LL * s[1] - 183326LL * s[0]) % m1;
    if (p1 < 0) p1 += m1;
    long long p2 = (86098LL * s[5] - 539608LL * s[3]) % m2;
    if (p2 < 0) p2 += m2;
    s[0] = s[1]; s[1] = s[2]; s[2] = p1;
    s[3] = s[4]; s[4] = s[5]; s[5] = p2;
}

int main() {
    CHECK(strstr(ital_version(), "gfx950") != nullptr);
    CHECK(ital_launch_count() == 0);

    // ---- mvndst stream bookkeeping
    int st[6];
    CHECK(ital_mvn_seed(st) == 0);
    long long ref[6];
    for (int i = 0; i < 6; i++) ref[i] = st[i];
    CHECK(ital_mvn_advance(st, 1000) == 0);
    for (int i = 0; i < 1000; i++) mrg_step(ref);
    for (int i = 0; i < 6; i++) CHECK(st[i] == (int)ref[i]);
    CHECK(ital_mvn_advance(st, 0) == 0);
    CHECK(ital_mvn_advance(st, -5) != 0 && strstr(ital_last_error(), "ital_mvn_advance"));
    CHECK(ital_mvn_advance(nullptr, 5) != 0);
    CHECK(ital_mvn_seed(nullptr) != 0);
    CHECK(ital_mvn_draws_per_call(2) == 0 && ital_mvn_draws_per_call(5) == 56);
    {
        std::vector<long long> jump(ITAL_JUMP_BITS * 18), pat((1 << ITAL_MAX_T) * 18);
        std::vector<double> vk(ITAL_MAX_T - 1);
        for (int t = 3; t <= ITAL_MAX_T; t++) CHECK(ital_mvn_tables(t, jump.data(), pat.data(), vk.data()) == 0);
        CHECK(vk[0] == 1.0 / 397 && pat[0] == 1 && pat[1] == 0);
        CHECK(ital_mvn_tables(2, jump.data(), nullptr, nullptr) != 0);
        CHECK(ital_mvn_tables(ITAL_MAX_T + 1, nullptr, pat.data(), nullptr) != 0);
        CHECK(ital_mvn_tables(ITAL_GENERIC_MAX_DIM, jump.data(), nullptr, nullptr) == 0);
        std::vector<double> vka((ITAL_GENERIC_MAX_DIM + 1) * ITAL_GENERIC_MAX_DIM);
        CHECK(ital_mvn_generic_tables(ITAL_GENERIC_MAX_DIM, jump.data(), vka.data()) == 0);
        CHECK(ital_mvn_generic_tables(ITAL_GENERIC_MAX_DIM + 1, nullptr, nullptr) != 0);
        CHECK(jump[0] == 0 && jump[1] == 1);        // one draw: the companion matrix itself
        int seeds[ITAL_MAX_T + 1][6];
        int s2[6];
        ital_mvn_seed(s2);
        CHECK(ital_mvn_round_seeds(s2, 100, 4, seeds) == 0);
        int s3[6];
        ital_mvn_seed(s3);
        for (int j = 0; j < 6; j++) CHECK(seeds[1][j] == s3[j] && seeds[3][j] == s3[j]);   // t = 1, 2 draw nothing
        ital_mvn_advance(s3, 98LL * 16 * 24);
        for (int j = 0; j < 6; j++) CHECK(seeds[4][j] == s3[j]);
        CHECK(ital_mvn_round_seeds(s2, 3, 4, seeds) != 0);
    }

    // ---- numpy's legacy generator: skip == draw-and-discard, with every parity and across block boundaries, threads or not
    {
        ital_np_legacy_state a, b;
        memset(&a, 0, sizeof(a));
        for (int i = 0; i < 624; i++) a.key[i] = 1812433253u * (uint32_t)(i + 1) + 12345u;
        a.pos = 624;
        b = a;
        std::vector<double> x(200001), y(200001);
        CHECK(ital_np_legacy_normals(&a, 0, x.data(), 200001, 1) == 0);
        CHECK(ital_np_legacy_normals(&b, 0, y.data(), 200001, 4) == 0);
        CHECK(memcmp(x.data(), y.data(), x.size() * sizeof(double)) == 0);
        CHECK(memcmp(&a, &b, sizeof(a)) == 0 && a.has_gauss == 1);
        ital_np_legacy_state c;
        memset(&c, 0, sizeof(c));
        for (int i = 0; i < 624; i++) c.key[i] = 1812433253u * (uint32_t)(i + 1) + 12345u;
        c.pos = 624;
        double tail[7];
        CHECK(ital_np_legacy_normals(&c, 1001, tail, 7, 1) == 0);
        for (int i = 0; i < 7; i++) CHECK(tail[i] == x[1001 + i] && isfinite(tail[i]));
        CHECK(ital_np_legacy_normals(&c, 200001 - 1008, nullptr, 0, 1) == 0);
        CHECK(memcmp(&a, &c, sizeof(a)) == 0);
        CHECK(ital_np_legacy_normals(nullptr, 0, nullptr, 0, 1) != 0);
        CHECK(ital_np_legacy_normals(&c, -1, nullptr, 0, 1) != 0);
        CHECK(ital_np_legacy_normals(&c, 0, nullptr, 3, 1) != 0);
        c.pos = 700;
        CHECK(ital_np_legacy_normals(&c, 0, tail, 1, 1) != 0 && strstr(ital_last_error(), "position"));
    }

    // ---- validation in front of the launches (nothing below reaches the device)
    CHECK(ital_score_workspace(4, 10) > 0 && ital_score_workspace(2, 10) == 0 && ital_score_workspace(9, 10) == 0);
    CHECK(ital_mcmi_workspace(6, 10) > 0 && ital_mcmi_workspace(4, 10) == 0);
    CHECK(ital_topk_workspace() > 0);
    CHECK(ital_score_step(nullptr, nullptr) != 0 && strstr(ital_last_error(), "null descriptor"));
    CHECK(ital_score_generic(nullptr, nullptr) != 0);
    CHECK(ital_mcmi_score_step(nullptr, nullptr) != 0);
    CHECK(ital_fetch_round(nullptr, nullptr) != 0);
    CHECK(ital_gp_append(nullptr, nullptr) != 0);
    {
        ital_score_desc d;
        memset(&d, 0, sizeof(d));
        CHECK(ital_score_step(&d, nullptr) == 0);                 // no candidates: nothing to do
        d.n_cand = 5;
        d.t = 0;
        CHECK(ital_score_step(&d, nullptr) != 0);
        d.t = ITAL_MAX_T + 1;
        CHECK(ital_score_step(&d, nullptr) != 0);
        d.t = 3;
        d.batch.kmax = 2;
        CHECK(ital_score_step(&d, nullptr) != 0 && strstr(ital_last_error(), "batch capacity"));
        d.batch.kmax = 4;
        CHECK(ital_score_step(&d, nullptr) != 0 && strstr(ital_last_error(), "jump tables"));
        double dummy[4] = {0, 0, 0, 0};
        d.sel_record = dummy;                                     // fused selection without its buffers
        d.t = 1;
        CHECK(ital_score_step(&d, nullptr) != 0 && strstr(ital_last_error(), "fused selection"));
        ital_round_desc r;
        memset(&r, 0, sizeof(r));
        r.step = d;
        r.k = 0;
        CHECK(ital_fetch_round(&r, nullptr) != 0);
        r.k = 3;
        r.step.n_cand = 2;
        CHECK(ital_fetch_round(&r, nullptr) != 0 && strstr(ital_last_error(), "fewer candidates"));
        r.step.n_cand = (1 << 18) + 1;
        CHECK(ital_fetch_round(&r, nullptr) != 0);
        r.step.n_cand = 10;
        r.step.sel_record = nullptr;
        CHECK(ital_fetch_round(&r, nullptr) != 0 && strstr(ital_last_error(), "sel_"));
        r.step.sel_record = dummy;
        r.step.sel_ret = reinterpret_cast<int64_t*>(dummy);
        r.begin = 3;
        CHECK(ital_fetch_round(&r, nullptr) != 0);
        r.begin = 2;
        CHECK(ital_fetch_round(&r, nullptr) != 0 && strstr(ital_last_error(), "previous list"));
    }
    {
        ital_mcmi_desc m;
        memset(&m, 0, sizeof(m));
        CHECK(ital_mcmi_score_step(&m, nullptr) == 0);
        m.n_i = 4; m.n_all = 3; m.t = 2; m.batch.kmax = 4;
        CHECK(ital_mcmi_score_step(&m, nullptr) != 0 && strstr(ital_last_error(), "candidate slice"));
        m.n_all = 8; m.ld_cov = 4; m.ldc = 8;
        CHECK(ital_mcmi_score_step(&m, nullptr) != 0 && strstr(ital_last_error(), "leading dimension"));
        ital_append_desc a;
        memset(&a, 0, sizeof(a));
        CHECK(ital_gp_append(&a, nullptr) != 0);
        a.lb.c = 17;
        CHECK(ital_gp_append(&a, nullptr) != 0);
        a.lb.c = 4; a.m = 30; a.ldl = 32;
        CHECK(ital_gp_append(&a, nullptr) != 0 && strstr(ital_last_error(), "capacity"));
        a.m = 8; a.ldx = 20;
        CHECK(ital_gp_append(&a, nullptr) != 0 && strstr(ital_last_error(), "multiple of 16"));
    }
    {
        ital_batch b;
        memset(&b, 0, sizeof(b));
        b.kmax = 4; b.ldx = 16; b.ldw = 16;
        CHECK(ital_select_resolve(nullptr, 1, 7, 0, 0, 0, b, nullptr, nullptr, nullptr) != 0);      // record length mismatch
        CHECK(ital_select_resolve(nullptr, 1, ITAL_REC_HEADER + 36, 0, 0, 9, b, nullptr, nullptr, nullptr) != 0);
        CHECK(ital_select_fused(nullptr, nullptr, nullptr, 1, 0, nullptr, 0, 0, 2, nullptr, nullptr, nullptr, nullptr, 16, nullptr,
                                16, 0, 16, nullptr, 16, 0, 0, b, nullptr, nullptr, nullptr, nullptr) != 0);   // mode
        CHECK(ital_select_local(nullptr, nullptr, nullptr, 1, 0, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, 16, nullptr,
                                16, 20, 16, nullptr, 16, 0, 4, nullptr, nullptr, nullptr, nullptr) != 0);      // m > ldw
        double rec[8];
        CHECK(ital_select_exchange(rec, rec, 8, nullptr, nullptr) != 0 && strstr(ital_last_error(), "communicator"));
        int fake_comm = 0;
        CHECK(ital_select_exchange(rec, rec, 8, &fake_comm, nullptr) != 0 && strstr(ital_last_error(), "RCCL"));   // none loaded here
        CHECK(ital_select_exchange(nullptr, rec, 8, &fake_comm, nullptr) != 0);
    }
    CHECK(ital_row_norms(nullptr, 0, 16, nullptr, nullptr) == 0);
    CHECK(ital_row_norms(nullptr, 4, 20, nullptr, nullptr) != 0);
    CHECK(ital_rbf_cols(nullptr, nullptr, 4, 20, nullptr, nullptr, 1, 1.0, 1.0, nullptr, 4, nullptr) != 0);
    CHECK(ital_rbf_cols(nullptr, nullptr, 4, 16, nullptr, nullptr, 17, 1.0, 1.0, nullptr, 4, nullptr) != 0);
    CHECK(ital_cov_block(nullptr, nullptr, 4, nullptr, nullptr, 4, 20, nullptr, 0, nullptr, 0, 0, 1.0, 1.0, nullptr, 4, nullptr) != 0);
    CHECK(ital_cov_block(nullptr, nullptr, 4, nullptr, nullptr, 8, 16, nullptr, 0, nullptr, 0, 0, 1.0, 1.0, nullptr, 4, nullptr) != 0);
    CHECK(ital_chol_append(nullptr, nullptr, 16, nullptr, 8, nullptr, nullptr, 6, 4, 1.0, 1.0, 1e-6, nullptr, nullptr) != 0);
    CHECK(ital_topk(nullptr, 10, 0, 0, nullptr, nullptr, nullptr, nullptr) != 0);
    CHECK(ital_launch_count() == 0);            // nothing above reached a launch
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("OK host side under ASan + UBSan\n");
    return 0;
}
