/* Plain-C host of libffvd_hip.so: what a non-Python maintainer binds (include/ffvd_abi.h only, no C++/torch types).
 *   gcc -std=c99 -O2 -Iinclude examples/abi_demo.c -o abi_demo -Lffvd_amd -lffvd_hip -Wl,-rpath,$PWD/ffvd_amd -lm
 * Builds a small seeded problem, evaluates the collapsed ELBO on both routes, prints the terms. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "ffvd_abi.h"

static double lcg(unsigned long long *s) {           /* uniform in (-1, 1): enough for a demo */
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)(*s >> 11) / 9007199254740992.0) * 2.0 - 1.0;
}

int main(void) {
    enum { T = 200, D = 2, Cn = 1, M = 30, S = 2, J = 1, P = D + Cn };
    unsigned long long seed = 20230209ULL;
    static double X[S * (T + 1) * D], Z[M * P], U[M * D], Y[T * J], ctrl[T * Cn];
    double logvar[D], loglen[D * P], logQ[D], CC[D * J], DD[J], logR[J * J];
    for (int i = 0; i < T * Cn; ++i) ctrl[i] = lcg(&seed);
    for (int s = 0; s < S; ++s)
        for (int t = 0; t <= T; ++t)
            for (int d = 0; d < D; ++d) X[(s * (T + 1) + t) * D + d] = sin(0.05 * t + d) + 0.1 * lcg(&seed);
    for (int m = 0; m < M; ++m) {
        const int t = (m * 7) % T;
        for (int d = 0; d < D; ++d) Z[m * P + d] = sin(0.05 * t + d) + 0.05 * lcg(&seed);
        Z[m * P + D] = ctrl[t] + 0.05 * lcg(&seed);
    }
    for (int i = 0; i < M * D; ++i) U[i] = lcg(&seed);
    for (int t = 0; t < T; ++t) Y[t] = 0.5 * sin(0.05 * (t + 1)) + 0.05 + 0.1 * lcg(&seed);
    for (int d = 0; d < D; ++d) {
        logvar[d] = log(0.5); logQ[d] = 2.0 * log(0.4 + 0.05 * d); CC[d] = d ? -0.25 : 0.5;
        for (int p = 0; p < P; ++p) loglen[d * P + p] = log(2.0 + 0.1 * d);
    }
    DD[0] = 0.05; logR[0] = log(0.4);

    for (int route = FFVD_ROUTE_REFERENCE; route <= FFVD_ROUTE_GRAM; ++route) {
        ffvd_config cfg = {0};
        cfg.T = T; cfg.D = D; cfg.C = Cn; cfg.M = M; cfg.S_local = S; cfg.Ydim = J; cfg.shared_terms = 1;
        cfg.dtype = FFVD_F64; cfg.kernel_kind = FFVD_KERNEL_SE; cfg.branch = FFVD_BRANCH_B;
        cfg.prior_type = FFVD_PRIOR_NORMAL; cfg.route = route; cfg.jitter = 1e-5;
        ffvd_handle *h = NULL;
        if (ffvd_create(&cfg, &h) != FFVD_OK) { fprintf(stderr, "create: %s\n", ffvd_last_error(NULL)); return 1; }
        if (ffvd_set_data(h, Y, ctrl, 0) != FFVD_OK) { fprintf(stderr, "data: %s\n", ffvd_last_error(h)); return 1; }
        ffvd_params p = {X, Z, U, logvar, loglen, logQ, CC, DD, logR};
        double terms[8], nll = 0.0;
        const int rc = ffvd_elbo(h, &p, 0, terms, &nll);
        if (rc != FFVD_OK) { fprintf(stderr, "elbo (%d): %s\n", rc, ffvd_last_error(h)); return 1; }
        printf("route %d nll %.15g (chains %.0f) later_term1 %.12g later_term2 %.12g\n", route, nll, terms[7],
               terms[FFVD_TERM_LATER1] / terms[7], terms[FFVD_TERM_LATER2] / terms[7]);
        ffvd_destroy(h);
    }
    return 0;
}
