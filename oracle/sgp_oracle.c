/* sgp_oracle.c -- plain-C, single-thread restatement of the reference's PER-POINT algorithm.
 * TEST INFRASTRUCTURE ONLY (see oracle/sgp_oracle.py header): it checks the HIP path and times the
 * reference's algorithmic shape on the CPU; the product never links or calls it.
 *
 * Where the NumPy oracle restates the batched mathematics, this file walks the reference's own control flow:
 *   - one :v message per data point: k = K(Xu, x_n), Lambda_n = w k k', xi_n = w y k
 *                                                   (GPnode/UniSGPnode.jl:144-158, :161-173)
 *   - the N-fold product folded left, a full M x M pass per point, then mean_cov, Rv and Uv on the N-th call
 *                                                   (GPnode/UniSGPnode.jl:62-73)
 *   - one :w rule per point: alpha = L \ k, I1 = k_nn - alpha'alpha, I2 = y^2 + v - 2 y k'mu + |Uv k|^2
 *                                                   (GPnode/UniSGPnode.jl:196-238)
 *   - average energy per point                      (GPnode/UniSGPnode.jl:337-387, 411-436)
 * Parity: pinned against oracle/sgp_oracle.py (itself pinned by the reference's fixtures) in tests/test_oracle_c.py.
 * All matrices column-major; X is D x N, Xu is D x M.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void kernel_col(const double* Xu, int M, int D, const double* x, double sigma2, const double* inv_ell, double* k) {
    for (int m = 0; m < M; ++m) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
            double t = (Xu[(size_t)m * D + d] - x[d]) * inv_ell[d];
            d2 += t * t;
        }
        k[m] = sigma2 * exp(-0.5 * d2);
    }
}

/* in-place lower Cholesky; returns 0 or the failing leading minor (LAPACK convention) */
static int chol_lower(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)k * n + j] * A[(size_t)k * n + j];
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)j * n + i];
            for (int k = 0; k < j; ++k) s -= A[(size_t)k * n + i] * A[(size_t)k * n + j];
            A[(size_t)j * n + i] = s / d;
        }
        for (int i = 0; i < j; ++i) A[(size_t)j * n + i] = 0.0;
    }
    return 0;
}

/* Ainv = (L L')^-1 from the lower factor L (cholinv) */
static void chol_inverse(const double* L, int n, double* Ainv, double* W) {
    /* W = L^-1 */
    memset(W, 0, sizeof(double) * n * n);
    for (int c = 0; c < n; ++c) {
        W[(size_t)c * n + c] = 1.0 / L[(size_t)c * n + c];
        for (int i = c + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += L[(size_t)k * n + i] * W[(size_t)c * n + k];
            W[(size_t)c * n + i] = -s / L[(size_t)i * n + i];
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i) {
            double s = 0.0;
            for (int k = j; k < n; ++k) s += W[(size_t)i * n + k] * W[(size_t)j * n + k];
            Ainv[(size_t)j * n + i] = s;
            Ainv[(size_t)i * n + j] = s;
        }
}

/* Returns 0, or k > 0 for a failed Cholesky, or -1 for allocation failure.
 * prior: mu0 (M) + Sigma0 (M x M) as in `v ~ MvNormalMeanCovariance(mu_v, Sigma_v)` (experiments/regression_kin40k.ipynb:148)
 * y_var may be NULL (regression).  Outputs: mu (M), Sigma (M x M), Uv (M x M upper), out[3] = {sum I1, sum I2, energy},
 * I1/I2 per point (may be NULL). */
int oracle_vmp_sweep_perpoint(const double* Xu, int M, int D, const double* X, const double* y, const double* y_var, long N,
                              double sigma2, const double* ell, int n_ell, double jitter, double w_bar, double E_logw,
                              const double* mu0, const double* Sigma0, double* mu, double* Sigma, double* Uv, double* out,
                              double* I1_out, double* I2_out) {
    const size_t MM = (size_t)M * M;
    double* inv_ell = (double*)malloc(sizeof(double) * D);
    double* Lam = (double*)malloc(sizeof(double) * MM);
    double* tmp = (double*)malloc(sizeof(double) * MM);
    double* W = (double*)malloc(sizeof(double) * MM);
    double* Kuu = (double*)malloc(sizeof(double) * MM);
    double* xi = (double*)calloc(M, sizeof(double));
    double* k = (double*)malloc(sizeof(double) * M);
    double* a = (double*)malloc(sizeof(double) * M);
    if (!inv_ell || !Lam || !tmp || !W || !Kuu || !xi || !k || !a) return -1;
    for (int d = 0; d < D; ++d) inv_ell[d] = 1.0 / ell[n_ell == 1 ? 0 : d];
    int info = 0;

    /* caller side: Kuu and its Cholesky factor (experiments/regression_kin40k.ipynb:183-184) */
    for (int j = 0; j < M; ++j) {
        kernel_col(Xu, M, D, Xu + (size_t)j * D, sigma2, inv_ell, Kuu + (size_t)j * M);
        Kuu[(size_t)j * M + j] += jitter;
    }
    if ((info = chol_lower(Kuu, M)) != 0) goto done;

    /* prior -> weighted mean / precision */
    memcpy(tmp, Sigma0, sizeof(double) * MM);
    if ((info = chol_lower(tmp, M)) != 0) goto done;
    chol_inverse(tmp, M, Lam, W);
    for (int i = 0; i < M; ++i) {
        double s = 0.0;
        for (int j = 0; j < M; ++j) s += Lam[(size_t)j * M + i] * mu0[j];
        xi[i] = s;
    }
    /* HOT LOOP 1 + 2: one rank-1 message and one M x M fold per point (GPnode/UniSGPnode.jl:62-63,153-156) */
    for (long n = 0; n < N; ++n) {
        kernel_col(Xu, M, D, X + (size_t)n * D, sigma2, inv_ell, k);
        for (int j = 0; j < M; ++j) {
            const double wk = w_bar * k[j];
            double* col = Lam + (size_t)j * M;
            for (int i = 0; i < M; ++i) col[i] += wk * k[i];
        }
        const double wy = w_bar * y[n];
        for (int i = 0; i < M; ++i) xi[i] += wy * k[i];
    }
    /* the N-th product: mean_cov, Rv, Uv (GPnode/UniSGPnode.jl:66-69) */
    memcpy(tmp, Lam, sizeof(double) * MM);
    if ((info = chol_lower(tmp, M)) != 0) goto done;
    chol_inverse(tmp, M, Sigma, W);
    for (int i = 0; i < M; ++i) {
        double s = 0.0;
        for (int j = 0; j < M; ++j) s += Sigma[(size_t)j * M + i] * xi[j];
        mu[i] = s;
    }
    for (int j = 0; j < M; ++j)
        for (int i = 0; i < M; ++i) tmp[(size_t)j * M + i] = Sigma[(size_t)j * M + i] + mu[i] * mu[j];
    if ((info = chol_lower(tmp, M)) != 0) goto done;
    for (int j = 0; j < M; ++j)
        for (int i = 0; i < M; ++i) Uv[(size_t)j * M + i] = (i <= j) ? tmp[(size_t)i * M + j] : 0.0;   /* Uv = L_R' */

    /* one :w rule + average energy per point (GPnode/UniSGPnode.jl:196-238, 337-387) */
    {
        double s1 = 0.0, s2 = 0.0, U = 0.0;
        for (long n = 0; n < N; ++n) {
            kernel_col(Xu, M, D, X + (size_t)n * D, sigma2, inv_ell, k);
            for (int i = 0; i < M; ++i) {               /* alpha = KuuL \ k */
                double s = k[i];
                for (int c = 0; c < i; ++c) s -= Kuu[(size_t)c * M + i] * a[c];
                a[i] = s / Kuu[(size_t)i * M + i];
            }
            double aa = 0.0, kmu = 0.0, bb = 0.0;
            for (int i = 0; i < M; ++i) { aa += a[i] * a[i]; kmu += k[i] * mu[i]; }
            for (int i = 0; i < M; ++i) {               /* beta = Uv k */
                double s = 0.0;
                for (int j = i; j < M; ++j) s += Uv[(size_t)j * M + i] * k[j];
                bb += s * s;
            }
            const double I1 = sigma2 - aa;
            const double I2 = y[n] * y[n] + (y_var ? y_var[n] : 0.0) - 2.0 * y[n] * kmu + bb;
            if (I1_out) I1_out[n] = I1;
            if (I2_out) I2_out[n] = I2;
            s1 += I1;
            s2 += I2;
            U += 0.5 * (I1 * w_bar - E_logw + 1.8378770664093454835606594728112 + I2 * w_bar);
        }
        out[0] = s1; out[1] = s2; out[2] = U;
    }
done:
    free(inv_ell); free(Lam); free(tmp); free(W); free(Kuu); free(xi); free(k); free(a);
    return info;
}
