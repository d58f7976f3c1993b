/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) - C restatement of the reference's IRL inner loop, used
 *   (1) as a second checker (tests/test_oracle_c.py compares it with oracle/pdp_oracle.py), and
 *   (2) as the timed CPU baseline of bench.py (`cpu_baseline.kind = "port"`): the same arithmetic the reference does
 *       in Python/CasADi/numpy, in the reference's order of operations, OpenMP over the batch.
 * Follows (paths relative to the reference repository):
 *   rollout of given controls / PMP costates   PDP/PDP.py:148-170, 199-209
 *   OCSys.getAuxSys                            PDP/PDP.py:272-314   (model derivatives: oracle/gen/<sys>_oc.c, sympy-generated)
 *   LQR.lqrSolver backward / forward           PDP/PDP.py:557-608   (explicit inverses by LU with partial pivoting, like numpy.linalg.inv)
 *   IRL loss / chain rule                      Examples/IRL/cartpole/cartpole_PDP.py:63-74
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DECL(s)                                                                                                              \
    extern const int s##_dims[3];                                                                                            \
    void s##_dyn(const double*, const double*, const double*, double*);                                                      \
    void s##_dHx(const double*, const double*, const double*, const double*, double*);                                       \
    void s##_aux(const double*, const double*, const double*, const double*, double*, double*, double*, double*, double*,     \
                 double*, double*, double*);                                                                                 \
    void s##_fin(const double*, const double*, double*, double*, double*);
DECL(pendulum) DECL(cartpole) DECL(robotarm) DECL(quadrotor) DECL(rocket)

typedef struct {
    const char* name;
    const int* dims;
    void (*dyn)(const double*, const double*, const double*, double*);
    void (*dHx)(const double*, const double*, const double*, const double*, double*);
    void (*aux)(const double*, const double*, const double*, const double*, double*, double*, double*, double*, double*, double*, double*, double*);
    void (*fin)(const double*, const double*, double*, double*, double*);
} model_t;
#define ENTRY(s) {#s, s##_dims, s##_dyn, s##_dHx, s##_aux, s##_fin}
static const model_t MODELS[] = {ENTRY(pendulum), ENTRY(cartpole), ENTRY(robotarm), ENTRY(quadrotor), ENTRY(rocket)};

/* C = A(ra x ca) * B(ca x cb), row-major; ta / tb transpose the operand */
static void mm(int ra, int ca, int cb, const double* A, int ta, const double* B, int tb, double* C) {
    for (int i = 0; i < ra; ++i)
        for (int j = 0; j < cb; ++j) {
            double s = 0.0;
            for (int k = 0; k < ca; ++k) s += (ta ? A[k * ra + i] : A[i * ca + k]) * (tb ? B[j * ca + k] : B[k * cb + j]);
            C[i * cb + j] = s;
        }
}
/* in-place inverse by LU with partial pivoting (what numpy.linalg.inv / LAPACK getrf+getri do); returns 0 if singular */
static int inv(int n, double* A, double* w /* n*n */) {
    int piv[32];
    for (int i = 0; i < n * n; ++i) w[i] = 0.0;
    for (int i = 0; i < n; ++i) w[i * n + i] = 1.0;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (fabs(A[i * n + k]) > fabs(A[p * n + k])) p = i;
        piv[k] = p;
        if (A[p * n + k] == 0.0) return 0;
        if (p != k) for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; t = w[k * n + j]; w[k * n + j] = w[p * n + j]; w[p * n + j] = t; }
        double d = 1.0 / A[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            double f = A[i * n + k] * d;
            if (f == 0.0) continue;
            for (int j = k; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
            for (int j = 0; j < n; ++j) w[i * n + j] -= f * w[k * n + j];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double d = 1.0 / A[k * n + k];
        for (int j = 0; j < n; ++j) w[k * n + j] *= d;
        for (int i = 0; i < k; ++i) {
            double f = A[i * n + k];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) w[i * n + j] -= f * w[k * n + j];
        }
    }
    memcpy(A, w, sizeof(double) * n * n);
    (void)piv;
    return 1;
}

static void one_traj(const model_t* M, int T, int given, const double* x0, const double* u, const double* th, const double* dx, const double* du,
                     double* xs, double* ls, double* loss, double* grad, double* Xo, double* Uo, double* ws) {
    const int n = M->dims[0], m = M->dims[1], p = M->dims[2];
    const int sF = n * n, sG = n * m, sE = n * p, sHuu = m * m, sHue = m * p;
    const int per = 2 * sF + 2 * sG + 2 * sE + sHuu + sHue;
    double* aux = ws;                                   /* T * per */
    double* PP = aux + (size_t)T * per;                 /* T * n*n */
    double* WW = PP + (size_t)T * sF;                   /* T * n*p */
    double* X = WW + (size_t)T * sE;                    /* (T+1) * n*p */
    double* U = X + (size_t)(T + 1) * sE;               /* T * m*p */
    double* tmp = U + (size_t)T * sHue;                 /* scratch: 12 * n*(n+p+m) */
    if (!given) {
        memcpy(xs, x0, sizeof(double) * n);
        for (int t = 0; t < T; ++t) M->dyn(xs + t * n, u + t * m, th, xs + (t + 1) * n);
        double hxx[32 * 32], hxe[32 * 64];
        M->fin(xs + T * n, th, ls + (T - 1) * n, hxx, hxe);                         /* lam[T-1] = h_x(x_T)      PDP.py:204 */
        for (int k = T - 1; k >= 1; --k) M->dHx(xs + k * n, u + k * m, ls + k * n, th, ls + (k - 1) * n);   /* PDP.py:205-209 */
    }
    /* getAuxSys, PDP.py:287-301 */
    for (int t = 0; t < T; ++t) {
        double* a = aux + (size_t)t * per;
        M->aux(xs + t * n, u + t * m, ls + t * n, th, a, a + sF, a + sF + sG, a + sF + sG + sE, a + 2 * sF + sG + sE, a + 2 * sF + 2 * sG + sE,
               a + 2 * sF + 2 * sG + 2 * sE, a + 2 * sF + 2 * sG + 2 * sE + sHuu);
    }
    double dh[32];
    M->fin(xs + T * n, th, dh, PP + (size_t)(T - 1) * sF, WW + (size_t)(T - 1) * sE);    /* PP[-1] = hxx, WW[-1] = hxe   PDP.py:561-562 */
    double *iH = tmp, *GiH = iH + sHuu, *XiH = GiH + sG, *A = XiH + sG, *R = A + sF, *Mt = R + sF, *Q = Mt + sE, *N = Q + sF, *IPR = N + sE,
           *w1 = IPR + sF, *w2 = w1 + sF + sE, *w3 = w2 + sF + sE, *S = w3 + sF + sE;
#define AUX(t) const double *F = aux + (size_t)(t)*per, *G = F + sF, *E = G + sG, *Hxx = E + sE, *Hxu = Hxx + sF, *Hxe = Hxu + sG, *Huu = Hxe + sE, *Hue = Huu + sHuu
    for (int t = T - 1; t >= 1; --t) {                                         /* PDP.py:563-580 */
        AUX(t);
        const double *P = PP + (size_t)t * sF, *W = WW + (size_t)t * sE;
        memcpy(iH, Huu, sizeof(double) * sHuu); inv(m, iH, w1);
        mm(n, m, m, G, 0, iH, 0, GiH);
        mm(n, m, m, Hxu, 0, iH, 0, XiH);
        mm(n, m, n, GiH, 0, Hxu, 1, A); for (int i = 0; i < sF; ++i) A[i] = F[i] - A[i];
        mm(n, m, n, GiH, 0, G, 1, R);
        mm(n, m, p, GiH, 0, Hue, 0, Mt); for (int i = 0; i < sE; ++i) Mt[i] = E[i] - Mt[i];
        mm(n, m, n, XiH, 0, Hxu, 1, Q); for (int i = 0; i < sF; ++i) Q[i] = Hxx[i] - Q[i];
        mm(n, m, p, XiH, 0, Hue, 0, N); for (int i = 0; i < sE; ++i) N[i] = Hxe[i] - N[i];
        mm(n, n, n, P, 0, R, 0, IPR); for (int i = 0; i < n; ++i) IPR[i * n + i] += 1.0;
        inv(n, IPR, w1);
        mm(n, n, n, A, 1, IPR, 0, S);                                          /* temp_mat = A'(I+PR)^-1 */
        mm(n, n, n, P, 0, A, 0, w1); mm(n, n, n, S, 0, w1, 0, w2);
        double* Pc = PP + (size_t)(t - 1) * sF; for (int i = 0; i < sF; ++i) Pc[i] = Q[i] + w2[i];
        mm(n, n, p, P, 0, Mt, 0, w1); for (int i = 0; i < sE; ++i) w1[i] += W[i];
        mm(n, n, p, S, 0, w1, 0, w2);
        double* Wc = WW + (size_t)(t - 1) * sE; for (int i = 0; i < sE; ++i) Wc[i] = N[i] + w2[i];
    }
    memset(X, 0, sizeof(double) * sE);                                         /* ini_state = zeros(n,p) */
    for (int t = 0; t < T; ++t) {                                              /* PDP.py:588-608 */
        AUX(t);
        const double *P = PP + (size_t)t * sF, *W = WW + (size_t)t * sE;
        const double* xt = X + (size_t)t * sE;
        memcpy(iH, Huu, sizeof(double) * sHuu); inv(m, iH, w1);
        mm(n, m, m, G, 0, iH, 0, GiH);
        mm(n, m, n, GiH, 0, Hxu, 1, A); for (int i = 0; i < sF; ++i) A[i] = F[i] - A[i];
        mm(n, m, p, GiH, 0, Hue, 0, Mt); for (int i = 0; i < sE; ++i) Mt[i] = E[i] - Mt[i];
        mm(n, m, n, GiH, 0, G, 1, R);
        mm(n, n, n, P, 0, R, 0, IPR); for (int i = 0; i < n; ++i) IPR[i * n + i] += 1.0;
        inv(n, IPR, w1);
        /* u = -iH (Hxu' x + Hue) - iH G' (I+PR)^-1 (P A x + P M + W) */
        mm(n, n, n, P, 0, A, 0, w1); mm(n, n, p, w1, 0, xt, 0, w2);
        mm(n, n, p, P, 0, Mt, 0, w1); for (int i = 0; i < sE; ++i) w2[i] += w1[i] + W[i];
        mm(n, n, p, IPR, 0, w2, 0, w1);
        mm(m, n, p, G, 1, w1, 0, w2);                                          /* G' (..)  (m x p) */
        mm(m, n, p, Hxu, 1, xt, 0, w3); for (int i = 0; i < sHue; ++i) w3[i] += Hue[i] + w2[i];
        double* ut = U + (size_t)t * sHue;
        mm(m, m, p, iH, 0, w3, 0, ut); for (int i = 0; i < sHue; ++i) ut[i] = -ut[i];
        double* xn = X + (size_t)(t + 1) * sE;
        mm(n, n, p, F, 0, xt, 0, xn); mm(n, m, p, G, 0, ut, 0, w1); for (int i = 0; i < sE; ++i) xn[i] += w1[i] + E[i];
    }
    /* loss and chain rule, cartpole_PDP.py:63-74 */
    double L = 0.0;
    for (int j = 0; j < p; ++j) grad[j] = 0.0;
    for (int t = 0; t <= T; ++t)
        for (int i = 0; i < n; ++i) {
            double d = xs[t * n + i] - dx[t * n + i];
            L += d * d;
            for (int j = 0; j < p; ++j) grad[j] += d * X[(size_t)t * sE + i * p + j];
        }
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < m; ++i) {
            double d = u[t * m + i] - du[t * m + i];
            L += d * d;
            for (int j = 0; j < p; ++j) grad[j] += d * U[(size_t)t * sHue + i * p + j];
        }
    *loss = L;
    if (Xo) memcpy(Xo, X, sizeof(double) * (size_t)(T + 1) * sE);
    if (Uo) memcpy(Uo, U, sizeof(double) * (size_t)T * sHue);
}

int pdp_oracle_n_models(void) { return (int)(sizeof(MODELS) / sizeof(MODELS[0])); }
const char* pdp_oracle_model_name(int i) { return MODELS[i].name; }
void pdp_oracle_model_dims(int i, int* d) { d[0] = MODELS[i].dims[0]; d[1] = MODELS[i].dims[1]; d[2] = MODELS[i].dims[2]; }

/* The U-OC unit for a batch.  theta_bstride = 0 shares theta.  given != 0: xs / ls are inputs (optimal trajectory).
 * xs [B][T+1][n], ls [B][T][n] in/out; X [B][T+1][n][p], U [B][T][m][p] optional outputs.  Returns 0, or -1 on bad model id. */
int pdp_oracle_oc_unit(int model, int B, int T, int given, const double* x0, const double* u, const double* theta, int theta_bstride,
                       const double* demo_x, const double* demo_u, double* xs, double* ls, double* loss, double* grad, double* X, double* U,
                       int threads) {
    if (model < 0 || model >= pdp_oracle_n_models()) return -1;
    const model_t* M = &MODELS[model];
    const int n = M->dims[0], m = M->dims[1], p = M->dims[2];
    const size_t per = 2 * n * n + 2 * n * m + 2 * n * p + m * m + m * p;
    const size_t wsn = (size_t)T * per + (size_t)T * n * n + (size_t)T * n * p + (size_t)(T + 1) * n * p + (size_t)T * m * p + 16 * (size_t)n * (n + p + m) + 64;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        double* ws = (double*)malloc(sizeof(double) * wsn);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            one_traj(M, T, given, x0 ? x0 + (size_t)b * n : 0, u + (size_t)b * T * m, theta + (size_t)b * theta_bstride, demo_x + (size_t)b * (T + 1) * n,
                     demo_u + (size_t)b * T * m, xs + (size_t)b * (T + 1) * n, ls + (size_t)b * T * n, loss + b, grad + (size_t)b * p,
                     X ? X + (size_t)b * (T + 1) * n * p : 0, U ? U + (size_t)b * T * m * p : 0, ws);
        }
        free(ws);
    }
    return 0;
}
