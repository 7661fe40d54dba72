/*
 * cbfqp_oracle.c -- plain C (double) restatement of the reference CBF-QP path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ as the
 * checker at sizes the numpy oracle is too slow for, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg (kind "port").
 * Never linked into, loaded by or called from the product library.
 *
 * Follows, per agent:
 *   robots/dynamic_unicycle2D.py:42-73,121-186   f, g, agent_barrier (circle, superellipsoid)
 *   robots/kinematic_bicycle2D.py:67-111,160-173 f, g(x), agent_barrier
 *   dynamic_env/kinematic_bicycle2D_c3bf.py:15-75, ..._dpcbf.py:16-84
 *   position_control/cbf_qp.py:108-199           row assembly + solve + status
 * The QP (cbf_qp.py:190, cvxpy -> GUROBI in the reference) is solved by exact
 * active-set enumeration, the same algorithm as oracle/qp.py; it is validated
 * against oracle/qp.py and the golden vectors in tests/test_oracle_c.py.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KMAX 64

enum { DU = 0, KB = 1, C3BF = 2, DPCBF = 3, SI = 4, DI = 5, QUAD2D = 6, UNI = 7 };

typedef struct {
    int model, cbf_mode;
    double R, dt, a1, a2, lo[2], hi[2], Lr, mass;
} par_t;

static void hocbf_circle(const double* X, const double* o, double R, double beta, double* h, double* hdot,
                         double* dhd) {
    double th = X[2], v = X[3], c = cos(th), s = sin(th);
    double ex = X[0] - o[0], ey = X[1] - o[1], dmin = o[2] + R;
    double nrm = sqrt(ex * ex + ey * ey);
    *h = nrm * nrm - beta * dmin * dmin;
    *hdot = 2.0 * (ex * (v * c) + ey * (v * s));
    dhd[0] = 2.0 * v * c;
    dhd[1] = 2.0 * v * s;
    dhd[2] = 2.0 * (ex * (-v * s) + ey * (v * c));
    dhd[3] = 2.0 * (ex * c + ey * s);
}

static void hocbf_super(const double* X, const double* o, double R, double* h, double* hdot, double* dhd) {
    double th = X[2], v = X[3], c = cos(th), s = sin(th);
    double a = o[2] + R, b = o[3] + R, e = o[4], ct = cos(o[5]), st = sin(o[5]);
    double px = ct * (X[0] - o[0]) + st * (X[1] - o[1]);
    double py = -st * (X[0] - o[0]) + ct * (X[1] - o[1]);
    *h = pow(px / a, e) + pow(py / b, e) - 1.0;
    double gx = e * pow(px, e - 1) / pow(a, e), gy = e * pow(py, e - 1) / pow(b, e);
    double dhx = gx * ct - gy * st, dhy = gx * st + gy * ct;
    *hdot = dhx * v * c + dhy * v * s;
    double ca = e * (e - 1) / pow(a, e) * pow(px, e - 2), cb = e * (e - 1) / pow(b, e) * pow(py, e - 2);
    double hxx = ca * ct * ct + cb * st * st, hxy = (ca - cb) * ct * st, hyy = ca * st * st + cb * ct * ct;
    dhd[0] = hxx * v * c + hxy * v * s;
    dhd[1] = hxy * v * c + hyy * v * s;
    dhd[2] = dhx * (-v * s) + dhy * (v * c);
    dhd[3] = dhx * c + dhy * s;
}

static void c3bf(const double* X, const double* o, double R, double* h, double* dh) {
    double th = X[2], v = X[3], c = cos(th), s = sin(th), ovx = o[3], ovy = o[4];
    double ego = (o[2] + R) * 1.0, px = o[0] - X[0], py = o[1] - X[1];
    double vx = ovx - v * c, vy = ovy - v * s;
    double pm = sqrt(px * px + py * py), vm = sqrt(vx * vx + vy * vy), eps = 1e-6;
    double sq = sqrt(fmax(pm * pm - ego * ego, eps)), cphi = sq / (pm + eps);
    *h = (px * vx + py * vy) + pm * vm * cphi;
    double k = (sq + eps) / vm;
    dh[0] = -vx - vm * px / (sq + eps);
    dh[1] = -vy - vm * py / (sq + eps);
    dh[2] = v * s * px - v * c * py + k * (v * (ovx * s - ovy * c));
    dh[3] = -c * px - s * py + k * (v - (ovx * c + ovy * s));
}

static void dpcbf(const double* X, const double* o, double R, double* h, double* dh) {
    const double kl = 0.1, km = 0.5, sm = 1.05;
    double th = X[2], v = X[3], c = cos(th), s = sin(th), ovx = o[3], ovy = o[4];
    double ego = (o[2] + R) * sm, px = o[0] - X[0], py = o[1] - X[1];
    double vx = ovx - v * c, vy = ovy - v * s;
    double pm = sqrt(px * px + py * py), vm = sqrt(vx * vx + vy * vy);
    double rot = atan2(py, px), cr = cos(rot), sr = sin(rot);
    double vnx = cr * vx + sr * vy, vny = -sr * vx + cr * vy;
    double sd = sqrt(fmax(pm * pm - ego * ego, 1e-6)), pm2 = pm * pm;
    double lam = kl * sd / vm * sqrt(sm * sm - 1) / ego, mu = km * sd * sqrt(sm * sm - 1) / ego;
    *h = vnx + lam * vny * vny + mu;
    dh[0] = py * vny / pm2 - kl * px * vny * vny / vm / sd - 2 * kl * sd / vm * vny * py / pm2 * vnx - km * px / sd;
    dh[1] = -px * vny / pm2 - kl * py * vny * vny / vm / sd + 2 * kl * sd / vm * vny * px / pm2 * vnx - km * py / sd;
    dh[2] = -v * sin(rot - th) - kl * sd * v * (ovx * s - ovy * c) * vny * vny / (vm * vm * vm)
            - 2 * kl * sd * vny * v * cos(rot - th) / vm;
    dh[3] = -cos(rot - th) - kl * sd / (vm * vm * vm) * (v - ovx * c - ovy * s) * vny * vny
            - 2 * kl * sd * vny * sin(rot - th) / vm;
}

/* superellipsoid pieces shared by the integrator models (robots/single_integrator2D.py:131-147,
   robots/double_integrator2D.py:187-218) */
static void super_terms(const double* X, const double* o, double R, double* h, double* dhx, double* dhy, double* hxx,
                        double* hxy, double* hyy) {
    double a = o[2] + R, b = o[3] + R, e = o[4], ct = cos(o[5]), st = sin(o[5]);
    double px = ct * (X[0] - o[0]) + st * (X[1] - o[1]);
    double py = -st * (X[0] - o[0]) + ct * (X[1] - o[1]);
    *h = pow(px / a, e) + pow(py / b, e) - 1.0;
    double gx = e * pow(px, e - 1) / pow(a, e), gy = e * pow(py, e - 1) / pow(b, e);
    *dhx = gx * ct - gy * st; *dhy = gx * st + gy * ct;
    double ca = e * (e - 1) / pow(a, e) * pow(px, e - 2), cb = e * (e - 1) / pow(b, e) * pow(py, e - 2);
    *hxx = ca * ct * ct + cb * st * st; *hxy = (ca - cb) * ct * st; *hyy = ca * st * st + cb * ct * ct;
}

/* one row; returns 0 on bad obstacle flag */
static int cbf_row(const par_t* p, const double* X, const double* o, double* n, double* c, double* h) {
    if (p->model == QUAD2D) { /* robots/quad2D.py:46-81,166-177: X = [x, z, th, vx, vz, thdot] */
        double ex = X[0] - o[0], ez = X[1] - o[1], dmin = o[2] + p->R, nrm = sqrt(ex * ex + ez * ez);
        double vx = X[3], vz = X[4], sn = sin(X[2]), cs = cos(X[2]);
        *h = nrm * nrm - 1.01 * dmin * dmin;
        double hdot = 2.0 * (ex * vx + ez * vz);
        double a = 2.0 * ex * (-sn / p->mass) + 2.0 * ez * (cs / p->mass);
        n[0] = a; n[1] = a;
        double Lf = 2.0 * vx * vx + 2.0 * vz * vz + 2.0 * ez * (-9.81);
        *c = p->cbf_mode ? (*h / (p->dt * p->dt) + 2.0 * hdot / p->dt + Lf)
                         : (Lf + (p->a1 + p->a2) * hdot + (p->a1 * p->a2) * (*h));
        return 1;
    }
    if (p->model == UNI) { /* robots/unicycle2D.py:52-62,100-128: rel-deg 1, sigma(s) = k2 tanh((k1 - s) / 2) */
        double ex = X[0] - o[0], ey = X[1] - o[1], dmin = o[2] + p->R, nrm = sqrt(ex * ex + ey * ey);
        double cs = cos(X[2]), sn = sin(X[2]), s = ex * cs + ey * sn, th = tanh(0.5 * (0.5 - s));
        double dsig = -0.5 * 1.8 * (1.0 - th * th);
        *h = nrm * nrm - 1.01 * dmin * dmin - 1.8 * th;
        double dhx = 2.0 * ex - dsig * cs, dhy = 2.0 * ey - dsig * sn, dht = -dsig * (-sn * ex + cs * ey);
        n[0] = dhx * cs + dhy * sn; n[1] = dht;                 /* dh_dx g, g = [[c, 0], [s, 0], [0, 1]] */
        *c = p->cbf_mode ? (*h / p->dt) : (p->a1 * (*h));      /* f = 0 */
        return 1;
    }
    if (p->model == SI || p->model == DI) {
        double hx, dhx, dhy, hxx = 0, hxy = 0, hyy = 0;
        if (o[6] == 0.0) {
            double ex = X[0] - o[0], ey = X[1] - o[1], dmin = o[2] + p->R, nrm = sqrt(ex * ex + ey * ey);
            hx = nrm * nrm - 1.01 * dmin * dmin; dhx = 2.0 * ex; dhy = 2.0 * ey; hxx = 2.0; hyy = 2.0;
        } else if (o[6] == 1.0) super_terms(X, o, p->R, &hx, &dhx, &dhy, &hxx, &hxy, &hyy);
        else return 0;
        *h = hx;
        n[0] = dhx; n[1] = dhy;
        if (p->model == SI) {
            *c = p->cbf_mode ? (hx / p->dt) : (p->a1 * hx);
        } else {
            double vx = X[2], vy = X[3], hdot = dhx * vx + dhy * vy;
            double Lf = (hxx * vx + hxy * vy) * vx + (hxy * vx + hyy * vy) * vy;
            *c = p->cbf_mode ? (hx / (p->dt * p->dt) + 2.0 * hdot / p->dt + Lf)
                             : (Lf + (p->a1 + p->a2) * hdot + (p->a1 * p->a2) * hx);
        }
        return 1;
    }
    double th = X[2], v = X[3], cs = cos(th), sn = sin(th), f0 = v * cs, f1 = v * sn;
    double g[4][2] = {{0, 0}, {0, 0}, {0, 1}, {1, 0}};
    if (p->model != DU) {
        g[0][1] = -v * sn; g[1][1] = v * cs; g[2][1] = v / p->Lr;
    }
    double d[4], hdot = 0.0;
    int rel2 = (p->model == DU || p->model == KB);
    if (p->model == DU) {
        if (o[6] == 0.0) hocbf_circle(X, o, p->R, 1.01, h, &hdot, d);
        else if (o[6] == 1.0) hocbf_super(X, o, p->R, h, &hdot, d);
        else return 0;
    } else if (p->model == KB) hocbf_circle(X, o, p->R, 1.1, h, &hdot, d);
    else if (p->model == C3BF) c3bf(X, o, p->R, h, d);
    else dpcbf(X, o, p->R, h, d);
    for (int j = 0; j < 2; ++j) n[j] = d[0] * g[0][j] + d[1] * g[1][j] + d[2] * g[2][j] + d[3] * g[3][j];
    double Lf = d[0] * f0 + d[1] * f1;
    if (rel2) *c = p->cbf_mode ? (*h / (p->dt * p->dt) + 2.0 * hdot / p->dt + Lf)
                               : (Lf + (p->a1 + p->a2) * hdot + (p->a1 * p->a2) * (*h));
    else *c = p->cbf_mode ? (*h / p->dt + Lf) : (Lf + p->a1 * (*h));
    return 1;
}

static int feasible(int m, const double (*G)[2], const double* c, const double* u, double tol) {
    for (int i = 0; i < m; ++i) {
        double r = G[i][0] * u[0] + G[i][1] * u[1] + c[i];
        double sc = fmax(1.0, fabs(G[i][0]) * fabs(u[0]) + fabs(G[i][1]) * fabs(u[1]) + fabs(c[i]));
        if (!(r >= -tol * sc)) return 0;
    }
    return 1;
}

/* exact enumeration, mirrors oracle/qp.py:solve_qp2 */
static int solve_qp2(int m, const double (*G)[2], const double* c, const double* ur, double* u_out) {
    const double tol = 1e-9;
    for (int i = 0; i < m; ++i)
        if (!isfinite(G[i][0]) || !isfinite(G[i][1]) || !isfinite(c[i])) return 1;
    if (!isfinite(ur[0]) || !isfinite(ur[1])) return 1;
    double best = INFINITY, bu[2] = {0, 0};
    int found = 0;
    double n2[KMAX + 4];
#define CONSIDER(U)                                                                     \
    do {                                                                                \
        double cost_ = ((U)[0] - ur[0]) * ((U)[0] - ur[0]) + ((U)[1] - ur[1]) * ((U)[1] - ur[1]); \
        if (cost_ < best && feasible(m, G, c, (U), tol)) { best = cost_; bu[0] = (U)[0]; bu[1] = (U)[1]; found = 1; } \
    } while (0)
    CONSIDER(ur);
    for (int i = 0; i < m; ++i) {
        n2[i] = G[i][0] * G[i][0] + G[i][1] * G[i][1];
        if (n2[i] <= 0.0) continue;
        double lam = (G[i][0] * ur[0] + G[i][1] * ur[1] + c[i]) / n2[i];
        double u[2] = {ur[0] - lam * G[i][0], ur[1] - lam * G[i][1]};
        CONSIDER(u);
    }
    for (int i = 0; i < m; ++i)
        for (int j = i + 1; j < m; ++j) {
            double det = G[i][0] * G[j][1] - G[i][1] * G[j][0];
            if (fabs(det) <= 1e-14 * sqrt(n2[i] * n2[j])) continue;
            double u[2] = {(-c[i] * G[j][1] + c[j] * G[i][1]) / det, (-c[j] * G[i][0] + c[i] * G[j][0]) / det};
            CONSIDER(u);
        }
#undef CONSIDER
    if (!found) return 1;
    u_out[0] = bu[0]; u_out[1] = bu[1];
    return 0;
}

/* status: 0 optimal, 1 infeasible, 3 bad obstacle flag.  u_out NaN unless optimal. */
int oracle_cbfqp_batch(int model, long B, int K, const double* X, const double* u_ref, const double* obs,
                       int obs_shared, const int* n_obs, double radius, double dt, double alpha1, double alpha2,
                       const double* u_min, const double* u_max, double rear_ax_dist, int cbf_mode,
                       double* u_out, int* status, double* h_out, int n_threads, int nx, double mass) {
    if (K < 1 || K > KMAX) return 1;
    if (nx < 4) nx = 4;
    par_t p = {model, cbf_mode, radius, dt, alpha1, alpha2, {u_min[0], u_min[1]}, {u_max[0], u_max[1]}, rear_ax_dist, mass};
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(static)
#endif
    for (long i = 0; i < B; ++i) {
        double G[KMAX + 4][2], c[KMAX + 4], h;
        const double* o = obs_shared ? obs : obs + (size_t)i * K * 7;
        int nk = n_obs ? n_obs[i] : K, bad = 0;
        if (nk > K) nk = K;
        for (int r = 0; r < K; ++r) {
            G[r][0] = G[r][1] = c[r] = 0.0;
            h = 0.0;
            if (r < nk && !cbf_row(&p, X + (size_t)nx * i, o + 7 * r, G[r], &c[r], &h)) bad = 1;
            if (h_out) h_out[(size_t)i * K + r] = (r < nk) ? h : 0.0;
        }
        G[K][0] = 1; G[K][1] = 0; c[K] = -p.lo[0];
        G[K + 1][0] = -1; G[K + 1][1] = 0; c[K + 1] = p.hi[0];
        G[K + 2][0] = 0; G[K + 2][1] = 1; c[K + 2] = -p.lo[1];
        G[K + 3][0] = 0; G[K + 3][1] = -1; c[K + 3] = p.hi[1];
        double u[2] = {NAN, NAN};
        int st = bad ? 3 : solve_qp2(K + 4, (const double (*)[2])G, c, u_ref + 2 * i, u);
        if (st != 0) u[0] = u[1] = NAN;
        u_out[2 * i] = u[0]; u_out[2 * i + 1] = u[1];
        status[i] = st;
    }
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
