// tridiag_host.hpp -- host-side arithmetic on the Lanczos tridiagonal: QL eigen-solver, the lowest eigenvalue from a Sturm sequence, and the
// stopping rule of sp_lanc_eigh built on both.  No device code: tests/test_stoprule_cpu.py compiles this header alone with g++.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace edgpu {

// ---------------------------------------------------------------------------------------
// symmetric tridiagonal eigen-solver (implicit QL with Wilkinson shifts), host, n <= ~1000.
// d: diagonal (n), e: sub-diagonal e[1..n-1] (e[0] unused).  On exit d = eigenvalues
// (unsorted); if z != nullptr (n*n, identity on entry) it returns the vectors TRANSPOSED: z[j*n + i] = component i of vector j
// (the rotation of two vectors then runs over two contiguous rows; same arithmetic as the textbook column form, ~8x faster at n = 80).
// ---------------------------------------------------------------------------------------
static bool tqli(std::vector<double> &d, std::vector<double> &e, int n, double *z)
{
    for (int i = 1; i < n; i++) e[i - 1] = e[i];
    if (n > 0) e[n - 1] = 0.0;
    for (int l = 0; l < n; l++) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; m++) {
                double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 120) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double sn = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; i--) {
                    double f = sn * e[i], b = c * e[i];
                    e[i + 1] = (r = std::hypot(f, g));
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    sn = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * sn + 2.0 * c * b;
                    d[i + 1] = g + (p = sn * r);
                    g = c * r - b;
                    if (z) {
                        double *zi = z + (size_t)i * n, *zi1 = zi + n;
                        for (int k = 0; k < n; k++) {
                            const double f2 = zi1[k];
                            zi1[k] = sn * zi[k] + c * f2;
                            zi[k] = c * zi[k] - sn * f2;
                        }
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// Dense symmetric eigenproblem of the projected matrices (thick-restart Lanczos: Ritz block + arrow + tridiagonal tail, 20-96 rows;
// one per restart): Householder reduction to tridiagonal form with the reflectors accumulated, then the QL sweep above on the
// accumulated basis.  A: m x m row-major symmetric (destroyed).  w: eigenvalues ascending; Y[i + m*j] = component i of vector j.
// Returns false when QL does not converge (the caller falls back to the Jacobi sweeps, which cannot fail).  Against the cyclic
// Jacobi it replaces: 26 against 73 us at m = 20 (the size every launch-bound sector solves 10-14 times per search), 88 / 255 us at
// m = 32, 0.9 / 5.1 ms at m = 80 (one host core of the build container).
// ---------------------------------------------------------------------------------------
static bool sym_eig_householder(std::vector<double> &A, int m, std::vector<double> &w, std::vector<double> &Y)
{
    if (m < 1) { w.clear(); Y.clear(); return true; }
    // zt = Q^T (row i = column i of Q), Q = H_0 H_1 ... : A = Q T Q^T
    std::vector<double> zt((size_t)m * m, 0.0), v((size_t)m), pvec((size_t)m);
    for (int i = 0; i < m; i++) zt[(size_t)i * m + i] = 1.0;
    for (int k = 0; k + 2 < m; k++) {
        // reflector that maps x = A[k+1.., k] onto alpha e_1
        double xn2 = 0.0; for (int i = k + 1; i < m; i++) xn2 += A[(size_t)i * m + k] * A[(size_t)i * m + k];
        const double x0 = A[(size_t)(k + 1) * m + k];
        if (!(xn2 - x0 * x0 > 0.0)) continue;                     // nothing below the sub-diagonal: already tridiagonal here
        const double alpha = x0 > 0.0 ? -std::sqrt(xn2) : std::sqrt(xn2);
        double vv = 0.0;
        for (int i = k + 1; i < m; i++) { v[i] = A[(size_t)i * m + k]; if (i == k + 1) v[i] -= alpha; vv += v[i] * v[i]; }
        if (!(vv > 0.0)) continue;
        const double beta = 2.0 / vv;
        // p = beta A22 v ; K = beta/2 v.p ; q = p - K v ; A22 -= v q^T + q v^T
        double vp = 0.0;
        for (int i = k + 1; i < m; i++) { double a = 0.0; const double *Ai = &A[(size_t)i * m]; for (int j = k + 1; j < m; j++) a += Ai[j] * v[j]; pvec[i] = beta * a; vp += v[i] * pvec[i]; }
        const double K = 0.5 * beta * vp;
        for (int i = k + 1; i < m; i++) pvec[i] -= K * v[i];
        for (int i = k + 1; i < m; i++) { double *Ai = &A[(size_t)i * m]; const double vi = v[i], qi = pvec[i]; for (int j = k + 1; j < m; j++) Ai[j] -= vi * pvec[j] + qi * v[j]; }
        A[(size_t)(k + 1) * m + k] = A[(size_t)k * m + (k + 1)] = alpha;
        for (int i = k + 2; i < m; i++) A[(size_t)i * m + k] = A[(size_t)k * m + i] = 0.0;
        // Q <- Q H: columns k+1.. of Q, i.e. rows k+1.. of zt:  zt[j][r] -= v_j * beta * sum_j' v_j' zt[j'][r]
        for (int r = 0; r < m; r++) pvec[r] = 0.0;
        for (int j = k + 1; j < m; j++) { const double vj = v[j]; const double *zj = &zt[(size_t)j * m]; for (int r = 0; r < m; r++) pvec[r] += vj * zj[r]; }
        for (int j = k + 1; j < m; j++) { const double bv = beta * v[j]; double *zj = &zt[(size_t)j * m]; for (int r = 0; r < m; r++) zj[r] -= bv * pvec[r]; }
    }
    std::vector<double> d((size_t)m), e((size_t)m, 0.0);
    for (int i = 0; i < m; i++) { d[i] = A[(size_t)i * m + i]; if (i > 0) e[i] = A[(size_t)i * m + (i - 1)]; }
    if (!tqli(d, e, m, zt.data())) return false;
    std::vector<int> idx((size_t)m); for (int i = 0; i < m; i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] < d[b]; });
    w.resize(m); Y.assign((size_t)m * m, 0.0);
    for (int j = 0; j < m; j++) { w[j] = d[idx[j]]; const double *zj = &zt[(size_t)idx[j] * m]; for (int i = 0; i < m; i++) Y[i + (size_t)m * j] = zj[i]; }
    return true;
}

// lowest eigenvalue (and optionally its eigenvector) of T_k = tridiag(alphas[0..k), betas[1..k))
static bool tridiag_lowest(const double *al, const double *be, int k, double *e0, std::vector<double> *zcol)
{
    std::vector<double> d(al, al + k), e(k, 0.0);
    for (int i = 1; i < k; i++) e[i] = be[i];
    std::vector<double> z;
    if (zcol) { z.assign((size_t)k * k, 0.0); for (int i = 0; i < k; i++) z[(size_t)i * k + i] = 1.0; }
    if (!tqli(d, e, k, zcol ? z.data() : nullptr)) return false;
    int im = 0;
    for (int i = 1; i < k; i++) if (d[i] < d[im]) im = i;
    *e0 = d[im];
    if (zcol) { zcol->resize(k); for (int i = 0; i < k; i++) (*zcol)[i] = z[(size_t)im * k + i]; }
    return true;
}

// Lowest eigenvalue of T_k to a requested accuracy from the pivots of T_k - x (a Sturm sequence), O(k) per evaluation: the stopping rule of sp_lanc_eigh compares the lowest
// eigenvalues of T_k and T_{k-1} after EVERY iteration, and a full QL sweep per iteration (O(k^2), 60-150 us at k = 40-80) was most of the
// wall time of a ground-state search on a launch-bound sector (2-4.6 ms of 3-6 ms).  The loops below ask this estimate first and run the
// exact QL -- whose values decide, as before -- only when the estimate is within a wide margin of the threshold.
static double tridiag_lowest_estimate(const double *al, const double *be, int k, double upper, bool have_upper, double rel_tol, double *abs_err)
{
    double lo = al[0], hi = al[0];
    for (int i = 0; i < k; i++) {
        const double r = (i > 0 ? std::fabs(be[i]) : 0.0) + (i + 1 < k ? std::fabs(be[i + 1]) : 0.0);
        lo = std::min(lo, al[i] - r); hi = std::min(hi, al[i]);      // Gershgorin from below; the lowest eigenvalue is <= every diagonal entry
    }
    if (have_upper && upper < hi && upper > lo) hi = upper;          // interlacing: lowest(T_k) <= lowest(T_{k-1}) (the caller adds that value's error)
    if (!(lo <= hi)) { *abs_err = 0.0; return std::nan(""); }
    // last pivot q_{k-1}(x) of the LDL^T factorisation of T_k - x, valid while the earlier pivots are positive (x below the lowest eigenvalue
    // of T_{k-1}); there it is a decreasing, concave function of x whose only root is the lowest eigenvalue of T_k
    auto last_pivot = [&](double x, double &q) {
        q = al[0] - x;
        for (int i = 1; i < k; i++) {
            if (!(q > 0.0)) return false;
            q = al[i] - x - be[i] * be[i] / q;
        }
        return true;
    };
    // invariant: pivots valid and q > 0 at lo (no eigenvalue below lo); at hi an eigenvalue lies below (q <= 0 or an earlier pivot <= 0).
    // Regula falsi with the Illinois modification while both ends have a valid pivot, bisection otherwise: ~10 evaluations instead of ~45.
    double flo = 0.0, fhi = 0.0; bool hi_valid = false;
    if (!last_pivot(lo, flo) || !(flo > 0.0)) { *abs_err = 0.5 * (hi - lo); return 0.5 * (lo + hi); }      // Gershgorin guarantees it; rounding at most
    { double q; hi_valid = last_pivot(hi, q) ; fhi = q; if (hi_valid && q > 0.0) { *abs_err = 1e-14 * std::max(1.0, std::fabs(hi)); return hi; } }      // nothing below hi: hi is the bound itself
    int side = 0;
    auto width_ok = [&]() { return hi - lo <= rel_tol * std::max(1.0, std::max(std::fabs(lo), std::fabs(hi))); };
    auto place = [&](double x) {      // evaluate at x and shrink the bracket; returns which end moved (-1 lo, +1 hi)
        double q; const bool valid = last_pivot(x, q);
        if (valid && q > 0.0) { lo = x; flo = q; if (side == -1 && hi_valid) fhi *= 0.5; side = -1; return -1; }
        hi = x; hi_valid = valid; fhi = q; if (side == 1) flo *= 0.5; side = 1; return 1;
    };
    for (int it = 0; it < 200 && !width_ok(); it++) {
        double x = 0.5 * (lo + hi); bool secant = false;
        if (hi_valid) { const double xs = (lo * fhi - hi * flo) / (fhi - flo); if (xs > lo && xs < hi) { x = xs; secant = true; } }
        const int moved = place(x);
        if (secant && it >= 2 && !width_ok()) {      // the secant point is close to the root: probe a tolerance's width to the other side of it
            const double d = 0.9 * rel_tol * std::max(1.0, std::fabs(x));
            const double x2 = moved < 0 ? x + d : x - d;
            if (x2 > lo && x2 < hi) (void)place(x2);
        }
    }
    *abs_err = 0.5 * (hi - lo) + 1e-14 * std::max(1.0, std::max(std::fabs(lo), std::fabs(hi)));      // bracket + rounding of the recurrence
    return 0.5 * (lo + hi);
}
// The stopping rule of SciFortran's sp_lanc_eigh as the loops apply it after iteration nlanc-1 (T has nlanc rows): exact QL values decide;
// the estimates only say when they cannot yet -- a coarse one first (1e-7, half the bisection steps), refined when that is not decisive.
// Returns 1 stop, 0 continue, -1 QL failure.
struct StopRule {
    double esave = 0.0, est_prev = 0.0, err_prev = 0.0; bool esave_exact = true;      // esave: lowest eigenvalue of the previous T by QL (0 before the first one, as in the reference loop)
    int check(const double *al, const double *be, int nlanc, int ncheck, double threshold, double *e0)
    {
        double err = 0.0;
        double est = tridiag_lowest_estimate(al, be, nlanc, est_prev + err_prev, nlanc > 1, 1e-7, &err);
        // the QL values and the fine estimates agree to a few 1e-15 |T| (3.5e-15 at most over the matrices of tests/cpp/stoprule_check.cpp):
        // the two differences the rule could look at are within ~1e-14 |T| of each other; the margin is 20 times that
        const double margin = 2e-13 * std::max(1.0, std::fabs(est));
        bool near_rule = !(est == est) || !(std::fabs(est - est_prev) > threshold + margin + err + err_prev);      // NaN: let the exact path report it
        if (near_rule && nlanc >= ncheck && est == est) {      // not decisive at this accuracy: refine both values
            if (nlanc > 1 && err_prev > 1e-13 * std::max(1.0, std::fabs(est_prev))) est_prev = tridiag_lowest_estimate(al, be, nlanc - 1, 0.0, false, 4e-15, &err_prev);
            est = tridiag_lowest_estimate(al, be, nlanc, est_prev + err_prev, nlanc > 1, 4e-15, &err);
            near_rule = !(std::fabs(est - est_prev) > threshold + margin + err + err_prev);
        }
        est_prev = est; err_prev = err;
        if (nlanc >= ncheck && near_rule) {
            if (!esave_exact) { if (nlanc > 1) { if (!tridiag_lowest(al, be, nlanc - 1, &esave, nullptr)) return -1; } else esave = 0.0; }
            if (!tridiag_lowest(al, be, nlanc, e0, nullptr)) return -1;
            if (std::fabs(*e0 - esave) <= threshold) return 1;
            esave = *e0; esave_exact = true;
            return 0;
        }
        *e0 = est; esave_exact = false;
        return 0;
    }
};

}      // namespace edgpu
