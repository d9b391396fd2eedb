// Host-only check of sym_eig_householder (csrc/tridiag_host.hpp): Householder + QL against cyclic Jacobi sweeps on the matrix shapes the
// solvers produce (Ritz block + arrow + tridiagonal tail, dense projected matrices, degenerate and decoupled cases).  Built and run by
// tests/test_symeig_cpu.py.
#include <cstdio>
#include <random>
#include "../../dmft-lanc-ed_amd/csrc/tridiag_host.hpp"
using namespace edgpu;
static void jacobi(std::vector<double> A, int m, std::vector<double> &w)
{
    for (int sweep = 0; sweep < 80; sweep++) {
        double off = 0.0; for (int p = 0; p < m; p++) for (int q = p + 1; q < m; q++) off += A[p * m + q] * A[p * m + q];
        if (off < 1e-34) break;
        for (int p = 0; p < m; p++) for (int q = p + 1; q < m; q++) {
            const double apq = A[p * m + q]; if (std::fabs(apq) < 1e-300) continue;
            const double th = (A[q * m + q] - A[p * m + p]) / (2.0 * apq), t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < m; k++) { double a = A[k * m + p], b = A[k * m + q]; A[k * m + p] = c * a - sn * b; A[k * m + q] = sn * a + c * b; }
            for (int k = 0; k < m; k++) { double a = A[p * m + k], b = A[q * m + k]; A[p * m + k] = c * a - sn * b; A[q * m + k] = sn * a + c * b; }
        }
    }
    w.resize(m); for (int i = 0; i < m; i++) w[i] = A[i * m + i]; std::sort(w.begin(), w.end());
}
int main()
{
    std::mt19937 rng(11); std::uniform_real_distribution<double> U(-1, 1);
    double worst_val = 0, worst_res = 0, worst_orth = 0; int fails = 0, cases = 0;
    for (int trial = 0; trial < 400; trial++) {
        const int m = 1 + trial % 40 + (trial % 9 == 0 ? 56 : 0);
        std::vector<double> A((size_t)m * m, 0.0);
        const int kind = trial % 5;
        if (kind == 0) { for (int i = 0; i < m; i++) for (int j = i; j < m; j++) A[i * m + j] = A[j * m + i] = U(rng) * 3; }               // dense
        else if (kind == 1 || kind == 2) {      // thick-restart shape: diagonal block of l Ritz values, arrow to row l, tridiagonal tail
            const int l = m / 2;
            for (int i = 0; i < l; i++) { A[i * m + i] = -8 + 6.0 * i / (l + 1) + 0.01 * U(rng); if (l < m) A[i * m + l] = A[l * m + i] = (kind == 2 && i < 2 ? 1e-11 : 0.3) * U(rng); }
            for (int i = l; i < m; i++) { A[i * m + i] = -3 + 4 * U(rng); if (i + 1 < m) A[i * m + i + 1] = A[(i + 1) * m + i] = 0.5 + std::fabs(U(rng)); }
        }
        else if (kind == 3) { for (int i = 0; i < m; i++) A[i * m + i] = (i % 3) * 1.0; if (m > 3) A[0 * m + 3] = A[3 * m + 0] = 1e-9; }       // degenerate diagonal, one tiny coupling
        else { for (int i = 0; i < m; i++) { A[i * m + i] = U(rng); if (i + 1 < m) A[i * m + i + 1] = A[(i + 1) * m + i] = (i == m / 2 ? 0.0 : U(rng)); } }   // tridiagonal, decoupled in the middle
        std::vector<double> A0(A), w, Y, wj;
        cases++;
        if (!sym_eig_householder(A, m, w, Y)) { fails++; printf("trial %d m %d: QL did not converge\n", trial, m); continue; }
        jacobi(A0, m, wj);
        double nrmA = 0; for (double x : A0) nrmA = std::max(nrmA, std::fabs(x)); nrmA = std::max(nrmA * m, 1.0);
        for (int j = 0; j < m; j++) {
            worst_val = std::max(worst_val, std::fabs(w[j] - wj[j]) / nrmA);
            if (j && w[j] < w[j - 1]) { fails++; printf("trial %d: not ascending\n", trial); }
            double r2 = 0; for (int i = 0; i < m; i++) { double a = -w[j] * Y[i + (size_t)m * j]; for (int k = 0; k < m; k++) a += A0[i * m + k] * Y[k + (size_t)m * j]; r2 += a * a; }
            worst_res = std::max(worst_res, std::sqrt(r2) / nrmA);
            for (int j2 = 0; j2 <= j; j2++) { double d = 0; for (int i = 0; i < m; i++) d += Y[i + (size_t)m * j] * Y[i + (size_t)m * j2]; worst_orth = std::max(worst_orth, std::fabs(d - (j == j2 ? 1.0 : 0.0))); }
        }
    }
    printf("cases %d failures %d  worst |w - w_jacobi|/|A| %.3e  worst residual/|A| %.3e  worst |Y^T Y - 1| %.3e\n", cases, fails, worst_val, worst_res, worst_orth);
    return (fails || worst_val > 1e-14 || worst_res > 1e-14 || worst_orth > 1e-13) ? 1 : 0;
}
