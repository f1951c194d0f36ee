// Small dense LU with partial pivoting (the role LAPACK dgetrf/dgetrs play in the reference:
// src/ParOptInteriorPoint.cpp:1969,2159,2664,2722 and src/ParOptQuasiNewton.cpp:375,406).
// Unblocked right-looking elimination, column-major, first-maximum pivot rule as dgetf2/idamax.
// Matrices here are at most 96 x 96 and replicated on every rank.
#include <math.h>

#include "core.hpp"

namespace po {

int lu_factor(int n, double *A, int lda, int *piv) {
  int info = 0;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = fabs(A[k + (size_t)lda * k]);
    for (int i = k + 1; i < n; i++) {
      const double v = fabs(A[i + (size_t)lda * k]);
      if (v > best) {
        best = v;
        p = i;
      }
    }
    piv[k] = p;
    if (A[p + (size_t)lda * k] != 0.0) {
      if (p != k) {
        for (int j = 0; j < n; j++) {
          const double tmp = A[k + (size_t)lda * j];
          A[k + (size_t)lda * j] = A[p + (size_t)lda * j];
          A[p + (size_t)lda * j] = tmp;
        }
      }
      const double inv = 1.0 / A[k + (size_t)lda * k];
      for (int i = k + 1; i < n; i++) A[i + (size_t)lda * k] *= inv;
    } else if (info == 0) {
      info = k + 1;
    }
    for (int j = k + 1; j < n; j++) {
      const double akj = A[k + (size_t)lda * j];
      if (akj != 0.0) {
        for (int i = k + 1; i < n; i++) A[i + (size_t)lda * j] -= A[i + (size_t)lda * k] * akj;
      }
    }
  }
  return info;
}

void lu_solve(int n, const double *A, int lda, const int *piv, double *b) {
  for (int k = 0; k < n; k++) {
    const int p = piv[k];
    if (p != k) {
      const double tmp = b[k];
      b[k] = b[p];
      b[p] = tmp;
    }
  }
  for (int k = 0; k < n; k++) {  // L y = P b (unit lower)
    const double bk = b[k];
    if (bk != 0.0) {
      for (int i = k + 1; i < n; i++) b[i] -= A[i + (size_t)lda * k] * bk;
    }
  }
  for (int k = n - 1; k >= 0; k--) {  // U x = y
    b[k] /= A[k + (size_t)lda * k];
    const double bk = b[k];
    for (int i = 0; i < k; i++) b[i] -= A[i + (size_t)lda * k] * bk;
  }
}

}  // namespace po
