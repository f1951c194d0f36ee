// Host-side C++ of the product under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only, no GPU, no HIP
// runtime): the in-repo LU (lu.cpp, the reference's dgetrf/dgetrs sites), the one-time symbolic analysis of the
// sparse Cholesky (csr.cpp::csr_analyse) and the option registry (options.cpp).  Built by
// `make -C paropt_amd/csrc sanitize`, run by tests/test_host_sanitize.py.  Exit code 0 = every check passed and no
// sanitizer report.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "core.hpp"
#include "csr.hpp"
#include "ip.hpp"

using namespace po;

static int fails = 0;
#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) {                                                   \
      fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      fails++;                                                       \
    }                                                                \
  } while (0)

static uint64_t rng_state = 88172645463325252ULL;
static double rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0);
}

static void test_lu() {
  for (int n : {1, 2, 3, 7, 20, 42, 64}) {
    std::vector<double> A((size_t)n * n), A0, b(n), x(n);
    for (double &v : A) v = rnd() - 0.5;
    for (int i = 0; i < n; i++) A[(size_t)i * (n + 1)] += (i % 3 == 0) ? 0.0 : 2.0;  // force row exchanges
    A0 = A;
    for (int i = 0; i < n; i++) x[i] = rnd();
    for (int i = 0; i < n; i++) {
      double s = 0.0;
      for (int j = 0; j < n; j++) s += A0[i + (size_t)n * j] * x[j];
      b[i] = s;
    }
    std::vector<int> piv(n);
    const int info = lu_factor(n, A.data(), n, piv.data());
    CHECK(info == 0);
    for (int i = 0; i < n; i++) CHECK(piv[i] >= i && piv[i] < n);
    lu_solve(n, A.data(), n, piv.data(), b.data());
    double err = 0.0;
    for (int i = 0; i < n; i++) err = fmax(err, fabs(b[i] - x[i]));
    CHECK(err < 1e-8);
  }
  // exactly singular: info = index + 1 of the zero pivot, nothing out of bounds, no trap
  std::vector<double> S = {1.0, 2.0, 2.0, 4.0};
  int piv[2];
  CHECK(lu_factor(2, S.data(), 2, piv) == 2);
  CHECK(lu_factor(0, nullptr, 1, nullptr) == 0);
}

static void analyse(int64_t n, const std::vector<int> &rowp, const std::vector<int> &cols, bool expect_ok) {
  CsrSymbolic sym;
  const int64_t w = (int64_t)rowp.size() - 1;
  const int rc = csr_analyse(n, w, rowp.data(), cols.data(), &sym);
  CHECK((rc == PO_OK) == expect_ok);
  if (rc != PO_OK) return;
  CHECK((int64_t)sym.perm.size() == w);
  std::vector<char> seen(w, 0);
  for (int64_t i = 0; i < w; i++) {
    CHECK(sym.perm[i] >= 0 && sym.perm[i] < w);
    if (sym.perm[i] >= 0 && sym.perm[i] < w) seen[sym.perm[i]] = 1;
  }
  for (int64_t i = 0; i < w; i++) CHECK(seen[i]);
  CHECK(sym.nnzL >= w && sym.nnzL >= sym.nnzS - 0 * w);
  CHECK((int64_t)sym.Lrowp.size() == w + 1 && sym.Lrowp[w] == sym.nnzL);
  for (int64_t i = 0; i < w; i++) {
    CHECK(sym.Lrowp[i + 1] > sym.Lrowp[i]);
    CHECK(sym.Lcols[sym.Lrowp[i + 1] - 1] == (int)i);  // the diagonal closes the row
    for (int q = sym.Lrowp[i]; q < sym.Lrowp[i + 1] - 1; q++) CHECK(sym.Lcols[q] < sym.Lcols[q + 1]);
  }
}

static void test_csr() {
  {  // chain, span 2 stride 1 (examples/rosenbrock/sparse_rosenbrock.cpp)
    const int n = 500;
    std::vector<int> rowp, cols;
    for (int i = 0; i + 1 < n; i++) {
      rowp.push_back((int)cols.size());
      cols.push_back(i);
      cols.push_back(i + 1);
    }
    rowp.push_back((int)cols.size());
    analyse(n, rowp, cols, true);
  }
  {  // 2-D grid of pairwise constraints (fronts), reversed column order inside the rows
    const int nx = 24, ny = 17;
    std::vector<int> rowp, cols;
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        if (i + 1 < nx) {
          rowp.push_back((int)cols.size());
          cols.push_back(j * nx + i + 1);
          cols.push_back(j * nx + i);
        }
        if (j + 1 < ny) {
          rowp.push_back((int)cols.size());
          cols.push_back((j + 1) * nx + i);
          cols.push_back(j * nx + i);
        }
      }
    rowp.push_back((int)cols.size());
    analyse(nx * ny, rowp, cols, true);
  }
  for (int trial = 0; trial < 20; trial++) {  // random patterns with empty and long rows
    const int n = 40 + (int)(rnd() * 200), w = 1 + (int)(rnd() * 120);
    std::vector<int> rowp, cols;
    for (int r = 0; r < w; r++) {
      rowp.push_back((int)cols.size());
      const int len = (r % 11 == 0) ? 0 : (r % 17 == 1 ? n / 2 : 1 + (int)(rnd() * 6));
      std::vector<char> used(n, 0);
      for (int k = 0; k < len; k++) {
        int c = (int)(rnd() * n);
        while (used[c]) c = (c + 1) % n;
        used[c] = 1;
        cols.push_back(c);
      }
    }
    rowp.push_back((int)cols.size());
    analyse(n, rowp, cols, true);
  }
  {  // malformed input is refused, not read out of bounds
    std::vector<int> rowp = {0, 2, 4}, cols = {0, 1, 1, 7};
    analyse(4, rowp, cols, false);  // column index 7 >= n
    std::vector<int> rowp2 = {0, 3, 2}, cols2 = {0, 1, 2};
    analyse(4, rowp2, cols2, false);  // decreasing row pointer
    std::vector<int> rowp3 = {0, 2}, cols3 = {1, 1};
    analyse(4, rowp3, cols3, false);  // duplicate column in a row
  }
}

static void test_options() {
  Options o;
  o.addTrustRegionDefaults();
  o.addMMADefaults();
  CHECK(o.set("qn_subspace_size", 17) == PO_OK && o.integer("qn_subspace_size") == 17);
  CHECK(o.set("qn_subspace_size", -1) != PO_OK && o.integer("qn_subspace_size") == 17);  // out of range
  CHECK(o.set("abs_res_tol", 1e-9) == PO_OK && o.real("abs_res_tol") == 1e-9);
  CHECK(o.set("abs_res_tol", -1.0) != PO_OK);
  CHECK(o.set("abs_res_tol", 3) != PO_OK);            // wrong type
  CHECK(o.set("qn_type", "sr1") == PO_OK && std::string(o.str("qn_type")) == "sr1");
  CHECK(o.set("qn_type", "newton") != PO_OK);          // not a value of the enum
  CHECK(o.set("qn_type", (const char *)nullptr) != PO_OK);
  CHECK(o.set("no_such_option", 1) != PO_OK);
  CHECK(o.set("no_such_option", 1.0) != PO_OK);
  CHECK(o.set("no_such_option", "x") != PO_OK);
  CHECK(o.set("use_line_search", 5) == PO_OK && o.integer("use_line_search") == 1);  // booleans normalise
  CHECK(o.set("output_file", (const char *)nullptr) == PO_OK && std::string(o.str("output_file")).empty());
  CHECK(o.set("tr_max_size", 2.5) == PO_OK && o.set("mma_max_iterations", 7) == PO_OK);
  std::string longname(4000, 'x');
  CHECK(o.set(longname.c_str(), 1) != PO_OK);  // the error text is truncated, not overrun
}

int main() {
  test_lu();
  test_csr();
  test_options();
  if (fails) {
    fprintf(stderr, "%d check(s) failed\n", fails);
    return 1;
  }
  printf("host_sanitize: ok\n");
  return 0;
}
