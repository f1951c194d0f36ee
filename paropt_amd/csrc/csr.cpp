// Host side of the general sparse-constraint path (csr.hpp): the one-time symbolic analysis and the
// launch sequences.  No numeric work happens here.
#include "csr.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>

namespace po {

namespace {

constexpr int kFrontMin = 16;  // shortest chain scheduled as a front

int internal_error(const char *what) {
  set_error("internal: sparse Cholesky analysis: %s", what);
  return PO_ERR_ARG;
}

// subsets this small are numbered as they come (PAROPT_AMD_ND_LEAF overrides, for experiments)
int leaf_size() {
  static int v = 0;
  if (v == 0) {
    const char *e = getenv("PAROPT_AMD_ND_LEAF");
    v = e && atoi(e) > 0 ? atoi(e) : 8;
  }
  return v;
}

// Nested dissection from BFS level structures (George's automatic nested dissection): the middle
// level of a rooted level structure from a pseudo-peripheral vertex separates the graph; the two
// halves are numbered first, the separator last.  adj is the graph of S without the diagonal.
struct Dissector {
  const std::vector<int> &adjp, &adj;
  std::vector<int> region;  // current subset id of every vertex (-1 = already numbered)
  std::vector<int> level;
  std::vector<int> perm;    // perm[new] = old
  int next_region = 1;

  Dissector(const std::vector<int> &ap, const std::vector<int> &a, int w)
      : adjp(ap), adj(a), region(w, 0), level(w, -1), perm(w, -1) {}

  // BFS inside `reg` from `start`; vertices in visit order into out, level[] filled; returns #levels
  int bfs(int start, int reg, std::vector<int> &out, int stamp_region) {
    out.clear();
    out.push_back(start);
    region[start] = stamp_region;
    level[start] = 0;
    size_t head = 0;
    int nlev = 1;
    while (head < out.size()) {
      const int v = out[head++];
      for (int p = adjp[v]; p < adjp[v + 1]; p++) {
        const int u = adj[p];
        if (region[u] != reg) continue;
        region[u] = stamp_region;
        level[u] = level[v] + 1;
        nlev = level[u] + 1;
        out.push_back(u);
      }
    }
    return nlev;
  }

  void number(const std::vector<int> &verts, int lo) {
    for (size_t i = 0; i < verts.size(); i++) {
      perm[lo + (int)i] = verts[i];
      region[verts[i]] = -1;
    }
  }

  // number the vertices of `verts` (all carrying region id `reg`) into positions [lo, lo + |verts|)
  void order(std::vector<int> &verts, int reg, int lo, int depth) {
    if ((int)verts.size() <= leaf_size() || depth > 96) {
      number(verts, lo);
      return;
    }
    // connected components first: removing a separator may have split the subset
    std::vector<int> comp, queue;
    std::vector<std::vector<int>> big;
    std::vector<int> big_reg;
    std::vector<int> small;
    for (size_t s = 0; s < verts.size(); s++) {
      if (region[verts[s]] != reg) continue;
      const int r = next_region++;
      bfs(verts[s], reg, comp, r);
      if ((int)comp.size() <= leaf_size()) {
        small.insert(small.end(), comp.begin(), comp.end());
      } else {
        big.push_back(comp);
        big_reg.push_back(r);
      }
    }
    std::vector<int>().swap(verts);
    number(small, lo);
    lo += (int)small.size();
    for (size_t b = 0; b < big.size(); b++) {
      const int sz = (int)big[b].size();
      bisect(big[b], big_reg[b], lo, depth);
      lo += sz;
    }
  }

  void bisect(std::vector<int> &comp, int reg, int lo, int depth) {
    // pseudo-peripheral start: the last vertex of a BFS, twice
    std::vector<int> lv;
    int r1 = next_region++;
    bfs(comp[0], reg, lv, r1);
    int r2 = next_region++;
    bfs(lv.back(), r1, lv, r2);
    int r3 = next_region++;
    const int nlev = bfs(lv.back(), r2, lv, r3);
    if (nlev < 3) {  // clique-like: nothing to separate
      number(lv, lo);
      return;
    }
    std::vector<int> count(nlev, 0);
    for (int v : lv) count[level[v]]++;
    const int total = (int)lv.size();
    int mid = 1, cum = count[0];
    while (mid < nlev - 2 && cum + count[mid] < total / 2) cum += count[mid++];
    // vertices of the middle level without a neighbour in the next level can stay in the first half
    std::vector<int> A, B, S;
    const int ra = next_region++, rb = next_region++;
    for (int v : lv) {
      const int l = level[v];
      if (l < mid) {
        A.push_back(v);
      } else if (l > mid) {
        B.push_back(v);
      } else {
        bool touches = false;
        for (int p = adjp[v]; p < adjp[v + 1] && !touches; p++) {
          const int u = adj[p];
          touches = region[u] == r3 && level[u] == mid + 1;
        }
        (touches ? S : A).push_back(v);
      }
    }
    for (int v : A) region[v] = ra;
    for (int v : B) region[v] = rb;
    const int na = (int)A.size(), nb = (int)B.size();
    number(S, lo + na + nb);
    std::vector<int>().swap(lv);
    std::vector<int>().swap(comp);
    order(A, ra, lo, depth + 1);
    order(B, rb, lo + na, depth + 1);
  }
};

}  // namespace

static int csr_analyse_impl(int64_t n64, int64_t w64, const int *rowp, const int *cols, CsrSymbolic *out,
                            bool allow_fronts);

int csr_analyse(int64_t n, int64_t w, const int *rowp, const int *cols, CsrSymbolic *out) {
  const char *e = getenv("PAROPT_AMD_NO_FRONTS");
  int rc = csr_analyse_impl(n, w, rowp, cols, out, !(e && atoi(e) > 0));
  if (rc == PO_ERR_NUMERIC) {  // a chain that is not a dense front after all: schedule row by row
    *out = CsrSymbolic();
    rc = csr_analyse_impl(n, w, rowp, cols, out, false);
  }
  return rc;
}

static int csr_analyse_impl(int64_t n64, int64_t w64, const int *rowp, const int *cols, CsrSymbolic *out,
                            bool allow_fronts) {
  CsrSymbolic &s = *out;
  if (w64 < 0 || n64 < 0 || w64 > 2000000000LL || n64 > 2000000000LL) {
    set_error("sparse Jacobian: sizes out of range (nwcon %lld, nvars %lld)", (long long)w64, (long long)n64);
    return PO_ERR_ARG;
  }
  const int w = (int)w64, n = (int)n64;
  s.w = w;
  s.n = n;
  if (rowp[0] != 0) {
    set_error("sparse Jacobian: rowp[0] must be 0");
    return PO_ERR_ARG;
  }
  for (int i = 0; i < w; i++) {
    if (rowp[i + 1] < rowp[i]) {
      set_error("sparse Jacobian: rowp is not non-decreasing at row %d", i);
      return PO_ERR_ARG;
    }
  }
  const int nnz = rowp[w];
  s.nnz = nnz;
  // column-sorted rows, remembering the user's slot
  s.rowp.assign(rowp, rowp + w + 1);
  s.cols.resize(nnz);
  s.src.resize(nnz);
  s.identity_src = true;
  {
    std::vector<std::pair<int, int>> row;
    for (int i = 0; i < w; i++) {
      row.clear();
      for (int p = rowp[i]; p < rowp[i + 1]; p++) {
        if (cols[p] < 0 || cols[p] >= n) {
          set_error("sparse Jacobian: column %d out of range in row %d", cols[p], i);
          return PO_ERR_ARG;
        }
        row.push_back(std::make_pair(cols[p], p));
      }
      std::sort(row.begin(), row.end());
      for (size_t q = 0; q < row.size(); q++) {
        if (q > 0 && row[q].first == row[q - 1].first) {
          set_error("sparse Jacobian: column %d appears twice in row %d", row[q].first, i);
          return PO_ERR_ARG;
        }
        const int p = rowp[i] + (int)q;
        s.cols[p] = row[q].first;
        s.src[p] = row[q].second;
        if (row[q].second != p) s.identity_src = false;
      }
    }
  }
  // transpose
  s.colp.assign((size_t)n + 1, 0);
  for (int p = 0; p < nnz; p++) s.colp[s.cols[p] + 1]++;
  double pairs = 0.0;
  for (int j = 0; j < n; j++) {
    const double cnt = s.colp[j + 1];
    pairs += cnt * cnt;
    s.colp[j + 1] += s.colp[j];
  }
  if (pairs > 1.5e9) {
    set_error("sparse Jacobian: Aw Aw^T would have more than 1.5e9 entries (a dense column?)");
    return PO_ERR_ARG;
  }
  s.rowsT.resize(nnz);
  s.srcT.resize(nnz);
  {
    std::vector<int> fill(s.colp.begin(), s.colp.end() - 1);
    for (int i = 0; i < w; i++) {
      for (int p = s.rowp[i]; p < s.rowp[i + 1]; p++) {
        const int q = fill[s.cols[p]]++;
        s.rowsT[q] = i;
        s.srcT[q] = p;
      }
    }
  }
  // graph of S = pattern(Aw Aw^T) without the diagonal
  std::vector<int> adjp((size_t)w + 1, 0), adj;
  {
    std::vector<int> mark(w, -1);
    for (int i = 0; i < w; i++) {
      mark[i] = i;
      for (int p = s.rowp[i]; p < s.rowp[i + 1]; p++) {
        const int k = s.cols[p];
        for (int q = s.colp[k]; q < s.colp[k + 1]; q++) {
          const int j = s.rowsT[q];
          if (mark[j] != i) {
            mark[j] = i;
            adj.push_back(j);
          }
        }
      }
      adjp[i + 1] = (int)adj.size();
    }
  }
  s.nnzS = (int64_t)(adj.size() / 2) + w;
  // ordering
  {
    Dissector nd(adjp, adj, w);
    std::vector<int> all(w);
    std::iota(all.begin(), all.end(), 0);
    nd.order(all, 0, 0, 0);
    s.perm.swap(nd.perm);
  }
  // The dissection ordering fixes the elimination TREE; any topological order of that tree gives the same
  // factor up to a relabelling.  Rows are renumbered level by level (a row's level = the height of its
  // subtree), so the rows a launch works on - and their storage in L - are contiguous in memory.
  auto invert = [&](const std::vector<int> &perm, std::vector<int> &iperm) -> bool {
    iperm.assign(w, -1);
    for (int i = 0; i < w; i++) {
      if (perm[i] < 0 || perm[i] >= w || iperm[perm[i]] != -1) return false;
      iperm[perm[i]] = i;
    }
    return true;
  };
  // elimination tree (Liu, with path compression)
  auto etree = [&](const std::vector<int> &perm, const std::vector<int> &iperm, std::vector<int> &parent) {
    parent.assign(w, -1);
    std::vector<int> anc(w, -1);
    for (int i = 0; i < w; i++) {
      const int old = perm[i];
      for (int p = adjp[old]; p < adjp[old + 1]; p++) {
        int r = iperm[adj[p]];
        if (r >= i) continue;
        while (anc[r] != -1 && anc[r] != i) {
          const int nx = anc[r];
          anc[r] = i;
          r = nx;
        }
        if (anc[r] == -1) {
          anc[r] = i;
          parent[r] = i;
        }
      }
    }
  };
  auto levels_of = [&](const std::vector<int> &parent, std::vector<int> &lev) -> int {
    lev.assign(w, 0);
    int nlev = w > 0 ? 1 : 0;
    for (int i = 0; i < w; i++) {
      const int p = parent[i];
      if (p >= 0 && lev[p] < lev[i] + 1) lev[p] = lev[i] + 1;
      if (lev[i] + 1 > nlev) nlev = lev[i] + 1;
    }
    return nlev;
  };
  if (!invert(s.perm, s.iperm)) {
    set_error("internal: ordering is not a permutation");
    return PO_ERR_ARG;
  }
  // Fronts.  A separator of the dissection is a clique, i.e. a chain j -> j+1 -> ... of the tree whose
  // columns have nested structures (a fundamental supernode): row by row it would cost one level per
  // vertex.  Chains of at least kFrontMin rows become FRONTS: one scheduling unit whose rows are numbered
  // consecutively; the part of those rows left of the front is computed for all rows at once, the dense
  // lower triangle inside the front (the tails of the rows) by one workgroup.  Levels are heights in the
  // tree of units.
  std::vector<int> lev(w, 0), front_of(w, -1);
  {
    etree(s.perm, s.iperm, s.parent);
    const std::vector<int> &parent = s.parent;
    std::vector<int> colcount(w, 1), nchild(w, 0), head(w);
    {
      std::vector<int> mark(w, -1);
      for (int i = 0; i < w; i++) {
        const int old = s.perm[i];
        mark[i] = i;
        for (int p = adjp[old]; p < adjp[old + 1]; p++) {
          int j = s.iperm[adj[p]];
          if (j >= i) continue;
          while (mark[j] != i) {
            colcount[j]++;
            mark[j] = i;
            j = parent[j];
          }
        }
      }
    }
    for (int i = 0; i < w; i++) {
      if (parent[i] >= 0) nchild[parent[i]]++;
    }
    std::iota(head.begin(), head.end(), 0);
    if (allow_fronts) {
      for (int j = 0; j + 1 < w; j++) {
        if (parent[j] == j + 1 && nchild[j + 1] == 1 && colcount[j] == colcount[j + 1] + 1) head[j + 1] = head[j];
      }
    }
    std::vector<int> fsize(w, 0), ulev(w, 0);
    for (int i = 0; i < w; i++) fsize[head[i]]++;
    auto unit = [&](int i) { return fsize[head[i]] >= kFrontMin ? head[i] : i; };
    int nlev = w > 0 ? 1 : 0;
    for (int i = 0; i < w; i++) {
      const int u = unit(i), p = parent[i];
      if (ulev[u] + 1 > nlev) nlev = ulev[u] + 1;
      if (p >= 0) {
        const int up = unit(p);
        if (up != u && ulev[up] < ulev[u] + 1) ulev[up] = ulev[u] + 1;
      }
    }
    // bucket (level, ordinary | front), rows ascending inside a bucket: a front stays contiguous
    std::vector<int> start((size_t)2 * nlev + 1, 0), byLevel(w), newlev(w), newfront(w, -1);
    for (int i = 0; i < w; i++) start[2 * ulev[unit(i)] + (fsize[head[i]] >= kFrontMin ? 1 : 0) + 1]++;
    for (int b = 0; b < 2 * nlev; b++) start[b + 1] += start[b];
    std::vector<int> newpos(w);
    for (int i = 0; i < w; i++) {
      const int u = unit(i);
      const bool in_front = fsize[head[i]] >= kFrontMin;
      const int pos = start[2 * ulev[u] + (in_front ? 1 : 0)]++;
      newpos[i] = pos;
      byLevel[pos] = s.perm[i];
      newlev[pos] = ulev[u];
      if (in_front) newfront[pos] = newpos[head[i]];
    }
    s.perm.swap(byLevel);
    lev.swap(newlev);
    front_of.swap(newfront);
    invert(s.perm, s.iperm);
  }
  etree(s.perm, s.iperm, s.parent);
  s.Lrowp.assign((size_t)w + 1, 0);
  s.Lcols.clear();
  {
    std::vector<int> mark(w, -1);
    for (int i = 0; i < w; i++) {
      const int old = s.perm[i];
      const size_t begin = s.Lcols.size();
      mark[i] = i;
      for (int p = adjp[old]; p < adjp[old + 1]; p++) {
        int j = s.iperm[adj[p]];
        if (j >= i) continue;
        while (mark[j] != i) {
          s.Lcols.push_back(j);
          mark[j] = i;
          j = s.parent[j];
        }
      }
      std::sort(s.Lcols.begin() + begin, s.Lcols.end());
      s.Lcols.push_back(i);
      if (s.Lcols.size() > 2000000000ULL) {
        set_error("sparse Cholesky: the factor would have more than 2e9 entries");
        return PO_ERR_ARG;
      }
      s.Lrowp[i + 1] = (int)s.Lcols.size();
    }
  }
  s.nnzL = (int64_t)s.Lcols.size();
  // columns of L below the diagonal
  s.Ltp.assign((size_t)w + 1, 0);
  for (int i = 0; i < w; i++) {
    for (int p = s.Lrowp[i]; p < s.Lrowp[i + 1] - 1; p++) s.Ltp[s.Lcols[p] + 1]++;
  }
  for (int j = 0; j < w; j++) s.Ltp[j + 1] += s.Ltp[j];
  s.Ltrows.resize(s.Ltp[w]);
  s.Ltsrc.resize(s.Ltp[w]);
  {
    std::vector<int> fill(s.Ltp.begin(), s.Ltp.end() - 1);
    for (int i = 0; i < w; i++) {
      for (int p = s.Lrowp[i]; p < s.Lrowp[i + 1] - 1; p++) {
        const int q = fill[s.Lcols[p]]++;
        s.Ltrows[q] = i;
        s.Ltsrc[q] = p;
      }
    }
  }
  // structural entries of lower(P S P^T) -> slot in L
  s.ent_a.clear();
  s.ent_b.clear();
  s.ent_slot.clear();
  for (int i = 0; i < w; i++) {
    const int old = s.perm[i];
    s.ent_a.push_back(old);
    s.ent_b.push_back(old);
    s.ent_slot.push_back(s.Lrowp[i + 1] - 1);
    for (int p = adjp[old]; p < adjp[old + 1]; p++) {
      const int j = s.iperm[adj[p]];
      if (j >= i) continue;
      const int *b = &s.Lcols[s.Lrowp[i]], *e = &s.Lcols[s.Lrowp[i + 1] - 1];
      const int *f = std::lower_bound(b, e, j);
      if (f == e || *f != j) {
        set_error("internal: entry (%d,%d) of S missing from the factor pattern", i, j);
        return PO_ERR_ARG;
      }
      s.ent_a.push_back(old);
      s.ent_b.push_back(adj[p]);
      s.ent_slot.push_back((int)(f - &s.Lcols[0]));
    }
  }
  // assemble in the order of Aw's rows: the (many) reads of an entry walk Aw coalesced, its one write scatters
  {
    const size_t ne = s.ent_slot.size();
    std::vector<int> idx(ne);
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](int x, int y) {
      return s.ent_a[x] != s.ent_a[y] ? s.ent_a[x] < s.ent_a[y] : s.ent_b[x] < s.ent_b[y];
    });
    std::vector<int> ta(ne), tb(ne), ts(ne);
    for (size_t k = 0; k < ne; k++) {
      ta[k] = s.ent_a[idx[k]];
      tb[k] = s.ent_b[idx[k]];
      ts[k] = s.ent_slot[idx[k]];
    }
    s.ent_a.swap(ta);
    s.ent_b.swap(tb);
    s.ent_slot.swap(ts);
  }
  // Levels.  An ordinary row needs every row in its pattern, all of them in earlier levels; the rows of a
  // front need earlier levels for the columns left of the front and each other inside it.  The same sets in
  // DESCENDING order schedule the backward solve (x_i needs x_j for the ancestors j in column i).
  {
    int nlev = 0;
    for (int i = 0; i < w; i++) nlev = std::max(nlev, lev[i] + 1);
    s.fwd_ptr.assign((size_t)nlev + 1, 0);
    for (int i = 0; i < w; i++) s.fwd_ptr[lev[i] + 1]++;
    for (int l = 0; l < nlev; l++) s.fwd_ptr[l + 1] += s.fwd_ptr[l];
    s.fwd_order.resize(w);
    std::iota(s.fwd_order.begin(), s.fwd_order.end(), 0);
    s.fwd_maxlen.assign(nlev, 0);
    s.bwd_maxlen.assign(nlev, 0);
    s.ord_end.assign(nlev, 0);
    s.front_ptr.assign((size_t)nlev + 1, 0);
    s.front_maxdesc.assign(nlev, 0);
    s.front_start.clear();
    s.front_size.clear();
    s.front_of = front_of;
    for (int l = 0; l < nlev; l++) {
      int i = s.fwd_ptr[l];
      const int e = s.fwd_ptr[l + 1];
      while (i < e && front_of[i] < 0) {
        if (lev[i] != l) return internal_error("rows are not numbered level by level");
        s.fwd_maxlen[l] = std::max(s.fwd_maxlen[l], s.Lrowp[i + 1] - s.Lrowp[i]);
        s.bwd_maxlen[l] = std::max(s.bwd_maxlen[l], s.Ltp[i + 1] - s.Ltp[i]);
        for (int p = s.Lrowp[i]; p < s.Lrowp[i + 1] - 1; p++) {
          if (lev[s.Lcols[p]] >= l) return internal_error("a row depends on a row of its own level");
        }
        i++;
      }
      s.ord_end[l] = i;
      while (i < e) {
        const int f0 = i;
        if (front_of[i] != f0) return internal_error("front rows are not contiguous");
        int sz = 0;
        while (i < e && front_of[i] == f0) {
          // the tail of row f0 + sz must be exactly the columns f0 .. f0 + sz
          const int len = s.Lrowp[i + 1] - s.Lrowp[i];
          if (lev[i] != l || len < sz + 1 || s.Lcols[s.Lrowp[i + 1] - 1 - sz] != f0) return PO_ERR_NUMERIC;
          for (int p = s.Lrowp[i]; p < s.Lrowp[i + 1] - 1 - sz; p++) {
            if (lev[s.Lcols[p]] >= l) return internal_error("a front depends on a row of its own level");
          }
          s.front_maxdesc[l] = std::max(s.front_maxdesc[l], len - sz - 1);
          sz++;
          i++;
        }
        s.front_start.push_back(f0);
        s.front_size.push_back(sz);
        s.max_front = std::max(s.max_front, sz);
      }
      s.front_ptr[l + 1] = (int)s.front_start.size();
    }
  }
  return PO_OK;
}

// ---------------------------------------------------------------------------------------------------
CsrSparse::CsrSparse(Ctx *c, int64_t n_, int64_t w_) : ctx(c), n(n_), w(w_), nnz(0) {}

namespace {
template <class T>
void dfree(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}
template <class T>
int upload(T *&dst, const std::vector<T> &src) {
  dfree(dst);
  const size_t bytes = (src.size() + 4) * sizeof(T);
  PO_HIP(hipMalloc((void **)&dst, bytes));
  PO_HIP(hipMemset(dst, 0, bytes));
  if (!src.empty()) PO_HIP(hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return PO_OK;
}
}  // namespace

CsrSparse::~CsrSparse() {
  if (vals != data) dfree(vals);
  vals = nullptr;
  dfree(data);
  dfree(d_rowp);
  dfree(d_cols);
  dfree(d_src);
  dfree(d_colp);
  dfree(d_rowsT);
  dfree(d_srcT);
  dfree(d_perm);
  dfree(d_iperm);
  dfree(d_Lrowp);
  dfree(d_Lcols);
  dfree(d_Ltp);
  dfree(d_Ltrows);
  dfree(d_Ltsrc);
  dfree(d_ent_a);
  dfree(d_ent_b);
  dfree(d_ent_slot);
  dfree(d_fwd);
  dfree(d_front_of);
  dfree(d_front_end);
  dfree(d_fstart);
  dfree(d_fsize);
  dfree(Lvals);
  dfree(ones);
  dfree(wwork);
  dfree(d_flag);
  if (cw) vec_decref(cw);
}

static int group_for(double avg) {
  if (avg <= 6.0) return 1;
  if (avg <= 24.0) return 4;
  if (avg <= 96.0) return 16;
  return 64;
}

int CsrSparse::setPattern(const int *rowp, const int *cols) {
  PO_TRY(csr_analyse(n, w, rowp, cols, &sym));
  nnz = sym.nnz;
  user_rowp.assign(rowp, rowp + w + 1);
  user_cols.assign(cols, cols + nnz);
  PO_HIP(hipSetDevice(ctx->device));
  if (vals != data) dfree(vals);
  vals = nullptr;
  const size_t vbytes = ((size_t)nnz + 4) * sizeof(double);
  if (adopt_data) {  // upgradeFromLight: the user-visible value array (and what the user wrote into it) stays
    data = adopt_data;
    adopt_data = nullptr;
  } else {
    dfree(data);
    PO_HIP(hipMalloc((void **)&data, vbytes));
    PO_HIP(hipMemset(data, 0, vbytes));
  }
  if (sym.identity_src) {
    vals = data;
  } else {
    PO_HIP(hipMalloc((void **)&vals, vbytes));
    PO_HIP(hipMemset(vals, 0, vbytes));
  }
  PO_TRY(upload(d_rowp, sym.rowp));
  PO_TRY(upload(d_cols, sym.cols));
  PO_TRY(upload(d_src, sym.src));
  PO_TRY(upload(d_colp, sym.colp));
  PO_TRY(upload(d_rowsT, sym.rowsT));
  PO_TRY(upload(d_srcT, sym.srcT));
  PO_TRY(upload(d_perm, sym.perm));
  PO_TRY(upload(d_iperm, sym.iperm));
  PO_TRY(upload(d_Lrowp, sym.Lrowp));
  PO_TRY(upload(d_Lcols, sym.Lcols));
  PO_TRY(upload(d_Ltp, sym.Ltp));
  PO_TRY(upload(d_Ltrows, sym.Ltrows));
  PO_TRY(upload(d_Ltsrc, sym.Ltsrc));
  PO_TRY(upload(d_ent_a, sym.ent_a));
  PO_TRY(upload(d_ent_b, sym.ent_b));
  PO_TRY(upload(d_ent_slot, sym.ent_slot));
  PO_TRY(upload(d_fwd, sym.fwd_order));
  {
    std::vector<int> front_end(sym.front_of.size(), -1);
    for (size_t f = 0; f < sym.front_start.size(); f++) {
      for (int r = 0; r < sym.front_size[f]; r++) front_end[sym.front_start[f] + r] = sym.front_start[f] + sym.front_size[f];
    }
    PO_TRY(upload(d_front_of, sym.front_of));
    PO_TRY(upload(d_front_end, front_end));
    PO_TRY(upload(d_fstart, sym.front_start));
    PO_TRY(upload(d_fsize, sym.front_size));
  }
  dfree(Lvals);
  dfree(ones);
  dfree(wwork);
  dfree(d_flag);
  PO_HIP(hipMalloc((void **)&Lvals, ((size_t)sym.nnzL + 4) * sizeof(double)));
  PO_HIP(hipMalloc((void **)&ones, ((size_t)w + 4) * sizeof(double)));
  PO_HIP(hipMalloc((void **)&wwork, ((size_t)w + 4) * sizeof(double)));
  PO_HIP(hipMalloc((void **)&d_flag, 4 * sizeof(int)));
  PO_HIP(hipMemset(wwork, 0, ((size_t)w + 4) * sizeof(double)));
  PO_HIP(hipMemset(ones, 0, ((size_t)w + 4) * sizeof(double)));
  PO_HIP(hipDeviceSynchronize());
  PO_TRY(k_fill(ctx, ones, w, 1.0));
  if (adopt_cw) {  // (the same Vec object: handles the user holds stay valid)
    cw = adopt_cw;
    adopt_cw = nullptr;
  } else {
    if (cw) vec_decref(cw);
    cw = vec_new(ctx, w);
    if (!cw) return PO_ERR_HIP;
  }
  nlevels_f = (int)sym.fwd_ptr.size() - 1;
  spmv_group = group_for(w > 0 ? (double)nnz / (double)w : 0.0);
  spmvT_group = group_for(n > 0 ? (double)nnz / (double)n : 0.0);
  return PO_OK;
}

int CsrSparse::setPatternLight(const int *rowp, const int *cols) {
  nnz = rowp[w];
  user_rowp.assign(rowp, rowp + w + 1);
  user_cols.assign(cols, cols + nnz);
  PO_HIP(hipSetDevice(ctx->device));
  if (vals != data) dfree(vals);
  vals = nullptr;
  dfree(data);
  const size_t vbytes = ((size_t)nnz + 4) * sizeof(double);
  PO_HIP(hipMalloc((void **)&data, vbytes));
  PO_HIP(hipMemset(data, 0, vbytes));
  vals = data;
  if (cw) vec_decref(cw);
  cw = vec_new(ctx, w);
  if (!cw) return PO_ERR_HIP;
  light = true;
  return PO_OK;
}

int CsrSparse::upgradeFromLight() {
  if (!light) return PO_OK;
  // The analysis runs after all.  The user-visible value array `data` (po_problem_get_sparse_jacobian_data hands it
  // out as a borrowed pointer, like the reference's getSparseJacobianData) and the constraint-value vector `cw` are
  // ADOPTED by setPattern, not reallocated: a caller that cached either keeps a valid pointer, and what the gradient
  // callback just wrote stays in place.
  adopt_data = data;
  adopt_cw = cw;
  data = nullptr;
  vals = nullptr;
  cw = nullptr;
  light = false;
  const std::vector<int> rp(user_rowp), cl(user_cols);
  int rc = setPattern(rp.data(), cl.data());
  if (adopt_data) {  // setPattern failed before it took them over
    data = adopt_data;
    adopt_data = nullptr;
    if (!vals) vals = data;
  }
  if (adopt_cw) {
    if (cw && cw != adopt_cw) vec_decref(cw);
    cw = adopt_cw;
    adopt_cw = nullptr;
  }
  return rc;
}

int CsrSparse::valuesChanged() {
  if (light || vals == data || nnz == 0) return PO_OK;
  return k_csr_gather(ctx, vals, data, d_src, nnz);
}

int CsrSparse::spmv(double alpha, const double *px, double *out) {
  return k_csr_spmv(ctx, spmv_group, d_rowp, d_cols, vals, w, alpha, px, nullptr, 1.0, out, out, nullptr);
}

int CsrSparse::spmvT(double alpha, const double *pzw, double *out) {
  return k_csr_spmvT(ctx, spmvT_group, d_colp, d_rowsT, d_srcT, vals, n, alpha, pzw, out, nullptr, out);
}

int CsrSparse::innerProduct(double alpha, const double *cvec, double *out) {
  return k_csr_inner(ctx, d_rowp, d_cols, vals, w, alpha, cvec, out);
}

int CsrSparse::colSum(double scale, const double *y, double *out) {
  if (light) {  // no transposed index was built (setPatternLight): the caller owns the grouped form of this product
    set_error("CsrSparse::colSum on a light (grouped) pattern");
    return PO_ERR_ARG;
  }
  return k_csr_colsum(ctx, d_colp, d_rowsT, n, scale, y, out);
}

int CsrSparse::panelPermuted(const double *d, const double *const *P, int nv, double *const *U) {
  return k_csr_panel(ctx, d_rowp, d_cols, vals, w, d, P, nv, U, d_iperm);
}

// one thread per row when the rows of a level are short, or when there are so many that even long rows keep
// every lane busy; a wavefront per row otherwise
static int thinLevel(int maxlen, int64_t work_items) { return maxlen <= 24 || (work_items >= 65536 && maxlen <= 96); }

int CsrSparse::factor(const double *dinv, const double *cdiag) {
  if (w <= 0) return PO_OK;
  PO_HIP(hipMemsetAsync(Lvals, 0, ((size_t)sym.nnzL + 4) * sizeof(double), ctx->stream));
  PO_HIP(hipMemsetAsync(d_flag, 0, 4 * sizeof(int), ctx->stream));
  PO_TRY(k_csr_assemble(ctx, d_rowp, d_cols, vals, dinv, cdiag, d_ent_a, d_ent_b, d_ent_slot,
                        (int64_t)sym.ent_slot.size(), Lvals));
  for (int l = 0; l < nlevels_f; l++) {
    const int b = sym.fwd_ptr[l], e = sym.fwd_ptr[l + 1];
    const int oe = sym.ord_end[l], f0 = sym.front_ptr[l], nf = sym.front_ptr[l + 1] - f0;
    PO_TRY(k_chol_level(ctx, d_Lrowp, d_Lcols, Lvals, d_fwd + b, oe - b, d_flag,
                        thinLevel(sym.fwd_maxlen[l], oe - b), sym.fwd_maxlen[l]));
    PO_TRY(k_chol_fronts(ctx, d_Lrowp, d_Lcols, Lvals, oe, e - oe, d_front_of, d_fstart + f0, d_fsize + f0, nf,
                         sym.front_maxdesc[l], d_flag));
  }
  int flag[4] = {0, 0, 0, 0};
  PO_HIP(hipMemcpyAsync(flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, ctx->stream));
  PO_HIP(hipStreamSynchronize(ctx->stream));
  if (flag[0] != 0) {
    // S is not numerically SPD: D^-1 or C has a non-positive entry (an indefinite diagonal Hessian, say).
    // The reference ignores LAPACK's info in the same situation (ParOptSparseCholesky.cpp:631, and the return
    // value of mat->factor at ParOptInteriorPoint.cpp:1930) and carries on with whatever the factor holds;
    // here the offending pivots were replaced by 1 so the factor stays finite, the event is counted, the text
    // is kept for po_last_error() and the interior point carries on too.  po_quasidef_factor reports it.
    std::vector<double> hd((size_t)n), hc((size_t)w);
    double dmin = 0.0, cmin = 0.0;
    long bad = 0;
    if (hipMemcpy(hd.data(), dinv, hd.size() * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(hc.data(), cdiag, hc.size() * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) {
      dmin = hd.empty() ? 0.0 : hd[0];
      cmin = hc.empty() ? 0.0 : hc[0];
      for (double v : hd) {
        if (!(v == v) || v - v != 0.0) bad++;
        if (v < dmin) dmin = v;
      }
      for (double v : hc) {
        if (!(v == v) || v - v != 0.0) bad++;
        if (v < cmin) cmin = v;
      }
    }
    set_error("sparse Cholesky: non-positive pivot in row %d of the permuted Schur complement "
              "(min D^-1 %.3e, min C %.3e, %ld non-finite inputs)", flag[1], dmin, cmin, bad);
    if (breakdowns++ == 0 && ctx->rank == 0) {
      fprintf(stderr, "ParOpt warning: sparse Cholesky breakdown, the quasi-definite matrix is not positive "
                      "definite (min D^-1 %.3e, min C %.3e); continuing as the reference does\n", dmin, cmin);
    }
  }
  return PO_OK;
}

int CsrSparse::solveInPlace(double *const *Y, int nv, bool forward, bool backward) {
  if (nv > kMaxPanel) {  // independent right-hand sides: slabs of one kernel's pointer table
    for (int j0 = 0; j0 < nv; j0 += kMaxPanel)
      PO_TRY(solveInPlace(Y + j0, nv - j0 > kMaxPanel ? kMaxPanel : nv - j0, forward, backward));
    return PO_OK;
  }
  if (w <= 0 || nv <= 0) return PO_OK;
  if (forward) {
    for (int l = 0; l < nlevels_f; l++) {
      const int b = sym.fwd_ptr[l], e = sym.fwd_ptr[l + 1];
      const int oe = sym.ord_end[l], f0 = sym.front_ptr[l], nf = sym.front_ptr[l + 1] - f0;
      PO_TRY(k_trsv_fwd_level(ctx, d_Lrowp, d_Lcols, Lvals, d_fwd + b, oe - b, Y, nv,
                              thinLevel(sym.fwd_maxlen[l], (int64_t)(oe - b) * nv)));
      PO_TRY(k_trsv_fronts_fwd(ctx, d_Lrowp, d_Lcols, Lvals, oe, e - oe, d_front_of, d_fstart + f0, d_fsize + f0,
                               nf, Y, nv));
    }
  }
  if (backward) {
    for (int l = nlevels_f - 1; l >= 0; l--) {
      const int b = sym.fwd_ptr[l], e = sym.fwd_ptr[l + 1];
      const int oe = sym.ord_end[l], f0 = sym.front_ptr[l], nf = sym.front_ptr[l + 1] - f0;
      PO_TRY(k_trsv_fronts_bwd(ctx, d_Lrowp, d_Ltp, d_Ltrows, d_Ltsrc, Lvals, oe, e - oe, d_front_of, d_front_end,
                               d_fstart + f0, d_fsize + f0, nf, Y, nv));
      PO_TRY(k_trsv_bwd_level(ctx, d_Lrowp, d_Ltp, d_Ltrows, d_Ltsrc, Lvals, d_fwd + b, oe - b, Y, nv,
                              thinLevel(sym.bwd_maxlen[l], (int64_t)(oe - b) * nv)));
    }
  }
  return PO_OK;
}

int CsrSparse::halfSolve(double *const *U, int nv) { return solveInPlace(U, nv, true, false); }

int CsrSparse::correction(const double *const *Yp, int nv, const double *alpha, double *out) {
  PO_TRY(k_panel_axpy(ctx, wwork, 0.0, nullptr, 0.0, alpha, Yp, nv, w));
  double *Y[1] = {wwork};
  PO_TRY(solveInPlace(Y, 1, false, true));
  PO_TRY(k_csr_gather(ctx, out, wwork, d_iperm, w));
  return k_scale(ctx, out, w, -1.0);
}

int CsrSparse::applyK0(const double *dinv, const double *bx, const double *bw, double *yx, double *yw) {
  // wwork (elimination order) = bw - Aw (dinv o bx)
  PO_TRY(k_csr_spmv(ctx, spmv_group, d_rowp, d_cols, vals, w, -1.0, bx, dinv, bw ? 1.0 : 0.0, bw, wwork,
                    d_iperm));
  double *Y[1] = {wwork};
  PO_TRY(solveInPlace(Y, 1, true, true));
  PO_TRY(k_csr_gather(ctx, yw, wwork, d_iperm, w));
  // yx = dinv o (bx + Aw^T yw)
  return k_csr_spmvT(ctx, spmvT_group, d_colp, d_rowsT, d_srcT, vals, n, 1.0, yw, bx, dinv, yx);
}

const char *CsrSparse::factorInfo() {
  // the fields of ParOptQuasiDefSparseMat::getFactorInfo (src/ParOptSparseMat.cpp:433-450); this factor has
  // the level and front counts take the supernode column
  char buf[192];
  const double tri = 0.5 * (double)w * ((double)w + 1.0);
  snprintf(buf, sizeof(buf),
           "n %5lld nlevels %5d nfronts %5d nnz(K) %7lld nnz(L) %7lld nnz(L) / nnz(K) %8.4f sparsity(L) %8.2e",
           (long long)w, nlevels_f, (int)sym.front_start.size(), (long long)sym.nnzS, (long long)sym.nnzL,
           sym.nnzS > 0 ? (double)sym.nnzL / (double)sym.nnzS : 0.0, tri > 0 ? (double)sym.nnzL / tri : 0.0);
  info = buf;
  return info.c_str();
}

}  // namespace po
