// Weighted Gram  W = P^T diag(d) P  on the fp64 matrix cores of gfx950.
//
// Replaces the reference's c(c+1)/2 dots of setUpKKTDiagSystem (src/ParOptInteriorPoint.cpp:1935-1950)
// AND the k diagonal-KKT solves + k mdots of setUpKKTSystem (:2648-2654) by one pass over the panel
// P = [Ac | Z]; with a pre-weighted last column t = Dinv o d1 the same pass also yields the
// panel dots P^T t of the bordered solve that follows (:2139-2147).  Algorithmic traffic 8(m+1)n bytes,
// m(m+1)n useful flops.
//
// Instruction choice (tools/mfma_f64_rate.hip on MI355X, profiles/r02_mfma_f64_rate.txt):
//   v_mfma_f64_16x16x4_f64     2048 flop in ~105 cycles/SIMD  -> 47 TFLOP/s sustained
//   v_mfma_f64_4x4x4_4b_f64     512 flop in 16.6 cycles/SIMD -> 75.8 TFLOP/s (96 % of the 78.6 spec)
// and the 4-column granularity of the second form wastes far less of a symmetric product: for m = 43 columns
// 66 blocks of 4x4 (1056 products per row, 946 useful) against 6 blocks of 16x16 (1536 per row).
//
// Mapping.  A workgroup (4 wavefronts) stages a tile of 128 rows of all columns (padded to groups of 4) in LDS,
// [column][row] with a row stride of 136 doubles (== 8 mod 32: the ds_read_b64 operand fetch below touches 32
// distinct 8-byte bank pairs per half wave).  v_mfma_f64_4x4x4_4b multiplies four independent 4x4x4 blocks;
// measured operand layout (tools/mfma_f64_layout.hip): A lane = i + 4 b + 16 k, B lane = j + 4 b + 16 k,
// D lane = j + 4 b + 16 i.  The four blocks b are four row chunks of the tile (16 rows per instruction:
// row = 16 s + b + 4 k), so one instruction accumulates one 4x4 block pair (I, J) of W over 16 rows, and every
// lane fetches ONE operand per column group: a_g = P[row][4 g + (lane & 3)], used as A for the pairs (g, .)
// and, multiplied by the row weight, as B for the pairs (., g).  The NG (NG + 1) / 2 block pairs are dealt
// to the four wavefronts by column group J (OUTPUT split: every wavefront sees all rows, owns about a quarter
// of W and weights only its own B operands), so
// there is no cross-wave reduction; the four row-chunk partials of a block are summed with two lane shuffles
// at the very end.  The loads of the next tile are issued into registers before the current tile is
// multiplied.
#include "core.hpp"

#include <math.h>
#include <stdlib.h>

#include <memory>
#include <vector>

namespace po {

typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int kGramTile = 128;          // rows per LDS tile
constexpr int kGramLd = kGramTile + 8;  // LDS row stride in doubles (== 8 mod 32)

struct PtrTableW {
  double *p[kMaxPanel];
};

__device__ __forceinline__ f64x2 ld_nt(const double *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p));
}

// Block pairs (I <= J) are dealt to the four wavefronts by COLUMN group J (the B side): a wavefront then weights
// only the few operands b_J = w a_J of its own column groups instead of all of them.  Longest-processing-time
// assignment of the column groups (column group J carries J + 1 pairs), fixed at compile time.
template <int NG>
struct GramPlan {
  int owner[NG];
  int load[4];
  constexpr GramPlan() : owner{}, load{0, 0, 0, 0} {
    for (int J = NG - 1; J >= 0; J--) {
      int best = 0;
      for (int w = 1; w < 4; w++)
        if (load[w] < load[best]) best = w;
      owner[J] = best;
      load[best] += J + 1;
    }
  }
  constexpr int max_load() const {
    int m = load[0];
    for (int w = 1; w < 4; w++)
      if (load[w] > m) m = load[w];
    return m;
  }
};
template <int NG>
struct GramPlanHolder {
  static constexpr GramPlan<NG> plan = GramPlan<NG>();
  static constexpr int NQ = plan.max_load();
};

// operands of one 16-row step: a[g] = P[row][4 g + (lane & 3)] for every column group, w = the row weight
template <int NG, int LD = kGramLd>
__device__ __forceinline__ void gram_fetch(const double *__restrict__ base, const double *__restrict__ dwl, int s,
                                           double (&a)[NG], double &w) {
  w = dwl[16 * s];
#pragma unroll
  for (int g = 0; g < NG; g++) a[g] = base[(4 * g) * LD + 16 * s];
}

template <int NG, int W>
__device__ __forceinline__ void gram_step(const double (&a)[NG], double w, bool tsel,
                                          double (&acc)[GramPlanHolder<NG>::NQ]) {
  constexpr GramPlan<NG> plan = GramPlanHolder<NG>::plan;
  int q = 0;
#pragma unroll
  for (int J = 0; J < NG; J++) {
    if (plan.owner[J] == W) {
      // the pre-weighted column t (if any) is the last column: its products with the panel are plain dots
      const double bw = a[J] * ((J == NG - 1 && tsel) ? 1.0 : w);
#pragma unroll
      for (int I = 0; I <= J; I++) {
        acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[I], bw, acc[q], 0, 0, 0);
        q++;
      }
    }
  }
}

template <int NG, int W, int TR = kGramTile, int LD = kGramLd>
__device__ __forceinline__ void gram_tile(const double *__restrict__ pt, const double *__restrict__ dw, int lane,
                                          int tcol, double (&acc)[GramPlanHolder<NG>::NQ]) {
  const int ci = lane & 3;                                 // column within the group
  const int rowoff = ((lane >> 2) & 3) + 4 * (lane >> 4);  // row within the 16-row step: b + 4 k
  const double *base = pt + ci * LD + rowoff;
  const double *dwl = dw + rowoff;
  const bool tsel = (4 * (NG - 1) + ci == tcol);
  // the operand fetch of step s + 1 is issued before the matrix instructions of step s
  double a0[NG], a1[NG], w0, w1;
  gram_fetch<NG, LD>(base, dwl, 0, a0, w0);
#pragma unroll 1
  for (int s = 0; s < TR / 16; s += 2) {
    gram_fetch<NG, LD>(base, dwl, s + 1, a1, w1);
    gram_step<NG, W>(a0, w0, tsel, acc);
    if (s + 2 < TR / 16) gram_fetch<NG, LD>(base, dwl, s + 2, a0, w0);
    gram_step<NG, W>(a1, w1, tsel, acc);
  }
}

// ---- ROW-SPLIT consumers (producer/consumer form, NG <= kGramRowSplitMaxNG) ------------------------------------
// Each consumer wave takes two of the tile's eight 16-row steps and ALL NG (NG + 1) / 2 block pairs: every operand
// is fetched from LDS once per workgroup (24 ds_read_b64 per wave and tile instead of 96 with the output split
// above), the matrix-instruction count per wave is the same, and the four per-wave partial results are summed
// through LDS once, at the end of the kernel, in wave order (deterministic).
constexpr int kGramRowSplitMaxNG = 14;  // NG (NG + 1) / 2 accumulators of 2 VGPRs each must fit beside the operands
constexpr int kGramRowSplitMinNG = 5;   // narrow panels (<= 16 columns) are nowhere near the LDS / matrix limits: they
                                        // keep the output split (and with it round 2's summation order)

template <int NG>
__device__ __forceinline__ void gram_step_all(const double (&a)[NG], double w, bool tsel,
                                              double (&acc)[NG * (NG + 1) / 2]) {
  int q = 0;
#pragma unroll
  for (int J = 0; J < NG; J++) {
    const double bw = a[J] * ((J == NG - 1 && tsel) ? 1.0 : w);
#pragma unroll
    for (int I = 0; I <= J; I++) {
      acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[I], bw, acc[q], 0, 0, 0);
      q++;
    }
  }
}

struct GramNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// `between`: called after the tile's operand fetches have been issued and before its matrix instructions (the fused
// group sums request THEIR operands there: LDS serves the matrix operands first, the rest arrives behind the matrix work)
template <int NG, class Hook = GramNoHook>
__device__ __forceinline__ void gram_tile_rows(const double *__restrict__ pt, const double *__restrict__ dw, int lane,
                                               int wave, int tcol, double (&acc)[NG * (NG + 1) / 2],
                                               const Hook &between = Hook()) {
  const int ci = lane & 3;
  const int rowoff = ((lane >> 2) & 3) + 4 * (lane >> 4);
  const double *base = pt + ci * kGramLd + rowoff;
  const double *dwl = dw + rowoff;
  const bool tsel = (4 * (NG - 1) + ci == tcol);
  double a0[NG], a1[NG], w0, w1;
  // The two steps' operands are 16 doubles apart: left alone, hipcc merges each pair of fetches into one
  // ds_read2_b64, whose banking ((addr / 4) mod 32, MI355X_MICROARCH.md LDS table) makes the [column][row] stride of
  // 136 doubles a two-way conflict (SQ_LDS_BANK_CONFLICT = a third of the LDS-active cycles, r03 PMC pass).  The
  // second step's offset is laundered through an empty asm so that the fetches stay separate ds_read_b64.
  int off1 = 16 * (2 * wave + 1);  // (an opaque OFFSET: laundering the pointer itself would lose its LDS address
  asm volatile("" : "+v"(off1));   //  space and turn the fetches into flat loads)
  if constexpr (NG <= 12) {
    gram_fetch<NG>(base, dwl, 2 * wave, a0, w0);        // both steps requested before the first matrix instruction:
    gram_fetch<NG>(base + off1, dwl + off1, 0, a1, w1);  // the second fetch lands behind the first step's work
    between();
    gram_step_all<NG>(a0, w0, tsel, acc);
    gram_step_all<NG>(a1, w1, tsel, acc);
  } else {  // wider panels: the accumulators leave no room for both steps' operands
    gram_fetch<NG>(base, dwl, 2 * wave, a0, w0);
    between();
    gram_step_all<NG>(a0, w0, tsel, acc);
    gram_fetch<NG>(base + off1, dwl + off1, 0, a0, w0);
    gram_step_all<NG>(a0, w0, tsel, acc);
  }
}

// column-major pair order of gram_step_all: q(I, J) = J (J + 1) / 2 + I
template <int NG>
__device__ __forceinline__ void gram_store_rows(const double (&acc)[NG * (NG + 1) / 2], int lane, int wave,
                                                double *__restrict__ red /* [4][NP * 16] in LDS */) {
  constexpr int NP = NG * (NG + 1) / 2;
  int q = 0;
#pragma unroll
  for (int J = 0; J < NG; J++) {
#pragma unroll
    for (int I = 0; I <= J; I++) {
      const int p = I * NG - (I * (I - 1)) / 2 + (J - I);  // row-major index of (I, J): the slot layout of gram_store
      double v = acc[q];
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      if (((lane >> 2) & 3) == 0) red[(size_t)wave * NP * 16 + p * 16 + (lane >> 4) * 4 + (lane & 3)] = v;
      q++;
    }
  }
}

template <int NG, int W>
__device__ __forceinline__ void gram_store(const double (&acc)[GramPlanHolder<NG>::NQ], int lane,
                                           double *__restrict__ partials) {
  constexpr GramPlan<NG> plan = GramPlanHolder<NG>::plan;
  int q = 0;
#pragma unroll
  for (int J = 0; J < NG; J++) {
    if (plan.owner[J] == W) {
#pragma unroll
      for (int I = 0; I <= J; I++) {
        const int p = I * NG - (I * (I - 1)) / 2 + (J - I);  // row-major index of (I, J) among the upper pairs
        double v = acc[q];
        v += __shfl_xor(v, 4, 64);  // the four row chunks b of the block
        v += __shfl_xor(v, 8, 64);
        if (((lane >> 2) & 3) == 0) {
          const int slot = p * 16 + (lane >> 4) * 4 + (lane & 3);  // element (i, j) of block pair p
          partials[(size_t)slot * gridDim.x + blockIdx.x] = v;
        }
        q++;
      }
    }
  }
}

// ZP > 0: the first `kpend` (<= 4 * ZP) columns are L-SR1 columns that have not been materialised yet:
// V.p[j] = Y_j, S.p[j] = S_j, the staged value is Z_j = Y_j - b0 S_j, which is also written to Zout.p[j] for the
// later panel passes of the iteration (saves the separate 3-pass rebuild of Z).
template <int NG, int ZP, int OCC>
__global__ void __launch_bounds__(kBlock, OCC)
    wgram_kernel(const double *__restrict__ d, PtrTable V, int nv, int64_t n, int64_t ntiles,
                 double *__restrict__ partials, PtrTable S, PtrTableW Zout, int kpend, double b0, int tcol,
                 int ablate) {
  constexpr int M = 4 * NG;
  constexpr int NQ = GramPlanHolder<NG>::NQ;
  extern __shared__ double lds[];  // [M][kGramLd] panel tile, then [kGramTile] weights
  double *pt = lds;
  double *dw = lds + M * kGramLd;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: pointers stay in SGPRs

  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;

  // zero the padded columns once (they are never written by the staging loop)
  for (int idx = tid; idx < (M - nv) * kGramLd; idx += kBlock) pt[nv * kGramLd + idx] = 0.0;

  // wave w stages columns w, w+4, ...; lane l holds rows 2l, 2l+1 of the 128-row tile.  Columns beyond nv
  // re-read the last column (never written to LDS), so that the loads carry no branch and are all issued
  // before the first wait.
  const double *colp[NG];
#pragma unroll
  for (int it = 0; it < NG; it++) {
    const int j = wave + 4 * it;
    colp[it] = V.p[j < nv ? j : nv - 1];
  }
  const double *scol[ZP > 0 ? ZP : 1];
  double *zcol[ZP > 0 ? ZP : 1];
#pragma unroll
  for (int it = 0; it < (ZP > 0 ? ZP : 1); it++) {
    const int j = wave + 4 * it;
    scol[it] = (ZP > 0 && kpend > 0) ? S.p[j < kpend ? j : kpend - 1] : nullptr;
    zcol[it] = (ZP > 0 && j < kpend) ? Zout.p[j] : nullptr;
  }
  const int64_t ilast = ((n - 1) >> 1) << 1;
  // Software pipeline: the global loads of the NEXT tile are issued (all of them, back to back) before the
  // current tile is multiplied, and are only waited for at the next LDS store.  Row pairs past n are clamped to
  // the last in-range pair for the load and staged as zeros (the pad element of an odd-length vector is 0.0).
  f64x2 buf[NG];
  f64x2 sbuf[ZP > 0 ? ZP : 1];
  int64_t pre_i = 0;
  bool pre_in = false;
  f64x2 dbuf = (f64x2){0.0, 0.0};
#define PO_GRAM_PREFETCH(TILE)                                                               \
  {                                                                                          \
    int64_t _i = (TILE) * kGramTile + 2 * lane;                                              \
    pre_in = (_i < n);                                                                       \
    if (!pre_in) _i = ilast;                                                                 \
    pre_i = _i;                                                                              \
    _Pragma("unroll") for (int it = 0; it < NG; it++) buf[it] = ld_nt(colp[it] + _i);        \
    if (ZP > 0) {                                                                            \
      _Pragma("unroll") for (int it = 0; it < ZP; it++) sbuf[it] = ld_nt(scol[it] + _i);     \
    }                                                                                        \
    dbuf = *reinterpret_cast<const f64x2 *>(d + _i);                                         \
  }
  if ((int64_t)blockIdx.x < ntiles) PO_GRAM_PREFETCH((int64_t)blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int it = 0; it < NG; it++) {
      const int j = wave + 4 * it;
      f64x2 v = buf[it];
      if (ZP > 0 && it < ZP && j < kpend) {
        v.x -= b0 * sbuf[it].x;
        v.y -= b0 * sbuf[it].y;
        if (pre_in && zcol[it]) __builtin_nontemporal_store(v, reinterpret_cast<f64x2 *>(zcol[it] + pre_i));
      }
      if (!pre_in) v = (f64x2){0.0, 0.0};
      if (ablate == 2) {  // tuning: no LDS staging
        if (v.x == 1.2345e301) acc[0] += v.y;
      } else if (j < nv) *reinterpret_cast<f64x2 *>(pt + j * kGramLd + 2 * lane) = v;
    }
    if (wave == 0) *reinterpret_cast<f64x2 *>(dw + 2 * lane) = pre_in ? dbuf : (f64x2){0.0, 0.0};
    __syncthreads();
    if (tile + gridDim.x < ntiles && ablate != 3) PO_GRAM_PREFETCH(tile + gridDim.x);
    if (ablate == 1 || ablate == 2) continue;  // tuning: no matrix work
    switch (wave) {
      case 0: gram_tile<NG, 0>(pt, dw, lane, tcol, acc); break;
      case 1: gram_tile<NG, 1>(pt, dw, lane, tcol, acc); break;
      case 2: gram_tile<NG, 2>(pt, dw, lane, tcol, acc); break;
      default: gram_tile<NG, 3>(pt, dw, lane, tcol, acc); break;
    }
  }
#undef PO_GRAM_PREFETCH
  switch (wave) {
    case 0: gram_store<NG, 0>(acc, lane, partials); break;
    case 1: gram_store<NG, 1>(acc, lane, partials); break;
    case 2: gram_store<NG, 2>(acc, lane, partials); break;
    default: gram_store<NG, 3>(acc, lane, partials); break;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Producer / consumer form (default for NG <= 16): 8 wavefronts per workgroup, ONE workgroup per CU.
//   waves 4-7 (producers) keep three tiles of their column slice in flight in registers, stage the oldest into
//     one of two LDS tile buffers (forming the L-SR1 columns on the way) and immediately re-issue the loads of the
//     tile three ahead: the HBM stream never waits for matrix work;
//   waves 0-3 (consumers, one per SIMD) run the matrix instructions of the tile staged one step earlier.
// One workgroup barrier per tile: after it, buffer it%2 is full (read by the consumers during step `it`) and
// buffer (it+1)%2 is free (filled by the producers during step `it`).
// ---------------------------------------------------------------------------------------------------------
constexpr int kGramDepth = 3;  // tiles a producer keeps in flight
#ifdef PO_WGRAM_NO_FULL_STAGE  // (A/B builds: the staging with per-lane selects on every tile)
#define PO_WGRAM_FULL_STAGE 0
#else
#define PO_WGRAM_FULL_STAGE 1
#endif

// PAROPT_AMD_WGRAM_ABLATE=16: cycle stamps of workgroup 0 (s_memtime), read by po_debug_wgram_stamps:
// [0] consumer barrier wait, [1] consumer matrix work, [2] producer staging (incl. the wait for its loads),
// [3] producer load issue, [4] producer barrier wait, [5] tiles, [6] / [7] s_memtime / s_memrealtime ticks of the loop
__device__ unsigned long long g_wgram_stamp[8];

// ---- structured sparse Jacobian riding in the pass (round 4) ---------------------------------------------------
// One sparse constraint per group of `nw` consecutive variables (period nw + skip, first group at variable 0; the
// GroupMap of wcon.hpp).  The panel image U_j = alpha Aw (d o V_j) -- group sums of the weighted columns -- is what
// group_panel_tiled_kernel (wcon.hip) computes in a pass of its own over the same panel; with the tile pitch cut to
// whole groups (G groups = trows <= 128 rows per tile, the rest of the tile staged as zeros) the consumers take the
// sums from the LDS tile they have just multiplied: the panel is read once instead of twice.  Same arithmetic as
// group_panel_tiled_kernel -- rounded products d * v added in index order, then * alpha -- so U has the same bits.
// Variables past the last group are covered by ordinary 128-row tiles behind the group tiles.
struct GramGeom {
  int64_t ngt = 0;    // tiles of whole groups (0: plain 128-row tiling from row 0)
  int64_t rg = 0;     // rows covered by the group tiles
  int64_t nwcon = 0;  // groups
  int trows = kGramTile, G = 0, period = 0, nw = 0, ncols = 0;
  double alpha = 0.0;
};
// (branch-free: with branches here hipcc loses count of the producers' outstanding loads and waits for ALL of them
// before staging a tile -- vmcnt(0) instead of vmcnt(newer tiles) -- which serialises the three-tile prefetch)
__device__ __forceinline__ void gram_tile_geom(const GramGeom &gg, int64_t tile, int64_t &row0, int &nrows) {
  const bool grp = tile < gg.ngt;
  const int64_t rowg = tile * gg.trows, rowp = gg.rg + (tile - gg.ngt) * kGramTile;
  const int64_t left = gg.rg - rowg;
  const int nrg = left < gg.trows ? (int)left : gg.trows;
  row0 = grp ? rowg : rowp;
  nrows = grp ? nrg : kGramTile;
}
// The sums of one (column u, group gi) pair per consumer lane, in two phases around the tile's matrix work: the
// operands (nw rows of the column and their weights, <= kGramGroupNw each) are REQUESTED from LDS before the matrix
// instructions and consumed after them, so the LDS latency of the sums hides behind the tile's matrix work and what is
// left on the consumers' critical path is the chain of nw ordered adds.  (Measured on the way: summing straight from LDS
// after the matrix work -- 20 dependent load/add steps -- made the pass 1.6 us per tile slower whatever the panel
// width; indexing the kernel-argument pointer table per lane is a vector load from HOST memory per tile; and ONE flat
// store in the kernel makes hipcc wait for vmcnt(0) everywhere, the producers' prefetch included.)
constexpr int kGramGroupNw = 24;  // widest group the fused form takes (registers: 2 x 24 doubles per lane)
struct GramGroupOps {
  double p[kGramGroupNw], w[kGramGroupNw];
};
__device__ __forceinline__ void gram_groups_request(const double *__restrict__ pt, const double *__restrict__ dw,
                                                    const GramGeom &gg, int u, int gi, GramGroupOps &o) {
  // (ds_read_b64 each: two rows per ds_read_b128 was tried and is 70 % slower on the whole pass)
  const double *row = pt + u * kGramLd + gi * gg.period;
  const double *wr = dw + gi * gg.period;
#pragma unroll
  for (int k = 0; k < kGramGroupNw; k++) {
    if (k < gg.nw) {  // (wave-uniform)
      o.p[k] = row[k];
      o.w[k] = wr[k];
    }
  }
}
__device__ __forceinline__ double gram_groups_sum(const GramGeom &gg, const GramGroupOps &o) {
#pragma clang fp contract(off)  // rounded products, then ordered adds: the bits of group_panel_tiled_kernel
  double sacc = 0.0;
#pragma unroll
  for (int k = 0; k < kGramGroupNw; k++) {
    if (k < gg.nw) {
      const double prod = o.w[k] * o.p[k];
      sacc = sacc + prod;
    }
  }
  return gg.alpha * sacc;
}
// pairs beyond the first 256 of a tile (wide panels with many small groups): straight from LDS after the matrix work
__device__ __forceinline__ void gram_groups_rest(const double *__restrict__ pt, const double *__restrict__ dw,
                                                 const GramGeom &gg, double *const *Utab, int64_t g0, int ng, int ctid) {
#pragma clang fp contract(off)
  typedef __attribute__((address_space(1))) double gdouble;
  for (int pair = ctid + 256; pair < ng * gg.ncols; pair += 256) {
    const int u = pair / ng, gi = pair - u * ng;
    const double *row = pt + u * kGramLd + gi * gg.period;
    const double *wr = dw + gi * gg.period;
    double sacc = 0.0;
    for (int k = 0; k < gg.nw; k++) {
      const double prod = wr[k] * row[k];
      sacc = sacc + prod;
    }
    gdouble *up = (gdouble *)Utab[u];
    up[g0 + gi] = gg.alpha * sacc;
  }
}

template <int NG, int ZP>
struct GramProducer {
  f64x2 buf[kGramDepth][NG];
  f64x2 sbuf[kGramDepth][ZP > 0 ? ZP : 1];
  f64x2 dbuf[kGramDepth];
  int64_t row[kGramDepth];
  bool in[kGramDepth];
};

template <int NG, int ZP, int R>
__device__ __forceinline__ void gram_pc_load(GramProducer<NG, ZP> &P, const double *const (&colp)[NG],
                                             const double *const (&scol)[ZP > 0 ? ZP : 1], const double *d,
                                             int64_t tile, int64_t ntiles, int64_t n, int64_t ilast, int lane,
                                             const GramGeom &gg) {
  if (tile >= ntiles) {
    P.in[R] = false;
    return;
  }
  int64_t row0;
  int nrows;
  gram_tile_geom(gg, tile, row0, nrows);
  int64_t i = row0 + 2 * lane;
  const bool in = (2 * lane < nrows) && (i < n);
  // (lanes outside the tile: any in-range pair will do, its value is not used -- with group tiles one of THIS tile's,
  // whose lines the instruction requests anyway)
  int64_t alt = row0 + 2 * (lane & 3);
  alt = alt > ilast ? ilast : alt;
  if (!in) i = gg.ngt > 0 ? alt : ilast;
  P.in[R] = in;
  P.row[R] = i;
#pragma unroll
  for (int it = 0; it < NG; it++) P.buf[R][it] = ld_nt(colp[it] + i);
  if (ZP > 0) {
#pragma unroll
    for (int it = 0; it < ZP; it++) P.sbuf[R][it] = ld_nt(scol[it] + i);
  }
  P.dbuf[R] = *reinterpret_cast<const f64x2 *>(d + i);
}

// FULL: every lane of the wavefront holds rows of the tile (all tiles but a panel's last one, and the tiles of whole
// groups): the per-lane "outside the tile -> stage zero" selects -- four 32-bit selects per column and tile, a third of
// the producers' vector instructions, issued on the SIMD the matrix instructions of the consumer wavefront run on --
// are left out (round 5; the caller tests `in` across the wavefront, a scalar branch).
template <int NG, int ZP, int R, bool FULL = false>
__device__ __forceinline__ void gram_pc_stage(GramProducer<NG, ZP> &P, double *__restrict__ pt,
                                              double *__restrict__ dw, double *const (&zcol)[ZP > 0 ? ZP : 1], int pw,
                                              int nv, int kpend, double b0, int lane) {
  const bool in = FULL ? true : P.in[R];
#pragma unroll
  for (int it = 0; it < NG; it++) {
    const int j = pw + 4 * it;
    f64x2 v = P.buf[R][it];
    if (ZP > 0 && it < ZP && j < kpend) {
      v.x -= b0 * P.sbuf[R][it].x;
      v.y -= b0 * P.sbuf[R][it].y;
      if (in && zcol[it]) __builtin_nontemporal_store(v, reinterpret_cast<f64x2 *>(zcol[it] + P.row[R]));
    }
    if (!FULL && !in) v = (f64x2){0.0, 0.0};
    if (j < nv) *reinterpret_cast<f64x2 *>(pt + j * kGramLd + 2 * lane) = v;
  }
  if (pw == 0) *reinterpret_cast<f64x2 *>(dw + 2 * lane) = in ? P.dbuf[R] : (f64x2){0.0, 0.0};
}

template <int NG, int ZP, int RS, int GS>
__global__ void __launch_bounds__(512, 1)
    wgram_pc_kernel(const double *__restrict__ d, PtrTable V, int nv, int64_t n, int64_t ntiles,
                    double *__restrict__ partials, PtrTable S, PtrTableW Zout, int kpend, double b0, int tcol,
                    int ablate, int prio, GramGeom gg, PtrTableW Ugs) {
  constexpr int M = 4 * NG;
  constexpr int NQ = GramPlanHolder<NG>::NQ;
  const bool want_stamps = (ablate & 16) != 0;  // PAROPT_AMD_WGRAM_ABLATE: 16 = cycle stamps, low bits = what is cut
  ablate &= 15;
  if constexpr (GS == 0) gg = GramGeom();  // plain tiling, folded at compile time
  constexpr int kBufDoubles = M * kGramLd + kGramTile;  // panel tile, then the row weights
  extern __shared__ double lds[];                       // two tile buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // zero the padded columns of both buffers once (never written by the staging)
  for (int b2 = 0; b2 < 2; b2++)
    for (int idx = tid; idx < (M - nv) * kGramLd; idx += 512) lds[b2 * kBufDoubles + nv * kGramLd + idx] = 0.0;
  double **utab = reinterpret_cast<double **>(lds + 2 * kBufDoubles);  // GS: the panel image's output pointers
  if constexpr (GS != 0) {
    if (tid < kMaxPanel) utab[tid] = Ugs.p[tid];
  }
  // tiles of this workgroup: blockIdx.x + it * gridDim.x, it = 0 .. nt-1
  // GS = 0: tile blockIdx.x + it * gridDim.x.  GS: a workgroup takes CONSECUTIVE tiles -- the 120-row (15-line) tiles
  // of e.g. groups of 20 do not end on cache-line boundaries, and with neighbouring tiles on different CUs every
  // column of every tile fetched two partial lines twice (measured: +40 % on the pass)
  int64_t first = blockIdx.x, stride = gridDim.x;
  int64_t nt = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
  if constexpr (GS != 0) {
    const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    first = blockIdx.x * per;
    stride = 1;
    nt = ntiles - first;
    nt = nt < 0 ? 0 : (nt < per ? nt : per);
  }
  __syncthreads();
  if (wave >= 4) {
    // ------------------------------------------------ producers ------------------------------------------------
    // The producers' few vector instructions (addresses, staging) must not queue behind the consumers' matrix
    // instructions on the shared SIMD: static priority for the whole loop (MI355X_MICROARCH.md, two waves per SIMD)
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (prio == 2) __builtin_amdgcn_s_setprio(2);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    const int pw = wave - 4;
    const double *colp[NG];
#pragma unroll
    for (int it = 0; it < NG; it++) {
      const int j = pw + 4 * it;
      colp[it] = V.p[j < nv ? j : nv - 1];
    }
    const double *scol[ZP > 0 ? ZP : 1];
    double *zcol[ZP > 0 ? ZP : 1];
#pragma unroll
    for (int it = 0; it < (ZP > 0 ? ZP : 1); it++) {
      const int j = pw + 4 * it;
      scol[it] = (ZP > 0 && kpend > 0) ? S.p[j < kpend ? j : kpend - 1] : nullptr;
      zcol[it] = (ZP > 0 && j < kpend) ? Zout.p[j] : nullptr;
    }
    const int64_t ilast = ((n - 1) >> 1) << 1;
    GramProducer<NG, ZP> P;
    gram_pc_load<NG, ZP, 0>(P, colp, scol, d, first, ntiles, n, ilast, lane, gg);
    gram_pc_load<NG, ZP, 1>(P, colp, scol, d, first + stride, ntiles, n, ilast, lane, gg);
    gram_pc_load<NG, ZP, 2>(P, colp, scol, d, first + 2 * stride, ntiles, n, ilast, lane, gg);
    // step `it` (ring slot it % 3, LDS buffer it % 2): stage tile `it`, reload the slot with tile it + 3, barrier
    unsigned long long st_stage = 0, st_load = 0, st_wait = 0;
    const bool stamp = want_stamps && blockIdx.x == 0 && wave == 4;
    const unsigned long long st_c0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long st_r0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
    // GS == 2 (round 5): the group sums of the panel image are taken by the PRODUCER waves, off the consumers' critical
    // path (matrix work only): behind the barrier that publishes tile t the 256 producer lanes request the operands of
    // their (column, group) pair from tile t's buffer, stage tile t + 1 and issue the loads of tile t + 4 while the
    // requests are in flight, then add (same rounded products, same order: the bits of group_panel_tiled_kernel) and
    // store.  Branch-free like the rest of the producers' loop (a lane without a pair reads clamped addresses and
    // stores 0.0 to the pad element behind the first output column), so that hipcc keeps exact counts of the
    // outstanding prefetch loads; the first step "finishes" a tile that does not exist the same way.
    [[maybe_unused]] GramGroupOps gops;
    [[maybe_unused]] int g2_u = 0, g2_gi = 0, g2_uc = 0;
    [[maybe_unused]] bool g2_lane = false;
    if constexpr (GS == 2) {
      const int ptid = tid - 256;
      g2_u = ptid / gg.G;
      g2_gi = ptid - g2_u * gg.G;
      g2_lane = g2_u < gg.ncols;
      g2_uc = g2_lane ? g2_u : 0;
    }
    typedef __attribute__((address_space(1))) double g2double;
    [[maybe_unused]] g2double *g2_up = nullptr;
    if constexpr (GS == 2) g2_up = (g2double *)utab[g2_uc];  // (published by the barrier above)
#define PO_GS2_REQUEST(BUF) gram_groups_request((BUF), (BUF) + M * kGramLd, gg, g2_uc, g2_gi, gops)
#define PO_GS2_FINISH(TIDX)                                                                                   \
  {                                                                                                           \
    const int64_t _tile = first + (TIDX) * stride;                                                            \
    const int64_t _g0 = _tile * gg.G;                                                                         \
    const bool _mine = g2_lane && (TIDX) >= 0 && _tile < gg.ngt && _g0 + g2_gi < gg.nwcon;                    \
    const double _sum = gram_groups_sum(gg, gops);                                                            \
    const int64_t _idx = _mine ? _g0 + g2_gi : gg.nwcon;                                                      \
    g2_up[_idx] = _mine ? _sum : 0.0;                                                                         \
  }
    // (Round 5, measured and NOT kept: with the loads under `if (tile < ntiles)` and the steps under `if (it + R < nt)`
    // hipcc's wait-count bookkeeping merges "issued" and "not issued" at the joins and from then on takes every load in
    // flight for kGramDepth-1 tiles younger than it is: tile t is staged behind vmcnt(NG) instead of
    // vmcnt(NG + 2 (NG + 1)), i.e. behind the loads of tiles t + 1 and t + 2 as well.  A branch-free version (clamped
    // loads past the end, whole rounds of three unconditional steps) does get the exact counts -- and is not faster:
    // 43.7 / 43.9 against 43.0 / 43.9 it/s at config 3, 180.5 / 181.0 against 183.2 / 181.0 at config 4, 308 against 313
    // at config 2, one call, profiles/r05_ab_wgram_producers_pc64.jsonl.  A tile period is longer than the HBM latency,
    // so the loads issued one step ago have landed either way.  The 64-row form below is written branch-free.)
#define PO_PC_STEP(R)                                                                                         \
  if (it + (R) < nt) {                                                                                        \
    double *bt = lds + (size_t)((it + (R)) & 1) * kBufDoubles;                                                \
    if constexpr (GS == 2) PO_GS2_REQUEST(lds + (size_t)((it + (R) + 1) & 1) * kBufDoubles);                  \
    const unsigned long long _t0 = stamp ? __builtin_amdgcn_s_memtime() : 0;                                  \
    if (ablate == 2) {                                                                                        \
      if (P.buf[R][0].x == 1.2345e301) bt[0] = P.buf[R][NG - 1].y;                                            \
    } else if (PO_WGRAM_FULL_STAGE && __all(P.in[R])) {                                                       \
      gram_pc_stage<NG, ZP, (R), true>(P, bt, bt + M * kGramLd, zcol, pw, nv, kpend, b0, lane);               \
    } else {                                                                                                  \
      gram_pc_stage<NG, ZP, (R), false>(P, bt, bt + M * kGramLd, zcol, pw, nv, kpend, b0, lane);              \
    }                                                                                                         \
    if (stamp) __builtin_amdgcn_s_waitcnt(0);                                                                 \
    const unsigned long long _t1 = stamp ? __builtin_amdgcn_s_memtime() : 0;                                  \
    if (ablate != 3) gram_pc_load<NG, ZP, (R)>(P, colp, scol, d, first + (it + (R) + kGramDepth) * stride, ntiles, n, ilast, lane, gg); \
    const unsigned long long _t2 = stamp ? __builtin_amdgcn_s_memtime() : 0;                                  \
    if constexpr (GS == 2) PO_GS2_FINISH(it + (R) - 1);                                                       \
    __syncthreads();                                                                                          \
    if (stamp) {                                                                                              \
      const unsigned long long _t3 = __builtin_amdgcn_s_memtime();                                            \
      st_stage += _t1 - _t0;                                                                                  \
      st_load += _t2 - _t1;                                                                                   \
      st_wait += _t3 - _t2;                                                                                   \
    }                                                                                                         \
  }
    for (int64_t it = 0; it < nt; it += kGramDepth) {
      PO_PC_STEP(0)
      PO_PC_STEP(1)
      PO_PC_STEP(2)
    }
#undef PO_PC_STEP
    if constexpr (GS == 2) {
      // the sums of the last tile (staged and published by the last step's barrier; the consumers overwrite the tile
      // buffers only behind the trailing barrier below)
      if (nt > 0) {
        const double *bl = lds + (size_t)((nt - 1) & 1) * kBufDoubles;
        PO_GS2_REQUEST(bl);
        PO_GS2_FINISH(nt - 1);
      }
    }
#undef PO_GS2_REQUEST
#undef PO_GS2_FINISH
    if (stamp && lane == 0) {
      g_wgram_stamp[2] = st_stage;
      g_wgram_stamp[3] = st_load;
      g_wgram_stamp[4] = st_wait;
      g_wgram_stamp[5] = (unsigned long long)nt;
      g_wgram_stamp[6] = __builtin_amdgcn_s_memtime() - st_c0;      // shader-clock ticks of the whole loop
      g_wgram_stamp[7] = __builtin_amdgcn_s_memrealtime() - st_r0;  // 100 MHz ticks of the whole loop
    }
    __syncthreads();  // matches the consumers' trailing barrier
    if (RS) __syncthreads();  // ... and the one between their LDS partials and the cross-wave sum
  } else {
    // ------------------------------------------------ consumers ------------------------------------------------
    constexpr int NACC = RS ? NG * (NG + 1) / 2 : NQ;
    double acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; q++) acc[q] = 0.0;
    unsigned long long sc_wait = 0, sc_work = 0;
    const bool cstamp = want_stamps && blockIdx.x == 0 && wave == 0;
    // (GS) this lane's pair of a full tile of G groups; the last group tile may hold fewer groups (recomputed there)
    [[maybe_unused]] int gs_u = 0, gs_gi = 0;
    if constexpr (GS == 1) {
      gs_u = tid / gg.G;
      gs_gi = tid - gs_u * gg.G;
    }
    for (int64_t it = 0; it < nt; it++) {
      const unsigned long long _t0 = cstamp ? __builtin_amdgcn_s_memtime() : 0;
      __syncthreads();  // tile `it` is staged in buffer it % 2
      const unsigned long long _t1 = cstamp ? __builtin_amdgcn_s_memtime() : 0;
      const double *bt = lds + (size_t)(it & 1) * kBufDoubles;
      if (ablate == 1) continue;  // tuning: no matrix work
      // (GS) phase 1: request the operands of this lane's group sum before the matrix work
      [[maybe_unused]] GramGroupOps gops;  // (GS == 1: the consumers take the sums)
      [[maybe_unused]] bool gs_on = false, gs_mine = false;
      [[maybe_unused]] int gs_ng = 0, pu = 0, pgi = 0;
      [[maybe_unused]] int64_t gs_g0 = 0;
      if constexpr (GS == 1) {
        const int64_t tile = first + it * stride;
        gs_on = tile < gg.ngt;
        if (gs_on) {
          gs_g0 = tile * gg.G;
          gs_ng = (int)((gg.nwcon - gs_g0) < gg.G ? (gg.nwcon - gs_g0) : gg.G);
          pu = gs_u;
          pgi = gs_gi;
          if (gs_ng != gg.G) {
            pu = tid / gs_ng;
            pgi = tid - pu * gs_ng;
          }
          gs_mine = tid < gs_ng * gg.ncols;
          if (!RS && gs_mine && ablate != 5) gram_groups_request(bt, bt + M * kGramLd, gg, pu, pgi, gops);
        }
      }
      if constexpr (RS) {
        if constexpr (GS == 1) {
          auto request = [&]() {
            if (gs_on && gs_mine && ablate != 5) gram_groups_request(bt, bt + M * kGramLd, gg, pu, pgi, gops);
          };
          gram_tile_rows<NG>(bt, bt + M * kGramLd, lane, wave, tcol, acc, request);
        } else {
          gram_tile_rows<NG>(bt, bt + M * kGramLd, lane, wave, tcol, acc);
        }
      } else {
        switch (wave) {
          case 0: gram_tile<NG, 0>(bt, bt + M * kGramLd, lane, tcol, acc); break;
          case 1: gram_tile<NG, 1>(bt, bt + M * kGramLd, lane, tcol, acc); break;
          case 2: gram_tile<NG, 2>(bt, bt + M * kGramLd, lane, tcol, acc); break;
          default: gram_tile<NG, 3>(bt, bt + M * kGramLd, lane, tcol, acc); break;
        }
      }
      // (GS) phase 2: the ordered sum and its store (a GLOBAL store: the pointer from LDS is generic)
      if constexpr (GS == 1) {
        if (gs_on) {
          if (gs_mine && ablate != 6) {  // (tuning: 4 = no store, 5 = no operand requests, 6 = neither sums nor store)
            typedef __attribute__((address_space(1))) double gdouble;
            gdouble *up = (gdouble *)utab[pu];
            const double sum = gram_groups_sum(gg, gops);
            if (ablate != 4 || sum == 1.2345e301) up[gs_g0 + pgi] = sum;
          }
          if (gs_ng * gg.ncols > 256) gram_groups_rest(bt, bt + M * kGramLd, gg, utab, gs_g0, gs_ng, tid);
        }
      }
      if (cstamp) {
        // (the accumulators are consumed only at the end: read one so that the stamp follows the matrix work)
        if (acc[0] == 1.2345e301) sc_work++;
        const unsigned long long _t2 = __builtin_amdgcn_s_memtime();
        sc_wait += _t1 - _t0;
        sc_work += _t2 - _t1;
      }
    }
    if (cstamp && lane == 0) {
      g_wgram_stamp[0] = sc_wait;
      g_wgram_stamp[1] = sc_work;
    }
    __syncthreads();  // the last tile has been consumed by every wave: the tile buffers are free
    if constexpr (RS) {
      constexpr int NP = NG * (NG + 1) / 2;
      gram_store_rows<NG>(acc, lane, wave, lds);
      __syncthreads();
      for (int slot = tid; slot < NP * 16; slot += 256)
        partials[(size_t)slot * gridDim.x + blockIdx.x] =
            ((lds[slot] + lds[NP * 16 + slot]) + lds[2 * NP * 16 + slot]) + lds[3 * NP * 16 + slot];
    } else {
      switch (wave) {
        case 0: gram_store<NG, 0>(acc, lane, partials); break;
        case 1: gram_store<NG, 1>(acc, lane, partials); break;
        case 2: gram_store<NG, 2>(acc, lane, partials); break;
        default: gram_store<NG, 3>(acc, lane, partials); break;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Producer / consumer form for WIDE panels (17-20 column groups = 65-80 columns, e.g. c = 32 constraints + an
// L-BFGS(20) memory + the pre-weighted column: 73), round 5.  Two 128-row tile buffers of 80 columns do not fit the
// 160 KB of LDS, so these panels ran the single-role kernel with ONE wavefront per SIMD (0.34 of the HBM peak: load
// wait, staging and matrix work in series).  Here the tile is 64 rows (row stride 72 doubles == 8 mod 32, the same
// banking as 136), two buffers = 93 KB at 80 columns.  A producer wavefront loads TWO columns per instruction (lanes
// 0-31 the 32 row pairs of column j, lanes 32-63 those of column j + 4) and keeps four tiles in flight; the
// consumers split the OUTPUT (column groups dealt to the four wavefronts, as in the single-role kernel), four 16-row
// steps per tile.  Same slot layout of the partial sums; the summation order differs from the single-role form's
// (64- instead of 128-row tiles), which only panels of this width ever see.
// ---------------------------------------------------------------------------------------------------------
constexpr int kG64Tile = 64;
constexpr int kG64Ld = kG64Tile + 8;

template <int NG, int W>
__device__ __forceinline__ void gram64_consume(const double *__restrict__ lds, int64_t nt,
                                               double *__restrict__ partials, int tcol, int lane, bool cstamp) {
  constexpr int M = 4 * NG;
  constexpr int NQ = GramPlanHolder<NG>::NQ;
  constexpr int kBufDoubles = M * kG64Ld + kG64Tile;
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;
  unsigned long long sc_wait = 0, sc_work = 0;
  for (int64_t it = 0; it < nt; it++) {
    const unsigned long long _t0 = cstamp ? __builtin_amdgcn_s_memtime() : 0;
    __syncthreads();  // tile `it` is staged in buffer it % 2
    const unsigned long long _t1 = cstamp ? __builtin_amdgcn_s_memtime() : 0;
    const double *bt = lds + (size_t)(it & 1) * kBufDoubles;
    gram_tile<NG, W, kG64Tile, kG64Ld>(bt, bt + M * kG64Ld, lane, tcol, acc);
    if (cstamp) {
      if (acc[0] == 1.2345e301) sc_work++;  // (read an accumulator: the stamp follows the matrix work)
      const unsigned long long _t2 = __builtin_amdgcn_s_memtime();
      sc_wait += _t1 - _t0;
      sc_work += _t2 - _t1;
    }
  }
  if (cstamp && lane == 0) {
    g_wgram_stamp[0] = sc_wait;
    g_wgram_stamp[1] = sc_work;
  }
  __syncthreads();
  gram_store<NG, W>(acc, lane, partials);
}

template <int NH, int D>
struct Gram64Producer {
  f64x2 buf[D][NH];
  f64x2 dbuf[D];
  bool in[D];
};

template <int NG>
__global__ void __launch_bounds__(512, 1)
    wgram_pc64_kernel(const double *__restrict__ d, PtrTable V, int nv, int64_t n, int64_t ntiles,
                      double *__restrict__ partials, int tcol, int prio, int want_stamps) {
  constexpr int M = 4 * NG;
  constexpr int NH = (NG + 1) / 2;      // loads per producer wavefront and tile (two columns each)
  constexpr int D = 4;                  // tiles a producer keeps in flight (registers: 4 (NH + 1) per tile and lane)
  constexpr int kBufDoubles = M * kG64Ld + kG64Tile;
  extern __shared__ double lds[];  // two tile buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int b2 = 0; b2 < 2; b2++)
    for (int idx = tid; idx < (M - nv) * kG64Ld; idx += 512) lds[b2 * kBufDoubles + nv * kG64Ld + idx] = 0.0;
  const int64_t first = blockIdx.x, stride = gridDim.x;
  const int64_t nt = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
  __syncthreads();
  if (wave >= 4) {
    // ------------------------------------------------ producers ------------------------------------------------
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (prio == 2) __builtin_amdgcn_s_setprio(2);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    const int pw = wave - 4, half = lane >> 5, rp = lane & 31;
    const int j0 = pw + 4 * half;  // this lane's column of load h: j0 + 8 h
    const double *colp[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) colp[h] = V.p[j0 + 8 * h < nv ? j0 + 8 * h : nv - 1];
    const int64_t ilast = ((n - 1) >> 1) << 1;
    Gram64Producer<NH, D> P;
    // (loads unconditional -- a tile past the end re-requests the last lines and is staged as zeros -- and whole rounds
    // of D unconditional steps: hipcc then counts the loads in flight exactly, see the note in wgram_pc_kernel)
#define PO_G64_LOAD(R, TILE)                                                             \
  {                                                                                      \
    const int64_t _t = (TILE);                                                           \
    int64_t _i = _t * kG64Tile + 2 * rp;                                                 \
    const bool _in = _t < ntiles && _i < n;                                              \
    if (!_in) _i = ilast;                                                                \
    P.in[R] = _in;                                                                       \
    _Pragma("unroll") for (int h = 0; h < NH; h++) P.buf[R][h] = ld_nt(colp[h] + _i);    \
    P.dbuf[R] = *reinterpret_cast<const f64x2 *>(d + _i);                                \
  }
    unsigned long long st_stage = 0, st_load = 0, st_wait = 0;
    const bool stamp = want_stamps && blockIdx.x == 0 && wave == 4;  // (PAROPT_AMD_WGRAM_ABLATE=16, as in wgram_pc_kernel)
    const unsigned long long st_c0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long st_r0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
#define PO_G64_STEP(R)                                                                   \
  {                                                                                      \
    double *bt = lds + (size_t)((it + (R)) & 1) * kBufDoubles;                           \
    const bool _in = P.in[R];                                                            \
    const unsigned long long _t0 = stamp ? __builtin_amdgcn_s_memtime() : 0;             \
    if (PO_WGRAM_FULL_STAGE && __all(_in)) { /* (no per-lane selects: see gram_pc_stage) */ \
      _Pragma("unroll") for (int h = 0; h < NH; h++) {                                   \
        const int _j = j0 + 8 * h;                                                       \
        if (_j < nv) *reinterpret_cast<f64x2 *>(bt + _j * kG64Ld + 2 * rp) = P.buf[R][h]; \
      }                                                                                  \
    } else {                                                                             \
      _Pragma("unroll") for (int h = 0; h < NH; h++) {                                   \
        f64x2 v = P.buf[R][h];                                                           \
        if (!_in) v = (f64x2){0.0, 0.0};                                                 \
        const int _j = j0 + 8 * h;                                                       \
        if (_j < nv) *reinterpret_cast<f64x2 *>(bt + _j * kG64Ld + 2 * rp) = v;          \
      }                                                                                  \
    }                                                                                    \
    if (pw == 0 && half == 0)                                                            \
      *reinterpret_cast<f64x2 *>(bt + M * kG64Ld + 2 * rp) = _in ? P.dbuf[R] : (f64x2){0.0, 0.0}; \
    if (stamp) __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): the staging stores have left */ \
    const unsigned long long _t1 = stamp ? __builtin_amdgcn_s_memtime() : 0;             \
    PO_G64_LOAD(R, first + (it + (R) + D) * stride);                                     \
    const unsigned long long _t2 = stamp ? __builtin_amdgcn_s_memtime() : 0;             \
    __syncthreads();                                                                     \
    if (stamp) {                                                                         \
      const unsigned long long _t3 = __builtin_amdgcn_s_memtime();                       \
      st_stage += _t1 - _t0;                                                             \
      st_load += _t2 - _t1;                                                              \
      st_wait += _t3 - _t2;                                                              \
    }                                                                                    \
  }
    PO_G64_LOAD(0, first);
    PO_G64_LOAD(1, first + stride);
    PO_G64_LOAD(2, first + 2 * stride);
    if constexpr (D == 4) PO_G64_LOAD(3, first + 3 * stride);
    int64_t it = 0;
    for (; it + D <= nt; it += D) {
      PO_G64_STEP(0)
      PO_G64_STEP(1)
      PO_G64_STEP(2)
      if constexpr (D == 4) PO_G64_STEP(3)
    }
    if (it < nt) PO_G64_STEP(0)
    if (it + 1 < nt) PO_G64_STEP(1)
    if constexpr (D == 4) {
      if (it + 2 < nt) PO_G64_STEP(2)
    }
#undef PO_G64_STEP
#undef PO_G64_LOAD
    if (stamp && lane == 0) {
      g_wgram_stamp[2] = st_stage;
      g_wgram_stamp[3] = st_load;
      g_wgram_stamp[4] = st_wait;
      g_wgram_stamp[5] = (unsigned long long)nt;
      g_wgram_stamp[6] = __builtin_amdgcn_s_memtime() - st_c0;
      g_wgram_stamp[7] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
    __syncthreads();  // matches the consumers' trailing barrier
  } else {
    // ------------------------------------------------ consumers ------------------------------------------------
    // (one whole loop per wavefront role, dispatched once: with the role switch inside a shared loop the accumulators of
    // the four roles end up in disjoint register ranges -- 248 registers and spills at 19 groups against 156-195 here)
    switch (wave) {
      case 0: gram64_consume<NG, 0>(lds, nt, partials, tcol, lane, want_stamps && blockIdx.x == 0); break;
      case 1: gram64_consume<NG, 1>(lds, nt, partials, tcol, lane, false); break;
      case 2: gram64_consume<NG, 2>(lds, nt, partials, tcol, lane, false); break;
      default: gram64_consume<NG, 3>(lds, nt, partials, tcol, lane, false); break;
    }
  }
}

template <int NG>
static int wgram_pc64_launch_t(Ctx *c, const double *d, const PtrTable &pt, int nv, int64_t n, int tcol, int *grid_out) {
  const size_t lds = (size_t)2 * (4 * NG * kG64Ld + kG64Tile) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    PO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wgram_pc64_kernel<NG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t ntiles = (n + kG64Tile - 1) / kG64Tile;
  int64_t g = (int64_t)c->num_cu;  // one workgroup per CU
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  PO_TRY(ensure_partials(c, (size_t)g * (NG * (NG + 1) / 2) * 16));
  const int prio = dbg_switch(SW_WGRAM_PRIO, "PAROPT_AMD_WGRAM_PRIO", 2);
  const int stamps = (dbg_switch(SW_WGRAM_ABLATE, "PAROPT_AMD_WGRAM_ABLATE", 0) & 16) != 0;
  hipLaunchKernelGGL((wgram_pc64_kernel<NG>), dim3((int)g), dim3(512), lds, c->stream, d, pt, nv, n, ntiles,
                     c->d_partials, tcol, prio, stamps);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  *grid_out = (int)g;
  return PO_OK;
}

template <int NG, int ZP, int RS, int GS = 0>
static int wgram_pc_launch_t(Ctx *c, const double *d, const PtrTable &pt, int nv, int64_t n, int64_t ntiles,
                             const PtrTable &st, const PtrTableW &zt, int kpend, double b0, int tcol, int *grid_out,
                             const GramGeom *gg = nullptr, const PtrTableW *ut = nullptr) {
  const size_t lds = (size_t)2 * (4 * NG * kGramLd + kGramTile) * sizeof(double) + (GS ? kMaxPanel * sizeof(double *) : 0);
  static bool attr_set = false;
  if (!attr_set) {
    PO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wgram_pc_kernel<NG, ZP, RS, GS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int ablate = dbg_switch(SW_WGRAM_ABLATE, "PAROPT_AMD_WGRAM_ABLATE", 0);
  int64_t g = (int64_t)c->num_cu;  // one workgroup per CU
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  PO_TRY(ensure_partials(c, (size_t)g * (NG * (NG + 1) / 2) * 16));
  // producers at raised priority: their address arithmetic no longer queues behind the consumers' matrix
  // instructions on the shared SIMD (-4 % on the plain form, -1.7 % with the L-SR1 columns formed; 0 switches it off)
  const int prio = dbg_switch(SW_WGRAM_PRIO, "PAROPT_AMD_WGRAM_PRIO", 2);
  static const PtrTableW no_u = PtrTableW();
  hipLaunchKernelGGL((wgram_pc_kernel<NG, ZP, RS, GS>), dim3((int)g), dim3(512), lds, c->stream, d, pt, nv, n, ntiles,
                     c->d_partials, st, zt, kpend, b0, tcol, ablate, prio, gg ? *gg : GramGeom(), ut ? *ut : no_u);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  *grid_out = (int)g;
  return PO_OK;
}

template <int NG, int ZP, int OCC>
static int wgram_launch_t(Ctx *c, const double *d, const PtrTable &pt, int nv, int64_t n, int64_t ntiles,
                          const PtrTable &st, const PtrTableW &zt, int kpend, double b0, int tcol, int *grid_out) {
  const size_t lds = (size_t)(4 * NG * kGramLd + kGramTile) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    PO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wgram_kernel<NG, ZP, OCC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  static const int ablate = getenv("PAROPT_AMD_WGRAM_ABLATE") ? atoi(getenv("PAROPT_AMD_WGRAM_ABLATE")) : 0;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > OCC) per_cu = OCC;  // one wavefront of every resident workgroup per SIMD
  if (per_cu < 1) per_cu = 1;
  int64_t g = (int64_t)c->num_cu * per_cu;
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  PO_TRY(ensure_partials(c, (size_t)g * (NG * (NG + 1) / 2) * 16));
  hipLaunchKernelGGL((wgram_kernel<NG, ZP, OCC>), dim3((int)g), dim3(kBlock), lds, c->stream, d, pt, nv, n, ntiles,
                     c->d_partials, st, zt, kpend, b0, tcol, ablate);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  *grid_out = (int)g;
  return PO_OK;
}

static int wgram_groups(int nv) { return (nv + 3) / 4; }

// the tile geometry of a pass that also takes the group sums, or false when this panel / map does not qualify (the
// caller then runs group_panel_tiled_kernel as before)
static bool wgram_groups_geom(const GramGroups *g, int nv, int64_t n, int kpend, GramGeom *gg) {
  if (!g || g->nwcon <= 0 || g->ncols <= 0 || g->ncols > nv || !g->U || kpend > 0) return false;
  static const bool off = getenv("PAROPT_AMD_NO_GRAM_GROUPS") != nullptr;
  if (off) return false;
  const int64_t period = (int64_t)g->nw + g->skip;
  if (g->start != 0 || g->nw <= 0 || g->nw > kGramGroupNw || period > kGramTile) return false;
  if ((g->nwcon - 1) * period + g->nw > n) return false;
  int G = (int)(kGramTile / period);
  if (period & 1) G &= ~1;  // an even number of rows per tile: lanes hold row pairs
  if (G < 1) return false;
  int64_t rg = g->nwcon * period;
  if (rg > n) rg = n;
  if ((rg & 1) && rg != n) return false;  // the ordinary tiles behind the groups start on an even row
  const int NG = (nv + 3) / 4;
  if (NG > 9 || n < 4 * kGramTile) return false;  // the producer/consumer instantiations that carry the sums
  gg->ngt = (g->nwcon + G - 1) / G;
  gg->rg = rg;
  gg->nwcon = g->nwcon;
  gg->trows = G * (int)period;
  gg->G = G;
  gg->period = (int)period;
  gg->nw = g->nw;
  gg->ncols = g->ncols;
  gg->alpha = g->alpha;
  return true;
}

int k_wgram_launch(Ctx *c, const double *d, const double *const *V, int nv, int64_t n, int *nblocks,
                   int *nslots, const double *const *S, double *const *Zout, int kpend, double b0,
                   int preweighted_last, const GramGroups *groups, bool *groups_done) {
  if (groups_done) *groups_done = false;
  if (nv > kWgramMaxVecs || nv < 1) {
    set_error("wgram panel width %d outside 1..%d", nv, kWgramMaxVecs);
    return PO_ERR_ARG;
  }
  if (kpend > 12 || kpend > nv || (kpend > 0 && (!S || !Zout))) {
    set_error("wgram: %d pending columns cannot be materialised in the Gram pass", kpend);
    return PO_ERR_ARG;
  }
  count_bytes(c, nv + 1 + 2 * kpend, n);  // panel + weights (+ S read, Z written)
  const int tcol = preweighted_last ? nv - 1 : -1;
  const int NG = wgram_groups(nv);
  int64_t ntiles = (n + kGramTile - 1) / kGramTile;
  PtrTable pt, st;
  PtrTableW zt, ut;
  for (int j = 0; j < kMaxPanel; j++) {
    pt.p[j] = j < nv ? V[j] : nullptr;
    st.p[j] = (S && j < kpend) ? S[j] : nullptr;
    zt.p[j] = (Zout && j < kpend) ? Zout[j] : nullptr;
    ut.p[j] = nullptr;
  }
  int grid = 0;
  GramGeom gg;
  // resident wavefronts per SIMD the kernel is compiled for (register budget 512 / OCC per lane)
  static const int occ_env = getenv("PAROPT_AMD_WGRAM_OCC") ? atoi(getenv("PAROPT_AMD_WGRAM_OCC")) : 0;
  // PAROPT_AMD_WGRAM_PC=0: the single-role form (every wavefront loads, stages and multiplies) for all widths
  static const bool use_pc = !(getenv("PAROPT_AMD_WGRAM_PC") && atoi(getenv("PAROPT_AMD_WGRAM_PC")) == 0);
  // PAROPT_AMD_WGRAM_RS=0: consumers split the OUTPUT (block pairs) instead of the tile's rows (A/B switch)
  // measured in one process (tools/ab_switch.py, profiles/r03_ab_wgram.txt, n = 50 M, 43 columns): the row split is
  // 8 % faster than the output split on the plain form (2.93 vs 3.18 ms = 0.75 of the HBM peak) and 2 % with the L-SR1
  // columns formed in the pass -- once its operand fetches stay ds_read_b64 (see gram_tile_rows)
  const bool row_split = dbg_switch(SW_WGRAM_RS, "PAROPT_AMD_WGRAM_RS", 1) != 0;
  // Narrow panels (<= 20 columns) take the single-role form: one workgroup per CU with three 128-row tiles of a few
  // columns in flight does not cover the HBM latency (5 columns: 0.31 of the peak at any n, 9: 0.45, 17: 0.65), four
  // single-role workgroups per CU do (0.70 / 0.79 / 0.66-0.77); from 25 columns on the two forms are level and the
  // producer/consumer form wins beyond (round 4, tools/dbg/wgram_pc_grid.py; PAROPT_AMD_WGRAM_PC_MIN_NG to move it).
  // (The same threshold for a panel image riding in the pass: on a narrow panel the single-role Gram plus the
  // stand-alone panel image are faster than the fused producer/consumer form -- 0.18 + 0.14 against 0.9 ms at 5 columns.)
  // PAROPT_AMD_WGRAM_PC64=0: panels of 65-80 columns on the single-role kernel, as before round 5 (A/B)
  static const bool use_pc64 = !(getenv("PAROPT_AMD_WGRAM_PC64") && atoi(getenv("PAROPT_AMD_WGRAM_PC64")) == 0);
  static const int pc_min_ng = getenv("PAROPT_AMD_WGRAM_PC_MIN_NG") ? atoi(getenv("PAROPT_AMD_WGRAM_PC_MIN_NG")) : 6;
  if (use_pc && NG >= pc_min_ng && wgram_groups_geom(groups, nv, n, kpend, &gg)) {
    // the pass also takes the structured panel image (see GramGeom): group tiles first, ordinary tiles behind them
    ntiles = gg.ngt + (n - gg.rg + kGramTile - 1) / kGramTile;
    for (int j = 0; j < gg.ncols; j++) ut.p[j] = groups->U[j];
    count_bytes(c, (double)gg.ncols, gg.nwcon);
    // round 5 experiment, OFF by default: the sums on the PRODUCER waves (GS = 2: same bits; all 621 GPU tests pass with
    // it).  Measured in one process at config 4 (tools/ab_switch.py, profiles/r05_ab_gs_producer.jsonl): 5.93 against
    // 5.52 ms per iteration, the Gram phase 1.76 against 1.46 ms -- the producers' staging + load issue + 40 operand
    // requests is the longer path of a tile, not the consumers' matrix work + sums.  PAROPT_AMD_GS_PRODUCER=1 selects it.
    const bool gs_prod = gg.G * gg.ncols <= 256 && dbg_switch(SW_GS_PRODUCER, "PAROPT_AMD_GS_PRODUCER", 0) != 0;
#define PO_WGG(NGv)                                                                                                  \
  case NGv:                                                                                                          \
    if (NGv >= kGramRowSplitMinNG && row_split && gs_prod)                                                           \
      PO_TRY((wgram_pc_launch_t<(NGv >= kGramRowSplitMinNG ? NGv : kGramRowSplitMinNG), 0, 1, 2>(                    \
          c, d, pt, nv, n, ntiles, st, zt, 0, 0.0, tcol, &grid, &gg, &ut)));                                         \
    else if (NGv >= kGramRowSplitMinNG && row_split)                                                                 \
      PO_TRY((wgram_pc_launch_t<NGv, 0, 1, 1>(c, d, pt, nv, n, ntiles, st, zt, 0, 0.0, tcol, &grid, &gg, &ut)));     \
    else                                                                                                             \
      PO_TRY((wgram_pc_launch_t<NGv, 0, 0, 1>(c, d, pt, nv, n, ntiles, st, zt, 0, 0.0, tcol, &grid, &gg, &ut)));     \
    break;
    switch (NG) {
      PO_WGG(1) PO_WGG(2) PO_WGG(3) PO_WGG(4) PO_WGG(5) PO_WGG(6) PO_WGG(7) PO_WGG(8) PO_WGG(9)
    }
#undef PO_WGG
    if (groups_done) *groups_done = true;
    *nblocks = grid;
    *nslots = (NG * (NG + 1) / 2) * 16;
    return PO_OK;
  }
#define PO_WG(NGv)                                                                                     \
  case NGv: {                                                                                          \
    /* measured at NG = 11 (n = 50 M): the plain form is fastest compiled for 3 wavefronts per SIMD, the form   \
       that also forms the L-SR1 columns for 2 (its extra prefetch registers spill at 3) */                     \
    /* (round 5: 11 groups at 3 wavefronts per SIMD and the column-forming form of 10 / 11 groups at 3 spilled   \
       to scratch: 2 there; every instantiation left is scratch-free, tests/test_kernel_resources.py) */        \
    constexpr int OCC0 = NGv <= 8 ? 4 : (NGv <= 10 ? 3 : (NGv <= 14 ? 2 : 1));                         \
    constexpr int OCC0A = OCC0 > 1 ? OCC0 - 1 : 1;                                                     \
    constexpr int OCCZ = NGv <= 7 ? 3 : (NGv <= 13 ? 2 : 1);                                           \
    constexpr int OCCZA = NGv <= 7 ? 4 : (NGv <= 9 ? 3 : OCCZ);                                        \
    if (NGv >= 17 && use_pc && use_pc64 && kpend == 0 && n >= 4 * kGramTile) {                         \
      constexpr int NGw = NGv >= 17 ? NGv : 17;                                                        \
      PO_TRY((wgram_pc64_launch_t<NGw>(c, d, pt, nv, n, tcol, &grid)));                                \
    } else if (NGv <= 16 && use_pc && n >= 4 * kGramTile && NGv >= pc_min_ng) {                        \
      constexpr int NGc = NGv <= 16 ? NGv : 16;                                                        \
      if (NGv >= kGramRowSplitMinNG && NGv <= kGramRowSplitMaxNG && row_split) {                       \
        constexpr int NGr = NGv <= kGramRowSplitMaxNG ? NGv : kGramRowSplitMaxNG;                      \
        if (kpend > 0) PO_TRY((wgram_pc_launch_t<NGr, 3, 1>(c, d, pt, nv, n, ntiles, st, zt, kpend, b0, tcol, &grid))); \
        else PO_TRY((wgram_pc_launch_t<NGr, 0, 1>(c, d, pt, nv, n, ntiles, st, zt, 0, 0.0, tcol, &grid)));              \
      } else if (kpend > 0 && NGv == 16) /* (the producer/consumer form with column formation spills at 16 groups) */ \
        PO_TRY((wgram_launch_t<NGv, 3, 1>(c, d, pt, nv, n, ntiles, st, zt, kpend, b0, tcol, &grid)));                   \
      else if (kpend > 0) PO_TRY((wgram_pc_launch_t<(NGc < 16 ? NGc : 15), 3, 0>(c, d, pt, nv, n, ntiles, st, zt, kpend, b0, tcol, &grid))); \
      else PO_TRY((wgram_pc_launch_t<NGc, 0, 0>(c, d, pt, nv, n, ntiles, st, zt, 0, 0.0, tcol, &grid)));                \
    } else if (kpend > 0) {                                                                            \
      if (occ_env == OCCZA) PO_TRY((wgram_launch_t<NGv, 3, OCCZA>(c, d, pt, nv, n, ntiles, st, zt, kpend, b0, tcol, &grid))); \
      else PO_TRY((wgram_launch_t<NGv, 3, OCCZ>(c, d, pt, nv, n, ntiles, st, zt, kpend, b0, tcol, &grid)));     \
    } else {                                                                                           \
      if (occ_env == OCC0A) PO_TRY((wgram_launch_t<NGv, 0, OCC0A>(c, d, pt, nv, n, ntiles, st, zt, 0, 0.0, tcol, &grid)));    \
      else PO_TRY((wgram_launch_t<NGv, 0, OCC0>(c, d, pt, nv, n, ntiles, st, zt, 0, 0.0, tcol, &grid)));        \
    }                                                                                                  \
  } break;
  switch (NG) {
    PO_WG(1) PO_WG(2) PO_WG(3) PO_WG(4) PO_WG(5) PO_WG(6) PO_WG(7) PO_WG(8) PO_WG(9) PO_WG(10)
    PO_WG(11) PO_WG(12) PO_WG(13) PO_WG(14) PO_WG(15) PO_WG(16) PO_WG(17) PO_WG(18) PO_WG(19) PO_WG(20)
  }
#undef PO_WG
  *nblocks = grid;
  *nslots = (NG * (NG + 1) / 2) * 16;
  return PO_OK;
}

int wgram_debug_stamps(double out[8]) {
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  PO_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wgram_stamp), sizeof(h)));
  for (int i = 0; i < 8; i++) out[i] = (double)h[i];
  return PO_OK;
}

static int wgram_one_launch(Ctx *c, const double *d, const double *const *V, int nv, int64_t n, double *W,
                            const double *const *S, double *const *Zout, int kpend, double b0, int preweighted_last,
                            bool may_defer, const GramGroups *groups = nullptr, bool *groups_done = nullptr);

// Panels wider than one launch (kWgramMaxVecs columns: registers and LDS of the kernel) are processed by column
// BLOCKS: the columns are cut into nb blocks of <= kWgramMaxVecs / 2 columns (multiples of 4) and every pair of
// blocks (I < J) is one launch over [V_I | V_J], which yields W_II, W_IJ and W_JJ.  nb (nb - 1) / 2 launches, each
// block streamed nb - 1 times: slower per column than the single launch, but there is no limit on the width (the
// reference has none either: src/ParOptInteriorPoint.cpp:1935-1950, 2648-2654).
int k_wgram(Ctx *c, const double *d, const double *const *V, int nv, int64_t n, double *W,
            const double *const *S, double *const *Zout, int kpend, double b0, int preweighted_last, bool may_defer,
            const GramGroups *groups, bool *groups_done) {
  if (groups_done) *groups_done = false;
  if (nv <= 0) return PO_OK;
  if (nv <= kWgramMaxVecs)
    return wgram_one_launch(c, d, V, nv, n, W, S, Zout, kpend, b0, preweighted_last, may_defer, groups, groups_done);
  if (kpend > 0 || preweighted_last) {
    set_error("wgram: a panel of %d columns is processed in blocks, which cannot form L-SR1 columns or carry a "
              "pre-weighted column", nv);
    return PO_ERR_ARG;
  }
  const int half = kWgramMaxVecs / 2;
  const int nb = (nv + half - 1) / half;
  int bw = (nv + nb - 1) / nb;
  bw = ((bw + 3) / 4) * 4;  // whole column groups
  std::vector<const double *> V2;
  std::vector<double> W2;
  for (int I = 0; I < nb; I++) {
    const int i0 = I * bw, ni = (nv - i0 < bw) ? nv - i0 : bw;
    if (ni <= 0) continue;
    for (int J = I + 1; J < nb; J++) {
      const int j0 = J * bw, nj = (nv - j0 < bw) ? nv - j0 : bw;
      if (nj <= 0) continue;
      const int m2 = ni + nj;
      V2.assign(V + i0, V + i0 + ni);
      V2.insert(V2.end(), V + j0, V + j0 + nj);
      W2.assign((size_t)m2 * m2, 0.0);
      PO_TRY(wgram_one_launch(c, d, V2.data(), m2, n, W2.data(), nullptr, nullptr, 0, 0.0, 0, false));
      auto col = [&](int q) { return q < ni ? i0 + q : j0 + (q - ni); };
      for (int q = 0; q < m2; q++)
        for (int r = 0; r < m2; r++) W[col(r) + (size_t)nv * col(q)] = W2[r + (size_t)m2 * q];
    }
  }
  return PO_OK;
}

static int wgram_one_launch(Ctx *c, const double *d, const double *const *V, int nv, int64_t n, double *W,
                            const double *const *S, double *const *Zout, int kpend, double b0, int preweighted_last,
                            bool may_defer, const GramGroups *groups, bool *groups_done) {
  if (nv <= 0) return PO_OK;
  int grid = 0, nslots = 0;
  // (the block sums live on the heap: a deferred launch unpacks them at the flush of the enclosing batch)
  const bool defer = may_defer && c->batch_depth > 0;
  // po_ctx_time_wgram: HIP events on the launch stream (not for deferred launches: the one event pair is read back
  // right after the launch)
  const bool timed = c->time_wgram != 0 && !defer;
  if (timed) PO_HIP(hipEventRecord(c->ev0, c->stream));
  PO_TRY(k_wgram_launch(c, d, V, nv, n, &grid, &nslots, S, Zout, kpend, b0, preweighted_last, groups, groups_done));
  if (timed) PO_HIP(hipEventRecord(c->ev1, c->stream));
  auto blocks = std::make_shared<std::vector<double>>(nslots);
  PO_TRY(reduce_finish(c, grid, nslots, 0, 0, blocks->data(), !defer));  // !defer: the results are on the host
  if (timed) {
    // (the host may have seen the completion flag without synchronising the stream: the event is waited for itself)
    float ms = 0.0f;
    PO_HIP(hipEventSynchronize(c->ev1));
    PO_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    const int w = kpend > 0 ? 1 : 0;
    c->wgram_ms[w] += ms;
    c->wgram_count[w]++;
    c->wgram_cols[w] = nv;
    c->wgram_bytes[w] += 8.0 * (double)n * (nv + 1 + 2 * kpend);  // panel + weights (+ S read, Z written)
  }
  auto unpack = [blocks, nv, W] {
    const int NG = wgram_groups(nv);
    int p = 0;
    for (int I = 0; I < NG; I++) {
      for (int J = I; J < NG; J++, p++) {
        for (int i = 0; i < 4; i++) {
          for (int j = 0; j < 4; j++) {
            const int r = 4 * I + i, s = 4 * J + j;
            // upper entries only: a diagonal block pair computes both triangles, and with a pre-weighted last
            // column only W[r][last] = P_r . t is meaningful
            if (r < nv && s < nv && r <= s) {
              const double v = (*blocks)[(size_t)p * 16 + i * 4 + j];
              W[r + (size_t)nv * s] = v;
              W[s + (size_t)nv * r] = v;
            }
          }
        }
      }
    }
  };
  after_reduce(c, unpack);  // (runs at once unless the reduction was queued)
  return PO_OK;
}

}  // namespace po
