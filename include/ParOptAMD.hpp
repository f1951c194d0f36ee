// ParOptAMD.hpp -- header-only C++ facade over the C ABI (paropt_amd.h) with ParOpt's class and
// method names, so that problem classes written against the reference's C++ API
// (src/ParOptVec.h:28-98, src/ParOptProblem.h:42-296, src/ParOptQuasiNewton.h:32-220,
// src/ParOptOptions.h:9-61, src/ParOptInteriorPoint.h:128-217) port by changing the include and the
// communicator argument: the MPI_Comm of the reference becomes a po_ctx (one per GPU/rank).
//
// Semantics kept from the reference:
//   * intrusive reference counting: objects are born with count 0, holders incref(), decref()
//     deletes at 0 (ParOptBase, src/ParOptVec.h:28-47);
//   * ParOptVec::getArray returns the local length and a raw HOST pointer the caller may read and
//     write (src/ParOptVec.cpp:212-217).  Here that pointer is a pinned mirror of the HBM data: the
//     facade downloads a vector before handing it to a user callback as an input and uploads the
//     vectors a callback is documented to fill (x, lb, ub / g, Ac) when it returns;
//   * user callbacks return int fail (0 = ok); optimize() returns 0, 1 (mis-configuration) or the
//     initial evaluation's fail code; options are set by name with typed setOption overloads that
//     return non-zero for unknown names / wrong types (src/ParOptOptions.cpp:310-386).
#ifndef PAROPT_AMD_HPP
#define PAROPT_AMD_HPP

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

extern "C" {
#include "paropt_amd.h"
}

typedef double ParOptScalar;
#define ParOptRealPart(x) (x)

// ---- the communicator -----------------------------------------------------------------------------
// PAROPT_AMD_USE_MPI (set by the headers under include/paropt_compat/, which carry the reference's file names):
// the classes take the reference's MPI_Comm, one po_ctx per communicator is created on first use (device = rank
// within the node, modulo the visible devices) and reductions across ranks go through MPI_Allgather on the host
// (po_ctx_comm_init_callback) -- a maintainer's MPI program recompiles unchanged.  Without it the "communicator" is
// the po_ctx itself.
#ifdef PAROPT_AMD_USE_MPI
#include <mpi.h>
typedef MPI_Comm ParOptComm;
#define PAROPT_MPI_TYPE MPI_DOUBLE
struct ParOptAMDCommEntry {
  MPI_Comm comm;
  po_ctx ctx;
};
inline int paropt_amd_allgather(const double *in, double *out, int count, void *user) {
  MPI_Comm comm = static_cast<ParOptAMDCommEntry *>(user)->comm;
  return MPI_Allgather(in, count, MPI_DOUBLE, out, count, MPI_DOUBLE, comm) == MPI_SUCCESS ? 0 : 1;
}
inline po_ctx paropt_amd_context(MPI_Comm comm) {
  static std::vector<ParOptAMDCommEntry *> reg;  // lives as long as the process
  for (ParOptAMDCommEntry *e : reg) {
    int cmp = MPI_UNEQUAL;
    MPI_Comm_compare(e->comm, comm, &cmp);
    if (cmp == MPI_IDENT || cmp == MPI_CONGRUENT) return e->ctx;
  }
  int rank = 0, size = 1, local = 0, ndev = 0;
  MPI_Comm_rank(comm, &rank);
  MPI_Comm_size(comm, &size);
  MPI_Comm node;
  if (MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node) == MPI_SUCCESS) {
    MPI_Comm_rank(node, &local);
    MPI_Comm_free(&node);
  }
  po_device_count(&ndev);
  ParOptAMDCommEntry *e = new ParOptAMDCommEntry();
  e->comm = comm;
  e->ctx = NULL;
  if (po_ctx_create(ndev > 0 ? local % ndev : 0, &e->ctx) != 0 ||
      (size > 1 && po_ctx_comm_init_callback(e->ctx, rank, size, &paropt_amd_allgather, e) != 0)) {
    fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  reg.push_back(e);
  return e->ctx;
}
#else
typedef po_ctx ParOptComm;
inline po_ctx paropt_amd_context(po_ctx ctx) { return ctx; }
#endif

class ParOptBase {
 public:
  ParOptBase() : ref_count(0) {}
  virtual ~ParOptBase() {}
  void incref() { ref_count++; }
  void decref() {
    ref_count--;
    if (ref_count == 0) delete this;
  }

 private:
  int ref_count;
};

// ---- ParOptVec: the 11 pure virtuals of src/ParOptVec.h:53-70 ---------------------------------------
class ParOptVec : public ParOptBase {
 public:
  virtual ~ParOptVec() {}
  virtual void set(ParOptScalar alpha) = 0;
  virtual void zeroEntries() = 0;
  virtual void copyValues(ParOptVec *vec) = 0;
  virtual double norm() = 0;
  virtual double maxabs() = 0;
  virtual double l1norm() = 0;
  virtual ParOptScalar dot(ParOptVec *vec) = 0;
  virtual void mdot(ParOptVec **vecs, int nvecs, ParOptScalar *output) = 0;
  virtual void scale(ParOptScalar alpha) = 0;
  virtual void axpy(ParOptScalar alpha, ParOptVec *x) = 0;
  virtual int getArray(ParOptScalar **array) = 0;
  // facade additions (no-ops for vectors that do not live in HBM): the device handle, and explicit control of
  // the pinned host mirror behind getArray (not needed for correctness: the mirror is kept coherent, see
  // po_vec_get_array)
  virtual po_vec handle() { return NULL; }
  virtual void syncToDevice() {}
  virtual void syncToHost() {}
  virtual void releaseArray(int) {}
};

// ParOptBasicVec (src/ParOptVec.h:75-96): the design vector in HBM
class ParOptBasicVec : public ParOptVec {
 public:
  // a new zero-filled vector of n local components
  ParOptBasicVec(ParOptComm comm, int n) : owner(true), h(NULL) { po_vec_create(paropt_amd_context(comm), n, &h); }
  // adapter over a vector owned by the library (callback arguments, solver state)
  explicit ParOptBasicVec(po_vec borrowed) : owner(false), h(borrowed) {}
  ~ParOptBasicVec() {
    if (owner && h) po_vec_decref(h);
  }
  void set(ParOptScalar alpha) { po_vec_set(h, alpha); }
  void zeroEntries() { po_vec_zero(h); }
  void copyValues(ParOptVec *vec) { po_vec_copy(h, vec->handle()); }
  double norm() { double v = 0; po_vec_norm(h, &v); return v; }
  double maxabs() { double v = 0; po_vec_maxabs(h, &v); return v; }
  double l1norm() { double v = 0; po_vec_l1norm(h, &v); return v; }
  ParOptScalar dot(ParOptVec *vec) { double v = 0; po_vec_dot(h, vec->handle(), &v); return v; }
  void mdot(ParOptVec **vecs, int nvecs, ParOptScalar *output) {
    std::vector<po_vec> hs(nvecs > 0 ? nvecs : 1);
    for (int i = 0; i < nvecs; i++) hs[i] = vecs[i]->handle();
    po_vec_mdot(h, hs.data(), nvecs, output);
  }
  void scale(ParOptScalar alpha) { po_vec_scale(h, alpha); }
  void axpy(ParOptScalar alpha, ParOptVec *x) { po_vec_axpy(h, alpha, x->handle()); }
  // the reference's contract (src/ParOptVec.cpp:212-217): the pointer is the vector's data -- writes are seen by
  // the operations above without any call in between, results of those operations show up behind the pointer
  int getArray(ParOptScalar **array) {
    int64_t n = 0;
    po_vec_size(h, &n);
    if (array) po_vec_get_array(h, array);
    return (int)n;
  }
  po_vec handle() { return h; }
  void syncToDevice() { po_vec_sync_to_device(h); }
  void syncToHost() { po_vec_sync_to_host(h); }
  void releaseArray(int upload) { po_vec_release_array(h, upload); }

 private:
  bool owner;
  po_vec h;
};

// ---- ParOptOptions (src/ParOptOptions.h:9-61, .cpp:80-470): typed registry with immediate validation ---------
class ParOptOptions : public ParOptBase {
 public:
  static const int PAROPT_STRING_OPTION = 1;
  static const int PAROPT_BOOLEAN_OPTION = 2;
  static const int PAROPT_INT_OPTION = 3;
  static const int PAROPT_FLOAT_OPTION = 4;
  static const int PAROPT_ENUM_OPTION = 5;

#ifdef PAROPT_AMD_USE_MPI
  ParOptOptions(MPI_Comm _comm = MPI_COMM_WORLD) : comm(_comm), iter_pos(0) {}
#else
  ParOptOptions() : iter_pos(0) {}
#endif
  // the add* calls return 1 when the name is taken (the entry is then left alone), else 0
  int addStringOption(const char *name, const char *value, const char *descript) {
    Entry x;
    x.type = PAROPT_STRING_OPTION;
    x.has_str = value != NULL;
    x.s = x.s_default = value ? value : "";
    return add(name, x, descript);
  }
  int addBoolOption(const char *name, int value, const char *descript) {
    Entry x;
    x.type = PAROPT_BOOLEAN_OPTION;
    x.i = x.i_default = value ? 1 : 0;
    x.ilo = 0;
    x.ihi = 1;
    return add(name, x, descript);
  }
  int addIntOption(const char *name, int value, int low, int high, const char *descript) {
    Entry x;
    x.type = PAROPT_INT_OPTION;
    x.i = x.i_default = value;
    x.ilo = low;
    x.ihi = high;
    return add(name, x, descript);
  }
  int addFloatOption(const char *name, double value, double low, double high, const char *descript) {
    Entry x;
    x.type = PAROPT_FLOAT_OPTION;
    x.f = x.f_default = value;
    x.flo = low;
    x.fhi = high;
    return add(name, x, descript);
  }
  int addEnumOption(const char *name, const char *value, int size, const char *options[], const char *descript) {
    Entry x;
    x.type = PAROPT_ENUM_OPTION;
    x.s = x.s_default = value ? value : "";
    x.has_str = true;
    for (int k = 0; k < size; k++) x.choices.push_back(options[k]);
    return add(name, x, descript);
  }
  int isOption(const char *name) { return e.count(name) ? 1 : 0; }

  // setOption returns 0 on success; unknown names, wrong types, values outside the range and strings that are
  // not a member of the enumeration are refused with a message on stderr (:310-386) and leave the entry unchanged
  int setOption(const char *name, const char *value) {
    Entry *x = find(name);
    if (!x) return 1;
    if (x->type == PAROPT_STRING_OPTION) {
      x->has_str = value != NULL;
      x->s = value ? value : "";
      x->is_set = 1;
      return 0;
    }
    if (x->type == PAROPT_ENUM_OPTION) {
      for (const std::string &c : x->choices) {
        if (value && c == value) {
          x->s = value;
          x->is_set = 1;
          return 0;
        }
      }
      fprintf(stderr, "ParOptOptions Warning: Enum option %s not set: %s is not one of its values\n", name,
              value ? value : "(null)");
      return 1;
    }
    fprintf(stderr, "ParOptOptions Warning: Option %s is not a string or enum option\n", name);
    return 1;
  }
  int setOption(const char *name, int value) {
    Entry *x = find(name);
    if (!x) return 1;
    if (x->type == PAROPT_BOOLEAN_OPTION) {
      x->i = value ? 1 : 0;
      x->is_set = 1;
      return 0;
    }
    if (x->type == PAROPT_INT_OPTION) {
      if (value < x->ilo || value > x->ihi) {
        fprintf(stderr, "ParOptOptions Warning: Integer option %s = %d out of range [%d, %d]\n", name, value, x->ilo,
                x->ihi);
        return 1;
      }
      x->i = value;
      x->is_set = 1;
      return 0;
    }
    fprintf(stderr, "ParOptOptions Warning: Option %s is not a boolean or integer option\n", name);
    return 1;
  }
  int setOption(const char *name, double value) {
    Entry *x = find(name);
    if (!x) return 1;
    if (x->type != PAROPT_FLOAT_OPTION) {
      fprintf(stderr, "ParOptOptions Warning: Option %s is not a float option\n", name);
      return 1;
    }
    if (value < x->flo || value > x->fhi) {
      fprintf(stderr, "ParOptOptions Warning: Float option %s = %g out of range [%g, %g]\n", name, value, x->flo,
              x->fhi);
      return 1;
    }
    x->f = value;
    x->is_set = 1;
    return 0;
  }
  // typed getters: NULL / 0 / 0.0 for a missing name or the wrong type, like the reference (:391-440)
  const char *getStringOption(const char *name) {
    Entry *x = get(name, PAROPT_STRING_OPTION);
    return (x && x->has_str) ? x->s.c_str() : NULL;
  }
  int getBoolOption(const char *name) {
    Entry *x = get(name, PAROPT_BOOLEAN_OPTION);
    return x ? x->i : 0;
  }
  int getIntOption(const char *name) {
    Entry *x = get(name, PAROPT_INT_OPTION);
    return x ? x->i : 0;
  }
  double getFloatOption(const char *name) {
    Entry *x = get(name, PAROPT_FLOAT_OPTION);
    return x ? x->f : 0.0;
  }
  const char *getEnumOption(const char *name) {
    Entry *x = get(name, PAROPT_ENUM_OPTION);
    return x ? x->s.c_str() : NULL;
  }
  int getOptionType(const char *name) {
    std::map<std::string, Entry>::iterator it = e.find(name);
    return it == e.end() ? 0 : it->second.type;
  }
  const char *getDescription(const char *name) {
    std::map<std::string, Entry>::iterator it = e.find(name);
    return it == e.end() ? NULL : it->second.descript.c_str();
  }
  int getIntRange(const char *name, int *low, int *high) {
    Entry *x = get(name, PAROPT_INT_OPTION);
    if (!x) return 1;
    if (low) *low = x->ilo;
    if (high) *high = x->ihi;
    return 0;
  }
  int getFloatRange(const char *name, double *low, double *high) {
    Entry *x = get(name, PAROPT_FLOAT_OPTION);
    if (!x) return 1;
    if (low) *low = x->flo;
    if (high) *high = x->fhi;
    return 0;
  }
  int getEnumRange(const char *name, int *size, const char *const **values) {
    Entry *x = get(name, PAROPT_ENUM_OPTION);
    if (!x) return 1;
    x->cptr.clear();
    for (const std::string &c : x->choices) x->cptr.push_back(c.c_str());
    if (size) *size = (int)x->cptr.size();
    if (values) *values = x->cptr.data();
    return 0;
  }
  // one line per option; output_level > 0 also prints the ranges (:443-470)
  void printSummary(FILE *fp, int output_level) {
    if (!fp) return;
    for (std::map<std::string, Entry>::iterator it = e.begin(); it != e.end(); ++it) {
      const Entry &x = it->second;
      if (x.type == PAROPT_STRING_OPTION) {
        fprintf(fp, "%-40s %-15s\n", it->first.c_str(), x.has_str ? x.s.c_str() : "(null)");
      } else if (x.type == PAROPT_ENUM_OPTION) {
        fprintf(fp, "%-40s %-15s\n", it->first.c_str(), x.s.c_str());
        if (output_level > 0) {
          fprintf(fp, "%-40s", "  values:");
          for (const std::string &c : x.choices) fprintf(fp, " %s", c.c_str());
          fprintf(fp, "\n");
        }
      } else if (x.type == PAROPT_FLOAT_OPTION) {
        fprintf(fp, "%-40s %-15g\n", it->first.c_str(), x.f);
        if (output_level > 0) fprintf(fp, "%-40s [%g, %g]\n", "  range:", x.flo, x.fhi);
      } else {
        fprintf(fp, "%-40s %-15d\n", it->first.c_str(), x.i);
        if (output_level > 0 && x.type == PAROPT_INT_OPTION) fprintf(fp, "%-40s [%d, %d]\n", "  range:", x.ilo, x.ihi);
      }
    }
  }
  // iteration over the names (:472-495)
  void begin() { iter_pos = 0; names.clear(); for (auto &kv : e) names.push_back(kv.first); }
  const char *getName() { return iter_pos < names.size() ? names[iter_pos].c_str() : NULL; }
  int next() { iter_pos++; return iter_pos < names.size() ? 1 : 0; }

  // facade plumbing: the registries of the library's drivers (po_options_visit_defaults), and the hand-over of
  // every entry a driver knows to that driver
  void addLibraryDefaults(int which) { po_options_visit_defaults(which, &ParOptOptions::visit, this); }
  template <class SetStr, class SetInt, class SetFloat>
  int forward(int which, SetStr sstr, SetInt sint, SetFloat sfloat) {
    std::vector<std::string> known;
    po_options_visit_defaults(which, &ParOptOptions::collect, &known);
    if (which != 0) po_options_visit_defaults(0, &ParOptOptions::collect, &known);  // tr / mma share the ip registry
    int bad = 0;
    for (const std::string &nm : known) {
      std::map<std::string, Entry>::iterator it = e.find(nm);
      if (it == e.end()) continue;
      const Entry &x = it->second;
      if (x.type == PAROPT_STRING_OPTION || x.type == PAROPT_ENUM_OPTION) {
        bad |= sstr(nm.c_str(), (x.type == PAROPT_STRING_OPTION && !x.has_str) ? "" : x.s.c_str());
      } else if (x.type == PAROPT_FLOAT_OPTION) {
        bad |= sfloat(nm.c_str(), x.f);
      } else {
        bad |= sint(nm.c_str(), x.i);
      }
    }
    return bad;
  }

 private:
  struct Entry {
    int type = 0, is_set = 0;
    bool has_str = false;
    std::string s, s_default, descript;
    int i = 0, i_default = 0, ilo = 0, ihi = 0;
    double f = 0.0, f_default = 0.0, flo = 0.0, fhi = 0.0;
    std::vector<std::string> choices;
    std::vector<const char *> cptr;
  };
  int add(const char *name, Entry &x, const char *descript) {
    if (e.count(name)) return 1;
    x.descript = descript ? descript : "";
    e[name] = x;
    return 0;
  }
  Entry *find(const char *name) {
    std::map<std::string, Entry>::iterator it = e.find(name);
    if (it == e.end()) {
      fprintf(stderr, "ParOptOptions Warning: %s is not an option\n", name);
      return NULL;
    }
    return &it->second;
  }
  Entry *get(const char *name, int type) {
    std::map<std::string, Entry>::iterator it = e.find(name);
    return (it == e.end() || it->second.type != type) ? NULL : &it->second;
  }
  static void visit(void *user, const char *name, int type, const char *sval, int ival, int ilo, int ihi, double fval,
                    double flo, double fhi, int nenum, const char *const *enumvals) {
    ParOptOptions *o = static_cast<ParOptOptions *>(user);
    if (type == PAROPT_STRING_OPTION) {
      o->addStringOption(name, sval, "");
    } else if (type == PAROPT_BOOLEAN_OPTION) {
      o->addBoolOption(name, ival, "");
    } else if (type == PAROPT_INT_OPTION) {
      o->addIntOption(name, ival, ilo, ihi, "");
    } else if (type == PAROPT_FLOAT_OPTION) {
      o->addFloatOption(name, fval, flo, fhi, "");
    } else {
      std::vector<const char *> ch(enumvals, enumvals + nenum);
      o->addEnumOption(name, sval, nenum, ch.data(), "");
    }
  }
  static void collect(void *user, const char *name, int, const char *, int, int, int, double, double, double, int,
                      const char *const *) {
    static_cast<std::vector<std::string> *>(user)->push_back(name);
  }
#ifdef PAROPT_AMD_USE_MPI
  MPI_Comm comm;
#endif
  std::map<std::string, Entry> e;
  std::vector<std::string> names;
  size_t iter_pos;
};

// ---- ParOptQuasiDefMat: what createQuasiDefMat() returns (src/ParOptSparseMat.h:18-120) ------------------
// In the reference these objects factor and apply the quasi-definite matrix on the host.  Here the library does
// that on the device; the objects only say WHICH form the problem asks for: block diagonal with nwblock x nwblock
// blocks (ParOptQuasiDefBlockMat) or a general sparse matrix from the CSR pattern (ParOptQuasiDefSparseMat).
class ParOptProblem;
class ParOptSparseProblem;
class ParOptQuasiDefMat : public ParOptBase {
 public:
  virtual ~ParOptQuasiDefMat() {}
  virtual int getBlockSize() { return 1; }
  virtual int isSparse() { return 0; }
};
class ParOptQuasiDefBlockMat : public ParOptQuasiDefMat {
 public:
  ParOptQuasiDefBlockMat(ParOptProblem *, int _nwblock) : nwblock(_nwblock < 1 ? 1 : _nwblock) {}
  int getBlockSize() { return nwblock; }

 private:
  int nwblock;
};
class ParOptQuasiDefSparseMat : public ParOptQuasiDefMat {
 public:
  explicit ParOptQuasiDefSparseMat(ParOptSparseProblem *) {}
  int isSparse() { return 1; }
};

// ---- ParOptProblem (src/ParOptProblem.h:42-296) -------------------------------------------------------
class ParOptProblem : public ParOptBase {
 public:
  explicit ParOptProblem(ParOptComm _comm)
      : comm(_comm), ctx(paropt_amd_context(_comm)), nvars(0), ncon(0), ninequality(-1), nwcon(0), nwinequality(-1),
        nwblock(1), linear_constraints(0), hprob(NULL) {}
  ParOptProblem(ParOptComm _comm, int _nvars, int _ncon, int _ninequality, int _nwcon, int _nwinequality)
      : comm(_comm), ctx(paropt_amd_context(_comm)), nvars(_nvars), ncon(_ncon), ninequality(_ninequality),
        nwcon(_nwcon), nwinequality(_nwinequality), nwblock(1), linear_constraints(0), hprob(NULL) {}
  // the nwblock of `new ParOptQuasiDefBlockMat(this, nwblock)` when createQuasiDefMat() is not overridden
  void setSparseBlockSize(int _nwblock) { nwblock = _nwblock; }
  // facade extension (po_problem_set_linear_constraints): the dense constraints are linear, evalObjConGradient is
  // called with Ac == NULL after the first evaluation of each optimize()
  void setLinearConstraints(int flag) {
    linear_constraints = flag;
    if (hprob) po_problem_set_linear_constraints(hprob, flag);
  }
  // facade extension (po_problem_set_deferred_reductions): the callbacks take reduced values only through ParOptVec
  // reductions / po_ctx_reduce_device and post-process them in po_ctx_after_reduce hooks; the solver may then batch
  // them with its own reductions (one collective + host sync per step instead of one per reduction)
  void setDeferredReductions(int flag) {
    deferred_reductions = flag;
    if (hprob) po_problem_set_deferred_reductions(hprob, flag);
  }
  virtual ~ParOptProblem() {
    if (hprob) po_problem_destroy(hprob);
  }
  ParOptComm getMPIComm() { return comm; }
  po_ctx getContext() { return ctx; }
  void setProblemSizes(int _nvars, int _ncon, int _nwcon) {
    nvars = _nvars;
    ncon = _ncon;
    nwcon = _nwcon;
    if (ninequality < 0) ninequality = ncon;
    if (nwinequality < 0) nwinequality = nwcon;
  }
  void setNumInequalities(int _ninequality, int _nwinequality) {
    ninequality = _ninequality;
    nwinequality = _nwinequality;
  }
  void getProblemSizes(int *_nvars, int *_ncon, int *_nwcon) {
    if (_nvars) *_nvars = nvars;
    if (_ncon) *_ncon = ncon;
    if (_nwcon) *_nwcon = nwcon;
  }
  void getNumInequalities(int *_ninequality, int *_nwinequality) {
    if (_ninequality) *_ninequality = ninequality < 0 ? ncon : ninequality;
    if (_nwinequality) *_nwinequality = nwinequality < 0 ? nwcon : nwinequality;
  }
  virtual ParOptVec *createDesignVec() { return new ParOptBasicVec(comm, nvars); }
  virtual ParOptVec *createConstraintVec() { return new ParOptBasicVec(comm, nwcon); }
  // the reference makes this pure virtual (:72); here the default is the block form with setSparseBlockSize()
  virtual ParOptQuasiDefMat *createQuasiDefMat() { return new ParOptQuasiDefBlockMat(this, nwblock); }
  virtual int isSparseInequality() { return 1; }
  virtual int useLowerBounds() { return 1; }
  virtual int useUpperBounds() { return 1; }

  virtual void getVarsAndBounds(ParOptVec *x, ParOptVec *lb, ParOptVec *ub) = 0;
  virtual int evalObjCon(ParOptVec *x, ParOptScalar *fobj, ParOptScalar *cons) = 0;
  virtual int evalObjConGradient(ParOptVec *x, ParOptVec *g, ParOptVec **Ac) = 0;
  // Hessian of the Lagrangian (:160-196); non-zero = not available
  virtual int evalHvecProduct(ParOptVec *, ParOptScalar *, ParOptVec *, ParOptVec *, ParOptVec *) { return 1; }
  virtual int evalHessianDiag(ParOptVec *, ParOptScalar *, ParOptVec *, ParOptVec *) { return 1; }
  virtual void computeQuasiNewtonUpdateCorrection(ParOptVec *, ParOptScalar *, ParOptVec *, ParOptVec *,
                                                  ParOptVec *) {}
  virtual void writeOutput(int, ParOptVec *) {}
  // sparse constraints (src/ParOptProblem.h:215-262); `out` / `pzw` are w-sized
  virtual void evalSparseCon(ParOptVec *, ParOptVec *) {}
  virtual void addSparseJacobian(ParOptScalar, ParOptVec *, ParOptVec *, ParOptVec *) {}
  virtual void addSparseJacobianTranspose(ParOptScalar, ParOptVec *, ParOptVec *, ParOptVec *) {}
  virtual void addSparseInnerProduct(ParOptScalar, ParOptVec *, ParOptVec *, ParOptScalar *) {}
  virtual int getSparseJacobianBlockSize() { return -1; }

  // Finite-difference check of the gradients the problem provides (src/ParOptProblem.cpp:225-622): the step
  // direction is +-1 by the sign of the objective gradient, forward differences with step dh; prints the
  // projected derivatives, their finite-difference estimates and the relative errors on rank 0, and (with
  // check_hvec_product) does the same for the Hessian-vector product against differences of the Lagrangian's
  // gradient, and for the sparse Jacobian products against differences of evalSparseCon.  Returns the largest
  // relative error seen (the reference returns nothing).
  double checkGradients(double dh, ParOptVec *xvec = NULL, int check_hvec_product = 0);

  // the C-callback problem handed to the library (created on first use); the library-backed trust-region
  // subproblems return the library's own object instead
  virtual po_problem handle() {
    if (!hprob) {
      ParOptQuasiDefMat *qd = createQuasiDefMat();
      if (qd) {
        qd->incref();
        if (!qd->isSparse()) nwblock = qd->getBlockSize();
        qd->decref();
      }
      po_problem_callbacks cb;
      cb.user = this;
      cb.get_vars_and_bounds = &ParOptProblem::tramp_vars;
      cb.eval_obj_con = &ParOptProblem::tramp_eval;
      cb.eval_obj_con_gradient = &ParOptProblem::tramp_grad;
      cb.qn_update_correction = NULL;
      cb.write_output = &ParOptProblem::tramp_write;
      if (po_problem_create_callbacks(ctx, nvars, ncon, ninequality < 0 ? ncon : ninequality, &cb, &hprob) != 0) {
        fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
      }
      if (hprob) {
        attachSparse();
        po_problem_set_hessian_callbacks(hprob, &ParOptProblem::tramp_hvec, &ParOptProblem::tramp_hdiag);
        if (!(useLowerBounds() && useUpperBounds()))
          po_problem_set_var_bound_options(hprob, useLowerBounds(), useUpperBounds());
        if (linear_constraints) po_problem_set_linear_constraints(hprob, 1);
        if (deferred_reductions) po_problem_set_deferred_reductions(hprob, 1);
      }
    }
    return hprob;
  }

 protected:
  ParOptComm comm;
  po_ctx ctx;
  int nvars, ncon, ninequality, nwcon, nwinequality;
  int nwblock, linear_constraints;
  int deferred_reductions = 0;
  po_problem hprob;
  // registers the sparse-constraint callbacks with the library; ParOptSparseProblem registers its CSR form
  virtual void attachSparse() {
    if (nwcon <= 0) return;
    po_problem_sparse_callbacks scb;
    scb.eval_sparse_con = &ParOptProblem::tramp_wcon;
    scb.add_sparse_jacobian = &ParOptProblem::tramp_wjac;
    scb.add_sparse_jacobian_transpose = &ParOptProblem::tramp_wjact;
    scb.add_sparse_inner_product = &ParOptProblem::tramp_winner;
    if (po_problem_set_sparse_callbacks(hprob, nwcon, nwinequality < 0 ? nwcon : nwinequality, &scb) != 0 ||
        (nwblock > 1 && po_problem_set_sparse_block_size(hprob, nwblock) != 0)) {
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    }
  }
  // Callback arguments are vectors the SOLVER owns: whatever the user's code did with their getArray pointers is
  // uploaded (outputs) or dropped (inputs) when the callback returns (po_vec_release_array).
  struct Arg {
    ParOptBasicVec v;
    int upload;
    Arg(po_vec h, int _upload) : v(h), upload(_upload) { v.incref(); }
    ~Arg() { v.releaseArray(upload); }
    ParOptVec *p() { return &v; }
  };

 private:
  static int tramp_vars(void *self, po_vec x, po_vec lb, po_vec ub) {
    Arg vx(x, 1), vl(lb, 1), vu(ub, 1);
    static_cast<ParOptProblem *>(self)->getVarsAndBounds(vx.p(), vl.p(), vu.p());
    return 0;
  }
  static int tramp_eval(void *self, po_vec x, double *fobj, double *cons) {
    Arg vx(x, 0);
    return static_cast<ParOptProblem *>(self)->evalObjCon(vx.p(), fobj, cons);
  }
  static int tramp_grad(void *self, po_vec x, po_vec g, const po_vec *Ac) {
    ParOptProblem *me = static_cast<ParOptProblem *>(self);
    Arg vx(x, 0), vg(g, 1);
    std::vector<Arg *> args;
    std::vector<ParOptVec *> va(me->ncon > 0 ? me->ncon : 1, (ParOptVec *)NULL);
    // Ac == NULL: the problem declared linear constraints (setLinearConstraints) and only g is wanted
    for (int j = 0; Ac && j < me->ncon; j++) {
      args.push_back(new Arg(Ac[j], 1));
      va[j] = args.back()->p();
    }
    int fail = me->evalObjConGradient(vx.p(), vg.p(), (Ac || me->ncon == 0) ? va.data() : NULL);
    for (Arg *a : args) delete a;
    return fail;
  }
  static int tramp_hvec(void *self, po_vec x, const double *z, po_vec zw, po_vec px, po_vec hvec) {
    ParOptProblem *me = static_cast<ParOptProblem *>(self);
    Arg vx(x, 0), vp(px, 0), vh(hvec, 1);
    Arg *vzw = zw ? new Arg(zw, 0) : NULL;
    std::vector<double> zc(z, z + (me->ncon > 0 ? me->ncon : 0));
    int fail = me->evalHvecProduct(vx.p(), zc.data(), vzw ? vzw->p() : NULL, vp.p(), vh.p());
    delete vzw;
    return fail;
  }
  static int tramp_hdiag(void *self, po_vec x, const double *z, po_vec zw, po_vec hdiag) {
    ParOptProblem *me = static_cast<ParOptProblem *>(self);
    Arg vx(x, 0), vh(hdiag, 1);
    Arg *vzw = zw ? new Arg(zw, 0) : NULL;
    std::vector<double> zc(z, z + (me->ncon > 0 ? me->ncon : 0));
    int fail = me->evalHessianDiag(vx.p(), zc.data(), vzw ? vzw->p() : NULL, vh.p());
    delete vzw;
    return fail;
  }
  static int tramp_wcon(void *self, po_vec x, po_vec out) {
    Arg vx(x, 0), vo(out, 1);
    static_cast<ParOptProblem *>(self)->evalSparseCon(vx.p(), vo.p());
    return 0;
  }
  static int tramp_wjac(void *self, double alpha, po_vec x, po_vec px, po_vec out) {
    Arg vx(x, 0), vp(px, 0), vo(out, 1);
    static_cast<ParOptProblem *>(self)->addSparseJacobian(alpha, vx.p(), vp.p(), vo.p());
    return 0;
  }
  static int tramp_wjact(void *self, double alpha, po_vec x, po_vec pzw, po_vec out) {
    Arg vx(x, 0), vp(pzw, 0), vo(out, 1);
    static_cast<ParOptProblem *>(self)->addSparseJacobianTranspose(alpha, vx.p(), vp.p(), vo.p());
    return 0;
  }
  static int tramp_winner(void *self, double alpha, po_vec x, po_vec cvec, po_vec A) {
    // the reference hands `A` over as a raw array (packed upper blocks for nwblock > 1)
    Arg vx(x, 0), vc(cvec, 0), va(A, 1);
    double *a = NULL;
    va.p()->getArray(&a);
    static_cast<ParOptProblem *>(self)->addSparseInnerProduct(alpha, vx.p(), vc.p(), a);
    return 0;
  }
  static int tramp_write(void *self, int iter, po_vec x) {
    Arg vx(x, 0);
    static_cast<ParOptProblem *>(self)->writeOutput(iter, vx.p());
    return 0;
  }
};

// ---- ParOptSparseProblem (src/ParOptProblem.h:301-395): fixed CSR pattern for the sparse Jacobian -------------
// Overlapping rows are allowed; the quasi-definite system is solved with the device sparse Cholesky
// (ParOptQuasiDefSparseMat of the reference).  `data` is a HOST array of nnz entries in the order of cols, as in
// the reference; it is uploaded after the call.
class ParOptSparseProblem : public ParOptProblem {
 public:
  explicit ParOptSparseProblem(ParOptComm _comm) : ParOptProblem(_comm) {}
  ParOptQuasiDefMat *createQuasiDefMat() { return new ParOptQuasiDefSparseMat(this); }
  // after setProblemSizes(), as in the reference (:306-312)
  void setSparseJacobianData(const int *_rowp, const int *_cols) {
    rowp.assign(_rowp, _rowp + nwcon + 1);
    cols.assign(_cols, _cols + rowp[nwcon]);
    data.assign(cols.size() > 0 ? cols.size() : 1, 0.0);
  }
  int getSparseJacobianData(const int **_rowp, const int **_cols, const ParOptScalar **_data) {
    if (_rowp) *_rowp = rowp.data();
    if (_cols) *_cols = cols.data();
    if (_data) *_data = data.data();
    return (int)cols.size();
  }
  virtual int evalSparseObjCon(ParOptVec *x, ParOptScalar *fobj, ParOptScalar *cons, ParOptVec *sparse_con) = 0;
  virtual int evalSparseObjConGradient(ParOptVec *x, ParOptVec *g, ParOptVec **Ac, ParOptScalar *data) = 0;
  // Facade extension for device-resident problems: the same evaluation with the Jacobian entries written straight
  // into the library's DEVICE array (nnz doubles in the order of cols; launch on po_ctx_stream(ctx)).  The default is
  // the reference's form above -- the host array filled by evalSparseObjConGradient, then one host-to-device copy of
  // nnz doubles per gradient evaluation (160 MB at 1 M constraints x 20 variables: more than a whole iteration of the
  // solver) -- so a problem whose data live in HBM overrides this one instead and never touches the host array.
  virtual int evalSparseObjConGradientDevice(ParOptVec *x, ParOptVec *g, ParOptVec **Ac, ParOptScalar *device_data) {
    int fail = evalSparseObjConGradient(x, g, Ac, data.data());
    if (po_ctx_memcpy(ctx, device_data, data.data(), (int64_t)sizeof(double) * (int64_t)cols.size(), 1) != 0) return 1;
    return fail;
  }
  // the base-class evaluations are never reached: the library calls the sparse forms (:348-358)
  int evalObjCon(ParOptVec *, ParOptScalar *, ParOptScalar *) { return 1; }
  int evalObjConGradient(ParOptVec *, ParOptVec *, ParOptVec **) { return 1; }
  // one line about the device factorization (ParOptQuasiDefMat::getFactorInfo)
  const char *getFactorInfo() { return hprob ? po_quasidef_factor_info(hprob) : NULL; }

 protected:
  void attachSparse() {
    if (po_problem_set_sparse_jacobian_data(hprob, nwcon, nwinequality, rowp.data(), cols.data(),
                                            &ParOptSparseProblem::tramp_sobjcon,
                                            &ParOptSparseProblem::tramp_sgrad) != 0) {
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    }
  }

 private:
  std::vector<int> rowp, cols;
  std::vector<ParOptScalar> data;
  static int tramp_sobjcon(void *self, po_vec x, double *fobj, double *cons, po_vec sparse) {
    Arg vx(x, 0), vs(sparse, 1);
    return static_cast<ParOptSparseProblem *>(self)->evalSparseObjCon(vx.p(), fobj, cons, vs.p());
  }
  static int tramp_sgrad(void *self, po_vec x, po_vec g, const po_vec *Ac, double *ddata, int64_t nnz) {
    ParOptSparseProblem *me = static_cast<ParOptSparseProblem *>(self);
    int fail = 0;
    {
      Arg vx(x, 0), vg(g, 1);
      std::vector<Arg *> args;
      std::vector<ParOptVec *> va(me->ncon > 0 ? me->ncon : 1, (ParOptVec *)NULL);
      for (int j = 0; Ac && j < me->ncon; j++) {
        args.push_back(new Arg(Ac[j], 1));
        va[j] = args.back()->p();
      }
      fail = me->evalSparseObjConGradientDevice(vx.p(), vg.p(), (Ac || me->ncon == 0) ? va.data() : NULL, ddata);
      for (Arg *a : args) delete a;
    }
    (void)nnz;
    return fail;
  }
};

// ---- compact quasi-Newton -----------------------------------------------------------------------
enum ParOptBFGSUpdateType { PAROPT_SKIP_NEGATIVE_CURVATURE, PAROPT_DAMPED_UPDATE };
enum ParOptQuasiNewtonDiagonalType {
  PAROPT_YTY_OVER_YTS,
  PAROPT_YTS_OVER_STS,
  PAROPT_INNER_PRODUCT_YTY_OVER_YTS,
  PAROPT_INNER_PRODUCT_YTS_OVER_STS
};

class ParOptCompactQuasiNewton : public ParOptBase {
 public:
  virtual ~ParOptCompactQuasiNewton() {
    for (ParOptVec *v : zwrap) v->decref();
    if (h) po_qn_destroy(h);
  }
  void setInitDiagonalType(ParOptQuasiNewtonDiagonalType t) {
    po_qn_set_diag_type(h, t == PAROPT_YTS_OVER_STS ? PO_QN_YTS_OVER_STS : PO_QN_YTY_OVER_YTS);
  }
  void reset() { po_qn_reset(h); }
  virtual int update(ParOptVec *, const ParOptScalar *, ParOptVec *, ParOptVec *s, ParOptVec *y) {
    int rc = 0;
    po_qn_update(h, s->handle(), y->handle(), &rc);
    return rc;
  }
  // multiplier-only update (src/ParOptQuasiNewton.h:60-63): a no-op for the limited-memory classes
  virtual int update(ParOptVec *, const ParOptScalar *, ParOptVec *) { return 0; }
  void mult(ParOptVec *x, ParOptVec *y) { po_qn_mult(h, x->handle(), y->handle()); }
  void multAdd(ParOptScalar alpha, ParOptVec *x, ParOptVec *y) { po_qn_mult_add(h, alpha, x->handle(), y->handle()); }
  int getCompactMat(ParOptScalar *b0, const ParOptScalar **d, const ParOptScalar **M, ParOptVec ***Z) {
    int k = 0;
    const po_vec *zs = NULL;
    po_qn_get_compact(h, &k, b0, d, M, &zs);
    if (Z) {
      for (ParOptVec *v : zwrap) v->decref();
      zwrap.clear();
      for (int i = 0; i < k; i++) {
        zwrap.push_back(new ParOptBasicVec(zs[i]));
        zwrap.back()->incref();
      }
      *Z = zwrap.data();
    }
    return k;
  }
  int getMaxLimitedMemorySize() { int k = 0; po_qn_max_size(h, &k); return k; }
  po_qn handle() { return h; }

 protected:
  ParOptCompactQuasiNewton() : h(NULL) {}
  po_qn h;
  std::vector<ParOptVec *> zwrap;
};

class ParOptLBFGS : public ParOptCompactQuasiNewton {
 public:
  ParOptLBFGS(ParOptProblem *prob, int subspace) {
    int n;
    prob->getProblemSizes(&n, NULL, NULL);
    po_qn_create(prob->getContext(), PO_QN_BFGS, n, subspace, &h);
  }
  void setBFGSUpdateType(ParOptBFGSUpdateType t) {
    po_qn_set_update_type(h, t == PAROPT_DAMPED_UPDATE ? PO_BFGS_DAMPED_UPDATE : PO_BFGS_SKIP_NEGATIVE_CURVATURE);
  }
};

class ParOptLSR1 : public ParOptCompactQuasiNewton {
 public:
  ParOptLSR1(ParOptProblem *prob, int subspace) {
    int n;
    prob->getProblemSizes(&n, NULL, NULL);
    po_qn_create(prob->getContext(), PO_QN_SR1, n, subspace, &h);
  }
};

// ---- ParOptInteriorPoint ------------------------------------------------------------------------
class ParOptInteriorPoint : public ParOptBase {
 public:
  // the option set of ParOptInteriorPoint::addDefaultOptions (src/ParOptInteriorPoint.cpp:536-727)
  static void addDefaultOptions(ParOptOptions *options) { options->addLibraryDefaults(0); }
  ParOptInteriorPoint(ParOptProblem *_prob, ParOptOptions *_options = NULL)
      : prob(_prob), options(_options), ip(NULL), x(NULL), zl(NULL), zu(NULL), zw(NULL), sw(NULL), tw(NULL) {
    prob->incref();
    if (options) options->incref();
    if (po_ip_create(prob->handle(), &ip) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    applyOptions();
  }
  ParOptOptions *getOptions() { return options; }
  // the options object is shared and may change between the constructor and optimize(), as in the reference
  void applyOptions() {
    if (!ip || !options) return;
    po_ip h = ip;
    int bad = options->forward(
        0, [h](const char *n, const char *v) { return po_ip_set_option_str(h, n, v); },
        [h](const char *n, int v) { return po_ip_set_option_int(h, n, v); },
        [h](const char *n, double v) { return po_ip_set_option_float(h, n, v); });
    if (bad) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  ~ParOptInteriorPoint() {
    if (x) x->decref();
    if (zl) zl->decref();
    if (zu) zu->decref();
    if (zw) zw->decref();
    if (sw) sw->decref();
    if (tw) tw->decref();
    if (ip) po_ip_destroy(ip);
    if (options) options->decref();
    prob->decref();
  }
  ParOptProblem *getOptProblem() { return prob; }
  int optimize(const char *checkpoint = NULL) {
    applyOptions();
    return ip ? po_ip_optimize(ip, checkpoint) : 1;
  }
  void getProblemSizes(int *nvars, int *ncon, int *nwcon) { prob->getProblemSizes(nvars, ncon, nwcon); }
  // borrowed internals, as in the reference (src/ParOptInteriorPoint.cpp:793-826)
  void getOptimizedPoint(ParOptVec **_x, ParOptScalar **_z, ParOptVec **_zw, ParOptVec **_zl, ParOptVec **_zu) {
    po_vec hx, hzl, hzu;
    const double *z;
    po_ip_get_optimized_point(ip, &hx, &z, &hzl, &hzu);
    wrap(&x, hx);
    wrap(&zl, hzl);
    wrap(&zu, hzu);
    if (_x) *_x = x;
    if (_z) *_z = const_cast<double *>(z);
    if (_zw) {
      po_vec hzw = NULL;
      po_ip_get_optimized_sparse(ip, &hzw, NULL, NULL, NULL, NULL);
      wrap(&zw, hzw);
      *_zw = zw;
    }
    if (_zl) *_zl = zl;
    if (_zu) *_zu = zu;
  }
  void getOptimizedSlacks(ParOptScalar **s, ParOptScalar **t, ParOptVec **_sw, ParOptVec **_tw) {
    const double *ps, *pt, *pzs, *pzt;
    po_ip_get_optimized_slacks(ip, &ps, &pt, &pzs, &pzt);
    if (s) *s = const_cast<double *>(ps);
    if (t) *t = const_cast<double *>(pt);
    if (_sw || _tw) {
      po_vec hsw = NULL, htw = NULL;
      po_ip_get_optimized_sparse(ip, NULL, &hsw, &htw, NULL, NULL);
      wrap(&sw, hsw);
      wrap(&tw, htw);
      if (_sw) *_sw = sw;
      if (_tw) *_tw = tw;
    }
  }
  void getIterationCounters(int *niter = NULL, int *neval = NULL, int *ngeval = NULL, int *nhvec = NULL) {
    po_ip_get_counters(ip, niter, neval, ngeval);
    if (nhvec) po_ip_get_hvec_count(ip, nhvec);
  }
  double getBarrierParameter() { double v = 0; po_ip_get_barrier_parameter(ip, &v); return v; }
  ParOptScalar getComplementarity() { double v = 0; po_ip_get_complementarity(ip, &v); return v; }
  void setPenaltyGamma(double gamma) { po_ip_set_penalty_gamma(ip, gamma); }
  void setPenaltyGamma(const double *gamma) { po_ip_set_penalty_gamma_array(ip, gamma); }
  // the caller keeps ownership of (and must keep alive) the approximation; NULL detaches it
  void setQuasiNewton(ParOptCompactQuasiNewton *qn) { po_ip_set_quasi_newton(ip, qn ? qn->handle() : NULL); }
  void resetProblemInstance(ParOptProblem *problem) {
    if (po_ip_reset_problem_instance(ip, problem->handle()) == 0) {
      problem->incref();
      prob->decref();
      prob = problem;
    } else {
      fprintf(stderr, "ParOpt: Incompatible problem instance\n");
    }
  }
  void resetQuasiNewtonHessian() { po_ip_reset_quasi_newton(ip); }
  void resetDesignAndBounds() { po_ip_reset_design_and_bounds(ip); }
  // checkGradients(dh) (src/ParOptInteriorPoint.h:166, .cpp:6196-6199): the problem's finite-difference check at the
  // solver's current point (with the Hessian-vector product when use_hvec_product is set); prints the report
  void checkGradients(double dh) {
    applyOptions();
    const char *report = NULL;
    if (po_ip_check_gradients(ip, dh, &report) != 0) {
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    } else if (report) {
      int rank = 0;
      po_ctx_rank(prob->getContext(), &rank, NULL);
      if (rank == 0) fputs(report, stdout);
    }
  }
  // setBFGSUpdateType (.h:172, .cpp:1179-1186): applies to the solver's own L-BFGS object
  void setBFGSUpdateType(ParOptBFGSUpdateType update) {
    po_qn q = NULL;
    if (po_ip_get_quasi_newton(ip, &q) == 0 && q)
      po_qn_set_update_type(q, update == PAROPT_DAMPED_UPDATE ? PO_BFGS_DAMPED_UPDATE : PO_BFGS_SKIP_NEGATIVE_CURVATURE);
  }
  // setUseDiagHessian (.h:181; declared but never defined in the reference): the use_diag_hessian option
  void setUseDiagHessian(int truth) {
    if (options) options->setOption("use_diag_hessian", truth ? 1 : 0);
    po_ip_set_option_int(ip, "use_diag_hessian", truth ? 1 : 0);
  }
  // checkMeritFuncGradient(xpt, dh) (.h:199, .cpp:3280-3432): prints "Merit function test" and the two derivatives
  void checkMeritFuncGradient(ParOptVec *xpt = NULL, double dh = 1e-6) {
    applyOptions();
    if (po_ip_check_merit_func_gradient(ip, xpt ? xpt->handle() : NULL, dh, NULL, NULL) != 0)
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  po_ip handle() { return ip; }
  int writeSolutionFile(const char *filename) { return po_ip_write_solution_file(ip, filename); }
  int readSolutionFile(const char *filename) { return po_ip_read_solution_file(ip, filename); }
  const char *getHistory() { const char *t = ""; po_ip_get_history(ip, &t); return t; }

 private:
  void wrap(ParOptVec **slot, po_vec h) {
    if (*slot) (*slot)->decref();
    *slot = NULL;
    if (h) {
      *slot = new ParOptBasicVec(h);
      (*slot)->incref();
    }
  }
  ParOptProblem *prob;
  ParOptOptions *options;
  po_ip ip;
  ParOptVec *x, *zl, *zu, *zw, *sw, *tw;
};

// ---- the trust-region layer, class by class (src/ParOptTrustRegion.h:15-480, ----------------------------------
// src/ParOptCompactEigenvalueApprox.h:7-206).  The objects are handles to the library's device-side
// implementations; they are assembled exactly as the reference's user code assembles them
// (examples/eigenvalue/eigenvalue_opt.py:298-308, src/ParOptOptimizer.cpp:108-183):
//     qn = new ParOptLBFGS(problem, m);                      approx = new ParOptCompactEigenApprox(problem, N);
//     eig_qn = new ParOptEigenQuasiNewton(qn, approx, 0);    sub = new ParOptEigenSubproblem(problem, eig_qn);
//     sub->setEigenModelUpdate(data, update);                ip = new ParOptInteriorPoint(sub, options);
//     tr = new ParOptTrustRegion(sub, options);              tr->optimize(ip);
// c(s) = c0 + g0^T s + 1/2 s^T H M H^T s with N curvature directions H = [h_0 .. h_{N-1}]
class ParOptCompactEigenApprox : public ParOptBase {
 public:
  ParOptCompactEigenApprox(ParOptProblem *problem, int _N) : h(NULL), g0w(NULL) {
    if (po_eig_create(problem->handle(), _N, &h) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  ~ParOptCompactEigenApprox() {
    dropWrappers();
    if (h) po_eig_destroy(h);
  }
  void multAdd(ParOptScalar alpha, ParOptVec *x, ParOptVec *y) { po_eig_mult_add(h, alpha, x->handle(), y->handle()); }
  // borrowed pointers into the approximation: the caller writes c0, M, Minv directly and the vectors through
  // getArray (what it wrote is uploaded when the model-update callback returns, or by releaseArray(1))
  void getApproximation(ParOptScalar **_c0, ParOptVec **_g0, int *_N, ParOptScalar **_M, ParOptScalar **_Minv,
                        ParOptVec ***_hvecs) {
    po_vec g0 = NULL;
    const po_vec *hv = NULL;
    int N = 0;
    po_eig_get_approximation(h, _c0, &g0, &N, _M, _Minv, &hv);
    if (!g0w) {
      g0w = new ParOptBasicVec(g0);
      g0w->incref();
      for (int i = 0; i < N; i++) {
        hw.push_back(new ParOptBasicVec(hv[i]));
        hw.back()->incref();
      }
    }
    if (_g0) *_g0 = g0w;
    if (_N) *_N = N;
    if (_hvecs) *_hvecs = hw.data();
  }
  ParOptScalar evalApproximation(ParOptVec *s, ParOptVec *t) {
    double v = 0.0;
    po_eig_eval_approximation(h, s ? s->handle() : NULL, t ? t->handle() : NULL, &v);
    return v;
  }
  void evalApproximationGradient(ParOptVec *s, ParOptVec *grad) {
    po_eig_eval_approximation_gradient(h, s->handle(), grad->handle());
  }
  po_eig handle() { return h; }

 private:
  void dropWrappers() {
    if (g0w) g0w->decref();
    g0w = NULL;
    for (ParOptVec *v : hw) v->decref();
    hw.clear();
  }
  po_eig h;
  ParOptVec *g0w;
  std::vector<ParOptVec *> hw;
};

// B = B_qn - z0 * H M H^T as one compact matrix over [Z_qn | H]; z0 follows the multiplier of constraint `index`
class ParOptEigenQuasiNewton : public ParOptCompactQuasiNewton {
 public:
  ParOptEigenQuasiNewton(ParOptCompactQuasiNewton *_qn, ParOptCompactEigenApprox *_eigh, int _index = 0)
      : qn(_qn), eigh(_eigh), index(_index) {
    if (qn) qn->incref();
    eigh->incref();
    if (po_eigqn_create(qn ? qn->handle() : NULL, eigh->handle(), index, &h) != 0)
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  ~ParOptEigenQuasiNewton() {
    if (h) po_qn_destroy(h);  // before the objects it borrows
    h = NULL;
    eigh->decref();
    if (qn) qn->decref();
  }
  void setUseQuasiNewtonObjective(int truth) { po_eigqn_set_use_quasi_newton_objective(h, truth); }
  // the curvature pair goes to the quasi-Newton object itself (acceptTrialStep); here only z0 is recorded
  int update(ParOptVec *, const ParOptScalar *z, ParOptVec *, ParOptVec *, ParOptVec *) {
    po_eigqn_update_multipliers(h, z);
    return 0;
  }
  int update(ParOptVec *, const ParOptScalar *z, ParOptVec *) {
    po_eigqn_update_multipliers(h, z);
    return 0;
  }
  ParOptCompactQuasiNewton *getCompactQuasiNewton() { return qn; }
  ParOptCompactEigenApprox *getCompactEigenApprox() { return eigh; }
  int getMultiplierIndex() { return index; }

 private:
  ParOptCompactQuasiNewton *qn;
  ParOptCompactEigenApprox *eigh;
  int index;
};

// The subproblem interface (src/ParOptTrustRegion.h:15-151).  The two library forms below implement it on the
// device.  A USER-written subclass (round 5) is driven through a callback table: ParOptTrustRegion / ParOptOptimizer::
// setTrustRegionSubproblem take it like a library one (po_trsub_create_callbacks) -- its seven
// virtuals and its ParOptProblem side (getVarsAndBounds / evalObjCon / evalObjConGradient of the MODEL, in the step)
// are called from the device solver.
class ParOptTrustRegionSubproblem : public ParOptProblem {
 public:
  explicit ParOptTrustRegionSubproblem(ParOptComm _comm) : ParOptProblem(_comm), user_sub(NULL) {}
  ~ParOptTrustRegionSubproblem() {
    if (user_sub) po_trsub_destroy(user_sub);
    for (ParOptVec *v : keep) v->decref();
  }
  virtual ParOptCompactQuasiNewton *getQuasiNewton() = 0;
  virtual void initModelAndBounds(double tr_size) = 0;
  virtual void setTrustRegionBounds(double tr_size) = 0;
  virtual int evalTrialStepAndUpdate(int update_flag, ParOptVec *step, ParOptScalar *z, ParOptVec *zw,
                                     ParOptScalar *fobj, ParOptScalar *cons) = 0;
  virtual int acceptTrialStep(ParOptVec *xt, ParOptScalar *z, ParOptVec *zw) = 0;
  virtual void rejectTrialStep() = 0;
  virtual int getQuasiNewtonUpdateType() { return 0; }
  virtual int getLinearModel(ParOptVec **_xk = NULL, ParOptScalar *fk = NULL, ParOptVec **gk = NULL,
                             const ParOptScalar **ck = NULL, ParOptVec ***Ak = NULL, ParOptVec **lb = NULL,
                             ParOptVec **ub = NULL) = 0;
  // The library object behind the subproblem: a library-backed one returns its own; a user-written one gets a
  // callback-backed object on first use, built over the subproblem's own ParOptProblem side (ParOptProblem::handle():
  // sizes, inequality counts and -- already in model form -- its sparse-constraint callbacks).
  virtual po_trsub subHandle() {
    if (!user_sub) {
      po_problem self = ParOptProblem::handle();
      if (!self) return NULL;
      po_trsub_callbacks cb;
      memset(&cb, 0, sizeof(cb));
      cb.user = this;
      cb.get_quasi_newton = &ParOptTrustRegionSubproblem::ts_qn;
      cb.init_model_and_bounds = &ParOptTrustRegionSubproblem::ts_init;
      cb.set_trust_region_bounds = &ParOptTrustRegionSubproblem::ts_bounds;
      cb.eval_trial_step_and_update = &ParOptTrustRegionSubproblem::ts_trial;
      cb.accept_trial_step = &ParOptTrustRegionSubproblem::ts_accept;
      cb.reject_trial_step = &ParOptTrustRegionSubproblem::ts_reject;
      cb.get_quasi_newton_update_type = &ParOptTrustRegionSubproblem::ts_utype;
      cb.get_linear_model = &ParOptTrustRegionSubproblem::ts_model;
      cb.get_vars_and_bounds = &ParOptTrustRegionSubproblem::ts_vars;
      cb.eval_obj_con = &ParOptTrustRegionSubproblem::ts_eval;
      cb.eval_obj_con_gradient = &ParOptTrustRegionSubproblem::ts_grad;
      cb.sparse_constraints_are_model = 1;
      if (po_trsub_create_callbacks(self, &cb, &user_sub) != 0) {
        fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
        user_sub = NULL;
      }
    }
    return user_sub;
  }
  // the po_problem the interior point is built on: the callback-backed subproblem (NOT the plain problem side)
  po_problem handle() {
    po_trsub h = subHandle();
    po_problem p = NULL;
    if (h) po_trsub_problem(h, &p);
    return p;
  }

 private:
  typedef ParOptTrustRegionSubproblem Self;
  static Self *me(void *u) { return static_cast<Self *>(u); }
  static int ts_qn(void *u, po_qn *qn) {
    ParOptCompactQuasiNewton *q = me(u)->getQuasiNewton();
    *qn = q ? q->handle() : NULL;
    return 0;
  }
  static int ts_init(void *u, double tr) {
    me(u)->initModelAndBounds(tr);
    return 0;
  }
  static int ts_bounds(void *u, double tr) {
    me(u)->setTrustRegionBounds(tr);
    return 0;
  }
  static int ts_trial(void *u, int flag, po_vec step, const double *z, po_vec zw, double *fobj, double *cons) {
    Arg vs(step, 0), vw(zw, 0);
    return me(u)->evalTrialStepAndUpdate(flag, vs.p(), const_cast<double *>(z), zw ? vw.p() : NULL, fobj, cons);
  }
  static int ts_accept(void *u, po_vec step, const double *z, po_vec zw) {
    Arg vs(step, 0), vw(zw, 0);
    return me(u)->acceptTrialStep(vs.p(), const_cast<double *>(z), zw ? vw.p() : NULL);
  }
  static int ts_reject(void *u) {
    me(u)->rejectTrialStep();
    return 0;
  }
  static int ts_utype(void *u) { return me(u)->getQuasiNewtonUpdateType(); }
  static int ts_model(void *u, po_vec *xk, double *fk, po_vec *gk, const double **ck, const po_vec **Ak, po_vec *lb,
                      po_vec *ub) {
    Self *s = me(u);
    ParOptVec *vx = NULL, *vg = NULL, *vl = NULL, *vu = NULL, **va = NULL;
    const int m = s->getLinearModel(&vx, fk, &vg, ck, &va, &vl, &vu);
    if (!vx || !vg || !vl || !vu) return 1;
    *xk = vx->handle();
    *gk = vg->handle();
    *lb = vl->handle();
    *ub = vu->handle();
    s->akh.resize(m > 0 ? m : 1);
    for (int i = 0; i < m; i++) s->akh[i] = va[i]->handle();
    *Ak = s->akh.data();
    return 0;
  }
  static int ts_vars(void *u, po_vec x, po_vec l, po_vec up) {
    Arg vx(x, 1), vl(l, 1), vu(up, 1);
    me(u)->getVarsAndBounds(vx.p(), vl.p(), vu.p());
    return 0;
  }
  static int ts_eval(void *u, po_vec step, double *fobj, double *cons) {
    Arg vs(step, 0);
    return me(u)->evalObjCon(step ? vs.p() : NULL, fobj, cons);
  }
  static int ts_grad(void *u, po_vec step, po_vec g, const po_vec *Ac) {
    Self *s = me(u);
    int fail = 0;
    {
      Arg vs(step, 0), vg(g, 1);
      std::vector<Arg *> args;
      std::vector<ParOptVec *> va(s->ncon > 0 ? s->ncon : 1, (ParOptVec *)NULL);
      for (int j = 0; Ac && j < s->ncon; j++) {
        args.push_back(new Arg(Ac[j], 1));
        va[j] = args.back()->p();
      }
      fail = s->evalObjConGradient(vs.p(), vg.p(), (Ac || s->ncon == 0) ? va.data() : NULL);
      for (Arg *a : args) delete a;
    }
    return fail;
  }
  po_trsub user_sub;
  std::vector<po_vec> akh;
  std::vector<ParOptVec *> keep;
};

// common part of the two library-backed subproblems: everything forwards to the po_trsub object
class ParOptLibrarySubproblem : public ParOptTrustRegionSubproblem {
 public:
  ~ParOptLibrarySubproblem() {
    dropWrappers();
    if (sub) po_trsub_destroy(sub);
    if (base) base->decref();
  }
  po_trsub subHandle() { return sub; }
  po_problem handle() {
    po_problem p = NULL;
    if (sub) po_trsub_problem(sub, &p);
    return p;
  }
  void initModelAndBounds(double tr_size) { check(po_trsub_init_model_and_bounds(sub, tr_size)); }
  void setTrustRegionBounds(double tr_size) { check(po_trsub_set_trust_region_bounds(sub, tr_size)); }
  int evalTrialStepAndUpdate(int update_flag, ParOptVec *step, ParOptScalar *z, ParOptVec *zw, ParOptScalar *fobj,
                             ParOptScalar *cons) {
    return po_trsub_eval_trial_step_and_update(sub, update_flag, step->handle(), z, zw ? zw->handle() : NULL, fobj,
                                               cons) != 0;
  }
  int acceptTrialStep(ParOptVec *step, ParOptScalar *z, ParOptVec *zw) {
    return po_trsub_accept_trial_step(sub, step->handle(), z, zw ? zw->handle() : NULL) != 0;
  }
  void rejectTrialStep() { po_trsub_reject_trial_step(sub); }
  int getQuasiNewtonUpdateType() {
    int t = 0;
    po_trsub_get_quasi_newton_update_type(sub, &t);
    return t;
  }
  int getLinearModel(ParOptVec **_xk = NULL, ParOptScalar *_fk = NULL, ParOptVec **_gk = NULL,
                     const ParOptScalar **_ck = NULL, ParOptVec ***_Ak = NULL, ParOptVec **_lb = NULL,
                     ParOptVec **_ub = NULL) {
    po_vec xk, gk, lb, ub;
    const po_vec *Ak = NULL;
    int m = 0;
    po_trsub_get_linear_model(sub, &xk, _fk, &gk, _ck, &Ak, &lb, &ub, &m);
    dropWrappers();
    po_vec hs[4] = {xk, gk, lb, ub};
    for (int i = 0; i < 4; i++) {
      lin[i] = new ParOptBasicVec(hs[i]);
      lin[i]->incref();
    }
    for (int i = 0; i < m; i++) {
      akw.push_back(new ParOptBasicVec(Ak[i]));
      akw.back()->incref();
    }
    if (_xk) *_xk = lin[0];
    if (_gk) *_gk = lin[1];
    if (_lb) *_lb = lin[2];
    if (_ub) *_ub = lin[3];
    if (_Ak) *_Ak = akw.data();
    return m;
  }
  // the ParOptProblem side (what the interior-point solver sees) answers through the library object as well
  void getVarsAndBounds(ParOptVec *x, ParOptVec *lb, ParOptVec *ub) {
    check(po_problem_get_vars_and_bounds(handle(), x->handle(), lb->handle(), ub->handle()));
  }
  int evalObjCon(ParOptVec *x, ParOptScalar *fobj, ParOptScalar *cons) {
    return po_problem_eval_obj_con(handle(), x->handle(), fobj, cons) != 0;
  }
  int evalObjConGradient(ParOptVec *x, ParOptVec *g, ParOptVec **Ac) {
    std::vector<po_vec> hs(ncon > 0 ? ncon : 1, (po_vec)NULL);
    for (int i = 0; Ac && i < ncon; i++) hs[i] = Ac[i]->handle();
    return po_problem_eval_obj_con_gradient(handle(), x->handle(), g->handle(), Ac ? hs.data() : NULL) != 0;
  }
  void writeOutput(int iter, ParOptVec *x) { base->writeOutput(iter, x); }

 protected:
  ParOptLibrarySubproblem(ParOptProblem *_base) : ParOptTrustRegionSubproblem(_base->getMPIComm()), base(_base), sub(NULL) {
    base->incref();
    int nv = 0, nc = 0, nwc = 0, nineq = 0, nwineq = 0;
    base->getProblemSizes(&nv, &nc, &nwc);
    base->getNumInequalities(&nineq, &nwineq);
    setProblemSizes(nv, nc, nwc);
    setNumInequalities(nineq, nwineq);
    for (int i = 0; i < 4; i++) lin[i] = NULL;
  }
  void check(int rc) {
    if (rc != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  void dropWrappers() {
    for (int i = 0; i < 4; i++) {
      if (lin[i]) lin[i]->decref();
      lin[i] = NULL;
    }
    for (ParOptVec *v : akw) v->decref();
    akw.clear();
  }
  ParOptProblem *base;
  po_trsub sub;
  ParOptVec *lin[4];
  std::vector<ParOptVec *> akw;
};

// quadratic model with the compact quasi-Newton Hessian (src/ParOptTrustRegion.h:153-300); _qn may be NULL
class ParOptQuadraticSubproblem : public ParOptLibrarySubproblem {
 public:
  ParOptQuadraticSubproblem(ParOptProblem *_problem, ParOptCompactQuasiNewton *_qn)
      : ParOptLibrarySubproblem(_problem), qn(_qn) {
    if (qn) qn->incref();
    check(po_trsub_create_quadratic(_problem->handle(), qn ? qn->handle() : NULL, &sub));
  }
  ~ParOptQuadraticSubproblem() {
    if (sub) po_trsub_destroy(sub);  // before the quasi-Newton object it borrows
    sub = NULL;
    if (qn) qn->decref();
  }
  ParOptCompactQuasiNewton *getQuasiNewton() { return qn; }

 private:
  ParOptCompactQuasiNewton *qn;
};

// the compact eigenvalue model of one constraint under the same driver (...EigenvalueApprox.h:86-206)
class ParOptEigenSubproblem : public ParOptLibrarySubproblem {
 public:
  ParOptEigenSubproblem(ParOptProblem *_problem, ParOptEigenQuasiNewton *_qn)
      : ParOptLibrarySubproblem(_problem), approx(_qn), data(NULL), updateEigenModel(NULL) {
    approx->incref();
    check(po_trsub_create_eigen(_problem->handle(), approx->handle(), &sub));
  }
  ~ParOptEigenSubproblem() {
    if (sub) po_trsub_destroy(sub);
    sub = NULL;
    approx->decref();
  }
  // update(data, x, approx) is called at the starting point and at every accepted point with c0 and g0 preset to
  // the constraint's value and gradient; it fills hvecs, M and Minv (and may change c0 / g0)
  void setEigenModelUpdate(void *_data, void (*update)(void *, ParOptVec *, ParOptCompactEigenApprox *)) {
    data = _data;
    updateEigenModel = update;
    check(po_trsub_set_eigen_model_update(sub, update ? &ParOptEigenSubproblem::tramp_update : NULL, this));
  }
  ParOptCompactQuasiNewton *getQuasiNewton() { return approx; }

 private:
  static int tramp_update(void *self, po_vec x, po_eig) {
    ParOptEigenSubproblem *me = static_cast<ParOptEigenSubproblem *>(self);
    Arg vx(x, 0);
    // the library's approximation IS the one behind the caller's object: hand that object over, as the reference does
    me->updateEigenModel(me->data, vx.p(), me->approx->getCompactEigenApprox());
    return 0;
  }
  ParOptEigenQuasiNewton *approx;
  void *data;
  void (*updateEigenModel)(void *, ParOptVec *, ParOptCompactEigenApprox *);
};

// ParOptInfeasSubproblem (src/ParOptTrustRegion.h:293-374, .cpp:468-650): the steering problem of the trust-region
// driver as a ParOptProblem of its own -- works over any subproblem (the two library ones and user-written subclasses)
class ParOptInfeasSubproblem : public ParOptProblem {
 public:
  static const int PAROPT_SUBPROBLEM_OBJECTIVE = 1;
  static const int PAROPT_LINEAR_OBJECTIVE = 2;
  static const int PAROPT_CONSTANT_OBJECTIVE = 3;
  static const int PAROPT_SUBPROBLEM_CONSTRAINT = 1;
  static const int PAROPT_LINEAR_CONSTRAINT = 2;

  ParOptInfeasSubproblem(ParOptTrustRegionSubproblem *_prob, int subproblem_objective, int subproblem_constraint)
      : ParOptProblem(_prob->getMPIComm()), prob(_prob), infeas(NULL) {
    prob->incref();
    int nv = 0, nc = 0, nwc = 0, nineq = 0, nwineq = 0;
    prob->getProblemSizes(&nv, &nc, &nwc);
    prob->getNumInequalities(&nineq, &nwineq);
    setProblemSizes(nv, nc, nwc);
    setNumInequalities(nineq, nwineq);
    if (!prob->subHandle() ||
        po_infeas_create(prob->subHandle(), subproblem_objective, subproblem_constraint, &infeas) != 0)
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  ~ParOptInfeasSubproblem() {
    if (infeas) po_problem_destroy(infeas);
    prob->decref();
  }
  void setObjectiveScaling(ParOptScalar _scale) {
    if (infeas) po_infeas_set_objective_scaling(infeas, _scale);
  }
  po_problem handle() { return infeas; }
  ParOptQuasiDefMat *createQuasiDefMat() { return prob->createQuasiDefMat(); }
  int isSparseInequality() { return prob->isSparseInequality(); }
  int useLowerBounds() { return 1; }
  int useUpperBounds() { return 1; }
  void getVarsAndBounds(ParOptVec *x, ParOptVec *lb, ParOptVec *ub) {
    if (po_problem_get_vars_and_bounds(infeas, x->handle(), lb->handle(), ub->handle()) != 0)
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  int evalObjCon(ParOptVec *x, ParOptScalar *fobj, ParOptScalar *cons) {
    return po_problem_eval_obj_con(infeas, x->handle(), fobj, cons) != 0;
  }
  int evalObjConGradient(ParOptVec *x, ParOptVec *g, ParOptVec **Ac) {
    std::vector<po_vec> hs(ncon > 0 ? ncon : 1, (po_vec)NULL);
    for (int i = 0; Ac && i < ncon; i++) hs[i] = Ac[i]->handle();
    return po_problem_eval_obj_con_gradient(infeas, x->handle(), g->handle(), Ac ? hs.data() : NULL) != 0;
  }

 private:
  ParOptTrustRegionSubproblem *prob;
  po_problem infeas;
};

// ParOptTrustRegion (src/ParOptTrustRegion.h:376-480): SL1QP with the adaptive penalty update, or the filter method
class ParOptTrustRegion : public ParOptBase {
 public:
  static void addDefaultOptions(ParOptOptions *options) { options->addLibraryDefaults(1); }
  ParOptTrustRegion(ParOptTrustRegionSubproblem *_subproblem, ParOptOptions *_options = NULL)
      : subproblem(_subproblem), options(_options), tr(NULL), x(NULL) {
    subproblem->incref();
    if (!options) {
      options = new ParOptOptions();
      addDefaultOptions(options);
    }
    options->incref();
    if (!subproblem->subHandle()) {
      fprintf(stderr, "ParOptAMD: the trust-region subproblem could not be attached to the library: %s\n",
              po_last_error());
    } else if (po_tr_create_subproblem(subproblem->subHandle(), &tr) != 0) {
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    }
    applyOptions();
  }
  ~ParOptTrustRegion() {
    if (x) x->decref();
    if (tr) po_tr_destroy(tr);
    options->decref();
    subproblem->decref();
  }
  ParOptOptions *getOptions() { return options; }
  void initialize() {
    applyOptions();
    if (tr && po_tr_initialize(tr) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  void setPenaltyGamma(double gamma) {
    applyOptions();
    if (tr) po_tr_set_penalty_gamma(tr, gamma);
  }
  void setPenaltyGamma(const double *gamma) {
    applyOptions();
    if (tr) po_tr_set_penalty_gamma_array(tr, gamma);
  }
  int getPenaltyGamma(const double **gamma) {
    int m = 0;
    subproblem->getProblemSizes(NULL, &m, NULL);
    if (tr) po_tr_get_state(tr, NULL, NULL, NULL, NULL, gamma, NULL, NULL);
    return m;
  }
  // declared by the reference (src/ParOptTrustRegion.h:394-395) without a definition: the two options
  void setPenaltyGammaMax(double gamma_max) { options->setOption("tr_penalty_gamma_max", gamma_max); }
  void setPenaltyGammaMin(double gamma_min) { options->setOption("tr_penalty_gamma_min", gamma_min); }
  // optimize(ip): `optimizer` must have been built on this subproblem (ParOptInteriorPoint(subproblem, options))
  void optimize(ParOptInteriorPoint *optimizer) {
    if (!tr || !optimizer) return;
    optimizer->applyOptions();
    applyOptions();
    if (po_tr_optimize_with(tr, optimizer->handle()) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  void getOptimizedPoint(ParOptVec **_x) {
    if (_x && subproblem) subproblem->getLinearModel(_x);
  }
  const char *getHistory() {
    const char *t = "";
    if (tr) po_tr_get_history(tr, &t);
    return t;
  }
  po_tr handle() { return tr; }

 private:
  void applyOptions() {
    if (!tr) return;
    po_tr h = tr;
    if (options->forward(
            1, [h](const char *n, const char *v) { return po_tr_set_option_str(h, n, v); },
            [h](const char *n, int v) { return po_tr_set_option_int(h, n, v); },
            [h](const char *n, double v) { return po_tr_set_option_float(h, n, v); }) != 0)
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  ParOptTrustRegionSubproblem *subproblem;
  ParOptOptions *options;
  po_tr tr;
  ParOptVec *x;
};

// ---- ParOptOptimizer: algorithm = "ip" | "tr" | "mma" (src/ParOptOptimizer.cpp:65-206) --------------
// The reference's generic entry point.  "tr" builds the quasi-Newton object, the quadratic
// subproblem, the interior-point sub-solver and the trust-region driver exactly as :108-183 does.
class ParOptOptimizer : public ParOptBase {
 public:
  // src/ParOptOptimizer.cpp:38-49: "algorithm", "ip_checkpoint_file" and the three drivers' option sets
  static void addDefaultOptions(ParOptOptions *options) {
    const char *optimizers[3] = {"ip", "tr", "mma"};
    options->addEnumOption("algorithm", "tr", 3, optimizers, "optimization algorithm");
    options->addStringOption("ip_checkpoint_file", NULL, "checkpoint file of the interior-point method");
    options->addLibraryDefaults(0);
    options->addLibraryDefaults(1);
    options->addLibraryDefaults(2);
  }
  ParOptOptions *getOptions() { return options; }
  ParOptProblem *getProblem() { return prob; }
  ParOptOptimizer(ParOptProblem *_prob, ParOptOptions *_options)
      : prob(_prob), options(_options), ip(NULL), tr(NULL), mma(NULL), x(NULL), subproblem(NULL), trobj(NULL) {
    prob->incref();
    options->incref();
  }
  // setTrustRegionSubproblem (src/ParOptOptimizer.h:42, .cpp:226-237): algorithm = "tr" then drives the caller's
  // subproblem (e.g. a ParOptEigenSubproblem) instead of building a ParOptQuadraticSubproblem from the options
  void setTrustRegionSubproblem(ParOptTrustRegionSubproblem *_subproblem) {
    if (_subproblem) _subproblem->incref();
    if (trobj) trobj->decref();
    trobj = NULL;
    if (subproblem) {
      if (ip) ip->decref();
      ip = NULL;
      subproblem->decref();
    }
    subproblem = _subproblem;
  }
  ~ParOptOptimizer() {
    if (x) x->decref();
    if (trobj) trobj->decref();
    if (ip) ip->decref();
    if (subproblem) subproblem->decref();
    if (tr) po_tr_destroy(tr);
    if (mma) po_mma_destroy(mma);
    options->decref();
    prob->decref();
  }
  void optimize() {
    const char *alg = options->getEnumOption("algorithm");
    const std::string algorithm = alg ? alg : "tr";
    if (algorithm == "ip") {
      if (!ip) {
        ip = new ParOptInteriorPoint(prob, options);
        ip->incref();
      }
      ip->optimize(options->getStringOption("ip_checkpoint_file"));
    } else if (algorithm == "tr" && subproblem) {  // :158-183 with the caller's subproblem
      if (!ip) {
        ip = new ParOptInteriorPoint(subproblem, options);
        ip->incref();
      }
      if (!trobj) {
        trobj = new ParOptTrustRegion(subproblem, options);
        trobj->incref();
      }
      trobj->optimize(ip);
    } else if (algorithm == "tr") {
      if (!tr && po_tr_create(prob->handle(), &tr) != 0) {
        fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
        return;
      }
      po_tr h = tr;
      if (options->forward(
              1, [h](const char *n, const char *v) { return po_tr_set_option_str(h, n, v); },
              [h](const char *n, int v) { return po_tr_set_option_int(h, n, v); },
              [h](const char *n, double v) { return po_tr_set_option_float(h, n, v); }) != 0) {
        fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
        return;
      }
      if (po_tr_optimize(tr) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    } else if (algorithm == "mma") {
      if (!mma && po_mma_create(prob->handle(), &mma) != 0) {
        fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
        return;
      }
      po_mma h = mma;
      if (options->forward(
              2, [h](const char *n, const char *v) { return po_mma_set_option_str(h, n, v); },
              [h](const char *n, int v) { return po_mma_set_option_int(h, n, v); },
              [h](const char *n, double v) { return po_mma_set_option_float(h, n, v); }) != 0) {
        fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
        return;
      }
      if (po_mma_optimize(mma) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    } else {
      fprintf(stderr, "ParOptOptimizer Error: Unrecognized algorithm option %s\n", algorithm.c_str());
    }
  }
  void getOptimizedPoint(ParOptVec **_x, ParOptScalar **_z, ParOptVec **_zw, ParOptVec **_zl,
                         ParOptVec **_zu) {
    if (trobj && ip) {  // :209-213
      trobj->getOptimizedPoint(_x);
      ip->getOptimizedPoint(NULL, _z, _zw, _zl, _zu);
    } else if (tr || mma) {
      po_vec hx = NULL;
      const double *z = NULL;
      if (tr) {
        po_tr_get_optimized_point(tr, &hx, &z, NULL);
      } else {
        po_mma_get_optimized_point(mma, &hx, &z, NULL, NULL, NULL);
      }
      if (x) x->decref();
      x = new ParOptBasicVec(hx);
      x->incref();
      if (_x) *_x = x;
      if (_z) *_z = const_cast<double *>(z);
      if (_zw) *_zw = NULL;
      if (_zl) *_zl = NULL;
      if (_zu) *_zu = NULL;
    } else if (ip) {
      ip->getOptimizedPoint(_x, _z, _zw, _zl, _zu);
    }
  }
  const char *getTrustRegionHistory() {
    const char *t = "";
    if (trobj) return trobj->getHistory();
    if (tr) po_tr_get_history(tr, &t);
    return t;
  }

 private:
  ParOptProblem *prob;
  ParOptOptions *options;
  ParOptInteriorPoint *ip;
  po_tr tr;
  po_mma mma;
  ParOptVec *x;
  ParOptTrustRegionSubproblem *subproblem;
  ParOptTrustRegion *trobj;
};

// ---- ParOptProblem::checkGradients -------------------------------------------------------------------
inline double ParOptProblem::checkGradients(double dh, ParOptVec *xvec, int check_hvec_product) {
  int rank = 0;
  po_ctx_rank(ctx, &rank, NULL);
  ParOptVec *x = xvec ? xvec : createDesignVec();
  x->incref();
  ParOptVec *px = createDesignVec(), *g = createDesignVec(), *xt = createDesignVec(), *gt = createDesignVec();
  px->incref();
  g->incref();
  xt->incref();
  gt->incref();
  std::vector<ParOptVec *> Ac(ncon > 0 ? ncon : 1, (ParOptVec *)NULL), At(ncon > 0 ? ncon : 1, (ParOptVec *)NULL);
  for (int i = 0; i < ncon; i++) {
    Ac[i] = createDesignVec();
    Ac[i]->incref();
    At[i] = createDesignVec();
    At[i]->incref();
  }
  if (!xvec) getVarsAndBounds(x, g, px);  // the bounds land in g and px and are discarded
  ParOptScalar fobj = 0.0, ft = 0.0;
  std::vector<ParOptScalar> c(ncon > 0 ? ncon : 1, 0.0), ct(ncon > 0 ? ncon : 1, 0.0), Apx(ncon > 0 ? ncon : 1, 0.0);
  evalObjCon(x, &fobj, c.data());
  evalObjConGradient(x, g, Ac.data());
  ParOptScalar *pxv, *gv;
  px->getArray(&pxv);
  g->getArray(&gv);
  for (int i = 0; i < nvars; i++) pxv[i] = gv[i] >= 0.0 ? 1.0 : -1.0;
  const ParOptScalar pobj = g->dot(px);
  if (ncon > 0) px->mdot(Ac.data(), ncon, Apx.data());
  // forward difference along px
  xt->copyValues(x);
  xt->axpy(dh, px);
  evalObjCon(xt, &ft, ct.data());
  double worst = 0.0;
  auto report = [&](const char *what, int idx, double actual, double fd) {
    const double err = fabs(actual - fd), rel = err / (fabs(actual) > 1e-300 ? fabs(actual) : 1.0);
    if (rel > worst) worst = rel;
    if (rank == 0) {
      if (idx < 0) {
        printf("%s\n%15s %15s %15s %15s\n%15.6e %15.6e %15.4e %15.4e\n", what, "Actual", "FD", "Err", "Rel err", actual,
               fd, err, rel);
      } else {
        printf("%s[%d]\n%15.6e %15.6e %15.4e %15.4e\n", what, idx, actual, fd, err, rel);
      }
    }
  };
  report("Objective gradient test", -1, pobj, (ft - fobj) / dh);
  for (int i = 0; i < ncon; i++) report("Constraint gradient test", i, Apx[i], (ct[i] - c[i]) / dh);
  ParOptVec *zw = NULL;
  if (nwcon > 0) {
    zw = createConstraintVec();
    zw->incref();
    ParOptScalar *zwv;
    zw->getArray(&zwv);
    for (int i = 0; i < nwcon; i++) zwv[i] = 1.05 + 0.25 * (i % 21);
    // sparse Jacobian: Aw px against differences of evalSparseCon, and (Aw px).zw against px.(Aw^T zw)
    ParOptVec *cw = createConstraintVec(), *cwt = createConstraintVec(), *jp = createConstraintVec();
    cw->incref();
    cwt->incref();
    jp->incref();
    evalSparseCon(x, cw);
    evalSparseCon(xt, cwt);
    jp->zeroEntries();
    addSparseJacobian(1.0, x, px, jp);
    cwt->axpy(-1.0, cw);
    cwt->scale(1.0 / dh);
    report("Sparse Jacobian-vector product test (zw^T Aw px)", -1, jp->dot(zw), cwt->dot(zw));
    gt->zeroEntries();
    addSparseJacobianTranspose(1.0, x, zw, gt);
    report("Sparse Jacobian transpose test (px^T Aw^T zw vs zw^T Aw px)", -1, gt->dot(px), jp->dot(zw));
    cw->decref();
    cwt->decref();
    jp->decref();
  }
  if (check_hvec_product) {
    // gradient of the Lagrangian g - sum z_i Ac_i - Aw^T zw at x and at x + dh px, against H px
    std::vector<ParOptScalar> z(ncon > 0 ? ncon : 1, 0.0);
    for (int i = 0; i < ncon; i++) z[i] = 2.3 - 0.15 * (i % 5);
    ParOptVec *hvec = createDesignVec();
    hvec->incref();
    auto lagr = [&](ParOptVec *xp, ParOptVec *gp, std::vector<ParOptVec *> &Ap) {
      evalObjConGradient(xp, gp, Ap.data());
      for (int i = 0; i < ncon; i++) gp->axpy(-z[i], Ap[i]);
      if (nwcon > 0) addSparseJacobianTranspose(-1.0, xp, zw, gp);
    };
    lagr(x, g, Ac);
    lagr(xt, gt, At);
    const int fail = evalHvecProduct(x, z.data(), zw, px, hvec);
    if (fail == 0) {
      gt->axpy(-1.0, g);
      gt->scale(1.0 / dh);
      report("Hessian-vector product test (px^T H px)", -1, hvec->dot(px), gt->dot(px));
      gt->axpy(-1.0, hvec);
      const double nrm = hvec->norm(), en = gt->norm();
      if (rank == 0) printf("|H px - FD|: %15.4e   |H px|: %15.4e\n", en, nrm);
      if (nrm > 0.0 && en / nrm > worst) worst = en / nrm;
    } else if (rank == 0) {
      printf("Hessian-vector products are not implemented by this problem\n");
    }
    hvec->decref();
  }
  if (zw) zw->decref();
  for (int i = 0; i < ncon; i++) {
    Ac[i]->decref();
    At[i]->decref();
  }
  px->decref();
  g->decref();
  xt->decref();
  gt->decref();
  x->decref();
  return worst;
}

#endif  // PAROPT_AMD_HPP
