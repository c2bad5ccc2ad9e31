/* shim/mockr/mockr.c -- implementation of the mock R runtime declared in Rinternals.h of this directory.
 * A TEST DOUBLE: it exists so that shim/bssm_amd_shim.c can run (and its PROTECT discipline, list handling, dims and error
 * paths can be tested) where R is absent.  Not R, not part of the product. */
#define _POSIX_C_SOURCE 200809L
#include "R.h"
#include "R_ext/Rdynload.h"

#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct attr { SEXP tag, val; struct attr* next; };
struct SEXPREC {
  int type;
  R_xlen_t len;
  void* data;             /* double* / int* / SEXP* / char* */
  struct attr* attrs;
  int collected;          /* poisoned by the gctorture emulation */
  int permanent;          /* symbols, nil, objects made by the test harness (the caller's arguments) */
  int mark;
  mockr_closure_fn fn;    /* CLOSXP */
  void* user;
  long serial;
};

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, NULL, 0, 1, 0, NULL, NULL, 0};
static struct SEXPREC env_rec = {NILSXP, 0, NULL, NULL, 0, 1, 0, NULL, NULL, 0};
SEXP R_NilValue = &nil_rec, R_GlobalEnv = &env_rec, R_NamesSymbol = NULL, R_DimSymbol = NULL;
double R_NaReal, R_NegInf, R_PosInf;
int R_NaInt = INT32_MIN;

/* ---- object table ---- */
static SEXP* g_objs = NULL;
static long g_nobjs = 0, g_cap = 0, g_serial = 0;
static void** g_blocks = NULL;   /* R_alloc */
static long g_nblocks = 0, g_bcap = 0;
static SEXP g_stack[10000];
static int g_depth = 0, g_depth_after = 0;
static int g_torture = 1, g_in_call = 0, g_touch = 0, g_building = 0;
static SEXP* g_args = NULL;
static int g_nargs = 0;
static jmp_buf g_jmp;
static char g_err[2048], g_warn[2048];
static SEXP g_symbols[256];
static int g_nsym = 0;

static void fail(const char* fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
static void fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  if (g_in_call) longjmp(g_jmp, 1);
  fprintf(stderr, "mockr: %s (outside mockr_call)\n", g_err);
  abort();
}

static void init_once(void) {
  static int done = 0;
  if (done) return;
  done = 1;
  R_NaReal = NAN; R_NegInf = -INFINITY; R_PosInf = INFINITY;
  R_NamesSymbol = Rf_install("names");
  R_DimSymbol = Rf_install("dim");
}

static size_t elt_size(int type) {
  switch (type) {
    case REALSXP: return sizeof(double);
    case INTSXP: case LGLSXP: return sizeof(int);
    case STRSXP: case VECSXP: case LANGSXP: return sizeof(SEXP);
    case CHARSXP: return 1;
    default: return 0;
  }
}

static void mark(SEXP s) {
  if (!s || s->mark) return;
  s->mark = 1;
  for (struct attr* a = s->attrs; a; a = a->next) { mark(a->tag); mark(a->val); }
  if (s->type == VECSXP || s->type == STRSXP || s->type == LANGSXP)
    for (R_xlen_t i = 0; i < s->len; i++) mark(((SEXP*)s->data)[i]);
}

/* what a garbage collection at this point would free: poison it */
static void torture(void) {
  if (!g_torture || !g_in_call) return;
  for (long i = 0; i < g_nobjs; i++) g_objs[i]->mark = 0;
  for (int i = 0; i < g_depth; i++) mark(g_stack[i]);
  for (int i = 0; i < g_nargs; i++) mark(g_args[i]);
  for (long i = 0; i < g_nobjs; i++) {
    SEXP s = g_objs[i];
    if (s->permanent) mark(s);
  }
  for (long i = 0; i < g_nobjs; i++) {
    SEXP s = g_objs[i];
    if (s->mark || s->permanent || s->collected) continue;
    s->collected = 1;
    if (s->type == REALSXP) for (R_xlen_t q = 0; q < s->len; q++) ((double*)s->data)[q] = NAN;
    if (s->type == INTSXP || s->type == LGLSXP) for (R_xlen_t q = 0; q < s->len; q++) ((int*)s->data)[q] = INT32_MIN;
  }
}

static SEXP new_obj(int type, R_xlen_t len) {
  init_once();
  torture();
  SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
  s->type = type; s->len = len; s->serial = ++g_serial;
  s->permanent = g_building > 0;   /* built by the test harness: owned by the caller (R: protected by the calling frame) */
  const size_t es = elt_size(type);
  if (es) s->data = calloc((size_t)(len > 0 ? len : 1) + (type == CHARSXP ? 1 : 0), es);
  if (type == VECSXP || type == STRSXP || type == LANGSXP) for (R_xlen_t i = 0; i < len; i++) ((SEXP*)s->data)[i] = R_NilValue;
  if (g_nobjs == g_cap) { g_cap = g_cap ? 2 * g_cap : 1024; g_objs = (SEXP*)realloc(g_objs, (size_t)g_cap * sizeof(SEXP)); }
  g_objs[g_nobjs++] = s;
  return s;
}

static SEXP use(SEXP s) {   /* every accessor goes through here */
  if (!s) fail("NULL SEXP");
  if (s->collected) { g_touch++; fail("use of an object that a garbage collection would have freed (missing PROTECT): type %d, length %ld", s->type, (long)s->len); }
  return s;
}

/* ---- R API ---- */
SEXP Rf_install(const char* name) {
  for (int i = 0; i < g_nsym; i++) if (!strcmp((const char*)g_symbols[i]->data, name)) return g_symbols[i];
  SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
  s->type = SYMSXP; s->permanent = 1; s->data = strdup(name);
  if (g_nsym >= 256) fail("too many symbols");
  g_symbols[g_nsym++] = s;
  return s;
}
SEXP Rf_getAttrib(SEXP x, SEXP tag) {
  use(x);
  for (struct attr* a = x->attrs; a; a = a->next) if (a->tag == tag) return a->val;
  return R_NilValue;
}
SEXP Rf_setAttrib(SEXP x, SEXP tag, SEXP val) {
  use(x); use(val);
  if (x == R_NilValue) fail("attempt to set an attribute on NULL");
  for (struct attr* a = x->attrs; a; a = a->next) if (a->tag == tag) { a->val = val; return val; }
  struct attr* a = (struct attr*)calloc(1, sizeof *a);
  a->tag = tag; a->val = val; a->next = x->attrs; x->attrs = a;
  return val;
}
SEXP Rf_allocVector(unsigned int type, R_xlen_t len) {
  if (len < 0) fail("negative length vectors are not allowed");
  if (!elt_size((int)type) || type == CHARSXP) fail("allocVector: unsupported type %u", type);
  return new_obj((int)type, len);
}
static SEXP with_dim(unsigned int type, int nd, const int* d) {
  R_xlen_t len = 1;
  for (int i = 0; i < nd; i++) { if (d[i] < 0) fail("negative extents to matrix / array"); len *= d[i]; }
  SEXP s = Rf_protect(Rf_allocVector(type, len));
  SEXP dim = Rf_allocVector(INTSXP, nd);
  for (int i = 0; i < nd; i++) ((int*)dim->data)[i] = d[i];
  Rf_setAttrib(s, R_DimSymbol, dim);
  Rf_unprotect(1);
  return s;
}
SEXP Rf_allocMatrix(unsigned int type, int nr, int nc) { int d[2] = {nr, nc}; return with_dim(type, 2, d); }
SEXP Rf_alloc3DArray(unsigned int type, int a, int b, int c) { int d[3] = {a, b, c}; return with_dim(type, 3, d); }
SEXP Rf_allocArray(unsigned int type, SEXP dims) { use(dims); return with_dim(type, (int)dims->len, (const int*)dims->data); }
SEXP Rf_mkChar(const char* str) {
  SEXP s = new_obj(CHARSXP, (R_xlen_t)strlen(str));
  memcpy(s->data, str, strlen(str) + 1);
  return s;
}
SEXP Rf_mkString(const char* str) {
  SEXP v = Rf_protect(Rf_allocVector(STRSXP, 1));
  ((SEXP*)v->data)[0] = Rf_mkChar(str);
  Rf_unprotect(1);
  return v;
}
SEXP Rf_mkNamed(unsigned int type, const char** names) {
  int n = 0;
  while (names[n][0]) n++;
  SEXP v = Rf_protect(Rf_allocVector(type, n));
  SEXP nm = Rf_protect(Rf_allocVector(STRSXP, n));
  for (int i = 0; i < n; i++) ((SEXP*)nm->data)[i] = Rf_mkChar(names[i]);
  Rf_setAttrib(v, R_NamesSymbol, nm);
  Rf_unprotect(2);
  return v;
}
SEXP Rf_ScalarReal(double x) { SEXP s = Rf_allocVector(REALSXP, 1); ((double*)s->data)[0] = x; return s; }
SEXP Rf_ScalarInteger(int x) { SEXP s = Rf_allocVector(INTSXP, 1); ((int*)s->data)[0] = x; return s; }
SEXP Rf_ScalarLogical(int x) { SEXP s = Rf_allocVector(LGLSXP, 1); ((int*)s->data)[0] = x == R_NaInt ? x : (x != 0); return s; }
SEXP Rf_duplicate(SEXP x) {
  use(x);
  if (x == R_NilValue || x->type == SYMSXP || x->type == CLOSXP || x->type == CHARSXP) return x;
  SEXP s = Rf_protect(new_obj(x->type, x->len));
  if (x->type == VECSXP || x->type == STRSXP) {
    for (R_xlen_t i = 0; i < x->len; i++) ((SEXP*)s->data)[i] = Rf_duplicate(((SEXP*)x->data)[i]);
  } else if (elt_size(x->type)) {
    memcpy(s->data, x->data, (size_t)x->len * elt_size(x->type));
  }
  for (struct attr* a = x->attrs; a; a = a->next) Rf_setAttrib(s, a->tag, Rf_duplicate(a->val));
  Rf_unprotect(1);
  return s;
}
SEXP Rf_coerceVector(SEXP x, unsigned int type) {
  use(x);
  if ((unsigned)x->type == type) return x;
  if (x == R_NilValue) return Rf_allocVector(type, 0);
  const int from_num = x->type == REALSXP || x->type == INTSXP || x->type == LGLSXP;
  if (!from_num || !(type == REALSXP || type == INTSXP || type == LGLSXP)) fail("coerceVector: unsupported conversion %d -> %u", x->type, type);
  SEXP s = Rf_protect(Rf_allocVector(type, x->len));
  for (R_xlen_t i = 0; i < x->len; i++) {
    double v; int na = 0;
    if (x->type == REALSXP) { v = ((double*)x->data)[i]; na = isnan(v); }
    else { const int iv = ((int*)x->data)[i]; na = iv == R_NaInt; v = iv; }
    if (type == REALSXP) ((double*)s->data)[i] = na ? R_NaReal : v;
    else if (type == INTSXP) ((int*)s->data)[i] = na ? R_NaInt : (int)v;
    else ((int*)s->data)[i] = na ? R_NaInt : (v != 0.0);
  }
  for (struct attr* a = x->attrs; a; a = a->next) Rf_setAttrib(s, a->tag, a->val);
  Rf_unprotect(1);
  return s;
}
SEXP Rf_lang2(SEXP f, SEXP arg) {
  use(f); use(arg);
  SEXP s = new_obj(LANGSXP, 2);
  ((SEXP*)s->data)[0] = f; ((SEXP*)s->data)[1] = arg;
  return s;
}
SEXP Rf_eval(SEXP call, SEXP env) {
  (void)env;
  use(call);
  if (call->type != LANGSXP) return call;
  SEXP f = use(((SEXP*)call->data)[0]), arg = use(((SEXP*)call->data)[1]);
  if (f->type != CLOSXP || !f->fn) fail("attempt to apply non-function");
  /* the callee allocates like any R code: its result is unprotected when it comes back */
  SEXP r = f->fn(arg, f->user);
  if (!r) {
    char msg[sizeof g_err];
    snprintf(msg, sizeof msg, "%s", g_err[0] ? g_err : "(no message)");   /* (fail() formats into g_err) */
    fail("error in the evaluated closure: %s", msg);
  }
  return r;
}
SEXP Rf_protect(SEXP s) {
  if (g_depth >= 10000) fail("protect(): protection stack overflow");
  g_stack[g_depth++] = s;
  return s;
}
void Rf_unprotect(int n) {
  if (n > g_depth) fail("unprotect(): only %d protected item(s)", g_depth);
  g_depth -= n;
}
int Rf_length(SEXP s) { use(s); return (int)s->len; }
R_xlen_t Rf_xlength(SEXP s) { use(s); return s->len; }
static void need(SEXP s, const char* what) {
  use(s);
  if (s->len < 1) fail("%s: argument of length 0", what);
}
int Rf_asInteger(SEXP s) {
  need(s, "asInteger");
  if (s->type == INTSXP || s->type == LGLSXP) return ((int*)s->data)[0];
  if (s->type == REALSXP) { const double v = ((double*)s->data)[0]; return isnan(v) ? R_NaInt : (int)v; }
  fail("asInteger: not a number (type %d)", s->type);
}
int Rf_asLogical(SEXP s) {
  need(s, "asLogical");
  if (s->type == LGLSXP) return ((int*)s->data)[0];
  if (s->type == INTSXP) { const int v = ((int*)s->data)[0]; return v == R_NaInt ? v : (v != 0); }
  if (s->type == REALSXP) { const double v = ((double*)s->data)[0]; return isnan(v) ? R_NaInt : (v != 0.0); }
  fail("asLogical: not a logical (type %d)", s->type);
}
double Rf_asReal(SEXP s) {
  need(s, "asReal");
  if (s->type == REALSXP) return ((double*)s->data)[0];
  if (s->type == INTSXP || s->type == LGLSXP) { const int v = ((int*)s->data)[0]; return v == R_NaInt ? R_NaReal : (double)v; }
  fail("asReal: not a number (type %d)", s->type);
}
int Rf_isNull(SEXP s) { return use(s) == R_NilValue; }
int Rf_isFunction(SEXP s) { return use(s)->type == CLOSXP; }
int TYPEOF(SEXP s) { return use(s)->type; }
double* REAL(SEXP s) { if (use(s)->type != REALSXP) fail("REAL() can only be applied to a 'numeric', not a type-%d object", s->type); return (double*)s->data; }
int* INTEGER(SEXP s) { if (use(s)->type != INTSXP && s->type != LGLSXP) fail("INTEGER() can only be applied to a 'integer', not a type-%d object", s->type); return (int*)s->data; }
int* LOGICAL(SEXP s) { if (use(s)->type != LGLSXP) fail("LOGICAL() can only be applied to a 'logical', not a type-%d object", s->type); return (int*)s->data; }
SEXP VECTOR_ELT(SEXP s, R_xlen_t i) {
  if (use(s)->type != VECSXP) fail("VECTOR_ELT() can only be applied to a 'list', not a type-%d object", s->type);
  if (i < 0 || i >= s->len) fail("attempt to access index %ld/%ld in VECTOR_ELT", (long)i, (long)s->len);
  return ((SEXP*)s->data)[i];
}
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP v) {
  use(v);
  if (use(s)->type != VECSXP) fail("SET_VECTOR_ELT() can only be applied to a 'list', not a type-%d object", s->type);
  if (i < 0 || i >= s->len) fail("attempt to set index %ld/%ld in SET_VECTOR_ELT", (long)i, (long)s->len);
  ((SEXP*)s->data)[i] = v;
  return v;
}
SEXP STRING_ELT(SEXP s, R_xlen_t i) {
  if (use(s)->type != STRSXP) fail("STRING_ELT() can only be applied to a 'character vector', not a type-%d object", s->type);
  if (i < 0 || i >= s->len) fail("attempt to access index %ld/%ld in STRING_ELT", (long)i, (long)s->len);
  return ((SEXP*)s->data)[i];
}
void SET_STRING_ELT(SEXP s, R_xlen_t i, SEXP v) {
  if (use(s)->type != STRSXP || use(v)->type != CHARSXP) fail("SET_STRING_ELT: wrong types");
  if (i < 0 || i >= s->len) fail("attempt to set index %ld/%ld in SET_STRING_ELT", (long)i, (long)s->len);
  ((SEXP*)s->data)[i] = v;
}
const char* CHAR(SEXP s) { if (use(s)->type != CHARSXP) fail("CHAR() can only be applied to a 'CHARSXP', not a type-%d object", s->type); return (const char*)s->data; }
char* R_alloc(size_t n, int size) {
  void* p = calloc(n ? n : 1, (size_t)(size > 0 ? size : 1));
  if (!p) fail("R_alloc: cannot allocate %zu x %d bytes", n, size);
  if (g_nblocks == g_bcap) { g_bcap = g_bcap ? 2 * g_bcap : 256; g_blocks = (void**)realloc(g_blocks, (size_t)g_bcap * sizeof(void*)); }
  g_blocks[g_nblocks++] = p;
  return (char*)p;
}
void Rf_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  if (g_in_call) longjmp(g_jmp, 1);
  fprintf(stderr, "mockr: Rf_error outside mockr_call: %s\n", g_err);
  abort();
}
void Rf_warning(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_warn, sizeof g_warn, fmt, ap);
  va_end(ap);
}
void R_CheckUserInterrupt(void) {}

/* ---- registration ---- */
static const R_CallMethodDef* g_routines = NULL;
static int g_dynsym = -1;
int R_registerRoutines(DllInfo* dll, const void* c, const R_CallMethodDef* call, const void* f, const void* e) {
  (void)dll; (void)c; (void)f; (void)e;
  g_routines = call;
  return 1;
}
Rboolean R_useDynamicSymbols(DllInfo* dll, Rboolean v) { (void)dll; const int old = g_dynsym; g_dynsym = v; return old; }
Rboolean R_forceSymbols(DllInfo* dll, Rboolean v) { (void)dll; (void)v; return 0; }
void* mockr_lookup(const char* name, int* nargs) {
  for (const R_CallMethodDef* r = g_routines; r && r->name; r++)
    if (!strcmp(r->name, name)) { if (nargs) *nargs = r->numArgs; return (void*)r->fun; }
  return NULL;
}
int mockr_registered_count(void) { int n = 0; for (const R_CallMethodDef* r = g_routines; r && r->name; r++) n++; return n; }
int mockr_dynamic_symbols(void) { return g_dynsym; }

/* ---- harness ---- */
SEXP mockr_closure(mockr_closure_fn fn, void* user) {
  init_once();
  SEXP s = new_obj(CLOSXP, 0);
  s->fn = fn; s->user = user;
  return s;
}
SEXP mockr_extptr(void) { return new_obj(EXTPTRSXP, 0); }
const char* mockr_last_error(void) { return g_err; }
const char* mockr_last_warning(void) { return g_warn; }
int mockr_protect_depth(void) { return g_depth_after; }
int mockr_poisoned_touches(void) { return g_touch; }
long mockr_live_objects(void) { return g_nobjs; }
void mockr_gctorture(int on) { g_torture = on; }

typedef SEXP (*f0)(void);
SEXP mockr_call(void* fn, int nargs, SEXP* args) {
  init_once();
  g_err[0] = 0; g_warn[0] = 0;
  if (nargs < 0 || nargs > 40) { snprintf(g_err, sizeof g_err, "mockr_call: %d arguments", nargs); return NULL; }
  SEXP a[40];
  for (int i = 0; i < 40; i++) a[i] = i < nargs ? args[i] : R_NilValue;
  g_args = a; g_nargs = nargs; g_depth = 0; g_in_call = 1;
  SEXP res = NULL;
  if (setjmp(g_jmp) == 0) {
    /* the C calling conventions in use here pass the first arguments in registers and the rest on the stack in order, so
     * a function of nargs SEXP parameters can be entered through a pointer to one of 40: the callee reads the first nargs */
    typedef SEXP (*f40)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP,
                        SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
    res = ((f40)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18], a[19],
                    a[20], a[21], a[22], a[23], a[24], a[25], a[26], a[27], a[28], a[29], a[30], a[31], a[32], a[33], a[34], a[35], a[36], a[37], a[38], a[39]);
    g_depth_after = g_depth;
    if (g_depth != 0) { snprintf(g_err, sizeof g_err, "stack imbalance in .Call: %d protected item(s) left", g_depth); res = NULL; }
    else if (res && res->collected) { snprintf(g_err, sizeof g_err, "the returned object was not protected when the last allocation happened"); res = NULL; }
  } else {
    g_depth_after = 0;   /* R unwinds the protect stack on error */
    res = NULL;
  }
  /* what the call returned belongs to the caller now */
  if (res) { for (long i = 0; i < g_nobjs; i++) g_objs[i]->mark = 0; mark(res); for (long i = 0; i < g_nobjs; i++) if (g_objs[i]->mark) g_objs[i]->permanent = 1; }
  g_in_call = 0; g_args = NULL; g_nargs = 0; g_depth = 0;
  return res;
}

void mockr_reset(void) {
  for (long i = 0; i < g_nobjs; i++) {
    SEXP s = g_objs[i];
    for (struct attr* a = s->attrs; a;) { struct attr* nx = a->next; free(a); a = nx; }
    free(s->data);
    free(s);
  }
  g_nobjs = 0;
  for (long i = 0; i < g_nblocks; i++) free(g_blocks[i]);
  g_nblocks = 0;
  g_depth = g_depth_after = 0; g_touch = 0; g_err[0] = 0; g_warn[0] = 0;
}

/* objects the harness builds (arguments; a closure's result while it is being assembled) are the caller's until released */
void mockr_building(int on) { g_building += on ? 1 : -1; if (g_building < 0) g_building = 0; }
static void release_rec(SEXP s) {
  if (!s || !s->permanent || s->type == SYMSXP || s == R_NilValue || s == R_GlobalEnv) return;
  s->permanent = 0;
  for (struct attr* a = s->attrs; a; a = a->next) release_rec(a->val);
  if (s->type == VECSXP || s->type == STRSXP || s->type == LANGSXP) for (R_xlen_t i = 0; i < s->len; i++) release_rec(((SEXP*)s->data)[i]);
}
/* a closure's finished result goes back to the callee unprotected, as an R function's value does */
void mockr_release(SEXP s) { release_rec(s); }

/* ---- accessors for the Python side of the harness (ctypes) ---- */
int mockr_type(SEXP s) { return s->type; }
long mockr_len(SEXP s) { return (long)s->len; }
void* mockr_data(SEXP s) { return s->data; }
SEXP mockr_attr(SEXP s, const char* name) { return Rf_getAttrib(s, Rf_install(name)); }
void mockr_set_attr(SEXP s, const char* name, SEXP v) { Rf_setAttrib(s, Rf_install(name), v); }
SEXP mockr_nil(void) { return R_NilValue; }
SEXP mockr_vec(int type, long len) { init_once(); return Rf_allocVector((unsigned)type, len); }
SEXP mockr_char(const char* str) { init_once(); return Rf_mkChar(str); }
void mockr_set_elt(SEXP s, long i, SEXP v) { ((SEXP*)s->data)[i] = v; }
SEXP mockr_get_elt(SEXP s, long i) { return ((SEXP*)s->data)[i]; }
/* raise from inside a closure callback (the Python side catches its exception and reports it here) */
void mockr_closure_failed(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

/* ---- self-tests of the mock itself (tests/test_host_logic.py): entry points with the PROTECT bugs the mock must catch ---- */
SEXP mockr_selftest_ok(SEXP x) {           /* correct: 2 * x */
  SEXP a = PROTECT(Rf_allocVector(REALSXP, Rf_xlength(x)));
  SEXP names = PROTECT(Rf_mkString("twice"));
  for (R_xlen_t i = 0; i < Rf_xlength(x); i++) REAL(a)[i] = 2.0 * REAL(x)[i];
  Rf_setAttrib(a, Rf_install("what"), names);
  UNPROTECT(2);
  return a;
}
SEXP mockr_selftest_unprotected(SEXP x) {  /* `a` is unprotected while `b` is allocated: a collection there frees it */
  SEXP a = Rf_allocVector(REALSXP, 3);
  SEXP b = PROTECT(Rf_allocVector(REALSXP, 3));
  REAL(a)[0] = REAL(x)[0];
  UNPROTECT(1);
  return b;
}
SEXP mockr_selftest_imbalance(SEXP x) {    /* leaves one item on the protect stack */
  (void)x;
  PROTECT(Rf_allocVector(REALSXP, 1));
  return R_NilValue;
}
SEXP mockr_selftest_returns_collected(SEXP x) {   /* the value handed back was unprotected at the last allocation */
  (void)x;
  SEXP a = Rf_allocVector(REALSXP, 2);
  (void)Rf_allocVector(REALSXP, 2);
  return a;
}
SEXP mockr_selftest_error(SEXP x) {        /* Rf_error with items on the protect stack: R unwinds them */
  PROTECT(Rf_allocVector(REALSXP, 1));
  Rf_error("deliberate failure %d", Rf_asInteger(x));
}
