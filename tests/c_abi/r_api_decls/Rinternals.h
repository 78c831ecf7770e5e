/* tests/c_abi/r_api_decls/Rinternals.h -- COMPILE-CHECK DECLARATIONS ONLY.
 * R is not installed in this image.  bindings/R/src/gpvR_plan.c (our own file, written against R's public API) is
 * syntax- and prototype-checked with `gcc -fsyntax-only` against these declarations of the few R API functions it uses,
 * with the signatures documented in "Writing R Extensions" (5.9 Handling R objects in C, 5.13 External pointers).
 * Nothing is linked or run against this file; it is not an R implementation and not used for any reference build. */
#ifndef GPV_TEST_RINTERNALS_DECLS
#define GPV_TEST_RINTERNALS_DECLS
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int Rboolean;
#define TRUE 1
#define FALSE 0
#define REALSXP 14
#define STRSXP 16
#define EXTPTRSXP 22
extern SEXP R_NilValue;
extern double R_NaReal;
#define NA_REAL R_NaReal
int TYPEOF(SEXP);
double *REAL(SEXP);
int *INTEGER(SEXP);
int *LOGICAL(SEXP);
int LENGTH(SEXP);
R_xlen_t XLENGTH(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
const char *CHAR(SEXP);
SEXP Rf_install(const char *);
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_allocMatrix(unsigned int, int, int);
SEXP Rf_coerceVector(SEXP, unsigned int);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarReal(double);
SEXP Rf_mkString(const char *);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
Rboolean Rf_isReal(SEXP);
Rboolean Rf_isInteger(SEXP);
Rboolean Rf_isLogical(SEXP);
Rboolean Rf_isMatrix(SEXP);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
void Rf_error(const char *, ...) __attribute__((noreturn));
SEXP R_MakeExternalPtr(void *, SEXP, SEXP);
void *R_ExternalPtrAddr(SEXP);
SEXP R_ExternalPtrTag(SEXP);
void R_ClearExternalPtr(SEXP);
typedef void (*R_CFinalizer_t)(SEXP);
void R_RegisterCFinalizerEx(SEXP, R_CFinalizer_t, Rboolean);
#endif
