/* declarations-only stub of the R API subset used by r/bayesssm_amd_glue.c, see README.md */
#ifndef BSSM_R_STUB_RINTERNALS_H
#define BSSM_R_STUB_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
#define INTSXP 13
#define REALSXP 14
#define VECSXP 19
extern double R_NaN;
extern double R_NaReal;
#define NA_REAL R_NaReal
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_coerceVector(SEXP, SEXPTYPE);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_isNull(SEXP);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarLogical(int);
SEXP Rf_ScalarReal(double);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
int LENGTH(SEXP);
R_xlen_t XLENGTH(SEXP);
double *REAL(SEXP);
int *INTEGER(SEXP);
void Rf_error(const char *, ...) __attribute__((noreturn, format(printf, 1, 2)));
#endif
