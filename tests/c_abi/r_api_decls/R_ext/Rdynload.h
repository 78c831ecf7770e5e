/* compile-check declarations only: see ../Rinternals.h */
#ifndef GPV_TEST_RDYNLOAD_DECLS
#define GPV_TEST_RDYNLOAD_DECLS
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *, const void *, const R_CallMethodDef *, const void *, const void *);
int R_useDynamicSymbols(DllInfo *, int);
#endif
