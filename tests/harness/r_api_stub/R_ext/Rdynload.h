/* declarations-only stub, see ../README.md */
#ifndef BSSM_R_STUB_RDYNLOAD_H
#define BSSM_R_STUB_RDYNLOAD_H
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct { const char *name; DL_FUNC fun; int numArgs; void *types; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef struct _DllInfo DllInfo;
typedef int Rboolean_stub;
int R_registerRoutines(DllInfo *info, const R_CMethodDef *const croutines, const R_CallMethodDef *const callRoutines,
                       const R_FortranMethodDef *const fortranRoutines, const R_ExternalMethodDef *const externalRoutines);
int R_useDynamicSymbols(DllInfo *info, int value);
#define FALSE 0
#define TRUE 1
#endif
