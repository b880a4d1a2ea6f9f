/* declarations-only stub, see README.md */
#ifndef BSSM_R_STUB_R_H
#define BSSM_R_STUB_R_H
#include <stddef.h>
char *R_alloc(size_t n, int size);
#endif
