/* declarations-only stub, see ../README.md */
#ifndef BSSM_R_STUB_RANDOM_H
#define BSSM_R_STUB_RANDOM_H
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);
#endif
