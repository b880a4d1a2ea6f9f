// Host build of bayesssm_amd/csrc/rng.h for known-answer tests (TEST INFRASTRUCTURE).
#include "../../bayesssm_amd/csrc/rng.h"
extern "C" void h_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out)
{
    bssm::u32x4 c{c0, c1, c2, c3};
    bssm::u32x4 r = bssm::philox4x32_10(c, k0, k1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
extern "C" double h_qnorm(double p) { return bssm::qnorm_as241(p); }
