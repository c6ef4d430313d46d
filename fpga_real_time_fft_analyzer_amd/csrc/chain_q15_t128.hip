// The integer cascade with 128-sample tiles (sa_launch_filter_q15_t128): chain_q15.hip compiled a second time.  Used
// when launches of a handle overlap (sa_set_overlap > 1): half the LDS per workgroup, so two cascades and an FFT workgroup
// share a CU.  Same arithmetic, same results; the FFT kernels live in the first translation unit only.
#undef SA_STAMPS
#define SA_Q15_SECOND_TU 1
#define SA_Q15_TILE 128
#include "chain_q15.hip"
