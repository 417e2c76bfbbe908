// The direct kernel once more, with relaxed float arithmetic: what RT_FLAG_RELAXED selects (include/rt_amd.h).
//
// The exact build (rt_kernels.hip, -ffp-contract=off, correctly rounded sqrt / division) reproduces the reference bit for
// bit.  The north star's bar is lower: 99.9 % of the pixels within 1 LSB and 50 dB.  This translation unit spends that
// slack on the pixels that can afford it -- the direct kernel's, 98 % of the default frame: they draw no random numbers,
// so a comparison that falls the other way (a discriminant a few ulps from zero at a silhouette or a shadow edge) moves
// one of the pixel's four sub-pixel rays and nothing after it.  Bounce pixels stay with the exact kernels: there a
// flipped branch shifts the pixel's place in its rand() sequence for every later sample.
//   - fused multiply-adds wherever the compiler finds them (-ffp-contract=fast on this file's command line);
//   - v_sqrt_f32 / v_rcp_f32 / v_rsq_f32 (1 ulp) instead of the correctly rounded sequences (rt_device.hpp RT_RELAXED);
//   - divisions at 2.5 ulp (-fno-hip-fp32-correctly-rounded-divide-sqrt); min / max as single instructions.
// tests/test_gpu_relaxed.py measures the outcome against the oracle and asserts the north star's tolerance.
#define RT_RELAXED_TU 1
#define RT_RELAXED 1
#define RT_NS rt_relaxed
#include "rt_kernels.hip"
