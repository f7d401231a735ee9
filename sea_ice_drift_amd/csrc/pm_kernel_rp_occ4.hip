// pm_kernel_rp_occ4.hip - the row-pair kernel compiled for four wavefronts per SIMD (128 VGPRs): the launches whose LDS
// footprint fits four workgroups per CU (slot groups, i.e. at most 7 angles, at borders 20 and 21).  Same source, a
// translation unit of its own - see SID_OCC4_TU at the top of pm_kernel_mfma.hip.
#define SID_OCC4_TU 1
#include "pm_kernel_mfma.hip"
