// The best-path search over a lattice whose nodes carry typo costs (typo correction): the search kernel source
// compiled with the typo-cost additions switched on, into namespace kamd::typok (viterbi_kernel.hpp).  A separate translation unit for the
// same reason as viterbi_kernel_sbg.hip: the measured Knlm kernels stay, instruction for instruction, what they were.
#define KAMD_TYPO 1
#include "viterbi_kernel.hip"
