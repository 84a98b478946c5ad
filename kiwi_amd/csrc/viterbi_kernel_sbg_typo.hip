// Typo correction with a SkipBigram model (the reference's typo transformers work with every model type): the search kernel source compiled with
// BOTH additions switched on -- history rings (KAMD_SBG) and node typo costs (KAMD_TYPO) -- into namespace kamd::typok::sbgk.  A sixth
// translation unit, for the same reason as the others: the measured kernels stay, instruction for instruction, what they were.
#define KAMD_TYPO 1
#define KAMD_SBG 1
#include "viterbi_kernel.hip"
