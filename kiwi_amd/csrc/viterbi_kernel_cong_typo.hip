// Typo correction with a CoNgram model (the reference's default model type + its --typo configurations): the search kernel source compiled
// with BOTH additions switched on -- CoNgram scoring (KAMD_CONG) and node typo costs (KAMD_TYPO) -- into namespace kamd::typok::congk.  A fifth
// translation unit, for the same reason as the others: the measured kernels stay, instruction for instruction, what they were.
#define KAMD_TYPO 1
#define KAMD_CONG 1
#include "viterbi_kernel.hip"
