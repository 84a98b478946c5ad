// Typo correction with the global CoNgram model: KAMD_TYPO + KAMD_CONG + KAMD_CONGG, namespace kamd::typok::congk::gk (see viterbi_kernel_congg.hip).
#define KAMD_TYPO 1
#define KAMD_CONG 1
#define KAMD_CONGG 1
#include "viterbi_kernel.hip"
