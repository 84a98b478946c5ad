// The best-path search for CoNgram models (local, quantised): viterbi_kernel.hip compiled with KAMD_CONG into namespace kamd::congk -- the
// Knlm, SkipBigram and typo translation units stay token for token what they were.  See the KAMD_CONG sections of viterbi_kernel.hip.
#define KAMD_CONG 1
#include "viterbi_kernel.hip"
