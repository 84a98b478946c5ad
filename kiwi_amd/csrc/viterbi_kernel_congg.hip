// The best-path search for the GLOBAL CoNgram model (ModelType::congGlobal, window 7: a valid distant token is scored as a mixture over the context and the
// last seven such tokens of the path; reference src/CoNgramModel.cpp:802-868, 1037-1490): viterbi_kernel.hip compiled with KAMD_CONG + KAMD_CONGG into
// namespace kamd::congk::gk -- CoNgram scoring plus the history plumbing the SkipBigram compilation has (KAMD_HIST).  A translation unit of its own, like the
// others: the measured kernels stay, instruction for instruction, what they were.  The general search only (no position-step kernel for this model type).
#define KAMD_CONG 1
#define KAMD_CONGG 1
#include "viterbi_kernel.hip"
