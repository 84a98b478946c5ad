// The best-path search for SkipBigram models (Knlm + skip-bigram mixture, reference src/SkipBigramModel.hpp:40-201): the search
// kernel source compiled with the history-ring additions switched on, into namespace kamd::sbgk (viterbi_kernel.hpp).
// A separate translation unit on purpose: the Knlm kernels of viterbi_kernel.hip stay, instruction for instruction, the code
// that was measured -- this kernel is sensitive enough to register allocation that a shared template would not guarantee that.
#define KAMD_SBG 1
#include "viterbi_kernel.hip"
