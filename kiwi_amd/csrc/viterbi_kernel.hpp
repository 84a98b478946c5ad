// Launch-side declarations of the best-path search kernel (viterbi_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace kamd
{
#ifdef KAMD_TEST_SMALL_CAPS
	// test build (make smallcaps): tiny LDS staging capacities so that small lattices drive every fallback path
	constexpr uint32_t QCAP = 4;
#else
	constexpr uint32_t QCAP = 32;          // work items of one batch staged in LDS (per lane group)
#endif
	constexpr uint32_t BIGQ = 2048;        // work items of one batch staged in HBM scratch (per lane group); beyond: CS_ERR_PAIR_OVERFLOW

	struct EndCand { float score, fcs, typo; uint32_t parent; uint8_t rootId, sp; uint16_t pad; };
	// scratch in HBM per lane group (items of oversized batches, end-node candidates)
	// (live: the node's incoming paths that are not pruned, in path order -- the items of a node with many dead paths are formed over this list, evaluateNode)
	template<uint32_t Q> struct GroupScratchT { uint64_t key[Q]; float score[Q]; float fcs[Q]; uint32_t live[Q]; };
	using GroupScratch = GroupScratchT<BIGQ>;
	template<uint32_t Q> struct GroupScratchCong { uint64_t key[Q]; float score[Q]; float fcs[Q]; uint32_t live[Q]; uint32_t ctx[Q]; };   // CoNgram search: + the context id of every item

	// SkipBigram models: LM state of every work item of a batch beyond the Knlm node -- history ring, ring position, and a
	// 32-bit digest that is compared before the rings are
	// are.  With rings in the container keys far fewer paths coincide, so nodes collect thousands of incoming paths where the
	// Knlm search sees tens (7271 live ones, 17682 slots counting pruned paths, on the small synthetic model): this kernel stages up
	// to BIGQ_SBG items of one batch (1.8 MB of HBM scratch per lane group; compacting pruned paths out of the item list is future work).
	constexpr uint32_t BIGQ_SBG = 32768;
	// Top-1 de-duplication of a staged batch by hashing instead of scanning (thousands of items per candidate): one slot per
	// container key.  All-zero = free, which is how the engine hands the table over and how every batch leaves it.
	struct SbgSlot { uint32_t owner;      // item that claimed the slot, + 1
		uint32_t firstInv;                 // 0xFFFFFFFF - (earliest item of the key)            (atomicMax)
		unsigned long long best; };        // orderable(score) << 32 | 0xFFFFFFFF - item: the key's winner, first on ties (atomicMax)
	// top-N: the same table groups the items of a container key -- firstInv is then the head (item + 1) of the key's list, `next` its links --, so that an item
	// counts the better items of ITS key by walking that list instead of scanning every item of its candidate (a node with 3 000 incoming paths: 9 M
	// comparisons per candidate with the scan; BASELINE config 3's slowest sentences are those nodes).
	struct SbgScratch { uint32_t hist[BIGQ_SBG][8]; uint32_t pos[BIGQ_SBG]; uint32_t hash[BIGQ_SBG]; uint32_t slot[BIGQ_SBG]; uint32_t next[BIGQ_SBG]; SbgSlot table[2 * BIGQ_SBG]; };

	uint32_t searchKernelLdsBytes(int G);
	constexpr uint32_t kPosKernelLdsBytes = 12800;      // dynamic LDS of k_pos_path<16, .> (four lane groups: ring + staged new states)
	constexpr uint32_t kTypoRingExtra = 384;            // per lane group: the typo compilations without history states (typok::, typok::congk::) keep the state ranges of 64 instead of 32 nodes in LDS
	constexpr uint32_t kPosKernelLdsBytes8 = 13312;     // ... of k_pos_path<8, .> (eight lane groups of half the size each; 26 x 512 bytes: twelve one-wave blocks per CU as before)

	// G = lanes per chunk (4, 8, 16, 32 or 64): a 64-lane wavefront searches 64/G chunks concurrently.
	// WPS = waves per SIMD the instantiation is compiled for (2, or 3 for G = 8 / 16).
	template<int G, int WPS>
	__global__ void k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork);
	// The same search for a SkipBigram model (Knlm + skip-bigram mixture, reference src/SkipBigramModel.hpp): the kernel source
	// compiled a second time with KAMD_SBG defined (viterbi_kernel_sbg.hip), so that the Knlm kernels above stay exactly the
	// code that was measured.  G = 16 or 64, WPS = 2.
	namespace sbgk
	{
		uint32_t histKernelLdsBytes(int G);      // dynamic LDS of this compilation's k_best_path<G, .> (G = 16 or 64)
		template<int G, int WPS>
		__global__ void k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork, SbgDev S);
	}
	// ... and for lattices built over typo graphs (viterbi_kernel_typo.hip, KAMD_TYPO): nodeTypo = the typo cost of every node of the batch,
	// parallel to WorkView::nodes.  G = 16 or 64, WPS = 2; Knlm scoring.
	namespace typok
	{
		template<int G, int WPS>
		__global__ void k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork, const float* nodeTypo);
	}
	// ... and for CoNgram models (viterbi_kernel_cong.hip, KAMD_CONG): the context trie sits where the Knlm trie does in ModelView (edge hash, root
	// table, suffix links), CG names the embedding tables.  G = 16 or 64, WPS = 2.
	namespace congk
	{
		template<int G, int WPS>
		__global__ void k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork, CongDev CG);
	}
	// ... and for typo correction with a CoNgram model (viterbi_kernel_cong_typo.hip, KAMD_TYPO + KAMD_CONG)
	namespace typok { namespace congk
	{
		template<int G, int WPS>
		__global__ void k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork, const float* nodeTypo, CongDev CG);
	} }
	// ... and for the GLOBAL CoNgram model (viterbi_kernel_congg.hip / _congg_typo.hip: KAMD_CONG + KAMD_CONGG [+ KAMD_TYPO]): GG names the window sections and the
	// history storage of the search (device_types.hpp CongGDev).  G = 16 or 64, WPS = 2; the general search only
	namespace congk { namespace gk
	{
		template<int G, int WPS>
		__global__ void k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork, CongDev CG, CongGDev GG);
		uint32_t histKernelLdsBytes(int G);
	} }
	namespace typok { namespace congk { namespace gk
	{
		template<int G, int WPS>
		__global__ void k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork, const float* nodeTypo, CongDev CG, CongGDev GG);
		uint32_t histKernelLdsBytes(int G);
	} } }
	// ... and for typo correction with a SkipBigram model (viterbi_kernel_sbg_typo.hip, KAMD_TYPO + KAMD_SBG)
	namespace typok { namespace sbgk
	{
		template<int G, int WPS>
		__global__ void k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork, SbgDev S, const float* nodeTypo);
		uint32_t histKernelLdsBytes(int G);
	} }
	// The position-step search (viterbi_pos.inc): the common case of the search above, one END POSITION of the lattice per step over the position
	// program written by k_expand_pos; it leaves in DevChunkResult::pad the node k_best_path -- launched over all chunks afterwards -- carries on at
	// (the end node: only the end stage is left).  G = 16 (one DPP row per chunk); every compilation of the search except the SkipBigram ones has it.
	template<int G, int WPS>
	__global__ void k_pos_path(ModelView M, BatchView B, WorkView W, SearchParams P, const uint32_t* chunkOrder, uint32_t nWork);
	namespace typok { template<int G, int WPS> __global__ void k_pos_path(ModelView M, BatchView B, WorkView W, SearchParams P, const uint32_t* chunkOrder, uint32_t nWork, const float* nodeTypo); }
	namespace congk { template<int G, int WPS> __global__ void k_pos_path(ModelView M, BatchView B, WorkView W, SearchParams P, const uint32_t* chunkOrder, uint32_t nWork, CongDev CG); }
	namespace typok { namespace congk { template<int G, int WPS> __global__ void k_pos_path(ModelView M, BatchView B, WorkView W, SearchParams P, const uint32_t* chunkOrder, uint32_t nWork, const float* nodeTypo, CongDev CG); } }
	// End stage, one THREAD per chunk: restated std::sort of the end candidates, per-(root, state) selection and the
	// back-trace into 24-byte tokens.  A separate launch so that 64 chunks share a wavefront in this strictly serial stage.
	__global__ void k_finish_paths(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t chunkBegin, uint32_t chunkCount, uint32_t stride);
}
