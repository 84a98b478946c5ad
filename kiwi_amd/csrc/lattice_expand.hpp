// Pieces of the candidate / position-program expansion shared by k_expand_cands / k_expand_pos (lattice_kernels.hip: one launch each, any lattice
// kernel's output) and the tail of k_lattice_wave (lattice_wave.hip: the same records written straight from the lattice it holds in LDS).
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "feature.hpp"

namespace kamd
{
	__device__ __forceinline__ float leftBoundaryScore(uint32_t t)      // TagSequenceScorer (src/TagUtils.cpp:49-62) incl. the PA spill (include/kiwi/TagUtils.h:10-18)
	{
		if (t == 2 * T_MAX) return 5.f;
		if (t < T_MAX) return (t == T_NNP || t == T_NP || t == T_IC) ? -1.f : (t == T_SB ? -3.f : 0.f);
		const uint8_t r = (uint8_t)(t - T_MAX);
		return (isEClass(r) || isJClass(r) || isSuffixTag(r) || r == T_VCP) ? -1.f : 0.f;
	}
	// 0 = not evaluated, 1 = a regular candidate, 2 = z-coda / z-siot shortcut (PathEvaluator.hpp:385-446)
	__device__ __forceinline__ uint32_t posCandKind(const SearchParams& P, uint32_t flags, uint8_t tag, bool spaceBefore)
	{
		if (P.splitComplex && (flags & MF_HAS_COMPLEX)) return 0;
		if (tag == T_Z_CODA || tag == T_Z_SIOT) return (tag == T_Z_SIOT && !(P.splitSaisiot || P.mergeSaisiot)) ? 0u : 2u;
		if (!(flags & MF_SINGLE) && (flags & MF_HA_CONTRACTION) && spaceBefore) return 0;
		return 1;
	}

	// one candidate of a form as the static record the search kernels read (k_expand_cands): MorphRec dwords 0..7, then {morpheme id, first LM id,
	// sentence-break type, LM id of the second chunk}
	__device__ __forceinline__ CandStatic candStaticOf(const ModelView& M, uint32_t mid)
	{
		const uint4* mr = reinterpret_cast<const uint4*>(M.morphs + mid);
		CandStatic o; const uint4 r0 = mr[0], r1 = mr[1];
		o.m0 = Quad{ r0.x, r0.y, r0.z, r0.w }; o.m1 = Quad{ r1.x, r1.y, r1.z, r1.w };
		const uint32_t flags = o.m1.y & 0xFFFF; const uint8_t tag = (uint8_t)o.m1.z;
		const uint32_t firstWid = (flags & MF_SINGLE) ? o.m0.x : M.chunkLm[o.m0.z];
		const uint32_t sbType = tag == T_SB ? M.sbInfo[mid] : 0;
		// 4th word: LM id of the second chunk of a chunked candidate (saves the search a dependent chunk-table load)
		const uint32_t secondWid = (!(flags & MF_SINGLE) && (o.m1.w & 0xFF) >= 2) ? M.chunkLm[o.m0.z + 1] : 0;
		o.x = Quad{ mid, firstWid, sbType, secondWid };
		return o;
	}
	// class of a candidate in the transposed evaluator's order (CoNgram models; src/PathEvaluator.hpp:884-915 + src/CoNgramModel.cpp:86-135)
	__device__ __forceinline__ uint32_t candClassOf(uint32_t tag, uint32_t socket, uint32_t flags)
	{
		if (tag == T_Z_CODA) return 0;
		if (tag == T_Z_SIOT) return 1;
		if (!socket) return 2;
		return (flags & MF_SINGLE) ? 3 : 4;
	}
}
