// SkipBigram mixture score of one LM step: SkipBigramModel::evaluate (/root/reference/src/SkipBigramModel.hpp:113-139)
// with the scalar logSumExp of src/MathFunc.hpp:43-56 (ArchType none / balanced) -- the same fp32 operations in the
// same order: 8 discounted Knlm terms + 8 bigram compensations (-inf when the history word is no partner of `next`),
// maximum, sum of exponentials in index order, logarithm, minus log(window).  exp / log are the glibc-exact ones of
// exact_math.hpp, so host and device agree bit for bit with the reference's std::exp / std::log.
// Shared by the search kernel (device) and the host-side check exported as kamd_debug_sbg_evaluate (capi_low.cpp).
#pragma once
#include <cmath>
#include "exact_math.hpp"

namespace kamd
{
	// S: any view with ptrs / keys / comps / discnts / logWindowSize (flat_model.hpp SbgView, device_types.hpp SbgDev).
	// hist: the ring of the last 8 valid word ids in storage order (the summation order is the ring's, not the age's).
	template<class SV>
	KAMD_HD float sbgEvaluate(const SV& S, const uint32_t (&hist)[8], uint32_t next, float ll)
	{
		const uint32_t kb = S.ptrs[next], ke = S.ptrs[next + 1];
		float a[16];
		// std::lower_bound over the partners of `next` for all eight history words TOGETHER: the eight searches run over the same range, so they take the
		// same number of halving steps and every step's eight probes are independent loads -- log2(partners) + 3 memory round trips for the whole mixture
		// instead of eight searches one after the other (the search kernel is bound by exactly this chain: DESIGN.md, SkipBigram).  Invariant of a step:
		// the answer lies in [base, base + len].
		uint32_t base[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) { base[i] = kb; a[i] = S.discnts[hist[i]] + ll; }
		uint32_t len = ke - kb;
		while (len > 1)
		{
			const uint32_t half = len >> 1;
#pragma unroll
			for (int i = 0; i < 8; ++i) base[i] += (S.keys[base[i] + half - 1] < hist[i]) ? half : 0u;
			len -= half;
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) if (len == 1) base[i] += (S.keys[base[i]] < hist[i]) ? 1u : 0u;
		// (key and compensation of the slot a search ended at are requested together, the compensation speculatively: one more round trip, not two)
		if (ke > kb)
		{
			uint32_t fk[8]; float fc[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) { const uint32_t at = base[i] < ke ? base[i] : ke - 1; fk[i] = S.keys[at]; fc[i] = S.comps[at]; }
#pragma unroll
			for (int i = 0; i < 8; ++i) a[8 + i] = (base[i] < ke && fk[i] == hist[i]) ? fc[i] : -INFINITY;
		}
		else
		{
#pragma unroll
			for (int i = 0; i < 8; ++i) a[8 + i] = -INFINITY;
		}
		float mx = a[0];
#pragma unroll
		for (int i = 1; i < 16; ++i) mx = a[i] > mx ? a[i] : mx;
		float sum = 0;
#pragma unroll
		for (int i = 0; i < 16; ++i) sum += exact::expf_glibc(a[i] - mx);
		return (exact::logf_glibc(sum) + mx) - S.logWindowSize;
	}

	// SbgState::nextImpl (src/SkipBigramModel.hpp:169-182) around a Knlm step that already happened: `ll` is the Knlm
	// log-likelihood of `next`; the ring advances for every word the model knows, scored or not.
	template<class SV>
	KAMD_HD float sbgNext(const SV& S, uint32_t (&hist)[8], uint32_t& pos, uint32_t next, float ll)
	{
		if (next < S.vocabSize && S.valid[next])
		{
			if (ll > -13.f) ll = sbgEvaluate(S, hist, next, ll);
#pragma unroll
			for (int i = 0; i < 8; ++i) hist[i] = (uint32_t)i == pos ? next : hist[i];   // no dynamic indexing: the ring stays in registers on the device
			pos = (pos + 1) & 7u;
		}
		return ll;
	}
}
