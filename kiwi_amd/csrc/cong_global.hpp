// The global CoNgram model (ModelType::congGlobal, window 7): the arithmetic of one score, written once for the host (oracle, tests) and the device.
//
// A word whose bit is set in the model's distant-token mask is scored as a mixture over the path's context and the last seven such words of the path
// (CoNgramModel::progress, src/CoNgramModel.cpp:802-868; progressMatrixWSort / WOSort, :1037-1466):
//
//     w[0]   = confidence(context)           + positionConfidence[0]
//     w[k+1] = confidence(history word k)    + positionConfidence[k + 1]        (-99999 for an empty history slot)
//     w      = logSoftmax(w)
//     ll     = logSumExp_k( w[k] + score(row k, next) [+ validTokenSum(context)] )
//
// score(row, next) is the quantised product of congScore (flat_model.hpp) over the context row / the distant row of the history word.  Which of the
// reference's code paths computes a score decides its roundings, and both are reproduced:
//
//   * `single` (LmState::next -> progress(): one incoming path and one candidate, later chunks of a candidate, combining stems, the closing token):
//     logSoftmax over packets of four with a horizontal sum, the valid-token sum SUBTRACTED from term 0 before and added after the logSumExp, the final
//     logarithm is libm's;
//   * `matrix` (progressMatrix*, "transposed": one score per SIMD lane): maximum, sum and both logarithms term by term, the valid-token sum ADDED to terms
//     1..7, the final logarithm is the SIMD approximation.
//
// exp / log of the SIMD paths are the Cephes-style polynomials of src/SIMD.hpp:121-246 as the SSE2 / SSE4.1 operator set evaluates them -- per lane, multiply
// and add rounded separately (maddf = addf(mulf), SIMD.hpp:295); the pin of the CoNgram oracle is the reference's SSE4.1 build.  Compiled with
// -ffp-contract=off everywhere (csrc/Makefile and the lane emulator's; the oracle's x86-64 baseline has no FMA to contract to).
#pragma once
#include "flat_model.hpp"
#include "exact_math.hpp"

namespace kamd
{
	namespace congg
	{
		constexpr uint32_t WINDOW = 7;

		KAMD_HD float fmaxSse(float a, float b) { return a > b ? a : b; }      // _mm_max_ps(a, b)
		KAMD_HD float fminSse(float a, float b) { return a < b ? a : b; }      // _mm_min_ps(a, b)

		// simd::OperatorBase::expf (src/SIMD.hpp:121-167), one lane
		KAMD_HD float expfSimd(float x0)
		{
			const float x = fmaxSse(fminSse(x0, 88.723f), -88.723f);
			const float m = floorf(x * 1.44269504088896341f + 0.5f);
			float r = m * -0.693359375f + x;
			r = m * 2.12194440e-4f + r;
			const float r2 = r * r, r3 = r2 * r;
			float y = 1.9875691500E-4f * r + 1.3981999507E-3f;
			float y1 = 4.1665795894E-2f * r + 1.6666665459E-1f;
			const float y2 = r + 1.0f;
			y = y * r + 8.3334519073E-3f;
			y1 = y1 * r + 5.0000001201E-1f;
			y = y * r3 + y1;
			y = y * r2 + y2;
			// ldexpf_fast (SIMD.hpp:96-106): the biased exponent clamped to 0 .. 255, 2^e formed by a shift
			const float eb = fminSse(fmaxSse(m + 127.f, 0.f), 255.f);
			const float p = exact::u2f((uint32_t)(int32_t)eb << 23);
			return fmaxSse(y * p, x0);
		}

		// simd::OperatorBase::logf (src/SIMD.hpp:169-246), one lane, for finite x > 0 (a sum of exponentials with one term exp(0))
		KAMD_HD float logfSimd(float x0)
		{
			float x = fmaxSse(x0, exact::u2f(0x00800000u));
			uint32_t ix = exact::f2u(x);
			float e = (float)(int32_t)((ix & 0x7F800000u) >> 23) - 126.f;
			ix = (ix & ~0x7F800000u) | (126u << 23);
			x = exact::u2f(ix);
			const bool lt = x < 0.707106781186547524f;
			const float tmp = lt ? x : 0.f;
			x = x - 1.0f;
			e = e - (lt ? 1.0f : 0.f);
			x = x + tmp;
			const float x2 = x * x, x3 = x2 * x;
			float y = 7.0376836292E-2f * x + -1.1514610310E-1f;
			float y1 = -1.2420140846E-1f * x + 1.4249322787E-1f;
			float y2 = 2.0000714765E-1f * x + -2.4999993993E-1f;
			y = y * x + 1.1676998740E-1f;
			y1 = y1 * x + -1.6668057665E-1f;
			y2 = y2 * x + 3.3333331174E-1f;
			y = y * x3 + y1;
			y = y * x3 + y2;
			y = y * x3;
			y = -0.5f * x2 + y;
			x = x + y;
			x = e * 0.69314718f + x;
			return x;
		}

		// a row of the context table (a context's or a history word's distant row) against an output row: congScore's arithmetic
		KAMD_HD float rowScore(const uint8_t* a8, const uint8_t* b8, uint32_t dim, bool outputFirst)
		{
			const int8_t* a = reinterpret_cast<const int8_t*>(a8);
			const int8_t* b = reinterpret_cast<const int8_t*>(b8);
			int32_t acc = 0;
			for (uint32_t k = 0; k < dim; ++k) acc += (int32_t)a[k] * (int32_t)b[k];
			float cs, os, bias;
			__builtin_memcpy(&cs, a + dim, 4); __builtin_memcpy(&bias, a + dim + 4, 4); __builtin_memcpy(&os, b + dim, 4);
			return outputFirst ? (float)acc * os * cs + bias : (float)acc * cs * os + bias;
		}

		// the eight mixture weights before normalisation; hist[0..6] = CoNgramState::history[0..6], 0 = empty
		KAMD_HD void rawWeights(const CongView& C, uint32_t ctx, const uint32_t* hist, float* w)
		{
			w[0] = C.posConf[0] + C.ctxConf[2 * ctx];
			for (uint32_t k = 0; k < WINDOW; ++k) w[k + 1] = C.posConf[k + 1] + (hist[k] ? C.distConf[hist[k]] : -99999.f);
		}

		// logSoftmaxImpl<sse4_1, 8> (src/MathFunc.hpp:64-86): packets {0..3} and {4..7}, lane sums added horizontally (redsumf, SIMD.hpp:365-369)
		KAMD_HD float packetSumExp8(const float* w, float mx)
		{
			float s[4];
			for (int l = 0; l < 4; ++l) s[l] = (0.f + expfSimd(w[l] - mx)) + expfSimd(w[4 + l] - mx);
			return (s[0] + s[2]) + (s[1] + s[3]);
		}
		KAMD_HD void logSoftmax8(float* w)
		{
			float mx = w[0];
			for (int i = 1; i < 8; ++i) mx = fmaxSse(mx, w[i]);
			const float sub = logfSimd(packetSumExp8(w, mx)) + mx;
			for (int i = 0; i < 8; ++i) w[i] = w[i] - sub;
		}
		// logSumExpImpl<sse4_1, 8> (src/MathFunc.hpp:11-31): the same sum, libm's logarithm
		KAMD_HD float logSumExp8(const float* w)
		{
			float mx = w[0];
			for (int i = 1; i < 8; ++i) mx = fmaxSse(mx, w[i]);
			return exact::logf_glibc(packetSumExp8(w, mx)) + mx;
		}
		// LogSoftmaxTransposed<arch, 8>::block / LogSumExpTransposed<arch, 8>::block (src/MathFunc.hpp:128-190, 245-291): one lane
		KAMD_HD void logSoftmaxT8(float* w)
		{
			float m = fmaxSse(w[0], w[1]);
			for (int i = 2; i < 8; ++i) m = fmaxSse(m, w[i]);
			for (int i = 0; i < 8; ++i) w[i] = w[i] - m;
			float s = expfSimd(w[0]);
			for (int i = 1; i < 8; ++i) s = s + expfSimd(w[i]);
			s = logfSimd(s);
			for (int i = 0; i < 8; ++i) w[i] = w[i] - s;
		}
		KAMD_HD float logSumExpT8(const float* w)
		{
			float m = fmaxSse(w[0], w[1]);
			for (int i = 2; i < 8; ++i) m = fmaxSse(m, w[i]);
			float s = expfSimd(w[0] - m);
			for (int i = 1; i < 8; ++i) s = s + expfSimd(w[i] - m);
			return m + logfSimd(s);
		}

		// progress() for a valid distant token (src/CoNgramModel.cpp:812-841): scatteredGEMMOpt(8, 1) is the baseline kernel (qgemm.hpp:184-187)
		KAMD_HD float scoreSingle(const CongView& C, uint32_t ctx, const uint32_t* hist, uint32_t next)
		{
			float w[8];
			rawWeights(C, ctx, hist, w);
			logSoftmax8(w);
			const uint8_t* out = C.outEmb + (size_t)next * C.stride;
			w[0] = w[0] + rowScore(C.ctxEmb + (size_t)ctx * C.stride, out, C.dim, false);
			for (uint32_t k = 0; k < WINDOW; ++k) w[k + 1] = w[k + 1] + rowScore(C.distEmb + (size_t)hist[k] * C.stride, out, C.dim, false);      // (an empty slot: distant row 0)
			const float vts = C.ctxConf[2 * ctx + 1];
			w[0] = w[0] - vts;
			return logSumExp8(w) + vts;
		}

		// one entry of progressMatrixWSort / WOSort for a valid distant token (src/CoNgramModel.cpp:1262-1296, 1420-1462); outputFirst: the rounding of the
		// quantised products, decided by the shape of the whole matrix (see the callers)
		KAMD_HD float scoreMatrix(const CongView& C, uint32_t ctx, const uint32_t* hist, uint32_t next, bool outputFirst)
		{
			float w[8];
			rawWeights(C, ctx, hist, w);
			logSoftmaxT8(w);
			const float vts = C.ctxConf[2 * ctx + 1];
			for (int i = 1; i < 8; ++i) w[i] = w[i] + vts;
			const uint8_t* out = C.outEmb + (size_t)next * C.stride;
			w[0] = w[0] + rowScore(C.ctxEmb + (size_t)ctx * C.stride, out, C.dim, outputFirst);
			for (uint32_t k = 0; k < WINDOW; ++k) if (hist[k]) w[k + 1] = w[k + 1] + rowScore(C.distEmb + (size_t)hist[k] * C.stride, out, C.dim, outputFirst);
			return logSumExpT8(w);
		}

		// the history after `next` (progress(): src/CoNgramModel.cpp:912-919; nextState: :1004-1013): slot 7 holds the newest word, the ring moves on only
		// when slot 7 was taken
		KAMD_HD void pushHistory(const CongView& C, uint32_t* hist /* [8] */, uint32_t next)
		{
			if (hist[WINDOW]) for (uint32_t k = 0; k < WINDOW; ++k) hist[k] = hist[k + 1];
			hist[WINDOW] = C.distant(next) ? next : 0;
		}
	}
}
