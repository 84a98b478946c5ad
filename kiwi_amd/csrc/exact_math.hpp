// expf / logf that return, bit for bit, what glibc's libm returns on x86-64 (glibc >= 2.27: the ARM "optimized routines"
// single-precision exp / log, sysdeps/ieee754/flt-32/e_expf.c, e_logf.c) -- written from the published algorithm: table-driven,
// evaluated in double precision, one final rounding to float.  The reference scores SkipBigram / CoNgram mixtures with
// std::exp / std::log on floats (src/MathFunc.hpp:43-56, ArchType none / balanced); a device restatement that has to agree
// bit-exactly cannot use the GPU's own expf / logf (1-2 ulp, different roundings).  Arguments outside the ranges noted below
// fall back to the result for the nearest special case; the callers here only ever pass x <= 0 to the exponential and
// finite positive sums to the logarithm.  Checked against libm on the host by tests/test_exact_math.py.
#pragma once
#include <cstdint>
#include <cstring>
#include "kchars.hpp"   // KAMD_HD

namespace kamd
{
	namespace exact
	{
		KAMD_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
		KAMD_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
		KAMD_HD uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
		KAMD_HD double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

		// 2^(i/32) as bit patterns with the exponent contribution of i/32 taken out (exp2f_data.c)
		KAMD_HD uint64_t exp2fTab(uint32_t i)
		{
			const uint64_t T[32] = {
				0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
				0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
				0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
				0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
				0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
				0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
			return T[i];
		}

		// expf for finite x in about [-103.97, 88.72]; below: 0 (underflow), above: +inf.  Subnormal results are produced by the
		// final double -> float conversion exactly as in the original (which only special-cases the exceptions, not the value).
		KAMD_HD float expf_glibc(float x)
		{
			const double N = 32.0;
			const double InvLn2N = 0x1.71547652b82fep+0 * N;
			const double Shift = 0x1.8p+52;
			const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
			if (x != x) return x;
			if (x > 0x1.62e42ep6f) return u2f(0x7f800000u);      // x > log(0x1p128)
			if (x < -0x1.9fe368p6f) return 0.0f;                  // x < log(0x1p-150)
			const double xd = (double)x;
			const double z = InvLn2N * xd;
			double kd = z + Shift;
			const uint64_t ki = d2u(kd);
			kd -= Shift;
			const double r = z - kd;
			uint64_t t = exp2fTab((uint32_t)(ki % 32));
			t += ki << (52 - 5);
			const double s = u2d(t);
			const double zz = C0 * r + C1;
			const double r2 = r * r;
			double y = C2 * r + 1.0;
			y = zz * r2 + y;
			y = y * s;
			return (float)y;
		}

		// logf for finite x > 0 (normal or subnormal)
		KAMD_HD float logf_glibc(float x)
		{
			const double invc[16] = { 0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,
				0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1, 0x1.b2036576afce6p-1,
				0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1 };
			const double logc[16] = { -0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3, -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,
				-0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4, -0x1.252f438e10c1ep-5, 0x0p+0, 0x1.aa5aa5df25984p-5, 0x1.c5e53aa362eb4p-4, 0x1.526e57720db08p-3,
				0x1.bc2860d22477p-3, 0x1.1058bc8a07ee1p-2, 0x1.4043057b6ee09p-2 };
			const double Ln2 = 0x1.62e42fefa39efp-1;
			const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
			uint32_t ix = f2u(x);
			if (ix == 0x3f800000u) return 0.0f;
			if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u)
			{
				if (ix * 2 == 0) return u2f(0xff800000u);             // log(0) = -inf
				if (ix == 0x7f800000u) return x;                      // log(inf) = inf
				if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return u2f(0x7fc00000u);   // negative or NaN
				ix = f2u(x * 0x1p23f);                                 // subnormal: normalise
				ix -= 23u << 23;
			}
			const uint32_t tmp = ix - 0x3f330000u;
			const uint32_t i = (tmp >> (23 - 4)) % 16;
			const int32_t k = (int32_t)tmp >> 23;
			const uint32_t iz = ix - (tmp & 0xff800000u);
			const double z = (double)u2f(iz);
			const double r = z * invc[i] - 1.0;
			const double y0 = logc[i] + (double)k * Ln2;
			const double r2 = r * r;
			double y = A1 * r + A2;
			y = A0 * r2 + y;
			y = y * r2 + (y0 + r);
			return (float)y;
		}

		// expm1f (glibc 2.35 sysdeps/ieee754/flt-32/s_expm1f.c: the fdlibm algorithm in single precision) for finite x; the callers here
		// (tanhf_glibc) pass |x| < 44.  Every operation is a single-precision one in the original's order (-ffp-contract=off).
		KAMD_HD float expm1f_glibc(float x)
		{
			const float ln2_hi = u2f(0x3f317180u), ln2_lo = u2f(0x3717f7d1u), invln2 = u2f(0x3fb8aa3bu);
			const float Q1 = u2f(0xbd088889u), Q2 = u2f(0x3ad00d01u), Q3 = u2f(0xb8a670cdu), Q4 = u2f(0x36867e54u), Q5 = u2f(0xb457edbbu);
			uint32_t hx = f2u(x);
			const bool neg = (hx & 0x80000000u) != 0;
			hx &= 0x7fffffffu;
			if (hx >= 0x4195b844u)      // |x| >= 27 ln2
			{
				if (hx >= 0x42b17218u)
				{
					if (hx > 0x7f800000u) return x + x;
					if (hx == 0x7f800000u) return neg ? -1.0f : x;
					if (x > u2f(0x42b17180u)) return u2f(0x7f800000u);
				}
				if (neg) return -1.0f;
			}
			float hi, lo, c = 0.f, t, e;
			int32_t k;
			if (hx > 0x3eb17218u)      // |x| > 0.5 ln2
			{
				if (hx < 0x3F851592u)      // and |x| < 1.5 ln2
				{
					if (!neg) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
					else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
				}
				else
				{
					k = (int32_t)(invln2 * x + (neg ? -0.5f : 0.5f));
					t = (float)k;
					hi = x - t * ln2_hi;
					lo = t * ln2_lo;
				}
				x = hi - lo;
				c = (hi - x) - lo;
			}
			else if (hx < 0x33000000u) return x;      // |x| < 2^-25
			else k = 0;
			const float hfx = 0.5f * x;
			const float hxs = x * hfx;
			const float r1 = 1.0f + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
			t = 3.0f - r1 * hfx;
			e = hxs * ((r1 - t) / (6.0f - x * t));
			if (k == 0) return x - (x * e - hxs);
			e = (x * (e - c) - c);
			e -= hxs;
			if (k == -1) return 0.5f * (x - e) - 0.5f;
			if (k == 1)
			{
				if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
				return 1.0f + 2.0f * (x - e);
			}
			float y;
			if (k <= -2 || k > 56)
			{
				y = 1.0f - (e - x);
				y = u2f(f2u(y) + ((uint32_t)k << 23));
				return y - 1.0f;
			}
			if (k < 23)
			{
				t = u2f(0x3f800000u - (0x1000000u >> k));      // 1 - 2^-k
				y = t - (e - x);
				y = u2f(f2u(y) + ((uint32_t)k << 23));
			}
			else
			{
				t = u2f((uint32_t)(0x7f - k) << 23);      // 2^-k
				y = x - (e + t);
				y += 1.0f;
				y = u2f(f2u(y) + ((uint32_t)k << 23));
			}
			return y;
		}

		// tanhf (glibc 2.35 sysdeps/ieee754/flt-32/s_tanhf.c) for finite x
		KAMD_HD float tanhf_glibc(float x)
		{
			const uint32_t jx = f2u(x), ix = jx & 0x7fffffffu;
			if (ix >= 0x7f800000u) return (ix > 0x7f800000u) ? x + x : ((jx >> 31) ? -1.0f : 1.0f);
			float z;
			if (ix < 0x41b00000u)      // |x| < 22
			{
				if (ix == 0) return x;
				if (ix < 0x24000000u) return x * (1.0f + x);
				const float ax = u2f(ix);
				if (ix >= 0x3f800000u) { const float t = expm1f_glibc(2.0f * ax); z = 1.0f - 2.0f / (t + 2.0f); }
				else { const float t = expm1f_glibc(-2.0f * ax); z = -t / (t + 2.0f); }
			}
			else z = 1.0f;
			return (jx >> 31) ? -z : z;
		}
	}
}
