// TEST INFRASTRUCTURE -- functional stand-in for third_party/streamvbyte (an empty git submodule of the reference checkout), so that the
// REAL src/CoNgramModel.cpp compiles into oracle/_ref.  StreamVByte (D. Lemire, N. Kurz, C. Rupp: "Stream VByte: Faster Byte-Oriented
// Integer Compression", 2017; github.com/fast-pack/streamvbyte, include/streamvbyte.h) stores n 32-bit integers as ceil(n/4) control bytes --
// four 2-bit length codes each, first integer in the low bits -- followed by the integers' significant bytes, little endian.
// Standard codes: 0..3 -> 1, 2, 3, 4 bytes.  "0124" codes: 0..3 -> 0, 1, 2, 4 bytes.  Scalar restatement of that published format.
#pragma once
#include <stddef.h>
#include <stdint.h>

static inline size_t streamvbyte_max_compressedbytes(const uint32_t length) { return (size_t)((length + 3) / 4) + (size_t)length * sizeof(uint32_t); }

static inline size_t svb_standin_encode(const uint32_t* in, uint32_t count, uint8_t* out, int v0124)
{
	uint8_t* keyPtr = out;
	const uint32_t keyLen = (count + 3) / 4;
	uint8_t* dataPtr = keyPtr + keyLen;
	for (uint32_t i = 0; i < keyLen; ++i) keyPtr[i] = 0;
	for (uint32_t i = 0; i < count; ++i)
	{
		const uint32_t v = in[i];
		uint32_t code, nbytes;
		if (v0124) { if (v == 0) { code = 0; nbytes = 0; } else if (v < (1u << 8)) { code = 1; nbytes = 1; } else if (v < (1u << 16)) { code = 2; nbytes = 2; } else { code = 3; nbytes = 4; } }
		else { if (v < (1u << 8)) { code = 0; nbytes = 1; } else if (v < (1u << 16)) { code = 1; nbytes = 2; } else if (v < (1u << 24)) { code = 2; nbytes = 3; } else { code = 3; nbytes = 4; } }
		keyPtr[i / 4] |= (uint8_t)(code << ((i % 4) * 2));
		for (uint32_t b = 0; b < nbytes; ++b) *dataPtr++ = (uint8_t)(v >> (8 * b));
	}
	return (size_t)(dataPtr - out);
}

static inline size_t svb_standin_decode(const uint8_t* in, uint32_t* out, uint32_t count, int v0124)
{
	const uint8_t* keyPtr = in;
	const uint32_t keyLen = (count + 3) / 4;
	const uint8_t* dataPtr = keyPtr + keyLen;
	static const uint8_t lenStd[4] = { 1, 2, 3, 4 }, len0124[4] = { 0, 1, 2, 4 };
	for (uint32_t i = 0; i < count; ++i)
	{
		const uint32_t code = (keyPtr[i / 4] >> ((i % 4) * 2)) & 3u;
		const uint32_t nbytes = v0124 ? len0124[code] : lenStd[code];
		uint32_t v = 0;
		for (uint32_t b = 0; b < nbytes; ++b) v |= (uint32_t)(*dataPtr++) << (8 * b);
		out[i] = v;
	}
	return (size_t)(dataPtr - in);
}

static inline size_t streamvbyte_encode(const uint32_t* in, uint32_t length, uint8_t* out) { return svb_standin_encode(in, length, out, 0); }
static inline size_t streamvbyte_encode_0124(const uint32_t* in, uint32_t length, uint8_t* out) { return svb_standin_encode(in, length, out, 1); }
static inline size_t streamvbyte_decode(const uint8_t* in, uint32_t* out, uint32_t length) { return svb_standin_decode(in, out, length, 0); }
static inline size_t streamvbyte_decode_0124(const uint8_t* in, uint32_t* out, uint32_t length) { return svb_standin_decode(in, out, length, 1); }
