// "Raw model" = what the reference's KiwiBuilder holds right before build() bakes it
// (/root/reference/include/kiwi/Form.h:27-134 MorphemeRaw, :205-223 FormRaw) plus the language
// model blob in the reference's own sj.knlm layout (/root/reference/include/kiwi/Knlm.h:9-15).
// Written by kiwi_amd/synth.py; consumed by the host-side baker (model.cpp) and by the
// reference bridge (oracle/ref_bridge.cpp).
#pragma once
#include "container.hpp"

namespace kamd
{
#pragma pack(push, 1)
	struct RawMorph
	{
		uint32_t kform, lmId, origId;
		int32_t combined;
		float userScore;
		uint32_t chunkPtr;
		uint8_t tag, vpPack, senseId, socket;
		uint16_t dialect;
		uint8_t nChunks, pad;
	};
#pragma pack(pop)
	static_assert(sizeof(RawMorph) == 32, "RawMorph layout");

	struct RawModel
	{
		const uint32_t* meta = nullptr; // nForms, nMorphs, vocabSize, flags
		const uint32_t* formPtr = nullptr;
		const uint16_t* formChars = nullptr;
		const uint32_t* formCandPtr = nullptr;
		const uint32_t* formCand = nullptr;
		const RawMorph* morph = nullptr;
		const uint32_t* chunkIds = nullptr;
		const uint8_t* chunkPos = nullptr;
		const uint8_t* knlm = nullptr;
		size_t knlmSize = 0;
		const uint8_t* sbg = nullptr;   // optional SkipBigram blob (reference skipbigram.mdl layout)
		size_t sbgSize = 0;
		const uint8_t* cong = nullptr;  // optional CoNgram blob (reference cong.mdl layout)
		size_t congSize = 0;
		const uint8_t* nounchr = nullptr;  // optional character-level CoNgram blob (reference nounchr.mdl layout)
		size_t nounchrSize = 0;

		size_t nForms() const { return meta[0]; }
		size_t nMorphs() const { return meta[1]; }
		size_t vocabSize() const { return meta[2]; }

		void bind(const Container& c)
		{
			if (std::strncmp(c.kind(), "KAMDRAW1", 8) != 0) throw std::runtime_error{ "not a KAMDRAW1 raw model file" };
			size_t n;
			meta = c.ptr<uint32_t>("meta", &n);
			if (n < 4) throw std::runtime_error{ "raw model: bad meta" };
			formPtr = c.ptr<uint32_t>("form_ptr", &n);
			if (n != nForms() + 1) throw std::runtime_error{ "raw model: form_ptr size" };
			formChars = c.ptr<uint16_t>("form_chars");
			formCandPtr = c.ptr<uint32_t>("form_cand_ptr", &n);
			if (n != nForms() + 1) throw std::runtime_error{ "raw model: form_cand_ptr size" };
			formCand = c.ptr<uint32_t>("form_cand");
			morph = c.ptr<RawMorph>("morph", &n);
			if (n != nMorphs()) throw std::runtime_error{ "raw model: morph size" };
			chunkIds = c.ptr<uint32_t>("chunk_ids");
			chunkPos = c.ptr<uint8_t>("chunk_pos");
			if (c.has("knlm")) { auto s = c.get("knlm"); knlm = s.data; knlmSize = s.size; }      // (a CoNgram-only model has none, like the reference's models/cong/base)
			if (c.has("sbg")) { auto g = c.get("sbg"); sbg = g.data; sbgSize = g.size; }
			if (c.has("cong")) { auto g = c.get("cong"); cong = g.data; congSize = g.size; }
			if (c.has("nounchr")) { auto g = c.get("nounchr"); nounchr = g.data; nounchrSize = g.size; }
		}
	};
}
