// Result assembly of the common case -- ONE chunk, ONE analysis (top-1), no affix joining, no pretokenized spans -- straight from the device's token
// records into the packed records of a result segment: no Token / PathTok objects, no per-token strings, and ONE table access per token where
// insertPathIntoResults (post.cpp: src/Kiwi.cpp:615-783) walks morpheme -> form -> characters -> dialect (four dependent misses on a model that does not
// fit the caches; MI355X box, 65 536 sentences: the assembly was 5.3 of the batch's 16 ms of host time, profiles/r06_h_*).  What a token needs of its
// morpheme is tabulated once per model (TokenTemplates: tag, sense, dialect, the JOINED form in one character pool, the allomorph facts of
// Kiwi.cpp:700-716).  Everything else is the general path's code over the other record type (post_common.hpp).  Product only: the oracle keeps the general
// path, and the CPU suite (lane-emulated kernels) compares the two byte for byte (KAMD_FAST_ASSEMBLY=0 switches this one off).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <string_view>
#include <vector>
#include "post.hpp"
#include "post_common.hpp"
#include "feature.hpp"
#include "textprep.hpp"
#include "device_types.hpp"

namespace kamd
{
	struct TokenTemplates
	{
		enum : uint8_t { KFORM_EMPTY = 1, EO_ALLOMORPH = 2, ENDS_HA = 4, POSITIVE = 8 };
		struct Rec { uint32_t joinedOff; uint16_t joinedLen; uint8_t tag, senseId; uint16_t dialect; uint8_t flags, pad; };
		std::vector<Rec> recs;
		std::vector<char16_t> pool;      // joinHangul(kform) of every form some morpheme has as its kform
		bool built = false;

		void build(const FlatModel& m)
		{
			const size_t nM = m.morphs.size();
			recs.assign(nM, Rec{});
			std::vector<uint32_t> at(m.forms.size(), 0xFFFFFFFFu), len(m.forms.size(), 0);
			for (size_t i = 0; i < nM; ++i)
			{
				const uint32_t f = m.morphKform[i];
				const FormRec& fr = m.forms[f];
				const char16_t* kf = (const char16_t*)m.formChars.data() + fr.charOff;
				if (at[f] == 0xFFFFFFFFu)
				{
					const U16 j = joinHangul(kf, fr.len);
					at[f] = (uint32_t)pool.size(); len[f] = (uint32_t)j.size();
					pool.insert(pool.end(), j.begin(), j.end());
				}
				Rec& r = recs[i];
				r.joinedOff = at[f]; r.joinedLen = (uint16_t)len[f];
				r.tag = m.morphs[i].tag; r.senseId = m.morphs[i].senseId;
				r.dialect = m.morphDialect.empty() ? (uint16_t)0 : m.morphDialect[i];
				uint8_t fl = 0;
				if (!fr.len) fl |= KFORM_EMPTY;
				if (T_EP <= r.tag && r.tag <= T_ETM && fr.len && kf[0] == 0xC5B4) fl |= EO_ALLOMORPH;
				if (fr.len && kf[fr.len - 1] == 0xD558) fl |= ENDS_HA;
				if (matchPolar((const uint16_t*)kf, fr.len, CP_POSITIVE)) fl |= POSITIVE;
				r.flags = fl;
			}
			built = true;
		}
	};

	struct FastAssembly
	{
		std::vector<uint16_t> wordPositions;
		std::vector<size_t> newlines;

		static bool applies(size_t topN, uint64_t match, bool pretok)
		{
			constexpr uint64_t M_JOIN_PARTICLE_YO = 1ull << 27;
			const bool off = [] { const char* e = std::getenv("KAMD_FAST_ASSEMBLY"); return e && std::atoi(e) == 0; }();      // (read per fetch: a test compares the two paths in one process)
			return !off && topN == 1 && !pretok && !(match & (M_JOIN_NOUN_PREFIX | M_JOIN_NOUN_SUFFIX | M_JOIN_VERB_SUFFIX | M_JOIN_ADJ_SUFFIX | M_JOIN_ADV_SUFFIX | M_MERGE_SAISIOT | M_JOIN_PARTICLE_YO));
		}

		// the text's only chunk, its only path: tk[0 .. nTok) with offsets relative to `so` in the normalised text `pt`
		void text(const FlatModel& mdl, const TokenTemplates& T, uint64_t match, bool integrateAllomorph, const char16_t* raw, size_t rawLen,
			const PreparedView& pt, uint32_t so, const DevToken* tk, uint32_t nTok, const DevToken* tkEnd, float score, ResultSegment& seg)
		{
			// (the template records of the tokens behind this text's -- the next text's, as the device wrote them -- are asked for now: they arrive while this text is assembled)
			for (const DevToken* q = tk + nTok; q < tkEnd && q < tk + nTok + 24; ++q) __builtin_prefetch(&T.recs[q->morph]);
			// getWordPositions (Kiwi.cpp:465-487)
			wordPositions.resize(rawLen);
			{
				uint32_t position = 0; bool contSpace = false;
				for (size_t i = 0; i < rawLen; ++i)
				{
					wordPositions[i] = (uint16_t)position;
					if (isSpace(raw[i])) { if (!contSpace) ++position; contSpace = true; }
					else contSpace = false;
				}
			}
			const size_t tok0 = seg.toks.size();
			// (records and characters are written through pointers into room made once per text -- what the tokens can need at most: a form joins into no more
			// units than it has, except a syllable followed by an old coda, which becomes three -- and the vectors are cut back at the end)
			size_t formRoom = 0;
			for (uint32_t k = 0; k < nTok; ++k) formRoom += (tk[k].ownKind ? (tk[k].ownKind == 2 ? (size_t)mdl.forms[tk[k].ownA].len : (size_t)tk[k].ownLen) * 3 / 2 + 2 : (size_t)T.recs[tk[k].morph].joinedLen + 3) + 1;
			const size_t form0 = seg.forms.size();
			seg.toks.resize(tok0 + nTok); seg.forms.resize(form0 + formRoom);
			FlatToken* outTok = seg.toks.data() + tok0; char16_t* const formBase = seg.forms.data(); char16_t* outForm = formBase + form0;
			const uint32_t* ptBegin = pt.position.p; const uint32_t* ptEnd = pt.position.p + pt.position.n;
			const bool compat = (match & M_COMPATIBLE_JAMO) != 0;
			int32_t prevMorph = -1;
			const size_t nPos = (size_t)(ptEnd - ptBegin);
			uint32_t lastBegin = 0; size_t lastB = 0;
			bool anyPairTag = false;
			for (uint32_t k = 0; k < nTok; ++k)
			{
				const DevToken& d = tk[k];
				std::u16string_view own;
				if (d.ownKind == 2) { const FormRec& fr = mdl.forms[d.ownA]; own = { (const char16_t*)mdl.formChars.data() + fr.charOff, fr.len }; }
				else if (d.ownKind) own = { pt.norm.p + so + d.ownA, d.ownLen };
				if (!own.empty() && own[0] == u' ') continue;
				const TokenTemplates::Rec& r = T.recs[d.morph];
				FlatToken o{};
				o.formOff = (uint64_t)(outForm - formBase);
				char16_t* const formAt = outForm;
				bool done = false;
				if (!integrateAllomorph && (r.flags & TokenTemplates::EO_ALLOMORPH))
				{
					const uint8_t pf = prevMorph >= 0 ? T.recs[prevMorph].flags : (uint8_t)TokenTemplates::POSITIVE;      // (no previous morpheme: an empty string matches every polarity)
					const char16_t first = (prevMorph >= 0 && (pf & TokenTemplates::ENDS_HA)) ? (char16_t)0xC5EC : (pf & TokenTemplates::POSITIVE) ? (char16_t)0xC544 : (char16_t)0;
					if (first)
					{
						const FormRec& kfr = mdl.forms[mdl.morphKform[d.morph]];
						U16 s(1, first);
						s.append((const char16_t*)mdl.formChars.data() + kfr.charOff + 1, kfr.len - 1);
						const U16 j = joinHangul(s);
						std::copy(j.begin(), j.end(), outForm); outForm += j.size();
						done = true;
					}
				}
				if (!done)
				{
					if (own.empty()) { const char16_t* src = T.pool.data() + r.joinedOff; for (uint32_t q = 0; q < r.joinedLen; ++q) outForm[q] = src[q]; outForm += r.joinedLen; }
					else { const U16 j = joinHangul(own.data(), own.size()); std::copy(j.begin(), j.end(), outForm); outForm += j.size(); }
				}
				o.formLen = (uint16_t)(outForm - formAt);
				if (compat) for (char16_t* q = formAt; q < outForm; ++q) *q = postc::toCompatibleConsonant(*q);
				*outForm++ = 0;
				o.tag = r.tag; o.morph = (int32_t)d.morph;
				const uint32_t begin = (uint32_t)d.begin + so, end = (uint32_t)d.end + so;
				// (upper_bound(begin) - 1 and lower_bound(end) of the general path: tokens come in text order, so both are a few steps from the previous token's)
				size_t b, e;
				if (begin >= lastBegin && end >= begin)
				{
					b = lastB;
					while (b + 1 < nPos && ptBegin[b + 1] <= begin) ++b;
					e = b;
					while (e < nPos && ptBegin[e] < end) ++e;
				}
				else
				{
					b = (std::upper_bound(ptBegin, ptEnd, begin) - ptBegin) - 1;
					e = std::lower_bound(ptBegin, ptEnd, end) - ptBegin;
				}
				lastBegin = begin; lastB = b;
				o.position = (uint32_t)b; o.length = (uint16_t)(e - b);
				o.score = d.wordScore; o.typoCost = d.typoCost; o.typoFormId = 0;
				o.senseOrScript = r.senseId;
				if ((r.tag == T_NNG || r.tag == T_NNP) && !own.empty()) o.senseOrScript = 0xFF;
				// updateTokenInfoScript (Kiwi.cpp:590-605)
				if ((o.tag == T_SL || o.tag == T_SH || o.tag == T_SW || o.tag == T_W_EMOJI) && (r.flags & TokenTemplates::KFORM_EMPTY) && o.formLen)
				{
					uint32_t c = formAt[0];
					if (isHighSurrogate(c)) c = mergeSurrogate(c, o.formLen > 1 ? formAt[1] : 0);
					o.senseOrScript = chr2ScriptType(c);
					if (o.senseOrScript == 1 /* latin */) o.tag = T_SL;
				}
				o.dialect = r.dialect;
				o.wordPosition = wordPositions[o.position];
				o.pairedToken = (uint32_t)-1;
				prevMorph = (int32_t)d.morph;
				anyPairTag = anyPairTag || o.tag == T_SSO || o.tag == T_SSC || o.tag == T_SB;
				*outTok++ = o;
			}
			seg.toks.resize((size_t)(outTok - seg.toks.data())); seg.forms.resize((size_t)(outForm - formBase));
			postc::newLinePositions(raw, rawLen, newlines);
			FlatToken* toks = seg.toks.data() + tok0; const size_t n = seg.toks.size() - tok0;
			const char16_t* forms = seg.forms.data();
			auto strOf = [forms](const FlatToken& t) { return std::u16string_view{ forms + t.formOff, t.formLen }; };
			if (anyPairTag) postc::fillPaired(toks, n, strOf);
			postc::fillSentLine(&mdl, toks, n, newlines, strOf);
			seg.anaScore.push_back(score);
			seg.anaTok.push_back((uint32_t)seg.toks.size());
			seg.textAna.push_back((uint32_t)seg.anaScore.size());
		}
	};
}
