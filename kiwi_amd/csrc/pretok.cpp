// makePretokenizedSpanGroup (/root/reference/src/Kiwi.cpp:785-946) over the flat model: see pretok.hpp.
#include <algorithm>
#include <stdexcept>
#include "pretok.hpp"
#include "textprep.hpp"
#include "kchars.hpp"

namespace kamd
{
	namespace
	{
		// areTagsEqual (include/kiwi/Types.h:249-252): irregularity aside when the caller lets it be inferred
		inline bool tagsEqual(uint8_t a, uint8_t b, bool infer) { return infer ? ((a & 0x7F) == (b & 0x7F)) : (a == b); }
		// getDefaultMorphemeId (include/kiwi/Kiwi.h:64-67): the LM id of a word the model does not have = its tag's default morpheme
		inline uint32_t defaultMorphemeId(uint8_t tag) { return (uint32_t)(tag & 0x7F) + 1u; }
	}

	void makePretokGroup(const FlatModel& m, const char16_t* text, size_t len, uint64_t match, const std::vector<PtSpan>& spans, PretokGroup& g)
	{
		g = PretokGroup{};
		if (spans.empty()) return;
		U16 norm; std::vector<uint32_t> pos;
		normalizeWithPosition(text, len, norm, pos);
		if (match & M_NORMALIZE_CODA) normalizeCoda(norm);
		const uint32_t nM = m.h.nMorphs, nF = m.h.nForms;
		TempEntries& temps = g.temps;
		for (const PtSpan& sp : spans)
		{
			if (sp.begin >= sp.end || sp.end > len) throw std::invalid_argument{ "pretokenized span outside the text or empty" };
			const uint32_t b = pos[sp.begin], e = pos[sp.end];
			if (b >= e) throw std::invalid_argument{ "pretokenized span outside the text or empty" };
			PretokGroup::Span sn{ b, e, 0, false };
			if (sp.tokens.empty())
			{
				// the dictionary form spelled by the text, or the default form of NNP (Kiwi.cpp:820-828)
				const int32_t f = formIdOfString(m, norm.substr(b, e - b));
				sn.form = f >= 0 ? (uint32_t)f : (uint32_t)T_NNP - 1u;
			}
			else if (sp.tokens.size() == 1)
			{
				const PtToken& t = sp.tokens[0];
				U16 fs; std::vector<uint32_t> dp;
				normalizeWithPosition(t.form.data(), t.form.size(), fs, dp);      // normalizeHangul
				const int32_t f = formIdOfString(m, fs);
				if (f >= 0 && m.forms[f].candCnt == 1 && tagsEqual(m.morphs[m.formCand[m.forms[f].candOff]].tag, t.tag, t.inferRegularity)) sn.form = (uint32_t)f;      // :833-838
				else
				{
					// a new form: the entry's morphemes of that tag (at most two) or a new morpheme (:839-882)
					TempEntries::Form tf; tf.str = fs;
					if (f >= 0)
						for (uint32_t ci = 0; ci < m.forms[f].candCnt && tf.cands.size() < 2; ++ci)
						{
							const uint32_t mi = m.formCand[m.forms[f].candOff + ci];
							if (tagsEqual(m.morphs[mi].tag, t.tag, t.inferRegularity)) tf.cands.push_back(mi);
						}
					if (tf.cands.empty())
					{
						tf.cands.push_back(nM + (uint32_t)temps.morphs.size());
						temps.morphs.push_back(TempEntries::Morph{ (uint32_t)temps.forms.size(), t.tag, defaultMorphemeId(t.tag), {} });
					}
					sn.form = nF + (uint32_t)temps.forms.size();
					temps.forms.push_back(std::move(tf));
				}
			}
			else
			{
				// several tokens: one new form with one new morpheme whose chunks are the tokens (:884-934)
				const uint32_t wholeForm = (uint32_t)temps.forms.size(), wholeMorph = nM + (uint32_t)temps.morphs.size();
				temps.forms.push_back(TempEntries::Form{ U16{}, { wholeMorph } });
				temps.morphs.push_back(TempEntries::Morph{ wholeForm, 0 /* POSTag::unknown */, 0, {} });      // (its chunks below: the vector may grow in between)
				const size_t wholeAt = temps.morphs.size() - 1;
				std::vector<TempEntries::Chunk> chunks;
				for (const PtToken& t : sp.tokens)
				{
					U16 fs; std::vector<uint32_t> dp;
					normalizeWithPosition(t.form.data(), t.form.size(), fs, dp);
					const int32_t f = formIdOfString(m, fs);
					uint32_t found = 0xFFFFFFFFu;
					if (f >= 0)
						for (uint32_t ci = 0; ci < m.forms[f].candCnt; ++ci)
						{
							const uint32_t mi = m.formCand[m.forms[f].candOff + ci];
							if (m.morphs[mi].tag == t.tag) { found = mi; break; }
						}
					if (found == 0xFFFFFFFFu)
					{
						found = nM + (uint32_t)temps.morphs.size();
						temps.morphs.push_back(TempEntries::Morph{ (uint32_t)temps.forms.size(), t.tag, defaultMorphemeId(t.tag), {} });
						temps.forms.push_back(TempEntries::Form{ fs, {} });      // (formStrs: the string alone, no candidates)
					}
					if (t.begin > t.end || (uint64_t)sp.begin + t.end > len) throw std::invalid_argument{ "token range outside its pretokenized span's text" };
					const uint32_t cb = pos[sp.begin + t.begin] - b, ce = pos[sp.begin + t.end] - b;
					if (cb > 255 || ce > 255) throw std::invalid_argument{ "pretokenized span longer than 255 units" };      // (Morpheme::chunks keeps byte offsets)
					chunks.push_back(TempEntries::Chunk{ found, (uint8_t)cb, (uint8_t)ce });
				}
				temps.morphs[wholeAt].chunks = std::move(chunks);
				sn.form = nF + wholeForm;
			}
			sn.fallback = sn.form + 1 >= (uint32_t)T_NNG && sn.form + 1 < (uint32_t)T_MAX;      // within(form, value(nng), value(max)): KTrie.cpp:1197
			g.spans.push_back(sn);
		}
		std::stable_sort(g.spans.begin(), g.spans.end(), [](const PretokGroup::Span& a, const PretokGroup::Span& c) { return a.begin < c.begin; });
		for (size_t i = 1; i < g.spans.size(); ++i)
			if (g.spans[i - 1].end > g.spans[i].begin) throw std::invalid_argument{ "`PretokenizedSpan`s should not have overlapped ranges." };      // Kiwi.cpp:941-944
		if (!temps.forms.empty())
		{
			bakeTempsOverlay(m, temps, g.overlay);
			const TempOverlay& o = g.overlay;
			g.devMorphs = o.morphs;
			for (size_t i = 0; i < g.devMorphs.size(); ++i) { g.devMorphs[i].feat = (uint16_t)o.morphPath[i]; g.devMorphs[i].prevFlags = (uint8_t)(o.morphPath[i] >> 16); }
			FlatModel& hm = g.hostModel;
			hm.h = m.h; hm.h.nForms = nF + (uint32_t)temps.forms.size(); hm.h.nMorphs = nM + (uint32_t)temps.morphs.size();
			hm.morphs = m.morphs; hm.morphs.insert(hm.morphs.end(), o.morphs.begin(), o.morphs.end());
			hm.morphKform = m.morphKform; hm.morphKform.insert(hm.morphKform.end(), o.morphKform.begin(), o.morphKform.end());
			hm.forms.assign(m.forms.begin(), m.forms.begin() + nF); hm.forms.insert(hm.forms.end(), o.forms.begin(), o.forms.end());
			hm.formChars.assign(m.formChars.begin(), m.formChars.end() - 1); hm.formChars.insert(hm.formChars.end(), o.formChars.begin(), o.formChars.end());
			if (!m.morphDialect.empty()) { hm.morphDialect = m.morphDialect; hm.morphDialect.resize(hm.morphs.size(), 0); }
		}
	}
}
