// Host side of typo correction: see typo.hpp.
#include <algorithm>
#include <map>
#include <numeric>
#include <stdexcept>
#include "typo.hpp"
#include "hostutil.hpp"
#include "textprep.hpp"

namespace kamd
{
	namespace
	{
		inline bool isSyllable(char16_t c) { return 0xAC00 <= c && c < 0xD7A4; }                 // include/kiwi/Utils.h:64-92
		inline bool isOnsetJamo(char16_t c) { return 0x1100 <= c && c < 0x1100 + 19; }
		inline bool isVowelCompatJamo(char16_t c) { return 0x314F <= c && c < 0x3164; }
		inline char16_t joinOnsetVowel(size_t onset, size_t vowel) { return (char16_t)(0xAC00 + (char16_t)((onset * 21 + vowel) * 28)); }
		std::u16string normalizeRule(const std::u16string& s)      // normalizeHangul (src/StrUtils.h:494-521)
		{
			std::u16string o;
			for (char16_t c : s)
			{
				if (c == 0xB42C) c = 0xB410;
				if (0xAC00 <= c && c < 0xD7A4)
				{
					const int coda = (c - 0xAC00) % 28;
					o.push_back((char16_t)(c - coda));
					if (coda) o.push_back((char16_t)(coda + 0x11A7));
				}
				else o.push_back(c);
			}
			return o;
		}
		inline bool leftCondMatched(const char16_t* s, size_t n, uint8_t cond) { return typoLeftCondMatched((const uint16_t*)s, n, cond); }
	}

	size_t TypoTransformer::KeyHash::operator()(const Key& k) const
	{
		// tail first, then every head folded in: h ^= H(head) + (h << 6) + (h >> 2)
		size_t h = std::hash<uint16_t>{}(k.dialect);
		h ^= std::hash<uint8_t>{}(k.cond) + (h << 6) + (h >> 2);
		h ^= std::hash<std::u16string>{}(k.error) + (h << 6) + (h >> 2);
		h ^= std::hash<std::u16string>{}(k.orig) + (h << 6) + (h >> 2);
		return h;
	}

	void TypoTransformer::addWithCond(const std::u16string& orig, const std::u16string& error, float cost, uint8_t cond, uint16_t dialect)
	{
		if (orig == error) return;
		auto put = [&](Key&& k)
		{
			auto ins = typos_.emplace(std::move(k), cost);
			if (!ins.second) ins.first->second = std::isfinite(cost) ? std::min(ins.first->second, cost) : cost;
		};
		if (cond == TC_NONE || cond == TC_VOWEL || cond == TC_ANY || cond == TC_CONTINUAL || cond == TC_BOUNDARY) put(Key{ orig, error, cond, dialect });
		else if (cond == TC_APPLOSIVE)
		{
			// the reference iterates over a string literal, i.e. over its terminating NUL too: the 14th variant matches at the start of a text only
			static const char16_t codas[14] = { 0x11A8, 0x11A9, 0x11AA, 0x11AE, 0x11B8, 0x11B9, 0x11BA, 0x11BB, 0x11BD, 0x11BE, 0x11BF, 0x11C0, 0x11C1, 0 };
			for (char16_t c : codas)
			{
				std::u16string o, e;
				o.push_back(c); o += orig;
				if (c) e.push_back(c);
				e += error;
				put(Key{ o, e, (uint8_t)(c ? TC_NONE : cond), dialect });
			}
		}
		else throw std::invalid_argument{ "typo rule: unsupported left condition" };
	}

	void TypoTransformer::addNormalized(const std::u16string& orig, const std::u16string& error, float cost, uint8_t cond, uint16_t dialect)
	{
		if (orig.empty() || error.empty()) throw std::invalid_argument{ "typo rule: empty string" };
		if (isOnsetJamo(orig.back()) != isOnsetJamo(error.back())) throw std::invalid_argument{ "typo rule: `orig` and `error` must both end in an onset or neither" };
		if (isVowelCompatJamo(orig[0]) != isVowelCompatJamo(error[0])) throw std::invalid_argument{ "typo rule: `orig` and `error` must both start with a vowel or neither" };
		if (isOnsetJamo(orig.back()))
		{
			std::u16string o = orig, e = error;
			for (size_t v = 0; v < 21; ++v) { o.back() = joinOnsetVowel(orig.back() - 0x1100, v); e.back() = joinOnsetVowel(error.back() - 0x1100, v); addWithCond(o, e, cost, cond, dialect); }
		}
		else if (isVowelCompatJamo(orig[0]))
		{
			std::u16string o = orig, e = error;
			for (size_t c = 0; c < 19; ++c) { o[0] = joinOnsetVowel(c, orig[0] - 0x314F); e[0] = joinOnsetVowel(c, error[0] - 0x314F); addWithCond(o, e, cost, cond, dialect); }
		}
		else addWithCond(orig, error, cost, cond, dialect);
	}

	void TypoTransformer::add(const std::u16string& orig, const std::u16string& error, float cost, uint8_t cond, uint16_t dialect)
	{
		addNormalized(normalizeRule(orig), normalizeRule(error), cost, cond, dialect);
	}

	void TypoTransformer::addEntry(const std::u16string& orig, const std::u16string& error, float cost, uint8_t cond, uint16_t dialect)
	{
		auto ins = typos_.emplace(Key{ orig, error, cond, dialect }, cost);
		if (!ins.second) ins.first->second = std::min(ins.first->second, cost);
	}

	void TypoTransformer::update(const TypoTransformer& o)
	{
		for (auto& p : o.typos_) addEntry(p.first.orig, p.first.error, p.second, p.first.cond, p.first.dialect);
		continualCost_ = std::min(continualCost_, o.continualCost_);
		lengtheningCost_ = std::min(lengtheningCost_, o.lengtheningCost_);
	}

	TypoTransformer TypoTransformer::withDialect(uint16_t dialect) const
	{
		TypoTransformer ret;
		for (auto& p : typos_) ret.typos_.emplace(Key{ p.first.orig, p.first.error, p.first.cond, dialect }, p.second);
		ret.continualCost_ = continualCost_; ret.lengtheningCost_ = lengtheningCost_;
		return ret;
	}

	namespace
	{
#include "typo_sets.inc"
		template<size_t N> void addRows(TypoTransformer& t, const TypoDefRow(&rows)[N])
		{
			auto split = [](const char16_t* s) { std::vector<std::u16string> v(1); for (; *s; ++s) { if (*s == u'|') v.emplace_back(); else v.back().push_back(*s); } return v; };
			for (const TypoDefRow& r : rows)
				for (const auto& o : split(r.origs)) for (const auto& e : split(r.errors)) t.add(o, e, r.cost, r.cond, 0);      // TypoTransformer::addTypos (include/kiwi/TypoTransformer.h:309-321)
		}
		TypoTransformer unite(TypoTransformer a, const TypoTransformer& b) { a.update(b); return a; }      // operator| (TypoTransformer.h:396-401)
	}

	const TypoTransformer& defaultTypoSet(int set)
	{
		enum : uint16_t { D_GANGWON = 1 << 2, D_GYEONGSANG = 1 << 3, D_JEJU = 1 << 5, D_HAMGYEONG = 1 << 7 };      // Dialect (include/kiwi/Types.h:322-335)
		static const TypoTransformer withoutTypo;
		static const TypoTransformer basic = [] { TypoTransformer t; addRows(t, kTypoBasic); return t; }();
		static const TypoTransformer continual = [] { TypoTransformer t; t.setContinualCost(1.f); addRows(t, kTypoContinual); return t; }();
		static const TypoTransformer basicWithContinual = unite(basic, continual);
		static const TypoTransformer lengthening = [] { TypoTransformer t; t.setLengtheningCost(0.25f); return t; }();
		static const TypoTransformer basicWithContinualAndLengthening = unite(basicWithContinual, lengthening);
		static const TypoTransformer dialect = []
		{
			TypoTransformer a, b, c, d;
			addRows(a, kTypoDialectJeju); addRows(b, kTypoDialectHamgyeong); addRows(c, kTypoDialectNorthEast); addRows(d, kTypoDialectGyeongsang);
			return unite(unite(unite(a.withDialect(D_JEJU), b.withDialect(D_HAMGYEONG)), c.withDialect(D_HAMGYEONG | D_GYEONGSANG | D_GANGWON)), d.withDialect(D_GYEONGSANG));
		}();
		switch (set)
		{
		case 0: return withoutTypo;
		case 1: return basic;
		case 2: return continual;
		case 3: return basicWithContinual;
		case 4: return lengthening;
		case 5: return basicWithContinualAndLengthening;
		case 6: return dialect;
		default: throw std::invalid_argument{ "Invalid `DefaultTypoSet`" };
		}
	}

	const PreparedTypo& defaultDialectTypo()
	{
		static const PreparedTypo p{ defaultTypoSet(6), true };
		return p;
	}

	void TypoTransformer::scaleCost(float scale)
	{
		if (!std::isfinite(scale) || scale <= 0) throw std::invalid_argument{ "`scale` must be positive real." };
		for (auto& p : typos_) p.second *= scale;
		if (std::isfinite(continualCost_)) continualCost_ *= scale;
		if (std::isfinite(lengtheningCost_)) lengtheningCost_ *= scale;
	}

	// ---- prepare -----------------------------------------------------------------------------------------------------------------------
	PreparedTypo::PreparedTypo(const TypoTransformer& tt, bool inverse) : continualCost_(tt.continualCost()), lengtheningCost_(tt.lengtheningCost())
	{
		struct BNode { std::map<char16_t, int> next; int fail = -1; int pat = -1; uint32_t depth = 0; };
		struct BRepl { std::u16string str; float cost; uint8_t cond; uint16_t dialect; };
		std::vector<BNode> b(1);
		std::vector<std::vector<BRepl>> lists;
		auto walk = [&](const std::u16string& s)
		{
			int n = 0;
			for (char16_t c : s)
			{
				auto it = b[n].next.find(c);
				if (it == b[n].next.end()) { b.emplace_back(); b.back().depth = b[n].depth + 1; const int id = (int)b.size() - 1; b[n].next[c] = id; n = id; }
				else n = it->second;
			}
			return n;
		};
		walk(std::u16string(1, u'\0'));      // the edge the automaton is entered through: patterns that begin with NUL match at the start of a text only
		for (auto& t : tt.rules())            // iteration order of the rule map (see typo.hpp)
		{
			const std::u16string& pat = inverse ? t.first.error : t.first.orig;
			const std::u16string& rep = inverse ? t.first.orig : t.first.error;
			if (pat == rep) continue;
			const int n = walk(pat);
			if (b[n].pat < 0) { b[n].pat = (int)lists.size(); lists.emplace_back(); }
			auto& list = lists[b[n].pat];
			bool merged = false;
			for (auto& p : list)
			{
				if (p.cond == t.first.cond && p.str == rep)
				{
					if (p.dialect == t.first.dialect) { p.cost = std::isfinite(t.second) ? std::min(p.cost, t.second) : t.second; merged = true; break; }
					else if (p.cost == t.second) { p.dialect = (uint16_t)(p.dialect | t.first.dialect); merged = true; break; }
				}
			}
			if (!merged) list.push_back(BRepl{ rep, t.second, t.first.cond, t.first.dialect });
		}
		// failure links, breadth first
		std::vector<int> order{ 0 };
		auto bstep = [&](int n, char16_t c) { auto it = b[n].next.find(c); return it == b[n].next.end() ? -1 : it->second; };
		for (size_t qi = 0; qi < order.size(); ++qi)
		{
			const int u = order[qi];
			for (auto& kv : b[u].next)
			{
				int f = b[u].fail;
				while (f >= 0 && bstep(f, kv.first) < 0) f = b[f].fail;
				b[kv.second].fail = (u == 0) ? 0 : (f >= 0 ? bstep(f, kv.first) : 0);
				order.push_back(kv.second);
			}
		}
		// flatten
		pats_.resize(lists.size());
		trie_.resize(b.size());
		for (size_t n = 0; n < b.size(); ++n)
		{
			TrieNode& t = trie_[n];
			t.edgeOff = (uint32_t)keys_.size(); t.numNexts = (uint16_t)b[n].next.size(); t.depth = (uint16_t)b[n].depth; t.fail = b[n].fail; t.pattern = b[n].pat >= 0 ? b[n].pat : -1;
			for (auto& kv : b[n].next) { keys_.push_back(kv.first); children_.push_back((uint32_t)kv.second); }      // std::map: sorted by key
			if (b[n].pat >= 0)
			{
				auto& list = lists[b[n].pat];
				Pattern& p = pats_[b[n].pat];
				p.replOff = (uint32_t)repls_.size(); p.replCnt = (uint32_t)list.size(); p.patLength = b[n].depth;
				if (!inverse && list[0].cond == TC_APPLOSIVE) p.patLength--;
				for (auto& r : list)
				{
					size_t from = 0;
					if (inverse && r.cond == TC_APPLOSIVE && !r.str.empty() && r.str[0] == 0) from = 1;
					repls_.push_back(Repl{ (uint32_t)pool_.size(), (uint32_t)(r.str.size() - from), r.cost, r.cond, 0, r.dialect });
					pool_.append(r.str, from, std::u16string::npos);
				}
			}
		}
		for (int v : order)      // "a shorter pattern ends here too" marks of the frozen trie (src/FrozenTrie.hpp:139-151)
		{
			if (v == 0 || trie_[v].pattern >= 0) continue;
			for (int f = trie_[v].fail; f > 0; f = trie_[f].fail) if (trie_[f].pattern >= 0) { trie_[v].pattern = -2; break; }
		}
		// what the device graph kernel needs beside the tables: the last-character facts of every replacement string, and its capacity bounds
		auto lastOfStr = [](const char16_t* f, size_t n, uint8_t out[2])
		{
			uint32_t lastC = 0; bool any = false;
			for (size_t j = 0; j < n; ++j)
			{
				uint32_t c32 = f[j];
				if (isHighSurrogate(c32) && j + 1 < n) { c32 = mergeSurrogate(c32, f[j + 1]); ++j; }
				lastC = c32; any = true;
			}
			out[0] = (any && lastC) ? identifySpecialChr(lastC) : (uint8_t)0xFF;
			out[1] = (any && lastC) ? chr2ScriptType(lastC) : (uint8_t)0;
		};
		replLast_.assign(6 * repls_.size(), 0);
		for (size_t i = 0; i < repls_.size(); ++i)
		{
			const char16_t* f = pool_.data() + repls_[i].strOff; const size_t n = repls_[i].strLen;
			lastOfStr(f, n, &replLast_[6 * i]); lastOfStr(f, n ? 1 : 0, &replLast_[6 * i + 2]); lastOfStr(f + (n ? 1 : 0), n ? n - 1 : 0, &replLast_[6 * i + 4]);
		}
		for (size_t v = 0; v < trie_.size(); ++v)
		{
			uint32_t per = 0;
			for (int32_t sub = (int32_t)v; sub >= 0; sub = trie_[sub].fail)
			{
				if (trie_[sub].pattern == -1) break;
				if (trie_[sub].pattern < 0) continue;
				++per;
			}
			matchesPerEnd_ = std::max(matchesPerEnd_, per);
		}
		for (auto& p : pats_)
		{
			std::vector<char16_t> firsts;
			for (uint32_t ri = 0; ri < p.replCnt; ++ri)
			{
				const Repl& r = repls_[p.replOff + ri];
				if (r.cond != TC_CONTINUAL && r.cond != TC_BOUNDARY) continue;
				const char16_t c = pool_[r.strOff];      // (as graph() reads it: the unit at strOff even of an empty replacement)
				if (std::find(firsts.begin(), firsts.end(), c) == firsts.end()) firsts.push_back(c);
			}
			maxCtiBound_ = std::max<uint32_t>(maxCtiBound_, (uint32_t)firsts.size() + 1);
		}
	}

	void PreparedTypo::lastOf(const TypoGraphNode& gn, const char16_t* str, uint8_t out[2]) const
	{
		uint32_t lastC = 0; bool any = false;
		const std::u16string f = formOf(gn, str);
		for (size_t j = 0; j < f.size(); ++j)
		{
			uint32_t c32 = f[j];
			if (isHighSurrogate(c32) && j + 1 < f.size()) { c32 = mergeSurrogate(c32, f[j + 1]); ++j; }
			lastC = c32; any = true;
		}
		out[0] = (any && lastC) ? identifySpecialChr(lastC) : (uint8_t)0xFF;
		out[1] = (any && lastC) ? chr2ScriptType(lastC) : (uint8_t)0;
	}

	int32_t PreparedTypo::step(int32_t node, char16_t c) const
	{
		const TrieNode& t = trie_[node];
		const uint16_t* kb = keys_.data() + t.edgeOff;
		const uint16_t* it = std::lower_bound(kb, kb + t.numNexts, (uint16_t)c);
		if (it == kb + t.numNexts || *it != (uint16_t)c) return -1;
		return (int32_t)children_[t.edgeOff + (it - kb)];
	}

	std::u16string PreparedTypo::formOf(const TypoGraphNode& g, const char16_t* str) const
	{
		if (g.formOff & TYPO_FORM_IN_POOL) return pool_.substr(g.formOff & ~TYPO_FORM_IN_POOL, g.formLen);
		return std::u16string{ str + g.formOff, g.formLen };
	}

	// ---- generateGraph (without pretokenized spans) ---------------------------------------------------------------------------------------
	size_t PreparedTypo::graph(const char16_t* str, size_t n, uint16_t allowedDialect, std::vector<TypoGraphNode>& outNodes) const
	{
		constexpr uint32_t npos = 0xFFFFFFFFu;
		struct Match { size_t end; int32_t pat; };
		std::vector<TypoGraphNode> temp;
		std::vector<Match> matches;
		std::vector<size_t> breakPoints;
		std::vector<std::pair<uint32_t, uint32_t>> endPosMap{ { 0, 0 } };
		size_t last = 0, maxCti = 0;
		temp.push_back(TypoGraphNode{ 0, 0, 0, 0.f, 0, 0, 0, 0, 0 });

		auto append = [&](uint32_t formOff, uint32_t formLen, size_t startPos, size_t endPos, float cost) -> bool      // appendNewNode (:594-628)
		{
			if (startPos != (size_t)-1 && endPosMap[startPos - last].first == npos) return false;
			const size_t newId = temp.size();
			TypoGraphNode nn{ formOff, formLen, (uint32_t)endPos, cost, 0, 0, 0, 0, 0 };
			nn.prevOffset = startPos == (size_t)-1 ? (uint32_t)(newId - 1) : endPosMap[startPos - last].first;
			temp.push_back(nn);
			if (nn.endPos >= endPosMap.size() + last) return true;
			auto& slot = endPosMap[nn.endPos - last];
			if (slot.first == npos) slot.first = (uint32_t)newId; else temp[slot.second].siblingOffset = (uint32_t)newId;
			slot.second = (uint32_t)newId;
			return true;
		};
		auto patStart = [&](const Match& m) { return m.end - pats_[m.pat].patLength; };

		auto insertBranch = [&]()
		{
			const size_t totStart = patStart(matches[0]), totEnd = matches.back().end;
			const auto carry = endPosMap.back();
			endPosMap.assign((totEnd - last) + 1, { npos, npos });
			endPosMap[0] = carry;
			breakPoints.clear();
			breakPoints.push_back(totStart);
			for (auto& m : matches) breakPoints.push_back(m.end);
			breakPoints.push_back(totEnd);
			std::sort(breakPoints.begin(), breakPoints.end());
			breakPoints.erase(std::unique(breakPoints.begin(), breakPoints.end()), breakPoints.end());
			std::sort(matches.begin(), matches.end(), [&](const Match& a, const Match& b) { return patStart(a) < patStart(b); });

			if (last < totStart) append((uint32_t)last, (uint32_t)(totStart - last), last, totStart, 0.f);
			for (size_t i = 1; i < breakPoints.size(); ++i) append((uint32_t)breakPoints[i - 1], (uint32_t)(breakPoints[i] - breakPoints[i - 1]), breakPoints[i - 1], breakPoints[i], 0.f);

			for (auto& m : matches)
			{
				const size_t e = m.end, s = patStart(m);
				const Pattern& P = pats_[m.pat];
				std::unordered_map<char16_t, std::pair<size_t, size_t>> contIdx;      // first replacement character -> (continual index, node of the first half)
				for (uint32_t ri = 0; ri < P.replCnt; ++ri)
				{
					const Repl& r = repls_[P.replOff + ri];
					const uint32_t rOff = r.strOff | TYPO_FORM_IN_POOL;
					if (r.dialect != 0 && !(allowedDialect & r.dialect)) continue;
					if (r.cond == TC_VOWEL) { if (s == 0 || !isSyllable(str[s - 1])) continue; }
					else if (r.cond == TC_ANY) { if (s == 0) continue; }
					else if (r.cond == TC_CONTINUAL || r.cond == TC_BOUNDARY)
					{
						if (r.cond == TC_CONTINUAL && (s == 0 || !isSyllable(str[s - 1]))) continue;
						if (r.cond == TC_CONTINUAL && !std::isfinite(continualCost_)) continue;
						const float scale = r.cond == TC_CONTINUAL ? continualCost_ : 1.f;
						auto ins = contIdx.emplace(pool_[r.strOff], std::make_pair(contIdx.size() + 1, (size_t)0));
						if (ins.second)
						{
							if (append(rOff, 1, s, (size_t)-1, r.cost * scale / 2))
							{
								temp.back().endPos = (uint32_t)e; temp.back().continualTypoIdx = (uint8_t)ins.first->second.first; temp.back().dialect = r.dialect;
								ins.first->second.second = temp.size() - 1;
								if (append(rOff + 1, r.strLen - 1, (size_t)-1, e, r.cost * scale / 2)) { temp.back().prevOffset = (uint32_t)ins.first->second.second; temp.back().dialect = r.dialect; }
							}
							else contIdx.erase(ins.first);
						}
						else if (append(rOff + 1, r.strLen - 1, (size_t)-1, e, r.cost * scale / 2)) { temp.back().prevOffset = (uint32_t)ins.first->second.second; temp.back().dialect = r.dialect; }
						continue;
					}
					else if (!leftCondMatched(str, s, r.cond)) continue;
					if (append(rOff, r.strLen, s, e, r.cost)) temp.back().dialect = r.dialect;
				}
				maxCti = std::max(maxCti, contIdx.size() + 1);
			}
			last = totEnd;
			matches.clear();
		};

		int32_t node = step(0, 0);
		for (size_t i = 0; i < n; ++i)
		{
			int32_t nx = step(node, str[i]);
			while (nx < 0)
			{
				node = trie_[node].fail;
				if (node >= 0) nx = step(node, str[i]);
				else { node = 0; break; }
			}
			if (nx < 0) continue;
			node = nx;
			if (trie_[node].pattern == -1) continue;
			const size_t endPos = i + 1;
			// a node that only carries the mark has pattern length (uint32_t)-1 in the reference: its start lies far beyond the text
			const size_t startPos = trie_[node].pattern >= 0 ? endPos - pats_[trie_[node].pattern].patLength : endPos - (size_t)0xFFFFFFFFu;
			if (!matches.empty() && matches.back().end < startPos) insertBranch();
			for (int32_t sub = node; sub >= 0; sub = trie_[sub].fail)
			{
				if (trie_[sub].pattern == -1) break;
				if (trie_[sub].pattern < 0) continue;
				matches.push_back(Match{ endPos, trie_[sub].pattern });
			}
		}
		if (!matches.empty()) insertBranch();
		{
			const auto carry = endPosMap.back();
			endPosMap.assign(1, carry);
			append((uint32_t)last, (uint32_t)(n - last), last, n + 1, 0.f);
			temp.back().endPos = (uint32_t)n;
		}
		std::vector<size_t> sortIdx(temp.size()), rev(temp.size());
		std::iota(sortIdx.begin(), sortIdx.end(), 0);
		std::stable_sort(sortIdx.begin(), sortIdx.end(), [&](size_t a, size_t b) { return temp[a].endPos < temp[b].endPos; });
		for (size_t i = 0; i < temp.size(); ++i) rev[sortIdx[i]] = i;
		outNodes.clear();
		outNodes.reserve(temp.size());
		for (size_t i = 0; i < temp.size(); ++i)
		{
			outNodes.push_back(temp[sortIdx[i]]);
			auto& g = outNodes.back();
			g.prevOffset = (uint32_t)(i - rev[g.prevOffset]);
			if (g.siblingOffset != 0) g.siblingOffset = (uint32_t)(rev[g.siblingOffset] - i);
		}
		return maxCti;
	}
}
