// TEST INFRASTRUCTURE -- CPU restatement of the reference's typo graph generation:
//   TypoTransformer (rule container)            /root/reference/src/TypoTransformer.cpp:214-373, include/kiwi/TypoTransformer.h:304-434
//   IntermediateTypoTransformer / prepare()     src/TypoTransformer.cpp:375-486
//   appendNewNode, generateGraph                src/TypoTransformer.cpp:594-628, 811-1049
// Written for this repo from the behaviour of those functions; pinned against the real translation unit through
// oracle/ref_bridge.cpp (kref_typo_*) by tests/test_typo_oracle.py.  Only tests may use it.
//
// Two things about the reference that a restatement has to reproduce and that are easy to miss:
//   * the rules live in a std::unordered_map keyed by (orig, error, leftCond, dialect) and prepare() walks it in ITERATION order, which
//     decides the order of the replacements of a pattern and so the order of the graph nodes.  The same container type with the same
//     hash (Hash<std::tuple<...>>, include/kiwi/Types.h:499-518, over std::hash) filled by the same sequence of insertions iterates
//     identically -- so this restatement keeps the rules in exactly that container;
//   * the "applosive" rule expansion iterates over a char16_t string LITERAL, i.e. over its terminating NUL as well: the 14th variant
//     has a NUL where the coda would be, and the pattern trie is entered through a NUL edge, so that variant matches at the start of
//     the text only (TypoTransformer.cpp:238-252, 390-395, 994).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

namespace korc
{
	namespace typo
	{
		using Str = std::u16string;
		enum Cond : uint8_t { C_NONE, C_ANY, C_VOWEL, C_VOCALIC, C_VOCALIC_H, C_NON_VOWEL, C_NON_VOCALIC, C_NON_VOCALIC_H, C_APPLOSIVE, C_CONTINUAL, C_BOUNDARY };

		inline bool isSyllable(char16_t c) { return 0xAC00 <= c && c < 0xD7A4; }            // include/kiwi/Utils.h:64-92
		inline bool isOnset(char16_t c) { return 0x1100 <= c && c < 0x1100 + 19; }
		inline bool isVowelJamo(char16_t c) { return 0x314F <= c && c < 0x3164; }
		inline char16_t joinOnsetVowel(size_t onset, size_t vowel) { return (char16_t)(0xAC00 + (char16_t)((onset * 21 + vowel) * 28)); }
		inline Str normalizeHangul(const Str& s)     // src/StrUtils.h:494-521 without the position table
		{
			Str o;
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
		// FeatureTestor::isMatched(begin, end, CondVowel) (src/FeatureTestor.cpp:6-60) on the prefix [0, n) of s
		inline bool leftCondMatched(const Str& s, size_t n, uint8_t cond)
		{
			if (cond == C_NONE) return true;
			if (n == 0) return false;
			if (cond == C_ANY) return true;
			const char16_t l = s[n - 1];
			if (cond == C_APPLOSIVE)
			{
				switch (l) { case 0x11A8: case 0x11A9: case 0x11AA: case 0x11AE: case 0x11B8: case 0x11B9: case 0x11BA: case 0x11BB: case 0x11BD: case 0x11BE: case 0x11BF: case 0x11C0: case 0x11C1: return true; default: return false; }
			}
			if (!(0xAC00 <= l && l <= 0xD7A4) && !(0x11A8 <= l && l <= 0x11C2)) return true;
			const bool coda = 0x11A8 <= l && l <= 0x11C2;
			switch (cond)
			{
			case C_VOCALIC_H: if (l == 0x11C2) return true; [[fallthrough]];
			case C_VOCALIC: if (l == 0x11AF) return true; [[fallthrough]];
			case C_VOWEL: return !coda;
			case C_NON_VOCALIC_H: if (l == 0x11C2) return false; [[fallthrough]];
			case C_NON_VOCALIC: if (l == 0x11AF) return false; [[fallthrough]];
			case C_NON_VOWEL: return !(0xAC00 <= l && l <= 0xD7A4);
			default: return false;
			}
		}

		struct Key
		{
			Str orig, err; uint8_t cond; uint16_t dialect;
			bool operator==(const Key& o) const { return orig == o.orig && err == o.err && cond == o.cond && dialect == o.dialect; }
		};
		struct KeyHash   // Hash<std::tuple<KString, KString, CondVowel, Dialect>> (include/kiwi/Types.h:499-518): tail first, then fold the head in
		{
			size_t operator()(const Key& k) const
			{
				size_t h = std::hash<uint16_t>{}(k.dialect);
				h ^= std::hash<uint8_t>{}(k.cond) + (h << 6) + (h >> 2);
				h ^= std::hash<Str>{}(k.err) + (h << 6) + (h >> 2);
				h ^= std::hash<Str>{}(k.orig) + (h << 6) + (h >> 2);
				return h;
			}
		};

		struct Rules   // TypoTransformer
		{
			std::unordered_map<Key, float, KeyHash> typos;
			float continualCost = INFINITY, lengtheningCost = INFINITY;

			void emplaceMin(Key&& k, float cost, bool finiteRule)
			{
				auto ins = typos.emplace(std::move(k), cost);
				if (!ins.second) ins.first->second = finiteRule ? (std::isfinite(cost) ? std::min(ins.first->second, cost) : cost) : std::min(ins.first->second, cost);
			}
			void addWithCond(const Str& orig, const Str& err, float cost, uint8_t cond, uint16_t dialect)   // TypoTransformer.cpp:224-256
			{
				if (orig == err) return;
				if (cond == C_NONE || cond == C_VOWEL || cond == C_ANY || cond == C_CONTINUAL || cond == C_BOUNDARY) emplaceMin(Key{ orig, err, cond, dialect }, cost, true);
				else if (cond == C_APPLOSIVE)
				{
					static const char16_t codas[14] = { 0x11A8, 0x11A9, 0x11AA, 0x11AE, 0x11B8, 0x11B9, 0x11BA, 0x11BB, 0x11BD, 0x11BE, 0x11BF, 0x11C0, 0x11C1, 0 };   // ... and the literal's NUL
					for (char16_t c : codas)
					{
						Str o, e;
						o.push_back(c); o += orig;
						if (c) e.push_back(c);
						e += err;
						emplaceMin(Key{ o, e, (uint8_t)(c ? C_NONE : cond), dialect }, cost, true);
					}
				}
				else throw std::invalid_argument{ "Unsupported leftCond" };
			}
			void addNormalized(const Str& orig, const Str& err, float cost, uint8_t cond, uint16_t dialect)   // :258-292
			{
				if (orig.empty() || err.empty()) throw std::invalid_argument{ "empty rule" };
				if (isOnset(orig.back()) != isOnset(err.back())) throw std::invalid_argument{ "onset mismatch" };
				if (isVowelJamo(orig[0]) != isVowelJamo(err[0])) throw std::invalid_argument{ "vowel mismatch" };
				if (isOnset(orig.back()))
				{
					Str o = orig, e = err;
					for (size_t i = 0; i < 21; ++i) { o.back() = joinOnsetVowel(orig.back() - 0x1100, i); e.back() = joinOnsetVowel(err.back() - 0x1100, i); addWithCond(o, e, cost, cond, dialect); }
				}
				else if (isVowelJamo(orig[0]))
				{
					Str o = orig, e = err;
					for (size_t i = 0; i < 19; ++i) { o[0] = joinOnsetVowel(i, orig[0] - 0x314F); e[0] = joinOnsetVowel(i, err[0] - 0x314F); addWithCond(o, e, cost, cond, dialect); }
				}
				else addWithCond(orig, err, cost, cond, dialect);
			}
			void add(const Str& orig, const Str& err, float cost, uint8_t cond, uint16_t dialect) { addNormalized(normalizeHangul(orig), normalizeHangul(err), cost, cond, dialect); }   // :294-297
			// one entry of another transformer's map, as TypoTransformer::update inserts it (:348-361)
			void addEntry(const Str& orig, const Str& err, float cost, uint8_t cond, uint16_t dialect) { emplaceMin(Key{ orig, err, cond, dialect }, cost, false); }
		};

		struct Repl { Str str; float cost; uint8_t cond; uint16_t dialect; };
		struct GraphNode { Str form; uint32_t endPos = 0; float typoCost = 0; uint32_t prevOffset = 0, siblingOffset = 0; uint8_t continualTypoIdx = 0; uint16_t dialect = 0; };

		class Prepared   // IntermediateTypoTransformer + PreparedTypoTransformer
		{
			struct Pattern { std::vector<Repl> repl; uint32_t patLength = 0; };
			struct TNode { std::map<char16_t, int> next; int fail = -1; int pat = -1; uint32_t depth = 0; bool hasSub = false; };
			std::vector<Pattern> pats;
			std::vector<TNode> trie;
			float continualCost = INFINITY, lengtheningCost = INFINITY;

			int walk(const Str& s)
			{
				int n = 0;
				for (char16_t c : s)
				{
					auto it = trie[n].next.find(c);
					if (it == trie[n].next.end()) { trie.emplace_back(); trie.back().depth = trie[n].depth + 1; const int id = (int)trie.size() - 1; trie[n].next[c] = id; n = id; }
					else n = it->second;
				}
				return n;
			}
			bool isNull(int n) const { return trie[n].pat < 0 && !trie[n].hasSub; }
			int step(int n, char16_t c) const { auto it = trie[n].next.find(c); return it == trie[n].next.end() ? -1 : it->second; }

		public:
			Prepared(const Rules& r, bool inverse) : continualCost(r.continualCost), lengtheningCost(r.lengtheningCost)
			{
				trie.emplace_back();
				walk(Str(1, u'\0'));      // the entry edge (IntermediateTypoTransformer(): patTrie.build("\0"))
				for (auto& t : r.typos)   // iteration order of the map: see the header
				{
					const Str& pat = inverse ? t.first.err : t.first.orig;
					const Str& rep = inverse ? t.first.orig : t.first.err;
					if (pat == rep) continue;
					const int n = walk(pat);
					if (trie[n].pat < 0) { trie[n].pat = (int)pats.size(); pats.emplace_back(); }
					auto& list = pats[trie[n].pat].repl;
					bool updated = false;
					for (auto& p : list)
					{
						if (p.cond == t.first.cond && p.str == rep)
						{
							if (p.dialect == t.first.dialect) { p.cost = std::isfinite(t.second) ? std::min(p.cost, t.second) : t.second; updated = true; break; }
							else if (p.cost == t.second) { p.dialect = (uint16_t)(p.dialect | t.first.dialect); updated = true; break; }
						}
					}
					if (!updated) list.push_back(Repl{ rep, t.second, t.first.cond, t.first.dialect });
				}
				for (size_t n = 0; n < trie.size(); ++n)
				{
					if (trie[n].pat < 0) continue;
					auto& p = pats[trie[n].pat];
					p.patLength = trie[n].depth;
					if (!inverse && p.repl[0].cond == C_APPLOSIVE) p.patLength--;
					if (inverse) for (auto& rr : p.repl) if (rr.cond == C_APPLOSIVE && !rr.str.empty() && rr.str[0] == 0) rr.str.erase(0, 1);
				}
				// Aho-Corasick failure links (breadth first) and the "a shorter pattern ends here too" marks of the frozen trie
				std::vector<int> order{ 0 };
				for (size_t qi = 0; qi < order.size(); ++qi)
				{
					const int u = order[qi];
					for (auto& kv : trie[u].next)
					{
						const int v = kv.second;
						int f = trie[u].fail;
						while (f >= 0 && step(f, kv.first) < 0) f = trie[f].fail;
						trie[v].fail = (u == 0) ? 0 : (f >= 0 ? step(f, kv.first) : 0);
						order.push_back(v);
					}
				}
				for (int v : order)
				{
					if (v == 0 || trie[v].pat >= 0) continue;
					for (int f = trie[v].fail; f > 0; f = trie[f].fail) if (trie[f].pat >= 0) { trie[v].hasSub = true; break; }
				}
			}

			float lengthening() const { return lengtheningCost; }

			// generateGraph (TypoTransformer.cpp:811-1049) without pretokenized spans
			std::vector<GraphNode> graph(const Str& str, uint16_t allowedDialect, size_t& maxContinualTypoIdx) const
			{
				constexpr uint32_t npos = 0xFFFFFFFFu;
				struct Match { size_t end; int pat; };
				std::vector<GraphNode> temp;
				std::vector<Match> matches;
				std::vector<size_t> breakPoints;
				std::vector<std::pair<uint32_t, uint32_t>> endPosMap{ { 0, 0 } };
				size_t last = 0;
				temp.emplace_back();

				// appendNewNode (:594-628)
				auto append = [&](const Str& form, size_t startPos, size_t endPos, float cost) -> bool
				{
					if (startPos != (size_t)-1 && endPosMap[startPos - last].first == npos) return false;
					const size_t newId = temp.size();
					temp.emplace_back();
					GraphNode& nn = temp.back();
					nn.form = form; nn.endPos = (uint32_t)endPos; nn.typoCost = cost;
					nn.prevOffset = startPos == (size_t)-1 ? (uint32_t)(newId - 1) : endPosMap[startPos - last].first;
					if (nn.endPos >= endPosMap.size() + last) return true;
					auto& slot = endPosMap[nn.endPos - last];
					if (slot.first == npos) slot.first = (uint32_t)newId; else temp[slot.second].siblingOffset = (uint32_t)newId;
					slot.second = (uint32_t)newId;
					return true;
				};
				auto patStart = [&](const Match& m) { return m.end - pats[m.pat].patLength; };

				auto insertBranch = [&]()
				{
					const size_t totStart = patStart(matches[0]), totEnd = matches.back().end;
					const auto v = endPosMap.back();
					const size_t base = last;      // (`append` reads `last`; it only changes at the end of this function)
					endPosMap.assign((totEnd - base) + 1, { npos, npos });
					endPosMap[0] = v;
					breakPoints.clear();
					breakPoints.push_back(totStart);
					for (auto& m : matches) breakPoints.push_back(m.end);
					breakPoints.push_back(totEnd);
					std::sort(breakPoints.begin(), breakPoints.end());
					breakPoints.erase(std::unique(breakPoints.begin(), breakPoints.end()), breakPoints.end());
					std::sort(matches.begin(), matches.end(), [&](const Match& a, const Match& b) { return patStart(a) < patStart(b); });

					if (last < totStart) append(str.substr(last, totStart - last), last, totStart, 0.f);
					for (size_t i = 1; i < breakPoints.size(); ++i) append(str.substr(breakPoints[i - 1], breakPoints[i] - breakPoints[i - 1]), breakPoints[i - 1], breakPoints[i], 0.f);

					for (auto& m : matches)
					{
						const size_t e = m.end, s = patStart(m);
						std::unordered_map<char16_t, std::pair<size_t, size_t>> contIdx;      // first replacement char -> (index, node of the first half)
						for (auto& repl : pats[m.pat].repl)
						{
							if (repl.dialect != 0 && !(allowedDialect & repl.dialect)) continue;
							if (repl.cond == C_VOWEL) { if (s == 0 || !isSyllable(str[s - 1])) continue; }
							else if (repl.cond == C_ANY) { if (s == 0) continue; }
							else if (repl.cond == C_CONTINUAL || repl.cond == C_BOUNDARY)
							{
								if (repl.cond == C_CONTINUAL && (s == 0 || !isSyllable(str[s - 1]))) continue;
								if (repl.cond == C_CONTINUAL && !std::isfinite(continualCost)) continue;
								const float scale = repl.cond == C_CONTINUAL ? continualCost : 1.f;
								auto ins = contIdx.emplace(repl.str[0], std::make_pair(contIdx.size() + 1, (size_t)0));
								if (ins.second)
								{
									if (append(repl.str.substr(0, 1), s, (size_t)-1, repl.cost * scale / 2))
									{
										temp.back().endPos = (uint32_t)e; temp.back().continualTypoIdx = (uint8_t)ins.first->second.first; temp.back().dialect = repl.dialect;
										ins.first->second.second = temp.size() - 1;
										if (append(repl.str.substr(1), (size_t)-1, e, repl.cost * scale / 2)) { temp.back().prevOffset = (uint32_t)ins.first->second.second; temp.back().dialect = repl.dialect; }
									}
									else contIdx.erase(ins.first);
								}
								else if (append(repl.str.substr(1), (size_t)-1, e, repl.cost * scale / 2)) { temp.back().prevOffset = (uint32_t)ins.first->second.second; temp.back().dialect = repl.dialect; }
								continue;
							}
							else if (!leftCondMatched(str, s, repl.cond)) continue;
							if (append(repl.str, s, e, repl.cost)) temp.back().dialect = repl.dialect;
						}
						maxContinualTypoIdx = std::max(maxContinualTypoIdx, contIdx.size() + 1);
					}
					last = totEnd;
					matches.clear();
				};

				int node = step(0, 0);      // entered through the NUL edge
				for (size_t i = 0; i < str.size(); ++i)
				{
					int nn = step(node, str[i]);
					while (nn < 0)
					{
						node = trie[node].fail;
						if (node >= 0) nn = step(node, str[i]);
						else { node = 0; break; }
					}
					if (nn < 0) continue;
					node = nn;
					if (isNull(node)) continue;
					const size_t endPos = i + 1;
					// (a node that only carries the "shorter pattern ends here" mark has patLength (uint32_t)-1 in the reference: start = far beyond the text)
					const size_t startPos = trie[node].pat >= 0 ? endPos - pats[trie[node].pat].patLength : endPos - (size_t)0xFFFFFFFFu;
					if (!matches.empty() && matches.back().end < startPos) insertBranch();
					for (int sub = node; sub >= 0; sub = trie[sub].fail)
					{
						if (isNull(sub)) break;
						if (trie[sub].pat < 0) continue;
						matches.push_back(Match{ endPos, trie[sub].pat });
					}
				}
				if (!matches.empty()) insertBranch();
				{
					const auto v = endPosMap.back();
					endPosMap.assign(1, v);
					append(str.substr(last), last, str.size() + 1, 0.f);
					temp.back().endPos = (uint32_t)str.size();
				}
				std::vector<size_t> sortIdx(temp.size()), rev(temp.size());
				std::iota(sortIdx.begin(), sortIdx.end(), 0);
				std::stable_sort(sortIdx.begin(), sortIdx.end(), [&](size_t a, size_t b) { return temp[a].endPos < temp[b].endPos; });
				for (size_t i = 0; i < temp.size(); ++i) rev[sortIdx[i]] = i;
				std::vector<GraphNode> out;
				out.reserve(temp.size());
				for (size_t i = 0; i < temp.size(); ++i)
				{
					out.push_back(temp[sortIdx[i]]);
					auto& n = out.back();
					n.prevOffset = (uint32_t)(i - rev[n.prevOffset]);
					if (n.siblingOffset != 0) n.siblingOffset = (uint32_t)(rev[n.siblingOffset] - i);
				}
				return out;
			}
		};
	}
}
