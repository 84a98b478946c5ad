// Host-side model baker: raw model (what the reference's KiwiBuilder holds before build()) -> FlatModel.
// Reproduces the observable result of KiwiBuilder::build() for a typo-free, standard-dialect build
// (/root/reference/src/KiwiBuilder.cpp:2385-2640): form ordering and formHash (:2445-2455), per-form
// reductions (:2478-2507), the form trie in creation order (include/kiwi/Trie.hpp:444-470) frozen into an
// Aho-Corasick automaton (src/FrozenTrie.hpp:95-154), and the Knlm load-time expansion
// (src/Knlm.hpp:1003-1167).  Everything is index based; nothing here is shared with oracle/_ref.
#include <sys/stat.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <numeric>
#include <stdexcept>

#include "flat_model.hpp"
#include "raw_model.hpp"
#include "hostutil.hpp"
#include "feature.hpp"

namespace kamd
{
	namespace
	{
		struct BuildNode
		{
			std::map<uint16_t, uint32_t> next;
			int32_t val = TRIE_NONE;
			uint16_t depth = 0;
		};

		uint8_t reduceVowel(uint8_t v, uint8_t mv) // KiwiBuilder.cpp:2258-2278
		{
			if (v == mv) return v;
			if (CV_VOWEL <= v && v <= CV_VOCALIC_H)
			{
				if (CV_VOWEL <= mv && mv <= CV_VOCALIC_H) return std::max(v, mv);
				return CV_NONE;
			}
			if (CV_NON_VOWEL <= v && v <= CV_NON_VOCALIC_H)
			{
				if (CV_NON_VOWEL <= mv && mv <= CV_NON_VOCALIC_H) return std::min(v, mv);
				return CV_NONE;
			}
			return CV_NONE;
		}

		struct KnlmHeader // include/kiwi/Knlm.h:9-15
		{
			uint64_t num_nodes, node_offset, key_offset, ll_offset, gamma_offset, qtable_offset, htx_offset;
			uint64_t unk_id, bos_id, eos_id, vocab_size;
			uint8_t order, key_size, diff_size, quantized;
			uint32_t extra_buf_size;
		};

		bool lmSearch(const FlatModel& m, const LmNodeRec& n, uint32_t key, int32_t& v)
		{
			const uint32_t* k = m.lmKeys.data() + n.nextOff;
			const uint32_t* e = k + n.numNexts;
			const uint32_t* it = std::lower_bound(k, e, key);
			if (it == e || *it != key) return false;
			v = m.lmValues[n.nextOff + (it - k)];
			return true;
		}

		float lmValueAsFloat(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }

		// KnLangModel::progress (src/Knlm.cpp:44-130), host copy used only while baking (bos node)
		float lmProgressHost(const FlatModel& m, int32_t& node, uint32_t next)
		{
			float acc = 0;
			for (;;)
			{
				int32_t v;
				const LmNodeRec* n = &m.lmNodes[node];
				// a history-transformed model (Knlm.cpp:61-70, 116-126): when no context continues with `next`, the new state is the root's child for the
				// TRANSFORMED id (the oldest token of a trie path is stored transformed, src/count.hpp:148-240)
				auto htxRoot = [&](uint32_t w) -> int32_t { if (m.lmHtx.empty()) return 0; const uint32_t t = m.lmHtx[w]; return t < m.lmRoot.size() ? m.lmRoot[t] : 0; };
				if (node == 0)
				{
					v = m.lmRoot[next];
					if (v == 0) { node = htxRoot(next); return acc + m.h.unkLl; }
				}
				else if (!lmSearch(m, *n, next, v))
				{
					acc += n->gamma;
					node += n->lower;
					continue;
				}
				if (v > 0) { node += v; return acc + m.lmNodes[node].ll; }
				int32_t cur = node;
				while (m.lmNodes[cur].lower)
				{
					cur += m.lmNodes[cur].lower;
					int32_t lv;
					if (lmSearch(m, m.lmNodes[cur], next, lv) && lv > 0) { node = cur + lv; return acc + lmValueAsFloat(v); }
				}
				node = htxRoot(next);
				return acc + lmValueAsFloat(v);
			}
		}

		float lmGetLL(const FlatModel& m, int32_t node, uint32_t next) // src/Knlm.cpp:10-42
		{
			int32_t v;
			const LmNodeRec& n = m.lmNodes[node];
			if (node == 0)
			{
				v = m.lmRoot[next];
				if (v == 0) return m.h.unkLl;
			}
			else if (!lmSearch(m, n, next, v))
			{
				if (!n.lower) throw std::runtime_error{ "knlm: the start context of a history-transformed model does not hold the unknown word (the reference's loader recurses forever on such a file)" };
				return n.gamma + lmGetLL(m, node + n.lower, next);
			}
			if (v > 0) return m.lmNodes[node + v].ll;
			return lmValueAsFloat(v);
		}

		void loadKnlm(FlatModel& m, const uint8_t* blob, size_t size)
		{
			if (size < sizeof(KnlmHeader)) throw std::runtime_error{ "knlm: truncated header" };
			KnlmHeader hd;
			std::memcpy(&hd, blob, sizeof(hd));
			if (hd.key_size != 2 && hd.key_size != 4) throw std::runtime_error{ "knlm: unsupported key size" };
			const size_t nNodes = hd.num_nodes;
			const uint32_t qbits = hd.quantized & 0x1F; const bool compressed = (hd.quantized & 0x80) != 0;      // Knlm.hpp:1008-1009
			if (qbits > 16) throw std::runtime_error{ "16+ bits quantization not supported." };
			auto keyAt = [&](uint64_t off, size_t i) -> uint32_t
			{
				if (hd.key_size == 2) { uint16_t v; std::memcpy(&v, blob + off + 2 * i, 2); return v; }
				uint32_t v; std::memcpy(&v, blob + off + 4 * i, 4); return v;
			};
			// node sizes: plain keys, or -- compressed -- the variable-length code qe::QCode<0, 2, 8, 16> (src/QEncoder.hpp): two header bits per value
			// name its class {0 bits: 0, 2 bits: 1..4, 8 bits: 5..260, 16 bits: 261..}, the bodies form one LSB-first bit stream of 64-bit words that
			// starts right behind the header bytes (Knlm.hpp:1016-1023; the reference decodes into 16-bit slots: 16-bit keys only)
			std::vector<uint32_t> nodeSize(nNodes);
			if (compressed)
			{
				if (hd.key_size != 2) throw std::runtime_error{ "knlm: compressed node sizes need 16-bit keys" };
				const uint8_t* qh = blob + hd.node_offset;
				const uint8_t* qb = qh + (nNodes + 3) / 4;
				if (qb > blob + size) throw std::runtime_error{ "knlm: truncated node section" };
				static const uint32_t qBits[4] = { 0, 2, 8, 16 }, qBias[4] = { 0, 1, 5, 261 };
				uint64_t bitPos = 0;
				for (size_t i = 0; i < nNodes; ++i)
				{
					const uint32_t q = (qh[i / 4] >> (2 * (i % 4))) & 3;
					uint32_t e = 0;
					if (qBits[q])
					{
						if (qb + (bitPos + qBits[q] + 7) / 8 > blob + size) throw std::runtime_error{ "knlm: truncated node section" };
						for (uint32_t k = 0; k < qBits[q]; ++k) e |= (uint32_t)((qb[(bitPos + k) >> 3] >> ((bitPos + k) & 7)) & 1) << k;
						bitPos += qBits[q];
					}
					nodeSize[i] = e + qBias[q];
				}
			}
			else for (size_t i = 0; i < nNodes; ++i) nodeSize[i] = keyAt(hd.node_offset, i);
			size_t nonLeaf = 0, leaf = 0;
			for (size_t i = 0; i < nNodes; ++i) (nodeSize[i] ? nonLeaf : leaf)++;
			const size_t nKeys = (hd.ll_offset - hd.key_offset) / hd.key_size;
			m.lmKeys.resize(nNodes ? nNodes - 1 : 0);
			if (nKeys < m.lmKeys.size()) throw std::runtime_error{ "knlm: key section too small" };
			for (size_t i = 0; i < m.lmKeys.size(); ++i) m.lmKeys[i] = keyAt(hd.key_offset, i);
			// log-likelihoods and back-off weights: floats, or -- quantised -- fixed-width codes into two tables of 2^bits floats at qtable_offset
			// (lm::FixedLengthEncoder<bits, uint32_t>, src/BitEncoder.hpp: a plain LSB-first bit stream; one stream for the non-leaf then the leaf
			// log-likelihoods, one for the back-off weights; Knlm.hpp:398-455, 1036-1061)
			std::vector<float> llV(nonLeaf + leaf), gammaV(nonLeaf);
			if (qbits)
			{
				const size_t tab = (size_t)1 << qbits;
				if (hd.qtable_offset + 8 * tab > size) throw std::runtime_error{ "knlm: truncated quantisation tables" };
				const float* llTable = (const float*)(blob + hd.qtable_offset);
				const float* gammaTable = llTable + tab;
				auto code = [&](uint64_t off, uint64_t limit, size_t i) -> uint32_t
				{
					const uint64_t bp = (uint64_t)i * qbits;
					if (off + (bp + qbits + 7) / 8 > limit) throw std::runtime_error{ "knlm: truncated quantised section" };
					uint32_t v = 0;
					for (uint32_t k = 0; k < qbits; ++k) v |= (uint32_t)((blob[off + ((bp + k) >> 3)] >> ((bp + k) & 7)) & 1) << k;
					return v;
				};
				for (size_t i = 0; i < nonLeaf + leaf; ++i) llV[i] = llTable[code(hd.ll_offset, hd.gamma_offset, i)];
				for (size_t i = 0; i < nonLeaf; ++i) gammaV[i] = gammaTable[code(hd.gamma_offset, hd.qtable_offset, i)];
			}
			else
			{
				if (hd.ll_offset + 4 * (nonLeaf + leaf) > size || hd.gamma_offset + 4 * nonLeaf > size) throw std::runtime_error{ "knlm: truncated float sections" };
				std::memcpy(llV.data(), blob + hd.ll_offset, 4 * (nonLeaf + leaf));
				std::memcpy(gammaV.data(), blob + hd.gamma_offset, 4 * nonLeaf);
			}
			const float* ll = llV.data();
			const float* gamma = gammaV.data();
			const float* leafLl = ll + nonLeaf;

			// history transformer (Knlm.hpp:1070-1076): [vocab] keys; the root's direct table then spans the transformed ids too
			m.lmHtx.clear(); m.lmHtxNode.clear();
			size_t rootSize = hd.vocab_size;
			if (hd.htx_offset)
			{
				if (hd.htx_offset + (size_t)hd.key_size * hd.vocab_size > size) throw std::runtime_error{ "knlm: truncated history transformer" };
				m.lmHtx.resize(hd.vocab_size);
				for (size_t i = 0; i < hd.vocab_size; ++i) { m.lmHtx[i] = keyAt(hd.htx_offset, i); rootSize = std::max<size_t>(rootSize, (size_t)m.lmHtx[i] + 1); }
			}
			m.lmNodes.assign(nonLeaf, LmNodeRec{});
			m.lmValues.assign(nNodes - 1, 0);
			m.lmRoot.assign(rootSize, 0);
			// pre-order node stream -> non-leaf node table + per-edge values (Knlm.hpp:1089-1122)
			struct Range { size_t node, cur, end; };
			std::vector<Range> st;
			size_t ni = 0, li = 0, nextOff = 0;
			for (size_t i = 0; i < nNodes; ++i)
			{
				const uint32_t sz = nodeSize[i];
				if (sz)
				{
					if (!st.empty()) m.lmValues[st.back().cur] = (int32_t)(ni - st.back().node);
					auto& n = m.lmNodes[ni];
					n.numNexts = sz; n.nextOff = (uint32_t)nextOff; n.ll = ll[ni]; n.gamma = gamma[ni];
					st.push_back(Range{ ni, nextOff, nextOff + sz });
					nextOff += sz;
					++ni;
				}
				else
				{
					if (st.empty()) throw std::runtime_error{ "knlm: malformed node stream" };
					std::memcpy(&m.lmValues[st.back().cur], &leafLl[li], 4);
					st.back().cur++;
					while (st.back().cur == st.back().end)
					{
						st.pop_back();
						if (st.empty()) break;
						st.back().cur++;
					}
					++li;
				}
			}
			for (uint32_t i = 0; i < m.lmNodes[0].numNexts; ++i) if (m.lmKeys[i] < rootSize) m.lmRoot[m.lmKeys[i]] = m.lmValues[i];

			m.h.nLmNodes = (uint32_t)nonLeaf;
			m.h.nLmEdges = (uint32_t)m.lmKeys.size();
			m.h.lmOrder = hd.order;
			m.h.lmKeyBytes = hd.key_size;
			m.h.unkLl = 0;
			if (!m.lmHtx.empty())
			{
				// Knlm.hpp:1138-1145: the unknown word's score is read in the state after <s> (links and unk_ll still zero), the start state is entered
				// through the TRANSFORMED <s>
				int32_t nd = 0;
				lmProgressHost(m, nd, (uint32_t)hd.bos_id);
				m.h.unkLl = lmGetLL(m, nd, (uint32_t)hd.unk_id);
				int32_t bos0 = 0;
				lmProgressHost(m, bos0, m.lmHtx[hd.bos_id]);
				m.h.bosNode = bos0;
			}
			else m.h.unkLl = lmGetLL(m, 0, (uint32_t)hd.unk_id);   // Knlm.hpp:1147
			// suffix ("lower") links by BFS (Knlm.hpp:38-63, 1153-1166)
			std::deque<uint32_t> dq{ 0u };
			while (!dq.empty())
			{
				const uint32_t p = dq.front(); dq.pop_front();
				const LmNodeRec pn = m.lmNodes[p];
				for (uint32_t i = 0; i < pn.numNexts; ++i)
				{
					const int32_t v = m.lmValues[pn.nextOff + i];
					if (v <= 0) continue;
					uint32_t k = m.lmKeys[pn.nextOff + i];
					const uint32_t child = p + v;
					uint32_t node = p;
					while (m.lmNodes[node].lower)
					{
						const uint32_t low = node + m.lmNodes[node].lower;
						if (low == 0 && !m.lmHtx.empty()) k = k < m.lmHtx.size() ? m.lmHtx[k] : 0;      // findLowerNode, Knlm.hpp:43-46: the root's children are keyed by transformed ids
						int32_t found;
						if (lmSearch(m, m.lmNodes[low], k, found)) { node = low + found; goto done; }
						node = low;
					}
				done:
					m.lmNodes[child].lower = (int32_t)node - (int32_t)child;
					dq.push_back(child);
				}
			}
			// device lookup structures: edge hash, root table with the child's ll, per-node back-off record
			m.lmBackoff.resize(nonLeaf);
			for (size_t i = 0; i < nonLeaf; ++i) m.lmBackoff[i] = LmBackoff{ m.lmNodes[i].lower, m.lmNodes[i].gamma };
			m.lmRoot2.assign(rootSize, LmRootRec{ 0, 0.f });
			if (!m.lmHtx.empty())
			{
				m.lmHtxNode.assign(rootSize, 0);
				for (size_t w = 0; w < m.lmHtx.size(); ++w) { const uint32_t t = m.lmHtx[w]; m.lmHtxNode[w] = t < rootSize ? m.lmRoot[t] : 0; }
			}
			auto edgeLl = [&](uint32_t node, int32_t v) { return v > 0 ? m.lmNodes[node + v].ll : lmValueAsFloat(v); };
			for (uint32_t i = 0; i < m.lmNodes[0].numNexts; ++i) if (m.lmKeys[i] < rootSize) m.lmRoot2[m.lmKeys[i]] = LmRootRec{ m.lmValues[i], edgeLl(0, m.lmValues[i]) };
			{
				const size_t nEdges = m.lmKeys.size() - m.lmNodes[0].numNexts;
				size_t nBuckets = 1;
				// 4 slots per bucket, at most one edge per two buckets on average: a bucket is full -- and a lookup that ends in it has to go on to the
				// next one -- with probability < 0.2 % (Poisson, mean <= 0.5).  One such lane sends its whole wavefront through the general walk's loop,
				// so the table is sized for that to be rare (at mean 2 it was 4 % of the lookups); 128 - 256 bytes of HBM per edge
				// (KAMD_LM_HASH_LOAD=<edges per bucket, 1..4>: a denser table for a model whose n-gram count makes 128 - 256 B per edge too much memory)
				size_t perTwo = 1;
				if (const char* e = std::getenv("KAMD_LM_HASH_LOAD")) perTwo = (size_t)std::min(8, std::max(1, 2 * std::atoi(e)));
				while (nBuckets * perTwo < 2 * (nEdges + 1)) nBuckets <<= 1;
				if (nBuckets > (size_t(1) << 32)) throw std::runtime_error{ "language model: the edge table would need more than 2^32 buckets (" + std::to_string(nEdges) + " edges)" };
				m.lmHashMask = (uint32_t)(nBuckets - 1);
				m.lmHash.assign(nBuckets * 4, LmSlot{ LM_SLOT_EMPTY, LM_SLOT_EMPTY, 0, 0.f });
				if (std::getenv("KAMD_LM_STATS"))      // developer aid: the shape of the language model's trie
				{
					size_t hist[12] = {}; size_t edgesIn[12] = {};
					for (uint32_t nd = 1; nd < nonLeaf; ++nd) { uint32_t k = m.lmNodes[nd].numNexts, b = 0; while ((1u << b) < k + 1 && b < 11) ++b; ++hist[b]; edgesIn[b] += k; }
					fprintf(stderr, "[lm stats] unigrams %u, non-leaf nodes %zu, edges below the root's %zu, hash buckets %zu (%.1f MB), back-off records %.1f MB; nodes (edges) by fan-out < 2^b:", (unsigned)m.lmNodes[0].numNexts, nonLeaf, nEdges, nBuckets, nBuckets * 64e-6, m.lmBackoff.size() * 8e-6);
					for (int b = 0; b < 12; ++b) fprintf(stderr, " %d:%zu(%zu)", b, hist[b], edgesIn[b]);
					fprintf(stderr, "\n");
				}
				for (uint32_t nd = 1; nd < nonLeaf; ++nd)
				{
					const LmNodeRec& r = m.lmNodes[nd];
					for (uint32_t e = 0; e < r.numNexts; ++e)
					{
						const uint32_t wid = m.lmKeys[r.nextOff + e];
						const int32_t v = m.lmValues[r.nextOff + e];
						uint32_t b = lmHashOf(nd, wid) & m.lmHashMask;
						for (;;)
						{
							LmSlot* s = &m.lmHash[(size_t)b * 4];
							int k = 0;
							while (k < 4 && s[k].node != LM_SLOT_EMPTY) ++k;
							if (k < 4) { s[k] = LmSlot{ nd, wid, v, edgeLl(nd, v) }; break; }
							b = (b + 1) & m.lmHashMask;
						}
					}
				}
			}
			if (m.lmHtx.empty())
			{
				int32_t bos = 0;
				lmProgressHost(m, bos, (uint32_t)hd.bos_id);  // Knlm.hpp:1148-1149 (links are still zero there too)
				m.h.bosNode = bos;
			}
		}
	}

	namespace
	{
		// ---- CoNgram model blob (reference cong.mdl, loader src/CoNgramModel.cpp:425-789): local (window 0) use of an 8-bit model ------------
		// header 64 B (include/kiwi/CoNgramModel.h:18-34) | node sizes, Stream VByte "0124" | keys, Stream VByte | values, Stream VByte "0124" |
		// per context: dim x s8, fp16 scale, fp16 -bias [, 2 x fp16 when the model has a window] | per word: dim x s8, fp16 scale.
		// Stream VByte (Lemire, Kurz, Rupp 2017): ceil(n/4) control bytes of four 2-bit length codes, then the significant bytes, little endian;
		// codes 0..3 = 1, 2, 3, 4 bytes ("0124": 0, 1, 2, 4 bytes).
		size_t svbDecode(const uint8_t* in, const uint8_t* end, uint32_t* out, size_t n, bool v0124)
		{
			static const uint8_t lenStd[4] = { 1, 2, 3, 4 }, len0124[4] = { 0, 1, 2, 4 };
			const uint8_t* key = in; const uint8_t* data = in + (n + 3) / 4;
			if (data > end) throw std::runtime_error{ "cong.mdl: truncated Stream VByte section" };
			for (size_t i = 0; i < n; ++i)
			{
				const uint32_t code = (key[i / 4] >> ((i % 4) * 2)) & 3u;
				const uint32_t nb = v0124 ? len0124[code] : lenStd[code];
				if (data + nb > end) throw std::runtime_error{ "cong.mdl: truncated Stream VByte section" };
				uint32_t v = 0;
				for (uint32_t b = 0; b < nb; ++b) v |= (uint32_t)(*data++) << (8 * b);
				out[i] = v;
			}
			return (size_t)(data - in);
		}
		float halfToFloat(uint16_t h)      // CoNgramModel.cpp:344-354 (normal numbers only, as there)
		{
			uint32_t u = (uint32_t)(h & 0x8000) << 16;
			u |= ((uint32_t)(h & 0x7FFF) + 0x1C000) << 13;
			float f; std::memcpy(&f, &u, 4);
			return f;
		}
		bool congSearch(const FlatModel& m, const CongNodeRec& n, uint32_t key, int32_t& v)
		{
			const uint32_t* k = m.congKeys.data() + n.nextOff;
			const uint32_t* it = std::lower_bound(k, k + n.numNexts, key);
			if (it == k + n.numNexts || *it != key) return false;
			v = m.congValues[n.nextOff + (it - k)];
			return v != 0;
		}

		void loadCong(FlatModel& m, const uint8_t* blob, size_t size)
		{
			struct Header { uint64_t vocabSize, contextSize; uint16_t dim, flags; uint8_t keySize, windowSize, qbit, qgroup; uint64_t numNodes, nodeOffset, keyOffset, valueOffset, embOffset; };
			static_assert(sizeof(Header) == 64, "CoNgramModelHeader");
			if (size < sizeof(Header)) throw std::runtime_error{ "cong.mdl: truncated header" };
			Header hd; std::memcpy(&hd, blob, sizeof(hd));
			if (hd.qbit != 8 && !(hd.qbit == 4 && (hd.qgroup == 4 || hd.qgroup == 8 || hd.qgroup == 16) && hd.dim % 16 == 0)) throw std::runtime_error{ "cong.mdl: unsupported embedding packing (8-bit, or 4-bit in groups of 4 / 8 / 16, are)" };
			if (hd.flags != 0) throw std::runtime_error{ "cong.mdl: output bias / reordered vocabulary / trie frequencies (the sections of a character model) are not supported by this loader" };
			if (hd.keySize != 4 && hd.keySize != 3 && hd.keySize != 2) throw std::runtime_error{ "cong.mdl: only 16-bit, variable-length 16-bit and 32-bit keys are supported by this loader" };
			if (hd.dim == 0 || hd.dim % 4 || hd.numNodes < 1) throw std::runtime_error{ "cong.mdl: bad header" };
			// keySize 3: 16-bit trie keys, a token id >= 63488 is spelt as two of them (CoNgramModel::progressContextNode, src/CoNgramModel.hpp:271-300)
			m.congVlTMax = hd.keySize == 3 ? 65536u - 2048u : 0xFFFFFFFFu; m.congVlBits = hd.keySize == 3 ? 10u : 0u;
			const size_t rootSize = std::max<size_t>(hd.vocabSize, hd.keySize == 3 ? 65536u : 0u);
			const uint8_t* end = blob + size;
			const size_t nNodes = hd.numNodes;
			std::vector<uint32_t> sizes(nNodes), keys(nNodes - 1), values(nNodes);
			svbDecode(blob + hd.nodeOffset, end, sizes.data(), nNodes, true);
			svbDecode(blob + hd.keyOffset, end, keys.data(), nNodes - 1, false);
			svbDecode(blob + hd.valueOffset, end, values.data(), nNodes, true);
			size_t nonLeaf = 0;
			for (auto sz : sizes) nonLeaf += sz ? 1 : 0;
			m.congNodes.assign(nonLeaf, CongNodeRec{});
			m.congKeys = keys;
			m.congValues.assign(nNodes - 1, 0);
			m.congRoot.assign(rootSize, 0);
			// pre-order node stream -> non-leaf node table + per-edge values (CoNgramModel.cpp:489-523)
			struct Range { size_t node, cur, end; };
			std::vector<Range> st;
			size_t ni = 0, nextOff = 0;
			for (size_t i = 0; i < nNodes; ++i)
			{
				if (sizes[i])
				{
					if (!st.empty()) m.congValues[st.back().cur] = (int32_t)(ni - st.back().node);
					CongNodeRec& n = m.congNodes[ni];
					n.value = values[i]; n.numNexts = sizes[i]; n.nextOff = (uint32_t)nextOff;
					st.push_back(Range{ ni, nextOff, nextOff + sizes[i] });
					nextOff += sizes[i];
					++ni;
				}
				else
				{
					if (st.empty()) throw std::runtime_error{ "cong.mdl: malformed node stream" };
					m.congValues[st.back().cur] = -(int32_t)values[i];
					st.back().cur++;
					while (st.back().cur == st.back().end)
					{
						st.pop_back();
						if (st.empty()) break;
						st.back().cur++;
					}
				}
			}
			for (uint32_t i = 0; i < m.congNodes[0].numNexts; ++i) if (keys[i] < rootSize) m.congRoot[keys[i]] = m.congValues[i];
			// suffix links and inherited context ids, breadth first (CoNgramModel.cpp:547-568 with findLowerNode / findLowerValue, CoNgramModel.hpp:181-227)
			std::deque<uint32_t> dq{ 0u };
			while (!dq.empty())
			{
				const uint32_t p = dq.front(); dq.pop_front();
				const CongNodeRec pn = m.congNodes[p];
				for (uint32_t i = 0; i < pn.numNexts; ++i)
				{
					const int32_t v = m.congValues[pn.nextOff + i];
					if (v <= 0) continue;
					const uint32_t k = m.congKeys[pn.nextOff + i];
					const uint32_t child = p + v;
					// findLowerNode(p, k)
					uint32_t node = p, lowerNode;
					for (;;)
					{
						if (!m.congNodes[node].lower) { lowerNode = node; break; }
						const uint32_t low = node + m.congNodes[node].lower;
						int32_t found;
						if (congSearch(m, m.congNodes[low], k, found) && found > 0) { lowerNode = low + found; break; }
						node = low;
					}
					m.congNodes[child].lower = (int32_t)lowerNode - (int32_t)child;
					if (m.congNodes[child].value == 0)
					{
						// findLowerValue(p, k)
						uint32_t nd = p; uint32_t val = 0; bool done = false;
						while (m.congNodes[nd].lower)
						{
							const uint32_t low = nd + m.congNodes[nd].lower;
							int32_t found;
							if (congSearch(m, m.congNodes[low], k, found)) { val = found >= 0 ? m.congNodes[low + found].value : (uint32_t)(-found); done = true; break; }
							nd = low;
						}
						if (!done) val = m.congNodes[nd].value;
						m.congNodes[child].value = val;
					}
					dq.push_back(child);
				}
			}
			// embeddings
			m.congDim = hd.dim; m.congCtx = (uint32_t)hd.contextSize; m.congVocab = (uint32_t)hd.vocabSize;
			const size_t stride = (size_t)hd.dim + 8;
			m.congCtxEmb.assign(hd.contextSize * stride, 0); m.congOutEmb.assign(hd.vocabSize * stride, 0);
			const uint8_t* e = blob + hd.embOffset;
			// a row: qbit 8 = dim x s8 + fp16 scale; qbit 4 = dim / 2 bytes of nibble pairs + fp16 global scale + dim / qgroup local bytes, requantised to
			// s8 at load time as the reference does (requantizePackedInts, src/CoNgramModel.cpp:378-400)
			const size_t rowRec = hd.qbit == 8 ? (size_t)hd.dim + 2 : (size_t)hd.dim / 2 + 2 + hd.dim / hd.qgroup;
			// a file of the global model (windowSize 7; the reference's builder always writes one) carries on top: per context fp16 confidence + fp16 valid-token
			// sum; after the output rows the distant rows (per word: row, fp16 -bias, fp16 confidence), windowSize fp16 position confidences and the
			// distant-token mask (CoNgramModel.cpp:640-762)
			if (hd.windowSize != 0 && hd.windowSize != 7) throw std::runtime_error{ "cong.mdl: unsupported window size (0 and 7 are)" };
			const size_t ctxRec = rowRec + 2 + (hd.windowSize > 0 ? 4 : 0), outRec = rowRec, distRec = rowRec + 4;
			const size_t globalBytes = hd.windowSize ? hd.vocabSize * distRec + 2 * (size_t)hd.windowSize + (hd.vocabSize + 7) / 8 : 0;
			if (e + hd.contextSize * ctxRec + hd.vocabSize * outRec + globalBytes > end) throw std::runtime_error{ "cong.mdl: truncated embeddings" };
			m.congWindow = hd.windowSize; m.congKeyBytes = hd.keySize == 2 ? 2 : 4;
			if (hd.windowSize) m.congCtxConf.assign(hd.contextSize * 2, 0.f);
			auto readRow = [&](const uint8_t* src, uint8_t* o)      // -> the row's s8 values and its fp32 scale at o[dim]
			{
				float scale;
				if (hd.qbit == 8)
				{
					std::memcpy(o, src, hd.dim);
					uint16_t hs; std::memcpy(&hs, src + hd.dim, 2);
					scale = halfToFloat(hs);
				}
				else
				{
					// requantizePackedU4 as the reference's SSE4.1 build computes it (src/archImpl/sse4_1.cpp:480-573; the pin of the CoNgram oracle): value =
					// mulhrs((nibble - zeroPoint) * localScale +- 4, 32768 / 9) -- a rounded division by 9 where the scalar version (archImpl/none.cpp:22-57)
					// truncates; zeroPoint = (local >> 6) + 6, localScale = (local & 63) + 9; scale = global / 8
					uint16_t hs; std::memcpy(&hs, src + hd.dim / 2, 2);
					const uint8_t* local = src + hd.dim / 2 + 2;
					for (uint32_t i = 0; i < hd.dim; ++i)
					{
						const uint8_t packed = src[i / 2];
						const int32_t nib = (i & 1) ? (packed >> 4) : (packed & 0x0F);
						const uint8_t l = local[i / hd.qgroup];
						int32_t v = (nib - (int32_t)((l >> 6) + 6)) * (int32_t)((l & 0x3F) + 9);
						v += v > 0 ? 4 : v < 0 ? -4 : 0;
						const int32_t q = (((v * (32768 / 9)) >> 14) + 1) >> 1;      // _mm_mulhrs_epi16
						o[i] = (uint8_t)(int8_t)q;
					}
					scale = halfToFloat(hs) / 8;
				}
				std::memcpy(o + hd.dim, &scale, 4);
			};
			for (size_t i = 0; i < hd.contextSize; ++i, e += ctxRec)
			{
				uint8_t* o = &m.congCtxEmb[i * stride];
				readRow(e, o);
				uint16_t hb; std::memcpy(&hb, e + rowRec, 2);
				const float bias = -halfToFloat(hb);
				std::memcpy(o + hd.dim + 4, &bias, 4);
				if (hd.windowSize)
				{
					uint16_t hc, hv; std::memcpy(&hc, e + rowRec + 2, 2); std::memcpy(&hv, e + rowRec + 4, 2);
					m.congCtxConf[i * 2] = halfToFloat(hc); m.congCtxConf[i * 2 + 1] = halfToFloat(hv);
				}
			}
			for (size_t i = 0; i < hd.vocabSize; ++i, e += outRec) readRow(e, &m.congOutEmb[i * stride]);
			if (hd.windowSize)
			{
				m.congDistEmb.assign(hd.vocabSize * stride, 0); m.congDistConf.assign(hd.vocabSize, 0.f);
				for (size_t i = 0; i < hd.vocabSize; ++i, e += distRec)
				{
					uint8_t* o = &m.congDistEmb[i * stride];
					readRow(e, o);
					uint16_t hb, hc; std::memcpy(&hb, e + rowRec, 2); std::memcpy(&hc, e + rowRec + 2, 2);
					const float bias = -halfToFloat(hb);
					std::memcpy(o + hd.dim + 4, &bias, 4);
					m.congDistConf[i] = halfToFloat(hc);
				}
				m.congPosConf.assign(hd.windowSize + 1, 0.f);
				for (size_t i = 0; i < hd.windowSize; ++i, e += 2) { uint16_t h; std::memcpy(&h, e, 2); m.congPosConf[i + 1] = halfToFloat(h); }
				m.congDistMask.assign(e, e + (hd.vocabSize + 7) / 8);
			}
			// device lookup structures, in the shapes of the Knlm ones: edge hash (slot.ll carries the child's context id), root table, suffix links
			auto asF = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
			auto edgeCtx = [&](uint32_t node, int32_t v) { return v > 0 ? asF(m.congNodes[node + v].value) : asF(0u); };
			m.congBackoff.resize(nonLeaf);
			for (size_t i = 0; i < nonLeaf; ++i) m.congBackoff[i] = LmBackoff{ m.congNodes[i].lower, 0.f };
			m.congRoot2.assign(rootSize, LmRootRec{ 0, 0.f });
			for (uint32_t i = 0; i < m.congNodes[0].numNexts; ++i) if (keys[i] < rootSize) m.congRoot2[keys[i]] = LmRootRec{ m.congValues[i], edgeCtx(0, m.congValues[i]) };
			const size_t nEdges = m.congKeys.size() - m.congNodes[0].numNexts;
			size_t nBuckets = 1;
			while (nBuckets * 2 < nEdges + 1) nBuckets <<= 1;
			m.congHashMask = (uint32_t)(nBuckets - 1);
			m.congHash.assign(nBuckets * 4, LmSlot{ LM_SLOT_EMPTY, LM_SLOT_EMPTY, 0, 0.f });
			for (uint32_t nd = 1; nd < nonLeaf; ++nd)
			{
				const CongNodeRec& r = m.congNodes[nd];
				for (uint32_t ei = 0; ei < r.numNexts; ++ei)
				{
					const uint32_t wid = m.congKeys[r.nextOff + ei];
					const int32_t v = m.congValues[r.nextOff + ei];
					uint32_t b = lmHashOf(nd, wid) & m.congHashMask;
					for (;;)
					{
						LmSlot* s2 = &m.congHash[(size_t)b * 4];
						int k = 0;
						while (k < 4 && s2[k].node != LM_SLOT_EMPTY) ++k;
						if (k < 4) { s2[k] = LmSlot{ nd, wid, v, edgeCtx(nd, v) }; break; }
						b = (b + 1) & m.congHashMask;
					}
				}
			}
		}

		// ---- the reference's on-disk model files ------------------------------------------------------------
		// sj.morph = serializer::writeMany(os, toKey("KIWI"), forms, morphemes) (src/KiwiBuilder.cpp:923-937).  The serializer's format
		// (src/serializer.hpp:212-350): fundamentals and enums raw little-endian, vectors and strings a u32 count followed by the elements,
		// pairs member by member.  FormRaw = {form: u16 string, candidate: vector<u32>} (src/Form.cpp:71); MorphemeRaw = {kform u32, tag u8,
		// vpPack u8, senseId u8, combineSocket u8, combined i32, userScore f32, chunks vector<u32>, chunkPositions vector<pair<u8,u8>>,
		// lmMorphemeId u32, groupId u32, dialect u16, _reserved u16} (src/Form.cpp:36).  sj.knlm and skipbigram.mdl are memory images
		// (include/kiwi/Knlm.h:9-15, src/SkipBigramModel.hpp:45-92) -- the layouts loadKnlm / the SkipBigram loader below already read.
		struct ModelFiles
		{
			std::vector<uint32_t> meta, formPtr, formCandPtr, formCand, chunkIds;
			std::vector<uint16_t> formChars;
			std::vector<RawMorph> morph;
			std::vector<uint8_t> chunkPos, knlm, sbg, cong, nounchr;
		};

		// ---- character-level CoNgram model (reference nounchr.mdl; writer src/CoNgramModel.cpp:2087-2400, reader :425-789 with VlKeyType = uint8_t) ----
		// header | node sizes, one BYTE per node of the pre-order stream | keys, bytes | values, Stream VByte "0124" | [flag 4] one frequency byte per node
		// (16-byte aligned behind the values) | per context: row, fp16 -bias | per token: row | [flag 1] one bias code per token + fp16 minimum
		// (bias = code * -min / 255 + min) | [flag 2] u16 per token: its id in the trie's key space | [flag 4] fp16 entropy per context (not used here)
		void loadChr(FlatModel& m, const uint8_t* blob, size_t size)
		{
			struct Header { uint64_t vocabSize, contextSize; uint16_t dim, flags; uint8_t keySize, windowSize, qbit, qgroup; uint64_t numNodes, nodeOffset, keyOffset, valueOffset, embOffset; };
			if (size < sizeof(Header)) throw std::runtime_error{ "nounchr.mdl: truncated header" };
			Header hd; std::memcpy(&hd, blob, sizeof(hd));
			if (hd.keySize != 1) throw std::runtime_error{ "nounchr.mdl: a character model has 8-bit keys (keySize 1)" };
			if (hd.qbit != 8 || hd.windowSize != 0) throw std::runtime_error{ "nounchr.mdl: only 8-bit embeddings without a window are supported by this loader" };
			if (hd.dim == 0 || hd.dim % 4 || hd.numNodes < 1 || hd.vocabSize > 65535) throw std::runtime_error{ "nounchr.mdl: bad header" };
			const uint8_t* end = blob + size;
			const size_t nNodes = hd.numNodes;
			if (hd.nodeOffset + nNodes > size || hd.keyOffset + nNodes - 1 > size) throw std::runtime_error{ "nounchr.mdl: truncated trie" };
			std::vector<uint32_t> values(nNodes);
			const size_t valBytes = svbDecode(blob + hd.valueOffset, end, values.data(), nNodes, true);
			if (hd.flags & 4)
			{
				// one frequency byte per node, 16-byte aligned behind the values, packed on top of the context id as the reference does (clamped to 127,
				// CoNgramModel.cpp:467-479): "no context of its own" (value == 0, inherit from the suffix) is tested on the PACKED value there, so a
				// node with context 0 and a non-zero frequency keeps context 0; chrProgress unpacks (CoNgramModel::unpackContextId)
				const size_t fo = hd.valueOffset + (valBytes + 15) / 16 * 16;
				if (fo + nNodes > size) throw std::runtime_error{ "nounchr.mdl: truncated trie frequencies" };
				for (size_t i = 0; i < nNodes; ++i) values[i] = (values[i] & 0x00FFFFFFu) | ((uint32_t)std::min<uint8_t>(blob[fo + i], 127) << 24);
			}
			const uint8_t* sizes = blob + hd.nodeOffset; const uint8_t* keys = blob + hd.keyOffset;
			size_t nonLeaf = 0;
			for (size_t i = 0; i < nNodes; ++i) nonLeaf += sizes[i] ? 1 : 0;
			m.chrNodes.assign(nonLeaf, CongNodeRec{});
			m.chrKeys.assign(keys, keys + (nNodes - 1));
			m.chrValues.assign(nNodes - 1, 0);
			m.chrRoot.assign(256, 0);
			struct Range { size_t node, cur, end; };
			std::vector<Range> st;
			size_t ni = 0, nextOff = 0;
			for (size_t i = 0; i < nNodes; ++i)
			{
				if (sizes[i])
				{
					if (!st.empty()) m.chrValues[st.back().cur] = (int32_t)(ni - st.back().node);
					CongNodeRec& n = m.chrNodes[ni];
					n.value = values[i]; n.numNexts = sizes[i]; n.nextOff = (uint32_t)nextOff;
					st.push_back(Range{ ni, nextOff, nextOff + sizes[i] });
					nextOff += sizes[i];
					++ni;
				}
				else
				{
					if (st.empty()) throw std::runtime_error{ "nounchr.mdl: malformed node stream" };
					m.chrValues[st.back().cur] = -(int32_t)values[i];
					st.back().cur++;
					while (st.back().cur == st.back().end)
					{
						st.pop_back();
						if (st.empty()) break;
						st.back().cur++;
					}
				}
			}
			for (uint32_t i = 0; i < m.chrNodes[0].numNexts; ++i) m.chrRoot[m.chrKeys[i]] = m.chrValues[i];
			m.chrDim = hd.dim; m.chrCtx = (uint32_t)hd.contextSize; m.chrVocab = (uint32_t)hd.vocabSize;
			ChrView C = m.chrView();      // (host walk over the tables filled so far; embeddings follow)
			// suffix links and inherited context ids, breadth first (CoNgramModel.cpp:547-568; findLowerNode / findLowerValue, CoNgramModel.hpp:181-227)
			// ... and the depth of every node in TOKENS (CoNgramModel.cpp:551-572: the second byte of a two-byte spelling, key >= 224, does not count;
			// CoNgramModel::getNodeDepth -- the frequency-based unknown-form score asks for it)
			m.chrDepth.assign(nonLeaf, 0);
			std::deque<uint32_t> dq{ 0u };
			while (!dq.empty())
			{
				const uint32_t p = dq.front(); dq.pop_front();
				const CongNodeRec pn = m.chrNodes[p];
				for (uint32_t i = 0; i < pn.numNexts; ++i)
				{
					const int32_t v = m.chrValues[pn.nextOff + i];
					if (v <= 0) continue;
					const uint32_t k = m.chrKeys[pn.nextOff + i];
					const uint32_t child = p + v;
					uint32_t node = p, lowerNode;
					for (;;)
					{
						if (!m.chrNodes[node].lower) { lowerNode = node; break; }
						const uint32_t low = node + m.chrNodes[node].lower;
						int32_t found;
						if (chrSearch(C, m.chrNodes[low], k, found) && found > 0) { lowerNode = low + found; break; }
						node = low;
					}
					m.chrNodes[child].lower = (int32_t)lowerNode - (int32_t)child;
					if (m.chrNodes[child].value == 0)
					{
						uint32_t nd = p; uint32_t val = 0; bool done = false;
						while (m.chrNodes[nd].lower)
						{
							const uint32_t low = nd + m.chrNodes[nd].lower;
							int32_t found;
							if (chrSearch(C, m.chrNodes[low], k, found)) { val = found >= 0 ? m.chrNodes[low + found].value : (uint32_t)(-found); done = true; break; }
							nd = low;
						}
						if (!done) val = m.chrNodes[nd].value;
						m.chrNodes[child].value = val;
					}
					m.chrDepth[child] = (uint16_t)(m.chrDepth[p] + (k >= 224 ? 0 : 1));
					dq.push_back(child);
				}
			}
			// CoNgramModel::getContextFrequency (CoNgramModel.hpp:76-86): dequantizeFrequencyScale of the byte on top of a trie value (:34-39; libm's powf, as there)
			m.chrHasFreq = (hd.flags & 4) != 0;
			m.chrFreqTab.assign(256, 0.f);
			for (uint32_t q = 0; q < 256; ++q) m.chrFreqTab[q] = q <= 16 ? (float)q : powf(2.0f, ((float)q + 16) / 8.0f);
			const size_t stride = (size_t)hd.dim + 8;
			m.chrCtxEmb.assign(hd.contextSize * stride, 0); m.chrOutEmb.assign(hd.vocabSize * stride, 0);
			const uint8_t* e = blob + hd.embOffset;
			const size_t ctxRec = (size_t)hd.dim + 4, outRec = (size_t)hd.dim + 2;
			size_t need = hd.contextSize * ctxRec + hd.vocabSize * outRec;
			if (hd.flags & 1) need += hd.vocabSize + 2;
			if (hd.flags & 2) need += 2 * hd.vocabSize;
			if (e + need > end) throw std::runtime_error{ "nounchr.mdl: truncated embeddings" };
			for (size_t i = 0; i < hd.contextSize; ++i, e += ctxRec)
			{
				uint8_t* o = &m.chrCtxEmb[i * stride];
				std::memcpy(o, e, hd.dim);
				uint16_t hs, hb; std::memcpy(&hs, e + hd.dim, 2); std::memcpy(&hb, e + hd.dim + 2, 2);
				const float scale = halfToFloat(hs), bias = -halfToFloat(hb);
				std::memcpy(o + hd.dim, &scale, 4); std::memcpy(o + hd.dim + 4, &bias, 4);
			}
			for (size_t i = 0; i < hd.vocabSize; ++i, e += outRec)
			{
				uint8_t* o = &m.chrOutEmb[i * stride];
				std::memcpy(o, e, hd.dim);
				uint16_t hs; std::memcpy(&hs, e + hd.dim, 2);
				const float scale = halfToFloat(hs);
				std::memcpy(o + hd.dim, &scale, 4);
			}
			if (hd.flags & 1)
			{
				uint16_t hm; std::memcpy(&hm, e + hd.vocabSize, 2);
				const float minVal = halfToFloat(hm);
				for (size_t i = 0; i < hd.vocabSize; ++i)
				{
					const float b = (float)e[i] * (-minVal) / 255.f + minVal;      // CoNgramModel.cpp:764-772
					std::memcpy(&m.chrOutEmb[i * stride + hd.dim + 4], &b, 4);
				}
				e += hd.vocabSize + 2;
			}
			m.chrInv.clear();
			if (hd.flags & 2)
			{
				m.chrInv.resize(hd.vocabSize);
				std::memcpy(m.chrInv.data(), e, 2 * hd.vocabSize);
				e += 2 * hd.vocabSize;
			}
			// the state after <s> (UnkFormScorer::UnkFormScorer, src/UnkFormScorer.cpp:22-25)
			C = m.chrView();
			int32_t node = 0; uint32_t ctx = 0;
			chrProgressPacked(C, node, ctx, 0);
			m.chrBosNode = node; m.chrBosCtx = ctx & 0x00FFFFFFu; m.chrBosCtxPacked = ctx;
		}

		std::vector<uint8_t> readFile(const std::string& path, bool required)
		{
			FILE* f = std::fopen(path.c_str(), "rb");
			if (!f) { if (required) throw std::runtime_error{ "cannot open model file: " + path }; return {}; }
			std::fseek(f, 0, SEEK_END);
			const long n = std::ftell(f);
			std::fseek(f, 0, SEEK_SET);
			std::vector<uint8_t> b((size_t)std::max(0l, n));
			if (n > 0 && std::fread(b.data(), 1, (size_t)n, f) != (size_t)n) { std::fclose(f); throw std::runtime_error{ "short read: " + path }; }
			std::fclose(f);
			return b;
		}

		void loadModelDir(const std::string& dir, ModelFiles& o, RawModel& raw)
		{
			// A model directory as Kiwi ships it also holds the build-time inputs of the reference's KiwiBuilder -- combiningRule.txt (REQUIRED there:
			// rule-combined morphemes are generated from it at build time, src/KiwiBuilder.cpp:1080-1092, 2385-2640) and default / multi / typo .dict
			// (:1035-1078).  This loader reads sj.morph and the language-model files only: analysing with such a directory would silently miss every
			// rule-combined and dictionary entry.  Refused loudly instead, unless the caller asks for exactly that.
			{
				struct stat st;
				if (stat((dir + "/combiningRule.txt").c_str(), &st) == 0 && !std::getenv("KAMD_ALLOW_UNEXPANDED_MODEL"))
					throw std::runtime_error{ "kiwi_amd: this model directory holds combiningRule.txt: the reference expands rule-combined morphemes and loads its .dict "
						"files at build time, which this library does not do itself -- analyses would differ from the reference's.  Let the reference's builder do that step: "
						"tools/export_built.cpp (built against libkiwi) writes a container of this directory that kiwi_init loads.  (tools/check_model_dir.py "
						"describes the directory; KAMD_ALLOW_UNEXPANDED_MODEL=1 loads sj.morph + the language model alone)" };
			}
			const std::vector<uint8_t> mb = readFile(dir + "/sj.morph", true);
			const uint8_t* p = mb.data(); const uint8_t* end = p + mb.size();
			auto need = [&](size_t n) { if ((size_t)(end - p) < n) throw std::runtime_error{ "sj.morph: truncated" }; };
			auto get = [&](auto& v) { need(sizeof(v)); std::memcpy(&v, p, sizeof(v)); p += sizeof(v); };
			need(4);
			if (std::memcmp(p, "KIWI", 4) != 0) throw std::runtime_error{ "sj.morph: 'KIWI' is needed at the head of the file" };
			p += 4;
			uint32_t nForms; get(nForms);
			o.formPtr.assign(1, 0); o.formCandPtr.assign(1, 0);
			for (uint32_t i = 0; i < nForms; ++i)
			{
				uint32_t n; get(n); need(2 * (size_t)n);
				const size_t at = o.formChars.size();
				o.formChars.resize(at + n);
				if (n) std::memcpy(&o.formChars[at], p, 2 * (size_t)n);
				p += 2 * (size_t)n;
				o.formPtr.push_back((uint32_t)o.formChars.size());
				uint32_t c; get(c); need(4 * (size_t)c);
				const size_t ca = o.formCand.size();
				o.formCand.resize(ca + c);
				if (c) std::memcpy(&o.formCand[ca], p, 4 * (size_t)c);
				p += 4 * (size_t)c;
				o.formCandPtr.push_back((uint32_t)o.formCand.size());
			}
			uint32_t nMorphs; get(nMorphs);
			o.morph.assign(nMorphs, RawMorph{});
			for (uint32_t i = 0; i < nMorphs; ++i)
			{
				RawMorph& r = o.morph[i];
				get(r.kform); get(r.tag); get(r.vpPack); get(r.senseId); get(r.socket); get(r.combined); get(r.userScore);
				uint32_t nc; get(nc); need(4 * (size_t)nc);
				if (nc > 255) throw std::runtime_error{ "sj.morph: a morpheme with more than 255 chunks" };
				r.chunkPtr = (uint32_t)o.chunkIds.size(); r.nChunks = (uint8_t)nc;
				o.chunkIds.resize(o.chunkIds.size() + nc);
				if (nc) std::memcpy(&o.chunkIds[r.chunkPtr], p, 4 * (size_t)nc);
				p += 4 * (size_t)nc;
				uint32_t np; get(np); need(2 * (size_t)np);
				if (np != nc) throw std::runtime_error{ "sj.morph: chunkPositions.size() != chunks.size()" };
				o.chunkPos.insert(o.chunkPos.end(), p, p + 2 * (size_t)np);
				p += 2 * (size_t)np;
				uint32_t groupId; uint16_t reserved;
				get(r.lmId); get(groupId); get(r.dialect); get(reserved);
				r.origId = 0; r.pad = 0;
			}
			if (p != end) throw std::runtime_error{ "sj.morph: trailing bytes" };
			if (o.chunkIds.empty()) { o.chunkIds.push_back(0); o.chunkPos.assign(2, 0); }
			if (o.formChars.empty()) o.formChars.push_back(0);
			if (o.formCand.empty()) o.formCand.push_back(0);
			// language model files (KiwiBuilder.cpp:939-1031): cong.mdl alone is a complete model (models/cong/base ships no sj.knlm); else sj.knlm,
			// optionally with skipbigram.mdl
			o.cong = readFile(dir + "/cong.mdl", false);
			o.knlm = readFile(dir + "/sj.knlm", o.cong.empty());
			o.sbg = o.knlm.empty() ? std::vector<uint8_t>{} : readFile(dir + "/skipbigram.mdl", false);
			uint32_t vocab = 0;
			if (!o.knlm.empty())
			{
				if (o.knlm.size() < sizeof(KnlmHeader)) throw std::runtime_error{ "sj.knlm: truncated header" };
				KnlmHeader hd; std::memcpy(&hd, o.knlm.data(), sizeof(hd));
				vocab = (uint32_t)hd.vocab_size;      // the language model's vocabulary size is the reference's langVocabSize
			}
			else
			{
				if (o.cong.size() < 8) throw std::runtime_error{ "cong.mdl: truncated header" };
				uint64_t v; std::memcpy(&v, o.cong.data(), 8); vocab = (uint32_t)v;      // CoNgramModelHeader::vocabSize
			}
			o.meta = { nForms, nMorphs, vocab, 0 };
			raw.meta = o.meta.data(); raw.formPtr = o.formPtr.data(); raw.formChars = o.formChars.data(); raw.formCandPtr = o.formCandPtr.data(); raw.formCand = o.formCand.data();
			raw.morph = o.morph.data(); raw.chunkIds = o.chunkIds.data(); raw.chunkPos = o.chunkPos.data();
			raw.knlm = o.knlm.data(); raw.knlmSize = o.knlm.size();
			raw.knlm = o.knlm.empty() ? nullptr : o.knlm.data();
			raw.sbg = o.sbg.empty() ? nullptr : o.sbg.data(); raw.sbgSize = o.sbg.size();
			raw.cong = o.cong.empty() ? nullptr : o.cong.data(); raw.congSize = o.cong.size();
			// the character model of Match::oovChrModel (KiwiBuilder.cpp:1094-1100): optional
			o.nounchr = readFile(dir + "/nounchr.mdl", false);
			raw.nounchr = o.nounchr.empty() ? nullptr : o.nounchr.data(); raw.nounchrSize = o.nounchr.size();
		}
	}

	// `path`: a KAMDRAW1 container (kiwi_amd/synth.py), or a directory -- one that holds the reference's own model files sj.morph + sj.knlm
	// (+ skipbigram.mdl), else one that holds kiwi_amd.raw
	float chrScoreHost(const ChrView& C, const uint16_t* s, size_t n)
	{
		int32_t node = C.bosNode; uint32_t ctx = C.bosCtx;
		float score = 0;
		for (size_t i = 0; i < n; ++i) score += chrProgress(C, node, ctx, chrToken(s[i], identifySpecialChr(s[i])));
		score += chrProgress(C, node, ctx, 0);
		return score;
	}

	namespace
	{
		void loadRaw(const std::string& path, Container& file, ModelFiles& files, RawModel& raw)
		{
			struct stat st;
			if (stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode))
			{
				if (stat((path + "/sj.morph").c_str(), &st) == 0) loadModelDir(path, files, raw);
				else { file.load(path + "/kiwi_amd.raw"); raw.bind(file); }
			}
			else { file.load(path); raw.bind(file); }
		}
		void bakeRaw(FlatModel& m, const RawModel& raw, uint32_t enabledDialects, size_t tempFrom, size_t tempMorphFrom = (size_t)-1);
	}

	void bakeModel(FlatModel& m, const std::string& path, uint32_t enabledDialects)
	{
		Container file;
		ModelFiles files;
		RawModel raw;
		loadRaw(path, file, files, raw);
		bakeRaw(m, raw, enabledDialects, (size_t)-1);
	}

	// The model with TEMPORARY forms and morphemes behind its own -- what Kiwi::analyze builds per call for pretokenized spans that name forms or tags the
	// dictionary does not have (makePretokenizedSpanGroup, src/Kiwi.cpp:785-946: PretokenizedSpanGroup::forms / morphemes / formStrs).  They are baked like
	// the model's own entries (the same derived fields), keep their raw order behind the sorted forms (form id = raw index) and stay out of the trie:
	// only a span's lattice node refers to them.
	void bakeModelWithTemps(FlatModel& m, const std::string& path, uint32_t enabledDialects, const TempEntries& t)
	{
		Container file;
		ModelFiles files;
		RawModel raw;
		loadRaw(path, file, files, raw);
		const size_t nF = raw.nForms(), nM = raw.nMorphs();
		const size_t nCand = raw.formCandPtr[nF], nChars = raw.formPtr[nF];
		size_t nChunk = 0;
		for (size_t i = 0; i < nM; ++i) nChunk = std::max<size_t>(nChunk, (size_t)raw.morph[i].chunkPtr + raw.morph[i].nChunks);
		std::vector<uint32_t> meta(raw.meta, raw.meta + 4), formPtr(raw.formPtr, raw.formPtr + nF + 1), formCandPtr(raw.formCandPtr, raw.formCandPtr + nF + 1),
			formCand(raw.formCand, raw.formCand + nCand), chunkIds(raw.chunkIds, raw.chunkIds + nChunk);
		std::vector<uint16_t> formChars(raw.formChars, raw.formChars + nChars);
		std::vector<RawMorph> morph(raw.morph, raw.morph + nM);
		std::vector<uint8_t> chunkPos(raw.chunkPos, raw.chunkPos + 2 * nChunk);
		for (const auto& f : t.forms)
		{
			formChars.insert(formChars.end(), f.str.begin(), f.str.end());
			formPtr.push_back((uint32_t)formChars.size());
			for (uint32_t c : f.cands) formCand.push_back(c);
			formCandPtr.push_back((uint32_t)formCand.size());
		}
		for (const auto& tm : t.morphs)
		{
			RawMorph r{};
			r.kform = (uint32_t)(nF + tm.tempForm); r.lmId = tm.lmId; r.tag = tm.tag;
			r.chunkPtr = (uint32_t)chunkIds.size(); r.nChunks = (uint8_t)tm.chunks.size();
			for (const auto& c : tm.chunks) { chunkIds.push_back(c.morph); chunkPos.push_back(c.begin); chunkPos.push_back(c.end); }
			morph.push_back(r);
		}
		meta[0] = (uint32_t)(nF + t.forms.size()); meta[1] = (uint32_t)(nM + t.morphs.size());
		raw.meta = meta.data(); raw.formPtr = formPtr.data(); raw.formChars = formChars.data(); raw.formCandPtr = formCandPtr.data(); raw.formCand = formCand.data();
		raw.morph = morph.data(); raw.chunkIds = chunkIds.data(); raw.chunkPos = chunkPos.data();
		bakeRaw(m, raw, enabledDialects, nF, nM);
	}

	// What bakeRaw derives for the temporaries of bakeModelWithTemps, from the baked model alone (see TempOverlay): the per-morpheme fields (:965-1062 below), the
	// per-form fields and reductions (:1080-1165), in the same order and with the same tests -- a temporary has no vowel / polarity condition, no socket, no
	// sense id, no score, is its own combined morpheme and belongs to no dialect (RawMorph{} in bakeModelWithTemps).
	void bakeTempsOverlay(const FlatModel& m, const TempEntries& t, TempOverlay& o)
	{
		o = TempOverlay{};
		const uint32_t nF = m.h.nForms, nM = m.h.nMorphs, vocab = m.h.vocabSize;
		o.nBaseForms = nF; o.nBaseMorphs = nM;
		if (t.forms.empty() && t.morphs.empty()) return;
		const size_t nTM = t.morphs.size(), nTF = t.forms.size();
		auto tagOf = [&](uint32_t mi) -> uint8_t { return mi < nM ? m.morphs[mi].tag : t.morphs[mi - nM].tag; };
		auto lmIdOf = [&](uint32_t mi) -> uint32_t { return mi < nM ? m.morphs[mi].lmId : t.morphs[mi - nM].lmId; };
		auto kformOf = [&](uint32_t mi) -> U16 { return mi < nM ? m.formStr(m.morphKform[mi]) : U16(t.forms[t.morphs[mi - nM].tempForm].str); };
		auto checkId = [&](uint32_t mi) { if (mi >= nM + nTM) throw std::invalid_argument{ "temporary entries: morpheme id out of range" }; };
		// ---- morphemes ----
		o.morphs.assign(nTM, MorphRec{}); o.morphKform.resize(nTM); o.sbInfo.assign(nTM, 0); o.morphPath.assign(nTM, 0);
		auto morphAt = [&](uint32_t mi) -> const MorphRec& { return mi < nM ? m.morphs[mi] : o.morphs[mi - nM]; };
		auto chunkMorphAt = [&](uint32_t off) -> uint32_t { return off < m.chunkMorph.size() ? m.chunkMorph[off] : o.chunkMorph[off - m.chunkMorph.size()]; };
		auto chunkLmAt = [&](uint32_t off) -> uint32_t { return off < m.chunkLm.size() ? m.chunkLm[off] : o.chunkLm[off - m.chunkLm.size()]; };
		for (size_t k = 0; k < nTM; ++k)
		{
			const TempEntries::Morph& tm = t.morphs[k];
			if (tm.tempForm >= nTF) throw std::invalid_argument{ "temporary entries: form index out of range" };
			if (tm.chunks.size() > 255) throw std::invalid_argument{ "temporary entries: more than 255 chunks" };
			MorphRec& r = o.morphs[k];
			r.lmId = tm.lmId; r.userScore = 0; r.combinedId = (int32_t)(nM + k);
			r.tag = tm.tag; r.vowel = 0; r.polar = 0; r.socket = 0; r.senseId = 0; r.nChunks = (uint8_t)tm.chunks.size();
			r.chunkOff = (uint32_t)(m.chunkMorph.size() + o.chunkMorph.size());
			for (const auto& c : tm.chunks)
			{
				checkId(c.morph);
				o.chunkMorph.push_back(c.morph); o.chunkLm.push_back(lmIdOf(c.morph)); o.chunkPos.push_back(c.begin); o.chunkPos.push_back(c.end);
			}
			if (r.nChunks == 0) r.flags |= MF_SINGLE;
			o.morphKform[k] = nF + tm.tempForm;
		}
		for (size_t k = 0; k < nTM; ++k)
		{
			MorphRec& r = o.morphs[k];
			const U16& kf = t.forms[t.morphs[k].tempForm].str;
			const uint8_t tag = r.tag;
			bool hc = false;      // (its combined morpheme is itself, and not complex)
			for (uint32_t c = 0; c < r.nChunks; ++c) hc = hc || (morphAt(chunkMorphAt(r.chunkOff + c)).flags & MF_COMPLEX);
			if (hc) r.flags |= MF_HAS_COMPLEX;
			if (kf.empty()) r.flags |= MF_KFORM_EMPTY;
			r.feat = featMask((const uint16_t*)kf.data(), (uint32_t)kf.size());
			if (!kf.empty() && identifySpecialChr(kf.back()) == T_SSC) r.flags |= MF_ENDS_WITH_SSC;
			const uint16_t f0 = kf.empty() ? 0 : kf[0];
			if (isEClass(tag) && 0xC544 <= f0 && f0 <= 0xC774) r.flags |= MF_VOWEL_E;
			if ((tag == T_JKS || tag == T_JKC) && kf.size() == 1 && f0 == 0xAC00) r.flags |= MF_INF_J;
			if (f0 == 0xC73C || f0 == 0xB290 || (0xC0AC <= f0 && f0 <= 0xC2DC)) r.flags |= MF_BAD_PAIR_OF_L;
			if (isEClass(tag) && f0 == 0xC5B4) r.flags |= MF_CONTRACTABLE_E;
			uint32_t lastMorph, firstWid;
			if (r.flags & MF_SINGLE) { lastMorph = (uint32_t)r.combinedId; firstWid = r.lmId; }
			else { lastMorph = chunkMorphAt(r.chunkOff + r.nChunks - 1); firstWid = chunkLmAt(r.chunkOff); }
			r.lastSeqId = lastMorph >= vocab ? lastMorph : lmIdOf(lastMorph);
			if (lastMorph < vocab) r.flags |= MF_IN_VOCAB_LAST;
			if (firstWid < nM + nTM && tagOf(firstWid) == T_P) r.flags |= MF_FIRST_WID_IS_P;
			for (uint32_t c = 1; c < r.nChunks; ++c)
			{
				const uint32_t w = chunkLmAt(r.chunkOff + c);
				if (w < nM + nTM && tagOf(w) == T_P) r.flags |= MF_ANY_REST_WID_IS_P;
			}
			if (!(r.flags & MF_SINGLE) && kf.size() == 1 && (f0 == 0xB2E4 || f0 == 0xAC8C || f0 == 0xC9C0))
			{
				const U16 c0 = kformOf(chunkMorphAt(r.chunkOff));
				if (c0.size() == 1 && c0[0] == 0xD558) r.flags |= MF_HA_CONTRACTION;
			}
			uint8_t pf = 0;
			if (isIrregularTag(tag)) pf |= PF_IRREGULAR;
			if (tag == T_NP && kf.size() == 1 && (f0 == 0xB098 || f0 == 0xB108 || f0 == 0xC800)) pf |= PF_INFLECTENDA_NP;
			if (isVerbClass(tag) && !kf.empty() && kf.back() == 0x11AF) pf |= PF_VERB_L;
			if (isVerbClass(tag) && matchPolar((const uint16_t*)kf.data(), (uint32_t)kf.size(), CP_POSITIVE)) pf |= PF_POSITIVE_VERB;
			if (isVerbClass(tag) && !kf.empty() && !isHangulCoda(kf.back())) pf |= PF_VERB_VOWEL;
			if (tag == T_VA || tag == T_XSA) pf |= PF_VA_OR_XSA;
			if (isEClass(tag) && tag != T_EF) pf |= PF_E_NOT_EF;
			if (tag == T_UNKNOWN || tag == T_EF || tag == T_SF) pf |= PF_UNK_EF_SF;
			r.prevFlags = pf;
			r.special = 6;
			if (tag == T_SB) o.sbInfo[k] = (uint8_t)getSBType(joinHangul(kf));
		}
		for (size_t k = 0; k < nTM; ++k)
		{
			const MorphRec& cm = o.morphs[k];
			const MorphRec& wm = morphAt(cm.lastSeqId);
			uint16_t f;
			if (!(wm.flags & MF_KFORM_EMPTY)) f = wm.feat | ((wm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
			else if (cm.tag == T_UNKNOWN && cm.nChunks)
			{
				const MorphRec& lm = morphAt(chunkMorphAt(cm.chunkOff + cm.nChunks - 1));
				f = lm.feat | ((lm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
			}
			else f = cm.feat | ((cm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
			if (cm.tag == T_SSC) f |= LF_TAG_SSC;
			if (cm.tag == T_Z_SIOT) f |= LF_PREV_ZSIOT;
			o.morphPath[k] = (uint32_t)f | ((uint32_t)wm.prevFlags << 16);
		}
		// ---- forms (they keep their order behind the model's and stay out of the trie) ----
		o.forms.assign(nTF + 1, FormRec{});
		const uint32_t charBase = (uint32_t)m.formChars.size() - 1;      // (the model's closing 0 is where the first temporary string begins)
		uint8_t hash = m.forms[nF - 1].formHash;
		for (size_t j = 0; j < nTF; ++j)
		{
			const U16& s = t.forms[j].str;
			if (s.size() > 255) throw std::invalid_argument{ "temporary entries: form longer than 255 units" };
			if (t.forms[j].cands.size() > 65535) throw std::invalid_argument{ "temporary entries: too many candidates" };
			FormRec& f = o.forms[j];
			f.charOff = charBase + (uint32_t)o.formChars.size();
			f.len = (uint8_t)s.size();
			f.numSpaces = (uint8_t)std::count(s.begin(), s.end(), u' ');
			o.formChars.insert(o.formChars.end(), s.begin(), s.end());
			f.candOff = (uint32_t)(m.formCand.size() + o.formCand.size());
			f.candCnt = (uint16_t)t.forms[j].cands.size();
			for (uint32_t c : t.forms[j].cands) { checkId(c); o.formCand.push_back(c); }
			const uint32_t* cand = o.formCand.data() + (f.candOff - m.formCand.size());
			if (!s.empty() && isHangulSyllable(s.back()))
			{
				bool zc = false, zs = false;
				for (uint32_t c = 0; c < f.candCnt; ++c)
				{
					const uint32_t mi = cand[c];
					const MorphRec& mr = morphAt(mi);
					uint8_t tag = mr.tag;
					if (tag == T_UNKNOWN && mr.nChunks) tag = tagOf(chunkMorphAt(mr.chunkOff + mr.nChunks - 1));
					if (isJClass(tag) || isEClass(tag)) zc = true;
					const uint8_t t2 = mr.tag;
					if (isNNClass(t2) && mr.lmId != (uint32_t)clearIrregular(t2) + 1) zs = true;
				}
				if (zc) f.flags |= FF_ZCODA_APPENDABLE;
				if (zs) f.flags |= FF_ZSIOT_APPENDABLE;
			}
			if (!s.empty() && isHangulCoda(s[0])) f.flags |= FF_FIRST_IS_CODA;
			if (s.size() == 1) { const uint8_t st = identifySpecialChr(s[0]); if (T_SF <= st && st <= T_SW) f.flags |= FF_IS_STAG; }
			if (!s.empty() && s[0] == 0xC544) f.flags |= FF_STARTS_WITH_A;
			if (!s.empty() && identifySpecialChr(s.back()) == T_SSC) f.flags |= FF_ENDS_WITH_SSC;
			{
				const U16 prev = j ? t.forms[j - 1].str : m.formStr(nF - 1);
				if (!equalIgnoringSpace(s, prev)) ++hash;
				f.formHash = hash;
			}
			if (f.candCnt)
			{
				const MorphRec& c0 = morphAt(cand[0]);
				if (c0.vowel != CV_NONE) { uint8_t v = c0.vowel; for (uint32_t c = 0; c < f.candCnt; ++c) v = reduceVowel(v, morphAt(cand[c]).vowel); f.vowelPolar = (uint8_t)((f.vowelPolar & 0xF0) | v); }
				if (c0.polar != CP_NONE) { uint8_t p = c0.polar; for (uint32_t c = 0; c < f.candCnt; ++c) p = (p == morphAt(cand[c]).polar) ? p : (uint8_t)CP_NONE; f.vowelPolar = (uint8_t)((f.vowelPolar & 0x0F) | (p << 4)); }
				bool hasJ = false, anyFull = false, allPartial = true;
				for (uint32_t c = 0; c < f.candCnt; ++c)
				{
					const MorphRec& mm = morphAt(cand[c]);
					const uint8_t tg = mm.tag;
					hasJ = hasJ || isJClass(tg) || tg == T_EC || tg == T_EF;
					const uint8_t ct = clearIrregular(tg);
					const uint32_t md = (cand[c] < nM && !m.morphDialect.empty()) ? m.morphDialect[cand[c]] : 0u;
					anyFull = anyFull || (md == 0 && ct != T_UNKNOWN && ct != T_P && ct != T_P + 1);
					if (!(mm.socket || !(mm.flags & MF_SINGLE))) allPartial = false;
				}
				if (f.candCnt == 1 && morphAt(cand[0]).tag == T_UNKNOWN && morphAt(cand[0]).nChunks) allPartial = false;
				if (allPartial) f.flags2 |= FF2_ALL_PARTIAL;
				if (hasJ) f.flags |= FF_HAS_JCLASS;
				if (anyFull) f.flags |= FF_HAS_ANY_FULL;
			}
		}
		o.forms[nTF].charOff = charBase + (uint32_t)o.formChars.size();
		o.forms[nTF].candOff = (uint32_t)(m.formCand.size() + o.formCand.size());
		{
			const U16 prev = nTF ? t.forms[nTF - 1].str : m.formStr(nF - 1);
			if (!equalIgnoringSpace(U16{}, prev)) ++hash;
			o.forms[nTF].formHash = hash;
		}
		o.formChars.push_back(0);
		if (m.chrDim && !m.formUnkChr.empty())
		{
			const ChrView C = m.chrView();
			o.formUnkChr.assign(nTF, 0.f);
			for (size_t j = 0; j < nTF; ++j) o.formUnkChr[j] = chrScoreHost(C, o.formChars.data() + (o.forms[j].charOff - charBase), o.forms[j].len);
			o.formChrTok.resize(o.formChars.size());
			for (size_t i = 0; i < o.formChars.size(); ++i) o.formChrTok[i] = (uint16_t)chrToken(o.formChars[i], identifySpecialChr(o.formChars[i]));
		}
	}

	namespace {
	void bakeRaw(FlatModel& m, const RawModel& raw, uint32_t enabledDialects, size_t tempFrom, size_t tempMorphFrom)
	{
		const size_t nF = raw.nForms(), nM = raw.nMorphs();
		if (nF < kDefaultFormSize) throw std::runtime_error{ "raw model: missing default forms" };
		const size_t nSorted = std::min(nF, tempFrom);      // (temporary forms keep their raw order behind the sorted ones)

		std::vector<U16> rawForm(nF);
		for (size_t i = 0; i < nF; ++i) rawForm[i].assign((const char16_t*)raw.formChars + raw.formPtr[i], (const char16_t*)raw.formChars + raw.formPtr[i + 1]);
		auto rawTag = [&](uint32_t mi) { return raw.morph[mi].tag; };

		// ---- form order: defaults fixed, the rest sorted by (string ignoring spaces, original index) ------
		std::vector<uint32_t> order(nF);
		std::iota(order.begin(), order.end(), 0u);
		std::sort(order.begin() + kDefaultFormSize, order.begin() + nSorted, [&](uint32_t a, uint32_t b)
		{
			const int c = cmpIgnoringSpace(rawForm[a], rawForm[b]);
			if (c == -1) return true;
			if (cmpIgnoringSpace(rawForm[b], rawForm[a]) == -1) return false;
			return a < b;
		});
		std::vector<uint32_t> newId(nF);
		for (size_t i = 0; i < nF; ++i) newId[order[i]] = (uint32_t)i;

		// ---- morphemes -----------------------------------------------------------------------------------
		m.morphs.assign(nM, MorphRec{});
		m.morphKform.resize(nM);
		m.sbInfo.assign(nM, 0);
		std::vector<uint8_t> complexRaw(nM);
		for (size_t i = 0; i < nM; ++i)
		{
			const RawMorph& r = raw.morph[i];
			MorphRec& o = m.morphs[i];
			o.lmId = r.lmId; o.userScore = r.userScore; o.combinedId = (int32_t)i + r.combined;
			o.tag = r.tag; o.vowel = r.vpPack & 0xF; o.polar = (r.vpPack >> 4) & 0x3; // Morpheme::polar is a 2-bit field (Form.h:147)
			o.socket = r.socket; o.senseId = r.senseId; o.nChunks = r.nChunks;
			o.chunkOff = (uint32_t)m.chunkMorph.size();
			complexRaw[i] = (r.vpPack & 0x80) ? 1 : 0;
			bool hasSaisiot = false;
			for (size_t c = 0; c < r.nChunks; ++c)
			{
				const uint32_t cm = raw.chunkIds[r.chunkPtr + c];
				m.chunkMorph.push_back(cm);
				m.chunkLm.push_back(raw.morph[cm].lmId);
				m.chunkPos.push_back(raw.chunkPos[(r.chunkPtr + c) * 2]);
				m.chunkPos.push_back(raw.chunkPos[(r.chunkPtr + c) * 2 + 1]);
				hasSaisiot = hasSaisiot || rawTag(cm) == T_Z_SIOT;
			}
			if (complexRaw[i] && !hasSaisiot) o.flags |= MF_COMPLEX;   // src/Form.cpp:135-137
			if (complexRaw[i] && hasSaisiot) o.flags |= MF_SAISIOT;
			if (r.nChunks == 0 || (o.flags & (MF_COMPLEX | MF_SAISIOT))) o.flags |= MF_SINGLE;
			m.morphKform[i] = newId[r.kform];
			if (r.dialect) { if (m.morphDialect.empty()) m.morphDialect.assign(nM, 0); m.morphDialect[i] = r.dialect; }      // Morpheme::dialect (Dialect bits; 0 = standard)
		}
		const uint32_t vocab = (uint32_t)raw.vocabSize();
		for (size_t i = 0; i < nM; ++i)
		{
			MorphRec& o = m.morphs[i];
			const U16& kf = rawForm[raw.morph[i].kform];
			const uint8_t tag = o.tag;
			// hasComplex (Form.h:176-185)
			bool hc = (m.morphs[o.combinedId].flags & MF_COMPLEX) != 0;
			for (uint32_t c = 0; c < o.nChunks; ++c) hc = hc || (m.morphs[m.chunkMorph[o.chunkOff + c]].flags & MF_COMPLEX);
			if (hc) o.flags |= MF_HAS_COMPLEX;
			if (kf.empty()) o.flags |= MF_KFORM_EMPTY;
			o.feat = featMask((const uint16_t*)kf.data(), (uint32_t)kf.size());
			if (!kf.empty() && identifySpecialChr(kf.back()) == T_SSC) o.flags |= MF_ENDS_WITH_SSC;
			const uint16_t f0 = kf.empty() ? 0 : kf[0];
			if (isEClass(tag) && 0xC544 <= f0 && f0 <= 0xC774) o.flags |= MF_VOWEL_E;            // hasNoOnset: '아'..'이'
			if ((tag == T_JKS || tag == T_JKC) && kf.size() == 1 && f0 == 0xAC00) o.flags |= MF_INF_J; // '가'
			if (f0 == 0xC73C || f0 == 0xB290 || (0xC0AC <= f0 && f0 <= 0xC2DC)) o.flags |= MF_BAD_PAIR_OF_L; // '으','느','사'..'시'
			if (isEClass(tag) && f0 == 0xC5B4) o.flags |= MF_CONTRACTABLE_E;                     // '어'
			// first / last LM ids (PathEvaluator.hpp:537-558)
			uint32_t lastMorph, firstWid;
			if (o.flags & MF_SINGLE) { lastMorph = (uint32_t)o.combinedId; firstWid = o.lmId; }
			else { lastMorph = m.chunkMorph[o.chunkOff + o.nChunks - 1]; firstWid = m.chunkLm[o.chunkOff]; }
			o.lastSeqId = lastMorph >= vocab ? lastMorph : raw.morph[lastMorph].lmId;
			if (lastMorph < vocab) o.flags |= MF_IN_VOCAB_LAST;
			if (firstWid < nM && raw.morph[firstWid].tag == T_P) o.flags |= MF_FIRST_WID_IS_P;
			for (uint32_t c = 1; c < o.nChunks; ++c)
			{
				const uint32_t w = m.chunkLm[o.chunkOff + c];
				if (w < nM && raw.morph[w].tag == T_P) o.flags |= MF_ANY_REST_WID_IS_P;
			}
			if (!(o.flags & MF_SINGLE) && kf.size() == 1 && (f0 == 0xB2E4 || f0 == 0xAC8C || f0 == 0xC9C0)) // 다 게 지
			{
				const U16& c0 = rawForm[raw.morph[m.chunkMorph[o.chunkOff]].kform];
				if (c0.size() == 1 && c0[0] == 0xD558) o.flags |= MF_HA_CONTRACTION; // 하
			}
			// previous-morpheme predicates (PathEvaluator.hpp:46-83)
			uint8_t pf = 0;
			if (isIrregularTag(tag)) pf |= PF_IRREGULAR;
			if (tag == T_NP && kf.size() == 1 && (f0 == 0xB098 || f0 == 0xB108 || f0 == 0xC800)) pf |= PF_INFLECTENDA_NP; // 나 너 저
			if (isVerbClass(tag) && !kf.empty() && kf.back() == 0x11AF) pf |= PF_VERB_L;
			if (isVerbClass(tag) && matchPolar((const uint16_t*)kf.data(), (uint32_t)kf.size(), CP_POSITIVE)) pf |= PF_POSITIVE_VERB;
			if (isVerbClass(tag) && !kf.empty() && !isHangulCoda(kf.back())) pf |= PF_VERB_VOWEL;
			if (tag == T_VA || tag == T_XSA) pf |= PF_VA_OR_XSA;
			if (isEClass(tag) && tag != T_EF) pf |= PF_E_NOT_EF;
			if (tag == T_UNKNOWN || tag == T_EF || tag == T_SF) pf |= PF_UNK_EF_SF;
			o.prevFlags = pf;
			o.special = 6;
			if (tag == T_SB) m.sbInfo[i] = (uint8_t)getSBType(joinHangul(kf));
		}
		// what a path that ends in morpheme i (recorded word id = lastSeqId) exposes to the next morpheme:
		// FormEvaluator's choice of left string (PathEvaluator.hpp:261-291) and RuleBasedScorer's previous-morpheme tests
		m.morphPath.assign(nM, 0);
		for (size_t i = 0; i < nM; ++i)
		{
			const MorphRec& cm = m.morphs[i];
			const MorphRec& wm = m.morphs[cm.lastSeqId];
			uint16_t f;
			if (!(wm.flags & MF_KFORM_EMPTY)) f = wm.feat | ((wm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
			else if (cm.tag == T_UNKNOWN && cm.nChunks)
			{
				const MorphRec& lm = m.morphs[m.chunkMorph[cm.chunkOff + cm.nChunks - 1]];
				f = lm.feat | ((lm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
			}
			else f = cm.feat | ((cm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
			if (cm.tag == T_SSC) f |= LF_TAG_SSC;
			if (cm.tag == T_Z_SIOT) f |= LF_PREV_ZSIOT;
			m.morphPath[i] = (uint32_t)f | ((uint32_t)wm.prevFlags << 16);
		}
		// KiwiBuilder::getSpecialMorphs (KiwiBuilder.cpp:2642-2662)
		for (auto& s : m.h.specialMorph) s = 0;
		for (size_t i = 0; i < std::min(nM, tempMorphFrom); ++i)      // (the special morphemes are fixed when the model is built: a call's temporary morphemes never take their place)
		{
			const U16& fs = rawForm[raw.morph[i].kform];
			size_t base;
			if (fs == u"'") base = 0; else if (fs == u"\"") base = 3; else continue;
			const uint8_t tag = raw.morph[i].tag;
			if (tag == T_SSO) m.h.specialMorph[base + 0] = (uint32_t)i;
			else if (tag == T_SSC) m.h.specialMorph[base + 1] = (uint32_t)i;
			else if (tag == T_SS) m.h.specialMorph[base + 2] = (uint32_t)i;
		}
		// determineSpecialMorphType: first matching slot wins (include/kiwi/Kiwi.h:517-524)
		for (size_t i = 0; i < nM; ++i)
			for (int s = 5; s >= 0; --s) if (m.h.specialMorph[s] == i) m.morphs[i].special = (uint8_t)s;

		// ---- forms ---------------------------------------------------------------------------------------
		m.forms.assign(nF + 1, FormRec{});
		uint32_t maxLen = 0;
		for (size_t ni = 0; ni < nF; ++ni)
		{
			const uint32_t ri = order[ni];
			const U16& s = rawForm[ri];
			if (s.size() > 255) throw std::runtime_error{ "raw model: form longer than 255 units" };
			FormRec& f = m.forms[ni];
			f.charOff = (uint32_t)m.formChars.size();
			f.len = (uint8_t)s.size();
			f.numSpaces = (uint8_t)std::count(s.begin(), s.end(), u' ');
			m.formChars.insert(m.formChars.end(), s.begin(), s.end());
			f.candOff = (uint32_t)m.formCand.size();
			const uint32_t cb = raw.formCandPtr[ri], ce = raw.formCandPtr[ri + 1];
			f.candCnt = (uint16_t)(ce - cb);
			for (uint32_t c = cb; c < ce; ++c) m.formCand.push_back(raw.formCand[c]);
			maxLen = std::max<uint32_t>(maxLen, f.len - f.numSpaces);
			// zCoda / zSiot appendable (KiwiBuilder.cpp:2294-2352)
			if (!s.empty() && isHangulSyllable(s.back()))
			{
				bool zc = false, zs = false;
				for (uint32_t c = cb; c < ce; ++c)
				{
					const uint32_t mi = raw.formCand[c];
					uint8_t tag = rawTag(mi);
					if (tag == T_UNKNOWN && raw.morph[mi].nChunks) tag = rawTag(raw.chunkIds[raw.morph[mi].chunkPtr + raw.morph[mi].nChunks - 1]);
					if (isJClass(tag) || isEClass(tag)) zc = true;
					const uint8_t t2 = rawTag(mi);
					if (isNNClass(t2) && raw.morph[mi].lmId != (uint32_t)clearIrregular(t2) + 1) zs = true;
				}
				if (zc) f.flags |= FF_ZCODA_APPENDABLE;
				if (zs) f.flags |= FF_ZSIOT_APPENDABLE; // the extra !isHangulCoda(back) test there is implied by isHangulSyllable
			}
			if (!s.empty() && isHangulCoda(s[0])) f.flags |= FF_FIRST_IS_CODA;
			if (s.size() == 1) { const uint8_t t = identifySpecialChr(s[0]); if (T_SF <= t && t <= T_SW) f.flags |= FF_IS_STAG; }
			if (!s.empty() && s[0] == 0xC544) f.flags |= FF_STARTS_WITH_A;
			if (!s.empty() && identifySpecialChr(s.back()) == T_SSC) f.flags |= FF_ENDS_WITH_SSC;
		}
		m.forms[nF].charOff = (uint32_t)m.formChars.size();
		m.forms[nF].candOff = (uint32_t)m.formCand.size();
		m.formChars.push_back(0); // keeps &formChars[charOff] valid for the sentinel
		{
			uint8_t hash = 0;
			for (size_t i = 1; i <= nF; ++i)
			{
				const U16 a = m.formStr((uint32_t)i), b = m.formStr((uint32_t)i - 1);
				if (!equalIgnoringSpace(a, b)) ++hash;
				m.forms[i].formHash = hash;
			}
		}
		// per-form reductions + the list that goes into the trie (KiwiBuilder.cpp:2473-2515)
		std::vector<uint32_t> sortedForms;
		for (size_t i = kDefaultFormSize; i < nF; ++i)
		{
			FormRec& f = m.forms[i];
			if (!f.candCnt) continue;
			const uint32_t* cand = &m.formCand[f.candOff];
			const MorphRec& c0 = m.morphs[cand[0]];
			if (c0.vowel != CV_NONE) { uint8_t v = c0.vowel; for (uint32_t c = 0; c < f.candCnt; ++c) v = reduceVowel(v, m.morphs[cand[c]].vowel); f.vowelPolar = (uint8_t)((f.vowelPolar & 0xF0) | v); }
			if (c0.polar != CP_NONE) { uint8_t p = c0.polar; for (uint32_t c = 0; c < f.candCnt; ++c) p = (p == m.morphs[cand[c]].polar) ? p : (uint8_t)CP_NONE; f.vowelPolar = (uint8_t)((f.vowelPolar & 0x0F) | (p << 4)); }
			bool hasJ = false, anyFull = false;
			// Form::dialect (KiwiBuilder.cpp:2284-2292, 2500): standard as soon as one candidate is, else the union of the candidates' dialects
			uint32_t fDialect = m.morphDialect.empty() ? 0u : m.morphDialect[cand[0]];
			for (uint32_t c = 0; c < f.candCnt; ++c)
			{
				const uint8_t t = m.morphs[cand[c]].tag;
				hasJ = hasJ || isJClass(t) || t == T_EC || t == T_EF;
				const uint8_t ct = clearIrregular(t);
				const uint32_t md = m.morphDialect.empty() ? 0u : m.morphDialect[cand[c]];
				anyFull = anyFull || (md == 0 && ct != T_UNKNOWN && ct != T_P && ct != T_P + 1);      // (hasAnyFullMorphemes counts standard morphemes only, :2494-2498)
				fDialect = (fDialect == 0 || md == 0) ? 0u : (fDialect | md);
			}
			if (fDialect) { if (m.formDialect.empty()) m.formDialect.assign(nF + 1, 0); m.formDialect[i] = (uint16_t)fDialect; }
			{
				bool allPartial = true;
				for (uint32_t c = 0; c < f.candCnt; ++c)
				{
					const MorphRec& mm = m.morphs[cand[c]];
					if (!(mm.socket || !(mm.flags & MF_SINGLE))) allPartial = false;
				}
				if (f.candCnt == 1 && m.morphs[cand[0]].tag == T_UNKNOWN && m.morphs[cand[0]].nChunks) allPartial = false;
				if (allPartial) f.flags2 |= FF2_ALL_PARTIAL;
			}
			if (hasJ) f.flags |= FF_HAS_JCLASS;
			if (anyFull) f.flags |= FF_HAS_ANY_FULL;
			if (fDialect && !(enabledDialects & fDialect)) continue;      // a form of dialects that are not enabled stays out of the trie (:2501-2504)
			if (i >= tempFrom) continue;      // (a temporary form: reached through its span's node only)
			sortedForms.push_back((uint32_t)i);
		}
		{
			std::vector<U16> strs(nF + 1);
			for (auto i : sortedForms) strs[i] = m.formStr(i);
			// same algorithm + same comparator outcomes on the same input order as KiwiBuilder.cpp:2511-2514,
			// so equal-ignoring-space forms end up in the same relative order.
			std::sort(sortedForms.begin(), sortedForms.end(), [&](uint32_t a, uint32_t b) { return lessIgnoringSpace(strs[a], strs[b]); });
		}

		// ---- trie in creation order ----------------------------------------------------------------------
		std::vector<BuildNode> bn(kDefaultFormSize + 1);
		for (uint32_t i = 0; i < kDefaultFormSize; ++i) bn[i + 1].val = (int32_t)i;
		{
			U16 prev;
			std::vector<uint32_t> ptrs;
			for (auto fi : sortedForms)
			{
				U16 key;
				for (auto c : m.formStr(fi)) if (c != u' ') key.push_back(c);
				size_t common = 0;
				while (common < std::min(prev.size(), key.size()) && prev[common] == key[common]) ++common;
				ptrs.resize(key.size());
				uint32_t node = common ? ptrs[common - 1] : 0;
				for (size_t i = common; i < key.size(); ++i)
				{
					auto it = bn[node].next.find(key[i]);
					uint32_t child;
					if (it != bn[node].next.end()) child = it->second;
					else
					{
						child = (uint32_t)bn.size();
						bn.emplace_back();
						bn[child].depth = bn[node].depth + 1;
						bn[node].next[key[i]] = child;
					}
					node = child;
					ptrs[i] = node;
				}
				if (bn[node].val == TRIE_NONE) bn[node].val = (int32_t)fi;
				prev = key;
			}
		}
		const size_t nT = bn.size();
		m.trie.assign(nT, TrieNodeRec{});
		m.trieRoot.assign(65536, 0);
		for (size_t i = 0; i < nT; ++i)
		{
			auto& t = m.trie[i];
			t.edgeOff = (uint32_t)m.trieKeys.size();
			t.numNexts = (uint16_t)bn[i].next.size();
			if (bn[i].next.size() > 65535) throw std::runtime_error{ "trie fan-out overflow" };
			t.depth = bn[i].depth;
			t.value = bn[i].val;
			t.fail = -1;
			for (auto& p : bn[i].next) { m.trieKeys.push_back(p.first); m.trieChild.push_back(p.second); }
		}
		for (auto& p : bn[0].next) m.trieRoot[p.first] = p.second;
		auto findChild = [&](uint32_t node, uint16_t k) -> int64_t
		{
			const auto& t = m.trie[node];
			const uint16_t* kb = m.trieKeys.data() + t.edgeOff;
			const uint16_t* it = std::lower_bound(kb, kb + t.numNexts, k);
			if (it == kb + t.numNexts || *it != k) return -1;
			return m.trieChild[t.edgeOff + (it - kb)];
		};
		{
			std::deque<uint32_t> dq{ 0u };
			while (!dq.empty())
			{
				const uint32_t p = dq.front(); dq.pop_front();
				const auto pt = m.trie[p];
				for (uint32_t e = 0; e < pt.numNexts; ++e)
				{
					const uint16_t k = m.trieKeys[pt.edgeOff + e];
					const uint32_t child = m.trieChild[pt.edgeOff + e];
					// findFail (FrozenTrie.hpp:31-52): first proper suffix state that has an edge k
					uint32_t n = p; int32_t f = 0;
					for (;;)
					{
						if (m.trie[n].fail < 0) { f = (int32_t)n; break; }   // reached the root: child fails to the root
						n = (uint32_t)m.trie[n].fail;
						const int64_t c = findChild(n, k);
						if (c >= 0) { f = (int32_t)c; break; }
					}
					m.trie[child].fail = f;
					dq.push_back(child);
				}
				if (m.trie[p].value == TRIE_NONE)
				{
					// FrozenTrie.hpp:144-152 : mark "a proper suffix of this state is (or leads to) a form"
					for (int32_t n = (int32_t)p; m.trie[n].fail >= 0; n = m.trie[n].fail)
					{
						if (m.trie[n].value == TRIE_NONE) continue;
						m.trie[p].value = TRIE_SUBMATCH;
						break;
					}
				}
			}
		}

		buildTrieEdges(m);
		m.h.nForms = (uint32_t)nF; m.h.nMorphs = (uint32_t)nM; m.h.vocabSize = vocab;
		m.h.nTrieNodes = (uint32_t)nT; m.h.nTrieEdges = (uint32_t)m.trieKeys.size();
		m.h.maxFormLen = maxLen;
		if (maxLen > 64) throw std::runtime_error{ "dictionary form longer than 64 units: the trie-scan kernel's depth mask is 64 bits" };
		if (raw.knlm) loadKnlm(m, raw.knlm, raw.knlmSize);
		else if (!raw.cong) throw std::runtime_error{ "Cannot find any valid model files" };      // KiwiBuilder.cpp:984-990
		if (raw.cong) loadCong(m, raw.cong, raw.congSize);
		if (raw.nounchr)
		{
			// the reference loads nounchr.mdl QUANTISED only next to a CoNgram model (KiwiBuilder.cpp:1096-1099: fp32 through Eigen otherwise); the
			// quantised scorer is what is restated here, so the character model is kept for CoNgram models only
			if (raw.cong)
			{
				loadChr(m, raw.nounchr, raw.nounchrSize);
				// per form: the steps over its own string and </s> (UnkFormScorer::chrBasedScore before the bias, src/UnkFormScorer.cpp:53-66)
				const ChrView C = m.chrView();
				m.formUnkChr.assign(nF, 0.f);
				for (size_t f = 0; f < nF; ++f) m.formUnkChr[f] = chrScoreHost(C, m.formChars.data() + m.forms[f].charOff, m.forms[f].len);
				m.formChrTok.resize(m.formChars.size());
				for (size_t i = 0; i < m.formChars.size(); ++i) m.formChrTok[i] = (uint16_t)chrToken(m.formChars[i], identifySpecialChr(m.formChars[i]));
			}
		}
		if (raw.sbg)
		{
			// SkipBigramModel blob, uncompressed + unquantised (reference src/SkipBigramModel.hpp:40-105): header{u64 vocabSize; u8 keySize,
			// windowSize, compressed, quantize; u8 rsv[4]} | kSizes[vocab] | keyData[total] | discnts[vocab] f32 | compensations[total] f32 | validness[vocab]
			const uint8_t* p = raw.sbg;
			uint64_t vocab; std::memcpy(&vocab, p, 8);
			const uint8_t keySize = p[8], window = p[9], compressed = p[10], quantize = p[11];
			if (compressed || quantize || (keySize != 2 && keySize != 4) || window != 8) throw std::runtime_error{ "raw model: unsupported SkipBigram layout" };
			p += 16;
			auto key = [&](const uint8_t* q, size_t i) -> uint32_t { if (keySize == 2) { uint16_t v; std::memcpy(&v, q + 2 * i, 2); return v; } uint32_t v; std::memcpy(&v, q + 4 * i, 4); return v; };
			m.sbgPtrs.assign(vocab + 1, 0);
			for (size_t i = 0; i < vocab; ++i) m.sbgPtrs[i + 1] = m.sbgPtrs[i] + key(p, i);
			const size_t total = m.sbgPtrs[vocab];
			p += vocab * keySize;
			m.sbgKeys.resize(total);
			for (size_t i = 0; i < total; ++i) m.sbgKeys[i] = key(p, i);
			p += total * keySize;
			m.sbgDiscnts.resize(vocab); std::memcpy(m.sbgDiscnts.data(), p, vocab * 4); p += vocab * 4;
			m.sbgComps.resize(total); if (total) std::memcpy(m.sbgComps.data(), p, total * 4); p += total * 4;
			m.sbgValid.assign(p, p + vocab); p += vocab;
			if ((size_t)(p - raw.sbg) > raw.sbgSize) throw std::runtime_error{ "raw model: truncated SkipBigram blob" };
			m.sbgWindow = window;
		}
		if (raw.knlm && vocab > m.lmRoot.size()) throw std::runtime_error{ "raw model: vocab larger than LM vocab" };
	}
	}

	void buildTrieEdges(FlatModel& m)
	{
		size_t edges = 0;
		for (size_t n = 1; n < m.trie.size(); ++n) edges += m.trie[n].numNexts;
		size_t cap = 64;
		while (cap < 4 * edges) cap <<= 1;      // at most a quarter full: a lookup that misses ends at an empty slot after 1.2 probes on average
		if (cap > (1ull << 31)) throw std::runtime_error{ "form trie too large for the edge table" };
		m.trieEdges.assign(cap, TrieEdgeSlot{ TRIE_EDGE_EMPTY, 0, 0, TRIE_NONE });
		m.trieEdgeMask = (uint32_t)cap - 1;
		for (size_t n = 1; n < m.trie.size(); ++n)
		{
			const TrieNodeRec& t = m.trie[n];
			for (uint32_t e = 0; e < t.numNexts; ++e)
			{
				const uint32_t key = m.trieKeys[t.edgeOff + e];
				uint32_t h = trieEdgeHash((uint32_t)n, key) & m.trieEdgeMask;
				while (m.trieEdges[h].node != TRIE_EDGE_EMPTY) h = (h + 1) & m.trieEdgeMask;
				const uint32_t child = m.trieChild[t.edgeOff + e];
				m.trieEdges[h] = TrieEdgeSlot{ (uint32_t)n, key, child, m.trie[child].value };
			}
		}
	}

	int32_t formIdOfString(const FlatModel& m, const std::u16string& nrm)
	{
		uint32_t node = 0;
		for (size_t i = 0; i < nrm.size(); ++i)
		{
			const uint16_t c = (uint16_t)nrm[i];
			if (node == 0) { node = m.trieRoot[c]; if (!node) return -1; continue; }
			const TrieNodeRec& t = m.trie[node];
			const uint16_t* kb = m.trieKeys.data() + t.edgeOff;
			const uint16_t* it = std::lower_bound(kb, kb + t.numNexts, c);
			if (it == kb + t.numNexts || *it != c) return -1;
			node = m.trieChild[t.edgeOff + (it - kb)];
		}
		if (node == 0 || m.trie[node].value < 0) return -1;
		return m.trie[node].value;
	}

	std::vector<uint32_t> findMorphemes(const FlatModel& m, const char16_t* s, size_t n, uint8_t tag)
	{
		std::vector<uint32_t> ret;
		// normalizeHangul (src/StrUtils.h:494-521): syllable with coda -> open syllable + coda jamo
		std::u16string nrm;
		for (size_t i = 0; i < n; ++i)
		{
			char16_t c = s[i];
			if (c == 0xB42C) c = 0xB410;
			if (0xAC00 <= c && c < 0xD7A4)
			{
				const int coda = (c - 0xAC00) % 28;
				nrm.push_back((char16_t)(c - coda));
				if (coda) nrm.push_back((char16_t)(coda + 0x11A7));
			}
			else nrm.push_back(c);
		}
		uint32_t node = 0;
		for (size_t i = 0; i < nrm.size(); ++i)
		{
			const uint16_t c = (uint16_t)nrm[i];
			if (node == 0) { node = m.trieRoot[c]; if (!node) return ret; continue; }
			const TrieNodeRec& t = m.trie[node];
			const uint16_t* kb = m.trieKeys.data() + t.edgeOff;
			const uint16_t* it = std::lower_bound(kb, kb + t.numNexts, c);
			if (it == kb + t.numNexts || *it != c) return ret;
			node = m.trieChild[t.edgeOff + (it - kb)];
		}
		if (node == 0 || m.trie[node].value < 0) return ret;      // no form ends here (or only a submatch does)
		const FormRec& f = m.forms[m.trie[node].value];
		for (uint32_t k = 0; k < f.candCnt; ++k)
		{
			const uint32_t mid = m.formCand[f.candOff + k];
			const MorphRec& r = m.morphs[mid];
			if (r.socket || (tag && (r.tag & 0x7F) != (tag & 0x7F))) continue;
			ret.push_back(mid);
		}
		return ret;
	}

	std::vector<uint32_t> blockBitsOf(const FlatModel& m, const std::vector<uint32_t>& ids)
	{
		const size_t nM = m.morphs.size();
		std::vector<uint32_t> raw((nM + 31) / 32, 0), eff((nM + 31) / 32, 0);
		for (uint32_t id : ids) if (id < nM) raw[id >> 5] |= 1u << (id & 31);
		auto has = [&](uint32_t id) { return id < nM && ((raw[id >> 5] >> (id & 31)) & 1); };
		for (size_t i = 0; i < nM; ++i)
		{
			const MorphRec& r = m.morphs[i];
			bool b = r.combinedId >= 0 && has((uint32_t)r.combinedId);
			for (uint32_t k = 0; k < r.nChunks && !b; ++k) b = has(m.chunkMorph[r.chunkOff + k]);
			if (b) eff[i >> 5] |= 1u << (i & 31);
		}
		return eff;
	}

	std::vector<uint8_t> dumpDict(const FlatModel& m)
	{
		std::vector<uint8_t> out;
		auto put = [&](const void* p, size_t n) { out.insert(out.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
		auto put32 = [&](uint32_t v) { put(&v, 4); };
		auto put16 = [&](uint16_t v) { put(&v, 2); };
		auto put8 = [&](uint8_t v) { put(&v, 1); };
		put32(m.h.nForms + 1); put32(m.h.nMorphs);
		for (uint32_t i = 0; i <= m.h.nForms; ++i)
		{
			const FormRec& f = m.forms[i];
			put32(f.len); put(m.formChars.data() + f.charOff, 2 * f.len);
			put32(f.numSpaces); put8(f.vowelPolar & 0xF); put8(f.vowelPolar >> 4); put8(f.formHash); put8(f.flags & 15); put16(0);
			put32(f.candCnt);
			for (uint32_t c = 0; c < f.candCnt; ++c) put32(m.formCand[f.candOff + c]);
		}
		for (uint32_t i = 0; i < m.h.nMorphs; ++i)
		{
			const MorphRec& o = m.morphs[i];
			put32(m.morphKform[i]); put8(o.tag); put8(o.vowel); put8(o.polar);
			put8((o.flags & MF_COMPLEX) ? 1 : 0); put8((o.flags & MF_SAISIOT) ? 1 : 0); put8(o.senseId); put8(o.socket);
			int32_t comb = o.combinedId - (int32_t)i; put(&comb, 4); put(&o.userScore, 4);
			put32(o.lmId); put32(0 /* origMorphemeId: not used on the analyze path */); put16(0);
			put32(o.nChunks);
			for (uint32_t c = 0; c < o.nChunks; ++c) { put32(m.chunkMorph[o.chunkOff + c]); put8(m.chunkPos[2 * (o.chunkOff + c)]); put8(m.chunkPos[2 * (o.chunkOff + c) + 1]); }
		}
		for (auto v : m.h.specialMorph) put32(v);
		return out;
	}
}
