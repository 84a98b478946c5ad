"""Deterministic synthetic model + corpus generator.

The real Kiwi model binaries are absent from the reference checkout (git-LFS
pointers), so parity and throughput are measured on a synthetic model that has
the *shape* of a Kiwi model:

* the reserved default forms / morphemes laid out exactly as the reference's
  ``KiwiBuilder::initMorphemes`` does (``/root/reference/src/KiwiBuilder.cpp:1108-1131``),
* a dictionary of forms over normalised Hangul (CV syllable blocks + split-out
  codas, ``src/StrUtils.h:494-521``) with homonyms, allomorphs, left-condition
  features, pre-combined (chunked) morphemes, split irregular stems
  (``combineSocket``), complex nouns and forms containing spaces,
* a Kneser-Ney style n-gram model serialised in the reference's ``sj.knlm``
  layout (``src/Knlm.hpp:1003-1167``), estimated from a synthetic morpheme
  corpus so that the lattice search has a realistic score landscape.

The output is a "raw model" section container (see ``container.py``) holding
what ``KiwiBuilder`` holds right before ``build()`` bakes it; both the product's
host-side baker (``kiwi_amd/csrc/model.cpp``) and the reference bridge
(``oracle/ref_bridge.cpp``) consume that same file.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field

import numpy as np

from .container import write_container

# --- POSTag numbering: /root/reference/include/kiwi/Types.h:195-227 -------------------------
(UNKNOWN, NNG, NNP, NNB, VV, VA, MAG, NR, NP, VX, MM, MAJ, IC, XPN, XSN, XSV, XSA, XSM, XR,
 VCP, VCN, SF, SP, SS, SSO, SSC, SE, SO, SW, SB, SL, SH, SN, W_URL, W_EMAIL, W_MENTION,
 W_HASHTAG, W_SERIAL, W_EMOJI, JKS, JKC, JKG, JKO, JKB, JKV, JKQ, JX, JC, EP, EF, EC, ETN, ETM,
 Z_CODA, Z_SIOT, USER0, USER1, USER2, USER3, USER4, P, TAG_MAX) = range(62)
PV, PA = P, P + 1
IRREGULAR = 0x80
DEFAULT_TAG_SIZE = P                      # Types.h:257
DEFAULT_FORM_SIZE = DEFAULT_TAG_SIZE + 26  # KiwiBuilder.cpp:40
N_DEFAULT_MORPH = DEFAULT_FORM_SIZE + 3    # KiwiBuilder.cpp:1111

# CondVowel / CondPolarity: Types.h:263-288
(CV_NONE, CV_ANY, CV_VOWEL, CV_VOCALIC, CV_VOCALIC_H, CV_NON_VOWEL, CV_NON_VOCALIC,
 CV_NON_VOCALIC_H, CV_APPLOSIVE) = range(9)
CP_NONE, CP_POSITIVE, CP_NEGATIVE, CP_NON_ADJ = range(4)

MORPH_DTYPE = np.dtype([
    ("kform", "<u4"), ("lm_id", "<u4"), ("orig_id", "<u4"), ("combined", "<i4"),
    ("user_score", "<f4"), ("chunk_ptr", "<u4"), ("tag", "u1"), ("vp_pack", "u1"),
    ("sense_id", "u1"), ("socket", "u1"), ("dialect", "<u2"), ("n_chunks", "u1"), ("pad", "u1"),
])
assert MORPH_DTYPE.itemsize == 32

SEED_BASE = 0x4B495749  # "KIWI" 

TAG_NAMES = ("UN", "NNG", "NNP", "NNB", "VV", "VA", "MAG", "NR", "NP", "VX", "MM", "MAJ", "IC", "XPN", "XSN", "XSV", "XSA", "XSM", "XR", "VCP", "VCN",
             "SF", "SP", "SS", "SSO", "SSC", "SE", "SO", "SW", "SB", "SL", "SH", "SN", "W_URL", "W_EMAIL", "W_MENTION", "W_HASHTAG", "W_SERIAL", "W_EMOJI",
             "JKS", "JKC", "JKG", "JKO", "JKB", "JKV", "JKQ", "JX", "JC", "EP", "EF", "EC", "ETN", "ETM", "Z_CODA", "Z_SIOT")


def tag_id(name: str):
    """Tag id of a tag as the reference prints it ("VV", "VV-I" = irregular, "VV-R" = marked regular); None when it is not a tag of the table."""
    base, _, suffix = name.partition("-")
    if base not in TAG_NAMES:
        return None
    return TAG_NAMES.index(base) | (IRREGULAR if suffix == "I" else 0)


# compatibility jamo that stand for a coda (U+11A8 + index), in the order of the coda jamo block
_COMPAT_CODA = (0x3131, 0x3132, 0x3133, 0x3134, 0x3135, 0x3136, 0x3137, 0x3139, 0x313A, 0x313B, 0x313C, 0x313D, 0x313E, 0x313F, 0x3140,
                0x3141, 0x3142, 0x3144, 0x3145, 0x3146, 0x3147, 0x3148, 0x314A, 0x314B, 0x314C, 0x314D, 0x314E)


def cv(onset: int, vowel: int) -> str:
    return chr(0xAC00 + (onset * 21 + vowel) * 28)


def coda(c: int) -> str:  # c in 1..27
    return chr(0x11A7 + c)


def is_coda(ch: str) -> bool:
    return 0x11A8 <= ord(ch) <= 0x11C2


def is_syll(ch: str) -> bool:
    return 0xAC00 <= ord(ch) < 0xD7A4


def join_hangul(norm: str) -> str:
    """Inverse of the coda split for well-formed strings (syllable + coda -> composed)."""
    out = []
    for ch in norm:
        o = ord(ch)
        if is_coda(ch) and out and is_syll(out[-1]) and (ord(out[-1]) - 0xAC00) % 28 == 0:
            out[-1] = chr(ord(out[-1]) + (o - 0x11A7))
        else:
            out.append(ch)
    return "".join(out)


def normalize_hangul(text: str) -> str:
    out = []
    for ch in text:
        o = ord(ch)
        if o == 0xB42C:
            o = 0xB410
        if 0xAC00 <= o < 0xD7A4:
            c = (o - 0xAC00) % 28
            out.append(chr(o - c))
            if c:
                out.append(chr(0x11A7 + c))
        else:
            out.append(chr(o))
    return "".join(out)


@dataclass
class Morph:
    kform: int
    tag: int
    vowel: int = CV_NONE
    polar: int = CP_NONE
    complex: bool = False
    sense_id: int = 0
    socket: int = 0
    combined: int = 0
    user_score: float = 0.0
    lm_id: int = 0
    orig_id: int = 0
    chunks: list = field(default_factory=list)
    chunk_pos: list = field(default_factory=list)
    dialect: int = 0


class RawModel:
    def __init__(self):
        self.forms: list[str] = []
        self.form_cands: list[list[int]] = []
        self.form_map: dict[str, int] = {}
        self.morphs: list[Morph] = []
        self.vocab_size = 0
        self.knlm: bytes = b""
        self.sbg: bytes = b""          # optional: SkipBigramModel blob
        self.cong: bytes = b""         # optional: CoNgram model blob (cong.mdl layout)
        self.nounchr: bytes = b""      # optional: character-level CoNgram model for unknown-form scoring (nounchr.mdl layout)
        self._init_defaults()

    # KiwiBuilder::initMorphemes (KiwiBuilder.cpp:1108-1131)
    def _init_defaults(self):
        self.forms = [""] * DEFAULT_FORM_SIZE
        self.form_cands = [[] for _ in range(DEFAULT_FORM_SIZE)]
        self.morphs = [Morph(0, UNKNOWN) for _ in range(N_DEFAULT_MORPH)]
        for i in range(1, DEFAULT_TAG_SIZE):
            self.form_cands[i - 1].append(i + 1)
            self.morphs[i + 1].tag = i
        for i in range(27):
            f = i + DEFAULT_TAG_SIZE - 1
            m = i + DEFAULT_TAG_SIZE + 1
            self.form_cands[f].append(m)
            self.forms[f] = chr(0x11A8 + i)
            self.morphs[m].tag = Z_CODA
            self.morphs[m].kform = f
            self.morphs[m].user_score = -1.5
        siot = 0x11BA - 0x11A8
        self.form_cands[DEFAULT_TAG_SIZE + siot - 1].append(DEFAULT_TAG_SIZE + 28)
        m = self.morphs[DEFAULT_TAG_SIZE + 28]
        m.tag = Z_SIOT
        m.kform = DEFAULT_TAG_SIZE + siot - 1
        m.user_score = -1.5

    def form_id(self, s: str) -> int:
        fid = self.form_map.get(s)
        if fid is None:
            fid = len(self.forms)
            self.forms.append(s)
            self.form_cands.append([])
            self.form_map[s] = fid
        return fid

    def add_morph(self, s: str, tag: int, **kw) -> int:
        fid = self.form_id(s)
        mid = len(self.morphs)
        self.morphs.append(Morph(fid, tag, **kw))
        self.form_cands[fid].append(mid)
        return mid

    def sections(self) -> dict:
        form_ptr = np.zeros(len(self.forms) + 1, "<u4")
        chars = []
        for i, s in enumerate(self.forms):
            chars.extend(ord(c) for c in s)
            form_ptr[i + 1] = len(chars)
        cand_ptr = np.zeros(len(self.forms) + 1, "<u4")
        cands = []
        for i, c in enumerate(self.form_cands):
            cands.extend(c)
            cand_ptr[i + 1] = len(cands)
        rec = np.zeros(len(self.morphs), MORPH_DTYPE)
        chunk_ids, chunk_pos = [], []
        for i, m in enumerate(self.morphs):
            r = rec[i]
            r["kform"] = m.kform
            r["lm_id"] = m.lm_id
            r["orig_id"] = m.orig_id
            r["combined"] = m.combined
            r["user_score"] = m.user_score
            r["chunk_ptr"] = len(chunk_ids)
            r["tag"] = m.tag
            r["vp_pack"] = (m.vowel & 0xF) | ((m.polar & 7) << 4) | (0x80 if m.complex else 0)
            r["sense_id"] = m.sense_id
            r["socket"] = m.socket
            r["dialect"] = m.dialect
            r["n_chunks"] = len(m.chunks)
            chunk_ids.extend(m.chunks)
            for a, b in m.chunk_pos:
                chunk_pos.extend((a, b))
        meta = np.array([len(self.forms), len(self.morphs), self.vocab_size, 0], "<u4")
        return {
            "meta": meta,
            "form_ptr": form_ptr,
            "form_chars": np.array(chars, "<u2"),
            "form_cand_ptr": cand_ptr,
            "form_cand": np.array(cands, "<u4"),
            "morph": rec,
            "chunk_ids": np.array(chunk_ids, "<u4"),
            "chunk_pos": np.array(chunk_pos, "u1"),
            **({"knlm": np.frombuffer(self.knlm, "u1")} if self.knlm else {}),
            **({"sbg": np.frombuffer(self.sbg, "u1")} if self.sbg else {}),
            **({"cong": np.frombuffer(self.cong, "u1")} if getattr(self, "cong", None) else {}),
            **({"nounchr": np.frombuffer(self.nounchr, "u1")} if getattr(self, "nounchr", None) else {}),
        }

    def save(self, path: str):
        write_container(path, self.sections(), kind=b"KAMDRAW1")


# ---------------------------------------------------------------------------------------------
def _zipf_weights(n: int, s: float = 1.0) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), s)
    return w / w.sum()


class _Lex:
    """Word lists per grammatical class with Zipf sampling weights."""

    def __init__(self):
        self.items: dict[str, list[int]] = {}
        self.cum: dict[str, np.ndarray] = {}
        self.rare: set[int] = set()          # morphemes whose weight is scaled down (SynthSpec.homonym_skew)

    def add(self, cls: str, mid: int):
        self.items.setdefault(cls, []).append(mid)

    def finalize(self, skew: float = 0.0):
        for k, v in self.items.items():
            w = _zipf_weights(len(v))
            if skew > 0 and self.rare:
                w = w * np.where(np.fromiter((m in self.rare for m in v), bool, len(v)), np.exp(-skew), 1.0)
                w = w / w.sum()
            self.cum[k] = np.cumsum(w)

    def sample(self, cls: str, rng) -> int:
        v = self.items[cls]
        return v[min(int(np.searchsorted(self.cum[cls], rng.random())), len(v) - 1)]


def _ends_with_coda(s: str) -> bool:
    return bool(s) and is_coda(s[-1])


def _is_positive(s: str) -> bool:
    # last vowel of the stem in {ㅏ,ㅑ,ㅗ,ㅛ} (FeatureTestor.cpp:60-78 restated for CV strings)
    for ch in reversed(s):
        if is_coda(ch):
            continue
        if not is_syll(ch):
            break
        v = ((ord(ch) - 0xAC00) // 28) % 21
        return v in (0, 2, 8, 12)
    return False


def _vowel_ok(left: str, cond: int) -> bool:
    """FeatureTestor::isMatched(begin,end,CondVowel) for our alphabet (FeatureTestor.cpp:6-58)."""
    if cond == CV_NONE:
        return True
    if not left:
        return False
    if cond == CV_ANY:
        return True
    e = ord(left[-1])
    if not (0xAC00 <= e <= 0xD7A4) and not (0x11A8 <= e <= 0x11C2):
        return True
    if cond in (CV_VOWEL, CV_VOCALIC, CV_VOCALIC_H):
        if cond == CV_VOCALIC_H and e == 0x11C2:
            return True
        if cond in (CV_VOCALIC_H, CV_VOCALIC) and e == 0x11AF:
            return True
        return not (0x11A8 <= e <= 0x11C2)
    if cond in (CV_NON_VOWEL, CV_NON_VOCALIC, CV_NON_VOCALIC_H):
        if cond == CV_NON_VOCALIC_H and e == 0x11C2:
            return False
        if cond in (CV_NON_VOCALIC_H, CV_NON_VOCALIC) and e == 0x11AF:
            return False
        return not (0xAC00 <= e <= 0xD7A4)
    return False


@dataclass
class SynthSpec:
    n_words: int = 3000          # open-class dictionary entries (nouns/verbs/adverbs)
    n_josa: int = 40
    n_eomi: int = 90
    n_contract: int = 300        # pre-combined (stem + ending) surface forms
    n_irregular: int = 40        # split irregular stems (combineSocket)
    n_complex: int = 150
    n_spaced: int = 60           # forms containing a space
    homonym_rate: float = 0.12
    # > 0: a homonym added to an existing surface form is sampled e^-skew times as often as its Zipf rank says (language-model corpus and test corpora
    # alike).  Homonyms of one class at neighbouring ranks score within a fraction of a nat of each other, which a SkipBigram search -- whose paths do
    # not merge for eight words -- turns into 2^k paths per node (86 x the states of the Knlm search on the 'full' lexicon, `_data/sbg_probe.py`); in a
    # real model one reading of a homograph dominates in a given context
    homonym_skew: float = 0.0
    # > 0: that share of the nouns of the language-model corpus is replaced by the default NNG morpheme (an out-of-vocabulary noun as the search
    # records it), a tenth of it by the default NNP one: the two unknown-noun readings of an unknown form then differ by ~ ln 10 instead of tying
    lm_unk_rate: float = 0.0
    lm_sentences: int = 20000
    lm_order: int = 3
    use_htx: bool = False
    knlm_qbits: int = 0          # > 0: the Knlm blob is quantised to that many bits (reference KnLangModelHeader::quantized, Knlm.hpp:398-455, 1036-1061)
    knlm_compress: bool = False  # node sizes in the variable-length code qe::QCode<0, 2, 8, 16> (quantized |= 0x80; 16-bit keys only)
    use_sbg: bool = False        # also emit a SkipBigram model (reference skipbigram.mdl layout) over the same vocabulary
    use_cong: bool = False       # also emit a local (window 0), 8-bit CoNgram model (reference cong.mdl layout) over the same vocabulary
    cong_dim: int = 32
    cong_only: bool = False      # with use_cong: no Knlm blob in the container (the layout of the reference's models/cong/base: sj.morph + cong.mdl)
    cong_key_size: int = 4       # CoNgramModelHeader::keySize: 2 / 4 = 16- / 32-bit trie keys, 3 = 16-bit keys with ids >= 63488 spelt as two "surrogate" keys (src/CoNgramModel.hpp:271-300)
    cong_qbit: int = 8           # 4: embeddings packed two per byte with one 8-bit local scale / zero point per cong_qgroup values (src/CoNgramModel.cpp:378-400)
    cong_qgroup: int = 0
    use_nounchr: bool = False    # also emit a character-level CoNgram model (reference nounchr.mdl: Match::oovChrModel scores unknown forms with it, src/UnkFormScorer.cpp)
    cong_window: int = 0         # > 0: the file also carries the sections of the global model (confidences, distant embeddings, mask), as the reference's builder always writes them
    extra_words: tuple = ()      # ((form, tag name[, Dialect bits]), ...): real-text dictionary entries added to the generated lexicon (the eval_data parity corpus, workloads.eval_model)
    seed: int = SEED_BASE


FULL_SPEC = SynthSpec(n_words=118000, n_josa=90, n_eomi=260, n_contract=9000, n_irregular=400,
                      n_complex=3000, n_spaced=800, lm_sentences=400000, lm_order=4)  # order 4, the reference's maximum and what SURVEY.md section 8(d) names (rounds 1-4: order 3 -- build_knlm packed an n-gram into 63 bits)
FULL_SBG_SPEC = SynthSpec(n_words=118000, n_josa=90, n_eomi=260, n_contract=9000, n_irregular=400,
                          n_complex=3000, n_spaced=800, lm_sentences=400000, lm_order=4, use_sbg=True,   # FULL_SPEC's lexicon + skip-bigram tables (32-bit keys)
                          homonym_skew=4.0, lm_unk_rate=0.03)      # round 4: one reading of a homograph dominates, unknown NNG / NNP readings differ (a SkipBigram search needs both to prune: see the fields)
SMALL_SPEC = SynthSpec()
SMALL_ORDER4_SPEC = SynthSpec(lm_order=4, lm_sentences=60000)   # an order-4 Knlm (the reference's maximum, include/kiwi/Kiwi.h:611): contexts of three words, back-off chains one node longer than what a search state carries
SMALL_Q8_SPEC = SynthSpec(knlm_qbits=8, knlm_compress=True)   # SMALL_SPEC with the Knlm file as the reference ships it: 8-bit quantised, node sizes compressed
SMALL_Q5_SPEC = SynthSpec(knlm_qbits=5)                       # ... and a bit width that exercises the generic fixed-length bit stream
SMALL_HTX_SPEC = SynthSpec(use_htx=True)                       # SMALL_SPEC with a history-transformed Knlm (tag histories: what the reference's builder writes by default)
SMALL_HTX_Q8_SPEC = SynthSpec(use_htx=True, knlm_qbits=8, knlm_compress=True)   # ... and quantised / compressed on top: the shape of a shipped sj.knlm
SMALL_CONG_CHR_SPEC = SynthSpec(use_cong=True, use_nounchr=True)   # SMALL_CONG_SPEC + the character model of Match::oovChrModel (the reference loads it quantised with CoNgram model types only)
SMALL_SBG_SPEC = SynthSpec(use_sbg=True)   # same lexicon / Knlm as SMALL_SPEC (same seed) + skip-bigram tables
# the CoNgram file the way the reference's builder writes it for a large vocabulary: 4-bit grouped embeddings, variable-length 16-bit keys, window sections
MID_CONG_VL4_SPEC = SynthSpec(n_words=66000, use_cong=True, cong_only=True, cong_key_size=3, cong_qbit=4, cong_qgroup=8, cong_window=7)   # (> 65536 morphemes: an LM id is a morpheme id, and the reference sizes its root table by the vocabulary while indexing it with 16-bit keys)
# the CoNgram file with the global model's sections (window 7): scored as ModelType::congGlobal where asked for; 32-bit and 16-bit ids (the reference's state hash reads 8 BYTES of history)
SMALL_CONG_GLOBAL_SPEC = SynthSpec(use_cong=True, cong_window=7)
SMALL_CONG_GLOBAL16_SPEC = SynthSpec(use_cong=True, cong_window=7, cong_key_size=2)
SMALL_CONG_SPEC = SynthSpec(use_cong=True) # same lexicon as SMALL_SPEC + a local CoNgram model (the Knlm blob stays in the container: the dictionary bake needs a vocabulary size)
FULL_CONG_SPEC = SynthSpec(n_words=118000, n_josa=90, n_eomi=260, n_contract=9000, n_irregular=400,
                           n_complex=3000, n_spaced=800, lm_sentences=400000, lm_order=3, use_cong=True, cong_dim=64, cong_only=True)


# grammatical class (of the sentence grammar below) a caller-given dictionary entry joins, by tag
_CLASS_OF_TAG = {NNG: "noun", NNP: "noun", NNB: "noun", NP: "noun", NR: "noun", XR: "noun", VV: "verb", VX: "verb", VA: "adj", MAG: "adv", MAJ: "adv", IC: "adv",
                 MM: "det", XPN: "xpn", XSN: "xsn", XSV: "xsv", XSA: "xsa", XSM: "xsm", VCP: "vcp", VCN: "vcn", JKS: "josa", JKC: "josa", JKG: "josa", JKO: "josa",
                 JKB: "josa", JKV: "josa", JKQ: "josa", JX: "josa", JC: "josa", EP: "ep", EF: "ef", EC: "ec", ETN: "etn", ETM: "etm"}


class SynthModel:
    """Generates the raw model and keeps the generative grammar for corpus sampling."""

    def __init__(self, spec: SynthSpec = SMALL_SPEC):
        self.spec = spec
        self.rng = np.random.default_rng(spec.seed)
        self.raw = RawModel()
        self.lex = _Lex()
        self.contract: dict[tuple[int, int], list[str]] = {}   # (stem, eomi) -> surface strings
        self.irregular: dict[int, tuple[int, int]] = {}        # full verb -> (left part, socket)
        self.socket_right: dict[int, dict[int, str]] = {}      # socket -> {eomi -> right surface}
        self._build_lexicon()
        self._build_lm()

    # -- random strings ----------------------------------------------------------------------
    def _syll(self, coda_p=0.35) -> str:
        r = self.rng
        s = cv(int(r.integers(0, 19)), int(r.integers(0, 21)))
        if r.random() < coda_p:
            # common codas are more likely (ㄴ ㄹ ㅁ ㅇ ㄱ ㅂ ㅅ)
            if r.random() < 0.8:
                s += coda(int(r.choice([4, 8, 16, 21, 1, 17, 19])))
            else:
                s += coda(int(r.integers(1, 28)))
        return s

    def _word(self, lens=(1, 2, 3, 4), p=(0.10, 0.45, 0.33, 0.12), coda_p=0.35) -> str:
        n = int(self.rng.choice(lens, p=p))
        return "".join(self._syll(coda_p) for _ in range(n))

    def _new_word(self, **kw) -> str:
        for _ in range(100):
            w = self._word(**kw)
            if w not in self.raw.form_map:
                return w
        return w

    # -- lexicon -----------------------------------------------------------------------------
    def _build_lexicon(self):
        sp, r, raw, lex = self.spec, self.rng, self.raw, self.lex
        open_tags = [NNG, NNP, NNB, NP, NR, VV, VA, VX, MAG, MAJ, MM, IC, XR, VV | IRREGULAR, VA | IRREGULAR]
        open_p = np.array([0.44, 0.14, 0.02, 0.01, 0.01, 0.13, 0.07, 0.01, 0.08, 0.01, 0.03, 0.01, 0.02, 0.01, 0.01])
        open_p = open_p / open_p.sum()
        cls_of = {NNG: "noun", NNP: "noun", NNB: "noun", NP: "noun", NR: "noun", VV: "verb", VA: "adj",
                  VX: "verb", MAG: "adv", MAJ: "adv", MM: "det", IC: "adv", XR: "noun",
                  VV | IRREGULAR: "verb", VA | IRREGULAR: "adj"}
        existing: list[str] = []
        for _ in range(sp.n_words):
            tag = int(r.choice(open_tags, p=open_p))
            homonym = False
            if existing and r.random() < sp.homonym_rate:
                s = existing[int(r.integers(0, len(existing)))]
                if any(raw.morphs[m].tag == tag for m in raw.form_cands[raw.form_map[s]]):
                    s = self._new_word()
                else:
                    homonym = True
            else:
                s = self._new_word()
            kw = {}
            if r.random() < 0.03:
                kw["user_score"] = float(np.float32(r.choice([-3.0, -1.0, 0.5, 2.0])))
            if r.random() < 0.05:
                kw["sense_id"] = int(r.integers(1, 4))
            mid = raw.add_morph(s, tag, **kw)
            existing.append(s)
            lex.add(cls_of[tag], mid)
            if homonym:
                lex.rare.add(mid)

        # affixes
        for tag, n, cls in ((XPN, 6, "xpn"), (XSN, 14, "xsn"), (XSV, 4, "xsv"), (XSA, 4, "xsa"), (XSM, 3, "xsm")):
            for _ in range(n):
                lex.add(cls, raw.add_morph(self._new_word(lens=(1, 2), p=(0.7, 0.3)), tag))
        lex.add("vcp", raw.add_morph(cv(11, 20), VCP))     # 이
        lex.add("vcn", raw.add_morph(cv(11, 0) + cv(2, 20), VCN))  # 아니

        # josa: allomorph pairs conditioned on the left coda
        josa_tags = [JKS, JKC, JKG, JKO, JKB, JKV, JKQ, JX, JC]
        n_pairs = sp.n_josa // 3
        for _ in range(n_pairs):
            tag = int(r.choice(josa_tags))
            a = raw.add_morph(self._new_word(lens=(1, 2), p=(0.75, 0.25), coda_p=0.2), tag, vowel=CV_NON_VOWEL)
            b = raw.add_morph(self._new_word(lens=(1, 2), p=(0.75, 0.25), coda_p=0.2), tag, vowel=CV_VOWEL)
            raw.morphs[b].lm_id = a          # allomorph -> canonical (Form.h:75-80)
            lex.add("josa", a)
            self.contract.setdefault(("allo", a), []).append(b)
        for _ in range(sp.n_josa - 2 * n_pairs):
            tag = int(r.choice(josa_tags))
            if r.random() < 0.15:   # coda-initial josa, e.g. 'ᆫ', 'ᆯ'
                s = coda(int(r.choice([4, 8]))) + ("" if r.random() < 0.5 else self._syll(0.1))
                lex.add("josa", raw.add_morph(s, tag, vowel=CV_VOWEL))
            else:
                lex.add("josa", raw.add_morph(self._new_word(lens=(1, 2, 3), p=(0.5, 0.4, 0.1), coda_p=0.2), tag,
                                              vowel=int(r.choice([CV_NONE, CV_ANY]))))

        # eomi
        eomi_tags = [EP, EF, EC, ETN, ETM]
        eomi_cls = {EP: "ep", EF: "ef", EC: "ec", ETN: "etn", ETM: "etm"}
        ep_ = np.array([0.08, 0.3, 0.4, 0.07, 0.15])
        n_polar = sp.n_eomi // 5
        a_syll, eo_syll = cv(11, 0), cv(11, 4)           # 아 / 어
        for _ in range(n_polar):                         # 아/어 allomorph pairs
            tag = int(r.choice([EF, EC, EP], p=[0.3, 0.6, 0.1]))
            rest = "" if r.random() < 0.4 else self._word(lens=(1, 2), p=(0.7, 0.3), coda_p=0.15)
            if (eo_syll + rest) in raw.form_map and any(
                    raw.morphs[m].tag == tag for m in raw.form_cands[raw.form_map[eo_syll + rest]]):
                continue
            neg = raw.add_morph(eo_syll + rest, tag, polar=CP_NEGATIVE)
            pos = raw.add_morph(a_syll + rest, tag, polar=CP_POSITIVE)
            raw.morphs[pos].lm_id = neg
            lex.add(eomi_cls[tag], neg)
            self.contract.setdefault(("allo_polar", neg), []).append(pos)
        for _ in range(sp.n_eomi - 2 * n_polar):
            tag = int(r.choice(eomi_tags, p=ep_))
            u = r.random()
            if u < 0.25:      # coda-initial ending: needs a vowel-final stem
                s = coda(int(r.choice([4, 8, 17, 16]))) + ("" if r.random() < 0.3 else
                                                            self._word(lens=(1, 2), p=(0.7, 0.3), coda_p=0.15))
                vw = int(r.choice([CV_VOWEL, CV_VOCALIC]))
            elif u < 0.4:     # 으-initial style ending for consonant-final stems
                s = cv(11, 18) + self._word(lens=(1, 2), p=(0.7, 0.3), coda_p=0.15)
                vw = CV_NON_VOWEL
            else:
                s = self._new_word(lens=(1, 2, 3), p=(0.45, 0.4, 0.15), coda_p=0.15)
                vw = int(r.choice([CV_NONE, CV_ANY, CV_NON_VOCALIC, CV_VOCALIC_H]))
            pol = CP_NON_ADJ if r.random() < 0.06 else CP_NONE
            fid = raw.form_map.get(s)
            if fid is not None and any(raw.morphs[m].tag == tag for m in raw.form_cands[fid]):
                continue
            lex.add(eomi_cls[tag], raw.add_morph(s, tag, vowel=vw, polar=pol))

        # quotes registered as dictionary forms (KiwiBuilder.cpp:2642-2662)
        for q in ("'", '"'):
            for tag in (SSO, SSC, SS):
                raw.add_morph(q, tag)

        # dictionary entries given by the caller (real text): a form is normalised as the dictionary holds it (syllable + split-out coda; a
        # compatibility consonant such as the 'ㄴ' of the gold annotations is the coda jamo it stands for)
        for entry in sp.extra_words:
            form, tag_name = entry[0], entry[1]
            dialect = int(entry[2]) if len(entry) > 2 else 0      # (Dialect bits of the entry, MorphemeRaw::dialect; 0 = standard)
            tid = tag_id(tag_name)
            cls = _CLASS_OF_TAG.get(tid & 0x7F)
            if tid is None or cls is None:
                continue
            s = normalize_hangul("".join(chr(0x11A8 + _COMPAT_CODA.index(ord(ch))) if ord(ch) in _COMPAT_CODA else ch for ch in form))
            if not s or " " in s:
                continue
            fid = raw.form_map.get(s)
            if fid is not None and any(raw.morphs[m].tag == tid and raw.morphs[m].dialect in (0, dialect) for m in raw.form_cands[fid]):
                continue
            lex.add(cls, raw.add_morph(s, tid, vowel=CV_VOWEL if is_coda(s[0]) else CV_NONE, dialect=dialect))

        base_end = len(raw.morphs)

        # complex nouns: chunks + complex flag -> treated as single unless splitComplex (Form.h:174)
        nouns = lex.items["noun"]
        for _ in range(sp.n_complex):
            a, b = int(r.choice(nouns)), int(r.choice(nouns))
            sa, sb = raw.forms[raw.morphs[a].kform], raw.forms[raw.morphs[b].kform]
            s = sa + sb
            if s in raw.form_map:
                continue
            mid = raw.add_morph(s, NNG, complex=True, chunks=[a, b],
                                chunk_pos=[(0, len(sa)), (len(sa), len(s))])
            lex.add("noun", mid)
        # forms with a space inside (multi-word proper nouns), Form::numSpaces (Form.cpp:95)
        for _ in range(sp.n_spaced):
            s = self._new_word(lens=(1, 2), p=(0.4, 0.6)) + " " + self._new_word(lens=(1, 2), p=(0.4, 0.6))
            if s.replace(" ", "") in raw.form_map or s in raw.form_map:
                continue
            lex.add("noun", raw.add_morph(s, NNP))

        # split irregular stems: full verb V (in vocab), left part L/P (socket s, combined -> V)
        n_sockets = 4
        pieces = {}
        for s_id in range(1, n_sockets + 1):
            pieces[s_id] = raw.add_morph(coda(int([17, 7, 19, 8][s_id - 1])), PV, socket=s_id)  # ㅂ ㄷ ㅅ ㄹ
            self.socket_right[s_id] = {}
        verbs = lex.items["verb"] + lex.items["adj"]
        made = 0
        for v in list(r.permutation(verbs)):
            if made >= sp.n_irregular:
                break
            v = int(v)
            sv = raw.forms[raw.morphs[v].kform]
            if raw.morphs[v].tag & IRREGULAR == 0 or len(sv) < 2:
                # turn some regular stems with a matching coda into irregular ones
                if not _ends_with_coda(sv):
                    continue
            if not _ends_with_coda(sv) or " " in sv:
                continue
            s_id = int(r.integers(1, n_sockets + 1))
            left = sv[:-1]
            if not left or left in raw.form_map and any(
                    raw.morphs[m].socket for m in raw.form_cands[raw.form_map[left]]):
                continue
            lid = raw.add_morph(left, PV if raw.morphs[v].tag & 0x7F == VV else PA, socket=s_id)
            raw.morphs[lid].combined = v - lid
            self.irregular[v] = (lid, s_id)
            made += 1
        self.vocab_end_candidates = len(raw.morphs)

        # right parts for sockets: combined morpheme [piece/P, eomi] (KiwiBuilder.cpp:1618-1701)
        eomis = lex.items.get("ec", []) + lex.items.get("ef", []) + lex.items.get("etm", [])
        self._combined_start = len(raw.morphs)
        for s_id in range(1, n_sockets + 1):
            for e in r.permutation(eomis)[: max(4, len(eomis) // 6)]:
                e = int(e)
                se = raw.forms[raw.morphs[e].kform]
                if is_coda(se[0]):
                    continue
                surf = self._syll(0.0) + se[1:]     # e.g. ㅂ + 어 -> 워
                if surf in raw.form_map:
                    continue
                self._add_combined(surf, [pieces[s_id], e], [(0, 1), (0, len(surf))], socket=s_id,
                                   vowel=CV_NONE, score=0.0)
                self.socket_right[s_id][e] = surf
        # contractions stem+eomi -> new surface, expands to several tokens (PathEvaluator.hpp:1135-1153)
        stems = verbs
        tries = 0
        while sum(1 for k in self.contract if isinstance(k[0], int)) < sp.n_contract and tries < sp.n_contract * 4:
            tries += 1
            v, e = int(r.choice(stems)), int(r.choice(eomis))
            if (v, e) in self.contract:
                continue
            sv, se = raw.forms[raw.morphs[v].kform], raw.forms[raw.morphs[e].kform]
            if " " in sv or is_coda(se[0]):
                continue
            body = sv[:-1] if _ends_with_coda(sv) else sv
            surf = body[:-1] + self._syll(0.3) + se[1:]
            if not surf or surf in raw.form_map or is_coda(surf[0]):
                continue
            k = max(1, len(body) - 1)
            self._add_combined(surf, [v, e], [(0, min(k + 1, len(surf))), (k, len(surf))], socket=0,
                               vowel=CV_NONE, score=float(np.float32(r.choice([0.0, -0.5, -1.0]))))
            self.contract[(v, e)] = [surf]
        lex.finalize(sp.homonym_skew)
        self.base_end = base_end

    def _add_combined(self, surf, chunks, pos, socket, vowel, score):
        raw = self.raw
        mid = raw.add_morph(surf, UNKNOWN, chunks=list(chunks), chunk_pos=list(pos), socket=socket, vowel=vowel)
        m = raw.morphs[mid]
        m.user_score = float(np.float32(sum(raw.morphs[c].user_score for c in chunks) + score))
        return mid

    # -- language model ----------------------------------------------------------------------
    def sample_sentence(self, rng, n_eojeol=None):
        """Returns (morpheme-id sequence for the LM, list of eojeol surface strings (normalised))."""
        lex, raw = self.lex, self.raw
        n_eojeol = n_eojeol or int(rng.integers(2, 9))
        lm_seq, surf = [], []
        for ei in range(n_eojeol):
            last = ei == n_eojeol - 1
            u = rng.random()
            if last or u < 0.33:
                lm, s = self._verb_phrase(rng, final=last)
            elif u < 0.88:
                lm, s = self._noun_phrase(rng)
            else:
                m = lex.sample("adv" if rng.random() < 0.7 else "det", rng)
                lm, s = [m], raw.forms[raw.morphs[m].kform]
            lm_seq.extend(lm)
            surf.append(s)
        return lm_seq, surf

    def _lmid(self, m):
        mm = self.raw.morphs[m]
        return mm.lm_id if mm.lm_id else m

    def _attach(self, left: str, m: int, rng) -> tuple[int, str]:
        """Pick the allomorph of suffix ``m`` that is compatible with ``left``."""
        raw = self.raw
        cands = [m] + self.contract.get(("allo", m), []) + self.contract.get(("allo_polar", m), [])
        rng.shuffle(cands)
        for c in cands:
            mc = raw.morphs[c]
            if not _vowel_ok(left, mc.vowel):
                continue
            if mc.polar == CP_POSITIVE and not _is_positive(left):
                continue
            if mc.polar == CP_NEGATIVE and _is_positive(left):
                continue
            s = raw.forms[mc.kform]
            if is_coda(s[0]) and (not left or not is_syll(left[-1])):
                continue
            return c, s
        return -1, ""

    def _noun_phrase(self, rng):
        lex, raw = self.lex, self.raw
        lm, s = [], ""
        if rng.random() < 0.05:
            m = lex.sample("xpn", rng)
            lm.append(m)
            s += raw.forms[raw.morphs[m].kform]
        n = lex.sample("noun", rng)
        lm.append(n)
        s += raw.forms[raw.morphs[n].kform]
        if rng.random() < 0.1:
            m = lex.sample("xsn", rng)
            lm.append(m)
            s += raw.forms[raw.morphs[m].kform]
        if rng.random() < 0.75:
            for _ in range(4):
                j = lex.sample("josa", rng)
                c, js = self._attach(s, j, rng)
                if c >= 0:
                    lm.append(c)
                    s += js
                    break
        return lm, s

    def _verb_phrase(self, rng, final):
        lex, raw = self.lex, self.raw
        lm, s = [], ""
        u = rng.random()
        if u < 0.12:
            n = lex.sample("noun", rng)
            x = lex.sample("xsv" if rng.random() < 0.6 else "xsa", rng)
            lm += [n, x]
            s = raw.forms[raw.morphs[n].kform] + raw.forms[raw.morphs[x].kform]
            stem = x
        elif u < 0.18:
            n = lex.sample("noun", rng)
            x = lex.items["vcp"][0]
            lm += [n, x]
            s = raw.forms[raw.morphs[n].kform] + raw.forms[raw.morphs[x].kform]
            stem = x
        else:
            stem = lex.sample("verb" if rng.random() < 0.7 else "adj", rng)
            lm.append(stem)
            s = raw.forms[raw.morphs[stem].kform]
        if rng.random() < 0.15 and "ep" in lex.items:
            c, es = self._attach(s, lex.sample("ep", rng), rng)
            if c >= 0:
                lm.append(c)
                s += es
                stem = -1
        cls = "ef" if final else ("ec" if rng.random() < 0.7 else "etm")
        for _ in range(6):
            e = lex.sample(cls, rng)
            # contraction / irregular realisations only straight after the stem
            if stem >= 0 and len(lm) == 1:
                if (stem, e) in self.contract and rng.random() < 0.8:
                    return [stem, e], self.contract[(stem, e)][0]
                if stem in self.irregular:
                    lid, s_id = self.irregular[stem]
                    right = self.socket_right[s_id].get(e)
                    if right is not None and rng.random() < 0.8:
                        return [stem, e], raw.forms[raw.morphs[lid].kform] + right
            c, es = self._attach(s, e, rng)
            if c >= 0:
                lm.append(c)
                s += es
                break
        return lm, s

    def _build_lm(self):
        sp, raw = self.spec, self.raw
        # every morpheme that is not a combined (chunked, tag unknown) entry is in the LM vocabulary
        vocab = self._combined_start
        raw.vocab_size = vocab
        for i, m in enumerate(raw.morphs):
            if i >= vocab:
                m.lm_id = i                       # KiwiBuilder.cpp:1641
            elif m.lm_id == 0 and i > 0:
                m.lm_id = i                       # KiwiBuilder.cpp:906-921
        rng = np.random.default_rng(sp.seed + 1)
        sents = []
        sf_id = SF + 1
        noun_ids = set(self.lex.items.get("noun", ())) if sp.lm_unk_rate > 0 else set()
        urng = np.random.default_rng(sp.seed + 11)
        for _ in range(sp.lm_sentences):
            lm, _ = self.sample_sentence(rng)
            ids = [self._lmid(m) for m in lm]
            if noun_ids:
                for k, m in enumerate(lm):
                    if m in noun_ids:
                        u = urng.random()
                        if u < sp.lm_unk_rate:
                            ids[k] = NNG + 1 if u < 0.9 * sp.lm_unk_rate else NNP + 1      # (default tag morphemes: id = tag + 1)
            sents.append([0] + ids + [sf_id, 1])
        htx = None
        if sp.use_htx:
            htx = np.array([(raw.morphs[i].tag & 0x7F) + vocab for i in range(vocab)], dtype=np.int64)
        raw.knlm = build_knlm(sents, vocab, sp.lm_order, htx=htx)
        if sp.knlm_qbits or sp.knlm_compress:
            raw.knlm = quantize_knlm(raw.knlm, sp.knlm_qbits, sp.knlm_compress)
        if sp.use_sbg:
            raw.sbg = build_sbg(sents, vocab, key_size=2 if vocab + 1 <= 0xFFFF else 4, seed=sp.seed + 2)
        if sp.use_cong:
            raw.cong = build_cong(sents, vocab, dim=sp.cong_dim, seed=sp.seed + 3, key_size=sp.cong_key_size, qbit=sp.cong_qbit, qgroup=sp.cong_qgroup, window=sp.cong_window)
            if sp.cong_only:
                raw.knlm = b""
        if sp.use_nounchr:
            raw.nounchr = build_nounchr([f for f in raw.forms[DEFAULT_FORM_SIZE:] if f], seed=sp.seed + 4)

    # -- text corpus -------------------------------------------------------------------------
    def make_corpus(self, n, seed, min_jamo=5, max_jamo=200, exact_jamo=None, oov_rate=0.03, lognormal=None):
        """Synthetic sentences (composed Hangul text). Length is measured in non-space
        normalised units ("jamo")."""
        rng = np.random.default_rng(seed)
        out = []
        for _ in range(n):
            if lognormal is not None:      # (median, sigma): lengths ~ log-normal, clipped to [min_jamo, max_jamo]
                target = int(min(max_jamo, max(min_jamo, round(float(rng.lognormal(np.log(lognormal[0]), lognormal[1]))))))
            else:
                target = exact_jamo or int(rng.integers(min_jamo, max_jamo + 1))
            words, total = [], 0
            while total < target - 1:
                _, surf = self.sample_sentence(rng, n_eojeol=int(rng.integers(1, 4)))
                for w in surf:
                    if rng.random() < oov_rate:
                        w = "".join(self._rand_syll(rng) for _ in range(int(rng.integers(2, 6))))
                    ns = len(w.replace(" ", ""))
                    if total + ns > target - 1:
                        # fill the remainder with an OOV run of exactly the missing size
                        rem = target - 1 - total
                        w = "".join(cv(int(rng.integers(0, 19)), int(rng.integers(0, 21))) for _ in range(rem))
                        ns = rem
                    if ns:
                        words.append(w)
                        total += ns
                    if total >= target - 1:
                        break
            text = " ".join(words) + str(rng.choice([".", "?", "!"]))
            out.append(join_hangul(text))
        return out

    @staticmethod
    def _rand_syll(rng):
        s = cv(int(rng.integers(0, 19)), int(rng.integers(0, 21)))
        if rng.random() < 0.3:
            s += coda(int(rng.integers(1, 28)))
        return s


# ---------------------------------------------------------------------------------------------
def _pack_bits(codes, bits) -> bytes:
    """lm::FixedLengthEncoder<bits, uint32_t> (src/BitEncoder.hpp): value k occupies bits [k*bits, (k+1)*bits) of an LSB-first bit stream."""
    codes = np.asarray(codes, np.uint32)
    if bits == 8:
        return codes.astype(np.uint8).tobytes()
    b = ((codes[:, None] >> np.arange(bits, dtype=np.uint32)[None, :]) & 1).astype(np.uint8).reshape(-1)
    out = np.packbits(b, bitorder="little").tobytes()
    return out + b"\0" * ((-len(out)) % 4)      # (whole 32-bit packets)


def _qcode_encode(sizes):
    """qe::QCode<0, 2, 8, 16> (src/QEncoder.hpp:13-45, 92-135): two header bits per value name its class {0: the value 0, 1: 1..4 in 2 bits,
    2: 5..260 in 8 bits, 3: 261.. in 16 bits}; the bodies form one LSB-first bit stream of 64-bit words right behind the header bytes."""
    sizes = np.asarray(sizes, np.int64)
    q = (sizes >= 1).astype(np.int64) + (sizes >= 5) + (sizes >= 261)
    assert sizes.max() < 261 + 65536
    n = len(sizes)
    qp = np.concatenate([q, np.zeros((-n) % 4, np.int64)]).reshape(-1, 4)
    header = (qp[:, 0] | (qp[:, 1] << 2) | (qp[:, 2] << 4) | (qp[:, 3] << 6)).astype(np.uint8).tobytes()
    nbits = np.array([0, 2, 8, 16])[q]
    bias = np.array([0, 1, 5, 261])[q]
    e = sizes - bias
    bits = []
    for v, b in zip(e.tolist(), nbits.tolist()):
        bits.extend((v >> k) & 1 for k in range(b))
    body = np.packbits(np.array(bits, np.uint8), bitorder="little").tobytes() if bits else b""
    body += b"\0" * ((-len(body)) % 8 + 8)      # whole 64-bit words and one to spare (the reference's decoder may read a word ahead)
    return header + body


def quantize_knlm(blob: bytes, bits: int, compress: bool) -> bytes:
    """Re-writes an unquantised Knlm blob the way KnLangModelBase::build does with quantize = bits / compress (Knlm.hpp:688-790): log-likelihoods
    and back-off weights as `bits`-bit codes into two tables of 2^bits floats, node sizes optionally QCode-compressed."""
    (num_nodes, node_off, key_off, ll_off, gamma_off, qtable_off, htx_off, unk_id, bos_id, eos_id, vocab_size,
     order, key_size, diff_size, quantized, extra) = struct.unpack_from("<11Q4BI", blob, 0)
    assert quantized == 0
    kdt = "<u2" if key_size == 2 else "<u4"
    sizes = np.frombuffer(blob, kdt, num_nodes, node_off)
    n_nonleaf = int((sizes != 0).sum()); n_leaf = num_nodes - n_nonleaf
    keys = np.frombuffer(blob, kdt, num_nodes - 1, key_off)
    ll = np.frombuffer(blob, "<f4", num_nodes, ll_off)
    gamma = np.frombuffer(blob, "<f4", n_nonleaf, gamma_off)
    htx = None if not htx_off else np.frombuffer(blob, kdt, vocab_size, htx_off)

    def table_and_codes(x):
        if not bits:
            return None, None
        tab = np.quantile(x, np.linspace(0, 1, 1 << bits)).astype("<f4")
        tab = np.unique(tab)
        tab = np.concatenate([tab, np.full((1 << bits) - len(tab), tab[-1], "<f4")]).astype("<f4")
        idx = np.clip(np.searchsorted(tab, x), 1, len(tab) - 1)
        idx = np.where(np.abs(tab[idx - 1] - x) <= np.abs(tab[idx] - x), idx - 1, idx)
        return tab, idx.astype(np.uint32)
    ll_tab, ll_codes = table_and_codes(np.minimum(ll, np.float32(-1e-6)))      # (a table entry of +0 would turn a leaf's value into a child offset)
    gm_tab, gm_codes = table_and_codes(gamma)
    if bits:
        assert (ll_tab[ll_codes[n_nonleaf:]] < 0).all()      # leaf values are told apart from child offsets by their sign
    node_bytes = _qcode_encode(sizes) if compress else sizes.astype(kdt).tobytes()
    if compress:
        assert key_size == 2, "the reference decodes compressed node sizes into 16-bit slots"
    ll_bytes = _pack_bits(ll_codes, bits) if bits else ll.astype("<f4").tobytes()
    gm_bytes = _pack_bits(gm_codes, bits) if bits else gamma.astype("<f4").tobytes()

    def al(x):
        return (x + 15) & ~15
    off = 96
    node_o = off; off = al(off + len(node_bytes))
    key_o = off; off = al(off + keys.nbytes)
    ll_o = off; off = al(off + len(ll_bytes))
    gm_o = off; off = al(off + len(gm_bytes))
    qt_o = off
    if bits:
        off = al(off + 8 * (1 << bits))
    htx_o = 0
    if htx is not None:
        htx_o = off; off = al(off + htx.nbytes)
    buf = bytearray(off)
    struct.pack_into("<11Q4BI", buf, 0, num_nodes, node_o, key_o, ll_o, gm_o, qt_o if bits else off if htx is None else qt_o, htx_o,
                     unk_id, bos_id, eos_id, vocab_size, order, key_size, diff_size, (bits & 0x1F) | (0x80 if compress else 0), 0)
    buf[node_o:node_o + len(node_bytes)] = node_bytes
    buf[key_o:key_o + keys.nbytes] = keys.tobytes()
    buf[ll_o:ll_o + len(ll_bytes)] = ll_bytes
    buf[gm_o:gm_o + len(gm_bytes)] = gm_bytes
    if bits:
        buf[qt_o:qt_o + 4 * (1 << bits)] = ll_tab.tobytes()
        buf[qt_o + 4 * (1 << bits):qt_o + 8 * (1 << bits)] = gm_tab.tobytes()
    if htx is not None:
        buf[htx_o:htx_o + htx.nbytes] = htx.tobytes()
    return bytes(buf)


def build_knlm(sents, vocab_size, order, htx=None, discount=0.75, unk_id=2, bos_id=0, eos_id=1) -> bytes:
    """Interpolated Kneser-Ney estimate serialised in the reference's uncompressed,
    unquantised ``sj.knlm`` layout (reader: /root/reference/src/Knlm.hpp:1003-1167; header:
    include/kiwi/Knlm.h:9-15).

    ``htx`` (history transformer; the reference's builder uses one by default: useLmTagHistory, src/KiwiBuilder.cpp:1167-1174, ids = tag + vocab):
    the OLDEST token of a trie path is stored transformed, the rest raw (utils::countNgrams, src/count.hpp:148-240: the first edge from the root is
    makeNext(historyTx(w)), later ones makeNext(w)), so the key of an n-gram (w1 .. wn), n >= 2, is (htx[w1], w2, .., wn); unigrams stay raw."""
    if htx is not None:
        # the reference's loader reads the unknown word's score in the context of <s> BEFORE the suffix links exist (Knlm.hpp:1138-1141): the
        # bigram (<s>, <unk>) has to be in the model
        sents = list(sents) + [[bos_id, unk_id, eos_id]] * 3
    flat = np.concatenate([np.asarray(s, dtype=np.int64) for s in sents])
    lens = np.array([len(s) for s in sents], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    sid = np.repeat(np.arange(len(sents)), lens)
    hist = flat if htx is None else htx[flat]
    key_space = int(max(vocab_size, 0 if htx is None else int(htx.max()) + 1)) + 1
    bits = max(1, (key_space - 1).bit_length())
    # an n-gram is packed into ONE integer key: 64-bit where that holds `order` tokens, arbitrary-precision Python integers (numpy object arrays: the same
    # operators, unique / searchsorted by Python comparisons -- minutes instead of seconds for the 'full' models) where it does not (order 4 over 2^17 words)
    wide = bits * order > 63
    kt = object if wide else np.int64
    # n-gram tables: dict order -> (keys[n, order] sorted, counts)
    grams = {}
    n_tok = len(flat)
    for n in range(1, order + 1):
        idx = np.arange(n_tok - n + 1)
        ok = sid[idx] == sid[idx + n - 1]
        idx = idx[ok]
        cols = [flat[idx + k] for k in range(n)]
        if htx is not None and n >= 2:
            cols[0] = hist[idx]
        code = np.zeros(len(idx), dtype=kt)
        for c in cols:
            code = (code << bits) | (c.astype(kt) if wide else c)
        u, cnt = np.unique(code, return_counts=True)
        grams[n] = (u, cnt.astype(np.float64))

    def lead_tx(code, n):
        """key of the n-gram whose tokens `code` packs raw: its first token transformed when the model has a history transformer (n >= 2)"""
        if htx is None or n < 2:
            return code
        sh = bits * (n - 1)
        top = ((code >> sh) & ((1 << bits) - 1)).astype(np.int64)
        return (code & ((1 << sh) - 1)) | (htx[top].astype(np.int64).astype(kt) << sh)

    def split(code, n):
        cols = []
        for k in range(n):
            cols.append(((code >> (bits * (n - 1 - k))) & ((1 << bits) - 1)).astype(np.int64))
        return np.stack(cols, axis=1) if n else np.zeros((len(code), 0), np.int64)

    # continuation counts for lower orders: N1+(• w) of the (n+1)-grams whose suffix is this n-gram
    counts = {order: grams[order][1]}
    for n in range(order - 1, 0, -1):
        hi, _ = grams[n + 1]
        # suffix of an (n+1)-gram in *mixed* key space: history part transformed, last raw
        suffix = lead_tx(hi & ((1 << (bits * n)) - 1), n)
        u, c = np.unique(suffix, return_counts=True)
        cc = np.zeros(len(grams[n][0]))
        pos = np.searchsorted(grams[n][0], u)
        okm = (pos < len(grams[n][0]))
        okm[okm] &= grams[n][0][pos[okm]] == u[okm]
        cc[pos[okm]] = c[okm]
        # n-grams that start a sentence (begin with <s>) keep their real count
        first = (grams[n][0] >> (bits * (n - 1))) & ((1 << bits) - 1)
        bos_key = bos_id if htx is None else int(htx[bos_id])
        isbos = first == (bos_key if n > 1 else bos_id)
        cc = np.where(isbos | (cc == 0), np.maximum(cc, grams[n][1] * (isbos | (cc == 0))), cc)
        counts[n] = cc

    # probabilities, top-down recursion on the context
    uni_keys, uni_c = grams[1][0], counts[1]
    uni_total = uni_c.sum()
    p_uni = np.maximum(uni_c - discount, 0.05) / uni_total
    p_uni = p_uni / p_uni.sum() * (1 - 1e-4)
    prob = {1: p_uni}
    gamma = {}
    for n in range(2, order + 1):
        keys, c = grams[n][0], counts[n]
        ctx = keys >> bits
        uctx, inv = np.unique(ctx, return_inverse=True)
        tot = np.bincount(inv, weights=c)
        n1 = np.bincount(inv)
        g = discount * n1 / tot
        # lower-order probability of the same word in the shortened context
        low_key = lead_tx(keys & ((1 << (bits * (n - 1))) - 1), n - 1)
        if n - 1 == 1:
            lp_pos = np.searchsorted(grams[1][0], low_key)
            lower = prob[1][np.minimum(lp_pos, len(prob[1]) - 1)]
        else:
            # the shortened history is already in transformed space; the (n-1)-gram table shares it
            lp_pos = np.searchsorted(grams[n - 1][0], low_key)
            lp_pos = np.minimum(lp_pos, len(grams[n - 1][0]) - 1)
            found = grams[n - 1][0][lp_pos] == low_key
            lower = np.where(found, prob[n - 1][lp_pos], 1e-7)
        prob[n] = np.maximum(c - discount, 0) / tot[inv] + g[inv] * lower
        gamma[n - 1] = (uctx, g)

    # --- trie assembly: node = context sequence; children keyed by next id ---------------------
    # children of a context of length n-1 are the n-grams sharing it.  A child is a non-leaf node
    # iff it is itself a context of some (n+1)-gram.  Context keys live in history space except
    # that the *last* element of an n-gram key is raw; a child (ctx, w) as a context is (ctx, h(w)).
    nodes_children = {}   # ctx tuple -> dict key -> [ll, gamma or None]
    root = {}
    nodes_children[()] = root
    ll1 = np.log(prob[1])
    for k, l in zip(grams[1][0].tolist(), ll1.tolist()):
        root[int(k)] = [float(np.float32(l)), None]
    ctx_gamma = {}
    for n in range(1, order):
        uctx, g = gamma[n]
        cols = split(uctx, n)
        for row, gv in zip(cols.tolist(), g.tolist()):
            ctx_gamma[tuple(row)] = float(np.float32(math.log(max(gv, 1e-6))))
    for n in range(2, order + 1):
        cols = split(grams[n][0], n)
        lls = np.log(np.maximum(prob[n], 1e-12))
        for row, l in zip(cols.tolist(), lls.tolist()):
            nodes_children.setdefault(tuple(row[:-1]), {})[int(row[-1])] = [float(np.float32(l)), None]
    # make sure every context node exists as a child chain from the root (suffix/prefix closure)
    for ctx in sorted(ctx_gamma, key=len):
        for d in range(1, len(ctx) + 1):
            par = nodes_children.setdefault(tuple(ctx[:d - 1]), {})
            if ctx[d - 1] not in par:
                par[ctx[d - 1]] = [-20.0, None]
        nodes_children.setdefault(tuple(ctx), {})
    # a context must also have all its suffixes as contexts so that `lower` links resolve to real
    # nodes (Knlm.hpp:38-63)
    for ctx in sorted(list(nodes_children), key=len, reverse=True):
        for s in range(1, len(ctx)):
            suf = tuple(ctx[s:]) if htx is None else (int(htx[ctx[s]]),) + tuple(ctx[s + 1:])
            if suf not in nodes_children:
                nodes_children[suf] = {}
                for d in range(1, len(suf) + 1):
                    par = nodes_children.setdefault(tuple(suf[:d - 1]), {})
                    if suf[d - 1] not in par:
                        par[suf[d - 1]] = [-20.0, None]
    # drop empty contexts (they are leaves)
    nonleaf = {c for c, ch in nodes_children.items() if ch or c == ()}

    node_sizes, keys_out, ll_nonleaf, gamma_nonleaf, ll_leaf = [], [], [], [], []

    import sys
    sys.setrecursionlimit(10000)

    def emit(ctx, ll):
        ch = nodes_children[ctx]
        node_sizes.append(len(ch))
        ll_nonleaf.append(ll)
        gamma_nonleaf.append(ctx_gamma.get(ctx, -0.05) if ctx else 0.0)
        ks = sorted(ch)
        keys_out.extend(ks)
        for k in ks:
            child = ctx + (k,)
            if child in nonleaf and len(child) < order:
                emit(child, ch[k][0])
            else:
                node_sizes.append(0)
                ll_leaf.append(ch[k][0])

    emit((), 0.0)
    key_size = 2 if key_space <= 0xFFFF else 4
    kdt = "<u2" if key_size == 2 else "<u4"
    node_arr = np.array(node_sizes, kdt)
    assert max(node_sizes) < (1 << (8 * key_size))
    key_arr = np.array(keys_out, kdt)
    ll_arr = np.array(ll_nonleaf + ll_leaf, "<f4")
    # leaf values are stored as float bit patterns and told apart from child offsets by their sign
    assert (ll_arr[len(ll_nonleaf):] < 0).all()
    gm_arr = np.array(gamma_nonleaf, "<f4")
    htx_arr = None if htx is None else np.asarray(htx, kdt)

    def al(x):
        return (x + 15) & ~15
    off = 96
    node_off = off
    off = al(off + node_arr.nbytes)
    key_off = off
    off = al(off + key_arr.nbytes)
    ll_off = off
    off = al(off + ll_arr.nbytes)
    gamma_off = off
    off = al(off + gm_arr.nbytes)
    qtable_off = off
    htx_off = 0
    if htx_arr is not None:
        htx_off = off
        off = al(off + htx_arr.nbytes)
    buf = bytearray(off)
    struct.pack_into("<11Q4BI", buf, 0, len(node_sizes), node_off, key_off, ll_off, gamma_off, qtable_off,
                     htx_off, unk_id, bos_id, eos_id, vocab_size, order, key_size, 4, 0, 0)
    buf[node_off:node_off + node_arr.nbytes] = node_arr.tobytes()
    buf[key_off:key_off + key_arr.nbytes] = key_arr.tobytes()
    buf[ll_off:ll_off + ll_arr.nbytes] = ll_arr.tobytes()
    buf[gamma_off:gamma_off + gm_arr.nbytes] = gm_arr.tobytes()
    if htx_arr is not None:
        buf[htx_off:htx_off + htx_arr.nbytes] = htx_arr.tobytes()
    return bytes(buf)


def _svb_encode(values, v0124: bool) -> bytes:
    """Stream VByte (Lemire / Kurz / Rupp 2017; the coding of the reference's cong.mdl node sizes, keys and values): ceil(n/4) control bytes of
    four 2-bit length codes, then the significant bytes, little endian.  Standard codes 0..3 = 1, 2, 3, 4 bytes; '0124' codes = 0, 1, 2, 4 bytes."""
    v = np.asarray(values, dtype=np.uint32)
    n = len(v)
    if v0124:
        code = np.where(v == 0, 0, np.where(v < (1 << 8), 1, np.where(v < (1 << 16), 2, 3))).astype(np.uint8)
        nbytes = np.array([0, 1, 2, 4], np.uint8)[code]
    else:
        code = np.where(v < (1 << 8), 0, np.where(v < (1 << 16), 1, np.where(v < (1 << 24), 2, 3))).astype(np.uint8)
        nbytes = np.array([1, 2, 3, 4], np.uint8)[code]
    pad = (-n) % 4
    c4 = np.concatenate([code, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    ctrl = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8)
    le = v.astype("<u4").view(np.uint8).reshape(-1, 4)
    mask = np.arange(4)[None, :] < nbytes[:, None]
    return ctrl.tobytes() + le[mask].tobytes()


def _cong_vl_keys(tok, key_size):
    """The trie keys a token id is spelt with (CoNgramModel::progressContextNode, src/CoNgramModel.hpp:271-300): itself, or for keySize 3 and
    ids >= 63488 two 16-bit "surrogate" keys (63488 + high 10 bits, 63488 + 1024 + low 10 bits)."""
    if key_size != 3 or tok < 63488:
        return (tok,)
    r = tok - 63488
    return (63488 + (r >> 10), 63488 + 1024 + (r & 1023))


def _pack_u4_row(vals, qgroup, rng):
    """One embedding row in the 4-bit grouped packing (reader: requantizePackedInts qbit 4, src/CoNgramModel.cpp:388-395; unpacking
    src/archImpl/none.cpp:22-57 / sse4_1.cpp:480-573): dim/2 bytes of nibble pairs (low nibble first), fp16 global scale, dim/qgroup local bytes
    (bits 0-5: scale - 9, bits 6-7: zero point - 6).  `vals` are the target values; what is stored is their nearest representable code."""
    dim = len(vals)
    ng = dim // qgroup
    local = np.zeros(ng, np.uint8)
    nib = np.zeros(dim, np.uint8)
    for g in range(ng):
        v = vals[g * qgroup:(g + 1) * qgroup].astype(np.int64)
        zp = int(rng.integers(6, 10))
        span = max(int(np.abs(v).max()), 1)
        sc = int(min(63, max(9, round(span * 9 / 7))))      # (nibble - zp) in about -9 .. 9; value = that * sc / 9, |value| <= 63
        q = np.clip(np.round(v * 9 / sc).astype(np.int64) + zp, 0, 15)
        nib[g * qgroup:(g + 1) * qgroup] = q
        local[g] = ((zp - 6) << 6) | (sc - 9)
    packed = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8)
    return packed, local


def build_cong(sents, vocab_size, dim=32, max_ctx_len=2, seed=0, min_count=2, key_size=4, qbit=8, qgroup=0, window=0) -> bytes:
    """A synthetic *local* (window 0) CoNgram model in the reference's ``cong.mdl`` layout (reader: /root/reference/src/CoNgramModel.cpp:425-789;
    header: include/kiwi/CoNgramModel.h:18-34): 8-bit embeddings (qbit 8), 32-bit keys (keySize 4), no optional sections (flags 0).

    header 64 B | node sizes (Stream VByte 0124; pre-order stream, 0 = leaf) | keys (Stream VByte; per non-leaf node its sorted child keys) |
    values (Stream VByte 0124; per node of the stream its context id, 0 = inherit from the longest suffix) |
    per context: dim x int8, fp16 scale, fp16 (-bias) | per vocabulary word: dim x int8, fp16 scale.

    The context trie holds the histories (token sequences of length <= max_ctx_len + 1 seen >= min_count times) of the LM training sentences;
    every node is a context with its own embedding row.  Scores are not a trained model's: random embeddings scaled so that a transition costs
    roughly -2 .. -12 like a log-probability -- the arithmetic (u8 x s8 dot product, hsum correction, two scales, bias) is what is exercised."""
    rng = np.random.default_rng(seed)
    flat = np.concatenate([np.asarray(s, dtype=np.int64) for s in sents])
    lens = np.array([len(s) for s in sents], dtype=np.int64)
    sid = np.repeat(np.arange(len(sents)), lens)
    depth = max_ctx_len + 1
    bits = max(1, int(vocab_size).bit_length())
    assert bits * depth <= 63
    children = {(): {}}                       # history tuple (of trie KEYS) -> {next key: True}
    n_tok = len(flat)
    assert key_size in (2, 3, 4) and (key_size != 2 or vocab_size <= 0xFFFF) and (key_size != 3 or vocab_size <= 63488 + (1 << 20))
    assert qbit == 8 or (qbit == 4 and qgroup in (4, 8, 16) and dim % 16 == 0)
    for n in range(1, depth + 1):
        idx = np.arange(n_tok - n + 1)
        idx = idx[sid[idx] == sid[idx + n - 1]]
        code = np.zeros(len(idx), dtype=np.int64)
        for k in range(n):
            code = (code << bits) | flat[idx + k]
        u, cnt = np.unique(code, return_counts=True)
        u = u[cnt >= (1 if n == 1 else min_count)]
        for c in u.tolist():
            toks = tuple((c >> (bits * (n - 1 - k))) & ((1 << bits) - 1) for k in range(n))
            hist = tuple(k for t in toks[:-1] for k in _cong_vl_keys(t, key_size))
            if hist not in children:
                continue
            for k in _cong_vl_keys(toks[-1], key_size):      # (a two-key token adds an inner node for its first key)
                children[hist][k] = True
                hist = hist + (k,)
                children.setdefault(hist, {})
    node_sizes, keys_out, values = [], [], []
    n_ctx = 1                                  # context 0: the empty / unknown context

    def emit(h):
        nonlocal n_ctx
        ch = children[h]
        node_sizes.append(len(ch))
        if h and rng.random() < 0.85 and not (key_size == 3 and 63488 <= h[-1] < 63488 + 1024):      # (the node of a first "surrogate" key is no context)
            values.append(n_ctx); n_ctx += 1
        else:
            values.append(0)                   # the root, and a share of the inner nodes: the context of the longest suffix applies
        ks = sorted(ch)
        keys_out.extend(ks)
        for k in ks:
            g = h + (k,)
            if children.get(g):
                emit(g)
            else:
                node_sizes.append(0)
                values.append(n_ctx); n_ctx += 1      # a leaf always names a context (its value is stored negated: must be non-zero)
    import sys
    sys.setrecursionlimit(10000)
    emit(())
    num_nodes = len(node_sizes)
    assert len(keys_out) == num_nodes - 1 and n_ctx < (1 << 24)

    def half(x):
        return np.asarray(x, np.float16).view(np.uint16)
    # value range: the reference's SSE4.1 / AVX2 kernels form the u8 x s8 dot product with pmaddubsw, which adds two products with SIGNED 16-bit
    # SATURATION; |value| <= 63 keeps every pair sum below 2^15 (2 * 191 * 63), so that the exact integer dot product is also theirs
    ctx_emb = rng.integers(-63, 64, size=(n_ctx, dim), dtype=np.int8)
    ctx_scale = half(rng.uniform(0.008, 0.024, n_ctx))
    ctx_negbias = half(rng.uniform(3.0, 9.0, n_ctx))                 # stored as -bias
    out_emb = rng.integers(-63, 64, size=(vocab_size, dim), dtype=np.int8)
    out_scale = half(rng.uniform(0.008, 0.024, vocab_size))
    emb = bytearray()

    def row(vals, scale16):
        if qbit == 8:
            return vals.tobytes() + scale16.tobytes()
        packed, local = _pack_u4_row(vals, qgroup, rng)
        # qbit 4: the reader's scale is globalScale / 8 (src/archImpl/sse4_1.cpp:572) -- the stored fp16 is 8 x the wanted scale
        return packed.tobytes() + half(np.float32(np.float16(scale16.view(np.float16)) * np.float16(8))).tobytes() + local.tobytes()
    for i in range(n_ctx):
        emb += row(ctx_emb[i], ctx_scale[i]) + ctx_negbias[i].tobytes()
        if window:
            emb += half(rng.uniform(-1.0, 1.0)).tobytes() + half(rng.uniform(0.0, 1.0)).tobytes()      # confidence, valid-token sum (global model only)
    for i in range(vocab_size):
        emb += row(out_emb[i], out_scale[i])
    if window:
        for i in range(vocab_size):          # distant embeddings: row, fp16 -bias, fp16 confidence
            emb += row(out_emb[(i * 7 + 3) % vocab_size], out_scale[i]) + half(rng.uniform(3.0, 9.0)).tobytes() + half(rng.uniform(-1.0, 1.0)).tobytes()
        emb += half(rng.uniform(-1.0, 1.0, window)).tobytes()      # position confidences
        emb += rng.integers(0, 256, (vocab_size + 7) // 8, dtype=np.uint8).tobytes()      # distant-token mask
    def al(x):
        return (x + 15) & ~15
    node_b, key_b, val_b = _svb_encode(node_sizes, True), _svb_encode(keys_out, False), _svb_encode(values, True)
    node_off = 64
    key_off = al(node_off + len(node_b))
    val_off = al(key_off + len(key_b))
    emb_off = al(val_off + len(val_b))
    buf = bytearray(al(emb_off + len(emb)))
    struct.pack_into("<QQHHBBBBQQQQQ", buf, 0, vocab_size, n_ctx, dim, 0, key_size, window, qbit, qgroup, num_nodes, node_off, key_off, val_off, emb_off)
    buf[node_off:node_off + len(node_b)] = node_b
    buf[key_off:key_off + len(key_b)] = key_b
    buf[val_off:val_off + len(val_b)] = val_b
    buf[emb_off:emb_off + len(emb)] = emb
    return bytes(buf)


CHR_VOCAB = 530      # ChrTokenizer::Token::max (include/kiwi/Dataset.h:144-153)


def chr_token(c: int, special_type) -> int:
    """ChrTokenizer::encodeOne (src/Dataset.cpp:805-847) of one UTF-16 unit; special_type(c) = identifySpecialChr as a POSTag name."""
    if 0xAC00 <= c < 0xD7A4:
        return 10 + (c - 0xAC00) // 28
    if 0x11A8 <= c <= 0x11C2:
        return 10 + 399 + (c - 0x11A8)
    if 0x21 <= c < 0x7F:
        return 10 + 399 + 27 + (c - 0x21)
    return {"sf": 1, "sp": 2, "ss": 3, "sso": 4, "ssc": 5, "se": 6, "so": 7, "sh": 9}.get(special_type(c), 8)


def build_nounchr(forms, dim=32, max_ctx_len=2, seed=0) -> bytes:
    """A synthetic character-level CoNgram model in the layout of the reference's ``nounchr.mdl`` (writer: src/CoNgramModel.cpp:2087-2400, reader
    :425-789): keySize 1 -- the trie's keys are BYTES, a (reordered) token id >= 192 is spelt as two of them (192 + high 5 bits, 224 + low 5 bits:
    CoNgramModel::progressContextNode, src/CoNgramModel.hpp:271-300) --, node sizes and keys as plain bytes, values Stream VByte, and all three
    optional sections (flags 7): one frequency byte per trie node behind the values, an 8-bit output bias + its fp16 minimum, the token
    reordering (u16 per token), an fp16 entropy per context.  Trained on nothing: the trie holds the character n-grams of the dictionary's own
    forms, embeddings / biases are random in the range of log-probabilities."""
    rng = np.random.default_rng(seed)
    V = CHR_VOCAB
    inv = np.arange(V)
    inv[1:] = 1 + rng.permutation(V - 1)          # token -> context key space (0 = BOS / EOS stays)

    def keys_of(tok):
        k = int(inv[tok])
        if k < 192:
            return (k,)
        r = k - 192
        return (192 + (r >> 5), 224 + (r & 31))

    def toks(form):
        out = []
        for ch in form:
            c = ord(ch)
            if 0xAC00 <= c < 0xD7A4:
                coda = (c - 0xAC00) % 28
                out.append(10 + (c - coda - 0xAC00) // 28)
                if coda:
                    out.append(10 + 399 + coda - 1)
            elif 0x21 <= c < 0x7F:
                out.append(10 + 399 + 27 + c - 0x21)
            else:
                out.append(8)
        return out
    children = {(): {}}
    for f in forms:
        t = [0] + toks(f) + [0]
        for i in range(len(t)):
            for n in range(1, max_ctx_len + 2):
                if i + n > len(t):
                    break
                hist = tuple(k for x in t[i:i + n - 1] for k in keys_of(x))
                if hist not in children:
                    break
                for k in keys_of(t[i + n - 1]):
                    children[hist][k] = True
                    hist = hist + (k,)
                    children.setdefault(hist, {})
    node_sizes, keys_out, values, freqs = [], [], [], []
    n_ctx = 1

    def emit(h):
        nonlocal n_ctx
        ch = children[h]
        assert len(ch) < 256
        node_sizes.append(len(ch))
        freqs.append(int(rng.integers(0, 200)))
        if h and rng.random() < 0.85 and not (192 <= h[-1] < 224):      # (the node of a first "surrogate" byte is no context)
            values.append(n_ctx); n_ctx += 1
        else:
            values.append(0)
        ks = sorted(ch)
        keys_out.extend(ks)
        for k in ks:
            g = h + (k,)
            if children.get(g):
                emit(g)
            else:
                node_sizes.append(0)
                freqs.append(int(rng.integers(0, 200)))
                values.append(n_ctx); n_ctx += 1
    import sys
    sys.setrecursionlimit(10000)
    emit(())
    num_nodes = len(node_sizes)
    assert len(keys_out) == num_nodes - 1 and n_ctx < (1 << 24)

    def half(x):
        return np.asarray(x, np.float16).view(np.uint16)
    ctx_emb = rng.integers(-63, 64, size=(n_ctx, dim), dtype=np.int8)
    ctx_scale = half(rng.uniform(0.008, 0.024, n_ctx))
    ctx_negbias = half(rng.uniform(2.0, 6.0, n_ctx))
    out_emb = rng.integers(-63, 64, size=(V, dim), dtype=np.int8)
    out_scale = half(rng.uniform(0.008, 0.024, V))
    emb = bytearray()
    for i in range(n_ctx):
        emb += ctx_emb[i].tobytes() + ctx_scale[i].tobytes() + ctx_negbias[i].tobytes()
    for i in range(V):
        emb += out_emb[i].tobytes() + out_scale[i].tobytes()
    emb += rng.integers(0, 256, V, dtype=np.uint8).tobytes() + half(-3.5).tobytes()      # output bias codes, their minimum (maximum 0)
    emb += inv.astype("<u2").tobytes()
    emb += half(rng.uniform(0.0, 4.0, n_ctx)).tobytes()                                     # context entropies

    def al(x):
        return (x + 15) & ~15
    node_b, key_b, val_b = bytes(node_sizes), bytes(keys_out), _svb_encode(values, True)
    node_off = 64
    key_off = al(node_off + len(node_b))
    val_off = al(key_off + len(key_b))
    freq_off = al(val_off + len(val_b))
    emb_off = al(freq_off + num_nodes)
    buf = bytearray(al(emb_off + len(emb)))
    struct.pack_into("<QQHHBBBBQQQQQ", buf, 0, V, n_ctx, dim, 7, 1, 0, 8, 0, num_nodes, node_off, key_off, val_off, emb_off)
    buf[node_off:node_off + len(node_b)] = node_b
    buf[key_off:key_off + len(key_b)] = key_b
    buf[val_off:val_off + len(val_b)] = val_b
    buf[freq_off:freq_off + num_nodes] = bytes(freqs)
    buf[emb_off:emb_off + len(emb)] = emb
    return bytes(buf)


def build_sbg(sents, vocab_size, key_size=2, window=8, max_keys_per_word=48, seed=0) -> bytes:
    """Skip-bigram tables in the reference's uncompressed, unquantised ``skipbigram.mdl`` layout (reader:
    /root/reference/src/SkipBigramModel.hpp:40-105; header: include/kiwi/SkipBigramModel.h:9-13):
    header | kSizes[vocab] | keyData[total] (sorted per word) | discnts[vocab] f32 | compensations[total] f32 | validness[vocab] u8.
    For a word ``next``, keyData holds the history words h (within the window) it was seen after, compensations the
    log-likelihood-like value the reference substitutes for that pair, discnts[h] the discount added to the Knlm score."""
    rng = np.random.default_rng(seed)
    flat = np.concatenate([np.asarray(s, dtype=np.int64) for s in sents])
    lens = np.array([len(s) for s in sents], dtype=np.int64)
    sid = np.repeat(np.arange(len(sents)), lens)
    pairs = []
    for d in range(1, window + 1):
        ok = sid[d:] == sid[:-d]
        h, w = flat[:-d][ok], flat[d:][ok]
        pairs.append(h * (vocab_size + 1) + w)
    code, cnt = np.unique(np.concatenate(pairs), return_counts=True)
    h_all, w_all = code // (vocab_size + 1), code % (vocab_size + 1)
    keep = (h_all > 2) & (w_all > 2) & (cnt >= 2)          # no bos/eos/unk rows
    h_all, w_all, cnt = h_all[keep], w_all[keep], cnt[keep].astype(np.float64)
    uni = np.bincount(flat, minlength=vocab_size).astype(np.float64) + 1.0
    order = np.lexsort((h_all, w_all))
    h_all, w_all, cnt = h_all[order], w_all[order], cnt[order]
    k_sizes = np.zeros(vocab_size, np.int64)
    keys, comps = [], []
    starts = np.searchsorted(w_all, np.arange(vocab_size + 1))
    for w in range(vocab_size):
        a, b = starts[w], starts[w + 1]
        if a == b:
            continue
        hs, cs = h_all[a:b], cnt[a:b]
        if len(hs) > max_keys_per_word:
            top = np.sort(np.argsort(-cs, kind="stable")[:max_keys_per_word])
            hs, cs = hs[top], cs[top]
        k_sizes[w] = len(hs)
        keys.append(hs)
        # log P(w | h within the window), clipped to the range the reference's gate (ll > -13) lets through
        comps.append(np.clip(np.log(cs / (uni[hs] * window)), -12.5, -0.05))
    key_arr = np.concatenate(keys) if keys else np.zeros(0, np.int64)
    cmp_arr = np.concatenate(comps) if comps else np.zeros(0, np.float64)
    valid = (k_sizes > 0).astype(np.uint8)
    # a few words are valid without any pair (the discount still applies to them as history), a few frequent ones invalid
    extra = rng.random(vocab_size) < 0.05
    valid[(k_sizes == 0) & extra & (np.arange(vocab_size) > 2)] = 1
    discnts = np.where(valid > 0, -np.abs(rng.normal(0.7, 0.3, vocab_size)) - 0.05, 0.0)
    kdt = "<u2" if key_size == 2 else "<u4"
    assert k_sizes.max(initial=0) < (1 << (8 * key_size))
    head = struct.pack("<Q8B", vocab_size, key_size, window, 0, 0, 0, 0, 0, 0)
    return head + k_sizes.astype(kdt).tobytes() + key_arr.astype(kdt).tobytes() + discnts.astype("<f4").tobytes() \
        + cmp_arr.astype("<f4").tobytes() + valid.tobytes()
