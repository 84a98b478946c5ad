"""Shared test inputs: synthetic sentences plus hand-written edge cases (patterns, brackets, quotes, numbers,
multi-sentence texts, emoji, surrogates, empty / whitespace-only inputs)."""
import random

EDGE_TEXTS = [
    "가나다\x00", "abc\x00def 가\x00",      # U+0000 inside a text (the reference's own C test hands the terminator over as part of each line)
    "",
    " ",
    "   \n\t ",
    ".",
    "?!",
    "가",
    "가.",
    "ㅋㅋㅋ",
    "123",
    "3.14 와 1,234,567 그리고 12:30 에 2020-01-02.",
    "http://example.com/a?b=c 와 foo.bar@test.co.kr 과 #해시태그 @mention_1 을",
    "Mr. Kim 과 e.g. 그리고 U.S.A. 는",
    "(괄호 안) [대괄호] {중괄호} <꺾쇠>",
    "'작은따옴표' \"큰따옴표\" 그리고 ‘둥근’ “따옴표”",
    "\"열린 인용. 다음 문장! 셋째?\" 밖.",
    "첫 문장. 둘째 문장. 셋째 문장? 넷째! 다섯째…",
    "줄\n바꿈\r\n포함\n\n두 줄 띄움",
    "1. 첫째 2. 둘째 가. 항목 (1) 괄호",
    "english words and 한글 mixed ＡＢＣ 全角 漢字 かな",
    "이모지 😀 와 👍🏽 그리고 👨‍👩‍👧 가족",
    "\ud83d 깨진 \ude00 서로게이트",
    "됬다 했닼ㅋㅋ 앜",
    "가나다라마바사아자차카타파하" * 8,
    "a" * 70 + " " + "가" * 70,
    "· ~ - … ― ※ ★ ♥ → ∼ ⟪⟫",
    "끝에 공백   ",
    "   앞에 공백",
    "탭\t과  여러   공백",
    "zero​width‍joiner ✊‍🏽",
]


def synthetic(sm, n, seed, **kw):
    return sm.make_corpus(n, seed, **kw)


def dictionary_mix(sm, n, seed):
    """Sentences mixing dictionary words with punctuation, digits, quotes and brackets."""
    import numpy as np
    rng = np.random.default_rng(seed)
    base = sm.make_corpus(n, seed + 1, min_jamo=5, max_jamo=60)
    out = []
    extras = ["123", "4.5", "(", ")", "\"", "'", ",", "~", "abc", "Dr.", "1,000", "#태그", "…", "·", "[", "]", "“", "”", "2.", "가.", "\n", "!", "? "]
    for s in base:
        words = s.split(" ")
        k = int(rng.integers(0, 4))
        for _ in range(k):
            pos = int(rng.integers(0, len(words) + 1))
            e = extras[int(rng.integers(0, len(extras)))]
            if rng.random() < 0.5 and pos < len(words):
                words[pos] = e + words[pos] if rng.random() < 0.5 else words[pos] + e
            else:
                words.insert(pos, e)
        out.append(" ".join(words))
    return out


_FUZZ_FRAGS = ["http://a.b/c?d=1", "https://www.example.com/path", "user@mail.com", "@handle", "#tag", "#한글태그", "010-1234-5678", "1,234.5", "3.14",
               "2024-09-24", "12:30", "AB-123", "a.b.c", "e.g.", "U.S.A.", "Mr.", "...", "!?", "~~", "(", ")", "[", "]", "{", "}", "'", '"', "‘", "’", "“", "”",
               "「", "」", "·", "…", "ㅋㅋㅋ", "ㅠㅠ", "😀", "👍🏽", "👨‍👩‍👧", "🇰🇷", "‍", "️", "①", "Ⅳ", "㈜", "㎏", "ＡＢＣ", "１２３", "日本語", "中文", "русский",
               "العربية", "ελληνικά", "\t", "\n", "\r\n", "  ", "　", "\xa0", "A", "z", "0", "9", "-", "_", "+", "=", "*", "&", "%", "$", "₩", "@", "#", ".",
               ",", ";", ":", "/", "\\", "|", "^", "`", "<", ">", "ᄀ", "ᅡ", "ᆨ", "가", "힣", "\ud800", "\udc00", "\ud83d", "\ude00x"]


def fuzzed(sm, n, seed):
    """Random mixtures of dictionary words, pattern-like fragments (URLs, e-mail, hashtags, mentions, serials, numbers), symbols,
    emoji sequences, other scripts, jamo, lone surrogates and random code points: stresses normalisation, character typing, the
    pattern recognisers, chunking and the special-character nodes of the lattice."""
    import random
    rng = random.Random(seed)
    words = [w for t in dictionary_mix(sm, 200, seed + 1) for w in t.split()]
    out = []
    for _ in range(n):
        parts = []
        for _ in range(rng.randint(1, 12)):
            c = rng.random()
            if c < 0.45:
                parts.append(rng.choice(words))
            elif c < 0.85:
                parts.append(rng.choice(_FUZZ_FRAGS))
            else:
                parts.append("".join(chr(rng.choice([rng.randint(0x20, 0x7e), rng.randint(0xac00, 0xd7a3), rng.randint(0x1100, 0x11ff),
                                                     rng.randint(0x2000, 0x2bff), rng.randint(0x3000, 0x33ff)])) for _ in range(rng.randint(1, 4))))
            if rng.random() < 0.6:
                parts.append(" ")
        out.append("".join(parts))
    return out


def pick_blocklist(analyzer, texts, k, any_tag_every=4):
    """A blocklist for `texts`: the k dictionary morphemes that occur most often in their (unconstrained) analyses by `analyzer`, as
    [(form, tag id or -1)] for kiwi_morphset_add -- every `any_tag_every`-th item without a tag (= every morpheme of that form)."""
    from collections import Counter
    cnt = Counter()
    for t in texts:
        for tok in analyzer.analyze(t)[0][0]:
            if tok.morph_id and 1 <= (tok.tag & 0x7F) <= 20 or 39 <= (tok.tag & 0x7F) <= 52:      # content words, particles, endings
                cnt[(tok.form, tok.tag & 0x7F)] += 1
    items = [it for it, _ in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0]))[:k]]
    return [(f, -1 if i % any_tag_every == any_tag_every - 1 else t) for i, (f, t) in enumerate(items)]


def force_lanes(monkeypatch, lanes):
    """KAMD_GROUP_LANES for a test's engine: a lane-group width forces the general search kernel k_best_path<lanes>; "pos" leaves the engine's own
    choice in place -- top-1 searches then run the position-step kernel k_pos_path first (KAMD_WPS=3 selects its three-waves build); "pos8": its build with
    eight chunks per wavefront (8-lane groups)."""
    if lanes in ("pos", "pos8"):
        monkeypatch.delenv("KAMD_GROUP_LANES", raising=False)
        monkeypatch.setenv("KAMD_POS_PATH", "2")      # (also for typo correction, where the engine's own choice is the general kernel)
        if lanes == "pos8":
            monkeypatch.setenv("KAMD_POS_G", "8")     # k_pos_path<8, .>: eight chunks per wavefront (the engine's own choice is 16-lane groups)
        else:
            monkeypatch.delenv("KAMD_POS_G", raising=False)
    else:
        monkeypatch.setenv("KAMD_GROUP_LANES", lanes)
        monkeypatch.delenv("KAMD_POS_PATH", raising=False)
        monkeypatch.delenv("KAMD_POS_G", raising=False)


def repeated_unknown_texts(sm, n, seed):
    """Texts in which unknown words recur (what Match::oovChrFreqModel is about): a few made-up words and dictionary words, repeated, between known text."""
    rnd = random.Random(seed)
    base = synthetic(sm, n, seed, min_jamo=5, max_jamo=60)
    syll = "가나다라마바사아자차카타파하거너더러머버서어저고노도로모보소오조구누두루무부수우주"
    out = []
    for i, t in enumerate(base):
        made = ["".join(rnd.choice(syll) for _ in range(rnd.randint(2, 5))) for _ in range(rnd.randint(1, 3))]
        parts = t.split(" ")
        for _ in range(rnd.randint(2, 8)):
            parts.insert(rnd.randint(0, len(parts)), rnd.choice(made) + rnd.choice(["", "", "은", "를", "이"]))
        out.append(" ".join(parts) + ("" if i % 4 else " " + t))
    return out


def free_port():
    """A TCP port nobody listens on right now (rendezvous of a torch.distributed.run job: the test files run side by side under pytest-xdist)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
