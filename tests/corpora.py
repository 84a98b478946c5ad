"""Shared test inputs: synthetic sentences plus hand-written edge cases (patterns, brackets, quotes, numbers,
multi-sentence texts, emoji, surrogates, empty / whitespace-only inputs)."""

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
