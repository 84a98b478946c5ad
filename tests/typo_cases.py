"""Shared by tests/test_typo_oracle.py and tools/make_golden.py: this repo's own typo rules (every branch of the reference's rule
expansion: vowel / onset jamo expansion, left conditions any / vowel / applosive / continual / boundary, an infinite-cost
exclusion, dialect-tagged rules) and texts dense in the patterns those rules and the built-in sets look for."""
import random

INF = float("inf")
COND = {"none": 0, "any": 1, "vowel": 2, "vocalic": 3, "vocalic_h": 4, "non_vowel": 5, "non_vocalic": 6, "non_vocalic_h": 7, "applosive": 8, "continual": 9, "boundary": 10}

RULES = [   # (origs, errors, cost, left condition, dialect bits)
    (["ㅐ", "ㅔ"], ["ㅐ", "ㅔ"], 1.0, "none", 0), (["ㅚ", "ㅙ"], ["ㅞ", "ㅐ"], 1.5, "none", 0), (["ㅟ", "ㅢ"], ["ㅣ"], 1.0, "none", 0),
    (["위", "의"], ["이"], INF, "none", 0), (["위", "의"], ["이"], 1.0, "any", 0), (["자", "쟈"], ["자", "쟈"], 1.0, "none", 0),
    (["ᆻ어"], ["ᆺ어", "ᆺ서"], 1.0, "none", 0), (["ᆫᄒ"], ["ᆫᄒ", "ᆭᄋ"], 2.0, "none", 0), (["ᄒ"], ["ᄋ"], 0.5, "vowel", 0),
    (["ᄒ", "ᄀ"], ["ᄏ", "ᄁ"], 1.0, "applosive", 0), (["ᆨᄋ"], ["ᄀ"], 1.0, "continual", 0), (["ᆫᄋ"], ["ᄂ"], 1.0, "continual", 0),
    (["ᆯᄋ"], ["ᄅ"], 1.0, "continual", 0), (["시어"], ["셔"], 0.25, "boundary", 8), (["지어"], ["져"], 0.25, "boundary", 0),
    (["안"], ["않"], 1.5, "none", 0), (["돼"], ["되"], 1.0, "none", 0), (["던"], ["든"], 1.0, "none", 16),
]
HAND_TEXTS = ["외않됀데 궨찮아", "됬어요 왠지 몰라도 어의없네", "구지 그렇게 해야되?", "먹었읍니다 먹었서요 했어 했서", "있따가 갈께 않되 안돼",
              "국어 국이 산이 물이 달아", "가시어 가셔 지어 져 던지 든지", "학교에 각하 악하다 막히다", "희망 의사 위치 쥐", "", "ㅐ", "안",
              "앗뿔싸 그럼 오늘부터 다시 열심히 해보자꾸나.", "그럼 내괴다룄네"]
SYLLABLES = "안않돼되왜외웨의위이희히쥐지어져셔시가각학국산물달던든했었읍습니다요서어아자쟈괴내레뢰"


def fill(transformer, as_names):
    """Feeds RULES through `add(orig, error, cost, cond, dialect)`; as_names: pass the condition by name (refbridge) or by value."""
    for origs, errs, cost, cond, dia in RULES:
        for o in origs:
            for e in errs:
                transformer.add(o, e, cost, cond if as_names else COND[cond], dia)


def texts(n, seed):
    rnd = random.Random(seed)
    return HAND_TEXTS + ["".join(rnd.choice(SYLLABLES + "  ") for _ in range(rnd.randint(1, 60))) for _ in range(n)]


CODA2ONSET = {1: 0, 4: 2, 7: 3, 8: 5, 16: 6, 17: 7, 19: 9, 22: 12, 23: 14, 24: 15, 25: 16, 26: 17, 27: 18}


LENGTHENING_VOWEL = [0, 1, 0, 1, 4, 5, 4, 5, 8, 0, 1, 1, 8, 13, 4, 5, 20, 13, 18, 20, 20]


def misspell(text, rnd, vowels=True, carry=True, lengthen=False):
    """Injects the kinds of errors the built-in typo sets correct into a text of the synthetic model: confusable vowels (ㅐ/ㅔ, ㅚ/ㅙ),
    a coda written as the onset of the following vowel-initial syllable (연철, what the continual rules undo), and 1-3 syllables that
    merely lengthen the vowel of an open syllable ("가아아", what the lengthening cost pays for)."""
    o = list(text)
    if vowels:
        for i, ch in enumerate(o):
            c = ord(ch)
            if 0xAC00 <= c < 0xD7A4 and rnd.random() < 0.15:
                v = (c - 0xAC00) // 28 % 21
                if v == 1: c += 4 * 28
                elif v == 5: c -= 4 * 28
                elif v == 11: c -= 1 * 28
                o[i] = chr(c)
    if carry:
        for i in range(len(o) - 1):
            a, b = ord(o[i]), ord(o[i + 1])
            if 0xAC00 <= a < 0xD7A4 and 0xAC00 <= b < 0xD7A4 and rnd.random() < 0.5:
                coda = (a - 0xAC00) % 28
                onset = (b - 0xAC00) // 28 // 21
                if coda in CODA2ONSET and onset == 11:
                    o[i] = chr(a - coda)
                    o[i + 1] = chr(b + (CODA2ONSET[coda] - 11) * 21 * 28)
    if lengthen:
        p = []
        for ch in o:
            p.append(ch)
            c = ord(ch)
            if 0xAC00 <= c < 0xD7A4 and (c - 0xAC00) % 28 == 0 and rnd.random() < 0.2:
                p.append(chr(0xAC00 + (11 * 21 + LENGTHENING_VOWEL[(c - 0xAC00) // 28 % 21]) * 28) * rnd.randint(1, 3))
        o = p
    return "".join(o)
