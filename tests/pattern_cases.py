"""Seeded inputs for the pattern recognisers of the text preparation (URL, e-mail, mention, hashtag, number, serial, abbreviation, emoji)."""
import random

ALPHABET = list("0199abZA..,:/-@#%+_  \t(?~=htpscom") + [chr(c) for c in (0xAC00, 0xD55C, 0x11A8, 0xFF11, 0xFE0F, 0x200D, 0x2764, 0x263A, 0x00A9, 0x2122, 0x2A, 0x20E3)] \
    + ["\U0001F600", "\U0001F3FB", "\U0001F91D", "\U0001F468", "\ud83d", "\ude00"]
SEEDS = ["http://", "https://", "http://a.co", "ab@cd.com", "1,234.5", "12:30", "2020.01.02", "e.g. ", "U.S.A.", "@name", "#tag", "3.", "10.5.", "1,23", "a.b.c "]
ALL = 0xFFFFFFFF


def pattern_cases(n, seed):
    """n triples (unit left of the text, text, match options)."""
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        s = rng.choice(SEEDS) if rng.random() < 0.4 else ""
        s += "".join(rng.choice(ALPHABET) for _ in range(rng.randint(1, 14)))
        left = rng.choice(ALPHABET)[0]
        match = ALL if rng.random() < 0.8 else rng.getrandbits(32)
        out.append((left, s, match))
    return out
