#!/usr/bin/env python3
"""Writes tests/golden/pattern_golden.json: pattern-like strings (seeded random mixtures over the alphabet the recognisers care about) with what the
REAL reference's matchPattern (src/PatternMatcher.cpp:380, through oracle/_ref) answers at their first unit.  Run in the build container (needs
/root/reference compiled into oracle/_ref); the JSON travels, the reference does not."""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refbridge
from pattern_cases import pattern_cases
items = []
for left, text, match in pattern_cases(3000, 77):
    n, tag = refbridge.match_pattern(left, text, match)
    items.append({"left": ord(left), "units": [int.from_bytes(text.encode("utf-16-le", errors="surrogatepass")[i:i + 2], "little") for i in range(0, 2 * len(text.encode("utf-16-le", errors="surrogatepass")) // 2, 2)], "match": match, "len": n, "tag": tag})
json.dump({"source": "kiwi::matchPattern of /root/reference (oracle/_ref/libkiwi_ref.so), tools/make_golden_patterns.py", "items": items},
          open(os.path.join(ROOT, "tests", "golden", "pattern_golden.json"), "w"), separators=(",", ":"))
print(len(items), "cases,", sum(1 for i in items if i["len"]), "matches")
