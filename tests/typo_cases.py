"""Shared by tests/test_typo_oracle.py and tools/make_golden.py: this repo's own typo rules (every branch of the reference's rule
expansion: vowel / onset jamo expansion, left conditions any / vowel / applosive / continual / boundary, an infinite-cost
exclusion, dialect-tagged rules) and texts dense in the patterns those rules and the built-in sets look for."""
import random

from kiwi_amd.workloads import COND, INF, TYPO_RULES as RULES, misspell  # noqa: F401  (one copy: the c5 benchmark workload uses the same rules)

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


