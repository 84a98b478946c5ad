"""TEST INFRASTRUCTURE: ctypes access to oracle/_build/liboracle.so (this repo's CPU restatement of the
hot path, oracle/*.hpp).  Same call surface and byte layouts as refbridge.RefKiwi so tests can diff them."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from refbridge import MATCH_ALL, MATCH_ALL_WITH_NORMALIZING, Token, _Reader, parse_results  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")

COUNTER_NAMES = ["inputUnits", "trieProbes", "trieProbeKeyBytes", "failHops", "candEmits", "otherNodes",
                 "transitions", "candMorphs", "statesWritten", "lmProbes", "lmProbeKeyBytes", "lmRootProbes", "tokens",
                 "maxPrevPaths", "nodesOver128", "nodesOver512", "lattNodes", "sbgEvals", "sbgProbeKeyBytes", "sbgHits", "sbgModel",
                 "congCtxRows", "congOutRows", "congScores", "congProbes", "congProbeKeyBytes", "congRootProbes", "congDim", "congGlobalScores", "typoGraphNodes", "typoStateSteps", "typoStatesKept", "congPast64"]


def available() -> bool:
    return os.path.exists(LIB_PATH)


class OracleKiwi:
    def __init__(self, raw_model_path: str, enabled_dialects: int = 0):
        self.lib = C.CDLL(LIB_PATH)
        L = self.lib
        L.korc_open.restype = C.c_void_p
        L.korc_open.argtypes = [C.c_char_p]
        L.korc_close.argtypes = [C.c_void_p]
        L.korc_set_config.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.korc_dump_dict.restype = C.c_size_t
        L.korc_set_faithful_order.argtypes = [C.c_void_p, C.c_int]
        L.korc_set_container_limits.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.korc_lm_next.restype = C.c_float
        L.korc_lm_next.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32]
        L.korc_dump_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.korc_lm_progress.restype = C.c_float
        L.korc_lm_progress.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32]
        L.korc_split.restype = C.c_size_t
        L.korc_split.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t]
        L.korc_analyze.restype = C.c_size_t
        L.korc_analyze.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
        L.korc_analyze_batch.restype = C.c_double
        L.korc_analyze_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
        L.korc_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.korc_set_cong_global.argtypes = [C.c_void_p, C.c_int]
        if enabled_dialects:      # KiwiBuilder's enabledDialects at the bake
            L.korc_open_dialects.restype = C.c_void_p
            L.korc_open_dialects.argtypes = [C.c_char_p, C.c_int]
            self.h = L.korc_open_dialects(raw_model_path.encode(), enabled_dialects)
        else:
            self.h = L.korc_open(raw_model_path.encode())
        if not self.h:
            raise RuntimeError("korc_open failed")
        self._buf = np.zeros(1 << 20, np.uint8)

    def close(self):
        if self.h:
            self.lib.korc_close(self.h)
            self.h = None

    def _call(self, fn, *args):
        while True:
            need = fn(*args, self._buf.ctypes.data, self._buf.nbytes)
            if need <= self._buf.nbytes:
                return self._buf[:need]
            self._buf = np.zeros(int(need * 1.5), np.uint8)

    def set_config(self, cut_off=8.0, space_penalty=7.0, typo_cost_weight=6.0, max_unk=6, max_unk_j=0xFFFFFFFF, space_tol=0, integrate_allomorph=True):
        self.lib.korc_set_config(self.h, cut_off, space_penalty, typo_cost_weight, max_unk, max_unk_j, space_tol, int(integrate_allomorph))

    def set_cong_global(self, on=True):
        """Score with the sections of the global CoNgram model (ModelType::congGlobal) when the model file carries them (window 7)."""
        if self.lib.korc_set_cong_global(self.h, int(on)) != 0:
            raise RuntimeError("the model has no CoNgram window sections")

    def set_faithful_order(self, on=True):
        """Test hook: hand kept paths on in the reference's own container order (persistent std::unordered_set / _map, as its
        thread_local containers) instead of insertion order; resets the persistent state."""
        self.lib.korc_set_faithful_order(self.h, int(on))

    def set_container_limits(self, small_max=128, medium_max=512, bucket_cap=128):
        """Test hook: number of incoming paths up to which the small / medium container is used, and the per-bucket key cap."""
        self.lib.korc_set_container_limits(self.h, small_max, medium_max, bucket_cap)

    def analyze(self, text: str, top_n: int = 1, match: int = MATCH_ALL_WITH_NORMALIZING, open_ending=False):
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.korc_analyze(self.h, u.ctypes.data, len(u), top_n, match, int(open_ending), *a))
        if len(buf) == 0:
            raise RuntimeError("korc_analyze failed")
        return parse_results(buf)

    def analyze_pretokenized(self, text: str, spans, top_n: int = 1, match: int = MATCH_ALL_WITH_NORMALIZING):
        """Kiwi::analyze with pretokenized spans as far as the oracle restates them (spans as in refbridge.RefKiwi.analyze_pretokenized); None when a span needs
        a temporary form or morpheme (not restated)."""
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        desc, forms = [], []
        for b, e, toks in spans:
            desc += [b, e, len(toks)]
            for form, tb, te, tag, infer in toks:
                f = np.frombuffer(form.encode("utf-16-le"), np.uint16)
                desc += [sum(len(x) for x in forms), len(f), tb, te, tag, infer]
                forms.append(f)
        d = np.array(desc if desc else [0], np.uint32)
        fl = np.concatenate(forms) if forms else np.zeros(1, np.uint16)
        self.lib.korc_analyze_pretokenized.restype = C.c_size_t
        self.lib.korc_analyze_pretokenized.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]
        buf = self._call(lambda *a: self.lib.korc_analyze_pretokenized(self.h, u.ctypes.data, len(u), top_n, match, d.ctypes.data, len(spans), fl.ctypes.data, *a))
        return parse_results(buf) if len(buf) else None

    def split(self, text: str, match: int = MATCH_ALL_WITH_NORMALIZING):
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.korc_split(self.h, u.ctypes.data, len(u), match, *a))
        r = _Reader(buf)
        chunks = []
        for _ in range(r.get("I")):
            n, split_end = r.get("II")
            nodes = [r.get("IIIIiIIIf") for _ in range(n)]
            chunks.append((split_end, nodes))
        return chunks

    def split_typo(self, typo, text: str, threshold=2.5, allowed_dialect=0, match: int = MATCH_ALL_WITH_NORMALIZING):
        """Lattices over the typo graph of every chunk; `typo` is a prepared OracleTypo (typo_lattice_oracle.hpp)."""
        self.lib.korc_split_typo.restype = C.c_size_t
        self.lib.korc_split_typo.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t]
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.korc_split_typo(self.h, typo.h, threshold, allowed_dialect, u.ctypes.data, len(u), match, *a))
        r = _Reader(buf)
        chunks = []
        for _ in range(r.get("I")):
            n, split_end = r.get("II")
            chunks.append((split_end, [r.get("IIIIiIIIf") for _ in range(n)]))
        return chunks

    def analyze_dialect(self, text: str, allowed_dialect: int, dialect_cost: float = 3.0, typo=None, threshold=2.5, top_n: int = 1, match: int = MATCH_ALL_WITH_NORMALIZING, open_ending=False):
        """AnalyzeOption::allowedDialects / dialectCost.  typo None with a dialect allowed: the caller's copy of the built-in `dialect` set (dialect_typo(), below)
        at threshold 2.5 -- what the reference takes by itself (src/Kiwi.cpp:1037-1041)."""
        if typo is None and allowed_dialect:
            typo, threshold = self.dialect_typo(), 2.5
        self.lib.korc_analyze_dialect.restype = C.c_size_t
        self.lib.korc_analyze_dialect.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.korc_analyze_dialect(self.h, typo.h if typo is not None else None, threshold, allowed_dialect, dialect_cost, u.ctypes.data, len(u), top_n, match, int(open_ending), *a))
        if len(buf) == 0:
            raise RuntimeError("korc_analyze_dialect failed")
        return parse_results(buf)

    def dialect_typo(self):
        """The built-in typo set DefaultTypoSet::dialect as an OracleTypo, prepared: its entries come from the committed fixture tests/golden/typo_default_sets.json
        where present, else from the reference itself (refbridge)."""
        if getattr(self, "_dialect_typo", None) is None:
            import refbridge
            entries, continual, lengthening = refbridge.default_typo_entries("dialect")
            t = OracleTypo()
            t.update_entries(entries, continual, lengthening)
            t.prepare(True)
            self._dialect_typo = t
        return self._dialect_typo

    def analyze_typo(self, typo, text: str, threshold=2.5, allowed_dialect=0, top_n: int = 1, match: int = MATCH_ALL_WITH_NORMALIZING, open_ending=False):
        self.lib.korc_analyze_typo.restype = C.c_size_t
        self.lib.korc_analyze_typo.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.korc_analyze_typo(self.h, typo.h, threshold, allowed_dialect, u.ctypes.data, len(u), top_n, match, int(open_ending), *a))
        if len(buf) == 0:
            raise RuntimeError("korc_analyze_typo failed")
        return parse_results(buf)

    def lm_progress(self, node: int, wid: int):
        n = C.c_int32(node)
        ll = self.lib.korc_lm_progress(self.h, C.byref(n), wid)
        return float(ll), int(n.value)

    def lm_next(self, node: int, pos: int, hist, wid: int):
        """One step of the model's LM state type (Knlm, or SkipBigram on top of it) -> (ll, node, pos, hist[8])."""
        n = C.c_int32(node); p = C.c_uint32(pos); h = np.array(hist, np.uint32)
        ll = self.lib.korc_lm_next(self.h, C.byref(n), C.byref(p), h.ctypes.data, wid)
        return float(ll), int(n.value), int(p.value), [int(x) for x in h]

    def dump_dict(self) -> bytes:
        return bytes(self._call(lambda *a: self.lib.korc_dump_dict(self.h, *a)))

    def analyze_batch(self, texts: list, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, threads=1, typo=None, typo_threshold=2.5):
        """typo: a prepared OracleTypo / RefTypo -> the batch is analysed with it."""
        enc = [np.frombuffer(t.encode("utf-16-le", errors="surrogatepass"), np.uint16) for t in texts]
        offs = np.zeros(len(enc) + 1, np.uint64)
        offs[1:] = np.cumsum([len(e) for e in enc])
        flat = np.concatenate(enc) if enc else np.zeros(0, np.uint16)
        ntok = C.c_uint64(0)
        self.lib.korc_analyze_batch_typo.restype = C.c_double
        self.lib.korc_analyze_batch_typo.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p]
        sec = self.lib.korc_analyze_batch_typo(self.h, typo.h if typo is not None else None, typo_threshold, flat.ctypes.data, offs.ctypes.data, len(enc), top_n, match, threads, C.byref(ntok))
        return float(sec), int(ntok.value)

    def analyze_batch_timed(self, texts: list, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, threads=1, min_seconds=2.0, typo=None, typo_threshold=2.5):
        """The batch timed soundly (timed_pool.hpp): persistent threads, one untimed warm-up pass, whole passes over `texts` until at least
        `min_seconds` of wall time -> (seconds, passes, tokens of one pass)."""
        enc = [np.frombuffer(t.encode("utf-16-le", errors="surrogatepass"), np.uint16) for t in texts]
        offs = np.zeros(len(enc) + 1, np.uint64)
        offs[1:] = np.cumsum([len(e) for e in enc])
        flat = np.concatenate(enc) if enc else np.zeros(0, np.uint16)
        ntok = C.c_uint64(0); passes = C.c_uint32(0)
        fn = self.lib.korc_analyze_batch_timed
        fn.restype = C.c_double
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        sec = fn(self.h, typo.h if typo is not None else None, typo_threshold, flat.ctypes.data, offs.ctypes.data, len(enc), top_n, match, threads, min_seconds, C.byref(passes), C.byref(ntok))
        return float(sec), int(passes.value), int(ntok.value)

    def set_blocklist(self, items):
        """AnalyzeOption::blocklist of the analyses that follow: items = [(form, tag id or -1)]; returns the morphemes found per item."""
        self.lib.korc_blocklist_clear.argtypes = [C.c_void_p]
        self.lib.korc_blocklist_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        self.lib.korc_blocklist_clear(self.h)
        out = []
        for form, tag in items:
            u = np.frombuffer(form.encode("utf-16-le"), np.uint16)
            out.append(self.lib.korc_blocklist_add(self.h, u.ctypes.data, len(u), tag))
        return out

    def counters(self, reset=False) -> dict:
        arr = np.zeros(len(COUNTER_NAMES), np.uint64)
        self.lib.korc_counters(self.h, arr.ctypes.data, int(reset))
        return dict(zip(COUNTER_NAMES, (int(x) for x in arr)))


def cpu_capacity(threads: int, min_seconds: float = 1.0) -> float:
    """Cores this process really gets (korc_cpu_capacity): aggregate rate of `threads` threads of register-only work relative to one thread."""
    lib = C.CDLL(LIB_PATH)
    lib.korc_cpu_capacity.restype = C.c_double
    lib.korc_cpu_capacity.argtypes = [C.c_int, C.c_double]
    return float(lib.korc_cpu_capacity(threads, min_seconds))


def alg_bytes(c: dict) -> dict:
    """ALG_BYTES v1: algorithmic bytes from oracle event counts, no cache credit.
    Returns the split used by bench.py: dictionary scan + lattice build ('lattice') and best-path search ('search')."""
    lattice = (2 * c["inputUnits"] + c["trieProbes"] * (12 + 4) + c["trieProbeKeyBytes"] + c["failHops"] * 8
               + c["candEmits"] * (16 + 24) + c["otherNodes"] * 24
               # typo lattices: 28 B per typo-graph node, 44 B (the fixed part of a search state) read per state transition and written per
               # state that lives on; their trie probes, fail hops and node appends are in the counters above
               + c.get("typoGraphNodes", 0) * 28 + c.get("typoStateSteps", 0) * 44 + c.get("typoStatesKept", 0) * 44)
    # S: the SkipBigram state carries the 8-word history ring; so does the state of the global CoNgram model (7 words + a spare)
    state = 48 if (c.get("sbgModel") or c.get("congGlobalScores")) else 32
    search = (c["transitions"] * state + c["candMorphs"] * 16 + c["statesWritten"] * state
              + c["lmProbes"] * (20 + 4) + c["lmProbeKeyBytes"] + c["lmRootProbes"] * 4 + c["tokens"] * 24
              # "SBG extra": per evaluate() 8 B row pointers + 8 discounts + the key bytes of 8 partner searches + 4 B per hit
              + c.get("sbgEvals", 0) * (8 + 8 * 4) + c.get("sbgProbeKeyBytes", 0) + c.get("sbgHits", 0) * 4)
    dim = c.get("congDim", 0)
    if dim:     # "CoNgram": unique context rows dim + 16 B, unique output rows dim + 8 B, 4 B per score, context-trie probe = Knlm probe with a 16-byte node
        search += (c["congCtxRows"] * (dim + 16) + c["congOutRows"] * (dim + 8) + c["congScores"] * 4
                   + c["congProbes"] * (16 + 4) + c["congProbeKeyBytes"] + c["congRootProbes"] * 4)
        # "CoNgram global extra" (DESIGN.md, round 5): a mixture reads the distant rows of the seven history words (dim + 8 B each, as an output row) and the
        # eight position confidences, on top of the context / output rows counted above
        search += c.get("congGlobalScores", 0) * (7 * (dim + 8) + 8 * 4)
    return {"lattice": lattice, "search": search, "total": lattice + search}


class OracleTypo:
    """typo_oracle.hpp: rule container -> prepare -> typo graph (same interface and byte layout as refbridge.RefTypo)."""

    def __init__(self, continual=float("inf"), lengthening=float("inf")):
        L = self.lib = C.CDLL(LIB_PATH)
        L.korc_typo_new.restype = C.c_void_p
        L.korc_typo_new.argtypes = [C.c_float, C.c_float]
        L.korc_typo_close.argtypes = [C.c_void_p]
        L.korc_typo_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_int]
        L.korc_typo_add_entry.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_int]
        L.korc_typo_set_costs.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.korc_typo_prepare.argtypes = [C.c_void_p, C.c_int]
        L.korc_typo_graph.restype = C.c_size_t
        L.korc_typo_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        self.h = L.korc_typo_new(continual, lengthening)

    @staticmethod
    def _u16(s):
        return np.frombuffer(s.encode("utf-16-le", errors="surrogatepass"), np.uint16)

    def add(self, orig, error, cost=1.0, cond=0, dialect=0):
        o, e = self._u16(orig), self._u16(error)
        if self.lib.korc_typo_add(self.h, o.ctypes.data, len(o), e.ctypes.data, len(e), cost, cond, dialect) != 0:
            raise ValueError((orig, error))

    def update_entries(self, entries, continual, lengthening):
        """TypoTransformer::update with another transformer's entries, given in the iteration order of its map."""
        for orig, err, cost, cond, dialect in entries:
            o, e = self._u16(orig), self._u16(err)
            self.lib.korc_typo_add_entry(self.h, o.ctypes.data, len(o), e.ctypes.data, len(e), cost, cond, dialect)
        self.lib.korc_typo_set_costs(self.h, continual, lengthening)

    def prepare(self, inverse=True):
        self.lib.korc_typo_prepare(self.h, int(inverse))

    def graph_bytes(self, text, allowed_dialect=0, norm_coda=True):
        u = self._u16(text)
        need = self.lib.korc_typo_graph(self.h, u.ctypes.data, len(u), allowed_dialect, int(norm_coda), None, 0)
        buf = np.zeros(need, np.uint8)
        self.lib.korc_typo_graph(self.h, u.ctypes.data, len(u), allowed_dialect, int(norm_coda), buf.ctypes.data, need)
        return buf.tobytes()

    def __del__(self):
        try:
            self.lib.korc_typo_close(self.h)
        except Exception:
            pass
