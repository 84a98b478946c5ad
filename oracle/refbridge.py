"""TEST INFRASTRUCTURE: ctypes access to oracle/_ref/libkiwi_ref.so (the real reference TUs,
see oracle/ref_bridge.cpp).  Only tests/, bench.py's cpu_baseline leg and the golden-vector
generator import this."""
from __future__ import annotations

import ctypes as C
import os
import struct
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# Two builds of the same reference sources (oracle/Makefile): `ref` (scalar architectures only) and `refx86` (every SIMD architecture +
# src/CoNgramModel.cpp).  ONE of them is loaded per process -- the x86 build where it exists, it contains the other: their C++ statics are
# STB_GNU_UNIQUE symbols, which the dynamic linker unifies process-wide, so the second library would silently run on the first one's dispatch tables.
LIB_PLAIN_PATH = os.path.join(HERE, "_ref", "libkiwi_ref.so")

MATCH_ALL = (1 << 0) | (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 5) | (1 << 23)
MATCH_ALL_WITH_NORMALIZING = MATCH_ALL | (1 << 16)


LIB_X86_PATH = os.path.join(HERE, "_ref", "libkiwi_ref_x86.so")
LIB_PATH = LIB_X86_PATH if os.path.exists(LIB_X86_PATH) else LIB_PLAIN_PATH


def x86_available() -> bool:
    return os.path.exists(LIB_X86_PATH)


def available() -> bool:
    return os.path.exists(LIB_PATH)


@dataclass
class Token:
    form: str
    tag: int
    position: int
    length: int
    word_position: int
    sent_position: int
    line_number: int
    sense_id: int
    score: float
    typo_cost: float
    typo_form_id: int
    paired_token: int
    sub_sent_position: int
    dialect: int
    morph_id: int


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.o = 0

    def get(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def str16(self):
        n = self.get("I")
        s = bytes(self.b[self.o:self.o + 2 * n]).decode("utf-16-le", errors="surrogatepass")
        self.o += 2 * n
        return s


def parse_results(buf) -> list:
    r = _Reader(buf)
    out = []
    for _ in range(r.get("I")):
        score = r.get("f")
        toks = []
        for _ in range(r.get("I")):
            form = r.str16()
            pos, wpos, spos, line = r.get("IIII")
            length, tag, sense = r.get("HBB")
            sc, tc, tfid, paired, subsent, dialect, mid = r.get("ffIIIHi")
            toks.append(Token(form, tag, pos, length, wpos, spos, line, sense, sc, tc, tfid, paired, subsent, dialect, mid))
        out.append((toks, score))
    return out


def write_model_dir(raw_model_path: str, out_dir: str):
    """The synthetic model written as the reference's own model FILES: sj.morph by the reference's serializer (serializer::writeMany with
    the "KIWI" key, src/KiwiBuilder.cpp:934-937), sj.knlm / skipbigram.mdl as the memory images they are."""
    L = C.CDLL(LIB_PATH)
    L.kref_write_model_dir.argtypes = [C.c_char_p, C.c_char_p]
    os.makedirs(out_dir, exist_ok=True)
    if L.kref_write_model_dir(raw_model_path.encode(), out_dir.encode()) != 0:
        raise RuntimeError("kref_write_model_dir failed")


class RefKiwi:
    def __init__(self, raw_model_path: str, arch: int = 0, model_dir_sbg=None, x86=False):
        """raw_model_path: a raw container; or, with model_dir_sbg = False / True / 2 (= cong.mdl, CoNgram), a DIRECTORY holding the reference's own model files,
        loaded through the reference's serializer (Knlm only / with skipbigram.mdl).
        model_dir_sbg = "cong_global": a raw container whose CoNgram blob is loaded as the GLOBAL model (window 7); ("built", ModelType, BuildOption bits): a
        directory as Kiwi ships it, loaded and built by the real KiwiBuilder (RefKiwi.built).
        x86: the library built with every SIMD architecture and src/CoNgramModel.cpp (oracle/Makefile refx86): a container with a CoNgram blob is
        analysed with it; arch 3 = sse4_1, 4 = avx2, 5 = avx512bw, 6 = avx512vnni (the quantised CoNgram path exists for those only)."""
        if x86 and not x86_available():
            raise RuntimeError("oracle/_ref/libkiwi_ref_x86.so is not built")
        self.lib = C.CDLL(LIB_PATH)
        L = self.lib
        L.kref_open_dir.restype = C.c_void_p
        L.kref_open_dir.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.kref_open.restype = C.c_void_p
        L.kref_open.argtypes = [C.c_char_p, C.c_int]
        L.kref_close.argtypes = [C.c_void_p]
        L.kref_set_config.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.kref_dump_dict.restype = C.c_size_t
        L.kref_dump_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.kref_sbg_next.restype = C.c_float
        L.kref_sbg_next.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32]
        L.kref_lm_progress.restype = C.c_float
        L.kref_lm_progress.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32]
        L.kref_split.restype = C.c_size_t
        L.kref_split.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t]
        L.kref_analyze.restype = C.c_size_t
        L.kref_analyze.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
        L.kref_analyze_batch.restype = C.c_double
        L.kref_analyze_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
        if model_dir_sbg == "cong_global":
            # a raw container whose CoNgram blob carries the window sections, scored as ModelType::congGlobal (x86 library only)
            L.kref_open_cong_global.restype = C.c_void_p
            L.kref_open_cong_global.argtypes = [C.c_char_p, C.c_int]
            self.h = L.kref_open_cong_global(raw_model_path.encode(), arch)
        elif isinstance(model_dir_sbg, tuple) and model_dir_sbg[0] == "dialects":
            # a raw container baked with KiwiBuilder's enabledDialects = model_dir_sbg[1] (kiwi_init's last argument)
            L.kref_open_dialects.restype = C.c_void_p
            L.kref_open_dialects.argtypes = [C.c_char_p, C.c_int, C.c_int]
            self.h = L.kref_open_dialects(raw_model_path.encode(), arch, int(model_dir_sbg[1]))
        elif isinstance(model_dir_sbg, tuple) and model_dir_sbg[0] == "built":
            # a directory as Kiwi ships it, loaded and baked by the REAL KiwiBuilder (x86 library only): ("built", ModelType, BuildOption bits)
            L.kref_open_built.restype = C.c_void_p
            L.kref_open_built.argtypes = [C.c_char_p, C.c_int, C.c_int]
            self.h = L.kref_open_built(raw_model_path.encode(), int(model_dir_sbg[1]), int(model_dir_sbg[2]))
        else:
            self.h = L.kref_open(raw_model_path.encode(), arch) if model_dir_sbg is None else L.kref_open_dir(raw_model_path.encode(), arch, int(model_dir_sbg))
        if not self.h:
            raise RuntimeError("kref_open failed")
        self._buf = np.zeros(1 << 20, np.uint8)

    @classmethod
    def built(cls, model_dir: str, model_type: int = 2, options: int = 1):
        """KiwiBuilder{ model_dir, 1, options, model_type }.build() of the reference itself (src/KiwiBuilder.cpp, unmodified): dictionaries per `options`
        (BuildOption bits), the combining rules of the directory's combiningRule.txt."""
        if not x86_available():
            raise RuntimeError("oracle/_ref/libkiwi_ref_x86.so is not built")
        return cls(model_dir, model_dir_sbg=("built", model_type, options))

    def close(self):
        if self.h:
            self.lib.kref_close(self.h)
            self.h = None

    def set_blocklist(self, items):
        """AnalyzeOption::blocklist of the analyses that follow: items = [(form, tag id or -1)], added like kiwi_morphset_add does
        (Kiwi::findMorphemes); returns the number of morphemes found per item.  [] clears it."""
        self.lib.kref_blocklist_clear.argtypes = [C.c_void_p]
        self.lib.kref_blocklist_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        self.lib.kref_blocklist_clear(self.h)
        out = []
        for form, tag in items:
            u = np.frombuffer(form.encode("utf-16-le"), np.uint16)
            out.append(self.lib.kref_blocklist_add(self.h, u.ctypes.data, len(u), tag))
        return out

    def _call(self, fn, *args):
        while True:
            need = fn(*args, self._buf.ctypes.data, self._buf.nbytes)
            if need <= self._buf.nbytes:
                return self._buf[:need]
            self._buf = np.zeros(int(need * 1.5), np.uint8)

    def set_config(self, cut_off=8.0, space_penalty=7.0, typo_cost_weight=6.0, max_unk=6, max_unk_j=0xFFFFFFFF, space_tol=0, integrate_allomorph=True):
        self.lib.kref_set_config(self.h, cut_off, space_penalty, typo_cost_weight, max_unk, max_unk_j, space_tol, int(integrate_allomorph))

    def analyze(self, text: str, top_n: int = 1, match: int = MATCH_ALL_WITH_NORMALIZING, open_ending=False):
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.kref_analyze(self.h, u.ctypes.data, len(u), top_n, match, int(open_ending), *a))
        return parse_results(buf)

    def analyze_pretokenized(self, text: str, spans, top_n: int = 1, match: int = MATCH_ALL_WITH_NORMALIZING):
        """Kiwi::analyze with pretokenized spans: spans = [(begin, end, [(form, begin, end, tag id, infer_regularity), ...]), ...], offsets in UTF-16 units
        of `text` (a token's relative to its span).  The reference's own answer (this repo's product refuses the argument so far)."""
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
        self.lib.kref_analyze_pretokenized.restype = C.c_size_t
        self.lib.kref_analyze_pretokenized.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]
        buf = self._call(lambda *a: self.lib.kref_analyze_pretokenized(self.h, u.ctypes.data, len(u), top_n, match, d.ctypes.data, len(spans), fl.ctypes.data, *a))
        return parse_results(buf)

    def split(self, text: str, match: int = MATCH_ALL_WITH_NORMALIZING):
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.kref_split(self.h, u.ctypes.data, len(u), match, *a))
        r = _Reader(buf)
        chunks = []
        for _ in range(r.get("I")):
            n, split_end = r.get("II")
            nodes = [r.get("IIIIiIIIf") for _ in range(n)]
            chunks.append((split_end, nodes))
        return chunks

    def split_typo(self, typo, text: str, threshold=2.5, allowed_dialect=0, match: int = MATCH_ALL_WITH_NORMALIZING):
        """Lattices over the typo graph of every chunk; `typo` is a prepared RefTypo."""
        self.lib.kref_split_typo.restype = C.c_size_t
        self.lib.kref_split_typo.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t]
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.kref_split_typo(self.h, typo.h, threshold, allowed_dialect, u.ctypes.data, len(u), match, *a))
        r = _Reader(buf)
        chunks = []
        for _ in range(r.get("I")):
            n, split_end = r.get("II")
            chunks.append((split_end, [r.get("IIIIiIIIf") for _ in range(n)]))
        return chunks

    def analyze_dialect(self, text: str, allowed_dialect: int, dialect_cost: float = 3.0, typo=None, threshold=2.5, top_n: int = 1, match: int = MATCH_ALL_WITH_NORMALIZING, open_ending=False):
        """AnalyzeOption::allowedDialects / dialectCost; typo None: the reference takes its built-in `dialect` typo set itself when a dialect is allowed."""
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        self.lib.kref_analyze_dialect.restype = C.c_size_t
        self.lib.kref_analyze_dialect.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
        buf = self._call(lambda *a: self.lib.kref_analyze_dialect(self.h, typo.h if typo is not None else None, threshold, allowed_dialect, dialect_cost, u.ctypes.data, len(u), top_n, match, int(open_ending), *a))
        return parse_results(buf)

    def analyze_typo(self, typo, text: str, threshold=2.5, allowed_dialect=0, top_n: int = 1, match: int = MATCH_ALL_WITH_NORMALIZING, open_ending=False):
        self.lib.kref_analyze_typo.restype = C.c_size_t
        self.lib.kref_analyze_typo.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = self._call(lambda *a: self.lib.kref_analyze_typo(self.h, typo.h, threshold, allowed_dialect, u.ctypes.data, len(u), top_n, match, int(open_ending), *a))
        return parse_results(buf)

    def lm_progress(self, node: int, wid: int):
        n = C.c_int32(node)
        ll = self.lib.kref_lm_progress(self.h, C.byref(n), wid)
        return float(ll), int(n.value)

    def sbg_next(self, node: int, pos: int, hist, wid: int):
        """One SbgState::next step through the reference (16-bit vocabularies) -> (ll, node, pos, hist[8]); ll is NaN for non-SBG models."""
        n = C.c_int32(node); p = C.c_uint32(pos); h = np.array(hist, np.uint32)
        ll = self.lib.kref_sbg_next(self.h, C.byref(n), C.byref(p), h.ctypes.data, wid)
        return float(ll), int(n.value), int(p.value), [int(x) for x in h]

    def dump_dict(self) -> bytes:
        return bytes(self._call(lambda *a: self.lib.kref_dump_dict(self.h, *a)))

    def analyze_batch(self, texts: list, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, threads=1, typo=None, typo_threshold=2.5):
        """typo: a prepared OracleTypo / RefTypo -> the batch is analysed with it."""
        enc = [np.frombuffer(t.encode("utf-16-le", errors="surrogatepass"), np.uint16) for t in texts]
        offs = np.zeros(len(enc) + 1, np.uint64)
        offs[1:] = np.cumsum([len(e) for e in enc])
        flat = np.concatenate(enc) if enc else np.zeros(0, np.uint16)
        ntok = C.c_uint64(0)
        self.lib.kref_analyze_batch_typo.restype = C.c_double
        self.lib.kref_analyze_batch_typo.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p]
        sec = self.lib.kref_analyze_batch_typo(self.h, typo.h if typo is not None else None, typo_threshold, flat.ctypes.data, offs.ctypes.data, len(enc), top_n, match, threads, C.byref(ntok))
        return float(sec), int(ntok.value)

    def analyze_batch_timed(self, texts: list, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, threads=1, min_seconds=2.0, typo=None, typo_threshold=2.5):
        """The batch timed soundly (timed_pool.hpp): persistent threads, one untimed warm-up pass, whole passes over `texts` until at least
        `min_seconds` of wall time -> (seconds, passes, tokens of one pass)."""
        enc = [np.frombuffer(t.encode("utf-16-le", errors="surrogatepass"), np.uint16) for t in texts]
        offs = np.zeros(len(enc) + 1, np.uint64)
        offs[1:] = np.cumsum([len(e) for e in enc])
        flat = np.concatenate(enc) if enc else np.zeros(0, np.uint16)
        ntok = C.c_uint64(0); passes = C.c_uint32(0)
        fn = self.lib.kref_analyze_batch_timed
        fn.restype = C.c_double
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        sec = fn(self.h, typo.h if typo is not None else None, typo_threshold, flat.ctypes.data, offs.ctypes.data, len(enc), top_n, match, threads, min_seconds, C.byref(passes), C.byref(ntok))
        return float(sec), int(passes.value), int(ntok.value)


# ---- typo graphs ---------------------------------------------------------------------------
COND = {"none": 0, "any": 1, "vowel": 2, "vocalic": 3, "vocalic_h": 4, "non_vowel": 5, "non_vocalic": 6, "non_vocalic_h": 7, "applosive": 8, "continual": 9, "boundary": 10}
DEFAULT_TYPO_SETS = {"without": 0, "basic": 1, "continual": 2, "basic_with_continual": 3, "lengthening": 4, "basic_with_continual_and_lengthening": 5, "dialect": 6}


def _u16(s):
    return np.frombuffer(s.encode("utf-16-le", errors="surrogatepass"), np.uint16)


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(LIB_PATH)
    return _LIB


def parse_typo_graph(buf):
    """Byte layout shared by kref_typo_graph and the oracle's korc_typo_graph -> (normalised text, nodes, maxContinualTypoIdx);
    node = (form, endPos, typoCost, prevOffset, siblingOffset, continualTypoIdx, dialect)."""
    import struct
    o = 0
    n, = struct.unpack_from("<I", buf, o); o += 4
    norm = bytes(buf[o:o + 2 * n]).decode("utf-16-le", errors="surrogatepass"); o += 2 * n
    cnt, = struct.unpack_from("<I", buf, o); o += 4
    nodes = []
    for _ in range(cnt):
        fl, = struct.unpack_from("<I", buf, o); o += 4
        form = bytes(buf[o:o + 2 * fl]).decode("utf-16-le", errors="surrogatepass"); o += 2 * fl
        end, cost, prev, sib, cti, dia = struct.unpack_from("<IfIIBH", buf, o); o += 19
        nodes.append((form, end, cost, prev, sib, cti, dia))
    mx, = struct.unpack_from("<I", buf, o)
    return norm, nodes, mx


def default_typo_entries(set_name):
    """Entries of a built-in typo set in the iteration order of the reference's map: [(orig, error, cost, cond, dialect)], continual, lengthening."""
    import struct
    L = _lib()
    L.kref_typo_default_entries.restype = C.c_size_t
    L.kref_typo_default_entries.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    need = L.kref_typo_default_entries(DEFAULT_TYPO_SETS[set_name], None, 0)
    buf = np.zeros(need, np.uint8)
    L.kref_typo_default_entries(DEFAULT_TYPO_SETS[set_name], buf.ctypes.data, need)
    b = buf.tobytes(); o = 0
    n, = struct.unpack_from("<I", b, o); o += 4
    out = []
    for _ in range(n):
        no, = struct.unpack_from("<I", b, o); o += 4
        orig = b[o:o + 2 * no].decode("utf-16-le", errors="surrogatepass"); o += 2 * no
        ne, = struct.unpack_from("<I", b, o); o += 4
        err = b[o:o + 2 * ne].decode("utf-16-le", errors="surrogatepass"); o += 2 * ne
        cost, cond, dia = struct.unpack_from("<fBH", b, o); o += 7
        out.append((orig, err, cost, cond, dia))
    cont, leng = struct.unpack_from("<ff", b, o)
    return out, cont, leng


class RefTypo:
    """The reference's TypoTransformer -> PreparedTypoTransformer -> generateGraph, public API only."""

    def __init__(self, continual=float("inf"), lengthening=float("inf")):
        L = self.lib = _lib()
        L.kref_typo_new.restype = C.c_void_p
        L.kref_typo_new.argtypes = [C.c_float, C.c_float]
        L.kref_typo_close.argtypes = [C.c_void_p]
        L.kref_typo_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_int]
        L.kref_typo_update_default.argtypes = [C.c_void_p, C.c_int]
        L.kref_typo_prepare.argtypes = [C.c_void_p, C.c_int]
        L.kref_typo_graph.restype = C.c_size_t
        L.kref_typo_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        self.h = L.kref_typo_new(continual, lengthening)

    def add(self, orig, error, cost=1.0, cond="none", dialect=0):
        o, e = _u16(orig), _u16(error)
        if self.lib.kref_typo_add(self.h, o.ctypes.data, len(o), e.ctypes.data, len(e), cost, COND[cond] if isinstance(cond, str) else cond, dialect) != 0:
            raise ValueError((orig, error))

    @classmethod
    def from_default(cls, set_name):
        """A copy of the built-in set itself (kiwi_typo_get_default)."""
        self = cls()
        self.lib.kref_typo_close(self.h)
        self.lib.kref_typo_from_default.restype = C.c_void_p
        self.lib.kref_typo_from_default.argtypes = [C.c_int]
        self.h = self.lib.kref_typo_from_default(DEFAULT_TYPO_SETS[set_name])
        return self

    def update_default(self, set_name):
        self.lib.kref_typo_update_default(self.h, DEFAULT_TYPO_SETS[set_name])

    def prepare(self, inverse=True):
        self.lib.kref_typo_prepare(self.h, int(inverse))

    def graph(self, text, allowed_dialect=0, norm_coda=True):
        u = _u16(text)
        need = self.lib.kref_typo_graph(self.h, u.ctypes.data, len(u), allowed_dialect, int(norm_coda), None, 0)
        buf = np.zeros(need, np.uint8)
        self.lib.kref_typo_graph(self.h, u.ctypes.data, len(u), allowed_dialect, int(norm_coda), buf.ctypes.data, need)
        return parse_typo_graph(buf.tobytes())

    def __del__(self):
        try:
            self.lib.kref_typo_close(self.h)
        except Exception:
            pass


def match_pattern(left: str, text: str, match: int):
    """The reference's matchPattern (src/PatternMatcher.cpp:380) at text[0], `left` = the UTF-16 unit before it: (matched length, tag)."""
    import numpy as np
    lib = _lib()
    lib.kref_match_pattern.restype = C.c_uint64
    lib.kref_match_pattern.argtypes = [C.c_uint16, C.c_void_p, C.c_uint32, C.c_uint64]
    u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
    r = int(lib.kref_match_pattern(ord(left), u.ctypes.data, len(u), match))
    return r & 0xFFFFFFFF, r >> 32
