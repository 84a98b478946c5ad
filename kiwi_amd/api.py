"""ctypes access to the product library kiwi_amd/libkiwi_hip.so through its C ABI (include/kiwi_amd.h).

There is deliberately no fallback: if the shared library (built by ``__graft_entry__.build()`` /
``make -C kiwi_amd/csrc``) is missing or no HIP device is visible, opening an engine raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import struct
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KAMD_LIB", os.path.join(HERE, "libkiwi_hip.so"))

MATCH_ALL = (1 << 0) | (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 5) | (1 << 23)
MATCH_ALL_WITH_NORMALIZING = MATCH_ALL | (1 << 16)


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


TOKEN_DTYPE = np.dtype([
    ("position", "<u4"), ("word_position", "<u4"), ("sent_position", "<u4"), ("line_number", "<u4"),
    ("length", "<u2"), ("tag", "u1"), ("sense_or_script", "u1"), ("score", "<f4"), ("typo_cost", "<f4"),
    ("typo_form_id", "<u4"), ("paired_token", "<u4"), ("sub_sent_position", "<u4"), ("dialect", "<u2"),
    ("form_len", "<u2"), ("morph_id", "<i4"), ("form_off", "<u8")])
assert TOKEN_DTYPE.itemsize == 56


def load_library(lib_path=None):
    lib_path = lib_path or LIB_PATH
    if not os.path.exists(lib_path):
        raise RuntimeError(f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the analyze path has no CPU fallback)")
    L = C.CDLL(lib_path)
    L.kamd_open.restype = C.c_void_p
    L.kamd_open.argtypes = [C.c_char_p, C.c_int]
    L.kamd_close.argtypes = [C.c_void_p]
    L.kamd_last_error.restype = C.c_char_p
    L.kamd_set_config.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.kamd_analyze_batch.restype = C.c_void_p
    L.kamd_analyze_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
    L.kamd_stage.restype = C.c_void_p
    L.kamd_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
    L.kamd_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.kamd_fetch.restype = C.c_void_p
    L.kamd_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.kamd_batch_info.argtypes = [C.c_void_p, C.c_void_p]
    L.kamd_batch_reruns.argtypes = [C.c_void_p, C.c_void_p]
    L.kamd_batch_reruns.restype = C.c_int
    L.kamd_batch_close.argtypes = [C.c_void_p]
    L.kamd_res_texts.restype = C.c_uint32
    L.kamd_res_texts.argtypes = [C.c_void_p]
    L.kamd_res_size.restype = C.c_uint32
    L.kamd_res_size.argtypes = [C.c_void_p, C.c_uint32]
    L.kamd_res_prob.restype = C.c_float
    L.kamd_res_prob.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.kamd_res_token_num.restype = C.c_uint32
    L.kamd_res_token_num.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.kamd_res_tokens.restype = C.c_void_p
    L.kamd_res_tokens.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.kamd_res_forms.restype = C.c_void_p
    L.kamd_res_forms.argtypes = [C.c_void_p, C.c_uint32]
    L.kamd_res_d2h_bytes.restype = C.c_uint64
    L.kamd_res_d2h_bytes.argtypes = [C.c_void_p]
    L.kamd_res_close.argtypes = [C.c_void_p]
    L.kamd_res_pack.restype = C.c_size_t
    L.kamd_res_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.kamd_res_merge_strided.restype = C.c_void_p
    L.kamd_res_merge_strided.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.kamd_dump_dict.restype = C.c_size_t
    L.kamd_dump_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.kamd_dump_lattices.restype = C.c_size_t
    L.kamd_dump_lattices.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t]
    return L


def declared_symbols(header="kiwi_amd.h"):
    """Function names declared in include/<header> (used by the CPU-side ABI test)."""
    hdr = open(os.path.join(os.path.dirname(HERE), "include", header), encoding="utf-8").read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b((?:kamd|kiwi)_[a-z0-9_]+)\s*\(", hdr)))


def pack_texts(texts):
    enc = [np.frombuffer(t.encode("utf-16-le", errors="surrogatepass"), np.uint16) for t in texts]
    offs = np.zeros(len(enc) + 1, np.uint64)
    if enc:
        offs[1:] = np.cumsum([len(e) for e in enc])
    flat = np.ascontiguousarray(np.concatenate(enc)) if enc and offs[-1] else np.zeros(1, np.uint16)
    return flat, offs


class Results:
    def __init__(self, lib, handle):
        self.lib, self.h = lib, handle

    def close(self):
        if self.h:
            self.lib.kamd_res_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def n_texts(self):
        return self.lib.kamd_res_texts(self.h)

    def d2h_bytes(self):
        return int(self.lib.kamd_res_d2h_bytes(self.h))

    def pack(self) -> np.ndarray:
        """The results as one position-independent byte buffer (kamd_res_pack): what a rank ships to the gathering rank."""
        n = self.lib.kamd_res_pack(self.h, None, 0)
        buf = np.zeros(n, np.uint8)
        if self.lib.kamd_res_pack(self.h, buf.ctypes.data, n) != n:
            raise RuntimeError("kamd_res_pack failed: " + self.lib.kamd_last_error().decode())
        return buf

    @classmethod
    def merge_strided(cls, lib, parts):
        """Packed results of an index-strided split (text g -> part g % len(parts)) merged back into input order."""
        parts = [np.ascontiguousarray(p, np.uint8) for p in parts]
        ptrs = (C.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
        sizes = (C.c_size_t * len(parts))(*[p.nbytes for p in parts])
        h = lib.kamd_res_merge_strided(ptrs, sizes, len(parts))
        if not h:
            raise RuntimeError("kamd_res_merge_strided failed: " + lib.kamd_last_error().decode())
        return cls(lib, h)

    def token_array(self, text, index=0):
        n = self.lib.kamd_res_token_num(self.h, text, index)
        if not n:
            return np.zeros(0, TOKEN_DTYPE)
        p = self.lib.kamd_res_tokens(self.h, text, index)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n * TOKEN_DTYPE.itemsize,)).view(TOKEN_DTYPE).copy()

    def to_python(self):
        """[(tokens, score)] per analysis, per text -- same shape as refbridge.parse_results."""
        out = []
        for t in range(self.n_texts()):
            forms_p = self.lib.kamd_res_forms(self.h, t)
            res = []
            for i in range(self.lib.kamd_res_size(self.h, t)):
                arr = self.token_array(t, i)
                toks = []
                for r in arr:
                    fl, fo = int(r["form_len"]), int(r["form_off"])
                    raw = C.string_at(forms_p + 2 * fo, 2 * fl) if fl else b""
                    toks.append(Token(raw.decode("utf-16-le", errors="surrogatepass"), int(r["tag"]), int(r["position"]), int(r["length"]),
                                      int(r["word_position"]), int(r["sent_position"]), int(r["line_number"]), int(r["sense_or_script"]),
                                      float(r["score"]), float(r["typo_cost"]), int(r["typo_form_id"]), int(r["paired_token"]),
                                      int(r["sub_sent_position"]), int(r["dialect"]), int(r["morph_id"])))
                res.append((toks, float(self.lib.kamd_res_prob(self.h, t, i))))
            out.append(res)
        return out


class MorphSet:
    def __init__(self, lib, handle):
        self.lib, self.h = lib, handle

    def close(self):
        if self.h:
            self.lib.kamd_morphset_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Batch:
    def __init__(self, lib, handle):
        self.lib, self.h = lib, handle

    def info(self):
        a = np.zeros(3, np.uint64)
        self.lib.kamd_batch_info(self.h, a.ctypes.data)
        return {"chunks": int(a[0]), "units": int(a[1]), "device_bytes": int(a[2])}

    def pool(self):
        """kamd_batch_pool: states in the chunks' own arenas, in the pool behind them, and what the last fetched run asked the pool for."""
        a = np.zeros(3, np.uint64)
        self.lib.kamd_batch_pool.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.kamd_batch_pool(self.h, a.ctypes.data)
        return {"arena_states": int(a[0]), "pool_states": int(a[1]), "pool_asked": int(a[2])}

    def close(self):
        if self.h:
            self.lib.kamd_batch_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


class KiwiAmd:
    """Batched analyzer on one MI355X."""

    def __init__(self, raw_model_path: str, device: int = -1, lib_path: str = None, enabled_dialects: int = 0, lm_mode: int = 0):
        self.lib = load_library(lib_path)     # lib_path: another build of the same library (tests: the small-capacity build)
        if lm_mode:               # kamd_open_mode: 1 Knlm, 2 SkipBigram, 3 CoNgram local, 4 CoNgram global (distant tokens)
            self.lib.kamd_open_mode.restype = C.c_void_p
            self.lib.kamd_open_mode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
            self.h = self.lib.kamd_open_mode(raw_model_path.encode(), device, lm_mode, enabled_dialects)
        elif enabled_dialects:    # KiwiBuilder's enabledDialects (kiwi_init's last argument)
            self.lib.kamd_open_dialects.restype = C.c_void_p
            self.lib.kamd_open_dialects.argtypes = [C.c_char_p, C.c_int, C.c_int]
            self.h = self.lib.kamd_open_dialects(raw_model_path.encode(), device, enabled_dialects)
        else:
            self.h = self.lib.kamd_open(raw_model_path.encode(), device)
        if not self.h:
            raise RuntimeError("kamd_open failed: " + self.lib.kamd_last_error().decode())

    def close(self):
        if self.h:
            self.lib.kamd_close(self.h)
            self.h = None

    def _err(self, what):
        return RuntimeError(f"{what} failed: " + self.lib.kamd_last_error().decode())

    def set_config(self, cut_off=8.0, space_penalty=7.0, typo_cost_weight=6.0, max_unk=6, max_unk_j=0xFFFFFFFF, space_tol=0, integrate_allomorph=True):
        self.lib.kamd_set_config(self.h, cut_off, space_penalty, typo_cost_weight, max_unk, max_unk_j, space_tol, int(integrate_allomorph))

    def set_oov_chr_bias(self, bias: float):
        self.lib.kamd_set_oov_chr_bias.argtypes = [C.c_void_p, C.c_float]
        self.lib.kamd_set_oov_chr_bias(self.h, bias)

    def set_oov_freq_params(self, global_weight: float = 35.0, local_weight: float = 3.0, global_min_freq: float = 4.0):
        """KiwiConfig::oovGlobalWeight / oovLocalWeight / oovGlobalMinFreq (Match::oovChrFreqModel, 2 << 8 in `match`)."""
        self.lib.kamd_set_oov_freq_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        self.lib.kamd_set_oov_freq_params(self.h, global_weight, local_weight, global_min_freq)

    def analyze_batch(self, texts, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, open_ending=False, host_threads=0) -> Results:
        flat, offs = pack_texts(texts)
        r = self.lib.kamd_analyze_batch(self.h, flat.ctypes.data, offs.ctypes.data, len(texts), top_n, match, int(open_ending), host_threads)
        if not r:
            raise self._err("kamd_analyze_batch")
        return Results(self.lib, r)

    def morphset(self, items):
        """A morpheme set for `blocklist=`: items = [(form, tag id or -1)], each added like kiwi_morphset_add (Kiwi::findMorphemes).
        Returns (handle object, morphemes found per item); keep the object alive while it is in use."""
        L = self.lib
        L.kamd_morphset_new.restype = C.c_void_p
        L.kamd_morphset_new.argtypes = [C.c_void_p]
        L.kamd_morphset_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        L.kamd_morphset_close.argtypes = [C.c_void_p]
        ms = MorphSet(L, L.kamd_morphset_new(self.h))
        if not ms.h:
            raise self._err("kamd_morphset_new")
        found = []
        for form, tag in items:
            u = np.frombuffer(form.encode("utf-16-le"), np.uint16)
            n = L.kamd_morphset_add(ms.h, u.ctypes.data, len(u), tag)
            if n < 0:
                raise self._err("kamd_morphset_add")
            found.append(n)
        return ms, found

    def analyze_batch_opt(self, texts, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, open_ending=False, host_threads=0, typo=None, typo_threshold=2.5, blocklist=None) -> Results:
        """kamd_analyze_batch_opt: the per-call options of the reference's AnalyzeOption (prepared typo transformer, blocklist)."""
        L = self.lib
        L.kamd_analyze_batch_opt.restype = C.c_void_p
        L.kamd_analyze_batch_opt.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
        flat, offs = pack_texts(texts)
        r = L.kamd_analyze_batch_opt(self.h, typo.h if typo is not None else None, typo_threshold, 0, blocklist.h if blocklist is not None else None,
                                     flat.ctypes.data, offs.ctypes.data, len(texts), top_n, match, int(open_ending), host_threads)
        if not r:
            raise self._err("kamd_analyze_batch_opt")
        return Results(self.lib, r)

    def analyze_batch_dialect(self, texts, allowed_dialect, dialect_cost=3.0, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, open_ending=False, host_threads=0, typo=None, typo_threshold=2.5, blocklist=None) -> Results:
        """kamd_analyze_batch_dialect: AnalyzeOption::allowedDialects / dialectCost; typo None with a dialect allowed: the built-in `dialect` typo set, threshold 2.5."""
        L = self.lib
        L.kamd_analyze_batch_dialect.restype = C.c_void_p
        L.kamd_analyze_batch_dialect.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
        flat, offs = pack_texts(texts)
        r = L.kamd_analyze_batch_dialect(self.h, typo.h if typo is not None else None, typo_threshold, allowed_dialect, dialect_cost, blocklist.h if blocklist is not None else None,
                                         flat.ctypes.data, offs.ctypes.data, len(texts), top_n, match, int(open_ending), host_threads)
        if not r:
            raise self._err("kamd_analyze_batch_dialect")
        return Results(self.lib, r)

    def analyze_packed(self, flat, offs, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, open_ending=False, host_threads=0, typo=None, typo_threshold=2.5) -> Results:
        """The C-ABI call itself on an already packed batch (`pack_texts`): UTF-16 strings resident on the host in, packed token
        records resident on the host out -- the end-to-end region of SURVEY.md section 8(d).  typo: a prepared `Typo` (kamd_analyze_batch_opt)."""
        if typo is None:
            r = self.lib.kamd_analyze_batch(self.h, flat.ctypes.data, offs.ctypes.data, len(offs) - 1, top_n, match, int(open_ending), host_threads)
        else:
            L = self.lib
            L.kamd_analyze_batch_opt.restype = C.c_void_p
            L.kamd_analyze_batch_opt.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
            r = L.kamd_analyze_batch_opt(self.h, typo.h, typo_threshold, 0, None, flat.ctypes.data, offs.ctypes.data, len(offs) - 1, top_n, match, int(open_ending), host_threads)
        if not r:
            raise self._err("kamd_analyze_batch")
        return Results(self.lib, r)

    def analyze(self, text, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, open_ending=False):
        return self.analyze_batch([text], top_n, match, open_ending, 1).to_python()[0]

    def analyze_pretokenized(self, text, spans, top_n=1, match=MATCH_ALL_WITH_NORMALIZING, open_ending=False):
        """kamd_analyze_pretokenized: spans = [(begin, end, [(form, begin, end, tag id, infer_regularity), ...]), ...] in UTF-16 units of `text`
        (token offsets relative to their span) -- Kiwi::analyze's `pretokenized` argument."""
        L = self.lib
        L.kamd_analyze_pretokenized.restype = C.c_void_p
        L.kamd_analyze_pretokenized.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
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
        r = L.kamd_analyze_pretokenized(self.h, u.ctypes.data, len(u), top_n, match, int(open_ending), d.ctypes.data, len(spans), fl.ctypes.data)
        if not r:
            raise self._err("kamd_analyze_pretokenized")
        return Results(self.lib, r).to_python()[0]

    def stage(self, texts, match=MATCH_ALL_WITH_NORMALIZING, open_ending=False, host_threads=0, typo=None, typo_threshold=2.5) -> Batch:
        """typo: a prepared `Typo` (kamd_stage_typo); it must outlive the batch."""
        flat, offs = pack_texts(texts)
        if typo is None:
            b = self.lib.kamd_stage(self.h, flat.ctypes.data, offs.ctypes.data, len(texts), match, int(open_ending), host_threads)
        else:
            self.lib.kamd_stage_typo.restype = C.c_void_p
            self.lib.kamd_stage_typo.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
            b = self.lib.kamd_stage_typo(self.h, typo.h, typo_threshold, 0, flat.ctypes.data, offs.ctypes.data, len(texts), match, int(open_ending), host_threads)
        if not b:
            raise self._err("kamd_stage")
        return Batch(self.lib, b)

    def run(self, batch: Batch):
        ms = np.zeros(4, np.float32)
        if self.lib.kamd_run(self.h, batch.h, ms.ctypes.data) != 0:
            raise self._err("kamd_run")
        return {"scan_ms": float(ms[0]), "lattice_ms": float(ms[1]), "search_ms": float(ms[2]), "finish_ms": float(ms[3])}

    def reruns(self, batch: Batch):
        """(chunks of the last run() that outgrew their scratch and were searched again inside it, wall ms of those passes)."""
        ms = C.c_float(0)
        n = self.lib.kamd_batch_reruns(batch.h, C.byref(ms))
        if n < 0:
            raise self._err("kamd_batch_reruns")
        return n, float(ms.value)

    def fetch(self, batch: Batch, top_n=1) -> Results:
        r = self.lib.kamd_fetch(self.h, batch.h, top_n)
        if not r:
            raise self._err("kamd_fetch")
        return Results(self.lib, r)

    def dump_dict(self) -> bytes:
        buf = np.zeros(1 << 20, np.uint8)
        while True:
            n = self.lib.kamd_dump_dict(self.h, buf.ctypes.data, buf.nbytes)
            if n <= buf.nbytes:
                return bytes(buf[:n])
            buf = np.zeros(int(n * 1.2), np.uint8)

    def split(self, text, match=MATCH_ALL_WITH_NORMALIZING):
        u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        buf = np.zeros(1 << 20, np.uint8)
        while True:
            n = self.lib.kamd_dump_lattices(self.h, u.ctypes.data, len(u), match, buf.ctypes.data, buf.nbytes)
            if n == 0:
                raise self._err("kamd_dump_lattices")
            if n <= buf.nbytes:
                break
            buf = np.zeros(int(n * 1.2), np.uint8)
        o = 0
        (nch,) = struct.unpack_from("<I", buf, o)
        o += 4
        chunks = []
        for _ in range(nch):
            nn, se = struct.unpack_from("<II", buf, o)
            o += 8
            nodes = []
            for _ in range(nn):
                nodes.append(struct.unpack_from("<IIIIiIIIf", buf, o))
                o += 36
            chunks.append((se, nodes))
        return chunks


class Typo:
    """A typo transformer of the low-level ABI (kamd_typo_*): rules added one by one, then prepared for analysis."""

    def __init__(self, lib, continual=float("inf"), lengthening=float("inf")):
        self.lib = lib
        lib.kamd_typo_new.restype = C.c_void_p
        lib.kamd_typo_new.argtypes = [C.c_float, C.c_float]
        lib.kamd_typo_close.argtypes = [C.c_void_p]
        lib.kamd_typo_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_int]
        lib.kamd_typo_prepare.argtypes = [C.c_void_p, C.c_int]
        self.h = lib.kamd_typo_new(continual, lengthening)

    @classmethod
    def from_default(cls, lib, default_typo_set: int):
        """A copy of one of Kiwi's built-in typo sets (reference DefaultTypoSet 0..6: kamd_typo_default)."""
        self = cls(lib)
        lib.kamd_typo_close(self.h)
        lib.kamd_typo_default.restype = C.c_void_p
        lib.kamd_typo_default.argtypes = [C.c_int]
        self.h = lib.kamd_typo_default(default_typo_set)
        if not self.h:
            raise ValueError(default_typo_set)
        return self

    def add(self, orig, error, cost=1.0, cond=0, dialect=0):
        o = np.frombuffer(orig.encode("utf-16-le"), np.uint16)
        e = np.frombuffer(error.encode("utf-16-le"), np.uint16)
        if self.lib.kamd_typo_add(self.h, o.ctypes.data, len(o), e.ctypes.data, len(e), cost, cond, dialect) != 0:
            raise ValueError((orig, error))

    def prepare(self, inverse=True):
        if self.lib.kamd_typo_prepare(self.h, int(inverse)) != 0:
            raise RuntimeError("kamd_typo_prepare")
        return self

    def close(self):
        if self.h:
            self.lib.kamd_typo_close(self.h)
            self.h = None
