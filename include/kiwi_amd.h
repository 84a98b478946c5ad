/* kiwi_amd.h -- low-level C ABI of the MI355X batched analyze path (libkiwi_hip.so).
 *
 * These entry points are what a binding of the reference would call in place of its per-sentence
 * loop; the Kiwi-compatible drop-in symbols (kiwi_init / kiwi_analyze* / kiwi_res_*) are declared in
 * kiwi_capi.h and are implemented on top of this layer.  Plain pointers and sizes only.
 *
 *   kamd_open            <- kiwi_init + KiwiBuilder::build          (/root/reference/src/capi/kiwi_c.cpp:717-736)
 *   kamd_analyze_batch   <- Kiwi::analyze(topN, reader, receiver)   (/root/reference/include/kiwi/Kiwi.h:402-454)
 *   kamd_stage/run/fetch <- the same, split so a benchmark can time the device part with inputs resident in HBM
 *   kamd_res_*           <- kiwi_res_* accessors                     (/root/reference/src/capi/kiwi_c.cpp:1036-1285)
 */
#ifndef KIWI_AMD_H
#define KIWI_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct kamd_engine* kamd_engine_h;
typedef struct kamd_batch* kamd_batch_h;     /* chunks of a batch staged in HBM */
typedef struct kamd_results* kamd_results_h; /* per-text token lists */

typedef struct
{
	uint32_t position, word_position, sent_position, line_number;
	uint16_t length; uint8_t tag; uint8_t sense_or_script;
	float score, typo_cost;
	uint32_t typo_form_id, paired_token, sub_sent_position;
	uint16_t dialect; uint16_t form_len;
	int32_t morph_id;
	uint64_t form_off;   /* offset of the NUL-terminated UTF-16 form in kamd_res_forms(r, text) */
} kamd_token_t;

/* returns NULL on failure; kamd_last_error() (thread local) tells why.  device < 0: current/first device */
kamd_engine_h kamd_open(const char* raw_model_path, int device);
/* the same with KiwiBuilder's enabledDialects (kiwi_init's last argument, /root/reference/src/KiwiBuilder.cpp:963-967, 2500-2504; KIWI_DIALECT_* bits):
 * dictionary forms whose morphemes all belong to dialects that are not enabled stay out of the trie */
kamd_engine_h kamd_open_dialects(const char* raw_model_path, int device, int enabled_dialects);
/* ... and with the language-model type chosen as kiwi_init's KIWI_BUILD_MODEL_TYPE_* bits do (/root/reference/src/KiwiBuilder.cpp:939-961): lm_mode 0 = what the
 * container offers (CoNgram local, else SkipBigram, else Knlm), 1 = Knlm, 2 = SkipBigram, 3 = CoNgram (local scoring), 4 = CoNgram GLOBAL (ModelType::congGlobal:
 * distant tokens, window 7 -- the file must carry the window sections) */
kamd_engine_h kamd_open_mode(const char* raw_model_path, int device, int lm_mode, int enabled_dialects);
void kamd_close(kamd_engine_h h);
const char* kamd_last_error(void);

/* KiwiConfig fields used on this path (include/kiwi/Kiwi.h:150-167) */
int kamd_set_config(kamd_engine_h h, float cut_off_threshold, float space_penalty, float typo_cost_weight,
	uint32_t max_unk_form_size, uint32_t max_unk_form_size_followed_by_jclass, uint32_t space_tolerance, int integrate_allomorph);
/* KiwiConfig::oovChrBias: subtracted from the character model's score of an unknown form when match_options carry Match::oovChrModel (1 << 8);
 * that option needs a model with the character model (nounchr.mdl next to cong.mdl / the raw container's `nounchr` section) */
int kamd_set_oov_chr_bias(kamd_engine_h h, float bias);
/* KiwiConfig::oovGlobalWeight / oovLocalWeight / oovGlobalMinFreq (defaults 35 / 3 / 4): Match::oovChrFreqModel (2 << 8) and oovChrFreqBranchModel (3 << 8) mix the
 * character model's score of an unknown form with how often its prefixes occur in the text under analysis (reference src/UnkFormScorer.cpp:68-121) */
int kamd_set_oov_freq_params(kamd_engine_h h, float global_weight, float local_weight, float global_min_freq);

/* texts: concatenated UTF-16; offsets[n+1].  top_n in 1..16 (null + error otherwise); analyses beyond the best differ from a given reference run only in exact ties (top-N). */
kamd_results_h kamd_analyze_batch(kamd_engine_h h, const uint16_t* texts, const uint64_t* offsets, uint32_t n,
	uint32_t top_n, uint64_t match_options, int open_ending, int host_threads);

kamd_batch_h kamd_stage(kamd_engine_h h, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint64_t match_options, int open_ending, int host_threads);
/* launches the kernels on the staged batch; ms_out[4] = {dictionary scan, lattice build (+ candidate expansion), best-path search,
   end stage (candidate sort + back-trace)} in milliseconds, summed over the sub-batches (HIP events on the launching streams) */
int kamd_run(kamd_engine_h h, kamd_batch_h b, float* ms_out);
kamd_results_h kamd_fetch(kamd_engine_h h, kamd_batch_h b, uint32_t top_n);
/* info[0]=chunks, [1]=non-space normalised units ("jamo"), [2]=device bytes of the staged batch */
int kamd_batch_info(kamd_batch_h b, uint64_t* info3);
/* kamd_run searches every chunk to the end: chunks that outgrow their device scratch in the first pass are searched again, together, with
   larger capacities before it returns.  Returns how many chunks of the last kamd_run that were (< 0: error); *ms_out (optional) = the wall
   time of those extra passes, which kamd_run's ms_out[4] (first pass only) does not contain */
int kamd_batch_reruns(kamd_batch_h b, float* ms_out);
/* State arenas of the staged batch: a chunk's arena holds what the typical chunk needs; one that fills its arena carries on in an arena twice as large taken from the
   batch's pool (append-only, one atomic add per growth; the first pass only -- chunks the pool cannot serve are re-run as above).
   out[0] = states in the chunks' own arenas, [1] = states of the pool, [2] = states of the pool asked for by the last fetched run (> [1]: the pool ran out) */
int kamd_batch_pool(kamd_batch_h b, uint64_t* out3);
void kamd_batch_close(kamd_batch_h b);

uint32_t kamd_res_texts(kamd_results_h r);
uint32_t kamd_res_size(kamd_results_h r, uint32_t text);                       /* number of analyses (<= top_n) */
float kamd_res_prob(kamd_results_h r, uint32_t text, uint32_t index);
uint32_t kamd_res_token_num(kamd_results_h r, uint32_t text, uint32_t index);
const kamd_token_t* kamd_res_tokens(kamd_results_h r, uint32_t text, uint32_t index);
/* form pool the tokens of `text` index with form_off (results are stored in flat segments of consecutive texts; each has its own pool) */
const uint16_t* kamd_res_forms(kamd_results_h r, uint32_t text);
/* bytes the device -> host copy of this batch moved (chunk summaries + the path headers and token records actually produced) */
uint64_t kamd_res_d2h_bytes(kamd_results_h r);
void kamd_res_close(kamd_results_h r);
/* Multi-GPU result gather (one process per GPU; the path shards by independent texts): a rank packs its results into one
 * position-independent buffer (returns the size needed; writes when cap suffices), ships it to the gathering rank (RCCL / gloo: kiwi_amd/dist.py),
 * which merges the parts of an index-strided split (text g -> part g % n_parts) back into input order. */
size_t kamd_res_pack(kamd_results_h r, uint8_t* out, size_t cap);
kamd_results_h kamd_res_merge_strided(const uint8_t* const* parts, const size_t* sizes, uint32_t n_parts);

/* test hooks: baked dictionary dump and the lattices of one text, in the byte layouts of oracle/ref_bridge.cpp */
/* developer probe: exp_out[i] = expf, log_out[i] = logf of x[i] computed ON THE DEVICE by csrc/exact_math.hpp (bit-identical to glibc) */
int kamd_debug_exact_math(const float* x, float* exp_out, float* log_out, uint32_t n);
/* developer probe: tanh_out[i] = tanhf of x[i] computed ON THE DEVICE by csrc/exact_math.hpp (bit-identical to glibc 2.35; the frequency-based unknown-form scores use it) */
int kamd_debug_exact_tanh(const float* x, float* tanh_out, uint32_t n);
/* developer probe: the GLOBAL CoNgram model's score (reference CoNgramModel::progress / progressMatrix*, src/CoNgramModel.cpp:802-868, 1037-1466; csrc/cong_global.hpp)
 * of next[i] after context id ctx[i] with the seven history words hist7[7 * i ..], computed ON THE DEVICE; flags[i] bit 0 = the progressMatrix* entry, bit 1 = output
 * scale first.  The model must carry the window sections (cong.mdl windowSize 7).  The search does not use this model type yet (kiwi_init refuses CONG_GLOBAL). */
int kamd_debug_cong_global(kamd_engine_h h, const uint32_t* ctx, const uint32_t* hist7, const uint32_t* next, const uint8_t* flags, float* out, uint32_t n);
/* developer probe, HOST side: the pattern recogniser of the text preparation (reference matchPattern, src/PatternMatcher.cpp:380) at text[0]:
 * matched length | tag << 32, 0 = no pattern starts here.  `left` = the unit before text[0] (u' ' at the start) */
uint64_t kamd_debug_match_pattern(uint16_t left, const uint16_t* text, uint32_t len, uint64_t match_options);
/* developer probe, HOST side, no device: one SkipBigram LM step (reference SbgState::nextImpl, src/SkipBigramModel.hpp:169-182) on top of
 * the Knlm log-likelihood `knlm_ll`, by the code the search kernel shares (csrc/sbg_eval.hpp); hist8 / pos are updated in place */
int kamd_debug_sbg_next(const char* raw_model_path, uint32_t* hist8, uint32_t* pos, uint32_t wid, float knlm_ll, float* ll_out);
/* Typo transformers (csrc/typo.hpp): rule container (reference TypoTransformer::addTypo / update / scaleCost), preparation
 * (TypoTransformer::prepare) and the typo graph of a text (PreparedTypoTransformer::generateGraph).  kamd_analyze_batch_typo / kamd_stage_typo analyse with a prepared one; kamd_typo_graph dumps the graph for tests:
 * {u32 normLen, u16[]; u32 nNodes; per node: u32 formLen, u16[], u32 endPos, f32 typoCost, u32 prevOffset, u32 siblingOffset, u8 continualTypoIdx, u16 dialect}; u32 maxContinualTypoIdx.
 * left_cond: CondVowel value (0 none, 1 any, 2 vowel, 8 applosive, 9 continual, 10 boundary); dialect: Dialect bit mask. */
typedef struct kamd_typo* kamd_typo_h;
kamd_typo_h kamd_typo_new(float continual_cost, float lengthening_cost);     /* INFINITY = that kind of typo is off */
void kamd_typo_close(kamd_typo_h t);
kamd_typo_h kamd_typo_default(int default_typo_set);   /* a copy of one of Kiwi's built-in sets (reference DefaultTypoSet 0..6, capi.h:485-491); NULL + error otherwise */
int kamd_typo_add(kamd_typo_h t, const uint16_t* orig, uint32_t n_orig, const uint16_t* error, uint32_t n_error, float cost, int left_cond, int dialect);
int kamd_typo_add_entry(kamd_typo_h t, const uint16_t* orig, uint32_t n_orig, const uint16_t* error, uint32_t n_error, float cost, int left_cond, int dialect);   /* an already expanded rule, as update() inserts it */
int kamd_typo_set_costs(kamd_typo_h t, float continual_cost, float lengthening_cost);
int kamd_typo_scale(kamd_typo_h t, float scale);
int kamd_typo_prepare(kamd_typo_h t, int inverse);
size_t kamd_typo_graph(kamd_typo_h t, const uint16_t* text, uint32_t len, int allowed_dialect, int normalize_coda, uint8_t* out, size_t cap);
/* kamd_analyze_batch with a prepared typo transformer -- AnalyzeOption::typoTransformer / typoThreshold / allowedDialects of the reference.
 * SkipBigram models are refused. */
kamd_results_h kamd_analyze_batch_typo(kamd_engine_h h, kamd_typo_h t, float threshold, int allowed_dialect, const uint16_t* texts, const uint64_t* offsets, uint32_t n_texts,
                                       uint32_t top_n, uint64_t match_options, int open_ending, int host_threads);
/* ... and kamd_stage with one (the transformer must outlive the batch): kamd_run / kamd_fetch as usual */
kamd_batch_h kamd_stage_typo(kamd_engine_h h, kamd_typo_h t, float threshold, int allowed_dialect, const uint16_t* texts, const uint64_t* offsets, uint32_t n_texts,
                             uint64_t match_options, int open_ending, int host_threads);
/* Morpheme sets -- AnalyzeOption::blocklist of the reference (include/kiwi/Kiwi.h AnalyzeOption, src/capi/kiwi_c.cpp:851-864, 1779-1826): candidates
 * whose combined morpheme or one of whose chunks is in the set are left out of the search (src/PathEvaluator.hpp:385, 892).
 * kamd_morphset_add: Kiwi::findMorphemes(form, tag) -- every morpheme of the dictionary form `form` with POS tag id `tag` (< 0 or 0: any), irregularity
 * ignored; returns how many were added (0: no such form), < 0 on error.  The set must outlive the calls that use it. */
typedef struct kamd_morphset* kamd_morphset_h;
kamd_morphset_h kamd_morphset_new(kamd_engine_h h);
int kamd_morphset_add(kamd_morphset_h m, const uint16_t* form, uint32_t len, int tag);
void kamd_morphset_close(kamd_morphset_h m);
/* kamd_analyze_batch with the per-call options of the reference's AnalyzeOption: a prepared typo transformer (or NULL) and a blocklist (or NULL) */
kamd_results_h kamd_analyze_batch_opt(kamd_engine_h h, kamd_typo_h t, float threshold, int allowed_dialect, kamd_morphset_h blocklist,
                                      const uint16_t* texts, const uint64_t* offsets, uint32_t n_texts, uint32_t top_n, uint64_t match_options, int open_ending, int host_threads);
/* ... and AnalyzeOption::allowedDialects / dialectCost (/root/reference/include/kiwi/Kiwi.h:89-99, src/PathEvaluator.hpp:231-236, 386, 893): morphemes of a dialect
 * that is neither standard nor allowed are not candidates, those of an allowed dialect cost dialect_cost.  With a dialect allowed and t == NULL the
 * built-in typo set DefaultTypoSet::dialect is applied with threshold 2.5, as the reference does (src/Kiwi.cpp:1037-1041) */
kamd_results_h kamd_analyze_batch_dialect(kamd_engine_h h, kamd_typo_h t, float threshold, int allowed_dialect, float dialect_cost, kamd_morphset_h blocklist,
                                          const uint16_t* texts, const uint64_t* offsets, uint32_t n_texts, uint32_t top_n, uint64_t match_options, int open_ending, int host_threads);
/* Kiwi::analyze(text, option, pretokenized) for ONE text -- the spans the caller has tokenised already (/root/reference/src/Kiwi.cpp:785-946, 1043-1051;
 * src/KTrie.cpp:782-790, 1177-1210; include/kiwi/Types.h:393-413): `spans` holds, per span, {begin, end, n_tokens} followed by n_tokens x {offset of the token's
 * form in `forms`, its length, begin, end (relative to the span's begin), tag id, inferRegularity}; offsets in UTF-16 units of `text`.  A span without tokens or with
 * one token that is a single-candidate dictionary entry points at that form; anything else gets temporary forms / morphemes, uploaded behind the model's tables for
 * this call (tokens of temporary morphemes report morph_id -1); every token inside span i of its chunk reports typo_form_id i + 1.  Not together with a typo
 * transformer.  NULL + kamd_last_error() for overlapping or empty spans. */
kamd_results_h kamd_analyze_pretokenized(kamd_engine_h h, const uint16_t* text, uint32_t len, uint32_t top_n, uint64_t match_options, int open_ending,
                                         const uint32_t* spans, uint32_t n_spans, const uint16_t* forms);
/* parity hook: the lattices the device builds OVER the typo graphs of a text's chunks (csrc/typo_lattice_kernel.hip), in the layout of kamd_dump_lattices;
 * 0 + kamd_last_error() on failure */
/* parity hook of the device typo-graph kernel (the analyze path generates typo graphs on the GPU): kamd_typo_graph's layout + two bytes per
 * node (type, script of the last character of its form); use_device 0 = the same from the host module */
size_t kamd_typo_graph_device(kamd_engine_h h, kamd_typo_h t, const uint16_t* text, uint32_t len, int allowed_dialect, int normalize_coda, int use_device, uint8_t* out, size_t cap);
size_t kamd_typo_lattices(kamd_engine_h h, kamd_typo_h t, float threshold, int allowed_dialect, const uint16_t* text, uint32_t len, uint64_t match_options, uint8_t* out, size_t cap);
size_t kamd_dump_dict(kamd_engine_h h, uint8_t* out, size_t cap);
size_t kamd_dump_lattices(kamd_engine_h h, const uint16_t* text, uint32_t len, uint64_t match_options, uint8_t* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
