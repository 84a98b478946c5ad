/* kiwi_capi.h -- the subset of Kiwi's C API that carries the batched analyze path, re-implemented on the
 * MI355X engine (libkiwi_hip.so).  Signatures, struct layouts, ownership and error conventions are those of
 * /root/reference/include/kiwi/capi.h (v0.23.1); each declaration cites the line it replaces.  A program
 * compiled against the reference's capi.h and linked with this library instead of libkiwi runs unchanged as
 * long as it stays inside this subset (anything else is simply not exported: link error, not silent change).
 *
 * Differences a caller can observe:
 *   - kiwi_init's model_path names a directory holding the reference's own model files (sj.morph + sj.knlm, optionally skipbigram.mdl; or
 *     sj.morph + cong.mdl, optionally nounchr.mdl -- the layout of the reference's models/cong/base), or a raw-model container (or a directory holding `kiwi_amd.raw`).  Its `options` are honoured as in the reference:
 *     KIWI_BUILD_INTEGRATE_ALLOMORPH sets integrate_allomorph; the model type bits select, as KiwiBuilder::getModelType does (KiwiBuilder.cpp:939-961): a
 *     CoNgram model when the container has one -- local scoring for the default and CONG, the GLOBAL scoring (ModelType::congGlobal: distant tokens,
 *     window 7) for LARGEST and CONG_GLOBAL (round 5; a cong.mdl without window sections is scored locally under LARGEST and refused under CONG_GLOBAL,
 *     where the reference would read past the file's sections) --, else Knlm (default, KNLM) or SkipBigram (LARGEST when the container has the tables,
 *     SBG); the LOAD_*_DICT bits are accepted (a raw container's dictionary is baked).
 *   - top_n > 1: the analyses and their scores are the reference's.  Among analyses whose scores are EXACTLY equal the order is this library's
 *     own deterministic one (candidates in lattice order); the reference's order of such ties follows the iteration order of its per-morpheme hash
 *     containers (src/BestPathContainer.hpp:279-483) and differs between its own builds.
 *   - option.blocklist (morpheme sets: kiwi_new_morphset / kiwi_morphset_add / _add_w / _close) is honoured;
 *     kiwi_init's enabled_dialects and option.allowed_dialects / dialect_cost are honoured (round 5): a model with dialect morphemes (MorphemeRaw::dialect of
 *     sj.morph / a raw container) keeps the dictionary forms of enabled dialects, an analysis skips the morphemes of dialects it does not allow, charges
 *     dialect_cost for the others and -- without a transformer of its own -- is corrected with the built-in `dialect` typo set (src/Kiwi.cpp:1037-1041);
 *     tokens report their dialect.
 *   - pretokenized spans (kiwi_pt_init / kiwi_pt_add_span / kiwi_pt_add_token_to_span{,_w} / kiwi_pt_close, the last argument of kiwi_analyze{,_w}) are
 *     honoured (round 6): makePretokenizedSpanGroup (src/Kiwi.cpp:785-946) restated; a span's lattice node is forced on the device, temporary forms / morphemes
 *     live behind the model's device tables for the call, tokens inside span i of their chunk report typo_form_id i + 1, kiwi_res_morpheme_id is -1 for a
 *     temporary morpheme.  Not together with option.typo_transformer (refused).
 *   - top_n > 16 and useOldSplitter are refused with NULL/KIWIERR_FAIL + kiwi_error() instead of being silently ignored.
 */
#ifndef KIWI_CAPI_SUBSET_H
#define KIWI_CAPI_SUBSET_H
#include <stddef.h>
#include <stdint.h>

#define KIWIERR_FAIL -1            /* capi.h:17 */
#define KIWIERR_INVALID_HANDLE -2  /* capi.h:18 */
#define KIWIERR_INVALID_INDEX -3   /* capi.h:19 */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kiwi_s* kiwi_h;                       /* capi.h:29 */
typedef struct kiwi_res* kiwi_res_h;                 /* capi.h:31 */
typedef struct kiwi_morphset* kiwi_morphset_h;       /* capi.h:36 */
typedef struct kiwi_pretokenized* kiwi_pretokenized_h; /* capi.h:37 */
typedef struct kiwi_typo* kiwi_typo_h;                   /* capi.h:35 */
typedef struct kiwi_prepared_typo* kiwi_prepared_typo_h; /* capi.h:38 */
typedef unsigned short kchar16_t;                    /* capi.h:39 */

typedef struct {                                     /* capi.h:43-61 */
	uint32_t chr_position, word_position, sent_position, line_number;
	uint16_t length;
	uint8_t tag;
	union { uint8_t sense_id; uint8_t script; };
	float score, typo_cost;
	uint32_t typo_form_id, paired_token, sub_sent_position;
	uint16_t dialect;
} kiwi_token_info_t;

typedef struct {                                     /* capi.h:72-86 */
	uint8_t integrate_allomorph;
	float cut_off_threshold, oov_rule_scale, oov_rule_bias, oov_chr_bias, oov_global_weight, oov_local_weight, oov_global_min_freq;
	float space_penalty, typo_cost_weight;
	uint32_t max_unk_form_size, max_unk_form_size_followed_by_j_class, space_tolerance;
} kiwi_config_t;

typedef struct {                                     /* capi.h:662-670 */
	int match_options;
	kiwi_morphset_h blocklist;
	int open_ending;
	int allowed_dialects;
	float dialect_cost;
	kiwi_prepared_typo_h typo_transformer;
	float typo_threshold;
} kiwi_analyze_option_t;

typedef int (*kiwi_reader_t)(int, char*, void*);       /* capi.h:104 */
typedef int (*kiwi_reader_w_t)(int, kchar16_t*, void*); /* capi.h:105 */
typedef int (*kiwi_receiver_t)(int, kiwi_res_h, void*); /* capi.h:144 */

enum { KIWI_NUM_THREADS = 0x8001, KIWI_GPU_BATCH_SIZE = 0x9001 };   /* capi.h:222 (+ one extension option) */
enum {                                               /* capi.h:158-171 */
	KIWI_BUILD_INTEGRATE_ALLOMORPH = 1, KIWI_BUILD_LOAD_DEFAULT_DICT = 2, KIWI_BUILD_LOAD_TYPO_DICT = 4, KIWI_BUILD_LOAD_MULTI_DICT = 8, KIWI_BUILD_DEFAULT = 15,
	KIWI_BUILD_MODEL_TYPE_DEFAULT = 0x0000, KIWI_BUILD_MODEL_TYPE_LARGEST = 0x0100, KIWI_BUILD_MODEL_TYPE_KNLM = 0x0200, KIWI_BUILD_MODEL_TYPE_SBG = 0x0300,
	KIWI_BUILD_MODEL_TYPE_CONG = 0x0400, KIWI_BUILD_MODEL_TYPE_CONG_GLOBAL = 0x0500,
};
enum {                                               /* capi.h:485-491 */
	KIWI_TYPO_WITHOUT_TYPO = 0, KIWI_TYPO_BASIC_TYPO_SET = 1, KIWI_TYPO_CONTINUAL_TYPO_SET = 2, KIWI_TYPO_BASIC_TYPO_SET_WITH_CONTINUAL = 3,
	KIWI_TYPO_LENGTHENING_TYPO_SET = 4, KIWI_TYPO_BASIC_TYPO_SET_WITH_CONTINUAL_AND_LENGTHENING = 5, KIWI_TYPO_DIALECT = 6,
};

const char* kiwi_version(void);                      /* capi.h:238 */
const char* kiwi_error(void);                        /* capi.h:245 */
void kiwi_clear_error(void);                         /* capi.h:252 */
kiwi_h kiwi_init(const char* model_path, int num_threads, int options, int enabled_dialects);   /* capi.h:599 */
void kiwi_set_global_config(kiwi_h handle, kiwi_config_t config);                                /* capi.h:607 */
kiwi_config_t kiwi_get_global_config(kiwi_h handle);                                             /* capi.h:615 */
void kiwi_set_option(kiwi_h handle, int option, int value);                                      /* capi.h:623 */
int kiwi_get_option(kiwi_h handle, int option);                                                  /* capi.h:636 */
void kiwi_set_option_f(kiwi_h handle, int option, float value);                                  /* capi.h:644 */
float kiwi_get_option_f(kiwi_h handle, int option);                                              /* capi.h:652 */
kiwi_res_h kiwi_analyze_w(kiwi_h handle, const kchar16_t* text, int top_n, kiwi_analyze_option_t option, kiwi_pretokenized_h pretokenized); /* capi.h:684 */
kiwi_res_h kiwi_analyze(kiwi_h handle, const char* text, int top_n, kiwi_analyze_option_t option, kiwi_pretokenized_h pretokenized);        /* capi.h:698 */
int kiwi_analyze_mw(kiwi_h handle, kiwi_reader_w_t reader, kiwi_receiver_t receiver, void* user_data, int top_n, kiwi_analyze_option_t option); /* capi.h:711 */
int kiwi_analyze_m(kiwi_h handle, kiwi_reader_t reader, kiwi_receiver_t receiver, void* user_data, int top_n, kiwi_analyze_option_t option);    /* capi.h:724 */
int kiwi_close(kiwi_h handle);                                                                   /* capi.h:771 */
const char* kiwi_tag_to_string(kiwi_h handle, uint8_t pos_tag);                                  /* capi.h:780 */
int kiwi_res_size(kiwi_res_h result);                                                            /* capi.h:788 */
float kiwi_res_prob(kiwi_res_h result, int index);                                               /* capi.h:797 */
int kiwi_res_word_num(kiwi_res_h result, int index);                                             /* capi.h:806 */
const kiwi_token_info_t* kiwi_res_token_info(kiwi_res_h result, int index, int num);             /* capi.h:816 */
int kiwi_res_morpheme_id(kiwi_res_h result, int index, int num, kiwi_h kiwi_handle);             /* capi.h:827 */
const kchar16_t* kiwi_res_form_w(kiwi_res_h result, int index, int num);                         /* capi.h:837 */
const kchar16_t* kiwi_res_tag_w(kiwi_res_h result, int index, int num);                          /* capi.h:847 */
const char* kiwi_res_form(kiwi_res_h result, int index, int num);                                /* capi.h:857 */
const char* kiwi_res_tag(kiwi_res_h result, int index, int num);                                 /* capi.h:867 */
int kiwi_res_position(kiwi_res_h result, int index, int num);                                    /* capi.h:877 */
int kiwi_res_length(kiwi_res_h result, int index, int num);                                      /* capi.h:887 */
int kiwi_res_word_position(kiwi_res_h result, int index, int num);                               /* capi.h:897 */
int kiwi_res_sent_position(kiwi_res_h result, int index, int num);                               /* capi.h:907 */
float kiwi_res_score(kiwi_res_h result, int index, int num);                                     /* capi.h:917 */
float kiwi_res_typo_cost(kiwi_res_h result, int index, int num);                                 /* capi.h:927 */
int kiwi_res_close(kiwi_res_h result);                                                           /* capi.h:937 */
const char* kiwi_get_script_name(uint8_t script);                                                /* capi.h:1417 */
/* morpheme sets: kiwi_analyze_option_t::blocklist (candidates whose morpheme -- or a chunk of it -- is in the set are left out of the search) */
kiwi_morphset_h kiwi_new_morphset(kiwi_h handle);                                                /* capi.h:660 */
int kiwi_morphset_add(kiwi_morphset_h handle, const char* form, const char* tag);                /* capi.h:1243 */
int kiwi_morphset_add_w(kiwi_morphset_h handle, const kchar16_t* form, const char* tag);         /* capi.h:1253 */
int kiwi_morphset_close(kiwi_morphset_h handle);                                                 /* capi.h:1263 */
/* Pretokenized spans: the object can be built and closed; kiwi_analyze* accepts it as long as it holds no span (analysing with spans is refused loudly:
 * the lattice kernels do not build span nodes yet) */
kiwi_pretokenized_h kiwi_pt_init(void);                                                          /* capi.h:1351 */
int kiwi_pt_add_span(kiwi_pretokenized_h handle, int begin, int end);                            /* capi.h:1367 */
int kiwi_pt_add_token_to_span(kiwi_pretokenized_h handle, int span_id, const char* form, const char* tag, int begin, int end);          /* capi.h:1382 */
int kiwi_pt_add_token_to_span_w(kiwi_pretokenized_h handle, int span_id, const kchar16_t* form, const char* tag, int begin, int end);   /* capi.h:1397 */
int kiwi_pt_close(kiwi_pretokenized_h handle);                                                   /* capi.h:1405 */
/* typo transformers (rule container, preparation, option.typo_transformer / typo_threshold of kiwi_analyze*): parity-checked on the MI355X against the
 * oracle and the real reference (tests/test_gpu_typo.py, tests/test_gpu_capi.py). */
kiwi_typo_h kiwi_typo_init(void);                                                                /* capi.h:469 */
kiwi_typo_h kiwi_typo_get_basic(void);                                                           /* capi.h:480 */
kiwi_typo_h kiwi_typo_get_default(int kiwi_typo_set);                                            /* capi.h:501 */
int kiwi_typo_add(kiwi_typo_h handle, const char** orig, int orig_size, const char** error, int error_size, float cost, int condition); /* capi.h:510 */
kiwi_typo_h kiwi_typo_copy(kiwi_typo_h handle);                                                  /* capi.h:519 */
int kiwi_typo_update(kiwi_typo_h handle, kiwi_typo_h src);                                       /* capi.h:530 */
int kiwi_typo_scale_cost(kiwi_typo_h handle, float scale);                                       /* capi.h:539 */
int kiwi_typo_set_continual_typo_cost(kiwi_typo_h handle, float threshold);                      /* capi.h:550 */
int kiwi_typo_set_lengthening_typo_cost(kiwi_typo_h handle, float threshold);                    /* capi.h:561 */
int kiwi_typo_close(kiwi_typo_h handle);                                                         /* capi.h:570 */
kiwi_prepared_typo_h kiwi_typo_prepare(kiwi_typo_h handle);                                      /* capi.h:580 */
int kiwi_prepared_typo_close(kiwi_prepared_typo_h handle);                                       /* capi.h:588 */

#ifdef __cplusplus
}
#endif
#endif
