/* libdia_hip.so — C ABI of the MI355X-native Dia decode path.
 *
 * The reference (babybirdprd/dia-tts-prune) has no native code and no FFI seam: its hot path is
 * PyTorch ops called from dia/layers.py and dia/model.py.  This header therefore DEFINES the
 * boundary a maintainer would bind (ctypes stub in INTEGRATION.md); every entry point names the
 * reference call site whose arithmetic it replaces.  All pointers are raw device pointers unless
 * marked "host"; no torch types cross this boundary.  Every function returns 0 on success and a
 * negative DIA_E_* code on failure; dia_last_error() gives the message (thread-local).
 *
 * One engine per device / stream; an engine is not re-entrant (the reference's Dia object is not
 * either: SURVEY.md §8b "Threading").
 */
#ifndef DIA_HIP_H
#define DIA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIA_ABI_VERSION 6

#define DIA_OK 0
#define DIA_E_ARG (-1)     /* bad argument / unsupported shape */
#define DIA_E_HIP (-2)     /* HIP runtime error */
#define DIA_E_STATE (-3)   /* call order violated */

#define DIA_KV_F32 0       /* parity mode: K/V caches in float32 */
#define DIA_KV_BF16 1      /* perf mode (reference GPU bf16 path, state.py:142-151) */
#define DIA_KV_BF16X2 2    /* every K / V value as hi + lo bf16 (16 significand bits) in two planes of the bf16 layouts, `kv_plane_stride`
                            * elements apart: the bytes of the fp32 caches, the MFMA attention kernel, logits within 1e-3 of the fp32 path */

/* GEMM epilogues */
#define DIA_EPI_SCALE_STORE 0  /* out[m][n] = acc * inv_rms[m]                      (q/k/v, cross-q, logits) */
#define DIA_EPI_RESID_EMIT 1   /* x[m][n] += acc; emit planes(x*g_next), strip ssq   (o_proj, wo)             */
#define DIA_EPI_SWIGLU_EMIT 2  /* h = silu(gate*inv)* (up*inv); emit planes(h)       (wi_fused)               */
#define DIA_EPI_CROSSKV 3      /* K = RoPE(acc*inv) , V = acc*inv -> cross caches    (precompute_cross_attn_cache) */

/* attention modes */
#define DIA_ATTN_SELF 0   /* decoder self-attention over the growing cache (layers.py:541-555) */
#define DIA_ATTN_CROSS 1  /* decoder cross-attention over encoder K/V, cond rows (layers.py:560-574) */
#define DIA_ATTN_ENC 2    /* encoder bidirectional self-attention over L packed tokens (layers.py:396-404) */

const char* dia_last_error(void);
int dia_abi_version(void);
/* number of visible HIP devices, or a negative error */
int dia_device_count(void);
/* Tuning / debug overrides of the launch heuristics (process-wide; csrc/tuning.hpp lists the knobs, e.g. "attn_nz",
 * "gemm_spw", "wo_sk").  value < 0 clears a knob.  Nothing on the launch path reads the environment; the DIA_TUNE
 * variable ("name=value,...") is read once when the first engine or kernel is initialised.  The reference has no
 * counterpart (its only switches are torch.compile / dtype, model.py:631-647). */
int dia_set_tuning(const char* name, int value);
int dia_get_tuning(const char* name);
/* 1 when the library was built with EXPERIMENTS=1 (measured-and-rejected kernels: dia_mlp_fused, the sparse weight
 * stream of dia_gemm_args.sp_blocks, earlier two-m-tile GEMM forms); 0 for the product build, in which those entry
 * points fail with DIA_E_ARG */
int dia_has_experiments(void);

/* ------------------------------------------------------------------------------------------------
 * Kernel-level entry points (unit parity tests call these; the engine below chains them).
 * `stream` is a hipStream_t passed as void*.
 * ------------------------------------------------------------------------------------------------ */

/* out[M][N] = X[M][K] . W[K][N] with X given as three bf16 planes and W as bf16 tiles.
 * Replaces DenseGeneral.forward = torch.tensordot (layers.py:55-66) and, through its epilogues, the
 * surrounding RMSNorm scale (layers.py:541,560,579,714), residual adds (555,574,582), the SwiGLU
 * gate (layers.py:95-101) and the cross K/V RoPE + cache write (layers.py:652-663). */
typedef struct {
  const void* A;            /* planes base, bf16 */
  int64_t a_plane_stride;   /* elements between planes */
  int32_t a_ktiles;         /* K/32 of the plane layout */
  int32_t M;                /* valid rows */
  const void* W;            /* weight tiles, bf16 [nstrips][KT][64][8] */
  int32_t KT;               /* K/32 */
  int32_t nstrips;          /* N/16 */
  int32_t epi;              /* DIA_EPI_* */
  int32_t nw;               /* waves per workgroup (4, 8 or 16); 0 = choose */
  /* row scale = rsqrt(sum_i ssq_in[i][m] * inv_d + eps); ssq_in == NULL -> scale 1 */
  const float* ssq_in;
  int32_t ssq_in_n;
  int32_t ssq_ld;           /* row stride of ssq arrays (padded row count) */
  float inv_d;
  float eps;
  float* out;               /* SCALE_STORE: [M][ldo]; RESID_EMIT: x in/out [M][ldo] */
  int32_t ldo;
  int32_t spw;              /* strips per workgroup for the M <= 4 kernel; 0 = choose */
  const float* gnext;       /* RESID_EMIT: norm weight of the consumer (NULL -> 1) */
  void* P;                  /* emitted planes (RESID_EMIT, SWIGLU_EMIT) */
  int64_t p_plane_stride;
  int32_t p_ktiles;
  int32_t _pad1;
  float* ssq_out;           /* RESID_EMIT: [nstrips][ssq_ld] */
  /* CROSSKV */
  void* kc;                 /* K cache of one layer, [kv_batch][heads][kv_cap][128] */
  void* vc;
  int32_t kv_dtype;
  int32_t kv_heads;
  int32_t kv_cap;
  int32_t kv_batch_index;
  const float* cos_t;       /* [npos][64] */
  const float* sin_t;
  /* structured-pruned (compacted) checkpoints — all optional, NULL = dense:
   * cmap      RESID_EMIT: int32 [N]; column n of the residual stream is emitted at plane position
   *           cmap[n] (the consumer's compacted K order) or dropped when cmap[n] < 0;
   * strip_map SCALE_STORE / CROSSKV: int32 [nstrips]; compact strip s holds the 16 output columns
   *           of original strip strip_map[s] (whole heads dropped by a later o_proj). */
  const int32_t* cmap;
  const int32_t* strip_map;
  /* cross-workgroup split-K (long K, few strips; one m-tile only): sk_scratch holds nstrips*sk*256 floats,
   * sk_tickets nstrips int32 zeroed once by the caller (the kernel re-zeroes them).  sk: 0 or 1 = off,
   * n > 1 = n workgroups per strip (opt-in: measured slower than 1 on the decode shapes).  With 17..32 rows
   * (two m-tiles) both sizes double and sk_scratch_floats must state the capacity. */
  float* sk_scratch;
  int32_t* sk_tickets;
  int32_t sk;
  int32_t kv_vblocked;      /* CROSSKV: 1 = write V in the blocked layout of dia_attn_args.v_blocked (kv_cap % 32 == 0) */
  /* CROSSKV over a packed prefill batch: row m belongs to utterance row_b[m] (-1 = padding, skipped) at text
   * position m - seg_off[row_b[m]]; both NULL = one utterance, kv_batch_index, position m. */
  const int32_t* row_b;
  const int32_t* seg_off;
  /* capacity of sk_scratch in floats (0 = not stated: only the explicit `sk` split-K, nstrips * sk * 256 floats, is
   * assumed).  With 17..32 rows and K > 2048 dia_gemm splits K by itself when nstrips * (KT / 64) * 512 floats fit;
   * the column-block form (k_gemm_blk32, debug knob) needs nstrips * (KT / 8) * 512 floats. */
  int64_t sk_scratch_floats;
  /* zero-skipping weight stream of an unstructured-pruned matrix (dia_hip.layout.sparse_tile_weight) instead of W:
   * sp_blocks = concatenated tile blocks, sp_toff[strip * KT + ktile] = (block offset / 16) << 8 | chunks.
   * M <= 4, KT in {16, 32, 64}, no split-K; results are bit-identical to the dense tiles of the same matrix. */
  const void* sp_blocks;
  const uint32_t* sp_toff;
  /* fp32 ACTIVATION TILES: bit 0 = A is, bit 1 = P receives the fragment order of one plane ([mtile][ktile][lane][8]) with 4-byte
   * elements — 4 bytes per value instead of the 6 of three planes (a_plane_stride / p_plane_stride unused; a buffer sized for three
   * planes holds them).  The 5..128-row kernel is instantiated per format and splits each value into its planes in registers: same
   * arithmetic bit for bit, a third less activation traffic per workgroup.  Both bits (or bit 0 alone when nothing is emitted):
   * the M <= 4 GEMV (splits while staging its image through LDS) and the 8-wave forms of the 5..128-row kernel; a mixed-format
   * call runs the generic kernel.  Not available in the tiled prefill kernel. */
  int32_t act_f32;
  /* 0 / 1: W holds one bf16 tile set (exact for bf16-representable checkpoints).  3: W holds THREE tile sets back to back, the
   * hi / mid / lo bf16 planes of fp32 weights (hi + mid + lo == w exactly; plane stride = KT * nstrips * 512 elements): the
   * products of a genuine fp32 checkpoint, exact like the activations' — through the generic kernel only (no split-K, no
   * persistent forms): the parity configuration for checkpoints that bf16 cannot hold, not a fast path. */
  int32_t w_planes;
  int64_t kv_plane_stride;  /* CROSSKV with kv_dtype DIA_KV_BF16X2: elements between the hi and the lo plane of kc / vc */
  int32_t w_layout;         /* 0: 16-column strips (above); 1: diagonal tiles of 4-column groups (dia_hip/layout.py diag_tile_weight): W = bf16
                             * [N/4][K/128][64][8], `nstrips` = N/8 (8-column half strips, one workgroup each, the whole K, no split-K);
                             * M <= 4, DIA_EPI_RESID_EMIT, fp32 tiles in and out; ssq_out receives N/8 partials per row */
  /* CROSSKV for SEVERAL decoder layers in one launch (the prefill's cross-K/V projections share their input): kv_layer_strips > 0 =
   * original strips per layer (kv_heads * 16); strip s (after strip_map, whose entries then count layer * kv_layer_strips + strip)
   * belongs to layer s / kv_layer_strips and is written kv_layer_stride ELEMENTS behind kc / vc per layer.  W = the layers' tile
   * sets back to back, nstrips = their sum.  0 = one layer (above). */
  int32_t kv_layer_strips;
  int64_t kv_layer_stride;
} dia_gemm_args;
int dia_gemm(const dia_gemm_args* a, void* stream);
/* same launch, bracketed by dispatch-level start/stop events (hipExtLaunchKernelGGL); returns the
 * kernel's own duration in milliseconds and synchronises on its end */
/* Fused SwiGLU MLP for 1-2 rows (batch 1): wi (DIA_EPI_SWIGLU_EMIT into planes P) and wo (DIA_EPI_RESID_EMIT reading
 * those planes, with sk_scratch / sk_tickets for a split-K of 2) in one persistent launch with a grid barrier
 * between the phases (MlpBlock.forward, layers.py:92-105).  `barrier`: two int32 on the device, zeroed by the
 * caller once per session; barrier[1] != 0 afterwards means a workgroup gave up waiting (results invalid).
 * Returns DIA_E_ARG when the shapes do not chain, no instantiation exists or 2 * wo->nstrips exceeds the CU
 * count (every workgroup must be resident) — callers then issue the two dia_gemm launches instead. */
int dia_mlp_fused(const dia_gemm_args* wi, const dia_gemm_args* wo, int32_t* barrier, void* stream);
/* same, bracketed by dispatch-level start/stop events; *ms_out = kernel duration (blocks until it finished) */
int dia_mlp_fused_timed(const dia_gemm_args* wi, const dia_gemm_args* wo, int32_t* barrier, void* stream, float* ms_out);


int dia_gemm_timed(const dia_gemm_args* a, void* stream, float* ms_out);

/* Single-query attention (decode) and the encoder's bidirectional attention on the same kernel.
 * Replaces RotaryEmbedding.forward (layers.py:135-173), KVCache.update (state.py:99-103),
 * repeat_interleave (layers.py:319-320) and F.scaled_dot_product_attention (layers.py:329-337). */
typedef struct {
  int32_t mode;             /* DIA_ATTN_* */
  int32_t kv_dtype;
  int32_t n_kv_heads;       /* grid.x */
  int32_t group;            /* q heads per kv head: 4 (self), 1 (cross, enc) */
  int32_t n_rows;           /* SELF: R = 2B rows; CROSS: B utterances; ENC: L tokens */
  int32_t kv_cap;           /* cache capacity per head (audio_length / text capacity) */
  const float* q;           /* fp32 [rows][ldq]; q head h at column q_off + h*128 */
  int32_t ldq;
  int32_t q_off;
  int32_t k_off;            /* SELF: new k of kv head h at k_off + h*128, new v at v_off + h*128 */
  int32_t v_off;
  void* kc;                 /* [kv rows][n_kv_heads][kv_cap][128] */
  void* vc;
  const int32_t* cur;       /* SELF/CROSS: per-utterance current step (device) */
  const int32_t* len;       /* CROSS: per-utterance text length (device); ENC: unused */
  int32_t enc_len;          /* ENC: L (== n_rows, <= kv_cap) */
  int32_t rope_rows;        /* rows of cos_t / sin_t, or 0 = not stated.  When stated, dia_attn refuses shapes whose RoPE
                             * position could leave the tables (SELF: kv_cap + 1 rows needed, ENC: enc_len) */
  const float* cos_t;
  const float* sin_t;
  void* P;                  /* output planes [3][mtiles][p_ktiles][64][8] */
  int64_t p_plane_stride;
  int32_t p_ktiles;
  int32_t _pad1;
  /* split-key partials: required when more than 128 keys are possible (kv_cap, or enc_len for ENC).
   * scratch: dia_attn_scratch_floats(pairs_rows, n_kv_heads, kv_cap) floats, pairs_rows = n_rows;
   * tickets: n_rows*n_kv_heads int32, zero before the first launch (the kernel re-zeroes them). */
  float* scratch;
  int32_t* tickets;
  /* int32 [n_kv_heads*group] or NULL: query head h is emitted at head position head_map[h] of the
   * (compacted) o_proj input; < 0 = head pruned, nothing emitted */
  const int32_t* head_map;
  /* 1: bf16 caches with V stored blocked as [key/32][128 dims][32 keys] (K stays [key][128]) -> the MFMA
   * attention kernel; 0: V stored [key][128] -> the VALU kernel (required for fp32 caches and ENC) */
  int32_t v_blocked;
  int32_t act_f32;          /* 1: P receives fp32 activation tiles (dia_gemm_args.act_f32) instead of three planes */
  int64_t kv_plane_stride;  /* DIA_KV_BF16X2: elements between the hi and the lo plane of kc / vc */
} dia_attn_args;
int dia_attn(const dia_attn_args* a, void* stream);
int dia_attn_scratch_floats(int n_rows, int n_kv_heads, int kv_cap);

/* Encoder self-attention of a PACKED prefill batch (Encoder.forward, layers.py:385-462, for every utterance
 * at once): utterance b owns packed rows [seg_off[b], seg_off[b] + seg_len[b]), seg_off a multiple of 32;
 * row_b[m] = utterance of packed row m or -1 for padding.  Launches the K/V plane preparation and the
 * MFMA attention; all arithmetic fp32-exact (three bf16 planes per operand).  kp / vp are scratch:
 * 3 * heads * rows * 128 bf16 each. */
typedef struct {
  const float* qkv;         /* [rows][ldq] fp32: q head h at q_off + h*128, k at k_off + h*128, v at v_off + h*128 */
  int32_t ldq, q_off, k_off, v_off;
  int32_t heads;
  int32_t rows;             /* packed rows, multiple of 32 */
  const int32_t* row_b;     /* [rows] */
  const int32_t* seg_off;   /* [B] */
  const int32_t* seg_len;   /* [B] */
  const float* cos_t;       /* [npos][64] */
  const float* sin_t;
  void* kp;                 /* scratch K planes */
  void* vp;                 /* scratch V planes (blocked) */
  void* P;                  /* output planes [3][rows/16][p_ktiles][64][8]; padding rows are not written */
  int64_t p_plane_stride;
  int32_t p_ktiles;
  int32_t _pad0;
} dia_enc_attn_args;
int dia_enc_attn(const dia_enc_attn_args* a, void* stream);

/* Decoder prefill over an audio prompt, batched (Decoder.forward in prefill mode, layers.py:722-766, with the
 * replay semantics of DESIGN.md §8: token row r -> cache slot r at RoPE position r + 1).  All prompt rows of all
 * utterances, both CFG rows, are packed: segment s = (utterance b, CFG row c) owns packed rows
 * [seg_off[s], seg_off[s] + seg_len[s]), seg_off a multiple of 32; row_seg[m] = segment of packed row m or -1;
 * seg_row[s] = 2b + c (the self-cache row; b = seg_row >> 1 indexes tokens, cross caches and text_len).
 * bf16 caches with blocked V only.  The dense layers between these three calls are dia_gemm over the packed rows. */
typedef struct {
  const int32_t* row_seg; const int32_t* seg_off; const int32_t* seg_len; const int32_t* seg_row;
  int32_t rows;             /* packed rows, multiple of 32 */
  int32_t _pad0;
  /* dia_dec_prefill_embed: x[m] = sum_c emb[c][tokens[b][r][c]], planes(x * g), strip ssq */
  const int32_t* tokens;    /* [B][T][C] */
  int32_t T, C, V, D;
  const float* emb;         /* [C][V][D] */
  const float* g;           /* first layer's pre-SA norm weight */
  float* x;                 /* [rows][D] */
  void* P; int64_t p_plane_stride; int32_t p_ktiles; int32_t ssq_ld;   /* embed: x planes; attn: output planes */
  float* ssq;               /* [D/16][ssq_ld] */
  /* dia_dec_prefill_kv (self K/V append of every packed row) and dia_dec_prefill_attn */
  const float* q;           /* fp32 [rows][ldq]: q head h at q_off + h*128 (kv: k at k_off, v at v_off) */
  int32_t ldq, q_off, k_off, v_off;
  int32_t q_heads, kv_heads, kv_cap;
  int32_t causal;           /* attn: 1 = self (keys 0..r of the segment's cache row), 0 = cross (text keys of the cond segment; uncond -> 0) */
  void* kc; void* vc;       /* bf16 caches [cache row][kv_heads][kv_cap][128], V blocked [.. kv_cap/32][128][32] */
  const float* cos_t; const float* sin_t;
  const int32_t* text_len;  /* cross: [B] */
} dia_dec_prefill_args;
int dia_dec_prefill_embed(const dia_dec_prefill_args* a, void* stream);
int dia_dec_prefill_kv(const dia_dec_prefill_args* a, void* stream);
int dia_dec_prefill_attn(const dia_dec_prefill_args* a, void* stream);

/* Encoder helper: RoPE(k) and v of all L tokens from the qkv rows into an fp32 [heads][cap][128]
 * scratch "cache" (layers.py:274-279,306-307 for the encoder). */
int dia_enc_kv_prep(const float* qkv, int ldq, int k_off, int v_off, int heads, int L, int cap,
                    const float* cos_t, const float* sin_t, float* kc, float* vc, void* stream);

/* x[m][:] = table[ids[m]][:] for the text encoder (layers.py:452), plus planes(x*g) and strip ssq.
 * cmap: as dia_gemm_args.cmap (first encoder layer's q/k/v input order of a compacted checkpoint) or NULL. */
int dia_embed_text(const int32_t* ids, int L, const float* table, int D, const float* g, float* x,
                   void* P, int64_t p_plane_stride, int p_ktiles, float* ssq, int ssq_ld, const int32_t* cmap, void* stream);

/* Decoder input embedding: x[2b..2b+1][:] = sum_c emb_c[tokens[b][cur[b]-1][c]] (layers.py:691-696). */
typedef struct {
  const int32_t* tokens;    /* [B][T][C] */
  const int32_t* cur;       /* [B] */
  int32_t B, T, C, V, D;
  int32_t act_f32;          /* 1: P receives fp32 activation tiles (dia_gemm_args.act_f32) instead of three planes */
  const float* emb;         /* [C][V][D] fp32 */
  const float* g;           /* first pre_sa_norm weight */
  float* x;                 /* [rows][D] */
  void* P;
  int64_t p_plane_stride;
  int32_t p_ktiles;
  int32_t ssq_ld;
  float* ssq;               /* [D/16][ssq_ld] */
  const int32_t* cmap;      /* as dia_gemm_args.cmap (first layer's q/k/v input order) or NULL */
} dia_embed_args;
int dia_embed_tokens(const dia_embed_args* a, void* stream);

/* CFG + constraints + temperature / top-k / top-p / multinomial + EOS/delay state machine + token
 * write + next-step embedding.  Replaces Dia._decoder_step (model.py:447-488), _sample_next_token
 * (model.py:32-82) and the loop body at model.py:771-807 (device-side, no host sync). */
typedef struct {
  const float* logits;      /* [rows][ld_logits], channel c at column c*V */
  int32_t ld_logits;
  int32_t B;
  int32_t T;                /* token buffer rows = audio_length */
  int32_t C;
  int32_t V;
  int32_t max_tokens;
  float cfg_scale, temperature, top_p;
  int32_t top_k;
  int32_t eos, pad, bos;
  int32_t max_delay;
  int32_t ignore_eos;       /* perf runs: natural EOS does not start the countdown */
  int32_t teacher;          /* parity runs: record samples in `pred`, leave `tokens` untouched */
  const int32_t* delay;     /* [C] device */
  const float* noise;       /* [B][noise_steps][C][V] Exp(1) variates; step index = cur - first_step[b] */
  int32_t noise_steps;
  int32_t _pad0;
  int32_t* tokens;          /* [B][T][C] */
  int32_t* pred;            /* [B][T][C] raw samples per step (row cur) */
  int32_t* cur;             /* [B] in/out */
  int32_t* fsm;             /* [B][4]: eos_detected, eos_countdown, bos_countdown, done */
  /* audio prompt (model.py:311-353, 406-422): first_step[b] = 1 + prompt frames = the first step that is
   * sampled.  Steps cur < first_step[b] REPLAY rows already in the token buffer: nothing is sampled or
   * written, the state machine does not move, only cur advances and the next row is embedded; the noise
   * row of step cur is cur - first_step[b].  NULL = no prompt anywhere (first_step 1). */
  const int32_t* first_step;
  dia_embed_args embed;     /* next-step embedding; embed.tokens/cur are taken from above */
} dia_sample_args;
int dia_sample(const dia_sample_args* a, void* stream);

/* Read-only pass over [ptr, ptr+nbytes) that pulls it into the 256 MiB Infinity Cache ahead of its
 * consumer (weights of the next kernels of the decode chain); writes nothing. */
int dia_prefetch(const void* ptr, int64_t nbytes, int nblocks, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Persistent MLP segment (csrc/seg.hip): cross o_proj -> wi_fused + SwiGLU -> wo (-> the NEXT layer's q/k/v projection)
 * of one decoder layer in ONE launch, for M <= 4 rows (batch 1-2) of a dense Dia-1.6B-shaped decoder.  Replaces the
 * launches co, wi, wo (and the following layer's qkv) of the step = reference DecoderLayer.forward layers.py:574-584
 * (+ layers.py:541, 273-275 of the next layer).  256 workgroups, one per CU, all resident together; the weights come in
 * the "ring layout" (dia_hip/layout.py seg_ring): per CU one contiguous run of 16 KiB slots in consumption order.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* a_in;        /* cross-attention output, fp32 activation tiles [ktile][64][8] of m-tile 0 (act_f32 format) */
  int32_t a_ktiles;         /* its K/32 (= 64) */
  int32_t M;                /* valid rows, 1..4 */
  const void* W;            /* ring arena of this layer: bf16 [256][nslots][16][64][8] */
  int32_t nslots;           /* dia_seg_slots(has_qkv): 29 with the next layer's q/k/v, 26 without */
  int32_t has_qkv;
  int32_t D, F;             /* 2048, 8192 (checked) */
  float* x;                 /* residual stream [rows][ldx], updated in place */
  int32_t ldx;
  const float* g_mlp;       /* pre_mlp_norm weight [D] */
  const float* g_next;      /* next consumer's norm weight [D]: the next layer's pre_sa_norm, or the final norm */
  float* qkv_out;           /* [rows][ldq] q|k|v of the next layer (has_qkv) */
  int32_t ldq;
  float* planes_x;          /* fp32 activation tiles of x * g_next, as DIA_EPI_RESID_EMIT leaves them (launched consumers read these) */
  int32_t xkt;
  float* ssq;               /* [D/16][ssq_ld] strip sums of squares of x */
  int32_t ssq_ld;
  float eps;
  void* ws;                 /* dia_seg_workspace_bytes() bytes; the first dia_seg_workspace_control_bytes() zeroed ONCE by the caller
                             * (and again after any failed launch) */
  int32_t timeout_us;       /* bound of every in-kernel wait; 0 = 20 ms */
  void* stamps;             /* debug: int64 [256][16] wall-clock (100 MHz) stamps of every workgroup's sync wave; NULL = off */
} dia_seg_args;
int dia_seg_mlp(const dia_seg_args* a, void* stream);
int64_t dia_seg_workspace_bytes(void);
int64_t dia_seg_workspace_control_bytes(void);
int32_t dia_seg_slots(int with_qkv);
/* 1 when the shapes (D, F, attention width = heads * 128, q/k/v width) have a segment kernel and the device has the CUs */
int dia_seg_supported(int D, int F, int attn_width, int nqkv);
/* after a stream sync: 0, or the code of the in-kernel wait that timed out (1 x1, 2 hidden, 3 wo partials, 4 x2) */
int dia_seg_error(const void* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Engine: owns nothing but the launch sequence.  All memory is allocated by the caller.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const void *w_qkv, *w_o, *w_cq, *w_co, *w_wi, *w_wo;   /* weight tiles */
  const void* w_wo_diag;                                 /* NULL, or wo once more in the diagonal layout (dia_gemm_args.w_layout = 1): used at <= 4 rows */
  const float *g_sa, *g_ca, *g_mlp;                      /* RMSNorm weights [D] */
  void *k_self, *v_self;                                 /* [R][kv_heads][T][128] */
  void *k_cross, *v_cross;                               /* [B][cq_heads][S][128] */
  int32_t kt_qkv, ns_qkv, kt_o, ns_o, kt_cq, ns_cq, kt_co, ns_co, kt_wi, ns_wi, kt_wo, ns_wo;
  /* compaction maps (device int32, NULL = dense): */
  const int32_t* cmap_ca;     /* emitted by self o_proj for the cross-q input order */
  const int32_t* cmap_mlp;    /* emitted by cross o_proj for the wi_fused input order */
  const int32_t* cmap_next;   /* emitted by wo for the next layer's q/k/v input order (or the logits head's) */
  const int32_t* smap_qkv;    /* strip map of the q/k/v output */
  const int32_t* smap_cq;     /* strip map of the cross-q output */
  const int32_t* hmap_self;   /* head map of self-attention */
  const int32_t* hmap_cross;  /* head map of cross-attention */
} dia_dec_layer;

typedef struct {
  int32_t n_layer, D, F, q_heads, kv_heads, cq_heads, C, V;
  int32_t B;                /* utterances; rows R = 2B */
  int32_t T;                /* audio_length (self cache capacity, token buffer rows) */
  int32_t S;                /* text capacity of the cross caches */
  int32_t kv_dtype;
  int32_t rows_pad;         /* 16 * ceil(R/16) */
  int32_t ld_logits;        /* 16 * ceil(C*V/16) */
  float eps;
  int32_t v_blocked;        /* 1: V caches in the blocked layout (bf16 only) -> MFMA attention */
  const dia_dec_layer* layers;   /* host array [n_layer] */
  const void* w_logits;
  int32_t kt_logits, ns_logits;
  const float* g_final;
  /* scratch (device) */
  float* x;                 /* [rows_pad][D] */
  void* planes_x;           /* [3][rows_pad/16][D/32][64][8] */
  void* planes_a;           /* attention output planes, width q_heads*128 */
  void* planes_h;           /* MLP hidden planes, width F */
  float* ssq;               /* [D/16][rows_pad] */
  float* qkv;               /* [rows_pad][(q_heads+2*kv_heads)*128] */
  float* qc;                /* [rows_pad][cq_heads*128] */
  float* logits;            /* [rows_pad][ld_logits] */
  const float* cos_t;       /* [T+1][64] */
  const float* sin_t;
  const int32_t* text_len;  /* [B] */
  float* sk_scratch;        /* split-K slabs: at least (D/16)*4*512 floats; with 17..128 rows (batch 9-64) also
                             * ceil(rows/16) * (D/16) * 4 * 256 (wo splits K four ways for every m-tile) */
  int32_t* sk_tickets;      /* max(max strips, 8 * D/16) int32, zeroed by the caller once */
  float* attn_scratch;      /* max over self/cross of dia_attn_scratch_floats(...) floats */
  int32_t* attn_tickets;    /* max(R*kv_heads, B*cq_heads) int32, zeroed by the caller once */
  int64_t sk_scratch_floats;/* capacity of sk_scratch in floats (0 = the minimum above) */
  int32_t* mlp_barrier;     /* 2 int32 zeroed by the caller once: dia_mlp_fused's barrier words (NULL = never fuse) */
  int32_t act_f32;          /* 1: planes_x / planes_a / planes_h carry fp32 activation tiles (dia_gemm_args.act_f32); needs
                             * sample.embed.act_f32 == 1 */
  int32_t w_planes;         /* 0 / 1, or 3: every weight pointer holds three bf16 planes of fp32 weights (dia_gemm_args.w_planes) */
  dia_sample_args sample;   /* sampler + FSM + embedding parameters */
  /* persistent MLP segments (NULL / 0 = the eight-launch layer): host array [n_layer] of ring arenas (layer l's arena holds
   * co, wi, wo of layer l and, for l + 1 < n_layer, qkv of layer l + 1), used when 2B <= 4 */
  const void* const* seg_w;
  void* seg_ws;             /* workspace of dia_seg_mlp */
  int64_t kv_plane_self, kv_plane_cross;   /* DIA_KV_BF16X2: plane strides (elements) of the self / cross caches */
} dia_engine_desc;

typedef struct dia_engine dia_engine;
int dia_engine_create(const dia_engine_desc* d, void* stream, dia_engine** out);
int dia_engine_destroy(dia_engine* e);
/* 1 when the engine's decode step runs the MLP as one fused launch (dia_mlp_fused: batch 1, shapes with an instantiation) */
int dia_engine_mlp_fused(const dia_engine* e);
/* enqueue `n_steps` decode steps; use_graph != 0 replays a captured hipGraph of one step */
int dia_engine_decode(dia_engine* e, int n_steps, int use_graph);
/* before the first graph decode: prefetch the weights of launch i+lookahead into the Infinity Cache
 * on a side branch of the step graph as soon as launch i has been issued (0 = off) */
int dia_engine_set_prefetch(dia_engine* e, int lookahead);
/* enqueue ONE decode step stopping after the logits GEMM (no sampling); for per-kernel timing */
int dia_engine_step_logits_only(dia_engine* e);
/* run ONE eager decode step with a HIP event recorded on the engine's stream after every launch and
 * return the elapsed milliseconds of each launch (launch order: per layer qkv, attn_self, o, cq,
 * attn_cross, co, wi, wo; then logits, sampler).  Synchronises the stream. */
int dia_engine_profile_step(dia_engine* e, float* ms_per_launch, int cap);
/* run ONE eager decode step with every kernel bracketed by dispatch-level start / stop events (timestamps of the
 * dispatch packet itself): ms_per_kernel[i] = begin -> end of the i-th launch (pure execution), interval_ms[i] (may be
 * NULL) = end of launch i-1 -> end of launch i = what the launch costs inside the dependent chain, boundary included —
 * the quantity rocprofv3 --kernel-trace reports as a kernel's duration in a replayed graph (begin[i] == end[i-1] there).
 * Same order as dia_engine_profile_step.  Returns the number of kernels launched or a negative DIA_E_*.  Synchronises. */
int dia_engine_time_step(dia_engine* e, float* ms_per_kernel, float* interval_ms, int cap);
/* kernel instantiation name (as rocprofv3 prints it, without the namespace) of the i-th launch of the calling thread's
 * last dia_engine_time_step / dia_gemm_timed; "" when out of range */
const char* dia_timed_kernel_name(int i);
/* number of kernel launches in one decode step */
int dia_engine_launches_per_step(const dia_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* DIA_HIP_H */
