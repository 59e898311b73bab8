// Process-wide tuning / debug overrides of the launch heuristics.  Nothing on the launch path reads the
// environment: values are set through dia_set_tuning() (C ABI) or, once at library initialisation, from the
// DIA_TUNE environment variable ("name=value,name=value") so that bench and profiling scripts can vary a knob
// without code changes.  -1 = not set (the heuristic decides).
#pragma once

enum dia_tune_id {
  DIA_TUNE_ATTN_NZ = 0,        // attn_nz: key-split factor of dia_attn (workgroups per (row, head) pair)
  DIA_TUNE_ATTN_GPW,           // attn_gpw: 32-key granules per wave before the keys are split (self)
  DIA_TUNE_ATTN_GPW_CROSS,     // attn_gpw_cross: same for cross-attention
  DIA_TUNE_GEMM_SPW,           // gemm_spw: strips per workgroup of the persistent GEMV / 16-row forms
  DIA_TUNE_GEMM_MZ_MAX,        // gemm_mz_max: highest m-tile count served by k_gemm16 over gridDim.z (0 = off)
  DIA_TUNE_TILE_MIN_BLOCKS,    // tile_min_blocks: fewest 64x256 blocks for which prefill uses the MFMA-tiled kernel
  DIA_TUNE_WO_SK,              // wo_sk: cross-workgroup split-K factor of wo in the decode step (1..4)
  DIA_TUNE_WO_PAIR,            // wo_pair: 0 = no split-K 4 per m-tile at 17..128 rows
  DIA_TUNE_WO_NW,              // wo_nw / wo_spw: waves and strips per workgroup of wo
  DIA_TUNE_WO_SPW,
  DIA_TUNE_ACT_F32,            // act_f32: 0 = the decode step keeps three bf16 activation planes between its kernels; 1 / unset =
                               // fp32 activation tiles (read by the host side when it builds a session)
  DIA_TUNE_WO_DIAG,            // wo_diag: 1 = <= 2 rows run wo from the diagonal layout (k_gemv_diag; experiment, measured slower end to end) — read by
                               // the engine and by DeviceWeights, which builds the second copy of wo only then
  DIA_TUNE_GEMM_2T,            // gemm_2t: 0 = 17..128 rows never take the two-m-tile kernel k_gemm2t (K = 2048 dense shapes)
  DIA_TUNE_GEMM_ZR,            // gemm_zr: 0 = 17..128 rows keep the one-strip-ahead z-form of k_gemm16 instead of the ring form k_gemm16_zr
  DIA_TUNE_SEG,                // seg: 1 = batch 1-2 sessions run persistent MLP segments (dia_seg_mlp) when the model carries ring
                               // arenas (DeviceWeights(seg="on")); unset / 0 = the eight-launch layer (read by the host side)
  DIA_TUNE_SEG_NB,             // seg_nb: 16 KiB weight slots each streaming wave of dia_seg_mlp keeps in flight (1..3)
  DIA_TUNE_SEG_DBG,            // seg_dbg: debug bits of dia_seg_mlp (1 = no weight loads: hand-off timing only, wrong results)
  DIA_TUNE_SEG_SLEEP,          // seg_sleep: s_sleep units before the first loads of the waves that stream the later ops
  DIA_TUNE_G2T_WGS,            // g2t_wgs: workgroup budget of a k_gemm2t launch over planes (the short-prompt prefill); unset = see dia_gemm
  DIA_TUNE_CKV_MERGE,          // ckv_merge: 0 = the prefill projects cross K/V with one launch per decoder layer (read by dia_hip/engine.py; A/B)
  // ---- EXPERIMENTS=1 builds only
  DIA_TUNE_MLP_FUSE,           // mlp_fuse: 1 = wi + wo as one persistent launch at batch 1 (dia_mlp_fused)
  DIA_TUNE_TILE_V,             // tile_v: prefill tile kernel variant (0..5; 3 = wave-specialised default)
  DIA_TUNE_BLK32_KR,           // blk32_kr / blk32_ws: k_gemm_blk32 form for 17..32 rows
  DIA_TUNE_BLK32_WS,
  DIA_TUNE_NO_G32,             // no_g32, g32_all, g32m: k_gemm32 / k_gemm32m selection for 17..32 rows
  DIA_TUNE_G32_ALL,
  DIA_TUNE_G32M,
  DIA_TUNE_COUNT
};

int dia_tune(int id);                 // current value, -1 when unset
void dia_tuning_init_from_env();      // reads DIA_TUNE once (called by dia_kernels_init_once)
