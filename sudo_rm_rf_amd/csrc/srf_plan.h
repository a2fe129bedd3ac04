// The plan object behind the opaque srf_plan* of include/sudormrf_hip.h (shared by srf_api.hip and srf_train.hip).
#pragma once
#include <vector>
#include "srf_common.h"

struct srf_plan {
  srf_config cfg;
  int Bt, T, Tp, L, A, SA;
  int Bg, nB, nC;  // folded batch (Bt*G), channels outside / inside the U-block per group
  int n_params, n_launches;
  // parameter indices
  int p_block0, p_block_stride, p_ublock_off, p_tail;
  // workspace offsets (bytes)
  size_t off_stats, stats_bytes, off_enc, off_xa, off_xb, off_xq, off_xu, off_y1, off_lv[SRF_MAX_DEPTH],
      off_masked, off_dec, off_pyr, off_wdpack, total_bytes;
  int fused_pyramid;
  int slots_per_block, n_slots;
  // pre-packed (split-bf16) weights of the 1x1 convolutions: param index -> workspace offset (0 = none)
  std::vector<int> pk_param, pk_cout, pk_cin;
  std::vector<size_t> pk_off;
  std::vector<size_t> pk_of_param;  // [n_params] offset or 0
};
