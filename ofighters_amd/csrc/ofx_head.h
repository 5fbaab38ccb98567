// ofx_head.h - head-2 tail of the bi-head policy (upconv2 -> upconv3 -> upconv4 -> arg-max) as a row-streaming
// kernel pair (ofx_head.hip); called by policy_forward_impl (ofx_policy.hip).
#pragma once
#include "ofx_internal.h"

struct HeadParams2 {
  int S;                        // policy samples (ships)
  const float *up1;             // planar [S][2][50][50] uprelu1
  const float *w2mf, *b2;       // upconv2 phase weights [20][16] (PrepLayout::w2mf), folded bias [4]
  const float *w2raw;           // BN-folded upconv2 kernel [9][2][4]
  const float *w3mf, *b3;       // upconv3 phase weights [2][36][16] (PrepLayout::w3mf), folded bias [8]
  const float *w3raw;           // BN-folded upconv3 kernel [9][4][8]
  const float *w4eff_c, *b4;    // upconv4 phase weights [ci 8][phase 4][tap 9], bias [1]
  const float *w4raw;           // upconv4 kernel [9][8]
  const float *w2fr, *w3fr;     // frame-line variants of the phase weights (PrepLayout::w2fr, w3fr)
  const float *efr;             // frame phase weights of upconv4 [line h|v][side][parity 2][3][8] (PrepLayout::efr)
  float *u2fr;                  // [S][4][100][4] exact frame lines of uprelu2: row 0, row 99, col 0, col 99
  float *vfr;                   // [S][4][200][9] exact frame lines of the tap planes V_t = sum_c w4[t][c] uprelu3_c (top, bottom, left, right)
  float *c4;                    // [S][4][400]   zero-padding corrections of the heat-map frame pixels
  const uint8_t *mask;          // [S] or null
  int32_t *live;                // with a mask: scratch [1 + S], filled here with the count and the ordered list of the
                                // selected ships - workgroup i works on live[1 + i], so the live workgroups are the
                                // FIRST ones of the grid whatever the mask's pattern (spread over XCDs and CUs)
  int live_ready;               // the caller has filled `live` already (ofx_head_compact)
  unsigned long long *best;     // [S] packed (ordered value << 32) | ~index, zeroed by the caller
  float *heat;                  // [S][400][400] or null
  const int32_t *probe;         // [S][2] (x, y) or null
  float *ptr_probe;             // [S] heat-map value at the probe
  int frames_ref;               // frame lines through the from-the-definition kernel (OFX_OPT_FRAMES_REF)
  int legacy;                   // TF1 legacy bilinear instead of half-pixel centres (OFX_OPT_BILINEAR_LEGACY)
  int bf16;                     // OFX_OPT_POLICY_BF16: 1 bf16 / 2 fp16 operands in upconv3 / upconv4 (opt-in, not the default)
  int event_base;               // >= 0: ofx_event_record(event_base / event_base + 1) around k_head_stream
  int ablate;                   // diagnostics (OFX_HEAD_HOOKS builds only)
  unsigned long long *dbg;      // diagnostics: [blocks][8 waves][6] s_memtime sums
};

// bytes of the three frame buffers for S samples
size_t ofx_head_frame_bytes(size_t S, size_t *u2fr, size_t *vfr, size_t *c4);
// mask [S] -> live[0] = count, live[1 + i] = i-th selected ship (one workgroup, a scan)
int ofx_head_compact(ofx_handle *h, int S, const uint8_t *mask, int32_t *live);
// k_head_frames + k_head_stream on the handle's stream
int ofx_launch_head(ofx_handle *h, const HeadParams2 &p);
