// ofx_policy.hip - bi-head policy forward (agents/qlearnIA_V2.py:123-190,206-220)
#include "ofx_internal.h"

extern "C" int ofx_policy_layout(const ofx_handle *h, ofx_policy_desc *desc) {
  (void)h; (void)desc;
  ofx_set_error("ofx_policy_layout: not built yet");
  return OFX_ERR_INVALID;
}
extern "C" int ofx_policy_forward(ofx_handle *h, const float *weights, const uint8_t *ship_mask, float *act_values,
                                  int32_t *iaction, int32_t *ipointer, float *heatmap) {
  (void)h; (void)weights; (void)ship_mask; (void)act_values; (void)iaction; (void)ipointer; (void)heatmap;
  ofx_set_error("ofx_policy_forward: not built yet");
  return OFX_ERR_INVALID;
}
extern "C" int ofx_policy_actions(ofx_handle *h, const int32_t *iaction, const int32_t *ipointer,
                                  const uint8_t *ship_mask, ofx_action *actions) {
  (void)h; (void)iaction; (void)ipointer; (void)ship_mask; (void)actions;
  ofx_set_error("ofx_policy_actions: not built yet");
  return OFX_ERR_INVALID;
}
