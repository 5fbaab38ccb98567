// ofx_diag.h - the DIAGNOSTIC build switches of libofx, in one place.  All default to 0: the product build carries none
// of their code.  A diagnostic build is made by a tools/ script (make HEAD_EXTRA=... / FIT_EXTRA=...), runs a timing
// experiment on the GPU box and is replaced by the default build again; its RESULTS ARE WRONG by construction where noted.
#pragma once

// k_head_stream (ofx_head.hip): s_memtime stamps per wave and section (OFX_HEAD_STAMPS=1 in the environment prints them
// for the 20th forward) and the role ablations OFX_HEAD_ABLATE = 1 (no stencil pass) / 2 (no stage B): tools/head_stamps.sh.
// Ablated runs compute wrong heat maps.
#ifndef OFX_HEAD_HOOKS
#define OFX_HEAD_HOOKS 0
#endif

// f_conv_fwd / f_bw (ofx_fit.hip): 1 = tiles are not filled, 2 = tiles are not computed (tools/fit_ablate.sh): wrong results.
#ifndef OFX_FIT_ABLATE
#define OFX_FIT_ABLATE 0
#endif

// k_trunk12<., true> (ofx_policy.hip): ofx_policy_trunk_stats returns s_memtime sums of wave 0 of every workgroup instead of
// the counts, summed over the workgroups: cycles in phase A (incl. its barrier) / in the rest of the step / of that in the
// classification + constant stores / in the run tiles.
#ifndef OFX_TRUNK_STAMPS
#define OFX_TRUNK_STAMPS 0
#endif
