/* pt_api_tuning.h — tuning surface of the traversal kernel (k_extend), split from the product API (include/pt_api.h).
 *
 * NO counterpart in the reference: TraceRay's scheduling is the DXR driver's business (PathTracing.hlsl:71). These knobs
 * exist for the measurements behind DESIGN.md (tools/tune_extend.py, tools/exp_tri_util.py, the schedule-independence
 * test); none of them changes a hit record. A host that only renders never includes this file. Values may be renumbered
 * or dropped between versions; the defaults are the shipped schedule (pathtracer_amd/csrc/kernels.h LaunchCfg). */
#ifndef PT_API_TUNING_H
#define PT_API_TUNING_H

#include "pt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  PT_TUNE_EXTEND_FETCH_MIN = 5,     /* refill idle lanes of a wave once this many (1..64) are idle (16) */
  PT_TUNE_EXTEND_POSTPONE_PCT = 6,  /* run a triangle phase once >= this % of the busy lanes have parked triangle work (40; 0 = at once) */
  PT_TUNE_EXTEND_BLOCKS_PER_CU = 7, /* persistent grid = CUs x this (0 = occupancy query) */
  PT_TUNE_EXTEND_LDS_STACK = 8,     /* per-lane traversal-stack entries kept in LDS, the rest spills to HBM (6) */
  PT_TUNE_EXTEND_DEFER = 9,         /* 0 = postpone triangle groups on the stack; N > 0 = park up to N groups per lane (4) */
  PT_TUNE_EXTEND_TOP_NODES = 12,    /* BVH nodes (breadth-first order: the top of the tree) every traversal block keeps in LDS (73) */
  PT_TUNE_EXTEND_CONTINUE_PCT = 13, /* a triangle phase goes on while >= this % of the busy lanes have work (25) */
  /* the same four for launches that run the shared triangle phase (PT_OPT_SHARED_TRIANGLE_PHASE), which has its own schedule:
     its 2 KB of LDS per wave come out of the parked-group queue and the top-node cache */
  PT_TUNE_SHARED_DEFER = 18,        /* parked groups per lane, 1..16 (3) */
  PT_TUNE_SHARED_TOP_NODES = 19,    /* (8) */
  PT_TUNE_SHARED_POSTPONE_PCT = 20, /* (25) */
  PT_TUNE_SHARED_CONTINUE_PCT = 21  /* a pass of fewer records than this % of the busy lanes is left for a fuller one (50) */
};

/* Sets one knob on this ctx (PT_ERR_INVALID_ARG outside its range). pt_set_option rejects these ids. */
int pt_debug_set_tuning(pt_ctx* ctx, int knob, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_TUNING_H */
