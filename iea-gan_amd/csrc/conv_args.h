// Kernel argument blocks of the conv launchers: the public C structs, passed by value to the kernels.
#pragma once
#include "common.h"
#include "../../include/ieagan_hip.h"

typedef ieagan_src_desc SrcDesc;
typedef ieagan_conv_desc ConvArgs;
typedef ieagan_wgrad_desc WgradArgs;

// internal bit of ConvArgs.flags (set by the launcher on its own copy of the descriptor, never by a caller): conv_gather splits the
// K loop over the four waves of a block (tiny feature maps)
#define CONV_INTERNAL_SPLITK (1 << 30)

int conv_gather_launch(const ConvArgs& a, hipStream_t st);
int conv_wgrad_launch(const WgradArgs& a, hipStream_t st, int use_tr);
int conv3x3_lds_launch(const ConvArgs& a, hipStream_t st);       // 1 = launched, 0 = not applicable (caller falls back to conv3x3_halo), < 0 = error
int conv3x3_ws_launch(const ConvArgs& a, hipStream_t st);         // 1 = launched, 0 = not applicable (caller falls back to conv3x3_halo)
int conv1x1_stream_launch(const ConvArgs& a, hipStream_t st);     // 1 = launched, 0 = not applicable (caller falls back to conv_gather)
int conv1x1_tile_launch(const ConvArgs& a, hipStream_t st);       // 1 = launched, 0 = not applicable (caller falls back to conv_gather)

// Planning calls (ieagan_conv_stats_slots): the dispatch runs exactly as in a real call, but every launch site records its blocks per
// statistics group here and returns instead of launching.  Thread-local (api.hip).
struct ConvPlanCtx { bool on; int slots; };
ConvPlanCtx& conv_plan_ctx();
#define CONV_PLAN_POINT(BPE, RET)                      \
    if (conv_plan_ctx().on) {                          \
        conv_plan_ctx().slots = (BPE);                 \
        return RET;                                    \
    }
