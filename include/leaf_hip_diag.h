/* Diagnostic exports of the libraries under tools/diag/ (make -C leaf_amd/csrc variants | stamps | ...; -DLEAF_VARIANTS).
 * NOT part of libleaf_hip.so and not part of the integration surface (include/leaf_hip.h): test hooks and tuning
 * knobs for tests/test_gpu_variants.py, the tile-plan test of tests/test_host_cpu.py and the measuring tools under tools/. */
#ifndef LEAF_HIP_DIAG_H
#define LEAF_HIP_DIAG_H
#include "leaf_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* host-side cut of a pass's sequences into the M tiles of the fused QKV + attention launch (leaf_amd/csrc/qkv_attn.hip): whole
 * sequences, <= tile_rows rows, prefixed sequences of <= ncap consecutive captions; out [2 (n + 1)] receives (first sequence, first row) of every
 * tile and (n, rows) behind the last, returns the number of tiles (test hook, tests/test_host_cpu.py) */
int leaf_debug_qkv_attn_plan(const int32_t* lens, int ctx, int s0, int n, int prefixed, int group, int group_off, int tile_rows /* 0: the
                             kernel form in use */, int ncap /* 0: what sequences of ctx positions allow */, int32_t* out);
int leaf_debug_gemm_stamps(void* buf);
/* dispatch tuning (tools/small_gemm_sweep.py): fewest 256 x 256 tiles for which the half-stage ring kernel takes a launch */
int leaf_debug_gemm_min_tiles(int n);

#ifdef __cplusplus
}
#endif
#endif
